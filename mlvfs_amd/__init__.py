"""mlvfs_amd -- MI355X-native implementation of MLVFS's per-frame raw-processing path.

Only what the hot path needs lives here:
  csrc/        hand-written HIP kernels (gfx950) + the C-ABI shim -> libmlvfs_amd.so
  lib.py       ctypes binding of include/mlvfs_amd.h (fails loudly if the .so is missing)
  abi.py       ctypes mirror of `struct frame_headers` (include/mlvfs_abi.h)
  pipeline.py  host-side mirror of process_frame's stage order (mlvfs/main.c:908-1005)
  stream.py    device-resident frame streams for throughput runs (bench.py)
  synth.py     seeded synthetic MLV payload generator
"""
__version__ = "0.1.0"
