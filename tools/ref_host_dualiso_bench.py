"""The reference's own process_frame text (oracle/_ref/ref_host_amd_wrap: main.c's functions sliced at build time, linked against
libmlvfs_amd.so) converting a dual-ISO clip of 3584x1320 frames from T host threads: `--dual-iso 2 --amaze-edge`, full-res, alias map.
One conversion per call by construction (cr2hdr20_convert_data), host memory in and out.  usage: ref_host_dualiso_bench.py [threads ...]"""
import os, subprocess, sys, tempfile, pathlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlvfs_amd import mlvfile, synth
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
host = os.path.join(root, "oracle", "_ref", "ref_host_amd_wrap")
w, h, n = 3584, 1320, 8
d = pathlib.Path(tempfile.mkdtemp(prefix="mlvfs_amd_di_")) / "card"
d.mkdir(parents=True)
pl = [np.ascontiguousarray(synth.pack_bits(synth.dual_iso_frame(w, h, frame=k)), "<u2").tobytes() for k in range(n)]
mlvfile.write_clip(str(d / "M07-1234.MLV"), pl, w, h, chunks=2, frame_space=32, shuffle=True, video_class=1)
vp = ["/M07-1234.MLV/M07-1234_%06d.dng" % k for k in range(n)]
for T in [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16]:
    r = subprocess.run([host, str(d), "-", "dual_iso=2", "hdr_interp=0", "threads=%d" % T, "loops=%d" % max(2, int(os.environ.get("FRAMES", "512")) // (8 * T)), "--", *vp],
                       capture_output=True, text=True, timeout=900)
    line = [l for l in r.stderr.splitlines() if '"fps"' in l]
    print(T, line[-1] if line else r.stderr[-500:], flush=True)
