"""Every drop-in symbol on its own, the way a caller outside any bracket meets it: one 3584x1320 frame in pageable host memory, FIRST
call of a process / clip and steady state, milliseconds -- and the reference's own code (oracle/_ref) beside it where it is there.
Written after stripes_compute_correction turned out to spend 300 of its 305 ms in 9.4 M calls of rand() (round 4): the per-frame
benches (tools/dropin_bench.py, bench.py extra.pcie) only see a clip's steady state.
usage: python tools/dropin_symbols_bench.py"""
import ctypes as C, json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mlvfs_amd import abi, lib, synth

W, H = 3584, 1320
L = lib.load(); assert L.mlvfs_amd_init(0) == 0
devnull = os.open(os.devnull, os.O_WRONLY); saved = os.dup(1)


def quiet(on):
    sys.stdout.flush()
    os.dup2(devnull if on else saved, 1)


def ms(fn, reps=1):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t.append((time.perf_counter() - t0) * 1e3)
    return t


frame = synth.normal_frame(W, H, seed=1)
packed = np.concatenate([synth.pack14(frame).astype("<u2"), np.zeros(4, "<u2")])
di = synth.dual_iso_frame(W, H, seed=3)
rows = {}


def fh_new(guid):
    fh = abi.make_frame_headers(W, H, bpp=14, black=synth.BLACK, white=synth.WHITE)
    fh.file_hdr.fileGuid = guid
    return fh


def run(name, make, call, reps=4):
    """make() -> fresh arguments (not timed), call(args) timed: the first call and the median of the later ones."""
    quiet(True)
    try:
        t = []
        for k in range(reps):
            a = make(k)
            t += ms(lambda: call(a))
    finally:
        quiet(False)
    rows[name] = {"first_ms": round(t[0], 3), "later_ms": round(float(np.median(t[1:])), 3)}


out = np.empty(W * H, np.uint16)
run("dng_get_image_data", lambda k: fh_new(1), lambda fh: L.dng_get_image_data(C.byref(fh), lib.ptr(packed), lib.ptr(out), 0, out.nbytes))
for m in (2, 3, 5):
    run(f"chroma_smooth {m}x{m}", lambda k: (fh_new(1), frame.copy()), lambda a, m=m: L.chroma_smooth(C.byref(a[0]), lib.ptr(a[1]), m))
# a new clip (guid) every time: detection + map; then the same clip again: the cached map
run("fix_bad_pixels, new clip", lambda k: (fh_new(100 + k), frame.copy()), lambda a: L.fix_bad_pixels(C.byref(a[0]), lib.ptr(a[1]), 0, 0))
run("fix_bad_pixels, same clip", lambda k: (fh_new(100), frame.copy()), lambda a: L.fix_bad_pixels(C.byref(a[0]), lib.ptr(a[1]), 0, 0))
libc = C.CDLL(None)
def mk_corr(k):
    libc.srand(1)
    return (fh_new(1), L.stripes_new_correction(f"symbols_bench_{k}.MLV".encode()), frame.copy())
run("stripes_compute_correction", mk_corr, lambda a: L.stripes_compute_correction(C.byref(a[0]), a[1], lib.ptr(a[2]), 0, a[2].size))
corr = L.stripes_get_correction(b"symbols_bench_0.MLV")
run("stripes_apply_correction", lambda k: (fh_new(1), frame.copy()), lambda a: L.stripes_apply_correction(C.byref(a[0]), corr, lib.ptr(a[1]), 0, a[1].size))
run("fix_pattern_noise", lambda k: frame.copy(), lambda a: L.fix_pattern_noise(lib.ptr(a), W, H, synth.WHITE, 0))
run("hdr_convert_data (preview)", lambda k: (fh_new(1), di.copy()), lambda a: L.hdr_convert_data(C.byref(a[0]), lib.ptr(a[1]), 0, a[1].nbytes))
run("cr2hdr20_convert_data (amaze-edge)", lambda k: (fh_new(1), di.copy()), lambda a: L.cr2hdr20_convert_data(C.byref(a[0]), lib.ptr(a[1]), 0, 1, 1, 0, 0))
hdr = np.zeros(65536, np.uint8)
run("dng_get_header_data", lambda k: fh_new(1), lambda fh: L.dng_get_header_data(C.byref(fh), lib.ptr(hdr), 0, hdr.size, 0.0, b"clip.MLV"))
def deflicker(a):
    h = L.hist_create(synth.WHITE)
    L.hist_add(h, lib.ptr(a), a.size, 1)
    L.hist_median(h)
    L.hist_destroy(h)
run("hist_create + hist_add + hist_median (deflicker)", lambda k: frame.copy(), deflicker)
print(json.dumps(rows, indent=1))

# ---- the reference's own code beside it (one call each; it has no first-call cost worth the name except its dual-ISO tables)
try:
    from oracle import bindings
    if not bindings.have_ref():
        raise RuntimeError("oracle/_ref not built")
    R = bindings.Reference()
    ref = {}
    def rt(name, fn):
        quiet(True)
        try:
            t = ms(fn)[0]
        finally:
            quiet(False)
        ref[name] = round(t, 1)
    img = frame.reshape(H, W)
    rt("dng_get_image_data", lambda: R.unpack(packed, W, H))
    for m in (2, 3, 5):
        rt(f"chroma_smooth {m}x{m}", lambda m=m: R.chroma_smooth(img, synth.BLACK, m))
    rt("fix_bad_pixels, new clip", lambda: R.fix_bad_pixels(img, synth.BLACK))
    rt("stripes_compute_correction", lambda: R.stripes_compute(img, synth.BLACK, synth.WHITE))
    rt("stripes_apply_correction", lambda: R.stripes_apply(img, synth.BLACK, synth.WHITE, 1, [65536, 65536, 64893, 66202, 64574, 66541, 65219, 65873]))
    rt("hdr_convert_data (preview)", lambda: R.hdr_preview(di.reshape(H, W), synth.BLACK, synth.WHITE))
    if os.environ.get("SYMBOLS_BENCH_SLOW") == "1":          # seconds each
        rt("fix_pattern_noise", lambda: R.fix_pattern_noise(img, synth.WHITE))
        rt("cr2hdr20_convert_data (amaze-edge)", lambda: R.cr2hdr20(di.reshape(H, W), synth.BLACK, synth.WHITE))
    rt("hist_create + hist_add + hist_median (deflicker)", lambda: R.hist_median(frame, 1, synth.WHITE))
    print("reference, one host core, ms:", json.dumps(ref))
except Exception as e:  # noqa: BLE001
    print("reference not timed:", e)
