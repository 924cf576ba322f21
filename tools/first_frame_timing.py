import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from mlvfs_amd import lib, synth, dist as mdist
from mlvfs_amd.stream import ClipStream
W,H=3584,1320
s=ClipStream(W,H)
packed=s.synth_packed(2,seed=1)
def T(name,f):
    torch.cuda.synchronize(); t0=time.perf_counter(); r=f(); torch.cuda.synchronize(); print(f"{name:28s} {(time.perf_counter()-t0)*1e3:7.2f} ms"); return r
for rep in range(2):
    print("rep",rep)
    frame0=T("unpack", lambda: s.unpack(packed[:1]))
    pix=T("detect_bad_pixels", lambda: s.detect_bad_pixels(frame0[0],0))
    T("fix_pixels", lambda: s.fix_pixels(frame0))
    frame0=T("chroma_smooth", lambda: s.chroma_smooth(frame0,5))
    count_rows,hist_rows=mdist.gpu_callbacks(s,frame0[0])
    n=T("count_rows", lambda: count_rows(0,H))
    rnd=T("glibc_rand_slice", lambda: mdist.glibc_rand_slice(0,n))
    T("rand upload", lambda: torch.from_numpy(rnd.view(np.int16)).to("cuda"))
    hn=T("hist_rows (all)", lambda: hist_rows(0,H,0,n))
    T("solve", lambda: mdist.solve_coefficients(hn[0],hn[1],s.frame_size))
# inside hist_rows: the library call alone
import ctypes as C
n = count_rows(0, H)
rnd = torch.from_numpy(mdist.glibc_rand_slice(0, n).view(np.int16)).to("cuda")
hist = torch.zeros(8 * 65536, dtype=torch.int32, device="cuda"); num = torch.zeros(8, dtype=torch.int32, device="cuda")
L = s.L; geom = s.geom
for rep in range(3):
    acc = C.c_int64(0)
    T("  stripes_count_dev", lambda: L.mlvfs_amd_stripes_count_dev(C.byref(geom), C.c_void_p(frame0[0].data_ptr()), 0, H, C.byref(acc), None))
    T("  stripes_hist_dev", lambda: L.mlvfs_amd_stripes_hist_dev(C.byref(geom), C.c_void_p(frame0[0].data_ptr()), 0, H, C.c_void_p(rnd.data_ptr()), 2 * n, C.c_void_p(hist.data_ptr()), C.c_void_p(num.data_ptr()), None))
