"""The definitive boundary test: the REFERENCE's own process_frame / deflicker / create_preview text
(mlvfs/main.c:895-1033, with mlv_get_frame_headers, get_image_data, the path helpers and resource_manager.c's
mlvfs_load_chunks / mlvfs_close_chunks) linked three ways by oracle/Makefile --

    oracle/_ref/ref_host_ref        with the reference's dng.o cs.o stripes.o hdr.o amaze_demosaic_RT.o histogram.o patternnoise.o
    oracle/_ref/ref_host_amd        with libmlvfs_amd.so in their place
    oracle/_ref/ref_host_amd_wrap   the same plus integration/mlvfs_amd_wrap.c and the two --wrap flags (the frame bracket)
    oracle/_ref/ref_host_amd_wrap_alloc   ... plus integration/mlvfs_amd_wrap_alloc.c and --wrap of malloc / calloc / realloc / free (page-locked
                                    frame buffers from the library's pool: the fused kernel writes image_buffer->data itself)

-- and run as three processes over the same synthetic .MLV clips.  What process_frame leaves in struct image_buffer
(->data and ->header, main.c:929-998) must be byte-equal between them for every option set.  The programs hold the reference's
text only as compiled code under the git-ignored oracle/_ref/ (they travel to the GPU box like libmlvfs_ref.so); the
command-line driver around them is oracle/ref_host_driver.inc."""
import os
import struct
import subprocess

import numpy as np
import pytest

from mlvfs_amd import mlvfile, synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFDIR = os.path.join(ROOT, "oracle", "_ref")
HOSTS = {k: os.path.join(REFDIR, "ref_host_" + k) for k in ("ref", "amd", "amd_wrap", "amd_wrap_alloc")}
W, H = 416, 264


def need_hosts():
    if not all(os.path.exists(p) for p in HOSTS.values()):
        if os.path.isdir("/root/reference/mlvfs"):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
        else:
            pytest.skip("oracle/_ref/ref_host_* not present (needs /root/reference to build)")


def run_host(kind, mlv_dir, prefix, opts, vpaths, timeout=900):
    cmd = [HOSTS[kind], str(mlv_dir), str(prefix), *[f"{k}={v}" for k, v in opts.items()], "--", *vpaths]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (kind, r.stderr[-3000:])
    out = []
    for i in range(len(vpaths)):
        out.append((open(f"{prefix}.{i}.data", "rb").read(), open(f"{prefix}.{i}.hdr", "rb").read()))
    return out, r


def make_clip(tmp_path, kind, n=5, w=W, h=H, reference=None):
    d = tmp_path / "card"
    d.mkdir()
    frames = [synth.dual_iso_frame(w, h, frame=k) if kind == "dual_iso" else synth.normal_frame(w, h, seed=9, frame=k, hot=60, cold=60) for k in range(n)]
    vc = 1
    if kind == "lj92":
        from oracle import lj92_testenc as enc
        from test_lj92 import quadrants
        pl, vc = [struct.pack("<I", w * h * 2) + enc.encode(quadrants(f), 6, 14) for f in frames], 1 | 0x100
    elif kind == "lzma":
        pl, vc = [reference.lzma_payload(synth.pack_bits(f).tobytes()) for f in frames], 1 | 0x80
    else:
        pl = [np.ascontiguousarray(synth.pack_bits(f), "<u2").tobytes() for f in frames]
    mlvfile.write_clip(str(d / "M07-1234.MLV"), pl, w, h, chunks=2, frame_space=32, shuffle=True, video_class=vc)
    return d, frames


def vpath(k):
    return "/M07-1234.MLV/M07-1234_%06d.dng" % k


def test_reference_text_host_equals_the_oracle_on_the_cpu(oracle, tmp_path):
    """The sliced text really is process_frame: through the reference's own objects it gives what the checker's pipeline gives."""
    need_hosts()
    d, frames = make_clip(tmp_path, "plain", n=3)
    got, _ = run_host("ref", d, tmp_path / "o", dict(cs=5, badpix=1, stripes=1), [vpath(0), vpath(2)])
    pixels = oracle.detect_bad_pixels(frames[0], synth.BLACK, 0)          # the clip's map comes from the first frame served (cs.c:233-312)
    corr = None
    for (data, hdr), k in zip(got, (0, 2)):
        img = oracle.chroma_smooth(oracle.apply_bad_pixels(frames[k], synth.BLACK, pixels), synth.BLACK, 5)
        if corr is None:
            corr = oracle.stripes_compute(img, synth.BLACK, synth.WHITE, frame_size=W * H * 14 // 8)
        want = oracle.stripes_apply(img, synth.BLACK, synth.WHITE, *corr)
        assert len(hdr) == 65536 and np.array_equal(np.frombuffer(data, "<u2").reshape(H, W), want), k


OPTION_SETS = [
    ("plain", dict(cs=5, badpix=1, stripes=1)),
    ("plain", dict(cs=2)),
    ("plain", dict(cs=3, badpix=2, stripes=1, deflicker=3000)),
    ("plain", dict(pnoise=1, cs=5)),
    ("plain", dict(deflicker=2500, fps1000=23976)),
    ("dual_iso", dict(dual_iso=1, badpix=1)),
    ("dual_iso", dict(dual_iso=2, cs=0, badpix=0)),
    ("dual_iso", dict(dual_iso=2, hdr_interp=1, cs=5, badpix=1, stripes=1)),
    ("plain", dict(dual_iso=2, cs=2, badpix=1)),                  # not a dual-ISO clip: cr2hdr20 returns 0, the normal path runs
    ("lj92", dict(cs=5, badpix=1, stripes=1)),
    ("lzma", dict(cs=5, badpix=1, stripes=1, deflicker=3000)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,opts", OPTION_SETS, ids=[k + ":" + ",".join(f"{a}={b}" for a, b in o.items()) for k, o in OPTION_SETS])
def test_reference_process_frame_text_linked_against_the_hip_library(gpu, reference, tmp_path, kind, opts):
    need_hosts()
    d, _ = make_clip(tmp_path, kind, reference=reference)
    order = [vpath(2), vpath(0), vpath(4), vpath(1), "/M07-1234.MLV/_PREVIEW.gif", vpath(3)]   # first frame served is NOT frame 0
    want, r0 = run_host("ref", d, tmp_path / "ref", opts, order)
    for host in ("amd", "amd_wrap", "amd_wrap_alloc"):
        for f in os.listdir(d):                                     # every host builds its own .IDX, like a fresh mount
            if f.endswith(".IDX"):
                os.remove(d / f)
        got, r1 = run_host(host, d, tmp_path / host, opts, order)
        for i, ((wd, wh), (gd, gh)) in enumerate(zip(want, got)):
            assert len(gd) == len(wd) and len(gh) == len(wh), (host, order[i])
            if gd != wd:
                a, b = np.frombuffer(gd, np.uint8), np.frombuffer(wd, np.uint8)
                raise AssertionError(f"{host} {order[i]}: image_buffer->data differs in {(a != b).sum()} bytes")
            assert gh == wh, f"{host} {order[i]}: image_buffer->header differs in {(np.frombuffer(gh, np.uint8) != np.frombuffer(wh, np.uint8)).sum()} bytes"


@pytest.mark.gpu
def test_reference_text_hosts_full_size_and_throughput(gpu, tmp_path):
    """3584x1320 (BASELINE configs[2]): the wrapped link against the plain link against the reference objects on two frames,
    then the wrapped host's bench mode from 8 threads (a smoke run of what bench.py's extra.pcie reports at 16)."""
    need_hosts()
    d, _ = make_clip(tmp_path, "plain", n=4, w=3584, h=1320)
    opts = dict(cs=5, badpix=1, stripes=1)
    want, _ = run_host("ref", d, tmp_path / "ref", opts, [vpath(0), vpath(3)])
    for host in ("amd", "amd_wrap", "amd_wrap_alloc"):
        got, _ = run_host(host, d, tmp_path / host, opts, [vpath(0), vpath(3)])
        assert got == want, host
    for bench_host in ("amd_wrap", "amd_wrap_alloc"):
      r = subprocess.run([HOSTS[bench_host], str(d), "-", "cs=5", "badpix=1", "stripes=1", "threads=8", "loops=3", "--", *[vpath(k) for k in range(4)]],
                         capture_output=True, text=True, timeout=600)
      assert r.returncode == 0 and '"fps"' in r.stderr, (bench_host, r.stderr[-2000:])
