"""ctypes bindings for the CHECKERS: liboracle.so (the repo's C restatement) and
_ref/libmlvfs_ref.so (the reference's own sources compiled by oracle/Makefile).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the mlvfs_amd package.
Both classes expose the same method names so a test can be parameterised over
(oracle, reference).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libmlvfs_ref.so")

u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(quiet: bool = True) -> None:
    """(Re)build the checkers; the reference part only when /root/reference exists."""
    subprocess.run(["make", "-C", HERE] + (["-s"] if quiet else []), check=True)


def have_ref() -> bool:
    return os.path.exists(REF_SO)


REF_LUTS_SO = os.path.join(HERE, "_ref", "libref_luts.so")


def restated_ev_tables(black: int):
    """The same three tables from oracle/ref_luts.c (the repo's restatement of main.c:128-196, built alone into
    _ref/libref_luts.so): what the C-host test and the GPU library's stand-alone mode assume the caller provides."""
    L = C.CDLL(REF_LUTS_SO)
    L.get_raw2ev.restype = C.POINTER(C.c_int)
    L.get_raw2ev.argtypes = [C.c_int]
    L.get_raw2evf.restype = C.POINTER(C.c_double)
    L.get_raw2evf.argtypes = [C.c_int]
    L.get_ev2raw.restype = C.POINTER(C.c_int)
    n = 16384 + black
    r = np.ctypeslib.as_array(L.get_raw2ev(black), shape=(n,)).copy()
    rf = np.ctypeslib.as_array(L.get_raw2evf(black), shape=(n,)).copy()
    base = C.cast(L.get_ev2raw(), C.c_void_p).value - 10 * 32768 * 4
    e = np.ctypeslib.as_array(C.cast(base, C.POINTER(C.c_int)), shape=(24 * 32768,)).copy()
    return r, rf, e


class PixelList(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32)]


class Oracle:
    """liboracle.so"""
    kind = "port"

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build()
        L = self.L = C.CDLL(ORACLE_SO)
        L.orc_unpack_bits.restype = C.c_size_t
        L.orc_unpack_bits.argtypes = [u16p, u8p, C.c_int64, C.c_size_t, C.c_int]
        L.orc_chroma_smooth.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_detect_bad_pixels.restype = C.c_size_t
        L.orc_detect_bad_pixels.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            i32p, C.c_size_t]
        L.orc_apply_bad_pixels.argtypes = [u16p, C.c_int, C.c_int, C.c_int, i32p, C.c_size_t,
                                           C.c_int, C.c_int, C.c_int]
        L.orc_apply_focus_pixels.argtypes = L.orc_apply_bad_pixels.argtypes
        L.orc_stripes_compute.restype = C.c_int
        L.orc_stripes_compute.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          i32p, C.c_void_p, C.c_void_p]
        L.orc_stripes_apply.argtypes = [u16p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, i32p, C.c_int64]
        L.orc_stripes_hist_rows.restype = C.c_int64
        L.orc_stripes_hist_rows.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                            C.c_void_p, C.c_void_p]
        L.orc_hdr_preview.restype = C.c_int
        L.orc_hdr_preview.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                      C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.orc_fix_pattern_noise.argtypes = [i16p, C.c_int, C.c_int, C.c_int]
        L.orc_fix_pattern_noise_dbg.argtypes = [i16p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_cr2hdr20.restype = C.c_int
        L.orc_cr2hdr20.argtypes = [u16p] + [C.c_int] * 8 + [i32p, C.c_void_p]
        L.orc_lj92_decode.argtypes = [u8p, C.c_int, u16p]
        L.orc_lj92_info.argtypes = [u8p, C.c_int, C.c_void_p]
        L.orc_lj92_untile.argtypes = [u16p, u16p, C.c_int, C.c_int]
        L.orc_lj92_encode.argtypes = [u16p] + [C.c_int] * 5 + [C.c_void_p, C.c_int, u8p, C.c_int]
        L.orc_lj92_encode_table.argtypes = [i32p, C.c_int, i32p]
        f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        L.orc_amaze_demosaic.restype = C.c_int
        L.orc_amaze_demosaic.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p]
        L.orc_build_raw2ev.argtypes = [C.c_int, i32p, C.c_int]
        L.orc_build_ev2raw.argtypes = [i32p]
        L.orc_rand_seed.argtypes = [C.c_void_p, C.c_uint]
        L.orc_rand_next.argtypes = [C.c_void_p]
        L.orc_rand_next.restype = C.c_int
        self._libc = C.CDLL(None)

    # -- tables
    def raw2ev(self, black: int, n: int = 16384) -> np.ndarray:
        out = np.empty(n, np.int32)
        self.L.orc_build_raw2ev(black, out, n)
        return out

    def ev2raw(self) -> np.ndarray:
        out = np.empty(24 * 32768, np.int32)
        self.L.orc_build_ev2raw(out)
        return out

    # -- stages (all return new arrays; inputs are not modified)
    def lj92_info(self, data: bytes) -> dict | None:
        buf = np.frombuffer(data, np.uint8).copy()
        raw = np.zeros(320, np.uint8)
        if self.L.orc_lj92_info(buf, buf.size, raw.ctypes.data) != 0:
            return None
        w, h, bits, pred, hb, off = raw[:24].view(np.int32)
        return dict(width=int(w), height=int(h), bits=int(bits), predictor=int(pred), huffbits=int(hb), scan_offset=int(off))

    def lj92_decode(self, data: bytes):
        """-> (status, width x height uint16 image or None)."""
        info = self.lj92_info(data)
        if info is None:
            return -1, None
        buf = np.frombuffer(data, np.uint8).copy()
        out = np.zeros((info["height"], info["width"]), np.uint16)
        return self.L.orc_lj92_decode(buf, buf.size, out), out

    def lj92_encode(self, flat: np.ndarray, w: int, h: int, bits: int = 14, read_len: int = 0, skip_len: int = 0, delin=None):
        """The reference's encoder restated (lj92.c:1104-1144): w x h values read from `flat` in runs of read_len, skip_len apart.
        -> bytes, or None where the reference leaves its arrays."""
        flat = np.ascontiguousarray(flat, np.uint16).reshape(-1)
        out = np.zeros(w * h * 5 + 256, np.uint8)
        d = None if delin is None else np.ascontiguousarray(delin, np.uint16)
        n = self.L.orc_lj92_encode(flat, w, h, bits, read_len or w * h, skip_len, None if d is None else d.ctypes.data,
                                   0 if d is None else d.size, out, out.size)
        return out[:n].tobytes() if n > 0 else None

    def lj92_encode_table(self, hist, npix: int):
        """-> dict(bits[1..16], nvalues, values, len, code) or None."""
        t = np.zeros(17 + 17 + 1 + 17 + 17, np.int32)
        if self.L.orc_lj92_encode_table(np.ascontiguousarray(hist, np.int32), npix, t) != 0:
            return None
        return dict(bits=t[1:17].tolist(), nvalues=int(t[34]), values=t[17:34].tolist(), len=t[35:52].tolist(), code=t[52:69].tolist())

    def lj92_untile(self, img: np.ndarray, xres: int, yres: int) -> np.ndarray:
        src = np.ascontiguousarray(img, np.uint16).reshape(-1)
        dst = np.zeros(xres * yres, np.uint16)
        self.L.orc_lj92_untile(src, dst, xres, yres)
        return dst.reshape(yres, xres)

    def unpack(self, packed: np.ndarray, w: int, h: int, bpp: int = 14, offset: int = 0,
               max_size: int | None = None) -> np.ndarray:
        max_size = w * h * 2 if max_size is None else max_size
        out = np.zeros(max_size, np.uint8)
        self.L.orc_unpack_bits(np.ascontiguousarray(packed, np.uint16), out, offset, max_size, bpp)
        return out.view(np.uint16)

    def chroma_smooth(self, img: np.ndarray, black: int, method: int) -> np.ndarray:
        out = np.ascontiguousarray(img, np.uint16).copy()
        h, w = out.shape
        self.L.orc_chroma_smooth(out, w, h, black, method)
        return out

    def detect_bad_pixels(self, img, black, aggressive=0, crop=(0, 0)) -> np.ndarray:
        img = np.ascontiguousarray(img, np.uint16)
        h, w = img.shape
        cap = 1 << 16
        while True:
            buf = np.zeros((cap, 2), np.int32)
            n = self.L.orc_detect_bad_pixels(img, w, h, black, aggressive, crop[0], crop[1], buf.reshape(-1), cap)
            if n <= cap:
                return buf[:n].copy()
            cap = int(n)

    def apply_bad_pixels(self, img, black, pixels, crop=(0, 0), dual_iso=0) -> np.ndarray:
        out = np.ascontiguousarray(img, np.uint16).copy()
        h, w = out.shape
        px = np.ascontiguousarray(pixels, np.int32).reshape(-1)
        self.L.orc_apply_bad_pixels(out, w, h, black, px if px.size else np.zeros(2, np.int32), px.size // 2,
                                    crop[0], crop[1], dual_iso)
        return out

    def apply_focus_pixels(self, img, black, pixels, crop=(0, 0), dual_iso=0) -> np.ndarray:
        out = np.ascontiguousarray(img, np.uint16).copy()
        h, w = out.shape
        px = np.ascontiguousarray(pixels, np.int32).reshape(-1)
        self.L.orc_apply_focus_pixels(out, w, h, black, px if px.size else np.zeros(2, np.int32), px.size // 2,
                                      crop[0], crop[1], dual_iso)
        return out

    def fix_bad_pixels(self, img, black, aggressive=0, dual_iso=0, crop=(0, 0)) -> np.ndarray:
        return self.apply_bad_pixels(img, black, self.detect_bad_pixels(img, black, aggressive, crop), crop, dual_iso)

    def stripes_compute(self, img, black, white, frame_size=None, coeffs=None, reseed: bool = True,
                        want_hist: bool = False):
        """Returns (needed, coeffs[8]) (+ hist[8,65536], num[8] when want_hist)."""
        img = np.ascontiguousarray(img, np.uint16)
        h, w = img.shape
        if frame_size is None:
            frame_size = w * h * 14 // 8
        co = np.zeros(8, np.int32) if coeffs is None else np.array(coeffs, np.int32)
        if reseed:
            self._libc.srand(1)            # a fresh process starts at seed 1
        hist = np.zeros((8, 65536), np.int32) if want_hist else None
        num = np.zeros(8, np.int32) if want_hist else None
        needed = self.L.orc_stripes_compute(img, w, h, black, white, frame_size, None, co,
                                            hist.ctypes.data if want_hist else None,
                                            num.ctypes.data if want_hist else None)
        return (needed, co, hist, num) if want_hist else (needed, co)

    def stripes_hist_rows(self, img, row0, row1, black, white, rnd=None):
        """Shard form: returns accepted-call count (rnd None) or (count, hist[8*65536], num[8])."""
        img = np.ascontiguousarray(img, np.uint16)
        h, w = img.shape
        if rnd is None:
            return int(self.L.orc_stripes_hist_rows(img, w, row0, row1, black, white, None, 0, None, None))
        rnd = np.ascontiguousarray(rnd, np.uint16)
        hist = np.zeros(8 * 65536, np.int32)
        num = np.zeros(8, np.int32)
        n = self.L.orc_stripes_hist_rows(img, w, row0, row1, black, white, rnd.ctypes.data, rnd.size,
                                         hist.ctypes.data, num.ctypes.data)
        return int(n), hist, num

    def stripes_apply(self, img, black, white, needed, coeffs) -> np.ndarray:
        out = np.ascontiguousarray(img, np.uint16).copy()
        h, w = out.shape
        self.L.orc_stripes_apply(out, w * h, w, black, white, int(needed), np.array(coeffs, np.int32), 0)
        return out

    def hdr_preview(self, img, black, white):
        """Returns (converted?, image, (black, white) after conversion)."""
        out = np.ascontiguousarray(img, np.uint16).copy()
        h, w = out.shape
        a, b, s = C.c_double(), C.c_double(), C.c_int()
        r = self.L.orc_hdr_preview(out, w, h, black, white, w * h * 2, C.byref(a), C.byref(b), C.byref(s))
        lv = (black * 4, white * 4) if r else (black, white)
        return r, out, lv

    def cr2hdr20(self, img, black, white, interp_method=1, fullres=1, alias_map=1, chroma_smooth=0, bad_pix=0,
                 reset=True, want_scalars=False):
        """Full dual-ISO conversion (mean23).  reset=True forgets the per-black table caches first
        (a fresh process); returns (ret, image, (black, white))[, scalars]."""
        out = np.ascontiguousarray(img, np.uint16).copy()
        h, w = out.shape
        if reset:
            self.L.orc_dualiso_reset()
        lv = np.zeros(2, np.int32)
        sc = np.zeros(8, np.float64)
        r = self.L.orc_cr2hdr20(out, w, h, black, white, interp_method, fullres, alias_map, chroma_smooth, lv, sc.ctypes.data)
        res = (r, out, (int(lv[0]), int(lv[1])))
        return res + (sc,) if want_scalars else res

    def amaze_demosaic(self, raw):
        """raw: float32 (h, w) Bayer plane in the caller's scale; returns (red, green, blue) float32 (h, w)."""
        return _amaze(self.L.orc_amaze_demosaic, raw)

    def fix_pattern_noise(self, img, white, flags: int = 0) -> np.ndarray:
        out = np.ascontiguousarray(img).view(np.int16).copy()
        h, w = out.shape
        self.L.orc_fix_pattern_noise_dbg(out, w, h, white, flags)
        return out.view(np.uint16)

    def glibc_rand(self, n: int, seed: int = 1) -> np.ndarray:
        st = (C.c_int32 * 32)()
        self.L.orc_rand_seed(st, seed)
        return np.array([self.L.orc_rand_next(st) for _ in range(n)], np.int64)

    def process_frame(self, packed, w, h, black, white, cs=0, bad_pix=0, stripes=0, bpp=14,
                      correction=None):
        """process_frame order (mlvfs/main.c:942-997); correction = (needed, coeffs) of the
        clip or None for a clip's first frame.  Returns (image, correction)."""
        img = self.unpack(packed, w, h, bpp).reshape(h, w)
        if bad_pix:
            img = self.fix_bad_pixels(img, black, aggressive=int(bad_pix == 2))
        if cs:
            img = self.chroma_smooth(img, black, cs)
        if stripes:
            if correction is None:
                correction = self.stripes_compute(img, black, white, frame_size=w * h * bpp // 8)
            img = self.stripes_apply(img, black, white, *correction)
        return img, correction


class Reference:
    """_ref/libmlvfs_ref.so -- the reference's own code behind oracle/ref_adapter.c."""
    kind = "reference"

    def mlv_frame_headers(self, path: str, index: int):
        """mlv_get_frame_headers (main.c:429-558, the reference's text sliced into the build by oracle/Makefile) ->
        (return value, the frame_headers struct as bytes)"""
        out = np.zeros(int(self.L.ref_sizeof_frame_headers()), np.uint8)
        ok = self.L.ref_mlv_frame_headers(path.encode(), index, out)
        return ok, out.tobytes()

    def lzma_payload(self, data: bytes, level=5, dict_size=1 << 20, lc=3, lp=0, pb=2) -> bytes:
        """[u32 size][5 properties][LZMA stream] of `data` by the reference's encoder (test streams)"""
        src = np.frombuffer(data, np.uint8)
        out = np.zeros(len(data) + len(data) // 3 + 1024, np.uint8)
        self.L.ref_lzma_make_payload.restype = C.c_long
        self.L.ref_lzma_make_payload.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_int, C.c_uint, C.c_int, C.c_int, C.c_int]
        n = self.L.ref_lzma_make_payload(np.ascontiguousarray(src), src.size, out, out.size, level, dict_size, lc, lp, pb)
        assert n > 0, n
        return out[:n].tobytes()

    def lzma_uncompress(self, payload: bytes):
        """LzmaUncompress as main.c:598-616 calls it -> (return code, decoded bytes)"""
        p = np.frombuffer(payload, np.uint8)
        want = int.from_bytes(payload[:4], "little")
        out = np.zeros(max(want, 1), np.uint8)
        n = C.c_size_t(0)
        self.L.ref_lzma_uncompress.argtypes = [u8p, C.c_size_t, u8p, C.POINTER(C.c_size_t)]
        r = self.L.ref_lzma_uncompress(np.ascontiguousarray(p), p.size, out, C.byref(n))
        return r, out[:n.value].tobytes()

    def gif(self, path: str) -> bytes:
        """gif_get_data (gif.c:82-221) of a clip on disk: the whole preview file"""
        self.L.ref_gif.restype = C.c_size_t
        self.L.ref_gif.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
        n = self.L.ref_gif(path.encode(), None, 0)
        out = np.zeros(max(n, 1), np.uint8)
        got = self.L.ref_gif(path.encode(), out.ctypes.data, n)
        return out[:got].tobytes()

    def ev_tables(self, black: int):
        """get_raw2ev(black)[0..16383+black], get_raw2evf likewise, get_ev2raw()[-10*32768 .. 14*32768-1] (main.c:128-196)"""
        n = 16384 + black
        r = np.ctypeslib.as_array(self.L.get_raw2ev(black), shape=(n,)).copy()
        rf = np.ctypeslib.as_array(self.L.get_raw2evf(black), shape=(n,)).copy()
        base = C.cast(self.L.get_ev2raw(), C.c_void_p).value - 10 * 32768 * 4
        e = np.ctypeslib.as_array(C.cast(base, C.POINTER(C.c_int)), shape=(24 * 32768,)).copy()
        return r, rf, e

    def __init__(self):
        if not have_ref():
            build()
        if not have_ref():
            raise FileNotFoundError(REF_SO)
        L = self.L = C.CDLL(REF_SO)
        L.ref_unpack.restype = C.c_size_t
        L.ref_unpack.argtypes = [u16p, u8p, C.c_int64, C.c_size_t, C.c_int, C.c_int, C.c_int]
        L.ref_chroma_smooth.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_fix_bad_pixels.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_uint64]
        L.ref_fix_focus_pixels.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                           C.c_int, C.c_int]
        L.ref_stripes_compute.restype = C.c_int
        L.ref_stripes_compute.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int, i32p]
        L.ref_stripes_apply.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p]
        L.ref_hdr_preview.restype = C.c_int
        L.ref_hdr_preview.argtypes = [u16p, C.c_int, C.c_int, C.c_int, C.c_int, i32p]
        L.ref_cr2hdr20.restype = C.c_int
        L.ref_cr2hdr20.argtypes = [u16p] + [C.c_int] * 9 + [i32p]
        L.ref_fix_pattern_noise.argtypes = [i16p, C.c_int, C.c_int, C.c_int]
        L.ref_fix_pattern_noise_dbg.argtypes = [i16p, C.c_int, C.c_int, C.c_int, C.c_int]
        f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        L.ref_amaze_demosaic.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, f32p, f32p]
        L.ref_hist_median_of.restype = C.c_uint16
        L.ref_hist_median_of.argtypes = [u16p, C.c_uint32, C.c_uint16, C.c_uint16]
        L.ref_sizeof_frame_headers.restype = C.c_size_t
        L.ref_process_frame.argtypes = [u16p, u16p] + [C.c_int] * 8 + [C.c_uint64, i32p, C.POINTER(C.c_int), C.c_int]
        L.get_raw2ev.restype = C.POINTER(C.c_int)
        L.get_raw2ev.argtypes = [C.c_int]
        L.get_ev2raw.restype = C.POINTER(C.c_int)
        for f in (L.ref_mlv_new_index, L.ref_mlv_get_index):
            f.restype = C.c_size_t
            f.argtypes = [C.c_char_p, u8p, C.c_size_t]
        L.ref_mlv_frame_count.argtypes = [C.c_char_p]
        L.ref_mlv_frame_headers.argtypes = [C.c_char_p, C.c_int, u8p]
        L.get_raw2evf.restype = C.POINTER(C.c_double)
        L.get_raw2evf.argtypes = [C.c_int]
        L.ref_lj92_decode.argtypes = [u8p, C.c_int, u16p, C.c_int, i32p]
        L.ref_lj92_encode.argtypes = [u16p, C.c_int, C.c_int, C.c_int, u8p, C.c_int]
        L.ref_lj92_encode_tile.argtypes = [u16p] + [C.c_int] * 5 + [C.c_void_p, C.c_int, u8p, C.c_int]
        L.ref_header_data.restype = C.c_size_t
        L.ref_header_data.argtypes = [u8p, u8p, C.c_int64, C.c_size_t, C.c_double, C.c_char_p]
        self._libc = C.CDLL(None)

    def lj92_encode(self, img: np.ndarray, bits: int = 14) -> bytes:
        img = np.ascontiguousarray(img, np.uint16)
        h, w = img.shape
        out = np.zeros(w * h * 3 + 200, np.uint8)
        n = self.L.ref_lj92_encode(img, w, h, bits, out, out.size)
        assert n > 0, n
        return out[:n].tobytes()

    def lj92_encode_tile(self, flat: np.ndarray, w: int, h: int, bits: int = 14, read_len: int = 0, skip_len: int = 0, delin=None):
        flat = np.ascontiguousarray(flat, np.uint16).reshape(-1)
        out = np.zeros(w * h * 3 + 200, np.uint8)
        d = None if delin is None else np.ascontiguousarray(delin, np.uint16)
        n = self.L.ref_lj92_encode_tile(flat, w, h, bits, read_len or w * h, skip_len, None if d is None else d.ctypes.data,
                                        0 if d is None else d.size, out, out.size)
        assert n > 0, n
        return out[:n].tobytes()

    def lj92_decode(self, data: bytes, cap_px: int = 1 << 24):
        buf = np.frombuffer(data, np.uint8).copy()                 # the reference writes into its input (bits[0] = 0)
        dims = np.zeros(3, np.int32)
        out = np.zeros(cap_px, np.uint16)
        st = self.L.ref_lj92_decode(buf, buf.size, out, cap_px, dims)
        w, h = int(dims[0]), int(dims[1])
        return st, (out[: w * h].reshape(h, w).copy() if st == 0 else None)

    def mlv_index(self, path: str, with_idx_file: bool = False) -> bytes:
        """XREF block of a clip from the reference's index.c (with_idx_file: get_index, reads or writes <name>.IDX)."""
        buf = np.zeros(1 << 22, np.uint8)
        f = self.L.ref_mlv_get_index if with_idx_file else self.L.ref_mlv_new_index
        n = f(path.encode(), buf, buf.size)
        return buf[:n].tobytes()

    def mlv_frame_count(self, path: str) -> int:
        return self.L.ref_mlv_frame_count(path.encode())

    def header_data(self, fh_blob: np.ndarray, offset: int = 0, max_size: int = 65536, fps_override: float = 0.0,
                    basename: bytes = b"clip"):
        """dng_get_header_data on a 592-byte frame_headers image -> (bytes written, header bytes, frame_headers after)."""
        blob = np.ascontiguousarray(fh_blob, np.uint8).copy()
        out = np.zeros(max(max_size, 1), np.uint8)
        n = self.L.ref_header_data(blob, out, offset, max_size, fps_override, basename)
        return n, out[:max_size], blob

    def raw2ev(self, black: int, n: int = 16384) -> np.ndarray:
        p = self.L.get_raw2ev(black)
        return np.array([p[i] for i in range(n)], np.int32)

    def ev2raw(self) -> np.ndarray:
        p = self.L.get_ev2raw()
        addr = C.addressof(p.contents) - 10 * 32768 * 4
        return np.ctypeslib.as_array((C.c_int32 * (24 * 32768)).from_address(addr)).copy()

    def unpack(self, packed, w, h, bpp=14, offset=0, max_size=None) -> np.ndarray:
        max_size = w * h * 2 if max_size is None else max_size
        out = np.zeros(max_size, np.uint8)
        self.L.ref_unpack(np.ascontiguousarray(packed, np.uint16), out, offset, max_size, w, h, bpp)
        return out.view(np.uint16)

    def chroma_smooth(self, img, black, method) -> np.ndarray:
        out = np.ascontiguousarray(img, np.uint16).copy()
        h, w = out.shape
        self.L.ref_chroma_smooth(out, w, h, black, method)
        return out

    def fix_bad_pixels(self, img, black, aggressive=0, dual_iso=0, crop=(0, 0)) -> np.ndarray:
        """crop is given as (panPosX, panPosY); guid 0 forces re-detection."""
        out = np.ascontiguousarray(img, np.uint16).copy()
        h, w = out.shape
        self.L.ref_fix_bad_pixels(out, w, h, black, aggressive, dual_iso, crop[0], crop[1], 0)
        return out

    def fix_focus_pixels(self, img, black, dual_iso, camera, raw_w, raw_h, pan=(0, 0)) -> np.ndarray:
        out = np.ascontiguousarray(img, np.uint16).copy()
        h, w = out.shape
        self.L.ref_fix_focus_pixels(out, w, h, black, dual_iso, camera, raw_w, raw_h, pan[0], pan[1])
        return out

    def stripes_compute(self, img, black, white, frame_size=None, coeffs=None, reseed=True):
        img = np.ascontiguousarray(img, np.uint16)
        h, w = img.shape
        co = np.zeros(8, np.int32) if coeffs is None else np.array(coeffs, np.int32)
        if reseed:
            self._libc.srand(1)
        needed = self.L.ref_stripes_compute(img, w, h, black, white, co)
        return needed, co

    def stripes_apply(self, img, black, white, needed, coeffs) -> np.ndarray:
        out = np.ascontiguousarray(img, np.uint16).copy()
        h, w = out.shape
        self.L.ref_stripes_apply(out, w, h, black, white, int(needed), np.array(coeffs, np.int32))
        return out

    def hdr_preview(self, img, black, white):
        out = np.ascontiguousarray(img, np.uint16).copy()
        h, w = out.shape
        lv = np.zeros(2, np.int32)
        r = self.L.ref_hdr_preview(out, w, h, black, white, lv)
        return r, out, (int(lv[0]), int(lv[1]))

    def amaze_demosaic(self, raw):
        return _amaze(self.L.ref_amaze_demosaic, raw)

    def cr2hdr20(self, img, black, white, interp_method=0, fullres=1, alias_map=1, chroma_smooth=0, bad_pix=0):
        out = np.ascontiguousarray(img, np.uint16).copy()
        h, w = out.shape
        lv = np.zeros(2, np.int32)
        r = self.L.ref_cr2hdr20(out, w, h, black, white, interp_method, fullres, alias_map, chroma_smooth,
                                bad_pix, lv)
        return r, out, (int(lv[0]), int(lv[1]))

    def fix_pattern_noise(self, img, white, flags: int = 0) -> np.ndarray:
        out = np.ascontiguousarray(img).view(np.int16).copy()
        h, w = out.shape
        self.L.ref_fix_pattern_noise_dbg(out, w, h, white, flags)
        return out.view(np.uint16)

    def hist_median(self, data, skip, white) -> int:
        d = np.ascontiguousarray(data, np.uint16)
        return int(self.L.ref_hist_median_of(d, d.size, skip, white))

    def process_frame(self, packed, w, h, black, white, cs=0, bad_pix=0, stripes=0, bpp=14, correction=None, guid=0):
        """guid 0: the bad-pixel map is detected anew on every frame (cs.c:235); any other value: once per clip."""
        img = np.zeros((h, w), np.uint16)
        co = np.zeros(8, np.int32) if correction is None else np.array(correction[1], np.int32)
        needed = C.c_int(0 if correction is None else int(correction[0]))
        if correction is None:
            self._libc.srand(1)
        self.L.ref_process_frame(np.ascontiguousarray(packed, np.uint16), img, w, h, bpp, black, white,
                                 cs, bad_pix, stripes, C.c_uint64(guid), co, C.byref(needed), int(correction is None))
        return img, ((needed.value, co) if stripes else None)


def _amaze(fn, raw):
    raw = np.asarray(raw, np.float32)
    h, w = raw.shape
    pitch = w + 16
    planes = [np.zeros((h, pitch), np.float32) for _ in range(4)]
    planes[0][:, :w] = raw
    fn(planes[0], w, h, pitch, planes[1], planes[2], planes[3])
    return tuple(p[:, :w].copy() for p in planes[1:])
