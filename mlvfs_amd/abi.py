"""ctypes mirror of `struct frame_headers` (include/mlvfs_abi.h; reference
mlvfs/mlvfs.h:51-63, mlvfs/mlv.h:40-239, mlvfs/raw.h:166-207).

Used by the host-side mirror to call the drop-in symbols of libmlvfs_amd.so the
way MLVFS's main.c does.  tests/test_abi.py checks sizes/offsets against the C
header (and against the reference's headers when /root/reference is present).
"""
import ctypes as C

u8, u16, u32, u64, i32 = C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64, C.c_int32


class RawInfo(C.Structure):
    _fields_ = [
        ("api_version", u32), ("do_not_use_this", u32),
        ("height", i32), ("width", i32), ("pitch", i32),
        ("frame_size", i32), ("bits_per_pixel", i32),
        ("black_level", i32), ("white_level", i32),
        ("crop", i32 * 4), ("active_area", i32 * 4),
        ("exposure_bias", i32 * 2), ("cfa_pattern", i32), ("calibration_illuminant1", i32),
        ("color_matrix1", i32 * 18), ("dynamic_range", i32),
    ]


def _block(fields):
    return [("blockType", u8 * 4), ("blockSize", u32), ("timestamp", u64)] + fields


class FileHdr(C.Structure):
    _pack_ = 1
    _fields_ = [("fileMagic", u8 * 4), ("blockSize", u32), ("versionString", u8 * 8), ("fileGuid", u64),
                ("fileNum", u16), ("fileCount", u16), ("fileFlags", u32), ("videoClass", u16), ("audioClass", u16),
                ("videoFrameCount", u32), ("audioFrameCount", u32), ("sourceFpsNom", u32), ("sourceFpsDenom", u32)]


class VidfHdr(C.Structure):
    _pack_ = 1
    _fields_ = _block([("frameNumber", u32), ("cropPosX", u16), ("cropPosY", u16), ("panPosX", u16),
                       ("panPosY", u16), ("frameSpace", u32)])


class RawiHdr(C.Structure):
    _pack_ = 1
    _fields_ = _block([("xRes", u16), ("yRes", u16), ("raw_info", RawInfo)])


class ExpoHdr(C.Structure):
    _pack_ = 1
    _fields_ = _block([("isoMode", u32), ("isoValue", u32), ("isoAnalog", u32), ("digitalGain", u32),
                       ("shutterValue", u64)])


class LensHdr(C.Structure):
    _pack_ = 1
    _fields_ = _block([("focalLength", u16), ("focalDist", u16), ("aperture", u16), ("stabilizerMode", u8),
                       ("autofocusMode", u8), ("flags", u32), ("lensID", u32), ("lensName", u8 * 32),
                       ("lensSerial", u8 * 32)])


class RtciHdr(C.Structure):
    _pack_ = 1
    _fields_ = _block([(n, u16) for n in ("tm_sec", "tm_min", "tm_hour", "tm_mday", "tm_mon", "tm_year", "tm_wday",
                                          "tm_yday", "tm_isdst", "tm_gmtoff")] + [("tm_zone", u8 * 8)])


class IdntHdr(C.Structure):
    _pack_ = 1
    _fields_ = _block([("cameraName", u8 * 32), ("cameraModel", u32), ("cameraSerial", u8 * 32)])


class WbalHdr(C.Structure):
    _pack_ = 1
    _fields_ = _block([(n, u32) for n in ("wb_mode", "kelvin", "wbgain_r", "wbgain_g", "wbgain_b", "wbs_gm", "wbs_ba")])


class FrameHeaders(C.Structure):
    _fields_ = [("fileNumber", u32), ("position", u64), ("vidf_hdr", VidfHdr), ("file_hdr", FileHdr),
                ("rtci_hdr", RtciHdr), ("idnt_hdr", IdntHdr), ("rawi_hdr", RawiHdr), ("expo_hdr", ExpoHdr),
                ("lens_hdr", LensHdr), ("wbal_hdr", WbalHdr)]


class StripesCorrection(C.Structure):
    pass


StripesCorrection._fields_ = [("next", C.POINTER(StripesCorrection)), ("mlv_filename", C.c_char_p),
                              ("correction_needed", C.c_int), ("coeffficients", C.c_int * 8)]


def make_frame_headers(w, h, bpp=14, black=2048, white=15000, guid=0, pan=(0, 0), camera=0,
                       raw_size=None) -> FrameHeaders:
    """Fill the fields the hot path reads."""
    fh = FrameHeaders()
    fh.rawi_hdr.xRes, fh.rawi_hdr.yRes = w, h
    ri = fh.rawi_hdr.raw_info
    ri.width, ri.height = raw_size if raw_size else (w, h)
    ri.pitch = w * bpp // 8
    ri.frame_size = w * h * bpp // 8
    ri.bits_per_pixel = bpp
    ri.black_level, ri.white_level = black, white
    ri.cfa_pattern = 0x02010100
    fh.file_hdr.fileGuid = guid
    fh.vidf_hdr.panPosX, fh.vidf_hdr.panPosY = pan
    fh.idnt_hdr.cameraModel = camera
    return fh
