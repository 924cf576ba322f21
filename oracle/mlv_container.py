"""CPU restatement of the reference's MLV container walk (TEST INFRASTRUCTURE ONLY -- never imported by mlvfs_amd/).

    make_index(paths)            mlvfs/index.c:216-341 (make_index) with index.c:78-99 (xref_sort: a bubble sort on `>`,
                                 hence stable) -> the XREF block, byte for byte
    idx_file(paths)              mlvfs/index.c:167-214 (save_index) -> the bytes of <name>.IDX
    frame_headers(paths, k)      mlvfs/main.c:429-558 (mlv_get_frame_headers) -> (return value, 592-byte frame_headers image)

Pinning: make_index / idx_file are checked against the reference's own index.c, which builds from its single source file
(oracle/_ref; tests/test_mlv_reader.py).  mlv_get_frame_headers lives in the reference's main.c, which needs <fuse.h> and
cannot be built here: that part of the restatement is PARITY UNPINNED (read off the source; its input, the index, is pinned).
"""
import struct

FRAME_HEADERS_SIZE = 592
# offsets inside struct frame_headers (include/mlvfs_abi.h; pinned by tests/test_abi.py) and the sizes copied per block
_SLOTS = {b"MLVI": (48, 52), b"RTCI": (100, 44), b"IDNT": (144, 84), b"RAWI": (228, 180), b"EXPO": (408, 40),
          b"LENS": (448, 96), b"WBAL": (544, 44)}
_VIDF = (16, 32)


def _scan(paths):
    entries = []                                                    # (time, file, kind, offset), scan order
    first_guid = 0
    for c, p in enumerate(paths):
        data = open(p, "rb").read()
        pos = 0
        while pos + 16 <= len(data):
            tag, size, ts = data[pos:pos + 4], *struct.unpack_from("<IQ", data, pos + 4)
            if size < 16 or size > 1024 * 1024 * 1024:
                break
            if tag == b"MLVI":
                hdr = data[pos:pos + min(52, size)]
                if len(hdr) < min(52, size):
                    break
                guid, file_num = struct.unpack_from("<QH", hdr.ljust(52, b"\0"), 16)
                if file_num == 0:
                    first_guid = guid
                elif guid != first_guid:
                    break
                ts = 0
            if tag != b"NULL":
                kind = 1 if tag == b"VIDF" else 2 if tag == b"AUDF" else 0
                entries.append((ts, c, kind, pos))
            pos += size
    n = len(entries)                                                # xref_sort, literally
    while n > 1:
        newn = 1
        for i in range(n - 1):
            if entries[i][0] > entries[i + 1][0]:
                entries[i], entries[i + 1] = entries[i + 1], entries[i]
                newn = i + 1
        n = newn
    return entries


def make_index(paths) -> bytes:
    e = _scan(paths)
    out = b"XREF" + struct.pack("<IQII", 24 + 12 * len(e), 0, 0, len(e))
    return out + b"".join(struct.pack("<HBBQ", f, 0, k, off) for _, f, k, off in e)


def idx_file(paths) -> bytes:
    head = bytearray(open(paths[0], "rb").read(52).ljust(52, b"\0"))
    struct.pack_into("<I", head, 4, 52)                             # blockSize
    struct.pack_into("<H", head, 24, len(paths) + 1)                # fileNum
    struct.pack_into("<II", head, 36, 0, 0)                         # videoFrameCount, audioFrameCount
    return bytes(head) + make_index(paths)


def frame_headers(paths, index: int):
    files = [open(p, "rb").read() for p in paths]
    fh = bytearray(FRAME_HEADERS_SIZE)
    found = rawi = False
    counter = 0
    for _, f, kind, off in _scan(paths):
        data = files[f]
        if kind == 1:
            if counter == index:
                found = True
                struct.pack_into("<I", fh, 0, f)
                struct.pack_into("<Q", fh, 8, off)
                (size,) = struct.unpack_from("<I", data, off + 4)
                n = min(_VIDF[1], size)
                fh[_VIDF[0]:_VIDF[0] + n] = data[off:off + n].ljust(n, b"\0")
                break
            counter += 1
        elif kind == 0:
            tag = data[off:off + 4]
            if tag in _SLOTS:
                (size,) = struct.unpack_from("<I", data, off + 4)
                at, cap = _SLOTS[tag]
                n = min(cap, size)
                if off + n <= len(data):                            # fread of one item of n bytes: all or nothing
                    fh[at:at + n] = data[off:off + n]
                    if tag == b"RAWI":
                        rawi = True
    return int(found and rawi), bytes(fh)
