"""Test-only LJ92 stream writer (TEST INFRASTRUCTURE): any predictor 0..7, one component, one Huffman table, byte
stuffing -- the subset the reference's decoder (mlvfs/lj92.c:512-593) accepts.  The reference's own encoder
(lj92.c:1104-1146) only writes predictor 6; this one lets the tests reach the other branches of parseScan."""
import heapq

import numpy as np


def _lengths(freq, ramp=False):
    """Huffman code lengths (<= 16) for the symbols with freq > 0; a fixed ramp when the optimum is deeper."""
    syms = [s for s, f in enumerate(freq) if f > 0]
    if len(syms) == 1:
        return {syms[0]: 1}
    if ramp:
        order = sorted(syms, key=lambda s: -freq[s])
        table = [2, 3, 3, 3, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]
        return {s: table[i] for i, s in enumerate(order)}
    heap = [(freq[s], i, (s,)) for i, s in enumerate(syms)]
    heapq.heapify(heap)
    depth = {s: 0 for s in syms}
    n = len(heap)
    while len(heap) > 1:
        a, b = heapq.heappop(heap), heapq.heappop(heap)
        for s in a[2] + b[2]:
            depth[s] += 1
        n += 1
        heapq.heappush(heap, (a[0] + b[0], n, a[2] + b[2]))
    if ramp or max(depth.values()) > 16:
        order = sorted(syms, key=lambda s: -freq[s])
        ramp = [2, 3, 3, 3, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]
        depth = {s: ramp[i] for i, s in enumerate(order)}
    return depth


def predictions(img: np.ndarray, predictor: int, bits: int) -> np.ndarray:
    x = img.astype(np.int64)
    h, w = x.shape
    left = np.zeros_like(x); left[:, 1:] = x[:, :-1]
    up = np.zeros_like(x); up[1:] = x[:-1]
    ul = np.zeros_like(x); ul[1:, 1:] = x[:-1, :-1]
    px = [np.zeros_like(x), left, up, ul, left + up - ul, left + ((up - ul) >> 1), up + ((left - ul) >> 1), (left + up) >> 1][predictor]
    px = px.copy()
    px[0, :] = left[0, :]                      # first row: the pixel to the left
    px[:, 0] = up[:, 0]                        # first column: the pixel above
    px[0, 0] = 1 << (bits - 1)
    return px


def encode(img: np.ndarray, predictor: int = 6, bits: int = 14, comment: bytes | None = None, ramp: bool = False) -> bytes:
    img = np.ascontiguousarray(img, np.uint16)
    h, w = img.shape
    diff = (img.astype(np.int64) - predictions(img, predictor, bits)).reshape(-1)
    mag = np.abs(diff)
    ssss = np.zeros(diff.size, np.int64)
    nz = mag > 0
    ssss[nz] = np.floor(np.log2(mag[nz])).astype(np.int64) + 1
    assert ssss.max() <= 16
    freq = np.bincount(ssss, minlength=17)
    depth = _lengths(list(freq), ramp)
    order = sorted(depth, key=lambda s: (depth[s], s))
    codes, code, prev = {}, 0, depth[order[0]]
    for s in order:
        code <<= depth[s] - prev
        prev = depth[s]
        codes[s] = code
        code += 1
    counts = [sum(1 for s in order if depth[s] == L) for L in range(1, 17)]
    out = bytearray(b"\xff\xd8")
    if comment is not None:
        out += b"\xff\xfe" + (len(comment) + 2).to_bytes(2, "big") + comment
    out += b"\xff\xc3" + (11).to_bytes(2, "big") + bytes([bits]) + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([1, 0, 0x11, 0])
    out += b"\xff\xc4" + (19 + len(order)).to_bytes(2, "big") + bytes([0]) + bytes(counts) + bytes(order)
    out += b"\xff\xda" + (8).to_bytes(2, "big") + bytes([1, 0, 0, predictor, 0, 0])
    extra = np.where(diff >= 0, diff, diff + (1 << ssss) - 1)
    acc, nacc = 0, 0
    body = bytearray()
    for s, e in zip(ssss.tolist(), extra.tolist()):
        acc = (acc << depth[s]) | codes[s]
        nacc += depth[s]
        if s:
            acc = (acc << s) | e
            nacc += s
        while nacc >= 8:
            byte = (acc >> (nacc - 8)) & 0xFF
            body.append(byte)
            if byte == 0xFF:
                body.append(0)
            nacc -= 8
        acc &= (1 << nacc) - 1
    if nacc:
        byte = (acc << (8 - nacc)) & 0xFF
        body.append(byte)
        if byte == 0xFF:
            body.append(0)
    return bytes(out + body + b"\xff\xd9")
