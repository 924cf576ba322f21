"""lj92_encode (lj92.h:65-68, lj92.c:711-1144) -- the encoder that completes lj92.o's export table.
CPU: the oracle's restatement against the reference's own encoder, byte for byte; the library's table builder (host code) against
both.  GPU: the library's lj92_encode (csrc/k_lj92enc.hip + lj92enc.cpp) against oracle and reference, and the round trip through
the library's own decoder."""
import numpy as np
import pytest

from mlvfs_amd import lj92


def material(w, h, kind, seed, bits=14):
    rng = np.random.default_rng(seed)
    top = (1 << bits) - 1
    if kind == "flat":                        # one class only: every difference zero but the first
        return np.full((h, w), 1 << (bits - 1), np.uint16)
    if kind == "ramp":
        return ((np.arange(w)[None, :] * 3 + np.arange(h)[:, None] * 5) % (top + 1)).astype(np.uint16)
    if kind == "noise":
        return rng.integers(0, top + 1, (h, w)).astype(np.uint16)
    if kind == "sparse":                      # long runs of zero differences (class 0 takes the all-ones code: many 0xFF bytes)
        x = np.full((h, w), 1000, np.uint16)
        m = rng.random((h, w)) < 0.01
        x[m] = rng.integers(0, top + 1, int(m.sum()))
        return x
    x = rng.normal(3000, 40, (h, w)) + np.linspace(0, 6000, w)[None, :]
    return np.clip(x, 0, top).astype(np.uint16)


CASES = [(64, 48, "smooth"), (64, 48, "flat"), (33, 17, "noise"), (130, 9, "ramp"), (256, 64, "sparse"), (1, 1, "noise"), (1, 50, "smooth"),
         (50, 1, "smooth"), (2, 2, "flat"), (640, 360, "smooth")]


@pytest.mark.parametrize("w,h,kind", CASES)
def test_oracle_encoder_equals_reference(oracle, reference, w, h, kind):
    for seed in range(3):
        img = material(w, h, kind, seed)
        want = reference.lj92_encode_tile(img, w, h, 14)
        got = oracle.lj92_encode(img, w, h, 14)
        assert got == want, (len(got or b""), len(want))
        st, back = oracle.lj92_decode(got)
        assert st == 0 and np.array_equal(back, img)


def test_oracle_encoder_tiles_bit_depths_and_tables(oracle, reference):
    rng = np.random.default_rng(5)
    big = material(96, 40, "smooth", 9)
    # a 32 x 20 tile out of a 96-wide image: runs of 32, 64 apart (lj92.c:766-769); a 2 x (32 x 10) interleave: width 64 read in runs of 32
    for (w, h, rl, sk) in [(32, 20, 32, 64), (64, 10, 32, 64), (48, 13, 16, 80), (96, 40, 96, 0), (24, 7, 5, 3)]:
        want = reference.lj92_encode_tile(big, w, h, 14, rl, sk)
        assert oracle.lj92_encode(big, w, h, 14, rl, sk) == want
    for bits in (8, 10, 12, 14, 15):
        img = material(40, 30, "noise", bits, bits)
        assert oracle.lj92_encode(img, 40, 30, bits) == reference.lj92_encode_tile(img, 40, 30, bits)
    # delinearisation table (lj92.c:750): values mapped before prediction
    table = np.sort(rng.integers(0, 4096, 16384)).astype(np.uint16)
    img = material(64, 32, "smooth", 2)
    assert oracle.lj92_encode(img, 64, 32, 12, delin=table) == reference.lj92_encode_tile(img, 64, 32, 12, delin=table)


def histograms(n=300, seed=11):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        k = int(rng.integers(1, 17))                                # classes in use (17 is refused)
        classes = rng.choice(17, k, replace=False)
        h = np.zeros(17, np.int64)
        style = i % 4
        if style == 0:
            h[classes] = rng.integers(1, 1000, k)
        elif style == 1:
            h[classes] = 2 ** rng.integers(0, 12, k)                # many exact ties in float
        elif style == 2:
            h[classes] = 7                                          # all equal
        else:
            h[classes] = np.maximum(1, (rng.random(k) ** 6 * 1e6).astype(np.int64))
        out.append(h)
    return out


def test_table_builder_of_the_library_equals_the_oracle(amd, oracle):
    """Host code only (no device): ties, equal frequencies, skewed histograms."""
    deep = 0
    for h in histograms():
        npix = int(h.sum())
        want = oracle.lj92_encode_table(h, npix)
        got = lj92.encode_table(h, npix)
        assert (got is None) == (want is None)
        if want is None:
            deep += 1
            continue
        assert got["nvalues"] == want["nvalues"] and got["bits"] == want["bits"] and got["values"] == want["values"]
        used = [s for s in range(17) if h[s]]
        assert [got["len"][s] for s in used] == [want["len"][s] for s in used]
        assert [got["code"][s] for s in used] == [want["code"][s] for s in used]
        # what the module's comment says about the reference's table: class 0 is written with the last, all-ones code
        n = want["nvalues"]
        longest = max(l for l in range(1, 17) if want["bits"][l - 1])
        assert want["values"][n - 1] == 0 and want["len"][0] == longest and want["code"][0] == (1 << longest) - 1
    assert deep < 30
    assert lj92.encode_table(np.zeros(17, np.int64), 0) is None
    assert lj92.encode_table(np.ones(17, np.int64), 17) is None      # all 17 classes: the reference writes behind its tables


def test_oracle_table_equals_the_reference_through_streams(oracle, reference):
    """The table is pinned through whole streams: images built to hit a given class histogram exactly are awkward, so the
    first row carries the differences (row 0 predicts from the left neighbour) and the rest repeats it (class 0)."""
    rng = np.random.default_rng(3)
    for trial in range(40):
        w = 200
        k = int(rng.integers(1, 13))
        classes = rng.choice(np.arange(1, 14), k, replace=False)
        steps = []
        for c in classes:
            steps += [int(rng.integers(1 << (c - 1), 1 << c)) * (1 if rng.random() < 0.5 else -1) for _ in range(int(rng.integers(1, 12)))]
        steps = (steps * (w // len(steps) + 1))[: w - 1]
        row = [8192]
        for s in steps:
            nxt = row[-1] + s
            if not 0 <= nxt < 16384:
                nxt = row[-1] - s
            row.append(min(max(nxt, 0), 16383))
        img = np.tile(np.array(row, np.uint16), (int(rng.integers(1, 6)), 1))
        h = img.shape[0]
        assert oracle.lj92_encode(img, w, h, 14) == reference.lj92_encode_tile(img, w, h, 14)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,kind", CASES + [(1920, 1080, "smooth"), (3584, 1320, "sparse")])
def test_gpu_encoder_equals_oracle(gpu, oracle, w, h, kind):
    img = material(w, h, kind, 1)
    want = oracle.lj92_encode(img, w, h, 14)
    got = lj92.encode(img, w, h, 14)
    assert got == want, (len(got), len(want))


@pytest.mark.gpu
def test_gpu_encoder_equals_reference_and_round_trips(gpu, oracle, reference):
    big = material(96, 40, "smooth", 9)
    for (w, h, rl, sk) in [(32, 20, 32, 64), (64, 10, 32, 64), (48, 13, 16, 80), (96, 40, 96, 0), (24, 7, 5, 3)]:
        assert lj92.encode(big, w, h, 14, rl, sk) == reference.lj92_encode_tile(big, w, h, 14, rl, sk)
    for bits in (8, 10, 12, 14, 15):
        img = material(40, 30, "noise", bits, bits)
        assert lj92.encode(img, 40, 30, bits) == reference.lj92_encode_tile(img, 40, 30, bits)
    table = np.sort(np.random.default_rng(5).integers(0, 4096, 16384)).astype(np.uint16)
    img = material(64, 32, "smooth", 2)
    assert lj92.encode(img, 64, 32, 12, delin=table) == reference.lj92_encode_tile(img, 64, 32, 12, delin=table)
    # encode -> the library's GPU decoder -> the pixels (size-independent property, full size)
    full = material(3584, 1320, "smooth", 4)
    s = lj92.encode(full, 3584, 1320, 14)
    assert s == reference.lj92_encode_tile(full, 3584, 1320, 14)
    import torch
    out = torch.empty((1, 1320, 3584), dtype=torch.int16, device="cuda")
    import ctypes as C
    from mlvfs_amd import lib
    buf = np.frombuffer(s, np.uint8)
    ptrs = (C.c_void_p * 1)(buf.ctypes.data)
    sizes = (C.c_size_t * 1)(buf.size)
    lib.check(gpu.mlvfs_amd_lj92_decode_dev(ptrs, sizes, 1, 0, 0, C.c_void_p(out.data_ptr()), out.stride(0) * 2, None), "decode")
    assert np.array_equal(out.cpu().numpy().view(np.uint16)[0], full)


@pytest.mark.gpu
def test_gpu_encoder_refuses_what_the_reference_cannot_encode(gpu):
    img = np.zeros((4, 64), np.uint16)
    img[:, ::2] = 65535                                   # differences of 17 bits in rows below the first
    img[1::2] = 65535 - img[1::2]
    with pytest.raises(Exception, match="17 bits"):
        lj92.encode(img, 64, 4, 16)
    with pytest.raises(Exception, match="delinearisation"):
        lj92.encode(np.full((4, 4), 100, np.uint16), 4, 4, 14, delin=np.arange(50, dtype=np.uint16))
    with pytest.raises(Exception):
        lj92.encode(np.zeros(4, np.uint16), 0, 4, 14)
