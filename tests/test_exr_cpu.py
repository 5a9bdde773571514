"""OpenEXR reader / writer and the on-disk datasets (the data format in front of the hot path).

No EXR file and no EXR library exists in the image, so the reader is pinned by files ASSEMBLED BY HAND in this test from the
published file layout (header attributes, line-offset table, chunk framing; the ZIP predictor / byte interleave; an RLE
stream written out byte by byte; a PIZ block produced by a test-side encoder restating ImfPizCompressor / ImfHuf / ImfWav's
forward direction) -- independent of reni_amd.exr.write_exr, which is then checked against the reader as well."""
import heapq
import os
import struct
import zlib

import numpy as np
import pytest
import torch

from reni_amd import exr
from reni_amd.custom_transforms import MinMaxNormalise, Resize, transform_builder
from reni_amd.data import RENIDatasetHDR, RENIDatasetLDR, get_dataset, natsorted


# ------------------------------------------------------------------ a tiny independent assembler
def _attr(name, typ, val):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(val)) + val


def _assemble(nx, ny, chans, comp, blocks, version=2, lpb=1):
    """chans: [(name, ptype)] alphabetical; blocks: list of (first line, payload bytes)"""
    chl = b"".join(n.encode() + b"\0" + struct.pack("<i", pt) + b"\0\0\0\0" + struct.pack("<ii", 1, 1) for n, pt in chans) + b"\0"
    box = struct.pack("<iiii", 0, 0, nx - 1, ny - 1)
    h = struct.pack("<ii", 20000630, version)
    h += _attr("channels", "chlist", chl) + _attr("compression", "compression", bytes([comp]))
    h += _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box)
    h += _attr("lineOrder", "lineOrder", b"\0") + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    h += _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + _attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    h += b"\0"
    pos = len(h) + 8 * len(blocks)
    table, body = b"", b""
    for y, payload in blocks:
        table += struct.pack("<Q", pos)
        chunk = struct.pack("<ii", y, len(payload)) + payload
        body += chunk
        pos += len(chunk)
    return h + table + body


def test_hand_assembled_uncompressed_half_file(tmp_path):
    # 2 x 2, channels B, G, R (alphabetical in the file), half floats given as their bit patterns
    ONE, HALF_, TWO, MINUS1, MAXH, ZERO = 0x3C00, 0x3800, 0x4000, 0xBC00, 0x7BFF, 0x0000
    line0 = struct.pack("<6H", ONE, TWO, HALF_, HALF_, MINUS1, ZERO)       # B(0,0) B(0,1) | G | R
    line1 = struct.pack("<6H", MAXH, ZERO, ONE, ONE, TWO, TWO)
    p = tmp_path / "a.exr"
    p.write_bytes(_assemble(2, 2, [("B", 1), ("G", 1), ("R", 1)], 0, [(0, line0), (1, line1)]))
    img = exr.read_exr(str(p))
    assert img.shape == (2, 2, 3) and img.dtype == np.float32
    want = np.array([[[-1, 0.5, 1], [0, 0.5, 2]], [[2, 1, 65504], [2, 1, 0]]], np.float32)  # [y][x][R, G, B]
    assert np.array_equal(img, want)


def test_hand_assembled_zip_float_and_rle_blocks(tmp_path):
    # ZIP: one block of 16 lines; payload = deflate(predictor(interleave(raw))) written out here step by step
    nx, ny = 5, 19
    rng = np.random.RandomState(1)
    Y = np.round(rng.rand(ny, nx) * 8).astype("<f4")  # few distinct values: compressible
    blocks = []
    for r0 in range(0, ny, 16):
        raw = Y[r0:r0 + 16].tobytes()
        half = (len(raw) + 1) // 2
        inter = bytes(raw[0::2]) + bytes(raw[1::2])
        assert len(inter[:half]) == half
        pred = bytes([inter[0]] + [(inter[i] - inter[i - 1] + 128) & 255 for i in range(1, len(inter))])
        z = zlib.compress(pred)
        assert len(z) < len(raw)
        blocks.append((r0, z))
    p = tmp_path / "z.exr"
    p.write_bytes(_assemble(nx, ny, [("Y", 2)], 3, blocks))
    got = exr.read_exr(str(p))
    assert got.shape == (ny, nx) and np.array_equal(got, Y.astype(np.float32))
    # RLE: one line of eight half 1.0 pixels.  raw = 00 3C x 8; interleaved = 00 x 8, 3C x 8; predicted = 00, 80 x 7, BC, 80 x 7
    rle = bytes([0xFF, 0x00, 0x06, 0x80, 0xFF, 0xBC, 0x06, 0x80])  # literal 00 | run 7 x 80 | literal BC | run 7 x 80
    q = tmp_path / "r.exr"
    q.write_bytes(_assemble(8, 1, [("Y", 1)], 1, [(0, rle)]))
    assert np.array_equal(exr.read_exr(str(q)), np.ones((1, 8), np.float32))


# ------------------------------------------------------------------ PIZ: forward direction restated for the test
def _wenc14(a, b):
    a, b = np.int16(a), np.int16(b)
    m = (int(a) + int(b)) >> 1
    d = int(a) - int(b)
    return np.uint16(m & 0xffff), np.uint16(d & 0xffff)


def _wenc16(a, b):
    ao = (int(a) + 0x8000) & 0xffff
    m = (ao + int(b)) >> 1
    d = ao - int(b)
    if d < 0:
        m = (m + 0x8000) & 0xffff
    return np.uint16(m), np.uint16(d & 0xffff)


def _wav2_encode(a, mx):
    enc = _wenc14 if mx < (1 << 14) else _wenc16
    ny, nx = a.shape
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        y = 0
        while y <= ny - p2:
            x = 0
            while x <= nx - p2:
                i00, i01 = enc(a[y, x], a[y, x + p])
                i10, i11 = enc(a[y + p, x], a[y + p, x + p])
                a[y, x], a[y + p, x] = enc(i00, i10)
                a[y, x + p], a[y + p, x + p] = enc(i01, i11)
                x += p2
            if nx & p:
                a[y, x], a[y + p, x] = enc(a[y, x], a[y + p, x])
            y += p2
        if ny & p:
            x = 0
            while x <= nx - p2:
                a[y, x], a[y, x + p] = enc(a[y, x], a[y, x + p])
                x += p2
        p = p2
        p2 <<= 1


class _BitOut:
    def __init__(self):
        self.c, self.lc, self.out, self.n = 0, 0, bytearray(), 0

    def put(self, nbits, v):
        self.c = (self.c << nbits) | v
        self.lc += nbits
        self.n += nbits
        while self.lc >= 8:
            self.lc -= 8
            self.out.append((self.c >> self.lc) & 0xff)
        self.c &= (1 << self.lc) - 1

    def flush(self):
        if self.lc:
            self.out.append((self.c << (8 - self.lc)) & 0xff)
            self.c, self.lc = 0, 0


def _huf_compress(sym):
    freq = np.bincount(sym, minlength=65537).astype(np.int64)
    used = np.nonzero(freq)[0]
    im, iM = int(used[0]), int(used[-1]) + 1
    freq[iM] = 1  # the run-length pseudo symbol
    heap = [(int(freq[s]), int(s), (int(s),)) for s in np.nonzero(freq)[0]]
    heapq.heapify(heap)
    length = {s: 0 for _, s, _ in heap}
    if len(heap) == 1:
        length[heap[0][1]] = 1
    while len(heap) > 1:
        fa, ka, sa = heapq.heappop(heap)
        fb, kb, sb = heapq.heappop(heap)
        for s in sa + sb:
            length[s] += 1
        heapq.heappush(heap, (fa + fb, min(ka, kb), sa + sb))
    assert max(length.values()) <= 58
    n = [0] * 60
    for l in length.values():
        n[l] += 1
    c = 0
    for l in range(58, 0, -1):
        nc = (c + n[l]) >> 1
        n[l] = c
        c = nc
    code = {}
    for s in sorted(length):
        code[s] = (length[s], n[length[s]])
        n[length[s]] += 1
    tb = _BitOut()  # the packed table: 6-bit lengths, zero runs run-length coded
    s = im
    while s <= iM:
        l = length.get(s, 0)
        if l == 0:
            z = 1
            while s + z <= iM and length.get(s + z, 0) == 0 and z < 255 + 6:
                z += 1
            if z >= 6:
                tb.put(6, 63); tb.put(8, z - 6)
            elif z >= 2:
                tb.put(6, 59 + z - 2)
            else:
                tb.put(6, 0)
            s += z
        else:
            tb.put(6, l)
            s += 1
    tb.flush()
    db = _BitOut()

    def send(s, cs):
        ls, lr = code[s][0], code[iM][0]
        if ls + lr + 8 < ls * cs:
            db.put(*code[s]); db.put(*code[iM]); db.put(8, cs)
        else:
            for _ in range(cs + 1):
                db.put(*code[s])

    s, cs = int(sym[0]), 0
    for v in sym[1:]:
        v = int(v)
        if v == s and cs < 255:
            cs += 1
        else:
            send(s, cs)
            cs = 0
        s = v
    send(s, cs)
    nbits = db.n
    db.flush()
    return struct.pack("<IIIII", im, iM, len(tb.out), nbits, 0) + bytes(tb.out) + bytes(db.out)


def _piz_block(planes16):
    """planes16: list of (uint16 array [ny][nx * size], size) per channel, size = 16-bit words per pixel"""
    allv = np.concatenate([p.reshape(-1) for p, _ in planes16])
    present = np.zeros(65536, bool)
    present[allv] = True
    bitmap = np.packbits(present, bitorder="little")
    bitmap[0] &= 0xfe  # zero is implied
    nz = np.nonzero(bitmap)[0]
    lo, hi = (int(nz[0]), int(nz[-1])) if nz.size else (8191, 0)
    present[0] = True
    fwd = np.cumsum(present) - 1
    mx = int(present.sum()) - 1
    stream = []
    for p, size in planes16:
        ny = p.shape[0]
        t = fwd[p].astype(np.uint16).reshape(ny, -1, size)
        for k in range(size):
            pl = np.ascontiguousarray(t[:, :, k])
            _wav2_encode(pl, mx)
            t[:, :, k] = pl
        stream.append(t.reshape(-1))
    huf = _huf_compress(np.concatenate(stream).astype(np.int64))
    out = struct.pack("<HH", lo, hi)
    if lo <= hi:
        out += bitmap[lo:hi + 1].tobytes()
    return out + struct.pack("<i", len(huf)) + huf


@pytest.mark.parametrize("case", ["half_small_range", "half_wide_range", "float"])
def test_piz_blocks_from_the_test_side_encoder(tmp_path, case):
    rng = np.random.RandomState(3)
    nx, ny = 13, 37  # odd sizes: the wavelet's leftover column and line; 32-line blocks: one full, one ragged
    if case == "half_small_range":   # < 2^14 distinct values: the 14-bit wavelet
        img = (np.round(rng.rand(ny, nx, 2) * 20) / 4).astype(np.float16)
        img[5:20, 2:9] = 1.5       # flat regions: zero wavelet coefficients, runs in the Huffman stream
    elif case == "half_wide_range":  # every half bit pattern of a large set: the 16-bit (modulo) wavelet
        img = rng.randint(0, 0x7bff, size=(ny, nx, 2)).astype(np.uint16).view(np.float16)
        extra = np.arange(20000, dtype=np.uint16)[: ny * nx * 2].reshape(-1)
        img.reshape(-1)[: extra.size].view(np.uint16)[:] = (extra * 3) % 0x7bff
    else:
        img = np.round(rng.rand(ny, nx, 2) * 6).astype(np.float32)
    pt = 2 if case == "float" else 1
    size = 2 if case == "float" else 1
    blocks = []
    for r0 in range(0, ny, 32):
        rows = img[r0:r0 + 32]
        planes = [(np.ascontiguousarray(rows[:, :, c]).view(np.uint16).reshape(rows.shape[0], nx * size), size) for c in (0, 1)]
        raw_len = rows.shape[0] * nx * 2 * size * 2
        blk = _piz_block(planes)
        if len(blk) >= raw_len:  # (the format stores a block raw when the codec does not shrink it)
            blk = b"".join(np.ascontiguousarray(rows[y, :, c]).tobytes() for y in range(rows.shape[0]) for c in (0, 1))
        blocks.append((r0, blk))
    if case == "half_small_range":
        assert all(len(b) < 32 * nx * 4 for _, b in blocks[:1])  # really compressed, not the raw fallback
    p = tmp_path / "p.exr"
    p.write_bytes(_assemble(nx, ny, [("A", pt), ("Z", pt)], 4, blocks))
    planes, _ = exr.read_exr_channels(str(p))
    for c, name in enumerate(("A", "Z")):
        want = img[:, :, c].astype(np.float32)
        assert np.array_equal(planes[name].view(np.uint32), want.view(np.uint32)), (case, name)


# ------------------------------------------------------------------ writer <-> reader
@pytest.mark.parametrize("pixel_type", ["half", "float"])
@pytest.mark.parametrize("compression", ["none", "rle", "zips", "zip"])
def test_write_read_round_trip(tmp_path, pixel_type, compression):
    yy, xx = np.meshgrid(np.arange(45), np.arange(70), indexing="ij")
    img = np.stack([np.exp(np.sin(xx / 9.0) * 3), (yy // 8).astype(np.float64), np.full(xx.shape, 0.25)], -1).astype(np.float32)
    img[3, 4, 0] = np.inf
    img[6, 7, 1] = np.nan
    p = str(tmp_path / "w.exr")
    exr.write_exr(p, img, pixel_type=pixel_type, compression=compression)
    raw_size = img.size * (2 if pixel_type == "half" else 4)
    if compression != "none":
        assert os.path.getsize(p) < raw_size  # the compressed paths really ran (smooth data)
    got = exr.read_exr(p)
    want = img.astype(np.float16).astype(np.float32) if pixel_type == "half" else img
    assert got.dtype == np.float32 and np.array_equal(got, want, equal_nan=True)
    got4 = None
    exr.write_exr(p, np.concatenate([img, img[:, :, :1]], -1), pixel_type=pixel_type, compression=compression)
    got4 = exr.read_exr(p)
    assert got4.shape == (45, 70, 4) and np.array_equal(got4[:, :, :3], want, equal_nan=True)


def test_unsupported_files_fail_loudly(tmp_path):
    p = tmp_path / "t.exr"
    p.write_bytes(_assemble(2, 1, [("Y", 1)], 0, [(0, b"\0" * 4)], version=2 | 0x200))
    with pytest.raises(NotImplementedError, match="tiled"):
        exr.read_exr(str(p))
    p.write_bytes(_assemble(2, 1, [("Y", 1)], 5, [(0, b"\0" * 4)]))
    with pytest.raises(NotImplementedError, match="PXR24"):
        exr.read_exr(str(p))
    p.write_bytes(b"\0" * 64)
    with pytest.raises(ValueError, match="magic"):
        exr.read_exr(str(p))
    good = _assemble(2, 2, [("Y", 1)], 0, [(0, b"\0" * 4), (1, b"\0" * 4)])
    p.write_bytes(good[:-6])
    with pytest.raises(ValueError):
        exr.read_exr(str(p))


# ------------------------------------------------------------------ datasets (src/data/datasets.py)
def _hdr(seed, h=16, w=32):
    rng = np.random.RandomState(seed)
    return np.exp(rng.randn(h, w, 3) * 2 - 3).astype(np.float32)


def test_hdr_dataset_matches_the_reference_pipeline(tmp_path):
    d = tmp_path / "Train"
    d.mkdir()
    names = ["env10.exr", "env2.exr", "env1.exr"]
    imgs = {n: _hdr(i) for i, n in enumerate(names)}
    imgs["env2.exr"][0, 0, 0] = 0.0       # a zero and an inf pixel: MinMaxNormalise clips to [min positive, max finite]
    imgs["env2.exr"][1, 1, 1] = np.inf
    for n, a in imgs.items():
        exr.write_exr(str(d / n), a, pixel_type="float", compression="zip")
    (d / "notes.txt").write_text("not an image")
    assert natsorted(names) == ["env1.exr", "env2.exr", "env10.exr"]
    tf = transform_builder([["resize", [8, 16]], ["minmaxnormalise", []]])
    ds = get_dataset("RENI_HDR", str(d), tf, True)
    assert isinstance(ds, RENIDatasetHDR) and len(ds) == 3 and ds.img_names == ["env1.exr", "env2.exr", "env10.exr"]
    # the dataset computed its own log-domain min / max (datasets.py:87-99)
    lo, hi = float("inf"), float("-inf")
    for n in ds.img_names:
        a = torch.from_numpy(imgs[n])
        a = torch.clip(a, a[a > 0].min(), a[a < torch.inf].max()).log()
        lo, hi = min(lo, float(a.min())), max(hi, float(a.max()))
    mm = [t for t in tf.transforms if isinstance(t, MinMaxNormalise)][0].minmax
    assert mm == [lo, hi] and ds.unnormalise is not None
    img, idx = ds[1]
    assert idx == 1 and img.shape == (3, 8, 16) and img.dtype == torch.float32 and bool(torch.isfinite(img).all())
    x = torch.from_numpy(imgs["env2.exr"].transpose(2, 0, 1))
    x = torch.nn.functional.interpolate(x[None], size=(8, 16), mode="bilinear", align_corners=False)[0]
    x = torch.clip(x, x[x > 0].min(), x[x < torch.inf].max()).log()
    assert torch.equal(img, torch.nan_to_num(2 * (x - lo) / (hi - lo) - 1))
    # un-normalise inverts it (custom_transforms.py:14-21), and the curriculum hook doubles the Resize
    back = ds.unnormalise(ds[0][0][None])
    x0 = torch.nn.functional.interpolate(torch.from_numpy(imgs["env1.exr"].transpose(2, 0, 1))[None], size=(8, 16), mode="bilinear", align_corners=False)
    assert float(((back - x0).abs() / x0).max()) <= 1e-4
    ds.double_resolution()
    assert ds[0][0].shape == (3, 16, 32)
    assert [t.size for t in tf.transforms if isinstance(t, Resize)] == [(16, 32)]
    with pytest.raises(NotImplementedError, match="network"):
        RENIDatasetHDR(str(d), tf, True)


def test_ldr_dataset(tmp_path):
    from PIL import Image
    d = tmp_path / "Test"
    d.mkdir()
    rng = np.random.RandomState(0)
    a = rng.randint(0, 256, size=(8, 16, 4)).astype(np.uint8)
    Image.fromarray(a, "RGBA").save(str(d / "im1.png"))
    tf = transform_builder([["resize", [8, 16]], ["normalize", [[0.5, 0.5, 0.5], [0.5, 0.5, 0.5]]]])
    ds = get_dataset("CUSTOM", str(d), tf, False)
    assert isinstance(ds, RENIDatasetLDR) and len(ds) == 1
    img, idx = ds[0]
    assert img.shape == (3, 8, 16) and idx == 0
    want = (torch.from_numpy(a[:, :, :3].transpose(2, 0, 1).copy()).float() / 255 - 0.5) / 0.5
    assert torch.allclose(img, want, atol=1e-6)
    assert torch.allclose(ds.unnormalise(img[None].clone()), want[None] * 0.5 + 0.5, atol=1e-6)


# ------------------------------------------------------------------ seeded shape fuzz of the codecs
@pytest.mark.parametrize("seed", range(12))
def test_codec_fuzz_round_trips(tmp_path, seed):
    """Random sizes (1 x 1 up to a ragged last block of every codec), channel sets, pixel types and value patterns through
    the writer and back, and PIZ blocks (test-side encoder) of random small images -- bit-exact."""
    rng = np.random.RandomState(100 + seed)
    ny, nx = int(rng.randint(1, 70)), int(rng.randint(1, 50))
    nc = int(rng.choice([1, 3, 4]))
    kind = rng.choice(["smooth", "noise", "const", "steps"])
    yy, xx = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    base = {"smooth": np.sin(xx / 7.0) * np.cos(yy / 5.0) * 10, "noise": rng.randn(ny, nx) * 100,
            "const": np.full((ny, nx), 0.375), "steps": (xx // 5 + yy // 3).astype(np.float64)}[kind]
    img = np.stack([base * (c + 1) for c in range(nc)], -1).astype(np.float32)
    for pixel_type in ("half", "float"):
        for compression in ("none", "rle", "zips", "zip"):
            p = str(tmp_path / f"f_{pixel_type}_{compression}.exr")
            exr.write_exr(p, img, pixel_type=pixel_type, compression=compression)
            got = exr.read_exr(p)
            want = img.astype(np.float16).astype(np.float32) if pixel_type == "half" else img
            want = want[:, :, 0] if nc == 1 else want
            assert got.shape == want.shape and np.array_equal(got, want), (seed, pixel_type, compression, kind)
    # PIZ: two half channels, 32-line blocks
    q = (np.round(base * 4) / 4).astype(np.float16)
    planes_all = [q, (q * 0.5).astype(np.float16)]
    blocks = []
    for r0 in range(0, ny, 32):
        rows = [pl[r0:r0 + 32] for pl in planes_all]
        blk = _piz_block([(np.ascontiguousarray(r).view(np.uint16).reshape(r.shape[0], nx), 1) for r in rows])
        raw = b"".join(np.ascontiguousarray(r[y]).tobytes() for y in range(rows[0].shape[0]) for r in rows)
        blocks.append((r0, blk if len(blk) < len(raw) else raw))
    p = tmp_path / "f_piz.exr"
    p.write_bytes(_assemble(nx, ny, [("A", 1), ("Z", 1)], 4, blocks))
    planes, _ = exr.read_exr_channels(str(p))
    assert np.array_equal(planes["A"], planes_all[0].astype(np.float32)) and np.array_equal(planes["Z"], planes_all[1].astype(np.float32))
