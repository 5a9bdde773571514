"""OpenEXR scan-line reader / writer in numpy (host I/O either side of the hot path: SURVEY.md section 8, row f3).

The reference loads its HDR environment maps with ``imageio.imread(path)`` (src/data/datasets.py:73-79), i.e. a
float32 ``[H, W, C]`` array with channels in R, G, B(, A) order; ``read_exr`` returns exactly that.  imageio / OpenEXR
are not installed on either box, so the format is restated here from the published OpenEXR file layout ("OpenEXR
File Layout", openexr.com): magic 20000630, version 2, attribute list, line-offset table, chunks of 1 / 16 scan lines.

Supported: single-part scan-line images; pixel types UINT / HALF / FLOAT; compression NONE, RLE, ZIPS, ZIP (the zlib
family: byte-interleave + delta predictor + deflate) and PIZ (wavelet + Huffman, decode only).  Tiled, deep and multi-part
files and the lossy codecs (PXR24, B44, DWA) raise ``NotImplementedError`` naming what was found.
``write_exr`` (NONE / RLE / ZIPS / ZIP, half or float) is used to save predictions and to make the test fixtures.

EXR reader: PARITY UNPINNED.  No EXR file and no EXR library exists in the build image, so this module is checked only
against hand-assembled known-answer files in tests/test_exr_cpu.py and round trips through its own writer -- not against a
file the reference's own loader (imageio) has read.  It is host I/O in front of the hot path and is not extended further.
"""
import struct
import zlib

import numpy as np

MAGIC = 20000630
NONE, RLE, ZIPS, ZIP, PIZ, PXR24, B44, B44A, DWAA, DWAB = range(10)
COMPRESSION_NAMES = ("NONE", "RLE", "ZIPS", "ZIP", "PIZ", "PXR24", "B44", "B44A", "DWAA", "DWAB")
LINES_PER_BLOCK = {NONE: 1, RLE: 1, ZIPS: 1, ZIP: 16, PIZ: 32, PXR24: 16, B44: 32, B44A: 32, DWAA: 32, DWAB: 256}
UINT, HALF, FLOAT = 0, 1, 2
_DTYPES = {UINT: np.dtype("<u4"), HALF: np.dtype("<f2"), FLOAT: np.dtype("<f4")}


# ---------------------------------------------------------------------------------------------
# header
# ---------------------------------------------------------------------------------------------
def _cstr(buf, pos):
    end = buf.index(b"\0", pos)
    return buf[pos:end].decode("latin-1"), end + 1


def parse_header(buf):
    """-> (attributes {name: (type, raw bytes)}, offset of the line-offset table)"""
    magic, version = struct.unpack_from("<ii", buf, 0)
    if magic != MAGIC:
        raise ValueError("not an OpenEXR file (bad magic number)")
    if (version & 0xff) != 2:
        raise NotImplementedError(f"OpenEXR format version {version & 0xff} (only 2 is defined)")
    for bit, what in ((0x200, "tiled"), (0x800, "deep-data"), (0x1000, "multi-part")):
        if version & bit:
            raise NotImplementedError(f"{what} OpenEXR files are not supported (single-part scan-line images only)")
    pos, attrs = 8, {}
    while buf[pos] != 0:
        name, pos = _cstr(buf, pos)
        typ, pos = _cstr(buf, pos)
        (size,) = struct.unpack_from("<i", buf, pos)
        pos += 4
        attrs[name] = (typ, bytes(buf[pos:pos + size]))
        pos += size
    return attrs, pos + 1


def parse_channels(raw):
    """chlist attribute -> [(name, pixel_type, x_sampling, y_sampling)] in file (alphabetical) order"""
    out, pos = [], 0
    while raw[pos] != 0:
        name, pos = _cstr(raw, pos)
        ptype, _plinear, xs, ys = struct.unpack_from("<iB3xii", raw, pos)
        pos += 16
        if ptype not in _DTYPES:
            raise ValueError(f"channel {name!r}: unknown pixel type {ptype}")
        out.append((name, ptype, xs, ys))
    return out


# ---------------------------------------------------------------------------------------------
# the zlib family: predictor + byte interleave (ImfZip.cpp), run-length coding (ImfRle.cpp)
# ---------------------------------------------------------------------------------------------
def _unpredict_uninterleave(t):
    x = t.astype(np.int64)
    x[1:] -= 128
    d = (np.cumsum(x) & 0xff).astype(np.uint8)  # d[i] = d[i-1] + t[i] - 128  (mod 256)
    n = d.size
    out = np.empty(n, np.uint8)
    out[0::2] = d[:(n + 1) // 2]
    out[1::2] = d[(n + 1) // 2:]
    return out


def _interleave_predict(raw):
    n = raw.size
    t = np.concatenate([raw[0::2], raw[1::2]]).astype(np.int64)
    p = t.copy()
    p[1:] = (t[1:] - t[:-1] + 128 + 256) & 0xff
    return p.astype(np.uint8)


def _rle_decode(src, n_out):
    out = np.empty(n_out, np.uint8)
    i, o, n = 0, 0, len(src)
    while i < n:
        c = src[i] - 256 if src[i] > 127 else src[i]
        i += 1
        if c < 0:  # -c literal bytes
            out[o:o - c] = np.frombuffer(src, np.uint8, -c, i)
            i -= c
            o -= c
        else:      # the next byte, c + 1 times
            out[o:o + c + 1] = src[i]
            i += 1
            o += c + 1
    if o != n_out:
        raise ValueError("RLE block decodes to the wrong length")
    return out


def _rle_encode(raw):
    b, out, i, n = raw.tobytes(), bytearray(), 0, raw.size
    while i < n:
        j = i + 1
        while j < n and b[j] == b[i] and j - i < 128:
            j += 1
        if j - i >= 3:  # a run
            out += bytes([j - i - 1, b[i]])
            i = j
            continue
        j = i  # literals up to the next run of three
        while j < n and j - i < 127 and not (j + 2 < n and b[j] == b[j + 1] == b[j + 2]):
            j += 1
        out += bytes([(256 - (j - i)) & 0xff]) + b[i:j]
        i = j
    return bytes(out)


# ---------------------------------------------------------------------------------------------
# PIZ (ImfPizCompressor.cpp, ImfHuf.cpp, ImfWav.cpp): decode only
# ---------------------------------------------------------------------------------------------
_HUF_ENCBITS, _HUF_DECBITS = 16, 14
_HUF_ENCSIZE = (1 << _HUF_ENCBITS) + 1
_SHORT_ZEROCODE_RUN, _LONG_ZEROCODE_RUN = 59, 63
_SHORTEST_LONG_RUN = 2 + _LONG_ZEROCODE_RUN - _SHORT_ZEROCODE_RUN


class _Bits:
    """MSB-first bit reader over bytes"""

    def __init__(self, data, pos=0):
        self.d, self.p, self.c, self.lc = data, pos, 0, 0

    def get(self, n):
        while self.lc < n:
            self.c = (self.c << 8) | (self.d[self.p] if self.p < len(self.d) else 0)
            self.p += 1
            self.lc += 8
        self.lc -= n
        return (self.c >> self.lc) & ((1 << n) - 1)


def _huf_decode(data, n_out):
    if len(data) < 20:
        raise ValueError("PIZ: truncated Huffman header")
    im, iM, _tbl, n_bits, _res = struct.unpack_from("<IIIII", data, 0)
    if im >= _HUF_ENCSIZE or iM >= _HUF_ENCSIZE:
        raise ValueError("PIZ: bad Huffman table bounds")
    # packed code lengths, 6 bits each, zero runs run-length coded (hufUnpackEncTable)
    length = np.zeros(_HUF_ENCSIZE, np.int64)
    br = _Bits(data, 20)
    i = im
    while i <= iM:
        l = br.get(6)
        if l == _LONG_ZEROCODE_RUN:
            i += br.get(8) + _SHORTEST_LONG_RUN
        elif l >= _SHORT_ZEROCODE_RUN:
            i += l - _SHORT_ZEROCODE_RUN + 2
        else:
            length[i] = l
            i += 1
    table_end = br.p - (br.lc // 8)  # the bit stream starts at the next byte boundary after the table
    # canonical codes (hufCanonicalCodeTable): within a length ascending symbol order; shorter codes have larger values
    count = np.bincount(length, minlength=59)
    start = np.zeros(60, np.int64)
    c = 0
    for l in range(58, 0, -1):
        nc = (c + count[l]) >> 1
        start[l] = c
        c = nc
    code = {}
    nxt = start.copy()
    for s in np.nonzero(length)[0]:
        l = int(length[s])
        code[(l, int(nxt[l]))] = int(s)
        nxt[l] += 1
    rlc = iM  # the run-length symbol
    # codes of up to 14 bits through a 2^14-entry table on the next 14 bits (entry = length << 17 | symbol), longer ones by search
    fast = np.zeros(1 << _HUF_DECBITS, np.int64)
    for (l, cd), s in code.items():
        if l <= _HUF_DECBITS:
            lo = cd << (_HUF_DECBITS - l)
            fast[lo:lo + (1 << (_HUF_DECBITS - l))] = (l << 17) | s
    fast = fast.tolist()
    out = [0] * n_out
    pos, nd = table_end, len(data)
    o, c, lc, used = 0, 0, 0, 0
    while o < n_out and used < n_bits:
        while lc < 64:
            c = (c << 8) | (data[pos] if pos < nd else 0)
            pos += 1
            lc += 8
        e = fast[(c >> (lc - _HUF_DECBITS)) & 0x3fff]
        if e:
            l, s = e >> 17, e & 0x1ffff
        else:
            for l in range(_HUF_DECBITS + 1, 59):
                s = code.get((l, (c >> (lc - l)) & ((1 << l) - 1)))
                if s is not None:
                    break
            else:
                raise ValueError("PIZ: invalid Huffman code")
        lc -= l
        used += l
        if s == rlc:
            lc -= 8
            used += 8
            run = (c >> lc) & 0xff
            if o == 0 or o + run > n_out:
                raise ValueError("PIZ: bad run in the Huffman stream")
            out[o:o + run] = [out[o - 1]] * run
            o += run
        else:
            out[o] = s
            o += 1
        c &= (1 << lc) - 1
    out = np.array(out, np.uint16)
    if o != n_out:
        raise ValueError("PIZ: Huffman stream decodes to the wrong length")
    return out


def _wdec14(l, h):
    ls = l.astype(np.int16).astype(np.int32)
    hs = h.astype(np.int16).astype(np.int32)
    ai = ls + (hs & 1) + (hs >> 1)
    return ai.astype(np.int16).astype(np.uint16), (ai - hs).astype(np.int16).astype(np.uint16)


def _wdec16(l, h):
    m, d = l.astype(np.int32), h.astype(np.int32)
    bb = (m - (d >> 1)) & 0xffff
    aa = (d + bb - 0x8000) & 0xffff
    return aa.astype(np.uint16), bb.astype(np.uint16)


def _wav2_decode(a, mx):
    """in place 2-D inverse wavelet of a [ny, nx] uint16 array (wav2Decode), one level per pass, vectorised per level"""
    dec = _wdec14 if mx < (1 << 14) else _wdec16
    ny, nx = a.shape
    n = min(nx, ny)
    p = 1
    while p <= n:
        p <<= 1
    p >>= 1
    p2 = p
    p >>= 1
    while p >= 1:
        ys = np.arange(0, ny - p2 + 1, p2) if ny - p2 >= 0 else np.arange(0)
        xs = np.arange(0, nx - p2 + 1, p2) if nx - p2 >= 0 else np.arange(0)
        if ys.size and xs.size:
            Y, X = np.meshgrid(ys, xs, indexing="ij")
            i00, i10 = dec(a[Y, X], a[Y + p, X])
            i01, i11 = dec(a[Y, X + p], a[Y + p, X + p])
            a[Y, X], a[Y, X + p] = dec(i00, i01)
            a[Y + p, X], a[Y + p, X + p] = dec(i10, i11)
        if (nx & p) and ys.size:  # a last column without a right-hand partner
            x = xs[-1] + p2 if xs.size else 0
            a[ys, x], a[ys + p, x] = dec(a[ys, x], a[ys + p, x])
        if ny & p:                # a last row without a partner below
            y = ys[-1] + p2 if ys.size else 0
            if xs.size:
                a[y, xs], a[y, xs + p] = dec(a[y, xs], a[y, xs + p])
        p2 = p
        p >>= 1


def _piz_decode(data, chans, nx, ny):
    """-> the block's bytes in scan-line order, as the uncompressed layout stores them"""
    lo, hi = struct.unpack_from("<HH", data, 0)
    pos = 4
    bitmap = np.zeros(8192, np.uint8)
    if lo <= hi:
        bitmap[lo:hi + 1] = np.frombuffer(data, np.uint8, hi - lo + 1, pos)
        pos += hi - lo + 1
    present = np.unpackbits(bitmap, bitorder="little").astype(bool)
    present[0] = True  # zero is always in the table (reverseLutFromBitmap)
    lut = np.zeros(65536, np.uint16)
    vals = np.nonzero(present)[0]
    lut[:vals.size] = vals
    max_value = vals.size - 1
    (hlen,) = struct.unpack_from("<i", data, pos)
    pos += 4
    words = [nx * ny * (_DTYPES[pt].itemsize // 2) for _, pt, _, _ in chans]
    tmp = _huf_decode(bytes(data[pos:pos + hlen]), sum(words))
    out = np.empty((ny, sum(w // ny for w in words)), np.uint16)
    o, col = 0, 0
    for (_, pt, _, _), w in zip(chans, words):
        size = _DTYPES[pt].itemsize // 2
        blk = tmp[o:o + w].reshape(ny, nx, size)
        for k in range(size):  # 32-bit types: two interleaved 16-bit planes, each its own wavelet
            plane = np.ascontiguousarray(blk[:, :, k])
            _wav2_decode(plane, max_value)
            blk[:, :, k] = plane
        out[:, col:col + nx * size] = lut[blk.reshape(ny, nx * size)]
        o += w
        col += nx * size
    return out.astype("<u2").tobytes()


# ---------------------------------------------------------------------------------------------
# reading
# ---------------------------------------------------------------------------------------------
def read_exr_channels(path):
    """-> ({channel name: float32 / uint32 array [H, W]}, attributes)"""
    with open(path, "rb") as f:
        buf = f.read()
    attrs, pos = parse_header(buf)
    for need in ("channels", "compression", "dataWindow", "lineOrder"):
        if need not in attrs:
            raise ValueError(f"OpenEXR header lacks the required attribute {need!r}")
    chans = parse_channels(attrs["channels"][1])
    comp = attrs["compression"][1][0]
    if comp not in (NONE, RLE, ZIPS, ZIP, PIZ):
        name = COMPRESSION_NAMES[comp] if comp < len(COMPRESSION_NAMES) else str(comp)
        raise NotImplementedError(f"OpenEXR compression {name} is not supported (NONE, RLE, ZIPS, ZIP, PIZ are)")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    nx, ny = x1 - x0 + 1, y1 - y0 + 1
    if nx < 1 or ny < 1:
        raise ValueError("empty data window")
    if any(xs != 1 or ys != 1 for _, _, xs, ys in chans):
        raise NotImplementedError("sub-sampled channels are not supported")
    lpb = LINES_PER_BLOCK[comp]
    nblk = (ny + lpb - 1) // lpb
    if pos + 8 * nblk > len(buf):
        raise ValueError("truncated file (line-offset table)")
    offsets = struct.unpack_from(f"<{nblk}Q", buf, pos)
    line_bytes = sum(_DTYPES[pt].itemsize for _, pt, _, _ in chans) * nx
    planes = {name: np.empty((ny, nx), _DTYPES[pt]) for name, pt, _, _ in chans}
    seen = np.zeros(ny, bool)
    for off in offsets:
        if off + 8 > len(buf):
            raise ValueError("truncated file (chunk header outside the file)")
        y, size = struct.unpack_from("<ii", buf, off)
        data = memoryview(buf)[off + 8:off + 8 + size]
        r0 = y - y0
        if r0 < 0 or r0 >= ny or len(data) != size:
            raise ValueError("corrupt chunk (scan line outside the data window, or truncated file)")
        rows = min(lpb, ny - r0)
        want = rows * line_bytes
        if comp == PIZ and size != want:
            raw = np.frombuffer(_piz_decode(data, chans, nx, rows), np.uint8)
        elif comp == NONE or size == want:  # (the codecs store a block raw when it does not shrink)
            raw = np.frombuffer(data, np.uint8)
        else:
            if comp == RLE:
                t = _rle_decode(bytes(data), want)
            else:
                t = np.frombuffer(zlib.decompress(bytes(data)), np.uint8)
            if t.size != want:
                raise ValueError("compressed block inflates to the wrong size")
            raw = _unpredict_uninterleave(t)
        raw = raw.reshape(rows, line_bytes)
        col = 0
        for name, pt, _, _ in chans:
            w = _DTYPES[pt].itemsize * nx
            planes[name][r0:r0 + rows] = np.ascontiguousarray(raw[:, col:col + w]).view(_DTYPES[pt])
            col += w
        seen[r0:r0 + rows] = True
    if not seen.all():
        raise ValueError("the file does not cover every scan line of its data window")
    return {k: (v if v.dtype.kind == "u" else v.astype(np.float32)) for k, v in planes.items()}, attrs


def read_exr(path):
    """float32 [H, W, C] with channels R, G, B(, A) -- what ``imageio.imread`` hands the reference (datasets.py:73-79);
    a luminance-only file gives [H, W]; other channel sets come in file (alphabetical) order."""
    planes, _ = read_exr_channels(path)
    names = list(planes)
    if {"R", "G", "B"} <= set(names):
        order = ["R", "G", "B"] + (["A"] if "A" in names else [])
    elif names == ["Y"]:
        return planes["Y"].astype(np.float32)
    else:
        order = names
    return np.stack([planes[c].astype(np.float32) for c in order], axis=-1)


# ---------------------------------------------------------------------------------------------
# writing
# ---------------------------------------------------------------------------------------------
def _attr(name, typ, value):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(value)) + value


def write_exr(path, img, channels=None, pixel_type="half", compression="zip"):
    """img [H, W, C] (or [H, W]) -> single-part scan-line OpenEXR file.  channels: names of the last axis ("RGB", "RGBA",
    "Y" by default); pixel_type "half" | "float"; compression "none" | "rle" | "zips" | "zip"."""
    img = np.asarray(img)
    if img.ndim == 2:
        img = img[:, :, None]
    ny, nx, nc = img.shape
    if channels is None:
        channels = {1: ["Y"], 3: ["R", "G", "B"], 4: ["R", "G", "B", "A"]}[nc]
    channels = list(channels)
    if len(channels) != nc or len(set(channels)) != nc:
        raise ValueError("one distinct channel name per plane")
    pt = {"half": HALF, "float": FLOAT}[pixel_type]
    comp = {"none": NONE, "rle": RLE, "zips": ZIPS, "zip": ZIP}[compression]
    order = sorted(range(nc), key=lambda i: channels[i])  # channels are stored in alphabetical order
    chl = b"".join(channels[i].encode() + b"\0" + struct.pack("<iB3xii", pt, 0, 1, 1) for i in order) + b"\0"
    box = struct.pack("<4i", 0, 0, nx - 1, ny - 1)
    head = struct.pack("<ii", MAGIC, 2)
    head += _attr("channels", "chlist", chl)
    head += _attr("compression", "compression", bytes([comp]))
    head += _attr("dataWindow", "box2i", box)
    head += _attr("displayWindow", "box2i", box)
    head += _attr("lineOrder", "lineOrder", b"\0")
    head += _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0))
    head += _attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    head += b"\0"
    lpb = LINES_PER_BLOCK[comp]
    planes = [np.ascontiguousarray(img[:, :, i]).astype(_DTYPES[pt]) for i in order]
    chunks = []
    for r0 in range(0, ny, lpb):
        rows = min(lpb, ny - r0)
        raw = np.concatenate([p[r0:r0 + rows].view(np.uint8).reshape(rows, -1) for p in planes], axis=1).reshape(-1)
        data = raw.tobytes()
        if comp != NONE:
            t = _interleave_predict(raw)
            packed = _rle_encode(t) if comp == RLE else zlib.compress(t.tobytes(), 6)
            if len(packed) < len(data):
                data = packed
        chunks.append(struct.pack("<ii", r0, len(data)) + data)
    pos = len(head) + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", pos)
        pos += len(c)
    with open(path, "wb") as f:
        f.write(head + table + b"".join(chunks))
