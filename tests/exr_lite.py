"""Minimal OpenEXR reader for single-part scanline files with ZIP/ZIPS/NONE compression and FLOAT channels.

Enough for the float images the reference renderer ships (scenes/*/normal.exr, albedo.exr); written here because no
OpenEXR module is installed.  Returns a dict channel-name -> (H, W) float32 array.
"""
import struct
import zlib

import numpy as np


def read_exr(path):
    data = open(path, "rb").read()
    magic, version = struct.unpack_from("<II", data, 0)
    assert magic == 20000630, "not an EXR file"
    assert (version & 0x200) == 0, "tiled EXR not supported"
    pos = 8
    hdr = {}
    while data[pos] != 0:
        e = data.index(b"\0", pos)
        name = data[pos:e].decode()
        pos = e + 1
        e = data.index(b"\0", pos)
        typ = data[pos:e].decode()
        pos = e + 1
        (size,) = struct.unpack_from("<i", data, pos)
        pos += 4
        hdr[name] = (typ, data[pos:pos + size])
        pos += size
    pos += 1
    chans = []
    cb = hdr["channels"][1]
    p = 0
    while cb[p] != 0:
        e = cb.index(b"\0", p)
        nm = cb[p:e].decode()
        p = e + 1
        ptype, _plinear, xs, ys = struct.unpack_from("<iB3xii", cb, p)
        p += 16
        assert xs == 1 and ys == 1
        chans.append((nm, ptype))
    comp = hdr["compression"][1][0]
    xmin, ymin, xmax, ymax = struct.unpack("<4i", hdr["dataWindow"][1])
    W, H = xmax - xmin + 1, ymax - ymin + 1
    lines_per_block = {0: 1, 2: 1, 3: 16}[comp]
    nblocks = (H + lines_per_block - 1) // lines_per_block
    offsets = struct.unpack_from("<%dQ" % nblocks, data, pos)
    bpp = {0: 4, 1: 2, 2: 4}
    out = {nm: np.zeros((H, W), np.float32) for nm, _ in chans}
    for off in offsets:
        y, size = struct.unpack_from("<ii", data, off)
        raw = data[off + 8: off + 8 + size]
        nlines = min(lines_per_block, ymax - y + 1)
        expect = nlines * W * sum(bpp[t] for _, t in chans)
        if comp in (2, 3) and size < expect:
            d = np.frombuffer(zlib.decompress(raw), np.uint8).astype(np.int32)
            d = np.cumsum(np.concatenate([d[:1], d[1:] - 128])) & 0xFF  # predictor
            d = d.astype(np.uint8)
            half = (len(d) + 1) // 2
            r = np.empty(len(d), np.uint8)  # de-interleave
            r[0::2] = d[:half]
            r[1::2] = d[half:]
            raw = r.tobytes()
        p = 0
        for ln in range(nlines):
            for nm, t in chans:
                dt = {0: "<u4", 1: "<f2", 2: "<f4"}[t]
                row = np.frombuffer(raw, dt, W, p)
                p += W * bpp[t]
                out[nm][y - ymin + ln] = row.astype(np.float32)
    return out


def read_rgb(path):
    c = read_exr(path)
    return np.stack([c["R"], c["G"], c["B"]], -1)
