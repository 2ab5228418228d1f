"""Minimal PNG decoder for the tests (python zlib + numpy): an independent check of the product's own
reader (edgegraph3d_amd/host/png_read.cpp). Returns the edge mask as cv::imread(IMREAD_COLOR) ==
EDGE_COLOR (255,255,255) would see it: uint8 [h, w], 1 = white."""
import struct
import zlib

import numpy as np


def read_png_edge_mask(path):
    d = open(path, "rb").read()
    assert d[:8] == b"\x89PNG\r\n\x1a\n"
    i, idat, plte, ihdr = 8, b"", None, None
    while i < len(d):
        n, t = struct.unpack(">I4s", d[i:i + 8])
        body = d[i + 8:i + 8 + n]
        if t == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif t == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif t == b"IDAT":
            idat += body
        i += 12 + n
    w, h, depth, ctype, _, _, interlace = ihdr
    assert interlace == 0
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp_bits = channels * depth
    stride = (w * bpp_bits + 7) // 8
    bpp = max(1, bpp_bits // 8)
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for r in range(h):
        f, line = int(raw[r, 0]), raw[r, 1:].astype(np.int32)
        cur = np.zeros(stride, np.int32)
        if f == 0:
            cur = line.copy()
        elif f == 2:
            cur = (line + prev) & 255
        else:
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                c = prev[x - bpp] if x >= bpp else 0
                if f == 1:
                    p = a
                elif f == 3:
                    p = (a + b) // 2
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + p) & 255
        out[r] = cur
        prev = cur
    if depth < 8:
        bits = np.unpackbits(out, axis=1)[:, :w * depth * channels]
        vals = bits.reshape(h, w * channels, depth)
        samples = np.zeros((h, w * channels), np.int32)
        for k in range(depth):
            samples = samples * 2 + vals[:, :, k]
        maxv = (1 << depth) - 1
    elif depth == 8:
        samples, maxv = out.astype(np.int32), 255
    else:
        samples, maxv = out.reshape(h, -1, 2)[:, :, 0].astype(np.int32), 255   # 16 bit: high byte, as OpenCV's 8-bit load
    samples = samples.reshape(h, w, channels)
    if ctype == 3:
        rgb = plte[samples[:, :, 0]]
        return (rgb == 255).all(axis=2).astype(np.uint8)
    if ctype in (0, 4):
        return (samples[:, :, 0] == maxv).astype(np.uint8)
    return (samples[:, :, :3] == maxv).all(axis=2).astype(np.uint8)


def write_png_gray(path, mask):
    """mask: [H, W] uint8 (0 / non-zero) -> an 8-bit grey PNG (0 / 255), no filtering; enough for the edge-image tests."""
    import struct
    import zlib

    import numpy as np
    m = (np.asarray(mask) != 0).astype(np.uint8) * 255
    h, w = m.shape
    raw = b"".join(b"\x00" + m[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
