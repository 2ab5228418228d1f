"""Scenes built from EDGE IMAGES through the N2 builder (include/eg3d_host.h eg3d_plg_build_from_mask):

  real_edges_scene()      25 of the real dtu006 edge maps (tests/golden/dtu006_edges) with SYNTHETIC
                          look-at cameras placed at the 25 listed camera centres (DTU-like intrinsics:
                          f = 2890 px, pp = (823, 619)). The reference's input.json with the true poses
                          is missing, so the geometry is NOT consistent with the images: the scene
                          exercises the path on real polyline statistics (BASELINE configs[2] stand-in),
                          its output is not a reconstruction of anything.
  rendered_edges_scene()  a synthetic scene whose projected curves are rasterised into binary edge
                          images and sent through the same builder: consistent geometry end to end
                          (edge images -> polyline graphs -> path).
Both return (SceneArrays, SeedsArrays, info)."""
import ctypes as C
import glob
import os

import numpy as np

from edgegraph3d_amd import _cdefs as D
from edgegraph3d_amd import host

HERE = os.path.dirname(os.path.abspath(__file__))
EDGES = os.path.join(HERE, "golden", "dtu006_edges")
F_PX, PPX, PPY = 2890.0, 823.0, 619.0


def _sfm():
    L = host.lib()
    L.eg3d_sfm_create.restype = C.c_void_p
    L.eg3d_sfm_create.argtypes = [C.c_int, C.c_int, C.c_int]
    L.eg3d_sfm_set_camera.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, D.f32p, D.f32p, C.c_char_p]
    L.eg3d_sfm_analytic_F.argtypes = [C.c_void_p, D.f64p, D.u8p]
    L.eg3d_sfm_cam_P.argtypes = [C.c_void_p]
    L.eg3d_sfm_cam_P.restype = D.f32p
    L.eg3d_sfm_destroy.argtypes = [C.c_void_p]
    return L


def lookat_cameras(centres, target, width, height):
    """P [V,16] f32 and analytic F [V,V,9] f64 of look-at cameras (x right, y down, z forward)."""
    L = _sfm()
    V = len(centres)
    h = L.eg3d_sfm_create(V, width, height)
    for v, Cw in enumerate(centres):
        z = target - Cw
        z = z / np.linalg.norm(z)
        up = np.array([0.0, 0.0, 1.0]) if abs(z[2]) < 0.95 else np.array([0.0, 1.0, 0.0])
        x = np.cross(z, up)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.ascontiguousarray(np.stack([x, y, z]), np.float32).reshape(-1)
        Cc = np.ascontiguousarray(Cw, np.float32)
        assert L.eg3d_sfm_set_camera(h, v, F_PX, PPX, PPY, D.np_ptr(R, C.c_float), D.np_ptr(Cc, C.c_float), b"view.png") == 0
    P = D.as_np(L.eg3d_sfm_cam_P(h), V * 16, np.float32).reshape(V, 16).copy()
    F = np.zeros((V, V, 9), np.float64)
    Fv = np.zeros((V, V), np.uint8)
    assert L.eg3d_sfm_analytic_F(h, D.np_ptr(F, C.c_double), D.np_ptr(Fv, C.c_uint8)) == 0
    L.eg3d_sfm_destroy(h)
    return P, F, Fv


def scene_from_views(views, P, F, Fv, width, height):
    vpo, pvo, vtx, ps, pe, pv = [0], [0], [], [], [], []
    for g in views:
        base = pvo[-1]
        pvo.extend(int(base + o) for o in g["pl_vtx_off"][1:])
        vtx.append(g["vtx_xy"])
        ps.append(g["pl_start"])
        pe.append(g["pl_end"])
        pv.append(g["pl_valid"])
        vpo.append(vpo[-1] + g["n_polylines"])
    d = {"n_views": len(views), "width": width, "height": height, "cam_P": P, "F": F, "F_valid": Fv,
         "view_pl_off": np.array(vpo, np.uint32), "pl_vtx_off": np.array(pvo, np.uint32),
         "vtx_xy": np.concatenate(vtx).astype(np.float32) if vtx else np.zeros((1, 2), np.float32),
         "pl_start": np.concatenate(ps).astype(np.uint32), "pl_end": np.concatenate(pe).astype(np.uint32),
         "pl_valid": np.concatenate(pv).astype(np.uint8)}
    return host.SceneArrays(d)


def _project(P, X):
    M = P.reshape(-1, 4, 4).astype(np.float64)
    h = np.einsum("vij,j->vi", M[:, :3, :], np.append(X, 1.0))
    return h[:, :2] / h[:, 2:3], h[:, 2]


def seeds_on_polylines(views, P, n_seeds, rng, width, height, depth, max_track=8):
    """Seeds = a vertex of a valid polyline of some view, pushed to a 3-D point on its ray at `depth`
    (+-15 %), observed (projection + 0.4 px noise) in a random subset of the views that see it."""
    V = len(views)
    M = P.reshape(V, 4, 4).astype(np.float64)
    off, view, xy = [0], [], []
    tries = 0
    while len(off) - 1 < n_seeds and tries < n_seeds * 50:
        tries += 1
        a = int(rng.integers(V))
        g = views[a]
        valid = np.nonzero(g["pl_valid"])[0]
        if not len(valid):
            continue
        p = int(rng.choice(valid))
        k = int(rng.integers(g["pl_vtx_off"][p], g["pl_vtx_off"][p + 1]))
        u = g["vtx_xy"][k].astype(np.float64)
        # back-project through K [R|t]: X = C + s * R^T K^-1 (u,1)
        A, t = M[a][:3, :3], M[a][:3, 3]
        ray = np.linalg.solve(A, np.append(u, 1.0))
        Cw = -np.linalg.solve(A, t)
        X = Cw + ray / np.linalg.norm(ray) * depth * rng.uniform(0.85, 1.15)
        uv, z = _project(P, X)
        vis = [v for v in range(V) if z[v] > 0 and 5 < uv[v][0] < width - 5 and 5 < uv[v][1] < height - 5]
        if len(vis) < 3:
            continue
        kk = int(rng.integers(3, min(len(vis), max_track) + 1))
        sel = sorted(rng.choice(vis, kk, replace=False))
        for v in sel:
            view.append(int(v))
            xy.append(uv[v] + rng.normal(0, 0.4, 2))
        off.append(len(view))
    return host.SeedsArrays(np.array(off, np.uint32), np.array(view, np.int32), np.array(xy, np.float32).reshape(-1, 2))


def real_edges_scene(n_views=25, n_seeds=6268, rng_seed=20180606):
    """See the module docstring. 25 views = the 25 camera centres the reference lists (26 edge maps are
    shipped; the first 25 are used); n_seeds defaults to the 6268 reference points its README quotes."""
    files = sorted(glob.glob(os.path.join(EDGES, "*.png")))[:n_views]
    assert len(files) == n_views, "edge-map fixtures missing"
    views, width, height = [], 0, 0
    for f in files:
        m = host.png_edge_mask(f)
        height, width = m.shape
        views.append(host.plg_from_mask(m))
    centres = np.loadtxt(os.path.join(EDGES, "target_camera_poses.txt"))[:n_views]
    assert len(centres) == n_views
    target = np.zeros(3)   # the cameras look at the origin of the target coordinate system (an assumption)
    P, F, Fv = lookat_cameras(centres, target, width, height)
    rng = np.random.default_rng(rng_seed)
    depth = float(np.linalg.norm(centres, axis=1).mean())
    seeds = seeds_on_polylines(views, P, n_seeds, rng, width, height, depth)
    info = {"views": n_views, "segments_per_view": float(np.mean([(np.diff(g["pl_vtx_off"])[g["pl_valid"] == 1] - 1).sum() for g in views])),
            "polylines_per_view": float(np.mean([g["pl_valid"].sum() for g in views]))}
    return scene_from_views(views, P, F, Fv, width, height), seeds, info


def rasterise(vtx, off, valid, width, height):
    """Polylines -> binary edge image (integer DDA along every segment)."""
    m = np.zeros((height, width), np.uint8)
    for p in range(len(off) - 1):
        if not valid[p]:
            continue
        v = vtx[off[p]:off[p + 1]]
        for a, b in zip(v[:-1], v[1:]):
            n = int(max(abs(b[0] - a[0]), abs(b[1] - a[1]))) + 1
            xs = np.clip(np.round(np.linspace(a[0], b[0], n + 1) - 0.5).astype(int), 0, width - 1)
            ys = np.clip(np.round(np.linspace(a[1], b[1], n + 1) - 0.5).astype(int), 0, height - 1)
            m[ys, xs] = 1
    return m


def rendered_edges_scene(cfg=1):
    """Synthetic scene `cfg`: its polylines are drawn into edge images, the images rebuilt into polyline
    graphs by the N2 builder; cameras, F and seeds are the synthetic scene's own."""
    s = host.Synth(cfg)
    sc = s.scene_np()
    V, W, H = sc["n_views"], sc["width"], sc["height"]
    views = []
    for v in range(V):
        a, b = int(sc["view_pl_off"][v]), int(sc["view_pl_off"][v + 1])
        off = sc["pl_vtx_off"][a:b + 1].astype(np.int64)
        views.append(host.plg_from_mask(rasterise(sc["vtx_xy"], off, sc["pl_valid"][a:b], W, H)))
    off, view, xy = s.seeds_np()
    return scene_from_views(views, sc["cam_P"], sc["F"], sc["F_valid"], W, H), host.SeedsArrays(off, view, xy), {"views": V}
