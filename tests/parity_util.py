"""Helpers shared by the parity tests: structural + numeric comparison of two edge-point sets."""
import numpy as np


def compare_edgepoints(ref, got, rel_tol=1e-4):
    """Returns a dict report. Structure (counts, keys, view/polyline/segment ids, order) must be
    exact; X within rel_tol relative (BASELINE.json north_star: 1e-4 relative on 3-D coordinates,
    exact view-id sets); obs coordinates compared bit-exactly and by max abs diff."""
    rep = {"ok": True, "msgs": []}

    def fail(m):
        rep["ok"] = False
        rep["msgs"].append(m)

    if ref["n_points"] != got["n_points"] or ref["n_obs"] != got["n_obs"]:
        fail("count mismatch: ref points/obs %d/%d got %d/%d" % (ref["n_points"], ref["n_obs"], got["n_points"], got["n_obs"]))
        # report first differing key
        n = min(ref["n_points"], got["n_points"])
        if n:
            d = np.nonzero((ref["key"][:n] != got["key"][:n]).any(axis=1))[0]
            if len(d):
                fail("first key diff at point %d: ref %s got %s" % (d[0], ref["key"][d[0]], got["key"][d[0]]))
        return rep
    for name in ("key", "obs_off", "obs_view", "obs_pl", "obs_seg"):
        if not np.array_equal(ref[name], got[name]):
            d = np.nonzero(np.asarray(ref[name]).reshape(len(ref[name]), -1) != np.asarray(got[name]).reshape(len(got[name]), -1))[0]
            fail("%s differs at %d entries (first %d)" % (name, len(d), d[0]))
    n = ref["n_points"]
    if n:
        nx = np.linalg.norm(ref["X"].astype(np.float64), axis=1)
        dx = np.linalg.norm(ref["X"].astype(np.float64) - got["X"].astype(np.float64), axis=1)
        rel = dx / np.maximum(nx, 1e-12)
        rep["max_rel_X"] = float(rel.max())
        rep["bitexact_X"] = bool(np.array_equal(ref["X"].view(np.uint32), got["X"].view(np.uint32)))
        if not (rel.max() <= rel_tol):
            fail("X relative error %.3e > %.1e" % (rel.max(), rel_tol))
        rep["max_abs_xy"] = float(np.abs(ref["obs_xy"].astype(np.float64) - got["obs_xy"]).max()) if ref["n_obs"] else 0.0
        rep["bitexact_xy"] = bool(np.array_equal(ref["obs_xy"].view(np.uint32), got["obs_xy"].view(np.uint32)))
        if rep["max_abs_xy"] > 1e-3:
            fail("observation coordinates differ by %.3e px" % rep["max_abs_xy"])
    else:
        rep.update(max_rel_X=0.0, bitexact_X=True, max_abs_xy=0.0, bitexact_xy=True)
    return rep
