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


def chain_ranges(r):
    """dict: chain key (seed, entry, hit) -> (first point, one past the last) in the cloud r (emission order)."""
    k = np.asarray(r["key"]).reshape(-1, 4)
    if len(k) == 0:
        return {}
    brk = np.nonzero((k[1:, :3] != k[:-1, :3]).any(axis=1))[0] + 1
    starts = np.concatenate([[0], brk])
    ends = np.concatenate([brk, [len(k)]])
    return {tuple(int(x) for x in k[a, :3]): (int(a), int(b)) for a, b in zip(starts, ends)}


def compare_by_chain(ref, got, rel_tol=1e-4):
    """STRUCTURE-TOLERANT comparison of two clouds, for a caller who holds a run of the real reference: chains are
    matched by their key (seed, track entry, starting hit) instead of by position, so one chain that grew or lost a
    point does not make everything behind it "differ". Reports how many chains exist on both sides, how many of those
    are structurally identical (same points, same (view, polyline, segment) lists in the same order), and — over the
    points of the identical chains — how many 3-D coordinates are bit-equal / within rel_tol, the largest relative
    difference, and the same for the 2-D observation coordinates. compare_edgepoints() is the strict form the parity
    tests of this repository use (everything exact, in order)."""
    ca, cb = chain_ranges(ref), chain_ranges(got)
    common = sorted(set(ca) & set(cb))
    rep = {"chains_ref": len(ca), "chains_got": len(cb), "chains_in_both": len(common),
           "chains_only_ref": len(set(ca) - set(cb)), "chains_only_got": len(set(cb) - set(ca)),
           "chains_structurally_identical": 0, "points_ref": int(ref["n_points"]), "points_got": int(got["n_points"]),
           "points_compared": 0, "points_X_bit_equal": 0, "points_X_within_tol": 0, "max_rel_dX": 0.0,
           "obs_compared": 0, "obs_xy_bit_equal": 0, "max_abs_dxy": 0.0, "rel_tol": rel_tol}
    ao, bo = np.asarray(ref["obs_off"]).astype(np.int64), np.asarray(got["obs_off"]).astype(np.int64)
    for key in common:
        (a0, a1), (b0, b1) = ca[key], cb[key]
        if a1 - a0 != b1 - b0 or not np.array_equal(np.diff(ao[a0:a1 + 1]), np.diff(bo[b0:b1 + 1])):
            continue
        oa, ob_, n = ao[a0], bo[b0], ao[a1] - ao[a0]
        if not (np.array_equal(ref["obs_view"][oa:oa + n], got["obs_view"][ob_:ob_ + n]) and
                np.array_equal(ref["obs_pl"][oa:oa + n], got["obs_pl"][ob_:ob_ + n]) and
                np.array_equal(ref["obs_seg"][oa:oa + n], got["obs_seg"][ob_:ob_ + n])):
            continue
        rep["chains_structurally_identical"] += 1
        Xa, Xb = ref["X"][a0:a1], got["X"][b0:b1]
        rel = np.linalg.norm(Xa.astype(np.float64) - Xb.astype(np.float64), axis=1) / np.maximum(np.linalg.norm(Xa.astype(np.float64), axis=1), 1e-12)
        rep["points_compared"] += len(rel)
        rep["points_X_bit_equal"] += int((Xa.view(np.uint32) == Xb.view(np.uint32)).all(axis=1).sum())
        rep["points_X_within_tol"] += int((rel <= rel_tol).sum())
        if len(rel):
            rep["max_rel_dX"] = max(rep["max_rel_dX"], float(np.nanmax(rel)))
        xa, xb = ref["obs_xy"][oa:oa + n], got["obs_xy"][ob_:ob_ + n]
        rep["obs_compared"] += int(n)
        rep["obs_xy_bit_equal"] += int((xa.view(np.uint32) == xb.view(np.uint32)).all(axis=1).sum())
        if n:
            rep["max_abs_dxy"] = max(rep["max_abs_dxy"], float(np.nanmax(np.abs(xa.astype(np.float64) - xb))))
    nb = max(1, rep["chains_in_both"])
    rep["share_chains_identical"] = rep["chains_structurally_identical"] / nb
    rep["share_points_within_tol"] = rep["points_X_within_tol"] / max(1, rep["points_compared"])
    return rep
