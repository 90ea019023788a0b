"""CPU: state-machine semantics of the restated OccupancyGrid that decide which rows come out
(SURVEY.md 8(a) "must reproduce" column and Appendix A)."""
import numpy as np
import pytest

import scenes

BBOX = (-0.1, 0.1, -0.1, 0.1, 0.0, 0.2)
RES = 0.005


def _grid(oracle_mod, **kw):
    return oracle_mod.OracleGrid(resolution=RES, bbox=BBOX, **kw)


def _plane_points(g, z_cell, xs, ys, jitter=0.0, rng=None):
    """One point per cell (x,y,z_cell) for x in xs, y in ys, at the cell centre (+ optional jitter)."""
    idx = np.array([[x, y, z_cell] for x in xs for y in ys], np.int32)
    c = g.probe_center(idx).astype(np.float32)
    if jitter and rng is not None:
        c = (c + rng.uniform(-jitter, jitter, size=c.shape)).astype(np.float32)
    return c


def test_empty_and_no_normals(oracle_mod):
    g = _grid(oracle_mod)
    g.clean()
    assert len(g.extract()) == 0
    g.add_points(np.zeros((0, 3), np.float32))
    assert g.is_dirty()  # addPoints sets state_changed even for an empty cloud (grid.hpp:189)
    g.add_points(_plane_points(g, 20, range(10, 14), range(10, 14)))  # 16 cells: never > 20 neighbours
    g.clean()
    assert not g.is_dirty()
    assert len(g.extract()) == 0
    assert g.counters()["occupied"] == 16


def test_plane_gets_normals_and_self_registration(oracle_mod):
    g = _grid(oracle_mod)
    pts = _plane_points(g, 20, range(5, 30), range(5, 30))
    g.add_points(pts, viewpoint=(0, 0, 0))
    g.clean()
    rows = g.extract()
    # interior cells see 25 occupied neighbours (> 20); the outermost ring sees 15 or 9..20
    assert len(rows) > 21 * 21 - 1
    inner = rows[(rows["ix"] == 15) & (rows["iy"] == 15)][0]
    assert inner["iz"] == 20 and abs(abs(inner["nz"]) - 1) < 1e-3
    assert inner["nz"] < 0  # oriented toward the viewpoint at the origin (grid.hpp:393-396)
    # the voxel's own buffered point lies on its own line -> i=0 self registration replays it (count 1)
    assert inner["count"] == 1
    deps = g.dependants(15, 15, 20)
    assert [15, 15, 20] in deps.tolist()
    # the line walk registers this voxel on the unoccupied cells above/below (grid.hpp:443-449)
    assert g.dependants(15, 15, 21).tolist() == [[15, 15, 20]]
    assert g.dependants(15, 15, 23).tolist() == [[15, 15, 20]]
    assert len(g.dependants(15, 15, 24)) == 0


def test_points_after_normal_found_are_not_buffered_but_update_dependants(oracle_mod):
    g = _grid(oracle_mod)
    pts = _plane_points(g, 20, range(5, 30), range(5, 30))
    g.add_points(pts)
    g.clean()
    before = g.counters()["buffered"]
    c0 = g.extract()
    g.add_points(pts)  # same points again: voxels have normals now
    after = g.counters()["buffered"]
    assert after - before == len(pts) - len(c0)  # only the normal-less border cells still buffer (grid.hpp:210-216)
    g.clean()
    c1 = g.extract()
    sel0 = c0[(c0["ix"] == 15) & (c0["iy"] == 15)][0]
    sel1 = c1[(c1["ix"] == 15) & (c1["iy"] == 15)][0]
    assert sel1["count"] == sel0["count"] + 1  # grid.hpp:244-277: later points update the dependant's running mean


def test_viewpoint_latched_at_first_occupancy(oracle_mod):
    g = _grid(oracle_mod)
    pts = _plane_points(g, 20, range(5, 30), range(5, 30))
    g.add_points(pts, viewpoint=(0, 0, 1.0))   # first toucher is above the plane
    g.add_points(pts, viewpoint=(0, 0, -1.0))  # later frame from below must not change the latch (grid.hpp:229)
    g.clean()
    rows = g.extract()
    assert (rows["nz"] > 0).all()


def test_unoccupied_cell_registration_last_wins_and_survives_occupation(oracle_mod):
    g = _grid(oracle_mod)
    # two parallel planes two cells apart: cell z=21 between them is registered by both (z=20 and z=22 voxels)
    pa = _plane_points(g, 20, range(5, 30), range(5, 30))
    pb = _plane_points(g, 22, range(5, 30), range(5, 30))
    g.add_points(np.vstack([pa, pb]))
    g.clean()
    d = g.dependants(15, 15, 21)
    # canonical ascending key order: (15,15,20) registers first, (15,15,22) overwrites (grid.hpp:443-449)
    assert d.tolist() == [[15, 15, 22]]
    # a point now lands in the middle cell: it becomes occupied, keeps the pre-attached dependant (grid.hpp:234-241)
    mid = g.probe_center(np.array([[15, 15, 21]], np.int32))
    rows0 = g.extract()
    g.add_points(mid)
    rows1 = g.extract()
    r0 = rows0[(rows0["ix"] == 15) & (rows0["iy"] == 15) & (rows0["iz"] == 22)][0]
    r1 = rows1[(rows1["ix"] == 15) & (rows1["iy"] == 15) & (rows1["iz"] == 22)][0]
    assert r1["count"] == r0["count"] + 1
    q0 = rows0[(rows0["ix"] == 15) & (rows0["iy"] == 15) & (rows0["iz"] == 20)][0]
    q1 = rows1[(rows1["ix"] == 15) & (rows1["iy"] == 15) & (rows1["iz"] == 20)][0]
    assert q1["count"] == q0["count"]  # the overwritten registration is lost, as in the reference


def test_rows_are_lexicographic_and_exclude_cell_dim(oracle_mod, synth_mod):
    sc = scenes.Scene(3, 160, 120, 0.005, clean_every=0)
    g = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    rows = scenes.run(g, sc, "capture")
    key = rows["ix"].astype(np.int64) * (1 << 42) + rows["iy"].astype(np.int64) * (1 << 21) + rows["iz"]
    assert (np.diff(key) > 0).all()
    (xd, yd, zd), _ = g.dims
    assert (rows["ix"] < xd).all() and (rows["iy"] < yd).all() and (rows["iz"] < zd).all()


def test_count_zero_rows_emit_zero_centroid(oracle_mod):
    g = _grid(oracle_mod)
    rng = np.random.default_rng(0)
    # points far (>1 mm) from every cell centre line: normals exist, no sample falls in any cylinder
    idx = np.array([[x, y, 20] for x in range(5, 30) for y in range(5, 30)], np.int32)
    c = g.probe_center(idx)
    pts = (c + np.array([0.0021, 0.0021, 0.0], np.float32)).astype(np.float32)
    g.add_points(pts)
    g.clean()
    rows = g.extract()
    z = rows[rows["count"] == 0]
    assert len(z) > 100
    assert (z["x"] == 0).all() and (z["y"] == 0).all() and (z["z"] == 0).all()  # grid.hpp:472-476


def test_first_row_rule_of_the_decoder(oracle_mod, synth_mod):
    """node.cpp:185,190 sizes and loops by row_step: an organised 160x120 message yields 160 points, the
    height=1,width=19200 publication yields all of them.  The harness applies the rule via n_points."""
    sc = scenes.Scene(1, 160, 120, 0.005)
    buf = sc.frame(0)
    g_row = oracle_mod.OracleGrid(resolution=0.005, bbox=sc.bbox)
    g_row.capture(buf, sc.poses[0], n_points=160)  # row_step / point_step of the organised message
    g_all = oracle_mod.OracleGrid(resolution=0.005, bbox=sc.bbox)
    g_all.capture(buf, sc.poses[0])
    assert g_row.counters()["presented"] == 160 and g_all.counters()["presented"] == 160 * 120


def test_welford_mean_matches_exact_mean(oracle_mod):
    g = _grid(oracle_mod)
    rng = np.random.default_rng(4)
    pts = _plane_points(g, 20, range(5, 30), range(5, 30))
    g.add_points(pts)
    g.clean()
    c = g.probe_center(np.array([[15, 15, 20]], np.int32))[0]
    extra = (c + rng.normal(scale=0.0004, size=(500, 3))).astype(np.float32)
    g.add_points(extra)
    row = [r for r in g.extract() if (r["ix"], r["iy"], r["iz"]) == (15, 15, 20)][0]
    # exact: members = points within 1 mm of the line through c along the row's normal; mean of their projections
    n = np.array([row["nx"], row["ny"], row["nz"]], np.float64)
    allp = np.vstack([pts[(np.abs(pts - c) < 1e-9).all(1)], extra]).astype(np.float64)
    t = (allp - c) @ n
    proj = c + t[:, None] * n
    dist = np.linalg.norm(allp - proj, axis=1)
    mem = dist < 0.001
    assert row["count"] == mem.sum()
    assert np.allclose([row["x"], row["y"], row["z"]], proj[mem].mean(0), atol=2e-6)
    assert np.isclose(row["mean_dist"], dist[mem].mean(), rtol=1e-4)
    assert np.allclose([row["sdx"], row["sdy"], row["sdz"]], proj[mem].var(0), rtol=2e-2, atol=1e-12)


def test_libstdcpp_order_mode_changes_only_rounding_and_overwrites(oracle_mod, synth_mod):
    """order_mode=1 walks a real std::unordered_set keyed like the reference (grid.hpp:151-156,315).  The set of
    emitted voxels and their normals cannot depend on the order (SURVEY hard part 2); counts may differ only
    where an unoccupied-cell registration was overwritten in a different order."""
    sc = scenes.Scene(5, 160, 120, 0.005, clean_every=2)
    ga = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox, order_mode=0)
    gb = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox, order_mode=1)
    ra = scenes.run(ga, sc, "capture")
    rb = scenes.run(gb, sc, "capture")
    assert len(ra) == len(rb)
    for f in ("ix", "iy", "iz", "nx", "ny", "nz"):
        assert np.array_equal(ra[f], rb[f])
    differ = int(np.sum(ra["count"] != rb["count"]))
    print("rows whose count depends on the clean order: %d of %d" % (differ, len(ra)))
    assert differ <= 0.02 * len(ra)
    same = ra["count"] == rb["count"]
    assert np.allclose(ra["x"][same], rb["x"][same], atol=1e-5)


def test_all_cores_timing_variant_computes_the_same_map(oracle_mod, synth_mod):
    """capture_mt/clean_mt is bench.py's all-cores CPU baseline, not a checker.  It still has to do the same work:
    the emitted voxel set, every points-in-cylinder count and every normal are order-independent and must be
    identical to the serial oracle; the float Welford fields may differ in the last bits (arrival order)."""
    sc = scenes.Scene(6, 160, 120, 0.005, clean_every=2)
    ga = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    ra = scenes.run(ga, sc, "capture")
    gb = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    for ev in sc.schedule():
        if ev[0] == "integrate":
            gb.capture_mt(sc.frame(ev[1]), sc.poses[ev[1]])
        else:
            gb.clean_mt()
    rb = gb.extract_mt()
    assert gb.normals_mt() == len(rb)
    assert len(ra) > 1000
    scenes.compare_rows(ra, rb)
    gb.clear()
    assert gb.normals_mt() == 0


def test_dense_storage_and_reserve_give_the_same_rows(oracle_mod, synth_mod):
    """The reference's storage -- one 16-byte Voxel per cell of the (dim+1)^3 box and buffer.reserve(1000) per new voxel
    (grid.hpp:626,228) -- against the hash map of touched cells the oracle uses by default: same rows, same occupancy.
    (bench.py times the dense + reserve variant as the faithful CPU baseline where the host has the memory.)"""
    import scenes
    bbox = (-0.3, 0.3, -0.3, 0.3, 0.2, 0.7)  # 600 x 600 x 500 cells at 1 mm: 2.9 GB of virtual Voxel array, touched sparsely
    sc = scenes.Scene(4, 160, 120, 0.001, bbox=bbox, fx=615.0, clean_every=2)
    a = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=bbox)
    b = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=bbox, dense=True, reserve=1000)
    assert b.is_dense and not a.is_dense
    ra, rb = scenes.run(a, sc, "capture"), scenes.run(b, sc, "capture")
    assert len(ra) > 1000 and ra.tobytes() == rb.tobytes()
    assert np.array_equal(a.occupied(), b.occupied())
    assert a.counters() == b.counters()
    b.clear()
    assert len(b.extract()) == 0 and b.counters()["occupied"] == 0
