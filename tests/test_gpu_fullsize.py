"""GPU: BASELINE.json-sized inputs.  Where the oracle is too slow for a full comparison the checks are
size-independent properties of the domain: conservation checksums, sortedness, run-to-run bit reproducibility and
sharding invariance; one full-resolution frame per big config is still compared with the oracle."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def _props(rows, ctr, dims):
    key = rows["ix"].astype(np.int64) * (1 << 42) + rows["iy"].astype(np.int64) * (1 << 21) + rows["iz"]  # any order-preserving packing
    assert (np.diff(key) > 0).all(), "rows not in strict lexicographic (x,y,z) order"
    assert (rows["ix"] < dims[0]).all() and (rows["iy"] < dims[1]).all() and (rows["iz"] < dims[2]).all()
    assert ctr["points_zclip_pass"] <= ctr["points_presented"] and ctr["points_in_bbox"] <= ctr["points_zclip_pass"]
    assert ctr["points_buffered"] <= ctr["points_in_bbox"]
    assert ctr["dep_pairs_member"] <= ctr["dep_pairs_tested"]
    # checksum of checksums: every cylinder member was counted exactly once, at integrate time or at replay time.
    # (records of cells with an index == dim exist but are not emitted, so >=)
    assert ctr["dep_pairs_member"] + ctr["replay_members"] >= int(rows["count"].astype(np.int64).sum())
    nn = np.sqrt(rows["nx"].astype(np.float64) ** 2 + rows["ny"].astype(np.float64) ** 2 + rows["nz"].astype(np.float64) ** 2)
    assert np.allclose(nn, 1.0, atol=1e-5)
    has = rows["count"] > 0
    assert (rows["mean_dist"][has] < 0.001).all() and (rows["mean_dist"][has] >= 0).all()  # members are inside the 1 mm cylinder


def _stream(hfpf_mod, sc, batch, caps, world=1):
    """Runs the scene through integrate_device in batches; returns rows, counters, dims."""
    import hfpf_dist
    fb = sc.W * sc.H * 16
    grids = [hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **caps) for _ in range(world)]
    vr = hfpf_dist.LocalVirtualRanks(grids)
    devs = [g.device_alloc(batch * fb) for g in grids]
    try:
        pending = [[] for _ in range(world)]

        def flush():
            for r, g in enumerate(grids):
                if pending[r]:
                    for j, f in enumerate(pending[r]):
                        g.device_upload(devs[r] + j * fb, sc.frame(f))
                    g.integrate_device(devs[r], len(pending[r]), fb, sc.W * sc.H, np.stack([sc.poses[f] for f in pending[r]]),
                                       frame_ids=np.array(pending[r], np.uint32))
                    g.sync()
                    pending[r] = []
        for ev in sc.schedule():
            if ev[0] == "integrate":
                r = ev[1] % world
                pending[r].append(ev[1])
                if len(pending[r]) == batch:
                    flush()
            else:
                flush()
                vr.clean_all() if world > 1 else grids[0].clean()
        rows = vr.extract() if world > 1 else grids[0].extract()
        ctrs = [g.counters() for g in grids]
        dims = grids[0].dims[0]
    finally:
        for g, d in zip(grids, devs):
            g.device_free(d)
            g.close()
    tot = {k: sum(c[k] for c in ctrs) for k in ctrs[0]}
    return rows, tot, dims


def test_config1_stream_properties_and_reproducibility(hfpf_mod, synth_mod):
    """configs[1] shape: 640x480 frames, random SE(3) poses, 1 m^3 @ 1 mm (36 frames, clean every 12)."""
    sc = scenes.Scene(36, 640, 480, 0.001, clean_every=12)
    caps = dict(max_bricks=200000, max_log_points=24 << 20, max_normals=6 << 20, max_frames=1024)
    rows, ctr, dims = _stream(hfpf_mod, sc, 12, caps)
    assert ctr["points_presented"] == 36 * 640 * 480
    assert len(rows) > 500000
    _props(rows, ctr, dims)
    rows2, ctr2, _ = _stream(hfpf_mod, sc, 6, caps)  # different batching, same schedule
    assert rows.tobytes() == rows2.tobytes(), "not bit-reproducible across runs / batch sizes"
    rows4, _, _ = _stream(hfpf_mod, sc, 12, dict(caps, frame_width=640))  # 16x16-pixel tiles (bench.py's setting)
    assert rows.tobytes() == rows4.tobytes(), "frame_width hint changed the result"
    rows3, ctr3, _ = _stream(hfpf_mod, sc, 6, caps, world=3)  # frames dealt to 3 virtual ranks
    assert rows.tobytes() == rows3.tobytes(), "sharded run differs from the single-GPU run"
    assert ctr3["points_presented"] == ctr["points_presented"] and ctr3["dep_pairs_member"] + ctr3["replay_members"] == \
        ctr["dep_pairs_member"] + ctr["replay_members"]


def test_config3_full_resolution_frame_vs_oracle(oracle_mod, hfpf_mod, synth_mod):
    """configs[2] shape: one 2048x1536 frame (3.1 M points), 2 m^3 bbox @ 0.5 mm (3999x1999x1999 cells)."""
    bbox = (-1.0, 1.0, -0.5, 0.5, 0.0, 1.0)
    sc = scenes.Scene(1, 2048, 1536, 0.0005, bbox=bbox, clean_every=0)
    caps = dict(max_bricks=400000, max_log_points=8 << 20, max_normals=4 << 20, max_frames=64)
    rows, ctr, dims = _stream(hfpf_mod, sc, 1, caps)
    assert dims == (3999, 1999, 1999)
    _props(rows, ctr, dims)
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=bbox)
    ref = scenes.run(og, sc, "capture")
    oc = og.counters()
    assert (oc["presented"], oc["zclip_pass"], oc["inserted"], oc["buffered"]) == (
        ctr["points_presented"], ctr["points_zclip_pass"], ctr["points_in_bbox"], ctr["points_buffered"])
    scenes.compare_rows(ref, rows)


def test_config5_grid_ten_cubic_metres(oracle_mod, hfpf_mod, synth_mod):
    """configs[4] grid: 10 m^3 @ 1 mm = 2499 x 1999 x 1999 (~1e10 cells; 78 MB brick directory)."""
    bbox = (-1.25, 1.25, -1.0, 1.0, 0.0, 2.0)
    sc = scenes.Scene(4, 320, 240, 0.001, bbox=bbox, fx=615.0, clean_every=2)
    caps = dict(max_bricks=100000, max_log_points=4 << 20, max_normals=1 << 20, max_frames=64)
    rows, ctr, dims = _stream(hfpf_mod, sc, 2, caps)
    assert dims == (2499, 1999, 1999)
    _props(rows, ctr, dims)
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=bbox)
    ref = scenes.run(og, sc, "capture")
    scenes.compare_rows(ref, rows)


@pytest.mark.slow
def test_config1_full_1000_frames(hfpf_mod, synth_mod):
    """configs[1] at its full size: the 1000-frame 640x480 stream of bench.py (same seeds), clean every 150 frames + final
    clean = 7 passes, dependant-table relocation at scale.  Too slow for the oracle, so: conservation checksums, sortedness,
    and byte reproducibility between one integrate call per clean epoch (150 frames, bench.py's default) and 50 frames per call."""
    sc = scenes.Scene(1000, 640, 480, 0.001, clean_every=150)
    caps = dict(max_bricks=100000, max_log_points=96 << 20, max_normals=6 << 20, max_frames=2048, frame_width=640)
    rows, ctr, dims = _stream(hfpf_mod, sc, 150, caps)
    assert dims == (999, 999, 999)
    assert ctr["points_presented"] == 1000 * 640 * 480 and ctr["frames_integrated"] == 1000 and ctr["clean_passes"] == 7
    assert ctr["dep_pairs_tested"] > 5e8 and len(rows) > 1500000
    _props(rows, ctr, dims)
    rows2, ctr2, _ = _stream(hfpf_mod, sc, 50, caps)
    assert rows.tobytes() == rows2.tobytes(), "1000-frame stream not byte-reproducible across frames_per_call 150 vs 50"
    for k in ("points_zclip_pass", "points_in_bbox", "points_buffered", "dep_pairs_tested", "dep_pairs_member", "replay_members",
              "voxels_occupied", "voxels_with_normal", "registrations"):
        assert ctr[k] == ctr2[k], k


@pytest.mark.slow
def test_bench_cadence_stream_vs_oracle_and_overflowing_bins(oracle_mod, hfpf_mod, synth_mod, monkeypatch):
    """The bench's cadence against the ORACLE: 120 frames of 640x480 (36.9 M points), 60 frames per integrate
    call, clean every 60 frames, 16x16-pixel tiles -- 10^3..10^4 parked points per brick, several sort rounds per brick in
    k_update_cells, the dry run of the first call, dependant-table relocation in the second and third clean pass.  The oracle
    runs the same stream serially on the host (~30 s).  Second engine run: bin regions planned at a third of their demand
    (HFPF_TEST_BIN_SCALE x the plan's slack of 2), so that most points of every brick go through the overflow list and take the
    direct forms in k_integrate_overflow (direct log append + chained entry, one lane per (point, dependant) pair, wave-cooperative
    atomic flush per member pair) beside the binned ones -- same rows, byte for byte."""
    sc = scenes.Scene(120, 640, 480, 0.001, clean_every=60)
    caps = dict(max_bricks=100000, max_log_points=48 << 20, max_normals=6 << 20, max_frames=256, frame_width=640)
    rows, ctr, dims = _stream(hfpf_mod, sc, 60, caps)
    assert ctr["clean_passes"] == 2 and ctr["dep_pairs_tested"] > 2e7 and len(rows) > 500000
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    ref = scenes.run(og, sc, "capture")
    oc = og.counters()
    og.close()
    assert (oc["presented"], oc["zclip_pass"], oc["inserted"], oc["buffered"]) == (
        ctr["points_presented"], ctr["points_zclip_pass"], ctr["points_in_bbox"], ctr["points_buffered"])
    scenes.compare_rows(ref, rows)
    monkeypatch.setenv("HFPF_TEST_BIN_SCALE", "0.16")
    rows2, ctr2, _ = _stream(hfpf_mod, sc, 60, caps)
    scenes.compare_rows(ref, rows2)
    assert rows.tobytes() == rows2.tobytes(), "overflowing bin regions (direct forms) changed the result"
    for k in ("points_buffered", "dep_pairs_tested", "dep_pairs_member", "replay_members", "voxels_with_normal"):
        assert ctr[k] == ctr2[k], k


@pytest.mark.parametrize("color", [False, True])
def test_update_forms_cell_sorted_and_per_point_give_the_same_bits(hfpf_mod, synth_mod, monkeypatch, color):
    """k_update_cells (points counting-sorted by cell in LDS, several rounds of 1024 points per brick at this size; default) against
    k_update (one lane per point, HFPF_UPDATE_FORM=points): 120 full-resolution frames in calls of 60, three clean passes --
    rows and pair counters must be identical."""
    sc = scenes.Scene(120, 640, 480, 0.001, clean_every=60)
    caps = dict(max_bricks=100000, max_log_points=48 << 20, max_normals=6 << 20, max_frames=256, frame_width=640, fuse_color=color)
    rows, ctr, _ = _stream(hfpf_mod, sc, 60, caps)
    if color:
        assert len(np.unique(rows["rgb"])) > 1000  # the colour sums went through the sorted LDS copy (s_rgb) and the table
    monkeypatch.setenv("HFPF_UPDATE_FORM", "points")
    rows2, ctr2, _ = _stream(hfpf_mod, sc, 60, caps)
    assert len(rows) > 500000 and ctr["dep_pairs_tested"] > 2e7
    assert rows.tobytes() == rows2.tobytes()
    for k in ("dep_pairs_tested", "dep_pairs_member", "replay_members", "points_buffered", "voxels_with_normal"):
        assert ctr[k] == ctr2[k], k


class _TwoEpochScene(scenes.Scene):
    """`first` frames, a clean pass, the remaining frames, a final clean pass."""

    def __init__(self, first, *a, **kw):
        super().__init__(*a, **kw)
        self.first = first

    def schedule(self):
        for f in range(self.n_frames):
            yield ("integrate", f)
            if f + 1 == self.first:
                yield ("clean",)
        yield ("clean",)


@pytest.mark.slow
def test_config3_update_shapes_and_colour_vs_oracle_at_scale(oracle_mod, hfpf_mod, synth_mod, monkeypatch):
    """configs[2]'s production kernels against the ORACLE at its resolution: 2048x1536 frames into the 2 m^3 box @ 0.5 mm
    -- two frames, a clean pass, then four frames in ONE integrate call, so that the dependant update of that call sees
    (point, dependant) pairs by the ten million, 200-300 records on a brick and bricks with more parked points than one LDS round
    of the wide shape holds (`update_extra_rounds`).  Five engine runs: the adaptive shape choice, the dense shape forced, the wide
    shape forced (HFPF_UPD_SHAPE=1: k_update_cells<false,512,1536,512,1024,6>), and colour fusion under both shapes (the 1024-point
    colour instantiations) -- each compared with the oracle (which fuses colour when asked), the shapes with each other byte for
    byte.  Oracle ~20 s for the 18.9 M points."""
    bbox = (-1.0, 1.0, -0.5, 0.5, 0.0, 1.0)
    sc = _TwoEpochScene(2, 6, 2048, 1536, 0.0005, bbox=bbox)
    caps = dict(max_bricks=400000, max_log_points=24 << 20, max_normals=6 << 20, max_frames=64, frame_width=2048)
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=bbox, fuse_color=True)
    ref_c = scenes.run(og, sc, "capture", color=True)
    oc = og.counters()
    og.close()
    ref = ref_c.copy()
    ref["rgb"] = 0  # the reference's own behaviour: colour never fused (geometry and counts do not depend on the extension)
    runs = {}
    for name, shape, color in (("adaptive", None, False), ("dense", "0", False), ("wide", "1", False), ("dense+colour", "0", True),
                               ("wide+colour", "1", True)):
        if shape is None:
            monkeypatch.delenv("HFPF_UPD_SHAPE", raising=False)
        else:
            monkeypatch.setenv("HFPF_UPD_SHAPE", shape)
        rows, ctr, dims = _stream(hfpf_mod, sc, 4, dict(caps, fuse_color=color))
        assert dims == (3999, 1999, 1999)
        assert (oc["presented"], oc["zclip_pass"], oc["inserted"], oc["buffered"]) == (
            ctr["points_presented"], ctr["points_zclip_pass"], ctr["points_in_bbox"], ctr["points_buffered"]), name
        assert ctr["dep_pairs_tested"] > 1e7, (name, ctr["dep_pairs_tested"])
        assert ctr["update_extra_rounds"] > 0, "%s: no brick took more than one round" % name
        scenes.compare_rows(ref_c if color else ref, rows)
        runs[name] = (rows, ctr)
    assert len(ref) > 1000000
    for a, b in (("adaptive", "dense"), ("adaptive", "wide"), ("dense+colour", "wide+colour")):
        assert runs[a][0].tobytes() == runs[b][0].tobytes(), "%s and %s differ" % (a, b)
        for k in ("dep_pairs_tested", "dep_pairs_member", "replay_members", "voxels_with_normal"):
            assert runs[a][1][k] == runs[b][1][k], (a, b, k)
    assert len(np.unique(runs["wide+colour"][0]["rgb"])) > 1000


def test_config4_shared_two_cubic_metre_grid_cameras_vs_oracle(oracle_mod, hfpf_mod, synth_mod):
    """configs[3] shape: one camera per rank (distinct frame and pose seeds), shared 2 m^3 bbox @ 1 mm (1999x999x999 cells), merge
    at every clean.  3 virtual ranks on one device against the oracle fed the union of the cameras' frames in canonical
    (frame, camera) order; small frames so the oracle finishes in seconds."""
    import hfpf_dist
    bbox = (-1.0, 1.0, -0.5, 0.5, 0.0, 1.0)
    world, n_frames, clean_every = 3, 4, 2
    cams = [scenes.Scene(n_frames, 320, 240, 0.001, bbox=bbox, fx=615.0 / 2, seed=0xF051 + 7919 * r, pose_seed=0x5E3 + 104729 * r)
            for r in range(world)]
    caps = dict(max_bricks=100000, max_log_points=4 << 20, max_normals=1 << 20, max_frames=64)
    og = oracle_mod.OracleGrid(resolution=0.001, bbox=bbox)
    grids = [hfpf_mod.OccupancyGrid(resolution=0.001, bbox=bbox, **caps) for _ in range(world)]
    vr = hfpf_dist.LocalVirtualRanks(grids)
    fb = 320 * 240 * 16
    devs = [g.device_alloc(fb) for g in grids]
    try:
        assert grids[0].dims[0] == (1999, 999, 999)
        for f in range(n_frames):
            for r in range(world):
                buf = cams[r].frame(f)
                og.capture(buf, cams[r].poses[f])
                grids[r].device_upload(devs[r], buf)
                grids[r].integrate_device(devs[r], 1, fb, 320 * 240, cams[r].poses[f][None], frame_ids=np.array([f * world + r], np.uint32))
                grids[r].sync()
            if (f + 1) % clean_every == 0 and f + 1 < n_frames:
                vr.clean_all()
                og.clean()
        vr.clean_all()
        og.clean()
        ref = og.extract()
        for on in range(world):  # every rank extracts the same merged cloud
            rows = vr.extract(on=on)
            scenes.compare_rows(ref, rows)
        assert len(ref) > 20000
        occ = grids[0].occupied()
        assert np.array_equal(og.occupied(), occ)
    finally:
        for g, d in zip(grids, devs):
            g.device_free(d)
            g.close()
