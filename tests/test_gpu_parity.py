"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu

SMALL = dict(max_bricks=60000, max_log_points=4 << 20, max_normals=1 << 20, max_frames=4096)


def _both(oracle_mod, hfpf_mod, scene, **cfg):
    og = oracle_mod.OracleGrid(resolution=scene.resolution, bbox=scene.bbox, **cfg)
    ref = scenes.run(og, scene, "capture")
    occ_ref = og.occupied()
    with hfpf_mod.OccupancyGrid(resolution=scene.resolution, bbox=scene.bbox, **cfg, **SMALL) as eg:
        got = scenes.run(eg, scene, "integrate")
        occ_got = eg.occupied()
        ctr = eg.counters()
    oc = og.counters()
    og.close()
    return ref, got, occ_ref, occ_got, oc, ctr


@pytest.mark.parametrize("res,W,H,fx,nf,ce", [
    (0.005, 160, 120, 0.0, 6, 3),      # reference default 5 mm voxels
    (0.001, 160, 120, 615.0, 6, 3),    # 1 mm voxels, crop of the 640x480 sensor
    (0.001, 160, 120, 615.0, 5, 1),    # clean after every frame
    (0.0005, 128, 96, 1968.0, 4, 2),   # 0.5 mm voxels (config-3-like resolution)
])
def test_stream_parity(oracle_mod, hfpf_mod, synth_mod, res, W, H, fx, nf, ce):
    sc = scenes.Scene(nf, W, H, res, fx=fx, clean_every=ce)
    ref, got, occ_ref, occ_got, oc, ctr = _both(oracle_mod, hfpf_mod, sc)
    assert np.array_equal(occ_ref, occ_got), "voxel occupancy differs"
    assert oc["presented"] == ctr["points_presented"]
    assert oc["zclip_pass"] == ctr["points_zclip_pass"]
    assert oc["inserted"] == ctr["points_in_bbox"]
    assert oc["buffered"] == ctr["points_buffered"]
    assert len(ref) > 100
    scenes.compare_rows(ref, got)


def test_identity_single_frame_config0(oracle_mod, hfpf_mod, synth_mod):
    """BASELINE configs[0]: one 640x480 frame, identity pose, 1 m^3 bbox @ 1 mm."""
    sc = scenes.Scene(1, 640, 480, 0.001, identity=True)
    ref, got, occ_ref, occ_got, oc, ctr = _both(oracle_mod, hfpf_mod, sc)
    assert np.array_equal(occ_ref, occ_got)
    scenes.compare_rows(ref, got)


def test_full_resolution_frames_with_tile_hint_vs_oracle(oracle_mod, hfpf_mod, synth_mod):
    """Three 640x480 frames with random poses, clean after the second, frame_width = 640 (bench.py's setting): the 16x16-pixel
    tile path at the sensor's real size against the oracle."""
    sc = scenes.Scene(3, 640, 480, 0.001, clean_every=2)
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    ref = scenes.run(og, sc, "capture")
    occ_ref = og.occupied()
    og.close()
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, frame_width=640, **SMALL) as eg:
        got = scenes.run(eg, sc, "integrate")
        assert np.array_equal(occ_ref, eg.occupied())
    assert len(ref) > 100000
    scenes.compare_rows(ref, got)


def test_pcl32_layout(oracle_mod, hfpf_mod, synth_mod):
    """32-byte PCL-style records (x,y,z at 0,4,8; rgb at 16) through the generic decode path."""
    sc = scenes.Scene(3, 160, 120, 0.005, layout=synth_mod.LAYOUT_PCL32, clean_every=2)
    ref, got, occ_ref, occ_got, oc, ctr = _both(oracle_mod, hfpf_mod, sc)
    assert np.array_equal(occ_ref, occ_got)
    scenes.compare_rows(ref, got)


def test_compacting_rebuild_path(oracle_mod, hfpf_mod, synth_mod):
    """max_normals just above the need: the incremental dependant-table update runs out of room and the engine
    falls back to the full (compacting) rebuild; rows must not change."""
    sc = scenes.Scene(6, 160, 120, 0.001, fx=615.0, clean_every=1)
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    ref = scenes.run(og, sc, "capture")
    cap = dict(SMALL)
    cap["max_normals"] = int(len(ref) * 1.3) + 64
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **cap) as eg:
        got = scenes.run(eg, sc, "integrate")
    scenes.compare_rows(ref, got)


def test_capacity_overflow_is_reported(hfpf_mod, synth_mod):
    sc = scenes.Scene(2, 160, 120, 0.001, fx=615.0)
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, max_bricks=8, max_log_points=1 << 20, max_normals=1 << 16, max_frames=8) as eg:
        eg.integrate(sc.frame(0), sc.poses[0])
        with pytest.raises(hfpf_mod.HfpfError) as e:
            eg.clean()
        assert e.value.code == -3 and "brick" in str(e.value)


def test_colour_extension_does_not_change_geometry(oracle_mod, hfpf_mod, synth_mod):
    """HFPF_FLAG_FUSE_COLOR (extension; the reference never fuses colour): same rows, plus the member points' mean RGB,
    bit-exact against the oracle's definition of the extension (sum of the members' r, g, b over count, round half up)."""
    sc = scenes.Scene(5, 160, 120, 0.001, fx=615.0, clean_every=2)
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox, fuse_color=True)
    ref = scenes.run(og, sc, "capture", color=True)
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as a:
        ra = scenes.run(a, sc, "integrate")
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, fuse_color=True, **SMALL) as b:
        rb = scenes.run(b, sc, "integrate")
    assert (ra["rgb"] == 0).all()
    for f in ra.dtype.names:
        if f != "rgb":
            assert np.array_equal(ra[f], rb[f]), f
    assert (rb["rgb"][rb["count"] == 0] == 0).all()
    scenes.compare_rows(ref, rb)  # includes rgb, bit-exact
    has = rb["count"] > 0
    assert has.sum() > 1000 and len(np.unique(rb["rgb"][has])) > 1000  # real colours, not a constant
    # PCL32 layout (rgb at offset 16) reads the same colours
    sc32 = scenes.Scene(5, 160, 120, 0.001, fx=615.0, clean_every=2, layout=synth_mod.LAYOUT_PCL32)
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, fuse_color=True, **SMALL) as c:
        rc = scenes.run(c, sc32, "integrate")
    assert rc.tobytes() == rb.tobytes()


def test_clear_then_rerun_is_identical(hfpf_mod, synth_mod):
    sc = scenes.Scene(4, 160, 120, 0.005, clean_every=2)
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as g:
        r1 = scenes.run(g, sc, "integrate")
        g.clear()
        assert g.state_changed
        assert len(g.extract()) == 0
        r2 = scenes.run(g, sc, "integrate")
    assert r1.tobytes() == r2.tobytes()  # integer sums: bitwise reproducible run to run


def test_clear_after_a_larger_session_leaves_nothing_behind(hfpf_mod, synth_mod):
    """hfpf_clear resets the part of the tables the session used (bricks and records are handed out in sequence), not their
    capacity: a session that follows a LARGER one, and one that follows a SMALLER one and grows past its bricks, must both
    give what a fresh handle gives."""
    big = scenes.Scene(6, 160, 120, 0.001, fx=615.0, clean_every=3)
    small = scenes.Scene(3, 160, 120, 0.001, fx=615.0, seed=0xBEEF, pose_seed=0x77, clean_every=2, max_angle=5.0)
    fresh = {}
    for name, sc in (("big", big), ("small", small)):
        with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as g:
            fresh[name] = scenes.run(g, sc, "integrate").tobytes()
            assert len(fresh[name])
    with hfpf_mod.OccupancyGrid(resolution=big.resolution, bbox=big.bbox, **SMALL) as g:
        assert scenes.run(g, big, "integrate").tobytes() == fresh["big"]
        used_big = g.counters()["bricks_allocated"]
        g.clear()
        assert scenes.run(g, small, "integrate").tobytes() == fresh["small"]  # after the larger session
        used_small = g.counters()["bricks_allocated"]
        assert used_small < used_big
        g.clear()
        assert scenes.run(g, big, "integrate").tobytes() == fresh["big"]  # grows past the bricks the smaller session reset
        g.clear()
        assert len(g.extract()) == 0 and g.counters()["bricks_allocated"] == 0


def test_long_batches_take_the_larger_dry_run_vs_oracle(oracle_mod, hfpf_mod, synth_mod):
    """A session's first batch of 64 frames or more is planned from a dry run of 16 frames instead of 8 (csrc/hfpf.hip
    integrate_device_locked): two batches of 72 frames with a clean pass behind each, against the oracle."""
    sc = scenes.Scene(144, 96, 72, 0.001, fx=92.25, clean_every=72)
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    ref = scenes.run(og, sc, "capture")
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as g:
        fb = sc.W * sc.H * 16
        dev = g.device_alloc(sc.n_frames * fb)
        for f in range(sc.n_frames):
            g.device_upload(dev + f * fb, sc.frame(f))
        for e in range(2):
            g.integrate_device(dev + e * 72 * fb, 72, fb, sc.W * sc.H, np.stack(sc.poses[e * 72:(e + 1) * 72]))
            g.clean()
        got = g.extract()
        ctr = g.counters()
        g.device_free(dev)
    og.close()
    assert ctr["dep_pairs_tested"] > 0
    scenes.compare_rows(ref, got)


def test_batched_device_frames_equal_single_host_frames(hfpf_mod, synth_mod):
    """hfpf_integrate_device with several frames per launch == one hfpf_integrate per frame (order-free sums)."""
    sc = scenes.Scene(6, 160, 120, 0.001, fx=615.0)
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as a:
        for f in range(3):
            a.integrate(sc.frame(f), sc.poses[f])
        a.clean()
        for f in range(3, 6):
            a.integrate(sc.frame(f), sc.poses[f])
        a.clean()
        ra = a.extract()
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as b:
        fb = sc.W * sc.H * 16
        dev = b.device_alloc(6 * fb)
        for f in range(6):
            b.device_upload(dev + f * fb, sc.frame(f))
        b.integrate_device(dev, 3, fb, sc.W * sc.H, np.stack(sc.poses[:3]))
        b.clean()
        b.integrate_device(dev + 3 * fb, 3, fb, sc.W * sc.H, np.stack(sc.poses[3:]))
        b.clean()
        rb = b.extract()
        b.device_free(dev)
    assert ra.tobytes() == rb.tobytes()


@pytest.mark.parametrize("binned", [True, False])
def test_binned_and_direct_update_are_bit_identical(oracle_mod, hfpf_mod, synth_mod, binned):
    """The two forms of the dependant update (LDS-staged brick bins vs one atomic per pair) give the same bits."""
    sc = scenes.Scene(7, 160, 120, 0.001, fx=615.0, clean_every=2)
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    ref = scenes.run(og, sc, "capture")
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, binned_update=binned, **SMALL) as g:
        got = scenes.run(g, sc, "integrate")
        ctr = g.counters()
    scenes.compare_rows(ref, got)
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, binned_update=not binned, **SMALL) as g2:
        other = scenes.run(g2, sc, "integrate")
        ctr2 = g2.counters()
    assert got.tobytes() == other.tobytes()
    assert ctr["dep_pairs_tested"] == ctr2["dep_pairs_tested"] and ctr["dep_pairs_member"] == ctr2["dep_pairs_member"]


@pytest.mark.parametrize("layout_name", ["LAYOUT_PACKED16", "LAYOUT_PCL32"])
def test_frame_width_hint_changes_nothing(oracle_mod, hfpf_mod, synth_mod, layout_name):
    """hfpf_config.frame_width only re-tiles the integrate kernel (16x16-pixel patches, one 8x8 patch per wave): rows must
    be byte-identical with the hint, without it, and with a width that does not tile the frame (falls back to runs);
    and the hinted run must still match the oracle."""
    sc = scenes.Scene(5, 160, 128, 0.001, fx=615.0, clean_every=2, layout=getattr(synth_mod, layout_name))
    rows = {}
    for fw in (0, 160, 100, 48):  # 48: width is a multiple of 16 but 160*128/48 is not integral -> falls back
        with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, frame_width=fw, **SMALL) as eg:
            rows[fw] = scenes.run(eg, sc, "integrate")
    assert len(rows[0]) > 1000
    for fw in (160, 100, 48):
        assert rows[fw].tobytes() == rows[0].tobytes(), "frame_width=%d changed the result" % fw
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    scenes.compare_rows(scenes.run(og, sc, "capture"), rows[160])
    og.close()


def test_full_record_table_falls_back_to_device_atomics(oracle_mod, hfpf_mod, synth_mod, monkeypatch):
    """The per-brick LDS record table of the dependant update has 512 slots; a record that finds no slot is updated with device
    atomics instead.  A surface brick rarely sees that many records, so the fall-back is forced (HFPF_TEST_TABLE_SKIP=1: records
    with an odd id never get a slot) for both forms of the kernel: rows must match the oracle and the ordinary run bit for bit."""
    sc = scenes.Scene(7, 160, 120, 0.001, fx=615.0, clean_every=2)
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    ref = scenes.run(og, sc, "capture")
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as g:
        plain = scenes.run(g, sc, "integrate")
    monkeypatch.setenv("HFPF_TEST_TABLE_SKIP", "1")
    for form in ("cells", "points"):
        monkeypatch.setenv("HFPF_UPDATE_FORM", form)
        with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as g:
            got = scenes.run(g, sc, "integrate")
        scenes.compare_rows(ref, got)
        assert got.tobytes() == plain.tobytes(), form


@pytest.mark.parametrize("host_batch,copy_streams", [("4", "2"), ("1", "2"), ("4", "1"), ("8", "4")])
def test_host_frames_batched_over_several_copy_streams_equal_device_frames(hfpf_mod, synth_mod, monkeypatch, host_batch, copy_streams):
    """The host-frame path under load: 24 frames of 640x480 handed to hfpf_integrate back to back, so that frames arrive while the
    engine's stream is still busy and are launched in batches of up to HFPF_HOST_BATCH ring slots whose uploads alternate over
    HFPF_COPY_STREAMS copy streams -- the kernels must wait for the last upload of EVERY stream that carried a slot of the batch
    (round-3 advisor finding: only the last slot's event was waited for).  Same again from page-locked caller memory
    (hfpf_integrate_pinned).  Rows and counters byte for byte those of the device-resident path."""
    sc = scenes.Scene(24, 640, 480, 0.001, clean_every=12)
    caps = dict(max_bricks=100000, max_log_points=16 << 20, max_normals=4 << 20, max_frames=256, frame_width=640)
    keys = ("points_presented", "points_zclip_pass", "points_in_bbox", "points_buffered", "dep_pairs_tested", "dep_pairs_member",
            "replay_members", "voxels_occupied", "voxels_with_normal", "registrations")
    frames = [sc.frame(f) for f in range(sc.n_frames)]
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **caps) as b:
        fb = sc.W * sc.H * 16
        dev = b.device_alloc(12 * fb)
        for half in range(2):
            for j in range(12):
                b.device_upload(dev + j * fb, frames[12 * half + j])
            b.integrate_device(dev, 12, fb, sc.W * sc.H, np.stack(sc.poses[12 * half:12 * half + 12]))
            b.clean()
        ref = b.extract()
        ref_ctr = b.counters()
        b.device_free(dev)
    assert len(ref) > 300000 and ref_ctr["dep_pairs_tested"] > 1e6
    monkeypatch.setenv("HFPF_HOST_BATCH", host_batch)
    monkeypatch.setenv("HFPF_COPY_STREAMS", copy_streams)
    for pinned in (False, True):
        with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **caps) as a:
            bufs = frames
            if pinned:
                bufs = [a.host_alloc(sc.W * sc.H * 16) for _ in frames]
                for hb, fr in zip(bufs, frames):
                    hb[:] = np.frombuffer(fr, np.uint8)
            for half in range(2):
                for f in range(12 * half, 12 * half + 12):
                    if pinned:
                        a.integrate_pinned(bufs[f], sc.poses[f], n_points=sc.W * sc.H)
                    else:
                        a.integrate(bufs[f], sc.poses[f])
                a.clean()
            got = a.extract()
            ctr = a.counters()
            if pinned:
                for hb in bufs:
                    a.host_free(hb)
        assert got.tobytes() == ref.tobytes(), "host frames (batch %s, %s copy streams, pinned=%s) differ from device frames" % (host_batch, copy_streams, pinned)
        for k in keys:
            assert ctr[k] == ref_ctr[k], k


def _plane_frame(offset_along_normal, n=200, half=0.02, seed=1):
    """A tilted planar patch z = 0.4 + 0.5 x + 0.3 y (camera frame = fusion frame: identity pose), shifted along its unit normal,
    as one packed XYZRGB frame: n x n samples, jittered by a few micrometres so that no point sits on a voxel boundary."""
    rng = np.random.default_rng(seed)
    u = np.linspace(-half, half, n, dtype=np.float64)
    x, y = np.meshgrid(u, u, indexing="ij")
    nrm = np.array([-0.5, -0.3, 1.0]) / np.sqrt(0.25 + 0.09 + 1.0)
    pts = np.stack([x, y, 0.4 + 0.5 * x + 0.3 * y], -1).reshape(-1, 3) + offset_along_normal * nrm + rng.uniform(-3e-6, 3e-6, (n * n, 3))
    rec = np.zeros((n * n, 4), np.float32)
    rec[:, :3] = pts
    rec[:, 3] = np.frombuffer(np.full(n * n, 0x00808080, np.uint32).tobytes(), np.float32)
    return rec.tobytes()


def test_registration_contest_on_unoccupied_cells_follows_the_key_order(oracle_mod, hfpf_mod, synth_mod):
    """'The last registrant wins' on an unoccupied cell (OccupancyGrid.hpp:443-449) with record ids that follow the Z-order of a pass
    while the contest is about the canonical (x, y, z) order (round-3 advisor finding).  A tilted plane: the normals lean in x
    and y, so the +-3-step walks of neighbouring voxels -- which differ in x AND y, i.e. pairs whose Morton order and key order
    disagree are common -- end on the same unoccupied cells one to three cells off the surface.  A second frame then puts its points
    exactly there (the plane shifted 2.2 mm along its normal): every one of them updates the ONE registrant its cell kept, so the
    members-per-voxel counts of the first plane's voxels say who won each contest.  Compared with the oracle, which walks the
    candidates in ascending key order.  (A build in which the atomicMax over the Z-order ids alone settles the contest -- the key
    comparison compiled out -- fails this test: counts differ at hundreds of rows; checked once on the GPU box in round 4.)"""
    pose = synth_mod.identity_pose()
    frames = [_plane_frame(0.0), _plane_frame(0.0022, seed=2)]
    og = oracle_mod.OracleGrid(resolution=0.001, bbox=scenes.BBOX_1M)
    with hfpf_mod.OccupancyGrid(resolution=0.001, bbox=scenes.BBOX_1M, **SMALL) as eg:
        for fr in frames:
            og.capture(fr, pose)
            og.clean()
            eg.integrate(fr, pose)
            eg.clean()
        ref, got = og.extract(), eg.extract()
        ctr = eg.counters()
    og.close()
    assert len(ref) > 2000 and ctr["dep_pairs_member"] > 20000  # the second plane's points found registrants on their (formerly unoccupied) cells
    scenes.compare_rows(ref, got)
