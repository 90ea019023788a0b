"""GPU: edge cases of the hot path against the oracle: non-default parameters, degenerate inputs, capacity
errors, frame-id ordering, concurrent callers."""
import threading

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu
SMALL = dict(max_bricks=60000, max_log_points=4 << 20, max_normals=1 << 20, max_frames=4096)


def _pair(oracle_mod, hfpf_mod, sc, **cfg):
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox, **cfg)
    ref = scenes.run(og, sc, "capture")
    occ_ref = og.occupied()
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **cfg, **SMALL) as eg:
        got = scenes.run(eg, sc, "integrate")
        occ = eg.occupied()
    assert np.array_equal(occ_ref, occ)
    scenes.compare_rows(ref, got)
    return ref


@pytest.mark.parametrize("cfg", [
    dict(K=1), dict(K=5), dict(gate=10), dict(gate=40),
    dict(cylinder_radius=0.0005), dict(cylinder_radius=0.003, ball_radius=0.03),
    dict(z_clip=(0.35, 0.5)),
    dict(pcl_shifted_cov=True),  # plane fit as PCL >= 1.11 computes the covariance
])
def test_non_default_parameters(oracle_mod, hfpf_mod, synth_mod, cfg):
    sc = scenes.Scene(5, 160, 120, 0.001, fx=615.0, clean_every=2)
    ref = _pair(oracle_mod, hfpf_mod, sc, **cfg)
    assert len(ref) > 50


def test_coarse_voxels_many_points_per_cell(oracle_mod, hfpf_mod, synth_mod):
    """2 cm voxels: hundreds of points per cell, long buffers, most points outside the 1 mm cylinders."""
    sc = scenes.Scene(4, 160, 120, 0.02, clean_every=1)
    ref = _pair(oracle_mod, hfpf_mod, sc)
    assert (ref["count"] == 0).any()  # rows with an empty cylinder emit the zero centroid (grid.hpp:472-476)


def test_tiny_grid_clips_the_stencil(oracle_mod, hfpf_mod, synth_mod):
    """A bbox only 4 cells thick in z: validCoord clips the 5x5x5 stencil everywhere (grid.hpp:337)."""
    sc = scenes.Scene(3, 160, 120, 0.005, bbox=(-0.2, 0.2, -0.2, 0.2, 0.545, 0.566), clean_every=0)
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    assert og.dims[0][2] == 4
    _pair(oracle_mod, hfpf_mod, sc)


def test_empty_nan_and_fully_clipped_frames(oracle_mod, hfpf_mod, synth_mod):
    sc = scenes.Scene(3, 160, 120, 0.005)
    n = sc.W * sc.H
    nan_frame = np.full((n, 4), np.nan, np.float32)
    far_frame = np.zeros((n, 4), np.float32)
    far_frame[:, 2] = 2.0  # beyond the z-clip
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as g:
        g.integrate(np.zeros(0, np.uint8), sc.poses[0], n_points=0)
        assert g.state_changed
        g.integrate(nan_frame, sc.poses[0])
        g.integrate(far_frame, sc.poses[0])
        g.clean()
        assert len(g.extract()) == 0
        c = g.counters()
        assert c["points_presented"] == 2 * n and c["points_zclip_pass"] == 0 and c["voxels_occupied"] == 0
        for f in range(3):  # and the grid still works afterwards
            g.integrate(sc.frame(f), sc.poses[f])
        g.clean()
        got = g.extract()
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    for f in range(3):
        og.capture(sc.frame(f), sc.poses[f])
    og.clean()
    scenes.compare_rows(og.extract(), got)


def test_explicit_frame_ids_out_of_order(oracle_mod, hfpf_mod, synth_mod):
    """The viewpoint latch is the smallest frame id, whatever order the launches arrive in: integrating frames
    3,1,2,0 with their ids equals the oracle fed 0,1,2,3."""
    sc = scenes.Scene(4, 160, 120, 0.005)
    og = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    for f in range(4):
        og.capture(sc.frame(f), sc.poses[f])
    og.clean()
    fb = sc.W * sc.H * 16
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as g:
        dev = g.device_alloc(fb)
        for f in (3, 1, 2, 0):
            g.device_upload(dev, sc.frame(f))
            g.integrate_device(dev, 1, fb, sc.W * sc.H, sc.poses[f].reshape(1, 12), frame_ids=np.array([f], np.uint32))
            g.sync()
        g.clean()
        got = g.extract()
        g.device_free(dev)
    scenes.compare_rows(og.extract(), got)


def test_capacity_errors_are_reported_not_fatal(hfpf_mod, synth_mod):
    sc = scenes.Scene(2, 160, 120, 0.001, fx=615.0)
    # point log too small
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, max_bricks=60000, max_log_points=4096, max_normals=1 << 16, max_frames=8) as g:
        g.integrate(sc.frame(0), sc.poses[0])
        with pytest.raises(hfpf_mod.HfpfError) as e:
            g.clean()
        assert e.value.code == -3 and "point log" in str(e.value)
    # normal records too few
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, max_bricks=60000, max_log_points=1 << 20, max_normals=64, max_frames=8) as g:
        g.integrate(sc.frame(0), sc.poses[0])
        with pytest.raises(hfpf_mod.HfpfError) as e:
            g.clean()
        assert e.value.code == -3
    # frame id beyond the viewpoint table
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, max_bricks=60000, max_log_points=1 << 20, max_normals=1 << 16, max_frames=1) as g:
        g.integrate(sc.frame(0), sc.poses[0])
        with pytest.raises(hfpf_mod.HfpfError) as e:
            g.integrate(sc.frame(1), sc.poses[1])
        assert e.value.code == -3
    # misaligned record layout
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as g:
        with pytest.raises(hfpf_mod.HfpfError) as e:
            g.integrate(sc.frame(0), sc.poses[0], point_step=18, off_x=1, off_y=5, off_z=9, off_rgb=13)
        assert e.value.code == -2


def test_concurrent_callers_are_serialised_and_order_free(hfpf_mod, synth_mod):
    """Four host threads push disjoint frames into one handle (the reference funnels everything through grid_mtx_,
    node.cpp:291); integer sums make the result independent of the interleaving."""
    sc = scenes.Scene(8, 160, 120, 0.005)
    fb = sc.W * sc.H * 16
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as a:
        for f in range(8):
            a.integrate(sc.frame(f), sc.poses[f])
        a.clean()
        ref = a.extract()
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as b:
        devs = [b.device_alloc(fb) for _ in range(8)]
        for f in range(8):
            b.device_upload(devs[f], sc.frame(f))
        errs = []

        def work(fs):
            try:
                for f in fs:
                    b.integrate_device(devs[f], 1, fb, sc.W * sc.H, sc.poses[f].reshape(1, 12), frame_ids=np.array([f], np.uint32))
            except Exception as e:  # pragma: no cover
                errs.append(e)
        ts = [threading.Thread(target=work, args=([t, t + 4],)) for t in range(4)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs
        b.clean()
        got = b.extract()
        for d in devs:
            b.device_free(d)
    assert ref.tobytes() == got.tobytes()


def test_pinned_host_entry_point_matches_bounce_path(hfpf_mod, synth_mod):
    """hfpf_integrate_pinned (upload straight from page-locked memory on the copy stream, asynchronous) gives the same bytes as
    hfpf_integrate (bounce copy) frame by frame, with more frames in flight than the staging ring has slots; a pageable buffer
    is refused."""
    sc = scenes.Scene(12, 160, 120, 0.001, fx=615.0, clean_every=5)
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as a:
        ref = scenes.run(a, sc, "integrate")
    fb = sc.W * sc.H * 16
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **SMALL) as g:
        pinned = g.host_alloc(sc.n_frames * fb)
        for f in range(sc.n_frames):
            pinned[f * fb:(f + 1) * fb] = sc.frame(f)
        for ev in sc.schedule():
            if ev[0] == "integrate":
                g.integrate_pinned(pinned[ev[1] * fb:(ev[1] + 1) * fb], sc.poses[ev[1]])
            else:
                g.clean()
        got = g.extract()
        with pytest.raises(hfpf_mod.HfpfError) as e:
            g.integrate_pinned(sc.frame(0), sc.poses[0])  # an ordinary numpy buffer
        assert e.value.code == -2 and "page-locked" in str(e.value)
        g.host_free(pinned)
    assert got.tobytes() == ref.tobytes()


@pytest.mark.parametrize("helpers", ["0", "3"])
def test_bounce_copy_of_large_frames_with_and_without_helper_threads(hfpf_mod, synth_mod, monkeypatch, helpers):
    """hfpf_integrate stages a frame of >= 1 MB with non-temporal stores split over the caller and HFPF_STAGE_THREADS helper
    threads (csrc/hfpf.hip StagePool).  Full 640x480 frames (4.9 MB), more of them than the staging ring has slots, from a source
    buffer that is NOT 16-byte aligned: the rows must equal the device-resident path's byte for byte."""
    sc = scenes.Scene(11, 640, 480, 0.002, clean_every=4)
    caps = dict(max_bricks=60000, max_log_points=8 << 20, max_normals=1 << 20, max_frames=64)
    fb = sc.W * sc.H * 16
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **caps) as a:
        dev = a.device_alloc(fb)
        for ev in sc.schedule():
            if ev[0] == "integrate":
                a.device_upload(dev, sc.frame(ev[1]))
                a.integrate_device(dev, 1, fb, sc.W * sc.H, sc.poses[ev[1]][None])
                a.sync()
            else:
                a.clean()
        ref = a.extract()
        a.device_free(dev)
    monkeypatch.setenv("HFPF_STAGE_THREADS", helpers)
    backing = np.zeros(fb + 64, np.uint8)
    with hfpf_mod.OccupancyGrid(resolution=sc.resolution, bbox=sc.bbox, **caps) as g:
        for ev in sc.schedule():
            if ev[0] == "integrate":
                off = 4 + (ev[1] % 3) * 8  # 4, 12, 20: never a multiple of 16
                view = backing[off:off + fb]
                view[:] = np.frombuffer(sc.frame(ev[1]), np.uint8)
                g.integrate(view, sc.poses[ev[1]])
                view[:] = 0xFF  # the caller's buffer is free again when the call returns
            else:
                g.clean()
        got = g.extract()
    assert len(ref) > 20000 and got.tobytes() == ref.tobytes()
