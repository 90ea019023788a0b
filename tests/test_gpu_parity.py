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


def test_pcl32_layout(oracle_mod, hfpf_mod, synth_mod):
    """32-byte PCL-style records (x,y,z at 0,4,8; rgb at 16) through the generic decode path."""
    sc = scenes.Scene(3, 160, 120, 0.005, layout=synth_mod.LAYOUT_PCL32, clean_every=2)
    ref, got, occ_ref, occ_got, oc, ctr = _both(oracle_mod, hfpf_mod, sc)
    assert np.array_equal(occ_ref, occ_got)
    scenes.compare_rows(ref, got)
