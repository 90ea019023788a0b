"""CPU: the two output files of downloadData (grid.hpp:456-488): meta.csv header verbatim (grid.hpp:462), PCD v0.7
ASCII for PointXYZRGBNormal, numbers round-trip.  The writers are host code and need no GPU."""
import os

import numpy as np

import pcd_io
import scenes

META_HEADER = "Id,sdx,sdy,sdz,mean distance from normal, distance from normal sd, points in cylinder"  # grid.hpp:462


def test_pcd_and_meta_roundtrip(tmp_path, hfpf_mod, oracle_mod, synth_mod):
    sc = scenes.Scene(3, 96, 72, 0.005, clean_every=0)
    g = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    rows = scenes.run(g, sc, "capture")
    assert len(rows) > 500
    pcd, meta = str(tmp_path / "test_cloud.pcd"), str(tmp_path / "meta.csv")
    hfpf_mod.write_pcd(rows, pcd)
    hfpf_mod.write_meta_csv(rows, meta)
    hdr, data = pcd_io.read_pcd_ascii(pcd)
    assert hdr["VERSION"] == "0.7"
    assert hdr["FIELDS"] == "x y z rgb normal_x normal_y normal_z curvature"
    assert hdr["SIZE"] == "4 4 4 4 4 4 4 4" and hdr["COUNT"] == "1 1 1 1 1 1 1 1"
    assert hdr["WIDTH"] == str(len(rows)) and hdr["HEIGHT"] == "1" and hdr["POINTS"] == str(len(rows))  # grid.hpp:483-484
    assert hdr["VIEWPOINT"] == "0 0 0 1 0 0 0" and hdr["DATA"] == "ascii"
    assert data.shape == (len(rows), 8)
    for j, f in enumerate(("x", "y", "z")):
        assert np.allclose(data[:, j], rows[f], rtol=1e-7, atol=0)  # %.8g keeps 8 significant digits
    for j, f in zip((4, 5, 6), ("nx", "ny", "nz")):
        assert np.allclose(data[:, j], rows[f], rtol=1e-7, atol=0)
    assert (data[:, 7] == 0).all()  # curvature is never written by the reference
    assert (data[:, 3] == 0xFF000000).all()  # rgb untouched: PCL's default point is r=g=b=0, a=255
    header, m = pcd_io.read_meta_csv(meta)
    assert header == META_HEADER
    assert np.array_equal(m[:, 0], np.arange(len(rows)))  # running id from 0 (grid.hpp:461,478)
    assert np.array_equal(m[:, 6], rows["count"])
    assert np.allclose(m[:, 4], rows["mean_dist"], rtol=1e-5)  # default ostream precision: 6 significant digits


def test_empty_outputs(tmp_path, hfpf_mod):
    rows = np.zeros(0, dtype=hfpf_mod.ROW_DTYPE)
    hfpf_mod.write_pcd(rows, str(tmp_path / "e.pcd"))
    hfpf_mod.write_meta_csv(rows, str(tmp_path / "e.csv"))
    hdr, data = pcd_io.read_pcd_ascii(str(tmp_path / "e.pcd"))
    assert hdr["POINTS"] == "0" and data.shape[0] == 0
    assert open(str(tmp_path / "e.csv")).read() == META_HEADER + "\n"


def test_unwritable_path_reports_io_error(hfpf_mod):
    import pytest
    with pytest.raises(hfpf_mod.HfpfError) as e:
        hfpf_mod.write_pcd(np.zeros(0, dtype=hfpf_mod.ROW_DTYPE), "/nonexistent_dir/x.pcd")
    assert e.value.code == -6


def test_node_library_exports_and_rejects_bad_bounding_box():
    import pytest
    import hfpf
    import hfpf_node
    L = hfpf_node.lib()
    for s in hfpf_node.EXPORTS:
        assert hasattr(L, s)
    # the reference indexes box[0..5] of an empty vector (node.cpp:451,162); the shell refuses instead
    with pytest.raises(hfpf.HfpfError) as e:
        hfpf_node.FusionNode(bounding_box=[])
    assert e.value.code == -1 and "bounding_box" in str(e.value)
    with pytest.raises(hfpf.HfpfError):
        hfpf_node.FusionNode(bounding_box=[0, 1, 0, 1, 0])


def test_alternate_extractors_and_binary_pcd(tmp_path, hfpf_mod, oracle_mod, synth_mod):
    """downloadHQ(threshold) / downloadClassified (grid.hpp:514-575) and the binary PCD variant."""
    sc = scenes.Scene(4, 96, 72, 0.005, clean_every=2)
    g = oracle_mod.OracleGrid(resolution=sc.resolution, bbox=sc.bbox)
    rows = scenes.run(g, sc, "capture")
    thr = 2
    hq = str(tmp_path / "hq.pcd")
    hfpf_mod.write_pcd_xyzrgb(rows, hq, min_count=thr)  # downloadHQ(cloud, 2): count >= 2, white
    hdr, data = pcd_io.read_pcd_ascii(hq)
    keep = rows["count"] >= thr
    assert hdr["FIELDS"] == "x y z rgb" and int(hdr["POINTS"]) == keep.sum() == data.shape[0]
    assert np.allclose(data[:, 0], rows["x"][keep], rtol=1e-7) and (data[:, 3] == 0xFFFFFFFF).all()
    cl = str(tmp_path / "cl.pcd")
    hfpf_mod.write_pcd_xyzrgb(rows, cl, classify_threshold=1)  # downloadClassified with threshold 1
    hdr, data = pcd_io.read_pcd_ascii(cl)
    assert data.shape[0] == len(rows)
    assert np.array_equal(data[:, 3] == 0xFFFF0000, rows["count"] > 1) and np.array_equal(data[:, 3] == 0xFFFFFFFF, rows["count"] <= 1)
    b = str(tmp_path / "b.pcd")
    hfpf_mod.write_pcd_binary(rows, b)
    raw = open(b, "rb").read()
    head, _, body = raw.partition(b"DATA binary\n")
    assert b"POINTS %d" % len(rows) in head and len(body) == 32 * len(rows)
    arr = np.frombuffer(body, dtype=np.float32).reshape(-1, 8)
    assert np.array_equal(arr[:, 0], rows["x"]) and np.array_equal(arr[:, 4], rows["nx"]) and (arr[:, 7] == 0).all()
