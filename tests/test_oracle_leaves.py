"""CPU: leaf arithmetic of the oracle against independent numpy restatements and the reference's quirk ledger
(SURVEY.md Appendix A).  These pin the oracle itself; the GPU engine is compared with the oracle elsewhere."""
import numpy as np
import pytest


def test_resolution_passes_through_float_and_dims_truncate(oracle_mod):
    # setResolution(float) into double members (grid.hpp:614-619); int truncation of dims (grid.hpp:623-625)
    g = oracle_mod.OracleGrid(resolution=0.005, bbox=(-0.80, 1.80, -1.5, 1.5, 0, 1.0))  # shipped launch file
    dims, res = g.dims
    assert res == float(np.float32(0.005)) == 0.004999999888241291
    assert dims == (520, 600, 200)
    g1 = oracle_mod.OracleGrid(resolution=0.001, bbox=(-0.5, 0.5, -0.5, 0.5, 0, 1.0))
    assert g1.dims == ((999, 999, 999), 0.0010000000474974513)
    g2 = oracle_mod.OracleGrid(resolution=0.0005, bbox=(-1, 1, -0.5, 0.5, 0, 1.0))
    assert g2.dims == ((3999, 1999, 1999), 0.0005000000237487257)


def test_index_matches_numpy_f64(oracle_mod):
    rng = np.random.default_rng(1)
    bbox = (-0.8, 1.8, -1.5, 1.5, 0.0, 1.0)
    for res in (0.005, 0.001, 0.0005):
        g = oracle_mod.OracleGrid(resolution=res, bbox=bbox)
        pts = rng.uniform([-0.9, -1.6, -0.1], [1.9, 1.6, 1.1], size=(20000, 3)).astype(np.float32)
        idx, valid = g.probe_index(pts)
        r = float(np.float32(res))
        mn = np.array(bbox[0::2])
        mx = np.array(bbox[1::2])
        ref = np.floor((pts.astype(np.float64) - mn) / r).astype(np.int64)
        assert np.array_equal(idx, ref)
        p64 = pts.astype(np.float64)
        vref = ~((p64 >= mx).any(1) | (p64 <= mn).any(1))  # strict interior, grid.hpp:644
        assert np.array_equal(valid, vref)


def test_bbox_is_strict_and_cell_dim_is_reachable(oracle_mod):
    g = oracle_mod.OracleGrid(resolution=0.001, bbox=(-0.5, 0.5, -0.5, 0.5, 0, 1.0))
    pts = np.array([[-0.5, 0, 0.5], [0.5, 0, 0.5], [0.49995, 0, 0.5], [np.nan, 0, 0.5]], np.float32)
    idx, valid = g.probe_index(pts)
    assert list(valid) == [False, False, True, True]  # NaN passes validPoints in the reference (all compares false)
    assert idx[2, 0] == 999  # == xdim: stored (grid.hpp:626) but never scanned (grid.hpp:463,649)
    assert idx[3, 0] == -2 ** 31  # cvttsd2si(NaN)


def test_transform_is_f64_left_to_right(oracle_mod, synth_mod):
    rng = np.random.default_rng(2)
    T = synth_mod.pose(0x5E3, 7)
    pts = rng.uniform(-1, 1, size=(5000, 3)).astype(np.float32)
    out = oracle_mod.probe_transform(T, pts)
    p = pts.astype(np.float64)
    ref = np.stack([((T[r, 0] * p[:, 0] + T[r, 1] * p[:, 1]) + T[r, 2] * p[:, 2]) + T[r, 3] for r in range(3)], 1).astype(np.float32)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


def _project_np(p, c, n, ball=np.float32(0.015)):
    f = np.float32
    d = (ball * n).astype(f)
    a = (c - d).astype(f)
    b = (c + d).astype(f)
    ap = (a - p).astype(f)
    ab = (a - b).astype(f)

    def dot(u, v):
        t = (u * v).astype(f)
        return f(t[0] + f(t[1] + t[2]))
    s = f(dot(ap, ab) / dot(ab, ab))
    proj = (a - (s * ab).astype(f)).astype(f)
    df = (p - proj).astype(f)
    return proj, float(np.sqrt(dot(df, df), dtype=f))


def test_projection_and_cylinder_membership(oracle_mod):
    rng = np.random.default_rng(3)
    n = 2000
    c = rng.uniform(-0.4, 0.4, size=(n, 3)).astype(np.float32)
    nn = rng.normal(size=(n, 3))
    nn = (nn / np.linalg.norm(nn, axis=1, keepdims=True)).astype(np.float32)
    p = (c + rng.normal(scale=0.0012, size=(n, 3))).astype(np.float32)
    proj, dist = oracle_mod.probe_project(p, c, nn)
    for i in range(0, n, 7):
        pr, di = _project_np(p[i], c[i], nn[i])
        assert np.array_equal(pr.view(np.uint32), proj[i].view(np.uint32))
        assert di == dist[i]
    # geometric sanity: distance to the line, membership near the 1 mm boundary exists on both sides
    t = np.einsum("ij,ij->i", (p - c).astype(np.float64), nn.astype(np.float64))
    perp = np.linalg.norm((p - c).astype(np.float64) - t[:, None] * nn, axis=1)
    assert np.allclose(dist, perp, atol=2e-6)
    assert (dist < 0.001).any() and (dist >= 0.001).any()


def _stencil(fn):
    occ = np.zeros(125, np.uint8)
    d = 0
    for i in range(-2, 3):
        for j in range(-2, 3):
            for k in range(-2, 3):
                occ[d] = 1 if fn(i, j, k) else 0
                d += 1
    return occ


def test_plane_fit_on_lattice_stencils(oracle_mod):
    g = oracle_mod.OracleGrid(resolution=0.005, bbox=(-0.5, 0.5, -0.5, 0.5, 0, 1.0))
    total, nrm = g.probe_normal(100, 100, 100, _stencil(lambda i, j, k: k == 0))
    assert total == 25 and abs(abs(nrm[2]) - 1) < 1e-3 and abs(nrm[0]) < 2e-2 and abs(nrm[1]) < 2e-2
    total, nrm = g.probe_normal(100, 100, 100, _stencil(lambda i, j, k: i == 0))
    assert total == 25 and abs(abs(nrm[0]) - 1) < 1e-3
    total, nrm = g.probe_normal(100, 100, 100, _stencil(lambda i, j, k: i == k))  # 45 degree plane
    assert total == 25 and abs(abs(nrm[0]) - 2 ** -0.5) < 2e-2 and abs(abs(nrm[2]) - 2 ** -0.5) < 2e-2 and nrm[0] * nrm[2] < 0
    assert abs(np.linalg.norm(nrm) - 1) < 1e-5
    # validCoord clips the stencil at the grid border (grid.hpp:337): corner cell sees 27 cells
    total, _ = g.probe_normal(0, 0, 0, np.ones(125, np.uint8))
    assert total == 27


def test_eigen33_against_numpy(oracle_mod):
    rng = np.random.default_rng(5)
    for _ in range(200):
        a = rng.normal(size=(30, 3)) * np.array([1.0, 0.6, 0.05])
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        a = a @ q.T
        cov = np.cov(a.T, bias=True).astype(np.float32)
        v = oracle_mod.probe_eigen33(cov)
        w, vec = np.linalg.eigh(cov.astype(np.float64))
        ref = vec[:, 0]
        assert abs(abs(float(v @ ref)) - 1) < 1e-4


def test_det_trig_close_to_correctly_rounded(oracle_mod):
    """The deterministic atan2/cos/sin (oracle/det_math.h) against f64 numpy rounded to f32."""
    rng = np.random.default_rng(6)
    y = np.abs(rng.normal(size=200000)).astype(np.float32)
    x = rng.normal(size=200000).astype(np.float32)
    th = rng.uniform(0, np.pi / 3, size=200000).astype(np.float32)
    a, _, _ = oracle_mod.probe_trig(y, x)
    _, c, s = oracle_mod.probe_trig(y, th)
    for got, ref in ((a, np.arctan2(y.astype(np.float64), x.astype(np.float64))), (c, np.cos(th.astype(np.float64))),
                     (s, np.sin(th.astype(np.float64)))):
        ref32 = ref.astype(np.float32)
        ulp = np.abs(got.view(np.int32).astype(np.int64) - ref32.view(np.int32).astype(np.int64))
        assert ulp.max() <= 1
        assert (ulp != 0).mean() < 1e-4  # differs from the correctly rounded value only at f64->f32 double-rounding ties
