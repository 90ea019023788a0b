"""GPU: every arithmetic leaf of the hot path, as the device code computes it, against the CPU oracle on large
random inputs (bit-exact: these feed integer voxel indices and discrete membership decisions)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TINY = dict(max_bricks=1024, max_log_points=1 << 16, max_normals=1 << 12, max_frames=16)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("res,bbox", [
    (0.005, (-0.80, 1.80, -1.5, 1.5, 0.0, 1.0)),
    (0.001, (-0.5, 0.5, -0.5, 0.5, 0.0, 1.0)),
    (0.0005, (-1.0, 1.0, -0.5, 0.5, 0.0, 1.0)),
    (0.001, (-1.25, 1.25, -1.0, 1.0, 0.0, 2.0)),  # config 5 grid: 2499 x 1999 x 1999
])
def test_transform_zclip_index_bbox_bit_exact(oracle_mod, hfpf_mod, synth_mod, res, bbox):
    rng = np.random.default_rng(11)
    n = 1 << 20
    pts = rng.uniform(-0.7, 0.7, size=(n, 3)).astype(np.float32)
    pts[:, 2] = rng.uniform(0.2, 0.7, size=n).astype(np.float32)
    pts[:1000, 2] = np.float32(0.28)  # exactly on the z-clip bounds (strict compares, node.cpp:252)
    pts[1000:2000, 2] = np.float32(0.6)
    pts[2000:2100] = np.nan
    pts[2100:2200, 0] = np.inf
    og = oracle_mod.OracleGrid(resolution=res, bbox=bbox)
    with hfpf_mod.OccupancyGrid(resolution=res, bbox=bbox, **TINY) as g:
        assert g.dims == og.dims
        for f in range(3):
            T = synth_mod.pose(0x5E3, f)
            q, idx, flags = g.probe_points(T, pts)
            q_ref = oracle_mod.probe_transform(T, pts)
            assert np.array_equal(_bits(q), _bits(q_ref))
            idx_ref, valid_ref = og.probe_index(q_ref)
            assert np.array_equal(idx, idx_ref)
            assert np.array_equal((flags & 2) != 0, valid_ref)
            z = pts[:, 2].astype(np.float64)
            assert np.array_equal((flags & 1) != 0, (z < 0.6) & (z > 0.28))


@pytest.mark.parametrize("res,bbox", [
    (0.005, (-0.80, 1.80, -1.5, 1.5, 0.0, 1.0)),
    (0.001, (-0.5, 0.5, -0.5, 0.5, 0.0, 1.0)),
    (0.0005, (-1.0, 1.0, -0.5, 0.5, 0.0, 1.0)),
    (0.001, (-1.25, 1.25, -1.0, 1.0, 0.0, 2.0)),
])
def test_voxel_index_on_exact_cell_boundaries(oracle_mod, hfpf_mod, synth_mod, res, bbox):
    """The device computes the index as floor(a * (1/res)) and falls back to the exact IEEE division whenever that product is
    within 1e-6 of an integer (geometry.hpp voxel_axis).  Identity pose, coordinates ON the voxel boundaries of all three
    axes and 0-3 f32 ulps either side, plus the bbox faces: every index must equal the oracle's floor((double(p)-min)/res)."""
    rng = np.random.default_rng(21)
    r = float(np.float32(res))  # the grid stores (double)(float)resolution
    n_cells = 60000
    pts = []
    for a in range(3):
        lo, hi = bbox[2 * a], bbox[2 * a + 1]
        k = rng.integers(0, int((hi - lo) / r) + 1, size=n_cells)
        edge = (lo + k * r).astype(np.float32)  # nearest f32 to the boundary
        for step in range(-3, 4):
            c = edge.copy()
            for _ in range(abs(step)):
                c = np.nextafter(c, np.float32(np.inf if step > 0 else -np.inf), dtype=np.float32)
            p = np.empty((n_cells, 3), np.float32)
            for b in range(3):
                p[:, b] = rng.uniform(bbox[2 * b], bbox[2 * b + 1], size=n_cells).astype(np.float32)
            p[:, a] = c
            pts.append(p)
    pts = np.concatenate(pts)
    T = synth_mod.identity_pose()
    og = oracle_mod.OracleGrid(resolution=res, bbox=bbox)
    with hfpf_mod.OccupancyGrid(resolution=res, bbox=bbox, **TINY) as g:
        q, idx, flags = g.probe_points(T, pts)
        assert np.array_equal(_bits(q), _bits(pts))  # identity pose is exact
        idx_ref, valid_ref = og.probe_index(pts)
        assert np.array_equal(idx, idx_ref), "index differs at %d boundary points" % int(np.any(idx != idx_ref, axis=1).sum())
        assert np.array_equal((flags & 2) != 0, valid_ref)
    # the construction really hits both sides of boundaries: neighbouring ulp steps must land in different cells somewhere
    assert len(np.unique(idx_ref[:, 0])) > 100


def test_projection_membership_bit_exact(oracle_mod, hfpf_mod):
    rng = np.random.default_rng(12)
    n = 1 << 20
    c = rng.uniform(-0.5, 0.5, size=(n, 3)).astype(np.float32)
    nn = rng.normal(size=(n, 3))
    nn = (nn / np.linalg.norm(nn, axis=1, keepdims=True)).astype(np.float32)
    perp = np.cross(nn, rng.normal(size=(n, 3)))
    perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    rad = np.where(rng.random(n) < 0.7, rng.uniform(0.000995, 0.001005, n), rng.uniform(0, 0.003, n))
    p = (c + perp * rad[:, None] + nn * rng.uniform(-0.004, 0.004, n)[:, None]).astype(np.float32)
    proj_ref, dist_ref = oracle_mod.probe_project(p, c, nn)
    with hfpf_mod.OccupancyGrid(**TINY) as g:
        proj, dist, member = g.probe_project(p, c, nn)
        member_kernel = g.last_member_kernel_form
    assert np.array_equal(_bits(proj), _bits(proj_ref))
    assert np.array_equal(dist, dist_ref)
    assert np.array_equal(member, dist_ref < 0.001)
    assert 0.2 < member.mean() < 0.8  # the boundary really is exercised on both sides
    # the kernels decide membership on the squared distance (d2 <= largest f32 whose correctly rounded sqrt passes): same decisions
    assert np.array_equal(member_kernel, member)
    near = np.abs(dist_ref - 0.001) < 1e-9  # a few ulps of the f32 distance around the radius
    assert near.sum() > 20 and np.array_equal(member_kernel[near], member[near])


def test_hoisted_division_is_the_plain_division_bit_for_bit(hfpf_mod):
    """k_update_cells keeps the divisor's share of the f32 division in a register per dependant entry (geometry.hpp, LineDiv);
    the projection parameter, the distance and the membership must come out bit-identical to the plain form, including where the
    numerator is tiny or zero (point in the plane through the segment end) and where it leaves the window (plain division then)."""
    rng = np.random.default_rng(21)
    n = 1 << 20
    c = rng.uniform(-0.5, 0.5, size=(n, 3)).astype(np.float32)
    nn = rng.normal(size=(n, 3))
    nn = (nn / np.linalg.norm(nn, axis=1, keepdims=True)).astype(np.float32)
    perp = np.cross(nn, rng.normal(size=(n, 3)))
    perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    rad = rng.uniform(0, 0.002, n)
    along = rng.uniform(-0.02, 0.02, n)
    kind = rng.integers(0, 4, n)
    along = np.where(kind == 1, -0.015 + rng.normal(0, 1e-9, n), along)  # at the segment end a = centre - r*n: numerator ~ 0
    p = (c.astype(np.float64) + perp * rad[:, None] + nn.astype(np.float64) * along[:, None])
    p = np.where((kind == 2)[:, None], p * 10.0 ** rng.uniform(3, 18, (n, 1)), p).astype(np.float32)  # far outside the window
    with hfpf_mod.OccupancyGrid(**TINY) as g:
        g.probe_project(p, c, nn)
        same = g.last_hoisted_division_same
    assert same.all(), f"{(~same).sum()} of {n} inputs differ"


def test_plane_fit_bit_exact(oracle_mod, hfpf_mod):
    rng = np.random.default_rng(13)
    n = 20000
    res, bbox = 0.001, (-0.5, 0.5, -0.5, 0.5, 0.0, 1.0)
    og = oracle_mod.OracleGrid(resolution=res, bbox=bbox)
    cells = rng.integers(0, 999, size=(n, 3)).astype(np.int32)
    cells[:50] = rng.integers(0, 3, size=(50, 3))  # border cells: validCoord clipping
    nn = rng.normal(size=(n, 3))
    nn /= np.linalg.norm(nn, axis=1, keepdims=True)
    ofs = np.stack(np.meshgrid(np.arange(-2, 3), np.arange(-2, 3), np.arange(-2, 3), indexing="ij"), -1).reshape(125, 3)
    dist = np.abs(ofs @ nn.T).T  # (n,125)
    thick = rng.uniform(0.5, 1.6, size=(n, 1))
    occ = ((dist < thick) | (rng.random((n, 125)) < 0.03)).astype(np.uint8)
    vps = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    with hfpf_mod.OccupancyGrid(resolution=res, bbox=bbox, **TINY) as g:
        normals, totals = g.probe_normals(cells, occ, vps)
    checked = 0
    for i in range(0, n, 4):
        t_ref, n_ref = og.probe_normal(int(cells[i, 0]), int(cells[i, 1]), int(cells[i, 2]), occ[i], vps[i])
        assert t_ref == totals[i]
        if t_ref >= 3:
            assert np.array_equal(_bits(normals[i]), _bits(n_ref)), "normal %d differs: %s vs %s" % (i, normals[i], n_ref)
            checked += 1
    assert checked > 4000


def test_det_trig_bit_exact(oracle_mod, hfpf_mod):
    rng = np.random.default_rng(14)
    n = 1 << 20
    y = np.abs(rng.normal(size=n)).astype(np.float32) * np.float32(10.0) ** rng.integers(-6, 3, size=n).astype(np.float32)
    x = rng.normal(size=n).astype(np.float32) * np.float32(10.0) ** rng.integers(-6, 3, size=n).astype(np.float32)
    y[:100] = 0
    x[50:150] = 0
    th = rng.uniform(-np.pi, np.pi, size=n).astype(np.float32)
    a_ref, _, _ = oracle_mod.probe_trig(y, x)
    _, c_ref, s_ref = oracle_mod.probe_trig(y, th)
    with hfpf_mod.OccupancyGrid(**TINY) as g:
        a, _, _ = g.probe_trig(y, x)
        _, c, s = g.probe_trig(y, th)
    assert np.array_equal(_bits(a), _bits(a_ref))
    assert np.array_equal(_bits(c), _bits(c_ref))
    assert np.array_equal(_bits(s), _bits(s_ref))
