"""ctypes binding of the CPU oracle (oracle/libhfpf_oracle.so).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package never does.  See oracle/hfpf_oracle.cpp for the restatement and its citations.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhfpf_oracle.so")


class Config(C.Structure):
    _fields_ = [
        ("resolution", C.c_float),
        ("bbox", C.c_double * 6),
        ("k", C.c_int32),
        ("K", C.c_int32),
        ("gate", C.c_int32),
        ("cylinder_radius", C.c_double),
        ("ball_radius", C.c_double),
        ("z_clip_min", C.c_double),
        ("z_clip_max", C.c_double),
        ("order_mode", C.c_int32),
        ("reserve", C.c_int32),
        ("pcl_shifted_cov", C.c_int32),
        ("fuse_color", C.c_int32),
        ("dense", C.c_int32),
    ]


ROW_DTYPE = np.dtype(
    [
        ("ix", "<i4"), ("iy", "<i4"), ("iz", "<i4"), ("count", "<u4"),
        ("x", "<f4"), ("y", "<f4"), ("z", "<f4"),
        ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"),
        ("sdx", "<f4"), ("sdy", "<f4"), ("sdz", "<f4"),
        ("mean_dist", "<f4"), ("sd_dist", "<f4"), ("rgb", "<u4"),
    ]
)


def build(force=False):
    """Compile the oracle with the recipe in oracle/Makefile."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.horacle_create.restype = C.c_void_p
        L.horacle_create.argtypes = [C.POINTER(Config)]
        L.horacle_destroy.argtypes = [C.c_void_p]
        L.horacle_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        L.horacle_capture.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_uint32, C.c_void_p]
        L.horacle_capture_rgb.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_uint32, C.c_void_p]
        L.horacle_is_dense.argtypes = [C.c_void_p]
        L.horacle_is_dense.restype = C.c_int32
        L.horacle_add_points.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.horacle_clean.argtypes = [C.c_void_p]
        L.horacle_capture_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.horacle_clean_mt.argtypes = [C.c_void_p]
        L.horacle_normals_mt.argtypes = [C.c_void_p]
        L.horacle_normals_mt.restype = C.c_uint64
        L.horacle_set_threads.argtypes = [C.c_int32]
        L.horacle_set_threads.restype = C.c_int32
        L.horacle_extract_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.horacle_extract_mt.restype = C.c_uint64
        L.horacle_is_dirty.argtypes = [C.c_void_p]
        L.horacle_is_dirty.restype = C.c_int32
        L.horacle_clear.argtypes = [C.c_void_p]
        L.horacle_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.horacle_extract.restype = C.c_uint64
        L.horacle_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.horacle_occupied.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.horacle_occupied.restype = C.c_uint64
        L.horacle_dependants.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64]
        L.horacle_dependants.restype = C.c_uint64
        L.horacle_probe_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.horacle_probe_index.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.horacle_probe_center.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.horacle_probe_normal.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.horacle_probe_normal.restype = C.c_int32
        L.horacle_probe_project.argtypes = [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                            C.c_void_p]
        L.horacle_probe_trig.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.horacle_probe_eigen33.argtypes = [C.c_void_p, C.c_void_p]
        L.horacle_sizeof_row.restype = C.c_uint64
        L.horacle_sizeof_config.restype = C.c_uint64
        assert L.horacle_sizeof_row() == ROW_DTYPE.itemsize
        assert L.horacle_sizeof_config() == C.sizeof(Config)
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def make_config(resolution=0.005, bbox=(-0.8, 1.8, -1.5, 1.5, 0.0, 1.0), k=2, K=3, gate=20, cylinder_radius=0.001,
                ball_radius=0.015, z_clip=(0.28, 0.6), order_mode=0, reserve=0, pcl_shifted_cov=False, fuse_color=False, dense=False):
    """Defaults are the reference's constants (node.cpp:91-93,163,311; grid.hpp:34-36,352; launch:7)."""
    c = Config()
    c.resolution = resolution
    for i in range(6):
        c.bbox[i] = float(bbox[i])
    c.k, c.K, c.gate = k, K, gate
    c.cylinder_radius, c.ball_radius = cylinder_radius, ball_radius
    c.z_clip_min, c.z_clip_max = z_clip
    c.order_mode, c.reserve = order_mode, reserve
    c.pcl_shifted_cov = 1 if pcl_shifted_cov else 0
    c.fuse_color = 1 if fuse_color else 0  # EXTENSION (not in the reference): mean colour of the cylinder members
    c.dense = 1 if dense else 0  # the reference's storage: 16 B per cell of the whole box (grid.hpp:626) instead of a hash map
    return c


def set_threads(n):
    """Thread count used by capture_mt/clean_mt (0 = leave as is); returns the count in effect."""
    return int(lib().horacle_set_threads(int(n)))


class OracleGrid:
    """Explicit-schedule driver of the restated OccupancyGrid + capture stage."""

    def __init__(self, **kw):
        self.cfg = make_config(**kw)
        self._h = C.c_void_p(lib().horacle_create(C.byref(self.cfg)))

    def close(self):
        if self._h:
            lib().horacle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def is_dense(self):
        """True when the dense (dim+1)^3 voxel array of the reference could be allocated (dense=True asked for it)."""
        return bool(lib().horacle_is_dense(self._h))

    @property
    def dims(self):
        d = (C.c_int32 * 3)()
        r = C.c_double()
        lib().horacle_dims(self._h, d, C.byref(r))
        return (d[0], d[1], d[2]), r.value

    def capture(self, buf, pose, n_points=None, point_step=16, off_x=0, off_y=4, off_z=8, off_rgb=None):
        """buf: contiguous uint8/any ndarray holding PointCloud2-style records; pose: 3x4 f64.  off_rgb: offset of the packed
        rgb field, read only by the colour extension (fuse_color=True)."""
        buf = np.ascontiguousarray(buf)
        pose = np.ascontiguousarray(pose, dtype=np.float64).reshape(12)
        if n_points is None:
            n_points = buf.nbytes // point_step
        if off_rgb is None:
            lib().horacle_capture(self._h, _p(buf), n_points, point_step, off_x, off_y, off_z, _p(pose))
        else:
            lib().horacle_capture_rgb(self._h, _p(buf), n_points, point_step, off_x, off_y, off_z, off_rgb, _p(pose))

    # All-cores timing baseline: own sharded voxel store, float results not run-to-run reproducible.  Never mix with
    # capture()/clean() on one grid and never use it as a parity checker.
    def capture_mt(self, buf, pose, n_points=None, point_step=16, off_x=0, off_y=4, off_z=8):
        buf = np.ascontiguousarray(buf)
        pose = np.ascontiguousarray(pose, dtype=np.float64).reshape(12)
        if n_points is None:
            n_points = buf.nbytes // point_step
        lib().horacle_capture_mt(self._h, _p(buf), n_points, point_step, off_x, off_y, off_z, _p(pose))

    def clean_mt(self):
        lib().horacle_clean_mt(self._h)

    def normals_mt(self):
        return int(lib().horacle_normals_mt(self._h))

    def extract_mt(self):
        n = lib().horacle_extract_mt(self._h, None, 0)
        rows = np.zeros(n, dtype=ROW_DTYPE)
        if n:
            lib().horacle_extract_mt(self._h, _p(rows), n)
        return rows

    def add_points(self, xyz, viewpoint=(0, 0, 0)):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        vp = np.asarray(viewpoint, dtype=np.float32)
        lib().horacle_add_points(self._h, _p(xyz), xyz.shape[0], _p(vp))

    def clean(self):
        lib().horacle_clean(self._h)

    def is_dirty(self):
        return bool(lib().horacle_is_dirty(self._h))

    def clear(self):
        lib().horacle_clear(self._h)

    def extract(self):
        n = lib().horacle_extract(self._h, None, 0)
        rows = np.zeros(n, dtype=ROW_DTYPE)
        if n:
            lib().horacle_extract(self._h, _p(rows), n)
        return rows

    def counters(self):
        out = np.zeros(6, dtype=np.uint64)
        lib().horacle_counters(self._h, _p(out))
        return dict(zip(["presented", "zclip_pass", "inserted", "occupied", "normals", "buffered"], out.tolist()))

    def occupied(self):
        n = lib().horacle_occupied(self._h, None, 0)
        out = np.zeros((n, 3), dtype=np.int32)
        if n:
            lib().horacle_occupied(self._h, _p(out), n)
        return out

    def dependants(self, x, y, z):
        n = lib().horacle_dependants(self._h, x, y, z, None, 0)
        out = np.zeros((n, 3), dtype=np.int32)
        if n:
            lib().horacle_dependants(self._h, x, y, z, _p(out), n)
        return out

    # ---- leaf probes ----
    def probe_index(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        idx = np.zeros((xyz.shape[0], 3), dtype=np.int32)
        valid = np.zeros(xyz.shape[0], dtype=np.uint8)
        lib().horacle_probe_index(self._h, _p(xyz), xyz.shape[0], _p(idx), _p(valid))
        return idx, valid.astype(bool)

    def probe_center(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32).reshape(-1, 3)
        out = np.zeros((idx.shape[0], 3), dtype=np.float32)
        lib().horacle_probe_center(self._h, _p(idx), idx.shape[0], _p(out))
        return out

    def probe_normal(self, x, y, z, occ, vp=None):
        occ = np.ascontiguousarray(occ, dtype=np.uint8)
        out = np.zeros(3, dtype=np.float32)
        vpa = None if vp is None else np.ascontiguousarray(vp, dtype=np.float32)
        total = lib().horacle_probe_normal(self._h, x, y, z, _p(occ), None if vpa is None else _p(vpa), _p(out))
        return total, out


def probe_transform(pose, xyz):
    pose = np.ascontiguousarray(pose, dtype=np.float64).reshape(12)
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    out = np.zeros_like(xyz)
    lib().horacle_probe_transform(_p(pose), _p(xyz), xyz.shape[0], _p(out))
    return out


def probe_project(pts, centres, normals, ball_r=0.015):
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
    centres = np.ascontiguousarray(centres, dtype=np.float32).reshape(-1, 3)
    normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
    proj = np.zeros_like(pts)
    dist = np.zeros(pts.shape[0], dtype=np.float64)
    lib().horacle_probe_project(C.c_float(ball_r), _p(pts), _p(centres), _p(normals), pts.shape[0], _p(proj), _p(dist))
    return proj, dist


def probe_trig(y, x):
    y = np.ascontiguousarray(y, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    a = np.zeros_like(x)
    c = np.zeros_like(x)
    s = np.zeros_like(x)
    lib().horacle_probe_trig(_p(y), _p(x), x.size, _p(a), _p(c), _p(s))
    return a, c, s


def probe_eigen33(m):
    m = np.ascontiguousarray(m, dtype=np.float32).reshape(9)
    out = np.zeros(3, dtype=np.float32)
    lib().horacle_probe_eigen33(_p(m), _p(out))
    return out
