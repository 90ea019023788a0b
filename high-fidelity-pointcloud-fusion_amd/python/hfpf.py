"""ctypes binding of libhfpf.so (include/hfpf.h) -- the host-side mirror used by tests and bench.py.

`OccupancyGrid` mirrors the public members of the reference's `class OccupancyGrid`
(pointcloud_fusion/pointcloud_fusion/include/utilities/OccupancyGrid.hpp:99-136) as the node uses them
(.../src/pointcloud_fusion_and_filter.cpp:161-164,293,311,398,438): construct, addPoints (here `integrate`,
which also folds in the decode / z-clip / transform of the capture threads), state_changed,
updateThicknessVectors (`clean`), downloadData (`extract` / `download_data`), clearVoxels (`clear`).

There is no CPU path: constructing a grid raises HfpfError when libhfpf.so is missing or no HIP device
is usable.  This module never imports the oracle.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.dirname(_HERE)
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.environ.get("HFPF_LIB") or os.path.join(CSRC_DIR, "libhfpf.so")  # HFPF_LIB: A/B builds during tuning

FLAG_FUSE_COLOR = 1
FLAG_PCL_SHIFTED_COV = 2
FLAG_DIRECT_UPDATE = 4
STATUS = {0: "OK", -1: "BAD_CONFIG", -2: "BAD_ARG", -3: "CAPACITY", -4: "HIP", -5: "STATE", -6: "IO", -7: "DIST"}


class HfpfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("hfpf status %d (%s): %s" % (code, STATUS.get(code, "?"), msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("resolution", C.c_float),
        ("bbox", C.c_double * 6),
        ("k", C.c_int32),
        ("K", C.c_int32),
        ("gate", C.c_int32),
        ("cylinder_radius", C.c_double),
        ("ball_radius", C.c_double),
        ("z_clip_min", C.c_double),
        ("z_clip_max", C.c_double),
        ("device", C.c_int32),
        ("flags", C.c_uint32),
        ("max_bricks", C.c_uint64),
        ("max_log_points", C.c_uint64),
        ("max_normals", C.c_uint64),
        ("max_frames", C.c_uint64),
        ("frame_width", C.c_uint32),
        ("reserved0", C.c_uint32),
        ("max_call_points", C.c_uint64),
    ]


class ExtractOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("classify_threshold", C.c_int32), ("min_count", C.c_double),
                ("paint_white", C.c_int32), ("reserved0", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "points_presented", "points_zclip_pass", "points_in_bbox", "points_buffered", "dep_pairs_tested",
        "dep_pairs_member", "voxels_occupied", "voxels_with_normal", "bricks_allocated", "registrations",
        "dep_entries", "frames_integrated", "clean_passes", "device_bytes", "replay_members", "points_direct", "table_misses", "update_extra_rounds")]


ROW_DTYPE = np.dtype(
    [
        ("ix", "<i4"), ("iy", "<i4"), ("iz", "<i4"), ("count", "<u4"),
        ("x", "<f4"), ("y", "<f4"), ("z", "<f4"),
        ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"),
        ("sdx", "<f4"), ("sdy", "<f4"), ("sdz", "<f4"),
        ("mean_dist", "<f4"), ("sd_dist", "<f4"), ("rgb", "<u4"),
    ]
)

# every symbol include/hfpf.h and include/hfpf_probe.h declare
EXPORTS = [
    "hfpf_default_config", "hfpf_abi_version", "hfpf_create", "hfpf_destroy", "hfpf_last_error", "hfpf_get_dims",
    "hfpf_integrate", "hfpf_integrate_pinned", "hfpf_host_alloc", "hfpf_host_free", "hfpf_integrate_device", "hfpf_is_dirty", "hfpf_clean", "hfpf_extract", "hfpf_extract_filtered", "hfpf_free_rows",
    "hfpf_write_pcd", "hfpf_write_meta_csv", "hfpf_write_pcd_xyzrgb", "hfpf_write_pcd_binary", "hfpf_clear", "hfpf_sync", "hfpf_get_counters", "hfpf_get_occupied",
    "hfpf_device_alloc", "hfpf_device_free", "hfpf_device_upload", "hfpf_kernel_timing", "hfpf_get_kernel_time",
    "hfpf_probe_points", "hfpf_probe_normals", "hfpf_probe_project", "hfpf_probe_trig",
    "hfpf_dist_unique_id", "hfpf_dist_init", "hfpf_dist_info", "hfpf_dist_disable", "hfpf_epoch_export", "hfpf_epoch_import", "hfpf_stats_export",
    "hfpf_extract_with_stats", "hfpf_device_download", "hfpf_device_copy", "hfpf_epoch_import_gathered",
]

EPOCH_REC_DTYPE = np.dtype([("key", "<u8"), ("first_frame", "<u4"), ("vx", "<f4"), ("vy", "<f4"), ("vz", "<f4"), ("pad", "<u4", (2,))])
assert EPOCH_REC_DTYPE.itemsize == 32


def build(force=False):
    """hipcc cross-compiles gfx950 without a GPU; see csrc/Makefile."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", CSRC_DIR, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HfpfError(-4, "libhfpf.so not built (%s); run __graft_entry__.build() -- there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
    L.hfpf_default_config.argtypes = [C.POINTER(Config)]
    L.hfpf_default_config.restype = None
    L.hfpf_abi_version.restype = C.c_int
    L.hfpf_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.hfpf_destroy.argtypes = [vp]
    L.hfpf_last_error.argtypes = [vp]
    L.hfpf_last_error.restype = C.c_char_p
    L.hfpf_get_dims.argtypes = [vp, C.POINTER(i32), C.POINTER(C.c_double)]
    L.hfpf_integrate.argtypes = [vp, vp, u32, u32, u32, u32, u32, u32, vp]
    L.hfpf_integrate_device.argtypes = [vp, vp, u32, u64, u32, u32, u32, u32, u32, u32, vp, vp]
    L.hfpf_integrate_pinned.argtypes = [vp, vp, u32, u32, u32, u32, u32, u32, vp]
    L.hfpf_host_alloc.argtypes = [vp, u64, C.POINTER(vp)]
    L.hfpf_host_free.argtypes = [vp, vp]
    L.hfpf_is_dirty.argtypes = [vp]
    L.hfpf_clean.argtypes = [vp]
    L.hfpf_extract.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.hfpf_extract_filtered.argtypes = [vp, C.POINTER(ExtractOpts), C.POINTER(vp), C.POINTER(u64)]
    L.hfpf_free_rows.argtypes = [vp]
    L.hfpf_free_rows.restype = None
    L.hfpf_write_pcd.argtypes = [vp, u64, C.c_char_p]
    L.hfpf_write_meta_csv.argtypes = [vp, u64, C.c_char_p]
    L.hfpf_write_pcd_xyzrgb.argtypes = [vp, u64, C.c_char_p, u32, i32, i32]
    L.hfpf_write_pcd_binary.argtypes = [vp, u64, C.c_char_p]
    L.hfpf_clear.argtypes = [vp]
    L.hfpf_sync.argtypes = [vp]
    L.hfpf_get_counters.argtypes = [vp, C.POINTER(Counters)]
    L.hfpf_get_occupied.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.hfpf_device_alloc.argtypes = [vp, u64, C.POINTER(vp)]
    L.hfpf_device_free.argtypes = [vp, vp]
    L.hfpf_device_upload.argtypes = [vp, vp, vp, u64]
    L.hfpf_kernel_timing.argtypes = [vp, C.c_int]
    L.hfpf_get_kernel_time.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(u64)]
    L.hfpf_probe_points.argtypes = [vp, vp, vp, u64, vp, vp, vp]
    L.hfpf_probe_normals.argtypes = [vp, u64, vp, vp, vp, vp, vp]
    L.hfpf_probe_project.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp]
    L.hfpf_probe_trig.argtypes = [vp, u64, vp, vp, vp, vp, vp]
    L.hfpf_dist_unique_id.argtypes = [vp]
    L.hfpf_dist_init.argtypes = [vp, C.c_int, C.c_int, vp]
    L.hfpf_dist_disable.argtypes = [vp]
    L.hfpf_dist_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.hfpf_epoch_export.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    L.hfpf_epoch_import.argtypes = [vp, vp, u64]
    L.hfpf_stats_export.argtypes = [vp, C.POINTER(vp), C.POINTER(u64), C.POINTER(vp), C.POINTER(u64)]
    L.hfpf_extract_with_stats.argtypes = [vp, vp, vp, C.POINTER(vp), C.POINTER(u64)]
    L.hfpf_device_download.argtypes = [vp, vp, vp, u64]
    L.hfpf_device_copy.argtypes = [vp, vp, vp, u64]
    L.hfpf_epoch_import_gathered.argtypes = [vp, vp, u64, i32, i32, vp]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def default_config():
    c = Config()
    lib().hfpf_default_config(C.byref(c))
    return c


class _RowBuffer:
    """Owner of one hfpf_extract result (engine-allocated); exposes it to numpy without copying."""

    def __init__(self, ptr, n):
        self._ptr, self._n = ptr, n

    @property
    def __array_interface__(self):
        return {"shape": (self._n,), "typestr": "|V%d" % ROW_DTYPE.itemsize, "descr": ROW_DTYPE.descr, "data": (self._ptr, False), "version": 3}

    def __del__(self):
        try:
            lib().hfpf_free_rows(C.c_void_p(self._ptr))
        except Exception:
            pass


class OccupancyGrid:
    """Device-resident occupancy grid.  Keyword defaults are the reference's constants."""

    def __init__(self, resolution=None, bbox=None, k=None, K=None, gate=None, cylinder_radius=None, ball_radius=None,
                 z_clip=None, device=0, max_bricks=0, max_log_points=0, max_normals=0, max_frames=0, fuse_color=False, pcl_shifted_cov=False, binned_update=None, frame_width=0, max_call_points=0):
        L = lib()
        c = default_config()
        if resolution is not None:
            c.resolution = resolution
        if bbox is not None:
            if len(bbox) != 6:
                raise HfpfError(-1, "bounding_box needs 6 values (xmin,xmax,ymin,ymax,zmin,zmax)")
            for i in range(6):
                c.bbox[i] = float(bbox[i])
        for name, val in (("k", k), ("K", K), ("gate", gate), ("cylinder_radius", cylinder_radius),
                          ("ball_radius", ball_radius)):
            if val is not None:
                setattr(c, name, val)
        if z_clip is not None:
            c.z_clip_min, c.z_clip_max = z_clip
        c.device = device
        if binned_update is None:
            binned_update = os.environ.get("HFPF_BINNED", "1") != "0"  # HFPF_BINNED=0: A/B against the direct form
        c.flags = ((FLAG_FUSE_COLOR if fuse_color else 0) | (FLAG_PCL_SHIFTED_COV if pcl_shifted_cov else 0) |
                   (0 if binned_update else FLAG_DIRECT_UPDATE))
        c.max_bricks, c.max_log_points, c.max_normals, c.max_frames = max_bricks, max_log_points, max_normals, max_frames
        c.frame_width = int(frame_width)  # scheduling hint only (16x16-pixel tiles); results do not depend on it
        c.max_call_points = int(max_call_points)  # 0 = per-call bins grown on demand
        self.cfg = c
        self._transport = None
        self._h = C.c_void_p()
        rc = L.hfpf_create(C.byref(c), C.byref(self._h))
        if rc != 0:
            msg = L.hfpf_last_error(None).decode()
            self._h = None
            raise HfpfError(rc, msg)

    # -- plumbing --
    def _chk(self, rc):
        if rc < 0:
            raise HfpfError(rc, lib().hfpf_last_error(self._h).decode())
        return rc

    def close(self):
        if getattr(self, "_h", None):
            lib().hfpf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def dims(self):
        d = (C.c_int32 * 3)()
        r = C.c_double()
        self._chk(lib().hfpf_get_dims(self._h, d, C.byref(r)))
        return (d[0], d[1], d[2]), r.value

    # -- the reference surface --
    def integrate(self, buf, pose, n_points=None, point_step=16, off_x=0, off_y=4, off_z=8, off_rgb=12):
        """addPoints + capture stage for one host frame (PointCloud2-style records)."""
        buf = np.ascontiguousarray(buf)
        pose = np.ascontiguousarray(pose, dtype=np.float64).reshape(12)
        if n_points is None:
            n_points = buf.nbytes // point_step
        self._chk(lib().hfpf_integrate(self._h, _p(buf), n_points, point_step, off_x, off_y, off_z, off_rgb, _p(pose)))

    def host_alloc(self, nbytes):
        """Page-locked host memory as a uint8 numpy view (free with host_free(view))."""
        p = C.c_void_p()
        self._chk(lib().hfpf_host_alloc(self._h, nbytes, C.byref(p)))
        return np.frombuffer((C.c_uint8 * nbytes).from_address(p.value), dtype=np.uint8)

    def host_free(self, view):
        self._chk(lib().hfpf_host_free(self._h, C.c_void_p(view.ctypes.data)))

    def integrate_pinned(self, buf, pose, n_points=None, point_step=16, off_x=0, off_y=4, off_z=8, off_rgb=12):
        """One frame straight from page-locked memory (a view from host_alloc): asynchronous, no bounce copy."""
        pose = np.ascontiguousarray(pose, dtype=np.float64).reshape(12)
        if n_points is None:
            n_points = buf.nbytes // point_step
        self._chk(lib().hfpf_integrate_pinned(self._h, C.c_void_p(buf.ctypes.data), n_points, point_step, off_x, off_y, off_z, off_rgb, _p(pose)))

    def integrate_device(self, dev_ptr, n_frames, frame_stride, n_points, poses, frame_ids=None, point_step=16, off_x=0,
                         off_y=4, off_z=8, off_rgb=12):
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(n_frames, 12)
        ids = None
        if frame_ids is not None:
            ids = np.ascontiguousarray(frame_ids, dtype=np.uint32)
        self._chk(lib().hfpf_integrate_device(self._h, C.c_void_p(dev_ptr), n_frames, frame_stride, n_points, point_step,
                                              off_x, off_y, off_z, off_rgb, _p(poses), _p(ids) if ids is not None else None))

    @property
    def state_changed(self):
        return bool(self._chk(lib().hfpf_is_dirty(self._h)))

    def clean(self):
        """updateThicknessVectors.  With a host-staged transport attached (see hfpf_dist.py) the epoch exchange
        runs first; with RCCL (dist_init_rccl) the engine does it internally.  Collective across ranks."""
        if self._transport is not None:
            self._transport.exchange(self)
        self._chk(lib().hfpf_clean(self._h))

    def _rows_out(self, rows, n):
        """Zero-copy: a numpy view over the engine-owned row buffer; hfpf_free_rows runs when the view is collected."""
        if not n.value:
            return np.zeros(0, dtype=ROW_DTYPE)
        return np.asarray(_RowBuffer(rows.value, n.value))

    def extract(self):
        if self._transport is not None:
            return self._transport.merged_extract(self)
        rows = C.c_void_p()
        n = C.c_uint64()
        self._chk(lib().hfpf_extract(self._h, C.byref(rows), C.byref(n)))
        return self._rows_out(rows, n)

    def extract_filtered(self, min_count=0.0, classify_threshold=-1, paint_white=False):
        """downloadHQ(threshold) / downloadClassified / download of the reference (grid.hpp:491-601): the same ordered
        extract with the count filter and the colour coding done on the device."""
        o = ExtractOpts(C.sizeof(ExtractOpts), int(classify_threshold), float(min_count), 1 if paint_white else 0, 0)
        rows = C.c_void_p()
        n = C.c_uint64()
        self._chk(lib().hfpf_extract_filtered(self._h, C.byref(o), C.byref(rows), C.byref(n)))
        return self._rows_out(rows, n)

    # -- multi-GPU --
    def dist_init_rccl(self, rank, world, unique_id):
        """unique_id: the 128 bytes rank 0 got from dist_unique_id(), broadcast by the launcher."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        self._chk(lib().hfpf_dist_init(self._h, rank, world, buf))

    def dist_world(self):
        """Rank count the engine's RCCL communicator reports (1 without a communicator)."""
        r, w = C.c_int32(), C.c_int32()
        self._chk(lib().hfpf_dist_info(self._h, C.byref(r), C.byref(w)))
        return w.value

    def dist_disable(self):
        self._chk(lib().hfpf_dist_disable(self._h))

    def attach_transport(self, transport):
        self._transport = transport

    def epoch_export(self):
        """-> (device pointer, n_records) of the 16-byte records of the cells occupied and the frames integrated since the last exchange."""
        p = C.c_void_p()
        n = C.c_uint64()
        self._chk(lib().hfpf_epoch_export(self._h, C.byref(p), C.byref(n)))
        return p.value or 0, n.value

    def epoch_import(self, dev_ptr, n_records):
        self._chk(lib().hfpf_epoch_import(self._h, C.c_void_p(dev_ptr), n_records))

    def epoch_import_gathered(self, dev_buffer, slice_stride_bytes, world, my_rank, counts):
        """Import every other rank's slice of a padded all-gather buffer (the layout ncclAllGather leaves behind)."""
        c = np.ascontiguousarray(counts, dtype=np.uint64)
        self._chk(lib().hfpf_epoch_import_gathered(self._h, C.c_void_p(dev_buffer), slice_stride_bytes, world, my_rank, _p(c)))

    def device_copy(self, dev_dst, dev_src, nbytes):
        self._chk(lib().hfpf_device_copy(self._h, C.c_void_p(dev_dst), C.c_void_p(dev_src), nbytes))

    def stats_export(self):
        """-> (dev ptr, n_words, colour dev ptr or 0, n_colour_words) of this handle's partial int64 sums."""
        p, pc = C.c_void_p(), C.c_void_p()
        n, nc = C.c_uint64(), C.c_uint64()
        self._chk(lib().hfpf_stats_export(self._h, C.byref(p), C.byref(n), C.byref(pc), C.byref(nc)))
        return p.value or 0, n.value, pc.value or 0, nc.value

    def extract_with_stats(self, dev_words, dev_cwords=0):
        rows = C.c_void_p()
        n = C.c_uint64()
        self._chk(lib().hfpf_extract_with_stats(self._h, C.c_void_p(dev_words), C.c_void_p(dev_cwords) if dev_cwords else None,
                                                C.byref(rows), C.byref(n)))
        return self._rows_out(rows, n)

    def device_download(self, dev_ptr, nbytes, dtype=np.uint8):
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        self._chk(lib().hfpf_device_download(self._h, _p(out), C.c_void_p(dev_ptr), out.nbytes))
        return out

    def download_data(self, cloud_location, metadata):
        """downloadData(cloud_location, metadata): writes test_cloud.pcd and meta.csv; returns the rows."""
        rows = self.extract()
        write_pcd(rows, cloud_location)
        write_meta_csv(rows, metadata)
        return rows

    def clear(self):
        self._chk(lib().hfpf_clear(self._h))

    # -- diagnostics / harness --
    def sync(self):
        self._chk(lib().hfpf_sync(self._h))

    def counters(self):
        c = Counters()
        self._chk(lib().hfpf_get_counters(self._h, C.byref(c)))
        return {n: getattr(c, n) for n, _ in Counters._fields_}

    def occupied(self):
        n = C.c_uint64()
        self._chk(lib().hfpf_get_occupied(self._h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 3), dtype=np.int32)
        if n.value:
            self._chk(lib().hfpf_get_occupied(self._h, _p(out), n.value, C.byref(n)))
        return out

    def device_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(lib().hfpf_device_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def device_free(self, ptr):
        self._chk(lib().hfpf_device_free(self._h, C.c_void_p(ptr)))

    def device_upload(self, dev_ptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(lib().hfpf_device_upload(self._h, C.c_void_p(dev_ptr), _p(arr), arr.nbytes))

    def kernel_timing(self, enable=True):
        """True / 1: integrate calls and clean passes; 2: also each kernel of an integrate call (kernel_time ids 2..4)."""
        self._chk(lib().hfpf_kernel_timing(self._h, int(enable)))

    def kernel_time(self, kernel_id=0):
        ms = C.c_double()
        n = C.c_uint64()
        self._chk(lib().hfpf_get_kernel_time(self._h, kernel_id, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # -- leaf probes (tests) --
    def probe_points(self, pose, xyz):
        pose = np.ascontiguousarray(pose, dtype=np.float64).reshape(12)
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        n = xyz.shape[0]
        q = np.zeros((n, 3), np.float32)
        idx = np.zeros((n, 3), np.int32)
        flags = np.zeros(n, np.uint8)
        self._chk(lib().hfpf_probe_points(self._h, _p(pose), _p(xyz), n, _p(q), _p(idx), _p(flags)))
        return q, idx, flags

    def probe_normals(self, cells, occ, vps):
        cells = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 3)
        n = cells.shape[0]
        occ = np.ascontiguousarray(occ, dtype=np.uint8).reshape(n, 125)
        vps = np.ascontiguousarray(vps, dtype=np.float32).reshape(n, 3)
        normals = np.zeros((n, 3), np.float32)
        totals = np.zeros(n, np.int32)
        self._chk(lib().hfpf_probe_normals(self._h, n, _p(cells), _p(occ), _p(vps), _p(normals), _p(totals)))
        return normals, totals

    def probe_project(self, pts, centres, normals):
        pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
        n = pts.shape[0]
        centres = np.ascontiguousarray(centres, dtype=np.float32).reshape(n, 3)
        normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(n, 3)
        proj = np.zeros((n, 3), np.float32)
        dist = np.zeros(n, np.float64)
        member = np.zeros(n, np.uint8)
        self._chk(lib().hfpf_probe_project(self._h, n, _p(pts), _p(centres), _p(normals), _p(proj), _p(dist), _p(member)))
        # bit 0: the reference's form (f32 sqrt widened, compared with the radius); bit 1: the kernels' form (squared
        # distance against the precomputed largest passing value) -- kept for the test that the two always agree
        self.last_member_kernel_form = (member & 2) != 0
        self.last_hoisted_division_same = (member & 4) != 0
        return proj, dist, (member & 1) != 0

    def probe_trig(self, y, x):
        y = np.ascontiguousarray(y, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.float32)
        a = np.zeros_like(x)
        c = np.zeros_like(x)
        s = np.zeros_like(x)
        self._chk(lib().hfpf_probe_trig(self._h, x.size, _p(y), _p(x), _p(a), _p(c), _p(s)))
        return a, c, s


def dist_unique_id():
    buf = (C.c_char * 128)()
    rc = lib().hfpf_dist_unique_id(buf)
    if rc != 0:
        raise HfpfError(rc, lib().hfpf_last_error(None).decode())
    return bytes(buf)


def write_pcd(rows, path):
    rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
    rc = lib().hfpf_write_pcd(_p(rows), rows.size, os.fsencode(path))
    if rc != 0:
        raise HfpfError(rc, "write_pcd(%s)" % path)


def write_meta_csv(rows, path):
    rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
    rc = lib().hfpf_write_meta_csv(_p(rows), rows.size, os.fsencode(path))
    if rc != 0:
        raise HfpfError(rc, "write_meta_csv(%s)" % path)


def write_pcd_xyzrgb(rows, path, min_count=0, classify_threshold=-1, white=True):
    """download / downloadHQ(threshold) / downloadClassified of the reference (grid.hpp:491-575)."""
    rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
    rc = lib().hfpf_write_pcd_xyzrgb(_p(rows), rows.size, os.fsencode(path), min_count, classify_threshold, 1 if white else 0)
    if rc != 0:
        raise HfpfError(rc, "write_pcd_xyzrgb(%s)" % path)


def write_pcd_binary(rows, path):
    rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
    rc = lib().hfpf_write_pcd_binary(_p(rows), rows.size, os.fsencode(path))
    if rc != 0:
        raise HfpfError(rc, "write_pcd_binary(%s)" % path)
