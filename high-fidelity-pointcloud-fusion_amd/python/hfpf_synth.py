"""ctypes binding of the synthetic RGB-D frame generator (synth/libhfpf_synth.so).

Stands in for the sensor driver + tf2 of the reference node (node.cpp:152,336): it produces
PointCloud2-shaped byte buffers (height=1, width=W*H) and 3x4 f64 fusion_frame<-camera poses.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(os.path.dirname(_HERE), "synth")
_LIB_PATH = os.path.join(_DIR, "libhfpf_synth.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fopenmp", "-fPIC", "-shared", "-o", _LIB_PATH,
                               os.path.join(_DIR, "synth.cpp")])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        # GPU boxes expose every host CPU but grant a small share; an OpenMP team per visible CPU thrashes.
        try:
            n = len(os.sched_getaffinity(0))
        except AttributeError:
            n = os.cpu_count() or 1
        os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, n))))
        L = C.CDLL(_LIB_PATH)
        L.hfpf_synth_pose.argtypes = [C.c_uint64, C.c_uint32, C.c_double, C.c_double, C.c_void_p]
        L.hfpf_synth_frame.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_void_p,
                                       C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_void_p]
        _lib = L
    return _lib


LAYOUT_PACKED16 = dict(point_step=16, off_x=0, off_y=4, off_z=8, off_rgb=12)
LAYOUT_PCL32 = dict(point_step=32, off_x=0, off_y=4, off_z=8, off_rgb=16)


def pose(seed, frame_idx, max_angle_deg=30.0, jitter=0.05):
    out = np.zeros(12, dtype=np.float64)
    lib().hfpf_synth_pose(seed, frame_idx, max_angle_deg, jitter, out.ctypes.data_as(C.c_void_p))
    return out.reshape(3, 4)


def identity_pose():
    return np.hstack([np.eye(3), np.zeros((3, 1))]).astype(np.float64)


def frame(seed, frame_idx, W, H, pose34, noise_sigma=0.0005, nan_permille=20, fx=0.0, layout=LAYOUT_PACKED16,
          out=None):
    """Returns a uint8 array of W*H*point_step bytes (camera-frame XYZRGB records)."""
    ps = layout["point_step"]
    if out is None:
        out = np.empty(W * H * ps, dtype=np.uint8)
    p = np.ascontiguousarray(pose34, dtype=np.float64).reshape(12)
    lib().hfpf_synth_frame(seed, frame_idx, W, H, fx, p.ctypes.data_as(C.c_void_p), noise_sigma, nan_permille, ps,
                           layout["off_x"], layout["off_y"], layout["off_z"], layout["off_rgb"],
                           out.ctypes.data_as(C.c_void_p))
    return out
