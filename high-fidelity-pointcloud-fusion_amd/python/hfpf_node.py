"""ctypes binding of the ROS-free node shell (host/libhfpf_node.so, include/hfpf_node.h): the harness that stands in
for roscpp/tf2 so the start/stop/reset/process state machine of pointcloud_fusion_and_filter can be driven in tests."""
import ctypes as C
import os

import numpy as np

import hfpf

_LIB_PATH = os.path.join(hfpf.PKG_DIR, "host", "libhfpf_node.so")
_lib = None

PUBLISH_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p)
TF_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_char), C.c_uint32)

EXPORTS = ["hfpf_node_default_params", "hfpf_node_create", "hfpf_node_destroy", "hfpf_node_last_error", "hfpf_node_on_point_cloud",
           "hfpf_node_start", "hfpf_node_stop", "hfpf_node_reset", "hfpf_node_process", "hfpf_node_clean_now", "hfpf_node_grid",
           "hfpf_node_get_stats", "hfpf_node_set_publisher"]


class Params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("fusion_frame", C.c_char_p), ("directory_name", C.c_char_p),
                ("bounding_box", C.POINTER(C.c_double)), ("bounding_box_len", C.c_uint32), ("engine", hfpf.Config),
                ("clean_period_s", C.c_double), ("final_clean_on_process", C.c_int32), ("write_variants", C.c_int32)]


class CloudMsg(C.Structure):
    _fields_ = [("data", C.c_void_p), ("height", C.c_uint32), ("width", C.c_uint32), ("point_step", C.c_uint32),
                ("row_step", C.c_uint32), ("off_x", C.c_uint32), ("off_y", C.c_uint32), ("off_z", C.c_uint32),
                ("off_rgb", C.c_uint32), ("frame_id", C.c_char_p)]


class TriggerResponse(C.Structure):
    _fields_ = [("success", C.c_int32), ("message", C.c_char * 256)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("received", "integrated", "dropped_not_started", "dropped_tf", "clean_passes", "process_calls")] + \
               [("started", C.c_int32), ("cloud_subscription_started", C.c_int32)]


def lib():
    global _lib
    if _lib is None:
        hfpf.lib()
        L = C.CDLL(_LIB_PATH)
        L.hfpf_node_default_params.argtypes = [C.POINTER(Params)]
        L.hfpf_node_default_params.restype = None
        L.hfpf_node_create.argtypes = [C.POINTER(Params), TF_FN, C.c_void_p, C.POINTER(C.c_void_p)]
        L.hfpf_node_destroy.argtypes = [C.c_void_p]
        L.hfpf_node_last_error.argtypes = [C.c_void_p]
        L.hfpf_node_last_error.restype = C.c_char_p
        L.hfpf_node_on_point_cloud.argtypes = [C.c_void_p, C.POINTER(CloudMsg)]
        for f in ("start", "stop", "reset", "process"):
            getattr(L, "hfpf_node_" + f).argtypes = [C.c_void_p, C.POINTER(TriggerResponse)]
        L.hfpf_node_clean_now.argtypes = [C.c_void_p]
        L.hfpf_node_grid.argtypes = [C.c_void_p]
        L.hfpf_node_grid.restype = C.c_void_p
        L.hfpf_node_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.hfpf_node_set_publisher.argtypes = [C.c_void_p, PUBLISH_FN, C.c_void_p]
        _lib = L
    return _lib


class FusionNode:
    """pointcloud_fusion_and_filter without ROS: params in, Trigger services and the cloud callback as methods."""

    def __init__(self, bounding_box, directory_name="./", fusion_frame="fusion_frame", tf_lookup=None, clean_period_s=0.0,
                 final_clean_on_process=False, resolution=None, write_variants=False, publisher=None, **caps):
        L = lib()
        p = Params()
        L.hfpf_node_default_params(C.byref(p))
        self._keep = [fusion_frame.encode(), os.fsencode(directory_name), (C.c_double * len(bounding_box))(*bounding_box)]
        p.fusion_frame, p.directory_name = self._keep[0], self._keep[1]
        p.bounding_box = C.cast(self._keep[2], C.POINTER(C.c_double))
        p.bounding_box_len = len(bounding_box)
        if resolution is not None:
            p.engine.resolution = resolution
        for k, v in caps.items():
            setattr(p.engine, k, v)
        p.clean_period_s = clean_period_s
        p.final_clean_on_process = 1 if final_clean_on_process else 0
        p.write_variants = 1 if write_variants else 0
        self._tf_py = tf_lookup
        self._pub_py = publisher

        def _tf(user, target, source, pose, err, cap):
            if self._tf_py is None:
                return 0
            try:
                T = np.ascontiguousarray(self._tf_py(target.decode(), source.decode()), dtype=np.float64).reshape(12)
            except Exception as e:  # stands for tf2::TransformException
                msg = str(e).encode()[:cap - 1]
                C.memmove(err, msg, len(msg))
                return 1
            for i in range(12):
                pose[i] = T[i]
            return 0
        self._tf_c = TF_FN(_tf)
        self._h = C.c_void_p()
        rc = L.hfpf_node_create(C.byref(p), self._tf_c, None, C.byref(self._h))
        if rc != 0:
            self._h = None
            raise hfpf.HfpfError(rc, L.hfpf_node_last_error(None).decode())

        def _pub(user, rows, n_rows, frame_id):  # the rows are only valid during the call: copy
            arr = np.frombuffer((C.c_char * (n_rows * hfpf.ROW_DTYPE.itemsize)).from_address(rows), dtype=hfpf.ROW_DTYPE).copy() if n_rows else \
                np.zeros(0, dtype=hfpf.ROW_DTYPE)
            self._pub_py(arr, frame_id.decode())
        self._pub_c = PUBLISH_FN(_pub)
        if publisher is not None:
            L.hfpf_node_set_publisher(self._h, self._pub_c, None)

    def close(self):
        if getattr(self, "_h", None):
            lib().hfpf_node_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _srv(self, name):
        r = TriggerResponse()
        rc = getattr(lib(), "hfpf_node_" + name)(self._h, C.byref(r))
        return rc, bool(r.success), r.message.decode()

    def start(self):
        return self._srv("start")

    def stop(self):
        return self._srv("stop")

    def reset(self):
        return self._srv("reset")

    def process(self):
        return self._srv("process")

    def publish(self, buf, height, width, point_step=16, offsets=(0, 4, 8, 12), frame_id="camera"):
        """A PointCloud2 arriving on ~input_point_cloud.  Returns 1 integrated / 0 dropped."""
        buf = np.ascontiguousarray(buf)
        m = CloudMsg(buf.ctypes.data, height, width, point_step, width * point_step, offsets[0], offsets[1], offsets[2], offsets[3],
                     frame_id.encode())
        rc = lib().hfpf_node_on_point_cloud(self._h, C.byref(m))
        if rc < 0:
            raise hfpf.HfpfError(rc, lib().hfpf_node_last_error(self._h).decode())
        return rc

    def clean_now(self):
        rc = lib().hfpf_node_clean_now(self._h)
        if rc < 0:
            raise hfpf.HfpfError(rc, lib().hfpf_node_last_error(self._h).decode())
        return rc

    def stats(self):
        s = Stats()
        lib().hfpf_node_get_stats(self._h, C.byref(s))
        return {n: getattr(s, n) for n, _ in Stats._fields_}
