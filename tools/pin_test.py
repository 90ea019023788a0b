import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
n = 128 << 20
d = C.c_void_p(); hip.hipMalloc(C.byref(d), C.c_size_t(n)); hip.hipMemset(d, 1, C.c_size_t(n)); hip.hipDeviceSynchronize()
for it in range(3):
    t0 = time.perf_counter(); p = C.c_void_p(); rc = hip.hipHostMalloc(C.byref(p), C.c_size_t(n), 0); t1 = time.perf_counter()
    hip.hipMemcpy(p, d, C.c_size_t(n), 2); t2 = time.perf_counter()
    hip.hipMemcpy(p, d, C.c_size_t(n), 2); t3 = time.perf_counter()
    hip.hipHostFree(p); t4 = time.perf_counter()
    print("pinned: alloc %.1f ms, D2H first %.1f ms, D2H again %.1f ms, free %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3))
import numpy as np
for it in range(3):
    t0 = time.perf_counter(); a = np.empty(n, np.uint8); t1 = time.perf_counter()
    hip.hipMemcpy(C.c_void_p(a.ctypes.data), d, C.c_size_t(n), 2); t2 = time.perf_counter()
    hip.hipMemcpy(C.c_void_p(a.ctypes.data), d, C.c_size_t(n), 2); t3 = time.perf_counter()
    print("pageable: alloc %.2f ms, D2H first %.1f ms, D2H again %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))
