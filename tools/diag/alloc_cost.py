#!/usr/bin/env python3
"""What a protocol call pays for its per-call device buffers: hipMalloc + hipFree of a dozen small blocks, and a pageable host-to-device copy of 100 B - 1 KB on the NULL stream."""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]; hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
sizes = [104, 200, 104, 104, 96, 200, 200, 576, 4, 8, 104, 104]
def once():
    ps = []
    for s in sizes:
        p = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(p), s) == 0; ps.append(p)
    return ps
host = (ctypes.c_uint8 * 1024)()
for rep in range(3):
    t0 = time.perf_counter(); ps = once(); t1 = time.perf_counter()
    for p, s in zip(ps, sizes): assert hip.hipMemcpyAsync(p, host, s, 1, None) == 0
    hip.hipStreamSynchronize(None); t2 = time.perf_counter()
    for p in ps: hip.hipFree(p)
    t3 = time.perf_counter()
    print(f"12 x hipMalloc {1e6*(t1-t0):.0f} us, 12 pageable uploads + sync {1e6*(t2-t1):.0f} us, 12 x hipFree {1e6*(t3-t2):.0f} us")
