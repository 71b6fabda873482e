#!/usr/bin/env python3
"""First-touch cost of fresh host arrays on this box: plain np.empty vs an mmap advised MADV_HUGEPAGE."""
import mmap
import time

import numpy as np

N = 175 * (1 << 20)
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "|", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
for rep in range(3):
    t = time.perf_counter(); a = np.empty(N, dtype=np.uint8); a[::4096] = 1; d1 = time.perf_counter() - t
    t = time.perf_counter(); a[:] = 2; d2 = time.perf_counter() - t
    print("np.empty + touch every page %.1f ms; full memset afterwards %.1f ms" % (1e3 * d1, 1e3 * d2))
    del a
for rep in range(3):
    t = time.perf_counter()
    m = mmap.mmap(-1, N + (2 << 20))
    m.madvise(mmap.MADV_HUGEPAGE)
    a = np.frombuffer(m, dtype=np.uint8)
    off = (-a.ctypes.data) % (2 << 20)
    a = a[off:off + N]
    a[::4096] = 1
    d1 = time.perf_counter() - t
    print("mmap + MADV_HUGEPAGE + touch %.1f ms" % (1e3 * d1))
    del a, m
for rep in range(3):
    t = time.perf_counter()
    m = mmap.mmap(-1, N, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | mmap.MAP_POPULATE)
    a = np.frombuffer(m, dtype=np.uint8)
    d0 = time.perf_counter() - t
    t = time.perf_counter(); a[::4096] = 1; d1 = time.perf_counter() - t
    print("mmap MAP_POPULATE %.1f ms, then touch %.1f ms" % (1e3 * d0, 1e3 * d1))
    del a, m
import ctypes
libc = ctypes.CDLL("libc.so.6", use_errno=True)
for rep in range(2):
    t = time.perf_counter()
    m = mmap.mmap(-1, N + (2 << 20))
    a = np.frombuffer(m, dtype=np.uint8)
    off = (-a.ctypes.data) % (2 << 20)
    MADV_HUGEPAGE, MADV_POPULATE_WRITE = 14, 23
    r1 = libc.madvise(ctypes.c_void_p(a.ctypes.data + off), ctypes.c_size_t(N), MADV_HUGEPAGE)
    r2 = libc.madvise(ctypes.c_void_p(a.ctypes.data + off), ctypes.c_size_t(N), MADV_POPULATE_WRITE)
    d0 = time.perf_counter() - t
    print("mmap + MADV_HUGEPAGE + MADV_POPULATE_WRITE %.1f ms (rc %d %d errno %d)" % (1e3 * d0, r1, r2, ctypes.get_errno()))
    del a, m
