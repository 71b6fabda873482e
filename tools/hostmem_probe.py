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
