#!/usr/bin/env python3
"""Floor of one halo exchange on this GPU: grouped ncclSend + ncclRecv to the rank itself on a one-rank RCCL
communicator (the calls the multi-GPU runner makes per colour and neighbour), plane-sized messages.
    python tools/exchange_probe.py            -> profiles/rNN_exchange_probe.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmg_amd import _hip, _hip_dist  # noqa: E402

_hip.require_gpu()
print("grouped ncclSend + ncclRecv to self, one-rank communicator, 200 back to back, each followed by a small kernel")
for label, n in (("8 B (norm all-reduce sized)", 8), ("32 KiB (coarse all-gather)", 32 << 10),
                 ("half a 64^2 plane, fp64 (level 3 of 512^3): 16 KiB", 16 << 10),
                 ("half a 128^2 plane: 64 KiB", 64 << 10), ("half a 256^2 plane: 256 KiB", 256 << 10),
                 ("half a 512^2 plane (level 0 of 512^3, one colour): 1 MiB", 1 << 20), ("a 512^2 plane: 2 MiB", 2 << 20)):
    us = _hip_dist.self_exchange_us(n, 200)
    print("%-62s %8.2f us per exchange   (%.1f GB/s)" % (label, us, n / us / 1e3))
