"""Device arrays at the Python boundary: mgCycle / mgSolve called with `b`, `initial` that live in
HBM (PyTorch-ROCm tensors through `__cuda_array_interface__`) return device arrays holding the bits of the host-array
calls — openmg.mgCycle chained as openmg/__init__.py:132-138 chains it — and parameters['trustOperators'] skips the
per-call checksum for list members passed before.  Each case runs in a process of its own (tests/devarray_worker.py says
why): PyTorch first, as a caller with device tensors has it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "devarray_worker.py")


def run(*args):
    p = subprocess.run([sys.executable, WORKER] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stderr[-4000:]


@pytest.mark.parametrize("smoother,pre,post", [("colour", 1, 1), ("gs", 1, 0), ("colour", 0, 1), ("jacobi", 2, 1)])
def test_chained_mgcycle_on_device_arrays_has_the_bits_of_the_host_calls(smoother, pre, post):
    run("chained", smoother, pre, post)


def test_trusted_operators_skip_the_checksum_but_not_a_new_list():
    run("trusted")


@pytest.mark.parametrize("give_info", [0, 1])
def test_mgsolve_with_a_device_right_hand_side(give_info):
    run("mgsolve", give_info)
