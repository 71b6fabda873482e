"""Nothing the fused passes compute may depend on what freshly allocated device memory happens to hold (zeros in a young
process, arbitrary bits later: the 27-point sweeps once multiplied a zero coefficient with such bits and a long test run
turned its iterate into NaNs).  tests/poison_worker.py runs the fused paths beside the set-by-set schedule of the same
hierarchy in a process whose every device allocation starts as NaN patterns (OMG_POISON=1)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fused_passes_do_not_read_unwritten_memory():
    env = dict(os.environ, OMG_POISON="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "poison_worker.py")], env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    report = json.loads(run.stdout.strip().splitlines()[-1])
    assert len(report) == 12
    for name, cases in report.items():
        for c in cases:
            assert c["finite"] and c["same_bits"], (name, c)
            assert c["norm_rel"] <= 1e-6 if "float32" in name else c["norm_rel"] <= 1e-12, (name, c)
