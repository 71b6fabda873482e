"""One checked cycle before a multi-rank timed run (bench.py --gpus N).

Every rank runs ONE cycle under a deadline, the ranks' norms are gathered (under the same deadline) and must
agree with rank 0's.  A rank whose cycle or gather does not return says where it was waiting and ends its
process with a non-zero code — the launcher then ends the others; nothing is ever re-exec'ed.  No GPU needed:
the cycle, the gather and the progress report are callables (tests/test_dist_cpu.py drives it over gloo with an
injected stalled rank)."""
import math
import os
import sys
import threading

EXIT_TIMEOUT = 4
EXIT_MISMATCH = 5


def run(rank, world, one_cycle, all_gather, timeout_s, where=None, rtol=1e-12, die=os._exit):
    """one_cycle() -> this rank's (global) residual norm; all_gather(obj) -> list of every rank's obj;
    where() -> text saying what the device was doing (read without synchronising).  Returns the norm."""
    box = {}

    def body():
        try:
            box["stage"] = "inside the cycle"
            # A cycle that RAISES on this rank (a peer-store wait that gave up, a HIP error) must not leave the other
            # ranks alone in the gather below: every rank makes the same sequence of collectives, the failing one
            # contributes (False, message), and every rank then raises the same RuntimeError.
            try:
                mine = (True, float(one_cycle()))
            except Exception as e:                      # noqa: BLE001 - told to every rank below
                mine = (False, "%s: %s" % (type(e).__name__, e))
            box["stage"] = "waiting for the other ranks' norms (the cycle of THIS rank has completed)"
            box["all"] = all_gather(mine)
            box["stage"] = "done"
        except BaseException as e:                      # reported by the caller's thread
            box["error"] = e

    t = threading.Thread(target=body, daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        detail = ""
        if where is not None and box.get("stage") == "inside the cycle":
            try:
                detail = "; device progress: %s" % (where(),)
            except Exception as e:                      # pragma: no cover - diagnostics must not raise
                detail = "; device progress unavailable (%s)" % e
        sys.stderr.write("preflight: rank %d of %d did not finish one cycle within %.0f s: %s%s\n"
                         % (rank, world, timeout_s, box.get("stage", "not started"), detail))
        sys.stderr.flush()
        die(EXIT_TIMEOUT)
        return None
    if "error" in box:
        raise box["error"]
    failed = [(r, v[1]) for r, v in enumerate(box["all"]) if not v[0]]
    if failed:
        raise RuntimeError("preflight: the checked cycle raised on rank(s) %s" % "; ".join("%d (%s)" % f for f in failed))
    norms = [float(v[1]) for v in box["all"]]
    ref = norms[0]
    bad = [r for r, v in enumerate(norms) if not math.isfinite(v) or abs(v - ref) > rtol * abs(ref)]
    if bad or not math.isfinite(ref):
        sys.stderr.write("preflight: rank %d sees residual norms that differ between ranks after one cycle: %s (ranks %s disagree with rank 0)\n"
                         % (rank, ", ".join("%d: %.17g" % (r, v) for r, v in enumerate(norms)), bad))
        sys.stderr.flush()
        die(EXIT_MISMATCH)
        return None
    return norms[rank]
