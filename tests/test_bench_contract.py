"""The committed bench lines and the PMC profiles they may quote (profiles/r03_*, profiles/r05_*): the driver's contract
keys, a roofline fraction that is a fraction, and the rule that HBM traffic measured by rocprofv3 is
only ever attached to a build whose kernel sources hash to the profile's."""
import json
import os
import re

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# the round whose final collection is tied to the committed kernel sources (earlier rounds' files are history)
LATEST = "r06" if os.path.exists(os.path.join(ROOT, "profiles", "r06_pmc_plane_down.json")) else None
ROUNDS = ["r03", "r05"] + (["r06"] if LATEST else [])


def _load(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1]) if name.endswith("bench.json") else json.load(f)


@pytest.mark.parametrize("rnd", ROUNDS)
def test_committed_bench_line_keeps_the_contract(rnd):
    d = _load(rnd + "_bench.json")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "V-cycles/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 1e-3          # value = steps / time of exactly those steps
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) / r["achieved"] < 1e-2
    assert r["csr_equiv_bytes"] > r["bytes_per_launch"]                          # the CSR-byte rate is reported, never used for frac
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "V-cycles/s" and c["value"] > 0 and c["sample"]
    assert d["csr_path"]["residual"]["frac"] <= 1.0 and d["csr_path"]["fine_grid_spmv"]["frac"] <= 1.0
    assert all(d["config"]["plane_levels"]) and d["set_schedule"]["vcycles_per_s"] < d["value"]
    for k in ("plane_down", "plane_up"):
        assert 0.0 < r["level0_kernels"][k]["frac"] <= 1.0 and r["level0_kernels"][k]["launches_per_cycle"] == 1.0
    assert d["reference_smoother"]["vcycles_per_s"] > 0 and all(d["reference_smoother"]["wavefront_levels"])
    assert re.fullmatch(r"[0-9a-f]{16}", d["config"]["kernel_src_sha"]) and re.fullmatch(r"[0-9a-f]{40}", d["config"]["git_head"])
    if rnd >= "r05":
        # round 5's legs: the matrix-free fine-grid SpMV over the line VERDICT r4 drew, BASELINE configs[0] and [4] (the latter
        # with its own roofline and the coefficient update), which kind of allocation the finest level's vectors got
        assert 0.70 <= d["fine_grid_spmv"]["frac"] <= 1.0
        assert d["config0"]["vcycles_per_s"] > 1500 and d["config0"]["grids"] == 3
        c4 = d["config4"]
        assert c4["stencil27_kernels"] and 0.0 < c4["roofline"]["frac"] <= 1.0 and c4["dtype"] == "f32"
        assert c4["update_fine_device_ms"] is not None and c4["update_fine_device_ms"] < c4["update_fine_device_plus_cycle_ms"] < 20.0
        assert d["config"]["process_population"]["which"] in ("fast", "slow")
        assert d["default_cycle"]["plane"] and d["default_cycle"]["vcycles_per_s"] > 0


@pytest.mark.parametrize("rnd", ROUNDS)
def test_pmc_traffic_is_tied_to_the_kernel_sources_it_was_measured_on(rnd):
    p = _load(rnd + "_pmc_plane_down.json")
    assert re.fullmatch(r"[0-9a-f]{16}", p["kernel_src_sha"]) and re.fullmatch(r"[0-9a-f]{40}", p["git_head_at_collection"])
    assert p["traffic_bytes"] == p["read_bytes"] + p["write_bytes"]
    assert abs(p["read_bytes"] - 2 * p["fetch_size_KiB"] * 1024) < 1024          # gfx950: reads = 2 x FETCH_SIZE
    assert p["traffic_bytes"] >= p["bytes_per_launch"]                           # over-fetch, never under
    h = bench.kernel_source_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", h) and h == bench.kernel_source_hash()
    if rnd == LATEST:
        # the round's final collection was made on the sources as they are committed: the profiles are this build's
        assert h == p["kernel_src_sha"], "openmg_amd/csrc changed after the round's profiles were collected"
        q = _load(rnd + "_pmc_s27_sweep.json")
        assert q["kernel_src_sha"] == h and q["traffic_bytes"] == q["read_bytes"] + q["write_bytes"]
    d = _load(rnd + "_bench.json")
    if d["roofline"]["traffic"] is not None:                                     # only a run of the profiled build may carry it
        # (the line quotes the PMC passes that were committed when it ran; the profile next to it may be a
        # later collection on the same sources: FETCH_SIZE varies by a few hundred bytes from pass to pass)
        assert d["config"]["kernel_src_sha"] == p["kernel_src_sha"]
        assert abs(d["roofline"]["traffic"] - p["traffic_bytes"]) <= 1e-3 * p["traffic_bytes"]
        assert "NOT measured in this run" in d["roofline"]["traffic_source"]
