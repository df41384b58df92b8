"""The driver parses the LAST stdout line of bench.py.  Round 5's line had grown to 21.7 KB and the driver recorded
`parsed: null`: the whole round's headline, roofline and cpu_baseline went unrecorded.  These tests size the line from canned
section results (the shapes bench.py's sections return), without a GPU."""
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _roof(point=None):
    r = {"kernel": "kp_gram3_kernel<6,3,false>", "bound": "mfma", "ms": 0.39703312516212463, "achieved": 54.16172766748127, "peak": 78.6,
         "unit": "TFLOP/s", "frac": 0.6890805046753343, "executed_flop_per_launch": 21504000000.0,
         "dense_equivalent_flop_per_launch": 33902400000.0, "dense_equivalent_achieved": 85.38934877576344,
         "dense_equivalent_frac": 1.0863784831522068, "note": "n" * 300}
    if point:
        r["point"] = point
    return r


def canned(n_gpus=1, bloat=1):
    """A result dict at least as large as round 5's (`bloat` multiplies the free-text and the per-point tables)."""
    res = {
        "metric": "EDMD snapshot-pairs/sec (bilinear fit, 3-link arm, poly-3)", "value": 238114103.03890663 * n_gpus,
        "unit": "snapshot-pairs/s", "n_gpus": n_gpus, "steps": 20, "warmup": 5, "ms_per_step": 0.41996672487584874,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "bilinear Koopman fit, poly degree 3, 100000 synthetic snapshot pairs per GPU, N=84, W=336 (BASELINE configs[1])",
                   "snapshots_per_gpu": 100000, "W": 336,
                   "parallelism": f"{n_gpus} rank(s), independent fits per rank, one RCCL all-gather of the K matrices at the end",
                   "comm": "rccl" if n_gpus > 1 else "local"},
        "host_enqueue_ms_per_step": 0.0126, "fit_latency_ms": 0.7505, "fit_with_K_fetched_ms": 0.8799,
        "h2d": {"upload_ms": 0.64, "note": "x" * 230 * bloat},
        "kernel_ms": {"gram": 0.397, "gram_launches_averaged": 50, "gram_reduce": 0.0153, "solve": 0.845},
        "untimed_prewarm_launches": {"pipelined_fits": 256, "gram_only": 32, "why": "y" * 160 * bloat},
        "roofline": dict(_roof(), traffic=91307645.87453875, algorithmic_bytes_per_launch=12000000.0, hbm_algorithmic_GBs=30.22,
                         pmc={"executed_flop_per_launch": 21504000000.0, "mfma_busy_cycles_per_instr": 16.0, "source": "profiles/x.json"}),
        "mpc": {"single_steps_per_s": 6392.78, "single_us_per_step": 156.4, "single_kernel_us": 146.07, "single_solved": 300,
                "closed_loop_steps_per_s": 19820.3, "closed_loop_us_per_step": 50.45, "closed_loop_kernel_us": 32.93,
                "batch_problems_per_s": 2507209.1, "batch": 4096, "workload": "w" * 120 * bloat, "reference_recorded": "r" * 90},
        "mpc_arm_blockM": {"steps": 300, "steps_per_s": 14424.6, "controller_us_per_step": 42.05, "get_koopman_end_to_end_ms": 0.288,
                           "workload": "w" * 150 * bloat},
        "width_points": {f"W{w}": {"W": w, "gram_ms": 0.2, "roofline": _roof(f"W{w}")} for w in (200, 136, 120, 108)},
        "snapshot_count_points": {f"Ns{n}": {"snapshots": n, "ms_per_fit": 0.094, "gram_ms": 0.069, "roofline": _roof()}
                                  for n in (11999, 1000000, 10000000)},
        "wide_dictionaries": {f"p{i}": {"W": 738, "ms_per_fit": 1.36, "gram_ms": 0.64, "solve_ms": 0.67, "roofline": _roof(), "note": "z" * 300}
                              for i in range(4 * bloat)},
        "rank_deficient_fit": {"bilinear_poly3": {"W": 336, "rank": 252, "ms_per_fit": 0.62}},
        "lasso_grid": {"values": 64, "seconds": 0.0039, "values_per_s": 16365.9, "n_gpus": n_gpus, "device_ms_per_rank": [2.3] * n_gpus,
                       "gather": "g" * 130, "workload": "w" * 150 * bloat, "ill_conditioned": {"W": 112, "note": "i" * 200}},
        "rand_sweep": {"systems": 1024, "seconds": 0.0173, "systems_per_s": 59296.0, "n_gpus": n_gpus, "workload": "w" * 180 * bloat},
        "kernels": [_roof(f"k{i}") for i in range(10 * bloat)],
        "one_caller": {"device_ids": list(range(n_gpus)), "host": "h" * 90,
                       "lasso_grid": {"values": 64, "values_per_s": 13207.8, "per_device_ms": {"upload": [0.32] * n_gpus}},
                       "snapshot_sharded_fit": {"Ns100000": {"pairs_per_s": 7.5e7, "note": "n" * 200 * bloat}},
                       "rand_sweep": {"systems": 1024, "systems_per_s": 136294.8}, "mpc_batch": {"problems": 4096}},
        "cpu_baseline": {"value": 58363.13, "unit": "snapshot-pairs/s", "cores": 32, "kind": "port", "sample": "s" * 160 * bloat,
                         "one_thread": {"value": 9510.3, "cores": 1, "sample": "t" * 60},
                         "mpc": {"value": 387.9, "unit": "MPC steps/s", "cores": 1, "kind": "port", "sample": "u" * 60}},
    }
    if n_gpus > 1:
        res["snapshot_sharded_fit"] = {"Ns100000": {"pairs_per_s": 1.1e8, "exchange": "e" * 80}, "Ns10000000": {"pairs_per_s": 1.9e9},
                                       "scaling": "strong"}
        del res["mpc"], res["mpc_arm_blockM"], res["cpu_baseline"]
    return res


def test_the_canned_result_is_as_large_as_the_line_the_driver_could_not_parse():
    assert len(json.dumps(canned())) > 15000


def test_compact_line_is_small_and_carries_the_contract():
    for n in (1, 8):
        for bloat in (1, 8):
            line = bench.compact_line(canned(n, bloat))
            assert "\n" not in line and len(line) < bench.COMPACT_LINE_LIMIT <= 4000, (n, bloat, len(line))
            d = json.loads(line)
            for k in REQUIRED:
                if n > 1 and k == "cpu_baseline":          # rank 0 at N = 1 only (the contract)
                    continue
                assert k in d, k
            assert d["n_gpus"] == n and d["steps"] == 20 and d["warmup"] == 5
            assert "workload" in d["config"] and "model" not in d["config"]
            rf = d["roofline"]
            for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
                assert k in rf, k
            assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
            if n == 1:
                for k in ("value", "unit", "cores", "kind", "sample"):
                    assert k in d["cpu_baseline"], k


def test_compact_line_keeps_the_headline_digits():
    res = canned()
    d = json.loads(bench.compact_line(res))
    assert abs(d["value"] / res["value"] - 1) < 1e-5 and abs(d["ms_per_step"] / res["ms_per_step"] - 1) < 1e-5


def test_emit_result_prints_exactly_one_stdout_line_and_the_detail_elsewhere(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        bench.emit_result(canned())
    lines = out.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4000 and json.loads(lines[0])["roofline"]["frac"] > 0
    full = json.loads(err.getvalue())
    assert "kernels" in full and "wide_dictionaries" in full
    assert json.load(open(tmp_path / "bench_detail.json")) == full
    assert json.load(open(tmp_path / "gpurun_out" / "bench_detail.json")) == full


def test_compact_line_of_the_committed_full_record():
    """The real thing: the full record of the round's own run (profiles/r06_bench_detail.json, written by emit_result on the GPU box)
    through compact_line - the size the driver will see, and the headline digits intact."""
    path = os.path.join(ROOT, "profiles", "r06_bench_detail.json")
    if not os.path.exists(path):
        import pytest
        pytest.skip("no committed record")
    res = json.loads(open(path).read())
    line = bench.compact_line(res)
    assert len(line) < bench.COMPACT_LINE_LIMIT
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k
    assert abs(d["value"] / res["value"] - 1) < 1e-5 and d["roofline"]["bound"] == "mfma" and 0.5 < d["roofline"]["frac"] < 1.0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
