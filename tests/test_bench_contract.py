"""The driver-facing contract of bench.py: one JSON line with the fields the round driver reads, the roofline and the
CPU-baseline objects.  Runs the real benchmark (few steps) in a child process on the GPU box."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extras"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "solves/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] * 1e-3 - 512) < 1.0     # 512 windows per step
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert "traffic_source" in r and "committed" in r["traffic_source"]          # the PMC bytes are NOT collected in this run
    # the dominant kernel is a single kernel by total time (the three-kernel step competes with its longest member)
    assert r["kernel"] in r["per_kernel"] and "dominant_by" in r
    assert r["kernel"] == "k_lin" or r["per_kernel"]["k_lin"]["ms_per_solve"] < r["per_kernel"]["k_step"]["ms_per_solve"]
    assert "binding" in r and "fp64" in r
    if r["fp64"] is not None:
        assert r["fp64"]["unit"] == "TFLOP/s" and r["fp64"]["peak"] == 78.6 and 0 < r["fp64"]["frac"] < 1
        assert r["binding"] in ("hbm", "fp64")
    # what the line says about the work it did is what the device counted
    cfg = d["config"]
    acc_hist = cfg["accepted_histogram"]
    mean_acc = sum(k * n for k, n in enumerate(acc_hist)) / 512.0
    assert abs(cfg["mean_accepted_steps"] - mean_acc) < 1e-9
    # every accepted step but the last one of a window that stops there is followed by a linearisation; the first
    # linearisation precedes all steps: sum(active k_lin) / nW - 1 <= mean accepted steps <= sum(active k_lin) / nW
    lin = d["linearisations_per_solve"]
    assert lin - 1.0 - 1e-9 <= cfg["mean_accepted_steps"] <= lin + 1e-9, (lin, cfg["mean_accepted_steps"])
    assert cfg["mean_accepted_steps"] + cfg["mean_rejected_steps"] == pytest.approx(cfg["mean_tr_iterations"])
    k_lin_active = sum(l["active"][0] for l in d["launches"] if l["kernel"] == "k_lin")
    assert abs(k_lin_active / 512.0 - lin) < 1e-6
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    # the GPU path is checked against the oracle inside the run as well
    assert d["parity"]["max_dp_m"] < 1e-4 and d["parity"]["max_dr_rad"] < 1e-6


def test_bench_single_rank_rccl_path():
    """VPL_FORCE_DIST=1: one rank, but through torch.distributed (backend nccl = RCCL): the block's states are packed on the
    device and all-gathered device to device, the timing / parity summaries all-reduced -- the code the N > 1 launches run."""
    env = dict(os.environ, VPL_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-extras",
                          "--cpu-windows", "8"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["parity"]["max_dp_m"] < 1e-4 and d["parity"]["allreduce_max"][0] == d["parity"]["max_dp_m"]
