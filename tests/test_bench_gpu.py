"""bench.py contract on a real GPU: one JSON line with the required keys, and the N>1 code path (RCCL init, barrier,
max-over-ranks, the one collective on the fused tally buffer) rehearsed with a single rank under torch.distributed.run."""
import json
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
SMALL = ["--steps", "2", "--warmup", "1", "--events-per-step", "256", "--lanes", "131072", "--completion-photons", "32", "--no-other-configs"]


def last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, "--cpu-photons", "8"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    d = last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["value"] > 1e8
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["reference_events_per_s_per_core"] == 2.8e6
    # the other half of BASELINE.json's metric and the run-to-completion leg (32 photons here: the numbers only have to be there and consistent)
    assert d["spectrum"]["fixture"] == "tests/golden/c2_tau6.npz" and d["spectrum"]["gpu_photons"] == 32 and d["spectrum_rms_vs_ref"] > 0
    assert abs(d["spectrum"]["mean_scatterings_gpu"] / d["spectrum"]["mean_scatterings_ref"] - 1.0) < 0.6      # (32 histories: 15 % per sigma)
    rc = d["run_to_completion"]
    assert rc["photons"] == 32 and rc["events"] > 1e7 and rc["lane_compactions"] >= 0 and abs(rc["value"] - rc["events"] / rc["seconds"]) < 1e-6 * rc["value"]


def test_bench_distributed_path_single_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", *SMALL, "--force-dist", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    d = last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 1e8 and d["config"]["tally_reduce_ms"] > 0
    assert d["run_to_completion"]["photons"] == 32       # the reduce of the completion leg went through the library's RCCL communicator


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    """--gpus N must equal the ranks the launcher started (the driver's scaling runs rely on n_gpus being what RCCL saw)"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL, "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "--gpus 2" in (r.stdout + r.stderr)
