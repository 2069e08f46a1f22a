"""GPU: bench.py's output contract at N = 1 and, rehearsed with two ranks on this one GPU (gloo, host staging: RCCL
wants one device per rank), at N = 2 -- the launch line is the driver's (`python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N`).  A small tower and gallery: this checks the JSON
line, the sharded leg and the rank-0-only fields, not speed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--variant", "PE-Tiny-T14-56", "--batch", "8", "--gallery", "60000", "--steps", "2", "--warmup", "1",
         "--search-queries", "600", "--no-cpu-baseline", "--ingest-images", "24"]


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def _check_contract(d, n):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == n and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "images/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "bf16"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 8 * n / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]      # whole-job images / max-over-ranks time
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9
    sq = d["search_query_batch"]
    assert sq["queries"] == 600 and sq["gallery_rows"] == 60000 and sq["shard_rows"] == 60000 // n
    assert sq["sharded_ms"] > 0


def test_bench_contract_one_gpu(dev):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _json_line(p.stdout)
    _check_contract(d, 1)
    assert d["ingest"]["ok"] and d["ingest"]["images"] == 24 and d["ingest"]["images_per_s"] > 0, d["ingest"]     # the reported ingest leg
    # the box-speed probes run right before the timed region (profiles/calibration_reference.json: the constants they are held against)
    c = d["calibration"]
    assert 500.0 < c["mfma_probe_tflops"] < 2600.0 and 2.0 < c["hbm_copy_tbs"] < 8.1, c
    assert c["attention_ms_per_launch"] > 0
    if c.get("reference"):
        assert d["value_at_reference_box"] == pytest.approx(d["value"] / c["mfma_probe_vs_reference"], rel=1e-9)
    assert d["certificate"]["uncertified_queries_last_step"] >= 0 and "uncertified_queries" in d["search_query_batch"]


def test_bench_contract_two_ranks_on_one_gpu(dev):
    port = 29700 + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-gpu", "--backend", "gloo"] + SMALL
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _json_line(p.stdout)
    _check_contract(d, 2)
    sq = d["search_query_batch"]
    assert sq["one_gpu_ms_same_process"] > 0 and sq["speedup_vs_1gpu_model"] > 0      # rank 0's 1-GPU run of the same search
    ab = sq["allgather_budget"]                                                          # what the >= 6x target leaves for the exchanges
    assert sq["allgather_budget_ms"] == ab["allgather_budget_ms"] == pytest.approx(ab["allowed_ms_per_search"] - ab["per_rank_compute_ms"])
    assert ab["exchange_ms_per_search_measured"] > 0 and ab["per_rank_compute_ms"] < sq["sharded_ms"]
    assert "cpu_baseline" not in d and "ingest" not in d                                 # N = 1 only


def test_bench_starts_its_own_ranks_without_a_launcher(dev):
    """`python bench.py --gpus 2 ...` with no torch.distributed.run around it (the way the driver starts N = 1): the
    parent starts the two rank processes itself before anything touches the device, relays rank 0's line and its status."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-gpu", "--backend", "gloo"] + SMALL
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _json_line(p.stdout)
    _check_contract(d, 2)
    sq = d["search_query_batch"]
    assert len(sq["sharded_ms_per_rank"]) == 2 and max(sq["sharded_ms_per_rank"]) == pytest.approx(sq["sharded_ms"], rel=1e-3)
    assert sq["one_gpu_ms_same_process"] > 0 and sq["speedup_vs_1gpu_model"] > 0
    c = d["collective"]
    assert c["ranks"] == 2 and c["backend"] == "gloo" and c["rccl_ranks"] == 0 and c["devices"] == [0, 0] and c["one_gpu_rehearsal"]
