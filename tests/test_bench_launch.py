"""CPU: bench.py's own launcher.  `python bench.py --gpus N` (N > 1) without torch.distributed.run around it must start
the N rank processes as CHILDREN, before torch or the HIP library is imported in the parent (a process that has touched
the GPU must never exec or fork GPU work on this pool), hand them the same arguments, and exit with their status."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run_parent(tmp_path, argv, extra_env=None):
    """Run bench.py as __main__ with subprocess.Popen replaced by a recorder (sitecustomize in a scratch directory)."""
    (tmp_path / "sitecustomize.py").write_text(textwrap.dedent("""
        import json, os, subprocess, sys
        class _P:
            def __init__(self, cmd, **kw):
                mods = [m for m in ("torch", "reverso_amd", "ctypes") if m in sys.modules]
                with open(os.environ["REC"], "w") as f:
                    json.dump({"cmd": cmd, "env": {k: kw.get("env", {}).get(k) for k in
                               ("HSA_ENABLE_IPC_MODE_LEGACY", "WORLD_SIZE")}, "loaded": mods}, f)
            def wait(self, timeout=None):
                return 7
        subprocess.Popen = _P
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PYTHONPATH"] = str(tmp_path)
    env["REC"] = str(tmp_path / "rec.json")
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH] + argv, env=env, capture_output=True, text=True, timeout=300), tmp_path / "rec.json"


def test_parent_starts_rank_processes_before_touching_torch(tmp_path):
    import json
    p, rec = _run_parent(tmp_path, ["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert p.returncode == 7, p.stderr[-2000:]                      # the children's status is the parent's
    r = json.loads(rec.read_text())
    cmd = r["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert r["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and r["env"]["WORLD_SIZE"] is None
    assert r["loaded"] == [], r["loaded"]                           # neither torch nor the HIP library in the parent


def test_gpus_equals_form_and_launcher_environment(tmp_path):
    import json
    p, rec = _run_parent(tmp_path, ["--gpus=2"])
    assert p.returncode == 7 and json.loads(rec.read_text())["cmd"][-1] == "--gpus=2"
    # under a launcher (WORLD_SIZE set) the parent starts nothing: it IS a rank
    rec.unlink()
    p, rec = _run_parent(tmp_path, ["--gpus", "2", "--help"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert not rec.exists() and p.returncode == 0 and "--gpus" in p.stdout


def test_parent_guard_ends_hung_ranks_and_exits_nonzero(tmp_path):
    """A rank that never finishes (a collective that does not complete) must not hang the caller: after REVO_BENCH_WALL_S
    the parent ends the process group IT started -- a fresh child was started, nothing is re-executed -- and exits 124."""
    import json
    import time
    (tmp_path / "sitecustomize.py").write_text(textwrap.dedent("""
        import json, os, subprocess, sys
        _real = subprocess.Popen
        def _P(cmd, **kw):
            if "torch.distributed.run" not in cmd:
                return _real(cmd, **kw)
            p = _real([sys.executable, "-S", "-c", "import time; time.sleep(600)"], **kw)      # the 'ranks' hang
            with open(os.environ["REC"], "w") as f:
                json.dump({"pid": p.pid, "new_session": bool(kw.get("start_new_session"))}, f)
            return p
        subprocess.Popen = _P
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PYTHONPATH=str(tmp_path), REC=str(tmp_path / "rec.json"), REVO_BENCH_WALL_S="2")
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 124, (p.returncode, p.stderr[-1000:])
    assert time.time() - t0 < 60 and "did not finish within" in p.stderr
    rec = json.loads((tmp_path / "rec.json").read_text())
    assert rec["new_session"]                                       # the children have a process group of their own
    for _ in range(50):
        try:
            os.kill(rec["pid"], 0)
        except ProcessLookupError:
            break
        time.sleep(0.1)
    else:
        raise AssertionError("the hung child is still alive")


def test_rccl_ranks_without_a_device_each_fail_in_one_line(tmp_path):
    """Under a launcher with the nccl backend and fewer visible devices than ranks (here: none) a rank stops with one
    readable line before any process group exists -- it does not double ranks up on a device for RCCL to reject."""
    env = {k: v for k, v in os.environ.items()}
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    last = [ln for ln in p.stderr.strip().splitlines() if ln.strip()][-1]
    assert "needs one GPU per rank" in last and "--one-gpu --backend gloo" in last, p.stderr[-1500:]
    assert "Traceback" not in p.stderr


def test_process_group_timeout_and_ipc_mode_are_set_on_both_launch_paths():
    """bench.py sets the dmabuf IPC mode at import (before torch): a torchrun-started rank gets it like a self-launched one;
    the process group is created with a finite timeout."""
    src = open(BENCH).read()
    head = src[: src.index("import torch")]
    assert 'os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")' in head
    assert src.count("init_process_group(") == 2 and src.count("timeout=tmo") == 2


def test_allgather_budget_field():
    """`search_query_batch.allgather_budget_ms` of an N > 1 line: what the >= 6x target of north_star leaves for the exchanges
    of one search, from the run's own numbers.  The rehearsed 8-shard stage times of profiles/r05_sharded_stage_8.json as the
    worked example: 18.38 ms on one GPU, 2.73 ms per rank of which 0.06 assumed for the exchange."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    b = bench.allgather_budget(18.38, 2.73, 0.06)
    assert abs(b["allowed_ms_per_search"] - 18.38 / 6) < 1e-12 and abs(b["per_rank_compute_ms"] - 2.67) < 1e-12
    assert abs(b["allgather_budget_ms"] - (18.38 / 6 - 2.67)) < 1e-12 and 0.39 < b["allgather_budget_ms"] < 0.40
    assert b["meets_target"] and abs(b["speedup_measured"] - 18.38 / 2.73) < 1e-12
    slow = bench.allgather_budget(18.38, 3.5, 0.9)
    assert not slow["meets_target"] and slow["allgather_budget_ms"] > 0          # compute fits, the exchange is what misses
    worse = bench.allgather_budget(18.38, 3.5, 0.1)
    assert worse["allgather_budget_ms"] < 0                                      # the compute alone misses the target
