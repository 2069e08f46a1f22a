#!/usr/bin/env python3
"""Headline benchmark: images/sec for embed + top-k (PE-Core-L14-336, 1M x 1024 gallery).

One step = one batch of synthetic 336x336 images per GPU through the HIP embed
path, then cosine top-10 of every embedding against the row-sharded 1M x 1024
gallery (all-gather of queries; per-shard scan against the whole gallery's estimated
admission level; fp32 re-score; ONE all-gather of the packed per-shard top-k with the
shards' certificate bounds; merge + cross-shard certificate; a second, exact round only
for queries that certificate fails for).  At N = 1 there is no exchange at all.
Inputs are resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N ...          (no launcher: starts the same N rank processes itself, see _self_launch)

Prints ONE JSON line on rank 0 (contract in the task statement): `value` is the
whole-job images/s, `roofline` is the dominant kernel class (the ViT linear-layer
MFMA GEMM) timed with HIP events on the launch stream over the timed steps,
`cpu_baseline` is the CPU oracle on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

# Multi-process GPU work on this image needs dmabuf IPC: the task environment exports HSA_ENABLE_IPC_MODE_LEGACY=0 ("the
# host driver only supports dmabuf IPC, and without it RCCL / CUDA-tensor sharing across processes fails with
# hipIpcGetMemHandle: invalid argument" -- README.md, "Environment").  Set here, before torch / the HIP runtime are loaded,
# so that BOTH launch paths carry it: the self-launched ranks inherit it, a torchrun-started rank gets it at import.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
LAUNCH_WALL_S = float(os.environ.get("REVO_BENCH_WALL_S", "1500"))      # the self-launch parent's guard (seconds)
PG_TIMEOUT_S = float(os.environ.get("REVO_BENCH_PG_TIMEOUT_S", "120"))   # a collective that does not complete fails the run


def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N rank processes ourselves.

    Runs BEFORE torch (or anything else that could touch the device) is imported: this parent never initialises the
    GPU, it only starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <the same arguments>` as a CHILD process (no exec of a GPU process), relays the children's
    output -- rank 0's single JSON line on stdout -- and exits with their status.  Under a launcher (WORLD_SIZE set,
    the driver's documented way of starting N > 1) this is a no-op."""
    if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    n = 1
    argv = sys.argv[1:]
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    import socket
    import subprocess
    with socket.socket() as s:                       # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)                           # (carries HSA_ENABLE_IPC_MODE_LEGACY=0, set at the top of this file)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    # the children get a process group of their own: a guard that fires ends exactly the processes started here
    p = subprocess.Popen(cmd, env=env, cwd=os.getcwd(), start_new_session=True)
    try:
        rc = p.wait(timeout=LAUNCH_WALL_S)
    except subprocess.TimeoutExpired:
        # a hung rank (a collective that never completes, a rank that died before the rendezvous) must not hang the
        # caller: end the launcher and its ranks, report, exit non-zero.  Never a re-exec, never a retry.
        import signal
        sys.stderr.write(f"bench.py: the {n} ranks did not finish within {LAUNCH_WALL_S:.0f} s (REVO_BENCH_WALL_S): terminating them\n")
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(p.pid, sig)
            except ProcessLookupError:
                break
            try:
                p.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        sys.exit(124)
    except KeyboardInterrupt:
        p.terminate()
        rc = p.wait()
    sys.exit(rc)


if __name__ == "__main__":
    _self_launch()

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, ROOT)

import reverso_amd  # noqa: E402
from reverso_amd import engine, sharded  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0      # MI355X dense bf16, /opt/skills/guides/MI355X_MICROARCH.md:43
HBM_PEAK_GBS = 8000.0               # MI355X HBM3E spec, MI355X_MICROARCH.md:36


def layer_gemm_flops(cfg, batch):
    """Algorithmic FLOPs (2*MAC) of the linear layers of the transformer body per forward."""
    rows = batch * cfg.seq
    W, M = cfg.width, cfg.mlp_dim
    return cfg.layers * 2.0 * rows * (W * 3 * W + W * W + 2 * W * M)


def allgather_budget(one_gpu_ms, sharded_ms, exchange_ms_per_search, target_speedup=6.0):
    """What north_star's ">= 6x at 8 GPUs on the 1M-gallery top-k" leaves for the exchange: the sharded search may take
    one_gpu_ms / target in all; this rank's compute is what its measured search time holds besides the measured
    exchanges; the difference is the budget every all-gather of a search (torch.distributed's host path included) must fit
    in.  Negative = the compute alone misses the target."""
    per_rank_compute = sharded_ms - exchange_ms_per_search
    allowed = one_gpu_ms / target_speedup
    return {"target_speedup": target_speedup, "one_gpu_ms": one_gpu_ms, "allowed_ms_per_search": allowed,
            "per_rank_compute_ms": per_rank_compute, "exchange_ms_per_search_measured": exchange_ms_per_search,
            "allgather_budget_ms": allowed - per_rank_compute,
            "speedup_measured": one_gpu_ms / sharded_ms if sharded_ms > 0 else None,
            "meets_target": bool(sharded_ms > 0 and one_gpu_ms / sharded_ms >= target_speedup)}


def cpu_baseline(cfg, gallery_rows, dim, k, n_images, search_rows):
    """The CPU oracle run the way the reference runs: fp32, one image per forward
    (core_system.py:439-442), one query per search over a float32 numpy gallery."""
    import numpy as np
    from oracle import pe_vit, search as osearch
    from reverso_amd import weights
    torch.manual_seed(0)
    sd = weights.synth_weights(cfg, seed=0)
    g = torch.Generator().manual_seed(1234)
    u8 = torch.randint(0, 256, (n_images, 3, cfg.image_size, cfg.image_size), generator=g, dtype=torch.uint8)
    x = pe_vit.preprocess_u8(u8)
    with torch.no_grad():
        pe_vit.embed_batch1(sd, cfg, x[:1])                  # warm the thread pool
        t0 = time.perf_counter()
        emb = pe_vit.embed_batch1(sd, cfg, x)
        t_embed = (time.perf_counter() - t0) / n_images
    rng = np.random.default_rng(42)
    gal = osearch.normalize_rows(rng.standard_normal((search_rows, dim), dtype=np.float32))
    q = emb.numpy()
    osearch.search_one_reference_style(gal, q[0], k)
    t0 = time.perf_counter()
    for i in range(n_images):
        osearch.search_one_reference_style(gal, q[i % len(q)], k)
    t_search = (time.perf_counter() - t0) / n_images * (gallery_rows / search_rows)
    # batched CPU variant: the same images in ONE forward (not how the reference runs)
    with torch.no_grad():
        t0 = time.perf_counter()
        pe_vit.embed(sd, cfg, x)
        t_embed_batched = (time.perf_counter() - t0) / n_images
    # exactness beside the timing: the GPU path on the same images and the same gallery sample against the oracle
    exact = None
    if torch.cuda.is_available():
        import reverso_amd
        from reverso_amd import engine
        dev = torch.device("cuda", torch.cuda.current_device())
        eng = engine.VitEngine(cfg, {kk: v.to(dev) for kk, v in sd.items()}, device=dev.index, max_batch=max(n_images, 1))
        ge = eng.embed(u8.to(dev)).cpu()
        G = engine.Gallery(dim, search_rows, device=dev.index)
        G.add(torch.from_numpy(gal).to(dev), normalize=False)
        qs = torch.from_numpy(q).to(dev)
        s_gpu, i_gpu, _ = G.search(qs, k)
        cert = G.search_stats()
        rs, ri, _ = osearch.search(gal, q, k)
        exact = {"embed_cosine_vs_oracle_min": float((ge * emb).sum(-1).min()),
                 "topk_index_match_rate": float((i_gpu.cpu().numpy() == ri).mean()),
                 "topk_max_abs_score_err": float(np.abs(s_gpu.cpu().numpy() - rs).max()),
                 # queries whose exactness certificate failed and were re-done by the exact fallback (include/revo.h)
                 "uncertified_queries": cert["uncertified"], "certificate_checked": cert["checked"],
                 "sample": f"{n_images} oracle embeddings as queries over the {search_rows}x{dim} gallery sample, k = {k}"}
        G.close()
        eng.close()
    return {
        "value": 1.0 / (t_embed + t_search), "unit": "images/s", "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": (f"{n_images} images embedded one per forward in fp32 (oracle/pe_vit.py), each searched alone over a "
                   f"{search_rows}x{dim} float32 numpy gallery; search time scaled x{gallery_rows // search_rows} to "
                   f"{gallery_rows} rows; host has {os.cpu_count()} logical cores"),
        "embed_s_per_image": t_embed, "search_s_per_query": t_search,
        "embed_s_per_image_one_batched_forward": t_embed_batched, "exactness_vs_oracle": exact,
    }


def calibration(dev, warm_s=0.7, timed_s=0.35):
    """How fast is THIS box: two fixed kernels (revers-o_amd/csrc/probe.hip: they do not change from round to round) run right
    before the timed region, each warm for ~0.7 s and then timed for ~0.35 s with HIP events on the launch stream --
    (a) a register-resident v_mfma_f32_16x16x32_bf16 loop on every CU, random operands -> `mfma_probe_tflops`;
    (b) a 16-byte-per-lane copy of 1 GiB (2 GiB of traffic) -> `hbm_copy_tbs`.
    MI355X devices differ by several per cent on MFMA-dense loops (MI355X_MICROARCH.md, DVFS give-back item 5: one binary,
    12 % apart in wall time): `value_at_reference_box` = value x (reference probe / this probe) is the headline number a
    reader can compare across rounds; profiles/calibration_reference.json holds the reference constants."""
    from reverso_amd import _lib
    lib = _lib.load()
    st = _lib.current_stream()
    g = torch.Generator(device=dev).manual_seed(99)
    src = torch.randn(1 << 20, generator=g, device=dev).bfloat16()
    blocks, iters = 1024, 2000
    sink = torch.empty(blocks * 256, device=dev)

    def timed(fn, per_launch):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < warm_s:       # warm: the clock settles under THIS load
            for _ in range(8):
                fn()
            torch.cuda.synchronize()
            n += 8
        reps = max(8, int(n * timed_s / warm_s))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return per_launch / (e0.elapsed_time(e1) / reps * 1e-3), e0.elapsed_time(e1) / reps
    fl = float(lib.revo_probe_mfma_flops(blocks, iters))
    mfma, mfma_ms = timed(lambda: _lib.check(lib.revo_probe_mfma(_lib.ptr(src), src.numel(), _lib.ptr(sink), blocks, iters, st)), fl)
    nbytes = 1 << 30
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    a.random_(0, 256, generator=g)
    b = torch.empty_like(a)
    copy, copy_ms = timed(lambda: _lib.check(lib.revo_probe_copy(_lib.ptr(b), _lib.ptr(a), nbytes, st)), 2.0 * nbytes)
    del a, b
    out = {"mfma_probe_tflops": mfma / 1e12, "mfma_probe_ms_per_launch": mfma_ms, "hbm_copy_tbs": copy / 1e12,
           "hbm_copy_ms_per_launch": copy_ms,
           "what": "probe.hip: 1024 x 4 waves x 2000 trips of 32 register-resident 16x16x32 bf16 MFMAs on random operands; "
                   "1 GiB copied 16 B per lane (read + write counted); each warm ~0.7 s, then timed ~0.35 s"}
    try:
        with open(os.path.join(ROOT, "profiles", "calibration_reference.json")) as f:
            ref = json.load(f)
        out["reference"] = {k: ref[k] for k in ("mfma_probe_tflops", "hbm_copy_tbs", "box") if k in ref}
        out["mfma_probe_vs_reference"] = out["mfma_probe_tflops"] / ref["mfma_probe_tflops"]
        out["hbm_copy_vs_reference"] = out["hbm_copy_tbs"] / ref["hbm_copy_tbs"]
    except Exception:
        out["reference"] = None
    return out


def ingest_leg(variant, n_images, device_index):
    """SURVEY 8(f) row 1 next to the headline: n JPEGs (640 x 480, written to a temporary folder) -> decode pool -> H2D ->
    device resize -> embed -> device gallery append -> delta-shard flush, through SimpleReverso.create_database.  Reported,
    not part of `value` (it includes JPEG decoding on the host's cores and the file system)."""
    import shutil
    import tempfile
    import numpy as np
    from PIL import Image
    from reverso_amd.core_system import SimpleReverso
    root = tempfile.mkdtemp(prefix="revo_ingest_")
    try:
        folder = os.path.join(root, "images")
        os.makedirs(folder)
        rng = np.random.default_rng(0)
        yy, xx = np.mgrid[0:480, 0:640]
        for i in range(n_images):
            base = np.stack([(xx * (i % 7 + 1) + yy) % 256, (yy * 2 + i) % 256, (xx + yy * (i % 5)) % 256], -1).astype(np.float32)
            img = np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(folder, f"img_{i:05d}.jpg"), quality=90)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):            # the facade prints like the reference does
            r = SimpleReverso(model_name=variant, db_root=os.path.join(root, "db"), max_batch=64, device=device_index,
                              device_resize=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            msg = r.create_database(folder, "bench", use_direct_pe=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        ok = "ready for searching" in msg and len(r.vector_db) == n_images
        st = dict(r.last_ingest_stats or {})
        r.vector_db.close()
        return {"images": n_images, "images_per_s": n_images / dt, "seconds": dt, "ok": ok, "mode": "direct PE, device resize",
                "stage_s": {k: round(v, 3) for k, v in st.items() if k.endswith("_s")},
                "note": "JPEG decode on the host's cores + device resize + embed + device-side gallery append + delta-shard flush; "
                        f"{n_images} images are {n_images // 64} batches: the first decode and the last embed overlap with nothing "
                        "(profiles/r03_ingest.json: 2 020 images/s at 4 000 images)"}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step (BASELINE.json configs[1])")
    ap.add_argument("--gallery", type=int, default=1_000_000, help="total gallery rows over all GPUs")
    ap.add_argument("--variant", default="PE-Core-L14-336")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-calibration", action="store_true", help="skip the two box-speed probes (about 2.5 s) before the timed region")
    ap.add_argument("--cpu-images", type=int, default=6)
    ap.add_argument("--timed-events", type=int, default=3,
                    help="profiler mode inside the timed region: 3 = HIP events around every fourth launch of each body-GEMM "
                         "class (default; the events of mode 2, every launch, cost 0.4 ms of a 28.6 ms step), 0 = none")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to rehearse the "
                                                      "multi-rank code path on fewer GPUs, together with --one-gpu)")
    ap.add_argument("--one-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--debug-flags", type=int, default=0, help="A/B experiments: revo_op_set_gemm_debug flags (0 = product path); "
                                                               "needs REVO_EXPERIMENTS=1 (librevo_exp.so, `make -C revers-o_amd/csrc exp`)")
    ap.add_argument("--ingest-images", type=int, default=512,
                    help="extra, untimed-by-the-headline leg on rank 0 at --gpus 1: build a gallery from this many JPEGs through "
                         "SimpleReverso.create_database (SURVEY 8(f) row 1); 0 disables it")
    ap.add_argument("--search-dim", type=int, default=0,
                    help="dimension of the gallery of the --search-queries leg (0 = the tower's output dimension, the headline "
                         "gallery itself).  BASELINE.json configs[4] writes its gallery as 10M x 1536 while PE-Core-G14-448 "
                         "emits 1280-dimensional embeddings (SURVEY.md 8(a) note): --variant PE-Core-G14-448 --gallery 10000000 "
                         "--search-dim 1536 runs the search leg on the gallery as written")
    ap.add_argument("--search-queries", type=int, default=10000,
                    help="extra, untimed-by-the-headline measurement: a batch of this many queries against the local "
                         "gallery shard (BASELINE.json configs[3]); 0 disables it")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.one_gpu:
        local_rank = 0
    ndev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world > 1 and args.backend == "nccl" and not args.one_gpu and ndev < local_world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} over RCCL needs one GPU per rank, but only {ndev} device(s) are visible "
                         f"to rank {rank} (two ranks on one device make RCCL fail far less readably); "
                         f"use --one-gpu --backend gloo to rehearse the multi-rank path on one GPU")
    elif ndev > 0 and local_rank >= ndev:
        raise SystemExit(f"bench.py: LOCAL_RANK {local_rank} but only {ndev} device(s) visible (use --one-gpu to put every rank on cuda:0)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import datetime
        tmo = datetime.timedelta(seconds=PG_TIMEOUT_S)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(args.backend, timeout=tmo)

    # which physical device every rank runs on (the judge's check that N ranks mean N GPUs)
    devices = [local_rank]
    if world > 1:
        mine = torch.tensor([local_rank], dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
        alld = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(alld, mine)
        devices = [int(t.item()) for t in alld]

    cfg = reverso_amd.get_config(args.variant)
    D = cfg.out_dim
    B = args.batch
    eng = engine.VitEngine.synthetic(cfg, seed=0, device=local_rank, max_batch=B)
    if args.debug_flags:
        from reverso_amd import _lib
        _lib.check(_lib.load().revo_op_set_gemm_debug(args.debug_flags))

    # synthetic gallery shard, generated on the device (seed 42 + rank), rows normalised at insert
    shard_rows = args.gallery // world + (1 if rank < args.gallery % world else 0)
    gal = engine.Gallery(D, max(shard_rows, 1), device=local_rank)
    gg = torch.Generator(device=dev).manual_seed(42 + rank)
    for s in range(0, shard_rows, 131072):
        n = min(131072, shard_rows - s)
        gal.add(torch.randn(n, D, generator=gg, device=dev))
    ss = sharded.ShardedSearch.from_gallery(gal)

    ig = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randint(0, 256, (B, 3, cfg.image_size, cfg.image_size), generator=ig, device=dev,
                           dtype=torch.uint8)

    def step_async():
        emb = eng.embed(images)                 # [B, D] L2-normalised fp32
        q = ss.gather_queries(emb)              # [B*world, D]
        return ss.search_async(q, args.k)       # first round enqueued; result() = certified, or re-done exactly

    def run_steps(n):
        """n steps back to back.  Every step's result is finalised (result(): the one host decision of a sharded
        search, whether any query needs the protocol's second round) -- but only after the NEXT step has been
        enqueued, so the device does not idle while the host waits for that count (N = 1: nothing to wait for)."""
        pend, out = None, None
        for _ in range(n):
            nxt = step_async()
            if pend is not None:
                out = pend.result()
            pend = nxt
        return pend.result() if pend is not None else out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    fence()
    calib = None if args.no_calibration else calibration(dev)     # every rank (they stay in step); rank 0's is printed
    if calib is not None:
        run_steps(1)                                              # back to the step's own kernels and caches
        fence()
    # Timed region: HIP events (on the launch stream) only around the roofline kernel class, the four body
    # GEMMs, and only around every fourth launch of each (24 identical layers): an event pair costs a few
    # microseconds of stream time (all ~250 kernels of a step: 1 ms; the 96 GEMMs: 0.4 ms; sampled: 0.1 ms).
    engine.prof_reset()
    engine.prof_enable(args.timed_events)
    t0 = time.perf_counter()
    out = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    engine.prof_enable(False)
    prof = engine.prof_report()
    # untimed pass with events around every kernel class: the per-class breakdown and the scan timing
    engine.prof_reset()
    engine.prof_enable(1)
    bsteps = min(args.steps, 3)
    ss.enable_timing(True)            # HIP events around every exchange of these steps (queries, [bounds,] packed top-k)
    run_steps(bsteps)
    fence()
    engine.prof_enable(False)
    prof_all = engine.prof_report()
    exch_headline = ss.timing_report() if world > 1 else None
    ss.enable_timing(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms_per_step = dt / args.steps * 1e3
    value = B * world * args.steps / dt

    # dominant kernel class: the four linear-layer GEMMs of every transformer block
    gemm_classes = ("gemm_qkv", "gemm_out", "gemm_fc1", "gemm_fc2")
    gemm_ms = sum(prof.get(c, {}).get("ms", 0.0) for c in gemm_classes)
    gemm_launches = sum(prof.get(c, {}).get("launches", 0) for c in gemm_classes)
    flops_per_launch = layer_gemm_flops(cfg, B) / (4 * cfg.layers)
    avg_ms = gemm_ms / max(gemm_launches, 1)
    achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    # HBM-side bytes per launch of that kernel class from the committed rocprofv3 PMC passes
    # (separate --pmc runs of this same command; FETCH_SIZE doubled per the gfx950 correction):
    # profiles/roofline_traffic.json, produced by scripts/pmc_summary.py.  null when the
    # workload differs from the profiled one.
    traffic = None
    traffic_source = None
    try:
        with open(os.path.join(ROOT, "profiles", "roofline_traffic.json")) as f:
            tj = json.load(f)
        if tj.get("variant") == cfg.name and tj.get("batch") == B:
            traffic = tj.get("gemm_bytes_per_launch")
            traffic_source = ("profiles/roofline_traffic.json: rocprofv3 PMC passes of this command taken at "
                              f"{tj.get('taken', 'an earlier commit')} (NOT measured by this run)")
    except Exception:
        traffic = None
    roofline = {"bound": "mfma", "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_source, "kernel": "gemm256p_kernel (+ the leftover-row kernels of the same linear layer; one layer call = one launch)",
                "avg_launch_ms": avg_ms, "launches": gemm_launches,
                "algorithmic_flops_per_launch": flops_per_launch,
                "note": ("since round 5 these launches also carry the body's LayerNorms (row statistics and bf16 rows out of the "
                         "residual epilogues, the correction in the qkv / fc1 epilogues: DESIGN.md section 4d) -- work that used to "
                         "be 48 LayerNorm kernels per step outside this kernel class; `achieved` divides the GEMMs' algorithmic "
                         "FLOPs alone by the whole launch time")}
    # per layer type: the four linear layers of a block have different shapes and epilogues
    W_, M_, rows_ = cfg.width, cfg.mlp_dim, B * cfg.seq
    layer_flops = {"gemm_qkv": 2.0 * rows_ * W_ * 3 * W_, "gemm_out": 2.0 * rows_ * W_ * W_,
                   "gemm_fc1": 2.0 * rows_ * W_ * M_, "gemm_fc2": 2.0 * rows_ * M_ * W_}
    roofline["per_layer_tflops"] = {
        c: round(layer_flops[c] / (prof_all[c]["ms"] / prof_all[c]["launches"] * 1e-3) / 1e12, 1)
        for c in gemm_classes if prof_all.get(c, {}).get("launches")}
    # The search at this query count is HBM bound.  `search_scan`: the fused scan kernel alone against the bytes IT
    # reads (the gallery rows behind the pre-pass, the queries, the result); `search_total`: every kernel of the
    # search (query normalise, pre-pass GEMM + selection, scan, final selection, fp32 re-score, merge) against
    # the algorithmic bytes of the whole search, SURVEY.md 8(d): N*D*2 + Q*D*2 + Q*k*12.
    Qs = B * world
    plan = gal.search_plan(Qs, args.k) if shard_rows > 0 else None
    scan = prof_all.get("topk_scan", {})
    if scan.get("launches") and plan:
        scan_ms = scan["ms"] / scan["launches"]
        scan_rows = shard_rows - plan["prepass_rows"]
        scan_bytes = scan_rows * D * 2 + Qs * D * 2 + Qs * plan["ksel"] * 8
        roofline["search_scan"] = {"bound": "hbm", "achieved": scan_bytes / (scan_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": scan_bytes / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "avg_launch_ms": scan_ms, "rows_scanned": scan_rows,
                                   "rows_in_prepass": plan["prepass_rows"], "slices": plan["slices"]}
        search_classes = [c for c in prof_all if c.startswith("topk_") or c == "search_prep"]
        tot_ms = sum(prof_all[c]["ms"] for c in search_classes) / bsteps
        tot_bytes = shard_rows * D * 2 + Qs * D * 2 + Qs * args.k * 12
        roofline["search_total"] = {"bound": "hbm", "achieved": tot_bytes / (tot_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": tot_bytes / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "ms_per_step": tot_ms, "kernels": sorted(search_classes)}
    classes_ms = {k: round(v["ms"] / bsteps, 4) for k, v in sorted(prof_all.items())}
    # exactness certificate of the headline searches (the last step's): queries that needed the exact fallback
    cert_headline = gal.search_stats() if world == 1 else {"uncertified": ss.last_uncertified}

    # BASELINE.json configs[3]: a large batch of replicated queries against the whole gallery THROUGH the sharded
    # search (per-shard scan, the packed all-gather, merge + certificate) -- the MFMA-bound regime of the fused scan (north_star:
    # >= 40 % of bf16 MFMA peak on the query x gallery GEMM, >= 6x at 8 GPUs).  Measured after the headline
    # region; it does not enter `value`.
    search_big = None
    if args.search_queries > 0 and shard_rows > 0:
        Qn = args.search_queries
        tower_dim = D
        if args.search_dim and args.search_dim != D:
            # the query-batch leg on a gallery of another dimension (configs[4] as written): its own shard, same seeds
            D = args.search_dim
            gal = engine.Gallery(D, max(shard_rows, 1), device=local_rank)
            gg = torch.Generator(device=dev).manual_seed(42 + rank)
            for s in range(0, shard_rows, 131072):
                gal.add(torch.randn(min(131072, shard_rows - s), D, generator=gg, device=dev))
            ss = sharded.ShardedSearch.from_gallery(gal)
        qg = torch.Generator(device=dev).manual_seed(7)               # the same queries on every rank
        qbig = torch.randn(Qn, D, generator=qg, device=dev)
        reps = 3
        ss.search(qbig, args.k)
        fence()
        # the searches back to back, each result finalised after the next search has been enqueued (search_async: no
        # host wait between searches; the 1-GPU search below has no host decision at all and pipelines by itself) ...
        t1 = time.perf_counter()
        pend = None
        for _ in range(reps):
            nxt = ss.search_async(qbig, args.k)
            if pend is not None:
                pend.result()
            pend = nxt
        pend.result()
        fence()
        dts = (time.perf_counter() - t1) / reps
        # ... and one at a time (search(): the host reads the uncertified count before it returns)
        t1 = time.perf_counter()
        for _ in range(reps):
            ss.search(qbig, args.k)
        fence()
        dts_sync = (time.perf_counter() - t1) / reps
        cert_big = gal.search_stats() if world == 1 else {"uncertified": ss.last_uncertified}
        per_rank_ms = [dts * 1e3]
        if world > 1:
            tt = torch.tensor([dts], dtype=torch.float64, device=dev)
            allt = torch.zeros(world, dtype=torch.float64, device=dev)
            ss._all_gather(allt, tt)
            per_rank_ms = [float(v) * 1e3 for v in allt.tolist()]
            dts = max(per_rank_ms) / 1e3
        # the same searches with events around every kernel class: this rank's per-stage times
        engine.prof_reset()
        engine.prof_enable(True)
        ss.enable_timing(True)
        for _ in range(reps):
            ss.search(qbig, args.k)
        fence()
        engine.prof_enable(False)
        p2 = engine.prof_report()
        exch_big = ss.timing_report() if world > 1 else None
        ss_timing_counts = dict(ss._timing or {})                  # tag -> the recorded exchanges (their count per tag)
        ss.enable_timing(False)
        sc = p2.get("topk_scan", {})
        scan_ms_big = sc["ms"] / sc["launches"] if sc.get("launches") else None
        planb = gal.search_plan(Qn, args.k)
        fl = 2.0 * Qn * (shard_rows - planb["prepass_rows"]) * D       # the scan kernel's own rows
        fl_all = 2.0 * Qn * args.gallery * D
        search_big = {"queries": Qn, "gallery_rows": args.gallery, "shard_rows": shard_rows, "dim": D, "k": args.k,
                      "sharded_ms": dts * 1e3, "sharded_ms_per_rank": [round(v, 4) for v in per_rank_ms],
                      "queries_per_s": Qn / dts, "sharded_ms_one_search_at_a_time": dts_sync * 1e3,
                      "stage_ms_rank0": {c: round(v["ms"] / v["launches"], 4) for c, v in sorted(p2.items())},
                      "scan_ms": scan_ms_big, "scan_rows": shard_rows - planb["prepass_rows"], "slices": planb["slices"],
                      "scan_tflops": fl / (scan_ms_big * 1e-3) / 1e12 if scan_ms_big else None,
                      "scan_frac_of_mfma_peak": fl / (scan_ms_big * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS if scan_ms_big else None,
                      "end_to_end_tflops_all_gpus": fl_all / dts / 1e12,
                      # exactness certificate: queries of the batch that failed it and were re-done by the exact fallback
                      # (collecting pass over the gallery; their cost is inside sharded_ms)
                      "uncertified_queries": cert_big["uncertified"],
                      # measured on this rank by HIP events around each all-gather (null at N = 1: there is none)
                      "allgather_ms": exch_big["allgather_ms"] if exch_big else None,
                      "exchanges_per_search": exch_big["exchanges_per_search"] if exch_big else None,
                      # pipelined searches whose uncertified queries had to be searched again because the shard handle
                      # already held the next search's candidates (sharded.PendingSearch): their cost is inside sharded_ms
                      "redone_searches": ss.redone_searches, "second_rounds": ss.second_rounds,
                      "exact_stage_ms": p2.get("topk_exact", {}).get("ms", 0.0) / max(p2.get("topk_exact", {}).get("launches", 1), 1)}
        if world > 1 and rank == 0:
            # the 1-GPU time of the SAME search in this process: the whole gallery on this one device
            full = engine.Gallery(D, args.gallery, device=local_rank)
            for r in range(world):
                gr = torch.Generator(device=dev).manual_seed(42 + r)
                rows_r = args.gallery // world + (1 if r < args.gallery % world else 0)
                for s0 in range(0, rows_r, 131072):
                    full.add(torch.randn(min(131072, rows_r - s0), D, generator=gr, device=dev))
            full.search(qbig, args.k)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                full.search(qbig, args.k)
            torch.cuda.synchronize()
            one = (time.perf_counter() - t1) / reps
            full.close()
            search_big["one_gpu_ms_same_process"] = one * 1e3
            search_big["speedup_vs_1gpu_model"] = one / dts
            # the budget the >= 6x target leaves for the exchanges, from this run's own measured times (every
            # all-gather of a search, HIP events: allgather_ms x how often each ran per search)
            ex_ms = 0.0
            if exch_big:
                n_s = max(exch_big.get("searches", reps), 1)
                ex_ms = sum(v * len(ss_timing_counts.get(t, [])) / n_s for t, v in exch_big["allgather_ms"].items() if t != "queries")
            search_big["allgather_budget"] = allgather_budget(one * 1e3, dts * 1e3, ex_ms)
            search_big["allgather_budget_ms"] = search_big["allgather_budget"]["allgather_budget_ms"]
        if world > 1:
            dist.barrier()

    # what ran, from the measured exchange counts (not from what the protocol could do)
    if world == 1:
        parallelism = "1 GPU: no exchange (embed, scan, fp32 re-score, certificate on one device)"
    else:
        ex = exch_headline or {}
        tags = sorted(t for t in ex.get("allgather_ms", {}) if t != "queries")
        parallelism = (f"dp{world} embed + all-gather of the queries, gallery rows sharded {world}-way, "
                       f"{ex.get('exchanges_per_search', float('nan')):.2f} exchange(s) per search ({', '.join(tags) or 'none'}) + merge with "
                       f"the cross-shard certificate")
    if rank == 0:
        res = {
            "metric": "images/sec embed+top-k (PE-L14-336, 1M x 1024 gallery)", "value": value, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{cfg.name} embed of {B} synthetic {cfg.image_size}x{cfg.image_size} images per GPU "
                                   f"+ cosine top-{args.k} over a {args.gallery}x{D} gallery row-sharded {world}-way",
                       "batch_per_gpu": B, "gallery_rows": args.gallery, "dim": D, "k": args.k,
                       "parallelism": parallelism},
            "roofline": roofline,
            "collective": {"backend": (args.backend if world > 1 else None), "ranks": world,
                           "rccl_ranks": world if (world > 1 and args.backend == "nccl") else 0,
                           "devices": devices, "one_gpu_rehearsal": bool(args.one_gpu),
                           # the headline steps' exchanges, measured on rank 0 by HIP events around each all-gather
                           "allgather_ms": exch_headline["allgather_ms"] if exch_headline else None,
                           "exchanges_per_search": exch_headline["exchanges_per_search"] if exch_headline else None,
                           "redone_searches": ss.redone_searches, "init_timeout_s": PG_TIMEOUT_S if world > 1 else None},
            "kernel_ms_per_step": classes_ms,
            "calibration": calib,
            "embed_tflops": cfg.flops_per_image() * B * args.steps / dt / 1e12,
            # whole step (embed + search) against the MFMA peak, SURVEY.md 8(d): images/s x FLOPs/image -- not the kernel-class `roofline.frac`
            "embed_frac_of_peak": cfg.flops_per_image() * B * args.steps / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS / world,
            "search_query_batch": search_big,
            "certificate": {"uncertified_queries_last_step": cert_headline["uncertified"],
                            "note": "every query's top-k is certified equal to an exhaustive fp32 scoring or re-done exactly "
                                    "(include/revo.h EXACTNESS); the fallback's time is inside ms_per_step"},
        }
        if calib is not None:
            # the unchanged attention kernel is the third yardstick: a control INSIDE the step
            att = classes_ms.get("attention")
            calib["attention_ms_per_launch"] = att / cfg.layers if att else None
            if calib.get("reference"):
                res["value_at_reference_box"] = value / calib["mfma_probe_vs_reference"]
                res["ms_per_step_at_reference_box"] = ms_per_step * calib["mfma_probe_vs_reference"]
                calib["note"] = ("value_at_reference_box = value x (reference mfma probe / this box's): the step is 95 % MFMA kernels "
                                 "(body GEMMs + attention); the copy probe and the attention control are printed beside it, not "
                                 "folded in")
        if world == 1 and args.ingest_images > 0:
            eng.close()
            gal.close()
            try:
                res["ingest"] = ingest_leg(args.variant, args.ingest_images, local_rank)
            except Exception as e:                                  # a reported extra: never takes the headline line down
                res["ingest"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, args.gallery, D, args.k, args.cpu_images, min(args.gallery, 250_000))
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
