"""GPU: the N > 1 product wiring (DP embed -> gather queries -> every rank scans its gallery shard with
the HIP kernels -> gather per-shard top-k -> HIP merge) rehearsed with TWO processes on one GPU.
The interconnect is gloo with host staging here (RCCL wants one device per rank; the driver's multi-GPU
run uses it); everything else is the code path of bench.py --gpus N."""
import os
import sys

import numpy as np
import pytest
import torch

import reverso_amd  # noqa: F401

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp, N, B, k):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import reverso_amd  # noqa: F401
    from reverso_amd import engine, sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    eng = engine.VitEngine.synthetic("PE-Tiny-T14-56", seed=0, device=0, max_batch=B, randomize_affine=True)
    D = eng.cfg.out_dim
    g = torch.Generator().manual_seed(99)
    gal = torch.randn(N, D, generator=g)
    imgs = torch.randint(0, 256, (world * B, 3, 56, 56), generator=g, dtype=torch.uint8)
    gal[100:160] = gal[100]                      # a tie group wider than the candidate list
    sizes = [N // world + (1 if r < N % world else 0) for r in range(world)]
    lo = sum(sizes[:rank])
    G = engine.Gallery(D, sizes[rank], device=0)
    G.add(gal[lo:lo + sizes[rank]].to(dev))
    ss = sharded.ShardedSearch.from_gallery(G)
    assert ss.offset == lo and ss.total_rows == N
    emb = eng.embed(imgs[rank * B:(rank + 1) * B].to(dev))          # this rank's images
    q = ss.gather_queries(emb)                                        # everybody's embeddings, rank order
    out = {}
    for thr in (None, 0.02):
        s, i, c = ss.search(q, k, thr)
        out[str(thr)] = (s.cpu().numpy(), i.cpu().numpy(), c.cpu().numpy())
    # pipelined form (bench.py's loops): the next search is enqueued before the previous one's result is asked for.  One
    # query sits on a group of 60 identical rows (wider than the candidate lists): its certificate fails, and by the time
    # its result is asked for the shard handle holds the NEXT search's candidates -- the pending search re-does itself,
    # in step on both ranks.  Also limit 50, whose second round draws on the scan's segments.
    qx = torch.cat([q, gal[100:101].to(dev)])
    want = [ss.search(qx, kk, thr) for kk, thr in ((k, None), (50, None), (k, 0.02))]
    assert ss.last_uncertified >= 0
    pend = [ss.search_async(qx, kk, thr) for kk, thr in ((k, None), (50, None), (k, 0.02))]
    unc = []
    for w, p in zip(want, pend):
        got = p.result()
        unc.append(ss.last_uncertified)
        for a, b in zip(got, w):
            assert torch.equal(a, b)
    assert unc[0] >= 1 and unc[2] >= 1, unc              # the tie-group query did need the second round
    out["x"] = tuple(t.cpu().numpy() for t in want[0])
    out["x50"] = tuple(t.cpu().numpy() for t in want[1])
    np.savez(os.path.join(tmp, f"rank{rank}.npz"), q=q.cpu().numpy(),
             **{f"{t}_{n}": v for t, (a, b, c) in out.items() for n, v in (("s", a), ("i", b), ("c", c))})
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_the_single_process_search(tmp_path, dev):
    import torch.multiprocessing as mp
    from reverso_amd import engine
    N, B, k, world = 40001, 4, 10, 2
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path), N, B, k), nprocs=world, join=True)
    eng = engine.VitEngine.synthetic("PE-Tiny-T14-56", seed=0, device=0, max_batch=world * B, randomize_affine=True)
    D = eng.cfg.out_dim
    g = torch.Generator().manual_seed(99)
    gal = torch.randn(N, D, generator=g)
    imgs = torch.randint(0, 256, (world * B, 3, 56, 56), generator=g, dtype=torch.uint8)
    gal[100:160] = gal[100]
    q = torch.cat([eng.embed(imgs[r * B:(r + 1) * B].to(dev)) for r in range(world)])
    G = engine.Gallery(D, N, device=0)
    G.add(gal.to(dev))
    for thr in (None, 0.02):
        s, i, c = (t.cpu().numpy() for t in G.search(q, k, thr))
        for rank in range(world):
            got = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
            assert np.array_equal(got["q"], q.cpu().numpy())
            assert np.array_equal(got[f"{thr}_i"], i) and np.array_equal(got[f"{thr}_c"], c)
            assert np.array_equal(got[f"{thr}_s"], s)
    qx = torch.cat([q, gal[100:101].to(dev)])
    for tag, kk in (("x", k), ("x50", 50)):
        s, i, c = (t.cpu().numpy() for t in G.search(qx, kk))
        assert i[-1, :min(kk, 60)].tolist() == list(range(100, 100 + min(kk, 60)))       # the tie group, index ascending
        for rank in range(world):
            got = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
            assert np.array_equal(got[f"{tag}_i"], i) and np.array_equal(got[f"{tag}_c"], c) and np.array_equal(got[f"{tag}_s"], s)


def _nccl_worker(rank, world, port, tmp, N, D, Q, k):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import reverso_amd  # noqa: F401
    from reverso_amd import engine, sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    g = torch.Generator().manual_seed(5)
    gal = torch.randn(N, D, generator=g)
    gal[100:160] = gal[100]                      # a tie group wider than the candidate list: forces the exact second round
    q = torch.randn(Q, D, generator=g)
    q[0] = gal[100]
    sizes = [N // world + (1 if r < N % world else 0) for r in range(world)]
    lo = sum(sizes[:rank])
    G = engine.Gallery(D, sizes[rank], device=rank)
    G.add(gal[lo:lo + sizes[rank]].to(dev))
    ss = sharded.ShardedSearch.from_gallery(G)
    out = {}
    for thr in (None, 0.1):
        s, i, c = ss.search(q.to(dev), k, thr)
        out[str(thr)] = (s.cpu().numpy(), i.cpu().numpy(), c.cpu().numpy(), ss.last_uncertified)
    np.savez(os.path.join(tmp, f"rank{rank}.npz"), **{f"{t}_{n}": v for t, (a, b, c, u) in out.items()
                                                        for n, v in (("s", a), ("i", b), ("c", c), ("u", np.int64(u)))})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL path wants one device per rank: needs two GPUs")
def test_two_ranks_over_rccl_equal_the_single_gpu_search(tmp_path, dev):
    """The product's N > 1 wiring exactly as bench.py --gpus N runs it: one process per GPU, torch.distributed "nccl"
    (= RCCL) all-gathers of device tensors, including the certificate's second round.  Runs wherever two GPUs are
    visible (the one-GPU boxes of this pool skip it; the driver's multi-GPU node runs it)."""
    import torch.multiprocessing as mp
    from reverso_amd import engine
    N, D, Q, k, world = 60001, 256, 37, 10, 2
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_nccl_worker, args=(world, port, str(tmp_path), N, D, Q, k), nprocs=world, join=True)
    g = torch.Generator().manual_seed(5)
    gal = torch.randn(N, D, generator=g)
    gal[100:160] = gal[100]
    q = torch.randn(Q, D, generator=g)
    q[0] = gal[100]
    G = engine.Gallery(D, N, device=0)
    G.add(gal.to(dev))
    for thr in (None, 0.1):
        s, i, c = (t.cpu().numpy() for t in G.search(q.to(dev), k, thr))
        for rank in range(world):
            got = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
            assert np.array_equal(got[f"{thr}_i"], i) and np.array_equal(got[f"{thr}_c"], c) and np.array_equal(got[f"{thr}_s"], s)
            assert int(got[f"{thr}_u"]) >= 1
