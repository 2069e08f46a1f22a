"""Search parity: bit-exact top-k indices and scores within 1e-3 (measured: ~1e-6)
against the fp32/fp64 CPU oracle and the committed golden vectors; size-independent
properties at BASELINE.json's sizes."""
import os
import sys

import numpy as np
import pytest
import torch

import reverso_amd  # noqa: F401
from reverso_amd import engine
from oracle import search as osearch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402



def _bruteforce_twin(G):
    """A handle of librevo_exp.so holding G's fp32 rows (bit for bit: they are stored normalised), forced through the
    brute-force pass -- the exhaustive fp32 scoring the product handle's certified result must equal.  (The product
    library cannot be switched: revo_search_set_mode exists only in the experiment build.)"""
    Gx = engine.Gallery(G.dim, max(len(G), 1), device=0, experiments=True)
    for s0 in range(0, len(G), 1 << 20):
        Gx.add(G.read(s0, min(1 << 20, len(G) - s0)), normalize=False)
    Gx.set_search_mode("bruteforce")
    return Gx

def _gold():
    return np.load(os.path.join(HERE, "golden", "search_4096x1024.npz"))


def _check(out, ref, atol=1e-3, near_tie=0.0):
    """near_tie > 0: two neighbours whose oracle scores differ by less than that may come out swapped
    (the GPU re-scores with an fp32 fma chain, the oracle rounds an fp64 sum once: ~1e-7 apart)."""
    s, i, c = (t.cpu().numpy() for t in out)
    rs, ri, rc = ref
    assert np.array_equal(c, rc)
    if near_tie > 0.0 and not np.array_equal(i, ri):
        for q in np.where((i != ri).any(1))[0]:
            assert sorted(i[q].tolist()) == sorted(ri[q].tolist()), q
            for j in np.where(i[q] != ri[q])[0]:
                jj = int(np.where(ri[q] == i[q][j])[0][0])
                assert abs(jj - j) == 1 and abs(rs[q][jj] - rs[q][j]) <= near_tie, (q, j)
    else:
        assert np.array_equal(i, ri)
    fin = np.isfinite(rs)
    assert np.array_equal(np.isfinite(s), fin)
    assert np.abs(s[fin] - rs[fin]).max(initial=0.0) <= atol
    return np.abs(s[fin] - rs[fin]).max(initial=0.0)


def _assert_indices_equal_up_to_fp32_ties(i, ri, rs, gal, qr, tie=3e-7):
    """Index equality with the fp64 oracle, except where fp32 cannot tell two rows apart: a differing position must hold
    a row whose true (fp64) score is within `tie` of the oracle's score at that position (the GPU's exact scores are
    fp32 fma chains, the oracle rounds an fp64 sum once: ~1e-7 apart).  Returns the number of queries whose index SETS
    differ (a tie exactly at the k-th place; swaps of neighbours inside the list are not counted)."""
    bad = np.where((i != ri).any(1))[0]
    if bad.size == 0:
        return 0
    g64 = None if callable(gal) else gal.astype(np.float64)       # callable: row index -> fp32 row (a gallery too large to copy)
    for q in bad:
        qv = qr[q].astype(np.float64)
        qv /= np.linalg.norm(qv)
        for j in np.where(i[q] != ri[q])[0]:
            row = gal(int(i[q, j])).astype(np.float64) if g64 is None else g64[i[q, j]]
            true = float(row @ qv / np.linalg.norm(row))
            assert abs(true - float(rs[q, j])) <= tie, (q, j, true, rs[q, j])
    return sum(sorted(i[q].tolist()) != sorted(ri[q].tolist()) for q in bad)      # queries whose index SETS differ


def test_golden_cases(dev):
    gold = _gold()
    gal, qr, perm = make_golden.search_case()
    G = engine.Gallery(1024, 5000, device=0)
    G.add(torch.from_numpy(gal))                      # host rows -> staged upload
    assert len(G) == 4096
    q = torch.from_numpy(qr).to(dev)
    worst = 0.0
    for k in (1, 5, 10, 50):
        for thr, tag in ((None, "none"), (0.7, "0p7")):
            t = f"k{k}_thr{tag}"
            worst = max(worst, _check(G.search(q, k, thr), (gold[t + "_scores"], gold[t + "_indices"], gold[t + "_counts"])))
    assert worst <= 1e-5          # re-scored in fp32: far inside the 1e-3 bound
    # single query, the reference's actual usage (core_system.py:657)
    s, i, c = G.search(q[0], 5, 0.7)
    assert i.cpu().tolist()[0] == [100, 101, 102, 103, 104] and int(c[0]) == 5
    G.close()


def test_shard_and_merge_equals_unsharded(dev):
    gold = _gold()
    gal, qr, _ = make_golden.search_case()
    q = torch.from_numpy(qr).to(dev)
    shard = 512
    ps, pi = [], []
    for p in range(8):
        G = engine.Gallery(1024, shard, device=0)
        G.add(torch.from_numpy(gal[p * shard:(p + 1) * shard]).to(dev))
        s, i, c = G.search(q, 10, None, index_offset=p * shard)
        ps.append(s)
        pi.append(i)
        G.close()
    out = engine.merge_topk(torch.stack(ps), torch.stack(pi), 10, None)
    _check(out, (gold["merged_scores"], gold["merged_indices"], gold["merged_counts"]))
    out = engine.merge_topk(torch.stack(ps), torch.stack(pi), 10, 0.7)
    _check(out, (gold["k10_thr0p7_scores"], gold["k10_thr0p7_indices"], gold["k10_thr0p7_counts"]))


@pytest.mark.parametrize("N,D,Q,k", [(1, 64, 1, 1), (5, 64, 3, 5), (127, 128, 2, 10), (129, 256, 130, 16),
                                     (1000, 1536, 7, 10), (20000, 1024, 300, 10), (3333, 64, 1, 50)])
def test_ragged_sizes_vs_oracle(dev, N, D, Q, k):
    rng = np.random.default_rng(N + D + Q)
    gal = rng.standard_normal((N, D), dtype=np.float32)
    qr = rng.standard_normal((Q, D), dtype=np.float32)
    if N > 10:
        qr[0] = gal[N // 2] * 3.0                         # exact (scaled) copy: cosine 1
    G = engine.Gallery(D, N, device=0)
    G.add(torch.from_numpy(gal).to(dev))
    for thr in (None, 0.2):
        _check(G.search(torch.from_numpy(qr).to(dev), k, thr), osearch.search(gal, qr, k, thr), atol=1e-5)
    G.close()


def test_adversarial_order_forces_queue_overflow(dev):
    """Scores rise with the row index, so nearly every row of every tile beats the running
    admission score: the selection queue overflows and tiles are re-scanned; results stay exact."""
    N, D, Q, k = 40000, 64, 300, 10
    rng = np.random.default_rng(5)
    qr = rng.standard_normal((Q, D), dtype=np.float32)
    base = qr.mean(0)
    ramp = np.linspace(0.0, 4.0, N, dtype=np.float32)[:, None]
    gal = base[None] * ramp + rng.standard_normal((N, D), dtype=np.float32)
    G = engine.Gallery(D, N, device=0)
    G.add(torch.from_numpy(gal).to(dev))
    _check(G.search(torch.from_numpy(qr).to(dev), k), osearch.search(gal, qr, k), atol=1e-5)
    G.close()


def test_every_late_row_beats_the_seed(dev):
    """The pre-pass rows are unrelated to the queries and every later row is close to every query:
    the first fused tile admits all 256 x 256 scores, far more than the selection queue holds, so
    the tile is recomputed in column groups (the retry ladder of topk256.hip).  Results stay exact."""
    D, Q, k = 64, 300, 10
    rng = np.random.default_rng(11)
    base = rng.standard_normal(D).astype(np.float32)
    qr = base[None] + 0.3 * rng.standard_normal((Q, D), dtype=np.float32)
    gal = np.concatenate([rng.standard_normal((8192, D), dtype=np.float32),
                          base[None] + 0.3 * rng.standard_normal((9000, D), dtype=np.float32)])
    G = engine.Gallery(D, len(gal), device=0)
    G.add(torch.from_numpy(gal).to(dev))
    out = G.search(torch.from_numpy(qr).to(dev), k)
    _check(out, osearch.search(gal, qr, k), atol=1e-5)
    assert int(out[1].min()) >= 8192
    G.close()


def test_all_rows_identical_ties_terminate(dev):
    """Every score ties: admission by score alone would refill the queue forever; the strict key
    test admits an equal score only with a smaller index."""
    D = 64
    v = np.random.default_rng(3).standard_normal(D).astype(np.float32)
    gal = np.repeat(v[None], 30000, axis=0)
    G = engine.Gallery(D, len(gal), device=0)
    G.add(torch.from_numpy(gal).to(dev))
    q = torch.from_numpy(np.stack([v, -v, v * 2])).to(dev)
    s, i, c = G.search(q, 10)
    assert i[0].cpu().tolist() == list(range(10)) and i[2].cpu().tolist() == list(range(10))
    assert i[1].cpu().tolist() == list(range(10))            # all -1.0: still index order
    assert torch.allclose(s.cpu(), torch.tensor([[1.0] * 10, [-1.0] * 10, [1.0] * 10]), atol=1e-6)
    G.close()


@pytest.mark.parametrize("N,Q", [(16384, 1), (16385, 257), (70001, 64), (250000, 513), (60000, 129), (90001, 192), (50000, 100)])
def test_large_scan_path_vs_oracle(dev, N, Q):
    D, k = 128, 10
    rng = np.random.default_rng(N + Q)
    gal = rng.standard_normal((N, D), dtype=np.float32)
    gal[N // 3: N // 3 + 40] = gal[N // 3]                 # a tie group longer than the candidate list
    qr = rng.standard_normal((Q, D), dtype=np.float32)
    qr[0] = gal[N // 3]
    G = engine.Gallery(D, N, device=0)
    G.add(torch.from_numpy(gal).to(dev))
    out = G.search(torch.from_numpy(qr).to(dev), k, None)
    _check(out, osearch.search(gal, qr, k), atol=1e-5)
    assert out[1][0].cpu().tolist() == list(range(N // 3, N // 3 + 10))
    _check(G.search(torch.from_numpy(qr).to(dev), k, 0.3), osearch.search(gal, qr, k, 0.3), atol=1e-5)
    G.close()


def test_empty_gallery_and_append_in_pieces(dev):
    G = engine.Gallery(64, 100, device=0)
    q = torch.randn(3, 64, device=dev)
    s, i, c = G.search(q, 5)
    assert (c == 0).all() and (i == -1).all() and torch.isinf(s).all()
    rng = np.random.default_rng(1)
    gal = rng.standard_normal((100, 64), dtype=np.float32)
    assert G.add(torch.from_numpy(gal[:30]).to(dev)) == 0
    assert G.add(torch.from_numpy(gal[30:31])) == 30
    assert G.add(torch.from_numpy(gal[31:]).to(dev)) == 31
    with pytest.raises(Exception):
        G.add(torch.zeros(1, 64))                         # over capacity
    _check(G.search(q, 5), osearch.search(gal, q.cpu().numpy(), 5), atol=1e-5)
    back = G.read(0, 100).cpu().numpy()
    np.testing.assert_allclose(back, osearch.normalize_rows(gal), atol=1e-6)
    G.clear()
    assert len(G) == 0
    G.close()


def test_all_duplicates_tie_order(dev):
    v = torch.randn(1, 128)
    G = engine.Gallery(128, 200, device=0)
    G.add(v.expand(200, 128).contiguous().to(dev))
    s, i, c = G.search(v.to(dev), 50)
    assert i.cpu().tolist()[0] == list(range(50))
    assert torch.allclose(s.cpu(), torch.ones(1, 50), atol=1e-6)
    G.close()


def test_properties_at_100k(dev):
    """configs[1] gallery (100k x 1024): planted neighbours are found first; top-k is sorted;
    a row-permuted gallery returns permuted indices; CPU oracle agrees on a query sample."""
    N, D, Q, k = 100_000, 1024, 64, 10
    g = torch.Generator(device=dev).manual_seed(42)
    gal = torch.randn(N, D, generator=g, device=dev)
    perm = torch.randperm(N, generator=g, device=dev)
    q = gal[perm[:Q]] + 0.05 * torch.randn(Q, D, generator=g, device=dev)
    G = engine.Gallery(D, N, device=0)
    G.add(gal)
    s, i, c = G.search(q, k)
    assert torch.equal(i[:, 0], perm[:Q])
    assert (s[:, :-1] >= s[:, 1:]).all() and (c == k).all()
    rs, ri, rc = osearch.search(gal.cpu().numpy(), q[:8].cpu().numpy(), k)
    assert np.array_equal(i[:8].cpu().numpy(), ri)
    assert np.abs(s[:8].cpu().numpy() - rs).max() <= 1e-5
    shuffle = torch.randperm(N, generator=g, device=dev)
    G2 = engine.Gallery(D, N, device=0)
    G2.add(gal[shuffle])
    s2, i2, _ = G2.search(q, k)
    assert torch.equal(shuffle[i2], i)
    assert torch.allclose(s2, s, atol=1e-6)
    G.close()
    G2.close()


def test_properties_at_full_gallery_1m(dev):
    """BASELINE.json's 1 M x 1024 gallery: size-independent properties for all queries AND the fp64 oracle's exhaustive
    ranking of all 1 M rows for 16 of them (chunked: seconds of numpy).  Planted neighbours come first, results are sorted and complete, an 8-way row shard + merge equals the
    unsharded search bit for bit, scores are the fp32 dot products, and a 10 000-query batch (configs[3])
    agrees with the 64-query calls on the shared queries."""
    N, D, k = 1_000_000, 1024, 10
    g = torch.Generator(device=dev).manual_seed(2024)
    G = engine.Gallery(D, N, device=0)
    for s0 in range(0, N, 125_000):
        G.add(torch.randn(125_000, D, generator=g, device=dev))
    Q = 64
    ids = torch.randint(0, N, (Q,), generator=g, device=dev)
    ids = torch.unique(ids)
    Q = ids.numel()
    rows = torch.cat([G.read(int(i), 1) for i in ids])
    q = rows + 0.02 * torch.randn(Q, D, generator=g, device=dev)
    s, i, c = G.search(q, k)
    assert torch.equal(i[:, 0], ids) and (c == k).all()
    assert (s[:, :-1] >= s[:, 1:]).all()
    # scores are exact fp32 cosines of the returned rows
    qn = torch.nn.functional.normalize(q.double(), dim=-1)
    for j in (0, Q // 2, Q - 1):
        got = torch.cat([G.read(int(r), 1) for r in i[j]]).double()
        assert ((got @ qn[j]) - s[j].double()).abs().max().item() <= 2e-6
    # 8 shards searched apart (global row ids via index_offset) and merged == unsharded
    shards = []
    for p in range(8):
        Gp = engine.Gallery(D, 125_000, device=0)
        Gp.add(G.read(p * 125_000, 125_000), normalize=False)
        shards.append(Gp)
    parts_s, parts_i = [], []
    for p, Gp in enumerate(shards):
        ps, pi, _ = Gp.search(q, k, index_offset=p * 125_000)
        parts_s.append(ps); parts_i.append(pi)
    ms, mi, mc = engine.merge_topk(torch.stack(parts_s), torch.stack(parts_i), k)
    assert torch.equal(mi, i) and torch.equal(ms, s) and torch.equal(mc, c)
    # the two-phase protocol of the sharded search (sharded.py; BASELINE.json configs[3]) on the same 8 shards, with
    # the 10 000-query batch: publish admission scores, "all-gather", bounded fp32 re-score, packed "all-gather",
    # merge == the unsharded search, bit for bit
    big = torch.cat([q, torch.randn(10_000 - Q, D, generator=g, device=dev)])
    bs, bi, bc = G.search(big, k)
    top_m = 8
    assert engine.search_ksel(k) == 32
    allb = torch.stack([Gp.search_candidates(big, k, top_m) for Gp in shards])          # [8, Q, top_m]
    # (each shard's candidates stay in its handle between the two phases)
    pb = engine.packed_bytes(big.shape[0], k)
    packed = torch.empty((8 * pb,), dtype=torch.uint8, device=dev)
    for p, Gp in enumerate(shards):
        Gp.search_finish(big.shape[0], k, allb, None, p * 125_000, out_packed=packed[p * pb:(p + 1) * pb])
    ts, ti, tc, unc = engine.merge_topk_packed(packed, 8, big.shape[0], k, certify=True)
    assert torch.equal(ti, bi) and torch.equal(ts, bs) and torch.equal(tc, bc)
    # the shards together re-score what reaches the 64th-best published score (2 x ksel): with that margin the
    # cross-shard certificate holds for (nearly) every query of a random gallery -- no second round
    assert int(unc[0].item()) <= 2
    for Gp in shards:
        Gp.close()
    # a large query batch takes the MFMA-bound regime of the same scan: same answers on the shared queries
    assert torch.equal(bi[:Q], i) and torch.equal(bs[:Q], s)
    assert (bc == k).all() and (bs[:, :-1] >= bs[:, 1:]).all()
    # THE ORACLE ON THIS GALLERY (core_system.py:659-664 semantics via oracle/search.py): the north star states "bit-exact
    # top-k indices for a fixed gallery" on the 1 M x 1024 gallery, so the exhaustive fp64 ranking is run here too -- 16
    # queries (8 planted, 8 random) over all 1 M stored rows, read back in chunks; k = 10 and 50, with and without a
    # threshold; and the same 16 rows of the 10 000-query batch's result.
    sel = list(range(8)) + list(range(Q, Q + 8))
    q16 = big[sel]
    q16_np = q16.cpu().numpy()
    chunks = ((s0, G.read(s0, 125_000).cpu().numpy()) for s0 in range(0, N, 125_000))
    sc = {}
    o50 = osearch.search_chunked(chunks, q16_np, 50)                   # one pass over the rows: k = 50 holds k = 10
    row_of = lambda r: G.read(r, 1).cpu().numpy()[0]
    for kk, thr in ((10, None), (50, None), (10, 0.3), (50, 0.05)):
        rs, ri = o50[0][:, :kk].copy(), o50[1][:, :kk].copy()
        if thr is not None:
            keep = rs >= np.float32(thr)
            rs[~keep], ri[~keep] = -np.inf, -1
        rc = (ri >= 0).sum(1).astype(np.int32)
        gs, gi, gc = (t.cpu().numpy() for t in G.search(q16, kk, thr))
        assert np.array_equal(gc, rc), (kk, thr)
        fin = np.isfinite(rs)
        assert np.array_equal(np.isfinite(gs), fin) and np.abs(gs[fin] - rs[fin]).max() <= 1e-5, (kk, thr)
        # (threshold cases: a score within fp32 resolution of the threshold could flip a count; the counts agree here)
        sc[(kk, thr)] = _assert_indices_equal_up_to_fp32_ties(np.where(fin, gi, -1), ri, rs, row_of, q16_np, tie=6e-7)
    assert sc[(10, None)] == 0                                           # no k-th-place tie among these 16 queries
    assert np.array_equal(bi[sel].cpu().numpy(), o50[1][:, :10]) and np.abs(bs[sel].cpu().numpy() - o50[0][:, :10]).max() <= 1e-5
    assert (o50[1][:8, 0] == ids[:8].cpu().numpy()).all()                # the planted rows are the oracle's best, too
    G.close()


@pytest.mark.parametrize("k", [17, 20, 50])
def test_large_scan_path_wide_k(dev, k):
    """limit 20 and 50 (the reference UI's other choices, ui.py:342) keep 64 candidates per query: the
    256 x 256 scan's 64-entry lists (128-element merges with de-duplication in the drains)."""
    for (N, Q, D) in [(70001, 64, 128), (120000, 300, 64), (40000, 180, 64)]:       # 64-, 256- and 192-row forms of the scan
        rng = np.random.default_rng(N + Q + k)
        gal = rng.standard_normal((N, D), dtype=np.float32)
        gal[N // 3: N // 3 + 90] = gal[N // 3]             # a tie group longer than the 64-entry list
        qr = rng.standard_normal((Q, D), dtype=np.float32)
        qr[0] = gal[N // 3]
        G = engine.Gallery(D, N, device=0)
        G.add(torch.from_numpy(gal).to(dev))
        out = G.search(torch.from_numpy(qr).to(dev), k, None)
        _check(out, osearch.search(gal, qr, k), atol=1e-5, near_tie=3e-7)
        assert out[1][0].cpu().tolist() == list(range(N // 3, N // 3 + k))
        _check(G.search(torch.from_numpy(qr).to(dev), k, 0.25), osearch.search(gal, qr, k, 0.25), atol=1e-5, near_tie=3e-7)
        G.close()


def test_wide_k_adversarial_overflow_and_ties(dev):
    """The queue-overflow ladder, the re-queued duplicates and the all-ties gallery with 64-entry lists."""
    k = 50
    N, D, Q = 40000, 64, 300
    rng = np.random.default_rng(5)
    qr = rng.standard_normal((Q, D), dtype=np.float32)
    ramp = np.linspace(0.0, 4.0, N, dtype=np.float32)[:, None]
    gal = qr.mean(0)[None] * ramp + rng.standard_normal((N, D), dtype=np.float32)
    G = engine.Gallery(D, N, device=0)
    G.add(torch.from_numpy(gal).to(dev))
    _check(G.search(torch.from_numpy(qr).to(dev), k), osearch.search(gal, qr, k), atol=1e-5, near_tie=3e-7)
    G.close()
    base = rng.standard_normal(D).astype(np.float32)
    qr = base[None] + 0.3 * rng.standard_normal((Q, D), dtype=np.float32)
    gal = np.concatenate([rng.standard_normal((8192, D), dtype=np.float32),
                          base[None] + 0.3 * rng.standard_normal((9000, D), dtype=np.float32)])
    G = engine.Gallery(D, len(gal), device=0)
    G.add(torch.from_numpy(gal).to(dev))
    # 9000 rows within a few 1e-5 of each other per rank: far more near-ties than the scan's 64 candidates hold.  The
    # exactness certificate (include/revo.h) fails for these queries and the collecting pass re-does them: the result
    # is that of an exhaustive fp32 scoring -- identical, bit for bit, to the brute-force mode of the same handle --
    # and equals the fp64 oracle's index for index except where two neighbours tie below fp32 resolution.
    qd = torch.from_numpy(qr).to(dev)
    out = G.search(qd, k)
    st = G.search_stats()
    assert st["checked"] == Q and st["uncertified"] >= Q // 2, st
    with pytest.raises(Exception, match="librevo_exp"):
        G.set_search_mode("bruteforce")               # the product handle cannot leave the certified mode
    Gx = _bruteforce_twin(G)
    ref = Gx.search(qd, k)
    assert Gx.search_stats()["bruteforced"] == Q
    Gx.close()
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    s, i, c = (t.cpu().numpy() for t in out)
    rs, ri, rc = osearch.search(gal, qr, k)
    assert np.array_equal(c, rc) and int(i.min()) >= 8192
    assert np.abs(s - rs).max() <= 5e-7          # a few fp32 ulps of 0.95: the fma chain against the fp64 sum
    _assert_indices_equal_up_to_fp32_ties(i, ri, rs, gal, qr, tie=6e-7)
    G.close()
    v = rng.standard_normal(D).astype(np.float32)
    G = engine.Gallery(D, 30000, device=0)
    G.add(torch.from_numpy(np.repeat(v[None], 30000, axis=0)).to(dev))
    s, i, c = G.search(torch.from_numpy(np.stack([v, -v])).to(dev), k)
    assert i[0].cpu().tolist() == list(range(k)) and i[1].cpu().tolist() == list(range(k))
    G.close()


@pytest.mark.parametrize("k,csize", [(10, 22), (50, 14), (10, 190), (50, 150)])
def test_near_duplicate_frame_clusters(dev, k, csize):
    """Near-duplicate video frames (the reference stores one vector per frame region, core_system.py:406-408): clusters
    of rows 1e-3 apart per element whose scores against a query lie within ~6e-5 of each other, i.e. inside the scan's
    bf16 input rounding (~1e-4): the scan's ranking INSIDE a cluster is noise.  Clusters of k + csize members: up to
    what the scan's candidate lists hold beyond k (22 at k = 10, 14 at k = 50) the fp32 re-score of the candidates
    decides; clusters of 200 members overflow the lists, the certificate fails and the collecting pass re-does the
    query -- either way the result is the exhaustive fp32 search's (bit-identical to the brute-force mode) and the
    oracle's up to fp32-level ties."""
    N, D, Q = 150000, 256, 96
    rng = np.random.default_rng(100 + k)
    gal = rng.standard_normal((N, D), dtype=np.float32)
    centres = rng.standard_normal((Q, D), dtype=np.float32)
    members = csize + k
    pos = rng.permutation(N)[: Q * members].reshape(Q, members)
    for q in range(Q):
        gal[pos[q]] = centres[q][None] + 1e-3 * rng.standard_normal((members, D), dtype=np.float32)
    qr = centres + 0.3 * rng.standard_normal((Q, D), dtype=np.float32)
    G = engine.Gallery(D, N, device=0)
    G.add(torch.from_numpy(gal).to(dev))
    qd = torch.from_numpy(qr).to(dev)
    out = G.search(qd, k, None)
    st = G.search_stats()
    if csize > 64:
        assert st["uncertified"] == Q and st["bruteforced"] == 0, st      # every query needed (only) the collecting pass
    Gx = _bruteforce_twin(G)
    ref = Gx.search(qd, k, None)
    Gx.close()
    G.close()
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    s, i, c = (t.cpu().numpy() for t in out)
    rs, ri, rc = osearch.search(gal, qr, k)
    assert np.array_equal(c, rc)
    assert np.abs(s - rs).max() <= 1e-6
    assert np.median(np.abs(np.diff(rs, axis=1))) > (5e-7 if csize < 64 else 5e-8)   # the planted gaps are above fp32 resolution
    for q in range(Q):
        assert set(i[q].tolist()) <= set(pos[q].tolist()), q
    ties = _assert_indices_equal_up_to_fp32_ties(i, ri, rs, gal, qr)
    if csize < 64:
        assert ties <= Q // 8     # neighbours ~2e-6 apart: only the occasional fp32-level tie (200-member clusters: ~5e-7 apart)


def test_certificate_modes_agree_and_count(dev):
    """The three routes to a result -- certified fast path, collecting pass, brute force -- give the same bits, on a
    random gallery (certificates pass), with a threshold, on the small-gallery scan and with exact duplicates wider
    than the candidate list (certificates fail: tie at the list's end)."""
    rng = np.random.default_rng(11)
    for (N, D, Q, k, thr) in [(70001, 128, 70, 10, None), (70001, 128, 70, 50, 0.2), (9000, 64, 33, 5, None),
                              (40000, 64, 300, 10, 0.3)]:
        gal = rng.standard_normal((N, D), dtype=np.float32)
        gal[N // 2: N // 2 + 45] = gal[N // 2]                       # 45 identical rows (one image's regions)
        qr = rng.standard_normal((Q, D), dtype=np.float32)
        qr[1] = gal[N // 2] + 0.01 * rng.standard_normal(D).astype(np.float32)
        G = engine.Gallery(D, N, device=0)
        G.add(torch.from_numpy(gal).to(dev))
        qd = torch.from_numpy(qr).to(dev)
        outs = {"product": G.search(qd, k, thr)}                  # librevo.so: certified, the only mode it has
        stp = G.search_stats()
        assert stp["checked"] == Q and 1 <= stp["uncertified"] <= Q // 4, stp
        Gx = _bruteforce_twin(G)                                  # librevo_exp.so: the same rows, every mode
        for mode in ("certified", "collect", "bruteforce", "uncertified"):
            Gx.set_search_mode(mode)
            outs[mode] = Gx.search(qd, k, thr)
            st = Gx.search_stats()
            assert st["checked"] == Q, (mode, st)
            if mode == "certified":
                assert 1 <= st["uncertified"] <= Q // 4, st            # the duplicate group's query, few others
            if mode == "collect":
                assert st["uncertified"] == Q and st["bruteforced"] == 0 and st["collected_rows"] >= Q * min(k, 5), st
            if mode == "bruteforce":
                assert st["bruteforced"] == Q, st
        G.close()
        Gx.close()
        for mode in ("product", "collect", "bruteforce"):
            for a, b in zip(outs["certified"], outs[mode]):
                assert torch.equal(a, b), (N, k, mode)
        _check(outs["certified"], osearch.search(gal, qr, k, thr), atol=1e-5, near_tie=3e-7)
        assert outs["certified"][1][1, :k].cpu().tolist() == list(range(N // 2, N // 2 + k)) or thr is not None


def test_certificate_error_bound_is_rigorous(dev):
    """eps of the certificate against measured |bf16-scan score - fp32 score|: random rows, rows
    whose bf16 rounding errors all point along the query (the worst case of the Cauchy-Schwarz terms), and long rows
    (D = 1536).  The scan's own scores come from a gallery without fp32 rows (keep_f32 = 0 returns them as they are)."""
    for D in (256, 1536):
        rng = np.random.default_rng(D)
        N, Q, k = 20000, 40, 16
        gal = osearch.normalize_rows(rng.standard_normal((N, D)).astype(np.float32))
        qr = rng.standard_normal((Q, D)).astype(np.float32)
        # adversarial rows (stored as given: normalize = False): every element of a unit row is moved to just above a
        # bf16 rounding midpoint, so bf16(g) - g has the sign of g and nearly the largest size it can have, and the query
        # is the row itself: the row's rounding error is parallel to the query
        u = np.abs(gal[:8]).view(np.uint32)
        u = (u & np.uint32(0xffff0000)) | np.uint32(0x00008001)
        gal[:8] = u.view(np.float32)
        qr[:8] = gal[:8]
        Gs = engine.Gallery(D, N, device=0, keep_f32=False)
        Gf = engine.Gallery(D, N, device=0, experiments=True)
        gd, qd = torch.from_numpy(gal).to(dev), torch.from_numpy(qr).to(dev)
        Gs.add(gd, normalize=False)
        Gf.add(gd, normalize=False)
        ss, si, _ = Gs.search(qd, k)
        Gf.set_search_mode("bruteforce")
        fs, fi, _ = Gf.search(qd, k)
        rows = Gf.read()
        Gs.close()
        Gf.close()
        # the bound, from the same quantities the kernels use (fp32 rows as stored, bf16 = round-to-nearest-even)
        qn = torch.nn.functional.normalize(qd, dim=-1)
        qb, gb = qn.bfloat16().float(), rows.bfloat16().float()
        e_q, n_qb = (qb - qn).norm(dim=-1), qb.norm(dim=-1)
        Gmax, Eg = rows.norm(dim=-1).max(), (gb - rows).norm(dim=-1).max()
        eps = (e_q * Gmax + n_qb * Eg + D * 2.0 ** -23 * n_qb * (Gmax + Eg) + D * 2.0 ** -24 * (n_qb + e_q) * Gmax) * 1.001
        # fp32 scores of the rows the scan returned
        exact = torch.einsum("qd,qkd->qk", qn.double(), rows[si].double())
        err = (ss.double() - exact).abs().max(dim=1).values
        assert (err <= eps.double()).all(), (err.max().item(), eps.min().item())
        # the adversarial pairs get close to the bound (where the query's own rounding, after its normalisation, happens to
        # point the same way as the row's: measured 0.97 of eps; where it points the other way the two cancel)
        assert (err[:8] / eps[:8].double()).max().item() >= 0.5
        # and the bound is what DESIGN.md says it is: bf16 keeps 8 significant bits, unit roundoff 2^-8 per operand
        assert eps.max().item() <= 2.2 * 2.0 ** -8 + 4 * D * 2.0 ** -23
        assert eps[8:].max().item() <= 1.5 * 2.0 ** -8                   # random queries: rounding norms ~0.45 * 2^-8, not the worst case
        assert torch.equal(fi[:8, 0].cpu(), torch.arange(8))


@pytest.mark.parametrize("k", [10, 50])
def test_scan_bound_histogram_edge_cases(dev, k):
    """The scan's admission bound comes from per-query score histograms whose 64 buckets start at the
    pre-pass bound and span one binade of the fp32 score.  Cases that leave that window must stay exact
    (the bound only ever lags; full segments fall back to the sorted-merge path):
    (a) clusters of near-copies of some queries far above the window (scores ~0.95 against a ~0.3 seed),
        late in the gallery and longer than a segment;
    (b) every score negative (bucket origin below zero);
    (c) a seed slightly below zero with positive scores later on (the window straddles zero)."""
    D, Q = 128, 300
    rng = np.random.default_rng(100 + k)
    # (a)
    N = 90000
    gal = rng.standard_normal((N, D), dtype=np.float32)
    qr = rng.standard_normal((Q, D), dtype=np.float32)
    for j in range(12):
        rows = 20000 + j * 5000 + np.arange(150)            # 150 > 2 * 64: the row's segment overflows
        # cosines spread from ~0.995 down to ~0.8 in steps of ~1.5e-3: further apart than the bf16 selection noise
        lvl = np.sqrt(0.01 + 0.5 * np.arange(150, dtype=np.float32) / 150)[:, None]
        gal[rows] = qr[j][None] + lvl * rng.standard_normal((150, D), dtype=np.float32)
    G = engine.Gallery(D, N, device=0)
    G.add(torch.from_numpy(gal).to(dev))
    out = G.search(torch.from_numpy(qr).to(dev), k)
    _check(out, osearch.search(gal, qr, k), atol=1e-5, near_tie=3e-7)
    assert float(out[0][0, 0]) > 0.9 and float(out[0][20, 0]) < 0.6
    G.close()
    # (b)
    N = 50000
    base = rng.standard_normal(D).astype(np.float32)
    qr = base[None] + 0.5 * rng.standard_normal((Q, D), dtype=np.float32)
    gal = -(base[None] + 0.5 * rng.standard_normal((N, D), dtype=np.float32))
    G = engine.Gallery(D, N, device=0)
    G.add(torch.from_numpy(gal).to(dev))
    out = G.search(torch.from_numpy(qr).to(dev), k)
    _check(out, osearch.search(gal, qr, k), atol=1e-5, near_tie=3e-7)
    assert float(out[0].max()) < 0.0
    G.close()
    # (c)
    gal = np.concatenate([-(base[None] + 2.0 * rng.standard_normal((20000, D), dtype=np.float32)),
                          rng.standard_normal((30000, D), dtype=np.float32)])
    G = engine.Gallery(D, len(gal), device=0)
    G.add(torch.from_numpy(gal).to(dev))
    out = G.search(torch.from_numpy(qr).to(dev), k)
    _check(out, osearch.search(gal, qr, k), atol=1e-5, near_tie=3e-7)
    G.close()


def test_scan_slice_balance_and_tails(dev):
    """Gallery sizes that leave ragged last tiles and uneven slices (the tiles are dealt out as evenly as
    possible, the first slices take one more), with the planted best row in the very last gallery row."""
    D, k = 64, 10
    for N, Q in [(16384 + 255, 600), (50001, 1500), (65536 + 256 * 7 + 1, 257)]:
        rng = np.random.default_rng(N)
        gal = rng.standard_normal((N, D), dtype=np.float32)
        qr = rng.standard_normal((Q, D), dtype=np.float32)
        gal[N - 1] = qr[Q - 1]
        gal[0] = qr[0]
        G = engine.Gallery(D, N, device=0)
        G.add(torch.from_numpy(gal).to(dev))
        out = G.search(torch.from_numpy(qr).to(dev), k)
        _check(out, osearch.search(gal, qr, k), atol=1e-5)
        assert int(out[1][Q - 1, 0]) == N - 1 and int(out[1][0, 0]) == 0
        G.close()


@pytest.mark.parametrize("N,Q,k", [(40001, 2048, 10), (70003, 2300, 10), (120000, 2064, 10), (33000, 3000, 50),
                                   (300000, 2560, 5)])
def test_phases_of_query_tiles_vs_oracle(dev, N, Q, k):
    """Eight query tiles and more: the scan launch is a sequence of phases (8 a query tiles pinned a per XCD with 32 / a
    slices each, largest a first, then the remaining < 8 query tiles with the slices of a small launch: topk256.hip), phases
    have different slice counts and the segment slots a phase does not use are closed by its slice 0.  Ragged last query
    tiles (2300 = 8 x 256 + 252; 2064 = 8 x 256 + 16: a tail launch or a ninth tile, as the cost model decides), the margin
    form (k = 50), best rows planted in the gallery's first scanned and very last rows and at every query tile's edges."""
    D = 64
    rng = np.random.default_rng(N + Q + k)
    gal = rng.standard_normal((N, D), dtype=np.float32)
    qr = rng.standard_normal((Q, D), dtype=np.float32)
    plant = sorted(set([0, 255, 256, 257, Q - 1] + [t * 256 for t in range(1, Q // 256)] + [t * 256 - 1 for t in range(1, Q // 256 + 1)]))
    rows = np.linspace(N - 1, N // 2, num=len(plant)).astype(np.int64)     # one distinct gallery row per planted query
    rows[0] = N - 1
    for qi, r in zip(plant, rows):
        gal[r] = qr[qi]
    G = engine.Gallery(D, N, device=0)
    G.add(torch.from_numpy(gal).to(dev))
    plan = G.search_plan(Q, k)
    assert plan["scan256"] and 2 <= plan["slices"] <= 512, plan             # segment slots per query: the most slices of any phase
    out = G.search(torch.from_numpy(qr).to(dev), k)
    _check(out, osearch.search(gal, qr, k), atol=1e-5, near_tie=3e-7)      # (150 000 places: fp32 and fp64 orders differ at ~1e-7)
    for qi, r in zip(plant, rows):
        assert int(out[1][qi, 0]) == int(r), (qi, r)
    _check(G.search(torch.from_numpy(qr).to(dev), k, 0.35), osearch.search(gal, qr, k, 0.35), atol=1e-5, near_tie=3e-7)
    G.close()


def test_config4_gallery_10m_x_1536(dev):
    """BASELINE.json configs[4]'s gallery as written: 10 M x 1536 (bf16 scan copy 30.7 GB + fp32 master 61 GB of
    the 288 GB), 256 queries, through the 256 x 256 scan.  The CPU oracle cannot hold this; what is checked:
    planted near-copies at the first / last / pre-pass-boundary rows come back in the planted order, results are
    sorted and complete, returned scores are the exact fp32 cosines of the returned rows, no row of four sampled
    50 000-row ranges beats a query's 10th result without being in it (oracle on the sample), and an 8-way
    row shard + merge equals the unsharded search bit for bit."""
    N, D, Q, k = 10_000_000, 1536, 256, 10
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 130 << 30:
        pytest.skip("needs ~110 GB of free HBM")
    g = torch.Generator(device=dev).manual_seed(404)
    q = torch.randn(Q, D, generator=g, device=dev)
    plant_rows = [0, 1, 32767, 32768, 32769, 65535, 65536, 65537, N // 2, N - 2, N - 1]
    G = engine.Gallery(D, N, device=0)
    chunk = 500_000
    for s0 in range(0, N, chunk):
        G.add(torch.randn(chunk, D, generator=g, device=dev))
    # the gallery is append-only: instead of planting copies of the queries, the first queries are made near-copies of
    # gallery rows at the interesting positions (first / last rows, both sides of the pre-pass boundary)
    ids = torch.tensor(plant_rows, device=dev)
    qrows = torch.cat([G.read(int(r), 1) for r in plant_rows])   # the normalised fp32 rows themselves
    q[: len(plant_rows)] = qrows + 0.01 * torch.randn(len(plant_rows), D, generator=g, device=dev)
    s, i, c = G.search(q, k)
    assert torch.equal(i[: len(plant_rows), 0], ids) and (c == k).all()
    assert float(s[: len(plant_rows), 0].min()) > 0.9 and float(s[len(plant_rows):, 0].max()) < 0.3
    assert (s[:, :-1] >= s[:, 1:]).all()
    qn = torch.nn.functional.normalize(q.double(), dim=-1)
    for j in (0, 7, 100, Q - 1):
        got = torch.cat([G.read(int(r), 1) for r in i[j]]).double()
        assert ((got @ qn[j]) - s[j].double()).abs().max().item() <= 2e-6
    # oracle on a sample: rows of four ranges against 16 of the queries
    qs = list(range(8)) + list(range(Q - 8, Q))
    for start in (0, 40_000, N // 3, N - 50_000):
        sub = G.read(start, 50_000).cpu().numpy()
        sc = osearch.cosine_scores(sub, qn[qs].float().cpu().numpy())
        for a, j in enumerate(qs):
            better = np.where(sc[a] > float(s[j, k - 1]) + 1e-6)[0] + start
            assert set(better.tolist()) <= set(i[j].cpu().tolist()), (start, j)
    # 8 shards searched apart and merged == unsharded (global row ids through index_offset)
    shard = N // 8
    parts_s, parts_i = [], []
    for p in range(8):
        Gp = engine.Gallery(D, shard, device=0)
        for s0 in range(0, shard, chunk):
            Gp.add(G.read(p * shard + s0, min(chunk, shard - s0)), normalize=False)
        ps, pi, _ = Gp.search(q, k, index_offset=p * shard)
        parts_s.append(ps); parts_i.append(pi)
        Gp.close()
    ms, mi, mc = engine.merge_topk(torch.stack(parts_s), torch.stack(parts_i), k)
    assert torch.equal(mi, i) and torch.equal(ms, s) and torch.equal(mc, c)
    plan = G.search_plan(Q, k)
    assert plan["scan256"] and plan["ksel"] == 64          # 10 M rows: the wide candidate lists (api.hip SEARCH_WIDE_ROWS)
    G.close()


def test_sharded_two_phase_with_certificate_equals_unsharded(dev):
    """Eight shards in one process through the whole protocol (revers-o_amd/sharded.py LocalShards: candidates, bound
    exchange or estimated level, bounded re-score with per-shard certificate bounds, packed merge with the cross-shard
    certificate, second exact round where needed) on a gallery of near-duplicate clusters that overflow the unsharded
    search's candidate lists: the merged result equals the unsharded search bit for bit (both are the exhaustive fp32
    search)."""
    from reverso_amd import sharded
    N, D, Q, k = 160000, 128, 64, 10
    rng = np.random.default_rng(77)
    gal = rng.standard_normal((N, D), dtype=np.float32)
    centres = rng.standard_normal((Q // 2, D), dtype=np.float32)
    pos = rng.permutation(N)[: (Q // 2) * 120].reshape(Q // 2, 120)
    for q in range(Q // 2):                                            # clusters of 120 spread over all shards
        gal[pos[q]] = centres[q][None] + 1e-3 * rng.standard_normal((120, D), dtype=np.float32)
    qr = np.concatenate([centres + 0.3 * rng.standard_normal((Q // 2, D), dtype=np.float32),
                         rng.standard_normal((Q // 2, D), dtype=np.float32)])
    gd, qd = torch.from_numpy(gal).to(dev), torch.from_numpy(qr).to(dev)
    G = engine.Gallery(D, N, device=0)
    G.add(gd)
    shards = []
    for p in range(8):
        Gp = engine.Gallery(D, N // 8, device=0)
        Gp.add(G.read(p * (N // 8), N // 8), normalize=False)
        shards.append(Gp)
    ls = sharded.LocalShards.from_galleries(shards)
    for thr in (None, 0.5):
        ref = G.search(qd, k, thr)
        assert G.search_stats()["uncertified"] >= Q // 2
        out = ls.search(qd, k, thr)
        # (up to round 3 the shards together re-scored ~64 rows per query and the clusters of 120 sent most queries through
        #  the second round; a shard that scans against the whole gallery's estimated level keeps and re-scores ALL its rows
        #  above that level -- every cluster member -- and the merge certifies them: the second round is exercised by
        #  test_shard_admission_estimate_that_is_far_too_high_... and the limit-50 tests instead)
        assert 0 <= ls.last_uncertified <= Q
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
    _check(ref if thr is None else G.search(qd, k, None), osearch.search(gal, qr, k), atol=1e-5, near_tie=3e-7)
    for Gp in shards:
        Gp.close()
    G.close()


def test_wide_candidate_lists_on_very_large_galleries(dev):
    """From 2^22 rows on the unsharded search keeps 64 candidates per query whatever k is (a failed certificate costs a
    whole pass there): same results as the brute-force mode, bit for bit, and the plan says so."""
    N, D, Q, k = (1 << 22) + 1000, 64, 33, 10
    g = torch.Generator(device=dev).manual_seed(4)
    G = engine.Gallery(D, N, device=0)
    for s0 in range(0, N, 1 << 20):
        G.add(torch.randn(min(1 << 20, N - s0), D, generator=g, device=dev))
    assert G.search_plan(Q, k)["ksel"] == 64
    q = torch.cat([G.read(123456, 8) + 0.05 * torch.randn(8, D, generator=g, device=dev), torch.randn(Q - 8, D, generator=g, device=dev)])
    out = G.search(q, k)
    st = G.search_stats()
    assert st["checked"] == Q
    Gx = _bruteforce_twin(G)
    ref = Gx.search(q, k)
    Gx.close()
    G.close()
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    assert out[1][:8, 0].cpu().tolist() == list(range(123456, 123464))


@pytest.mark.parametrize("Q", [64, 100, 200])
def test_collect_pass_appends_every_row_once_when_its_retry_ladder_deepens(dev, Q):
    """Many uncertified queries whose matches are dense in ONE 256-row gallery tile: the collecting pass recomputes the
    tile by column groups (its staging buffer holds 1024 survivors).  Column 0 and every odd column of the tile are exact
    duplicates of the (identical) queries: the first group of the two-group ladder (even columns: column 0 alone) fits
    and is appended, the second (odd columns) overflows and the ladder deepens -- column 0 must not be appended a second
    time, or the exact finish, which re-scores every list entry, returns that row twice and drops the true k-th hit."""
    N, D, k = 150000, 64, 10
    rng = np.random.default_rng(5 + Q)
    gal = rng.standard_normal((N, D), dtype=np.float32)
    v = rng.standard_normal(D).astype(np.float32)
    T = 256 * 400
    dup = [T] + [T + c for c in range(1, 256, 2)]
    gal[dup] = v
    qr = np.repeat(v[None], Q, axis=0)
    qr[Q - 3:] = rng.standard_normal((3, D), dtype=np.float32)         # a few ordinary queries ride along
    G = engine.Gallery(D, N, device=0)
    G.add(torch.from_numpy(gal).to(dev))
    qd = torch.from_numpy(qr).to(dev)
    s, i, c = G.search(qd, k)
    st = G.search_stats()
    assert st["uncertified"] >= Q - 3 and st["bruteforced"] == 0, st     # 129 rows within eps: collected, not brute-forced
    Gx = _bruteforce_twin(G)
    ref = Gx.search(qd, k)
    Gx.close()
    G.close()
    for a, b in zip((s, i, c), ref):
        assert torch.equal(a, b)
    want = sorted(dup)[:k]
    for q in range(Q - 3):
        assert i[q].cpu().tolist() == want, (q, i[q].cpu().tolist())
    _check((s, i, c), osearch.search(gal, qr, k), atol=1e-5, near_tie=3e-7)


@pytest.mark.parametrize("Q", [3, 70, 300])
def test_k50_is_resolved_from_the_scans_own_segments(dev, Q):
    """limit 50 (ui.py:342) on an ordinary gallery: 64 candidates leave the certificate less room (the 50th-to-64th score
    gap) than its error bound, so nearly every query fails it.  The scan runs with an admission margin of 2 eps for
    k > 25: everything an uncertified query can need is already in its segments, the exact finish re-scores that, and
    NO second pass over the gallery is made (from_segments == uncertified) -- with the exhaustive fp32 result, bit for
    bit, and the oracle's ranking."""
    N, D, k = 262144 + 777, 1024, 50
    g = torch.Generator(device=dev).manual_seed(77 + Q)
    G = engine.Gallery(D, N, device=0)
    for s0 in range(0, N, 65536):
        G.add(torch.randn(min(65536, N - s0), D, generator=g, device=dev))
    q = torch.randn(Q, D, generator=g, device=dev)
    q[0] = G.read(N // 2, 1)[0] + 0.05 * torch.randn(D, generator=g, device=dev)      # one planted neighbour
    for thr in (None, 0.08):
        out = G.search(q, k, thr)
        st = G.search_stats()
        assert st["checked"] == Q and st["uncertified"] >= (Q * 3) // 4, st          # the certificate does fail here
        assert st["from_segments"] == st["uncertified"] and st["bruteforced"] == 0, st
        assert st["collected_rows"] >= 50 * st["uncertified"], st
        Gx = _bruteforce_twin(G)
        ref = Gx.search(q, k, thr)
        Gx.close()
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
    assert int(out[1][0, 0]) == N // 2
    # k <= 25 keeps the plain scan (no margin): certified by the first pass on this gallery
    G.search(q, 20)
    st = G.search_stats()
    assert st["from_segments"] == 0 and st["uncertified"] <= max(1, Q // 20), st
    # the fp64 oracle over all rows for a few of the queries
    n16 = min(Q, 8)
    chunks = ((s0, G.read(s0, min(65536, N - s0)).cpu().numpy()) for s0 in range(0, N, 65536))
    rs, ri, rc = osearch.search_chunked(chunks, q[:n16].cpu().numpy(), k)
    gs, gi, gc = (t.cpu().numpy() for t in G.search(q[:n16], k))
    assert np.array_equal(gc, rc) and np.abs(gs - rs).max() <= 1e-5
    _assert_indices_equal_up_to_fp32_ties(gi, ri, rs, lambda r: G.read(r, 1).cpu().numpy()[0], q[:n16].cpu().numpy(), tie=6e-7)
    G.close()


@pytest.mark.parametrize("P,min_second_round", [(2, 24), (8, 1)])
def test_sharded_k50_second_round_draws_on_the_shards_segments(dev, P, min_second_round):
    """k = 50 over the shards of an ordinary gallery: the merge's certificate fails (two shards together re-score little
    more than 64 rows per query -- too few, as on one GPU -- so it fails for most queries; eight shards publish eight
    scores each, re-score about a hundred rows between them and fail for a few), so the protocol's second round runs --
    and every shard answers it from what its scan kept under the admission margin (from_segments), not by another pass
    over its rows.  Merged result == the unsharded search == the exhaustive fp32 scoring, bit for bit."""
    from reverso_amd import sharded
    N, D, Q, k = 8 * 40000, 1024, 48, 50
    g = torch.Generator(device=dev).manual_seed(5150)
    G = engine.Gallery(D, N, device=0)
    for s0 in range(0, N, 40000):
        G.add(torch.randn(40000, D, generator=g, device=dev))
    shards = []
    for p in range(P):
        Gp = engine.Gallery(D, N // P, device=0)
        Gp.add(G.read(p * (N // P), N // P), normalize=False)
        shards.append(Gp)
    q = torch.randn(Q, D, generator=g, device=dev)
    ls = sharded.LocalShards.from_galleries(shards)
    for thr in (None, 0.07):
        ref = G.search(q, k, thr)
        st = G.search_stats()
        assert st["uncertified"] >= Q // 2 and st["from_segments"] == st["uncertified"], st
        out = ls.search(q, k, thr)
        n2 = ls.last_uncertified
        assert n2 >= min_second_round                         # the second round did run
        for Gp in shards:
            sp = Gp.search_stats()                            # the shard's counters of that round
            assert sp["from_segments"] == n2 and sp["bruteforced"] == 0, sp
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
    Gx = _bruteforce_twin(G)
    bf = Gx.search(q, k, 0.07)
    Gx.close()
    for a, b in zip(ref, bf):
        assert torch.equal(a, b)
    for Gp in shards:
        Gp.close()
    G.close()


def test_segment_answered_query_with_more_rows_than_its_list_holds_goes_to_brute_force(dev):
    """3 000 exact duplicates of the query spread evenly over the gallery (every 87th row), limit 50: no slice's segment
    overflows (so the scan's segments DO hold every row that matters and the query is numbered among the
    segment-answered entries), but their number exceeds the 2 048-key collect list -- the exact finish passes the entry
    on to the brute-force pass.  Result: the 50 duplicates with the smallest indices, as the oracle has them."""
    N, D, k = 300000, 128, 50
    rng = np.random.default_rng(31)
    gal = rng.standard_normal((N, D), dtype=np.float32)
    v = rng.standard_normal(D).astype(np.float32)
    dup = np.arange(40057, N, 87)                 # behind the pre-pass rows (a pre-pass full of duplicates has its own bound
    assert len(dup) > 2048                        # at their score: then the segments alone cannot answer, and the search says so)
    gal[dup] = v
    qr = np.stack([v, rng.standard_normal(D).astype(np.float32), v * 2.0])
    G = engine.Gallery(D, N, device=0)
    G.add(torch.from_numpy(gal).to(dev))
    s, i, c = G.search(torch.from_numpy(qr).to(dev), k)
    st = G.search_stats()
    assert st["uncertified"] >= 2 and st["bruteforced"] == 2, st       # both duplicate queries: segments -> list overflow -> brute force
    assert st["from_segments"] >= 2, st
    assert i[0].cpu().tolist() == dup[:k].tolist() and i[2].cpu().tolist() == dup[:k].tolist()
    _check((s, i, c), osearch.search(gal, qr, k), atol=1e-5, near_tie=3e-7)
    G.close()


def test_k50_edges_smallest_scan256_gallery_and_no_fp32_rows(dev):
    """limit 50 where the margin machinery starts and where it must stay off: the smallest gallery the 256 x 256 scan takes
    (16 384 rows: a 4 096-row pre-pass, three slices) against the oracle, one row less (the small-gallery scan: no
    segments, the collect pass answers), and a gallery without fp32 rows (nothing to certify: the scan's own scores,
    no margin, stats say -1)."""
    D, k, Q = 256, 50, 40
    rng = np.random.default_rng(91)
    for N in (16384, 16383):
        gal = rng.standard_normal((N, D), dtype=np.float32)
        qr = rng.standard_normal((Q, D), dtype=np.float32)
        G = engine.Gallery(D, N, device=0)
        G.add(torch.from_numpy(gal).to(dev))
        out = G.search(torch.from_numpy(qr).to(dev), k)
        st = G.search_stats()
        assert st["checked"] == Q and st["bruteforced"] == 0, st
        if N < 16384:
            assert st["from_segments"] == 0, st
        else:
            assert st["from_segments"] == st["uncertified"], st
        _check(out, osearch.search(gal, qr, k), atol=1e-5, near_tie=3e-7)
        G.close()
    N = 40000
    gal = rng.standard_normal((N, D), dtype=np.float32)
    qr = rng.standard_normal((Q, D), dtype=np.float32)
    G = engine.Gallery(D, N, device=0, keep_f32=False)
    G.add(torch.from_numpy(gal).to(dev))
    s, i, c = G.search(torch.from_numpy(qr).to(dev), k)
    assert G.search_stats()["uncertified"] == -1
    rs, ri, rc = osearch.search(gal, qr, k)
    assert np.array_equal(c.cpu().numpy(), rc) and np.abs(s.cpu().numpy() - rs).max() <= 1e-2     # bf16-input scores
    assert (np.sort(i.cpu().numpy()[:, :10], axis=1) == np.sort(ri[:, :10], axis=1)).mean() >= 0.9
    G.close()


@pytest.mark.parametrize("Q", [193, 224, 256])
def test_193_to_256_queries_against_the_oracle(dev, Q):
    """193..256 queries (the 256-row form of the scan with a partly empty query tile; a two-128-row-tile form was measured
    and dropped): the oracle's results, with a tie group on the last query, a planted neighbour, a
    threshold, limit 10 and 50 (the latter with the admission margin), a ragged gallery size."""
    N, D = 70001, 128
    rng = np.random.default_rng(Q)
    gal = rng.standard_normal((N, D), dtype=np.float32)
    gal[N // 3: N // 3 + 40] = gal[N // 3]
    qr = rng.standard_normal((Q, D), dtype=np.float32)
    qr[Q - 1] = gal[N // 3]                                   # the last query of the second tile sits on the tie group
    qr[130] = gal[777] + 0.05 * rng.standard_normal(D).astype(np.float32)
    G = engine.Gallery(D, N, device=0)
    G.add(torch.from_numpy(gal).to(dev))
    qd = torch.from_numpy(qr).to(dev)
    for k, thr in ((10, None), (50, None), (10, 0.3)):
        out = G.search(qd, k, thr)
        _check(out, osearch.search(gal, qr, k, thr), atol=1e-5, near_tie=3e-7)
        assert out[1][Q - 1, :min(k, 40)].cpu().tolist() == list(range(N // 3, N // 3 + min(k, 40)))
        assert int(out[1][130, 0]) == 777
    G.close()


def test_shard_admission_estimate_that_is_far_too_high_still_gives_the_exhaustive_result(dev):
    """Shards of a row-sharded gallery start their scans from an ESTIMATE of the whole gallery's admission level (mean + z
    sigma of the shard's pre-pass scores, revo_search_set_total_rows).  Here the estimate is made useless on purpose: the
    first rows of shard 0 hold 300 near-copies of MINUS the query (scores ~ -0.95 among scores ~N(0, 0.09)), which blows
    the sample's sigma up and puts shard 0's estimate (~0.6) above its true matches (~0.55, further down the shard): its
    scan drops them, its list comes out below the estimate, the merge's certificate fails and the second round re-does
    the query exactly: merged result == the unsharded search,
    bit for bit; with a sane estimate (ordinary queries) no second round is needed."""
    from reverso_amd import sharded
    D, k, P, n = 128, 10, 2, 40000
    rng = np.random.default_rng(2718)
    gal = rng.standard_normal((P * n, D), dtype=np.float32)
    centre = rng.standard_normal(D).astype(np.float32)
    early = rng.choice(6000, size=300, replace=False)                            # inside shard 0's pre-pass rows
    gal[early] = -centre[None] + 0.33 * rng.standard_normal((300, D)).astype(np.float32)    # scores ~ -0.95: sigma blown up
    late = 20000 + np.arange(12)                                                  # the true matches (~0.55): shard 0, behind its pre-pass
    gal[late] = centre[None] + 1.5 * rng.standard_normal((12, D)).astype(np.float32)
    qr = np.concatenate([centre[None], rng.standard_normal((30, D), dtype=np.float32)])
    G = engine.Gallery(D, P * n, device=0)
    G.add(torch.from_numpy(gal).to(dev))
    shards = []
    for p in range(P):
        Gp = engine.Gallery(D, n, device=0)
        Gp.add(G.read(p * n, n), normalize=False)
        shards.append(Gp)
    ls = sharded.LocalShards.from_galleries(shards)                               # tells every shard the total row count
    qd = torch.from_numpy(qr).to(dev)
    ref = G.search(qd, k)
    out = ls.search(qd, k)
    assert ls.last_uncertified >= 1                                               # the poisoned query went through the second round
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    assert set(out[1][0].cpu().tolist()) == set(late[:k].tolist()) or set(out[1][0].cpu().tolist()) <= set(late.tolist())
    _check(out, osearch.search(gal, qr, k), atol=1e-5, near_tie=3e-7)
    out2 = ls.search(qd[1:], k)                                                   # ordinary queries: the estimate is sane
    assert ls.last_uncertified == 0
    for a, b in zip(out2, G.search(qd[1:], k)):
        assert torch.equal(a, b)
    for Gp in shards:
        Gp.close()
    G.close()


def test_shard_admission_estimate_on_clustered_and_on_light_tailed_galleries(dev):
    """The shards' admission estimate assumes a Gaussian tail.  (1) A clustered gallery (2 000 tight clusters, queries near
    cluster centres: a HEAVY tail -- a few dozen rows far above the bulk): the estimate comes out low, which costs
    survivors, not results, and no second round is needed.  (2) Rows and queries with +-1 coordinates (scores are a scaled
    binomial: bounded, a lighter tail than the Gaussian's far out): whatever the estimate does there -- measured: still no
    second round at 256 dimensions -- the result must be the exhaustive one.  Both: merged == unsharded, bit for bit."""
    from reverso_amd import sharded
    D, k, P, n = 256, 10, 4, 40000
    rng = np.random.default_rng(99)
    centres = rng.standard_normal((2000, D), dtype=np.float32)
    gal = centres[rng.integers(0, 2000, P * n)] + 0.25 * rng.standard_normal((P * n, D), dtype=np.float32)
    qr = centres[rng.integers(0, 2000, 200)] + 0.25 * rng.standard_normal((200, D), dtype=np.float32)
    light = np.sign(rng.standard_normal((P * n, D), dtype=np.float32))            # +-1 rows: scores of a +-1 query are a narrow binomial
    ql = np.sign(rng.standard_normal((64, D), dtype=np.float32))
    for name, g_np, q_np in (("clustered", gal, qr), ("light-tailed", light, ql)):
        G = engine.Gallery(D, P * n, device=0)
        G.add(torch.from_numpy(g_np).to(dev))
        shards = []
        for p in range(P):
            Gp = engine.Gallery(D, n, device=0)
            Gp.add(G.read(p * n, n), normalize=False)
            shards.append(Gp)
        ls = sharded.LocalShards.from_galleries(shards)
        qd = torch.from_numpy(q_np).to(dev)
        ref = G.search(qd, k)
        out = ls.search(qd, k)
        print(name, "uncertified after the merge:", ls.last_uncertified, "of", len(q_np))
        for a, b in zip(out, ref):
            assert torch.equal(a, b), name
        if name == "clustered":
            assert ls.last_uncertified <= len(q_np) // 20
        for Gp in shards:
            Gp.close()
        G.close()
