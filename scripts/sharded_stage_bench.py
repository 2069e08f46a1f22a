"""Per-stage times of the two-phase sharded search (sharded.py) rehearsed on ONE GPU: the 1 M x 1024 gallery as 8 shards
of 125 k rows, 10 000 replicated queries (BASELINE.json configs[3]).  Every shard's phases are timed with the library's
per-class events; the two all-gathers cannot be measured here (one GPU) and are entered as an assumption.
    python scripts/sharded_stage_bench.py [N] [Q] [P]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd
from reverso_amd import engine
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
P = int(sys.argv[3]) if len(sys.argv) > 3 else 8
D, k = 1024, 10
g = torch.Generator(device=dev).manual_seed(42)
full = engine.Gallery(D, N, device=0)
for s in range(0, N, 125_000):
    full.add(torch.randn(min(125_000, N - s), D, generator=g, device=dev))
shard = N // P
shards = []
for p in range(P):
    Gp = engine.Gallery(D, shard, device=0)
    Gp.add(full.read(p * shard, shard), normalize=False)
    if not os.environ.get("REVO_NO_EST"):          # (A/B: without the whole-gallery admission estimate, round 3's behaviour)
        Gp.set_total_rows(N)
    shards.append(Gp)
q = torch.randn(Q, D, generator=g, device=dev)
ksel = engine.search_ksel(k)
top_m = min(ksel, max(8, -(-min(64, 2 * ksel) // P)))
pb = engine.packed_bytes(Q, k)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    engine.prof_reset(); engine.prof_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); engine.prof_enable(False)
    prof = engine.prof_report()
    return e0.elapsed_time(e1) / reps, {c: v["ms"] / reps for c, v in prof.items()}

t1, st1 = timed(lambda: full.search(q, k))
# phase 1 on shard 0 (all shards are statistically alike), then the gathered bounds of all 8, phase 2 on shard 0, merge of 8
allb = torch.stack([Gp.search_candidates(q, k, top_m) for Gp in shards])
tc, stc = timed(lambda: shards[0].search_candidates(q, k, top_m))
packed = torch.empty((P * pb,), dtype=torch.uint8, device=dev)
for p, Gp in enumerate(shards):
    Gp.search_candidates(q, k, top_m)
    Gp.search_finish(Q, k, allb, None, p * shard, out_packed=packed[p * pb:(p + 1) * pb])
shards[0].search_candidates(q, k, top_m)
tf, stf = timed(lambda: shards[0].search_finish(Q, k, allb, None, 0, out_packed=packed[:pb]))
tfu, _ = timed(lambda: shards[0].search_finish(Q, k, None, None, 0, out_packed=packed[:pb]))      # no exchanged bounds: with the
# shards' admission estimate the lists are cut at the estimate instead (what sharded.py does then: one all-gather fewer)
tm, stm = timed(lambda: engine.merge_topk_packed(packed, P, Q, k))
s_ref, i_ref, c_ref = full.search(q, k)
for p, Gp in enumerate(shards):
    Gp.search_candidates(q, k, top_m)
    Gp.search_finish(Q, k, allb, None, p * shard, out_packed=packed[p * pb:(p + 1) * pb])
s2, i2, c2 = engine.merge_topk_packed(packed, P, Q, k)
kept = int((packed.view(P, pb)[:, : Q * k * 8].contiguous().view(torch.int64) >= 0).sum())
EST = not os.environ.get("REVO_NO_EST")
# assumption: latency-bound RCCL all-gathers at ~60 us each on xGMI -- two (admission scores 0.3 MB, packed top-k 1.2 MB per
# rank), or only the second when the shards scan against the estimated level (sharded.py leaves the first out then)
comm_ms = (1 if EST else 2) * 0.06
# the certificate's second round: the whole protocol in one process over the 8 shards tells how many queries
# need it; its per-rank cost = the exact pass of ONE shard for those queries (+ a third, small all-gather when it happens)
from reverso_amd import sharded
ls = sharded.LocalShards.from_galleries(shards)
s3, i3, c3 = ls.search(q, k)
n_unc = ls.last_uncertified
full_unc = None
full.search(q, k); full_unc = full.search_stats()["uncertified"]
t_exact = 0.0
if n_unc:
    qi = torch.arange(n_unc, dtype=torch.int32, device=dev)
    need = torch.full((n_unc,), 0.12, device=dev)
    shards[0].search_candidates(q, k, top_m)
    t_exact, _ = timed(lambda: shards[0].search_exact(qi, need, k, 0))
    comm_ms += 0.06
t8 = tc + (tfu if EST else tf) + tm + comm_ms + t_exact
print(json.dumps({
    "N": N, "Q": Q, "shards": P, "top_m": top_m,
    "one_gpu_ms": round(t1, 4), "one_gpu_stage_ms": {c: round(v, 4) for c, v in sorted(st1.items())},
    "per_rank_phase1_ms": round(tc, 4), "phase1_stage_ms": {c: round(v, 4) for c, v in sorted(stc.items())},
    "per_rank_finish_bounded_ms": round(tf, 4), "finish_unbounded_ms": round(tfu, 4), "merge_ms": round(tm, 4),
    "assumed_comm_ms": comm_ms, "shards_estimate_the_whole_gallerys_level": EST, "projected_8gpu_ms": round(t8, 4), "projected_speedup": round(t1 / t8, 3),
    "uncertified_queries_sharded": n_unc, "uncertified_queries_one_gpu": full_unc, "per_rank_exact_round_ms": round(t_exact, 4),
    "results_equal_unsharded": bool(torch.equal(i3, i_ref) and torch.equal(s3, s_ref) and torch.equal(c3, c_ref)),
    "first_round_equal_unsharded": bool(torch.equal(i2, i_ref) and torch.equal(s2, s_ref) and torch.equal(c2, c_ref)),
    "results_kept_per_query_all_ranks": round(kept / Q, 2)}), flush=True)
