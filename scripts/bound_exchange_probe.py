"""What would an exchange of the admission bound between the shards buy the shard scan?  (previous verdict, item 4b)
Rehearsal on ONE GPU with the experiment library: the 1 M x 1024 gallery as 8 shards, 10 000 queries.  The scan of
shard 0 is timed with its admission bounds seeded (revo_debug_seed_bounds) by what an all-gather could have told it:
  pre    the ksel-th best scan score over the union of all shards' pre-pass rows (an exchange right after the pre-pass)
  20%    the same over the pre-pass rows + the first fifth of every shard's scan (an exchange in the middle of the scan,
         applied from the start here: an upper bound of what it can give)
  full   the ksel-th best over the whole gallery (no exchange can know more)
and the whole protocol is run with the `pre` seeds on every shard and compared with the unsharded search.
    python scripts/bound_exchange_probe.py"""
import os, sys, json
os.environ.setdefault("REVO_EXPERIMENTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, reverso_amd
from reverso_amd import engine, sharded, _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
N, Q, P, D, k = 1_000_000, 10_000, 8, 1024, 10
g = torch.Generator(device=dev).manual_seed(42)
full = engine.Gallery(D, N, device=0)
for s in range(0, N, 125_000):
    full.add(torch.randn(125_000, D, generator=g, device=dev))
shard = N // P
shards = []
for p in range(P):
    Gp = engine.Gallery(D, shard, device=0)
    Gp.add(full.read(p * shard, shard), normalize=False)
    shards.append(Gp)
q = torch.randn(Q, D, generator=g, device=dev)
ksel = engine.search_ksel(k)
top_m = min(ksel, max(8, -(-min(64, 2 * ksel) // P)))
n_pre = shards[0].search_plan(Q, k)["prepass_rows"]

def union_bound(rows):
    G = engine.Gallery(D, P * rows, device=0)
    for p in range(P):
        G.add(full.read(p * shard, rows), normalize=False)
    return G.search_candidates(q, k, ksel)[:, ksel - 1].contiguous()

seeds = {"none": None, "pre": union_bound(n_pre), "20%": union_bound(n_pre + (shard - n_pre) // 5),
         "full": full.search_candidates(q, k, ksel)[:, ksel - 1].contiguous()}

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    engine.prof_reset(); engine.prof_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); engine.prof_enable(False)
    return e0.elapsed_time(e1) / reps, {c: v["ms"] / reps for c, v in engine.prof_report().items()}

out = {"N": N, "Q": Q, "shards": P, "pre_pass_rows": n_pre, "phase1_ms": {}, "scan_ms": {}}
for rnd in range(2):                       # alternated twice
    for name, sd in seeds.items():
        _lib.check(lib.revo_debug_seed_bounds(shards[0]._h, _lib.ptr(sd) if sd is not None else None))
        t, st = timed(lambda: shards[0].search_candidates(q, k, top_m))
        out["phase1_ms"].setdefault(name, []).append(round(t, 4))
        out["scan_ms"].setdefault(name, []).append(round(st.get("topk_scan", 0.0), 4))
_lib.check(lib.revo_debug_seed_bounds(shards[0]._h, None))
# the protocol with the `pre` seeds on every shard == the unsharded search, bit for bit
for Gp in shards:
    _lib.check(lib.revo_debug_seed_bounds(Gp._h, _lib.ptr(seeds["pre"])))
ls = sharded.LocalShards.from_galleries(shards)
s2, i2, c2 = ls.search(q, k)
s1, i1, c1 = full.search(q, k)
out["seeded_protocol_equals_unsharded"] = bool(torch.equal(i1, i2) and torch.equal(s1, s2) and torch.equal(c1, c2))
out["seeded_uncertified"] = int(ls.last_uncertified)
for Gp in shards:
    _lib.check(lib.revo_debug_seed_bounds(Gp._h, None))
print(json.dumps(out))
