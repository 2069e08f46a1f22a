"""Repro of test_scan_bound_histogram_edge_cases case (c) with the experiment switches (SCAN_DBG) available."""
import os, sys
os.environ.setdefault("REVO_EXPERIMENTS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes
import reverso_amd
from reverso_amd import engine, _lib
from oracle import search as osearch
dev = torch.device("cuda", 0)
dbg = int(os.environ.get("SCAN_DBG", "0"))
_lib.load().revo_op_set_gemm_debug(((dbg & 7) << 13) | ((dbg >> 3) << 20))
k = int(os.environ.get("TOPK", "50")); D, Q = 128, int(os.environ.get("NQ", "300"))
rng = np.random.default_rng(100 + k)
# consume the generator like the test does for (a) and (b)
N = 90000
gal = rng.standard_normal((N, D), dtype=np.float32); qr = rng.standard_normal((Q if Q == 300 else 300, D), dtype=np.float32)
for j in range(12):
    lvl = np.sqrt(0.01 + 0.5 * np.arange(150, dtype=np.float32) / 150)[:, None]
    _ = rng.standard_normal((150, D), dtype=np.float32)
N = 50000
base = rng.standard_normal(D).astype(np.float32)
qr = base[None] + 0.5 * rng.standard_normal((300, D), dtype=np.float32)
_ = rng.standard_normal((N, D), dtype=np.float32)
gal = np.concatenate([-(base[None] + 2.0 * rng.standard_normal((20000, D), dtype=np.float32)),
                      rng.standard_normal((30000, D), dtype=np.float32)])
qr = qr[:Q]
G = engine.Gallery(D, len(gal), device=0)
G.add(torch.from_numpy(gal).to(dev))
rs, ri, rc = osearch.search(gal, qr, k)
bad_total = 0
for rep in range(3):
    s, i, c = (t.cpu().numpy() for t in G.search(torch.from_numpy(qr).to(dev), k))
    bad = [q for q in range(Q) if sorted(i[q].tolist()) != sorted(ri[q].tolist())]
    bad_total += len(bad)
    msg = []
    for q in bad[:6]:
        missing = sorted(set(ri[q].tolist()) - set(i[q].tolist()))
        extra = sorted(set(i[q].tolist()) - set(ri[q].tolist()))
        msg.append((q, missing, extra))
    print("rep", rep, "bad queries", len(bad), msg, flush=True)
    if bad and os.environ.get("DUMP", "1") == "1":
        # layout of api.hip's workspace for this search (hist | per part: counts, segments | prelist | final lists | scores)
        lib = _lib.load()
        total = lib.revo_debug_read_workspace(G._h, 0, 0, None)
        buf = np.zeros(total, np.uint8)
        lib.revo_debug_read_workspace(G._h, 0, total, buf.ctypes.data_as(ctypes.c_void_p))
        up = lambda x: (x + 255) // 256 * 256
        ksel, NB = 64 if k > 16 else 32, 64
        n_pre = G.search_plan(Q, k)["prepass_rows"]
        for q_main in ((Q - Q % 256) if (Q > 256 and Q % 256) else Q, Q):      # the library splits a ragged tail only when cheaper
            parts = [(0, q_main)] + ([(q_main, Q - q_main)] if q_main < Q else [])
            off = up(Q * NB * 4)
            info = []
            for (q0, nq) in parts:
                sp = G.search_plan(nq, k)["slices"]
                cnt_off = off; off = up(off + nq * sp * 4)
                seg_off = off; off = up(off + nq * sp * 2 * ksel * 8)
                info.append((q0, nq, sp, cnt_off, seg_off))
            ok = all(0 <= int(buf[co: co + nq * sp * 4].view(np.int32).min()) and int(buf[co: co + nq * sp * 4].view(np.int32).max()) <= 2 * ksel
                     for (_, nq, sp, co, _) in info)
            if ok: break
        pre_off = off; fin_off = up(pre_off + Q * ksel * 8)
        print("  layout", info, "pre_off", pre_off, "fin_off", fin_off, "total", total)
        for (q0_, nq_, sp_, co_, so_) in info:
            cc_ = buf[co_: co_ + nq_ * sp_ * 4].view(np.int32)
            print("   part", q0_, nq_, sp_, "count range", int(cc_.min()), int(cc_.max()))
        q = bad[0]
        miss = sorted(set(ri[q].tolist()) - set(i[q].tolist()))[0]
        for (q0, nq, sp, cnt_off, seg_off) in info:
            if not (q0 <= q < q0 + nq): continue
            cnt = buf[cnt_off: cnt_off + nq * sp * 4].view(np.int32).reshape(nq, sp)[q - q0]
            seg = buf[seg_off: seg_off + nq * sp * 2 * ksel * 8].view(np.uint64).reshape(nq, sp, 2 * ksel)[q - q0]
            found = []
            for s_ in range(sp):
                idxs = (~seg[s_, :cnt[s_]]).astype(np.uint32)
                if miss in idxs.tolist(): found.append((s_, int(cnt[s_]), int(np.where(idxs == miss)[0][0])))
            print("  query", q, "missing row", miss, "slices", sp, "counts", cnt.tolist()[:60])
            print("  missing row present in segments:", found)
        fin = buf[fin_off: fin_off + Q * ksel * 8].view(np.uint64).reshape(Q, ksel)[q]
        fidx = (~fin).astype(np.uint32).tolist()
        print("  in final list:", miss in fidx, "final list size", int((fin != 0).sum()), "oracle rank of missing:", ri[q].tolist().index(miss))
        pre = buf[pre_off: pre_off + Q * ksel * 8].view(np.uint64).reshape(Q, ksel)[q]
        print("  tile of missing row:", (miss - n_pre) // 256, "col", (miss - n_pre) % 256)
if dbg & 2:
    st = (ctypes.c_int64 * 8)(); _lib.load().revo_debug_scan_stats(st)
    print("stats: drains", st[0], "queued", st[1], "retry passes", st[2], "slow frags", st[3], "appended", st[4], "refreshes", st[5])
print("plan", G.search_plan(Q, k))
sys.exit(1 if bad_total else 0)
