#!/bin/bash
# Round 6: the gallery stream of the small-query scan (<= 128 queries: every gallery row is read by ONE workgroup, once) with
# non-temporal LDS-DMA loads (shipped: topk256.hip S256_ONE_TILE_AUX = 2) against the default policy
# (make -C revers-o_amd/csrc var VARFLAGS=-DS256_ONE_TILE_AUX=0), alternated on one box.
set -e
for r in 1 2; do
  REVO_EXPERIMENTS=1 REVO_LIBRARY_PATH=revers-o_amd/librevo_var.so python scripts/search_small_q.py > gpurun_out/scan_nt_default_$r.json 2>/dev/null
  REVO_EXPERIMENTS=1 REVO_LIBRARY_PATH=revers-o_amd/librevo_exp.so python scripts/search_small_q.py > gpurun_out/scan_nt_nt_$r.json 2>/dev/null
done
python - <<'PY'
import json
for r in (1, 2):
    for t in ("default", "nt"):
        d = json.loads([l for l in open(f"gpurun_out/scan_nt_{t}_{r}.json") if l.startswith("{")][-1])
        print(r, t, [(x["Q"], x["k"], x["search_ms"], x["stage_ms"]["topk_scan"]) for x in d["rows"] if x["k"] == 10])
PY
