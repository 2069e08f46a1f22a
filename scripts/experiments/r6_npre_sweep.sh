#!/bin/bash
# pre-pass size of small-query searches re-swept with the non-temporal scan in place (REVO_NPRE: experiment build)
for n in 8192 16384 32768 65536; do
  REVO_EXPERIMENTS=1 REVO_NPRE=$n python scripts/search_small_q.py 2>/dev/null > gpurun_out/npre_$n.json
done
python - <<'PY'
import json
for n in (8192, 16384, 32768, 65536):
    d=json.loads([l for l in open(f"gpurun_out/npre_{n}.json") if l.startswith("{")][-1])
    print(n, [(x["Q"], x["search_ms"]) for x in d["rows"] if x["k"] == 10 and x["Q"] <= 256])
PY
