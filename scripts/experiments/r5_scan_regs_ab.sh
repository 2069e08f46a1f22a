for r in 1 2; do
REVO_LIBRARY_PATH=_bisect/head/librevo.so python scripts/scan_runs_ab.py 1000000 160 192 2>/dev/null | tail -1 >> gpurun_out/runs_192_old.txt
python scripts/scan_runs_ab.py 1000000 160 192 2>/dev/null | tail -1 >> gpurun_out/runs_192_new.txt
TOPK=50 REVO_LIBRARY_PATH=_bisect/head/librevo.so python scripts/scan_runs_ab.py 1000000 256 2048 10000 2>/dev/null | tail -1 >> gpurun_out/runs_k50_old.txt
TOPK=50 python scripts/scan_runs_ab.py 1000000 256 2048 10000 2>/dev/null | tail -1 >> gpurun_out/runs_k50_new.txt
done
python - <<'PY'
import json
for t in ("192_old","192_new","k50_old","k50_new"):
    for l in open("gpurun_out/runs_%s.txt"%t):
        d=json.loads(l); print(t,[(r["Q"],r["search_ms"],r["scan_ms"],r["checksum"]) for r in d["rows"]])
PY
