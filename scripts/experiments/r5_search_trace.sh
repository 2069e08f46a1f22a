#!/bin/bash
# kernel trace of one-query (and 64-query) searches: per-kernel durations and the gaps between consecutive kernels of a search
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for Q in 1 64; do
  rm -rf $REPO/gpurun_out/strace_$Q
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/strace_$Q -o t -- python3 $REPO/scripts/search_one_query_trace.py $Q > $REPO/gpurun_out/strace_$Q.log 2>&1
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for Q in (1, 64):
    f = glob.glob("gpurun_out/strace_%d/**/t_kernel_trace.csv" % Q, recursive=True)
    if not f: print("no trace", Q); continue
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-20 * 12:]                      # the timed searches are the last ones
    # one search = from an l2norm_rows launch to the next
    starts = [i for i, r in enumerate(rows) if "l2norm_rows" in r["Kernel_Name"]]
    per = collections.OrderedDict(); gaps = collections.OrderedDict(); spans = []
    for a, b in zip(starts[5:-1], starts[6:]):
        seq = rows[a:b]
        spans.append((int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e3)
        for i, r in enumerate(seq):
            n = r["Kernel_Name"].split("(")[0].replace("void revo::", "").replace("revo::", "")[:40] + "#%d" % i
            per.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            if i: gaps.setdefault(n, []).append((int(r["Start_Timestamp"]) - int(seq[i - 1]["End_Timestamp"])) / 1e3)
    print("Q =", Q, "searches", len(spans), "first-start to last-end us: %.1f" % (sum(spans) / len(spans)))
    for n, v in per.items():
        g = gaps.get(n)
        print("  %-46s run %7.1f us   gap before %6.1f us" % (n, sum(v) / len(v), (sum(g) / len(g)) if g else 0.0))
PY
find gpurun_out/strace_1 gpurun_out/strace_64 -name "*.db" -delete 2>/dev/null || true
