#!/bin/bash
# effective clock of the persistent GEMM on the whole chip against half of it (MI355X guide, 'DVFS give-back': GRBM_GUI_ACTIVE / 8 /
# kernel wall time); durations from the kernel trace of the same program, counters from a separate --pmc pass
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export REVO_EXPERIMENTS=1
for n in 32 16; do
  if [ $n = 16 ]; then export REVO_GEMM_NSLOT=16; else unset REVO_GEMM_NSLOT; fi
  rm -rf $REPO/gpurun_out/hc_pmc_$n $REPO/gpurun_out/hc_tr_$n
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/hc_pmc_$n -o p -- python3 $REPO/scripts/experiments/r5_gemm_half_chip.py > $REPO/gpurun_out/hc_pmc_$n.log 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/hc_tr_$n -o t -- python3 $REPO/scripts/experiments/r5_gemm_half_chip.py > $REPO/gpurun_out/hc_tr_$n.log 2>&1
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for n in (32, 16):
    f = glob.glob("gpurun_out/hc_pmc_%d/**/p_counter_collection.csv" % n, recursive=True)
    g = glob.glob("gpurun_out/hc_tr_%d/**/t_kernel_stats.csv" % n, recursive=True)
    if not f or not g: print("missing", n); continue
    cyc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "gemm256p_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE": cyc[r["Kernel_Name"].split("(")[0][-30:]].append(float(r["Counter_Value"]))
    dur = {r["Name"].split("(")[0][-30:]: float(r["AverageNs"]) for r in csv.DictReader(open(g[0])) if "gemm256p_kernel" in r["Name"]}
    for k, v in cyc.items():
        c = sum(v) / len(v)
        if k in dur: print("nslot", n, k, "avg us %.1f" % (dur[k] / 1e3), "GRBM_GUI_ACTIVE/8 %.3e" % (c / 8), "-> clock GHz %.2f" % (c / 8 / dur[k]))
PY
find gpurun_out/hc_pmc_32 gpurun_out/hc_pmc_16 gpurun_out/hc_tr_32 gpurun_out/hc_tr_16 -name "*.db" -delete 2>/dev/null || true
