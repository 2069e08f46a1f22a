#!/bin/bash
# bf16 epilogue slabs: 128-byte rows with an XOR swizzle against the 144-byte padded rows (the library of the commit before):
# GEMM tests, the LDS bank-conflict counters of the body GEMMs in bench.py's step, and the step time alternated.
OLD=${1:-_bisect/head/librevo.so}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ln_fold.py tests/test_gpu_l14_golden.py -x -q -m gpu > gpurun_out/slab_tests.log 2>&1; tail -2 gpurun_out/slab_tests.log
cd /tmp && export TMPDIR=/tmp
for t in new old; do
  rm -rf $REPO/gpurun_out/slab_pmc_$t
  if [ $t = old ]; then export REVO_LIBRARY_PATH=$REPO/$OLD; else unset REVO_LIBRARY_PATH; fi
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $REPO/gpurun_out/slab_pmc_$t -o p -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --ingest-images 0 --search-queries 0 > $REPO/gpurun_out/slab_pmc_$t.log 2>&1
done
unset REVO_LIBRARY_PATH
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for t in ("old", "new"):
    f = glob.glob("gpurun_out/slab_pmc_%s/**/p_counter_collection.csv" % t, recursive=True)
    if not f: print("no csv", t); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        if "gemm256p_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-34:]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in sorted(acc.items()):
        print(t, k, "conflict %% of LDS-active: %.2f" % (100 * c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1)))
PY
find gpurun_out/slab_pmc_old gpurun_out/slab_pmc_new -name "*.db" -delete 2>/dev/null || true
bash scripts/step_regression_ab.sh $OLD
