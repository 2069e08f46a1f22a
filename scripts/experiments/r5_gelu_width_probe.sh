#!/bin/bash
# Round 5: GELU on two (product) or four (var build) fragments side by side in the fc1 epilogue -- same time; two shipped.
# Library: rm -rf revers-o_amd/csrc/build/var; make -C revers-o_amd/csrc -j8 var VARFLAGS=-DREVO_GELU_WIDTH=4
# (_bisect/prev/librevo.so: the previous commit's library, scripts/step_regression_ab.sh says how to build it)
python -m pytest tests/test_gpu_ln_fold.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r05_lnfold_test.log 2>&1; tail -2 gpurun_out/r05_lnfold_test.log
for r in 1 2; do
 echo "--- width2 (product)"; python scripts/ln_fold_probe.py 2>&1 >/dev/null | tail -2
 echo "--- width4 (var)"; REVO_LIBRARY_PATH=revers-o_amd/librevo_var.so python scripts/ln_fold_probe.py 2>&1 >/dev/null | tail -2
 echo "--- previous commit"; REVO_LIBRARY_PATH=_bisect/prev/librevo.so python scripts/ln_fold_probe.py 2>&1 >/dev/null | tail -3
done
