#!/bin/bash
# Round 5: where the folded LayerNorm's consumer side spends its ~1.2 us per tile (timing-only builds, WRONG RESULTS).
# Libraries: for a in 1 2 4 7; do rm -rf revers-o_amd/csrc/build/var; make -C revers-o_amd/csrc -j8 var VARFLAGS=-DREVO_LNC_ABLATE=$a; cp revers-o_amd/librevo_var.so _bisect/librevo_abl$a.so; done
#   (1 = no merge of the statistics, 2 = no epilogue arithmetic, 4 = no statistics DMA)
for r in 1 2; do
 echo "--- product"; python scripts/ln_fold_probe.py 2>&1 >/dev/null | tail -2
 for a in 1 2 4 7; do echo "--- ablate $a (1 no merge, 2 no epilogue math, 4 no statistics DMA)"; REVO_LIBRARY_PATH=_bisect/librevo_abl$a.so python scripts/ln_fold_probe.py 2>&1 >/dev/null | tail -2; done
done
