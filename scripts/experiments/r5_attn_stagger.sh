#!/bin/bash
# Round 5: the half-tile stagger of waves 4-7 of the attention workgroup (attn_fwd_kernel<64, 8, 0, STAG = true>, experiment
# library, REVO_ATTN_STAG=1) against the product kernel: time and max error per sequence length, alternated.
for r in 1 2 3; do
  for S in 577 1024 197; do
    echo "--- S=$S in step"; python scripts/attn_bench.py 64 $S 16 2>&1 | grep -E "attention|max err" | tr '\n' ' '; echo
    echo "--- S=$S staggered"; REVO_ATTN_STAG=1 python scripts/attn_bench.py 64 $S 16 2>&1 | grep -E "attention|max err" | tr '\n' ' '; echo
  done
done
