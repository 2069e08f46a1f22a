#!/bin/bash
# Knock-out study of the body attention kernel at the headline shape (64 x 16 heads, S = 577, head_dim 64): the kernel with
# parts compiled in but switched off (librevo_exp.so, REVO_ATTN_DBG: WRONG RESULTS, timing only).
#  1 no K/V DMA + wait   2 no per-tile barrier   4 no exponentials   8 no PV MFMAs   16 no score MFMAs
#  32 no row sums (16 v_pk_add_f32 per tile)   64 no scale-and-shift (16 v_pk_fma_f32 per tile)
for d in 0 1 2 3 4 8 16 24 28 7 31 32 64 96; do
  echo -n "dbg=$d  "
  REVO_ATTN_DBG=$d python scripts/attn_bench.py 2>/dev/null | head -1
done
echo "S=1024:"
for d in 0 1 3 4; do
  echo -n "dbg=$d  "
  REVO_ATTN_DBG=$d python scripts/attn_bench.py 64 1024 16 2>/dev/null | head -1
done
