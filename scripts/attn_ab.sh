#!/bin/bash
# A/B of the attention kernel's FAST form (scale-and-shift by accumulator init, row sums on the matrix pipe) against its
# reference form (REVO_ATTN_REF=1, experiment library): headline shape and two others, alternated.
for r in 1 2; do
  for S in 577 1024 197; do
    B=64; [ $S = 197 ] && B=128
    echo -n "ref  "; REVO_ATTN_REF=1 python scripts/attn_bench.py $B $S 16 2>/dev/null | tr '\n' ' '; echo
    echo -n "fast "; python scripts/attn_bench.py $B $S 16 2>/dev/null | tr '\n' ' '; echo
  done
done
