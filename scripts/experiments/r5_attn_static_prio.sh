#!/bin/bash
# Round 5, no gain: static priority for half of the attention workgroup's waves (MI355X guide, "static priority for the younger half").
# Libraries: for v in 1 2; do rm -rf revers-o_amd/csrc/build/var; make -C revers-o_amd/csrc -j8 var VARFLAGS=-DREVO_ATTN_PRIO=$v; cp revers-o_amd/librevo_var.so _bisect/librevo_attnprio$v.so; done
for r in 1 2 3; do
 echo "--- exp (no static priority)"; python scripts/attn_bench.py 2>&1 | grep attention
 echo "--- waves 4-7 at priority 1"; REVO_LIBRARY_PATH=_bisect/librevo_attnprio1.so python scripts/attn_bench.py 2>&1 | grep attention
 echo "--- waves 0-3 at priority 1"; REVO_LIBRARY_PATH=_bisect/librevo_attnprio2.so python scripts/attn_bench.py 2>&1 | grep attention
done
