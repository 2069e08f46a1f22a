for r in 1 2 3; do
 echo "--- exp (no static priority)"; python scripts/attn_bench.py 2>&1 | grep attention
 echo "--- waves 4-7 at priority 1"; REVO_LIBRARY_PATH=_bisect/librevo_attnprio1.so python scripts/attn_bench.py 2>&1 | grep attention
 echo "--- waves 0-3 at priority 1"; REVO_LIBRARY_PATH=_bisect/librevo_attnprio2.so python scripts/attn_bench.py 2>&1 | grep attention
done
