for r in 1 2; do
 echo "--- product"; python scripts/ln_fold_probe.py 2>&1 >/dev/null | tail -2
 for a in 1 2 4 7; do echo "--- ablate $a (1 no merge, 2 no epilogue math, 4 no statistics DMA)"; REVO_LIBRARY_PATH=_bisect/librevo_abl$a.so python scripts/ln_fold_probe.py 2>&1 >/dev/null | tail -2; done
done
