"""Order of the vector-memory operations and waits of one kernel in a hipcc listing (-S output):
    python scripts/vmem_order.py /tmp/gemm.s gemm256q_kernelILi1E [first_label]
one line per run of equal instructions: what the counted vmcnt waits of the queued-stores kernel have to line up with."""
import re, sys
asm = open(sys.argv[1]).read()
m = re.search(r"^(_Z\S*" + re.escape(sys.argv[2]) + r"\S*):", asm, flags=re.M)
body = asm[m.end(): asm.index(".Lfunc_end", m.end())].splitlines()
def kind(l):
    l = l.strip()
    if l.startswith("s_waitcnt") and "vmcnt" in l: return l.split(";")[0].strip()
    if "scratch_" in l: return "SCRATCH " + l.split()[0]
    if l.startswith("buffer_load") and " lds" in l: return "dma"
    if l.startswith("buffer_load"): return "buffer_load"
    if l.startswith("buffer_store"): return "buffer_store" + (" nt" if " nt" in l else "")
    if l.startswith("global_load"): return "global_load"
    if l.startswith("global_store"): return "global_store"
    if l.startswith("global_atomic"): return "global_atomic"
    if l.startswith("s_barrier"): return "barrier"
    if l.startswith("v_mfma"): return "mfma"
    if l.startswith("ds_read"): return "ds_read"
    if l.startswith("ds_write"): return "ds_write"
    if l.startswith(".LBB") or l.startswith("; %bb."): return "@" + l.split(":")[0] + (" d" + re.search(r"Depth=(\d+)", l).group(1) if "Depth=" in l else "")
    if l.startswith("s_cbranch") or l.startswith("s_branch"): return l.split(";")[0].strip()
    return None
out, last, n = [], None, 0
for l in body:
    k = kind(l)
    if k is None: continue
    if k == last and not k.startswith("@"): n += 1; continue
    if last is not None: out.append(f"{last} x{n}" if n > 1 else last)
    last, n = k, 1
out.append(f"{last} x{n}" if n > 1 else last)
print("\n".join(out))
