"""Where a kernel's spilled registers are written and reloaded (hipcc listing of one source file of revers-o_amd/csrc):
    python scripts/spill_map.py gemm.hip gemm256q_kernelILi0E [-- extra flags]
prints one line per basic block that touches scratch: loop depth, scratch stores / loads, and what else the block holds
(MFMAs, DMA requests, buffer stores), in program order."""
import os, re, subprocess, sys
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "revers-o_amd", "csrc")
args = sys.argv[1:]
extra = []
if "--" in args:
    extra = args[args.index("--") + 1:]; args = args[: args.index("--")]
src, pat = args[0], args[1]
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-S", "--cuda-device-only",
                      src, "-o", "-"] + extra, cwd=CSRC, capture_output=True, text=True)
asm = out.stdout
for m in re.finditer(r"^(_Z\S*" + re.escape(pat) + r"\S*):", asm, flags=re.M):
    body = asm[m.end(): asm.index(".Lfunc_end", m.end())].splitlines()
    print("====", m.group(1), len(body), "lines")
    blk, depth, st = None, 0, None
    rows = []
    for ln in body:
        if ln.startswith(".LBB") or ln.startswith("; %bb."):
            dm = re.search(r"Depth=(\d+)", ln); depth = int(dm.group(1)) if dm else 0
            st = {"blk": ln.split(":")[0].strip(), "d": depth, "st": 0, "ld": 0, "mfma": 0, "dma": 0, "bst": 0, "n": 0, "wait": []}
            rows.append(st)
        elif st is not None:
            st["n"] += 1
            st["st"] += "scratch_store" in ln; st["ld"] += "scratch_load" in ln; st["mfma"] += "v_mfma" in ln
            st["dma"] += ("buffer_load_dword" in ln and " lds" in ln); st["bst"] += "buffer_store_dwordx4" in ln
            w = re.search(r"s_waitcnt vmcnt\((\d+)\)", ln)
            if w: st["wait"].append(int(w.group(1)))
    for r in rows:
        if r["st"] or r["ld"] or "-v" in extra:
            print(f"{r['blk']:>12} depth {r['d']} n {r['n']:4d}  scratch st {r['st']:2d} ld {r['ld']:2d}  mfma {r['mfma']:2d} dma {r['dma']} bstore {r['bst']:2d} vmcnt {r['wait']}")
    tot = [sum(r[k] for r in rows) for k in ("st", "ld")]
    print("total scratch stores / loads:", tot, " in depth >= 2:", [sum(r[k] for r in rows if r["d"] >= 2) for k in ("st", "ld")])
