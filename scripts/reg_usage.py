"""hipcc's own register report for one source file of revers-o_amd/csrc (cross-compiles without a GPU):
    python scripts/reg_usage.py gemm.hip [substring ...] [-- extra hipcc flags]
prints   VGPRs  spilled  scratch-bytes  kernel   for every kernel whose (demangled) name contains all substrings."""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "revers-o_amd", "csrc")
args = sys.argv[1:]
extra = []
if "--" in args:
    extra = args[args.index("--") + 1:]
    args = args[: args.index("--")]
src, subs = args[0], args[1:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-c", src,
                      "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"] + extra, cwd=CSRC, capture_output=True, text=True)
if out.returncode:
    sys.exit(out.stderr[-4000:])
cur, rows = None, {}
for line in out.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
    for key, pat in (("v", r" VGPRs: (\d+)"), ("a", r"AGPRs: (\d+)"), ("s", r"VGPRs Spill: (\d+)"), ("ss", r"SGPRs Spill: (\d+)"),
                     ("sc", r"ScratchSize \[bytes/lane\]: (\d+)"), ("sg", r" SGPRs: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur:
            rows[cur][key] = int(m.group(1))
names = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'VGPR':>5} {'spill':>5} {'scratch':>7} {'SGPR':>5} {'sspill':>6} {'occ':>3}  kernel")
for mangled, name in zip(rows, names):
    if all(s in name for s in subs):
        r = rows[mangled]
        short = re.sub(r"\(revo::\w+(, int)*\)$|void revo::", "", name)
        print(f"{r.get('v', -1):5d} {r.get('s', -1):5d} {r.get('sc', -1):7d} {r.get('sg', -1):5d} {r.get('ss', -1):6d} {r.get('occ', -1):3d}  {short}")
