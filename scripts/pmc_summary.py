"""Summarise three rocprofv3 --pmc passes (SQ counters, FETCH_SIZE, WRITE_SIZE) per kernel.
    python scripts/pmc_summary.py gpurun_out/prof_<tag> [profiles/roofline_traffic.json] > profiles/rNN_pmc_summary.csv
(directories as written by scripts/collect_profiles.sh; the optional second argument also writes the
HBM-side bytes per launch of the body GEMM kernels that bench.py reports as roofline.traffic)
FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read; MI355X_MICROARCH.md §HBM)."""
import csv, collections, glob, sys
root = sys.argv[1]
def agg(pattern):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d
sq, fe, wr = agg(root + "/sq/**/*counter_collection.csv"), agg(root + "/fetch/**/*counter_collection.csv"), agg(root + "/write/**/*counter_collection.csv")
mean = lambda l: sum(l) / len(l) if l else 0.0
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches", "SQ_WAVE_CYCLES", "wait_any_pct", "wait_inst_pct", "active_inst_pct", "SQ_VALU_MFMA_BUSY_CYCLES",
            "lds_bank_conflict_pct_of_lds_active", "fetch_MiB_per_launch_x2_corrected", "write_MiB_per_launch"])
for n in sorted((k for k in sq if "revo::" in k), key=lambda n: -sum(sq[n].get("SQ_WAVE_CYCLES", [0]))):
    c = sq[n]; wc = mean(c["SQ_WAVE_CYCLES"]) or 1.0
    w.writerow([n, len(c["SQ_WAVE_CYCLES"]), f"{wc:.4g}", f"{100 * mean(c['SQ_WAIT_ANY']) / wc:.1f}", f"{100 * mean(c['SQ_WAIT_INST_ANY']) / wc:.1f}",
                f"{100 * mean(c['SQ_ACTIVE_INST_ANY']) / wc:.1f}", f"{mean(c['SQ_VALU_MFMA_BUSY_CYCLES']):.4g}",
                f"{100 * mean(c['SQ_LDS_BANK_CONFLICT']) / (mean(c['SQ_LDS_IDX_ACTIVE']) or 1):.2f}",
                f"{2 * mean(fe[n].get('FETCH_SIZE', [0])) / 1024:.1f}", f"{mean(wr[n].get('WRITE_SIZE', [0])) / 1024:.1f}"])

if len(sys.argv) > 2:
    import json
    tot, n = 0.0, 0
    for k in sq:
        # one "launch" of bench.py's roofline = one linear layer = the persistent / per-tile 256x256 kernel with the
        # GELU (1), residual (2) or RoPE (5) epilogue, plus the kernels that take its leftover rows
        main = "revo::gemm256" in k and any(t in k for t in ("<1>", "<2>", "<5>", "<1,", "<2,", "<5,")) and "<3, 4>" not in k
        tail = ("gemm256_kernel<3, 4>" in k or "gemm128_kernel<2" in k or "splitk_reduce_resid" in k or
                "gemm_skinny_kernel<1," in k)
        if main or tail:
            tot += (2 * sum(fe[k].get("FETCH_SIZE", [])) + sum(wr[k].get("WRITE_SIZE", []))) * 1024
        if main:
            n += len(fe[k].get("FETCH_SIZE", []))
    import os
    json.dump({"variant": "PE-Core-L14-336", "batch": 64, "gemm_bytes_per_launch": tot / max(n, 1), "dispatches": n,
               "taken": os.environ.get("PROFILE_TAKEN", "an unrecorded commit"),
               "source": "scripts/pmc_summary.py over gpurun_out/prof_<tag> (scripts/collect_profiles.sh): (2*FETCH_SIZE + WRITE_SIZE) "
                         "summed over the body-GEMM kernels of bench.py (gemm256*_kernel with the GELU / residual / RoPE epilogues and the kernels that take their leftover rows) and divided by the number of linear-layer launches; "
                         "separate --pmc passes; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md HBM section; "
                         "the counters sit at the L2-fabric boundary, Infinity-Cache hits included)",
               "algorithmic_bytes_per_launch_note": "A + W read once + C written once (+ fp32 residual read) = 150..680 MB depending on the layer"},
              open(sys.argv[2], "w"), indent=1)
