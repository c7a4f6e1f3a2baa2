#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes over a prefill GEMM run: per kernel, mean counter values per launch plus derived
ratios (MFMA-busy fraction of SIMD time, VALU per MFMA, wait fractions, effective clock).
    python tools/pmc_gemm_summary.py <pass_dir> [<pass_dir> ...] > profiles/rNN_gemm_pmc.json"""
import csv, glob, json, re, sys, collections
csv.field_size_limit(1 << 30)


def short(k):
    m = re.search(r"(gemm\d?_\w*kernel|rotate_mfma_kernel|rotate_kernel|gemv_kernel)\s*<([^>]*)>", k)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    m = re.search(r"paro\d*(\w+kernel)I?(\w*)", k)
    return k[:70]


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "gemm" not in k and "rotate" not in k:
                    continue
                agg[short(k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in agg.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        m["launches"] = max(len(v) for v in cs.values())
        d = {}
        if "SQ_WAVE_CYCLES" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            # SQ_WAVE_CYCLES counts quad-cycles summed over waves; 2 waves per SIMD in the 8-wave kernels
            d["mfma_busy_per_wave_cycle"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * m["SQ_WAVE_CYCLES"])
            if "SQ_WAIT_ANY" in m:
                d["wait_any_frac_of_wave"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
        if "SQ_WAIT_INST_ANY" in m and "SQ_ACTIVE_INST_VALU" in m:
            d["valu_per_mfma"] = m.get("SQ_INSTS_VALU", 0) / max(m.get("SQ_INSTS_MFMA", 1), 1)
            d["lds_per_mfma"] = m.get("SQ_INSTS_LDS", 0) / max(m.get("SQ_INSTS_MFMA", 1), 1)
        out[k] = {"counters_mean_per_launch": m, "derived": d}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
