#!/usr/bin/env python3
"""Per-kernel mean counter values per launch from one or more `rocprofv3 --pmc` pass directories (every `paro::` kernel; the GEMV
instantiations keep their template arguments: <type, tiles per wave, rows, waves, pre-rotated, PD, FUSED, QS>).
    python tools/pmc_kernel_counters.py <label> <pass_dir> [<pass_dir> ...] > out.json"""
import collections, csv, glob, json, re, sys
csv.field_size_limit(1 << 30)


def short(k):
    m = re.search(r"gemv_kernelI(\w+?)Li(\d+)ELi(\d+)ELi(\d+)ELb([01])ELi(\d+)ELi(\d+)ELi(\d+)E", k)
    if m:
        t = {"DF16_": "f16", "DF16b": "bf16", "u6__bf16": "bf16"}.get(m.group(1), m.group(1))
        return "gemv_kernel<%s,tpw=%s,rows<=%s,waves=%s,prerot=%s,pd=%s,fused=%s,qs=%s>" % ((t,) + m.groups()[1:])
    m = re.search(r"paro::(\w+)", k) or re.search(r"_ZN4paro\d+(\w+?kernel)", k)
    return m.group(1) if m else k[:60]


def main():
    label, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "paro" in r["Kernel_Name"]:
                    agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"label": label, "kernels": {}}
    for k, cs in sorted(agg.items()):
        m = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
        m["launches"] = max(len(v) for v in cs.values())
        if "SQ_WAIT_ANY" in m and m.get("SQ_WAVE_CYCLES"):
            m["wait_any_frac_of_wave"] = round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 3)
        out["kernels"][k] = m
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
