#!/usr/bin/env python3
"""Summarise rocprofv3 output of a bench.py run into the small JSON files kept under profiles/.

    python tools/pmc_summary.py pmc   <fetch_dir> <write_dir> <workload> <out.json>
    python tools/pmc_summary.py stats <kernel_stats.csv> <out.csv>

`pmc`: per-launch HBM traffic of the paro::gemv_kernel family from two separate counter passes
(--pmc FETCH_SIZE, --pmc WRITE_SIZE; MI355X_MICROARCH.md, HBM section: both counters are in KiB and on
gfx950 FETCH_SIZE counts wide coalesced reads at half their bytes).
`stats`: the --kernel-trace --stats table with kernel names shortened to their template arguments."""
import csv, glob, json, os, re, sys

csv.field_size_limit(1 << 30)


def short(name: str) -> str:
    m = re.search(r"gemv_kernel<([^>]*)>", name)
    if m:
        return "paro::gemv_kernel<" + m.group(1).replace(" ", "") + ">"
    m = re.search(r"gemv_kernelI(DF16_|DF16b)Li(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d+)E(?:Li(\d+)E)?", name)   # mangled
    if m:
        at = "f16" if m.group(1) == "DF16_" else "bf16"
        fused = f",fused={m.group(7)}" if m.group(7) not in (None, "0") else ""   # 1 / 2: prologue / residual builds, | 8: consumes partial sums
        return f"paro::gemv_kernel<{at},tpw={m.group(2)},rows<={m.group(3)},waves={m.group(4)},prerot={m.group(5)},pd={m.group(6)}{fused}>"
    m = re.search(r"paro::(\w+)", name)
    if m:
        return "paro::" + m.group(1)
    m = re.search(r"_ZN4paro\d+(\w+?_kernel)I(\w*?)EEv", name)      # other mangled templates: attn / gemm3 / lm_head
    return ("paro::" + m.group(1) + "<" + m.group(2) + ">") if m else name[:60]


def counter_rows(d):
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))
    assert f, f"no counter_collection csv under {d}"
    with open(f[0], newline="") as fh:
        for row in csv.DictReader(fh):
            if "gemv_kernel" in row["Kernel_Name"]:
                yield short(row["Kernel_Name"]), float(row["Counter_Value"])


def pmc(fetch_dir, write_dir, workload, out):
    from bench import alg_bytes, kernel_sources_sha, layer_shapes
    res = {}
    for key, d in (("fetch", fetch_dir), ("write", write_dir)):
        per, n, tot = {}, 0, 0.0
        for k, v in counter_rows(d):
            per.setdefault(k, []).append(v)
            n += 1
            tot += v
        res[key] = {"launches": n, "mean_KB": tot / max(n, 1),
                    "per_instantiation_mean_KB": {k: sum(v) / len(v) for k, v in per.items()}}
    shapes = layer_shapes(workload)
    alg = sum(alg_bytes(K, sum(s), len(s)) for (_, K, s, _) in shapes) / len(shapes)
    traffic = (2 * res["fetch"]["mean_KB"] + res["write"]["mean_KB"]) * 1024
    json.dump({"workload": workload,
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 3 --warmup 1 "
                         "--no-graph --no-cpu-baseline`, all paro::gemv_kernel launches",
               "correction": "gfx950: FETCH_SIZE counts wide coalesced reads at half their bytes (MI355X_MICROARCH.md, HBM "
                             "section) -> bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
               # bench.py refuses this file as roofline.traffic once the kernel sources have changed
               "kernel_sources_sha": kernel_sources_sha(),
               **res, "traffic_bytes_per_launch": int(traffic), "algorithmic_bytes_per_launch": int(alg)},
              open(out, "w"), indent=1)
    print(f"traffic {traffic/1e6:.2f} MB per launch vs algorithmic {alg/1e6:.2f} MB ({traffic/alg:.3f}x)")


def stats(src, out):
    with open(src, newline="") as fh, open(out, "w", newline="") as oh:
        r = csv.reader(fh)
        w = csv.writer(oh)
        hdr = next(r)
        w.writerow(hdr)
        for row in r:
            row[0] = short(row[0])
            w.writerow(row)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if sys.argv[1] == "pmc":
        pmc(*sys.argv[2:6])
    else:
        stats(*sys.argv[2:4])
