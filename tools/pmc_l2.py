#!/usr/bin/env python3
"""L2-side traffic of the decode GEMV launches (measure the replicated L2 -> CU traffic instead of inferring it).

    python tools/pmc_l2.py <rocprofv3 counter dir> <workload> <out.json>

Input: one rocprofv3 pass `--pmc TCP_TCC_READ_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum` over `bench.py --steps 3 --warmup 1 --no-graph`
(separate from the FETCH_SIZE / WRITE_SIZE passes: MI355X_MICROARCH.md, PMC slots).  TCP_TCC_READ_REQ counts the read requests the
CUs' vector caches send to the L2 -- everything a CU ingests that its own L1 did not hold: weights (non-temporal, each byte once) AND
the rotation schedule / scale words / x that every workgroup re-reads.  A request moves up to 128 bytes (a 16-byte-per-lane wave load
is 1 KiB = 8 requests of 128 B; narrower loads move 64 B): the byte figures below are given for both widths, the ratio between
instantiations and against the algorithmic bytes is what matters.  TCC_HIT / (TCC_HIT + TCC_MISS) is the L2 hit rate (same guide)."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmc_summary import short
csv.field_size_limit(1 << 30)


def main(d, workload, out):
    from bench import alg_bytes, kernel_sources_sha, layer_shapes
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert f, f"no counter_collection csv under {d}"
    per = {}
    with open(f[0], newline="") as fh:
        for row in csv.DictReader(fh):
            if "gemv_kernel" not in row["Kernel_Name"] and "engine_kernel" not in row["Kernel_Name"]:
                continue
            per.setdefault(short(row["Kernel_Name"]), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    shapes = layer_shapes(workload)
    alg = {n: alg_bytes(K, sum(s), len(s)) for (n, K, s, _) in shapes}
    res = {}
    tot_req, tot_launch = 0.0, 0
    for k, c in sorted(per.items()):
        n = len(next(iter(c.values())))
        m = {name: sum(v) / len(v) for name, v in c.items()}
        req = m.get("TCP_TCC_READ_REQ_sum", 0.0)
        hit, miss = m.get("TCC_HIT_sum", 0.0), m.get("TCC_MISS_sum", 0.0)
        res[k] = {"launches": n, "tcp_tcc_read_req_per_launch": round(req, 1), "MB_at_64B": round(req * 64 / 1e6, 3), "MB_at_128B": round(req * 128 / 1e6, 3),
                  "tcc_read_req_per_launch": round(m.get("TCC_READ_sum", 0.0), 1), "l2_hit_rate": round(hit / max(hit + miss, 1.0), 4)}
        tot_req += req * n
        tot_launch += n
    mean_alg = sum(alg.values()) / len(alg)
    json.dump({"workload": workload, "kernel_sources_sha": kernel_sources_sha(),
               "source": "rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum over `bench.py --steps 3 --warmup 1 --no-graph` (its own pass)",
               "algorithmic_bytes_per_linear": alg, "mean_algorithmic_MB_per_launch": round(mean_alg / 1e6, 3),
               "mean_tcp_tcc_read_req_per_launch": round(tot_req / max(tot_launch, 1), 1),
               "mean_MB_through_the_CUs_per_launch": {"at_64B": round(tot_req / max(tot_launch, 1) * 64 / 1e6, 3), "at_128B": round(tot_req / max(tot_launch, 1) * 128 / 1e6, 3)},
               "per_instantiation": res}, open(out, "w"), indent=1)
    print(json.dumps({"mean_alg_MB": round(mean_alg / 1e6, 2), "mean_req": round(tot_req / max(tot_launch, 1)), "per": {k: (v["MB_at_64B"], v["MB_at_128B"], v["l2_hit_rate"]) for k, v in res.items()}}))


if __name__ == "__main__":
    main(*sys.argv[1:4])
