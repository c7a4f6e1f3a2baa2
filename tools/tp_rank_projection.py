"""What ONE rank of a tensor-parallel llama3-70b decode costs per token, measured on one GPU (the builder's boxes have one):
the rank's shard of every layer on the fused harness (seven launches per layer, HIP graph per token), with the one-shot
all-reduce kernel launched against a world of one (its launch, flag round trip and summation are paid; the peers' stores
and the xGMI flight time are not).  The result is an UPPER bound on the TP tokens/s of an N-GPU node; the difference to
the driver's SCALE numbers is the links.  Also prints the unsharded model (TP = 1) for the like-for-like base.

    python tools/tp_rank_projection.py [--model llama3-70b] [--tp 1,2,4,8] > profiles/rNN_tp_rank_projection.jsonl"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-70b")
    ap.add_argument("--tp", default="1,2,4,8")
    ap.add_argument("--new", type=int, default=64)
    ap.add_argument("--prompt", type=int, default=16)
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29671")
    dist.init_process_group("gloo", rank=0, world_size=1)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from paroquant_amd import tp as ptp
    from paroquant_amd.decoder import MODEL_CONFIGS, ParoDecoderLM
    hidden = MODEL_CONFIGS[a.model][0]
    for tp in [int(t) for t in a.tp.split(",")]:
        ar = ptp.OneShotAllReduce(dev, hidden) if tp > 1 else None
        lm = ParoDecoderLM.random(a.model, dev, max_positions=a.prompt + a.new + 8, tp_rank=0, tp_world=tp, allreduce=ar)
        lm.tp_group = None
        ids = torch.randint(0, lm.cfg.vocab, (a.prompt,), device=dev)
        stats = []
        for i in range(4):
            _, st = lm.generate(ids, a.new)
            if i >= 1:
                stats.append(st)
        ms = float(np.median([s["ms_per_token"] for s in stats]))
        launches = (5 if tp == 1 or lm.fused_allreduce else 7) * lm.cfg.n_layers + 3
        print(json.dumps({"model": a.model, "tp": tp, "rank_ms_per_token": round(ms, 4), "rank_tokens_per_s_upper_bound": round(1e3 / ms, 1),
                          "launches_per_token": launches, "us_per_launch": round(ms * 1e3 / launches, 3),
                          "rank_weight_bytes_per_token": int(lm.bytes_per_token),
                          "rank_GBps": round(lm.bytes_per_token / ms / 1e6, 1),
                          "allreduce": ("in the row-parallel GEMV's epilogue" if lm.fused_allreduce else "separate launch") if tp > 1 else None,
                          "note": "one rank on one GPU; one-shot all-reduce against a world of one (no peer stores, no xGMI)" if tp > 1 else "unsharded"}), flush=True)
        del lm, ar
        torch.cuda.empty_cache()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
