#!/usr/bin/env python3
"""End-to-end batch-1 decode on the fused harness (paroquant_amd/decoder.py), with the reference's benchmark protocol
(cli/benchmark.py:8-26: 2 warm-up + 5 runs, greedy, 128 new tokens; inference/base.py:62-77:
tps = decode tokens / (t_end - t_first_token)).  Synthetic weights of the named architecture (random INT4 in the
checkpoint format, random norms / embeddings, full vocabulary), so the whole model streams from HBM every token.
    python tools/bench_e2e.py [--model qwen3-4b] [--prompt 128] [--new 128] [--runs 5]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from paroquant_amd.decoder import ParoDecoderLM, named_config


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-4b")
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--new", type=int, default=128)
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    import bench
    if args.model in bench.HYBRID:       # Qwen3.5 family: gated delta net + gated head_dim-256 attention (paroquant_amd/decoder_qwen35.py)
        from paroquant_amd.decoder_qwen35 import ParoQwen35DecoderLM
        lm = ParoQwen35DecoderLM.random(args.model, dev, n_layers=args.layers or None, max_positions=args.prompt + args.new + 8)
    else:
        lm = ParoDecoderLM.random(args.model, dev, n_layers=args.layers or None, max_positions=args.prompt + args.new + 8)
    c = lm.cfg
    ids = torch.randint(0, c.vocab, (args.prompt,), device=dev)
    stats = []
    for i in range(args.warmup + args.runs):
        _, st = lm.generate(ids, args.new, use_graph=not args.no_graph)
        if i >= args.warmup:
            stats.append(st)
    tps = float(np.median([s["decode_tokens_per_s"] for s in stats]))
    ms = float(np.median([s["ms_per_token"] for s in stats]))
    lm_head_bytes = c.vocab * c.hidden * 2
    print(json.dumps({
        "metric": "end-to-end batch-1 greedy decode tokens/s (fused harness: %d launches per layer + lm_head, HIP graph)" % (4 if getattr(lm, "fuse_qkv_attn", False) else 5),
        "model": args.model, "layers": c.n_layers, "prompt_tokens": args.prompt, "new_tokens": args.new, "runs": args.runs,
        "decode_tokens_per_s": round(tps, 1), "ms_per_token": round(ms, 4),
        "ttft_ms": round(float(np.median([s["ttft_s"] for s in stats])) * 1e3, 2),
        "packed_weight_bytes_per_token": lm.bytes_per_token, "lm_head_bytes_per_token": lm_head_bytes,
        "effective_GBps": round((lm.bytes_per_token + lm_head_bytes) / ms / 1e6, 1),
        "hip_graph": not args.no_graph, "deferred_ksplit_reduction": bool(lm.deferred), "split_attention": bool(getattr(lm, "split_attn", False)),
        "attention_in_qkv_launch": bool(getattr(lm, "fuse_qkv_attn", False)), "data": "synthetic"}), flush=True)


if __name__ == "__main__":
    main()
