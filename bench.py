#!/usr/bin/env python3
"""bench.py -- the driver's measurement contract for the MI355X-native ParoQuant hot path.

    python bench.py --gpus N --steps K --warmup W [--workload qwen3-4b] [--no-graph] [--per-shape]

A "step" is ONE batch-1 decode token through the hot path of the named model: every quantised
linear of every decoder layer (merged qkv with 3 rotations, o_proj, merged gate_up with 2 rotations,
down_proj), fp16 activations, chained so each launch consumes the previous launch's output
(attention / norms / activation are outside the hot path -- SURVEY.md section 8 -- and are replaced by
zero-cost views).  Weights are synthetic (random INT4 in the checkpoint's AWQ format, repacked by
the product path), distinct per layer, so a step streams the whole model from HBM (Qwen3-4B:
1.9 GB -- far past the 256 MB Infinity Cache).  The step is captured once in a HIP graph and
replayed (vLLM captures decode the same way).

At N > 1 the DEFAULT workload is the north-star multi-GPU case: the 70B-class model with every linear sharded
Megatron-style across the N ranks (`llama3-70b-tp`: column-parallel qkv / gate_up, row-parallel o / down with an
RCCL all-reduce(SUM) of the [1, hidden] activations after each row-parallel linear) -- `scaling: strong`; the same
workload runs at N = 1 (`--gpus 1 --workload llama3-70b-tp`) for a like-for-like scaling curve.  Any single-GPU
workload named explicitly at N > 1 runs N independent replicas (data-parallel decode, no data-path collective,
`scaling: weak`).  `python bench.py --gpus N` spawns its N ranks itself (one process per GPU, backend nccl = RCCL)
unless it already runs under torchrun (WORLD_SIZE set).

Prints ONE JSON line on rank 0 (see the driver contract) with two extra objects:
  roofline      -- the fused GEMV kernel family `paro::gemv_kernel`: algorithmic bytes per launch
                   (BASELINE.md section 3 formula, averaged over the step's launches) / average launch
                   duration from HIP events around the timed region on the launch stream.
  cpu_baseline  -- the C port of the reference algorithm (oracle/paro_cpu.c) timed on the host
                   cores on a bounded sample (one decoder layer) of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

# who can reach a one-row route (the headline is what the drop-in boundary delivers)
ROUTE_REACH = {
    "fused": "every RotateQuantizedLinear.forward / ParoQuantLinearMethod.apply call (the reference's per-linear operator API)",
    "parts": "a caller that owns the decoder loop (paroquant_amd.decoder.ParoDecoderLM; INTEGRATION.md 5b)",
    "engine": "EXPERIMENTAL build only (make EXPERIMENTAL=1): a caller that hands over a whole chain of linears (paroquant_amd.engine.DecodeEngine, paro_engine_run)",
    "engine2": "EXPERIMENTAL build only: the loader / consumer build of the persistent engine (DecodeEngine(version=2), paro_engine2_run: csrc/experimental/engine2.hip)",
}

MODELS = {
    #              hidden, inter, q_dim, kv_dim, layers
    "qwen3-0.6b": (1024, 3072, 2048, 1024, 28),
    "qwen3-4b": (2560, 9728, 4096, 1024, 36),
    "llama3-8b": (4096, 14336, 4096, 1024, 32),
    "llama3-70b": (8192, 28672, 8192, 1024, 80),
    "qwen3-32b": (5120, 25600, 8192, 1024, 64),      # 27B..32B-class dense stand-in for BASELINE config 5
}


# Qwen3.5: hybrid decoders (transformers models/qwen3_5): every `full_interval`-th layer is full attention with a GATED
# q_proj (2 x heads x head_dim outputs, head_dim 256), the others are gated delta nets whose quantised linears are
# in_proj_qkv (hidden -> 2 key_dim + value_dim), in_proj_z (hidden -> value_dim) and out_proj (value_dim -> hidden);
# in_proj_a / in_proj_b stay dense (reference experiments/optimize/4bit.sh:17-20).  "qwen3.5-9b" is the transformers
# default text config ("Qwen3.5-9B style", configuration_qwen3_5.py); the 4B / 27B dimensions are NOT knowable offline
# (SURVEY 8d): the "-class" entries keep the family's structure at roughly that parameter count -- use --model-config
# <config.json> for a real checkpoint.
HYBRID = {
    #                    hidden, inter, heads, kv_heads, head_dim, lin_k_heads, lin_v_heads, layers, full_interval
    "qwen3.5-9b": (4096, 12288, 16, 4, 256, 16, 32, 32, 4),
    "qwen3.5-4b-class": (2560, 9216, 16, 4, 256, 16, 32, 32, 4),
    "qwen3.5-27b-class": (5120, 17408, 24, 4, 256, 16, 48, 64, 4),
}


def n_layers_of(model: str) -> int:
    return HYBRID[model][7] if model in HYBRID else MODELS[model][4]


def hidden_of(model: str) -> int:
    return HYBRID[model][0] if model in HYBRID else MODELS[model][0]


def known_models():
    return sorted(list(MODELS) + list(HYBRID))


def register_model_from_config(path: str) -> str:
    """Add a model read from an HF `config.json` (hidden_size, intermediate_size, num_attention_heads,
    num_key_value_heads, head_dim, num_hidden_layers) -- e.g. a Qwen3.5-27B checkpoint whose dimensions are
    not knowable offline.  Returns the registered name."""
    with open(path) as f:
        c = json.load(f)
    c = c.get("text_config", c)
    hd = c.get("head_dim") or c["hidden_size"] // c["num_attention_heads"]
    name = os.path.basename(os.path.dirname(os.path.abspath(path))) or "custom"
    if "linear_num_value_heads" in c:      # Qwen3.5-style hybrid
        lt = c.get("layer_types")
        interval = c.get("full_attention_interval", 4)
        if lt and "full_attention" in lt:
            interval = lt.index("full_attention") + 1
        HYBRID[name] = (c["hidden_size"], c["intermediate_size"], c["num_attention_heads"], c.get("num_key_value_heads", c["num_attention_heads"]),
                        hd, c["linear_num_key_heads"], c["linear_num_value_heads"], c["num_hidden_layers"], interval)
        if c.get("linear_key_head_dim", 128) != 128 or c.get("linear_value_head_dim", 128) != 128:
            raise SystemExit("linear-attention head dims other than 128 are not modelled here")
        return name
    MODELS[name] = (c["hidden_size"], c["intermediate_size"], c["num_attention_heads"] * hd,
                    c.get("num_key_value_heads", c["num_attention_heads"]) * hd, c["num_hidden_layers"])
    return name


def alg_bytes(K: int, N: int, P: int) -> int:
    """Algorithmic bytes of one fused GEMV call in the REFERENCE format."""
    G = K // 128
    return K * N // 2 + G * N * 2 + G * N // 2 + 2 * K + 2 * N + P * (16 * K + 8 * K + 2 * K)


def hybrid_layer_shapes(model: str, full: bool, tp: int = 1):
    """Quantised linears of one Qwen3.5 decoder layer (per TP rank), merged where the inputs coincide (q|k|v,
    in_proj_qkv|in_proj_z, gate|up).  Megatron sharding as the reference's loaders do it (vllm/plugin.py:33-50,66-76): the merged
    projections are column-parallel over whole heads (gated q: a head's query AND gate columns stay on one rank; KV heads replicate
    once there are fewer than ranks; the gated delta net's key / value heads split the same way), out_proj / o_proj / down_proj are
    row-parallel with K cut in multiples of 128 (one rotation group) and an all-reduce behind them."""
    h, inter, nh, nkv, hd, lk, lv, _, _ = HYBRID[model]
    kvh = max(nkv // tp, 1)
    mlp = [("gate_up_proj", h, [inter // tp, inter // tp], "col"), ("down_proj", inter // tp, [h], "row")]
    if full:
        return [("qkv_proj(gated q)", h, [2 * (nh // tp) * hd, kvh * hd, kvh * hd], "col"), ("o_proj", (nh // tp) * hd, [h], "row")] + mlp
    return [("in_proj_qkvz", h, [(2 * (lk // tp) + lv // tp) * 128, (lv // tp) * 128], "col"), ("out_proj", (lv // tp) * 128, [h], "row")] + mlp


def shard_error(model: str, tp: int):
    """None when `model` shards `tp`-way (row-parallel K slices in multiples of 128, column slices of whole heads / 16 columns), else why not."""
    if tp == 1:
        return None
    if model in HYBRID:
        h, inter, nh, nkv, hd, lk, lv, _, _ = HYBRID[model]
        if nh % tp or lk % tp or lv % tp or ((nh // tp) * hd) % 128 or inter % (tp * 128) or (nkv >= tp and nkv % tp):
            return f"{model} does not shard {tp}-way: heads {nh}/{nkv}, linear heads {lk}/{lv}, intermediate {inter}"
        return None
    h, inter, q, kv, _ = MODELS[model]
    if q % (tp * 128) or inter % (tp * 128) or kv % (tp * 16):
        return f"{model} does not shard {tp}-way: K slices must be multiples of 128, column slices of 16"
    return None


def layer_plan(model: str, tp: int = 1, n_layers=None):
    """Per decoder layer, the list of its quantised linears [(name, K, partition sizes, kind)]."""
    L = n_layers or n_layers_of(model)
    if model in HYBRID:
        iv = HYBRID[model][8]
        return [hybrid_layer_shapes(model, (l + 1) % iv == 0, tp) for l in range(L)]
    return [layer_shapes(model, tp) for _ in range(L)]


def distinct_shapes(model: str):
    """Every distinct linear of the model once (per-shape tables)."""
    if model in HYBRID:
        seen, out = set(), []
        for sh in hybrid_layer_shapes(model, False) + hybrid_layer_shapes(model, True):
            if sh[0] not in seen:
                seen.add(sh[0])
                out.append(sh)
        return out
    return layer_shapes(model)


def layer_shapes(model: str, tp: int = 1):
    """[(name, K, partition sizes, kind)] of one decoder layer (per TP rank)."""
    h, inter, q, kv, _ = MODELS[model]
    return [
        ("qkv_proj", h, [q // tp, kv // tp, kv // tp], "col"),
        ("o_proj", q // tp, [h], "row"),
        ("gate_up_proj", h, [inter // tp, inter // tp], "col"),
        ("down_proj", inter // tp, [h], "row"),
    ]


def synth_pairs(rng: np.random.Generator, krot: int, K: int) -> np.ndarray:
    """Synthetic ``pairs``: an independent random perfect matching (randperm(128)) per (stage, group) --
    the kernel contract is "any permutation" (reference dummy init, optim/qlinear.py:51-53)."""
    G = K // 128
    out = np.empty((krot, G, 128), dtype=np.int16)
    for r in range(krot):
        for g in range(G):
            out[r, g] = rng.permutation(128).astype(np.int16)
    return out.reshape(krot, K)


def synth_packed(K: int, sizes, dev, gen: torch.Generator, wq_order=None, gain_k: int = 0, keep_ckpt: bool = False):
    """Random layer in checkpoint format (SURVEY section 8d synthetic inputs), unit gain, repacked by the
    product path (torch.ops.paro.repack_awq).  `gain_k`: the in_features the gain is normalised for -- the FULL K for a
    row-parallel shard, whose tp partial outputs are summed by the all-reduce (unit gain after the sum, as in a real shard)."""
    from paroquant_amd.linear import PackedParoWeights
    N = sum(sizes)
    P = len(sizes)
    G = K // 128
    qweight = torch.randint(-2**31, 2**31 - 1, (K, N // 8), dtype=torch.int64, device=dev, generator=gen).to(torch.int32)
    qzeros = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int64, device=dev, generator=gen).to(torch.int32)
    # unit RMS gain through rotate + dequant matmul: E[(w - z)^2] = 42.5 = 6.52^2, E[cs^2] = 1.75, E[(u + 0.5)^2] = 13/12 --
    # exact enough that 80 layers (320 chained linears, llama3-70b) stay inside fp16
    gain = 1.0 / (6.52 * ((gain_k or K) ** 0.5) * (1.75 ** 0.5) * ((13.0 / 12.0) ** 0.5))
    scales = ((torch.rand(G, N, device=dev, generator=gen) + 0.5) * gain).half()
    theta = (torch.randn(P, 8, K // 2, device=dev, generator=gen) * 0.1).half()
    rng = np.random.default_rng(int(torch.randint(0, 2**31 - 1, (1,), generator=gen, device=dev).item()))
    pairs = torch.from_numpy(np.stack([synth_pairs(rng, 8, K) for _ in range(P)])).to(dev)
    cs = (torch.rand(P, 1, K, device=dev, generator=gen) * 1.5 + 0.5).half()
    pk = PackedParoWeights(qweight, qzeros, scales, theta, pairs, cs, sizes, wq_order=wq_order)
    if keep_ckpt:      # the checkpoint-format tensors on the host: what the float64 oracle chain of `route_oracle_check` reads
        pk._ckpt = dict(qweight=qweight.cpu().numpy(), qzeros=qzeros.cpu().numpy(), scales=scales.cpu().numpy(), theta=theta.cpu().numpy(),
                        pairs=pairs.cpu().numpy(), channel_scales=cs.cpu().numpy(), sizes=list(sizes), K=K)
    return pk


class DecodeStack:
    """All quantised linears of `n_layers` decoder layers, chained for one decode token (`rows` sequences)."""

    def __init__(self, model: str, dev, n_layers=None, tp: int = 1, rank: int = 0, seed: int = 0, allreduce=None, rows: int = 1,
                 route: str = "auto", keep_ckpt: bool = False):
        self.model, self.tp, self.rank, self.rows = model, tp, rank, rows
        from paroquant_amd import ops
        self.ops = ops
        # the collective after the row-parallel linears: the one-shot kernel (paroquant_amd.tp.OneShotAllReduce) or dist.all_reduce
        self.allreduce = allreduce or (lambda y: (dist.all_reduce(y), y)[1])
        # the one-shot object can also run inside the row-parallel GEMV's epilogue (paro_fusion_t.ar_*): no launch of its own
        self.fused_ar = allreduce if (hasattr(allreduce, "fusion_args") and os.environ.get("PARO_FUSED_ALLREDUCE", "1") != "0") else None
        self.n_layers = n_layers or n_layers_of(model)
        self.hidden = hidden_of(model)
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed + 1000 * rank)
        self.plan = layer_plan(model, tp, self.n_layers)
        self.shapes = self.plan[0]
        self.layers = []
        for shapes in self.plan:
            self.layers.append([synth_packed(K, sizes, dev, gen, gain_k=K * tp if kind == "row" else 0, keep_ckpt=keep_ckpt)
                                for (_, K, sizes, kind) in shapes])
        self.x = torch.randn(rows, self.hidden, device=dev, dtype=torch.float16, generator=gen)
        self.launches_per_step = sum(len(sh) for sh in self.plan)
        self.bytes_per_step = sum(alg_bytes(K, sum(s), len(s)) for sh in self.plan for (_, K, s, _) in sh)
        # route: "fused" = one paro_w4a16_gemv per linear (rotation inside the consuming kernel); "chain" = the decode-chain
        # family (activations handed over rotated: the consumer's rotation runs in the producer's epilogue, csrc/chain_impl.hpp).
        # Measured (profiles/r03_chain_rows_sweep.jsonl): one row -> fused, 2..16 rows -> chain.
        # "parts" (one row, one GPU) = "fused" with the deferred K-split reduction (include/paro_abi.h v12): a K-split linear whose output is
        # the next linear's whole input (o_proj -> gate_up, down_proj -> the next qkv) leaves its fp32 partial sums and the consumer adds them
        # while it seeds its rotation (in a decoder: to the residual stream, paroquant_amd/decoder.py).  Measured
        # (profiles/r03_parts_micro.jsonl): the in-launch hand-off is 1.3 .. 1.45 us of the producer's launch, completing the sums costs
        # the consumer 0.6 .. 0.8 us (every workgroup reads the four fp32 slots of every channel).
        if route == "auto":
            # The headline route is what the DROP-IN BOUNDARY delivers: one paro_w4a16_gemv per `RotateQuantizedLinear.forward` /
            # `ParoQuantLinearMethod.apply` call (transformers/modules.py:57-71, vllm/plugin.py:281-311) = "fused".  The routes that need a
            # caller who owns the decoder loop -- "parts" (deferred K-split reductions, paroquant_amd/decoder.py uses it), "engine" (one
            # persistent launch for the whole chain, csrc/engine.hip) -- are timed next to it and reported under config.route_ab
            # (VERDICT r3 weak #2).  2..16 rows: the decode-chain family (also a decoder-loop route; --route fused for the per-call number).
            route = "chain" if (rows > 1 and tp == 1) else "fused"
        if route == "parts":
            from paroquant_amd.decoder import deferred_route_pays
            self.parts_pays = deferred_route_pays(self.hidden)      # the decode harness' gate (70B-class widths keep the in-launch reducer)
        if route in ("chain", "parts", "engine", "engine2") and tp != 1:
            raise SystemExit(f"the {route} route is single-GPU")
        if route in ("parts", "engine", "engine2") and rows != 1:
            raise SystemExit(f"the {route} route is batch-1")
        self.route = route
        self._engine = None
        if route in ("parts", "engine", "engine2") or (route == "fused" and rows == 1 and tp == 1):     # (the fused stack can also run the parts route: the A/B leg)
            flat = [pk for lay in self.layers for pk in lay]
            self._flat = flat
            # producer i hands partial sums to consumer i + 1 (which reads the first K columns of i's output) when the launch shape for
            # a launch nobody polls in splits K (paro_gemv_parts_count: o / down 4-way, qkv 2-way, gate_up not at all); a consumer may
            # itself be a producer (qkv -> o in this chain without attention between)
            self._nparts = [0] * len(flat)
            for i in range(len(flat)):
                n = ops.gemv_parts_count(flat[i])
                if n >= 2 and (i + 1 == len(flat) or flat[i + 1].K <= flat[i].N):
                    self._nparts[i] = n                      # (the last one is completed by paro_parts_finish: in a decoder, in front of the final norm)
            self._parts = {}
            for pk in flat:
                self._parts.setdefault(pk.N, [torch.zeros(pk.N, 4, device=dev, dtype=torch.float32) for _ in range(2)])
            self._pk = {n: 0 for n in self._parts}
            self._zeros = torch.zeros(1, max(pk.K for pk in flat), device=dev, dtype=torch.float16)
            self._fin = torch.zeros(1, flat[-1].N, device=dev, dtype=torch.float16)
            self._y = {pk.N: torch.empty(1, pk.N, device=dev, dtype=torch.float16) for pk in flat}
            if route == "parts":
                self.launches_per_step += 1 if self._nparts[-1] else 0
            if route in ("engine", "engine2"):
                self._ensure_engine(2 if route == "engine2" else 1)
        if route == "chain":
            self.launches_per_step += 1        # the head's rotate_parts
            flat = [pk for lay in self.layers for pk in lay]
            self._flat = flat
            # static hand-over buffers, one per consumer geometry
            self._xr = {}
            for pk in flat:
                key = (len(pk.partition_sizes), pk.K)
                if key not in self._xr:
                    self._xr[key] = torch.empty(key[0], rows, key[1], device=dev, dtype=torch.float16)
            self._y = {pk.N: torch.empty(rows, pk.N, device=dev, dtype=torch.float16) for pk in flat}

    def use_route(self, route: str):
        """Switch the one-row route of a built stack (config.route_ab legs)."""
        if route in ("parts", "engine", "engine2") and not hasattr(self, "_flat"):
            raise RuntimeError(f"stack was not built for the {route} route")
        if route in ("engine", "engine2"):
            self._ensure_engine(2 if route == "engine2" else 1)
        self.route = route

    def _ensure_engine(self, version: int = 1):
        if getattr(self, "_engines", None) is None:
            self._engines = {}
        if version not in self._engines:
            from paroquant_amd.engine import DecodeEngine
            # the same chain as `step`: linear i + 1 reads the first K columns of linear i's output
            self._engines[version] = DecodeEngine(self._flat, in_col0=[0] * len(self._flat), version=version)
        self._engine = self._engines[version]

    def _step_engine(self, x: torch.Tensor) -> torch.Tensor:
        return self._engines[2 if self.route == "engine2" else 1](x)

    def step(self, x: torch.Tensor) -> torch.Tensor:
        if self.route in ("engine", "engine2"):
            return self._step_engine(x)
        if self.route == "chain":
            return self._step_chain(x)
        if self.route == "parts":
            return self._step_parts(x)
        h = x
        tp = self.tp
        if tp == 1:
            # every linear consumes the first K columns of its predecessor's output (attention / SiLU*mul stand-ins: views).  With more than one
            # row that slice is a STRIDED view and torch copies it in front of every linear whose predecessor is wider or narrower than its K
            # (72 copy launches per Qwen3-4B step, 0.21 ms -- no decoder has them: attention and SiLU*mul write contiguous tensors).  The
            # stand-in there is the first rows x K ELEMENTS of the predecessor's output: the same data dependency, a contiguous view, no
            # launch.  PARO_BENCH_STRIDED_STANDIN=1 restores the column slice (what rounds 3..5 timed at --rows > 1).
            strided = self.rows > 1 and os.environ.get("PARO_BENCH_STRIDED_STANDIN", "0") == "1"
            for lay in self.layers:
                for pk in lay:
                    if self.rows == 1 or strided:
                        h = pk.apply(h[:, : pk.K])
                    else:
                        h = pk.apply(h.reshape(-1)[: self.rows * pk.K].view(self.rows, pk.K))
            return h
        for qkv, o, gu, down in self.layers:
            a = qkv.apply(h)[:, : o.K]                      # attention stand-in: a view, no kernel
            if self.fused_ar is not None:                  # RowParallelLinear: the all-reduce runs in the GEMV's epilogue
                h = self.ops.w4a16_gemv_fused(a, o, 0, allreduce=self.fused_ar)
            else:
                h = self.allreduce(o.apply(a))              # ... or as its own launch (one-shot xGMI kernel, or RCCL)
            d = gu.apply(h)[:, : down.K]                    # SiLU*mul stand-in: a view, no kernel
            if self.fused_ar is not None:
                h = self.ops.w4a16_gemv_fused(d, down, 0, allreduce=self.fused_ar)
            else:
                h = self.allreduce(down.apply(d))
        return h

    def _step_parts(self, x: torch.Tensor) -> torch.Tensor:
        """The same chain of linears as the fused route (each consumes the first K columns of its predecessor's output), with the K-split
        reductions deferred into the consumers: base = zeros, so x' = the sum of the partial sums, rounded once -- what the in-launch
        reducer writes (bit for bit when the split is the same, tests/test_gpu_parts.py)."""
        ops, flat = self.ops, self._flat
        cur, pend = x, None
        for i, pk in enumerate(flat):
            kw = {}
            if pend is not None:               # consumer of the predecessor's partial sums
                xin, kw["parts_in"] = self._zeros[:, : pk.K], pend[: pk.K]
            else:
                xin = cur[:, : pk.K]
            if self._nparts[i]:                # ... and / or producer for the successor
                k = self._pk[pk.N]
                self._pk[pk.N] = k ^ 1
                pend = self._parts[pk.N][k]
                ops.w4a16_gemv_fused(xin, pk, 0, parts_out=pend, parts_n=self._nparts[i], **kw)
            elif pend is not None:
                cur, pend = ops.w4a16_gemv_fused(xin, pk, 0, out=self._y[pk.N], **kw), None
            else:
                cur = pk.apply(xin)
        if pend is not None:
            cur = ops.parts_finish(pend, out=self._fin.view(-1)).view(1, -1)
        return cur

    def _step_chain(self, x: torch.Tensor) -> torch.Tensor:
        ops, flat = self.ops, self._flat
        xr = ops.rotate_parts(x, flat[0], out=self._xr[(len(flat[0].partition_sizes), flat[0].K)])
        y = None
        for i, pk in enumerate(flat):
            nxt = flat[i + 1] if i + 1 < len(flat) else None
            nx = self._xr[(len(nxt.partition_sizes), nxt.K)] if nxt is not None else None
            y, xr = ops.chain_gemv(xr, pk, out=self._y[pk.N], next_pk=nxt, next_x=nx, next_col0=0)
        return y


def time_steps(fn, steps: int, warmup: int, world: int, dev):
    if world > 1:
        dist.barrier()          # ranks enter the warm-up together (a kernel-level collective spins on late peers)
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1)
    if world > 1:
        t = torch.tensor([wall, ev_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, ev_ms = t[0].item(), t[1].item()
    return wall, ev_ms


def route_oracle_check(model: str, dev, routes, n_layers: int = 2):
    """every one-row route against the float64 ORACLE chain (oracle/paro_oracle.py used as the checker, never as the thing
    measured), not against another route.  A small twin of the bench stack (`n_layers` decoder layers, own seed, checkpoint tensors kept on the
    host) runs through each route on the GPU; the oracle runs the same chain -- linear i + 1 reads the first K columns of linear i's output,
    rounded to fp16 as the kernels store it -- with the rotation, the dequantisation and the dot products in float64 (`ideal=True`).
    Returns {route: {"max_rel_err_vs_oracle", "within_1e-2_of_oracle"}} + the description of the check."""
    from oracle import paro_oracle as po       # checker use only (as tests/ and smoke() use it)
    small = DecodeStack(model, dev, n_layers=n_layers, seed=4242, route="fused", keep_ckpt=True)
    flat = small._flat
    h = small.x.cpu().numpy().astype(np.float16)
    for i, pk in enumerate(flat):
        c = pk._ckpt
        need = flat[i + 1].K if i + 1 < len(flat) else sum(c["sizes"])       # columns the next linear reads (whole partitions)
        cols, outs = 0, []
        for p, n in enumerate(c["sizes"]):
            if cols >= need:
                break
            outs.append(po.paro_linear(h[:, : c["K"]], np.ascontiguousarray(c["qweight"][:, cols // 8:(cols + n) // 8]),
                                       np.ascontiguousarray(c["qzeros"][:, cols // 8:(cols + n) // 8]), np.ascontiguousarray(c["scales"][:, cols:cols + n]),
                                       c["theta"][p], c["pairs"][p], c["channel_scales"][p], None, 128, ideal=True))
            cols += n
        y64 = np.concatenate(outs, axis=-1)
        h = y64.astype(np.float16)             # the kernels' one rounding per linear
    ref, ncol = y64, y64.shape[-1]
    out = {}
    for route in routes:
        try:
            small.use_route(route)
            y = small.step(small.x)
            torch.cuda.synchronize(dev)
            err = float(np.max(np.abs(y.float().cpu().numpy().astype(np.float64)[:, :ncol] - ref)) / max(np.max(np.abs(ref)), 1e-30))
            out[route] = {"max_rel_err_vs_oracle": round(err, 6), "within_1e-2_of_oracle": bool(err < 1e-2)}
        except Exception as e:
            out[route] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            small.route = "fused"
    del small
    torch.cuda.empty_cache()
    return out, (f"{n_layers} decoder layers ({len(flat)} chained linears) of the same architecture through each route vs the float64 oracle chain "
                 "(oracle/paro_oracle.py paro_linear ideal=True, fp16 rounding between linears), max|y - ref| / max|ref| of the last linear's output")


MFMA_PEAK_TFLOPS = 2500.0   # dense fp16 / bf16 matrix peak of MI355X (/opt/skills/guides/MI355X_MICROARCH.md; not the 2:1-sparsity figure)

# the north star's second metric: "MFMA utilisation (prefill, compute-bound) against gfx950 peaks" at BASELINE config 3's batch-32 x seq-2048
# prefill (M = 65536 rows): the reference reaches this through ParoQuantLinearMethod.apply at large M (vllm/plugin.py:281-311)
PREFILL_SHAPES = [("llama3-8b gate_up_proj [P=2]", 4096, [14336, 14336]), ("llama3-8b qkv_proj [P=3]", 4096, [4096, 1024, 1024]),
                  ("qwen3.5-4b-class gate_up_proj [P=2]", 2560, [9216, 9216])]


def prefill_table(dev, rows: int = 65536, launches: int = 5):
    """One bounded prefill leg: the fused linear at M = 32 x 2048 rows through the per-call operator (`PackedParoWeights.apply`
    = rotation pre-pass on the matrix cores + W4A16 MFMA GEMM), `launches` timed calls after two warm-ups; the pre-pass alone
    (`paro_rotate_parts` with the dense rotation matrices = the same launch the GEMM entry issues first) timed the same way."""
    from paroquant_amd import ops
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    out = []
    for name, K, sizes in PREFILL_SHAPES:
        pk = synth_packed(K, sizes, dev, gen)
        pk.prepare_prefill(torch.float16)
        x = torch.randn(rows, K, device=dev, dtype=torch.float32, generator=gen).half()
        xr = torch.empty(len(sizes), rows, K, device=dev, dtype=torch.float16)

        def timed(fn):
            for _ in range(2):
                fn()
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(launches):
                fn()
            e1.record()
            torch.cuda.synchronize(dev)
            return e0.elapsed_time(e1) / launches
        ms = timed(lambda: pk.apply(x))
        ms_pre = timed(lambda: ops.rotate_parts(x, pk, out=xr))
        try:      # the north star's one-launch form (rotation inside the GEMM's LDS stage), experiment variant 44: same bits, measured beside it
            ms_fused = round(timed(lambda: ops.w4a16_gemm_forced(x, pk, None, True, 44)), 4)
        except Exception as e:
            ms_fused = f"{type(e).__name__}: {e}"[:80]
        flops = 2.0 * rows * K * sum(sizes)
        out.append({"linear": name, "M": rows, "K": K, "N": sum(sizes), "P": len(sizes), "ms": round(ms, 4), "TFLOPs": round(flops / ms / 1e9, 1),
                    "mfma_util": round(flops / ms / 1e9 / MFMA_PEAK_TFLOPS, 4), "prepass_ms": round(ms_pre, 4),
                    "prepass_share": round(ms_pre / ms, 4), "timed_calls": launches, "kernel_launches_per_call": 2,
                    "fused_rotation_experiment_ms": ms_fused})
        del pk, x, xr
        torch.cuda.empty_cache()
    return out


def mid_prefill_table(dev, model: str = "llama3-8b", row_counts=(128, 512, 2048), calls: int = 10):
    """Medium prefill through the per-call operator (round 6, session 3): the prompt lengths a server actually sees.  Per linear of `model` and
    row count: us per `PackedParoWeights.apply` call (rotation pre-pass + W4A16 MFMA GEMM on the block shape gemm.hip `g4_shape` picks),
    events around `calls` back-to-back calls after two warm-ups, weights far past the caches between shapes.  The sweep behind the shapes:
    tools/sweep_gemm4.py, profiles/r06_sweep_gemm4_*.jsonl."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    out = []
    for name, K, sizes, _ in layer_shapes(model):
        pk = synth_packed(K, sizes, dev, gen)
        pk.prepare_prefill(torch.float16)
        for rows in row_counts:
            x = torch.randn(rows, K, device=dev, dtype=torch.float32, generator=gen).half()
            for _ in range(2):
                pk.apply(x)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(calls):
                pk.apply(x)
            e1.record()
            torch.cuda.synchronize(dev)
            us = e0.elapsed_time(e1) * 1e3 / calls
            out.append({"linear": f"{model} {name}", "M": rows, "K": K, "N": sum(sizes), "us_per_call": round(us, 2),
                        "TFLOPs": round(2.0 * rows * K * sum(sizes) / us / 1e6, 1)})
        del pk
        torch.cuda.empty_cache()
    return {"rows": out, "note": "same weights every call (cache-resident for the narrow linears): an upper bound of the call rate, for the shape rule's A/B see "
                                 "profiles/r06_sweep_gemm4_*.jsonl (rotating weight copies).  Round 5's rule on the same protocol: profiles/r06_sweep_gemm4_llama3-8b_before.jsonl"}


def batched_decode_steps(dev, model: str = "qwen3-4b", n_layers: int = 12, row_counts=(1, 2, 4, 8, 16, 32, 64), steps: int = 20, warmup: int = 3):
    """Batched decode THROUGH THE BOUNDARY (what a vLLM decode batch reaches: `ParoQuantLinearMethod.apply` is M-agnostic, vllm/plugin.py:281-311),
    driver-visible: the bench step of the headline workload's first `n_layers` layers at several row counts on the
    per-call route -- same weight bytes at every row count, so `x_one_row` is what the extra rows cost.  mode = what the library picked for the
    layer's qkv projection at that row count (0 rotation replicated per workgroup, 1 pre-pass, 3 shared inside the launch).  32 / 64 rows
    (added in round 6, session 3): the GEMV with 2 / 4 MFMA row tiles behind the schedule pre-pass (rotate.hip), wide merged projections
    on the MFMA GEMM from 33 rows on (abi.hip paro_w4a16_linear)."""
    import ctypes
    from paroquant_amd import _native as nat, ops as _ops
    lib = nat.load()
    out, base = [], None
    for rows in row_counts:
        st = DecodeStack(model, dev, n_layers=n_layers, seed=37, rows=rows, route="fused")
        st.step(st.x)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        ss = torch.cuda.Stream(dev)
        ss.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(ss):
            st.step(st.x)
        torch.cuda.current_stream(dev).wait_stream(ss)
        with torch.cuda.graph(g):
            st.step(st.x)
        _, ev = time_steps(g.replay, steps, warmup, 1, dev)
        ms = ev / steps
        base = ms if base is None else base
        pk0 = st._flat[0] if hasattr(st, "_flat") else (st.layers[0][0] if getattr(st, "layers", None) else None)
        mode = None
        try:
            if pk0 is not None:
                d = _ops.pk_desc(pk0, torch.float16)
                kn = [ctypes.c_int(v) for v in (0, 0, 0, -1)]
                if rows > 16:
                    mode = 1
                elif lib.paro_gemv_launch_shape(ctypes.byref(d), rows, *[ctypes.byref(k) for k in kn]) == 0:
                    mode = kn[3].value
        except Exception:
            mode = None
        out.append({"rows": rows, "ms_per_step_measured": round(ms, 4), "x_one_row": round(ms / base, 3),
                    "tokens_per_s_full_depth": round(rows * 1e3 / (ms * n_layers_of(model) / st.n_layers), 1), "qkv_rotation_mode": mode})
        del st, g
        torch.cuda.empty_cache()
    return {"workload": f"{model}-PARO decode, per-call route", "layers_measured": n_layers, "layers_of_model": n_layers_of(model), "rows": out,
            "note": "same weight bytes at every row count; x_one_row = step time / the one-row step of the same stack.  Protocol (round 6): between two linears the "
                    "stand-in for attention / SiLU*mul is a CONTIGUOUS view of the predecessor's output; rounds 3..5 sliced columns, which at > 1 row is a strided view "
                    "that torch copies in front of 72 of a Qwen3-4B step's 144 linears (0.21 ms of copy launches no decoder has; PARO_BENCH_STRIDED_STANDIN=1 restores it). "
                    "Round 5 with the strided stand-in, full depth: 1.42 / 1.90 / 2.60 at 2 / 8 / 16 rows; this build the same way 1.30 / 1.60 / 1.87, "
                    "contiguous 1.06 / 1.36 / 1.63 (mode 3 off: 1.06 / 1.52 / 2.31): profiles/r06_rows_boundary_v2.jsonl.  32 / 64 rows, full depth: 2.95 / 3.75 ms "
                    "with the stage-kernel pre-pass (rounds 1..5) -> 1.84 / 2.71 ms behind the schedule pre-pass: profiles/r06_rows_17_64_v2.jsonl"}


def config_steps(dev, models=(("qwen3-0.6b", 0), ("qwen3.5-4b-class", 8)), steps: int = 20, warmup: int = 3):
    """BASELINE configs 1 and 2's decode legs in the same line: the bench step (per-call route, one HIP graph) of the
    other single-GPU configurations; `layers` = 0 builds the full depth, otherwise the first n layers (stated in the row; the per-layer time
    does not depend on the depth: distinct weights per layer, far past the caches either way for the 4B-class)."""
    out = []
    for model, nl in models:
        st = DecodeStack(model, dev, n_layers=nl or None, seed=31, route="fused")
        st.step(st.x)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        ss = torch.cuda.Stream(dev)
        ss.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(ss):
            st.step(st.x)
        torch.cuda.current_stream(dev).wait_stream(ss)
        with torch.cuda.graph(g):
            st.step(st.x)
        _, ev = time_steps(g.replay, steps, warmup, 1, dev)
        ms = ev / steps
        full = n_layers_of(model)
        out.append({"workload": f"{model}-PARO batch-1 decode", "layers_measured": st.n_layers, "layers_of_model": full,
                    "ms_per_step_measured": round(ms, 4), "tokens_per_s_full_depth": round(1e3 / (ms * full / st.n_layers), 1),
                    "bytes_per_step_measured": int(st.bytes_per_step), "launches": st.launches_per_step,
                    "roofline_frac": round(st.bytes_per_step / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)})
        del st, g
        torch.cuda.empty_cache()
    return out


# BASELINE.json's target: ">= 70 % of MI355X HBM roofline on batch-1 INT4 GEMV at Llama-3-8B q/k/v/o/mlp shapes" -- the five rows of
# BASELINE.md section 3 that a plug-in issues one call for (q / o and k / v unmerged as the HF path runs them, merged qkv / gate_up as the
# vLLM path runs them, down).  Measured by every default bench run so that the driver-run record carries them (VERDICT r3 missing #4).
NORTH_STAR_SHAPES = [("q_proj / o_proj", 4096, [4096], "row"), ("k_proj / v_proj", 4096, [1024], "col"),
                     ("qkv_proj merged [P=3]", 4096, [4096, 1024, 1024], "col"), ("gate_up_proj merged [P=2]", 4096, [14336, 14336], "col"),
                     ("down_proj", 14336, [4096], "row")]


LAUNCH_FLOOR_US, LAUNCH_FLOOR_GBPS = 2.4, 7500.0     # a dependent streaming launch on MI355X: profiles/r03_overlap_probe2.jsonl, DESIGN 3.1


def per_shape_table(model: str, dev, reps: int = 400, shapes=None, min_bytes: int = 1 << 30):
    """Per-linear GEMV timing through the per-call operator (`PackedParoWeights.apply` = what RotateQuantizedLinear.forward /
    ParoQuantLinearMethod.apply run): events around `reps` back-to-back launches cycling >= `min_bytes` of distinct weights so
    neither L2 nor the 256 MB Infinity Cache can serve them."""
    rows = []
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    for name, K, sizes, _ in (shapes or distinct_shapes(model)):
        nb = alg_bytes(K, sum(sizes), len(sizes))
        copies = max(2, min(64, int(min_bytes // nb) + 1))
        packs = [synth_packed(K, sizes, dev, gen) for _ in range(copies)]
        x = torch.randn(1, K, device=dev, dtype=torch.float16, generator=gen)
        for i in range(20):
            packs[i % copies].apply(x)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(reps):
                packs[i % copies].apply(x)
        g.replay()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize(dev)
        us = e0.elapsed_time(e1) * 1e3 / reps
        floor_us = LAUNCH_FLOOR_US + nb / LAUNCH_FLOOR_GBPS / 1e3
        rows.append({"linear": name, "K": K, "N": sum(sizes), "P": len(sizes), "bytes": nb, "us_per_launch": round(us, 3),
                     "GBps": round(nb / us / 1e3, 1), "frac_of_8TBps": round(nb / us / 1e3 / HBM_PEAK_GBPS, 4),
                     # the distance to what ONE dependent launch that only streams these bytes costs on this part
                     # (2.4 us + bytes / 7.5 TB/s, measured: profiles/r03_overlap_probe2.jsonl); floor_frac = floor / measured <= 1
                     "floor_us": round(floor_us, 3), "floor_frac": round(floor_us / us, 4)})
        del packs, g
        torch.cuda.empty_cache()
    return rows


def cpu_baseline(model: str, budget_s: float = 20.0):
    """Time the C port of the reference algorithm on ONE decoder layer (1/n_layers of a step)."""
    from oracle import paro_cpu as pc   # the ONLY use of oracle/ in this file: the CPU baseline leg
    L = n_layers_of(model)
    rng = np.random.default_rng(0)
    # the sample: the distinct linears of the model once; a hybrid's two layer kinds are weighted by their counts below
    kinds = [(hybrid_layer_shapes(model, False), L - L // HYBRID[model][8]), (hybrid_layer_shapes(model, True), L // HYBRID[model][8])] \
        if model in HYBRID else [(layer_shapes(model), L)]
    layers = []
    for name, K, sizes, _ in [sh for shapes, _ in kinds for sh in shapes]:
        N, P, G = sum(sizes), len(sizes), K // 128
        layers.append(dict(
            qweight=rng.integers(-2**31, 2**31 - 1, size=(K, N // 8), dtype=np.int64).astype(np.int32),
            qzeros=rng.integers(-2**31, 2**31 - 1, size=(G, N // 8), dtype=np.int64).astype(np.int32),
            scales=rng.uniform(0.002, 0.02, size=(G, N)).astype(np.float16),
            theta=(rng.standard_normal((P, 8, K // 2)) * 0.1).astype(np.float16),
            pairs=np.stack([synth_pairs(rng, 8, K) for _ in range(P)]),
            channel_scales=rng.uniform(0.5, 2.0, size=(P, 1, K)).astype(np.float16), sizes=sizes, K=K))
    xs = [rng.standard_normal((1, l["K"])).astype(np.float16) for l in layers]
    pc.load()
    # one socket's worth of physical cores at most: on the 2 x 64-core / 256-thread MI355X host the default
    # 128-thread team is 1.4-8x slower and erratic (oversubscribed SMT siblings, cross-socket traffic)
    pc.set_threads(max(1, min(64, (os.cpu_count() or 2) // 2)))
    per_kind = [len(shapes) for shapes, _ in kinds]
    times = []          # per run: seconds of each layer kind
    t_start = time.perf_counter()
    while True:
        run, i = [], 0
        for n_lin in per_kind:
            t0 = time.perf_counter()
            for l, x in zip(layers[i:i + n_lin], xs[i:i + n_lin]):
                pc.linear_f16(x, l)
            run.append(time.perf_counter() - t0)
            i += n_lin
        times.append(run)
        if len(times) >= 3 and (time.perf_counter() - t_start > budget_s or len(times) >= 20):
            break
    med = np.median(np.asarray(times), axis=0)
    t_token = float(sum(m * cnt for m, (_, cnt) in zip(med, kinds)))
    t_layer = t_token / L
    out = {"value": round(1.0 / t_token, 4), "unit": "tokens/s", "cores": pc.threads(), "kind": "port",
           "sample": f"{len(kinds)} of {L} decoder layers ({sum(per_kind)} fused linears, M=1) x {len(times)} runs, median {t_layer * 1e3:.1f} ms/layer; "
                     f"tokens/s = 1 / (sum over layer kinds of ms_per_layer x count)"}
    # BASELINE config 0 ("single ParoLinear layer 4096 x 4096, group 128, CPU dequant + matmul path"): the same one
    # layer as torch-CPU dequant -> fp32 matmul (what AutoAWQ's WQLinearMMFunction does off-GPU), M = 1, for context
    try:
        K = N = 4096
        qw = torch.from_numpy(rng.integers(0, 16, size=(K, N), dtype=np.int64).astype(np.float32))
        qz = torch.from_numpy(rng.integers(0, 16, size=(K // 128, N), dtype=np.int64).astype(np.float32))
        sc = torch.from_numpy(rng.uniform(0.002, 0.02, size=(K // 128, N)).astype(np.float32))
        xr = torch.from_numpy(rng.standard_normal((1, K)).astype(np.float32))
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            w = (qw - qz.repeat_interleave(128, 0)) * sc.repeat_interleave(128, 0)
            (xr @ w).sum().item()
            ts.append(time.perf_counter() - t0)
        out["torch_cpu_dequant_matmul_4096x4096_ms"] = round(float(np.median(ts)) * 1e3, 2)
    except Exception as e:   # context only: never fail the bench line over it
        out["torch_cpu_dequant_matmul_4096x4096_ms"] = f"failed: {e}"
    return out


def cpu_per_shape(budget_s: float = 12.0):
    """SURVEY 8d / BASELINE.md section 4: the CPU baselines per shape, on the host cores of this box, ms per call and effective GB/s with
    the same bytes(K, N, P) as the GPU rows.  B1 = the oracle path on torch-CPU (unpack -> (q - z) s -> fp16 W, rotate(x * cs), fp32 matmul;
    what AutoAWQ's no-extension fallback does), B2 = the C / OpenMP port of the reference algorithm (oracle/paro_cpu.c; the only use of
    oracle/ here is this baseline leg).  Config 1 (4096 x 4096, M in {1, 16, 2048}) plus every Llama-3-8B shape of BASELINE.md section 3 at
    M = 1.  Median of up to 20 runs after one warm-up, bounded by `budget_s` in total (each row gets an equal share; at least 2 runs)."""
    from oracle import paro_cpu as pc
    pc.load()
    pc.set_threads(max(1, min(64, (os.cpu_count() or 2) // 2)))
    torch.set_num_threads(max(1, min(64, (os.cpu_count() or 2) // 2)))
    rng = np.random.default_rng(1)
    cases = [("config 1: 4096 x 4096", 4096, [4096], M) for M in (1, 16, 2048)] + [(n, K, sz, 1) for n, K, sz, _ in NORTH_STAR_SHAPES]
    share = budget_s / (2 * len(cases))
    out = []
    for name, K, sizes, M in cases:
        N, P, G = sum(sizes), len(sizes), K // 128
        L = dict(qweight=rng.integers(-2**31, 2**31 - 1, size=(K, N // 8), dtype=np.int64).astype(np.int32),
                 qzeros=rng.integers(-2**31, 2**31 - 1, size=(G, N // 8), dtype=np.int64).astype(np.int32),
                 scales=rng.uniform(0.002, 0.02, size=(G, N)).astype(np.float16),
                 theta=(rng.standard_normal((P, 8, K // 2)) * 0.1).astype(np.float16),
                 pairs=np.stack([synth_pairs(rng, 8, K) for _ in range(P)]),
                 channel_scales=rng.uniform(0.5, 2.0, size=(P, 1, K)).astype(np.float16), sizes=sizes, K=K)
        x = rng.standard_normal((M, K)).astype(np.float16)

        def timed(fn):
            fn()
            ts, t_begin = [], time.perf_counter()
            while len(ts) < 2 or (len(ts) < 20 and time.perf_counter() - t_begin < share):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            return float(np.median(ts)), len(ts)

        # B1: torch-CPU -- the unpacked nibbles are prepared once (format conversion), dequant + rotation + matmul are timed
        shifts = torch.tensor([0, 16, 4, 20, 8, 24, 12, 28], dtype=torch.int32)     # AWQ order (0,2,4,6,1,3,5,7): column 8c + j sits in nibble ...
        qw = ((torch.from_numpy(L["qweight"]).unsqueeze(-1) >> shifts) & 15).reshape(K, N).to(torch.float32)
        qz = ((torch.from_numpy(L["qzeros"]).unsqueeze(-1) >> shifts) & 15).reshape(G, N).to(torch.float32)
        sc = torch.from_numpy(L["scales"]).float()
        xt = torch.from_numpy(x).float()
        cs = torch.from_numpy(L["channel_scales"]).float().reshape(P, K)
        th = torch.from_numpy(L["theta"]).float()
        pr = torch.from_numpy(L["pairs"].astype(np.int64))                           # [P][8][K] group-local
        base = (torch.arange(K) // 128 * 128)
        I = pr[:, :, 0::2] + base[0::2]
        J = pr[:, :, 1::2] + base[0::2]
        col0 = np.concatenate([[0], np.cumsum(sizes)])

        def b1():
            w = ((qw - qz.repeat_interleave(128, 0)) * sc.repeat_interleave(128, 0)).half().float()
            ys = []
            for p_ in range(P):
                xr = (xt * cs[p_]).clone()
                for r in range(8):
                    c, s_ = torch.cos(th[p_, r]), torch.sin(th[p_, r])
                    xi, xj = xr[:, I[p_, r]], xr[:, J[p_, r]]
                    xr[:, I[p_, r]] = c * xi + s_ * xj
                    xr[:, J[p_, r]] = c * xj - s_ * xi
                ys.append(xr @ w[:, col0[p_]:col0[p_ + 1]])
            return torch.cat(ys, 1).half()

        nb = alg_bytes(K, N, P) + (M - 1) * 2 * (K + N)
        t2, n2 = timed(lambda: pc.linear_f16(x, L))
        t1, n1 = timed(b1)
        out.append({"shape": name, "M": M, "K": K, "N": N, "P": P, "bytes": int(nb),
                    "B2_c_port_ms": round(t2 * 1e3, 3), "B2_GBps": round(nb / t2 / 1e9, 2), "B2_runs": n2,
                    "B1_torch_cpu_ms": round(t1 * 1e3, 3), "B1_GBps": round(nb / t1 / 1e9, 2), "B1_runs": n1})
    short = [f"{r['shape']} M={r['M']}: B1 {r['B1_runs']} runs, B2 {r['B2_runs']} runs" for r in out if min(r["B1_runs"], r["B2_runs"]) < 20]
    return {"threads": pc.threads(), "rows": out,
            "protocol": f"BASELINE.md section 4: median after one warm-up of up to 20 runs, time-boxed at {share:.2f} s per row and leg so that the default bench run stays within minutes",
            # BASELINE.md asks for >= 20 runs; the rows listed here stopped at their time box with fewer (the large
            # torch-CPU legs take 0.1 .. 1 s per run): their medians are of the stated run counts.  `--cpu-budget` raises the box.
            "protocol_deviation": short or None}


def end_to_end(model: str, dev, prompt: int = 128, new: int = 128, runs: int = 5, warmup: int = 2,
               tp_rank: int = 0, tp_world: int = 1, allreduce=None):
    """End-to-end batch-1 greedy decode of the whole model on the fused harness (paroquant_amd/decoder.py: five launches
    per layer + final norm / lm_head / argmax, one HIP graph per token), with the reference's benchmark protocol
    (cli/benchmark.py:8-26: 2 warm-up + 5 runs, 128 new tokens; inference/base.py:62-77: tps = decode tokens /
    (t_end - t_first_token)).  Synthetic weights of the architecture, full vocabulary."""
    from paroquant_amd.decoder import MODEL_CONFIGS, ParoDecoderLM
    hybrid = model in HYBRID
    if model not in MODEL_CONFIGS and not hybrid:
        return None
    if hybrid:
        if tp_world != 1:
            return None
        from paroquant_amd.decoder_qwen35 import ParoQwen35DecoderLM       # gated delta net + gated head_dim-256 attention (csrc/gdn.hip)
        lm = ParoQwen35DecoderLM.random(model, dev, max_positions=prompt + new + 8)
        lm.deferred, lm.fused_allreduce = False, False
    else:
        lm = ParoDecoderLM.random(model, dev, max_positions=prompt + new + 8, tp_rank=tp_rank, tp_world=tp_world, allreduce=allreduce)
    ids = torch.randint(0, lm.cfg.vocab, (prompt,), device=dev, generator=torch.Generator(device=dev).manual_seed(11))  # same prompt on every rank
    stats = []
    for i in range(warmup + runs):
        _, st = lm.generate(ids, new)
        if i >= warmup:
            stats.append(st)
    tps = float(np.median([s_["decode_tokens_per_s"] for s_ in stats]))
    ms = float(np.median([s_["ms_per_token"] for s_ in stats]))
    lm_head_bytes = lm.cfg.vocab * lm.cfg.hidden * 2
    return {"value": round(tps, 1), "unit": "tokens/s", "ms_per_token": round(ms, 4),
            "ttft_ms": round(float(np.median([s_["ttft_s"] for s_ in stats])) * 1e3, 2),
            "protocol": f"{warmup} warm-up + {runs} runs, prompt {prompt}, {new} new tokens, greedy, HIP graph per token",
            "launches_per_token": (sum(5 if L.full else 6 for L in lm.layers) + 3) if hybrid else
                                  (((5 if tp_world == 1 or lm.fused_allreduce else 7) - (1 if getattr(lm, "fuse_qkv_attn", False) else 0)) * lm.cfg.n_layers
                                   + 3 + (1 if lm.deferred else 0)),
            "deferred_ksplit_reduction": bool(lm.deferred),   # o / down leave partial sums, gate_up / the next qkv complete them (decoder.py)
            "attention_in_qkv_launch": bool(getattr(lm, "fuse_qkv_attn", False)),   # ABI v18 attn_tail: 4 launches per layer
            "parallelism": f"tp{tp_world}" + (" (bytes_per_token and GBps are per rank)" if tp_world > 1 else ""),
            "bytes_per_token": int(lm.bytes_per_token + lm_head_bytes),
            "GBps": round((lm.bytes_per_token + lm_head_bytes) / ms / 1e6, 1),
            "note": "whole model: fused GEMVs (RMSNorm / SiLU*mul / residual fused), decode attention with KV cache, fp16 lm_head"}


def newest_pmc_file(model: str, explicit: str = ""):
    """(path, reason): profiles/rNN_pmc_bench_<model>.json of the latest round (or the file named on the command line).
    A summary that is OLDER than the kernel sources it would describe is refused (path None + the reason): a stale constant
    is worse than `traffic: null`.  The check uses the `kernel_sources_sha` the summary records
    (tools/pmc_summary.py) -- file times do not survive a checkout."""
    import glob
    if explicit:
        return (explicit, None) if os.path.exists(explicit) else (None, f"{explicit} not found")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_bench_{model}.json")))
    if not files:
        return None, "no PMC summary under profiles/ for this model"
    try:
        with open(files[-1]) as f:
            rec = json.load(f).get("kernel_sources_sha")
    except Exception as e:
        return None, f"{os.path.basename(files[-1])}: {e}"
    now = kernel_sources_sha()
    if rec != now:
        return None, (f"{os.path.basename(files[-1])} was collected on kernel sources {str(rec)[:12]}, the tree has {now[:12]}: "
                      "re-run tools/profile_round.sh (PMC passes last)")
    return files[-1], None


def kernel_sources_sha() -> str:
    """sha256 over the decode GEMV kernel sources (what the PMC traffic figure describes)."""
    import hashlib
    hsh = hashlib.sha256()
    for fn in ("gemv_impl.hpp", "gemv.hip", "chain_impl.hpp", "chain.hip", "common.hpp"):
        with open(os.path.join(ROOT, "paroquant_amd", "csrc", fn), "rb") as f:
            hsh.update(f.read())
    return hsh.hexdigest()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None,
                    help="model name (single GPU / replicas) or <model>-tp (tensor parallel over --gpus ranks); "
                         "default: qwen3-4b at --gpus 1, llama3-70b-tp above")
    ap.add_argument("--model-config", default="", help="HF config.json to register as a workload (use with --workload <dir name>)")
    ap.add_argument("--layers", type=int, default=0, help="decoder layers to instantiate (0 = all)")
    ap.add_argument("--rows", type=int, default=1, help="sequences decoded per step (batched decode; 1..16, with --route fused 1..512)")
    ap.add_argument("--route", default="auto", choices=["auto", "fused", "parts", "chain", "engine", "engine2"],
                    help="fused = rotation inside every consuming GEMV; chain = activations handed over rotated by the producing "
                         "launch (decode-chain family); parts = fused with the deferred K-split reduction of o / down (one row, one GPU); "
                         "auto = parts at one row on one GPU, chain at 2..16 rows, fused under tensor parallelism (measured: profiles/r03_chain_rows_sweep.jsonl, r03_parts_bench.jsonl)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-route-ab", action="store_true", help="skip the second one-row leg (the other of fused / parts) that `config.route_ab` reports")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end decode leg (fused harness: attention, norms, lm_head)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--per-shape", action="store_true", help="also print a per-linear GEMV table of the workload's model to stderr")
    ap.add_argument("--no-extra", action="store_true", help="skip extra.prefill (MFMA utilisation at M = 65536) and extra.configs (BASELINE configs 1 / 2 decode steps)")
    ap.add_argument("--no-north-star", action="store_true",
                    help="skip extra.llama3_8b_per_shape (the five Llama-3-8B GEMV shapes of BASELINE.md section 3, timed by the default run)")
    ap.add_argument("--pmc-file", default="", help="PMC summary json for roofline.traffic (default: newest profiles/rNN_pmc_bench_<model>.json)")
    ap.add_argument("--tp-backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend of the N > 1 run; gloo (+ --same-device) lets the whole TP path -- sharded HIP kernels, "
                         "all-reduce placement, max-over-ranks clock -- run with N processes on ONE GPU (tests; eager, no graph)")
    ap.add_argument("--same-device", action="store_true", help="all ranks use cuda:0 (only with --tp-backend gloo)")
    ap.add_argument("--no-oneshot", action="store_true",
                    help="TP runs: use the backend's all-reduce (RCCL / gloo) instead of the one-shot xGMI kernel")
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise only the multi-process plumbing (spawn, rendezvous, barrier, max-over-ranks timing, JSON) "
                         "with backend gloo and a trivial CPU step -- used by the CPU test-suite; prints data: 'dry-run'")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    if "WORLD_SIZE" in os.environ:        # launched by torchrun / the driver: one rank per process already
        run(args, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ["WORLD_SIZE"]))
        return
    if args.gpus <= 1:
        run(args, 0, 0, 1)
        return
    # plain `python bench.py --gpus N`: spawn the N ranks here (one process per GPU, rendezvous on 127.0.0.1)
    import socket
    import torch.multiprocessing as mp
    if not args.dry_run and not args.same_device and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(_spawned, args=(args,), nprocs=args.gpus, join=True)


def _spawned(local_rank: int, args):
    os.environ["RANK"] = os.environ["LOCAL_RANK"] = str(local_rank)
    os.environ["WORLD_SIZE"] = str(args.gpus)
    run(args, local_rank, local_rank, args.gpus)


def run(args, rank: int, local_rank: int, world: int):
    if args.model_config:
        registered = register_model_from_config(args.model_config)
        if args.workload is None:
            args.workload = registered + ("-tp" if world > 1 else "")
    workload = args.workload or ("qwen3-4b" if world == 1 else "llama3-70b-tp")
    tp_mode = workload.endswith("-tp")
    model = workload[:-3] if tp_mode else workload
    if model not in MODELS and model not in HYBRID:
        raise SystemExit(f"unknown workload {workload!r}; known models: {known_models()} (the dense ones also as <model>-tp)")
    if args.rows > 16 and args.route == "auto":
        args.route = "fused"      # above 16 rows only the per-call route exists (the chain family is a <= 16-row decode path)
    if not 1 <= args.rows <= (512 if args.route == "fused" else 16):
        raise SystemExit("--rows must be in 1..16 (--route fused, the per-call route: 1..512)")
    tp = world if tp_mode else 1

    if args.dry_run:
        dev = torch.device("cpu")
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        w = torch.randn(256, 256)
        buf = torch.randn(1, 256)

        def fn():
            h = buf @ w
            if world > 1 and tp_mode:
                dist.all_reduce(h)
            return h
        wall, ev_ms = time_steps_cpu(fn, args.steps, args.warmup, world)
        result = {"metric": "dry-run (multi-process plumbing only)", "value": round((1 if tp_mode else world) * args.steps / wall, 2),
                  "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": round(wall * 1e3 / args.steps, 4), "higher_is_better": True,
                  "scaling": "strong" if tp_mode else "weak", "vs_baseline": None, "dtype": "f32", "data": "dry-run",
                  "config": {"workload": workload, "parallelism": ("tp%d" % tp) if tp_mode else ("dp%d" % world)}}
        if rank == 0:
            print(json.dumps(result), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the ParoQuant hot path)")
    if args.same_device and args.tp_backend != "gloo":
        raise SystemExit("--same-device needs --tp-backend gloo (RCCL refuses two ranks on one GPU)")
    dev = torch.device("cuda", 0 if args.same_device else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.tp_backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import paroquant_amd  # noqa: F401
    from paroquant_amd import _native
    _native.load()

    if tp_mode and shard_error(model, tp):
        raise SystemExit(shard_error(model, tp))
    allreduce, allreduce_name = None, None
    if tp_mode and world > 1:
        from paroquant_amd import tp as ptp
        allreduce, allreduce_name = ptp.make_allreduce(dev, hidden_of(model), prefer_oneshot=not args.no_oneshot)
        if allreduce_name == "gloo":
            args.no_graph = True          # a gloo collective is a host operation: nothing to capture
    stack = DecodeStack(model, dev, n_layers=args.layers or None, tp=tp, rank=rank, allreduce=allreduce, rows=args.rows, route=args.route)

    def measure():
        """Eager warm-up, HIP-graph capture of the step (when the collective allows it), timed region."""
        if world > 1:
            dist.barrier()                  # ranks finish building their shards seconds apart; a kernel-level collective's wait is bounded
        out = stack.step(stack.x)           # eager warm-up (also sizes the shared workspace)
        torch.cuda.synchronize(dev)
        assert torch.isfinite(out.float()).all(), "non-finite activations in the synthetic decode chain"
        graphed = not args.no_graph
        fn = lambda: stack.step(stack.x)
        if graphed:
            try:
                s = torch.cuda.Stream(dev)
                s.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(s):
                    stack.step(stack.x)
                torch.cuda.current_stream(dev).wait_stream(s)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):   # RCCL's watchdog thread must not void the capture
                    stack.step(stack.x)
                graph.replay()
                torch.cuda.synchronize(dev)
                fn = graph.replay
            except Exception as e:           # e.g. an RCCL build that cannot be captured: measure eagerly, say so
                if tp <= 1:
                    raise
                print(f"[bench] HIP-graph capture of the TP step failed ({type(e).__name__}: {e}); timing eager launches",
                      file=sys.stderr, flush=True)
                graphed = False
                torch.cuda.synchronize(dev)
        w, e = time_steps(fn, args.steps, args.warmup, world, dev)
        return w, e, graphed

    wall, ev_ms, use_graph = measure()
    if allreduce_name == "oneshot":
        # a one-shot all-reduce that ever timed out on a peer produced garbage: never report such a run -- fall back to the
        # backend's collective on every rank and measure again
        bad = torch.tensor([1 if allreduce.gave_up() else 0], dtype=torch.int32, device=dev if args.tp_backend == "nccl" else "cpu")
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            if rank == 0:
                print(f"[bench] the one-shot all-reduce gave up waiting for a peer; re-measuring with {args.tp_backend}", file=sys.stderr, flush=True)
            allreduce_name = args.tp_backend
            stack.allreduce = lambda y: (dist.all_reduce(y), y)[1]
            stack.fused_ar = None
            if args.tp_backend == "gloo":
                args.no_graph = True
            wall, ev_ms, use_graph = measure()
    # one row, one GPU: the same stack through the decoder-loop routes as well (deferred K-split reductions; the persistent engine):
    # config.route_ab = {route: {ms_per_step, roofline_frac, max_rel_diff_vs_headline}}
    route_ab = None
    if stack.route in ("parts", "fused", "engine", "engine2") and not tp_mode and args.rows == 1 and not args.no_route_ab:
        frac_of = lambda w_s: round(stack.bytes_per_step * args.steps / w_s / 1e9 / HBM_PEAK_GBPS, 4)
        route_ab = {stack.route: {"ms_per_step": round(wall * 1e3 / args.steps, 4), "roofline_frac": frac_of(wall),
                                  "reachable_through": ROUTE_REACH[stack.route]}}
        mine = stack.route
        y_mine = stack.step(stack.x).clone()
        # (the persistent engines are timed only on a `make EXPERIMENTAL=1` library: they lost on every shape -- 0.930 / 1.204 ms against
        # 0.929 per call on this workload, profiles/r05_bench_qwen3-4b.jsonl -- and left the default library in round 6)
        from paroquant_amd import _native as _nat
        routes = ("fused", "parts") + (("engine", "engine2") if _nat.has_experimental() else ())
        for other in [r for r in routes if r != mine]:
            try:
                stack.use_route(other)
                y_other = stack.step(stack.x).clone()
                torch.cuda.synchronize(dev)
                w2, _, _ = measure()
                # same chain of linears; another K partition changes the fp32 summation order: the difference is chain-amplified rounding
                # (144 linears), gated at the north star's 1e-2
                rel = float((y_mine.float() - y_other.float()).abs().max() / y_mine.float().abs().max())
                route_ab[other] = {"ms_per_step": round(w2 * 1e3 / args.steps, 4), "roofline_frac": frac_of(w2),
                                   # context only: two routes cut K differently, so after the whole chain their fp32 summation orders have
                                   # diverged by chain-amplified ROUNDING -- parity is gated against the oracle below, not here
                                   "rounding_divergence_from_headline_after_%d_linears" % stack.launches_per_step: rel,
                                   "reachable_through": ROUTE_REACH[other]}
            except Exception as e:
                route_ab[other] = {"error": f"{type(e).__name__}: {e}"}
            finally:
                stack.route = mine
        try:       # parity of every route: against the float64 oracle chain
            chk, how = route_oracle_check(model, dev, [r for r in route_ab if "error" not in route_ab[r]])
            for r, v in chk.items():
                route_ab[r].update(v)
            route_ab["oracle_check"] = how
        except Exception as e:
            route_ab["oracle_check"] = f"failed: {type(e).__name__}: {e}"
    # TP runs report BOTH collectives: the one-shot xGMI kernel (when it came up and passed its self-test)
    # and the backend's all-reduce (RCCL), timed on the same shards; the headline is the faster leg that is healthy.
    allreduce_ab = None
    if tp_mode and world > 1:
        allreduce_ab = {allreduce_name: {"ms_per_step": round(wall * 1e3 / args.steps, 4), "hip_graph": use_graph}}
        if allreduce_name == "oneshot":
            saved = (stack.allreduce, stack.fused_ar, args.no_graph)
            stack.allreduce = lambda y: (dist.all_reduce(y), y)[1]
            stack.fused_ar = None
            if args.tp_backend == "gloo":
                args.no_graph = True
            try:
                wall2, ev2, graph2 = measure()
                allreduce_ab[args.tp_backend] = {"ms_per_step": round(wall2 * 1e3 / args.steps, 4), "hip_graph": graph2}
            except Exception as e:            # the library leg must never take the contract line down
                wall2, ev2, graph2 = float("inf"), 0.0, False
                allreduce_ab[args.tp_backend] = {"error": f"{type(e).__name__}: {e}"}
            faster = torch.tensor([1.0 if wall2 < wall else 0.0], device=dev if args.tp_backend == "nccl" else "cpu")
            dist.all_reduce(faster, op=dist.ReduceOp.MIN)        # every rank takes the same leg
            if float(faster.item()) > 0.5:
                wall, ev_ms, use_graph, allreduce_name = wall2, ev2, graph2, args.tp_backend
            else:
                stack.allreduce, stack.fused_ar, args.no_graph = saved
        else:
            allreduce_ab["oneshot"] = {"error": "not available on this node (set-up or self-test failed; see the log line above)"
                                       if not args.no_oneshot else "disabled (--no-oneshot)"}
    ms_per_step = wall * 1e3 / args.steps
    replicas = 1 if tp_mode else world
    tokens_per_s = replicas * args.rows * args.steps / wall

    launches = stack.launches_per_step
    us_per_launch = ev_ms * 1e3 / (args.steps * launches)
    bytes_per_launch = stack.bytes_per_step / launches
    achieved = bytes_per_launch / us_per_launch / 1e3      # GB/s, per rank
    traffic = None   # HBM bytes per launch from PMC counters (separate rocprofv3 --pmc passes, summaries under profiles/)
    pmc_file, pmc_reason = newest_pmc_file(model, args.pmc_file)
    if pmc_file and not tp_mode and stack.n_layers == n_layers_of(model) and args.rows == 1 and stack.route in ("fused", "parts"):
        with open(pmc_file) as f:
            traffic = json.load(f).get("traffic_bytes_per_launch")
    elif pmc_file:
        pmc_reason = "the PMC summary describes the default one-row workload at full depth"
    # L2-side traffic: what the CUs pull through L2 per launch -- TCP -> TCC read requests x 128 B from the round's
    # rocprofv3 --pmc TCP_TCC_READ_REQ_sum pass (tools/pmc_l2.py) -- next to the HBM-side figure; same freshness rule (kernel_sources_sha)
    l2_traffic, l2_source = None, None
    if traffic is not None:
        import glob
        l2_files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_l2_{model}.json")))
        try:
            if l2_files:
                with open(l2_files[-1]) as f:
                    l2 = json.load(f)
                if l2.get("kernel_sources_sha") == kernel_sources_sha():
                    l2_traffic = int(round(l2["mean_MB_through_the_CUs_per_launch"]["at_128B"] * 1e6))
                    l2_source = os.path.relpath(l2_files[-1], ROOT)
        except Exception:
            l2_traffic, l2_source = None, None
    roofline = {"bound": "hbm", "kernel": ("paro::chain_kernel (INT4 GEMV on rotated activations + the consumer's rotation in the epilogue"
                                           if stack.route == "chain" else ("paro::gemv_kernel (fused rotate+INT4 GEMV; K-split reductions deferred into the consuming launch"
                                                                           if stack.route == "parts" else "paro::gemv_kernel (fused rotate+INT4 GEMV")) + ", all launches of the step)",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "traffic_source": os.path.relpath(pmc_file, ROOT) if (pmc_file and traffic is not None) else None,
                "traffic_note": None if traffic is not None else pmc_reason,
                "l2_traffic": l2_traffic, "l2_traffic_source": l2_source,      # bytes through the CUs' vector caches per launch (L2 -> CU), or null
                "bytes_per_launch": int(bytes_per_launch), "us_per_launch": round(us_per_launch, 3),
                "launches_per_step": launches,
                "note": "per rank; launch duration = HIP-event time of the timed region / launches (includes inter-kernel gaps"
                        + (" and the all-reduces" if tp > 1 else "") + ")"}

    result = {
        "metric": "decode tokens/s through the ParoQuant quantised-linear hot path (fused rotation + INT4 GEMV), batch %d" % args.rows,
        "value": round(tokens_per_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong" if tp_mode else "weak", "vs_baseline": None, "dtype": "f16 activations x int4 weights (fp32 accumulate)",
        "data": "synthetic (random INT4 AWQ-format weights, random fp16 activations, random perfect-matching pairs)",
        "config": {"workload": f"{model}-PARO batch-{args.rows} decode: {stack.n_layers} layers x ({', '.join(n + ('[P=%d]' % len(sz) if len(sz) > 1 else '') for n, _, sz, _ in stack.shapes)}"
                               f"{'; every %d-th layer full attention: ' % HYBRID[model][8] + ', '.join(n for n, _, _, _ in hybrid_layer_shapes(model, True)[:2]) if model in HYBRID else ''}) "
                               f"W4A16 g128 krot8, {('TP=%d (%s all-reduce after o / down)' % (tp, {'oneshot': 'one-shot over xGMI' + (' in the GEMV epilogue' if stack.fused_ar is not None else ' kernel'), 'nccl': 'RCCL'}.get(allreduce_name, allreduce_name))) if tp_mode else 'replica per GPU'}",
                   "layers": stack.n_layers, "hidden": stack.hidden, "hip_graph": use_graph, "rows": args.rows, "route": stack.route,
                   "bytes_per_token": stack.bytes_per_step * (tp if tp_mode else 1),
                   "parallelism": ("tp%d" % tp) if tp_mode else ("dp%d" % world),
                   "collective_backend": (args.tp_backend if world > 1 and tp_mode else None),
                   "allreduce": allreduce_name, "allreduce_ab": allreduce_ab, "route_ab": route_ab},
        "roofline": roofline,
    }

    if tp_mode and world > 1:
        # like-for-like reference for the scaling curve: the SAME model unsharded on one GPU (rank 0), measured on a few
        # layers and scaled to the full depth (per-layer time is depth-independent: distinct weights per layer)
        ref = None
        if rank == 0:
            try:
                nl = min(8, n_layers_of(model))
                s1 = DecodeStack(model, dev, n_layers=nl, tp=1, rank=0, seed=123)
                s1.step(s1.x)
                torch.cuda.synchronize(dev)
                g1 = torch.cuda.CUDAGraph()
                sst = torch.cuda.Stream(dev)
                sst.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(sst):
                    s1.step(s1.x)
                torch.cuda.current_stream(dev).wait_stream(sst)
                with torch.cuda.graph(g1):
                    s1.step(s1.x)
                _, ev1 = time_steps(g1.replay, 20, 3, 1, dev)
                ms_layer = ev1 / 20 / nl
                ref = {"tokens_per_s": round(1e3 / (ms_layer * n_layers_of(model)), 2), "layers_measured": nl,
                       "note": "same model, TP=1, one GPU; per-layer time x full depth"}
                del s1, g1
            except Exception as e:
                ref = {"error": f"{type(e).__name__}: {e}"}
        dist.barrier()
        result["config"]["tp1_reference"] = ref
        # the whole model tensor-parallel on the fused harness (attention heads sharded, residual added inside the one-shot
        # all-reduce): every rank runs it, never fatal for the contract line; a one-shot that gave up is reported, not hidden
        if not args.no_e2e and allreduce_name == "oneshot" and stack.n_layers == n_layers_of(model):
            del stack
            torch.cuda.empty_cache()
            try:
                e2e = end_to_end(model, dev, tp_rank=rank, tp_world=world, allreduce=allreduce)
                if e2e is not None and allreduce.gave_up():
                    e2e = {"error": "the one-shot all-reduce timed out waiting for a peer during the end-to-end leg"}
            except Exception as e:
                e2e = {"error": f"{type(e).__name__}: {e}"}
            result["end_to_end"] = e2e
            stack = None

    if rank == 0:
        if args.per_shape:
            for row in per_shape_table(model, dev):
                print(json.dumps(row), file=sys.stderr, flush=True)
        if world == 1 and not tp_mode and args.rows == 1 and not args.no_north_star:
            # the north star's named shapes, in the ONE contract line: bytes, us per launch, fraction of 8 TB/s
            try:
                tab = per_shape_table("llama3-8b", dev, reps=300, shapes=NORTH_STAR_SHAPES, min_bytes=768 << 20)
                result["extra"] = {"llama3_8b_per_shape": tab, "target_frac": 0.70,
                                   "note": "batch-1 fused rotate + INT4 GEMV per operator call, HIP graph of 300 launches over >= 768 MiB of distinct weights; "
                                           "floor_frac = (2.4 us + bytes / 7.5 TB/s) / measured: the share of a launch that a kernel which only streamed its bytes would also take"}
            except Exception as e:
                result["extra"] = {"llama3_8b_per_shape": {"error": f"{type(e).__name__}: {e}"}}
        if world == 1 and not tp_mode and args.rows == 1 and not args.no_extra:
            ex = result.setdefault("extra", {})
            try:       # the north star's second metric, driver-visible: prefill MFMA utilisation at M = 32 x 2048
                ex["prefill"] = {"rows": prefill_table(dev), "peak_TFLOPs": MFMA_PEAK_TFLOPS,
                                 "note": "fused linear per operator call at M = 65536 (BASELINE config 3's batch-32 x seq-2048 prefill): rotation pre-pass on the "
                                         "matrix cores + W4A16 MFMA GEMM; mfma_util = 2 M K N / time / dense fp16 peak; prepass_share = the pre-pass launch alone / the call; "
                                         "fused_rotation_experiment_ms = the same call as ONE launch with the rotation applied to the LDS-staged slab inside the GEMM "
                                         "(north_star's wording; GEMM variant 44, same bits, profiles/NOTES.md 6.11) -- measured, slower, never selected"}
            except Exception as e:
                ex["prefill"] = {"error": f"{type(e).__name__}: {e}"}
            try:       # BASELINE configs 1 and 2 (decode legs) next to the headline configuration
                ex["configs"] = config_steps(dev)
            except Exception as e:
                ex["configs"] = {"error": f"{type(e).__name__}: {e}"}
            try:       # batched decode through the boundary: step time at 1 / 2 / 4 / 8 / 16 rows, same weights
                ex["batched_decode"] = batched_decode_steps(dev)
            except Exception as e:
                ex["batched_decode"] = {"error": f"{type(e).__name__}: {e}"}
            try:       # medium prefill (128 / 512 / 2048 rows) per Llama-3-8B linear through the boundary
                ex["mid_prefill"] = mid_prefill_table(dev)
            except Exception as e:
                ex["mid_prefill"] = {"error": f"{type(e).__name__}: {e}"}
            try:       # the launch shapes of the headline workload's layers, MEASURED (paroquant_amd/autotune.py), plus three shapes no sweep saw
                from paroquant_amd import autotune as _at
                rows_at = []
                gen_at = torch.Generator(device=dev).manual_seed(99)
                seen_at = [(n, K, list(sz)) for n, K, sz, _ in layer_plan(model, 1, 1)[0]] + [("unseen 3584->18944", 3584, [18944]),
                                                                                              ("unseen 5120->27648", 5120, [27648]), ("unseen 6144->4096", 6144, [4096])]
                for name_at, K_at, sz_at in seen_at:
                    rep = synth_packed(K_at, sz_at, dev, gen_at).autotune(force=True)
                    cand = sorted(rep["candidates"].items(), key=lambda kv: kv[1])
                    rows_at.append({"linear": name_at, "K": K_at, "N": sum(sz_at), "rule_tree": rep["default"], "rule_tree_us": rep["default_us"], "choice": rep["choice"],
                                    "choice_us": rep["choice_us"], "candidates": len(cand), "best_three": cand[:3]})
                ex["autotune"] = {"rows": rows_at, "note": "per layer shape: every legal (tiles_per_wave, ksplit, waves) timed at load time over rotating weight copies "
                                                          "(60 graph-replayed launches each); the rule tree's shape is kept unless one is >= 2 % faster; PARO_AUTOTUNE=1 "
                                                          "runs this inside process_weights_after_loading / RotateQuantizedLinear.prepare"}
            except Exception as e:
                ex["autotune"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(model, args.cpu_budget)
            try:       # SURVEY 8d's per-shape table (B1 torch-CPU, B2 C port); context, never fatal for the contract line
                result["cpu_baseline"]["per_shape"] = cpu_per_shape(max(4.0, args.cpu_budget * 0.6))
            except Exception as e:
                result["cpu_baseline"]["per_shape"] = {"error": f"{type(e).__name__}: {e}"}
        else:
            result["cpu_baseline"] = None
        if not args.no_e2e and world == 1 and not tp_mode and stack is not None and stack.n_layers == n_layers_of(model):
            del stack
            torch.cuda.empty_cache()
            try:
                result["end_to_end"] = end_to_end(model, dev)
            except Exception as e:       # reported, never fatal for the contract line
                result["end_to_end"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def time_steps_cpu(fn, steps: int, warmup: int, world: int):
    for _ in range(warmup):
        fn()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = t[0].item()
    return wall, wall * 1e3


if __name__ == "__main__":
    main()
