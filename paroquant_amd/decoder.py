"""Decode harness / generator on the fused kernels (SURVEY 8 rows f2 + f3).

What the reference does for end-to-end decode: ``TransformersGenerator`` drives HF ``generate()`` on a model whose
quantised linears are ``RotateQuantizedLinear`` (transformers/generator.py:37-67), ``cli/benchmark.py:8-26`` times
2 warm-up + 5 runs of 128 greedy tokens, and ``BaseGenerator.generate`` defines
``tps = new_tokens / (t_end - t_first_token)`` (inference/base.py:62-77).  Per decoder layer and token that is
RMSNorm, 3 x (rotate + GEMM) + cat, rope, attention, rotate + GEMM, residual add, RMSNorm, 2 x (rotate + GEMM) + cat,
SiLU, mul, rotate + GEMM, residual add -- about 25 launches.

Here a Llama / Qwen3-style decoder layer is FIVE launches, all captured with the rest of the step in one HIP graph:

    qkv      fused GEMV, RMSNorm prologue (input_layernorm weight folded into the channel scales)     ops.w4a16_gemv_fused
    attn     q/k norm + RoPE + KV append + GQA over the cache                                      ops.attn_decode
    o        fused GEMV, residual epilogue
    gate_up  fused GEMV, RMSNorm prologue (post_attention_layernorm folded)
    down     fused GEMV, SiLU(gate) * up prologue, residual epilogue

On one GPU the K-split reduction of o and down is DEFERRED (include/paro_abi.h v12, ``_layers_deferred``): their K-slices leave
fp32 partial sums and exit -- no in-launch hand-off (1.3 .. 1.45 us per launch) --, and the RMSNorm-prologue launch behind them
(gate_up; the next layer's qkv) adds them to the residual stream while it seeds its rotation and writes the new stream (0.6 .. 0.8 us:
every workgroup reads the four fp32 slots of every channel).  Same launches, same bits (tests/test_gpu_parts.py,
tests/test_gpu_parity.py::test_decoder_harness_deferred_matches_reducer); Qwen3-4B 691 -> 718 tokens/s, Llama-3-8B 612 -> 628
(profiles/r03_e2e.jsonl vs r03_e2e_reducer.jsonl).  ``PARO_DEFERRED_KSPLIT=0`` keeps the in-launch reducer.  The qkv projection
defers too where its launch shape splits (2-way): its partial sums and the K-slices' sums of squares go to the attention kernel,
which completes q / k / v as it reads them (``PARO_DEFERRED_QKV=0``: off; the summation order of qkv changes, so this part is equal
to the reducer route within rounding, not bit for bit).

Embedding lookup, final norm, the (unquantised, fp16) lm_head and the greedy argmax are plain torch ops inside the
same graph; the token and position live in device tensors, so a replay is one whole token with no host round trip.
Prefill runs the same weights through ``PackedParoWeights.apply`` (MFMA GEMM) with torch attention.
"""
from __future__ import annotations

import json
import os
import time
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import _native as nat
from . import ops
from .linear import PackedParoWeights


@dataclass
class DecoderConfig:
    hidden: int
    inter: int
    n_heads: int
    n_kv_heads: int
    head_dim: int
    n_layers: int
    vocab: int
    rms_eps: float = 1e-6
    rope_theta: float = 10000.0
    qk_norm: bool = False          # Qwen3: per-head RMSNorm on q and k before RoPE
    max_positions: int = 2048
    rope_scaling: Optional[dict] = None   # HF `rope_scaling` / `rope_parameters`: rope_type default | linear | llama3

    def __post_init__(self):
        # the attention kernel reads the position-contiguous V cache 8 positions (16 bytes) at a time
        self.max_positions = (int(self.max_positions) + 7) // 8 * 8

    @classmethod
    def from_hf(cls, c: dict, max_positions: int = 2048) -> "DecoderConfig":
        c = c.get("text_config", c)
        hd = c.get("head_dim") or c["hidden_size"] // c["num_attention_heads"]
        rp = c.get("rope_parameters") or {}
        rope_theta = c.get("rope_theta") or rp.get("rope_theta", 10000.0)
        # z-lab/Llama-3.1-8B-Instruct-PARO carries llama3 rope scaling: the low-frequency inv_freq change at EVERY position
        scaling = c.get("rope_scaling") or ({k: v for k, v in rp.items() if k not in ("rope_theta", "partial_rotary_factor")}
                                            if rp.get("rope_type", rp.get("type", "default")) != "default" else None)
        cfg = cls(c["hidden_size"], c["intermediate_size"], c["num_attention_heads"],
                  c.get("num_key_value_heads", c["num_attention_heads"]), hd, c["num_hidden_layers"], c["vocab_size"],
                  c.get("rms_norm_eps", 1e-6), float(rope_theta), c.get("model_type", "") in ("qwen3",), max_positions, scaling)
        rope_inv_freq(cfg, "cpu")      # unsupported scaling types fail HERE, at load time, not as silently wrong rotary angles
        # (newer configs keep the factor inside `rope_parameters`, next to theta and the scaling type)
        if float(c.get("partial_rotary_factor", 1.0)) != 1.0 or float(rp.get("partial_rotary_factor", 1.0)) != 1.0:
            raise NotImplementedError("partial rotary embeddings are not supported by the decode harness")
        return cfg


MODEL_CONFIGS = {
    # name: (hidden, inter, heads, kv_heads, head_dim, layers, vocab, qk_norm, rope_theta)
    "qwen3-0.6b": (1024, 3072, 16, 8, 128, 28, 151936, True, 1e6),
    "qwen3-4b": (2560, 9728, 32, 8, 128, 36, 151936, True, 1e6),
    "llama3-8b": (4096, 14336, 32, 8, 128, 32, 128256, False, 5e5),
    "llama3-70b": (8192, 28672, 64, 8, 128, 80, 128256, False, 5e5),
}


def deferred_route_pays(hidden: int) -> bool:
    """ONE gating predicate for the deferred K-split reduction, shared by the decode harness and bench.py's `parts` route: measured
    -2.8 % per step (Qwen3-4B) .. -1.1 % (Qwen3-0.6B), +0.3 % on 70B-class widths (hidden >= 8192: the launches are long enough that
    the in-launch hand-off no longer shows; profiles/r03_parts_bench.jsonl).  PARO_DEFERRED_KSPLIT=0 keeps the in-launch reducer."""
    return hidden < 8192 and os.environ.get("PARO_DEFERRED_KSPLIT", "1") != "0"


def named_config(name: str, max_positions: int = 2048) -> DecoderConfig:
    h, i, nh, nkv, hd, L, V, qk, th = MODEL_CONFIGS[name]
    return DecoderConfig(h, i, nh, nkv, hd, L, V, 1e-6, th, qk, max_positions)


def rope_inv_freq(cfg: DecoderConfig, device) -> torch.Tensor:
    """inv_freq[d] = theta^(-2d / head_dim), then the checkpoint's rope scaling as HF's ROPE_INIT_FUNCTIONS apply it:
    "linear" divides by `factor`; "llama3" (Llama-3.1+) divides the low frequencies by `factor`, keeps the high ones and
    blends in between (transformers/modeling_rope_utils.py, _compute_llama3_parameters).  Anything else raises."""
    import math
    half = cfg.head_dim // 2
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, half, dtype=torch.float32, device=device) * 2.0 / cfg.head_dim))
    sc = cfg.rope_scaling
    if not sc:
        return inv
    kind = sc.get("rope_type", sc.get("type", "default"))
    if kind == "default":
        return inv
    if kind == "linear":
        return inv / float(sc["factor"])
    if kind == "llama3":
        factor, lo, hi = float(sc["factor"]), float(sc["low_freq_factor"]), float(sc["high_freq_factor"])
        old = float(sc["original_max_position_embeddings"])
        wavelen = 2.0 * math.pi / inv
        scaled = torch.where(wavelen > old / lo, inv / factor, inv)
        smooth = (old / wavelen - lo) / (hi - lo)
        blended = (1.0 - smooth) * inv / factor + smooth * inv
        medium = ~(wavelen < old / hi) & ~(wavelen > old / lo)
        return torch.where(medium, blended, scaled)
    raise NotImplementedError(f"rope scaling type {kind!r} is not supported by the decode harness (default, linear, llama3 are)")


def rope_table(cfg: DecoderConfig, device) -> torch.Tensor:
    """fp32 [max_positions, head_dim]: cos(pos * inv_freq) then sin(...) (the rotary embedding of HF Llama / Qwen3, with
    the checkpoint's rope scaling: `rope_inv_freq`)."""
    inv = rope_inv_freq(cfg, device)
    ang = torch.arange(cfg.max_positions, dtype=torch.float32, device=device)[:, None] * inv[None, :]
    return torch.cat([ang.cos(), ang.sin()], dim=-1).contiguous()


class _Layer:
    __slots__ = ("qkv", "o", "gate_up", "down", "q_norm", "k_norm", "kcache", "vcache", "in_norm", "post_norm")


class ParoDecoderLM:
    """Greedy decoder over ParoQuant linears.  Build with :meth:`from_checkpoint` (an HF ``*-PARO`` directory,
    Llama / Qwen3 naming) or :meth:`random` (synthetic weights of a named architecture, for benchmarks)."""

    def __init__(self, cfg: DecoderConfig, device, dtype=torch.float16, tp_rank: int = 0, tp_world: int = 1, allreduce=None):
        """``tp_world > 1``: this object is ONE rank of a Megatron tensor-parallel model (one process per GPU): qkv / gate_up
        column-parallel (this rank's heads and MLP columns), o / down row-parallel followed by ``allreduce(part, residual,
        out)`` (``paroquant_amd.tp.make_allreduce``), embedding / norms / lm_head replicated; every rank computes the same
        token (the all-reduce sums in rank order, bit-identical on all ranks)."""
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        self.tp_rank, self.tp_world, self.allreduce = int(tp_rank), int(tp_world), allreduce
        # a one-shot all-reduce object can also run inside the row-parallel GEMV's epilogue (no launch of its own)
        self.fused_allreduce = hasattr(allreduce, "fusion_args") and os.environ.get("PARO_FUSED_ALLREDUCE", "1") != "0"
        self.tp_group = getattr(allreduce, "group", None)       # the prefill's [T, hidden] all-reduce goes through the process group
        if cfg.n_heads % self.tp_world or cfg.n_kv_heads % self.tp_world or cfg.inter % self.tp_world:
            raise ValueError(f"heads {cfg.n_heads} / kv heads {cfg.n_kv_heads} / intermediate {cfg.inter} do not split {self.tp_world}-way")
        self.nh, self.nkv, self.inter_l = cfg.n_heads // self.tp_world, cfg.n_kv_heads // self.tp_world, cfg.inter // self.tp_world
        if self.tp_world > 1 and ((self.nh * cfg.head_dim) % 128 or self.inter_l % 128 or allreduce is None):
            raise ValueError("tensor parallelism needs K slices in multiples of 128 (the rotation group) and an all-reduce")
        self.layers: List[_Layer] = []
        self.embed = self.lm_head = self.final_norm = None
        self.rope = rope_table(cfg, self.device)
        self._graph = None

    # ------------------------------------------------------------------ construction
    def _alloc_cache(self, layer: _Layer):
        c = self.cfg
        layer.kcache = torch.zeros(self.nkv, c.max_positions, c.head_dim, dtype=self.dtype, device=self.device)
        # V is kept position-contiguous ([head][dim][position]): the attention kernel's P V product reads it as MFMA fragments
        layer.vcache = torch.zeros(self.nkv, c.head_dim, c.max_positions, dtype=self.dtype, device=self.device)

    @classmethod
    def random(cls, name_or_cfg, device, n_layers: Optional[int] = None, max_positions: int = 1024, seed: int = 0,
               vocab: Optional[int] = None, dtype: torch.dtype = torch.float16, tp_rank: int = 0, tp_world: int = 1,
               allreduce=None) -> "ParoDecoderLM":
        import bench  # synthetic checkpoint-format layers (repo root on sys.path in tools / tests / bench)
        cfg = named_config(name_or_cfg, max_positions) if isinstance(name_or_cfg, str) else name_or_cfg
        if n_layers:
            cfg.n_layers = n_layers
        if vocab:
            cfg.vocab = vocab
        self = cls(cfg, device, dtype, tp_rank, tp_world, allreduce)
        gen = torch.Generator(device=self.device)      # the rank's own shard of the quantised linears ...
        gen.manual_seed(seed + 1000 * tp_rank)
        shared = torch.Generator(device=self.device)   # ... and what every rank holds identically (norms, embedding, lm_head)
        shared.manual_seed(seed + 77)
        q, kv = self.nh * cfg.head_dim, self.nkv * cfg.head_dim
        rnd = lambda *s: torch.randn(*s, device=self.device, generator=shared)
        for _ in range(cfg.n_layers):
            L = _Layer()
            L.qkv = bench.synth_packed(cfg.hidden, [q, kv, kv], self.device, gen)
            L.o = bench.synth_packed(q, [cfg.hidden], self.device, gen)
            L.gate_up = bench.synth_packed(cfg.hidden, [self.inter_l, self.inter_l], self.device, gen)
            L.down = bench.synth_packed(self.inter_l, [cfg.hidden], self.device, gen)
            L.in_norm = (1.0 + 0.05 * rnd(cfg.hidden)).to(self.dtype)
            L.post_norm = (1.0 + 0.05 * rnd(cfg.hidden)).to(self.dtype)
            L.qkv.fold_norm_weight(L.in_norm)
            L.gate_up.fold_norm_weight(L.post_norm)
            L.q_norm = (1.0 + 0.05 * rnd(cfg.head_dim)).to(self.dtype) if cfg.qk_norm else None
            L.k_norm = (1.0 + 0.05 * rnd(cfg.head_dim)).to(self.dtype) if cfg.qk_norm else None
            self._alloc_cache(L)
            self.layers.append(L)
        self.embed = (rnd(cfg.vocab, cfg.hidden) * 0.5).to(self.dtype)
        self.lm_head = (rnd(cfg.vocab, cfg.hidden) * (cfg.hidden ** -0.5)).to(self.dtype)
        self.final_norm = (1.0 + 0.05 * rnd(cfg.hidden)).to(self.dtype)
        self._static()
        return self

    @classmethod
    def from_raw(cls, cfg: DecoderConfig, raw_layers, shared: Dict[str, torch.Tensor], device, dtype: torch.dtype = torch.float16,
                 tp_rank: int = 0, tp_world: int = 1, allreduce=None) -> "ParoDecoderLM":
        """Build from checkpoint-format tensors of the FULL model and keep this rank's Megatron shard (reference:
        vllm/plugin.py:33-50,196-198; ``paroquant_amd.tp``).  ``raw_layers[l]`` = dict(qkv=, o=, gate_up=, down=: merged-layer
        dicts with qweight / qzeros / scales / theta [P, krot, K/2] / pairs [P, krot, K] / channel_scales [P, 1, K] / sizes;
        in_norm=, post_norm=, q_norm=, k_norm=); ``shared`` = dict(embed=, lm_head=, final_norm=)."""
        from . import tp as ptp
        self = cls(cfg, device, dtype, tp_rank, tp_world, allreduce)
        dev = self.device
        tt = lambda v: v if isinstance(v, torch.Tensor) else torch.from_numpy(v)

        def pack(d, kind):
            lay = {k: tt(d[k]) for k in ("qweight", "qzeros", "scales", "theta", "pairs", "channel_scales")}
            sizes = list(d["sizes"])
            if tp_world > 1:
                if kind == "col":
                    lay = ptp.shard_column_parallel({**lay, "bias": None}, sizes, tp_rank, tp_world)
                    sizes = [n // tp_world for n in sizes]
                else:
                    lay = ptp.shard_row_parallel({**lay, "bias": None}, tp_rank, tp_world)
            return PackedParoWeights(lay["qweight"].to(dev), lay["qzeros"].to(dev), lay["scales"].to(dev), lay["theta"].to(dev),
                                     lay["pairs"].to(dev), lay["channel_scales"].to(dev), sizes)

        for r in raw_layers:
            L = _Layer()
            L.qkv, L.o = pack(r["qkv"], "col"), pack(r["o"], "row")
            L.gate_up, L.down = pack(r["gate_up"], "col"), pack(r["down"], "row")
            L.in_norm, L.post_norm = tt(r["in_norm"]).to(dev, dtype), tt(r["post_norm"]).to(dev, dtype)
            L.qkv.fold_norm_weight(L.in_norm)
            L.gate_up.fold_norm_weight(L.post_norm)
            L.q_norm = tt(r["q_norm"]).to(dev, dtype) if r.get("q_norm") is not None else None
            L.k_norm = tt(r["k_norm"]).to(dev, dtype) if r.get("k_norm") is not None else None
            self._alloc_cache(L)
            self.layers.append(L)
        self.embed, self.lm_head = tt(shared["embed"]).to(dev, dtype), tt(shared["lm_head"]).to(dev, dtype)
        self.final_norm = tt(shared["final_norm"]).to(dev, dtype)
        self._static()
        return self

    @classmethod
    def from_checkpoint(cls, path: str, device, max_positions: int = 2048, dtype: torch.dtype = torch.float16) -> "ParoDecoderLM":
        """Load an HF ``*-PARO`` checkpoint directory (config.json + safetensors; tensor names of cli/convert.py:264-277
        under the usual ``model.layers.N.{self_attn,mlp}.*_proj`` paths).  q/k/v and gate/up are merged into one
        fused linear each (3 / 2 rotations), the layer norms are folded into the channel scales."""
        from safetensors import safe_open
        with open(os.path.join(path, "config.json")) as f:
            hf = json.load(f)
        cfg = DecoderConfig.from_hf(hf, max_positions)
        self = cls(cfg, device, dtype)
        t: Dict[str, torch.Tensor] = {}
        for fn in sorted(os.listdir(path)):
            if fn.endswith(".safetensors"):
                with safe_open(os.path.join(path, fn), framework="pt") as f:
                    for k in f.keys():
                        t[k] = f.get_tensor(k)
        dev = self.device

        def merged(prefix: str, names) -> PackedParoWeights:
            g = lambda n, s: t[f"{prefix}.{n}.{s}"].to(dev)
            sizes = [int(t[f"{prefix}.{n}.scales"].shape[1]) for n in names]
            qw = torch.cat([g(n, "qweight") for n in names], dim=1)
            qz = torch.cat([g(n, "qzeros") for n in names], dim=1)
            sc = torch.cat([g(n, "scales") for n in names], dim=1)
            th = torch.stack([g(n, "theta") for n in names])
            pr = torch.stack([g(n, "pairs") for n in names])
            cs = torch.stack([g(n, "channel_scales").reshape(1, -1) for n in names])
            has_b = [f"{prefix}.{n}.bias" in t for n in names]
            if any(has_b) and not all(has_b):
                raise ValueError(f"{prefix}: some of {names} carry a bias and some do not -- cannot merge them into one linear")
            bias = torch.cat([g(n, "bias").reshape(-1) for n in names]).to(self.dtype) if all(has_b) else None   # attention_bias models
            return PackedParoWeights(qw, qz, sc, th, pr, cs, sizes, bias)

        for l in range(cfg.n_layers):
            p = f"model.layers.{l}"
            L = _Layer()
            L.qkv = merged(f"{p}.self_attn", ["q_proj", "k_proj", "v_proj"])
            L.o = merged(f"{p}.self_attn", ["o_proj"])
            L.gate_up = merged(f"{p}.mlp", ["gate_proj", "up_proj"])
            L.down = merged(f"{p}.mlp", ["down_proj"])
            L.in_norm = t[f"{p}.input_layernorm.weight"].to(dev, self.dtype)
            L.post_norm = t[f"{p}.post_attention_layernorm.weight"].to(dev, self.dtype)
            L.qkv.fold_norm_weight(L.in_norm)
            L.gate_up.fold_norm_weight(L.post_norm)
            qn, kn = f"{p}.self_attn.q_norm.weight", f"{p}.self_attn.k_norm.weight"
            L.q_norm = t[qn].to(dev, self.dtype) if qn in t else None
            L.k_norm = t[kn].to(dev, self.dtype) if kn in t else None
            self._alloc_cache(L)
            self.layers.append(L)
        self.cfg.qk_norm = self.layers[0].q_norm is not None
        self.embed = t["model.embed_tokens.weight"].to(dev, self.dtype)
        self.lm_head = (t["lm_head.weight"] if "lm_head.weight" in t else t["model.embed_tokens.weight"]).to(dev, self.dtype)
        self.final_norm = t["model.norm.weight"].to(dev, self.dtype)
        self._static()
        return self

    def _static(self):
        """Static buffers of the captured decode step (the graph replays on them)."""
        c, dev, dt = self.cfg, self.device, self.dtype
        self.tok = torch.zeros(1, dtype=torch.long, device=dev)        # current token id
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)       # its position
        self.h = torch.zeros(1, c.hidden, dtype=dt, device=dev)        # residual stream (ping)
        self.h2 = torch.zeros(1, c.hidden, dtype=dt, device=dev)       # residual stream (pong)
        qkv_w = (self.nh + 2 * self.nkv) * c.head_dim
        self.qkv_buf = torch.zeros(1, qkv_w, dtype=dt, device=dev)
        self.attn_buf = torch.zeros(1, self.nh * c.head_dim, dtype=dt, device=dev)
        self.gu_buf = torch.zeros(1, 2 * self.inter_l, dtype=dt, device=dev)
        self.part = torch.zeros(1, c.hidden, dtype=dt, device=dev)     # this rank's partial sum of a row-parallel linear (TP)
        # deferred K-split reduction of o / down (one GPU, every layer's o and down K-split by the automatic launch shape and carry
        # no bias; PARO_DEFERRED_KSPLIT=0 keeps the in-launch reducer)
        n_o = min((ops.gemv_parts_count(L.o, dt) for L in self.layers), default=0)
        n_d = min((ops.gemv_parts_count(L.down, dt) for L in self.layers), default=0)
        self.deferred = self.tp_world == 1 and n_o >= 2 and n_d >= 2 and deferred_route_pays(c.hidden)
        if self.deferred:
            self.parts_o = torch.zeros(c.hidden, nat.PARO_MAX_PARTIALS, dtype=torch.float32, device=dev)
            self.parts_d = torch.zeros(c.hidden, nat.PARO_MAX_PARTIALS, dtype=torch.float32, device=dev)
        # ... and of qkv (2-way in the launch shape nobody polls in): its consumer is the attention kernel, where each q / k / v element
        # is read by ONE workgroup; the RMSNorm scalar travels as the K-slices' sums of squares (row N).  PARO_DEFERRED_QKV=0: off
        self.deferred_qkv = self.deferred and min((ops.gemv_parts_count(L.qkv, dt) for L in self.layers), default=0) >= 2 \
            and os.environ.get("PARO_DEFERRED_QKV", "1") != "0"
        if self.deferred_qkv:
            self.parts_q = torch.zeros(qkv_w + 1, nat.PARO_MAX_PARTIALS, dtype=torch.float32, device=dev)
        self.fuse_qkv_attn = False      # (decided below, once the attention's buffers exist)
        # ... and the attention's merge over position chunks (ABI v14): the chunks of a head run on different CUs and leave their slots'
        # (max, sum, un-normalised output); o_proj completes the merge while it seeds its rotation.  PARO_SPLIT_ATTN=0: off
        self.split_attn = self.deferred and c.head_dim in (64, 128) and os.environ.get("PARO_SPLIT_ATTN", "1") != "0"
        if self.split_attn:
            self.attn_parts = torch.zeros(ops.attn_parts_floats(self.nh, c.head_dim), dtype=torch.float32, device=dev)
        # ... and the attention itself inside the qkv launch (ABI v18, paro_attn_tail_t): its workgroups ride in the projection's grid,
        # request their K / V lines while the projection streams and take q / k / v as {partial sum, launch tag} granules -- one
        # in-launch hand-over instead of a launch boundary.  PARO_FUSE_QKV_ATTN=0: off
        self.fuse_qkv_attn = self.deferred_qkv and self.split_attn and os.environ.get("PARO_FUSE_QKV_ATTN", "1") != "0" \
            and all(ops.attn_tail_supported(L.qkv, self.nh, self.nkv, c.head_dim, c.max_positions, dt) for L in self.layers)
        if self.fuse_qkv_attn:
            self.parts_q = torch.zeros(qkv_w + 1, 2 * nat.PARO_MAX_PARTIALS, dtype=torch.float32, device=dev)      # 8-byte granules
        self.logits = torch.zeros(1, c.vocab, dtype=dt, device=dev)
        self.out_tokens = torch.zeros(c.max_positions, dtype=torch.long, device=dev)
        # per-instance scratch (arrival tickets of the attention chunks): two decoders of the same geometry may run on
        # different streams at the same time and must not share tickets
        self.attn_ws = torch.zeros(nat.load().paro_attn_decode_workspace_bytes(self.nh, self.nkv, c.head_dim, c.max_positions),
                                   dtype=torch.uint8, device=dev)
        self.lm_ws = ops.lm_head_workspace(dev, c.vocab)
        self.lm_head = self.lm_head.contiguous()
        # the fused tail (final norm + lm_head GEMV + argmax, 2 launches) needs hidden = 512 * 1..8;
        # otherwise the torch ops (about 10 launches incl. a hipBLASLt GEMV) take its place
        self.fused_tail = c.hidden % 512 == 0 and c.hidden <= 4096
        self.bytes_per_token = sum(pk.nbytes() for L in self.layers for pk in (L.qkv, L.o, L.gate_up, L.down))

    # ------------------------------------------------------------------ one decode token (capturable)
    def decode_step(self) -> None:
        """Consume ``self.tok`` at position ``self.pos``; leave the logits in ``self.logits``, the greedy next token in
        ``self.tok`` and advance ``self.pos``.  No host synchronisation, no allocation by the fused ops."""
        c = self.cfg
        R, S = nat.PROLOGUE_RMSNORM, nat.PROLOGUE_SILU_MUL
        torch.index_select(self.embed, 0, self.tok, out=self.h)
        h, h2 = self.h, self.h2
        if self.deferred:
            h = self._layers_deferred()
        for L in (() if self.deferred else self.layers):
            ops.w4a16_gemv_fused(h, L.qkv, R, c.rms_eps, out=self.qkv_buf)
            ops.attn_decode(self.qkv_buf, L.kcache, L.vcache, self.pos, self.rope, self.nh, self.nkv, c.head_dim,
                            L.q_norm, L.k_norm, c.rms_eps, out=self.attn_buf, workspace=self.attn_ws)
            if self.tp_world == 1:
                ops.w4a16_gemv_fused(self.attn_buf, L.o, 0, residual=h, out=h2)                 # h2 = h + o(attn)
                ops.w4a16_gemv_fused(h2, L.gate_up, R, c.rms_eps, out=self.gu_buf)
                ops.w4a16_gemv_fused(self.gu_buf, L.down, S, residual=h2, out=h)                # h = h2 + down(act)
            elif self.fused_allreduce:   # row-parallel o / down exchange their partial outputs in their own epilogue: five launches
                ops.w4a16_gemv_fused(self.attn_buf, L.o, 0, residual=h, out=h2, allreduce=self.allreduce)
                ops.w4a16_gemv_fused(h2, L.gate_up, R, c.rms_eps, out=self.gu_buf)
                ops.w4a16_gemv_fused(self.gu_buf, L.down, S, residual=h2, out=h, allreduce=self.allreduce)
            else:   # row-parallel o / down: partial sums, one all-reduce each with the residual added in its summation
                ops.w4a16_gemv_fused(self.attn_buf, L.o, 0, out=self.part)
                self.allreduce(self.part, residual=h, out=h2)
                ops.w4a16_gemv_fused(h2, L.gate_up, R, c.rms_eps, out=self.gu_buf)
                ops.w4a16_gemv_fused(self.gu_buf, L.down, S, out=self.part)
                self.allreduce(self.part, residual=h2, out=h)
        if self.fused_tail:
            ops.lm_head(h, self.final_norm, self.lm_head, self.logits, c.rms_eps, self.lm_ws)
            ops.argmax_advance(self.lm_ws, c.vocab, self.tok, self.pos, self.out_tokens)
        else:
            torch.matmul(self._final_norm(h), self.lm_head.t(), out=self.logits)
            self.out_tokens.index_copy_(0, self.pos.long(), self.tok)
            torch.argmax(self.logits, dim=-1, out=self.tok)
            self.pos.add_(1)

    def _layers_deferred(self) -> torch.Tensor:
        """The decoder layers with the deferred K-split reduction (include/paro_abi.h v12; one GPU): o_proj and down_proj leave
        their fp32 partial sums (no in-launch hand-off, ~1 us each); the RMSNorm-prologue launch behind them -- gate_up, the NEXT
        layer's qkv -- adds them to the residual stream while it seeds its rotation and writes the new stream.  Same launches per
        layer, same bits as the ordinary route (the reducer's summation order and single rounding); one small completion launch
        in front of the final norm.  Returns the buffer that holds the final residual stream."""
        c = self.cfg
        R, S = nat.PROLOGUE_RMSNORM, nat.PROLOGUE_SILU_MUL
        cur, other = self.h, self.h2
        pend = None
        qkv_to = dict(parts_out=self.parts_q) if self.deferred_qkv else dict(out=self.qkv_buf)
        for L in self.layers:
            if self.fuse_qkv_attn:      # the attention rides in the qkv launch
                qkv_to = dict(parts_out=self.parts_q, attn_tail=dict(
                    kcache=L.kcache, vcache=L.vcache, pos=self.pos, rope=self.rope, n_heads=self.nh, n_kv_heads=self.nkv, head_dim=c.head_dim,
                    q_norm_w=L.q_norm, k_norm_w=L.k_norm, eps=c.rms_eps, split_out=self.attn_parts, workspace=self.attn_ws))
            if pend is None:
                ops.w4a16_gemv_fused(cur, L.qkv, R, c.rms_eps, **qkv_to)
            else:
                ops.w4a16_gemv_fused(cur, L.qkv, R, c.rms_eps, parts_in=pend, x_out=other.view(-1), **qkv_to)
                cur, other = other, cur
            if not self.fuse_qkv_attn:
                ops.attn_decode(self.parts_q if self.deferred_qkv else self.qkv_buf, L.kcache, L.vcache, self.pos, self.rope, self.nh, self.nkv,
                                c.head_dim, L.q_norm, L.k_norm, c.rms_eps, out=self.attn_buf, workspace=self.attn_ws, norm_dim=c.hidden,
                                norm_eps=c.rms_eps, split_out=self.attn_parts if self.split_attn else None)
            if self.split_attn:
                ops.w4a16_gemv_fused(None, L.o, 0, parts_out=self.parts_o, attn_in=self.attn_parts, attn_head_dim=c.head_dim, dtype=self.dtype)
            else:
                ops.w4a16_gemv_fused(self.attn_buf, L.o, 0, parts_out=self.parts_o)
            ops.w4a16_gemv_fused(cur, L.gate_up, R, c.rms_eps, out=self.gu_buf, parts_in=self.parts_o, x_out=other.view(-1))
            cur, other = other, cur
            ops.w4a16_gemv_fused(self.gu_buf, L.down, S, parts_out=self.parts_d)
            pend = self.parts_d
        ops.parts_finish(pend, cur.view(-1), out=other.view(-1))
        return other

    def _final_norm(self, h: torch.Tensor) -> torch.Tensor:
        x = h.float()
        return (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.cfg.rms_eps)).to(self.dtype) * self.final_norm

    def capture(self) -> None:
        """Capture :meth:`decode_step` in a HIP graph (one warm-up on a side stream first, as torch requires)."""
        tok0, pos0 = self.tok.clone(), self.pos.clone()
        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            self.decode_step()
        torch.cuda.current_stream(self.device).wait_stream(s)
        self.tok.copy_(tok0); self.pos.copy_(pos0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):     # a collective library's watchdog thread must not void the capture
            self.decode_step()
        self.tok.copy_(tok0); self.pos.copy_(pos0)
        self._graph = g

    # ------------------------------------------------------------------ prefill (MFMA GEMM path + torch attention)
    @torch.no_grad()
    def prefill(self, ids: torch.Tensor) -> torch.Tensor:
        """Run the prompt ``ids`` [T] through the model, fill the KV caches, return the logits of the last position and
        set (tok, pos) for the first decode step."""
        c, dt = self.cfg, self.dtype
        T = int(ids.numel())
        if T > c.max_positions:
            raise ValueError("prompt longer than max_positions")
        sharded = self.tp_world > 1
        if sharded and not (dist.is_available() and dist.is_initialized()):
            # no process group to all-reduce [T, hidden] over (the decode step's kernel-level collective is sized for one
            # row): take the prompt through the decode step, one teacher-forced token at a time
            ids_d = ids.to(self.device)
            for i in range(T):
                self.tok.copy_(ids_d[i:i + 1])
                self.pos.fill_(i)
                self.decode_step()
            self.out_tokens[:T] = ids_d
            return self.logits.clone()                 # tok / pos already hold the first generated token and T

        def reduce_rows(part):                         # RowParallelLinear's all-reduce of [T, hidden] (RCCL / the group's backend)
            if sharded:
                dist.all_reduce(part, group=self.tp_group)
            return part
        h = self.embed[ids.to(self.device)]                                           # [T, hidden]
        rs = lambda x: torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + c.rms_eps)
        half = c.head_dim // 2
        cos = self.rope[:T, :half].to(dt)[:, None, :]
        sin = self.rope[:T, half:].to(dt)[:, None, :]

        def rope(x):                                                                  # [T, H, hd]
            x1, x2 = x[..., :half], x[..., half:]
            return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], dim=-1)

        def headnorm(x, w):
            return x if w is None else ((x.float() * rs(x)).to(dt) * w)

        # The element-wise work between the linears: three launches per layer (csrc/prompt.hip, ABI v19 -- row RMS, qkv post-processing
        # into the attention's inputs and the decode caches, SiLU * up) instead of ~45 framework operators, which at 128 rows cost more
        # than the linears (Qwen3-4B time to first token 16.8 -> profiles/NOTES.md 6.10).  PARO_PROMPT_TORCH=1: the framework expressions.
        if c.head_dim <= 128 and os.environ.get("PARO_PROMPT_TORCH", "0") != "1":
            sdpa = torch.nn.functional.scaled_dot_product_attention
            for L in self.layers:
                q, k, v = ops.prompt_qkv_post(L.qkv.apply(h), ops.prompt_row_rms(h, c.rms_eps), self.rope, L.kcache, L.vcache, self.nh, self.nkv,
                                              c.head_dim, L.q_norm, L.k_norm, c.rms_eps)
                att = sdpa(q.transpose(0, 1)[None], k.transpose(0, 1)[None], v.transpose(0, 1)[None], is_causal=True,
                           enable_gqa=self.nh != self.nkv)[0].transpose(0, 1).reshape(T, -1)
                h = h + reduce_rows(L.o.apply(att.contiguous()))
                act = ops.prompt_silu_mul(L.gate_up.apply(h), ops.prompt_row_rms(h, c.rms_eps))
                h = h + reduce_rows(L.down.apply(act))
            logits = torch.matmul(self._final_norm(h[-1:]), self.lm_head.t())
            self.out_tokens[:T] = ids.to(self.device)
            self.tok.copy_(torch.argmax(logits, dim=-1))
            self.pos.fill_(T)
            return logits
        for L in self.layers:
            qkv = (L.qkv.apply(h).float() * rs(h)).to(dt)          # norm weight is folded into the channel scales
            q, k, v = qkv.split([self.nh * c.head_dim, self.nkv * c.head_dim, self.nkv * c.head_dim], dim=-1)   # this rank's heads
            q = rope(headnorm(q.view(T, self.nh, c.head_dim), L.q_norm))
            k = rope(headnorm(k.view(T, self.nkv, c.head_dim), L.k_norm))
            v = v.view(T, self.nkv, c.head_dim)
            L.kcache[:, :T] = k.transpose(0, 1)
            L.vcache[:, :, :T] = v.permute(1, 2, 0)
            att = torch.nn.functional.scaled_dot_product_attention(
                q.transpose(0, 1)[None], k.transpose(0, 1)[None], v.transpose(0, 1)[None], is_causal=True,
                enable_gqa=self.nh != self.nkv)[0].transpose(0, 1).reshape(T, -1)
            h = h + reduce_rows(L.o.apply(att.contiguous()))
            gu = (L.gate_up.apply(h).float() * rs(h)).to(dt)
            act = torch.nn.functional.silu(gu[:, :self.inter_l]) * gu[:, self.inter_l:]
            h = h + reduce_rows(L.down.apply(act.contiguous()))
        logits = torch.matmul(self._final_norm(h[-1:]), self.lm_head.t())
        self.out_tokens[:T] = ids.to(self.device)
        self.tok.copy_(torch.argmax(logits, dim=-1))
        self.pos.fill_(T)
        return logits

    # ------------------------------------------------------------------ generation + the reference's benchmark protocol
    @torch.no_grad()
    def generate(self, ids: torch.Tensor, max_new_tokens: int, use_graph: bool = True):
        """Greedy generation.  Returns (tokens [T + new], stats) with the reference's accounting
        (inference/base.py:62-77): ttft = first token latency, tps = (new - 1) decode tokens / (t_end - t_first)."""
        c = self.cfg
        T = int(ids.numel())
        n_new = min(max_new_tokens, c.max_positions - T)
        if T < 1 or n_new < 1:
            raise ValueError(f"nothing to generate: prompt of {T} tokens, max_new_tokens {max_new_tokens}, max_positions {c.max_positions}")
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        self.prefill(ids)
        torch.cuda.synchronize(self.device)
        t_first = time.perf_counter()
        if use_graph and self._graph is None:
            tok, pos = self.tok.clone(), self.pos.clone()
            self.capture()
            self.tok.copy_(tok); self.pos.copy_(pos)
            torch.cuda.synchronize(self.device)
            t_first = time.perf_counter()          # graph capture is a one-time cost, not decode time
        for _ in range(n_new - 1):
            if use_graph:
                self._graph.replay()
            else:
                self.decode_step()
        torch.cuda.synchronize(self.device)
        t_end = time.perf_counter()
        self.out_tokens.index_copy_(0, self.pos.long(), self.tok)
        toks = self.out_tokens[: T + n_new].clone()
        dec = max(n_new - 1, 1)
        return toks, {"prompt_tokens": T, "new_tokens": n_new, "ttft_s": t_first - t0,
                      "decode_tokens_per_s": dec / max(t_end - t_first, 1e-9), "ms_per_token": (t_end - t_first) * 1e3 / dec}
