"""Decode harness for the Qwen3.5 family (BASELINE configs 3 and 5: Qwen3.5-4B / Qwen3.5-27B): the hybrid decoder of transformers'
``models/qwen3_5`` -- three gated-delta-net layers, then one gated full-attention layer with head_dim 256 -- at batch 1 on the fused
kernels, one HIP graph per token.

The reference decodes any HF architecture through ``generate()`` (``transformers/generator.py:37-67``) with its per-linear operator and
names this family's quantised modules in ``experiments/optimize/4bit.sh:17-20`` (``in_proj_a / in_proj_b`` stay dense).  Per layer here:

  gated delta net   in_proj_qkv | in_proj_z  ONE fused GEMV (two rotations, the input RMSNorm as prologue)
                    paro_gdn_prep   conv1d update + SiLU, the dense in_proj_a / in_proj_b, decay / beta          (csrc/gdn.hip)
                    paro_gdn_step   the recurrent delta rule on the 128 x 128 state of every value head + the gated RMSNorm
                    out_proj        fused GEMV + residual
  full attention    q(+gate) | k | v ONE fused GEMV (three rotations, RMSNorm prologue)
                    paro_attn_decode_gated   q / k norm (1 + w), partial rotary, KV append, attention, * sigmoid(gate)
                    o_proj          fused GEMV + residual
  MLP               gate | up fused GEMV (RMSNorm prologue), down with the SiLU * mul prologue + residual

Norm weights are ``(1 + w)`` in this family: folded into the channel scales with ``fold_norm_weight(plus_one=True)``.  The prompt is
taken through the decode step token by token (the state of the recurrence IS the prefill; a chunked prefill is the HF path's job).
"""
from __future__ import annotations

import ctypes
import json
import os
import time
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import _native as nat
from . import ops
from .linear import PackedParoWeights


@dataclass
class Qwen35Config:
    hidden: int
    inter: int
    n_heads: int
    n_kv_heads: int
    head_dim: int
    lin_k_heads: int
    lin_v_heads: int
    n_layers: int
    vocab: int
    layer_types: List[str]
    rms_eps: float = 1e-6
    rope_theta: float = 10000.0
    partial_rotary_factor: float = 0.25
    max_positions: int = 2048

    @classmethod
    def from_hf(cls, c: dict, max_positions: int = 2048) -> "Qwen35Config":
        c = c.get("text_config", c)
        rp = c.get("rope_parameters") or {}
        if rp.get("rope_type", "default") != "default":
            raise NotImplementedError(f"rope type {rp.get('rope_type')!r} is not supported by the Qwen3.5 decode harness")
        if c.get("linear_key_head_dim", 128) != 128 or c.get("linear_value_head_dim", 128) != 128 or c.get("linear_conv_kernel_dim", 4) != 4:
            raise NotImplementedError("the gated-delta-net kernels are built for key / value head dims of 128 and a conv kernel of 4")
        L = c["num_hidden_layers"]
        lt = c.get("layer_types") or ["linear_attention" if (i + 1) % c.get("full_attention_interval", 4) else "full_attention" for i in range(L)]
        return cls(c["hidden_size"], c["intermediate_size"], c["num_attention_heads"], c.get("num_key_value_heads", c["num_attention_heads"]),
                   c.get("head_dim", 256), c["linear_num_key_heads"], c["linear_num_value_heads"], L, c["vocab_size"], list(lt),
                   c.get("rms_norm_eps", 1e-6), float(rp.get("rope_theta", c.get("rope_theta", 10000.0))),
                   float(rp.get("partial_rotary_factor", c.get("partial_rotary_factor", 0.25))), max_positions)


class _Layer:
    __slots__ = ("full", "mix_in", "mix_out", "gate_up", "down", "q_norm", "k_norm", "kcache", "vcache", "w_ab", "conv_w", "A_log", "dt_bias",
                 "gdn_norm", "conv_state", "state")


class ParoQwen35DecoderLM:
    """Greedy batch-1 decoder over a Qwen3.5 ``*-PARO`` checkpoint (:meth:`from_checkpoint`) or synthetic weights (:meth:`random`)."""

    def __init__(self, cfg: Qwen35Config, device, dtype=torch.float16):
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        if cfg.head_dim != 256:
            raise NotImplementedError("the gated attention kernel is built for head_dim 256")
        self.rd = int(cfg.head_dim * cfg.partial_rotary_factor)
        inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, self.rd, 2, dtype=torch.float32, device=self.device) / self.rd))
        ang = torch.arange(cfg.max_positions, dtype=torch.float32, device=self.device)[:, None] * inv[None, :]
        self.rope = torch.cat([ang.cos(), ang.sin()], dim=-1).contiguous()          # [T, rotary_dim]: cos then sin
        self.layers: List[_Layer] = []
        self.embed = self.lm_head = self.final_norm = None
        self._graph = None

    # ------------------------------------------------------------------ construction
    def _finish_layer(self, L: _Layer):
        c, dev, dt = self.cfg, self.device, self.dtype
        if L.full:
            L.kcache = torch.zeros(c.n_kv_heads, c.max_positions, c.head_dim, dtype=dt, device=dev)
            L.vcache = torch.zeros(c.n_kv_heads, c.max_positions, c.head_dim, dtype=dt, device=dev)
        else:
            conv_dim = 2 * c.lin_k_heads * 128 + c.lin_v_heads * 128
            L.conv_state = torch.zeros(2, conv_dim, 4, dtype=dt, device=dev)       # double-buffered by the token's parity (paro_gdn_fused_step)
            L.state = torch.zeros(c.lin_v_heads, 128, 128, dtype=torch.float32, device=dev)
        self.layers.append(L)

    @classmethod
    def from_checkpoint(cls, path: str, device, max_positions: int = 2048, dtype: torch.dtype = torch.float16) -> "ParoQwen35DecoderLM":
        """An HF Qwen3.5 ``*-PARO`` directory: quantised linears under ``model.layers.N.{linear_attn,self_attn,mlp}.*`` in the reference's
        on-disk format (cli/convert.py:264-277), everything else dense."""
        from safetensors import safe_open
        with open(os.path.join(path, "config.json")) as f:
            cfg = Qwen35Config.from_hf(json.load(f), max_positions)
        self = cls(cfg, device, dtype)
        t: Dict[str, torch.Tensor] = {}
        for fn in sorted(os.listdir(path)):
            if fn.endswith(".safetensors"):
                with safe_open(os.path.join(path, fn), framework="pt") as f:
                    for k in f.keys():
                        t[k] = f.get_tensor(k)
        dev = self.device
        pre0 = "model.language_model." if any(k.startswith("model.language_model.") for k in t) else "model."

        def merged(prefix: str, names) -> PackedParoWeights:
            g = lambda n, s: t[f"{prefix}.{n}.{s}"].to(dev)
            sizes = [int(t[f"{prefix}.{n}.scales"].shape[1]) for n in names]
            return PackedParoWeights(torch.cat([g(n, "qweight") for n in names], dim=1), torch.cat([g(n, "qzeros") for n in names], dim=1),
                                     torch.cat([g(n, "scales") for n in names], dim=1), torch.stack([g(n, "theta") for n in names]),
                                     torch.stack([g(n, "pairs") for n in names]), torch.stack([g(n, "channel_scales").reshape(1, -1) for n in names]), sizes)

        for l in range(cfg.n_layers):
            p = f"{pre0}layers.{l}"
            L = _Layer()
            L.full = cfg.layer_types[l] == "full_attention"
            in_w = t[f"{p}.input_layernorm.weight"].to(dev)
            if L.full:
                L.mix_in = merged(f"{p}.self_attn", ["q_proj", "k_proj", "v_proj"])
                L.mix_out = merged(f"{p}.self_attn", ["o_proj"])
                L.q_norm = t[f"{p}.self_attn.q_norm.weight"].to(dev, dtype)
                L.k_norm = t[f"{p}.self_attn.k_norm.weight"].to(dev, dtype)
            else:
                a = f"{p}.linear_attn"
                L.mix_in = merged(a, ["in_proj_qkv", "in_proj_z"])
                L.mix_out = merged(a, ["out_proj"])
                w_ab = torch.cat([t[f"{a}.in_proj_a.weight"], t[f"{a}.in_proj_b.weight"]]).to(dev).float()
                L.w_ab = (w_ab * (1.0 + in_w.float())[None, :]).contiguous()                  # the input norm's (1 + w) folded in
                L.conv_w = t[f"{a}.conv1d.weight"].to(dev).float().reshape(-1, 4).contiguous()
                L.A_log, L.dt_bias = t[f"{a}.A_log"].to(dev).float().contiguous(), t[f"{a}.dt_bias"].to(dev).float().contiguous()
                L.gdn_norm = t[f"{a}.norm.weight"].to(dev, dtype).contiguous()
            L.mix_in.fold_norm_weight(in_w, plus_one=True)
            L.gate_up = merged(f"{p}.mlp", ["gate_proj", "up_proj"])
            L.down = merged(f"{p}.mlp", ["down_proj"])
            L.gate_up.fold_norm_weight(t[f"{p}.post_attention_layernorm.weight"].to(dev), plus_one=True)
            self._finish_layer(L)
        self.embed = t[f"{pre0}embed_tokens.weight"].to(dev, dtype)
        self.lm_head = (t["lm_head.weight"] if "lm_head.weight" in t else t[f"{pre0}embed_tokens.weight"]).to(dev, dtype).contiguous()
        self.final_norm = (1.0 + t[f"{pre0}norm.weight"].to(dev).float()).to(dtype)            # (1 + w) of the final norm
        self._static()
        return self

    @classmethod
    def random(cls, name_or_cfg, device, n_layers: Optional[int] = None, max_positions: int = 1024, seed: int = 0,
               vocab: Optional[int] = None, dtype: torch.dtype = torch.float16) -> "ParoQwen35DecoderLM":
        """Synthetic weights of a named hybrid architecture (``bench.HYBRID``: "qwen3.5-9b", ...) for benchmarks."""
        import bench
        if isinstance(name_or_cfg, str):
            h, inter, nh, nkv, hd, lk, lv, L, iv = bench.HYBRID[name_or_cfg]
            L = n_layers or L
            cfg = Qwen35Config(h, inter, nh, nkv, hd, lk, lv, L, vocab or 151936, ["linear_attention" if (i + 1) % iv else "full_attention" for i in range(L)],
                               max_positions=max_positions)
        else:
            cfg = name_or_cfg
        self = cls(cfg, device, dtype)
        dev = self.device
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        rnd = lambda *s: torch.randn(*s, device=dev, generator=gen)
        c = cfg
        for l in range(c.n_layers):
            L = _Layer()
            L.full = c.layer_types[l] == "full_attention"
            in_w = 0.05 * rnd(c.hidden)
            if L.full:
                L.mix_in = bench.synth_packed(c.hidden, [2 * c.n_heads * c.head_dim, c.n_kv_heads * c.head_dim, c.n_kv_heads * c.head_dim], dev, gen)
                L.mix_out = bench.synth_packed(c.n_heads * c.head_dim, [c.hidden], dev, gen)
                L.q_norm, L.k_norm = (0.05 * rnd(c.head_dim)).to(dtype), (0.05 * rnd(c.head_dim)).to(dtype)
            else:
                kd, vd = c.lin_k_heads * 128, c.lin_v_heads * 128
                L.mix_in = bench.synth_packed(c.hidden, [2 * kd + vd, vd], dev, gen)
                L.mix_out = bench.synth_packed(vd, [c.hidden], dev, gen)
                L.w_ab = (rnd(2 * c.lin_v_heads, c.hidden) * 0.05 * (1.0 + in_w)[None, :]).contiguous()
                L.conv_w = (rnd(2 * kd + vd, 4) * 0.3).contiguous()
                L.A_log = torch.log(torch.rand(c.lin_v_heads, device=dev, generator=gen) * 7.0 + 1.0)
                L.dt_bias = rnd(c.lin_v_heads) * 0.1
                L.gdn_norm = (1.0 + 0.05 * rnd(128)).to(dtype)
            L.mix_in.fold_norm_weight(in_w, plus_one=True)
            L.gate_up = bench.synth_packed(c.hidden, [c.inter, c.inter], dev, gen)
            L.down = bench.synth_packed(c.inter, [c.hidden], dev, gen)
            L.gate_up.fold_norm_weight(0.05 * rnd(c.hidden), plus_one=True)
            self._finish_layer(L)
        self.embed = (rnd(c.vocab, c.hidden) * 0.5).to(dtype)
        self.lm_head = (rnd(c.vocab, c.hidden) * (c.hidden ** -0.5)).to(dtype).contiguous()
        self.final_norm = (1.0 + 0.05 * rnd(c.hidden)).to(dtype)
        self._static()
        return self

    def _static(self):
        c, dev, dt = self.cfg, self.device, self.dtype
        self.tok = torch.zeros(1, dtype=torch.long, device=dev)
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)
        self.h = torch.zeros(1, c.hidden, dtype=dt, device=dev)
        self.h2 = torch.zeros(1, c.hidden, dtype=dt, device=dev)
        kd, vd = c.lin_k_heads * 128, c.lin_v_heads * 128
        self.conv_dim = 2 * kd + vd
        self.qkvz = torch.zeros(1, self.conv_dim + vd, dtype=dt, device=dev)
        self.qkv = torch.zeros(1, (2 * c.n_heads + 2 * c.n_kv_heads) * c.head_dim, dtype=dt, device=dev)
        self.mix = torch.zeros(1, max(vd, c.n_heads * c.head_dim), dtype=dt, device=dev)
        self.gu = torch.zeros(1, 2 * c.inter, dtype=dt, device=dev)
        self.logits = torch.zeros(1, c.vocab, dtype=dt, device=dev)
        self.out_tokens = torch.zeros(c.max_positions, dtype=torch.long, device=dev)
        self.lm_ws = ops.lm_head_workspace(dev, c.vocab)
        self.gdn_ws = torch.zeros(int(nat.load().paro_gdn_workspace_bytes(c.lin_v_heads)), dtype=torch.uint8, device=dev)   # zero-filled once
        self.fused_tail = c.hidden % 512 == 0 and c.hidden <= 4096
        self.bytes_per_token = sum(pk.nbytes() for L in self.layers for pk in (L.mix_in, L.mix_out, L.gate_up, L.down))
        # deferred K-split reduction (include/paro_abi.h v12; decoder.ParoDecoderLM._layers_deferred): out_proj / o_proj and down_proj leave
        # their fp32 partial sums, the RMSNorm-prologue launch behind them (gate_up, the next block's in_proj) completes the residual stream
        from .decoder import deferred_route_pays
        n_o = min((ops.gemv_parts_count(L.mix_out, dt) for L in self.layers), default=0)
        n_d = min((ops.gemv_parts_count(L.down, dt) for L in self.layers), default=0)
        self.deferred = n_o >= 2 and n_d >= 2 and deferred_route_pays(c.hidden)
        if self.deferred:
            self.parts_o = torch.zeros(c.hidden, nat.PARO_MAX_PARTIALS, dtype=torch.float32, device=dev)
            self.parts_d = torch.zeros(c.hidden, nat.PARO_MAX_PARTIALS, dtype=torch.float32, device=dev)

    # ------------------------------------------------------------------ one decode token (capturable)
    def decode_step(self) -> None:
        c, lib = self.cfg, nat.load()
        R, S = nat.PROLOGUE_RMSNORM, nat.PROLOGUE_SILU_MUL
        dtc, st = nat.dtype_code(self.dtype), nat.current_stream_ptr(self.device)
        torch.index_select(self.embed, 0, self.tok, out=self.h)
        h, h2 = self.h, self.h2
        vd = c.lin_v_heads * 128
        pend = None                      # deferred route: the previous block's down_proj partial sums, not yet in the stream
        with torch.cuda.device(self.device):
            for L in self.layers:
                # (deferred: the mixer's input projection completes h = h_prev + sum(pend) while it seeds its rotation and stores it)
                dk = dict(parts_in=pend, x_out=h2.view(-1)) if pend is not None else {}
                if pend is not None:
                    h, h2 = h2, h
                if L.full:
                    ops.w4a16_gemv_fused(h2 if pend is not None else h, L.mix_in, R, c.rms_eps, out=self.qkv, **dk)
                    mix = self.mix[:, : c.n_heads * c.head_dim]
                    nat.check(lib.paro_attn_decode_gated(self.qkv.data_ptr(), L.kcache.data_ptr(), L.vcache.data_ptr(), mix.data_ptr(), self.pos.data_ptr(),
                                                         self.rope.data_ptr(), L.q_norm.data_ptr(), L.k_norm.data_ptr(), 1, c.rms_eps, c.head_dim ** -0.5,
                                                         c.n_heads, c.n_kv_heads, c.head_dim, self.rd, c.max_positions, dtc, st))
                else:
                    ops.w4a16_gemv_fused(h2 if pend is not None else h, L.mix_in, R, c.rms_eps, out=self.qkvz, **dk)
                    mix = self.mix[:, :vd]
                    # conv1d update + the dense a / b rows + the recurrence + the gated norm: one launch (gdn.hip: gdn_fused_kernel)
                    nat.check(lib.paro_gdn_fused_step(self.qkvz.data_ptr(), h.data_ptr(), L.w_ab.data_ptr(), c.rms_eps, L.conv_state.data_ptr(),
                                                      L.conv_w.data_ptr(), L.A_log.data_ptr(), L.dt_bias.data_ptr(), L.state.data_ptr(),
                                                      L.gdn_norm.data_ptr(), c.rms_eps, mix.data_ptr(), self.pos.data_ptr(), c.hidden, self.conv_dim,
                                                      c.lin_k_heads, c.lin_v_heads, dtc, self.gdn_ws.data_ptr(), st))
                if self.deferred:
                    ops.w4a16_gemv_fused(mix, L.mix_out, 0, parts_out=self.parts_o)
                    ops.w4a16_gemv_fused(h, L.gate_up, R, c.rms_eps, out=self.gu, parts_in=self.parts_o, x_out=h2.view(-1))   # h2 = h + mixer(x)
                    h, h2 = h2, h
                    ops.w4a16_gemv_fused(self.gu, L.down, S, parts_out=self.parts_d)
                    pend = self.parts_d
                else:
                    ops.w4a16_gemv_fused(mix, L.mix_out, 0, residual=h, out=h2)                     # h2 = h + mixer(x)
                    ops.w4a16_gemv_fused(h2, L.gate_up, R, c.rms_eps, out=self.gu)
                    ops.w4a16_gemv_fused(self.gu, L.down, S, residual=h2, out=h)                    # h = h2 + mlp(...)
            if pend is not None:
                ops.parts_finish(pend, h.view(-1), out=h2.view(-1))
                h = h2
        if self.fused_tail:
            ops.lm_head(h, self.final_norm, self.lm_head, self.logits, c.rms_eps, self.lm_ws)
            ops.argmax_advance(self.lm_ws, c.vocab, self.tok, self.pos, self.out_tokens)
        else:
            x = h.float()
            xn = (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + c.rms_eps) * self.final_norm.float()).to(self.dtype)
            torch.matmul(xn, self.lm_head.t(), out=self.logits)
            self.out_tokens.index_copy_(0, self.pos.long(), self.tok)
            torch.argmax(self.logits, dim=-1, out=self.tok)
            self.pos.add_(1)

    def reset(self) -> None:
        """Forget the sequence: recurrent and convolution states to zero, position 0 (the KV cache is overwritten as it is used)."""
        for L in self.layers:
            if not L.full:
                L.conv_state.zero_()
                L.state.zero_()
        self.pos.zero_()

    def capture(self) -> None:
        """One HIP graph of :meth:`decode_step`.  The warm-up step advances the recurrent state, so the states are saved around it."""
        saved = [(L.conv_state.clone(), L.state.clone()) for L in self.layers if not L.full]
        tok0, pos0 = self.tok.clone(), self.pos.clone()

        def restore():
            for (cs, s), L in zip(saved, [L for L in self.layers if not L.full]):
                L.conv_state.copy_(cs)
                L.state.copy_(s)
            self.tok.copy_(tok0)
            self.pos.copy_(pos0)
        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            self.decode_step()
        torch.cuda.current_stream(self.device).wait_stream(s)
        restore()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self.decode_step()
        restore()
        self._graph = g

    @torch.no_grad()
    def prefill(self, ids: torch.Tensor, use_graph: bool = True, sequential: bool = False) -> torch.Tensor:
        """The prompt pass; returns the logits of the last position and leaves (tok, pos) at the first generated token.  Default: every
        linear over all T rows at once (the library's prefill GEMM path), the gated delta net's recurrence as ONE launch per layer with the
        state in registers (``paro_gdn_sequence``), convolution / norms / full attention as row-parallel torch operations
        (:meth:`_prefill_rows`).  ``sequential=True``: the decode step, one teacher-forced token at a time (what round 4 started with;
        kept as the cross-check of the row form)."""
        T = int(ids.numel())
        if T > self.cfg.max_positions:
            raise ValueError("prompt longer than max_positions")
        if not sequential and T >= 2 and os.environ.get("PARO_QWEN35_SEQUENTIAL_PREFILL", "0") != "1":
            return self._prefill_rows(ids)
        self.reset()
        ids_d = ids.to(self.device)
        if use_graph and self._graph is None:
            self.capture()
        for i in range(T):
            self.tok.copy_(ids_d[i:i + 1])
            if use_graph:
                self._graph.replay()
            else:
                self.decode_step()
        self.out_tokens[:T] = ids_d
        return self.logits.clone()

    @torch.no_grad()
    def _prefill_rows(self, ids: torch.Tensor) -> torch.Tensor:
        """All T prompt rows at once.  Rounding points follow the decode kernels (csrc/gdn.hip), which follow HF's modelling code
        (models/qwen3_5: Qwen3_5GatedDeltaNet.forward, Qwen3_5Attention.forward, Qwen3_5RMSNormGated)."""
        c, dt, dev, lib = self.cfg, self.dtype, self.device, nat.load()
        T = int(ids.numel())
        self.reset()
        ids_d = ids.to(dev)
        h = self.embed[ids_d]                                                          # [T, hidden]
        rs = lambda x: torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + c.rms_eps)
        kd, vd, nk, nv = c.lin_k_heads * 128, c.lin_v_heads * 128, c.lin_k_heads, c.lin_v_heads
        HD, nh, nkv, rd = c.head_dim, c.n_heads, c.n_kv_heads, self.rd
        half = rd // 2
        cos, sin = self.rope[:T, :half][:, None, :], self.rope[:T, half:][:, None, :]   # fp32 [T, 1, half]
        dtc, st = nat.dtype_code(dt), nat.current_stream_ptr(dev)

        def rope_partial(x):                                                          # x fp32 [T, H, HD] (already normalised, rounded to dt)
            x1, x2 = x[..., :half], x[..., half:rd]
            r = torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], dim=-1).to(dt).float()
            return torch.cat([r, x[..., rd:]], dim=-1)

        with torch.cuda.device(dev):
            for L in self.layers:
                r0 = rs(h)
                y = L.mix_in.apply((h.float() * r0).to(dt))                            # normalised FIRST (an un-normalised fp16 projection of a residual-stream outlier can overflow before the scalar shrinks it); the norm's weight is folded into the channel scales
                if L.full:
                    qg = y[:, : 2 * nh * HD].view(T, nh, 2, HD)
                    q, gate = qg[:, :, 0].float(), qg[:, :, 1].float()
                    k = y[:, 2 * nh * HD: 2 * nh * HD + nkv * HD].view(T, nkv, HD).float()
                    v = y[:, 2 * nh * HD + nkv * HD:].view(T, nkv, HD)
                    q = (q * rs(q) * (1.0 + L.q_norm.float())).to(dt).float()         # (1 + w) RMSNorm per head, one rounding
                    k = (k * rs(k) * (1.0 + L.k_norm.float())).to(dt).float()
                    q, k = rope_partial(q), rope_partial(k)
                    L.kcache[:, :T] = k.to(dt).transpose(0, 1)
                    L.vcache[:, :T] = v.transpose(0, 1)
                    rep = nh // nkv
                    kk = k.repeat_interleave(rep, dim=1).transpose(0, 1)               # [nh, T, HD]
                    vv = v.float().repeat_interleave(rep, dim=1).transpose(0, 1)
                    sc = torch.matmul(q.transpose(0, 1) * (HD ** -0.5), kk.transpose(1, 2))          # [nh, T, T] fp32
                    sc = sc.masked_fill(torch.ones(T, T, dtype=torch.bool, device=dev).triu(1), float("-inf"))
                    att = torch.matmul(torch.softmax(sc, dim=-1), vv).transpose(0, 1)                  # [T, nh, HD]
                    mix = (att * torch.sigmoid(gate)).to(dt).reshape(T, nh * HD)
                else:
                    conv_dim = 2 * kd + vd
                    xin = y[:, :conv_dim]                                              # [T, conv_dim] the convolution's inputs, z behind them
                    z = y[:, conv_dim:].float().view(T, nv, 128)
                    pad = torch.cat([torch.zeros(3, conv_dim, dtype=dt, device=dev), xin], dim=0).float()
                    w = L.conv_w                                                       # [conv_dim, 4]: taps for inputs t-3 .. t
                    cv = pad[0:T] * w[:, 0] + pad[1:T + 1] * w[:, 1] + pad[2:T + 2] * w[:, 2] + pad[3:T + 3] * w[:, 3]
                    conv_out = (cv * torch.sigmoid(cv)).to(dt).contiguous()
                    last3 = pad[T:T + 3].to(dt)                                        # the state the next token's update reads
                    L.conv_state[T & 1][:, 1:4] = last3.transpose(0, 1)             # (the buffer the token at position T reads)
                    ab = torch.matmul(h.float() * r0, L.w_ab.t())                      # [T, 2 nv]  ((1 + w) folded into w_ab)
                    tt = ab[:, :nv] + L.dt_bias
                    g = torch.exp(-torch.exp(L.A_log) * torch.nn.functional.softplus(tt, threshold=20.0))
                    g_beta = torch.cat([g, torch.sigmoid(ab[:, nv:])], dim=-1).contiguous()
                    raw = torch.empty(T, vd, dtype=torch.float32, device=dev)
                    nat.check(lib.paro_gdn_sequence(conv_out.data_ptr(), g_beta.data_ptr(), L.state.data_ptr(), raw.data_ptr(), T, nk, nv, dtc, st))
                    o = raw.view(T, nv, 128).to(dt).float()                            # Qwen3_5RMSNormGated with gdn_step_kernel's rounding points
                    n = (o * rs(o)).to(dt).float()
                    wn = (L.gdn_norm.float() * n).to(dt).float()
                    mix = (wn * (z * torch.sigmoid(z))).to(dt).reshape(T, vd)
                h = h + L.mix_out.apply(mix.contiguous())
                gu = L.gate_up.apply((h.float() * rs(h)).to(dt))
                act = (torch.nn.functional.silu(gu[:, : c.inter].float()) * gu[:, c.inter:].float()).to(dt)
                h = h + L.down.apply(act.contiguous())
            x = h[-1:].float()
            xn = (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + c.rms_eps) * self.final_norm.float()).to(dt)
            logits = torch.matmul(xn, self.lm_head.t())
        self.logits.copy_(logits)
        self.out_tokens[:T] = ids_d
        self.tok.copy_(torch.argmax(logits, dim=-1))
        self.pos.fill_(T)
        return logits.clone()

    @torch.no_grad()
    def generate(self, ids: torch.Tensor, max_new_tokens: int, use_graph: bool = True):
        """Greedy generation with the reference's accounting (inference/base.py:62-77): ttft = first-token latency,
        tps = (new - 1) decode tokens / (t_end - t_first)."""
        c = self.cfg
        T = int(ids.numel())
        n_new = min(max_new_tokens, c.max_positions - T)
        if T < 1 or n_new < 1:
            raise ValueError(f"nothing to generate: prompt of {T} tokens, max_new_tokens {max_new_tokens}, max_positions {c.max_positions}")
        if use_graph and self._graph is None:
            self.capture()                      # one-time cost, outside the clock like the reference's warm-up
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        self.prefill(ids, use_graph)
        torch.cuda.synchronize(self.device)
        t_first = time.perf_counter()
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
