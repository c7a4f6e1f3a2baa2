"""Test helper: write a tiny synthetic Llama-style ``*-PARO`` checkpoint (safetensors + config.json with a
``quantization_config`` block) whose quantised linears come from the oracle's packer -- the on-disk format of
cli/convert.py:264-277 that every back-end of the reference consumes."""
import json
import os

import numpy as np
import torch

from oracle import paro_oracle as po

LINEARS = (("self_attn.q_proj", "h", "q"), ("self_attn.k_proj", "h", "kv"), ("self_attn.v_proj", "h", "kv"),
           ("self_attn.o_proj", "q", "h"), ("mlp.gate_proj", "h", "i"), ("mlp.up_proj", "h", "i"), ("mlp.down_proj", "i", "h"))


def write_tiny_paro_llama(path: str, hidden=256, inter=512, heads=4, kv_heads=2, layers=2, vocab=128, seed=0,
                          model_type="llama", head_dim=None, group_size=128):
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    hd = head_dim or hidden // heads
    dims = {"h": hidden, "q": heads * hd, "kv": kv_heads * hd, "i": inter}
    rng = np.random.default_rng(seed)
    f16 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.float16)
    tensors = {"model.embed_tokens.weight": f16(rng.standard_normal((vocab, hidden)) * 0.5),
               "model.norm.weight": f16(1.0 + 0.1 * rng.standard_normal(hidden)),
               "lm_head.weight": f16(rng.standard_normal((vocab, hidden)) * 0.05)}
    oracle_layers = {}
    for l in range(layers):
        pre = f"model.layers.{l}."
        tensors[pre + "input_layernorm.weight"] = f16(1.0 + 0.1 * rng.standard_normal(hidden))
        tensors[pre + "post_attention_layernorm.weight"] = f16(1.0 + 0.1 * rng.standard_normal(hidden))
        if model_type == "qwen3":
            tensors[pre + "self_attn.q_norm.weight"] = f16(1.0 + 0.1 * rng.standard_normal(hd))
            tensors[pre + "self_attn.k_norm.weight"] = f16(1.0 + 0.1 * rng.standard_normal(hd))
        for name, kin, kout in LINEARS:
            L = po.make_layer(seed * 1000 + l * 10 + len(oracle_layers), dims[kin], [dims[kout]], group_size=group_size)
            oracle_layers[pre + name] = L
            tensors[pre + name + ".qweight"] = torch.from_numpy(L["qweight"])
            tensors[pre + name + ".qzeros"] = torch.from_numpy(L["qzeros"])
            tensors[pre + name + ".scales"] = torch.from_numpy(L["scales"])
            tensors[pre + name + ".theta"] = torch.from_numpy(L["theta"][0])
            tensors[pre + name + ".pairs"] = torch.from_numpy(L["pairs"][0])
            tensors[pre + name + ".channel_scales"] = torch.from_numpy(L["channel_scales"][0]).reshape(1, -1)
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(path, "model.safetensors"))
    cfg = {"architectures": ["Qwen3ForCausalLM" if model_type == "qwen3" else "LlamaForCausalLM"], "model_type": model_type, "hidden_size": hidden, "intermediate_size": inter,
           "num_hidden_layers": layers, "num_attention_heads": heads, "num_key_value_heads": kv_heads, "head_dim": hd,
           "vocab_size": vocab, "max_position_embeddings": 128, "rms_norm_eps": 1e-6, "rope_theta": 10000.0,
           "hidden_act": "silu", "tie_word_embeddings": False, "attention_bias": False, "mlp_bias": False,
           "torch_dtype": "float16", "bos_token_id": 1, "eos_token_id": 2,
           "quantization_config": {"quant_method": "paroquant", "bits": 4, "group_size": group_size, "krot": 8}}
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f, indent=1)
    return oracle_layers


# Qwen3.5 (hybrid gated-delta-net / full-attention decoder; transformers' models/qwen3_5): the quantised linear set of a
# *-PARO checkpoint of that family -- the reference's optimiser skips `linear_attn.in_proj_a` / `in_proj_b`
# (experiments/optimize/4bit.sh:17-20), which therefore stay dense `.weight` tensors next to the quantised ones
QWEN35_LINEAR_ATTN = (("linear_attn.in_proj_qkv", "h", "lqkv"), ("linear_attn.in_proj_z", "h", "lv"), ("linear_attn.out_proj", "lv", "h"))
QWEN35_FULL_ATTN = (("self_attn.q_proj", "h", "q2"), ("self_attn.k_proj", "h", "kv"), ("self_attn.v_proj", "h", "kv"), ("self_attn.o_proj", "q", "h"))
QWEN35_MLP = (("mlp.gate_proj", "h", "i"), ("mlp.up_proj", "h", "i"), ("mlp.down_proj", "i", "h"))


def qwen35_tiny_config(hidden=256, inter=512, heads=2, kv_heads=1, head_dim=256, layers=4, vocab=128, lk_heads=2, lv_heads=4):
    return {"architectures": ["Qwen3_5ForCausalLM"], "model_type": "qwen3_5_text", "hidden_size": hidden, "intermediate_size": inter,
            "num_hidden_layers": layers, "num_attention_heads": heads, "num_key_value_heads": kv_heads, "head_dim": head_dim,
            "linear_key_head_dim": 128, "linear_value_head_dim": 128, "linear_num_key_heads": lk_heads, "linear_num_value_heads": lv_heads,
            "linear_conv_kernel_dim": 4, "vocab_size": vocab, "max_position_embeddings": 128, "rms_norm_eps": 1e-6, "hidden_act": "silu",
            "tie_word_embeddings": False, "attention_bias": False, "torch_dtype": "float16", "bos_token_id": 1, "eos_token_id": 2}


def write_tiny_paro_qwen35(path: str, seed=0, **dims_kw):
    """Tiny synthetic Qwen3.5 `*-PARO` checkpoint: every 4th layer full attention (gated q_proj: 2 x heads x head_dim
    outputs, head_dim 256), the others gated delta net; quantised linears from the oracle's packer, `in_proj_a / in_proj_b`,
    conv1d, norms, `A_log`, `dt_bias` dense.  Returns (oracle layers by module path, dense state dict of the SAME
    function: every quantised linear as the dense matrix the oracle's float64 linear applies)."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    cfg = qwen35_tiny_config(**dims_kw)
    h, hd = cfg["hidden_size"], cfg["head_dim"]
    kd, vd = cfg["linear_num_key_heads"] * 128, cfg["linear_num_value_heads"] * 128
    dims = {"h": h, "q": cfg["num_attention_heads"] * hd, "q2": 2 * cfg["num_attention_heads"] * hd, "kv": cfg["num_key_value_heads"] * hd,
            "i": cfg["intermediate_size"], "lqkv": 2 * kd + vd, "lv": vd}
    rng = np.random.default_rng(seed)
    f16 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.float16)
    vocab, L = cfg["vocab_size"], cfg["num_hidden_layers"]
    dense = {"model.embed_tokens.weight": f16(rng.standard_normal((vocab, h)) * 0.5), "model.norm.weight": f16(0.1 * rng.standard_normal(h)),
             "lm_head.weight": f16(rng.standard_normal((vocab, h)) * 0.05)}
    tensors = dict(dense)
    oracle_layers = {}
    for l in range(L):
        pre = f"model.layers.{l}."
        full = (l + 1) % 4 == 0
        extra = {pre + "input_layernorm.weight": f16(0.1 * rng.standard_normal(h)), pre + "post_attention_layernorm.weight": f16(0.1 * rng.standard_normal(h))}
        if full:
            extra[pre + "self_attn.q_norm.weight"] = f16(0.1 * rng.standard_normal(hd))
            extra[pre + "self_attn.k_norm.weight"] = f16(0.1 * rng.standard_normal(hd))
        else:
            nv = cfg["linear_num_value_heads"]
            extra[pre + "linear_attn.in_proj_a.weight"] = f16(rng.standard_normal((nv, h)) * 0.05)
            extra[pre + "linear_attn.in_proj_b.weight"] = f16(rng.standard_normal((nv, h)) * 0.05)
            extra[pre + "linear_attn.conv1d.weight"] = f16(rng.standard_normal((2 * kd + vd, 1, 4)) * 0.3)
            extra[pre + "linear_attn.norm.weight"] = f16(1.0 + 0.1 * rng.standard_normal(128))
            extra[pre + "linear_attn.A_log"] = f16(np.log(rng.uniform(1.0, 8.0, nv)))
            extra[pre + "linear_attn.dt_bias"] = f16(rng.standard_normal(nv) * 0.1)
        tensors.update(extra)
        dense.update(extra)
        for name, kin, kout in (QWEN35_FULL_ATTN if full else QWEN35_LINEAR_ATTN) + QWEN35_MLP:
            K, N = dims[kin], dims[kout]
            Lq = po.make_layer(seed * 1000 + 7 * len(oracle_layers) + 1, K, [N])
            # keep the residual stream tame: scale the synthetic quantisation scales to unit gain
            Lq["scales"] = (Lq["scales"].astype(np.float32) * (1.0 / (0.011 * 6.5 * np.sqrt(K) * 1.3))).astype(np.float16)
            oracle_layers[pre + name] = Lq
            for key in ("qweight", "qzeros", "scales"):
                tensors[pre + name + "." + key] = torch.from_numpy(Lq[key])
            tensors[pre + name + ".theta"] = torch.from_numpy(Lq["theta"][0])
            tensors[pre + name + ".pairs"] = torch.from_numpy(Lq["pairs"][0])
            tensors[pre + name + ".channel_scales"] = torch.from_numpy(Lq["channel_scales"][0]).reshape(1, -1)
            w_eff = po.paro_linear(np.eye(K), Lq["qweight"], Lq["qzeros"], Lq["scales"], Lq["theta"][0], Lq["pairs"][0], Lq["channel_scales"][0],
                                   None, 128, ideal=True)          # [K, N]: row k = the linear's response to e_k
            dense[pre + name + ".weight"] = torch.from_numpy(np.ascontiguousarray(w_eff.T)).to(torch.float32)
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(path, "model.safetensors"))
    cfg_q = dict(cfg)
    cfg_q["quantization_config"] = {"quant_method": "paroquant", "bits": 4, "group_size": 128, "krot": 8}
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg_q, f, indent=1)
    return oracle_layers, dense, cfg
