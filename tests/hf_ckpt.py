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
