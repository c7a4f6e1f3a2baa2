"""Qwen3.5 (BASELINE configs 3 and 5 name Qwen3.5-4B / Qwen3.5-27B): the hybrid gated-delta-net / full-attention family
through the product path.  Dimensions of the linear set are read from the container's transformers (models/qwen3_5):
gated q_proj hidden -> 2 x heads x head_dim with head_dim 256, linear_attn.in_proj_qkv hidden -> 2 key_dim + value_dim,
in_proj_z hidden -> value_dim, out_proj value_dim -> hidden; in_proj_a / in_proj_b stay dense (the reference's optimiser
skips them, experiments/optimize/4bit.sh:17-20)."""
import numpy as np
import pytest
import torch

from oracle import paro_oracle as po

pytestmark = pytest.mark.gpu

TIGHT_F16 = 3e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    import paroquant_amd  # noqa: F401
    from paroquant_amd import _native
    _native.load()
    return torch.device("cuda:0")


def _np(t):
    return t.detach().float().cpu().numpy().astype(np.float64)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_qwen35_from_pretrained_mixed_dense_and_quantised(dev, tmp_path):
    """A synthetic 4-layer Qwen3.5 PARO checkpoint (3 gated-delta-net layers + 1 full-attention layer, quantised linears
    from the oracle's packer, in_proj_a / in_proj_b / conv1d dense) loads through AutoModelForCausalLM.from_pretrained ->
    ParoQuantHfQuantizer (transformers/quantizer.py:88-115: swap exactly the modules that own a `.qweight`):
      * every swapped linear's output matches the float64 oracle on the activations it actually received;
      * the logits match the DENSE fp32 model that carries, for every quantised linear, the matrix the oracle's linear
        applies (rotation folded in) -- the architecture runs HF's own modelling code on both sides."""
    import paroquant_amd.hf_quantizer  # noqa: F401
    from paroquant_amd import RotateQuantizedLinear
    from tests.hf_ckpt import write_tiny_paro_qwen35
    from transformers import AutoModelForCausalLM
    from transformers.models.qwen3_5.configuration_qwen3_5 import Qwen3_5TextConfig
    from transformers.models.qwen3_5.modeling_qwen3_5 import Qwen3_5ForCausalLM
    layers, dense, cfg = write_tiny_paro_qwen35(str(tmp_path))
    model = AutoModelForCausalLM.from_pretrained(str(tmp_path), dtype=torch.float16, device_map={"": "cuda:0"})
    swapped = {k: m for k, m in model.named_modules() if isinstance(m, RotateQuantizedLinear)}
    assert set(swapped) == set(layers)
    plain = {k for k, m in model.named_modules() if type(m) is torch.nn.Linear}
    assert {"model.layers.0.linear_attn.in_proj_a", "model.layers.0.linear_attn.in_proj_b", "lm_head"} <= plain
    assert swapped["model.layers.3.self_attn.q_proj"].out_features == 2 * cfg["num_attention_heads"] * cfg["head_dim"]
    seen = {}
    hooks = [m.register_forward_hook(lambda mod, inp, out, k=k: seen.__setitem__(k, (inp[0].detach(), out.detach()))) for k, m in swapped.items()]
    ids = torch.randint(0, cfg["vocab_size"], (1, 12), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    with torch.no_grad():
        logits = model(input_ids=ids).logits
    for h in hooks:
        h.remove()
    assert set(seen) == set(layers) and torch.isfinite(logits.float()).all()
    for k, (x, y) in seen.items():
        L = layers[k]
        K = x.shape[-1]
        ref = po.paro_linear(_np(x.reshape(-1, K)), L["qweight"], L["qzeros"], L["scales"], L["theta"][0], L["pairs"][0], L["channel_scales"][0],
                             None, 128, ideal=True)
        assert po.rel_err(_np(y.reshape(-1, y.shape[-1])), ref) < TIGHT_F16, k
    c = Qwen3_5TextConfig(**{k: v for k, v in cfg.items() if k not in ("architectures", "model_type", "torch_dtype")})
    ref_model = Qwen3_5ForCausalLM(c).float().to(dev)
    missing, unexpected = ref_model.load_state_dict({k: v.float() for k, v in dense.items()}, strict=False)
    assert not missing and not unexpected
    with torch.no_grad():
        ref_logits = ref_model(input_ids=ids).logits
    assert po.rel_err(_np(logits), _np(ref_logits)) < 3e-2      # fp16 model code vs fp32 model code around identical linears
    assert (logits[0, -1].float().argmax() == ref_logits[0, -1].argmax()) or po.rel_err(_np(logits), _np(ref_logits)) < 1e-2
    # single-token steps (the GEMV path) after a prefill
    with torch.no_grad():
        gen = model.generate(ids, max_new_tokens=3, do_sample=False)
    assert gen.shape == (1, 15)


# the linear set of the transformers default Qwen3.5 text config (the "Qwen3.5-9B style" configuration of
# configuration_qwen3_5.py: hidden 4096, intermediate 12288, 16 heads x 256, 4 KV heads, 16 key / 32 value heads x 128)
QWEN35_DEFAULT_SHAPES = [
    ("full_attn.qkv (gated q)", 4096, [8192, 1024, 1024]),
    ("full_attn.o", 4096, [4096]),
    ("linear_attn.in_proj_qkv+z", 4096, [8192, 4096]),
    ("linear_attn.out_proj", 4096, [4096]),
    ("mlp.gate_up", 4096, [12288, 12288]),
    ("mlp.down", 12288, [4096]),
]


@pytest.mark.parametrize("name,K,sizes", QWEN35_DEFAULT_SHAPES)
@pytest.mark.parametrize("rows", [1, 8])
def test_qwen35_default_config_linear_shapes(dev, name, K, sizes, rows):
    """Decode GEMV (fused family at one row, chain family at eight) at the REAL dimensions of the Qwen3.5 linear set,
    against oracle rows: 64 sampled output columns per partition in float64 (the full oracle matmul at these sizes takes
    minutes), plus linearity over the whole output."""
    from paroquant_amd import ops
    from paroquant_amd.linear import PackedParoWeights
    L = po.make_layer(K + len(sizes), K, sizes)
    pk = PackedParoWeights(_t(L["qweight"], dev), _t(L["qzeros"], dev), _t(L["scales"], dev), _t(L["theta"], dev), _t(L["pairs"], dev),
                           _t(L["channel_scales"], dev), sizes)
    rng = np.random.default_rng(rows)
    x = rng.standard_normal((rows, K)).astype(np.float16)
    if rows == 1:
        y = pk.apply(_t(x, dev))
    else:
        y, _ = ops.chain_gemv(ops.rotate_parts(_t(x, dev), pk), pk)
    got = _np(y)
    # oracle on sampled columns: rotate per partition (float64), dequantise only the sampled columns
    w = po.dequant_awq(L["qweight"], L["qzeros"], L["scales"], 128, out_dtype=np.float64)
    c0 = 0
    for p, n in enumerate(sizes):
        xr = po.rotate(x.astype(np.float64), L["pairs"][p], L["theta"][p].astype(np.float64), L["channel_scales"][p].reshape(-1).astype(np.float64), 128, mode="ideal")
        cols = c0 + rng.choice(n, size=64, replace=False)
        ref = xr @ w[:, cols]
        assert np.max(np.abs(got[:, cols] - ref)) / max(np.max(np.abs(ref)), 1e-30) < TIGHT_F16, (name, p)
        c0 += n
    # linearity over the full output: f(2x) == 2 f(x) up to rounding
    if rows == 1:
        y2 = pk.apply(_t(x * np.float16(2), dev))
    else:
        y2, _ = ops.chain_gemv(ops.rotate_parts(_t(x * np.float16(2), dev), pk), pk)
    assert po.rel_err(_np(y2), 2.0 * got) < TIGHT_F16


def test_decoder_harness_matches_hf_qwen3_5(dev, tmp_path):
    """VERDICT r3 missing #1 (f2): the Qwen3.5 decode harness (paroquant_amd/decoder_qwen35.py; csrc/gdn.hip: conv1d update, the gated
    delta-net recurrence on the 128 x 128 state of every value head, gated RMSNorm; gated head_dim-256 attention with partial rotary)
    against HF's OWN modelling code: the dense fp32 Qwen3_5ForCausalLM that carries, for every quantised linear, the matrix the oracle's
    linear applies.  Teacher-forced, position by position over 3 gated-delta-net layers + 1 full-attention layer: the logits of every
    position agree, the recurrent / convolution states are the model's, and greedy generation through the captured graph reproduces
    the eager steps."""
    from paroquant_amd.decoder_qwen35 import ParoQwen35DecoderLM
    from tests.hf_ckpt import write_tiny_paro_qwen35
    from transformers.models.qwen3_5.configuration_qwen3_5 import Qwen3_5TextConfig
    from transformers.models.qwen3_5.modeling_qwen3_5 import Qwen3_5ForCausalLM
    layers, dense, cfg = write_tiny_paro_qwen35(str(tmp_path), seed=3)
    lm = ParoQwen35DecoderLM.from_checkpoint(str(tmp_path), dev, max_positions=64)
    assert [L.full for L in lm.layers] == [False, False, False, True] and lm.rd == 64
    c = Qwen3_5TextConfig(**{k: v for k, v in cfg.items() if k not in ("architectures", "model_type", "torch_dtype")})
    ref_model = Qwen3_5ForCausalLM(c).float().to(dev)
    missing, unexpected = ref_model.load_state_dict({k: v.float() for k, v in dense.items()}, strict=False)
    assert not missing and not unexpected
    T = 24
    ids = torch.randint(0, cfg["vocab_size"], (T,), device=dev, generator=torch.Generator(device=dev).manual_seed(9))
    with torch.no_grad():
        ref_logits = ref_model(input_ids=ids[None]).logits[0]                       # [T, V], fp32 model code, whole sequence at once
    lm.reset()
    got = []
    for i in range(T):
        lm.tok.copy_(ids[i:i + 1])
        lm.decode_step()
        got.append(lm.logits[0].clone())
    got = torch.stack(got)
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    err = po.rel_err(_np(got), _np(ref_logits))
    assert err < 3e-2, err                       # fp16 activations / hand-written mixers vs fp32 model code around identical linears
    agree = (got.float().argmax(-1) == ref_logits.argmax(-1)).float().mean().item()
    assert agree >= 0.9, agree
    # the captured graph replays the same steps bit for bit (states restored around the capture's warm-up), and generate() continues
    eager_logits = got[-1].clone()
    last = lm.prefill(ids, sequential=True)
    assert torch.equal(last[0], eager_logits)
    # the row form of the prompt pass (every linear over all T rows, the delta rule as one launch per layer): the same last-position logits
    # to rounding, and -- against HF -- as close as the token-by-token route
    last_rows = lm.prefill(ids)
    assert po.rel_err(_np(last_rows[0]), _np(eager_logits)) < 2e-2
    assert po.rel_err(_np(last_rows[0]), _np(ref_logits[-1])) < 3e-2
    toks, stats = lm.generate(ids, 6)
    assert toks.shape == (T + 6,) and torch.equal(toks[:T], ids) and stats["decode_tokens_per_s"] > 0
    with torch.no_grad():
        ref_gen = ref_model.generate(ids[None], max_new_tokens=6, do_sample=False)[0]
    assert (toks == ref_gen).float().mean().item() >= 0.9                       # greedy continuations agree (ties aside)


@pytest.mark.gpu
def test_qwen35_harness_deferred_matches_reducer(monkeypatch):
    """The Qwen3.5 decode harness with the deferred K-split reduction (out_proj / o_proj / down_proj leave partial sums, gate_up and the next
    block's in_proj complete the residual stream -- also the stream gdn_prep's dense rows read) against the in-launch reducer
    (PARO_DEFERRED_KSPLIT=0): logits bit for bit and tokens one for one, eager and graph, over both layer kinds."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from paroquant_amd.decoder_qwen35 import ParoQwen35DecoderLM, Qwen35Config
    dev = torch.device("cuda:0")
    cfg = lambda: Qwen35Config(512, 2048, 8, 2, 256, 4, 16, 4, 640, ["linear_attention", "linear_attention", "linear_attention", "full_attention"],
                               max_positions=64)            # out_proj 2048 -> 512, o_proj 2048 -> 512, down 2048 -> 512: all K-split
    ids = torch.tensor([3, 17, 101, 7, 250, 9, 33], device=dev)
    lm_d = ParoQwen35DecoderLM.random(cfg(), dev, seed=5)
    assert lm_d.deferred
    monkeypatch.setenv("PARO_DEFERRED_KSPLIT", "0")
    lm_r = ParoQwen35DecoderLM.random(cfg(), dev, seed=5)
    assert not lm_r.deferred
    for use_graph in (False, True):
        for lm in (lm_d, lm_r):
            lm.reset()
        td, _ = lm_d.generate(ids, 10, use_graph=use_graph)
        tr, _ = lm_r.generate(ids, 10, use_graph=use_graph)
        assert torch.equal(td, tr)
        assert torch.equal(lm_d.logits, lm_r.logits)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_qwen35_row_prefill_matches_sequential(dtype):
    """paro_gdn_sequence + the row-parallel prompt pass against the decode step run token by token (the route the HF comparison pins):
    last-position logits, every layer's recurrent state, convolution state and KV cache, and ten greedy decode steps continued from each."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from paroquant_amd.decoder_qwen35 import ParoQwen35DecoderLM, Qwen35Config
    dev = torch.device("cuda:0")
    cfg = Qwen35Config(512, 1024, 4, 2, 256, 2, 4, 4, 640, ["linear_attention", "linear_attention", "linear_attention", "full_attention"], max_positions=96)
    lm = ParoQwen35DecoderLM.random(cfg, dev, seed=11, dtype=dtype)
    ids = torch.randint(0, 640, (70,), device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    tol = 2e-2 if dtype == torch.float16 else 8e-2
    f = lambda t: t.float().cpu().numpy().astype(np.float64)
    ls = lm.prefill(ids, use_graph=False, sequential=True)
    snap = [(L.state.clone(), L.conv_state.clone()) if not L.full else (L.kcache.clone(), L.vcache.clone()) for L in lm.layers]
    seq_logits = [ls.clone()]
    for _ in range(10):
        lm.decode_step()
        seq_logits.append(lm.logits.clone())
    lr = lm.prefill(ids, use_graph=False)
    assert int(lm.pos.item()) == 70
    assert po.rel_err(f(lr), f(ls)) < tol
    for L, (a, b) in zip(lm.layers, snap):
        if L.full:
            assert po.rel_err(f(L.kcache[:, :70]), f(a[:, :70])) < tol and po.rel_err(f(L.vcache[:, :70]), f(b[:, :70])) < tol
        else:
            assert po.rel_err(f(L.state), f(a)) < tol
            assert po.rel_err(f(L.conv_state[:, 1:]), f(b[:, 1:])) < tol
    lm.tok.copy_(torch.argmax(ls, dim=-1))                      # continue from the SAME token in both routes
    for i in range(10):
        lm.decode_step()
        assert po.rel_err(f(lm.logits), f(seq_logits[i + 1])) < 2 * tol, i
        lm.tok.copy_(torch.argmax(seq_logits[i + 1], dim=-1))
