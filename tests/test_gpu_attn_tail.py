"""GPU parity of the attention tail (ABI v18, ``paro_attn_tail_t``): the decode attention that consumes the qkv projection runs in
the projection's own launch -- its workgroups ride in one more grid row, request their K / V cache lines while the projection
streams, and take q / k / v as {partial sum, launch tag} granules from the projection's K-slices.

Same arithmetic as the two launches (``paro_w4a16_gemv_fused(parts_out)`` then ``paro_attn_decode_split(qkv_parts)``, the decode
harness's route since round 4, itself pinned against HF's modelling code by tests/test_gpu_parity.py::test_decoder_harness_matches_hf):
every comparison here is BIT for bit against that route -- logits of every step, the KV caches, the generated tokens -- eager and
replayed from a HIP graph, fp16 and bf16, 2 and 4 query heads per KV head, positions crossing the 64-position chunk boundaries."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    from paroquant_amd import _native
    _native.load()
    return torch.device("cuda:0")


def _pair(dev, dtype, nh, nkv, monkeypatch, max_positions=192, layers=2, hidden=2048, inter=4096):
    from paroquant_amd.decoder import ParoDecoderLM, DecoderConfig
    cfg = lambda: DecoderConfig(hidden, inter, nh, nkv, 128, layers, 640, 1e-6, 10000.0, True, max_positions)
    monkeypatch.delenv("PARO_FUSE_QKV_ATTN", raising=False)
    lm_f = ParoDecoderLM.random(cfg(), dev, seed=11, dtype=dtype)
    monkeypatch.setenv("PARO_FUSE_QKV_ATTN", "0")
    lm_u = ParoDecoderLM.random(cfg(), dev, seed=11, dtype=dtype)
    monkeypatch.delenv("PARO_FUSE_QKV_ATTN", raising=False)
    assert lm_f.deferred and lm_f.deferred_qkv and lm_f.split_attn and lm_f.fuse_qkv_attn
    assert lm_u.deferred and lm_u.deferred_qkv and lm_u.split_attn and not lm_u.fuse_qkv_attn
    return lm_f, lm_u


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("nh,nkv", [(16, 8), (16, 4)])
def test_attention_tail_matches_two_launches_bit_for_bit(dev, dtype, nh, nkv, monkeypatch):
    lm_f, lm_u = _pair(dev, dtype, nh, nkv, monkeypatch)
    g = torch.Generator(device=dev).manual_seed(5)
    ids = torch.randint(0, 640, (150,), device=dev, generator=g)
    for use_graph in (False, True):
        for lm in (lm_f, lm_u):
            lm.prefill(ids[:50])
            if use_graph:
                lm.capture()
        for i in range(50, 150):                     # teacher-forced: positions 50 .. 149 (one, two, three 64-position chunks)
            for lm in (lm_f, lm_u):
                lm.tok.copy_(ids[i:i + 1])
                lm._graph.replay() if use_graph else lm.decode_step()
            lf, lu = lm_f.logits, lm_u.logits
            assert torch.isfinite(lf.float()).all(), (use_graph, i)
            assert torch.equal(lf, lu), f"graph {use_graph} step {i}: max |d| = {(lf.float() - lu.float()).abs().max().item()}"
    for a, b in zip(lm_f.layers, lm_u.layers):
        assert torch.equal(a.kcache, b.kcache) and torch.equal(a.vcache, b.vcache)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_attention_tail_generates_the_same_tokens(dev, dtype, monkeypatch):
    lm_f, lm_u = _pair(dev, dtype, 16, 4, monkeypatch)
    ids = torch.tensor([3, 17, 101, 7, 250, 9, 33], device=dev)
    for use_graph in (False, True):
        tf, _ = lm_f.generate(ids, 150, use_graph=use_graph)        # 7 + 150 positions: all three chunks of the 192-position cache
        tu, _ = lm_u.generate(ids, 150, use_graph=use_graph)
        assert torch.equal(tf, tu)
        assert torch.equal(lm_f.logits, lm_u.logits)
    from paroquant_amd import ops
    ops.check_workspace(lm_f.layers[0].qkv.workspace)


@pytest.mark.parametrize("dtype", [torch.float16])
def test_attention_tail_long_cache(dev, dtype, monkeypatch):
    """More attention workgroups than column blocks (they fill several grid rows), 128-position chunks and the per-slot tickets of
    contexts beyond 256 / 512 positions: still bit for bit the two launches."""
    lm_f, lm_u = _pair(dev, dtype, 16, 4, monkeypatch, max_positions=2048, layers=1)
    g = torch.Generator(device=dev).manual_seed(6)
    ids = torch.randint(0, 640, (700,), device=dev, generator=g)
    for lm in (lm_f, lm_u):
        lm.prefill(ids[:240])
        lm.capture()
    for i in range(240, 700, 1):
        for lm in (lm_f, lm_u):
            lm.tok.copy_(ids[i:i + 1])
            lm._graph.replay()
        if i % 23 == 0 or i in (255, 256, 257, 511, 512, 513):
            assert torch.equal(lm_f.logits, lm_u.logits), i
    assert torch.equal(lm_f.logits, lm_u.logits)
    for a, b in zip(lm_f.layers, lm_u.layers):
        assert torch.equal(a.kcache, b.kcache) and torch.equal(a.vcache, b.vcache)


def test_attention_tail_not_offered_where_it_is_not_built(dev):
    """head_dim 64 / more than four query heads per KV head / a projection that does not K-split: the decoder keeps the two launches."""
    from paroquant_amd.decoder import ParoDecoderLM, DecoderConfig
    lm = ParoDecoderLM.random(DecoderConfig(2048, 4096, 16, 2, 128, 1, 640, 1e-6, 10000.0, True, 256), dev, seed=11)      # 8 query heads per KV head
    assert not lm.fuse_qkv_attn
