"""Worker of tests/test_gpu_parity.py::test_tp_decoder_ranks_share_one_gpu: `world` processes share cuda:0 (gloo as the
control channel).  Every rank builds the SAME tiny synthetic checkpoint (oracle make_layer, fixed seeds), keeps its Megatron
shard (ParoDecoderLM.from_raw -> paroquant_amd.tp.shard_*), and the tensor-parallel model -- kernel-level one-shot all-reduce
after o / down, residual added inside it -- must reproduce the unsharded model's logits and greedy tokens step by step;
then the same through a captured HIP graph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from oracle import paro_oracle as po
    from paroquant_amd import tp
    from paroquant_amd.decoder import ParoDecoderLM, DecoderConfig
    H, I, NH, NKV, HD, LAYERS, V = 512, 1024, 8, 4, 64, 2, 512
    cfg = lambda: DecoderConfig(H, I, NH, NKV, HD, LAYERS, V, 1e-6, 10000.0, True, 64)
    rng = np.random.default_rng(2026)
    raw = []
    for l in range(LAYERS):
        raw.append(dict(qkv=po.make_layer(10 * l + 1, H, [NH * HD, NKV * HD, NKV * HD]), o=po.make_layer(10 * l + 2, NH * HD, [H]),
                        gate_up=po.make_layer(10 * l + 3, H, [I, I]), down=po.make_layer(10 * l + 4, I, [H]),
                        in_norm=(1 + 0.1 * rng.standard_normal(H)).astype(np.float16), post_norm=(1 + 0.1 * rng.standard_normal(H)).astype(np.float16),
                        q_norm=(1 + 0.1 * rng.standard_normal(HD)).astype(np.float16), k_norm=(1 + 0.1 * rng.standard_normal(HD)).astype(np.float16)))
    shared = dict(embed=(rng.standard_normal((V, H)) * 0.5).astype(np.float16), lm_head=(rng.standard_normal((V, H)) * H ** -0.5).astype(np.float16),
                  final_norm=(1 + 0.1 * rng.standard_normal(H)).astype(np.float16))
    allreduce, name = tp.make_allreduce(dev, H)
    assert name == "oneshot", name
    lm_tp = ParoDecoderLM.from_raw(cfg(), raw, shared, dev, tp_rank=rank, tp_world=world, allreduce=allreduce)
    lm_ref = ParoDecoderLM.from_raw(cfg(), raw, shared, dev)                 # the unsharded model, on every rank
    ids = torch.tensor([5, 99, 3, 250, 17, 402, 8], device=dev)

    def teacher_forced(lm, n_new):
        logits, toks = [], []
        for i in range(int(ids.numel()) + n_new):
            if i < ids.numel():
                lm.tok.copy_(ids[i:i + 1])
            lm.pos.fill_(i)
            lm.decode_step()
            logits.append(lm.logits.float().clone())
            toks.append(int(lm.tok.item()))
        return torch.cat(logits), toks

    assert lm_tp.fused_allreduce                             # the row-parallel GEMVs exchange their partials in their epilogue
    lg_tp, tk_tp = teacher_forced(lm_tp, 8)
    lg_ref, tk_ref = teacher_forced(lm_ref, 8)
    err = float((lg_tp - lg_ref).abs().max()) / float(lg_ref.abs().max())
    assert err < 2e-2, err
    assert tk_tp == tk_ref, (tk_tp, tk_ref)
    # the same model with the all-reduce as its own launch (partial sums rounded per rank before the exchange)
    lm_sep = ParoDecoderLM.from_raw(cfg(), raw, shared, dev, tp_rank=rank, tp_world=world, allreduce=allreduce)
    lm_sep.fused_allreduce = False
    lg_sep, tk_sep = teacher_forced(lm_sep, 8)
    err_sep = float((lg_sep - lg_ref).abs().max()) / float(lg_ref.abs().max())
    assert err_sep < 2e-2 and tk_sep == tk_ref, (err_sep, tk_sep, tk_ref)
    del lm_sep
    # one row-parallel linear, directly: fused exchange == sum of the ranks' fp32-accumulated partials, bit-identical on all ranks
    from paroquant_amd import ops, tp as ptp
    lay = po.make_layer(77, 512, [H])
    keys = ("qweight", "qzeros", "scales", "theta", "pairs", "channel_scales")
    xs = torch.randn(1, 512, device=dev, dtype=torch.float16, generator=torch.Generator(device=dev).manual_seed(5))
    res = torch.randn(1, H, device=dev, dtype=torch.float16, generator=torch.Generator(device=dev).manual_seed(6))
    from paroquant_amd.linear import PackedParoWeights
    sh = ptp.shard_row_parallel({**{k: torch.from_numpy(np.ascontiguousarray(lay[k])) for k in keys}, "bias": None}, rank, world)
    pk_r = PackedParoWeights(*[sh[k].to(dev) for k in keys], [H])
    k0 = 512 // world * rank
    x_r = xs[:, k0:k0 + 512 // world].contiguous()
    y_f = ops.w4a16_gemv_fused(x_r, pk_r, 0, residual=res, allreduce=allreduce)
    part = ops.w4a16_gemv_fused(x_r, pk_r, 0).float()
    dist.all_reduce(part)
    want = part + res.float()
    assert float((y_f.float() - want).abs().max()) <= 2e-2 * float(want.abs().max()), float((y_f.float() - want).abs().max())
    ys = [torch.empty_like(y_f) for _ in range(world)]
    dist.all_gather(ys, y_f)
    assert all(torch.equal(ys[0], t) for t in ys)
    # the sharded prefill (local heads / MLP columns through the GEMM path, [T, hidden] all-reduced over the process group)
    pl_tp, pl_ref = lm_tp.prefill(ids).float(), lm_ref.prefill(ids).float()
    perr = float((pl_tp - pl_ref).abs().max()) / float(pl_ref.abs().max())
    assert perr < 2e-2, perr
    assert int(lm_tp.tok.item()) == int(lm_ref.tok.item()) and int(lm_tp.pos.item()) == int(ids.numel())
    # every rank holds the same tokens, and the captured graph replays the same sequence
    gathered = [None] * world
    dist.all_gather_object(gathered, tk_tp)
    assert all(g == gathered[0] for g in gathered)
    lm_g = ParoDecoderLM.from_raw(cfg(), raw, shared, dev, tp_rank=rank, tp_world=world, allreduce=allreduce)
    toks_g, stats = lm_g.generate(ids, 9, use_graph=True)
    assert toks_g[int(ids.numel()):].tolist() == tk_tp[int(ids.numel()) - 1:int(ids.numel()) - 1 + 9], (toks_g.tolist(), tk_tp)
    assert not allreduce.gave_up()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(f"TP_DECODER_OK world {world} rel_err {err:.2e} decode_tok_s {stats['decode_tokens_per_s']:.0f}", flush=True)


if __name__ == "__main__":
    main()
