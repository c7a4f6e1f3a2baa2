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

    lg_tp, tk_tp = teacher_forced(lm_tp, 8)
    lg_ref, tk_ref = teacher_forced(lm_ref, 8)
    err = float((lg_tp - lg_ref).abs().max()) / float(lg_ref.abs().max())
    assert err < 2e-2, err                                   # row-parallel partial sums are rounded per rank before the all-reduce
    assert tk_tp == tk_ref, (tk_tp, tk_ref)
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
