#!/usr/bin/env python3
"""G10 -- the layer export at group_size 64 (cli/convert.py:239-277 `_quantize_layer`, :158-191 `_quantize_rotated_weight`,
:194-203 `_to_awq_buffers`) run by IMPORTING the reference: pins the [K/64]-row scale / zero tensors and the packing at
the second group size the reference's operators accept (rotation.cu:117-123).  `paroquant.kernels.cuda` is stubbed
(CUDA-only JIT); its rotation is the oracle's fp32 one.  Run in the build container only:
    python tests/golden/make_golden_g10.py"""
import os, shutil, sys, tempfile, types
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import paro_oracle as po  # noqa: E402
REF = "/root/reference"


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted")
    tmp = tempfile.mkdtemp(prefix="paro_ref_")
    try:
        dst = os.path.join(tmp, "ref")
        shutil.copytree(REF, dst, ignore=shutil.ignore_patterns(".git"))
        sys.path.insert(0, dst)

        def _rot(x, idx_ij, theta, scales=None, group_size=128):
            out = po.rotate(x.detach().cpu().numpy(), idx_ij.cpu().numpy(), theta.detach().cpu().numpy(),
                            None if scales is None else scales.detach().cpu().numpy(), int(group_size), mode="f32")
            return torch.from_numpy(np.ascontiguousarray(out)).to(x.dtype)
        stub = types.ModuleType("paroquant.kernels.cuda")
        stub.scaled_pairwise_rotation = _rot
        stub.RotateTensorFunc = None
        sys.modules["paroquant.kernels.cuda"] = stub
        import paroquant.cli.convert as cv
        import paroquant.optim.quantizer as qz

        g = torch.Generator().manual_seed(1064)
        N, K, GS = 48, 256, 64
        w = torch.randn(N, K, generator=g) * 0.05
        cs = torch.rand(K, generator=g) * 1.5 + 0.5
        rng = np.random.default_rng(64)
        pairs = torch.from_numpy(po.random_pairs(rng, 8, K, GS))          # the optimiser pairs inside group_size groups
        theta = torch.randn(8, K // 2, generator=g) * 0.1
        rot = torch.from_numpy(po.rotate((w * cs).numpy(), pairs.numpy(), theta.numpy(), None, GS, "f32"))
        s, z = qz._calc_scales_and_zero_points(rot, GS, 0, 15)
        sd = {"weight": w.half(), "n_bits": torch.tensor(4), "group_size": torch.tensor(GS), "pairs_grouped": pairs,
              "angles_grouped": theta, "channel_scales": cs, "quantizer.scale": s, "quantizer.zero_point_float": z}
        buffers, bits, gsz, kr = cv._quantize_layer(sd, "cpu")
        np.savez_compressed(os.path.join(HERE, "quantize_layer_g64.npz"), weight=sd["weight"].numpy(), channel_scales_opt=cs.numpy(),
                            pairs_in=pairs.numpy(), theta_in=theta.numpy(), scale=s.numpy(), zero_point_float=z.numpy(),
                            bits=np.int32(bits), group_size=np.int32(gsz), krot=np.int32(kr),
                            **{f"out_{k}": v.numpy() for k, v in buffers.items()})
        print("G10 written:", {k: tuple(v.shape) for k, v in buffers.items()}, bits, gsz, kr)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
