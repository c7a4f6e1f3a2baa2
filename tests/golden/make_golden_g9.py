#!/usr/bin/env python3
"""G9 -- the MoE expert export (cli/convert.py:280-379, `_quantize_moe`) run by IMPORTING the reference on a small
synthetic optimiser state dict: pins which gate_up rows are gate / up, the per-expert AWQ stacking, the shared-rotation
buffers and their names.  `paroquant.kernels.cuda` is stubbed (CUDA-only JIT); its rotation is the oracle's fp32 one.
Run in the build container only:  python tests/golden/make_golden_g9.py"""
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

        g = torch.Generator().manual_seed(99)
        E, H, I, krot = 3, 256, 128, 8
        rng = np.random.default_rng(9)
        gate_up = torch.randn(E, 2 * I, H, generator=g) * 0.05
        down = torch.randn(E, H, I, generator=g) * 0.05
        gu_pairs = torch.from_numpy(po.random_pairs(rng, krot, H))
        dn_pairs = torch.from_numpy(po.random_pairs(rng, krot, I))
        gu_theta = torch.randn(krot, H // 2, generator=g) * 0.1
        dn_theta = torch.randn(krot, I // 2, generator=g) * 0.1
        gu_cs = torch.rand(H, generator=g) * 1.5 + 0.5
        dn_cs = torch.rand(I, generator=g) * 1.5 + 0.5
        gu_rot = torch.from_numpy(po.rotate((gate_up.reshape(-1, H) * gu_cs).numpy(), gu_pairs.numpy(), gu_theta.numpy(), None, 128, "f32"))
        dn_rot = torch.from_numpy(po.rotate((down.reshape(-1, I) * dn_cs).numpy(), dn_pairs.numpy(), dn_theta.numpy(), None, 128, "f32"))
        gu_s, gu_z = qz._calc_scales_and_zero_points(gu_rot, 128, 0, 15)
        dn_s, dn_z = qz._calc_scales_and_zero_points(dn_rot, 128, 0, 15)
        sd = {"n_bits": torch.tensor(4), "group_size": torch.tensor(128), "gate_up_weight": gate_up, "down_weight": down,
              "gate_up_pairs_grouped": gu_pairs, "gate_up_angles_grouped": gu_theta, "gate_up_channel_scales": gu_cs,
              "gate_up_quantizer.scale": gu_s, "gate_up_quantizer.zero_point_float": gu_z,
              "down_pairs_grouped": dn_pairs, "down_angles_grouped": dn_theta, "down_channel_scales": dn_cs,
              "down_quantizer.scale": dn_s, "down_quantizer.zero_point_float": dn_z}
        bufs, rots, bits, gs, kr = cv._quantize_moe(sd, "cpu")
        out = {f"in_{k.replace('.', '__')}": v.numpy() for k, v in sd.items()}
        for proj, d in bufs.items():
            for k, v in d.items():
                out[f"out_{proj}_{k}"] = v.numpy()
        for k, v in rots.items():
            out[f"rot_{k}"] = v.numpy()
        out["bits"], out["group_size"], out["krot"] = np.int32(bits), np.int32(gs), np.int32(kr)
        np.savez_compressed(os.path.join(HERE, "quantize_moe.npz"), **out)
        print("wrote quantize_moe.npz", {k: v.shape for k, v in out.items() if k.startswith("out_")})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
