#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by importing the REFERENCE's own
Python (z-lab/paroquant, mounted read-only at /root/reference).

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

The reference is imported from a throw-away copy under /tmp with
PYTHONDONTWRITEBYTECODE=1 so nothing is ever written into /root/reference, and
``paroquant.kernels.cuda`` (a CUDA-only JIT extension that cannot build here) is
replaced by a stub before import.  Only inputs and outputs are stored; no
reference source travels.

Fixtures:
  G1 pack_awq.npz            cli/convert.py:149        _pack_awq
  G2 to_awq_buffers.npz      cli/convert.py:194-203    _to_awq_buffers
  G3 quantizer.npz           optim/quantizer.py:10-25,87-117
  G4 kernel_pairs.npz        optim/train.py:56-91 -> optim/rotation.py:69-87
  G5 quantize_rotated.npz    cli/convert.py:158-191 (rotation = our oracle, stubbed in)
  G6 quantize_layer.npz      cli/convert.py:239-277    _quantize_layer end to end
"""
from __future__ import annotations

import os
import shutil
import sys
import tempfile
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import paro_oracle as po  # noqa: E402

REF = "/root/reference"


def _import_reference():
    tmp = tempfile.mkdtemp(prefix="paro_ref_")
    dst = os.path.join(tmp, "ref")
    shutil.copytree(REF, dst, ignore=shutil.ignore_patterns(".git"))
    sys.path.insert(0, dst)

    def _oracle_rotation(x, idx_ij, theta, scales=None, group_size=128):
        out = po.rotate(x.detach().cpu().numpy(), idx_ij.cpu().numpy(), theta.detach().cpu().numpy(),
                        None if scales is None else scales.detach().cpu().numpy(), int(group_size), mode="f32")
        return torch.from_numpy(np.ascontiguousarray(out)).to(x.dtype)

    stub = types.ModuleType("paroquant.kernels.cuda")
    stub.scaled_pairwise_rotation = _oracle_rotation
    stub.RotateTensorFunc = None
    sys.modules["paroquant.kernels.cuda"] = stub
    import paroquant.cli.convert as cv
    import paroquant.optim.quantizer as qz
    import paroquant.optim.rotation as rot
    import paroquant.optim.train as tr
    return tmp, cv, qz, rot, tr


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted; fixtures can only be regenerated in the build container")
    tmp, cv, qz, rot, tr = _import_reference()
    try:
        g = torch.Generator().manual_seed(1234)

        # G1 -----------------------------------------------------------------
        vals = torch.randint(0, 16, (256, 64), generator=g, dtype=torch.int32)
        packed = cv._pack_awq(vals)
        kat = cv._pack_awq(torch.tensor([[12, 15, 5, 0, 3, 11, 3, 7]], dtype=torch.int32))
        np.savez_compressed(os.path.join(HERE, "pack_awq.npz"), values=vals.numpy(), packed=packed.numpy(),
                            kat_values=np.array([[12, 15, 5, 0, 3, 11, 3, 7]], dtype=np.int32),
                            kat_packed=kat.numpy())

        # G2 -----------------------------------------------------------------
        N, K, gs = 48, 256, 128
        quantized = torch.randint(0, 16, (N, K), generator=g, dtype=torch.int32)
        scales_2d = torch.rand(N, K // gs, generator=g) * 0.02 + 0.001
        zeros_2d = torch.randint(0, 16, (N, K // gs), generator=g, dtype=torch.int32)
        bufs = cv._to_awq_buffers(quantized, scales_2d, zeros_2d)
        np.savez_compressed(os.path.join(HERE, "to_awq_buffers.npz"), quantized=quantized.numpy(),
                            scales_2d=scales_2d.numpy(), zeros_2d=zeros_2d.numpy(),
                            qweight=bufs["qweight"].numpy(), qzeros=bufs["qzeros"].numpy(),
                            scales=bufs["scales"].numpy())

        # G3 -----------------------------------------------------------------
        w = torch.randn(32, 256, generator=g) * 0.05
        sc, zp = qz._calc_scales_and_zero_points(w, 128, 0, 15)
        pq_auto = qz.UniformAffineQuantizer.pseudo_quantize(w, 4, 128)
        sc2 = sc * (1.0 + 0.1 * torch.rand(sc.shape, generator=g))
        zp2 = zp + torch.randn(zp.shape, generator=g)
        pq_given = qz.UniformAffineQuantizer.pseudo_quantize(w, 4, 128, sc2, zp2)
        np.savez_compressed(os.path.join(HERE, "quantizer.npz"), w=w.numpy(), scale=sc.numpy(), zero_point=zp.numpy(),
                            pq_auto=pq_auto.numpy(), scale2=sc2.numpy(), zero_point2=zp2.numpy(),
                            pq_given=pq_given.numpy())

        # G4 -----------------------------------------------------------------
        Kp, krot = 512, 8
        sens = torch.zeros(Kp // 128, 128)
        pairs_k = tr.get_random_rotation_pairs(sens, 128, krot, 0.5, seed=7)
        pairs_t = [torch.tensor(p, dtype=torch.int32) for p in pairs_k]
        angles_t = [torch.randn(len(p), generator=g) * 0.2 for p in pairs_k]
        kp, ka, km = rot.transform_to_kernel_data(pairs_t, angles_t, group_size=128)
        # partial matching (factor 0.25) exercises the dummy-pair padding of _align_shape
        pairs_q = tr.get_random_rotation_pairs(sens, 128, 2, 0.25, seed=11)
        pairs_qt = [torch.tensor(p, dtype=torch.int32) for p in pairs_q]
        angles_qt = [torch.randn(len(p), generator=g) * 0.2 for p in pairs_q]
        kp2, ka2, km2 = rot.transform_to_kernel_data(pairs_qt, angles_qt, group_size=128)
        np.savez_compressed(os.path.join(HERE, "kernel_pairs.npz"),
                            pairs=kp.numpy(), angles=ka.numpy(), mask=km.numpy(),
                            raw_pairs=np.stack([p.numpy() for p in pairs_t]),
                            raw_angles=np.stack([a.numpy() for a in angles_t]),
                            pairs_partial=kp2.numpy(), angles_partial=ka2.numpy(), mask_partial=km2.numpy(),
                            raw_pairs_partial=np.stack([p.numpy() for p in pairs_qt]),
                            raw_angles_partial=np.stack([a.numpy() for a in angles_qt]))

        # G5 -----------------------------------------------------------------
        N5, K5 = 32, 256
        w5 = torch.randn(N5, K5, generator=g) * 0.05
        cs5 = torch.rand(K5, generator=g) * 1.5 + 0.5
        rng = np.random.default_rng(5)
        pairs5 = torch.from_numpy(po.random_pairs(rng, 8, K5))
        theta5 = torch.randn(8, K5 // 2, generator=g) * 0.1
        rot5 = torch.from_numpy(po.rotate((w5 * cs5).numpy(), pairs5.numpy(), theta5.numpy(), None, 128, "f32"))
        s5, z5 = qz._calc_scales_and_zero_points(rot5, 128, 0, 15)
        q5, s2d, z2d = cv._quantize_rotated_weight(weight=w5, pairs=pairs5, theta=theta5, channel_scales=cs5,
                                                   scales_flat=s5, zp_flat=z5, bits=4, group_size=128, device="cpu")
        np.savez_compressed(os.path.join(HERE, "quantize_rotated.npz"), weight=w5.numpy(), channel_scales=cs5.numpy(),
                            pairs=pairs5.numpy(), theta=theta5.numpy(), scales_flat=s5.numpy(), zp_flat=z5.numpy(),
                            quantized=q5.numpy(), scales_2d=s2d.numpy(), zeros_2d=z2d.numpy())

        # G6 -----------------------------------------------------------------
        sd = {
            "weight": w5.half(), "n_bits": torch.tensor(4), "group_size": torch.tensor(128),
            "pairs_grouped": pairs5, "angles_grouped": theta5, "channel_scales": cs5,
            "quantizer.scale": s5, "quantizer.zero_point_float": z5,
            "bias": torch.randn(N5, generator=g) * 0.1,
        }
        buffers, bits, gsz, kr = cv._quantize_layer(sd, "cpu")
        np.savez_compressed(os.path.join(HERE, "quantize_layer.npz"),
                            weight=sd["weight"].numpy(), channel_scales_opt=cs5.numpy(), pairs_in=pairs5.numpy(),
                            theta_in=theta5.numpy(), scale=s5.numpy(), zero_point_float=z5.numpy(),
                            bias_in=sd["bias"].numpy(),
                            bits=np.int32(bits), group_size=np.int32(gsz), krot=np.int32(kr),
                            **{f"out_{k}": v.numpy() for k, v in buffers.items()})
        print("golden fixtures written to", HERE)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
