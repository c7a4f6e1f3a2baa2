#!/usr/bin/env python3
"""G7b -- pin the STAGE ORDER of the multi-stage rotation to reference-held code.

G7 pins orientation and pair layout of ONE stage.  The order in which the krot stages are applied is stated by the
reference in exactly one CPU-evaluable place: the loop of `RotateTensorFunc.backward`
(paroquant/kernels/cuda/autograd.py:34-38, 54-59)

    for i in range(KROT - 1, -1, -1):
        g = rotate(g, idx_ij[[i]], -theta[[i]])            # ONE stage at a time, last stage first
    grad_x = g * scale

i.e. grad_x = diag(scale) R_0^T R_1^T ... R_{K-1}^T grad_out, which is the transpose of the forward only if the forward is
F(x) = R_{K-1} ... R_1 R_0 (x * scale): stage 0 first.  The rotation stub is called with ONE stage per call inside that
loop, so the ORDER is the reference's, not the oracle's; each single stage is the convention G7 pins.

The fixture stores (idx [8, K] with non-commuting consecutive stages, theta, scale, G, grad_x of the reference loop).  The
test checks  <F(d), G> == <grad_x, d>  for the oracle's multi-stage forward F (linear in x), and that the same identity
FAILS for the forward with the stages reversed.   Run in the build container only:  python tests/golden/make_golden_g7b.py
"""
from __future__ import annotations

import importlib.util
import os
import shutil
import sys
import tempfile

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
        raise SystemExit("reference not mounted; fixtures can only be regenerated in the build container")
    tmp = tempfile.mkdtemp(prefix="paro_ref_")
    try:
        src = os.path.join(tmp, "autograd.py")
        shutil.copy(os.path.join(REF, "paroquant", "kernels", "cuda", "autograd.py"), src)
        lib = torch.library.Library("rotation", "DEF")
        lib.define("rotate(Tensor x, Tensor idx_ij, Tensor theta, Tensor? scales=None, int group_size=128) -> Tensor")
        calls = []

        def _oracle_rotate(x, idx_ij, theta, scales=None, group_size=128):
            calls.append(int(idx_ij.shape[0]))
            out = po.rotate(x.detach().double().numpy(), idx_ij.numpy(), theta.detach().double().numpy(),
                            None if scales is None else scales.detach().double().numpy(), int(group_size), mode="ideal")
            return torch.from_numpy(np.ascontiguousarray(out)).to(x.dtype)

        lib.impl("rotate", _oracle_rotate, "CPU")
        spec = importlib.util.spec_from_file_location("ref_autograd", src)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)

        rng = np.random.default_rng(707)
        K, B, KROT = 256, 4, 8
        idx = torch.from_numpy(po.random_pairs(rng, KROT, K))               # independent random matchings: consecutive stages do not commute
        theta = torch.from_numpy(rng.standard_normal((KROT, K // 2)) * 0.7)  # large angles: the order matters at O(1)
        x = torch.from_numpy(rng.standard_normal((B, K))).requires_grad_()
        scale = torch.from_numpy(rng.uniform(0.5, 2.0, K)).requires_grad_()
        G = torch.from_numpy(rng.standard_normal((B, K)))
        th = theta.clone().requires_grad_()
        y = ref.RotateTensorFunc.apply(x, idx, th, scale, 128)
        n_fwd = len(calls)
        y.backward(G)
        # inside backward() the stub saw ONE stage per call (2 calls per stage: t and g), never the whole schedule
        assert calls[:n_fwd] == [KROT] and calls[n_fwd:] == [1] * (2 * KROT), calls
        grad_x = x.grad.numpy()
        d = rng.standard_normal((B, K))
        fwd = lambda ii, tt: po.rotate(d, ii, tt, scale.detach().numpy(), 128, mode="ideal")
        lhs = float((fwd(idx.numpy(), theta.numpy()) * G.numpy()).sum())
        rhs = float((grad_x * d).sum())
        wrong = float((fwd(idx.numpy()[::-1].copy(), theta.numpy()[::-1].copy()) * G.numpy()).sum())
        print(f"G7b: <F(d), G> = {lhs:.9f}, <grad_x, d> = {rhs:.9f}, reversed stage order gives {wrong:.6f}")
        assert abs(lhs - rhs) < 1e-9 * max(1.0, abs(rhs)) and abs(wrong - rhs) > 1e-2 * max(1.0, abs(rhs))
        np.savez_compressed(os.path.join(HERE, "rotate_stage_order.npz"), idx=idx.numpy(), theta=theta.numpy(), scale=scale.detach().numpy(),
                            G=G.numpy(), d=d, grad_x=grad_x, y=y.detach().numpy(), x=x.detach().numpy())
        print("wrote", os.path.join(HERE, "rotate_stage_order.npz"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
