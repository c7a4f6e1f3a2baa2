#!/usr/bin/env python3
"""G7 -- pin the ROTATION CONVENTION (orientation + pair layout) to reference-held code.

The reference has no CPU forward for `rotation::rotate`, but it does hold one CPU-evaluable statement of
what the forward is: the analytic d/dtheta expression in `RotateTensorFunc.backward`
(paroquant/kernels/cuda/autograd.py:40-52):

    di = idx[:, 0::2] + offset ; dj = idx[:, 1::2] + offset          # pair layout  (:40-42)
    grad_theta = (ga*b - gb*a) * cos(theta) - (ga*a + gb*b) * sin(theta)                     (:49-52)

With (a, b) the stage's INPUTS and (ga, gb) the gradient at the stage's OUTPUTS this is exactly
d/dtheta of   y_i = c a + s b,  y_j = c b - s a   (rotation.cuh:53-56); for the opposite orientation
(y_i = c a - s b, y_j = c b + s a) the first term changes sign.  So the expression only agrees with a
finite difference of a forward that follows the reference's orientation and pair layout.

Observation recorded here (v0.1.16): as shipped, backward() feeds the expression the gradient at the stage's
INPUTS (it inverse-rotates `g` before using it, :37-38), so its grad_theta is not the derivative of its own
forward (checked against finite differences below: max error O(1)).  To evaluate the expression on the
operands it is correct for, the fixture calls backward() with grad_out = rotate(G, theta) for a single stage:
the inverse rotation inside backward() then hands the expression G itself, and the result is d/dtheta of
L(theta) = sum(G * rotate(x * scale; theta)).

The fixture stores inputs and the reference's outputs; `rotation::rotate` is stubbed with the oracle's fp64
rotate (the only thing that can run here).  The test (tests/test_oracle_golden.py::test_g7_*) checks central
finite differences of the oracle's forward against the stored grad_theta -- which fails for an oracle with the
wrong orientation or pair layout -- and grad_x / grad_scale (inverse = transposed stages, scale handling).

Run in the build container only:  python tests/golden/make_golden_g7.py
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

        def _oracle_rotate(x, idx_ij, theta, scales=None, group_size=128):
            out = po.rotate(x.detach().double().numpy(), idx_ij.numpy(), theta.detach().double().numpy(),
                            None if scales is None else scales.detach().double().numpy(), int(group_size), mode="ideal")
            return torch.from_numpy(np.ascontiguousarray(out)).to(x.dtype)

        lib.impl("rotate", _oracle_rotate, "CPU")
        spec = importlib.util.spec_from_file_location("ref_autograd", src)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)

        rng = np.random.default_rng(77)
        K, B = 256, 3
        idx = torch.from_numpy(po.random_pairs(rng, 1, K))
        theta = torch.from_numpy(rng.standard_normal((1, K // 2)) * 0.6)
        x = torch.from_numpy(rng.standard_normal((B, K)))
        scale = torch.from_numpy(rng.uniform(0.5, 2.0, K))
        G = torch.from_numpy(rng.standard_normal((B, K)))

        def run(grad_out, with_scale):
            th = theta.clone().requires_grad_()
            xx = x.clone().requires_grad_()
            sc = scale.clone().requires_grad_() if with_scale else None
            y = ref.RotateTensorFunc.apply(xx, idx, th, sc, 128)
            y.backward(grad_out)
            return y.detach(), th.grad, xx.grad, None if sc is None else sc.grad

        # (1) the expression on the operands it is correct for: grad_out = rotate(G, theta), single stage
        g_rot = _oracle_rotate(G, idx, theta)
        y, gth, gx, gsc = run(g_rot, True)
        # (2) as shipped (plain grad_out): recorded to document the discrepancy; not asserted as a derivative
        _, gth_plain, gx_plain, gsc_plain = run(G, True)

        # sanity in the build container.  L(theta) = sum(G * rotate(x * scale; theta)):  (1) -- the expression fed G as
        # the stage-OUTPUT gradient -- is dL/dtheta; (2) -- backward() as shipped on grad_out = G -- is not.
        def loss(th_np):
            return float((po.rotate(x.numpy(), idx.numpy(), th_np, scale.numpy(), 128, mode="ideal") * G.numpy()).sum())
        th0 = theta.numpy()
        eps, worst1, worst2 = 1e-6, 0.0, 0.0
        for t in range(0, K // 2, 7):
            d = np.zeros_like(th0)
            d[0, t] = eps
            fd = (loss(th0 + d) - loss(th0 - d)) / (2 * eps)
            worst1 = max(worst1, abs(fd - float(gth[0, t])))
            worst2 = max(worst2, abs(fd - float(gth_plain[0, t])))
        assert worst1 < 1e-6, worst1
        print(f"G7: expression vs FD max err {worst1:.2e} (operands as derived); as shipped {worst2:.2e}")

        np.savez_compressed(os.path.join(HERE, "rotate_backward.npz"),
                            x=x.numpy(), idx=idx.numpy(), theta=theta.numpy(), scale=scale.numpy(), G=G.numpy(),
                            grad_out_rotated=g_rot.numpy(), y=y.numpy(),
                            grad_theta=gth.numpy(), grad_x=gx.numpy(), grad_scale=gsc.numpy(),
                            grad_theta_as_shipped=gth_plain.numpy(), grad_x_as_shipped=gx_plain.numpy(),
                            grad_scale_as_shipped=gsc_plain.numpy(), fd_err_as_shipped=np.float64(worst2))
        print("wrote", os.path.join(HERE, "rotate_backward.npz"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
