"""CPU oracle for the ParoQuant inference hot path (TEST INFRASTRUCTURE ONLY).

This module is a plain-numpy restatement of the reference algorithm
(z-lab/paroquant v0.1.16) for the path

    y = rotate(x * channel_scales; pairs, theta) @ dequant(qweight, qzeros, scales) (+ bias)

It is the *checker* for the HIP kernels in ``paroquant_amd/csrc``.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it; the product path never does (it fails loudly when
the HIP extension is missing).

Pinning status
--------------
* AWQ pack / unpack, the ``(q - z) * s`` dequant formula, the quantiser
  round/clamp convention, the ``pairs``/``theta`` kernel-data layout and the
  quantise-after-rotate export formula are PINNED against golden vectors
  captured by importing the reference's own Python
  (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
* The rotation's CONVENTION is pinned to reference-held code: orientation and pair layout of a stage by the analytic
  d/dtheta expression of ``RotateTensorFunc.backward`` (G7, ``tests/golden/make_golden_g7.py``), the ORDER of the stages by
  that function's stage-by-stage backward loop (G7b, ``make_golden_g7b.py``: ``<F(d), G> == <grad_x, d>`` holds for stage 0
  first and fails for the reversed order).
* What remains **parity unpinned** is the per-stage ROUNDING of the half-precision rotation and the INT4 matmul: the
  reference implements the rotation in CUDA only (``rotation.cu:133-135`` registers a CUDA-key kernel, nothing for CPU),
  ships no tests or golden vectors, and delegates the INT4 matmul to un-vendored third-party packages (AutoAWQ
  ``WQLinearMMFunction`` -- unpinned in ``pyproject.toml:30``; vLLM ``>=0.19.1,<0.20`` AWQ-Marlin; MLX
  ``quantized_matmul``).  Their arithmetic is restated here from the reference's own producer / consumer code and from the
  kernel source, each function citing the lines it follows.

Every function cites the reference ``file:line`` it restates (paths relative to
the reference root).
"""
from __future__ import annotations

import hashlib
import os
from collections import OrderedDict

import numpy as np

# ---------------------------------------------------------------------------
# Memo of the two expensive pure functions of the test suite (dequantising a whole weight matrix, generating a synthetic layer): the
# parity tests call both again and again with the same arguments (one layer, several row counts / launch shapes / repetitions).  Keyed
# by CONTENT (a digest of the packed arrays) resp. by the argument tuple; bounded (PARO_ORACLE_CACHE_MB, default 1536; 0 disables);
# cached arrays are read-only.  Results are exactly what the uncached functions return.
# ---------------------------------------------------------------------------
_CACHE_BUDGET = int(os.environ.get("PARO_ORACLE_CACHE_MB", "1536")) << 20
_CACHE: "OrderedDict[tuple, tuple]" = OrderedDict()      # key -> (value, bytes)
_cache_bytes = 0


def _digest(a) -> tuple:
    a = np.ascontiguousarray(a)
    return (a.shape, a.dtype.str, hashlib.blake2b(a.view(np.uint8).reshape(-1), digest_size=16).digest())


def _cache_get(key):
    hit = _CACHE.get(key)
    if hit is not None:
        _CACHE.move_to_end(key)
        return hit[0]
    return None


def _cache_put(key, value, nbytes: int):
    global _cache_bytes
    if nbytes > _CACHE_BUDGET // 2:
        return
    _CACHE[key] = (value, nbytes)
    _cache_bytes += nbytes
    while _cache_bytes > _CACHE_BUDGET:
        _, (_, b) = _CACHE.popitem(last=False)
        _cache_bytes -= b

# ---------------------------------------------------------------------------
# AWQ packing (paroquant/cli/convert.py:19,149-155; inverse at
# paroquant/inference/backends/mlx/load.py:18-24)
# ---------------------------------------------------------------------------

AWQ_REORDER = (0, 2, 4, 6, 1, 3, 5, 7)      # convert.py:19
AWQ_INV_REORDER = (0, 4, 1, 5, 2, 6, 3, 7)  # mlx/load.py:18
BITS = 4
PACK = 32 // BITS


def pack_awq(values: np.ndarray) -> np.ndarray:
    """``[R, C]`` ints in [0,15] -> ``int32[R, C/8]`` (convert.py:149-155).

    Nibble ``p`` (bits ``4p..4p+3``) of word ``c`` holds column
    ``8c + AWQ_REORDER[p]``.
    """
    v = np.asarray(values).astype(np.int64)
    assert v.ndim == 2 and v.shape[1] % PACK == 0
    r = v.reshape(v.shape[0], -1, PACK)[:, :, list(AWQ_REORDER)]
    packed = np.zeros(r.shape[:2], dtype=np.int64)
    for i in range(PACK):
        packed |= (r[:, :, i] & 0xF) << (BITS * i)
    return packed.astype(np.uint32).view(np.int32)


def unpack_awq(packed: np.ndarray) -> np.ndarray:
    """``int32[R, C/8]`` -> ``uint8[R, C]`` (mlx/load.py:21-24)."""
    p = np.asarray(packed).view(np.uint32).astype(np.int64)
    shifts = np.arange(0, 32, BITS, dtype=np.int64)
    raw = ((p[:, :, None] >> shifts) & 0xF).astype(np.uint8)
    return raw[:, :, list(AWQ_INV_REORDER)].reshape(p.shape[0], -1)


def dequant_awq(qweight, qzeros, scales, group_size: int = 128, out_dtype=np.float16):
    """``W[k, n] = (q[k, n] - z[k // gs, n]) * s[k // gs, n]``.

    Follows the producer ``convert.py:179-188`` (``q = clamp(round(w/s) + z)``)
    and the consumer-side inverse ``mlx/load.py:46-54`` (``biases = -s*z``).
    The subtraction is exact in integers; the product is rounded once to
    ``out_dtype`` (fp16 = what an fp16 ``(q - z) * s`` dequant kernel produces).
    """
    key = None
    if _CACHE_BUDGET:
        key = ("deq", _digest(qweight), _digest(qzeros), _digest(scales), int(group_size), np.dtype(out_dtype).str)
        hit = _cache_get(key)
        if hit is not None:
            return hit
    q = unpack_awq(qweight).astype(np.float32)          # [K, N]
    z = unpack_awq(qzeros).astype(np.float32)           # [K/gs, N]
    s = np.asarray(scales).astype(np.float32)           # [K/gs, N]
    K = q.shape[0]
    assert K % group_size == 0 and z.shape[0] == K // group_size == s.shape[0]
    zf = np.repeat(z, group_size, axis=0)
    sf = np.repeat(s, group_size, axis=0)
    w = ((q - zf) * sf).astype(out_dtype)
    if key is not None:
        w.flags.writeable = False
        _cache_put(key, w, w.nbytes)
    return w


# ---------------------------------------------------------------------------
# bf16 helpers (numpy has no bfloat16)
# ---------------------------------------------------------------------------

def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even float32 -> bfloat16 bit pattern (uint16)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    rounding = 0x7FFF + ((u >> 16) & 1)
    out = ((u + rounding) >> 16).astype(np.uint16)
    nan = np.isnan(np.asarray(x, dtype=np.float32))
    out[nan] = 0x7FC0
    return out


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.asarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def round_to(x: np.ndarray, dtype: str) -> np.ndarray:
    """Round a float array to ``dtype`` in {"f16","bf16","f32","f64"}; returns float64/32 values."""
    if dtype == "f16":
        return np.asarray(x).astype(np.float16).astype(np.float32)
    if dtype == "bf16":
        return bf16_bits_to_f32(f32_to_bf16_bits(np.asarray(x, dtype=np.float32)))
    if dtype == "f32":
        return np.asarray(x).astype(np.float32)
    if dtype == "f64":
        return np.asarray(x).astype(np.float64)
    raise ValueError(dtype)


# ---------------------------------------------------------------------------
# Pairwise Givens rotation (paroquant/kernels/cuda/rotation.cu:10-43,
# rotation.cuh:16-75 (float), :91-173 (half/bf16))
# ---------------------------------------------------------------------------

def check_rotation_args(hidden: int, idx_ij: np.ndarray, theta: np.ndarray, group_size: int):
    """Host-side validation of ``rotate_launcher`` / ``rotate_dynamic``.

    ``h % GROUP_SIZE == 0`` (rotation.cu:66), ``theta.size(0) == idx.size(0)``
    (rotation.cu:114), group_size in {64, 128} (rotation.cu:116-122).
    """
    if group_size not in (64, 128):
        raise RuntimeError(f"Unsupported group_size: {group_size}; expected 64 or 128")
    if hidden % group_size != 0:
        raise RuntimeError("h must be divisible by GROUP_SIZE")
    if idx_ij.shape[0] != theta.shape[0]:
        raise RuntimeError("theta.size(0) must equal idx_ij.size(0)")


def rotate(x, idx_ij, theta, scales=None, group_size: int = 128, mode: str = "f16"):
    """``out = (prod_r Givens_r)(x * scales)`` per ``group_size``-channel group.

    x:       [..., H] float array
    idx_ij:  int16[KROT, H]; for stage r, group g, pair t:
             ``(i, j) = idx[r, g*GS + 2t], idx[r, g*GS + 2t + 1]`` (group-local)
             (rotation.cuh:33-34,127; autograd.py:40-42)
    theta:   [KROT, H/2]; pair t of group g uses ``theta[r, g*GS/2 + t]``
             (rotation.cuh:32,126)
    scales:  [H] or [1, H] or None

    Stage update (rotation.cuh:53-56 / :143-153):
        xi' = c*xi + s*xj ;  xj' = c*xj - s*xi

    mode:
      "f16"/"bf16"  -- reference-faithful half path: theta/scales are first
                      cast to the activation dtype (rotation.cu:75-78), the
                      channel-scale multiply is a half-precision multiply
                      (rotation.cuh:112-113), each stage is computed in fp32 as
                      ``fmaf(c, xi, s*xj)`` and re-rounded to half
                      (rotation.cuh:143-153).
      "f32"         -- RotateAccess<float> (rotation.cuh:16-75): fp32 throughout.
      "ideal"       -- float64 throughout (bounds both of the above).
      "f16_once"/"bf16_once" -- inputs rounded to half, fp32 (here fp64) math,
                      ONE rounding at the end: what the HIP kernels compute.
    """
    x = np.asarray(x)
    H = x.shape[-1]
    idx_ij = np.asarray(idx_ij)
    theta = np.asarray(theta)
    check_rotation_args(H, idx_ij, theta, group_size)
    krot = idx_ij.shape[0]
    G = H // group_size
    half = group_size // 2
    rows = x.reshape(-1, H)

    base = mode.split("_")[0]
    once = mode.endswith("_once")
    if base in ("f16", "bf16"):
        act = base
        xv = round_to(rows, act).astype(np.float64)
        th = round_to(theta, act)                       # rotation.cu:75
        sc = None if scales is None else round_to(np.asarray(scales).reshape(-1), act)
    elif base == "f32":
        act = "f32"
        xv = rows.astype(np.float32).astype(np.float64)
        th = np.asarray(theta, dtype=np.float32)
        sc = None if scales is None else np.asarray(scales, dtype=np.float32).reshape(-1)
    elif base == "ideal":
        act = "f64"
        xv = rows.astype(np.float64)
        th = np.asarray(theta, dtype=np.float64)
        sc = None if scales is None else np.asarray(scales, dtype=np.float64).reshape(-1)
    else:
        raise ValueError(mode)

    if sc is not None:
        if act in ("f16", "bf16") and not once:
            xv = round_to(xv * sc[None, :], act).astype(np.float64)   # __hmul, rotation.cuh:112-113
        elif act == "f32":
            xv = (xv.astype(np.float32) * sc[None, :].astype(np.float32)).astype(np.float64)
        else:
            xv = xv * sc[None, :].astype(np.float64)

    offs = (np.arange(G) * group_size)[:, None]
    for r in range(krot):
        pr = idx_ij[r].astype(np.int64).reshape(G, group_size)
        ii = (pr[:, 0::2] + offs).reshape(-1)            # autograd.py:40-41
        jj = (pr[:, 1::2] + offs).reshape(-1)
        t = th[r].reshape(G, half).reshape(-1).astype(np.float64)
        if act in ("f16", "bf16", "f32"):
            c = np.cos(t).astype(np.float32)
            s = np.sin(t).astype(np.float32)
        else:
            c, s = np.cos(t), np.sin(t)
        xi = xv[:, ii]
        xj = xv[:, jj]
        if act in ("f16", "bf16") and not once:
            # fmaf(c, xi, s*xj): inner product rounded to fp32, fma rounded to fp32,
            # then the pair is rounded to half (rotation.cuh:148-153)
            p1 = (s.astype(np.float64) * xj).astype(np.float32).astype(np.float64)
            p2 = (-s.astype(np.float64) * xi).astype(np.float32).astype(np.float64)
            yi = (c.astype(np.float64) * xi + p1).astype(np.float32)
            yj = (c.astype(np.float64) * xj + p2).astype(np.float32)
            yi = round_to(yi, act).astype(np.float64)
            yj = round_to(yj, act).astype(np.float64)
        elif act == "f32":
            # xi*c + xj*s ; xi*(-s) + xj*c (rotation.cuh:55-56), fp32 ops
            yi = ((xi * c).astype(np.float32).astype(np.float64) + (xj * s).astype(np.float32)).astype(np.float32).astype(np.float64)
            yj = ((xi * (-s)).astype(np.float32).astype(np.float64) + (xj * c).astype(np.float32)).astype(np.float32).astype(np.float64)
        else:
            cc, ss = c.astype(np.float64), s.astype(np.float64)
            yi = cc * xi + ss * xj
            yj = cc * xj - ss * xi
        xv[:, ii] = yi
        xv[:, jj] = yj

    if act in ("f16", "bf16"):
        out = round_to(xv, act)
    elif act == "f32":
        out = xv.astype(np.float32)
    else:
        out = xv
    return out.reshape(x.shape)


def inverse_rotation_params(idx_ij, theta):
    """Inverse = stages flipped, angles negated (optim/qlinear.py:110-120)."""
    return np.flip(np.asarray(idx_ij), axis=0).copy(), -np.flip(np.asarray(theta), axis=0).copy()


def is_valid_pairing(idx_ij, group_size: int = 128) -> bool:
    """Every (stage, group) slice of ``idx_ij`` is a permutation of 0..GS-1
    (perfect matching; dummies carry theta = 0 -- optim/rotation.py:37-54)."""
    idx = np.asarray(idx_ij).astype(np.int64)
    krot, H = idx.shape
    g = idx.reshape(krot, H // group_size, group_size)
    return bool(np.all(np.sort(g, axis=-1) == np.arange(group_size)))


def random_pairs(rng: np.random.Generator, krot: int, hidden: int, group_size: int = 128) -> np.ndarray:
    """Synthetic ``pairs``: an independent ``randperm(GS)`` per (stage, group).

    Same contract as the dummy init ``optim/qlinear.py:51-53`` (any permutation);
    used for synthetic benches/tests.
    """
    G = hidden // group_size
    out = np.empty((krot, G, group_size), dtype=np.int16)
    for r in range(krot):
        for g in range(G):
            out[r, g] = rng.permutation(group_size).astype(np.int16)
    return out.reshape(krot, hidden)


# ---------------------------------------------------------------------------
# Quantiser pieces (paroquant/optim/quantizer.py:10-25,87-117;
# paroquant/cli/convert.py:158-203,239-277)
# ---------------------------------------------------------------------------

def calc_scales_and_zero_points(weight, group_size: int, qmax: int = 15):
    """optim/quantizer.py:10-25 (min/max affine init)."""
    x = np.asarray(weight, dtype=np.float32).reshape(-1, group_size)
    mn = x.min(axis=1, keepdims=True)
    mx = x.max(axis=1, keepdims=True)
    scale = np.maximum(mx - mn, np.float32(1e-5)) / np.float32(qmax)
    zp = mn / scale
    return scale.astype(np.float32), zp.astype(np.float32)


def pseudo_quantize(x, n_bits: int = 4, group_size: int = 128, scale=None, zero_point=None):
    """optim/quantizer.py:87-117: ``(clamp(round(x/s) + rz, 0, qmax) - rz) * s`` with
    ``rz = clamp(-round(zp), 0, qmax)``."""
    x = np.asarray(x, dtype=np.float32)
    qmax = 2 ** n_bits - 1
    if scale is None or zero_point is None:
        scale, zero_point = calc_scales_and_zero_points(x, group_size, qmax)
    scale = np.clip(np.asarray(scale, dtype=np.float32), 1e-5, 1e5)
    rz = np.clip(-np.round(np.asarray(zero_point, dtype=np.float32)), 0, qmax)
    d1, d2 = x.shape
    xr = x.reshape(-1, group_size)
    xi = np.clip(np.round(xr / scale) + rz, 0, qmax)
    return ((xi - rz) * scale).reshape(d1, d2).astype(np.float32)


def quantize_rotated_weight(weight, pairs, theta, channel_scales, scales_flat, zp_flat,
                            bits: int = 4, group_size: int = 128, rotate_mode: str = "f32"):
    """cli/convert.py:158-191.  ``weight`` is ``[N, K]`` (out, in); the rotation is
    applied along K to ``weight * channel_scales``; returns
    ``(quantized int32[N, K], scales_2d f32[N, K/gs], zeros_2d int32[N, K/gs])``."""
    w = np.asarray(weight, dtype=np.float32)
    N, K = w.shape
    cs = np.asarray(channel_scales, dtype=np.float32).reshape(1, K)
    rotated = rotate(w * cs, pairs, theta, None, group_size, mode=rotate_mode).astype(np.float32)
    qmax = (1 << bits) - 1
    sf = np.asarray(scales_flat, dtype=np.float32).reshape(-1, 1)
    zf = np.asarray(zp_flat, dtype=np.float32).reshape(-1, 1)
    zero_points = np.clip(-np.round(zf), 0, qmax)
    q = np.clip(np.round(rotated.reshape(-1, group_size) / sf) + zero_points, 0, qmax)
    q = q.astype(np.int32).reshape(N, K)
    G = K // group_size
    return q, sf.reshape(N, G).astype(np.float32), zero_points.reshape(N, G).astype(np.int32)


def to_awq_buffers(quantized, scales_2d, zeros_2d):
    """cli/convert.py:194-203: transposes to ``[K, N]`` then packs along N."""
    return {
        "qweight": pack_awq(np.ascontiguousarray(np.asarray(quantized).T)),
        "qzeros": pack_awq(np.ascontiguousarray(np.asarray(zeros_2d).T)),
        "scales": np.ascontiguousarray(np.asarray(scales_2d).T).astype(np.float16),
    }


# ---------------------------------------------------------------------------
# The fused operator (transformers/modules.py:57-71; vllm/plugin.py:281-311)
# ---------------------------------------------------------------------------

def paro_linear(x, qweight, qzeros, scales, theta, pairs, channel_scales, bias=None,
                group_size: int = 128, act: str = "f16", rotate_mode: str | None = None,
                ideal: bool = False):
    """``RotateQuantizedLinear.forward`` (transformers/modules.py:57-71):
    ``y = WQLinearMM(rotate(x, pairs, theta, channel_scales), qweight, qzeros, scales) + bias``.

    Reference CPU statement of "dequant-then-fp16-matmul" (BASELINE.json):
    W is dequantised to the activation dtype, the matmul accumulates in fp32
    (here fp64 over fp16/bf16-exact operands, i.e. at least fp32 accuracy) and
    the result is rounded once to the activation dtype.
    ``ideal=True`` keeps everything in float64 (no intermediate rounding).
    """
    x = np.asarray(x)
    K = x.shape[-1]
    if ideal:
        xr = rotate(x.astype(np.float64), pairs, np.asarray(theta, dtype=np.float64),
                    None if channel_scales is None else np.asarray(channel_scales, dtype=np.float64),
                    128, mode="ideal")
        w = dequant_awq(qweight, qzeros, scales, group_size, out_dtype=np.float64)
        y = xr.reshape(-1, K) @ w
        if bias is not None:
            y = y + np.asarray(bias, dtype=np.float64)[None, :]
        return y.reshape(*x.shape[:-1], w.shape[1])
    mode = rotate_mode or act
    xr = rotate(x, pairs, theta, channel_scales, 128, mode=mode)   # group_size NOT forwarded: modules.py:60
    key = None
    w = None
    if _CACHE_BUDGET:
        key = ("deq-rounded", _digest(qweight), _digest(qzeros), _digest(scales), int(group_size), act)
        w = _cache_get(key)
    if w is None:
        w = dequant_awq(qweight, qzeros, scales, group_size, out_dtype=np.float32)
        w = round_to(w, act).astype(np.float64)
        if key is not None:
            w.flags.writeable = False
            _cache_put(key, w, w.nbytes)
    y = xr.reshape(-1, K).astype(np.float64) @ w
    if bias is not None:
        y = y + round_to(np.asarray(bias, dtype=np.float32), act).astype(np.float64)[None, :]
    y = round_to(y.astype(np.float32), act)
    return y.reshape(*x.shape[:-1], w.shape[1])


def paro_linear_merged(x, qweight, qzeros, scales, theta, pairs, channel_scales, partition_sizes,
                       bias=None, group_size: int = 128, act: str = "f16", ideal: bool = False,
                       rotate_mode: str | None = None):
    """``ParoQuantLinearMethod.apply`` for merged projections (vllm/plugin.py:288-311):
    per partition ``i`` rotate ``x`` with that partition's params, multiply by that
    partition's column slice, concatenate, then add the bias.

    theta ``[P, krot, K/2]``, pairs ``[P, krot, K]``, channel_scales ``[P, 1, K]``;
    qweight ``[K, sum(N_i)/8]`` split by ``sizes // 8`` (plugin.py:261-263).
    """
    sizes = list(partition_sizes)
    outs = []
    col = 0
    for i, n in enumerate(sizes):
        qw = np.ascontiguousarray(qweight[:, col // PACK:(col + n) // PACK])
        qz = np.ascontiguousarray(qzeros[:, col // PACK:(col + n) // PACK])
        sc = np.ascontiguousarray(scales[:, col:col + n])
        outs.append(paro_linear(x, qw, qz, sc, theta[i], pairs[i], channel_scales[i], None,
                                group_size, act, rotate_mode, ideal))
        col += n
    y = np.concatenate(outs, axis=-1)
    if bias is not None:
        if ideal:
            y = y + np.asarray(bias, dtype=np.float64)
        else:
            y = round_to(y.astype(np.float64) + round_to(np.asarray(bias, dtype=np.float32), act), act)
    return y


def pseudo_weight(weight, pairs, theta, channel_scales, scales_flat, zp_flat,
                  bits: int = 4, group_size: int = 128):
    """``PseudoQuantizedLinear._pseudo_quantize`` (optim/qlinear.py:89-123) in float64:
    ``R^-1(deq(Q(R(W * cs)))) / cs`` -- the semantic bridge used by the end-to-end
    identity test K5 (``x @ pseudo_weight.T == paro_linear(x, pack(Q(R(W*cs))), 1/cs)``)."""
    w = np.asarray(weight, dtype=np.float64)
    N, K = w.shape
    cs = np.asarray(channel_scales, dtype=np.float64).reshape(1, K)
    rot = rotate(w * cs, pairs, np.asarray(theta, dtype=np.float64), None, group_size, mode="ideal")
    qmax = (1 << bits) - 1
    sf = np.asarray(scales_flat, dtype=np.float64).reshape(-1, 1)
    rz = np.clip(-np.round(np.asarray(zp_flat, dtype=np.float64).reshape(-1, 1)), 0, qmax)
    q = np.clip(np.round(rot.reshape(-1, group_size) / sf) + rz, 0, qmax)
    deq = ((q - rz) * sf).reshape(N, K)
    ip, it = inverse_rotation_params(pairs, theta)
    back = rotate(deq, ip, np.asarray(it, dtype=np.float64), None, group_size, mode="ideal")
    return back / cs


# ---------------------------------------------------------------------------
# Synthetic layer generator (SURVEY section 8d "Synthetic inputs")
# ---------------------------------------------------------------------------

def make_layer(seed: int, K: int, sizes, krot: int = 8, group_size: int = 128, bias: bool = False):
    """Seeded synthetic merged layer in the on-disk format of cli/convert.py:268-277.

    Returns a dict with qweight ``int32[K, N/8]``, qzeros ``int32[K/gs, N/8]``,
    scales ``f16[K/gs, N]`` and per-partition theta ``f16[P, krot, K/2]``, pairs
    ``int16[P, krot, K]``, channel_scales ``f16[P, 1, K]``.
    """
    sizes = list(sizes)
    key = ("layer", int(seed), int(K), tuple(int(v) for v in sizes), int(krot), int(group_size), bool(bias))
    if _CACHE_BUDGET:
        hit = _cache_get(key)
        if hit is not None:
            return dict(hit, sizes=list(sizes))      # a fresh dict over the same (never modified in place) arrays
    rng = np.random.default_rng(seed)
    N = int(sum(sizes))
    G = K // group_size
    q = rng.integers(0, 16, size=(K, N), dtype=np.int64)
    z = rng.integers(0, 16, size=(G, N), dtype=np.int64)
    s = rng.uniform(0.002, 0.02, size=(G, N)).astype(np.float16)
    P = len(sizes)
    theta = (rng.standard_normal((P, krot, K // 2)) * 0.1).astype(np.float16)
    pairs = np.stack([random_pairs(rng, krot, K, 128) for _ in range(P)])
    cs = rng.uniform(0.5, 2.0, size=(P, 1, K)).astype(np.float16)
    out = dict(qweight=pack_awq(q), qzeros=pack_awq(z), scales=s, theta=theta, pairs=pairs,
               channel_scales=cs, sizes=sizes, K=K, N=N)
    if bias:
        out["bias"] = (rng.standard_normal(N) * 0.1).astype(np.float16)
    if _CACHE_BUDGET:
        _cache_put(key, dict(out), sum(v.nbytes for v in out.values() if isinstance(v, np.ndarray)))
    return out


def rel_err(y, ref):
    """``max|y - ref| / max|ref|`` -- the BASELINE 1e-2 relative gate."""
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(y - ref)) / max(np.max(np.abs(ref)), 1e-30))


# ---------------------------------------------------------------------------------------------------------
# Decoder-layer pieces around the linears (decode harness).  The reference has no code for
# these: it runs the HF model's own modules (transformers/generator.py:37-67, `transformers>=4.55`,
# pyproject.toml:27); restated here in float64 from the published Llama / Qwen3 modelling code
# (LlamaRMSNorm, LlamaMLP `down(act(gate) * up)`, `apply_rotary_pos_emb` with rotate_half, grouped-query attention).
# ---------------------------------------------------------------------------------------------------------

def rmsnorm(x, weight=None, eps: float = 1e-6):
    x = np.asarray(x, dtype=np.float64)
    y = x / np.sqrt((x * x).mean(-1, keepdims=True) + eps)
    return y if weight is None else y * np.asarray(weight, dtype=np.float64)


def silu_mul(gate_up, inter: int):
    gu = np.asarray(gate_up, dtype=np.float64)
    g, u = gu[..., :inter], gu[..., inter:2 * inter]
    return g / (1.0 + np.exp(-g)) * u


def gelu_tanh_mul(gate_up, inter: int):
    """Gemma's MLP activation: gelu(gate, approximate="tanh") * up (HF `GemmaMLP`, hidden_activation gelu_pytorch_tanh)."""
    gu = np.asarray(gate_up, dtype=np.float64)
    g, u = gu[..., :inter], gu[..., inter:2 * inter]
    return 0.5 * g * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (g + 0.044715 * g ** 3))) * u


def rope_tables(head_dim: int, positions: int, theta: float = 10000.0):
    half = head_dim // 2
    inv = 1.0 / (theta ** (np.arange(half, dtype=np.float64) * 2.0 / head_dim))
    ang = np.arange(positions, dtype=np.float64)[:, None] * inv[None, :]
    return np.cos(ang), np.sin(ang)


def rope_rotate_half(x, cos, sin):
    """x [..., head_dim]; cos / sin [head_dim / 2] of the position: (x1, x2) -> (x1 c - x2 s, x2 c + x1 s)."""
    half = x.shape[-1] // 2
    x1, x2 = x[..., :half], x[..., half:]
    return np.concatenate([x1 * cos - x2 * sin, x2 * cos + x1 * sin], axis=-1)


def attention_decode(qkv, kcache, vcache, pos: int, n_heads: int, n_kv_heads: int, head_dim: int, cos, sin,
                     q_norm_w=None, k_norm_w=None, eps: float = 1e-6):
    """One token of grouped-query attention with a KV cache (float64).  qkv [(Hq + 2 Hkv) hd]; caches
    [Hkv, T, hd] hold positions < pos (position pos is written here).  Returns (out [Hq hd], k_new, v_new)."""
    qkv = np.asarray(qkv, dtype=np.float64)
    q = qkv[: n_heads * head_dim].reshape(n_heads, head_dim)
    k = qkv[n_heads * head_dim:(n_heads + n_kv_heads) * head_dim].reshape(n_kv_heads, head_dim)
    v = qkv[(n_heads + n_kv_heads) * head_dim:].reshape(n_kv_heads, head_dim)
    if q_norm_w is not None:
        q, k = rmsnorm(q, q_norm_w, eps), rmsnorm(k, k_norm_w, eps)
    q, k = rope_rotate_half(q, cos[pos], sin[pos]), rope_rotate_half(k, cos[pos], sin[pos])
    K = np.asarray(kcache, dtype=np.float64)[:, :pos + 1].copy()
    V = np.asarray(vcache, dtype=np.float64)[:, :pos + 1].copy()
    K[:, pos], V[:, pos] = k, v
    n_rep = n_heads // n_kv_heads
    out = np.empty((n_heads, head_dim))
    for h in range(n_heads):
        s = K[h // n_rep] @ q[h] / np.sqrt(head_dim)
        p = np.exp(s - s.max())
        out[h] = (p / p.sum()) @ V[h // n_rep]
    return out.reshape(-1), k, v


# ---------------------------------------------------------------------------------------------------------
# Mixture-of-experts.  Export: cli/convert.py:280-379 (`_quantize_moe`): gate_up [E, 2 I, H] and
# down [E, H, I] are rotated / quantised with ONE rotation per projection shared by all experts; the first half of
# the gate_up rows is gate_proj, the second half up_proj (:343-347); every expert gets its own AWQ buffers.
# Forward: mlx/modules.py:159-212 (`RotateSwitchGLU.__call__`): rotate x once, gate / up of the selected experts,
# activation(up, gate) = silu(gate) * up, rotate the activation with the down rotation, down of the same experts.
# ---------------------------------------------------------------------------------------------------------

def quantize_moe(gate_up, down, gu_pairs, gu_theta, gu_cs, gu_scale, gu_zp, dn_pairs, dn_theta, dn_cs, dn_scale, dn_zp,
                 bits: int = 4, group_size: int = 128):
    """Restates `_quantize_moe`: returns ({proj: {qweight, qzeros, scales} stacked over experts}, rotation buffers)."""
    E, two_i, H = gate_up.shape
    _, H2, I = down.shape
    assert H2 == H
    gq, gs, gz = quantize_rotated_weight(gate_up.reshape(-1, H), gu_pairs, gu_theta, gu_cs, gu_scale, gu_zp, bits, group_size)
    dq, ds, dz = quantize_rotated_weight(down.reshape(-1, I), dn_pairs, dn_theta, dn_cs, dn_scale, dn_zp, bits, group_size)
    half = two_i // 2
    gq, gs, gz = gq.reshape(E, two_i, H), gs.reshape(E, two_i, -1), gz.reshape(E, two_i, -1)
    dq, ds, dz = dq.reshape(E, H, I), ds.reshape(E, H, -1), dz.reshape(E, H, -1)
    out = {}
    for proj, q, sc, zp in (("gate_proj", gq[:, :half], gs[:, :half], gz[:, :half]),
                            ("up_proj", gq[:, half:], gs[:, half:], gz[:, half:]), ("down_proj", dq, ds, dz)):
        bufs = [to_awq_buffers(q[e], sc[e], zp[e]) for e in range(E)]
        out[proj] = {k: np.stack([b[k] for b in bufs]) for k in ("qweight", "qzeros", "scales")}
    rot = {"gate_up_weight_theta": np.asarray(gu_theta, dtype=np.float16), "gate_up_weight_pairs": np.asarray(gu_pairs, dtype=np.int16),
           "gate_up_weight_channel_scales": (1.0 / np.asarray(gu_cs, dtype=np.float32)).astype(np.float16).reshape(1, -1),
           "down_weight_theta": np.asarray(dn_theta, dtype=np.float16), "down_weight_pairs": np.asarray(dn_pairs, dtype=np.int16),
           "down_weight_channel_scales": (1.0 / np.asarray(dn_cs, dtype=np.float32)).astype(np.float16).reshape(1, -1)}
    return out, rot


def moe_experts_forward(x, indices, experts, rot, group_size: int = 128):
    """float64 forward of the routed experts: x [T, H], indices [T, k] -> [T, k, H] (no routing weights: the MoE
    block applies them outside, as with the reference's RotateSwitchGLU)."""
    x = np.asarray(x, dtype=np.float64)
    T, k = indices.shape
    xr = rotate(x, rot["gate_up_weight_pairs"], rot["gate_up_weight_theta"].astype(np.float64),
                rot["gate_up_weight_channel_scales"].astype(np.float64).reshape(-1), group_size, "ideal")
    deq = lambda proj, e: dequant_awq(experts[proj]["qweight"][e], experts[proj]["qzeros"][e], experts[proj]["scales"][e],
                                      group_size, np.float16).astype(np.float64)
    out = None
    for t in range(T):
        for s in range(k):
            e = int(indices[t, s])
            g, u = xr[t] @ deq("gate_proj", e), xr[t] @ deq("up_proj", e)
            act = (g / (1.0 + np.exp(-g)) * u)[None, :]
            ar = rotate(act, rot["down_weight_pairs"], rot["down_weight_theta"].astype(np.float64),
                        rot["down_weight_channel_scales"].astype(np.float64).reshape(-1), group_size, "ideal")
            y = ar[0] @ deq("down_proj", e)
            if out is None:
                out = np.zeros((T, k, y.shape[0]))
            out[t, s] = y
    return out


def make_moe(seed: int, E: int, H: int, I: int, krot: int = 8, group_size: int = 128):
    """Synthetic MoE expert block in checkpoint format (random INT4, shared rotations)."""
    rng = np.random.default_rng(seed)
    def proj(K, N):
        return {"qweight": np.stack([rng.integers(-2**31, 2**31 - 1, size=(K, N // 8), dtype=np.int64).astype(np.int32) for _ in range(E)]),
                "qzeros": np.stack([rng.integers(-2**31, 2**31 - 1, size=(K // group_size, N // 8), dtype=np.int64).astype(np.int32) for _ in range(E)]),
                "scales": np.stack([rng.uniform(0.002, 0.02, size=(K // group_size, N)).astype(np.float16) for _ in range(E)])}
    experts = {"gate_proj": proj(H, I), "up_proj": proj(H, I), "down_proj": proj(I, H)}
    rot = {"gate_up_weight_theta": (rng.standard_normal((krot, H // 2)) * 0.1).astype(np.float16),
           "gate_up_weight_pairs": random_pairs(rng, krot, H, group_size),
           "gate_up_weight_channel_scales": rng.uniform(0.5, 2.0, size=(1, H)).astype(np.float16),
           "down_weight_theta": (rng.standard_normal((krot, I // 2)) * 0.1).astype(np.float16),
           "down_weight_pairs": random_pairs(rng, krot, I, group_size),
           "down_weight_channel_scales": rng.uniform(0.5, 2.0, size=(1, I)).astype(np.float16)}
    return experts, rot
