/*
 * paro_cpu.c -- plain-C restatement of the reference hot path for the host CPU
 * (TEST INFRASTRUCTURE / CPU BASELINE ONLY -- never linked into or called by the product path).
 *
 * The reference (z-lab/paroquant v0.1.16) has no CPU implementation of this path: the rotation is
 * a CUDA-only kernel (paroquant/kernels/cuda/rotation.cu:133-135) and the INT4 matmul lives in
 * un-vendored third-party packages.  This file restates, in C, exactly what oracle/paro_oracle.py
 * restates in numpy (same citations), so that bench.py can time "the reference algorithm on the
 * host cores" (cpu_baseline.kind = "port") and tests can cross-check the two restatements:
 *
 *   paro_cpu_rotate_f16   rotation.cu:10-43 + rotation.cuh:91-173 (half path: half-precision scale
 *                         multiply :112-113, fp32 fmaf per stage re-rounded to half :143-153)
 *   paro_cpu_linear_f16   transformers/modules.py:57-71 / vllm/plugin.py:281-311:
 *                         per partition  y = rotate(x) @ fp16((q - z) * s)  with fp32 accumulation,
 *                         AWQ nibble order of cli/convert.py:19,149-155, then bias, rounded to fp16.
 *
 * Parity status: "parity unpinned" for the rotation outputs / matmul (no reference CPU path or
 * golden vectors exist); the packing / dequant conventions are pinned by tests/golden via the numpy
 * oracle, and tests/test_oracle_c.py checks this file against the numpy oracle.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#ifdef __F16C__
#include <immintrin.h>
#endif

/* ---- fp16 <-> fp32 (round-to-nearest-even) ------------------------------------------------ */
static inline float h2f(uint16_t h) {
#ifdef __F16C__
  return _cvtsh_ss(h);
#else
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu, u;
  if (exp == 0) {
    if (man == 0) {
      u = sign;
    } else {
      int e = -1;
      do { e++; man <<= 1; } while (!(man & 0x400u));
      u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
    }
  } else if (exp == 31) {
    u = sign | 0x7f800000u | (man << 13);
  } else {
    u = sign | ((exp + 112) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &u, 4);
  return f;
#endif
}

static inline uint16_t f2h(float f) {
#ifdef __F16C__
  return _cvtss_sh(f, 0);
#else
  uint32_t u;
  memcpy(&u, &f, 4);
  uint32_t sign = (u >> 16) & 0x8000u;
  int32_t exp = (int32_t)((u >> 23) & 0xff) - 127 + 15;
  uint32_t man = u & 0x7fffffu;
  if (((u >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
  if (exp >= 31) return (uint16_t)(sign | 0x7c00u);
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign;
    man |= 0x800000u;
    uint32_t shift = (uint32_t)(14 - exp);
    uint32_t hm = man >> shift, rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1))) hm++;
    return (uint16_t)(sign | hm);
  }
  uint32_t hm = man >> 13, rem = man & 0x1fffu;
  uint16_t out = (uint16_t)(sign | ((uint32_t)exp << 10) | hm);
  if (rem > 0x1000u || (rem == 0x1000u && (hm & 1))) out++;
  return out;
#endif
}

static inline float rh(float f) { return h2f(f2h(f)); } /* round through half */

int paro_cpu_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* Cap the OpenMP team (bench.py's cpu_baseline: one socket's physical cores -- measured on the 2 x 64-core
 * MI355X host: 64 threads 2.4 / 4.1 ms per layer (Qwen3-4B / Llama-3-8B), 128 threads 3.4 / 32 ms and
 * erratic from run to run). */
void paro_cpu_set_threads(int n) {
#ifdef _OPENMP
  if (n >= 1) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int paro_cpu_has_f16c(void) {
#ifdef __F16C__
  return 1;
#else
  return 0;
#endif
}

/* ---- rotation (half path) ----------------------------------------------------------------- */
/* x, out: fp16 bits [rows, hidden]; idx int16 [krot, hidden]; theta fp16 [krot, hidden/2];
 * scales fp16 [hidden] or NULL.  One (row, group) at a time, exactly the per-stage update of
 * rotation.cuh:143-153 with the pair/angle indexing of :126-127. */
void paro_cpu_rotate_f16(const uint16_t* x, uint16_t* out, const int16_t* idx, const uint16_t* theta,
                         const uint16_t* scales, int64_t rows, int64_t hidden, int krot, int gs) {
  const int64_t groups = hidden / gs;
  const int half = gs / 2;
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t r = 0; r < rows; ++r) {
    for (int64_t g = 0; g < groups; ++g) {
      float v[128];
      const uint16_t* xr = x + r * hidden + g * gs;
      for (int c = 0; c < gs; ++c) {
        float xv = h2f(xr[c]);
        if (scales) xv = rh(xv * h2f(scales[g * gs + c])); /* __hmul, rotation.cuh:112-113 */
        v[c] = xv;
      }
      for (int k = 0; k < krot; ++k) {
        const int16_t* ij = idx + (int64_t)k * hidden + g * gs;
        const uint16_t* th = theta + (int64_t)k * (hidden / 2) + g * half;
        for (int t = 0; t < half; ++t) {
          const int i = ij[2 * t], j = ij[2 * t + 1];
          const float a = h2f(th[t]);
          const float s = sinf(a), c = cosf(a);
          const float xi = v[i], xj = v[j];
          v[i] = rh(fmaf(c, xi, s * xj));  /* rotation.cuh:148-153 */
          v[j] = rh(fmaf(c, xj, -s * xi));
        }
      }
      uint16_t* o = out + r * hidden + g * gs;
      for (int c = 0; c < gs; ++c) o[c] = f2h(v[c]);
    }
  }
}

/* ---- fused operator ----------------------------------------------------------------------- */
static const int kAwqShift[8] = {0, 16, 4, 20, 8, 24, 12, 28}; /* column j of a word -> bit offset */

/* y[rows, N] (fp16) = per-partition rotate(x) @ fp16((q - z) * s) + bias, fp32 accumulation.
 * qweight int32 [K, N/8], qzeros int32 [K/128, N/8], scales fp16 [K/128, N] (AWQ layout);
 * pairs int16 [P, krot, K], theta fp16 [P, krot, K/2], cs fp16 [P, K]; part_cols sum to N. */
void paro_cpu_linear_f16(const uint16_t* x, uint16_t* y, int64_t rows, int64_t K, int64_t N, const int32_t* qweight,
                         const int32_t* qzeros, const uint16_t* scales, int nparts, const int32_t* part_cols,
                         const int16_t* pairs, const uint16_t* theta, const uint16_t* cs, const uint16_t* bias,
                         int krot) {
  const int64_t NW = N / 8;
  uint16_t* xrot = (uint16_t*)malloc((size_t)rows * K * 2);
  float* xf = (float*)malloc((size_t)rows * K * 4);
  int64_t col0 = 0;
  for (int p = 0; p < nparts; ++p) {
    const int64_t ncols = part_cols[p];
    paro_cpu_rotate_f16(x, xrot, pairs + (int64_t)p * krot * K, theta + (int64_t)p * krot * (K / 2), cs + (int64_t)p * K,
                        rows, K, krot, 128);
    for (int64_t i = 0; i < rows * K; ++i) xf[i] = h2f(xrot[i]);
    /* column chunks of 64 (8 packed words) per task: each task streams its words for every k */
    const int64_t nchunks = (ncols + 63) / 64;
#pragma omp parallel for schedule(static)
    for (int64_t ch = 0; ch < nchunks; ++ch) {
      const int64_t c_begin = col0 + ch * 64;
      const int64_t c_end = (c_begin + 64 < col0 + ncols) ? c_begin + 64 : col0 + ncols;
      const int nc = (int)(c_end - c_begin);
      float acc[16][64];
      float w[64], sc[64], zp[64];
      for (int64_t r0 = 0; r0 < rows; r0 += 16) {
        const int rb = (int)((rows - r0 < 16) ? rows - r0 : 16);
        for (int r = 0; r < rb; ++r)
          for (int c = 0; c < nc; ++c) acc[r][c] = 0.f;
        for (int64_t k = 0; k < K; ++k) {
          const int64_t g = k >> 7;
          if ((k & 127) == 0) {
            for (int c = 0; c < nc; ++c) {
              const int64_t col = c_begin + c;
              sc[c] = h2f(scales[g * N + col]);
              zp[c] = (float)((((uint32_t)qzeros[g * NW + (col >> 3)]) >> kAwqShift[col & 7]) & 0xFu);
            }
          }
          for (int c = 0; c < nc; ++c) {
            const int64_t col = c_begin + c;
            const float q = (float)((((uint32_t)qweight[k * NW + (col >> 3)]) >> kAwqShift[col & 7]) & 0xFu);
            w[c] = rh((q - zp[c]) * sc[c]); /* fp16 dequantised weight */
          }
          for (int r = 0; r < rb; ++r) {
            const float xv = xf[(r0 + r) * K + k];
            for (int c = 0; c < nc; ++c) acc[r][c] += xv * w[c];
          }
        }
        for (int r = 0; r < rb; ++r)
          for (int c = 0; c < nc; ++c) {
            float v = acc[r][c];
            if (bias) v = rh(v) + h2f(bias[c_begin + c]); /* torch.cat(...) + bias on fp16 tensors (plugin.py:308-311) */
            y[(r0 + r) * N + c_begin + c] = f2h(v);
          }
      }
    }
    col0 += ncols;
  }
  free(xf);
  free(xrot);
}
