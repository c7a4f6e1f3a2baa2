// Probe for the compact rotation schedule (round 6): (1) does ds_bpermute_b32 use only address bits [7:2] (a stage word whose
// upper bits hold the angle can then be the address operand as it is), (2) accuracy of v_sin_f32 / v_cos_f32 on x = 1 + f
// (a 23-bit mantissa as the turn fraction: the word 0x3F800000 | f IS the float) against libm.
//   hipcc --offload-arch=gfx950 -O2 -o sincos_bperm_probe sincos_bperm_probe.hip && ./sincos_bperm_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>

__global__ void bperm_probe(const unsigned* addr, int* out) {
  const int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_ds_bpermute((int)addr[lane], lane * 1000 + 7);
}
__global__ void sincos_probe(const unsigned* words, float* s, float* c, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = __builtin_bit_cast(float, words[i]);
  s[i] = __builtin_amdgcn_sinf(x);
  c[i] = __builtin_amdgcn_cosf(x);
}

int main() {
  unsigned *d_addr; int* d_out;
  hipMalloc(&d_addr, 256); hipMalloc(&d_out, 256);
  unsigned addr[64]; int out[64];
  int bad = 0;
  for (int trial = 0; trial < 4; ++trial) {
    for (int l = 0; l < 64; ++l) {
      const unsigned src = (unsigned)((l * 37 + 11 + trial) & 63);
      const unsigned hi = trial == 0 ? 0u : (trial == 1 ? 0x3F800000u : (trial == 2 ? 0x3FFFFF00u : 0xBF812300u));
      addr[l] = hi | (src << 2) | (trial == 3 ? 3u : 0u);
    }
    hipMemcpy(d_addr, addr, 256, hipMemcpyHostToDevice);
    bperm_probe<<<1, 64>>>(d_addr, d_out);
    hipMemcpy(out, d_out, 256, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
      const int src = (int)((addr[l] >> 2) & 63);
      if (out[l] != src * 1000 + 7) ++bad;
    }
  }
  printf("{\"bpermute_upper_bits_ignored\": %s, \"mismatches\": %d", bad == 0 ? "true" : "false", bad);
  const int n = 1 << 20;
  std::vector<unsigned> w(n); std::vector<float> s(n), c(n);
  for (int i = 0; i < n; ++i) w[i] = 0x3F800000u | ((unsigned)i << 3) | (unsigned)(i & 7);
  unsigned* dw; float *ds, *dc;
  hipMalloc(&dw, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
  hipMemcpy(dw, w.data(), n * 4, hipMemcpyHostToDevice);
  sincos_probe<<<n / 256, 256>>>(dw, ds, dc, n);
  hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
  double es = 0, ec = 0, en = 0;
  for (int i = 0; i < n; ++i) {
    const double f = (double)(w[i] & 0x7FFFFFu) / 8388608.0;
    const double rs = sin(2 * M_PI * f), rc = cos(2 * M_PI * f);
    es = fmax(es, fabs(s[i] - rs)); ec = fmax(ec, fabs(c[i] - rc));
    en = fmax(en, fabs((double)s[i] * s[i] + (double)c[i] * c[i] - 1.0));
  }
  printf(", \"v_sin_max_abs_err\": %.3e, \"v_cos_max_abs_err\": %.3e, \"max_norm_dev\": %.3e}\n", es, ec, en);
  return 0;
}
