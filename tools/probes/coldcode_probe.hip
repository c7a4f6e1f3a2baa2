// What a block of code that is NOT executed costs at the tail of a launch-latency-bound kernel: NBLK unrolled blocks
// `if (p < world && p != rank) { ~25 instructions + a store }` with world = 1, rank = 0 (every block skipped by a uniform
// branch), the shape the first all-reduce epilogue of the GEMV had (16 peers x 3 loops x 2 call sites).  Reports the
// in-kernel cycles across the block sequence (s_memtime) and the launch spacing in a graph of 200 dependent launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int NBLK>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int world, int rank, unsigned long long* cyc) {
  const int tid = threadIdx.x;
  float v = in[tid];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("" : "+v"(v));
#pragma unroll
  for (int p = 0; p < NBLK; ++p) {
    if (p < world && p != rank) {
      float a = v * (float)(p + 1);
#pragma unroll
      for (int i = 0; i < 12; ++i) a = __builtin_fmaf(a, 1.0001f, (float)i);
      out[(size_t)(p + 1) * 4096 + blockIdx.x * 256 + tid] = a;
      v += a;
    }
  }
  asm volatile("" : "+v"(v));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + tid] = v;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NBLK>
void run(float* out, float* in, unsigned long long* cyc, hipStream_t st) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 200; ++i) k<NBLK><<<256, 256, 0, st>>>(out, in, 1, 0, cyc);
  hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("{\"probe\": \"cold_code\", \"skipped_blocks\": %d, \"us_per_launch\": %.3f, \"in_kernel_cycles_p50\": %llu, \"in_kernel_cycles_p95\": %llu}\n",
         NBLK, best * 1000.f / 200.f, h[128], h[243]);
}

int main() {
  float *out, *in; unsigned long long* cyc;
  hipMalloc(&out, (size_t)100 * 4096 * 4 + 256 * 256 * 4); hipMalloc(&in, 1024 * 4); hipMalloc(&cyc, 256 * 8);
  hipMemset(in, 0, 1024 * 4);
  hipStream_t st; hipStreamCreate(&st);
  run<0>(out, in, cyc, st);
  run<16>(out, in, cyc, st);
  run<48>(out, in, cyc, st);
  run<96>(out, in, cyc, st);
  run<0>(out, in, cyc, st);
  return 0;
}
