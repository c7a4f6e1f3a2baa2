// kernarg fetch latency at kernel start: s_memtime before / after the first use of the arguments
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(1024) void k(const unsigned* p0, const unsigned* p1, const unsigned* p2, const unsigned* p3, unsigned a0, unsigned a1,
                                          unsigned a2, unsigned a3, unsigned a4, unsigned a5, unsigned long long* out) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  // use every argument in scalar arithmetic (forces the argument fetch to complete)
  unsigned long long s = (unsigned long long)p0 + (unsigned long long)p1 + (unsigned long long)p2 + (unsigned long long)p3 + a0 + a1 + a2 + a3 + a4 + a5;
  unsigned long long t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "s"(s) : "memory");
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    out[(blockIdx.x * 16 + w) * 2] = t1 - t0;
    out[(blockIdx.x * 16 + w) * 2 + 1] = s;
  }
}
int main() {
  unsigned long long* out; hipMalloc(&out, 256 * 16 * 16);
  unsigned* d; hipMalloc(&d, 1024);
  hipStream_t st; hipStreamCreate(&st);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(out, 0, 256 * 16 * 16);
    if (mode == 0) {
      for (int i = 0; i < 5; ++i) k<<<256, 1024, 0, st>>>(d, d + 1, d + 2, d + 3, 1, 2, 3, 4, 5, 6, out);
    } else {
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
      for (int i = 0; i < 20; ++i) k<<<256, 1024, 0, st>>>(d, d + 1, d + 2, d + 3, 1, 2, 3, 4, 5, 6, out);
      hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      hipGraphLaunch(ge, st); hipGraphLaunch(ge, st);
      // the dependent-launch boundary with / without preload: 200 back-to-back (tiny) kernels per replay
      hipGraph_t g2; hipGraphExec_t ge2;
      hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
      for (int i = 0; i < 200; ++i) k<<<256, 1024, 0, st>>>(d, d + 1, d + 2, d + 3, 1, 2, 3, 4, 5, 6, out);
      hipStreamEndCapture(st, &g2); hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0);
      hipGraphLaunch(ge2, st); hipStreamSynchronize(st);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float best = 1e9f;
      for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, st); hipGraphLaunch(ge2, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("{\"probe\": \"launch_spacing\", \"build\": \"%s\", \"us_per_launch\": %.3f}\n", PRELOAD ? "preload" : "s_load", best * 1000.f / 200.f);
    }
    hipStreamSynchronize(st);
    std::vector<unsigned long long> h(256 * 16 * 2);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> v;
    for (int i = 0; i < 256 * 16; ++i) v.push_back(h[2 * i]);
    std::sort(v.begin(), v.end());
    printf("{\"probe\": \"kernarg_fetch\", \"build\": \"%s\", \"launch\": \"%s\", \"cycles_p5\": %llu, \"cycles_p50\": %llu, \"cycles_p95\": %llu, \"cycles_max\": %llu}\n",
           PRELOAD ? "preload" : "s_load", mode ? "graph" : "stream", v[v.size() / 20], v[v.size() / 2], v[v.size() * 19 / 20], v.back());
  }
  return 0;
}
