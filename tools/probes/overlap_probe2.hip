// Second overlap probe (round 3): what a dependent chain costs per hop when consecutive launches overlap on different streams and
// the dependency is carried by tagged granules, with the REAL shape of the decode chain's work:
//   kernel j = W workgroups x 512 threads; every wave requests `kb` KiB of distinct "weights" (16-byte nt loads held in
//   registers) at entry, THEN waits for its input granule from kernel j-1 (workgroup (w * 7 + j) % W: a different CU each hop),
//   consumes (xor-reduce of the loaded data + a short ALU chain), reduces across the waves through LDS and publishes.
// The epoch of a launch is read at entry together with the weight requests (not in front of the poll).
//   modes: boundary (one stream, no flags) | flags on S = 2, 3 streams, eager and captured into one graph
//     hipcc --offload-arch=gfx950 -O2 -o tools/probes/overlap_probe2 tools/probes/overlap_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NL>   // 16-byte loads per lane held in registers (NL * 1 KiB per wave)
__global__ __launch_bounds__(512) void node(const u32x4* __restrict__ weights, unsigned long long* flags, unsigned* cnt, unsigned* gaveup, unsigned* sink,
                                            int j, int W, int use_flags, size_t wstride) {
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ unsigned red[8];
  // weights of this (kernel, workgroup, wave): requested first
  const u32x4* base = weights + (size_t)j * wstride + ((size_t)(w * 8 + wave) * NL) * 64 + lane;
  u32x4 q[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) q[i] = __builtin_nontemporal_load(base + i * 64);
  unsigned ep = cnt[j * W + w] + 1u;    // private epoch of (kernel, workgroup)
  unsigned x = 1u;
  if (use_flags && j > 0) {
    const int src_wg = (w * 7 + j) % W;
    const unsigned long long* src = flags + (size_t)(j - 1) * W + src_wg;
    unsigned long long v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spin = 0;
    while ((unsigned)(v >> 32) != ep && spin < (1 << 13)) { __builtin_amdgcn_s_sleep(1); v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ++spin; }
    if ((unsigned)(v >> 32) != ep && lane == 0 && wave == 0) atomicAdd(gaveup, 1u);
    x = (unsigned)v | 1u;
  }
  unsigned acc = x;
#pragma unroll
  for (int i = 0; i < NL; ++i) acc ^= (q[i][0] * x) ^ q[i][1] ^ q[i][2] ^ q[i][3];
  for (int off = 32; off; off >>= 1) acc ^= __shfl_xor(acc, off);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (tid == 0) {
    unsigned r = 0;
    for (int i = 0; i < 8; ++i) r ^= red[i];
    if (use_flags) __hip_atomic_store(flags + (size_t)j * W + w, ((unsigned long long)ep << 32) | (r & 0xfffffffeu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else sink[w] = r;
    cnt[j * W + w] = ep;
  }
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 144, W = argc > 2 ? atoi(argv[2]) : 85, kb = argc > 3 ? atoi(argv[3]) : 8;
  const size_t wstride = (size_t)W * 8 * kb * 64;   // u32x4 per kernel
  u32x4* weights; unsigned long long* flags; unsigned *cnt, *gaveup, *sink;
  HIP_OK(hipMalloc(&weights, (size_t)N * wstride * 16)); HIP_OK(hipMemset(weights, 1, (size_t)N * wstride * 16));
  HIP_OK(hipMalloc(&flags, (size_t)N * W * 8)); HIP_OK(hipMalloc(&cnt, (size_t)N * W * 4)); HIP_OK(hipMalloc(&gaveup, 8)); HIP_OK(hipMalloc(&sink, W * 4));
  HIP_OK(hipMemset(flags, 0, (size_t)N * W * 8)); HIP_OK(hipMemset(cnt, 0, (size_t)N * W * 4)); HIP_OK(hipMemset(gaveup, 0, 8));
  hipStream_t st[4];
  for (auto& s : st) HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1, fork, join[4];
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1)); HIP_OK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  for (auto& e : join) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  auto launch = [&](int j, hipStream_t s, int use_flags) {
    if (kb == 2) node<2><<<W, 512, 0, s>>>(weights, flags, cnt, gaveup, sink, j, W, use_flags, wstride);
    else if (kb == 8) node<8><<<W, 512, 0, s>>>(weights, flags, cnt, gaveup, sink, j, W, use_flags, wstride);
    else node<16><<<W, 512, 0, s>>>(weights, flags, cnt, gaveup, sink, j, W, use_flags, wstride);
  };
  const int reps = 5;
  const double mb = (double)wstride * 16 / 1e6;
  auto report = [&](const char* mode, int S, float ms) {
    unsigned g[2]; HIP_OK(hipMemcpy(g, gaveup, 8, hipMemcpyDeviceToHost));
    printf("{\"mode\": \"%s\", \"streams\": %d, \"kernels\": %d, \"workgroups\": %d, \"MB_per_kernel\": %.2f, \"us_per_hop\": %.3f, \"TBps\": %.2f, \"gave_up\": %u}\n", mode, S, N, W, mb,
           ms * 1e3f / (N * reps), mb * N * reps / ms / 1e6, g[0]);
    fflush(stdout);
    HIP_OK(hipMemset(gaveup, 0, 8));
  };
  for (int S = 1; S <= 3; ++S)
    for (int graph = 0; graph < 2; ++graph) {
      HIP_OK(hipDeviceSynchronize());
      hipGraphExec_t ge = nullptr;
      auto body = [&] {
        if (S > 1) { HIP_OK(hipEventRecord(fork, st[0])); for (int s = 1; s < S; ++s) HIP_OK(hipStreamWaitEvent(st[s], fork, 0)); }
        for (int j = 0; j < N; ++j) launch(j, st[j % S], S > 1);
        for (int s = 1; s < S; ++s) { HIP_OK(hipEventRecord(join[s], st[s])); HIP_OK(hipStreamWaitEvent(st[0], join[s], 0)); }
      };
      if (graph) {
        hipGraph_t g;
        HIP_OK(hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal)); body(); HIP_OK(hipStreamEndCapture(st[0], &g));
        HIP_OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      }
      auto run = [&] { if (graph) HIP_OK(hipGraphLaunch(ge, st[0])); else body(); };
      run(); HIP_OK(hipStreamSynchronize(st[0]));
      HIP_OK(hipEventRecord(e0, st[0]));
      for (int r = 0; r < reps; ++r) run();
      HIP_OK(hipEventRecord(e1, st[0])); HIP_OK(hipEventSynchronize(e1));
      float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      report(S == 1 ? (graph ? "boundary_graph" : "boundary_eager") : (graph ? "flags_graph" : "flags_eager"), S, ms);
    }
  return 0;
}
