// Can consecutive launches of a dependent chain OVERLAP on this machine, with the dependency carried by memory instead of
// by the kernel boundary?  Chain of N kernels; workgroup w of kernel j waits until workgroup w of kernel j-1 has published
// its epoch (one write-through 8-byte granule, polled with sc1 loads), then publishes its own.  Launched
//   A  on ONE stream, no flags          : the plain dependent-kernel boundary (baseline)
//   B  on S streams round-robin, eager  : kernel j+1 is resident (and could prefetch) while kernel j runs
//   C  the same captured into ONE graph (fork at the start, join at the end), replayed
// Reports us per hop; a run that deadlocks gives up after a bounded spin and reports "gave_up".
//     hipcc --offload-arch=gfx950 -O2 -o tools/probes/overlap_probe tools/probes/overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

// flags[j][w]: granule {epoch}; cnt[j][w]: private epoch counter of (kernel j, workgroup w); work: a little dependent ALU
__global__ __launch_bounds__(256) void node(unsigned long long* flags, unsigned* cnt, unsigned* gaveup, int j, int wgs, int use_flags, int work) {
  const int w = blockIdx.x;
  __shared__ unsigned ep_s;
  if (threadIdx.x == 0) {
    const unsigned ep = cnt[j * wgs + w] + 1u;
    if (use_flags && j > 0) {
      const unsigned long long* src = flags + (size_t)(j - 1) * wgs + w;
      unsigned long long v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spin = 0;
      while ((unsigned)v != ep && spin < (1 << 13)) { __builtin_amdgcn_s_sleep(1); v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ++spin; }
      if ((unsigned)v != ep) atomicAdd(gaveup, 1u);
    }
    ep_s = ep;
  }
  __syncthreads();
  float x = (float)threadIdx.x;
  for (int i = 0; i < work; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
  if (x == 12345.678f) gaveup[1] = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (use_flags) __hip_atomic_store(flags + (size_t)j * wgs + w, (unsigned long long)ep_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    cnt[j * wgs + w] = ep_s;
  }
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 144, wgs = argc > 2 ? atoi(argv[2]) : 240, work = argc > 3 ? atoi(argv[3]) : 200;
  unsigned long long* flags; unsigned *cnt, *gaveup;
  HIP_OK(hipMalloc(&flags, (size_t)N * wgs * 8)); HIP_OK(hipMalloc(&cnt, (size_t)N * wgs * 4)); HIP_OK(hipMalloc(&gaveup, 8));
  HIP_OK(hipMemset(flags, 0, (size_t)N * wgs * 8)); HIP_OK(hipMemset(cnt, 0, (size_t)N * wgs * 4)); HIP_OK(hipMemset(gaveup, 0, 8));
  hipStream_t st[4];
  for (auto& s : st) HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1, fork, join[4];
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1)); HIP_OK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  for (auto& e : join) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  auto reset_flags = [&] { HIP_OK(hipDeviceSynchronize()); };
  auto report = [&](const char* mode, int S, float ms, int reps) {
    unsigned g[2]; HIP_OK(hipMemcpy(g, gaveup, 8, hipMemcpyDeviceToHost));
    printf("{\"mode\": \"%s\", \"streams\": %d, \"kernels\": %d, \"workgroups\": %d, \"us_per_hop\": %.3f, \"gave_up\": %u}\n", mode, S, N, wgs, ms * 1e3f / (N * reps), g[0]);
    fflush(stdout);
    HIP_OK(hipMemset(gaveup, 0, 8));
  };
  const int reps = 5;
  // ---- A: one stream, boundary dependencies only
  for (int graph = 0; graph < 2; ++graph) {
    hipGraphExec_t ge = nullptr;
    auto body = [&] { for (int j = 0; j < N; ++j) node<<<wgs, 256, 0, st[0]>>>(flags, cnt, gaveup, j, wgs, 0, work); };
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
    report(graph ? "boundary_graph" : "boundary_eager", 1, ms, reps);
  }
  // ---- B / C: S streams round-robin, flags carry the dependency
  for (int S = 2; S <= 4; ++S) {
    for (int graph = 0; graph < 2; ++graph) {
      reset_flags();
      hipGraphExec_t ge = nullptr;
      auto body = [&] {   // fork from st[0], round-robin, join into st[0]
        HIP_OK(hipEventRecord(fork, st[0]));
        for (int s = 1; s < S; ++s) HIP_OK(hipStreamWaitEvent(st[s], fork, 0));
        for (int j = 0; j < N; ++j) node<<<wgs, 256, 0, st[j % S]>>>(flags, cnt, gaveup, j, wgs, 1, work);
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
      report(graph ? "flags_graph" : "flags_eager", S, ms, reps);
    }
  }
  return 0;
}
