// Launch floor of dependent kernels inside a HIP graph on MI355X, by grid and workgroup size.
//   hipcc --offload-arch=gfx950 -O3 -o launch_floor tools/micro/launch_floor.hip && ./launch_floor
// Each graph holds 400 back-to-back launches of a kernel that (a) returns at once, (b) loads one kernel
// argument block and returns, (c) additionally does one dependent global load per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

struct Args { const unsigned* p; int n; int pad[60]; };

__global__ void k_empty(Args a) {}
__global__ void k_args(Args a) { if (a.n == 12345) ((unsigned*)a.p)[0] = 1; }
__global__ void k_load(Args a) {
  const unsigned v = a.p[(blockIdx.x * blockDim.x + threadIdx.x) & 0xffff];
  if (v == 0xdeadbeef) ((unsigned*)a.p)[1] = v;
}

template <typename K>
float run(K kern, int grid, int block, const Args& a, hipStream_t st) {
  const int reps = 400;
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, st, a);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ts;
  for (int r = 0; r < 7; ++r) {
    hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms * 1000.f / reps);
  }
  std::sort(ts.begin(), ts.end());
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ts[3];
}

int main() {
  hipStream_t st; hipStreamCreate(&st);
  unsigned* buf; hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20);
  Args a{}; a.p = buf; a.n = 1;
  printf("%6s %6s %10s %10s %10s   (us per launch, median of 7 graph replays)\n", "grid", "block", "empty", "args", "args+load");
  for (int block : {256, 512, 1024})
    for (int grid : {64, 128, 256, 512, 1024})
      printf("%6d %6d %10.2f %10.2f %10.2f\n", grid, block, run(k_empty, grid, block, a, st), run(k_args, grid, block, a, st),
             run(k_load, grid, block, a, st));
  return 0;
}
