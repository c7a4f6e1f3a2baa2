// Probe (round 6): is __builtin_amdgcn_dispatch_id() a usable per-launch tag -- the same in every workgroup of a launch, different in
// consecutive launches of one stream, also when the launches are replayed from a HIP graph?
//   hipcc --offload-arch=gfx950 -O2 -o dispatch_id_probe dispatch_id_probe.hip && ./dispatch_id_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

extern "C" __device__ unsigned long long paro_dispatch_id(void) __asm("llvm.amdgcn.dispatch.id");
__global__ void k(unsigned long long* out, int slot) {
  const unsigned long long id = paro_dispatch_id();
  if (threadIdx.x == 0) {
    if (blockIdx.x == 0) out[slot * 2] = id;
    if (blockIdx.x == gridDim.x - 1) out[slot * 2 + 1] = id;
  }
}

int main() {
  unsigned long long* d; hipMalloc((void**)&d, 64 * 16);
  hipMemset(d, 0, 64 * 16);
  hipStream_t st; hipStreamCreate(&st);
  for (int i = 0; i < 3; ++i) k<<<300, 64, 0, st>>>(d, i);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 3; ++i) k<<<300, 64, 0, st>>>(d, 3 + i);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  unsigned long long h[64 * 2];
  printf("{\"eager\": [");
  hipStreamSynchronize(st);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 3; ++i) printf("[%llu, %llu]%s", h[2 * i], h[2 * i + 1], i < 2 ? ", " : "]");
  for (int r = 0; r < 3; ++r) {
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf(", \"replay%d\": [", r);
    for (int i = 3; i < 6; ++i) printf("[%llu, %llu]%s", h[2 * i], h[2 * i + 1], i < 5 ? ", " : "]");
  }
  printf("}\n");
  return 0;
}
