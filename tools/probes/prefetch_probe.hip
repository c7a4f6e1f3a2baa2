// Weight prefetch into the memory-side cache (256 MiB Infinity Cache): a tiny-footprint kernel that touches one
// dword per 128-byte line of up to PARO_MAX_PREFETCH buffers and discards the data.  Launched on a SIDE branch of
// the decode-step graph for the linears of a later layer, it turns the ~3 us per launch during which a dependent
// GEMV chain leaves HBM idle (kernel boundary, prologue, first-byte latency, tail) into useful traffic: when the
// dependent kernel starts, its weights are on die.  Nothing is written; results are identical with or without it.
#include "common.hpp"

namespace paro {

struct PrefetchArgs {
  const unsigned char* ptr[PARO_MAX_PREFETCH];
  long long lines[PARO_MAX_PREFETCH];   // 128-byte lines per buffer
  int n;
  unsigned* sink;
};

__global__ __launch_bounds__(256) void prefetch_kernel(const PrefetchArgs a) {
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  unsigned acc = 0;
  for (int b = 0; b < a.n; ++b) {
    const unsigned char* p = a.ptr[b];
    const long long n = a.lines[b];
    long long i = tid;
    for (; i + 7 * stride < n; i += 8 * stride) {
      unsigned v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load((const unsigned*)(p + (i + u * stride) * 128));
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    for (; i < n; i += stride) acc ^= __builtin_nontemporal_load((const unsigned*)(p + i * 128));
  }
  // keeps the loads alive; a dword XOR over random weights equal to this constant AND the sentinel pointer set
  // never happens in a real launch (sink is null unless a test asks for the checksum)
  if (a.sink) atomicXor(a.sink, acc);
}

}  // namespace paro

extern "C" int paro_prefetch(const void* const* ptrs, const int64_t* bytes, int n, int workgroups, void* checksum,
                             void* stream) {
  using namespace paro;
  if (n < 0 || n > PARO_MAX_PREFETCH) return fail(PARO_ERR_INVALID, "paro_prefetch takes 0..%d buffers (got %d)", PARO_MAX_PREFETCH, n);
  if (n == 0) return PARO_OK;
  if (!ptrs || !bytes) return fail(PARO_ERR_INVALID, "null pointer");
  if (workgroups < 1 || workgroups > 4096) return fail(PARO_ERR_INVALID, "workgroups must be in 1..4096");
  PrefetchArgs a;
  a.n = n;
  a.sink = (unsigned*)checksum;
  for (int i = 0; i < PARO_MAX_PREFETCH; ++i) {
    a.ptr[i] = nullptr;
    a.lines[i] = 0;
    if (i < n) {
      if (!ptrs[i] || bytes[i] < 0) return fail(PARO_ERR_INVALID, "bad buffer %d", i);
      if (((uintptr_t)ptrs[i] & 127) != 0) return fail(PARO_ERR_INVALID, "buffer %d is not 128-byte aligned", i);
      a.ptr[i] = (const unsigned char*)ptrs[i];
      a.lines[i] = bytes[i] / 128;      // a trailing partial line is left to the consumer
    }
  }
  hipLaunchKernelGGL(prefetch_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("paro_prefetch");
}
