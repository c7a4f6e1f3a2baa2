// One-shot all-reduce(SUM) of a small activation vector across the ranks of one node, for the row-parallel linears of
// tensor-parallel decode (SURVEY 8 row e: "decode 1 x hidden x 2 B -- latency-bound; consider ... a one-shot direct-write
// all-reduce over the peer links").  The reference leaves this collective to vLLM's RowParallelLinear; a 16 KiB RCCL
// all-reduce is tens of microseconds, 160 of them per 70B-class token.
//
// Every rank owns one buffer (mapped into all peers through HIP IPC by the caller):
//   [0]                     u32 epoch of the last completed call          [4] u32 status (0 / PARO_WS_STATUS_GIVEUP)
//   [256 + (set, r) * 64]   u32 flag: rank r's data of epoch e has landed in slot (set, r)            set = e & 1
//   [4096 + (set, r) * S]   slot: rank r's partial vector of that epoch
// One call = one launch of one workgroup:  write my vector into slot (set, me) of EVERY rank's buffer (plain stores over
// xGMI) -> system-scope release -> barrier -> store epoch into flag (set, me) of every rank -> spin on my own flags
// (bounded) -> system-scope acquire -> sum the world slots in rank order in fp32 (bit-identical on every rank) -> epoch.
// Two slot sets suffice: a rank cannot enter call e + 2 before every peer has flagged e + 1, which a peer only does once
// it has finished summing call e.  Nothing on the host side: the launch is HIP-graph capturable, the epoch lives in the
// buffer.
#include "common.hpp"

namespace paro {

constexpr int kArFlagOff = 256, kArDataOff = 4096, kArMaxWorld = 16;

struct ArArgs {
  const unsigned short* x;
  const unsigned short* residual;   // added to the sum (the decoder's residual stream), or null
  unsigned short* y;
  unsigned char* const* peers;   // device array [world]: every rank's buffer as mapped in THIS process
  int world, rank, n;
  long long slot_bytes;
  int spin_limit;
};

template <typename AT>
__global__ __launch_bounds__(1024) void allreduce_oneshot_kernel(const ArArgs a) {
  typedef Act<AT> A;
  const int tid = threadIdx.x;
  unsigned char* mine = a.peers[a.rank];
  const unsigned epoch = __hip_atomic_load((const unsigned*)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const int set = (int)(epoch & 1u);
  const int nv = a.n >> 3;   // 16-byte vectors
  // 1. my partial vector into slot (set, rank) of every rank
  for (int p = 0; p < a.world; ++p) {
    u32x4* dst = (u32x4*)(a.peers[p] + kArDataOff + (long long)(set * a.world + a.rank) * a.slot_bytes);
    for (int i = tid; i < nv; i += 1024) dst[i] = ((const u32x4*)a.x)[i];
  }
  // 2. every thread's stores are out system-wide before anyone raises a flag
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  __syncthreads();
  if (tid < a.world)
    __hip_atomic_store((unsigned*)(a.peers[tid] + kArFlagOff + (set * a.world + a.rank) * 64), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // 3. all ranks' data of this epoch has landed here
  if (tid < a.world) {
    const unsigned* f = (const unsigned*)(mine + kArFlagOff + (set * a.world + tid) * 64);
    // fast polls first (the common case: the peers are a few microseconds apart), then ~1 us naps: ranks may be far apart
    // once (a peer still capturing its graph while this one already replays), so the bound is seconds, not milliseconds
    int spin = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != epoch && spin < a.spin_limit) {
      if (spin < 4096) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(32);
      ++spin;
    }
    if (spin >= a.spin_limit) ((unsigned*)mine)[1] = PARO_WS_STATUS_GIVEUP;   // a peer never arrived: sticky, read by the host
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  // 4. sum in rank order (the same order on every rank: bit-identical results)
  for (int i = tid; i < nv; i += 1024) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < a.world; ++r) {
      const u32x4 v = *(const u32x4*)(mine + kArDataOff + (long long)(set * a.world + r) * a.slot_bytes + (long long)i * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[2 * e] += A::to_f32(v[e] & 0xffffu);
        acc[2 * e + 1] += A::to_f32(v[e] >> 16);
      }
    }
    if (a.residual) {
      const u32x4 rv = ((const u32x4*)a.residual)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[2 * e] += A::to_f32(rv[e] & 0xffffu);
        acc[2 * e + 1] += A::to_f32(rv[e] >> 16);
      }
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (unsigned)A::from_f32(acc[2 * e]) | ((unsigned)A::from_f32(acc[2 * e + 1]) << 16);
    ((u32x4*)a.y)[i] = o;
  }
  // 5. this call is complete (every thread read the epoch before the first barrier)
  if (tid == 0) __hip_atomic_store((unsigned*)mine, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace paro

extern "C" int64_t paro_allreduce_buffer_bytes(int world, int64_t max_elems) {
  if (world < 1 || world > paro::kArMaxWorld || max_elems < 8) return -1;
  const int64_t slot = ((max_elems * 2 + 255) / 256) * 256;
  return paro::kArDataOff + 2 * (int64_t)world * slot;
}

extern "C" int paro_allreduce_oneshot(const void* x, const void* residual, void* y, int64_t n, int act_dtype,
                                      const void* const* peers_dev, int world, int rank, int64_t max_elems, void* stream) {
  using namespace paro;
  if (!x || !y || !peers_dev) return fail(PARO_ERR_INVALID, "null pointer");
  if (world < 1 || world > kArMaxWorld || rank < 0 || rank >= world) return fail(PARO_ERR_INVALID, "bad world / rank (%d / %d)", world, rank);
  if (n < 8 || n % 8 != 0 || n > max_elems) return fail(PARO_ERR_INVALID, "element count must be a multiple of 8 in 8..%lld (got %lld)", (long long)max_elems, (long long)n);
  if (act_dtype != PARO_DTYPE_F16 && act_dtype != PARO_DTYPE_BF16) return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  ArArgs a;
  a.x = (const unsigned short*)x;
  a.residual = (const unsigned short*)residual;
  a.y = (unsigned short*)y;
  a.peers = (unsigned char* const*)peers_dev;
  a.world = world;
  a.rank = rank;
  a.n = (int)n;
  a.slot_bytes = ((max_elems * 2 + 255) / 256) * 256;
  a.spin_limit = 1 << 22;   // ~4 s of ~1 us naps (callers barrier before phases in which ranks can be seconds apart)
  hipStream_t st = (hipStream_t)stream;
  if (act_dtype == PARO_DTYPE_F16)
    hipLaunchKernelGGL(allreduce_oneshot_kernel<f16>, dim3(1), dim3(1024), 0, st, a);
  else
    hipLaunchKernelGGL(allreduce_oneshot_kernel<bf16>, dim3(1), dim3(1024), 0, st, a);
  return check_launch("paro_allreduce_oneshot");
}

// ---- the per-rank buffer: FINE-GRAINED device memory.  Peers write into it and this rank polls it inside one kernel:
// ordinary (coarse-grained) device memory is only coherent across agents at kernel boundaries -- a flag polled through
// this GPU's L2 could stay stale for ever -- so the buffer is allocated here with hipDeviceMallocFinegrained and shared
// through HIP IPC handles (64 opaque bytes the caller passes between its processes).  Setup-time calls: they allocate
// and synchronise, unlike everything on the hot path.
extern "C" int paro_allreduce_buffer_create(int64_t bytes, void** out_ptr, void* out_handle64) {
  using namespace paro;
  if (bytes < kArDataOff || !out_ptr || !out_handle64) return fail(PARO_ERR_INVALID, "bad arguments");
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) return fail(PARO_ERR_LAUNCH, "hipExtMallocWithFlags(finegrained, %lld bytes): %s", (long long)bytes, hipGetErrorString(e));
  e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) {
    (void)hipFree(p);
    return fail(PARO_ERR_LAUNCH, "all-reduce buffer setup: %s", hipGetErrorString(e));
  }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  __builtin_memcpy(out_handle64, &h, 64);
  *out_ptr = p;
  return PARO_OK;
}

extern "C" int paro_allreduce_buffer_open(const void* handle64, void** out_ptr) {
  using namespace paro;
  if (!handle64 || !out_ptr) return fail(PARO_ERR_INVALID, "null pointer");
  hipIpcMemHandle_t h;
  __builtin_memcpy(&h, handle64, 64);
  void* p = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return fail(PARO_ERR_LAUNCH, "hipIpcOpenMemHandle: %s", hipGetErrorString(e));
  // Probe the mapping before any kernel touches it (a kernel store through a dead mapping is a fatal memory fault; the host
  // sees an error code instead and the caller falls back to the library collective): the owning device must be peer-
  // accessible from the current one, and a 4-byte read of the (read-only here) status word must come back.
  int cur = -1;
  hipPointerAttribute_t attr;
  bool ok = hipGetDevice(&cur) == hipSuccess && hipPointerGetAttributes(&attr, p) == hipSuccess;
  if (ok && attr.device != cur) {
    int can = 0;
    ok = hipDeviceCanAccessPeer(&can, cur, attr.device) == hipSuccess && can != 0;
  }
  unsigned probe = 0;
  ok = ok && hipMemcpy(&probe, static_cast<const char*>(p) + 4, 4, hipMemcpyDeviceToHost) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    (void)hipIpcCloseMemHandle(p);
    return fail(PARO_ERR_LAUNCH, "peer buffer is not accessible from device %d", cur);
  }
  *out_ptr = p;
  return PARO_OK;
}

extern "C" int paro_allreduce_buffer_close(void* peer_ptr) {
  if (!peer_ptr) return PARO_OK;
  return hipIpcCloseMemHandle(peer_ptr) == hipSuccess ? PARO_OK : paro::fail(PARO_ERR_LAUNCH, "hipIpcCloseMemHandle failed");
}

extern "C" int paro_allreduce_buffer_destroy(void* own_ptr) {
  if (!own_ptr) return PARO_OK;
  return hipFree(own_ptr) == hipSuccess ? PARO_OK : paro::fail(PARO_ERR_LAUNCH, "hipFree failed");
}

// Diagnostic (synchronises `stream`): PARO_OK, or PARO_ERR_LAUNCH when a call gave up waiting for a peer.
extern "C" int paro_allreduce_status(const void* own_ptr, void* stream) {
  using namespace paro;
  if (!own_ptr) return fail(PARO_ERR_INVALID, "null pointer");
  unsigned w[2] = {0u, 0u};
  hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  if (e == hipSuccess) e = hipMemcpy(w, own_ptr, 8, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(PARO_ERR_LAUNCH, "all-reduce status read: %s", hipGetErrorString(e));
  if (w[1] != 0u) return fail(PARO_ERR_LAUNCH, "one-shot all-reduce gave up waiting for a peer (epoch %u)", w[0]);
  return PARO_OK;
}
