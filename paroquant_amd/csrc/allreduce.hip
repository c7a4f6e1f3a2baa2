// One-shot all-reduce(SUM) of a small activation vector across the ranks of one node, for the row-parallel linears of
// tensor-parallel decode (SURVEY 8 row e: "decode 1 x hidden x 2 B -- latency-bound; consider ... a one-shot direct-write
// all-reduce over the peer links").  The reference leaves this collective to vLLM's RowParallelLinear; a 16 KiB RCCL
// all-reduce is tens of microseconds, 160 of them per 70B-class token.
//
// Every rank owns one buffer (mapped into all peers through HIP IPC by the caller):
//   [4]                     u32 status (0 / PARO_WS_STATUS_GIVEUP)
//   [64 + 4 * wg]           u32 epoch of workgroup wg's last completed call
//   [4096 + (set, r) * S]   slot of rank r: 8-byte granules {two activations, u32 epoch tag}             set = epoch & 1
//   (behind it: the {fp32 partial, tag} region of the row-parallel GEMV's all-reduce EPILOGUE, which runs this same
//   exchange from the GEMV's output threads -- layout and words in common.hpp, code in gemv_impl.hpp)
// The data IS the flag (the K-split hand-off's recipe, gemv_impl.hpp, stretched over the links): a rank stores each pair
// of activations together with the call's epoch as ONE 8-byte system-scope store into its slot in every peer, and polls
// the granules of its own buffer until their tags read the epoch.  No fence, no flag round trip, no barrier on the data
// path: the latency of a call is one store flight plus the polls.  Every thread owns one granule, a launch is
// ceil(n / 2048) workgroups, each with its own epoch word (all ranks pass the same n, so the words agree across ranks).
// The world's values are summed in rank order in fp32 -- bit-identical on every rank.  Two slot sets suffice: a rank
// cannot enter call e + 2 before every peer's granules of e + 1 have arrived, which a peer only sends once it has
// finished summing call e.  Nothing on the host side: the launch is HIP-graph capturable, the epochs live in the buffer.
#include "common.hpp"

namespace paro {


struct ArArgs {
  const unsigned* x;             // two activations per word
  const unsigned* residual;      // added to the sum (the decoder's residual stream), or null
  unsigned* y;
  unsigned char* mine;           // == peers[rank]
  unsigned char* peers[kArMaxWorld];   // every rank's buffer as mapped in THIS process, by value (scalar registers, no fetch per peer)
  int world, rank, ng;           // ng = granules = n / 2
  long long slot_granules;
  int spin_limit;
};

template <typename AT>
__global__ __launch_bounds__(kArThreads) void allreduce_oneshot_kernel(const ArArgs a) {
  typedef Act<AT> A;
  const int tid = threadIdx.x;
  const int g = blockIdx.x * kArThreads + tid;
  unsigned char* mine = a.mine;
  unsigned* my_epoch = (unsigned*)(mine + kArEpochOff) + blockIdx.x;
  unsigned epoch = __hip_atomic_load(my_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  if (epoch == 0u) epoch = 2u;        // tag 0 is "never written"; 2 keeps the set parity alternating across the wrap
  // a buffer that has given up once stops waiting (one poll per call): the ranks are out of step for good, the host falls
  // back to the library collective, and until it does a token must not cost `spin_limit` naps per all-reduce
  const int spin_limit = __hip_atomic_load((const unsigned*)mine + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == PARO_WS_STATUS_GIVEUP ? 0 : a.spin_limit;
  const long long slot0 = (long long)((int)(epoch & 1u) * a.world) * a.slot_granules;
  if (g < a.ng) {
    const unsigned mydata = a.x[g];
    const unsigned long long gran = ((unsigned long long)epoch << 32) | mydata;
    // 1. my granule into slot (set, rank) of every peer: one 8-byte store each, data and tag together
    for (int p = 0; p < a.world; ++p) {
      if (p == a.rank) continue;
      unsigned long long* dst = (unsigned long long*)(a.peers[p] + kArDataOff) + slot0 + (long long)a.rank * a.slot_granules + g;
      __hip_atomic_store(dst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const unsigned res = a.residual ? a.residual[g] : 0u;
    // 2. the peers' granules of this epoch, 8 peers per round (one round for a world of up to 8): every poll of a round
    // is issued before any is looked at; fast polls first (the common case: the peers are a few microseconds apart), then
    // ~1 us naps -- ranks may be far apart once (a peer still capturing its graph while this one already replays), so the
    // bound is seconds, not milliseconds.  3. summed in rank order (the same order on every rank: bit-identical results).
    // (Compact on purpose: 16-way unrolled, mostly skipped blocks cost this launch-latency-bound kernel microseconds.)
    const unsigned long long* src = (const unsigned long long*)(mine + kArDataOff) + slot0 + g;
    constexpr int CH = 8;
    float lo = 0.f, hi = 0.f;
    bool gave_up = spin_limit == 0;
    for (int r0 = 0; r0 < a.world; r0 += CH) {
      unsigned long long got[CH];
      bool all = false;
      for (int spin = 0; !all; ++spin) {
#pragma unroll
        for (int q = 0; q < CH; ++q)
          if (r0 + q < a.world && r0 + q != a.rank) got[q] = __hip_atomic_load(src + (long long)(r0 + q) * a.slot_granules, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        all = true;
#pragma unroll
        for (int q = 0; q < CH; ++q)
          if (r0 + q < a.world && r0 + q != a.rank) all = all && (unsigned)(got[q] >> 32) == epoch;
        if (all) break;
        if (gave_up || spin >= spin_limit) {
          ((unsigned*)mine)[1] = PARO_WS_STATUS_GIVEUP;   // a peer never arrived: sticky, read by the host
          gave_up = true;
          break;
        }
        if (spin < 4096) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(32);
      }
#pragma unroll
      for (int q = 0; q < CH; ++q)
        if (r0 + q < a.world) {
          const unsigned v = r0 + q == a.rank ? mydata : (unsigned)got[q];
          lo += A::to_f32(v & 0xffffu);
          hi += A::to_f32(v >> 16);
        }
    }
    if (a.residual) {
      lo += A::to_f32(res & 0xffffu);
      hi += A::to_f32(res >> 16);
    }
    a.y[g] = (unsigned)A::from_f32(lo) | ((unsigned)A::from_f32(hi) << 16);
  }
  // 4. this workgroup's call is complete (every wave read the epoch word before it arrives here)
  __syncthreads();
  if (tid == 0) __hip_atomic_store(my_epoch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace paro

extern "C" int64_t paro_allreduce_buffer_bytes(int world, int64_t max_elems) {
  if (world < 1 || world > paro::kArMaxWorld || max_elems < 8 || max_elems > (int64_t)paro::kArMaxWgs * paro::kArThreads * 2) return -1;
  return paro::ar_buffer_bytes(world, max_elems);
}

extern "C" int paro_allreduce_oneshot(const void* x, const void* residual, void* y, int64_t n, int act_dtype,
                                      const void* const* peers, int world, int rank, int64_t max_elems, void* stream) {
  using namespace paro;
  if (!x || !y || !peers) return fail(PARO_ERR_INVALID, "null pointer");
  if (world < 1 || world > kArMaxWorld || rank < 0 || rank >= world) return fail(PARO_ERR_INVALID, "bad world / rank (%d / %d)", world, rank);
  if (max_elems > (int64_t)kArMaxWgs * kArThreads * 2) return fail(PARO_ERR_INVALID, "max_elems out of range");
  if (n < 8 || n % 8 != 0 || n > max_elems) return fail(PARO_ERR_INVALID, "element count must be a multiple of 8 in 8..%lld (got %lld)", (long long)max_elems, (long long)n);
  if (act_dtype != PARO_DTYPE_F16 && act_dtype != PARO_DTYPE_BF16) return fail(PARO_ERR_INVALID, "act_dtype must be f16 or bf16");
  ArArgs a;
  a.x = (const unsigned*)x;
  a.residual = (const unsigned*)residual;
  a.y = (unsigned*)y;
  for (int r = 0; r < kArMaxWorld; ++r) a.peers[r] = r < world ? (unsigned char*)peers[r] : nullptr;
  a.mine = a.peers[rank];
  if (!a.mine) return fail(PARO_ERR_INVALID, "peers[rank] is null");
  a.world = world;
  a.rank = rank;
  a.ng = (int)(n / 2);
  a.slot_granules = ar_slot_a(max_elems);
  a.spin_limit = 1 << 22;   // ~4 s of ~1 us naps (callers barrier before phases in which ranks can be seconds apart)
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((a.ng + kArThreads - 1) / kArThreads);
  if (act_dtype == PARO_DTYPE_F16)
    hipLaunchKernelGGL(allreduce_oneshot_kernel<f16>, grid, dim3(kArThreads), 0, st, a);
  else
    hipLaunchKernelGGL(allreduce_oneshot_kernel<bf16>, grid, dim3(kArThreads), 0, st, a);
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
