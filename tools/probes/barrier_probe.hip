// Grid-barrier cost probe (MI355X): persistent grid of NWG workgroups x 512 threads, ITER barriers back to back.
//   variant 0: one monotonically increasing counter (agent-scope fetch_add by thread 0, poll until >= (it+1)*NWG)
//   variant 1: per-workgroup flag stores (sc1) + every workgroup polls all flags with its first NWG threads
//   variant 2: counter as 0, plus `work` dependent global loads per thread between barriers (a tiny "step")
// build: hipcc --offload-arch=gfx950 -O3 -o barrier_probe barrier_probe.hip ; run: ./barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(512) void bar_counter(unsigned* ctr, int iters, int nwg, unsigned long long* out) {
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(it + 1) * (unsigned)nwg;
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = __builtin_readcyclecounter() - t0;
}

__global__ __launch_bounds__(512) void bar_flags(unsigned* flags, int iters, int nwg, unsigned long long* out) {
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x * 16, (unsigned)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < nwg) {
      while (__hip_atomic_load(flags + threadIdx.x * 16, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it + 1))
        __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
}

// two-level: workgroups of one XCD (blockIdx % 8) meet on an XCD counter, the last arriver of each XCD bumps the global one
__global__ __launch_bounds__(512) void bar_two_level(unsigned* ctr, int iters, int nwg, unsigned long long* out) {
  const int x = blockIdx.x & 7;
  const unsigned per_x = (unsigned)nwg / 8;
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(ctr + 64 + x * 64, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)(it + 1) * per_x - 1) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it + 1) * 8u) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
}

// counter barrier + a dependent "activation" read after it: every thread reads one word written before the barrier by another workgroup
__global__ __launch_bounds__(512) void bar_counter_data(unsigned* ctr, unsigned* data, int iters, int nwg, unsigned long long* out) {
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    // produce: this workgroup's 512 words
    __hip_atomic_store(data + (size_t)blockIdx.x * 512 + threadIdx.x, (unsigned)it + acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(it + 1) * (unsigned)nwg;
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    // consume: the neighbour workgroup's words (agent-scope load: must see the other XCD's store)
    const int nb = (blockIdx.x + 37) % nwg;
    acc += __hip_atomic_load(data + (size_t)nb * 512 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u;
  }
  if (acc == 0xffffffffu) out[1] = acc;
}

// relaxed counter: no per-poll acquire (one fence after the wait), longer sleep between polls
template <int SLEEP>
__global__ __launch_bounds__(512) void bar_relaxed(unsigned* ctr, int iters, int nwg, unsigned long long* out) {
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(it + 1) * (unsigned)nwg;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(SLEEP);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}
// flags + master: every workgroup stores its flag; workgroup 0 polls all flags (one per thread) and publishes a go word
__global__ __launch_bounds__(512) void bar_master(unsigned* flags, int iters, int nwg, unsigned long long* out) {
  unsigned* go = flags + 16 * 1024;
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    const unsigned v = (unsigned)(it + 1);
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(flags + blockIdx.x * 16, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (blockIdx.x == 0) {
      if (threadIdx.x < nwg)
        while (__hip_atomic_load(flags + threadIdx.x * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) __builtin_amdgcn_s_sleep(1);
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(go, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (threadIdx.x == 0) {
      while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) __builtin_amdgcn_s_sleep(2);
    }
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
  }
}
// one-directional hand-off, no barrier: every workgroup publishes 512 tagged 8-byte granules, then reads 512 granules of
// OTHER workgroups (spread over all of them) and spins until their tags show this iteration
__global__ __launch_bounds__(512) void handoff_granules(unsigned long long* gr, int iters, int nwg, unsigned long long* out) {
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned long long v = ((unsigned long long)(it + 1) << 32) | (acc & 1u);
    __hip_atomic_store(gr + (size_t)blockIdx.x * 512 + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int src = (threadIdx.x * 7 + blockIdx.x) % nwg;
    const unsigned long long* p = gr + (size_t)src * 512 + ((threadIdx.x + blockIdx.x) & 511);
    unsigned long long g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while ((unsigned)(g >> 32) < (unsigned)(it + 1)) {
      __builtin_amdgcn_s_sleep(1);
      g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    acc += (unsigned)g;
    __syncthreads();
  }
  if (acc == 0xffffffffu) out[1] = acc;
}

int main(int argc, char** argv) {
  const int iters = 2000;
  unsigned* ctr; unsigned* data; unsigned long long* out;
  hipMalloc(&ctr, 1 << 20); unsigned long long* gr; hipMalloc(&gr, 256 * 512 * 8); hipMalloc(&data, 256 * 512 * 4); hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nwg : {64, 128, 256}) {
    for (int variant = 0; variant < 9; ++variant) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(ctr, 0, 1 << 20); hipMemset(gr, 0, 256 * 512 * 8); hipMemset(data, 0, 256 * 512 * 4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (variant == 0) bar_counter<<<nwg, 512>>>(ctr, iters, nwg, out);
        if (variant == 1) bar_flags<<<nwg, 512>>>(ctr, iters, nwg, out);
        if (variant == 2) bar_two_level<<<nwg, 512>>>(ctr, iters, nwg, out);
        if (variant == 3) bar_counter_data<<<nwg, 512>>>(ctr, data, iters, nwg, out);
        if (variant == 4) bar_relaxed<1><<<nwg, 512>>>(ctr, iters, nwg, out);
        if (variant == 5) bar_relaxed<8><<<nwg, 512>>>(ctr, iters, nwg, out);
        if (variant == 6) bar_relaxed<32><<<nwg, 512>>>(ctr, iters, nwg, out);
        if (variant == 7) bar_master<<<nwg, 512>>>(ctr, iters, nwg, out);
        if (variant == 8) handoff_granules<<<nwg, 512>>>(gr, iters, nwg, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("{\"probe\": \"grid_barrier\", \"nwg\": %d, \"variant\": %d, \"us_per_barrier\": %.3f}\n", nwg, variant, best * 1000.f / iters);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { printf("error %s\n", hipGetErrorString(e)); return 1; }
  return 0;
}
