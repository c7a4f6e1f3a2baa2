// A/B of two builds of libparo_mi355x.so WITHOUT Python: the same synthetic decode linears through both libraries'
// paro_w4a16_gemv (C ABI, dlopen), outputs compared BIT FOR BIT, and each library timed in a HIP graph of `reps` launches
// cycling >= `cycle_mib` MiB of distinct weights.  A run is a few seconds (no interpreter, no torch import), so a kernel
// experiment costs ~15 s of GPU box time instead of ~40:
//     make -C paroquant_amd/csrc -j OUT=$PWD/paroquant_amd/_lib_x EXTRA=-DSOME_EXPERIMENT    # the candidate
//     hipcc -O2 -o tools/ab_harness tools/ab_harness.cpp -ldl
//     gpurun -- tools/ab_harness paroquant_amd/_lib/libparo_mi355x.so paroquant_amd/_lib_x/libparo_mi355x.so [qwen3-4b]
// One JSON line per linear: {"linear", "K", "N", "us_a", "us_b", "identical": true|false, "max_abs_diff"}.
// Bit-identity against the VERIFIED build is the parity statement for a refactor that must not change results; anything
// that is meant to change them goes through tests/ and the oracle instead.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstdint>
#include <vector>
#include <string>
#include <random>
#include <algorithm>
#include "../include/paro_abi.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

struct Lib {
  void* h = nullptr;
  decltype(&paro_abi_version) abi_version;
  decltype(&paro_last_error) last_error;
  decltype(&paro_packed_qweight_bytes) packed_qweight_bytes;
  decltype(&paro_packed_sz_bytes) packed_sz_bytes;
  decltype(&paro_packed_rot_bytes) packed_rot_bytes;
  decltype(&paro_repack_awq) repack_awq;
  decltype(&paro_pack_rotation) pack_rotation;
  decltype(&paro_linear_workspace_bytes) linear_workspace_bytes;
  decltype(&paro_w4a16_gemv) w4a16_gemv;
  bool load(const char* path) {
    h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); return false; }
#define SYM(f, n) f = (decltype(f))dlsym(h, n); if (!f) { fprintf(stderr, "%s lacks %s\n", path, n); return false; }
    SYM(abi_version, "paro_abi_version") SYM(last_error, "paro_last_error") SYM(packed_qweight_bytes, "paro_packed_qweight_bytes")
    SYM(packed_sz_bytes, "paro_packed_sz_bytes") SYM(packed_rot_bytes, "paro_packed_rot_bytes") SYM(repack_awq, "paro_repack_awq")
    SYM(pack_rotation, "paro_pack_rotation") SYM(linear_workspace_bytes, "paro_linear_workspace_bytes") SYM(w4a16_gemv, "paro_w4a16_gemv")
#undef SYM
    return true;
  }
};

__global__ void fill_words(unsigned* p, size_t n, unsigned seed) {   // cheap hash: every bit pattern is a legal INT4 weight
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned v = (unsigned)i * 2654435761u + seed;
    v ^= v >> 15; v *= 2246822519u; v ^= v >> 13; v *= 3266489917u; v ^= v >> 16;
    p[i] = v;
  }
}

static unsigned short f2h(float f) {   // round-to-nearest-even float -> half bits (finite, in range)
  _Float16 h = (_Float16)f;
  unsigned short u; memcpy(&u, &h, 2); return u;
}

struct Shape { const char* name; int K; std::vector<int> parts; };

static std::vector<Shape> shapes_of(const std::string& model) {
  int h, inter, q, kv;
  if (model == "llama3-8b") { h = 4096; inter = 14336; q = 4096; kv = 1024; }
  else if (model == "qwen3-0.6b") { h = 1024; inter = 3072; q = 2048; kv = 1024; }
  else if (model == "llama3-70b") { h = 8192; inter = 28672; q = 8192; kv = 1024; }
  else { h = 2560; inter = 9728; q = 4096; kv = 1024; }   // qwen3-4b
  return {{"qkv_proj", h, {q, kv, kv}}, {"o_proj", q, {h}}, {"gate_up_proj", h, {inter, inter}}, {"down_proj", inter, {h}}};
}

struct Packed { void *wq, *sz; };

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s libA.so [libB.so] [model] [reps] [cycle_mib]\n", argv[0]); return 1; }
  Lib A, B;
  if (!A.load(argv[1])) return 1;
  const bool two = argc > 2 && strstr(argv[2], ".so");
  if (two && !B.load(argv[2])) return 1;
  const std::string model = argc > (two ? 3 : 2) ? argv[two ? 3 : 2] : "qwen3-4b";
  const int reps = argc > (two ? 4 : 3) ? atoi(argv[two ? 4 : 3]) : 200;
  const size_t cycle = (size_t)(argc > (two ? 5 : 4) ? atoi(argv[two ? 5 : 4]) : 1024) << 20;
  hipStream_t st; HIP_OK(hipStreamCreate(&st));
  std::mt19937 rng(1234);
  for (const Shape& s : shapes_of(model)) {
    if (getenv("AB_ONLY") && !strstr(getenv("AB_ONLY"), s.name)) continue;
    const int K = s.K, P = (int)s.parts.size(), G = K / 128;
    int N = 0; for (int c : s.parts) N += c;
    int32_t part_cols[PARO_MAX_PARTS] = {0};
    for (int i = 0; i < P; ++i) part_cols[i] = s.parts[i];
    const size_t qw_words = (size_t)K * N / 8, qz_words = (size_t)G * N / 8, sc_halves = (size_t)G * N;
    const int64_t wq_bytes = A.packed_qweight_bytes(K, N), sz_bytes = A.packed_sz_bytes(K, 128, P, part_cols), rot_bytes = A.packed_rot_bytes(K, P);
    const size_t per_copy = (size_t)wq_bytes + (size_t)sz_bytes;
    const int copies = (int)std::max<size_t>(2, std::min<size_t>(64, cycle / per_copy + 1));
    // rotation parameters: an independent random perfect matching per (partition, stage, group), small angles, scales in [0.5, 2)
    std::vector<int16_t> pairs((size_t)P * 8 * K);
    std::vector<unsigned short> theta((size_t)P * 8 * K / 2), cs((size_t)P * K), xh(K), sch(sc_halves);
    std::uniform_real_distribution<float> u01(0.f, 1.f);
    std::normal_distribution<float> nrm(0.f, 1.f);
    for (int p = 0; p < P; ++p)
      for (int r = 0; r < 8; ++r)
        for (int g = 0; g < G; ++g) {
          int16_t perm[128];
          for (int i = 0; i < 128; ++i) perm[i] = (int16_t)i;
          std::shuffle(perm, perm + 128, rng);
          memcpy(&pairs[((size_t)p * 8 + r) * K + (size_t)g * 128], perm, sizeof(perm));
        }
    for (auto& t : theta) t = f2h(0.1f * nrm(rng));
    for (auto& c : cs) c = f2h(0.5f + 1.5f * u01(rng));
    for (auto& v : xh) v = f2h(nrm(rng));
    const float gain = 1.0f / (6.52f * std::sqrt((float)K) * std::sqrt(1.75f) * std::sqrt(13.0f / 12.0f));
    for (auto& v : sch) v = f2h((u01(rng) + 0.5f) * gain);
    void *d_pairs, *d_theta, *d_cs, *d_x, *d_rot, *d_qw, *d_qz, *d_sc, *d_ws, *d_ya, *d_yb;
    int32_t* d_status;
    HIP_OK(hipMalloc(&d_pairs, pairs.size() * 2)); HIP_OK(hipMalloc(&d_theta, theta.size() * 2)); HIP_OK(hipMalloc(&d_cs, cs.size() * 2));
    HIP_OK(hipMalloc(&d_x, (size_t)K * 2 * 16)); HIP_OK(hipMemset(d_x, 0, (size_t)K * 2 * 16));   // 16 slabs: room for the PARO_XSLABS experiment
    HIP_OK(hipMalloc(&d_rot, rot_bytes)); HIP_OK(hipMalloc(&d_status, 4));
    HIP_OK(hipMalloc(&d_qw, qw_words * 4)); HIP_OK(hipMalloc(&d_qz, qz_words * 4)); HIP_OK(hipMalloc(&d_sc, sc_halves * 2));
    HIP_OK(hipMalloc(&d_ya, (size_t)N * 2)); HIP_OK(hipMalloc(&d_yb, (size_t)N * 2));
    HIP_OK(hipMemcpy(d_pairs, pairs.data(), pairs.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_theta, theta.data(), theta.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_cs, cs.data(), cs.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_x, xh.data(), (size_t)K * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_sc, sch.data(), sc_halves * 2, hipMemcpyHostToDevice));
    if (A.pack_rotation((const int16_t*)d_pairs, d_theta, K, P, 8, d_rot, d_status, st) != PARO_OK) { fprintf(stderr, "pack_rotation: %s\n", A.last_error()); return 3; }
    std::vector<Packed> pk(copies);
    for (int c = 0; c < copies; ++c) {
      fill_words<<<1024, 256, 0, st>>>((unsigned*)d_qw, qw_words, 17u + 101u * (unsigned)c);
      fill_words<<<256, 256, 0, st>>>((unsigned*)d_qz, qz_words, 91u + 977u * (unsigned)c);
      HIP_OK(hipMalloc(&pk[c].wq, wq_bytes)); HIP_OK(hipMalloc(&pk[c].sz, sz_bytes));
      if (A.repack_awq((const int32_t*)d_qw, (const int32_t*)d_qz, d_sc, K, N, 128, P, part_cols, 0, pk[c].wq, pk[c].sz, st) != PARO_OK) {
        fprintf(stderr, "repack_awq: %s\n", A.last_error()); return 3;
      }
    }
    HIP_OK(hipStreamSynchronize(st));
    int32_t status = 0; HIP_OK(hipMemcpy(&status, d_status, 4, hipMemcpyDeviceToHost));
    if (status) { fprintf(stderr, "illegal pair in the synthetic schedule\n"); return 3; }
    paro_linear_t L; memset(&L, 0, sizeof(L));
    L.K = K; L.N = N; L.n_parts = P; L.krot = 8; L.act_dtype = PARO_DTYPE_F16; L.wq_order = 0; L.group_size = 128;
    memcpy(L.part_cols, part_cols, sizeof(part_cols));
    L.rot = d_rot; L.pairs = (const int16_t*)d_pairs; L.theta = d_theta; L.channel_scales = d_cs;
    L.wq = pk[0].wq; L.sz = pk[0].sz;
    const int64_t ws_bytes = std::max<int64_t>(A.linear_workspace_bytes(&L, 1), two ? B.linear_workspace_bytes(&L, 1) : 0) + (1 << 20);
    HIP_OK(hipMalloc(&d_ws, ws_bytes)); HIP_OK(hipMemset(d_ws, 0, ws_bytes));
    auto launch = [&](Lib& lib, int c, void* y) {
      paro_linear_t l = L; l.wq = pk[c].wq; l.sz = pk[c].sz;
      static const int f_tpw = getenv("AB_TPW") ? atoi(getenv("AB_TPW")) : 0, f_ks = getenv("AB_KS") ? atoi(getenv("AB_KS")) : 0,
                       f_wv = getenv("AB_WV") ? atoi(getenv("AB_WV")) : 0;   // launch-shape overrides (0 = the library's choice)
      const int rc = lib.w4a16_gemv(&l, d_x, y, 1, d_ws, ws_bytes, f_tpw, f_ks, f_wv, -1, st);
      if (rc != PARO_OK) { fprintf(stderr, "gemv: %s\n", lib.last_error()); exit(3); }
    };
    // ---- bit identity of the two builds on every weight set
    bool identical = true; double maxdiff = 0.0;
    std::vector<unsigned short> ya(N), yb(N);
    for (int c = 0; c < copies && two; ++c) {
      launch(A, c, d_ya); launch(B, c, d_yb);
      HIP_OK(hipStreamSynchronize(st));
      HIP_OK(hipMemcpy(ya.data(), d_ya, (size_t)N * 2, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(yb.data(), d_yb, (size_t)N * 2, hipMemcpyDeviceToHost));
      if (memcmp(ya.data(), yb.data(), (size_t)N * 2) != 0) {
        identical = false;
        for (int i = 0; i < N; ++i) {
          _Float16 a, b; memcpy(&a, &ya[i], 2); memcpy(&b, &yb[i], 2);
          maxdiff = std::max(maxdiff, std::fabs((double)a - (double)b));
        }
      }
    }
    // ---- timing: a graph of `reps` launches per library, min of 5 replays, libraries interleaved
    auto build = [&](Lib& lib, void* y) {
      hipGraph_t g; hipGraphExec_t ge;
      HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < reps; ++i) launch(lib, i % copies, y);
      HIP_OK(hipStreamEndCapture(st, &g)); HIP_OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      HIP_OK(hipGraphLaunch(ge, st)); HIP_OK(hipStreamSynchronize(st));
      return ge;
    };
    hipGraphExec_t ga = build(A, d_ya), gb = two ? build(B, d_yb) : nullptr;
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    float best_a = 1e9f, best_b = 1e9f;
    for (int r = 0; r < 5; ++r) {
      float ms;
      HIP_OK(hipEventRecord(e0, st)); HIP_OK(hipGraphLaunch(ga, st)); HIP_OK(hipEventRecord(e1, st)); HIP_OK(hipEventSynchronize(e1));
      HIP_OK(hipEventElapsedTime(&ms, e0, e1)); best_a = std::min(best_a, ms);
      if (two) {
        HIP_OK(hipEventRecord(e0, st)); HIP_OK(hipGraphLaunch(gb, st)); HIP_OK(hipEventRecord(e1, st)); HIP_OK(hipEventSynchronize(e1));
        HIP_OK(hipEventElapsedTime(&ms, e0, e1)); best_b = std::min(best_b, ms);
      }
    }
    if (two)
      printf("{\"model\": \"%s\", \"linear\": \"%s\", \"K\": %d, \"N\": %d, \"weight_sets\": %d, \"us_a\": %.3f, \"us_b\": %.3f, \"identical\": %s, \"max_abs_diff\": %.3g}\n",
             model.c_str(), s.name, K, N, copies, best_a * 1e3f / reps, best_b * 1e3f / reps, identical ? "true" : "false", maxdiff);
    else
      printf("{\"model\": \"%s\", \"linear\": \"%s\", \"K\": %d, \"N\": %d, \"weight_sets\": %d, \"us_a\": %.3f}\n", model.c_str(), s.name, K, N, copies,
             best_a * 1e3f / reps);
    fflush(stdout);
    for (auto& p_ : pk) { HIP_OK(hipFree(p_.wq)); HIP_OK(hipFree(p_.sz)); }
    for (void* p_ : {d_pairs, d_theta, d_cs, d_x, d_rot, (void*)d_status, d_qw, d_qz, d_sc, d_ws, d_ya, d_yb}) HIP_OK(hipFree(p_));
  }
  return 0;
}
