// The decode chain without Python: the four quantised linears of `layers` decoder layers of a model, chained as in
// bench.py (attention / SiLU*mul replaced by views), through
//   A: paro_w4a16_gemv            (in-kernel rotation; what round 2 shipped)
//   B: paro_w4a16_gemv_chain      (pre-rotated activations, the consumer's rotation in the producer's epilogue)
// in one HIP graph each: us per layer, the two final outputs compared, then every shape alone (a graph of `reps`
// launches cycling the layers' weight sets, >= 1 GiB) with optional launch-shape overrides.
//     hipcc -O2 -o tools/chain_harness tools/chain_harness.cpp -ldl
//     tools/chain_harness paroquant_amd/_lib/libparo_mi355x.so [model] [layers] [rows] [silu]
//     CHAIN_SHAPE="qkv:5:4,o:8:4,gate_up:1:8,down:8:8" overrides (ksplit:waves) per linear; CHAIN_SWEEP=1 sweeps them.
// One JSON line per measurement.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>
#include "../include/paro_abi.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

struct Lib {
  void* h = nullptr;
  decltype(&paro_last_error) last_error;
  decltype(&paro_packed_qweight_bytes) packed_qweight_bytes;
  decltype(&paro_packed_sz_bytes) packed_sz_bytes;
  decltype(&paro_packed_rot_bytes) packed_rot_bytes;
  decltype(&paro_repack_awq) repack_awq;
  decltype(&paro_pack_rotation) pack_rotation;
  decltype(&paro_linear_workspace_bytes) linear_workspace_bytes;
  decltype(&paro_chain_workspace_bytes) chain_workspace_bytes;
  decltype(&paro_w4a16_gemv) w4a16_gemv;
  decltype(&paro_w4a16_gemv_chain) w4a16_gemv_chain;
  decltype(&paro_rotate_parts) rotate_parts;
  bool load(const char* path) {
    h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); return false; }
#define SYM(f, n) f = (decltype(f))dlsym(h, n); if (!f) { fprintf(stderr, "%s lacks %s\n", path, n); return false; }
    SYM(last_error, "paro_last_error") SYM(packed_qweight_bytes, "paro_packed_qweight_bytes") SYM(packed_sz_bytes, "paro_packed_sz_bytes")
    SYM(packed_rot_bytes, "paro_packed_rot_bytes") SYM(repack_awq, "paro_repack_awq") SYM(pack_rotation, "paro_pack_rotation")
    SYM(linear_workspace_bytes, "paro_linear_workspace_bytes") SYM(chain_workspace_bytes, "paro_chain_workspace_bytes")
    SYM(w4a16_gemv, "paro_w4a16_gemv") SYM(w4a16_gemv_chain, "paro_w4a16_gemv_chain") SYM(rotate_parts, "paro_rotate_parts")
#undef SYM
    return true;
  }
};

__global__ void fill_words(unsigned* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned v = (unsigned)i * 2654435761u + seed;
    v ^= v >> 15; v *= 2246822519u; v ^= v >> 13; v *= 3266489917u; v ^= v >> 16;
    p[i] = v;
  }
}
static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
static float h2f(unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

struct Shape { const char* name; int K; std::vector<int> parts; };
struct Rot { void *pairs, *theta, *cs, *rot; };   // one rotation parameter set per shape (shared by the layers: rotation bytes are small)
struct Lin { paro_linear_t L; };

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s lib.so [model] [layers] [rows] [silu]\n", argv[0]); return 1; }
  Lib A;
  if (!A.load(argv[1])) return 1;
  const std::string model = argc > 2 ? argv[2] : "qwen3-4b";
  int h, inter, q, kv;
  if (model == "llama3-8b") { h = 4096; inter = 14336; q = 4096; kv = 1024; }
  else if (model == "qwen3-0.6b") { h = 1024; inter = 3072; q = 2048; kv = 1024; }
  else if (model == "llama3-70b") { h = 8192; inter = 28672; q = 8192; kv = 1024; }
  else { h = 2560; inter = 9728; q = 4096; kv = 1024; }
  const std::vector<Shape> shapes = {{"qkv", h, {q, kv, kv}}, {"o", q, {h}}, {"gate_up", h, {inter, inter}}, {"down", inter, {h}}};
  size_t layer_bytes = 0;
  for (auto& s : shapes) { size_t N = 0; for (int c : s.parts) N += c; layer_bytes += (size_t)s.K * N / 2; }
  const int layers = argc > 3 && atoi(argv[3]) > 0 ? atoi(argv[3]) : (int)std::max<size_t>(2, ((size_t)1 << 30) / layer_bytes + 1);
  const int rows = argc > 4 ? atoi(argv[4]) : 1;
  const bool silu = argc > 5 && atoi(argv[5]) != 0;
  const int reps = 200;
  std::map<std::string, std::pair<int, int>> over;
  if (const char* e = getenv("CHAIN_SHAPE")) {
    std::string s(e);
    size_t pos = 0;
    while (pos < s.size()) {
      size_t c = s.find(',', pos); if (c == std::string::npos) c = s.size();
      std::string item = s.substr(pos, c - pos);
      size_t a = item.find(':'), b = item.find(':', a + 1);
      if (a != std::string::npos && b != std::string::npos) over[item.substr(0, a)] = {atoi(item.substr(a + 1, b - a - 1).c_str()), atoi(item.substr(b + 1).c_str())};
      pos = c + 1;
    }
  }
  hipStream_t st; HIP_OK(hipStreamCreate(&st));
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> u01(0.f, 1.f);
  std::normal_distribution<float> nrm(0.f, 1.f);
  int32_t* d_status; HIP_OK(hipMalloc(&d_status, 4));
  std::vector<Rot> rots(4);
  std::vector<std::vector<paro_linear_t>> lin(layers, std::vector<paro_linear_t>(4));
  int64_t ws_bytes = 1 << 20;
  for (int si = 0; si < 4; ++si) {
    const Shape& s = shapes[si];
    const int K = s.K, P = (int)s.parts.size(), G = K / 128;
    int N = 0; for (int c : s.parts) N += c;
    int32_t part_cols[PARO_MAX_PARTS] = {0};
    for (int i = 0; i < P; ++i) part_cols[i] = s.parts[i];
    std::vector<int16_t> pairs((size_t)P * 8 * K);
    std::vector<unsigned short> theta((size_t)P * 8 * K / 2), cs((size_t)P * K), sch((size_t)G * N);
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
    const float gain = 1.0f / (6.52f * std::sqrt((float)K) * std::sqrt(1.75f) * std::sqrt(13.0f / 12.0f));
    for (auto& v : sch) v = f2h((u01(rng) + 0.5f) * gain);
    Rot& R = rots[si];
    const int64_t rot_bytes = A.packed_rot_bytes(K, P);
    HIP_OK(hipMalloc(&R.pairs, pairs.size() * 2)); HIP_OK(hipMalloc(&R.theta, theta.size() * 2)); HIP_OK(hipMalloc(&R.cs, cs.size() * 2)); HIP_OK(hipMalloc(&R.rot, rot_bytes));
    HIP_OK(hipMemcpy(R.pairs, pairs.data(), pairs.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(R.theta, theta.data(), theta.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(R.cs, cs.data(), cs.size() * 2, hipMemcpyHostToDevice));
    if (A.pack_rotation((const int16_t*)R.pairs, R.theta, K, P, 8, R.rot, d_status, st) != PARO_OK) { fprintf(stderr, "pack_rotation: %s\n", A.last_error()); return 3; }
    void *d_qw, *d_qz, *d_sc;
    const size_t qw_words = (size_t)K * N / 8, qz_words = (size_t)G * N / 8;
    HIP_OK(hipMalloc(&d_qw, qw_words * 4)); HIP_OK(hipMalloc(&d_qz, qz_words * 4)); HIP_OK(hipMalloc(&d_sc, sch.size() * 2));
    HIP_OK(hipMemcpy(d_sc, sch.data(), sch.size() * 2, hipMemcpyHostToDevice));
    const int64_t wq_bytes = A.packed_qweight_bytes(K, N), sz_bytes = A.packed_sz_bytes(K, 128, P, part_cols);
    const int order = N / 16 >= 1024 ? 1 : 0;   // what PackedParoWeights picks
    for (int l = 0; l < layers; ++l) {
      paro_linear_t& L = lin[l][si];
      memset(&L, 0, sizeof(L));
      L.K = K; L.N = N; L.n_parts = P; L.krot = 8; L.act_dtype = PARO_DTYPE_F16; L.wq_order = order; L.group_size = 128;
      memcpy(L.part_cols, part_cols, sizeof(part_cols));
      L.rot = R.rot; L.pairs = (const int16_t*)R.pairs; L.theta = R.theta; L.channel_scales = R.cs;
      void *wq, *sz;
      HIP_OK(hipMalloc(&wq, wq_bytes)); HIP_OK(hipMalloc(&sz, sz_bytes));
      fill_words<<<1024, 256, 0, st>>>((unsigned*)d_qw, qw_words, 17u + 101u * (unsigned)(l * 4 + si));
      fill_words<<<256, 256, 0, st>>>((unsigned*)d_qz, qz_words, 91u + 977u * (unsigned)(l * 4 + si));
      if (A.repack_awq((const int32_t*)d_qw, (const int32_t*)d_qz, d_sc, K, N, 128, P, part_cols, order, wq, sz, st) != PARO_OK) { fprintf(stderr, "repack: %s\n", A.last_error()); return 3; }
      L.wq = wq; L.sz = sz;
      ws_bytes = std::max<int64_t>(ws_bytes, std::max(A.linear_workspace_bytes(&L, rows), A.chain_workspace_bytes(&L, rows)));
    }
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipFree(d_qw)); HIP_OK(hipFree(d_qz)); HIP_OK(hipFree(d_sc));
  }
  void* d_ws; HIP_OK(hipMalloc(&d_ws, ws_bytes)); HIP_OK(hipMemset(d_ws, 0, ws_bytes));
  const int Nqkv = q + 2 * kv;
  // activations
  std::vector<unsigned short> xh((size_t)rows * h);
  for (auto& v : xh) v = f2h(nrm(rng));
  void *d_x, *d_qkv, *d_o, *d_gu, *d_dn, *xr_qkv, *xr_o, *xr_gu, *xr_dn, *d_outA, *d_outB, *d_tmp;
  HIP_OK(hipMalloc(&d_x, (size_t)rows * h * 2)); HIP_OK(hipMemcpy(d_x, xh.data(), xh.size() * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMalloc(&d_qkv, (size_t)rows * Nqkv * 2)); HIP_OK(hipMalloc(&d_o, (size_t)rows * h * 2)); HIP_OK(hipMalloc(&d_gu, (size_t)rows * 2 * inter * 2));
  HIP_OK(hipMalloc(&d_dn, (size_t)rows * h * 2)); HIP_OK(hipMalloc(&d_outA, (size_t)rows * h * 2)); HIP_OK(hipMalloc(&d_outB, (size_t)rows * h * 2));
  HIP_OK(hipMalloc(&xr_qkv, (size_t)3 * rows * h * 2)); HIP_OK(hipMalloc(&xr_o, (size_t)rows * q * 2)); HIP_OK(hipMalloc(&xr_gu, (size_t)2 * rows * h * 2));
  HIP_OK(hipMalloc(&xr_dn, (size_t)rows * inter * 2)); HIP_OK(hipMalloc(&d_tmp, (size_t)rows * std::max(q, inter) * 2));
  auto chk = [&](int rc, const char* what) { if (rc != PARO_OK) { fprintf(stderr, "%s: %s\n", what, A.last_error()); exit(3); } };
  auto shape_of = [&](const char* name, int& ks, int& wv) { ks = 0; wv = 0; auto it = over.find(name); if (it != over.end()) { ks = it->second.first; wv = it->second.second; } };

  // ---- A: the fused GEMV chain (views: first q columns of qkv, first `inter` columns of gate_up; rows > 1 need a gather)
  auto strided_view = [&](void* src, int width, int take, void* dst) -> const void* {
    if (rows == 1) return src;
    HIP_OK(hipMemcpy2DAsync(dst, (size_t)take * 2, src, (size_t)width * 2, (size_t)take * 2, rows, hipMemcpyDeviceToDevice, st));
    return dst;
  };
  auto step_a = [&](void* out) {
    const void* hcur = d_x;
    for (int l = 0; l < layers; ++l) {
      chk(A.w4a16_gemv(&lin[l][0], hcur, d_qkv, rows, d_ws, ws_bytes, 0, 0, 0, -1, st), "gemv qkv");
      const void* a_in = strided_view(d_qkv, Nqkv, q, d_tmp);
      chk(A.w4a16_gemv(&lin[l][1], a_in, d_o, rows, d_ws, ws_bytes, 0, 0, 0, -1, st), "gemv o");
      chk(A.w4a16_gemv(&lin[l][2], d_o, d_gu, rows, d_ws, ws_bytes, 0, 0, 0, -1, st), "gemv gate_up");
      const void* d_in = strided_view(d_gu, 2 * inter, inter, d_tmp);
      void* y = l + 1 == layers ? out : d_dn;
      chk(A.w4a16_gemv(&lin[l][3], d_in, y, rows, d_ws, ws_bytes, 0, 0, 0, -1, st), "gemv down");
      hcur = y;
    }
  };
  // ---- B: the chain family
  auto chain_call = [&](const paro_linear_t* L, const void* xr, void* y, const paro_linear_t* next, void* nx, int64_t col0, int act, const char* name) {
    paro_chain_t C; memset(&C, 0, sizeof(C));
    C.x_rot = xr; C.y = y; C.next = next; C.next_x_rot = nx; C.next_col0 = col0; C.next_act = act;
    int ks, wv; shape_of(name, ks, wv);
    chk(A.w4a16_gemv_chain(L, &C, rows, d_ws, ws_bytes, ks, wv, st), name);
  };
  auto step_b = [&](void* out) {
    chk(A.rotate_parts(&lin[0][0], d_x, xr_qkv, rows, st), "rotate_parts");
    for (int l = 0; l < layers; ++l) {
      chain_call(&lin[l][0], xr_qkv, d_qkv, &lin[l][1], xr_o, 0, 0, "qkv");
      chain_call(&lin[l][1], xr_o, d_o, &lin[l][2], xr_gu, 0, 0, "o");
      chain_call(&lin[l][2], xr_gu, d_gu, &lin[l][3], xr_dn, 0, silu ? PARO_CHAIN_ACT_SILU_MUL : 0, "gate_up");
      const bool last = l + 1 == layers;
      chain_call(&lin[l][3], xr_dn, last ? out : d_dn, last ? nullptr : &lin[l + 1][0], last ? nullptr : xr_qkv, 0, 0, "down");
    }
  };
  auto graph_of = [&](auto fn) {
    hipGraph_t g; hipGraphExec_t ge;
    HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    fn();
    HIP_OK(hipStreamEndCapture(st, &g)); HIP_OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    HIP_OK(hipGraphLaunch(ge, st)); HIP_OK(hipStreamSynchronize(st));
    return ge;
  };
  hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  auto time_graph = [&](hipGraphExec_t ge) {
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      float ms;
      HIP_OK(hipEventRecord(e0, st)); HIP_OK(hipGraphLaunch(ge, st)); HIP_OK(hipEventRecord(e1, st)); HIP_OK(hipEventSynchronize(e1));
      HIP_OK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    return best;
  };
  if (!silu) {
    step_a(d_outA); step_b(d_outB); HIP_OK(hipStreamSynchronize(st));
    std::vector<unsigned short> ya((size_t)rows * h), yb((size_t)rows * h);
    HIP_OK(hipMemcpy(ya.data(), d_outA, ya.size() * 2, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(yb.data(), d_outB, yb.size() * 2, hipMemcpyDeviceToHost));
    double md = 0, mr = 0; bool nan = false;
    for (size_t i = 0; i < ya.size(); ++i) { const double a = h2f(ya[i]), b = h2f(yb[i]); if (a != a || b != b) nan = true; md = std::max(md, std::fabs(a - b)); mr = std::max(mr, std::fabs(a)); }
    hipGraphExec_t ga = graph_of([&] { step_a(d_outA); }), gb = graph_of([&] { step_b(d_outB); });
    const float ta = time_graph(ga), tb = time_graph(gb), ta2 = time_graph(ga), tb2 = time_graph(gb);
    printf("{\"model\": \"%s\", \"layers\": %d, \"rows\": %d, \"us_per_layer_fused\": %.3f, \"us_per_layer_chain\": %.3f, \"rel_diff_after_%d_layers\": %.3g, \"nan\": %s}\n",
           model.c_str(), layers, rows, std::min(ta, ta2) * 1e3f / layers, std::min(tb, tb2) * 1e3f / layers, layers, md / std::max(mr, 1e-30), nan ? "true" : "false");
    fflush(stdout);
  } else {
    hipGraphExec_t gb = graph_of([&] { step_b(d_outB); });
    printf("{\"model\": \"%s\", \"layers\": %d, \"rows\": %d, \"silu\": true, \"us_per_layer_chain\": %.3f}\n", model.c_str(), layers, rows, time_graph(gb) * 1e3f / layers);
  }
  // ---- every shape alone: fused vs chain (with its consumer's rotation in the epilogue), cycling the layers' weights
  const char* names[4] = {"qkv", "o", "gate_up", "down"};
  const void* xin[4] = {xr_qkv, xr_o, xr_gu, xr_dn};
  void* yout[4] = {d_qkv, d_o, d_gu, d_dn};
  void* nxo[4] = {xr_o, xr_gu, xr_dn, xr_qkv};
  const void* xplain[4] = {d_x, d_tmp, d_o, d_tmp};
  const bool sweep = getenv("CHAIN_SWEEP") != nullptr;
  for (int si = 0; si < 4; ++si) {
    const paro_linear_t& L0 = lin[0][si];
    const double bytes = (double)L0.K * L0.N / 2 + (double)(L0.K / 128) * L0.N * 2.5 + 2.0 * L0.K + 2.0 * L0.N + L0.n_parts * 26.0 * L0.K;
    hipGraphExec_t ga = graph_of([&] { for (int i = 0; i < reps; ++i) chk(A.w4a16_gemv(&lin[i % layers][si], xplain[si], yout[si], rows, d_ws, ws_bytes, 0, 0, 0, -1, st), "gemv"); });
    const float ta = time_graph(ga) * 1e3f / reps;
    std::vector<std::pair<int, int>> cfgs;
    if (sweep) { for (int ks : {1, 2, 3, 4, 5, 6, 8, 10, 12, 16}) for (int wv : {4, 8}) cfgs.push_back({ks, wv}); }
    else { int ks, wv; shape_of(names[si], ks, wv); cfgs.push_back({ks, wv}); }
    for (auto cfg : cfgs) {
      bool ok = true;
      auto body = [&] {
        for (int i = 0; i < reps && ok; ++i) {
          paro_chain_t C; memset(&C, 0, sizeof(C));
          C.x_rot = xin[si]; C.y = yout[si]; C.next = &lin[i % layers][(si + 1) % 4]; C.next_x_rot = nxo[si]; C.next_col0 = 0;
          C.next_act = (si == 2 && silu) ? PARO_CHAIN_ACT_SILU_MUL : 0;
          if (A.w4a16_gemv_chain(&lin[i % layers][si], &C, rows, d_ws, ws_bytes, cfg.first, cfg.second, st) != PARO_OK) ok = false;
        }
      };
      hipGraph_t g; hipGraphExec_t ge;
      HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      body();
      HIP_OK(hipStreamEndCapture(st, &g));
      if (!ok) { hipGraphDestroy(g); continue; }
      HIP_OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); HIP_OK(hipGraphLaunch(ge, st)); HIP_OK(hipStreamSynchronize(st));
      const float tb = time_graph(ge) * 1e3f / reps;
      printf("{\"model\": \"%s\", \"linear\": \"%s\", \"rows\": %d, \"K\": %lld, \"N\": %lld, \"ksplit\": %d, \"waves\": %d, \"us_fused\": %.3f, \"us_chain\": %.3f, \"frac_fused\": %.3f, \"frac_chain\": %.3f}\n",
             model.c_str(), names[si], rows, (long long)L0.K, (long long)L0.N, cfg.first, cfg.second, ta, tb, bytes / ta / 8e6, bytes / tb / 8e6);
      fflush(stdout);
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
  }
  // ---- diagnostic builds (make EXTRA=-DPARO_CHAIN_DIAG): per-workgroup phase stamps of ONE launch per shape
  typedef void (*set_dbg_fn)(void*);
  set_dbg_fn set_dbg = (set_dbg_fn)dlsym(A.h, "paro_chain_set_debug");
  if (set_dbg) {
    const size_t max_wg = 4096;
    unsigned long long* d_dbg; HIP_OK(hipMalloc(&d_dbg, max_wg * 16 * 8));
    std::vector<unsigned long long> hd(max_wg * 16);
    for (int si = 0; si < 4; ++si) {
      int ks, wv; shape_of(names[si], ks, wv);
      auto one = [&](int l) {
        paro_chain_t C; memset(&C, 0, sizeof(C));
        C.x_rot = xin[si]; C.y = yout[si]; C.next = &lin[l][(si + 1) % 4]; C.next_x_rot = nxo[si];
        C.next_act = (si == 2 && silu) ? PARO_CHAIN_ACT_SILU_MUL : 0;
        chk(A.w4a16_gemv_chain(&lin[l][si], &C, rows, d_ws, ws_bytes, ks, wv, st), "diag launch");
      };
      set_dbg(nullptr);
      for (int l = 0; l < layers; ++l) one(l);       // stream the other layers' weights through the caches
      HIP_OK(hipStreamSynchronize(st));
      HIP_OK(hipMemset(d_dbg, 0, max_wg * 16 * 8));
      set_dbg(d_dbg);
      one(0);
      HIP_OK(hipStreamSynchronize(st));
      set_dbg(nullptr);
      HIP_OK(hipMemcpy(hd.data(), d_dbg, max_wg * 16 * 8, hipMemcpyDeviceToHost));
      std::vector<std::vector<double>> ph[2];   // [owner?][phase] -> samples
      ph[0].assign(9, {}); ph[1].assign(9, {});
      unsigned long long rt_min = ~0ull, rt_max = 0, rt_last_entry = 0;
      int nwg = 0;
      for (size_t w = 0; w < max_wg; ++w) {
        const unsigned long long* t = &hd[w * 16];
        if (t[0] == 0) continue;
        ++nwg;
        const int own = (int)(t[9] >> 32) & 1;
        for (int k = 1; k <= 8; ++k) if (t[k]) ph[own][k].push_back((double)(t[k] - t[0]));
        rt_min = std::min(rt_min, t[10]); rt_max = std::max(rt_max, t[11]); rt_last_entry = std::max(rt_last_entry, t[10]);
      }
      auto med = [](std::vector<double>& v) { if (v.empty()) return -1.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
      auto mx = [](std::vector<double>& v) { if (v.empty()) return -1.0; return *std::max_element(v.begin(), v.end()); };
      printf("{\"timeline\": \"%s\", \"workgroups\": %d, \"entry_spread_us\": %.2f, \"first_entry_to_last_exit_us\": %.2f", names[si], nwg,
             (rt_last_entry - rt_min) * 0.01, (rt_max - rt_min) * 0.01);
      const char* pn[9] = {"", "loads_issued", "unit0_done", "units_done", "reduced", "polled", "stored", "rotated", "exit"};
      for (int own = 0; own < 2; ++own)
        for (int k = 1; k <= 8; ++k)
          if (!ph[own][k].empty()) printf(", \"%s_%s_cyc\": [%.0f, %.0f]", own ? "owner" : "producer", pn[k], med(ph[own][k]), mx(ph[own][k]));
      printf("}\n");
      fflush(stdout);
    }
  }
  return 0;
}
