// A/B harness of the stream-K GEMM (csrc/gemm_sk.hip) against the 256-tile kernel it is meant to replace
// (codetr_linear_f16), through the C ABI of libcodetr_hip.so, on the Swin-L layer shapes of the headline workload.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/gemm_sk_bench.cpp -o tools/micro/_bin/gemm_sk_bench -ldl
//   tools/micro/_bin/gemm_sk_bench [--images 4|8] [--reps 20] [--quick]
// Every variant is checked against the old kernel's output on the same operands (bit-for-bit on whole tiles: both
// accumulate the 32-deep k-steps in the same order; split tiles differ in summation order only) and a sample of
// elements against an fp64 host sum.  Variants run interleaved, rounds in one process (same clocks, same cache state).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

typedef int (*linear_fn)(void*, const void*, const void*, const void*, const void*, const void*, void*, int64_t, int64_t,
                         int64_t, int, int64_t, int);
typedef int (*linear_sk_fn)(void*, const void*, const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int,
                            void*, int64_t, int);
typedef int (*linear_pp_fn)(void*, const void*, const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int,
                            int);
typedef int64_t (*ws_fn)(void);

static uint16_t f2h(float f) {
  _Float16 h = (_Float16)f;
  uint16_t b;
  memcpy(&b, &h, 2);
  return b;
}
static float h2f(uint16_t b) {
  _Float16 h;
  memcpy(&h, &b, 2);
  return (float)h;
}
static uint32_t rng_state = 12345u;
static float rnd() {  // uniform in [-1, 1)
  rng_state = rng_state * 1664525u + 1013904223u;
  return (float)((int32_t)rng_state) * (1.0f / 2147483648.0f);
}

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                \
    }                                                                         \
  } while (0)

struct Shape {
  const char* name;
  int64_t M, N, K;
  int act;
  bool res;
};

int main(int argc, char** argv) {
  int images = 4, reps = 20;
  bool quick = false, nocheck = false, streamk = false;
  const char* only = nullptr;
  std::vector<std::pair<std::string, std::string>> extra;   // name=path of further builds of gemm_sk.hip (diagnostic)
  const char* lib = "co-detr-tensorrt_amd/codetr/libcodetr_hip.so";
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--images")) images = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--quick")) quick = true;
    else if (!strcmp(argv[i], "--lib")) lib = argv[++i];
    else if (!strcmp(argv[i], "--nocheck")) nocheck = true;
    else if (!strcmp(argv[i], "--streamk")) streamk = true;
    else if (!strcmp(argv[i], "--only")) only = argv[++i];
    else if (!strcmp(argv[i], "--sklib")) {
      std::string a = argv[++i];
      const size_t eq = a.find('=');
      extra.push_back({a.substr(0, eq), a.substr(eq + 1)});
    }
  }
  void* h = dlopen(lib, RTLD_NOW);
  if (!h) {
    printf("dlopen failed: %s\n", dlerror());
    return 2;
  }
  linear_fn lin = (linear_fn)dlsym(h, "codetr_linear_f16");
  linear_sk_fn sk = (linear_sk_fn)dlsym(h, "codetr_linear_sk_f16");
  ws_fn wsb = (ws_fn)dlsym(h, "codetr_linear_sk_workspace_bytes");
  linear_pp_fn pp = (linear_pp_fn)dlsym(h, "codetr_linear_pp_f16");   // round 6: the ping-pong kernel (csrc/gemm_pp.hip)
  if (!lin || !sk || !wsb) {
    printf("missing symbol\n");
    return 2;
  }
  const int64_t s = images;  // rows scale with the images of a launch
  std::vector<Shape> shapes = {
      // correctness-first odd shapes (edge tiles, few tiles, short K)
      {"edge.a", 300, 200, 128, 0, false},
      {"edge.b", 1000, 520, 192, 1, true},
      {"edge.c", 5000, 264, 256, 2, false},
      {"edge.d", 257, 1544, 1024, 0, true},
      {"edge.e", 33000, 768, 128, 0, true},
      // Swin-L at 1920x1280 (tokens per image: 153600 / 38400 / 9600 / 2400; window-padded rows for qkv / proj)
      {"swin2.qkv", 10080 * s, 2304, 768, 0, false},
      {"swin2.proj", 10080 * s, 768, 768, 0, true},
      {"swin2.fc1", 9600 * s, 3072, 768, 2, false},
      {"swin2.fc2", 9600 * s, 768, 3072, 0, true},
      {"swin3.qkv", 2880 * s, 4608, 1536, 0, false},
      {"swin3.proj", 2880 * s, 1536, 1536, 0, true},
      {"swin3.fc1", 2400 * s, 6144, 1536, 2, false},
      {"swin3.fc2", 2400 * s, 1536, 6144, 0, true},
      {"swin1.qkv", 40320 * s, 1152, 384, 0, false},
      {"swin1.proj", 40320 * s, 384, 384, 0, true},
      {"swin1.fc1", 38400 * s, 1536, 384, 2, false},
      {"swin1.fc2", 38400 * s, 384, 1536, 0, true},
      {"swin0.fc2", 153600 * s, 192, 768, 0, true},
      // encoder (40920 tokens per image, 256 channels)
      {"enc.value", 40920 * s, 256, 256, 0, false},
      {"enc.out", 40920 * s, 256, 256, 0, true},
      // the X-stationary kernel's shapes (encoder at 204600 tokens per image, Swin stage 0 at 153600)
      {"xs.offlog", 204600 * s, 480, 256, 0, false},
      {"dec.vproj", 204600 * s, 1536, 256, 0, false},   // value projections of the six decoder layers as one GEMM
      {"xs.s0qkv", 153600 * s, 576, 192, 0, false},
      {"xs.s0proj", 153600 * s, 192, 192, 0, true},
      {"xs.s0fc1", 153600 * s, 768, 192, 2, false},
  };
  if (quick) shapes.resize(9);
  const int64_t ws_bytes = wsb();
  void* ws;
  CK(hipMalloc(&ws, ws_bytes));
  CK(hipMemset(ws, 0, ws_bytes));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  struct Variant {
    std::string name;
    int flags;   // -1: the old kernel
    linear_sk_fn fn;
    bool check;
    linear_pp_fn ppfn = nullptr;
  };
  std::vector<Variant> variants = {{"old256", -1, nullptr, false}, {"sk.default", 0, sk, true}};
  if (streamk) variants.push_back({"sk.streamk", 0x40, sk, true});
  if (pp) {
    variants.push_back({"pp", 0, nullptr, true, pp});
  }
  for (auto& e : extra) {   // name=path[:flags]: a further build of gemm_sk.hip, or of an experiment file with the pp entry point
    std::string path = e.second;
    int xflags = 0;
    const size_t colon = path.rfind(':');
    if (colon != std::string::npos) {
      xflags = atoi(path.c_str() + colon + 1);
      path = path.substr(0, colon);
    }
    void* h2 = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h2) {
      printf("dlopen %s failed: %s\n", path.c_str(), dlerror());
      return 2;
    }
    linear_pp_fn p2 = (linear_pp_fn)dlsym(h2, "codetr_linear_pp_f16");
    if (p2) {
      variants.push_back({e.first, xflags, nullptr, true, p2});
      continue;
    }
    linear_sk_fn f2 = (linear_sk_fn)dlsym(h2, "codetr_linear_sk_f16");
    variants.push_back({e.first, 0, f2, false});
  }

  for (const Shape& sh : shapes) {
    if (only && !strstr(sh.name, only)) continue;
    const int64_t M = sh.M, N = sh.N, K = sh.K;
    std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K), hb(N), hr(sh.res ? (size_t)M * N : 0);
    const float wsc = 1.0f / sqrtf((float)K);
    for (auto& v : hx) v = f2h(rnd() * 1.7f);
    for (auto& v : hw) v = f2h(rnd() * 1.7f * wsc);
    for (auto& v : hb) v = f2h(rnd());
    for (auto& v : hr) v = f2h(rnd());
    void *dx, *dw, *db, *dr = nullptr, *dy0, *dy1;
    CK(hipMalloc(&dx, hx.size() * 2));
    CK(hipMalloc(&dw, hw.size() * 2));
    CK(hipMalloc(&db, hb.size() * 2));
    if (sh.res) CK(hipMalloc(&dr, hr.size() * 2));
    CK(hipMalloc(&dy0, (size_t)M * N * 2));
    CK(hipMalloc(&dy1, (size_t)M * N * 2));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    if (sh.res) CK(hipMemcpy(dr, hr.data(), hr.size() * 2, hipMemcpyHostToDevice));

    auto run = [&](const Variant& v, void* y) {
      if (v.flags < 0) return lin(st, dx, dw, db, dr, nullptr, y, M, N, K, sh.act, 0, 0);
      if (v.ppfn) return v.ppfn(st, dx, dw, db, dr, y, M, N, K, sh.act, v.flags);
      return v.fn(st, dx, dw, db, dr, y, M, N, K, sh.act, ws, ws_bytes, v.flags);
    };
    // ---- correctness ----
    CK(hipMemset(dy0, 0xff, (size_t)M * N * 2));
    int rc = run(variants[0], dy0);
    CK(hipStreamSynchronize(st));
    if (rc) printf("%s: old kernel rc %d\n", sh.name, rc);
    std::vector<uint16_t> y0((size_t)M * N), y1((size_t)M * N);
    CK(hipMemcpy(y0.data(), dy0, y0.size() * 2, hipMemcpyDeviceToHost));
    // fp64 host sums of a sample
    double worst_ref = 0;
    for (int t = 0; t < 64; ++t) {
      const int64_t m = (int64_t)((rnd() * 0.5 + 0.5) * (M - 1)), n = (int64_t)((rnd() * 0.5 + 0.5) * (N - 1));
      double acc = 0;
      for (int64_t k = 0; k < K; ++k) acc += (double)h2f(hx[m * K + k]) * (double)h2f(hw[n * K + k]);
      acc += h2f(hb[n]);
      if (sh.act == 1) acc = acc < 0 ? 0 : acc;
      if (sh.act == 2) acc = 0.5 * acc * (1.0 + erf(acc / sqrt(2.0)));
      if (sh.res) acc = (double)h2f(f2h((float)acc)) + h2f(hr[m * N + n]);
      const double got = h2f(y0[m * N + n]);
      worst_ref = std::max(worst_ref, fabs(got - acc) / std::max(1.0, fabs(acc)));
    }
    std::string verdict;
    for (size_t vi = 1; vi < variants.size(); ++vi) {
      if (nocheck || !variants[vi].check) continue;
      CK(hipMemset(dy1, 0xee, (size_t)M * N * 2));
      rc = run(variants[vi], dy1);
      hipError_t se = hipStreamSynchronize(st);
      if (rc || se != hipSuccess) {
        printf("%s %s: rc %d sync %s\n", sh.name, variants[vi].name.c_str(), rc, hipGetErrorString(se));
        return 3;
      }
      CK(hipMemcpy(y1.data(), dy1, y1.size() * 2, hipMemcpyDeviceToHost));
      size_t diff = 0, bad = 0;
      double worst = 0;
      for (size_t i = 0; i < y0.size(); ++i) {
        if (y0[i] != y1[i]) {
          ++diff;
          const double a = h2f(y0[i]), b = h2f(y1[i]);
          const double d = fabs(a - b) / std::max(1.0, fabs(a));
          worst = std::max(worst, d);
          if (!(d <= 4e-3)) ++bad;
        }
      }
      char buf[160];
      snprintf(buf, sizeof buf, " | %s: %zu differ, %zu bad, worst %.2e", variants[vi].name.c_str(), diff, bad, worst);
      verdict += buf;
      // the counters must be back at zero
    }
    printf("%-11s M=%7lld N=%5lld K=%5lld act %d res %d  old-vs-fp64 %.1e%s\n", sh.name, (long long)M, (long long)N,
           (long long)K, sh.act, (int)sh.res, worst_ref, verdict.c_str());
    fflush(stdout);
    // ---- timing: interleaved rounds ----
    if (strncmp(sh.name, "edge", 4) != 0) {
      std::vector<std::vector<float>> t(variants.size());
      for (int round = 0; round < 3; ++round) {
        for (size_t vi = 0; vi < variants.size(); ++vi) {
          for (int w = 0; w < 2; ++w) run(variants[vi], dy1);
          CK(hipEventRecord(e0, st));
          for (int r = 0; r < reps; ++r) run(variants[vi], dy1);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          t[vi].push_back(ms * 1e3f / reps);
        }
      }
      printf("   us (median of 3 rounds x %d):", reps);
      const double fl = 2.0 * M * N * K;
      for (size_t vi = 0; vi < variants.size(); ++vi) {
        std::sort(t[vi].begin(), t[vi].end());
        printf("  %s %.1f (%.0f TF/s)", variants[vi].name.c_str(), t[vi][1], fl / t[vi][1] / 1e6);
      }
      printf("\n");
      fflush(stdout);
    }
    hipFree(dx); hipFree(dw); hipFree(db); if (dr) hipFree(dr); hipFree(dy0); hipFree(dy1);
  }
  // the workspace counters must all be zero again
  {
    std::vector<unsigned> c(4096);
    const int64_t off = ws_bytes - 4096;
    CK(hipMemcpy(c.data(), (char*)ws + off, 4096, hipMemcpyDeviceToHost));
    size_t nz = 0;
    for (size_t i = 0; i < 1024; ++i) nz += c[i] != 0;
    printf("non-zero ticket counters after all launches: %zu\n", nz);
  }
  return 0;
}
