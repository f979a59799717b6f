// codetr_runner -- Python-free execution of a Co-DETR launch plan on MI355X.
//
// The counterpart of the reference's C++ runner (codetr_inference.cpp:322-438: deserialise an engine, bind five I/O
// tensors, enqueue, time N iterations, copy the detections back) for this build's "engine": a launch plan written by
// codetr/export.py -- the list of libcodetr_hip.so entry points one fp16 forward calls, their arguments, and the device
// memory layout they run in.  No Python, no PyTorch, no tracing compiler: dlopen the kernel library, hipMalloc the
// plan's segments, upload weights / constants, then replay the launches on one HIP stream -- eagerly, or captured once
// into a hipGraph and launched per image.
//
//   codetr_runner --plan model.plan [--lib libcodetr_hip.so] [--iters N] [--no-graph]
//                 [--input batch_inputs=file.bin] [--input img_masks=file.bin] [--out-dir DIR]
//
// Inputs default to the tensors of the recorded run (stored in the plan); outputs are written as raw little-endian
// arrays DIR/<name>.bin; one JSON line with shapes and timing goes to stdout.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

union Arg {
  const void* p;
  long long i;
  float f;
};
struct Entry {
  const char* name;
  int nargs;
  int (*call)(void* fn, const Arg* a);
};
#include "dispatch_gen.inc"

#define HIP_OK(expr)                                                                                      \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) {                                                                               \
      fprintf(stderr, "codetr_runner: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                                            \
    }                                                                                                     \
  } while (0)

enum Kind : uint8_t { kInt = 0, kFloat = 1, kDev = 2, kNull = 3, kHost = 4, kStream = 5 };
struct PArg {
  uint8_t kind;
  long long i = 0;
  float f = 0.f;
  uint32_t seg = 0;
  uint64_t off = 0;
  std::vector<unsigned char> host;
};
struct Call {
  const Entry* entry;
  void* fn;
  std::vector<PArg> args;
};
struct Io {
  std::string name, dtype;
  uint32_t seg;
  uint64_t off, nbytes;
  std::vector<long long> shape;
};

struct Reader {
  std::ifstream f;
  template <class T>
  T get() {
    T v;
    f.read(reinterpret_cast<char*>(&v), sizeof(T));
    if (!f) {
      fprintf(stderr, "codetr_runner: truncated plan\n");
      exit(2);
    }
    return v;
  }
  void bytes(void* dst, size_t n) {   // bulk read, checked like get()
    f.read(reinterpret_cast<char*>(dst), (std::streamsize)n);
    if (!f) {
      fprintf(stderr, "codetr_runner: truncated plan\n");
      exit(2);
    }
  }
  std::string str(size_t n) {
    std::string s(n, '\0');
    if (n) bytes(&s[0], n);
    return s;
  }
};

static std::vector<unsigned char> read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) {
    fprintf(stderr, "codetr_runner: cannot open %s\n", path.c_str());
    exit(2);
  }
  std::vector<unsigned char> b((size_t)f.tellg());
  f.seekg(0);
  f.read(reinterpret_cast<char*>(b.data()), (std::streamsize)b.size());
  return b;
}

int main(int argc, char** argv) {
  std::string plan_path, lib_path = "libcodetr_hip.so", out_dir;
  int iters = 20;
  bool use_graph = true;
  std::map<std::string, std::string> inputs;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() -> std::string {
      if (i + 1 >= argc) {
        fprintf(stderr, "codetr_runner: %s needs a value\n", a.c_str());
        exit(2);
      }
      return argv[++i];
    };
    if (a == "--plan") plan_path = next();
    else if (a == "--lib") lib_path = next();
    else if (a == "--iters") iters = atoi(next().c_str());
    else if (a == "--no-graph") use_graph = false;
    else if (a == "--out-dir") out_dir = next();
    else if (a == "--input") {
      const std::string kv = next();
      const size_t eq = kv.find('=');
      if (eq == std::string::npos) {
        fprintf(stderr, "codetr_runner: --input name=file\n");
        return 2;
      }
      inputs[kv.substr(0, eq)] = kv.substr(eq + 1);
    } else {
      fprintf(stderr, "usage: codetr_runner --plan P [--lib L] [--iters N] [--no-graph] [--input name=file] [--out-dir D]\n");
      return 2;
    }
  }
  if (plan_path.empty()) {
    fprintf(stderr, "codetr_runner: --plan is required\n");
    return 2;
  }

  // ---- the kernel library: a missing or stale library is fatal (there is no other execution path) ----
  void* lib = dlopen(lib_path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!lib) {
    fprintf(stderr, "codetr_runner: cannot load %s: %s\n", lib_path.c_str(), dlerror());
    return 2;
  }
  auto abi = reinterpret_cast<int (*)()>(dlsym(lib, "codetr_hip_abi_version"));
  auto strerr = reinterpret_cast<const char* (*)(int)>(dlsym(lib, "codetr_hip_strerror"));
  if (!abi || !strerr || abi() != kPlanAbi) {
    fprintf(stderr, "codetr_runner: %s has ABI %d, this runner was generated for ABI %d\n", lib_path.c_str(), abi ? abi() : -1, kPlanAbi);
    return 2;
  }

  // ---- plan ----
  Reader r;
  r.f.open(plan_path, std::ios::binary);
  if (!r.f) {
    fprintf(stderr, "codetr_runner: cannot open %s\n", plan_path.c_str());
    return 2;
  }
  const std::string magic = r.str(12);
  if (magic != std::string("CODETRPLAN\0\2", 12)) {
    fprintf(stderr, "codetr_runner: %s is not a plan of this format\n", plan_path.c_str());
    return 2;
  }
  const int plan_abi = r.get<int32_t>();
  if (plan_abi != kPlanAbi) {
    fprintf(stderr, "codetr_runner: plan recorded with ABI %d, library / runner have %d: re-export the plan\n", plan_abi, kPlanAbi);
    return 2;
  }
  const uint32_t nseg = r.get<uint32_t>();
  std::vector<uint64_t> seg_size(nseg);
  std::vector<unsigned char*> seg_base(nseg, nullptr);
  uint64_t total = 0;
  for (uint32_t i = 0; i < nseg; ++i) {
    seg_size[i] = r.get<uint64_t>();
    if (seg_size[i]) {
      HIP_OK(hipMalloc(reinterpret_cast<void**>(&seg_base[i]), seg_size[i]));
      total += seg_size[i];
    }
  }
  const uint32_t nblob = r.get<uint32_t>();
  uint64_t uploaded = 0;
  {
    std::vector<unsigned char> buf;
    for (uint32_t i = 0; i < nblob; ++i) {
      const uint32_t seg = r.get<uint32_t>();
      const uint64_t off = r.get<uint64_t>(), n = r.get<uint64_t>();
      if (seg >= nseg || !seg_base[seg] || off > seg_size[seg] || n > seg_size[seg] - off) {   // (before sizing the buffer)
        fprintf(stderr, "codetr_runner: blob %u outside its segment\n", i);
        return 2;
      }
      buf.resize(n);
      if (n) r.bytes(buf.data(), n);
      HIP_OK(hipMemcpy(seg_base[seg] + off, buf.data(), n, hipMemcpyHostToDevice));
      uploaded += n;
    }
  }
  const uint32_t ncall = r.get<uint32_t>();
  std::vector<Call> calls(ncall);
  const size_t nentries = sizeof(kEntries) / sizeof(kEntries[0]);
  for (uint32_t c = 0; c < ncall; ++c) {
    const std::string name = r.str(r.get<uint16_t>());
    const Entry* e = nullptr;
    for (size_t k = 0; k < nentries; ++k)
      if (name == kEntries[k].name) e = &kEntries[k];
    const int nargs = r.get<uint8_t>();
    if (!e || e->nargs != nargs) {
      fprintf(stderr, "codetr_runner: plan calls %s/%d, unknown to this runner\n", name.c_str(), nargs);
      return 2;
    }
    calls[c].entry = e;
    calls[c].fn = dlsym(lib, name.c_str());
    if (!calls[c].fn) {
      fprintf(stderr, "codetr_runner: %s does not export %s\n", lib_path.c_str(), name.c_str());
      return 2;
    }
    calls[c].args.resize(nargs);
    for (int j = 0; j < nargs; ++j) {
      PArg& a = calls[c].args[j];
      a.kind = r.get<uint8_t>();
      switch (a.kind) {
        case kInt: a.i = r.get<int64_t>(); break;
        case kFloat: a.f = r.get<float>(); break;
        case kDev:
          a.seg = r.get<uint32_t>();
          a.off = r.get<uint64_t>();
          if (a.seg >= nseg || !seg_base[a.seg] || a.off >= seg_size[a.seg]) {
            fprintf(stderr, "codetr_runner: call %u (%s) points outside its segment\n", c, name.c_str());
            return 2;
          }
          break;
        case kHost: {
          const uint32_t n = r.get<uint32_t>();
          if (n > (1u << 20)) {   // host arguments are level shapes / window tables: a few hundred bytes
            fprintf(stderr, "codetr_runner: call %u (%s) carries a %u-byte host argument\n", c, name.c_str(), n);
            return 2;
          }
          a.host.resize(n);
          if (n) r.bytes(a.host.data(), n);
          break;
        }
        case kNull:
        case kStream: break;
        default: fprintf(stderr, "codetr_runner: bad argument kind %d\n", a.kind); return 2;
      }
    }
  }
  const uint32_t nio = r.get<uint32_t>();
  std::vector<Io> ios(nio);
  for (auto& io : ios) {
    io.name = r.str(r.get<uint16_t>());
    io.dtype = r.str(r.get<uint16_t>());
    io.seg = r.get<uint32_t>();
    io.off = r.get<uint64_t>();
    io.nbytes = r.get<uint64_t>();
    io.shape.resize(r.get<uint8_t>());
    for (auto& d : io.shape) d = r.get<int64_t>();
    if (io.seg >= nseg || !seg_base[io.seg] || io.off > seg_size[io.seg] || io.nbytes > seg_size[io.seg] - io.off) {
      fprintf(stderr, "codetr_runner: tensor %s lies outside its segment\n", io.name.c_str());
      return 2;
    }
  }

  // ---- inputs given on the command line replace the recorded ones ----
  for (const auto& kv : inputs) {
    const Io* io = nullptr;
    for (const auto& x : ios)
      if (x.name == kv.first) io = &x;
    if (!io) {
      fprintf(stderr, "codetr_runner: the plan has no tensor named %s\n", kv.first.c_str());
      return 2;
    }
    const auto bytes = read_file(kv.second);
    if (bytes.size() != io->nbytes) {
      fprintf(stderr, "codetr_runner: %s holds %zu bytes, %s needs %llu\n", kv.second.c_str(), bytes.size(), io->name.c_str(),
              (unsigned long long)io->nbytes);
      return 2;
    }
    HIP_OK(hipMemcpy(seg_base[io->seg] + io->off, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
  }

  // ---- replay ----
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  std::vector<Arg> av(32);
  auto replay = [&]() {
    for (const Call& c : calls) {
      for (size_t j = 0; j < c.args.size(); ++j) {
        const PArg& a = c.args[j];
        switch (a.kind) {
          case kInt: av[j].i = a.i; break;
          case kFloat: av[j].i = 0; av[j].f = a.f; break;
          case kDev: av[j].p = seg_base[a.seg] + a.off; break;
          case kHost: av[j].p = a.host.data(); break;
          case kStream: av[j].p = stream; break;
          default: av[j].p = nullptr; break;
        }
      }
      const int rc = c.entry->call(c.fn, av.data());
      if (rc != 0) {
        fprintf(stderr, "codetr_runner: %s failed: %s (code %d)\n", c.entry->name, strerr(rc), rc);
        exit(3);
      }
    }
  };
  replay();  // eager once: per-device function attributes, first-touch costs
  HIP_OK(hipStreamSynchronize(stream));
  hipGraphExec_t exec = nullptr;
  if (use_graph) {
    hipGraph_t graph;
    HIP_OK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    replay();
    HIP_OK(hipStreamEndCapture(stream, &graph));
    HIP_OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HIP_OK(hipGraphLaunch(exec, stream));
    HIP_OK(hipStreamSynchronize(stream));
  }
  std::vector<double> ms;
  for (int it = 0; it < iters; ++it) {
    const auto t0 = std::chrono::high_resolution_clock::now();
    if (exec) HIP_OK(hipGraphLaunch(exec, stream));
    else replay();
    HIP_OK(hipStreamSynchronize(stream));
    ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count());
  }
  std::sort(ms.begin(), ms.end());

  // ---- outputs ----
  std::string shapes;
  for (const auto& io : ios) {
    if (io.name == "batch_inputs" || io.name == "img_masks") continue;
    std::vector<unsigned char> host(io.nbytes);
    HIP_OK(hipMemcpy(host.data(), seg_base[io.seg] + io.off, io.nbytes, hipMemcpyDeviceToHost));
    if (!out_dir.empty()) {
      std::ofstream o(out_dir + "/" + io.name + ".bin", std::ios::binary);
      o.write(reinterpret_cast<const char*>(host.data()), (std::streamsize)host.size());
    }
    shapes += (shapes.empty() ? "" : ", ") + std::string("\"") + io.name + "\": {\"dtype\": \"" + io.dtype + "\", \"shape\": [";
    for (size_t d = 0; d < io.shape.size(); ++d) shapes += (d ? ", " : "") + std::to_string(io.shape[d]);
    shapes += "]}";
  }
  printf("{\"launches\": %u, \"segments_bytes\": %llu, \"uploaded_bytes\": %llu, \"hipgraph\": %s, \"iters\": %d, "
         "\"p50_ms\": %.3f, \"min_ms\": %.3f, \"outputs\": {%s}}\n",
         ncall, (unsigned long long)total, (unsigned long long)uploaded, exec ? "true" : "false", iters,
         ms.empty() ? 0.0 : ms[ms.size() / 2], ms.empty() ? 0.0 : ms[0], shapes.c_str());
  return 0;
}
