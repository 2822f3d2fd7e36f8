// run_program.cpp - a complete host of the backbone in C++: no Python at run time.
//
//   python tools/export_program.py --backbone ncsnpp_v2_5M --frames 64 /tmp/net      (once, writes /tmp/net.fdbmprog/.fdbmw)
//   hipcc -O2 -I include examples/host_cpp/run_program.cpp -L <pkg>/csrc -lfdbm_hip -Wl,-rpath,<pkg>/csrc -o run_program
//   ./run_program /tmp/net [x.bin y.bin]      -> prints a checksum of s = dnn(x, y, t) and the time per evaluation
//
// Mirrors what fdbm/bridge.py:76 does per sampler step (`current_estimate = model(xt, y, time_prev)`), through the C ABI of
// include/fdbm_hip.h: one weight blob, one workspace, fdbm_ncsnpp_create_from_program, fdbm_ncsnpp_forward.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fdbm_hip.h"

static std::vector<char> slurp(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<char> b((size_t)n);
  if (fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read on %s\n", path.c_str()); exit(2); }
  fclose(f);
  return b;
}

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(3); } } while (0)

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s prefix [x.bin y.bin]\n", argv[0]); return 1; }
  const std::string prefix = argv[1];
  std::vector<char> prog = slurp(prefix + ".fdbmprog"), wts = slurp(prefix + ".fdbmw");
  const fdbm_program_header* h = reinterpret_cast<const fdbm_program_header*>(prog.data());
  const int64_t ws_bytes = fdbm_program_workspace_bytes(prog.data(), (int64_t)prog.size());
  const int64_t w_bytes = fdbm_program_weights_bytes(prog.data(), (int64_t)prog.size());
  if (ws_bytes < 0 || w_bytes < 0 || (size_t)w_bytes > wts.size()) { fprintf(stderr, "bad program: %s\n", fdbm_last_error()); return 2; }
  void *w_dev, *ws_dev, *x_dev, *y_dev, *s_dev;
  float* t_dev;
  HIPCK(hipMalloc(&w_dev, (size_t)w_bytes));
  HIPCK(hipMalloc(&ws_dev, (size_t)ws_bytes));
  HIPCK(hipMemcpy(w_dev, wts.data(), (size_t)w_bytes, hipMemcpyHostToDevice));
  fdbm_ncsnpp_ctx* ctx = fdbm_ncsnpp_create_from_program(prog.data(), (int64_t)prog.size(), w_dev, ws_dev, ws_bytes);
  if (!ctx) { fprintf(stderr, "create failed: %s\n", fdbm_last_error()); return 2; }
  const size_t n = (size_t)h->n_complex;            // complex64 elements of x / y / s
  std::vector<float> x(2 * n), y(2 * n), s(2 * n), logt((size_t)h->B);
  if (argc >= 4) {
    std::vector<char> xb = slurp(argv[2]), yb = slurp(argv[3]);
    if (xb.size() != 8 * n || yb.size() != 8 * n) { fprintf(stderr, "x / y must hold %zu complex64 values\n", n); return 2; }
    memcpy(x.data(), xb.data(), 8 * n); memcpy(y.data(), yb.data(), 8 * n);
  } else {
    unsigned st = 12345u;                             // a fixed pseudo-random input
    for (size_t i = 0; i < 2 * n; ++i) { st = st * 1664525u + 1013904223u; x[i] = ((st >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    for (size_t i = 0; i < 2 * n; ++i) { st = st * 1664525u + 1013904223u; y[i] = ((st >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
  }
  for (int b = 0; b < h->B; ++b) logt[b] = logf(0.5f);   // the model time enters as its host-evaluated logarithm (fdbm_temb)
  HIPCK(hipMalloc(&x_dev, 8 * n)); HIPCK(hipMalloc(&y_dev, 8 * n)); HIPCK(hipMalloc(&s_dev, 8 * n));
  HIPCK(hipMalloc((void**)&t_dev, sizeof(float) * (size_t)h->B));
  HIPCK(hipMemcpy(x_dev, x.data(), 8 * n, hipMemcpyHostToDevice));
  HIPCK(hipMemcpy(y_dev, y.data(), 8 * n, hipMemcpyHostToDevice));
  HIPCK(hipMemcpy(t_dev, logt.data(), sizeof(float) * (size_t)h->B, hipMemcpyHostToDevice));
  hipStream_t st;
  HIPCK(hipStreamCreate(&st));
  for (int it = 0; it < 3; ++it)
    if (fdbm_ncsnpp_forward(ctx, x_dev, y_dev, t_dev, s_dev, st)) { fprintf(stderr, "forward failed: %s\n", fdbm_last_error()); return 2; }
  HIPCK(hipStreamSynchronize(st));
  const auto t0 = std::chrono::steady_clock::now();
  const int reps = 20;
  for (int it = 0; it < reps; ++it) fdbm_ncsnpp_forward(ctx, x_dev, y_dev, t_dev, s_dev, st);
  HIPCK(hipStreamSynchronize(st));
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
  HIPCK(hipMemcpy(s.data(), s_dev, 8 * n, hipMemcpyDeviceToHost));
  double sum = 0.0, sq = 0.0;
  for (size_t i = 0; i < 2 * n; ++i) { sum += s[i]; sq += (double)s[i] * s[i]; }
  printf("B=%d F=%d T=%d dtype=%d  ops=%u  weights %.1f MB  workspace %.1f MB  %.3f ms per evaluation\n", h->B, h->F, h->T,
         h->dtype, h->n_ops, w_bytes / 1e6, ws_bytes / 1e6, ms);
  printf("checksum sum=%.9e  l2=%.9e\n", sum, std::sqrt(sq));
  if (argc >= 5) { FILE* f = fopen(argv[4], "wb"); fwrite(s.data(), 4, 2 * n, f); fclose(f); }
  fdbm_ncsnpp_destroy(ctx);
  return 0;
}
