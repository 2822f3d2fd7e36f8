// runtime.cpp - error reporting and the recorded-program runner.
#include <stdarg.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

static thread_local char g_err[512] = "";

void fdbm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* fdbm_last_error(void) { return g_err; }
extern "C" int fdbm_version(void) { return 1; }

// A kernel of our own, not hipMemsetAsync: a memset node captured into a HIP graph zeroes on the FIRST replay only
// (ROCm 7.2, gfx950; tools/memset_node_repro.py, profiles/r02/memset_node_repro.txt: { memset(buf) ; buf += 1 }
// reads 1.0 after replay 1 and (1.0 .. inf) after every later one, for 64-byte and 412-KiB nodes alike, with or
// without host copies in between; a 2-MiB node and a kernel node are correct).  The statistics arena was such a
// node in round 1: from the second replay on it was filled with a non-zero pattern and every result was NaN.
__global__ void __launch_bounds__(256) zero_kernel(uint4* __restrict__ p, int64_t n16, unsigned char* __restrict__ tail,
                                                   int ntail) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
    p[i] = uint4{0u, 0u, 0u, 0u};
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

extern "C" int fdbm_memset_zero(void* ptr, int64_t bytes, void* stream) {
  FDBM_CHECK(ptr && bytes >= 0, "fdbm_memset_zero: bad arguments");
  FDBM_CHECK(((uintptr_t)ptr & 15) == 0, "fdbm_memset_zero: pointer must be 16-byte aligned");
  if (bytes == 0) return 0;
  const int64_t n16 = bytes / 16;
  int g = (int)((n16 + 255) / 256);
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  zero_kernel<<<g, 256, 0, (hipStream_t)stream>>>((uint4*)ptr, n16, (unsigned char*)ptr + n16 * 16, (int)(bytes - n16 * 16));
  FDBM_LAUNCH_CHECK("fdbm_memset_zero");
  return 0;
}

__global__ void copy_f32_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// small device-to-device copy as a kernel of the library (inside replayed graphs: see fdbm_memset_zero)
extern "C" int fdbm_copy_f32(float* dst, const float* src, int64_t n, void* stream) {
  FDBM_CHECK(dst && src && n >= 0, "fdbm_copy_f32: bad arguments");
  if (n == 0) return 0;
  int g = (int)((n + 255) / 256);
  if (g > 1024) g = 1024;
  copy_f32_kernel<<<g, 256, 0, (hipStream_t)stream>>>(dst, src, n);
  FDBM_LAUNCH_CHECK("fdbm_copy_f32");
  return 0;
}

// ---- side lane -------------------------------------------------------------------------------------
static hipStream_t g_side = nullptr;
static hipEvent_t g_ev[FDBM_MAX_EVENTS];
static bool g_side_ok = false;

extern "C" int fdbm_runtime_init_side(void) {
  if (g_side_ok) return 0;
  // Opt-in (FDBM_SIDE_STREAM=1): measured on MI355X, the cross-stream edges cost more inside the HIP graph
  // (14 per forward) than the ~120 us of work they take off the main chain: 3.35 ms vs 3.17 ms per forward.
  const char* on = getenv("FDBM_SIDE_STREAM");
  if (!(on && on[0] == '1')) return 0;
  hipError_t e = hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking);
  FDBM_CHECK(e == hipSuccess, "fdbm_runtime_init_side: hipStreamCreate failed: %s", hipGetErrorString(e));
  for (int i = 0; i < FDBM_MAX_EVENTS; ++i) {
    e = hipEventCreateWithFlags(&g_ev[i], hipEventDisableTiming);
    FDBM_CHECK(e == hipSuccess, "fdbm_runtime_init_side: hipEventCreate failed: %s", hipGetErrorString(e));
  }
  g_side_ok = true;
  return 0;
}

#define P(i) ((void*)(intptr_t)o.iarg[i])
#define CP(i) ((const void*)(intptr_t)o.iarg[i])
#define FP(i) ((float*)(intptr_t)o.iarg[i])
#define CFP(i) ((const float*)(intptr_t)o.iarg[i])
#define I(i) ((int)o.iarg[i])

// Argument order per opcode = the parameter order of the C entry point (pointers and
// integers in iarg[], floats in farg[]), minus the trailing stream.
extern "C" int fdbm_run_program(const fdbm_op* ops, int n_ops, void* main_stream) {
  for (int k = 0; k < n_ops; ++k) {
    const fdbm_op& o = ops[k];
    int rc = 0;
    void* stream = (o.lane == 1 && g_side_ok) ? (void*)g_side : main_stream;
    switch (o.opcode) {
      case FDBM_OP_FORK:
      case FDBM_OP_MARK:
      case FDBM_OP_JOIN: {
        if (!g_side_ok) break;
        const int ev = I(0);
        FDBM_CHECK(ev >= 0 && ev < FDBM_MAX_EVENTS, "fdbm_run_program: event id %d out of range at op %d", ev, k);
        hipError_t e;
        if (o.opcode == FDBM_OP_FORK) {
          e = hipEventRecord(g_ev[ev], (hipStream_t)main_stream);
          if (e == hipSuccess) e = hipStreamWaitEvent(g_side, g_ev[ev], 0);
        } else if (o.opcode == FDBM_OP_MARK) {
          e = hipEventRecord(g_ev[ev], g_side);
        } else {
          e = hipStreamWaitEvent((hipStream_t)main_stream, g_ev[ev], 0);
        }
        FDBM_CHECK(e == hipSuccess, "fdbm_run_program: stream dependency failed at op %d: %s", k, hipGetErrorString(e));
        break;
      }
      case FDBM_OP_CONV:
        rc = fdbm_conv_igemm((const fdbm_conv_args*)CP(0), stream);
        break;
      case FDBM_OP_GN_STATS:
        rc = fdbm_gn_stats(FP(0), CP(1), I(2), CP(3), I(4), I(5), I(6), I(7), I(8), I(9), stream);
        break;
      case FDBM_OP_GN_FINALIZE:
        rc = fdbm_gn_finalize(FP(0), CFP(1), I(2), I(3), I(4), o.iarg[5], o.farg[0], stream);
        break;
      case FDBM_OP_GN_APPLY:
        rc = fdbm_gn_apply(P(0), CP(1), I(2), CP(3), I(4), CFP(5), I(6), o.iarg[7], o.farg[0], CFP(8),
                           CFP(9), I(10), I(11), I(12), I(13), I(14), stream);
        break;
      case FDBM_OP_RESAMPLE:
        rc = fdbm_resample2x_units(P(0), P(1), CP(2), CFP(3), I(4), I(15) > 0 ? I(15) : 1, o.iarg[5], o.farg[0],
                                   CFP(6), CFP(7), I(8), I(9), I(10), I(11), I(12), I(13), I(14), stream);
        break;
      case FDBM_OP_COMBINE:
        rc = fdbm_combine(P(0), CP(1), CFP(2), CFP(3), CFP(4), o.iarg[5], I(6), I(7), stream);
        break;
      case FDBM_OP_ATTENTION:
        rc = fdbm_attention(P(0), CP(1), I(2), I(3), I(4), I(5), stream);
        break;
      case FDBM_OP_STEM:
        rc = fdbm_conv_stem_stats(P(0), CFP(1), CFP(2), CFP(3), I(4), I(5), I(6), I(7), I(8), (double*)P(9), I(10), stream);
        break;
      case FDBM_OP_PACK:
        rc = fdbm_pack_input(FP(0), CP(1), CP(2), I(3), I(4), I(5), I(6), stream);
        break;
      case FDBM_OP_UNPACK:
        rc = fdbm_unpack_output(P(0), CFP(1), CFP(2), CFP(3), I(4), I(5), I(6), I(7), stream);
        break;
      case FDBM_OP_TEMB:
        rc = fdbm_temb(FP(0), CFP(1), CFP(2), CFP(3), CFP(4), CFP(5), CFP(6), FP(7), I(8), I(9),
                       stream);
        break;
      case FDBM_OP_DENSE:
        rc = fdbm_dense_rows(FP(0), CFP(1), CFP(2), CFP(3), I(4), I(5), I(6), stream);
        break;
      case FDBM_OP_UPDATE:
        rc = fdbm_bridge_update(P(0), CP(1), CP(2), CP(3), CFP(4), CFP(5), CFP(6), I(7), o.iarg[8],
                                stream);
        break;
      case FDBM_OP_MEMSET:
        rc = fdbm_memset_zero(P(0), o.iarg[1], stream);
        break;
      case FDBM_OP_PYRDOWN: {
        void* outs[8];
        for (int l = 0; l < 8; ++l) outs[l] = P(5 + l);
        rc = fdbm_pyramid_down_chain(CFP(0), outs, I(1), I(2), I(3), I(4), stream);
        break;
      }
      default:
        fdbm_set_error("fdbm_run_program: unknown opcode %d at op %d", o.opcode, k);
        return 3;
    }
    if (rc) return rc;
  }
  return 0;
}

// ---------------------------------------------------------------------------------
// composed backbone entry: a context = a recorded program + its static I/O buffers
// ---------------------------------------------------------------------------------
struct fdbm_ncsnpp_ctx {
  fdbm_conv_args* convs = nullptr;   // owned copies (contexts created from a serialised program)
  fdbm_op* ops;
  int n_ops;
  void* x_in;
  void* y_in;
  float* logt_in;
  void* s_out;
  int64_t n_complex;   // complex64 elements of x / y / s (B * F * T)
  int B;
};

extern "C" fdbm_ncsnpp_ctx* fdbm_ncsnpp_create(const fdbm_op* ops_host, int n_ops, void* x_in, void* y_in,
                                               float* logt_in, void* s_out, int64_t n_complex, int B) {
  if (!ops_host || n_ops <= 0 || !x_in || !y_in || !logt_in || !s_out || n_complex <= 0 || B <= 0) {
    fdbm_set_error("fdbm_ncsnpp_create: bad arguments");
    return nullptr;
  }
  fdbm_ncsnpp_ctx* c = new fdbm_ncsnpp_ctx;
  c->ops = new fdbm_op[n_ops];
  memcpy(c->ops, ops_host, sizeof(fdbm_op) * (size_t)n_ops);
  c->n_ops = n_ops; c->x_in = x_in; c->y_in = y_in; c->logt_in = logt_in; c->s_out = s_out;
  c->n_complex = n_complex; c->B = B;
  return c;
}

extern "C" void fdbm_ncsnpp_destroy(fdbm_ncsnpp_ctx* c) {
  if (!c) return;
  delete[] c->ops;
  delete[] c->convs;
  delete c;
}

// ---- serialised programs (include/fdbm_hip.h "contexts without Python", fdbm_amd/export.py) --------------------
static const fdbm_program_header* program_header(const void* program, int64_t nbytes) {
  if (!program || nbytes < (int64_t)sizeof(fdbm_program_header)) { fdbm_set_error("program: too short"); return nullptr; }
  const fdbm_program_header* h = reinterpret_cast<const fdbm_program_header*>(program);
  if (h->magic != 0x474F5250424D4446ull || h->version != 1) { fdbm_set_error("program: bad magic / version"); return nullptr; }
  if (h->conv_args_size != sizeof(fdbm_conv_args)) {
    fdbm_set_error("program: fdbm_conv_args is %u bytes in the file, %zu in this library", h->conv_args_size, sizeof(fdbm_conv_args));
    return nullptr;
  }
  const int64_t need = (int64_t)sizeof(fdbm_program_header) + (int64_t)h->n_ops * (int64_t)sizeof(fdbm_op) +
                       (int64_t)h->n_conv * (int64_t)sizeof(fdbm_conv_args);
  if (nbytes < need) { fdbm_set_error("program: %lld bytes, %lld needed", (long long)nbytes, (long long)need); return nullptr; }
  return h;
}
extern "C" int64_t fdbm_program_workspace_bytes(const void* program, int64_t nbytes) {
  const fdbm_program_header* h = program_header(program, nbytes);
  return h ? h->workspace_bytes : -1;
}
extern "C" int64_t fdbm_program_weights_bytes(const void* program, int64_t nbytes) {
  const fdbm_program_header* h = program_header(program, nbytes);
  return h ? h->weights_bytes : -1;
}

extern "C" fdbm_ncsnpp_ctx* fdbm_ncsnpp_create_from_program(const void* program, int64_t nbytes, void* weights_dev,
                                                            void* workspace_dev, int64_t workspace_bytes) {
  const fdbm_program_header* h = program_header(program, nbytes);
  if (!h) return nullptr;
  if (!weights_dev || !workspace_dev || workspace_bytes < h->workspace_bytes) {
    fdbm_set_error("fdbm_ncsnpp_create_from_program: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
                   (long long)h->workspace_bytes);
    return nullptr;
  }
  bool bad = false;
  auto reloc = [&](int64_t v) -> void* {
    if (v == 0) return nullptr;
    if (((uint64_t)v >> 62) != 1) { bad = true; return nullptr; }      // tagged pointers: bit 62 set, bit 63 clear
    const int region = (int)((v >> 60) & 3);
    const int64_t off = v & ((1ll << 60) - 1);
    // (a tagged pointer must START inside its region; the extent of each access is the recorded op's business: program files
    // are TRUSTED input, produced by fdbm_amd.export from a program that ran - INTEGRATION.md)
    if (region == 0 && off < h->workspace_bytes) return (char*)workspace_dev + off;
    if (region == 1 && off < h->weights_bytes) return (char*)weights_dev + off;
    bad = true;
    return nullptr;
  };
  if (hipMemset(workspace_dev, 0, (size_t)h->workspace_bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    fdbm_set_error("fdbm_ncsnpp_create_from_program: zeroing the workspace failed");
    return nullptr;
  }
  const fdbm_op* ops_in = reinterpret_cast<const fdbm_op*>(h + 1);
  const fdbm_conv_args* conv_in = reinterpret_cast<const fdbm_conv_args*>(ops_in + h->n_ops);
  fdbm_ncsnpp_ctx* c = new fdbm_ncsnpp_ctx;
  c->ops = new fdbm_op[h->n_ops];
  c->convs = new fdbm_conv_args[h->n_conv > 0 ? h->n_conv : 1];
  memcpy(c->ops, ops_in, sizeof(fdbm_op) * (size_t)h->n_ops);
  memcpy(c->convs, conv_in, sizeof(fdbm_conv_args) * (size_t)h->n_conv);
  for (uint32_t i = 0; i < h->n_conv; ++i) {
    fdbm_conv_args& a = c->convs[i];
#define RL(f) a.f = (std::remove_reference_t<decltype((a.f))>)reloc((int64_t)(intptr_t)a.f)
    for (int s = 0; s < FDBM_MAX_SEG; ++s) { RL(seg[s].src); RL(gn_seg_sums[s]); }
    RL(w); RL(bias); RL(tbias); RL(res); RL(out); RL(workspace); RL(gn_sums); RL(gn_gamma); RL(gn_beta);
    RL(comb_pyr); RL(comb_w); RL(comb_b); RL(stat_out); RL(w_frag); RL(acc_ws); RL(res_up2x);
#undef RL
  }
  for (uint32_t i = 0; i < h->n_ops; ++i) {
    fdbm_op& op = c->ops[i];
    if (op.opcode == FDBM_OP_CONV) {
      if (op.iarg[0] < 0 || op.iarg[0] >= (int64_t)h->n_conv) { bad = true; break; }
      op.iarg[0] = (int64_t)(intptr_t)&c->convs[op.iarg[0]];
      continue;
    }
    for (int j = 0; j < 24; ++j)
      if (((uint64_t)op.iarg[j] >> 62) == 1) op.iarg[j] = (int64_t)(intptr_t)reloc(op.iarg[j]);   // (negative counts have bit 63 set)
  }
  c->n_ops = (int)h->n_ops;
  c->x_in = reloc(h->x_in); c->y_in = reloc(h->y_in); c->logt_in = (float*)reloc(h->t_in); c->s_out = reloc(h->s_out);
  c->n_complex = h->n_complex; c->B = h->B;
  if (bad || !c->x_in || !c->y_in || !c->logt_in || !c->s_out) {
    fdbm_set_error("fdbm_ncsnpp_create_from_program: a pointer of the program lies outside its regions");
    fdbm_ncsnpp_destroy(c);
    return nullptr;
  }
  if (fdbm_runtime_init_side() != 0) { fdbm_ncsnpp_destroy(c); return nullptr; }
  return c;
}

// s = dnn(x, y, t): device pointers (complex64 [B][1][F][T] x2, f32 log t [B], complex64 out);
// x / y / log_t / out may be the context's own static buffers (then no copy is enqueued).
extern "C" int fdbm_ncsnpp_forward(fdbm_ncsnpp_ctx* c, const void* x, const void* y, const float* log_t,
                                   void* out, void* stream) {
  FDBM_CHECK(c && x && y && log_t && out, "fdbm_ncsnpp_forward: null argument");
  // copies by the library's own kernel (not hipMemcpyAsync): a caller may capture this entry into a HIP graph, and
  // the runtime's copy / memset graph nodes are what broke in round 1 (see zero_kernel above)
  const int64_t nf = (int64_t)c->n_complex * 2;
  int rc = 0;
  if (x != c->x_in) rc = fdbm_copy_f32((float*)c->x_in, (const float*)x, nf, stream);
  if (!rc && y != c->y_in) rc = fdbm_copy_f32((float*)c->y_in, (const float*)y, nf, stream);
  if (!rc && log_t != c->logt_in) rc = fdbm_copy_f32(c->logt_in, (const float*)log_t, (int64_t)c->B, stream);
  if (rc) return rc;
  rc = fdbm_run_program(c->ops, c->n_ops, stream);
  if (rc) return rc;
  if (out != c->s_out) rc = fdbm_copy_f32((float*)out, (const float*)c->s_out, nf, stream);
  return rc;
}
