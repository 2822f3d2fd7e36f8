// elementwise.hip - HBM-bound passes of the sampling path: sampler state update,
// predictor / corrector moves, network input packing / output projection, Combine,
// time embedding and the 4->nf stem convolution.  All fp32 math; 16-byte accesses.
#include "conv_common.h"

// ---------------------------------------------------------------------------------
// sampler state update: out = (wa*a + wb*b) + wc*c   (complex64 viewed as floats)
// ---------------------------------------------------------------------------------
template <bool HAS_C>
__global__ void __launch_bounds__(256) bridge_update_kernel(
    float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b,
    const float* __restrict__ c, const float* __restrict__ wa, const float* __restrict__ wb,
    const float* __restrict__ wc, int64_t nvec /*float4 per sample*/, int64_t tail_start,
    int64_t nfloat) {
  const int bi = blockIdx.y;
  const float fa = wa[bi], fb = wb[bi];
  const float fc = HAS_C ? wc[bi] : 0.f;
  const int64_t base = (int64_t)bi * nfloat;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 va = *reinterpret_cast<const f32x4*>(a + base + 4 * i);
    const f32x4 vb = *reinterpret_cast<const f32x4*>(b + base + 4 * i);
    f32x4 vc = {0, 0, 0, 0};
    if (HAS_C) vc = *reinterpret_cast<const f32x4*>(c + base + 4 * i);
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // each product and each sum rounded on its own: the SB first step cancels
      // 1794.79*y - 1793.82*y, so the rounding order is part of the contract.
      float s = __fadd_rn(__fmul_rn(fa, va[k]), __fmul_rn(fb, vb[k]));
      if (HAS_C) s = __fadd_rn(s, __fmul_rn(fc, vc[k]));
      r[k] = s;
    }
    *reinterpret_cast<f32x4*>(out + base + 4 * i) = r;
  }
  // scalar tail (n not a multiple of 4 floats)
  if (blockIdx.x == 0) {
    for (int64_t i = tail_start + threadIdx.x; i < nfloat; i += blockDim.x) {
      float s = __fadd_rn(__fmul_rn(fa, a[base + i]), __fmul_rn(fb, b[base + i]));
      if (HAS_C) s = __fadd_rn(s, __fmul_rn(fc, c[base + i]));
      out[base + i] = s;
    }
  }
}

extern "C" int fdbm_bridge_update(void* out, const void* a, const void* b, const void* c,
                                  const float* wa, const float* wb, const float* wc, int B,
                                  int64_t n_complex, void* stream) {
  FDBM_CHECK(out && a && b && wa && wb, "fdbm_bridge_update: null pointer");
  FDBM_CHECK((c == nullptr) == (wc == nullptr), "fdbm_bridge_update: c and wc must both be set or both NULL");
  FDBM_CHECK(B > 0 && n_complex >= 0, "fdbm_bridge_update: bad sizes B=%d n=%lld", B, (long long)n_complex);
  if (n_complex == 0) return 0;
  const int64_t nfloat = 2 * n_complex;
  const bool aligned = ((nfloat % 4) == 0) || B == 1;
  const int64_t nvec = aligned ? nfloat / 4 : 0;
  const int64_t tail = nvec * 4;
  int gx = (int)((nvec + 255) / 256);
  if (gx < 1) gx = 1;
  if (gx > 2048) gx = 2048;
  dim3 grid(gx, B);
  hipStream_t st = (hipStream_t)stream;
  if (c)
    bridge_update_kernel<true><<<grid, 256, 0, st>>>((float*)out, (const float*)a, (const float*)b,
                                                     (const float*)c, wa, wb, wc, nvec, tail, nfloat);
  else
    bridge_update_kernel<false><<<grid, 256, 0, st>>>((float*)out, (const float*)a, (const float*)b,
                                                      nullptr, wa, wb, nullptr, nvec, tail, nfloat);
  FDBM_LAUNCH_CHECK("fdbm_bridge_update");
  return 0;
}

// ---------------------------------------------------------------------------------
// counter-based Gaussian noise (the per-step draws of the stochastic samplers: torch.randn_like in
// fdbm/bridge.py:47,108, fdbm/util/predictors.py:46, fdbm/util/correctors.py:48,76) generated ON the device, in
// registers, by the kernel that consumes it - no host generator, no N x B noise tensors uploaded before a replay.
// Definition (also restated in numpy: oracle/rng.py).  Complex element e (flat index into the [B,1,F,T] state) of
// draw d under seed s:
//     (x0, x1, x2, x3) = Philox4x32-10(counter = (e_lo, e_hi, d, 0x46444d42), key = (s_lo, s_hi))
//     u1 = ((x0 >> 8) + 0.5) 2^-24,  u2 = ((x1 >> 8) + 0.5) 2^-24          (both in (0, 1))
//     r = sqrt(-2 ln u1),  re = r cos(2 pi u2) sqrt(1/2),  im = r sin(2 pi u2) sqrt(1/2)
// i.e. Re, Im ~ N(0, 1/2) like torch.randn_like of a complex64 tensor.  A draw is named by its index d in the
// sampler's call order (0 = the prior, then the per-step draws); rng = device words {s_lo, s_hi, d_base}: d = d_base +
// the draw argument, so successive sampler calls continue the stream without re-capturing their graph.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t& o0, uint32_t& o1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o0 = c0; o1 = c1;
  (void)c2; (void)c3;
}

__device__ __forceinline__ f32x2 rng_complex_normal(int64_t e, uint32_t draw, uint32_t s_lo, uint32_t s_hi) {
  uint32_t x0, x1;
  philox4x32_10((uint32_t)(uint64_t)e, (uint32_t)((uint64_t)e >> 32), draw, 0x46444d42u, s_lo, s_hi, x0, x1);
  const float u1 = ((float)(x0 >> 8) + 0.5f) * 5.9604644775390625e-08f;
  const float u2 = ((float)(x1 >> 8) + 0.5f) * 5.9604644775390625e-08f;
  const float r = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincosf(6.283185307179586f * u2, &sn, &cs);
  return f32x2{(r * cs) * 0.7071067811865476f, (r * sn) * 0.7071067811865476f};
}

__global__ void __launch_bounds__(256) randn_complex_kernel(f32x2* __restrict__ out, int64_t n, const uint32_t* __restrict__ rng,
                                                            uint32_t draw) {
  const uint32_t s_lo = rng[0], s_hi = rng[1], d = rng[2] + draw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = rng_complex_normal(i, d, s_lo, s_hi);
}

extern "C" int fdbm_randn_complex(void* out, int64_t n_complex, const uint32_t* rng, uint32_t draw, void* stream) {
  FDBM_CHECK(out && rng && n_complex >= 0, "fdbm_randn_complex: null pointer / bad size");
  if (n_complex == 0) return 0;
  int g = (int)((n_complex + 255) / 256);
  if (g > 4096) g = 4096;
  randn_complex_kernel<<<g, 256, 0, (hipStream_t)stream>>>((f32x2*)out, n_complex, rng, draw);
  FDBM_LAUNCH_CHECK("fdbm_randn_complex");
  return 0;
}

// ---------------------------------------------------------------------------------
// step boundary of the exponential-integrator samplers inside a replayed graph: what sits between two backbone
// evaluations - unpack_output (pyramid -> score, ncsnpp_v2.py:391-399), the state update (fdbm_bridge_update), the
// next evaluation's pack_input, the zeroing of its statistics arena and the copy of its time-embedding rows - as ONE
// launch instead of five (each 4-5 us of launch floor at batch 1).  Elementwise in (b, f, t); the arithmetic of the
// score and of the update is the two kernels' own, expression for expression (bit-identical results).
// ---------------------------------------------------------------------------------
// RNG: `third` is the step's Gaussian noise, generated here (rng_complex_normal) instead of read
template <bool RNG>
__global__ void __launch_bounds__(256) step_boundary_kernel(
    f32x2* __restrict__ x, const f32x2* __restrict__ y, const f32x2* __restrict__ third, const uint32_t* __restrict__ rng, uint32_t draw,
    const f32x4* __restrict__ pyr, const float* __restrict__ ow, const float* __restrict__ ob,
    const float* __restrict__ wa, const float* __restrict__ wb, const float* __restrict__ wc,
    f32x4* __restrict__ inp, uint4* __restrict__ zero16, int64_t nzero16,
    float* __restrict__ dense_dst, const float* __restrict__ dense_src, int64_t ndense,
    int F, int Fn, int T, int64_t total) {
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gstride = (int64_t)gridDim.x * blockDim.x;
  float w00 = 0, w01 = 0, w02 = 0, w03 = 0, w10 = 0, w11 = 0, w12 = 0, w13 = 0, b0 = 0, b1 = 0;
  uint32_t s_lo = 0, s_hi = 0, dr = 0;
  if constexpr (RNG) { s_lo = rng[0]; s_hi = rng[1]; dr = rng[2] + draw; }
  if (pyr) {
    w00 = ow[0]; w01 = ow[1]; w02 = ow[2]; w03 = ow[3];
    w10 = ow[4]; w11 = ow[5]; w12 = ow[6]; w13 = ow[7];
    b0 = ob[0]; b1 = ob[1];
  }
  for (int64_t i = gtid; i < total; i += gstride) {
    const int t = (int)(i % T);
    const int64_t r = i / T;
    const int f = (int)(r % F);
    const int64_t b = r / F;
    f32x2 xv = x[i];
    const f32x2 yv = y[i];
    if (pyr) {
      f32x2 o = {0.f, 0.f};
      if (f < Fn) {
        const f32x4 p = pyr[(b * Fn + f) * T + t];
        o[0] = b0 + (((w00 * p[0] + w01 * p[1]) + w02 * p[2]) + w03 * p[3]);
        o[1] = b1 + (((w10 * p[0] + w11 * p[1]) + w12 * p[2]) + w13 * p[3]);
      }
      const float fa = wa[b], fb = wb[b], fc = wc[b];
      f32x2 cv;
      if constexpr (RNG) cv = rng_complex_normal(i, dr, s_lo, s_hi); else cv = third[i];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float sv = __fadd_rn(__fmul_rn(fa, xv[k]), __fmul_rn(fb, o[k]));
        sv = __fadd_rn(sv, __fmul_rn(fc, cv[k]));
        xv[k] = sv;
      }
      x[i] = xv;
    }
    if (inp && f < Fn) {
      f32x4 o4 = {xv[0], xv[1], yv[0], yv[1]};
      inp[(b * Fn + f) * T + t] = o4;
    }
  }
  for (int64_t i = gtid; i < nzero16; i += gstride) zero16[i] = uint4{0u, 0u, 0u, 0u};
  for (int64_t i = gtid; i < ndense; i += gstride) dense_dst[i] = dense_src[i];
}

static int step_boundary_impl(void* x, const void* y, const void* third, const uint32_t* rng, uint32_t draw, const float* pyramid,
                              const float* out_w, const float* out_b, const float* wa, const float* wb, const float* wc, float* packed,
                              void* zero_ptr, int64_t zero_bytes, float* dense_dst, const float* dense_src,
                              int64_t dense_n, int B, int F, int Fn, int T, void* stream) {
  FDBM_CHECK(x && y, "fdbm_step_boundary: null state");
  FDBM_CHECK(!pyramid || ((third || rng) && out_w && out_b && wa && wb && wc), "fdbm_step_boundary: the update needs third (or rng), out_w, out_b, wa, wb, wc");
  FDBM_CHECK(B > 0 && Fn > 0 && Fn <= F && T > 0, "fdbm_step_boundary: bad shape");
  FDBM_CHECK(zero_bytes >= 0 && zero_bytes % 16 == 0 && (((uintptr_t)zero_ptr) & 15) == 0, "fdbm_step_boundary: the zeroed range must be 16-byte aligned and sized");
  FDBM_CHECK(dense_n >= 0 && (dense_n == 0 || (dense_dst && dense_src)), "fdbm_step_boundary: bad time-embedding rows");
  const int64_t total = (int64_t)B * F * T;
  int g = (int)((total + 255) / 256);
  if (g > 4096) g = 4096;
  if (rng)
    step_boundary_kernel<true><<<g, 256, 0, (hipStream_t)stream>>>(
        (f32x2*)x, (const f32x2*)y, nullptr, rng, draw, (const f32x4*)pyramid, out_w, out_b, wa, wb, wc, (f32x4*)packed,
        (uint4*)zero_ptr, zero_ptr ? zero_bytes / 16 : 0, dense_dst, dense_src, dense_n, F, Fn, T, total);
  else
    step_boundary_kernel<false><<<g, 256, 0, (hipStream_t)stream>>>(
        (f32x2*)x, (const f32x2*)y, (const f32x2*)third, nullptr, 0u, (const f32x4*)pyramid, out_w, out_b, wa, wb, wc, (f32x4*)packed,
        (uint4*)zero_ptr, zero_ptr ? zero_bytes / 16 : 0, dense_dst, dense_src, dense_n, F, Fn, T, total);
  FDBM_LAUNCH_CHECK("fdbm_step_boundary");
  return 0;
}

extern "C" int fdbm_step_boundary(void* x, const void* y, const void* third, const float* pyramid, const float* out_w,
                                  const float* out_b, const float* wa, const float* wb, const float* wc, float* packed,
                                  void* zero_ptr, int64_t zero_bytes, float* dense_dst, const float* dense_src,
                                  int64_t dense_n, int B, int F, int Fn, int T, void* stream) {
  return step_boundary_impl(x, y, third, nullptr, 0u, pyramid, out_w, out_b, wa, wb, wc, packed, zero_ptr, zero_bytes, dense_dst,
                            dense_src, dense_n, B, F, Fn, T, stream);
}

extern "C" int fdbm_step_boundary_rng(void* x, const void* y, const uint32_t* rng, uint32_t draw, const float* pyramid,
                                      const float* out_w, const float* out_b, const float* wa, const float* wb, const float* wc,
                                      float* packed, void* zero_ptr, int64_t zero_bytes, float* dense_dst, const float* dense_src,
                                      int64_t dense_n, int B, int F, int Fn, int T, void* stream) {
  FDBM_CHECK(rng, "fdbm_step_boundary_rng: null rng state");
  return step_boundary_impl(x, y, nullptr, rng, draw, pyramid, out_w, out_b, wa, wb, wc, packed, zero_ptr, zero_bytes, dense_dst,
                            dense_src, dense_n, B, F, Fn, T, stream);
}

// ---------------------------------------------------------------------------------
// predictor / corrector moves
// ---------------------------------------------------------------------------------
// RNG (here and in the corrector): float i of sample bi is component (i & 1) of complex element (bi nfloat + i) / 2 of the draw
template <bool RNG>
__global__ void __launch_bounds__(256) pc_predictor_kernel(
    float* __restrict__ xn, float* __restrict__ xm, const float* __restrict__ x,
    const float* __restrict__ s, const float* __restrict__ y, const float* __restrict__ z, const uint32_t* __restrict__ rng, uint32_t draw,
    const float* wx, const float* ws, const float* wy, const float* gd, float dt, float sq,
    int64_t nfloat) {
  uint32_t s_lo = 0, s_hi = 0, dr = 0;
  if constexpr (RNG) { s_lo = rng[0]; s_hi = rng[1]; dr = rng[2] + draw; }
  const int bi = blockIdx.y;
  const float fx = wx[bi], fs = ws[bi], fy = wy[bi];
  const float gz = __fmul_rn(gd[bi], sq);
  const int64_t base = (int64_t)bi * nfloat;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nfloat;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float xv = x[base + i];
    float drift = __fadd_rn(__fadd_rn(__fmul_rn(fx, xv), __fmul_rn(fs, s[base + i])),
                            __fmul_rn(fy, y[base + i]));
    float mean = __fadd_rn(xv, __fmul_rn(drift, dt));
    xm[base + i] = mean;
    float zv;
    if constexpr (RNG) zv = rng_complex_normal((base + i) >> 1, dr, s_lo, s_hi)[i & 1]; else zv = z[base + i];
    xn[base + i] = __fadd_rn(mean, __fmul_rn(gz, zv));
  }
}

static int pc_predictor_impl(void* x_new, void* x_mean, const void* x, const void* s,
                             const void* y, const void* z, const uint32_t* rng, uint32_t draw, const float* wx, const float* ws,
                             const float* wy, const float* gd, float dt, int B,
                             int64_t n_complex, void* stream) {
  FDBM_CHECK(x_new && x_mean && x && s && y && (z || rng) && wx && ws && wy && gd, "fdbm_pc_predictor: null pointer");
  FDBM_CHECK(dt <= 0.f, "fdbm_pc_predictor: dt must be <= 0 (reverse time), got %g", dt);
  const int64_t nfloat = 2 * n_complex;
  int gx = (int)((nfloat + 255) / 256);
  if (gx > 2048) gx = 2048;
  if (gx < 1) gx = 1;
  if (rng)
    pc_predictor_kernel<true><<<dim3(gx, B), 256, 0, (hipStream_t)stream>>>(
        (float*)x_new, (float*)x_mean, (const float*)x, (const float*)s, (const float*)y,
        nullptr, rng, draw, wx, ws, wy, gd, dt, sqrtf(-dt), nfloat);
  else
    pc_predictor_kernel<false><<<dim3(gx, B), 256, 0, (hipStream_t)stream>>>(
        (float*)x_new, (float*)x_mean, (const float*)x, (const float*)s, (const float*)y,
        (const float*)z, nullptr, 0u, wx, ws, wy, gd, dt, sqrtf(-dt), nfloat);
  FDBM_LAUNCH_CHECK("fdbm_pc_predictor");
  return 0;
}

extern "C" int fdbm_pc_predictor(void* x_new, void* x_mean, const void* x, const void* s,
                                 const void* y, const void* z, const float* wx, const float* ws,
                                 const float* wy, const float* gd, float dt, int B,
                                 int64_t n_complex, void* stream) {
  return pc_predictor_impl(x_new, x_mean, x, s, y, z, nullptr, 0u, wx, ws, wy, gd, dt, B, n_complex, stream);
}

extern "C" int fdbm_pc_predictor_rng(void* x_new, void* x_mean, const void* x, const void* s, const void* y,
                                     const uint32_t* rng, uint32_t draw, const float* wx, const float* ws,
                                     const float* wy, const float* gd, float dt, int B, int64_t n_complex, void* stream) {
  FDBM_CHECK(rng, "fdbm_pc_predictor_rng: null rng state");
  return pc_predictor_impl(x_new, x_mean, x, s, y, nullptr, rng, draw, wx, ws, wy, gd, dt, B, n_complex, stream);
}

template <bool RNG>
__global__ void __launch_bounds__(256) pc_corrector_kernel(
    float* __restrict__ xn, float* __restrict__ xm, const float* __restrict__ x,
    const float* __restrict__ s, const float* __restrict__ y, const float* __restrict__ nz, const uint32_t* __restrict__ rng, uint32_t draw,
    const float* a, const float* b, const float* den, const float* step, const float* nscale,
    int64_t nfloat) {
  uint32_t s_lo = 0, s_hi = 0, dr = 0;
  if constexpr (RNG) { s_lo = rng[0]; s_hi = rng[1]; dr = rng[2] + draw; }
  const int bi = blockIdx.y;
  const float fa = a[bi], fb = b[bi], fd = den[bi], fst = step[bi], fn = nscale[bi];
  const int64_t base = (int64_t)bi * nfloat;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nfloat;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float xv = x[base + i];
    float mean = __fadd_rn(__fmul_rn(fa, s[base + i]), __fmul_rn(fb, y[base + i]));
    float score = __fdiv_rn(-__fsub_rn(xv, mean), fd);
    float m = __fadd_rn(xv, __fmul_rn(fst, score));
    xm[base + i] = m;
    float zv;
    if constexpr (RNG) zv = rng_complex_normal((base + i) >> 1, dr, s_lo, s_hi)[i & 1]; else zv = nz[base + i];
    xn[base + i] = __fadd_rn(m, __fmul_rn(zv, fn));
  }
}

static int pc_corrector_impl(void* x_new, void* x_mean, const void* x, const void* s,
                             const void* y, const void* noise, const uint32_t* rng, uint32_t draw, const float* a, const float* b,
                             const float* den, const float* step, const float* noise_scale,
                             int B, int64_t n_complex, void* stream) {
  FDBM_CHECK(x_new && x_mean && x && s && y && (noise || rng) && a && b && den && step && noise_scale,
             "fdbm_pc_corrector: null pointer");
  const int64_t nfloat = 2 * n_complex;
  int gx = (int)((nfloat + 255) / 256);
  if (gx > 2048) gx = 2048;
  if (gx < 1) gx = 1;
  if (rng)
    pc_corrector_kernel<true><<<dim3(gx, B), 256, 0, (hipStream_t)stream>>>(
        (float*)x_new, (float*)x_mean, (const float*)x, (const float*)s, (const float*)y,
        nullptr, rng, draw, a, b, den, step, noise_scale, nfloat);
  else
    pc_corrector_kernel<false><<<dim3(gx, B), 256, 0, (hipStream_t)stream>>>(
        (float*)x_new, (float*)x_mean, (const float*)x, (const float*)s, (const float*)y,
        (const float*)noise, nullptr, 0u, a, b, den, step, noise_scale, nfloat);
  FDBM_LAUNCH_CHECK("fdbm_pc_corrector");
  return 0;
}

extern "C" int fdbm_pc_corrector(void* x_new, void* x_mean, const void* x, const void* s,
                                 const void* y, const void* noise, const float* a, const float* b,
                                 const float* den, const float* step, const float* noise_scale,
                                 int B, int64_t n_complex, void* stream) {
  return pc_corrector_impl(x_new, x_mean, x, s, y, noise, nullptr, 0u, a, b, den, step, noise_scale, B, n_complex, stream);
}

extern "C" int fdbm_pc_corrector_rng(void* x_new, void* x_mean, const void* x, const void* s, const void* y,
                                     const uint32_t* rng, uint32_t draw, const float* a, const float* b,
                                     const float* den, const float* step, const float* noise_scale,
                                     int B, int64_t n_complex, void* stream) {
  FDBM_CHECK(rng, "fdbm_pc_corrector_rng: null rng state");
  return pc_corrector_impl(x_new, x_mean, x, s, y, nullptr, rng, draw, a, b, den, step, noise_scale, B, n_complex, stream);
}

// Langevin corrector step size ON THE DEVICE (fdbm/util/correctors.py:46-51): grad_norm = mean_b ||score_b||,
// noise_norm = mean_b ||noise_b||, step = (snr * noise_norm / (grad_norm + 1e-8))^2 * 2 for every sample,
// noise_scale = sqrt(step * 2).  The host version fetched two norms per step (a device synchronisation inside the
// sampler loop, and nothing a HIP graph can capture); here two launches leave step[B] / noise_scale[B] where
// fdbm_pc_corrector reads them.  Norms in fp64 partial sums (the reference: fp32 torch.norm).
__global__ void __launch_bounds__(256) langevin_norm_kernel(double* __restrict__ part, const float* __restrict__ x,
                                                            const float* __restrict__ s, const float* __restrict__ y,
                                                            const float* __restrict__ nz, const float* a, const float* b,
                                                            const float* den, int64_t nfloat) {
  __shared__ double red[2][256];
  const int bi = blockIdx.y;
  const float fa = a[bi], fb = b[bi], fd = den[bi];
  const int64_t base = (int64_t)bi * nfloat;
  double g2 = 0.0, n2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nfloat; i += (int64_t)gridDim.x * blockDim.x) {
    const float mean = __fadd_rn(__fmul_rn(fa, s[base + i]), __fmul_rn(fb, y[base + i]));
    const float score = __fdiv_rn(-__fsub_rn(x[base + i], mean), fd);
    const float z = nz[base + i];
    g2 += (double)score * (double)score;
    n2 += (double)z * (double)z;
  }
  red[0][threadIdx.x] = g2; red[1][threadIdx.x] = n2;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) { red[0][threadIdx.x] += red[0][threadIdx.x + st]; red[1][threadIdx.x] += red[1][threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[((int64_t)bi * gridDim.x + blockIdx.x) * 2] = red[0][0];
    part[((int64_t)bi * gridDim.x + blockIdx.x) * 2 + 1] = red[1][0];
  }
}

__global__ void __launch_bounds__(64) langevin_step_kernel(float* __restrict__ step, float* __restrict__ nscale,
                                                           const double* __restrict__ part, int B, int gx, float snr) {
  // one wave: lane l sums samples l, l + 64, ...
  double gsum = 0.0, nsum = 0.0;
  for (int bi = threadIdx.x; bi < B; bi += 64) {
    double g2 = 0.0, n2 = 0.0;
    for (int k = 0; k < gx; ++k) { g2 += part[((int64_t)bi * gx + k) * 2]; n2 += part[((int64_t)bi * gx + k) * 2 + 1]; }
    gsum += (double)(float)sqrt(g2);          // per-sample norms are fp32 tensors in the reference
    nsum += (double)(float)sqrt(n2);
  }
  for (int o = 32; o > 0; o >>= 1) { gsum += __shfl_xor(gsum, o, 64); nsum += __shfl_xor(nsum, o, 64); }
  const float gmean = (float)(gsum / (double)B), nmean = (float)(nsum / (double)B);
  const float r = __fdiv_rn(__fmul_rn(snr, nmean), __fadd_rn(gmean, 1e-8f));
  const float st = __fmul_rn(__fmul_rn(r, r), 2.0f);
  const float ns = sqrtf(__fmul_rn(st, 2.0f));
  for (int bi = threadIdx.x; bi < B; bi += 64) { step[bi] = st; nscale[bi] = ns; }
}

extern "C" int fdbm_langevin_step(float* step, float* noise_scale, void* scratch, const void* x, const void* s,
                                  const void* y, const void* noise, const float* a, const float* b, const float* den,
                                  float snr, int B, int64_t n_complex, void* stream) {
  FDBM_CHECK(step && noise_scale && scratch && x && s && y && noise && a && b && den, "fdbm_langevin_step: null pointer");
  FDBM_CHECK(B > 0 && n_complex > 0, "fdbm_langevin_step: bad shape");
  const int64_t nfloat = n_complex * 2;
  const int gx = 64;                        // scratch: B * 64 * 2 doubles
  langevin_norm_kernel<<<dim3(gx, B), 256, 0, (hipStream_t)stream>>>((double*)scratch, (const float*)x, (const float*)s,
                                                                   (const float*)y, (const float*)noise, a, b, den, nfloat);
  FDBM_LAUNCH_CHECK("fdbm_langevin_step/norms");
  langevin_step_kernel<<<1, 64, 0, (hipStream_t)stream>>>(step, noise_scale, (const double*)scratch, B, gx, snr);
  FDBM_LAUNCH_CHECK("fdbm_langevin_step/step");
  return 0;
}

// ---------------------------------------------------------------------------------
// ode_sampler_int on the device (fdbm/bridge.py:115-140 -> scipy.integrate.solve_ivp(RK45); restated in
// fdbm_amd/odeint.py): the stage arithmetic of one Dormand-Prince step over the complex128 state, one pass each
// instead of ~25 stock elementwise launches per step.
//   fdbm_rk45_lincomb   out = y + scale * (c_0 K_0 + c_1 K_1 + ...)   (terms added in index order; zero coefficients skipped)
//                       - the stage inputs y + h sum_j a_sj K_j (c_j = a_sj h, scale = 1) and y_new = y + h sum_j b_j K_j
//   fdbm_rk45_error     err = sum_j e_j K_j;  partial[blk] = sum over the block's elements of |err h / (atol + rtol max(|y|, |y_new|))|^2
//                       (SciPy's error norm is sqrt(sum(partial) / n); 256 partials, summed by the caller in index order)
// ---------------------------------------------------------------------------------
struct Rk45Args {
  const double2* K[7];
  double c[7];
  int nk;
};

__global__ void __launch_bounds__(256) rk45_lincomb_kernel(double2* __restrict__ out, const double2* __restrict__ y, const Rk45Args a,
                                                           double scale, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double re = 0.0, im = 0.0;
    bool first = true;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      if (j < a.nk && a.c[j] != 0.0) {
        const double2 k = a.K[j][i];
        if (first) { re = k.x * a.c[j]; im = k.y * a.c[j]; first = false; }
        else { re = __dadd_rn(re, __dmul_rn(k.x, a.c[j])); im = __dadd_rn(im, __dmul_rn(k.y, a.c[j])); }
      }
    }
    const double2 yv = y[i];
    out[i] = double2{__dadd_rn(yv.x, __dmul_rn(scale, re)), __dadd_rn(yv.y, __dmul_rn(scale, im))};
  }
}

extern "C" int fdbm_rk45_lincomb(void* out, const void* y, const void* const* K, const double* coef, int nk, double scale,
                                 int64_t n_complex, void* stream) {
  FDBM_CHECK(out && y && K && coef && nk >= 1 && nk <= 7 && n_complex > 0, "fdbm_rk45_lincomb: bad arguments (nk=%d)", nk);
  Rk45Args a;
  memset(&a, 0, sizeof(a));
  a.nk = nk;
  for (int j = 0; j < nk; ++j) {
    FDBM_CHECK(K[j] || coef[j] == 0.0, "fdbm_rk45_lincomb: stage %d is NULL with a non-zero coefficient", j);
    a.K[j] = (const double2*)K[j];
    a.c[j] = K[j] ? coef[j] : 0.0;
  }
  int g = (int)((n_complex + 255) / 256);
  if (g > 2048) g = 2048;
  rk45_lincomb_kernel<<<g, 256, 0, (hipStream_t)stream>>>((double2*)out, (const double2*)y, a, scale, n_complex);
  FDBM_LAUNCH_CHECK("fdbm_rk45_lincomb");
  return 0;
}

__global__ void __launch_bounds__(256) rk45_error_kernel(double* __restrict__ partial, const Rk45Args a, const double2* __restrict__ y,
                                                         const double2* __restrict__ yn, double h, double atol, double rtol, int64_t n) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double re = 0.0, im = 0.0;
    bool first = true;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      if (a.c[j] != 0.0) {
        const double2 k = a.K[j][i];
        if (first) { re = k.x * a.c[j]; im = k.y * a.c[j]; first = false; }
        else { re = __dadd_rn(re, __dmul_rn(k.x, a.c[j])); im = __dadd_rn(im, __dmul_rn(k.y, a.c[j])); }
      }
    }
    const double2 a0 = y[i], a1 = yn[i];
    const double m0 = sqrt(a0.x * a0.x + a0.y * a0.y), m1 = sqrt(a1.x * a1.x + a1.y * a1.y);
    const double sc = atol + (m0 > m1 ? m0 : m1) * rtol;
    const double er = re * h / sc, ei = im * h / sc;
    acc += er * er + ei * ei;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

extern "C" int fdbm_rk45_error(double* partial, const void* const* K, const double* e, const void* y, const void* y_new, double h,
                               double atol, double rtol, int64_t n_complex, void* stream) {
  FDBM_CHECK(partial && K && e && y && y_new && n_complex > 0, "fdbm_rk45_error: bad arguments");
  Rk45Args a;
  memset(&a, 0, sizeof(a));
  a.nk = 7;
  for (int j = 0; j < 7; ++j) {
    FDBM_CHECK(K[j] || e[j] == 0.0, "fdbm_rk45_error: stage %d is NULL with a non-zero coefficient", j);
    a.K[j] = (const double2*)K[j];
    a.c[j] = K[j] ? e[j] : 0.0;
  }
  rk45_error_kernel<<<256, 256, 0, (hipStream_t)stream>>>(partial, a, (const double2*)y, (const double2*)y_new, h, atol, rtol, n_complex);
  FDBM_LAUNCH_CHECK("fdbm_rk45_error");
  return 0;
}

// ---------------------------------------------------------------------------------
// network input / output
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_input_kernel(f32x4* __restrict__ out,
                                                         const f32x2* __restrict__ x,
                                                         const f32x2* __restrict__ y, int F, int Fn,
                                                         int T, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const int64_t r = i / T;
    const int f = (int)(r % Fn);
    const int64_t b = r / Fn;
    const int64_t src = (b * F + f) * T + t;
    const f32x2 xv = x[src], yv = y[src];
    f32x4 o = {xv[0], xv[1], yv[0], yv[1]};
    out[i] = o;
  }
}

extern "C" int fdbm_pack_input(float* out, const void* x, const void* y, int B, int F, int Fn, int T,
                               void* stream) {
  FDBM_CHECK(out && x && y, "fdbm_pack_input: null pointer");
  FDBM_CHECK(Fn <= F && Fn > 0 && T > 0 && B > 0, "fdbm_pack_input: bad shape");
  const int64_t total = (int64_t)B * Fn * T;
  int g = (int)((total + 255) / 256);
  if (g > 4096) g = 4096;
  pack_input_kernel<<<g, 256, 0, (hipStream_t)stream>>>((f32x4*)out, (const f32x2*)x,
                                                        (const f32x2*)y, F, Fn, T, total);
  FDBM_LAUNCH_CHECK("fdbm_pack_input");
  return 0;
}

__global__ void __launch_bounds__(256) unpack_output_kernel(f32x2* __restrict__ out,
                                                            const f32x4* __restrict__ pyr,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ bias, int F,
                                                            int Fn, int T, int64_t total) {
  const float w00 = w[0], w01 = w[1], w02 = w[2], w03 = w[3];
  const float w10 = w[4], w11 = w[5], w12 = w[6], w13 = w[7];
  const float b0 = bias[0], b1 = bias[1];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const int64_t r = i / T;
    const int f = (int)(r % F);
    const int64_t b = r / F;
    f32x2 o = {0.f, 0.f};
    if (f < Fn) {
      const f32x4 p = pyr[(b * Fn + f) * T + t];
      o[0] = b0 + (((w00 * p[0] + w01 * p[1]) + w02 * p[2]) + w03 * p[3]);
      o[1] = b1 + (((w10 * p[0] + w11 * p[1]) + w12 * p[2]) + w13 * p[3]);
    }
    out[i] = o;
  }
}

extern "C" int fdbm_unpack_output(void* out, const float* pyr, const float* w, const float* b, int B,
                                  int F, int Fn, int T, void* stream) {
  FDBM_CHECK(out && pyr && w && b, "fdbm_unpack_output: null pointer");
  const int64_t total = (int64_t)B * F * T;
  int g = (int)((total + 255) / 256);
  if (g > 4096) g = 4096;
  unpack_output_kernel<<<g, 256, 0, (hipStream_t)stream>>>((f32x2*)out, (const f32x4*)pyr, w, b, F,
                                                           Fn, T, total);
  FDBM_LAUNCH_CHECK("fdbm_unpack_output");
  return 0;
}

// ---------------------------------------------------------------------------------
// Combine('sum'): out = h + bias + W[C][4] * pyr[4]
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) combine_kernel(T* __restrict__ out, const T* __restrict__ h,
                                                      const f32x4* __restrict__ pyr,
                                                      const f32x4* __restrict__ w,
                                                      const float* __restrict__ bias, int64_t M,
                                                      int C) {
  constexpr int VW = DT<T>::vecw;
  const int nvec = C / VW;
  const int64_t total = M * nvec;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % nvec);
    const int64_t m = i / nvec;
    const f32x4 p = pyr[m];
    float hv[VW];
    Vec16<T>::load(h + m * C + v * VW, hv);
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      const int c = v * VW + k;
      const f32x4 wc = w[c];
      hv[k] = hv[k] + (bias[c] + (((wc[0] * p[0] + wc[1] * p[1]) + wc[2] * p[2]) + wc[3] * p[3]));
    }
    Vec16<T>::store(out + m * C + v * VW, hv);
  }
}

extern "C" int fdbm_combine(void* out, const void* h, const float* pyr, const float* w,
                            const float* bias, int64_t M, int C, int dtype, void* stream) {
  FDBM_CHECK(out && h && pyr && w && bias, "fdbm_combine: null pointer");
  FDBM_CHECK(C % 8 == 0, "fdbm_combine: C=%d must be a multiple of 8", C);
  const int vw = dtype != FDBM_F32 ? 8 : 4;
  const int64_t total = M * (C / vw);
  int g = (int)((total + 255) / 256);
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FDBM_BF16)
    combine_kernel<bf16_t><<<g, 256, 0, st>>>((bf16_t*)out, (const bf16_t*)h, (const f32x4*)pyr,
                                              (const f32x4*)w, bias, M, C);
  else if (dtype == FDBM_F16)
    combine_kernel<f16_t><<<g, 256, 0, st>>>((f16_t*)out, (const f16_t*)h, (const f32x4*)pyr,
                                             (const f32x4*)w, bias, M, C);
  else if (dtype == FDBM_F32)
    combine_kernel<float><<<g, 256, 0, st>>>((float*)out, (const float*)h, (const f32x4*)pyr,
                                             (const f32x4*)w, bias, M, C);
  else
    FDBM_CHECK(false, "fdbm_combine: bad dtype %d", dtype);
  FDBM_LAUNCH_CHECK("fdbm_combine");
  return 0;
}

// ---------------------------------------------------------------------------------
// time embedding: dense rows (one wave per output row), optional Fourier front
// ---------------------------------------------------------------------------------
#define DENSE_BCHUNK 8
template <bool FOURIER, bool SILU>
__global__ void __launch_bounds__(256) dense_rows_kernel(float* __restrict__ out,
                                                         const float* __restrict__ act,
                                                         const float* __restrict__ t,
                                                         const float* __restrict__ fw,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ bias, int B, int R,
                                                         int K) {
  extern __shared__ float s_act[];  // [DENSE_BCHUNK][K]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  for (int b0 = 0; b0 < B; b0 += DENSE_BCHUNK) {
    const int nb = min(DENSE_BCHUNK, B - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < nb * K; i += blockDim.x) {
      const int bb = i / K, k = i % K;
      float v;
      if (FOURIER) {
        const int nf = K / 2;
        const int j = k < nf ? k : k - nf;
        // ((log t * W) * 2) * pi, every product rounded to fp32 (layerspp.py:40)
        const float arg = __fmul_rn(__fmul_rn(__fmul_rn(t[b0 + bb], fw[j]), 2.0f), 3.14159274101257324f);
        v = k < nf ? sinf(arg) : cosf(arg);
      } else {
        v = act[(int64_t)(b0 + bb) * K + k];
      }
      s_act[i] = v;
    }
    __syncthreads();
    if (r < R) {
      float acc[DENSE_BCHUNK];
#pragma unroll
      for (int bb = 0; bb < DENSE_BCHUNK; ++bb) acc[bb] = 0.f;
      for (int k = lane; k < K; k += 64) {
        const float wv = w[(int64_t)r * K + k];
#pragma unroll
        for (int bb = 0; bb < DENSE_BCHUNK; ++bb)
          if (bb < nb) acc[bb] += wv * s_act[bb * K + k];
      }
#pragma unroll
      for (int bb = 0; bb < DENSE_BCHUNK; ++bb) {
        float v = acc[bb];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0 && bb < nb) {
          v += bias[r];
          out[(int64_t)(b0 + bb) * R + r] = SILU ? silu_f(v) : v;
        }
      }
    }
  }
}

extern "C" int fdbm_dense_rows(float* out, const float* act, const float* w, const float* bias,
                               int B, int R, int K, void* stream) {
  FDBM_CHECK(out && act && w && bias, "fdbm_dense_rows: null pointer");
  FDBM_CHECK(K * DENSE_BCHUNK * 4 <= 65536, "fdbm_dense_rows: K=%d too large", K);
  dense_rows_kernel<false, false><<<cdiv(R, 4), 256, DENSE_BCHUNK * K * sizeof(float),
                                    (hipStream_t)stream>>>(out, act, nullptr, nullptr, w, bias, B, R, K);
  FDBM_LAUNCH_CHECK("fdbm_dense_rows");
  return 0;
}

extern "C" int fdbm_temb(float* out_act, const float* log_t, const float* fourier_w, const float* w1,
                         const float* b1, const float* w2, const float* b2, float* scratch, int B,
                         int nf, void* stream) {
  const float* t = log_t;
  FDBM_CHECK(out_act && t && fourier_w && w1 && b1 && w2 && b2 && scratch, "fdbm_temb: null pointer");
  const int E = 2 * nf, D = 4 * nf;
  FDBM_CHECK(D * DENSE_BCHUNK * 4 <= 65536, "fdbm_temb: nf=%d too large", nf);
  hipStream_t st = (hipStream_t)stream;
  // scratch[b][D] = silu(W1 * fourier(log t) + b1)
  dense_rows_kernel<true, true><<<cdiv(D, 4), 256, DENSE_BCHUNK * E * sizeof(float), st>>>(
      scratch, nullptr, t, fourier_w, w1, b1, B, D, E);
  FDBM_LAUNCH_CHECK("fdbm_temb/1");
  // out_act[b][D] = silu(W2 * scratch + b2)   (consumers all take act(temb))
  dense_rows_kernel<false, true><<<cdiv(D, 4), 256, DENSE_BCHUNK * D * sizeof(float), st>>>(
      out_act, scratch, nullptr, nullptr, w2, b2, B, D, D);
  FDBM_LAUNCH_CHECK("fdbm_temb/2");
  return 0;
}

// ---------------------------------------------------------------------------------
// stem: conv3x3 4 -> nf, direct (K = 36 is far too small for MFMA tiles; HBM-bound
// on the nf-channel output write)
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) conv_stem_kernel(T* __restrict__ out,
                                                        const f32x4* __restrict__ in,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias, int H,
                                                        int W, int nf, double* __restrict__ stat_out,
                                                        int stat_nsplit) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];  // [36][nf] (k-major) + [nf] bias + [nf/4][2] statistics
  float* s_st = s_w + nf * 37;
  for (int i = threadIdx.x; i < nf * 36; i += blockDim.x) {
    const int oc = i / 36, k = i % 36;
    s_w[k * nf + oc] = w[i];
  }
  for (int i = threadIdx.x; i < nf; i += blockDim.x) s_w[nf * 36 + i] = bias[i];
  if (stat_out)
    for (int i = threadIdx.x; i < nf / 2; i += blockDim.x) s_st[i] = 0.f;
  __syncthreads();
  const int ncg = nf / 8;   // 8 output channels per thread; the grid stride is a multiple of ncg: cg is fixed per thread
  const int64_t b = blockIdx.y;
  const int64_t total = (int64_t)H * W * ncg;
  float u1[2] = {0.f, 0.f}, u2[2] = {0.f, 0.f};     // (sum, sumsq) of this thread's two 4-channel units
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
  const int cg = (int)(idx % ncg);
  const int64_t p = idx / ncg;
  const int x = (int)(p % W);
  const int y = (int)(p / W);
  float tap[36];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = y + ky - 1, ix = x + kx - 1;
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const int iyc = min(max(iy, 0), H - 1), ixc = min(max(ix, 0), W - 1);
      f32x4 v = in[(b * H + iyc) * W + ixc];
      if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
      const int k = (ky * 3 + kx) * 4;
      tap[k] = v[0]; tap[k + 1] = v[1]; tap[k + 2] = v[2]; tap[k + 3] = v[3];
    }
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = 0.f;
  // consecutive threads read consecutive 32-byte groups of a k-row: conflict-free b128 reads
#pragma unroll
  for (int k = 0; k < 36; ++k) {
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(s_w + k * nf + cg * 8);
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(s_w + k * nf + cg * 8 + 4);
    const float t = tap[k];
    o[0] += w0[0] * t; o[1] += w0[1] * t; o[2] += w0[2] * t; o[3] += w0[3] * t;
    o[4] += w1[0] * t; o[5] += w1[1] * t; o[6] += w1[2] * t; o[7] += w1[3] * t;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] += s_w[nf * 36 + cg * 8 + j];
  T* dst = out + (b * H * W + p) * nf + cg * 8;
  if constexpr (sizeof(T) == 2) {
    Vec16<T>::store(dst, o);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)(T)o[j];       // statistics of the STORED tensor
  } else {
    Vec16<T>::store(dst, o);
    Vec16<T>::store(dst + 4, o + 4);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    u1[h] += (o[4 * h] + o[4 * h + 1]) + (o[4 * h + 2] + o[4 * h + 3]);
    u2[h] += (o[4 * h] * o[4 * h] + o[4 * h + 1] * o[4 * h + 1]) + (o[4 * h + 2] * o[4 * h + 2] + o[4 * h + 3] * o[4 * h + 3]);
  }
  }
  if (stat_out) {
    // unit statistics [B][stat_nsplit][nf/4][2] (fdbm_conv_args.stat_out layout with stat_G = nf/4)
    const int cg = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) % ncg);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      atomicAdd(&s_st[(cg * 2 + h) * 2], u1[h]);
      atomicAdd(&s_st[(cg * 2 + h) * 2 + 1], u2[h]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nf / 2; i += blockDim.x)
      atomicAdd(stat_out + ((b * stat_nsplit + blockIdx.x % stat_nsplit) * (nf / 4)) * 2 + i, (double)s_st[i]);
  }
}

// Matrix-core stem (W % 16 == 0, nf = 16 NT): the 4 input channels of one tap are exactly the K = 4 of
// v_mfma_f32_16x16x4_f32, so a 3x3 x 4-channel stem is 9 MFMA k-steps per (16 pixels x 16 channels) tile
// in full fp32 (an exact fma chain - the parity mode uses it too).  A wave keeps its 9 x NT weight
// operands in registers and walks over 16-pixel row segments; the input operand of a tap is one float
// per lane (pixel = lane & 15, channel = lane >> 4), read straight from the packed NHWC4 input.
template <typename T, int NT>
__global__ void __launch_bounds__(256) conv_stem_mfma_kernel(T* __restrict__ out, const float* __restrict__ in,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias, int H, int W,
                                                             double* __restrict__ stat_out, int stat_nsplit,
                                                             int tiles_per_wave) {
  constexpr int nf = 16 * NT;
  __shared__ float s_w[nf * 36 + nf];
  __shared__ double s_st[nf / 2];
  for (int i = threadIdx.x; i < nf * 36; i += 256) s_w[i] = w[i];
  for (int i = threadIdx.x; i < nf; i += 256) s_w[nf * 36 + i] = bias[i];
  if (threadIdx.x < nf / 2) s_st[threadIdx.x] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int frow = lane & 15, fk = lane >> 4;
  float wf[NT][9];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) wf[j][t] = s_w[(j * 16 + frow) * 36 + t * 4 + fk];
  f32x4 bv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bv[j] = *reinterpret_cast<const f32x4*>(s_w + nf * 36 + j * 16 + fk * 4);
  const int b = blockIdx.y;
  const int tiles_x = W / 16;
  const int ntiles = H * tiles_x;
  const float* img = in + (int64_t)b * H * W * 4;
  float u1[NT], u2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) u1[j] = u2[j] = 0.f;
  const int t0 = (blockIdx.x * 4 + wave) * tiles_per_wave;
  for (int t = t0; t < min(ntiles, t0 + tiles_per_wave); ++t) {
    const int y = t / tiles_x, x = (t - y * tiles_x) * 16 + frow;
    float v[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const float q = img[((int64_t)min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 1)) * 4 + fk];
      v[tap] = ok ? q : 0.f;
    }
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][tap], v[tap], acc[j], 0, 0, 0);
    T* dst = out + (((int64_t)b * H + y) * W + x) * nf + fk * 4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float o[4] = {acc[j][0] + bv[j][0], acc[j][1] + bv[j][1], acc[j][2] + bv[j][2], acc[j][3] + bv[j][3]};
      OutVec<T>::store(dst + j * 16, o);
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (float)(T)o[r];       // statistics of the STORED tensor
      }
      u1[j] += (o[0] + o[1]) + (o[2] + o[3]);
      u2[j] += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
    }
  }
  if (stat_out) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float r1 = row16_sum(u1[j]), r2 = row16_sum(u2[j]);     // over the 16 pixels of the tile row
      if (frow == 0) {
        atomicAdd(&s_st[(j * 4 + fk) * 2], (double)r1);
        atomicAdd(&s_st[(j * 4 + fk) * 2 + 1], (double)r2);
      }
    }
    __syncthreads();
    if (threadIdx.x < nf / 2)
      atomicAdd(stat_out + (((int64_t)b * stat_nsplit + blockIdx.x % stat_nsplit) * (nf / 4)) * 2 + threadIdx.x, s_st[threadIdx.x]);
  }
}

extern "C" int fdbm_conv_stem(void* out, const float* in, const float* w, const float* bias, int B,
                              int H, int W, int nf, int dt_out, void* stream) {
  return fdbm_conv_stem_stats(out, in, w, bias, B, H, W, nf, dt_out, nullptr, 0, stream);
}

extern "C" int fdbm_conv_stem_stats(void* out, const float* in, const float* w, const float* bias, int B,
                                    int H, int W, int nf, int dt_out, double* stat_out, int stat_nsplit,
                                    void* stream) {
  FDBM_CHECK(out && in && w && bias, "fdbm_conv_stem: null pointer");
  FDBM_CHECK(nf % 8 == 0 && nf > 0 && nf <= 256, "fdbm_conv_stem: nf=%d must be a multiple of 8, <= 256", nf);
  FDBM_CHECK(!stat_out || stat_nsplit >= 1, "fdbm_conv_stem: stat_nsplit must be >= 1");
  hipStream_t st = (hipStream_t)stream;
  if (W % 16 == 0 && (nf == 32 || nf == 64 || nf == 96 || nf == 128)) {     // matrix-core variant
    const int ntiles = H * (W / 16);
    int tpw = 4;                                                  // 16-pixel tiles per wave (weights loaded once per wave)
    while (tpw < 64 && (int64_t)B * cdiv(ntiles, 4 * tpw) > 2048) tpw *= 2;
    dim3 grid(cdiv(ntiles, 4 * tpw), B);
#define STEM_MFMA(TT, NTT) conv_stem_mfma_kernel<TT, NTT><<<grid, 256, 0, st>>>((TT*)out, in, w, bias, H, W, stat_out, stat_nsplit, tpw)
#define STEM_NT(TT) do { if (nf == 128) STEM_MFMA(TT, 8); else if (nf == 96) STEM_MFMA(TT, 6); else if (nf == 64) STEM_MFMA(TT, 4); else STEM_MFMA(TT, 2); } while (0)
    if (dt_out == FDBM_BF16) STEM_NT(bf16_t);
    else if (dt_out == FDBM_F16) STEM_NT(f16_t);
    else if (dt_out == FDBM_F32) STEM_NT(float);
    else FDBM_CHECK(false, "fdbm_conv_stem: bad dtype %d", dt_out);
#undef STEM_NT
#undef STEM_MFMA
    FDBM_LAUNCH_CHECK("fdbm_conv_stem(mfma)");
    return 0;
  }
  const size_t smem = (size_t)nf * 37 * sizeof(float) + (size_t)(nf / 2) * sizeof(float);
  const int64_t total = (int64_t)H * W * (nf / 8);
  int gs = cdiv(total, 256);
  int cap = 1024 / (B < 1 ? 1 : B);     // grid-stride: the [36][nf] weight staging is paid ~1024 times, not 4096
  if (cap < 64) cap = 64;
  if (gs > cap) gs = cap;
  // the stride gs * 256 must be a multiple of nf/8 (a thread keeps its channel group): 256 is, for nf <= 256... only
  // when nf/8 divides 256
  FDBM_CHECK(256 % (nf / 8) == 0 || !stat_out, "fdbm_conv_stem: statistics need nf/8 (%d) to divide 256", nf / 8);
  dim3 grid(gs, B);
  if (dt_out == FDBM_BF16)
    conv_stem_kernel<bf16_t><<<grid, 256, smem, st>>>((bf16_t*)out, (const f32x4*)in, w, bias, H, W, nf, stat_out, stat_nsplit);
  else if (dt_out == FDBM_F16)
    conv_stem_kernel<f16_t><<<grid, 256, smem, st>>>((f16_t*)out, (const f32x4*)in, w, bias, H, W, nf, stat_out, stat_nsplit);
  else if (dt_out == FDBM_F32)
    conv_stem_kernel<float><<<grid, 256, smem, st>>>((float*)out, (const f32x4*)in, w, bias, H, W, nf, stat_out, stat_nsplit);
  else
    FDBM_CHECK(false, "fdbm_conv_stem: bad dtype %d", dt_out);
  FDBM_LAUNCH_CHECK("fdbm_conv_stem");
  return 0;
}
