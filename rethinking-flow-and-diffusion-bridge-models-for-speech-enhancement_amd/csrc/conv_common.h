// conv_common.h - pieces shared by the convolution kernels (conv.hip: tap-outer implicit GEMM;
// conv_patch.hip: halo-patch 3x3 kernel): kernel parameter block, MFMA wrappers, epilogue.
#pragma once
#include <type_traits>

#include "common.h"

template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                  *reinterpret_cast<const bf16x8*>(&b), acc, 0, 0, 0);
  }
};
template <> struct Mfma<f16_t> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&a),
                                                 *reinterpret_cast<const f16x8*>(&b), acc, 0, 0, 0);
  }
};
template <> struct Mfma<float> {};        // (the f32 kernels call the 16x16x4 builtin directly)

template <typename TO> struct OutVec;
template <> struct OutVec<float> {
  __device__ static __forceinline__ void load(const float* p, float* v) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  __device__ static __forceinline__ void store(float* p, const float* v) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
};
template <typename H> struct OutVec16 {
  __device__ static __forceinline__ void load(const H* p, float* v) {
    typename V16<H>::x4 t = *reinterpret_cast<const typename V16<H>::x4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
  }
  __device__ static __forceinline__ void store(H* p, const float* v) {
    typename V16<H>::x4 t = {(H)v[0], (H)v[1], (H)v[2], (H)v[3]};
    *reinterpret_cast<typename V16<H>::x4*>(p) = t;
  }
};
template <> struct OutVec<bf16_t> : OutVec16<bf16_t> {};
template <> struct OutVec<f16_t> : OutVec16<f16_t> {};

struct ConvParams {
  fdbm_conv_seg seg[FDBM_MAX_SEG];
  int nseg;
  const void* w;
  const float* bias;
  const float* tbias;
  int tbias_stride;
  const void* res;
  const float* res_lo;   // residual at half resolution, f32 [B][H/2][W/2][Cout]: upsampled 2x in the epilogue
  float scale;
  void* out;
  int B, H, W, Cout, CoutPad;
  int nk;
  int ksplit;       // grid.z: k-steps are split over this many workgroups
  float* partial;   // fp32 slabs [ksplit][M][Cout] when ksplit > 1
  // GroupNorm(+SiLU) applied to the A operand while it is staged (segments with seg_gn >= 0;
  // seg_gn = channel offset of the segment inside the normalised, virtually concatenated input)
  const float* gn_sums;     // [B][gn_nsplit][gn_G][2] (sum, sumsq) or NULL
  const float* gn_gamma;
  const float* gn_beta;
  int gn_nsplit, gn_G, gn_C, gn_silu;
  double gn_inv_count;
  float gn_eps;
  int seg_gn[FDBM_MAX_SEG];
  // unit statistics (gn_unit != 0): per flagged segment s a buffer of DOUBLES [B][gn_unsp[s]][gn_ucnt[s]][2] of
  // (sum, sumsq) over units of 4 channels, as the convs' stat_out leaves them with stat_G = Cout/4;
  // gn_uoff[s] = first unit of the segment inside the normalised (virtually concatenated) input
  int gn_unit;
  const double* gn_useg[FDBM_MAX_SEG];
  int gn_unsp[FDBM_MAX_SEG], gn_uoff[FDBM_MAX_SEG], gn_ucnt[FDBM_MAX_SEG];
  // Combine('sum') folded into the epilogue: out += comb_b[n] + comb_w[n][0..3] . pyr[m][0..3]
  const float* comb_pyr;
  const float* comb_w;
  const float* comb_b;
  // (sum, sumsq) of the stored output per (image, group of Cout/stat_G channels), accumulated
  // with atomics into stat_out[B][stat_G][2] for the GroupNorm that consumes this tensor
  double* stat_out;  // fp64: the atomics' order then changes the sums by ~1e-16, not by fp32 last bits
  int stat_G;
  int stat_nsplit;   // stat_out is [B][stat_nsplit][stat_G][2]; a block adds into split blockIdx.x % nsplit
  // split-precision matrix mode (fdbm_conv_args.mma_mode 1): f32 tensors, operands as IEEE-half (hi, lo) pairs
  int mma_split;
  float acc_scale;   // 1 / (SPLIT_ACT_SCALE * weight scale): applied to the f32 sums before the epilogue
};

// ---- split-precision operands ---------------------------------------------------------------------------
// An f32 activation a is staged as hi = half(16 a), lo = half(16 a - hi): hi + lo carries 22 significant bits
// of 16 a (the factor keeps lo a NORMAL half down to |a| ~ 8e-3; below that its absolute error is < 2e-9).
// |16 a| is clamped to the largest half so that an outlier saturates instead of becoming an infinity.
// LDS / weight rows of 128 bytes hold 32 channels as [32 halves hi | 32 halves lo]: the 16-byte chunk c < 4 is the
// k-group c of v_mfma_f32_16x16x32_f16's operand (8 consecutive channels), chunk 4 + c the same k-group of lo.
#define SPLIT_ACT_SCALE 16.0f
__device__ __forceinline__ void split_f16x4(const f32x4& x, uint2& hi, uint2& lo) {
  f16x4 h, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float s = __builtin_amdgcn_fmed3f(x[q] * SPLIT_ACT_SCALE, -65504.f, 65504.f);
    h[q] = (f16_t)s;
    l[q] = (f16_t)(s - (float)h[q]);
  }
  hi = *reinterpret_cast<uint2*>(&h);
  lo = *reinterpret_cast<uint2*>(&l);
}

// sum over the 16 lanes of a DPP row (= the 16 pixels of an MFMA m-tile), every lane gets it:
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror - four VALU adds.
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
  return v;
}

#define CONV_MAX_NB 4          // images one M tile may touch when a GN prologue / stats are on
#define CONV_GN_MAXC 512

// final value of 4 consecutive output channels of pixel m: bias, time-embedding bias, residual,
// scale, Combine; stores and returns the stored (rounded) values in v.
// (loads of absent operands read this instead of sitting behind a branch, see conv_epilogue4)
static __device__ const float g_conv_zero[4] = {0.f, 0.f, 0.f, 0.f};

template <typename TO>
__device__ __forceinline__ void conv_epilogue4(const ConvParams& p, int64_t m, int64_t b, int n, float* v) {
  const int Cout = p.Cout;
  // The three common operands are fetched UNCONDITIONALLY (an absent one from a zero vector): a load behind a branch is
  // waited for on its own, and bias -> time bias -> residual became three serialised memory round trips per element.
  const float* bp = p.bias ? p.bias + n : g_conv_zero;
  const float* tp = p.tbias ? p.tbias + b * p.tbias_stride + n : g_conv_zero;
  const TO* rp = p.res ? reinterpret_cast<const TO*>(p.res) + m * Cout + n : reinterpret_cast<const TO*>(g_conv_zero);
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bp);
  const f32x4 tv = *reinterpret_cast<const f32x4*>(tp);
  float r[4];
  OutVec<TO>::load(rp, r);
  v[0] += bv[0]; v[1] += bv[1]; v[2] += bv[2]; v[3] += bv[3];
  v[0] += tv[0]; v[1] += tv[1]; v[2] += tv[2]; v[3] += tv[3];
  v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
  if (p.res_lo) {       // upsample_2d of the low-resolution residual, tap order of resample2x_kernel
    const int HW = p.H * p.W;
    const int rr = (int)(m - b * HW);
    const int y = rr / p.W, x = rr - y * p.W;
    const int H2 = p.H >> 1, W2 = p.W >> 1;
    const int iy = y >> 1, ix = x >> 1;
    int ys[2], xs[2];
    float wy[2], wx[2];
    if (y & 1) { ys[0] = iy; wy[0] = 0.75f; ys[1] = iy + 1; wy[1] = 0.25f; }
    else       { ys[0] = iy - 1; wy[0] = 0.25f; ys[1] = iy; wy[1] = 0.75f; }
    if (x & 1) { xs[0] = ix; wx[0] = 0.75f; xs[1] = ix + 1; wx[1] = 0.25f; }
    else       { xs[0] = ix - 1; wx[0] = 0.25f; xs[1] = ix; wx[1] = 0.75f; }
    float up[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      if (ys[a] < 0 || ys[a] >= H2) continue;
#pragma unroll
      for (int bx = 0; bx < 2; ++bx) {
        if (xs[bx] < 0 || xs[bx] >= W2) continue;
        const float wgt = wy[a] * wx[bx];
        const f32x4 q = *reinterpret_cast<const f32x4*>(p.res_lo + (((int64_t)b * H2 + ys[a]) * W2 + xs[bx]) * Cout + n);
        up[0] += wgt * q[0]; up[1] += wgt * q[1]; up[2] += wgt * q[2]; up[3] += wgt * q[3];
      }
    }
    v[0] += up[0]; v[1] += up[1]; v[2] += up[2]; v[3] += up[3];
  }
  v[0] *= p.scale; v[1] *= p.scale; v[2] *= p.scale; v[3] *= p.scale;
  if (p.comb_pyr) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(p.comb_pyr + m * 4);
    const f32x4 cb = *reinterpret_cast<const f32x4*>(p.comb_b + n);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(p.comb_w + (int64_t)(n + r) * 4);
      v[r] += cb[r] + (((w[0] * q[0] + w[1] * q[1]) + w[2] * q[2]) + w[3] * q[3]);
    }
  }
  OutVec<TO>::store(reinterpret_cast<TO*>(p.out) + m * Cout + n, v);
  if constexpr (sizeof(TO) == 2) {     // statistics are those of the STORED tensor
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (float)(TO)v[r];
  }
}

// a double moved between lanes by a DPP control (both halves), and the sum over the 2 | 4 | 8 | 16 adjacent lanes of a DPP row
// that share a statistics slot: every step adds two values that were formed the same way, so all lanes of the slot end
// with the bit-identical total (a + b == b + a)
template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lanes_sum_d(double v, int L) {
  if (L >= 2) v += dpp_mov_d<0xB1>(v);      // quad_perm [1,0,3,2]
  if (L >= 4) v += dpp_mov_d<0x4E>(v);      // quad_perm [2,3,0,1]
  if (L >= 8) v += dpp_mov_d<0x141>(v);     // row_half_mirror
  if (L >= 16) v += dpp_mov_d<0x140>(v);    // row_mirror
  return v;
}

// GroupNorm prologue table: per (image of this workgroup, channel) scale and shift from the partial
// (sum, sumsq) rows the producer left, as TWO arrays (scale at s_gn[i], shift at s_gn[shoff + i],
// i = bl * gn_C + c; adjacent channels adjacent, so that the transform runs on packed-f32 instructions):
// scale = rstd * gamma, shift = beta - mean * rstd * gamma.
// An (image, group) slot belongs to L ADJACENT lanes (L = 16 | 8 | 4 | 2: NTHR / (32 nb) rounded down to a power of
// two): each lane sums every L-th of the slot's partial rows / (unit, row) items in fp64 with two loads in flight, a DPP
// reduction (lanes_sum_d) leaves the same total in all of them, each computes mean and 1 / sqrt and writes the scale / shift
// of every L-th channel of the group - fixed order, ONE barrier, no LDS partials.  (Round 2's form - partials to LDS, a
// barrier, one thread per slot summing them, a barrier, the table, a barrier - put 1.5-2 us of barriers and serial LDS
// round trips at the head of every GroupNorm'd launch: tools/tap_timeline.py "gn table".)
// s_mr / scratch: no longer used (kept in the signature for the callers' LDS maps).  Ends with a barrier.
template <int NTHR>
__device__ __forceinline__ void conv_gn_table(const ConvParams& p, int b0, int nb, float* s_gn, int shoff, float* s_mr,
                                              unsigned char* scratch) {
  static_assert(NTHR == 128 || NTHR == 256 || NTHR == 512 || NTHR == 1024, "conv_gn_table: 128 ... 1024 threads");
  (void)s_mr; (void)scratch;
  const int tid = threadIdx.x;
  const int G = p.gn_G, C = p.gn_C;
  constexpr int LOGN = NTHR == 128 ? 7 : NTHR == 256 ? 8 : NTHR == 512 ? 9 : 10;
  const int lsh = min(4, LOGN - 5 - (nb == 1 ? 0 : nb == 2 ? 1 : 2));        // log2 L: NTHR / (32 images) lanes (nb <= 4), at most a DPP row
  const int L = 1 << lsh;
  const int slot = tid >> lsh, j = tid & (L - 1);
  const int bl = slot >> 5, g = slot & 31;
  const bool act = bl < nb && g < G;
  const int cpg = C / G;
  const int nsp = p.gn_nsplit < 0 ? -p.gn_nsplit : p.gn_nsplit;
  // gamma / beta of this lane's first channel: requested now, beside the statistics rows
  const int ce0 = act ? g * cpg + min(j, cpg - 1) : 0;
  const float g0 = p.gn_gamma[ce0], be0 = p.gn_beta[ce0];
  double a0 = 0.0, a1 = 0.0, c0 = 0.0, c1 = 0.0;
  if (act && p.gn_unit) {
    // units of 4 channels, one buffer per segment: group g = units [g * upg, (g + 1) * upg).  The group's
    // upg x (rows of its segment) partial sums are dealt out item by item over the slot's lanes
    const int upg = cpg >> 2;
    auto item = [&](int t, double& s0, double& s1) __attribute__((always_inline)) {
      // t -> (unit u = g * upg + t % upg, row t / upg); rows beyond the unit's segment add nothing
      const int u = g * upg + t % upg, sp = t / upg;
      const int sg = (p.seg_gn[1] >= 0 && u >= p.gn_uoff[1]) ? ((p.seg_gn[2] >= 0 && u >= p.gn_uoff[2]) ? ((p.seg_gn[3] >= 0 && u >= p.gn_uoff[3]) ? 3 : 2) : 1) : 0;
      const double* base = sg == 0 ? p.gn_useg[0] : sg == 1 ? p.gn_useg[1] : sg == 2 ? p.gn_useg[2] : p.gn_useg[3];
      const int usp = sg == 0 ? p.gn_unsp[0] : sg == 1 ? p.gn_unsp[1] : sg == 2 ? p.gn_unsp[2] : p.gn_unsp[3];
      const int ucnt = sg == 0 ? p.gn_ucnt[0] : sg == 1 ? p.gn_ucnt[1] : sg == 2 ? p.gn_ucnt[2] : p.gn_ucnt[3];
      const int uoff = sg == 0 ? p.gn_uoff[0] : sg == 1 ? p.gn_uoff[1] : sg == 2 ? p.gn_uoff[2] : p.gn_uoff[3];
      const bool live = sp < usp;
      const double* q = base + ((((int64_t)(b0 + bl)) * usp + (live ? sp : 0)) * ucnt + (u - uoff)) * 2;
      const double v0 = q[0], v1 = q[1];
      s0 = live ? v0 : 0.0; s1 = live ? v1 : 0.0;
    };
    int maxsp = p.gn_unsp[0];
    if (p.seg_gn[1] >= 0) maxsp = max(maxsp, p.gn_unsp[1]);
    if (p.seg_gn[2] >= 0) maxsp = max(maxsp, p.gn_unsp[2]);
    if (p.seg_gn[3] >= 0) maxsp = max(maxsp, p.gn_unsp[3]);
    const int T = upg * maxsp;
    for (int t = j; t < T; t += 2 * L) {
      double x0, x1, y0 = 0.0, y1 = 0.0;
      item(t, x0, x1);
      if (t + L < T) item(t + L, y0, y1);
      a0 += x0; a1 += x1; c0 += y0; c1 += y1;
    }
  } else if (act) {
    if (p.gn_nsplit < 0) {
      const double* sd = reinterpret_cast<const double*>(p.gn_sums) + (((int64_t)(b0 + bl)) * nsp * G + g) * 2;
      int sp = j;
      for (; sp + L < nsp; sp += 2 * L) {
        const double* q = sd + (int64_t)sp * G * 2;
        const double* r = sd + (int64_t)(sp + L) * G * 2;
        a0 += q[0]; a1 += q[1]; c0 += r[0]; c1 += r[1];
      }
      if (sp < nsp) { const double* q = sd + (int64_t)sp * G * 2; a0 += q[0]; a1 += q[1]; }
    } else {
      const float* sf = p.gn_sums + (((int64_t)(b0 + bl)) * nsp * G + g) * 2;
      int sp = j;
      for (; sp + L < nsp; sp += 2 * L) {
        const float* q = sf + (int64_t)sp * G * 2;
        const float* r = sf + (int64_t)(sp + L) * G * 2;
        a0 += (double)q[0]; a1 += (double)q[1]; c0 += (double)r[0]; c1 += (double)r[1];
      }
      if (sp < nsp) { const float* q = sf + (int64_t)sp * G * 2; a0 += (double)q[0]; a1 += (double)q[1]; }
    }
  }
  // (every lane takes part in the exchange: a lane outside `act` contributes to no live slot)
  const double t0 = lanes_sum_d(a0 + c0, L), t1 = lanes_sum_d(a1 + c1, L);
  const double mean = t0 * p.gn_inv_count;
  double var = t1 * p.gn_inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  // 1 / sqrt in fp64 as v_rsq_f64 + one Newton step (error ~1e-15, far below the float it is rounded to)
  // instead of the sqrt + division sequences (~40 instructions on the launch's critical path)
  const double x = var + (double)p.gn_eps;
  double r = __builtin_amdgcn_rsq(x);
  r = r * (1.5 - 0.5 * x * r * r);
  const float mf = (float)mean, rf = (float)r;
  if (act) {
    for (int cc = j; cc < cpg; cc += L) {
      const int c = g * cpg + cc;
      const float ga = cc == j ? g0 : p.gn_gamma[c];
      const float be = cc == j ? be0 : p.gn_beta[c];
      const float sc = rf * ga;
      s_gn[bl * C + c] = sc;
      s_gn[shoff + bl * C + c] = be - mf * sc;
    }
  }
  __syncthreads();
}

// GroupNorm scale/shift (+ SiLU) of one 16-byte vector of activations (8 bf16 / 4 f32) whose first
// channel's table entries are tsc[0] / tsh[0].
// FAST (f32 tensors only): SiLU from the hardware exp2 / rcp (the 16-bit modes' form, ~3e-7 relative) instead of
// expf and a true division - the split-precision mode's choice (FDBM_SPLIT_SILU=precise keeps the other).
template <typename T, bool FAST = false>
__device__ __forceinline__ uint4 gn_transform16(uint4 v, const float* tsc, const float* tsh, bool silu) {
  if constexpr (sizeof(T) == 2) {
    typename V16<T>::x8 e = *reinterpret_cast<typename V16<T>::x8*>(&v);
    float y[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) y[q] = (float)e[q] * tsc[q] + tsh[q];
    if (silu) {
#pragma unroll
      for (int q = 0; q < 8; ++q) y[q] = silu_f(y[q]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) e[q] = (T)y[q];
    return *reinterpret_cast<uint4*>(&e);
  } else {
    f32x4 e = *reinterpret_cast<f32x4*>(&v);
    if (silu) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if constexpr (FAST) e[q] = silu_f(e[q] * tsc[q] + tsh[q]); else e[q] = silu_precise(e[q] * tsc[q] + tsh[q]);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) e[q] = e[q] * tsc[q] + tsh[q];
    }
    return *reinterpret_cast<uint4*>(&e);
  }
}

// p.seg[ks] with a run-time ks makes the compiler copy the whole kernel argument to scratch
// and index it there; chains of wave-uniform selects on constant indices stay in SGPRs.
#define SEG_FIELD(p, ks, f) ((ks) == 0 ? (p).seg[0].f : (ks) == 1 ? (p).seg[1].f : (ks) == 2 ? (p).seg[2].f : (p).seg[3].f)

