// tfgridnet.hip - the TF-GridNet backbone (fdbm/backbones/tfgridnet.py:83-510) as one C-ABI context: created from an
// architecture descriptor + ONE flat f32 weight blob (layout: fdbm_amd/tfgridnet.py:pack_state, documented in
// include/fdbm_hip.h), evaluated with fdbm_tfgridnet_forward on caller-owned buffers.  f32 throughout (the parity mode);
// first version: every kernel is written for correctness and a sane memory pattern, none is tuned yet - the two
// recurrences (260 dependent LSTM steps per sequence) bound the evaluation, not the GEMMs.
//
// Activations are [B][T][Q][C] (C = emb_dim contiguous; T frames, Q frequency bins) - the layout in which
//   * F.unfold(ks) + Linear along Q (or, transposed, along T) is a GEMM whose A rows are OVERLAPPING contiguous windows
//     of ks*C floats (row stride C): no unfolded tensor is ever materialised;
//   * ConvTranspose1d(2H -> C, ks) is the same on a zero-padded [L + 2(ks-1)][2H] sequence buffer;
//   * LayerNorm over C, the per-head E-normalisation and the 1x1 convolutions are row operations.
#include <stdint.h>
#include <stdlib.h>

#include "common.h"
#include "conv_common.h"      // Mfma<f16_t>, split_f16x4: the split-precision operand pairs

namespace {

constexpr int TFG_MAX_LAYERS = 8;
constexpr int TFG_SEQ_CHUNK = 1024;     // sequences per pass of the recurrent path (bounds the workspace)

// ---- small elementwise / layout kernels ---------------------------------------------------------------------------
// x, y complex64 [B][1][F][T] -> in [B][T][F][4] = (x.re, x.im, y.re, y.im)   (tfgridnet.py:199, 219)
__global__ void tfg_pack_in(float* __restrict__ out, const float2* __restrict__ x, const float2* __restrict__ y, int B, int F, int T) {
  const int64_t n = (int64_t)B * T * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    const int t = (int)((i / F) % T);
    const int b = (int)(i / ((int64_t)F * T));
    const float2 a = x[((int64_t)b * F + f) * T + t], c = y[((int64_t)b * F + f) * T + t];
    reinterpret_cast<float4*>(out)[i] = float4{a.x, a.y, c.x, c.y};
  }
}

// in [B][T][F][2] -> complex64 [B][1][F][T]   (tfgridnet.py:227-230)
__global__ void tfg_unpack_out(float2* __restrict__ out, const float* __restrict__ in, int B, int F, int T) {
  const int64_t n = (int64_t)B * T * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const int f = (int)((i / T) % F);
    const int b = (int)(i / ((int64_t)F * T));
    const float2 v = reinterpret_cast<const float2*>(in)[((int64_t)b * T + t) * F + f];
    out[i] = v;
  }
}

// 3x3 convolution, zero padding 1, few channels: in [B][T][F][Ci] -> out [B][T][F][Co], w [Co][3][3][Ci], one thread per
// (pixel, co).  (The stem 4 -> C and, with flipped / transposed weights, ConvTranspose2d C -> 2: tfgridnet.py:150-175.)
__global__ void tfg_conv3x3(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ w,
                            const float* __restrict__ bias, int B, int T, int F, int Ci, int Co) {
  const int64_t n = (int64_t)B * T * F * Co;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % Co);
    const int64_t px = i / Co;
    const int f = (int)(px % F);
    const int t = (int)((px / F) % T);
    const int b = (int)(px / ((int64_t)F * T));
    float acc = bias[co];
    for (int ky = 0; ky < 3; ++ky) {
      const int tt = t + ky - 1;
      if (tt < 0 || tt >= T) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ff = f + kx - 1;
        if (ff < 0 || ff >= F) continue;
        const float* ip = in + (((int64_t)b * T + tt) * F + ff) * Ci;
        const float* wp = w + ((co * 3 + ky) * 3 + kx) * Ci;
        for (int c = 0; c < Ci; ++c) acc = fmaf(ip[c], wp[c], acc);
      }
    }
    out[i] = acc;
  }
}

// per-sample (sum, sum of squares) in fp64: stats[b][2] += ...   (GroupNorm(1, C) of the stem)
__global__ void tfg_sum_stats(double* __restrict__ stats, const float* __restrict__ x, int64_t per_sample) {
  const int b = blockIdx.y;
  const float* p = x + (int64_t)b * per_sample;
  double s = 0.0, q = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = p[i];
    s += v; q += v * v;
  }
  __shared__ double sh[2][256];
  sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { atomicAdd(&stats[b * 2], sh[0][0]); atomicAdd(&stats[b * 2 + 1], sh[1][0]); }
}

// out [B][T+2p][Q+2p][C] = zero-pad( [GroupNorm(1,C)](x) + tb[b][c] )   (tfgridnet.py:222-223, 337-338)
__global__ void tfg_gn_temb_pad(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ tb,
                                const double* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                float eps, int B, int T, int Q, int C, int pad) {
  const int Tp = T + 2 * pad, Qp = Q + 2 * pad;
  const int64_t n = (int64_t)B * Tp * Qp * C;
  const double cnt = (double)T * Q * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int q = (int)((i / C) % Qp) - pad;
    const int t = (int)((i / ((int64_t)C * Qp)) % Tp) - pad;
    const int b = (int)(i / ((int64_t)C * Qp * Tp));
    float v = 0.f;
    if (t >= 0 && t < T && q >= 0 && q < Q) {
      v = x[(((int64_t)b * T + t) * Q + q) * C + c];
      if (stats) {
        const double mean = stats[b * 2] / cnt;
        const double var = stats[b * 2 + 1] / cnt - mean * mean;
        v = (float)((v - mean) / sqrt(var + (double)eps)) * gamma[c] + beta[c];
      }
      v += tb[b * C + c];
    }
    out[i] = v;
  }
}

// rows of C floats: out = LayerNorm_C([PReLU](in)) * gamma + beta [+ res]   (intra/inter_norm; attn_concat_proj 1-2 + residual)
__global__ void tfg_ln_rows(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ gamma,
                            const float* __restrict__ beta, const float* __restrict__ prelu, const float* __restrict__ res,
                            float eps, int64_t rows, int C) {
  // one row per thread, the row held in registers (C <= 64: 16-byte loads, one pass over memory)
  const int C4 = C >> 2;
  const float a = prelu ? prelu[0] : 1.f;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    const float4* p = reinterpret_cast<const float4*>(in + r * C);
    float4 v[16];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < C4) {
        float4 x = p[c];
        if (prelu) { x.x = x.x < 0.f ? a * x.x : x.x; x.y = x.y < 0.f ? a * x.y : x.y; x.z = x.z < 0.f ? a * x.z : x.z; x.w = x.w < 0.f ? a * x.w : x.w; }
        v[c] = x;
        s += (x.x + x.y) + (x.z + x.w);
      }
    const float mean = s / C;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < C4) {
        const float d0 = v[c].x - mean, d1 = v[c].y - mean, d2 = v[c].z - mean, d3 = v[c].w - mean;
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    const float rstd = 1.f / sqrtf(q / C + eps);
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < C4) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[c], bt = reinterpret_cast<const float4*>(beta)[c];
        float4 o = {(v[c].x - mean) * rstd * g.x + bt.x, (v[c].y - mean) * rstd * g.y + bt.y,
                    (v[c].z - mean) * rstd * g.z + bt.z, (v[c].w - mean) * rstd * g.w + bt.w};
        if (res) { const float4 rr = reinterpret_cast<const float4*>(res + r * C)[c]; o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w; }
        reinterpret_cast<float4*>(out + r * C)[c] = o;
      }
  }
}

// [B][A1][A2][C] -> [B][A2 - 2 c2][A1 - 2 c1][C] (swap the two middle axes, crop c1 / c2 entries at both ends of A1 / A2)
__global__ void tfg_transpose_crop(float* __restrict__ out, const float* __restrict__ in, int B, int A1, int A2, int C, int c1, int c2) {
  const int O1 = A2 - 2 * c2, O2 = A1 - 2 * c1;
  const int64_t n = (int64_t)B * O1 * O2 * (C / 4);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % (C / 4));
    const int o2 = (int)((i / (C / 4)) % O2);
    const int o1 = (int)((i / ((int64_t)(C / 4) * O2)) % O1);
    const int b = (int)(i / ((int64_t)(C / 4) * O2 * O1));
    reinterpret_cast<float4*>(out)[i] =
        reinterpret_cast<const float4*>(in)[(((int64_t)b * A1 + (o2 + c1)) * A2 + (o1 + c2)) * (C / 4) + c];
  }
}

// ---- GEMM: out[m][n] = sum_k A(m)[k] * W[n][k] + bias[n] (+ res[m][n]) ---------------------------------------------
// A(m) = A + (m / rps) * seq_stride + (m % rps) * lda : rows may be overlapping windows (unfold).  out likewise with
// (out_seq_stride, ldo).  grid.z = batch with (sA, sW, sO) strides.  64 x 64 tile, 4 waves x (2 x 2 MFMA tiles), K-step 16.
struct GemmArgs {
  float* out; const float* A; const float* W; const float* bias; const float* res;
  int64_t M; int N, K;
  int rps; int64_t seq_stride; int lda; int64_t out_seq_stride; int ldo;
  int64_t sA, sW, sO;
  float scale;
  int swap;          // 1: blockIdx.x walks the n-tiles, blockIdx.y the m-tiles
};

__global__ void __launch_bounds__(256) tfg_gemm(const GemmArgs g) {
  // v_mfma_f32_16x16x4_f32: the W tile is the A operand (lane: row n = lane % 16, k = lane / 16), the activation tile the
  // B operand (column m = lane % 16, k = lane / 16); a lane's 4 results are 4 consecutive n of one m.
  __shared__ float As[16][64 + 4], Ws[16][64 + 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;             // wave's 32 x 32 quadrant of the 64 (m) x 64 (n) tile
  // (n-tiles vary fastest: the workgroups in flight together write adjacent 256-byte pieces of the same output rows)
  const int64_t m0 = (int64_t)(g.swap ? blockIdx.y : blockIdx.x) * 64;
  const int n0 = (g.swap ? blockIdx.x : blockIdx.y) * 64;
  const float* A = g.A + (int64_t)blockIdx.z * g.sA;
  const float* W = g.W + (int64_t)blockIdx.z * g.sW;
  float* out = g.out + (int64_t)blockIdx.z * g.sO;
  f32x4 acc[2][2];                                      // [n-tile][m-tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // loader: thread loads 4 consecutive k of one row (row = tid / 4, k4 = (tid % 4) * 4) for A and for W
  const int lr = tid >> 2, lk = (tid & 3) * 4;
  const int64_t am = m0 + lr;
  const float* arow = nullptr;
  if (am < g.M) arow = A + (am / g.rps) * g.seq_stride + (am % g.rps) * (int64_t)g.lda;
  const int wr = n0 + lr;
  const float* wrow = wr < g.N ? W + (int64_t)wr * g.K : nullptr;
  const int fr = lane & 15, fk = lane >> 4;
  // software pipeline: the next k-step's operands are requested into registers before the current one is multiplied
  // (with the loads inside the step every one of the K / 16 steps waited a full memory latency); 16-byte loads where
  // the row starts and K allow it
  const bool vecA = ((g.lda | g.K) & 3) == 0 && (g.seq_stride & 3) == 0 && (g.sA & 3) == 0;
  const bool vecW = (g.K & 3) == 0 && (g.sW & 3) == 0;
  float ra[4], rw[4];
  auto fetch = [&](int k0) __attribute__((always_inline)) {
    const int k = k0 + lk;
    if (arow && vecA && k < g.K) {
      const float4 v = *reinterpret_cast<const float4*>(arow + k);
      ra[0] = v.x; ra[1] = v.y; ra[2] = v.z; ra[3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) ra[j] = (arow && k + j < g.K) ? arow[k + j] : 0.f;
    }
    if (wrow && vecW && k < g.K) {
      const float4 v = *reinterpret_cast<const float4*>(wrow + k);
      rw[0] = v.x; rw[1] = v.y; rw[2] = v.z; rw[3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) rw[j] = (wrow && k + j < g.K) ? wrow[k + j] : 0.f;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < g.K; k0 += 16) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { As[lk + j][lr] = ra[j]; Ws[lk + j][lr] = rw[j]; }
    __syncthreads();
    if (k0 + 16 < g.K) fetch(k0 + 16);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      float wv[2], av[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        wv[i] = Ws[kg * 4 + fk][wn * 32 + i * 16 + fr];
        av[i] = As[kg * 4 + fk][wm * 32 + i * 16 + fr];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i], av[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t m = m0 + wm * 32 + j * 16 + fr;
    if (m >= g.M) continue;
    const int64_t ooff = (m / g.rps) * g.out_seq_stride + (m % g.rps) * (int64_t)g.ldo;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 32 + i * 16 + fk * 4 + r;
        if (n >= g.N) continue;
        float v = acc[i][j][r] * g.scale + (g.bias ? g.bias[n] : 0.f);
        if (g.res) v += g.res[ooff + n];
        out[ooff + n] = v;
      }
  }
}

// The same GEMM with every product on the 16-bit matrix pipe (round 3): both operands are split into IEEE-half (hi, lo) pairs while
// they are staged - hi = half(16 x), lo = half(16 x - hi), 22 significant bits, conv_common.h - and a.w = a_hi.w_hi + a_hi.w_lo +
// a_lo.w_hi: three v_mfma_f32_16x16x32_f16 per 32 k (48 matrix-pipe cycles) where the f32 form spends eight v_mfma_f32_16x16x4_f32
// (256 cycles); the f32 sums are multiplied by 1 / 256.  Everything else - tiles, windows, epilogue - as above.  LDS rows: 64 rows x
// [32 halves hi | 32 halves lo] + 16 bytes (conflict-free b128 fragment reads).  |x| <= 4 094 (16 x clamps at the largest half):
// LayerNorm outputs, LSTM states, normalised Q / K / V, softmax rows and the residual stream are far inside.
__global__ void __launch_bounds__(256) tfg_gemm_split(const GemmArgs g) {
  constexpr int RS = 144;
  __shared__ __attribute__((aligned(16))) unsigned char As[64 * RS], Ws[64 * RS];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  // (n-tiles vary fastest: the workgroups in flight together write adjacent 256-byte pieces of the same output rows)
  const int64_t m0 = (int64_t)(g.swap ? blockIdx.y : blockIdx.x) * 64;
  const int n0 = (g.swap ? blockIdx.x : blockIdx.y) * 64;
  const float* A = g.A + (int64_t)blockIdx.z * g.sA;
  const float* W = g.W + (int64_t)blockIdx.z * g.sW;
  float* out = g.out + (int64_t)blockIdx.z * g.sO;
  f32x4 acc[2][2];                                      // [n-tile][m-tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // loader: thread loads 8 consecutive k of one row (row = tid / 4, k8 = (tid % 4) * 8) for A and for W
  const int lr = tid >> 2, lq = tid & 3;
  const int64_t am = m0 + lr;
  const float* arow = nullptr;
  if (am < g.M) arow = A + (am / g.rps) * g.seq_stride + (am % g.rps) * (int64_t)g.lda;
  const int wr = n0 + lr;
  const float* wrow = wr < g.N ? W + (int64_t)wr * g.K : nullptr;
  const int fr = lane & 15, fk = lane >> 4;
  const bool vecA = ((g.lda | g.K) & 3) == 0 && (g.seq_stride & 3) == 0 && (g.sA & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
  const bool vecW = (g.K & 3) == 0 && (g.sW & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
  f32x4 ra[2], rw[2];
  auto fetch_row = [&](const float* row, bool vec, int k, f32x4 (&r)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kk = k + 4 * h;
      if (row && vec && kk + 3 < g.K) {
        r[h] = *reinterpret_cast<const f32x4*>(row + kk);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[h][j] = (row && kk + j < g.K) ? row[kk + j] : 0.f;
      }
    }
  };
  auto stage = [&](unsigned char* dst, const f32x4 (&r)[2]) __attribute__((always_inline)) {
    uint2 h0, l0, h1, l1;
    split_f16x4(r[0], h0, l0);
    split_f16x4(r[1], h1, l1);
    *reinterpret_cast<uint4*>(dst + lr * RS + lq * 16) = uint4{h0.x, h0.y, h1.x, h1.y};
    *reinterpret_cast<uint4*>(dst + lr * RS + 64 + lq * 16) = uint4{l0.x, l0.y, l1.x, l1.y};
  };
  fetch_row(arow, vecA, lq * 8, ra);
  fetch_row(wrow, vecW, lq * 8, rw);
  for (int k0 = 0; k0 < g.K; k0 += 32) {
    stage(As, ra);
    stage(Ws, rw);
    __syncthreads();
    if (k0 + 32 < g.K) {
      fetch_row(arow, vecA, k0 + 32 + lq * 8, ra);
      fetch_row(wrow, vecW, k0 + 32 + lq * 8, rw);
    }
    uint4 wh[2], wl[2], ah[2], al[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned char* wp = Ws + (wn * 32 + i * 16 + fr) * RS + fk * 16;
      const unsigned char* ap = As + (wm * 32 + i * 16 + fr) * RS + fk * 16;
      wh[i] = *reinterpret_cast<const uint4*>(wp); wl[i] = *reinterpret_cast<const uint4*>(wp + 64);
      ah[i] = *reinterpret_cast<const uint4*>(ap); al[i] = *reinterpret_cast<const uint4*>(ap + 64);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        Mfma<f16_t>::run(wh[i], ah[j], acc[i][j]);
        Mfma<f16_t>::run(wh[i], al[j], acc[i][j]);
        Mfma<f16_t>::run(wl[i], ah[j], acc[i][j]);
      }
    __syncthreads();
  }
  const float sc = g.scale * (1.0f / (SPLIT_ACT_SCALE * SPLIT_ACT_SCALE));
  // a lane's 4 results are 4 consecutive n of one m: one 16-byte store (and bias / residual load) where the row layout allows
  const bool vecO = ((g.ldo | g.N) & 3) == 0 && (g.out_seq_stride & 3) == 0 && (g.sO & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                    (!g.res || (reinterpret_cast<uintptr_t>(g.res) & 15) == 0) && (!g.bias || (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t m = m0 + wm * 32 + j * 16 + fr;
    if (m >= g.M) continue;
    const int64_t ooff = (m / g.rps) * g.out_seq_stride + (m % g.rps) * (int64_t)g.ldo;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int nb = n0 + wn * 32 + i * 16 + fk * 4;
      if (vecO && nb + 3 < g.N) {
        f32x4 v = acc[i][j] * sc;
        if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + nb);
        if (g.res) v += *reinterpret_cast<const f32x4*>(g.res + ooff + nb);
        *reinterpret_cast<f32x4*>(out + ooff + nb) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = nb + r;
          if (n >= g.N) continue;
          float v = acc[i][j][r] * sc + (g.bias ? g.bias[n] : 0.f);
          if (g.res) v += g.res[ooff + n];
          out[ooff + n] = v;
        }
      }
    }
  }
}

// ---- LSTM recurrence (nn.LSTM, gate order i f g o, zero initial state; tfgridnet.py:258-269) ---------------------------
// One workgroup = one (sequence, direction); thread r < 4H owns gate row r of W_hh (H floats in registers).
// G [nseq][L][8H]: input pre-activations (fwd gates | bwd gates, biases included); hout [nseq][L + 2*hp][2H] (the
// zero border of hp entries is what ConvTranspose1d reads; written elsewhere, once).
template <int HMAX, int NS>
__global__ void __launch_bounds__(4 * HMAX) tfg_lstm(float* __restrict__ hout, const float* __restrict__ G,
                                                     const float* __restrict__ whh_f, const float* __restrict__ whh_b,
                                                     int nseq, int L, int H, int hp) {
  // NS sequences per workgroup share the register-resident W_hh rows (one 7-wave workgroup with ~136 VGPRs fits a CU, so
  // the launch runs in ceil(workgroups / 256) rounds): the host picks NS in 1..4 to minimise rounds x step time - at batch 1
  // (263 sequences x 2 directions) NS = 3 gives 176 workgroups = ONE round where NS = 2 gave 264 = two.
  __shared__ __attribute__((aligned(16))) float s_h[NS][HMAX];
  __shared__ float s_g[NS][4 * HMAX];
  const int seq0 = blockIdx.x * NS, dir = blockIdx.y;
  const int r = threadIdx.x;
  const float* whh = dir ? whh_b : whh_f;
  float w[HMAX];
#pragma unroll
  for (int k = 0; k < HMAX; ++k) w[k] = (r < 4 * H && k < H) ? whh[(int64_t)r * H + k] : 0.f;
  for (int i = r; i < NS * HMAX; i += blockDim.x) (&s_h[0][0])[i] = 0.f;
  float c = 0.f;                                   // cell state of (sequence r / H, unit r % H) for r < NS * H
  const float* gseq[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) gseq[q] = G + (int64_t)min(seq0 + q, nseq - 1) * L * 8 * H + dir * 4 * H;
  const int uq = r / H, uu = r - uq * H;            // this thread's (sequence, unit) in the gate phase
  float* hseq = hout + ((int64_t)min(seq0 + uq, nseq - 1) * (L + 2 * hp) + hp) * 2 * H + dir * H + uu;
  const bool ulive = r < NS * H && seq0 + uq < nseq;
  __syncthreads();
  // (the next step's input pre-activation is requested a step ahead: its load latency would otherwise sit in every one
  // of the L dependent steps)
  float gnext[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) gnext[q] = (r < 4 * H) ? gseq[q][(int64_t)(dir ? L - 1 : 0) * 8 * H + r] : 0.f;
  for (int s = 0; s < L; ++s) {
    const int t = dir ? L - 1 - s : s;
    if (r < 4 * H) {
      float acc[NS];
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        acc[q] = gnext[q];
        if (s + 1 < L) gnext[q] = gseq[q][(int64_t)(dir ? t - 1 : t + 1) * 8 * H + r];
      }
#pragma unroll
      for (int k = 0; k < HMAX; k += 4) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          const float4 hv = *reinterpret_cast<const float4*>(&s_h[q][k]);
          acc[q] = fmaf(w[k], hv.x, acc[q]); acc[q] = fmaf(w[k + 1], hv.y, acc[q]);
          acc[q] = fmaf(w[k + 2], hv.z, acc[q]); acc[q] = fmaf(w[k + 3], hv.w, acc[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < NS; ++q) s_g[q][r] = acc[q];
    }
    __syncthreads();
    if (r < NS * H) {
      const float* gq = s_g[uq];
      const float ig = 1.f / (1.f + expf(-gq[uu])), fg = 1.f / (1.f + expf(-gq[H + uu]));
      const float gg = tanhf(gq[2 * H + uu]), og = 1.f / (1.f + expf(-gq[3 * H + uu]));
      c = fg * c + ig * gg;
      const float h = og * tanhf(c);
      s_h[uq][uu] = h;
      if (ulive) hseq[(int64_t)t * 2 * H] = h;
    }
    __syncthreads();
  }
}

// ---- attention pieces (tfgridnet.py:376-428) ---------------------------------------------------------------------------
// qkv [B][T][Q][2 nh E + C] -> AllHeadPReLULayerNormalization4DC per (b, t, q, head) over the head's channels, written as
//   Qn, Kn [B][nh][T][E * Q]   (feature = e * Q + q, as the reference flattens [E][Q])
//   VnT    [B][nh][Dv * Q][T]  (Dv = C / nh; transposed so that P @ V is again an A @ W^T product)
__global__ void tfg_head_norm(float* __restrict__ Qn, float* __restrict__ Kn, float* __restrict__ VnT, const float* __restrict__ qkv,
                              const float* __restrict__ slope, const float* __restrict__ gamma, const float* __restrict__ beta,
                              float eps, int B, int T, int Q, int nh, int E, int Dv) {
  const int NC = 2 * nh * E + nh * Dv;
  const int64_t n = (int64_t)B * T * Q * nh * 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int which = (int)(i % 3);                      // 0 Q, 1 K, 2 V
    const int h = (int)((i / 3) % nh);
    const int64_t px = i / (3 * nh);
    const int q = (int)(px % Q);
    const int t = (int)((px / Q) % T);
    const int b = (int)(px / ((int64_t)Q * T));
    const int D = which == 2 ? Dv : E;
    const int cbase = which == 0 ? h * E : which == 1 ? nh * E + h * E : 2 * nh * E + h * Dv;   // channel in qkv = parameter index
    const float* p = qkv + px * NC + cbase;
    const float a = slope[which * nh + h];
    float v[16];
    float s = 0.f;
    for (int e = 0; e < D; ++e) { float x = p[e]; x = x < 0.f ? a * x : x; v[e] = x; s += x; }
    const float mean = s / D;
    float var = 0.f;
    for (int e = 0; e < D; ++e) { const float d = v[e] - mean; var += d * d; }
    const float rstd = 1.f / sqrtf(var / D + eps);
    for (int e = 0; e < D; ++e) {
      const float o = (v[e] - mean) * rstd * gamma[cbase + e] + beta[cbase + e];
      if (which == 0) Qn[(((int64_t)b * nh + h) * T + t) * ((int64_t)E * Q) + (int64_t)e * Q + q] = o;
      else if (which == 1) Kn[(((int64_t)b * nh + h) * T + t) * ((int64_t)E * Q) + (int64_t)e * Q + q] = o;
      else VnT[(((int64_t)b * nh + h) * ((int64_t)Dv * Q) + (int64_t)e * Q + q) * T + t] = o;
    }
  }
}

// rows of n floats: softmax in place (one workgroup per row)
__global__ void __launch_bounds__(256) tfg_softmax_rows(float* __restrict__ x, int n) {
  float* p = x + (int64_t)blockIdx.x * n;
  __shared__ float sh[256];
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, p[i]);
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + o]); __syncthreads(); }
  m = sh[0];
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { const float e = expf(p[i] - m); p[i] = e; s += e; }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  const float inv = 1.f / sh[0];
  for (int i = threadIdx.x; i < n; i += 256) p[i] *= inv;
}

// O [B][nh][T][Dv * Q] (feature = e * Q + q) -> [B][T][Q][C], channel = h * Dv + e
__global__ void tfg_attn_reorder(float* __restrict__ out, const float* __restrict__ O, int B, int T, int Q, int nh, int Dv) {
  const int C = nh * Dv;
  const int64_t n = (int64_t)B * T * Q * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int q = (int)((i / C) % Q);
    const int t = (int)((i / ((int64_t)C * Q)) % T);
    const int b = (int)(i / ((int64_t)C * Q * T));
    const int h = c / Dv, e = c % Dv;
    out[i] = O[(((int64_t)b * nh + h) * T + t) * ((int64_t)Dv * Q) + (int64_t)e * Q + q];
  }
}

// time embedding: tb [n_layers][B][C] = Linear_l(SiLU(Linear2(SiLU(Linear1([sin, cos](2 pi W log t))))))   (tfgridnet.py:201-217, 222)
// one workgroup per sample (tiny); log t evaluated on the host (a 1-ulp libm difference moves sin(2 pi W log t) visibly)
__global__ void __launch_bounds__(256) tfg_temb(float* __restrict__ tb, const float* __restrict__ log_t, const float* __restrict__ Wf,
                                                const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                const float* __restrict__ b2, const float* __restrict__ wl, const float* __restrict__ bl,
                                                int B, int C, int n_layers) {
  extern __shared__ float sm[];
  float* f = sm;               // [2C]
  float* h1 = sm + 2 * C;      // [4C]
  float* h2 = h1 + 4 * C;      // [4C]
  const int b = blockIdx.x;
  const float lt = log_t[b];
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    const float p = lt * Wf[i] * 2.f * 3.14159265358979323846f;
    f[i] = sinf(p); f[C + i] = cosf(p);
  }
  __syncthreads();
  for (int n = threadIdx.x; n < 4 * C; n += blockDim.x) {
    float a = b1[n];
    for (int k = 0; k < 2 * C; ++k) a = fmaf(w1[n * 2 * C + k], f[k], a);
    h1[n] = a / (1.f + expf(-a));
  }
  __syncthreads();
  for (int n = threadIdx.x; n < 4 * C; n += blockDim.x) {
    float a = b2[n];
    for (int k = 0; k < 4 * C; ++k) a = fmaf(w2[n * 4 * C + k], h1[k], a);
    h2[n] = a / (1.f + expf(-a));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_layers * C; i += blockDim.x) {
    const int l = i / C, c = i % C;
    float a = bl[l * C + c];
    for (int k = 0; k < 4 * C; ++k) a = fmaf(wl[((int64_t)l * C + c) * 4 * C + k], h2[k], a);
    tb[((int64_t)l * B + b) * C + c] = a;
  }
}

inline int grid_for(int64_t n, int block = 256, int cap = 8192) {
  int64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : g > cap ? cap : g);
}

}  // namespace

// ---- context --------------------------------------------------------------------------------------------------------------
struct fdbm_tfgridnet_ctx {
  fdbm_tfgridnet_desc d;
  const float* w;                       // the packed blob (device)
  // offsets (floats) into the blob, in pack_state order
  int64_t conv_w, conv_b, gn_g, gn_b, dec_w, dec_b, fourier, t1w, t1b, t2w, t2b, tlw, tlb;
  struct Blk {
    int64_t ln_g[2], ln_b[2], win[2], bin[2], whh_f[2], whh_b[2], wdec[2], bdec[2];
    int64_t wqkv, bqkv, hn_slope, hn_gamma, hn_beta, wproj, bproj, prelu, pln_g, pln_b;
  } blk[TFG_MAX_LAYERS];
  int64_t total;
};

static void tfg_layout(fdbm_tfgridnet_ctx* c) {
  const fdbm_tfgridnet_desc& d = c->d;
  const int C = d.emb_dim, H = d.hidden, ks = d.emb_ks, nh = d.n_head, E = d.qk_channels, Dv = C / nh;
  const int NC = 2 * nh * E + C;
  int64_t o = 0;
  auto take = [&](int64_t n) { const int64_t r = o; o += n; return r; };
  c->conv_w = take((int64_t)C * 9 * d.in_ch); c->conv_b = take(C); c->gn_g = take(C); c->gn_b = take(C);
  for (int l = 0; l < d.n_layers; ++l) {
    auto& b = c->blk[l];
    for (int r = 0; r < 2; ++r) {
      b.ln_g[r] = take(C); b.ln_b[r] = take(C);
      b.win[r] = take((int64_t)8 * H * ks * C); b.bin[r] = take(8 * H);
      b.whh_f[r] = take((int64_t)4 * H * H); b.whh_b[r] = take((int64_t)4 * H * H);
      b.wdec[r] = take((int64_t)C * ks * 2 * H); b.bdec[r] = take(C);
    }
    b.wqkv = take((int64_t)NC * C); b.bqkv = take(NC);
    b.hn_slope = take(3 * nh); b.hn_gamma = take(NC); b.hn_beta = take(NC);
    b.wproj = take((int64_t)C * C); b.bproj = take(C); b.prelu = take(1); b.pln_g = take(C); b.pln_b = take(C);
  }
  c->dec_w = take((int64_t)d.out_ch * 9 * C); c->dec_b = take(d.out_ch);
  c->fourier = take(C);
  c->t1w = take((int64_t)4 * C * 2 * C); c->t1b = take(4 * C); c->t2w = take((int64_t)4 * C * 4 * C); c->t2b = take(4 * C);
  c->tlw = take((int64_t)d.n_layers * C * 4 * C); c->tlb = take((int64_t)d.n_layers * C);
  c->total = o;
}

static bool tfg_desc_ok(const fdbm_tfgridnet_desc* d) {
  return d && d->n_layers >= 1 && d->n_layers <= TFG_MAX_LAYERS && d->emb_dim >= 4 && d->emb_dim % 4 == 0 && d->emb_dim <= 64 &&
         d->hidden >= 4 && d->hidden % 4 == 0 && d->hidden <= 128 && d->emb_ks >= 1 && d->emb_ks <= 8 && d->n_head >= 1 &&
         d->emb_dim % d->n_head == 0 && d->emb_dim / d->n_head <= 16 && d->qk_channels >= 1 && d->qk_channels <= 16 &&
         d->in_ch == 4 && d->out_ch == 2;
}

extern "C" int64_t fdbm_tfgridnet_weights_count(const fdbm_tfgridnet_desc* d) {
  if (!tfg_desc_ok(d)) { fdbm_set_error("fdbm_tfgridnet: unsupported architecture descriptor"); return -1; }
  fdbm_tfgridnet_ctx c;
  c.d = *d;
  tfg_layout(&c);
  return c.total;
}

// workspace regions (floats), for (B, F, T)
struct TfgWs {
  int64_t xin, a0, a1, xp, n1, G, hbuf, x2, xt, x3, qkv, Qn, Kn, VnT, S, O, o2, tb, stats, total;
};
static TfgWs tfg_ws(const fdbm_tfgridnet_desc& d, int B, int F, int T) {
  const int C = d.emb_dim, H = d.hidden, ks = d.emb_ks, nh = d.n_head, E = d.qk_channels, Dv = C / nh, olp = ks - 1;
  const int64_t Tp = T + 2 * olp, Qp = F + 2 * olp;
  const int64_t px = (int64_t)B * T * F, pxp = (int64_t)B * Tp * Qp;
  // the recurrent path runs over chunks of at most TFG_SEQ_CHUNK sequences: the gate pre-activations of ALL sequences of a
  // batch-64 evaluation would be 14 GB
  int64_t nseq = B * (Tp > Qp ? Tp : Qp);
  if (nseq > TFG_SEQ_CHUNK) nseq = TFG_SEQ_CHUNK;
  const int64_t Lmax = (Tp > Qp ? Tp : Qp) - olp;
  TfgWs w;
  int64_t o = 0;
  auto take = [&](int64_t n) { const int64_t r = o; o += (n + 63) & ~(int64_t)63; return r; };
  w.xin = take(px * 4); w.a0 = take(px * C); w.a1 = take(px * C);
  w.xp = take(pxp * C); w.n1 = take(pxp * C + ks * C); w.x2 = take(pxp * C); w.xt = take(pxp * C); w.x3 = take(pxp * C);
  w.G = take(nseq * Lmax * 8 * H); w.hbuf = take(nseq * (Lmax + 2 * olp) * 2 * H + (int64_t)ks * 2 * H);
  w.qkv = take(px * (2 * nh * E + C)); w.Qn = take(px * nh * E); w.Kn = take(px * nh * E); w.VnT = take(px * C);
  w.S = take((int64_t)B * nh * T * T); w.O = take(px * C); w.o2 = take(px * C);
  w.tb = take((int64_t)d.n_layers * B * C); w.stats = take(4 * B);
  (void)Dv;
  w.total = o;
  return w;
}

extern "C" int64_t fdbm_tfgridnet_workspace_bytes(const fdbm_tfgridnet_desc* d, int B, int F, int T) {
  if (!tfg_desc_ok(d) || B < 1 || F < 1 || T < 1) { fdbm_set_error("fdbm_tfgridnet_workspace_bytes: bad arguments"); return -1; }
  return tfg_ws(*d, B, F, T).total * 4;
}

extern "C" fdbm_tfgridnet_ctx* fdbm_tfgridnet_create(const fdbm_tfgridnet_desc* d, const float* weights_dev, int64_t n_weights) {
  if (!tfg_desc_ok(d)) { fdbm_set_error("fdbm_tfgridnet_create: unsupported architecture descriptor"); return nullptr; }
  fdbm_tfgridnet_ctx* c = new fdbm_tfgridnet_ctx();
  c->d = *d;
  c->w = weights_dev;
  tfg_layout(c);
  if (!weights_dev || n_weights != c->total) {
    fdbm_set_error("fdbm_tfgridnet_create: weight blob has %lld floats, this architecture needs %lld", (long long)n_weights, (long long)c->total);
    delete c;
    return nullptr;
  }
  return c;
}

extern "C" void fdbm_tfgridnet_destroy(fdbm_tfgridnet_ctx* c) { delete c; }

static int tfg_gemm_launch(hipStream_t st, float* out, const float* A, const float* W, const float* bias, const float* res,
                           int64_t M, int N, int K, int rps, int64_t seq_stride, int lda, int64_t out_seq_stride, int ldo,
                           int batch = 1, int64_t sA = 0, int64_t sW = 0, int64_t sO = 0, float scale = 1.f) {
  GemmArgs g{out, A, W, bias, res, M, N, K, rps, seq_stride, lda, out_seq_stride, ldo, sA, sW, sO, scale, 0};
  const int64_t mt = (M + 63) / 64, nt = (N + 63) / 64;
  g.swap = mt <= 65535;
  dim3 grid((unsigned)(g.swap ? nt : mt), (unsigned)(g.swap ? mt : nt), (unsigned)batch);
  static const char* gm = getenv("FDBM_TFG_GEMM");              // experiments: "f32" = the v_mfma_f32_16x16x4_f32 form
  if (gm && gm[0] == 'f') tfg_gemm<<<grid, 256, 0, st>>>(g);
  else tfg_gemm_split<<<grid, 256, 0, st>>>(g);
  FDBM_LAUNCH_CHECK("fdbm_tfgridnet_forward(gemm)");
  return 0;
}

// x, y complex64 [B][1][F][T], log_t f32 [B] (host-evaluated logarithm of the model time), out complex64 [B][1][F][T];
// workspace: fdbm_tfgridnet_workspace_bytes(d, B, F, T) bytes, contents irrelevant on entry.  block_out (optional, may be
// NULL): f32 [n_layers][B][T][F][C], every block's output (tests).
// block_in / first_block (fdbm_tfgridnet_forward_from): the evaluation starts at block `first_block` with block_in
// [B][T][F][C] as that block's input (= the previous block's output); the stem is skipped.
static int tfg_forward_impl(fdbm_tfgridnet_ctx* c, const void* x, const void* y, const float* log_t, void* out,
                            int B, int F, int T, void* workspace, int64_t workspace_bytes, float* block_out,
                            const float* block_in, int first_block, void* stream) {
  FDBM_CHECK(c && log_t && out && workspace && (block_in || (x && y)), "fdbm_tfgridnet_forward: null argument");
  FDBM_CHECK(B >= 1 && F >= 1 && T >= 1 && T <= 4096, "fdbm_tfgridnet_forward: bad shape");
  FDBM_CHECK((block_in == nullptr) == (first_block == 0) && first_block >= 0 && first_block < c->d.n_layers,
             "fdbm_tfgridnet_forward_from: first_block %d needs its input (blocks 1 .. n_layers - 1), block 0 starts from x, y", first_block);
  const fdbm_tfgridnet_desc& d = c->d;
  const TfgWs ws = tfg_ws(d, B, F, T);
  FDBM_CHECK(workspace_bytes >= ws.total * 4, "fdbm_tfgridnet_forward: workspace too small (%lld < %lld bytes)",
             (long long)workspace_bytes, (long long)(ws.total * 4));
  hipStream_t st = (hipStream_t)stream;
  float* W0 = reinterpret_cast<float*>(workspace);
  const float* wt = c->w;
  const int C = d.emb_dim, H = d.hidden, ks = d.emb_ks, nh = d.n_head, E = d.qk_channels, Dv = C / nh, olp = ks - 1;
  const int Q = F, Tp = T + 2 * olp, Qp = Q + 2 * olp;
  const int NC = 2 * nh * E + C;
  const int64_t px = (int64_t)B * T * Q, pxp = (int64_t)B * Tp * Qp;
  float *xin = W0 + ws.xin, *cur = W0 + ws.a0, *nxt = W0 + ws.a1, *xp = W0 + ws.xp, *n1 = W0 + ws.n1, *G = W0 + ws.G,
        *hbuf = W0 + ws.hbuf, *x2 = W0 + ws.x2, *xt = W0 + ws.xt, *x3 = W0 + ws.x3, *qkv = W0 + ws.qkv, *Qn = W0 + ws.Qn,
        *Kn = W0 + ws.Kn, *VnT = W0 + ws.VnT, *S = W0 + ws.S, *O = W0 + ws.O, *o2 = W0 + ws.o2, *tb = W0 + ws.tb;
  double* stats = reinterpret_cast<double*>(W0 + ws.stats);

  FDBM_CHECK(fdbm_memset_zero(stats, (int64_t)16 * B, st) == 0, "fdbm_tfgridnet_forward: memset failed");
  tfg_temb<<<B, 256, (size_t)10 * C * sizeof(float), st>>>(tb, log_t, wt + c->fourier, wt + c->t1w, wt + c->t1b, wt + c->t2w, wt + c->t2b,
                                                           wt + c->tlw, wt + c->tlb, B, C, d.n_layers);
  if (block_in) {
    FDBM_CHECK(fdbm_copy_f32(cur, block_in, px * C, st) == 0, "fdbm_tfgridnet_forward_from: copy failed");
  } else {
    tfg_pack_in<<<grid_for(px), 256, 0, st>>>(xin, (const float2*)x, (const float2*)y, B, F, T);
    tfg_conv3x3<<<grid_for(px * C), 256, 0, st>>>(cur, xin, wt + c->conv_w, wt + c->conv_b, B, T, Q, d.in_ch, C);
    tfg_sum_stats<<<dim3(64, B), 256, 0, st>>>(stats, cur, (int64_t)T * Q * C);
  }
  FDBM_LAUNCH_CHECK("fdbm_tfgridnet_forward(stem)");

  for (int l = first_block; l < d.n_layers; ++l) {
    const auto& b = c->blk[l];
    // x + time embedding (the stem's GroupNorm applied on the way in the first block), zero-padded by ks - 1
    tfg_gn_temb_pad<<<grid_for(pxp * C), 256, 0, st>>>(xp, cur, tb + (int64_t)l * B * C, l == 0 ? stats : nullptr, wt + c->gn_g,
                                                       wt + c->gn_b, d.eps, B, T, Q, C, olp);
    // the two recurrent paths: r = 0 intra (sequences along Q for every (b, t)), r = 1 inter (along T for every (b, q))
    const float* src = xp;
    for (int r = 0; r < 2; ++r) {
      const int A = r == 0 ? Tp : Qp, Sl = r == 0 ? Qp : Tp;       // [B][A][Sl][C]: A sequences per sample, Sl entries each
      const int L = Sl - olp;
      const int64_t nseq = (int64_t)B * A;
      tfg_ln_rows<<<grid_for(nseq * Sl), 256, 0, st>>>(n1, src, wt + b.ln_g[r], wt + b.ln_b[r], nullptr, nullptr, d.eps, nseq * Sl, C);
      float* dst = r == 0 ? x2 : x3;
      for (int64_t s0 = 0; s0 < nseq; s0 += TFG_SEQ_CHUNK) {
        const int64_t ns = nseq - s0 < TFG_SEQ_CHUNK ? nseq - s0 : TFG_SEQ_CHUNK;
        // unfold + W_ih for both directions: rows = windows of ks*C floats at stride C
        if (tfg_gemm_launch(st, G, n1 + s0 * Sl * C, wt + b.win[r], wt + b.bin[r], nullptr, ns * L, 8 * H, ks * C, L, (int64_t)Sl * C, C,
                            (int64_t)L * 8 * H, 8 * H)) return 1;
        FDBM_CHECK(fdbm_memset_zero(hbuf, (((ns * (L + 2 * olp) * 2 * H + (int64_t)ks * 2 * H) * 4 + 15) / 16) * 16, st) == 0,
                   "fdbm_tfgridnet_forward: memset failed");
        // sequences per workgroup: rounds over the 256 CUs x measured step time (~0.70 + 0.37 NS us)
        int NS = 1;
        double best = 1e30;
        for (int c2 = 1; c2 <= 4; ++c2) {
          const int64_t wgs = 2 * ((ns + c2 - 1) / c2);
          const int64_t slots = H <= 80 ? 512 : 256;          // (H = 80: 120 VGPRs, two 5-wave workgroups share a CU)
          const double cost = (double)((wgs + slots - 1) / slots) * (0.70 + 0.37 * c2);
          if (cost < best - 1e-9) { best = cost; NS = c2; }
        }
        const dim3 lgrid((unsigned)((ns + NS - 1) / NS), 2);
#define TFG_LSTM_LAUNCH(HM, N_)                                                                                      \
  tfg_lstm<HM, N_><<<lgrid, 4 * HM, 0, st>>>(hbuf, G, wt + b.whh_f[r], wt + b.whh_b[r], (int)ns, L, H, olp)
#define TFG_LSTM_NS(HM)                                                                                              \
  do {                                                                                                               \
    if (NS == 1) TFG_LSTM_LAUNCH(HM, 1); else if (NS == 2) TFG_LSTM_LAUNCH(HM, 2);                                   \
    else if (NS == 3) TFG_LSTM_LAUNCH(HM, 3); else TFG_LSTM_LAUNCH(HM, 4);                                           \
  } while (0)
        if (H <= 80) TFG_LSTM_NS(80); else if (H <= 100) TFG_LSTM_NS(100); else TFG_LSTM_NS(128);
#undef TFG_LSTM_NS
#undef TFG_LSTM_LAUNCH
        FDBM_LAUNCH_CHECK("fdbm_tfgridnet_forward(lstm)");
        // ConvTranspose1d(2H -> C, ks) + bias + residual: rows = windows of ks*2H floats of the zero-bordered sequence buffer
        if (tfg_gemm_launch(st, dst + s0 * Sl * C, hbuf, wt + b.wdec[r], wt + b.bdec[r], src + s0 * Sl * C, ns * Sl, C, ks * 2 * H, Sl,
                            (int64_t)(L + 2 * olp) * 2 * H, 2 * H, (int64_t)Sl * C, C)) return 1;
      }
      if (r == 0) {
        tfg_transpose_crop<<<grid_for(pxp * C / 4), 256, 0, st>>>(xt, x2, B, Tp, Qp, C, 0, 0);       // [B][Qp][Tp][C]
        src = xt;
      }
    }
    // back to [B][T][Q][C], cropped
    tfg_transpose_crop<<<grid_for(px * C / 4), 256, 0, st>>>(nxt, x3, B, Qp, Tp, C, olp, olp);
    float* inter = nxt;
    // full-band self-attention over frames
    if (tfg_gemm_launch(st, qkv, inter, wt + b.wqkv, wt + b.bqkv, nullptr, px, NC, C, 1, C, C, NC, NC)) return 1;
    tfg_head_norm<<<grid_for(px * nh * 3), 256, 0, st>>>(Qn, Kn, VnT, qkv, wt + b.hn_slope, wt + b.hn_gamma, wt + b.hn_beta, d.eps,
                                                         B, T, Q, nh, E, Dv);
    const int DQ = E * Q, DV = Dv * Q;
    if (tfg_gemm_launch(st, S, Qn, Kn, nullptr, nullptr, T, T, DQ, 1, DQ, DQ, T, T, B * nh, (int64_t)T * DQ, (int64_t)T * DQ,
                        (int64_t)T * T, 1.f / sqrtf((float)DQ))) return 1;
    tfg_softmax_rows<<<(unsigned)((int64_t)B * nh * T), 256, 0, st>>>(S, T);
    if (tfg_gemm_launch(st, O, S, VnT, nullptr, nullptr, T, DV, T, 1, T, T, DV, DV, B * nh, (int64_t)T * T, (int64_t)DV * T,
                        (int64_t)T * DV)) return 1;
    tfg_attn_reorder<<<grid_for(px * C), 256, 0, st>>>(o2, O, B, T, Q, nh, Dv);
    if (tfg_gemm_launch(st, qkv, o2, wt + b.wproj, wt + b.bproj, nullptr, px, C, C, 1, C, C, C, C)) return 1;   // (qkv reused as [px][C])
    tfg_ln_rows<<<grid_for(px), 256, 0, st>>>(cur, qkv, wt + b.pln_g, wt + b.pln_b, wt + b.prelu, inter, d.eps, px, C);
    FDBM_LAUNCH_CHECK("fdbm_tfgridnet_forward(attention)");
    if (block_out)
      FDBM_CHECK(fdbm_copy_f32(block_out + (int64_t)l * px * C, cur, px * C, st) == 0, "fdbm_tfgridnet_forward: copy failed");
  }
  tfg_conv3x3<<<grid_for(px * d.out_ch), 256, 0, st>>>(xin, cur, wt + c->dec_w, wt + c->dec_b, B, T, Q, C, d.out_ch);
  tfg_unpack_out<<<grid_for(px), 256, 0, st>>>((float2*)out, xin, B, F, T);
  FDBM_LAUNCH_CHECK("fdbm_tfgridnet_forward(head)");
  return 0;
}

extern "C" int fdbm_tfgridnet_forward(fdbm_tfgridnet_ctx* c, const void* x, const void* y, const float* log_t, void* out,
                                      int B, int F, int T, void* workspace, int64_t workspace_bytes, float* block_out, void* stream) {
  FDBM_CHECK(x && y, "fdbm_tfgridnet_forward: null argument");
  return tfg_forward_impl(c, x, y, log_t, out, B, F, T, workspace, workspace_bytes, block_out, nullptr, 0, stream);
}

extern "C" int fdbm_tfgridnet_forward_from(fdbm_tfgridnet_ctx* c, const float* block_in, int first_block, const float* log_t, void* out,
                                           int B, int F, int T, void* workspace, int64_t workspace_bytes, float* block_out, void* stream) {
  FDBM_CHECK(block_in, "fdbm_tfgridnet_forward_from: null block input");
  return tfg_forward_impl(c, nullptr, nullptr, log_t, out, B, F, T, workspace, workspace_bytes, block_out, block_in, first_block, stream);
}
