// resample.hip - FIR resampling.
//
//  * fdbm_upfirdn2d : the reference's native-op boundary, any kernel / up / down / pad,
//    layout [major][h][w][minor] (minor contiguous), fp32.  One thread per output element,
//    gathers only the taps that hit a non-inserted sample (polyphase), consecutive threads
//    along `minor` then `w` so loads and stores coalesce.
//  * fdbm_resample2x: the only configuration the network uses ([1,3,3,1], factor 2), NHWC,
//    16-byte channel vectors, optional fused GroupNorm+SiLU on the input, one read of x
//    feeding both outputs of a res-block (resample(x) and resample(act(gn(x)))).
//    HBM-bound: down reads 1x writes 1/4x; up reads 1x writes 4x.
#include "common.h"
#include <cstdlib>

__global__ void __launch_bounds__(256) upfirdn2d_kernel(
    float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ kern, int major,
    int in_h, int in_w, int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
    int pad_x0, int pad_y0, int out_h, int out_w, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int mi = (int)(i % minor);
    int64_t r = i / minor;
    const int ox = (int)(r % out_w);
    r /= out_w;
    const int oy = (int)(r % out_h);
    const int64_t ma = r / out_h;
    // position in the padded, zero-inserted grid of the window's top-left corner
    const int py = oy * down_y - pad_y0;
    const int px = ox * down_x - pad_x0;
    float acc = 0.f;
    for (int ky = 0; ky < kh; ++ky) {
      const int uy = py + ky;                 // coordinate in the zero-inserted image
      if (uy < 0 || uy % up_y != 0) continue;
      const int iy = uy / up_y;
      if (iy >= in_h) continue;
      for (int kx = 0; kx < kw; ++kx) {
        const int ux = px + kx;
        if (ux < 0 || ux % up_x != 0) continue;
        const int ix = ux / up_x;
        if (ix >= in_w) continue;
        // true convolution: the kernel is flipped
        acc += kern[(kh - 1 - ky) * kw + (kw - 1 - kx)] * in[((ma * in_h + iy) * in_w + ix) * (int64_t)minor + mi];
      }
    }
    out[i] = acc;
  }
}

extern "C" int fdbm_upfirdn2d(float* out, const float* in, const float* kernel, int major, int in_h,
                              int in_w, int minor, int kh, int kw, int up_x, int up_y, int down_x,
                              int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                              void* stream) {
  FDBM_CHECK(out && in && kernel, "fdbm_upfirdn2d: null pointer");
  FDBM_CHECK(up_x >= 1 && up_y >= 1 && down_x >= 1 && down_y >= 1 && kh >= 1 && kw >= 1,
             "fdbm_upfirdn2d: up/down/kernel sizes must be >= 1");
  const int out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) / down_y + 1;
  const int out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) / down_x + 1;
  FDBM_CHECK(out_h > 0 && out_w > 0, "fdbm_upfirdn2d: empty output (%d x %d)", out_h, out_w);
  const int64_t total = (int64_t)major * out_h * out_w * minor;
  int g = (int)((total + 255) / 256);
  if (g > 8192) g = 8192;
  upfirdn2d_kernel<<<g, 256, 0, (hipStream_t)stream>>>(out, in, kernel, major, in_h, in_w, minor, kh,
                                                       kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0,
                                                       out_h, out_w, total);
  FDBM_LAUNCH_CHECK("fdbm_upfirdn2d");
  return 0;
}

// ---------------------------------------------------------------------------------
// [1,3,3,1] factor-2 fast path
//   down: out[i] = (x[2i-1] + 3x[2i] + 3x[2i+1] + x[2i+2]) / 8        per axis
//   up  : out[2i] = (x[i-1] + 3x[i]) / 4 ; out[2i+1] = (3x[i] + x[i+1]) / 4
// zero outside the image (zeros of the ACTIVATED tensor for the fused variant).
// ---------------------------------------------------------------------------------
template <typename T, bool UP, bool PLAIN, bool ACT>
__global__ void __launch_bounds__(256) resample2x_kernel(
    T* __restrict__ out_plain, T* __restrict__ out_act, const T* __restrict__ in,
    const float* __restrict__ stats, int nsplit, double inv_count, float eps,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    int H, int W, int C, int G, int chunks, int upg) {
  constexpr int VW = DT<T>::vecw;
  extern __shared__ float s_ss[];   // [2][C] scale / shift
  __shared__ double s_red[ACT ? 8 * 32 * 2 : 1];
  const int b = blockIdx.y;
  if (ACT) gn_scale_shift(s_ss, s_red, stats, nsplit, inv_count, eps, b, C, G, gamma, beta, upg);
  const int OH = UP ? 2 * H : H / 2, OW = UP ? 2 * W : W / 2;
  const int nvec = C / VW;
  const int64_t total = (int64_t)OH * OW * nvec;
  const int64_t per = (total + chunks - 1) / chunks;
  const int64_t i0 = blockIdx.x * per, i1 = min(total, i0 + per);
  const T* img = in + (int64_t)b * H * W * C;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const int v = (int)(i % nvec);
    const int64_t p = i / nvec;
    const int ox = (int)(p % OW), oy = (int)(p / OW);
    const int c = v * VW;
    float accp[VW], acca[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) accp[k] = acca[k] = 0.f;
    constexpr int NT = UP ? 2 : 4;
    int ys[NT], xs[NT];
    float wy[NT], wx[NT];
    if (UP) {
      const int iy = oy >> 1, ix = ox >> 1;
      if (oy & 1) { ys[0] = iy; wy[0] = 0.75f; ys[1] = iy + 1; wy[1] = 0.25f; }
      else        { ys[0] = iy - 1; wy[0] = 0.25f; ys[1] = iy; wy[1] = 0.75f; }
      if (ox & 1) { xs[0] = ix; wx[0] = 0.75f; xs[1] = ix + 1; wx[1] = 0.25f; }
      else        { xs[0] = ix - 1; wx[0] = 0.25f; xs[1] = ix; wx[1] = 0.75f; }
    } else {
      const float w4[4] = {0.125f, 0.375f, 0.375f, 0.125f};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ys[k] = 2 * oy - 1 + k; wy[k] = w4[k];
        xs[k] = 2 * ox - 1 + k; wx[k] = w4[k];
      }
    }
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      if (ys[a] < 0 || ys[a] >= H) continue;
#pragma unroll
      for (int bx = 0; bx < NT; ++bx) {
        if (xs[bx] < 0 || xs[bx] >= W) continue;
        const float wgt = wy[a] * wx[bx];
        float x[VW];
        Vec16<T>::load(img + ((int64_t)ys[a] * W + xs[bx]) * C + c, x);
#pragma unroll
        for (int k = 0; k < VW; ++k) {
          if (PLAIN) accp[k] += wgt * x[k];
          if (ACT) acca[k] += wgt * silu_t<T>(x[k] * s_ss[c + k] + s_ss[C + c + k]);
        }
      }
    }
    const int64_t o = (((int64_t)b * OH + oy) * OW + ox) * C + c;
    if (PLAIN) Vec16<T>::store(out_plain + o, accp);
    if (ACT) Vec16<T>::store(out_act + o, acca);
  }
}

// 2 x 2 OUTPUT BLOCK per thread, for the fused GroupNorm+SiLU case on maps too large for the tiled kernel below: the
// per-output kernel above evaluates the activation once per TAP (16 per output down, 4 per output up) and is VALU-bound
// there (1.6 TB/s at batch 64).  A thread that forms the 2 x 2 outputs around one point shares the activated inputs between
// them: down 6 x 6 inputs for 4 outputs (9 activations per output instead of 16), up 3 x 3 inputs for the 4 outputs
// of one input pixel (2.25 instead of 4).  The inputs are streamed row by row and every output accumulates its taps in
// the same order as above (rows outer, columns inner, same products): bit-identical results.
// down: H, W multiples of 4.
template <typename T, bool UP, bool PLAIN>
__global__ void __launch_bounds__(256) resample2x_quad_kernel(
    T* __restrict__ out_plain, T* __restrict__ out_act, const T* __restrict__ in,
    const float* __restrict__ stats, int nsplit, double inv_count, float eps,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    int H, int W, int C, int G, int chunks, int upg) {
  constexpr int VW = DT<T>::vecw;
  constexpr int NR = UP ? 3 : 6;               // input rows / columns of a block
  extern __shared__ float s_ss[];               // [2][C] scale / shift
  __shared__ double s_red[8 * 32 * 2];
  const int b = blockIdx.y;
  gn_scale_shift(s_ss, s_red, stats, nsplit, inv_count, eps, b, C, G, gamma, beta, upg);
  const int OH = UP ? 2 * H : H / 2, OW = UP ? 2 * W : W / 2;
  const int BH = OH / 2, BW = OW / 2;           // blocks of 2 x 2 outputs (up: one per input pixel)
  const int nvec = C / VW;
  const int64_t total = (int64_t)BH * BW * nvec;
  const int64_t per = (total + chunks - 1) / chunks;
  const int64_t i0 = blockIdx.x * per, i1 = min(total, i0 + per);
  const T* img = in + (int64_t)b * H * W * C;
  const float w4[4] = {0.125f, 0.375f, 0.375f, 0.125f};
  for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const int v = (int)(i % nvec);
    const int64_t pb = i / nvec;
    const int bx = (int)(pb % BW), by = (int)(pb / BW);
    const int c = v * VW;
    float sc[VW], sh[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) { sc[k] = s_ss[c + k]; sh[k] = s_ss[C + c + k]; }
    float accp[2][2][VW], acca[2][2][VW];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int k = 0; k < VW; ++k) accp[j][q][k] = acca[j][q][k] = 0.f;
    const int y00 = UP ? by - 1 : 4 * by - 1, x00 = UP ? bx - 1 : 4 * bx - 1;     // first input row / column of the block
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int iy = y00 + r;
      if (iy < 0 || iy >= H) continue;
      float xr[NR][VW], xa[NR][VW];
      bool okc[NR];
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        const int ix = x00 + q;
        okc[q] = ix >= 0 && ix < W;
        if (okc[q]) {
          Vec16<T>::load(img + ((int64_t)iy * W + ix) * C + c, xr[q]);
#pragma unroll
          for (int k = 0; k < VW; ++k) xa[q][k] = silu_t<T>(xr[q][k] * sc[k] + sh[k]);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // output row 2 by + j: down taps k = r - 2 j in 0..3 (weight w4[k]); up rows {r = j, j + 1} with weights
        // j == 0: {0.25, 0.75}, j == 1: {0.75, 0.25}
        const int k = UP ? r - j : r - 2 * j;
        if (k < 0 || k >= (UP ? 2 : 4)) continue;
        const float wy = UP ? ((k == 0) == (j == 0) ? 0.25f : 0.75f) : w4[k];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
          for (int t = 0; t < (UP ? 2 : 4); ++t) {
            const int cq = UP ? q + t : 2 * q + t;
            if (!okc[cq]) continue;
            const float wx = UP ? ((t == 0) == (q == 0) ? 0.25f : 0.75f) : w4[t];
            const float wgt = wy * wx;
#pragma unroll
            for (int e = 0; e < VW; ++e) {
              if (PLAIN) accp[j][q][e] += wgt * xr[cq][e];
              acca[j][q][e] += wgt * xa[cq][e];
            }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int64_t o = (((int64_t)b * OH + 2 * by + j) * OW + 2 * bx + q) * C + c;
        if (PLAIN) Vec16<T>::store(out_plain + o, accp[j][q]);
        Vec16<T>::store(out_act + o, acca[j][q]);
      }
  }
}

// Tiled variant for the fused GroupNorm+SiLU case: the per-output kernel above evaluates the
// activation once per TAP (16x per input element when upsampling, 4x when downsampling) and is
// VALU-bound there.  Here a workgroup stages an input tile with its halo ONCE - raw and activated,
// as f32 - in LDS and forms every output of the tile from it, in the same tap order as above
// (bit-identical results).  Tile: 8 x 8 output pixels (down) / 8 x 8 input pixels (up) x 16 channels.
template <typename T, bool UP, bool PLAIN>
__global__ void __launch_bounds__(256) resample2x_tile_kernel(
    T* __restrict__ out_plain, T* __restrict__ out_act, const T* __restrict__ in,
    const float* __restrict__ stats, int nsplit, double inv_count, float eps,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    int H, int W, int C, int G, int upg, int tiles_x, int ntiles) {
  constexpr int VW = DT<T>::vecw;
  constexpr int CCH = 16, CV = CCH / VW;          // channels per workgroup, 16-byte vectors of them
  constexpr int PS = CCH + 4;                       // LDS pixel stride in floats (bank spread)
  constexpr int IH = UP ? 10 : 18, IW = UP ? 10 : 18;       // staged input tile (with halo)
  constexpr int OHT = UP ? 16 : 8, OWT = UP ? 16 : 8;       // output tile
  extern __shared__ float s_all[];                  // [2][C] scale/shift | raw [IH*IW][PS] | act [IH*IW][PS]
  __shared__ double s_red[8 * 32 * 2];
  float* s_ss = s_all;
  float* s_raw = s_all + 2 * C;
  float* s_act = s_raw + IH * IW * PS;
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * CCH;
  gn_scale_shift(s_ss, s_red, stats, nsplit, inv_count, eps, b, C, G, gamma, beta, upg);
  const T* img = in + (int64_t)b * H * W * C;
  // a workgroup walks tiles blockIdx.x, + gridDim.x, ...: the scale/shift table (statistics rows -> fp64 reduce
  // -> per-channel entries, three barriers of pure latency) is built once per workgroup, not once per tile
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const int ty = tile / tiles_x, tx = tile % tiles_x;
  // input coordinates of the staged tile's first pixel
  const int iy0 = UP ? ty * 8 - 1 : ty * 16 - 1, ix0 = UP ? tx * 8 - 1 : tx * 16 - 1;
  for (int i = threadIdx.x; i < IH * IW * CV; i += 256) {
    const int v = i % CV, px = i / CV;
    const int iy = iy0 + px / IW, ix = ix0 + px % IW;
    float x[VW], a[VW];
    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
    if (ok) {
      Vec16<T>::load(img + ((int64_t)iy * W + ix) * C + c0 + v * VW, x);
#pragma unroll
      for (int k = 0; k < VW; ++k) a[k] = silu_t<T>(x[k] * s_ss[c0 + v * VW + k] + s_ss[C + c0 + v * VW + k]);
    } else {
#pragma unroll
      for (int k = 0; k < VW; ++k) x[k] = a[k] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      if (PLAIN) s_raw[px * PS + v * VW + k] = x[k];
      s_act[px * PS + v * VW + k] = a[k];
    }
  }
  __syncthreads();
  const int OH = UP ? 2 * H : H / 2, OW = UP ? 2 * W : W / 2;
  for (int o = threadIdx.x; o < OHT * OWT * CV; o += 256) {
    const int v = o % CV, op = o / CV;
    const int oyl = op / OWT, oxl = op % OWT;
    const int oy = ty * OHT + oyl, ox = tx * OWT + oxl;
    constexpr int NT = UP ? 2 : 4;
    int ys[NT], xs[NT];        // tile-local input coordinates of the taps
    float wy[NT], wx[NT];
    if (UP) {
      // global input row of tap 0: (oy odd ? oy/2 : oy/2 - 1); tile-local = that - iy0
      const int ly = (oyl >> 1) + 1, lx = (oxl >> 1) + 1;        // tile-local coordinates of input pixel (oy/2, ox/2)
      if (oyl & 1) { ys[0] = ly; wy[0] = 0.75f; ys[1] = ly + 1; wy[1] = 0.25f; }
      else         { ys[0] = ly - 1; wy[0] = 0.25f; ys[1] = ly; wy[1] = 0.75f; }
      if (oxl & 1) { xs[0] = lx; wx[0] = 0.75f; xs[1] = lx + 1; wx[1] = 0.25f; }
      else         { xs[0] = lx - 1; wx[0] = 0.25f; xs[1] = lx; wx[1] = 0.75f; }
    } else {
      const float w4[4] = {0.125f, 0.375f, 0.375f, 0.125f};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ys[k] = 2 * oyl + k; wy[k] = w4[k];          // global 2 oy - 1 + k, minus iy0 = 16 ty - 1
        xs[k] = 2 * oxl + k; wx[k] = w4[k];
      }
    }
    float accp[VW], acca[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) accp[k] = acca[k] = 0.f;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int bx = 0; bx < NT; ++bx) {
        const float wgt = wy[a] * wx[bx];
        const float* ra = s_act + (ys[a] * IW + xs[bx]) * PS + v * VW;
        const float* rr = s_raw + (ys[a] * IW + xs[bx]) * PS + v * VW;
#pragma unroll
        for (int k = 0; k < VW; ++k) {
          if (PLAIN) accp[k] += wgt * rr[k];
          acca[k] += wgt * ra[k];
        }
      }
    const int64_t oidx = (((int64_t)b * OH + oy) * OW + ox) * C + c0 + v * VW;
    if (PLAIN) Vec16<T>::store(out_plain + oidx, accp);
    Vec16<T>::store(out_act + oidx, acca);
  }
  __syncthreads();          // the staged tile is free for the next one
  }
}

// The whole progressive-input pyramid in ONE launch: level l+1 = downsample_2d(level l) of the 4-channel f32
// network input (ncsnpp_v2.py:296-305), 6 levels for the 7-level net.  One workgroup per image walks the
// levels (a workgroup barrier between them: every level is read only by the workgroup that wrote it); the
// per-level launches it replaces were pure launch floor (6 x ~9 us for < 1 MB of data).  Tap order of
// resample2x_kernel (bit-identical results).
struct PyrPtrs { float* out[8]; };

__global__ void __launch_bounds__(1024) pyramid_down_chain_kernel(const float* __restrict__ in, PyrPtrs ptrs,
                                                                  int levels, int H, int W) {
  const int b = blockIdx.x;
  const float* src = in + (int64_t)b * H * W * 4;
  int h = H, w = W;
  for (int l = 0; l < levels; ++l) {
    const int oh = h / 2, ow = w / 2;
    float* dst = ptrs.out[l] + (int64_t)b * oh * ow * 4;
    for (int i = threadIdx.x; i < oh * ow; i += blockDim.x) {
      const int oy = i / ow, ox = i - oy * ow;
      const float w4[4] = {0.125f, 0.375f, 0.375f, 0.125f};
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int iy = 2 * oy - 1 + a;
        if (iy < 0 || iy >= h) continue;
#pragma unroll
        for (int bx = 0; bx < 4; ++bx) {
          const int ix = 2 * ox - 1 + bx;
          if (ix < 0 || ix >= w) continue;
          const float wgt = w4[a] * w4[bx];
          const f32x4 v = *reinterpret_cast<const f32x4*>(src + ((int64_t)iy * w + ix) * 4);
          acc[0] += wgt * v[0]; acc[1] += wgt * v[1]; acc[2] += wgt * v[2]; acc[3] += wgt * v[3];
        }
      }
      *reinterpret_cast<f32x4*>(dst + (int64_t)i * 4) = acc;
    }
    __threadfence_block();
    __syncthreads();
    src = dst;
    h = oh; w = ow;
  }
}

extern "C" int fdbm_pyramid_down_chain(const float* in, void* const* outs, int levels, int B, int H, int W,
                                       void* stream) {
  FDBM_CHECK(in && outs && levels >= 1 && levels <= 8, "fdbm_pyramid_down_chain: bad arguments (levels=%d)", levels);
  FDBM_CHECK(B > 0 && H % (1 << levels) == 0 && W % (1 << levels) == 0,
             "fdbm_pyramid_down_chain: H, W (%d x %d) must be divisible by 2^levels", H, W);
  PyrPtrs ptrs;
  for (int l = 0; l < 8; ++l) ptrs.out[l] = l < levels ? (float*)outs[l] : nullptr;
  for (int l = 0; l < levels; ++l) FDBM_CHECK(ptrs.out[l], "fdbm_pyramid_down_chain: null output %d", l);
  pyramid_down_chain_kernel<<<B, 1024, 0, (hipStream_t)stream>>>(in, ptrs, levels, H, W);
  FDBM_LAUNCH_CHECK("fdbm_pyramid_down_chain");
  return 0;
}

extern "C" int fdbm_resample2x(void* out_plain, void* out_act, const void* in, const float* stats,
                               int nsplit, int64_t count, float eps,
                               const float* gamma, const float* beta, int B, int H, int W, int C,
                               int G, int up, int dtype, void* stream) {
  return fdbm_resample2x_units(out_plain, out_act, in, stats, nsplit, 1, count, eps, gamma, beta, B, H, W, C, G, up,
                               dtype, stream);
}

extern "C" int fdbm_resample2x_units(void* out_plain, void* out_act, const void* in, const float* stats,
                                     int nsplit, int stat_units, int64_t count, float eps,
                                     const float* gamma, const float* beta, int B, int H, int W, int C,
                                     int G, int up, int dtype, void* stream) {
  FDBM_CHECK(in && (out_plain || out_act), "fdbm_resample2x: null pointer");
  FDBM_CHECK(stat_units >= 1 && (stat_units == 1 || nsplit < 0), "fdbm_resample2x: unit statistics are fp64 partial sums (nsplit < 0)");
  FDBM_CHECK((out_act != nullptr) == (stats != nullptr), "fdbm_resample2x: out_act needs GroupNorm statistics (and vice versa)");
  FDBM_CHECK(!out_act || (gamma && beta && G > 0 && G <= 32 && C % G == 0 && C <= 1024), "fdbm_resample2x: bad GroupNorm arguments");
  FDBM_CHECK(nsplit == 0 || count > 0, "fdbm_resample2x: bad nsplit/count");
  const double inv_count = nsplit != 0 ? 1.0 / (double)count : 0.0;
  const int vw = dtype != FDBM_F32 ? 8 : 4;
  FDBM_CHECK(C % vw == 0, "fdbm_resample2x: C=%d must be a multiple of %d", C, vw);
  FDBM_CHECK(up || (H % 2 == 0 && W % 2 == 0), "fdbm_resample2x: downsampling needs even H, W (got %d x %d)", H, W);
  hipStream_t st = (hipStream_t)stream;
  // fused GroupNorm+SiLU on maps that tile: the LDS-staged kernel (one activation per input element)
  // ... up to 64 x 64 pixels.  Above that the register kernels (2 x 2 outputs per thread, or one output per thread where
  // that leaves too few workgroups) are faster at every batch size (measured against the tiled kernel: 48.6 -> 49.0 x
  // real time at batch 1, 129.8 -> 134.2 at batch 64): the tiled kernel's 16-channel slices read 32 bytes of every
  // 256-byte pixel per workgroup and its 18 x 18 staging pass keeps 60 % of the threads busy.
  static const char* tmax = getenv("FDBM_RESAMPLE_TILE_MAXHW");       // experiments
  const int64_t tile_max_hw = tmax ? atoll(tmax) : 4096;
  if (out_act && C % 16 == 0 && (int64_t)H * W <= tile_max_hw && (up ? (H % 8 == 0 && W % 8 == 0) : (H % 16 == 0 && W % 16 == 0))) {
    const int tiles_x = up ? W / 8 : W / 16, tiles_y = up ? H / 8 : H / 16;
    const int npx = up ? 100 : 324;
    const size_t smem = (2 * (size_t)C + 2 * (size_t)npx * 20) * sizeof(float);
    // >= 2 workgroups per CU in flight, then tiles are walked inside the workgroup
    const int ntiles = tiles_x * tiles_y;
    int gx = ntiles;
    while (gx > 64 && (int64_t)(gx / 2) * (C / 16) * B >= 512) gx = (gx + 1) / 2;
    dim3 grid(gx, C / 16, B);
#define RT(TT, U, P)                                                                                     \
  do {                                                                                                   \
    static bool attr = false;                                                                            \
    if (!attr) {                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resample2x_tile_kernel<TT, U, P>),        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);                  \
      attr = true;                                                                                       \
    }                                                                                                    \
    resample2x_tile_kernel<TT, U, P><<<grid, 256, smem, st>>>((TT*)out_plain, (TT*)out_act, (const TT*)in, stats, nsplit, \
                                                              inv_count, eps, gamma, beta, H, W, C, G, stat_units, tiles_x, ntiles); \
  } while (0)
#define RT_DISPATCH(TT)                                                              \
  do {                                                                               \
    if (up) { if (out_plain) RT(TT, true, true); else RT(TT, true, false); }         \
    else    { if (out_plain) RT(TT, false, true); else RT(TT, false, false); }       \
  } while (0)
    if (smem <= 64 * 1024) {
      if (dtype == FDBM_BF16) RT_DISPATCH(bf16_t);
      else if (dtype == FDBM_F16) RT_DISPATCH(f16_t);
      else if (dtype == FDBM_F32) RT_DISPATCH(float);
      else FDBM_CHECK(false, "fdbm_resample2x: bad dtype %d", dtype);
      FDBM_LAUNCH_CHECK("fdbm_resample2x(tile)");
      return 0;
    }
#undef RT_DISPATCH
#undef RT
  }
  const int OH = up ? 2 * H : H / 2, OW = up ? 2 * W : W / 2;
  // fused GroupNorm+SiLU on the large maps: a 2 x 2 output block per thread (shared activations)
  static const char* qenv = getenv("FDBM_RESAMPLE_QUAD");              // experiments: 0 = the per-output kernel
  // (where it still fills the chip: with fewer than 512 workgroups - the 128 x 128 x 128 level at batch 1 - the
  //  per-output kernel's four times as many threads win, 9 us against 16)
  const int64_t totalq = (int64_t)(OH / 2) * (OW / 2) * (C / vw);
  if (out_act && !(qenv && qenv[0] == '0') && totalq * B >= 131072 && (up || (H % 4 == 0 && W % 4 == 0))) {
    int chunks = (int)((totalq + 255) / 256);
    if (chunks > 4096) chunks = 4096;
    if (chunks < 1) chunks = 1;
    dim3 grid(chunks, B);
    const size_t smem = 2 * (size_t)C * sizeof(float);
#define RQ(TT, U, P) resample2x_quad_kernel<TT, U, P><<<grid, 256, smem, st>>>((TT*)out_plain, (TT*)out_act, (const TT*)in, stats, nsplit, inv_count, eps, gamma, beta, H, W, C, G, chunks, stat_units)
#define RQ_DISPATCH(TT)                                                         \
  do {                                                                          \
    if (up) { if (out_plain) RQ(TT, true, true); else RQ(TT, true, false); }    \
    else    { if (out_plain) RQ(TT, false, true); else RQ(TT, false, false); }  \
  } while (0)
    if (dtype == FDBM_BF16) RQ_DISPATCH(bf16_t);
    else if (dtype == FDBM_F16) RQ_DISPATCH(f16_t);
    else if (dtype == FDBM_F32) RQ_DISPATCH(float);
    else FDBM_CHECK(false, "fdbm_resample2x: bad dtype %d", dtype);
#undef RQ_DISPATCH
#undef RQ
    FDBM_LAUNCH_CHECK("fdbm_resample2x(quad)");
    return 0;
  }
  const int64_t total = (int64_t)OH * OW * (C / vw);
  // one output vector per thread up to 2048 workgroups (4 per thread left a 1 MB pyramid level on 16 CUs: 13 us of latency)
  int chunks = (int)((total + 255) / 256);
  static const char* cmax = getenv("FDBM_RESAMPLE_MAX_CHUNKS");      // experiments
  const int chunk_cap = cmax ? atoi(cmax) : 2048;
  if (chunks > chunk_cap) chunks = chunk_cap;
  if (chunks < 1) chunks = 1;
  dim3 grid(chunks, B);
  const size_t smem = out_act ? 2 * (size_t)C * sizeof(float) : 0;
#define RS(TT, U, P, A) resample2x_kernel<TT, U, P, A><<<grid, 256, smem, st>>>((TT*)out_plain, (TT*)out_act, (const TT*)in, stats, nsplit, inv_count, eps, gamma, beta, H, W, C, G, chunks, stat_units)
#define RS_DISPATCH(TT)                                                         \
  do {                                                                          \
    const bool P_ = out_plain != nullptr, A_ = out_act != nullptr;              \
    if (up) { if (P_ && A_) RS(TT, true, true, true); else if (P_) RS(TT, true, true, false); else RS(TT, true, false, true); } \
    else    { if (P_ && A_) RS(TT, false, true, true); else if (P_) RS(TT, false, true, false); else RS(TT, false, false, true); } \
  } while (0)
  if (dtype == FDBM_BF16) RS_DISPATCH(bf16_t);
  else if (dtype == FDBM_F16) RS_DISPATCH(f16_t);
  else if (dtype == FDBM_F32) RS_DISPATCH(float);
  else FDBM_CHECK(false, "fdbm_resample2x: bad dtype %d", dtype);
#undef RS_DISPATCH
#undef RS
  FDBM_LAUNCH_CHECK("fdbm_resample2x");
  return 0;
}
