// conv_mid.hip - the 3x3 convolution of the 64 x 64 level at batch 1 (16-bit tensors): the wave-per-tap kernel's tile and wave
// roles with the whole-map kernel's launch body.
//
// Why it exists (tools/tap_timeline.py on 64 x 64, 256 -> 256 with GroupNorm, round 3): a workgroup of the wave-per-tap kernel
// (conv_tap.hip) takes 14.6 us - 4.0 us until the first 64-channel chunk's patch is in LDS (statistics table: a dependent
// round trip, two barriers, fp64 work on 32 threads), then 4 chunks x 1.85 us, each a serial "weights arrive -> transform ->
// barrier -> MFMA" step fed at ~70 GB/s per CU (the 295 KB of weight fragments of a 64-channel n-block come out of the L2 at
// ~18 TB/s over all CUs), then 2.9 us of reduction and epilogue.  The tile is right - 4 x 16 pixels x 64 output channels, 256
// workgroups, GroupNorm + SiLU applied 1.7 x 4 times per element instead of 3.1 x 16 - the chain is not.  Here:
//   * every global load is requested up front in straight-line code from a host-reduced argument block (conv_small.hip's
//     rule): the statistics items, the halo patches of ALL staged channels (up to 2 passes of <= 256), the raw 1-tap
//     pixels, and this wave's weight fragments of the first pass (wave w = tap w, 4 n-tiles x 8 k-steps = 128 registers);
//     a later pass's fragments are requested k-step by k-step into the registers the MFMAs have just released;
//   * GroupNorm table without LDS partials: thread (group g = tid / 16, lane j = tid % 16) sums <= 4 of the group's
//     (unit, partial row) items in fp64, a DPP row reduction gives every lane of the row the same total, lane j writes the
//     scale / shift of the group's channel j - one barrier, nothing serial;
//   * all passes are transformed and written to LDS (two patch buffers) BEFORE the first MFMA, so the MFMA phase is one
//     run of ds_read_b128 + v_mfma with the weight stream behind it;
//   * tap 8 and the 1-tap segments (the res-block's 1x1 shortcut) are dealt out k-step by k-step over the 8 waves, the 8 partial
//     accumulator sets are summed through LDS (over the patches), 2 output tiles per wave in the epilogue.
// Results: the convolution of every other kernel of fdbm_conv_igemm (fp32 sums of the same products in a different order);
// GroupNorm mean / rstd from the producers' fp64 unit sums (the formula of conv_gn_table).
// Roofline: latency / L2 bandwidth (2 us of MFMA work per CU); selected by fdbm_conv_igemm for 16-bit tensors when the launch
// is <= 512 such tiles (fdbm_conv_mid_ok).
#include "conv_common.h"

#define MID_NTHR 512
#define MID_RS 264                     // LDS pixel stride of a patch (elements): 256 channels + 8
#define MID_PC 18                      // patch columns (16 + halo)
#define MID_PP 108                     // patch pixels (6 rows x 18)
#define MID_PATCH_BYTES (MID_PP * MID_RS * 2)
#define MID_MAIN_BYTES (8 * 16 * 1024)  // the reduction's 8 x 16 tiles of f32x4 x 64 lanes; >= 2 patches, >= patch + 1-tap pixels

#ifdef FDBM_STAMPS
#define MSTAMP(i)                                                                                         \
  do {                                                                                                    \
    if (a.stamps && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)             \
      a.stamps[i] = __builtin_amdgcn_s_memrealtime();                                                     \
  } while (0)
#else
#define MSTAMP(i)
#endif

namespace {

struct MidArgs {
  const void* w;                      // fragment-major weights: a (64-channel chunk, 16-channel n-tile) block is 1 024 elements
  int64_t wchunk;                     // stride between consecutive chunks (= CoutPad / 16 blocks)
  // staged (9-tap) passes: <= 256 channels each
  int npass;
  const void* src9[2];                // + coff
  int sC9[2], cin9[2], gnoff[2];      // pixel stride, channels, first channel inside the normalised input
  int64_t wb9[2], wtap9[2];           // offset of tap 0 / chunk 0 / n-tile 0; stride between taps
  // raw 1-tap segments
  const void* src1[2];
  int sC1[2], c1_0, c1;
  int64_t wb1[2];
  int H, W;
  // GroupNorm over the staged channels, statistics from the producers' unit sums: segment s -> doubles [B][unsp][ucnt][2]
  const double* useg[2];
  int unsp[2], uoff[2], ucnt[2], nuseg;
  double inv_count_d;
  const float* gamma;
  const float* beta;
  int cpg, silu;
  float eps;
  // epilogue
  const float* bias;
  const float* tbias;
  int tb_stride;
  const void* res;
  const float* res_lo;
  const float* comb_pyr;
  const float* comb_w;
  const float* comb_b;
  float scale;
  int Cout;
  unsigned long long* stamps;
};

static bool mid_plan(const ConvParams& p, MidArgs* out) {
  MidArgs a;
  memset(&a, 0, sizeof(a));
  const bool gn = p.gn_sums != nullptr;
  if (p.H % 4 != 0 || p.W % 16 != 0 || p.Cout % 64 != 0) return false;
  const int64_t wgs = (int64_t)(p.Cout / 64) * (p.H / 4) * (p.W / 16) * p.B;
  static const char* mx = getenv("FDBM_MID_MAX_GRID");            // experiments
  static const char* mn = getenv("FDBM_MID_MIN_GRID");
  // (fewer workgroups - 128 output channels on the 64 x 64 level - and the wave-per-tap kernel's 32-channel tiles are as fast:
  // 12.3 vs 12.3 us, 13.5 vs 12.9 with a shortcut; tools/mid_micro.py)
  if (wgs > (mx ? atoi(mx) : 512) || wgs < (mn ? atoi(mn) : 192)) return false;
  const int64_t blk = 1024, ntq = p.CoutPad / 16;
  int kb = 0;
  bool seen1 = false;
  for (int i = 0; i < p.nseg; ++i) {
    const fdbm_conv_seg& sg = p.seg[i];
    if (sg.cin % 64 != 0 || sg.coff % 8 != 0 || sg.C % 8 != 0) return false;
    const int nch = sg.cin / 64;
    if (sg.taps == 9) {
      if (seen1 || a.npass >= 2 || (sg.cin != 64 && sg.cin != 128 && sg.cin != 256)) return false;
      if (gn != (p.seg_gn[i] >= 0)) return false;                  // (the GroupNorm covers the 9-tap segments, all of them)
      const int s = a.npass++;
      a.src9[s] = reinterpret_cast<const unsigned char*>(sg.src) + (int64_t)sg.coff * 2;
      a.sC9[s] = sg.C; a.cin9[s] = sg.cin; a.gnoff[s] = gn ? p.seg_gn[i] : 0;
      a.wb9[s] = (int64_t)kb * ntq * blk; a.wtap9[s] = (int64_t)nch * ntq * blk;
      if (gn) {
        if (!p.gn_unit) return false;
        a.useg[s] = p.gn_useg[i]; a.unsp[s] = p.gn_unsp[i]; a.uoff[s] = p.gn_uoff[i]; a.ucnt[s] = p.gn_ucnt[i];
        if (a.unsp[s] < 1 || a.uoff[s] * 4 != a.gnoff[s]) return false;
        a.nuseg = s + 1;
      }
    } else {
      seen1 = true;
      if (gn && p.seg_gn[i] >= 0) return false;
      const int r = a.c1 == 0 ? 0 : 1;
      if (r == 1 && a.src1[1]) return false;
      a.src1[r] = reinterpret_cast<const unsigned char*>(sg.src) + (int64_t)sg.coff * 2;
      a.sC1[r] = sg.C; a.wb1[r] = (int64_t)kb * ntq * blk;
      if (r == 0) a.c1_0 = sg.cin;
      a.c1 += sg.cin;
    }
    kb += sg.taps * nch;
  }
  if (a.npass == 0 || a.c1 > 512) return false;
  if (a.npass == 2 && a.c1 > 0) return false;                      // (second patch buffer and 1-tap pixels share the LDS)
  if (gn) {
    if (p.gn_G != 32 || p.gn_C % 32 != 0) return false;            // thread (tid / 16, tid % 16) = (group, lane)
    a.cpg = p.gn_C / 32;
    if (a.cpg % 4 != 0 || a.cpg > 16) return false;
    int csum = 0, maxsp = 0;
    for (int s = 0; s < a.npass; ++s) { csum += a.cin9[s]; maxsp = a.unsp[s] > maxsp ? a.unsp[s] : maxsp; }
    if (csum != p.gn_C || a.gnoff[0] != 0 || (a.npass == 2 && a.gnoff[1] != a.cin9[0])) return false;
    if ((a.cpg / 4) * maxsp > 64) return false;                    // <= 4 items per lane
    a.gamma = p.gn_gamma; a.beta = p.gn_beta; a.silu = p.gn_silu; a.eps = p.gn_eps; a.inv_count_d = p.gn_inv_count;
  }
  a.w = p.w; a.wchunk = ntq * blk; a.H = p.H; a.W = p.W;
  a.bias = p.bias; a.tbias = p.tbias; a.tb_stride = p.tbias_stride; a.res = p.res; a.res_lo = p.res_lo;
  a.comb_pyr = p.comb_pyr; a.comb_w = p.comb_w; a.comb_b = p.comb_b; a.scale = p.scale; a.Cout = p.Cout;
  a.stamps = reinterpret_cast<unsigned long long*>(p.partial);
  *out = a;
  return true;
}

template <typename T>
__device__ __forceinline__ uint4 mld16(const T* p) { return *reinterpret_cast<const uint4*>(p); }

template <typename T, typename TO, bool GNP>
__global__ void __launch_bounds__(MID_NTHR) conv_mid_kernel(const MidArgs a, TO* __restrict__ out, double* __restrict__ stat_out,
                                                            int stat_G, int stat_nsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* s_p0 = reinterpret_cast<T*>(smem);
  T* s_p1 = reinterpret_cast<T*>(smem + MID_PATCH_BYTES);           // second pass's patch, or the raw 1-tap pixels [64][c1 + 8]
  f32x4* s_red = reinterpret_cast<f32x4*>(smem);                    // [8 waves][16 tiles][64 lanes] (after the MFMAs)
  float* s_sc = reinterpret_cast<float*>(smem + MID_MAIN_BYTES);    // [512] scale | [512] shift
  float* s_sh = s_sc + 512;
  double* s_ostat = reinterpret_cast<double*>(s_sh + 512);          // [64][2] output unit statistics

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fk = lane >> 4;
  const int H = a.H, W = a.W;
  const int tiles_x = W >> 4;
  const int ty = blockIdx.y / tiles_x, tx = blockIdx.y - ty * tiles_x;
  const int row0 = ty * 4, col0 = tx * 16, b = blockIdx.z;
  const int ntile0 = blockIdx.x * 4;
  const int64_t img = (int64_t)b * H * W;
  const int RS1 = a.c1 + 8;
  MSTAMP(0);

  // ---- 1. every global load, straight-line ------------------------------------------------------------------------------------------
  // (a) statistics items of group g = tid / 16: item t = j + 16 q -> (unit g * upg + t % upg, partial row t / upg)
  [[maybe_unused]] double st0[4], st1[4];
  [[maybe_unused]] float ga = 0.f, be = 0.f;
  if constexpr (GNP) {
    const int g = tid >> 4, j = tid & 15;
    const int upg = a.cpg >> 2;
    const int maxsp = a.nuseg > 1 ? max(a.unsp[0], a.unsp[1]) : a.unsp[0];
    const int nitem = upg * maxsp;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = j + 16 * q;
      const int tt = min(t, nitem - 1);
      const int u = g * upg + tt % upg, sp = tt / upg;
      const bool s1 = a.nuseg > 1 && u >= a.uoff[1];
      const int nsp = s1 ? a.unsp[1] : a.unsp[0];
      const bool live = t < nitem && sp < nsp;
      const double* r = (s1 ? a.useg[1] : a.useg[0]) + (((int64_t)b * nsp + (live ? sp : 0)) * (s1 ? a.ucnt[1] : a.ucnt[0]) + (u - (s1 ? a.uoff[1] : a.uoff[0]))) * 2;
      const double v0 = r[0], v1 = r[1];
      st0[q] = live ? v0 : 0.0; st1[q] = live ? v1 : 0.0;
    }
    const int c = g * a.cpg + min(j, a.cpg - 1);
    ga = a.gamma[c]; be = a.beta[c];
  }
  // (b) the halo patches: thread tid keeps item column tid % ipp (8 channels) of patch pixels tid / ipp + (512 / ipp) j.  The second
  // register set holds the second pass's patch or (one pass) the raw 1-tap pixels: the tile's 64 pixels x c1 / 8 items
  uint4 rawA[7], rawB[8];
  bool inA[7], inB[7];
  const int ipp1 = max(a.c1 >> 3, 1);
  auto load_patch = [&](const int s, uint4* raw, bool* in) __attribute__((always_inline)) {
    const int cin = s ? a.cin9[1] : a.cin9[0], sC = s ? a.sC9[1] : a.sC9[0];
    const int ish = cin == 256 ? 5 : cin == 128 ? 4 : 3;            // log2(items per pixel): 256 | 128 | 64 channels
    const int icol = tid & ((1 << ish) - 1);
    const T* src = reinterpret_cast<const T*>(s ? a.src9[1] : a.src9[0]) + img * sC + icol * 8;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int pp = min((tid + MID_NTHR * j) >> ish, MID_PP - 1);
      const int pr = pp / MID_PC, pc = pp - pr * MID_PC;
      const int row = row0 - 1 + pr, col = col0 - 1 + pc;
      in[j] = row >= 0 && row < H && col >= 0 && col < W;
      const int rc = min(max(row, 0), H - 1), cc = min(max(col, 0), W - 1);
      raw[j] = mld16(src + (int64_t)(rc * W + cc) * sC);
    }
  };
  load_patch(0, rawA, inA);
  if (a.npass > 1) {
    load_patch(1, rawB, inB);
  } else if (a.c1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = min(tid + MID_NTHR * j, 64 * ipp1 - 1);
      const int px = q / ipp1, ch = (q - px * ipp1) * 8;
      const bool s1 = ch >= a.c1_0;
      const int64_t pix = img + (int64_t)(row0 + (px >> 4)) * W + col0 + (px & 15);
      rawB[j] = mld16(reinterpret_cast<const T*>(s1 ? a.src1[1] : a.src1[0]) + pix * (s1 ? a.sC1[1] : a.sC1[0]) + (ch - (s1 ? a.c1_0 : 0)));
    }
  }
  // (the weight fragments are requested in step 3, between the items of the transform: 43 x 16-byte loads per thread in one run
  // keep the memory pipe's queue full for 2 us - 350 KB through a 64 B/clk port - and a wave stuck issuing them reaches the
  // statistics barrier only then; requested behind the barrier they stream in while the VALU does the GroupNorm + SiLU)
  const T* wl = reinterpret_cast<const T*>(a.w) + (int64_t)ntile0 * 1024 + fk * 128 + frow * 8;
  const int ks0 = a.cin9[0] >> 5, ks1p = a.npass > 1 ? a.cin9[1] >> 5 : 0;
  uint4 wown[8][4];                                                  // this wave's tap (waves 0-7 = taps 0-7) of pass 0: k-steps x n-tiles
  auto load_wown = [&](const int k) __attribute__((always_inline)) {
    const T* wp = wl + a.wb9[0] + (int64_t)wave * a.wtap9[0];
    const int kk = min(k, ks0 - 1);
#pragma unroll
    for (int n = 0; n < 4; ++n) wown[k][n] = mld16(wp + (int64_t)(kk >> 1) * a.wchunk + (kk & 1) * 512 + n * 1024);
  };
  MSTAMP(1);

  // ---- 2. GroupNorm table --------------------------------------------------------------------------------------------------------------
  if (stat_out)
    for (int i = tid; i < 128; i += MID_NTHR) s_ostat[i] = 0.0;
  if constexpr (GNP) {
    double s0 = (st0[0] + st0[1]) + (st0[2] + st0[3]), s1 = (st1[0] + st1[1]) + (st1[2] + st1[3]);
    s0 = lanes_sum_d(s0, 16); s1 = lanes_sum_d(s1, 16);
    const double mean = s0 * a.inv_count_d;
    double var = s1 * a.inv_count_d - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const double x = var + (double)a.eps;
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    const int g = tid >> 4, j = tid & 15;
    if (j < a.cpg) {
      const float sc = (float)r * ga;
      s_sc[g * a.cpg + j] = sc;
      s_sh[g * a.cpg + j] = be - (float)mean * sc;
    }
    __syncthreads();
  }
  MSTAMP(2);

  // ---- 3. transform + stage every pass's patch; the raw 1-tap pixels ------------------------------------------------------------
  auto stage_patch = [&](const int s, const uint4* raw, const bool* in) __attribute__((always_inline)) {
    const int cin = s ? a.cin9[1] : a.cin9[0];
    const int ish = cin == 256 ? 5 : cin == 128 ? 4 : 3;
    const int icol = tid & ((1 << ish) - 1);
    T* dst = (s ? s_p1 : s_p0) + icol * 8;
    [[maybe_unused]] f32x4 sc0, sc1, sh0, sh1;
    if constexpr (GNP) {
      const int c = (s ? a.gnoff[1] : a.gnoff[0]) + icol * 8;
      sc0 = *reinterpret_cast<const f32x4*>(s_sc + c); sc1 = *reinterpret_cast<const f32x4*>(s_sc + c + 4);
      sh0 = *reinterpret_cast<const f32x4*>(s_sh + c); sh1 = *reinterpret_cast<const f32x4*>(s_sh + c + 4);
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int pp = (tid + MID_NTHR * j) >> ish;
      if (pp < MID_PP) {
        uint4 v = raw[j];
        if constexpr (GNP) {
          typename V16<T>::x8 e = *reinterpret_cast<typename V16<T>::x8*>(&v);
          float y[8];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            y[k] = (float)e[k] * sc0[k] + sh0[k];
            y[4 + k] = (float)e[4 + k] * sc1[k] + sh1[k];
          }
          if (a.silu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) y[k] = silu_f(y[k]);
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) e[k] = (T)y[k];
          v = *reinterpret_cast<uint4*>(&e);
        }
        if (!in[j]) v = uint4{0u, 0u, 0u, 0u};
        *reinterpret_cast<uint4*>(dst + pp * MID_RS) = v;
      }
      if (s == 0) load_wown(j);
    }
    if (s == 0) load_wown(7);
  };
  stage_patch(0, rawA, inA);
  if (a.npass > 1) {
    stage_patch(1, rawB, inB);
  } else if (a.c1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = tid + MID_NTHR * j;
      if (q < 64 * ipp1) {
        const int px = q / ipp1;
        *reinterpret_cast<uint4*>(s_p1 + px * RS1 + (q - px * ipp1) * 8) = rawB[j];
      }
    }
  }
  // this wave's tap-8 k-step of each pass (requested now: the patch registers are free)
  uint4 wp8[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int kk = min(wave, max((s == 0 ? ks0 : ks1p) - 1, 0));
    const int ss = s < a.npass ? s : 0;
    const T* wp = wl + (ss ? a.wb9[1] : a.wb9[0]) + 8 * (ss ? a.wtap9[1] : a.wtap9[0]);
#pragma unroll
    for (int n = 0; n < 4; ++n) wp8[s][n] = mld16(wp + (int64_t)(kk >> 1) * a.wchunk + (kk & 1) * 512 + n * 1024);
  }
  __syncthreads();
  MSTAMP(3);

  // ---- 4. MFMAs: acc[t][n] = image row row0 + t (16 pixels) x n-tile n ---------------------------------------------------------------
  f32x4 acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int dy = wave / 3, dx = wave - dy * 3;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (s < a.npass) {
      const int ks = s == 0 ? ks0 : ks1p;
      const T* arow = (s ? s_p1 : s_p0) + (dy * MID_PC + frow + dx) * MID_RS + fk * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k < ks) {
          uint4 bf[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) bf[t] = *reinterpret_cast<const uint4*>(arow + t * MID_PC * MID_RS + k * 32);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int n = 0; n < 4; ++n) Mfma<T>::run(wown[k][n], bf[t], acc[t][n]);
        }
        if (s == 0 && a.npass > 1) {                                 // the next pass's k-step k into the registers just released
          const T* wp = wl + a.wb9[1] + (int64_t)wave * a.wtap9[1];
          const int kk = min(k, ks1p - 1);
#pragma unroll
          for (int n = 0; n < 4; ++n) wown[k][n] = mld16(wp + (int64_t)(kk >> 1) * a.wchunk + (kk & 1) * 512 + n * 1024);
        }
      }
    }
  }
  // 1-tap k-steps of this wave: wave, wave + 8 (requested now: the registers of the 9-tap fragments are free)
  const int nks1 = a.c1 >> 5;
  uint4 w1[2][4];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int kk = min(wave + 8 * q, max(nks1 - 1, 0));
    const int ch = kk * 32;
    const bool s1 = ch >= a.c1_0 && a.c1_0 < a.c1;
    const int rel = ch - (s1 ? a.c1_0 : 0);
    const T* wp = wl + (s1 ? a.wb1[1] : a.wb1[0]) + (int64_t)(rel >> 6) * a.wchunk + ((rel >> 5) & 1) * 512;
#pragma unroll
    for (int n = 0; n < 4; ++n) w1[q][n] = mld16(a.c1 ? wp + n * 1024 : wl);
  }
  // tap 8 (+2, +2): k-step `wave` of each pass
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (s < a.npass && wave < (s == 0 ? ks0 : ks1p)) {
      const T* arow = (s ? s_p1 : s_p0) + (2 * MID_PC + frow + 2) * MID_RS + fk * 8 + wave * 32;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint4 bf = *reinterpret_cast<const uint4*>(arow + t * MID_PC * MID_RS);
#pragma unroll
        for (int n = 0; n < 4; ++n) Mfma<T>::run(wp8[s][n], bf, acc[t][n]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int kk = wave + 8 * q;
    if (kk < nks1) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint4 bf = *reinterpret_cast<const uint4*>(s_p1 + (t * 16 + frow) * RS1 + kk * 32 + fk * 8);
#pragma unroll
        for (int n = 0; n < 4; ++n) Mfma<T>::run(w1[q][n], bf, acc[t][n]);
      }
    }
  }
  MSTAMP(4);

  // ---- 5. the epilogue's operands (tiles e = 2 wave, 2 wave + 1: image row e / 4, n-tile e % 4), then the 8-way sum through LDS -----
  const int Cout = a.Cout;
  int64_t pme[2];
  int n_out[2];
  f32x4 e_bias[2], e_tb[2];
  float e_res[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = 2 * wave + i;
    pme[i] = img + (int64_t)(row0 + (e >> 2)) * W + col0 + frow;
    n_out[i] = (ntile0 + (e & 3)) * 16 + fk * 4;
    e_bias[i] = *reinterpret_cast<const f32x4*>(a.bias ? a.bias + n_out[i] : g_conv_zero);
    e_tb[i] = *reinterpret_cast<const f32x4*>(a.tbias ? a.tbias + (int64_t)b * a.tb_stride + n_out[i] : g_conv_zero);
    OutVec<TO>::load(a.res ? reinterpret_cast<const TO*>(a.res) + pme[i] * Cout + n_out[i] : reinterpret_cast<const TO*>(g_conv_zero), e_res[i]);
  }
  __syncthreads();                                                  // (everybody has read the patches: the sums go over them)
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int n = 0; n < 4; ++n) s_red[(wave * 16 + t * 4 + n) * 64 + lane] = acc[t][n];
  __syncthreads();
  MSTAMP(5);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = 2 * wave + i;
    f32x4 pv[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) pv[w] = s_red[(w * 16 + e) * 64 + lane];
    f32x4 s = pv[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) s += pv[w];
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = ((s[r] + e_bias[i][r]) + e_tb[i][r]) + e_res[i][r];
    if (a.res_lo) {                                                 // upsampled half-resolution residual (tap order of conv_epilogue4)
      const int y = row0 + (e >> 2), x = col0 + frow, H2 = H >> 1, W2 = W >> 1, iy = y >> 1, ix = x >> 1;
      const int ys0 = (y & 1) ? iy : iy - 1, xs0 = (x & 1) ? ix : ix - 1;
      const float wy0 = (y & 1) ? 0.75f : 0.25f, wx0 = (x & 1) ? 0.75f : 0.25f;
      f32x4 q[4];
      float wq[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ay = ys0 + (t >> 1), ax = xs0 + (t & 1);
        const bool in = ay >= 0 && ay < H2 && ax >= 0 && ax < W2;
        wq[t] = in ? ((t >> 1) ? 1.0f - wy0 : wy0) * ((t & 1) ? 1.0f - wx0 : wx0) : 0.f;
        q[t] = *reinterpret_cast<const f32x4*>(in ? a.res_lo + (((int64_t)b * H2 + ay) * W2 + ax) * Cout + n_out[i] : g_conv_zero);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float up = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) up += wq[t] * q[t][r];
        v[r] += up;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= a.scale;
    if (a.comb_pyr) {                                               // Combine('sum') with the input pyramid
      const f32x4 cq = *reinterpret_cast<const f32x4*>(a.comb_pyr + pme[i] * 4);
      const f32x4 cb = *reinterpret_cast<const f32x4*>(a.comb_b + n_out[i]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 cw = *reinterpret_cast<const f32x4*>(a.comb_w + (int64_t)(n_out[i] + r) * 4);
        v[r] += cb[r] + (((cw[0] * cq[0] + cw[1] * cq[1]) + cw[2] * cq[2]) + cw[3] * cq[3]);
      }
    }
    OutVec<TO>::store(out + pme[i] * Cout + n_out[i], v);
    if constexpr (sizeof(TO) == 2) {                                // statistics are those of the STORED tensor
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (float)(TO)v[r];
    }
    if (stat_out) {
      const int scpg = Cout / stat_G;
      const float q1 = (v[0] + v[1]) + (v[2] + v[3]);
      const float q2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      const float r1 = row16_sum(q1), r2 = row16_sum(q2);
      if (frow == 0) {
        atomicAdd(&s_ostat[((n_out[i] - ntile0 * 16) / scpg) * 2], (double)r1);
        atomicAdd(&s_ostat[((n_out[i] - ntile0 * 16) / scpg) * 2 + 1], (double)r2);
      }
    }
  }
  if (stat_out) {
    __syncthreads();
    const int scpg = Cout / stat_G;
    const int g_0 = (ntile0 * 16) / scpg;
    const int ng = min(stat_G - g_0, (64 + scpg - 1) / scpg);
    if (tid < ng * 2) {
      const int k = tid & 1, g = g_0 + (tid >> 1);
      atomicAdd(stat_out + (((int64_t)b * stat_nsplit + blockIdx.y % stat_nsplit) * stat_G + g) * 2 + k, s_ostat[(g - g_0) * 2 + k]);
    }
  }
  if (wave == 0) MSTAMP(6);
}

template <typename T, typename TO, bool GNP>
static int launch_mid_i(const ConvParams& p, const MidArgs& a, hipStream_t st) {
  auto kern = &conv_mid_kernel<T, TO, GNP>;
  const int lds = MID_MAIN_BYTES + 2 * 512 * 4 + 64 * 2 * 8;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      fdbm_set_error("fdbm_conv_igemm(mid): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr = true;
  }
  dim3 grid((unsigned)(p.Cout / 64), (unsigned)((p.H / 4) * (p.W / 16)), (unsigned)p.B);
  kern<<<grid, MID_NTHR, lds, st>>>(a, reinterpret_cast<TO*>(p.out), p.stat_out, p.stat_G, p.stat_nsplit);
  FDBM_LAUNCH_CHECK("fdbm_conv_igemm(mid)");
  return 0;
}

}  // namespace

// can this convolution run on the 64-channel-block kernel?  (16-bit tensors in and out; p filled by fdbm_conv_igemm, p.w =
// fragment-major weights)
bool fdbm_conv_mid_ok(const ConvParams& p) {
  MidArgs a;
  if (!mid_plan(p, &a)) return false;
  if (p.stat_out && (p.Cout % p.stat_G != 0 || 64 % (p.Cout / p.stat_G) != 0)) return false;
  return (int64_t)MID_PATCH_BYTES + 64 * (a.c1 + 8) * 2 <= MID_MAIN_BYTES;
}

int fdbm_launch_conv_mid(const ConvParams& p, int dt_in, int dt_out, hipStream_t st) {
  MidArgs a;
  if (!mid_plan(p, &a) || !fdbm_conv_mid_ok(p) || dt_in != dt_out) {
    fdbm_set_error("fdbm_conv_igemm(mid): shape not supported");
    return 1;
  }
  const bool gn = p.gn_sums != nullptr;
  if (dt_in == FDBM_BF16) return gn ? launch_mid_i<bf16_t, bf16_t, true>(p, a, st) : launch_mid_i<bf16_t, bf16_t, false>(p, a, st);
  if (dt_in == FDBM_F16) return gn ? launch_mid_i<f16_t, f16_t, true>(p, a, st) : launch_mid_i<f16_t, f16_t, false>(p, a, st);
  fdbm_set_error("fdbm_conv_igemm(mid): 16-bit tensors only");
  return 1;
}
