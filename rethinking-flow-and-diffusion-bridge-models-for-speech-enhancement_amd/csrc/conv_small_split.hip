// conv_small_split.hip - the whole-map / band convolution kernel of the small feature maps (conv_small.hip) for the
// split-precision parity mode (fdbm_conv_args.mma_mode 1): f32 tensors in and out, every product on the 16-bit matrix pipe as
// three f16 MFMAs over (hi, lo) operand pairs (conv_common.h: hi = half(16 x), lo = half(16 x - hi); weights pre-split by the
// host with a power-of-two scale; hi.hi + hi.lo + lo.hi; the f32 sums are multiplied by acc_scale = 1 / (16 s_w)).
//
// Same launch body as conv_small.hip - every global load requested up front from a host-reduced argument block, the GroupNorm
// statistics summed by the workgroup itself (whole-map form) or read per thread from the producers' unit sums (band form),
// only the rows the workgroup's pixels touch transformed, one pass of MFMAs per wave (wave w = tap w; tap 8 and the 1-tap
// segments dealt out k-step by k-step), partial sums through LDS - with the f32 geometry:
//   * a 16-byte item is 4 channels = one statistics unit; a 32-channel k-step is a 128-byte LDS row [32 halves hi | 32 halves lo]
//     (the operand layout of conv_tap.hip's split mode), a staged pixel is KS9 such rows + 16 bytes;
//   * weights: fragment-major split pack [k-step][CoutPad / 16][8 chunks][16 rows][16 B], chunks 0-3 = hi, 4-7 = lo;
//   * a band form of the 32 x 32 level exists (TWO column tiles = one image row per workgroup: the staged rows are twice the
//     bytes) but loses to the wave-per-tap kernel there and is not selected (split_plan).
// In the parity mode these launches were the wave-per-tap kernel's (98 per batch-1 forward at 23 us: profiles/r03/
// kernel_stats_b1_f32s.csv); results agree with it to fp32 rounding of the sums (tests/test_hip_ops.py, policy 43 vs 11).
#include "conv_common.h"

#define SS_NTHR 512

namespace {

struct SplitArgs {
  const unsigned char* w;
  int64_t wb9[2], wtap9[2];           // BYTES: staged segment s: its tap 0 / k-step 0 / n-tile 0 block; stride between taps
  int64_t wb1[2];                     // raw 1-tap segment s: its k-step 0 / n-tile 0 block
  int64_t wstep;                      // bytes between consecutive k-steps (= CoutPad / 16 blocks of 2 KiB)
  const float* src9[2];               // staged segments, + coff
  const float* src1[2];               // raw 1-tap segments, + coff
  int sC9[2], sC1[2];
  int c9_0, c1_0, c9, c1;
  int taps9;                          // 9, or 1: 1-tap segments that carry the GroupNorm (centre tap only)
  int ks1, pooled;
  int H, W, HW;
  int a9_bytes, a1_bytes;
  int band, rpw;
  const double* useg[2];
  int unsp[2], ucnt[2];
  double inv_count_d;
  const float* gamma;
  const float* beta;
  int cpg, silu;
  float eps, inv_count;
  const float* bias;
  const float* tbias;
  int tb_stride;
  const float* res;
  const float* res_lo;
  const float* comb_pyr;
  const float* comb_w;
  const float* comb_b;
  float scale, acc_scale;
  int Cout;
};

static bool split_plan(const ConvParams& p, SplitArgs* out, int* lds_bytes, int* mt_out, int* nraw_out) {
  SplitArgs a;
  memset(&a, 0, sizeof(a));
  const bool gn = p.gn_sums != nullptr;
  int n9 = 0, nstaged = 0;
  bool seen1 = false;
  for (int i = 0; i < p.nseg; ++i) {
    if (p.seg[i].cin % 64 != 0 || p.seg[i].coff % 4 != 0 || p.seg[i].C % 4 != 0) return false;
    if (p.seg[i].taps == 9) { if (seen1) return false; ++n9; } else { seen1 = true; }
  }
  const bool gn1 = gn && n9 == 0;
  a.taps9 = gn1 ? 1 : 9;
  int kb = 0;
  const int64_t blk = 2048, ntq = p.CoutPad / 16;
  for (int i = 0; i < p.nseg; ++i) {
    const fdbm_conv_seg& sg = p.seg[i];
    const int nch = sg.cin / 32;
    const bool flagged = p.seg_gn[i] >= 0;
    const bool staged = sg.taps == 9 || (gn1 && flagged);
    if (gn && sg.taps == 9 && !flagged) return false;
    if (gn && !gn1 && sg.taps == 1 && flagged) return false;
    if (staged) {
      if (nstaged >= 2) return false;
      a.src9[nstaged] = reinterpret_cast<const float*>(sg.src) + sg.coff;
      a.sC9[nstaged] = sg.C;
      a.wb9[nstaged] = (int64_t)kb * ntq * blk;
      a.wtap9[nstaged] = (int64_t)nch * ntq * blk;
      if (nstaged == 0) a.c9_0 = sg.cin;
      a.c9 += sg.cin;
      ++nstaged;
    } else {
      const int r = (a.c1 == 0) ? 0 : 1;
      if (r == 1 && a.src1[1]) return false;
      a.src1[r] = reinterpret_cast<const float*>(sg.src) + sg.coff;
      a.sC1[r] = sg.C;
      a.wb1[r] = (int64_t)kb * ntq * blk;
      if (r == 0) a.c1_0 = sg.cin;
      a.c1 += sg.cin;
    }
    kb += sg.taps * nch;
  }
  if (a.c1 > 512) return false;
  if (a.c9 != 0 && a.c9 != 256 && a.c9 != 512) return false;
  if (a.c9 == 0 && a.c1 == 0) return false;
  a.wstep = ntq * blk;
  a.H = p.H; a.W = p.W; a.HW = p.H * p.W;
  if (a.HW % 16 != 0 || p.W < 2) return false;
  const int ipp = a.c9 / 4;                                           // staged 16-byte items per pixel
  const int rsb = (a.c9 / 32) * 128 + 16;                             // bytes of a staged pixel
  const bool whole = a.HW <= 64 && (!a.c9 || a.HW * ipp <= 8 * SS_NTHR) && (a.c9 ? (p.H + 2) * (p.W + 2) * rsb : 0) <= 110 * 1024;
  int mt = 1, items = a.HW * ipp;
  if (!whole) {
    if (gn && !p.gn_unit) return false;
    if (a.HW > 1024) return false;
    mt = a.HW > 256 ? 2 : 1;
    // (the two-tile form of the 32 x 32 level is built but NOT selected: 16 staged items per thread, 512 workgroups - 29.7 us per
    //  launch, and the mode's RTF is 34.1-34.3 with it against 34.7-35.0 with that level on the wave-per-tap kernel; FDBM_SPLIT_MT2=1
    //  selects it)
    static const char* mt2 = getenv("FDBM_SPLIT_MT2");
    if (mt == 2 && !(mt2 && mt2[0] == '1')) return false;
    const int pg = 16 * mt;
    if (!(pg % p.W == 0 || p.W % pg == 0) || a.HW % pg != 0) return false;
    a.band = 1;
    a.rpw = pg >= p.W ? pg / p.W : 1;
    items = (a.rpw + 2) * p.W * ipp;
    if (a.c9 && items > 16 * SS_NTHR) return false;
    if (mt == 2 && a.c9 != 256) return false;                         // (the two-tile form is instantiated for 256 staged channels)
  }
  if ((int64_t)((p.Cout + 15) / 16) * (a.HW / (16 * mt)) * p.B > 512) return false;     // (a latency kernel: conv_small.hip)
  *mt_out = mt;
  const int need = a.c9 ? (items + SS_NTHR - 1) / SS_NTHR : 1;
  *nraw_out = need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 16;
  a.ks1 = a.c1 / 32;
  a.pooled = (a.c9 ? a.c9 / 32 : 0) + a.ks1;
  if (a.pooled > 32) return false;
  if (gn) {
    if (p.gn_C != a.c9 || a.c9 == 0) return false;
    a.cpg = p.gn_C / p.gn_G;
    if ((a.cpg != 4 && a.cpg != 8 && a.cpg != 16) || p.gn_G > 128) return false;       // groups of 1 | 2 | 4 items
    a.gamma = p.gn_gamma; a.beta = p.gn_beta; a.silu = p.gn_silu; a.eps = p.gn_eps;
    a.inv_count = 1.0f / (float)(a.HW * a.cpg);
    a.inv_count_d = p.gn_inv_count;
    if (a.band) {
      int ns = 0;
      for (int i = 0; i < p.nseg; ++i) {
        if (p.seg_gn[i] < 0) continue;
        if (ns >= 2) return false;
        if (p.gn_uoff[i] != (ns == 0 ? 0 : a.c9_0 / 4) || p.gn_ucnt[i] != p.seg[i].cin / 4) return false;
        if (p.seg[i].cin % a.cpg != 0) return false;
        a.useg[ns] = p.gn_useg[i]; a.unsp[ns] = p.gn_unsp[i]; a.ucnt[ns] = p.gn_ucnt[i];
        ++ns;
      }
    } else if (a.c9_0 % a.cpg != 0) {
      return false;                                                    // (groups inside one segment)
    }
  }
  a.res_lo = p.res_lo; a.comb_pyr = p.comb_pyr; a.comb_w = p.comb_w; a.comb_b = p.comb_b;
  a.a9_bytes = a.c9 ? (a.band ? a.rpw + 2 : p.H + 2) * (p.W + 2) * rsb : 0;
  a.a1_bytes = a.c1 ? 16 * mt * (a.ks1 * 128 + 16) : 0;
  *lds_bytes = a.a9_bytes + a.a1_bytes + 8 * mt * 64 * 16 /*partials*/ + SS_NTHR * 8 /*stat partials*/ + 64 * 8 * 2 /*out stats*/ + 64;
  if (*lds_bytes > 150 * 1024) return false;
  a.w = reinterpret_cast<const unsigned char*>(p.w);
  a.bias = p.bias; a.tbias = p.tbias; a.tb_stride = p.tbias_stride; a.res = reinterpret_cast<const float*>(p.res);
  a.scale = p.scale; a.acc_scale = p.acc_scale; a.Cout = p.Cout;
  *out = a;
  return true;
}

__device__ __forceinline__ uint4 ldw(const unsigned char* p) { return *reinterpret_cast<const uint4*>(p); }

// GNS: 0 no GroupNorm, 1 statistics by the workgroup (whole-map form), 2 from the producers' unit sums (band form).
// KS9: staged 32-channel k-steps (8 | 16; 0: none).  NRAW: staged items per thread.  MT: 16-pixel column tiles per workgroup.
template <int GNS, int KS9, int NRAW, int MT>
__global__ void __launch_bounds__(SS_NTHR) conv_small_split_kernel(const SplitArgs a, float* __restrict__ out, double* __restrict__ stat_out,
                                                                   int stat_G, int stat_nsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool GNP = GNS != 0;
  constexpr int RSB = KS9 * 128 + 16;                                // bytes of a staged pixel
  constexpr int IPP = KS9 ? KS9 * 8 : 128;                           // staged items (4 channels) per pixel
  constexpr int PPASS = SS_NTHR / IPP;
  constexpr int PG = 16 * MT;
  const int H = a.H, W = a.W, HW = a.HW, PW = W + 2;
  const int RS1B = a.ks1 * 128 + 16;
  unsigned char* s_a9 = smem;                                       // [staged rows (+ 2: whole-map form)][W + 2][RSB], zero border
  unsigned char* s_a1 = smem + a.a9_bytes;                          // [PG][RS1B] raw 1-tap pixels
  f32x4* s_red = reinterpret_cast<f32x4*>(smem + a.a9_bytes + a.a1_bytes);       // [8 waves][MT][64 lanes]
  float2* s_part = reinterpret_cast<float2*>(s_red + 8 * MT * 64);  // one per thread: (sum, sumsq) of its items' 4 channels
  double* s_ostat = reinterpret_cast<double*>(s_part + SS_NTHR);    // [64][2] output unit statistics

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fk = lane >> 4;
  const int ntile = blockIdx.x, pgroup = blockIdx.y, b = blockIdx.z;
  const int64_t img = (int64_t)b * HW;
  const int r0 = a.band ? (pgroup * PG) / W - 1 : 0;
  const int nst = a.band ? a.rpw + 2 : H;
  const int roff = a.band ? 0 : 1;

  // ---- 1. every global load of the kernel, straight-line -----------------------------------------------------------------------
  const unsigned char* wl = a.w + (int64_t)ntile * 2048 + (fk * 16 + frow) * 16;      // lane's hi fragment of a block; lo: + 1 024
  // this wave's tap: k-steps 0 .. 7 now; k-steps 8 .. 15 (512 staged channels) and the pooled k-steps are requested in step 4,
  // behind the transform - in the registers the staged items have just left (all at once they do not fit beside them)
  constexpr int KH = KS9 > 8 ? 8 : KS9;
  [[maybe_unused]] uint4 wh[KH ? KH : 1], wlo[KH ? KH : 1], w2h[8], w2l[8];
  auto own_ptr = [&](const int k) __attribute__((always_inline)) {
    const int ch = k * 32;
    const bool s1 = ch >= a.c9_0;
    return wl + (s1 ? a.wb9[1] : a.wb9[0]) + (int64_t)(a.taps9 == 9 ? wave : 0) * (s1 ? a.wtap9[1] : a.wtap9[0]) +
           (int64_t)((ch - (s1 ? a.c9_0 : 0)) >> 5) * a.wstep;
  };
  if constexpr (KS9 > 0) {
#pragma unroll
    for (int k = 0; k < KH; ++k) { const unsigned char* q = own_ptr(k); wh[k] = ldw(q); wlo[k] = ldw(q + 1024); }
  }
  // the staged rows: thread tid keeps item column tid % IPP (4 channels) of staged pixels tid / IPP + PPASS j
  const int icol = tid % IPP;
  const bool seg1 = icol * 4 >= a.c9_0;
  [[maybe_unused]] f32x4 raw[NRAW];
  if constexpr (KS9 > 0) {
    const int sC = seg1 ? a.sC9[1] : a.sC9[0];
    const float* src = (seg1 ? a.src9[1] : a.src9[0]) + img * sC + (icol * 4 - (seg1 ? a.c9_0 : 0));
#pragma unroll
    for (int j = 0; j < NRAW; ++j) {
      const int sp = tid / IPP + PPASS * j;
      const int row = min(max(r0 + sp / W, 0), H - 1), col = sp % W;
      raw[j] = *reinterpret_cast<const f32x4*>(src + (int64_t)(row * W + col) * sC);
    }
  }
  // raw 1-tap sources: this workgroup's PG pixels (<= 128 items per pixel: 4 MT per thread; clamped)
  f32x4 raw1[4 * MT];
  {
    const int ipp1 = max(a.c1 / 4, 1);
#pragma unroll
    for (int j = 0; j < 4 * MT; ++j) {
      const int q = min(tid + SS_NTHR * j, PG * ipp1 - 1);
      const int px = pgroup * PG + q / ipp1, ch = (q % ipp1) * 4;
      const bool s1 = ch >= a.c1_0;
      const float* base1 = a.c1 ? (s1 ? a.src1[1] : a.src1[0]) : g_conv_zero;
      raw1[j] = *reinterpret_cast<const f32x4*>(base1 + (a.c1 ? (img + px) * (s1 ? a.sC1[1] : a.sC1[0]) + (ch - (s1 ? a.c1_0 : 0)) : 0));
    }
  }
  [[maybe_unused]] f32x4 g4, b4;
  if constexpr (GNP) {
    g4 = *reinterpret_cast<const f32x4*>(a.gamma + icol * 4);
    b4 = *reinterpret_cast<const f32x4*>(a.beta + icol * 4);
  }
  // band form: the unit sums of this thread's group (1 | 2 | 4 units), every partial row
  [[maybe_unused]] double us0 = 0.0, us1 = 0.0;
  if constexpr (GNS == 2) {
    const double* ub = (seg1 ? a.useg[1] : a.useg[0]);
    const int nsp = seg1 ? a.unsp[1] : a.unsp[0], ucnt = seg1 ? a.ucnt[1] : a.ucnt[0];
    const int u_item = icol - (seg1 ? a.c9_0 >> 2 : 0);              // this item's unit inside its segment
    const int upg = a.cpg >> 2;
    const int u0 = (u_item / upg) * upg;
    double q0[4], q1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      q0[q] = q1[q] = 0.0;
      const int u = u0 + min(q, upg - 1);
      for (int sp = 0; sp < nsp; ++sp) {
        const double* r = ub + (((int64_t)b * nsp + sp) * ucnt + u) * 2;
        q0[q] += r[0]; q1[q] += r[1];
      }
    }
    us0 = q0[0]; us1 = q1[0];
    if (upg >= 2) { us0 += q0[1]; us1 += q1[1]; }
    if (upg == 4) { us0 += q0[2] + q0[3]; us1 += q1[2] + q1[3]; }
  }
  // the epilogue's operands (wave t < MT finishes column tile t)
  const int etile = wave < MT ? wave : 0;
  const int pme_e = pgroup * PG + etile * 16 + frow;
  const int n_out = ntile * 16 + fk * 4;
  const bool live = n_out < a.Cout;
  const int n_ld = live ? n_out : 0;
  const f32x4 e_bias = *reinterpret_cast<const f32x4*>(a.bias ? a.bias + n_ld : g_conv_zero);
  const f32x4 e_tb = *reinterpret_cast<const f32x4*>(a.tbias ? a.tbias + (int64_t)b * a.tb_stride + n_ld : g_conv_zero);
  const f32x4 e_res = *reinterpret_cast<const f32x4*>(a.res ? a.res + (img + pme_e) * a.Cout + n_ld : g_conv_zero);

  // ---- 2. zero the padded rows and the output statistics (LDS only) ------------------------------------------------------------
  for (int i = tid; i < a.a9_bytes / 16; i += SS_NTHR) reinterpret_cast<uint4*>(s_a9)[i] = uint4{0u, 0u, 0u, 0u};
  if (stat_out)
    for (int i = tid; i < 128; i += SS_NTHR) s_ostat[i] = 0.0;

  // ---- 3. GroupNorm mean / rstd of this thread's 4 channels ------------------------------------------------------------------------
  [[maybe_unused]] float mean = 0.f, rstd = 1.f;
  if constexpr (GNS == 1 && KS9 > 0) {
    float p0 = 0.f, p1 = 0.f;
#pragma unroll
    for (int j = 0; j < NRAW; ++j) {
      if (tid / IPP + PPASS * j < HW) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float x = raw[j][k]; p0 += x; p1 += x * x; }
      }
    }
    s_part[tid] = float2{p0, p1};
    __syncthreads();
    // the group's ipg items x PPASS partial sums, fixed order
    const int ipg = a.cpg >> 2, i0 = (icol / ipg) * ipg;
    float t0 = 0.f, t1 = 0.f;
    for (int i = 0; i < ipg; ++i) {
      float2 v[PPASS];
#pragma unroll
      for (int q = 0; q < PPASS; ++q) v[q] = s_part[q * IPP + i0 + i];
#pragma unroll
      for (int q = 0; q < PPASS; ++q) { t0 += v[q].x; t1 += v[q].y; }
    }
    mean = t0 * a.inv_count;
    rstd = __builtin_amdgcn_rsqf(fmaxf(t1 * a.inv_count - mean * mean, 0.f) + a.eps);
  } else {
    if constexpr (GNS == 2) {
      const double m0 = us0 * a.inv_count_d;
      double v0 = us1 * a.inv_count_d - m0 * m0;
      v0 = v0 < 0.0 ? 0.0 : v0;
      const double x0 = v0 + (double)a.eps;
      double q0 = __builtin_amdgcn_rsq(x0);
      q0 = q0 * (1.5 - 0.5 * x0 * q0 * q0);
      mean = (float)m0; rstd = (float)q0;
    }
    __syncthreads();                                               // (the zeroing above precedes the interior writes)
  }

  // ---- 4. normalise + SiLU in registers, split into (hi, lo) halves; only the rows this workgroup's pixels touch go to LDS -----
  if constexpr (KS9 > 0) {
    const int r_lo = (pgroup * PG) / W - 1, r_hi = (pgroup * PG + PG - 1) / W + 1;
    const int koff = (icol >> 3) * 128 + (icol & 7) * 8;            // this item's place inside a staged pixel: k-step row, 8 bytes of hi
#pragma unroll
    for (int j = 0; j < NRAW; ++j) {
      const int sp = tid / IPP + PPASS * j;
      const int br = sp / W, col = sp - br * W, row = r0 + br;
      if (br < nst && row >= 0 && row < H && row >= r_lo && row <= r_hi) {
        f32x4 y = raw[j];
        if constexpr (GNP) {
#pragma unroll
          for (int k = 0; k < 4; ++k) y[k] = ((y[k] - mean) * rstd) * g4[k] + b4[k];
          if (a.silu == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = silu_f(y[k]);
          } else if (a.silu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = silu_precise(y[k]);
          }
        }
        uint2 hi, lo;
        split_f16x4(y, hi, lo);
        unsigned char* d = s_a9 + ((br + roff) * PW + col + 1) * RSB + koff;
        *reinterpret_cast<uint2*>(d) = hi;
        *reinterpret_cast<uint2*>(d + 64) = lo;
      }
    }
  }
  if (a.c1) {
    const int ipp1 = a.c1 / 4;
#pragma unroll
    for (int j = 0; j < 4 * MT; ++j) {
      const int q = tid + SS_NTHR * j;
      if (q < PG * ipp1) {
        const int px = q / ipp1, it = q - px * ipp1;
        uint2 hi, lo;
        split_f16x4(raw1[j], hi, lo);
        unsigned char* d = s_a1 + px * RS1B + (it >> 3) * 128 + (it & 7) * 8;
        *reinterpret_cast<uint2*>(d) = hi;
        *reinterpret_cast<uint2*>(d + 64) = lo;
      }
    }
  }
  if constexpr (KS9 > 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { const unsigned char* q = own_ptr(8 + k); w2h[k] = ldw(q); w2l[k] = ldw(q + 1024); }
  }
  uint4 ph[4], pl[4];                                                // pooled k-steps j = wave, wave + 8, ... (clamped)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = min(wave + 8 * k, a.pooled - 1);
    int64_t off;
    if (KS9 > 0 && j < KS9) {
      const int ch = j * 32;
      const bool s1 = ch >= a.c9_0;
      off = (s1 ? a.wb9[1] : a.wb9[0]) + (a.taps9 == 9 ? 8 : 0) * (s1 ? a.wtap9[1] : a.wtap9[0]) + (int64_t)((ch - (s1 ? a.c9_0 : 0)) >> 5) * a.wstep;
    } else {
      const int ch = (j - KS9) * 32;
      const bool s1 = ch >= a.c1_0;
      off = (s1 ? a.wb1[1] : a.wb1[0]) + (int64_t)((ch - (s1 ? a.c1_0 : 0)) >> 5) * a.wstep;
    }
    ph[k] = ldw(wl + off); pl[k] = ldw(wl + off + 1024);
  }
  __syncthreads();

  // ---- 5. MFMAs: 16 output channels x MT tiles of 16 pixels; a product = w_hi.a_hi + w_hi.a_lo + w_lo.a_hi -------------------------
  f32x4 acc[MT];
  int arow_base[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pme = pgroup * PG + t * 16 + frow;
    const int py = pme / W, pxc = pme - py * W;
    arow_base[t] = (py - 1 - r0 + roff) * PW + pxc;
  }
  auto mma3 = [&](const uint4& whi, const uint4& wlo_, const unsigned char* ap, f32x4& c) __attribute__((always_inline)) {
    const uint4 ah = *reinterpret_cast<const uint4*>(ap), al = *reinterpret_cast<const uint4*>(ap + 64);
    Mfma<f16_t>::run(whi, ah, c);
    Mfma<f16_t>::run(whi, al, c);
    Mfma<f16_t>::run(wlo_, ah, c);
  };
  if constexpr (KS9 > 0) {
    if (a.taps9 == 9) {
      const int dy = wave / 3, dx = wave - dy * 3;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const unsigned char* arow = s_a9 + (arow_base[t] + dy * PW + dx) * RSB + fk * 16;
#pragma unroll
        for (int k = 0; k < KH; ++k) mma3(wh[k], wlo[k], arow + k * 128, acc[t]);
        if constexpr (KS9 > 8) {
#pragma unroll
          for (int k = 0; k < 8; ++k) mma3(w2h[k], w2l[k], arow + (8 + k) * 128, acc[t]);
        }
      }
    }
  }
  {
    const int sh = a.taps9 == 9 ? 2 : 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = wave + 8 * k;
      if (j < a.pooled) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const unsigned char* ap = (KS9 > 0 && j < KS9) ? s_a9 + (arow_base[t] + sh * PW + sh) * RSB + j * 128 + fk * 16
                                                         : s_a1 + (t * 16 + frow) * RS1B + (j - KS9) * 128 + fk * 16;
          mma3(ph[k], pl[k], ap, acc[t]);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < MT; ++t) s_red[(wave * MT + t) * 64 + lane] = acc[t];
  __syncthreads();

  // ---- 6. sum of the 8 partial tiles + epilogue: wave t finishes column tile t ----------------------------------------------------------
  if (wave < MT) {
    f32x4 pv[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) pv[w] = s_red[(w * MT + wave) * 64 + lane];
    f32x4 s = pv[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) s += pv[w];
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = ((s[r] * a.acc_scale + e_bias[r]) + e_tb[r]) + e_res[r];
    if (a.res_lo) {
      const int y = pme_e / W, x = pme_e - y * W, H2 = H >> 1, W2 = W >> 1, iy = y >> 1, ix = x >> 1;
      const int ys0 = (y & 1) ? iy : iy - 1, xs0 = (x & 1) ? ix : ix - 1;
      const float wy0 = (y & 1) ? 0.75f : 0.25f, wx0 = (x & 1) ? 0.75f : 0.25f;
      f32x4 q[4];
      float wq[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ay = ys0 + (t >> 1), ax = xs0 + (t & 1);
        const bool in = ay >= 0 && ay < H2 && ax >= 0 && ax < W2;
        wq[t] = in ? ((t >> 1) ? 1.0f - wy0 : wy0) * ((t & 1) ? 1.0f - wx0 : wx0) : 0.f;
        q[t] = *reinterpret_cast<const f32x4*>(in ? a.res_lo + (((int64_t)b * H2 + ay) * W2 + ax) * a.Cout + n_ld : g_conv_zero);
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
    if (a.comb_pyr) {
      const f32x4 cq = *reinterpret_cast<const f32x4*>(a.comb_pyr + (img + pme_e) * 4);
      const f32x4 cb = *reinterpret_cast<const f32x4*>(a.comb_b + n_ld);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 cw = *reinterpret_cast<const f32x4*>(a.comb_w + (int64_t)(n_ld + r) * 4);
        v[r] += cb[r] + (((cw[0] * cq[0] + cw[1] * cq[1]) + cw[2] * cq[2]) + cw[3] * cq[3]);
      }
    }
    if (live) OutVec<float>::store(out + (img + pme_e) * a.Cout + n_out, v);
    if (stat_out) {
      const int scpg = a.Cout / stat_G;
      const float q1 = live ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
      const float q2 = live ? (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]) : 0.f;
      const float r1 = row16_sum(q1), r2 = row16_sum(q2);
      if (frow == 0 && live) {
        atomicAdd(&s_ostat[((n_out - ntile * 16) / scpg) * 2], (double)r1);
        atomicAdd(&s_ostat[((n_out - ntile * 16) / scpg) * 2 + 1], (double)r2);
      }
    }
  }
  if (stat_out) {
    if constexpr (MT > 1) __syncthreads();
    else __builtin_amdgcn_s_waitcnt(0xc07f);                        // lgkmcnt(0): the LDS atomics of wave 0 have landed
    if (wave == 0) {
      const int scpg = a.Cout / stat_G;
      const int g_0 = (ntile * 16) / scpg;
      const int ng = min(stat_G - g_0, (16 + scpg - 1) / scpg);
      if (lane < ng * 2) {
        const int k = lane & 1, g = g_0 + (lane >> 1);
        atomicAdd(stat_out + (((int64_t)b * stat_nsplit + pgroup % stat_nsplit) * stat_G + g) * 2 + k, s_ostat[(g - g_0) * 2 + k]);
      }
    }
  }
}

template <int GNS, int KS9, int NRAW, int MT>
static int launch_split_i(const ConvParams& p, const SplitArgs& a, int lds, hipStream_t st) {
  auto kern = &conv_small_split_kernel<GNS, KS9, NRAW, MT>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024);
    if (e != hipSuccess) {
      fdbm_set_error("fdbm_conv_igemm(small, split): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr = true;
  }
  dim3 grid((unsigned)((p.Cout + 15) / 16), (unsigned)(p.H * p.W / (16 * MT)), (unsigned)p.B);
  kern<<<grid, SS_NTHR, lds, st>>>(a, reinterpret_cast<float*>(p.out), p.stat_out, p.stat_G, p.stat_nsplit);
  FDBM_LAUNCH_CHECK("fdbm_conv_igemm(small, split)");
  return 0;
}

template <int GNS, int KS9, int MT>
static int launch_split_n(const ConvParams& p, const SplitArgs& a, int lds, int nraw, hipStream_t st) {
  if constexpr (KS9 == 0) {
    return launch_split_i<0, 0, 2, MT>(p, a, lds, st);
  } else {
    if constexpr (MT == 1) {
      if (nraw <= 2) return launch_split_i<GNS, KS9, 2, 1>(p, a, lds, st);
      if (nraw <= 4) return launch_split_i<GNS, KS9, 4, 1>(p, a, lds, st);
      if (nraw <= 8) return launch_split_i<GNS, KS9, 8, 1>(p, a, lds, st);
    }
    return launch_split_i<GNS, KS9, 16, MT>(p, a, lds, st);
  }
}

}  // namespace

// can this convolution of the split-precision mode run on the whole-map / band kernel?  (f32 tensors; p filled by
// fdbm_conv_igemm, p.w = fragment-major split weights)
bool fdbm_conv_small_split_ok(const ConvParams& p) {
  SplitArgs a;
  int lds, mt, nraw;
  return p.mma_split && split_plan(p, &a, &lds, &mt, &nraw);
}

int fdbm_launch_conv_small_split(const ConvParams& p, hipStream_t st) {
  SplitArgs a;
  int lds, mt, nraw;
  if (!p.mma_split || !split_plan(p, &a, &lds, &mt, &nraw)) {
    fdbm_set_error("fdbm_conv_igemm(small, split): shape not supported");
    return 1;
  }
  const bool gnp = p.gn_sums != nullptr;
  if (a.c9 == 0) return launch_split_n<0, 0, 1>(p, a, lds, nraw, st);
  if (mt == 2) return gnp ? launch_split_n<2, 8, 2>(p, a, lds, nraw, st) : launch_split_n<0, 8, 2>(p, a, lds, nraw, st);
  if (a.band) {
    if (a.c9 == 256) return gnp ? launch_split_n<2, 8, 1>(p, a, lds, nraw, st) : launch_split_n<0, 8, 1>(p, a, lds, nraw, st);
    return gnp ? launch_split_n<2, 16, 1>(p, a, lds, nraw, st) : launch_split_n<0, 16, 1>(p, a, lds, nraw, st);
  }
  if (a.c9 == 256) return gnp ? launch_split_n<1, 8, 1>(p, a, lds, nraw, st) : launch_split_n<0, 8, 1>(p, a, lds, nraw, st);
  return gnp ? launch_split_n<1, 16, 1>(p, a, lds, nraw, st) : launch_split_n<0, 16, 1>(p, a, lds, nraw, st);
}
