// conv_small.hip - the convolution of the SMALLEST feature maps (the 4 x 4 and 8 x 8 levels of a 4 s clip): whole map per
// workgroup, GroupNorm statistics computed by the consumer.
//
// Why it exists (profiles/r03/persistent_chain_prototype.txt, tools/persist_proto/chain.hip): a conv3x3 256 -> 256 with a
// GroupNorm + SiLU prologue on a 4 x 4 map costs 4.2 us as a launch inside a HIP graph when the kernel body is minimal
// (5.3 us at 8 x 8), and a persistent multi-layer kernel takes only 0.4-0.6 us off that - whereas the general wave-per-tap
// kernel (conv_tap.hip) spent 11.5-16 us per launch on these maps: a prologue table built from the producers' statistics
// rows (a dependent global round trip + two barriers + fp64 math), four register sets of look-ahead with their cursor,
// statistics atomics behind every epilogue.  The maps are tiny, so the lean form is:
//   workgroup = 8 waves = (16 output channels) x (16 pixels of one image); grid = Cout/16 x H*W/16 x B
//   * EVERY global load of the kernel is requested up front, in straight-line code, from a compact argument block the host
//     has already reduced to offsets and strides (the first version of this kernel spent 2.4 us of its 7.7 on integer
//     arithmetic over the general parameter block before its first load: tools/small_timeline.py): the wave's weight
//     fragments (wave w owns tap w of the 9-tap segments; tap 8 and the 1-tap shortcut / NIN segments are dealt out
//     k-step by k-step over the 8 waves), the whole map, the GroupNorm parameters of the thread's channels, and the
//     epilogue's operands (bias, time bias, residual);
//   * the WHOLE map of the 9-tap sources is staged once (H*W x C items of 16 bytes); its GroupNorm statistics are summed by
//     the workgroup itself in a fixed order (thread -> LDS partials -> group totals: no statistics buffers read, no
//     atomics, one barrier); only the rows this workgroup's pixels touch are normalised + SiLU'd (registers) and written
//     to LDS, inside a zero border; 1-tap sources: the workgroup's own 16 pixels, raw - or, when they carry the
//     GroupNorm (the attention block's NIN), staged like a 9-tap source and multiplied at the centre tap only;
//   * one pass of MFMAs per wave straight from LDS (row stride C + 8 elements: conflict-free b128), partial accumulators
//     summed through LDS, epilogue by wave 0 (the arithmetic of conv_epilogue4 on the prefetched operands) and the
//     output's unit statistics for the consumers that still read them (resampling, larger maps).
// BAND form (the 16 x 16 and 32 x 32 levels, where the padded map no longer fits): the workgroup stages only the rows its
// 16 (or 64: four MFMA column tiles) pixels touch - its rows plus one above and below - and takes the GroupNorm statistics
// from the producers' unit sums, but read PER THREAD (the thread's own group: 2-4 units x 1-2 partial rows, a handful of
// independent loads among the others - no table, no barrier, no fp64 work on one thread's critical path for all).
// Results: the same convolution as every other kernel of fdbm_conv_igemm; GroupNorm mean / variance from fp32 sums over
// H*W*cpg <= 2 048 values in a fixed order (the producers' fp64 unit sums are not read).
// Roofline: latency (these launches hold < 0.1 us of MFMA work per CU); selected by fdbm_conv_igemm for 16-bit tensors when
// the padded map fits the LDS (fdbm_conv_small_ok).
#include "conv_common.h"

#define SM_NTHR 512

// Diagnostic build only (-DFDBM_STAMPS, tools/small_timeline.py): workgroup (0, 0, 0) writes realtime-clock stamps (100 MHz)
// of its phases into the scratch the caller passed as acc_ws.  The product library has none of it.
#ifdef FDBM_STAMPS
#define SSTAMP(i)                                                                                         \
  do {                                                                                                    \
    if (a.stamps && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)             \
      a.stamps[i] = __builtin_amdgcn_s_memrealtime();                                                     \
  } while (0)
#else
#define SSTAMP(i)
#endif

namespace {

// everything the kernel needs, already reduced by the host (element counts are in elements of T)
struct SmallArgs {
  // weights (fragment-major: a (k-step of 64 channels, 16-channel n-tile) block is 1 024 elements)
  const void* w;
  int64_t wb9[2], wtap9[2];           // staged segment s: offset of its tap 0 / chunk 0 / n-tile 0 block; stride between taps
  int64_t wb1[2];                     // raw 1-tap segment s: offset of its chunk 0 / n-tile 0 block
  int64_t wchunk;                     // stride between consecutive 64-channel chunks (= CoutPad / 16 blocks)
  // sources
  const void* src9[2];                // whole-map ("staged") segments, + coff
  const void* src1[2];                // raw 1-tap segments, + coff
  int sC9[2], sC1[2];                 // their pixel strides
  int c9_0, c1_0;                     // channels of the first staged / first raw segment (the second starts there)
  int c9, c1;                         // staged / raw channels in all
  int taps9;                          // 9: the staged segments are 3x3 taps; 1: 1-tap segments that carry the GroupNorm (centre tap only)
  int ks1;                            // 32-channel k-steps of the raw 1-tap segments
  int pooled;                         // k-steps dealt out over the waves
  int H, W, HW;
  int a9_bytes, a1_bytes;
  // band form: staged image rows per workgroup = rpw + 2 (whole-map form: all H rows, band == 0)
  int band, rpw;
  // GroupNorm statistics from the producers' unit sums (band form): staged segment s -> doubles [B][unsp][ucnt][2]
  const double* useg[2];
  int unsp[2], ucnt[2];
  double inv_count_d;
  // GroupNorm over the staged channels
  const float* gamma;
  const float* beta;
  int cpg, silu;
  float eps, inv_count;
  // epilogue
  const float* bias;
  const float* tbias;
  int tb_stride;
  const void* res;
  const float* res_lo;                // residual at half resolution (f32 [B][H/2][W/2][Cout]), upsampled 2x in the epilogue
  const float* comb_pyr;              // Combine('sum'): out += comb_b[n] + comb_w[n][0..3] . comb_pyr[m][0..3]
  const float* comb_w;
  const float* comb_b;
  float scale;
  int Cout;
  unsigned long long* stamps;
};

static bool small_plan(const ConvParams& p, SmallArgs* out, int* lds_bytes, int* mt_out) {
  SmallArgs a;
  memset(&a, 0, sizeof(a));
  const bool gn = p.gn_sums != nullptr;
  // segment classes: "staged" = whole map in LDS (the 9-tap segments; or, when a GroupNorm covers them, the 1-tap ones),
  // "raw" = 1-tap segments without GroupNorm (this workgroup's 16 pixels)
  int n9 = 0, nstaged = 0;
  bool seen1 = false;
  for (int i = 0; i < p.nseg; ++i) {
    if (p.seg[i].cin % 64 != 0 || p.seg[i].coff % 8 != 0 || p.seg[i].C % 8 != 0) return false;
    if (p.seg[i].taps == 9) { if (seen1) return false; ++n9; } else { seen1 = true; }
  }
  const bool gn1 = gn && n9 == 0;                                   // the GroupNorm sits on 1-tap segments (NIN)
  a.taps9 = gn1 ? 1 : 9;
  int kb = 0;
  const int64_t blk = 1024, ntq = p.CoutPad / 16;
  for (int i = 0; i < p.nseg; ++i) {
    const fdbm_conv_seg& sg = p.seg[i];
    const int nch = sg.cin / 64;
    const bool flagged = p.seg_gn[i] >= 0;
    const bool staged = sg.taps == 9 || (gn1 && flagged);
    if (gn && sg.taps == 9 && !flagged) return false;               // (a 9-tap segment outside the GroupNorm: not in this network)
    if (gn && !gn1 && sg.taps == 1 && flagged) return false;
    if (staged) {
      if (nstaged >= 2) return false;
      a.src9[nstaged] = reinterpret_cast<const unsigned char*>(sg.src) + (int64_t)sg.coff * 2;
      a.sC9[nstaged] = sg.C;
      a.wb9[nstaged] = (int64_t)kb * ntq * blk;
      a.wtap9[nstaged] = (int64_t)nch * ntq * blk;
      if (nstaged == 0) a.c9_0 = sg.cin;
      a.c9 += sg.cin;
      ++nstaged;
    } else {
      const int r = (a.c1 == 0) ? 0 : 1;
      if (r == 1 && a.src1[1]) return false;
      a.src1[r] = reinterpret_cast<const unsigned char*>(sg.src) + (int64_t)sg.coff * 2;
      a.sC1[r] = sg.C;
      a.wb1[r] = (int64_t)kb * ntq * blk;
      if (r == 0) a.c1_0 = sg.cin;
      a.c1 += sg.cin;
    }
    kb += sg.taps * nch;
  }
  if (a.c1 > 512) return false;
  if (a.c9 != 0 && a.c9 != 256 && a.c9 != 512) return false;      // (the kernel is instantiated for 8 | 16 staged k-steps)
  if (a.c9 == 0 && a.c1 == 0) return false;
  a.wchunk = ntq * blk;
  a.H = p.H; a.W = p.W; a.HW = p.H * p.W;
  if (a.HW % 16 != 0 || p.W < 2) return false;
  // whole-map form when the padded map fits (LDS, 8 staged items per thread); else the band form
  const bool whole = a.HW <= 128 && (!a.c9 || a.HW * (a.c9 / 8) <= 8 * SM_NTHR) &&
                     (a.c9 ? (p.H + 2) * (p.W + 2) * (a.c9 + 8) * 2 : 0) <= 110 * 1024;
  int mt = 1;
  if (!whole) {
    if (gn && !p.gn_unit) return false;                            // (statistics of an explicit pass: the general kernels)
    if (a.HW > 1024) return false;
    mt = a.HW > 256 ? 4 : 1;
    static const char* bmax = getenv("FDBM_SMALL_BAND_MAXHW");       // experiments: largest map of the band form (default 1024)
    if (bmax && a.HW > atoi(bmax)) return false;
    const int pg = 16 * mt;
    if (!(pg % p.W == 0 || p.W % pg == 0) || a.HW % pg != 0) return false;
    a.band = 1;
    a.rpw = pg >= p.W ? pg / p.W : 1;
    if (a.c9 && (a.rpw + 2) * p.W * (a.c9 / 8) > 8 * SM_NTHR) return false;
    if (mt == 4 && a.c9 != 256) return false;                      // (the four-tile form is instantiated for 256 staged channels)
  }
  // a latency kernel: every workgroup re-reads its 16 output channels' weights and re-stages the rows it needs, which only pays
  // while the launch is a few hundred workgroups (batch 1: every level up to 32 x 32; batch 64: the 4 x 4 level) - beyond
  // that the wave-per-tap kernel's wider tiles win (batch 64: 3.4 ms per forward on this kernel against 0.9 ms)
  static const char* mxg = getenv("FDBM_SMALL_MAX_GRID");          // experiments
  if ((int64_t)((p.Cout + 15) / 16) * (a.HW / (16 * mt)) * p.B > (mxg ? atoi(mxg) : 512)) return false;
  *mt_out = mt;
  a.ks1 = a.c1 / 32;
  a.pooled = (a.c9 ? a.c9 / 32 : 0) + a.ks1;                       // tap 8 (or the centre tap) of the staged channels + the raw ones
  if (a.pooled > 32) return false;
  if (gn) {
    if (p.gn_C != a.c9 || a.c9 == 0) return false;
    a.cpg = p.gn_C / p.gn_G;
    if ((a.cpg % 8 != 0 && a.cpg != 4) || p.gn_G > 128) return false;   // groups of whole 16-byte items, or of half an item
    a.gamma = p.gn_gamma; a.beta = p.gn_beta; a.silu = p.gn_silu; a.eps = p.gn_eps;
    a.inv_count = 1.0f / (float)(a.HW * a.cpg);
    a.inv_count_d = p.gn_inv_count;
    if (a.band) {
      if (a.cpg != 4 && a.cpg != 8 && a.cpg != 16) return false;
      int ns = 0;
      for (int i = 0; i < p.nseg; ++i) {
        if (p.seg_gn[i] < 0) continue;
        if (ns >= 2) return false;
        if (p.gn_uoff[i] != (ns == 0 ? 0 : a.c9_0 / 4) || p.gn_ucnt[i] != p.seg[i].cin / 4) return false;
        if (p.seg[i].cin % a.cpg != 0) return false;                 // (groups inside one segment)
        a.useg[ns] = p.gn_useg[i]; a.unsp[ns] = p.gn_unsp[i]; a.ucnt[ns] = p.gn_ucnt[i];
        ++ns;
      }
    }
  }
  a.res_lo = p.res_lo; a.comb_pyr = p.comb_pyr; a.comb_w = p.comb_w; a.comb_b = p.comb_b;
  a.a9_bytes = a.c9 ? (a.band ? a.rpw + 2 : p.H + 2) * (p.W + 2) * (a.c9 + 8) * 2 : 0;
  a.a1_bytes = a.c1 ? 16 * mt * (a.c1 + 8) * 2 : 0;
  *lds_bytes = a.a9_bytes + a.a1_bytes + 8 * mt * 64 * 16 /*partials*/ + SM_NTHR * 16 /*stat partials*/ + 64 * 8 * 2 /*out stats*/ + 64;
  if (*lds_bytes > 150 * 1024) return false;
  a.w = p.w;
  a.bias = p.bias; a.tbias = p.tbias; a.tb_stride = p.tbias_stride; a.res = p.res; a.scale = p.scale; a.Cout = p.Cout;
  a.stamps = reinterpret_cast<unsigned long long*>(p.partial);
  *out = a;
  return true;
}

template <typename T>
__device__ __forceinline__ uint4 ld16(const T* p) { return *reinterpret_cast<const uint4*>(p); }

// GNS: 0 no GroupNorm, 1 statistics by the workgroup (whole-map form), 2 from the producers' unit sums (band form).
// KS9: staged channels / 32 (8 | 16; 0: no staged segment).  NRAW: staged 16-byte items per thread (1 | 2 | 4 | 8).
// MT: 16-pixel MFMA column tiles per workgroup (1; 4 in the band form of the 32 x 32 level).
template <typename T, typename TO, int GNS, int KS9, int NRAW, int MT>
__global__ void __launch_bounds__(SM_NTHR) conv_small_kernel(const SmallArgs a, TO* __restrict__ out, double* __restrict__ stat_out,
                                                             int stat_G, int stat_nsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool GNP = GNS != 0;
  constexpr bool BAND = GNS == 2 || MT > 1;                         // (a band without GroupNorm comes as <0, ..., MT> with a.band set)
  constexpr int C9 = KS9 * 32;
  constexpr int RS9 = C9 + 8;                                      // LDS row stride of the staged map (elements)
  constexpr int IPP = C9 ? C9 / 8 : 64;                            // staged items per pixel
  constexpr int PPASS = SM_NTHR / IPP;                             // pixels per pass
  constexpr int PG = 16 * MT;                                      // pixels per workgroup
  const int H = a.H, W = a.W, HW = a.HW, PW = W + 2;
  const int RS1 = a.c1 + 8;
  T* s_a9 = reinterpret_cast<T*>(smem);                            // [staged rows (+ 2 in the whole-map form)][W + 2][RS9], zero border
  T* s_a1 = reinterpret_cast<T*>(smem + a.a9_bytes);               // [PG][RS1] raw
  f32x4* s_red = reinterpret_cast<f32x4*>(smem + a.a9_bytes + a.a1_bytes);       // [8 waves][MT][64 lanes]
  f32x4* s_part = s_red + 8 * MT * 64;                             // one per thread: (sum, sumsq) of channels 0-3 | 4-7 of its items
  double* s_ostat = reinterpret_cast<double*>(s_part + SM_NTHR);   // [64][2] output unit statistics

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fk = lane >> 4;
  const int ntile = blockIdx.x, pgroup = blockIdx.y, b = blockIdx.z;
  const int64_t img = (int64_t)b * HW;
  (void)BAND;
  // staged rows: image rows r0 .. r0 + nst - 1 (band form: the workgroup's rows and one either side, possibly outside the
  // image; whole-map form: all of them); image pixel (row, col) sits at LDS row (row - r0 + roff), column col + 1
  const int r0 = a.band ? (pgroup * PG) / W - 1 : 0;
  const int nst = a.band ? a.rpw + 2 : H;
  const int roff = a.band ? 0 : 1;
  SSTAMP(0);

  // ---- 1. every global load of the kernel, straight-line -----------------------------------------------------------------------
  // weights: lane (frow, fk) reads 16-byte chunk 4 * khalf + fk of row frow of a (chunk, n-tile) block
  const T* wl = reinterpret_cast<const T*>(a.w) + (int64_t)ntile * 1024 + fk * 128 + frow * 8;
  [[maybe_unused]] uint4 wown[KS9 ? KS9 : 1];                      // this wave's tap (9-tap staged segments), k-steps 0 .. KS9-1
  if constexpr (KS9 > 0) {
#pragma unroll
    for (int k = 0; k < KS9; ++k) {
      const int ch = k * 32;
      const bool s1 = ch >= a.c9_0;                                 // (scalar: which staged segment)
      // (centre-tap form: the staged segments have ONE tap and these fragments go unused - read tap 0, stay inside the buffer)
      const int64_t base = (s1 ? a.wb9[1] : a.wb9[0]) + (int64_t)(a.taps9 == 9 ? wave : 0) * (s1 ? a.wtap9[1] : a.wtap9[0]);
      const int rel = ch - (s1 ? a.c9_0 : 0);
      wown[k] = ld16(wl + base + (int64_t)(rel >> 6) * a.wchunk + ((rel >> 5) & 1) * 512);
    }
  }
  uint4 wpool[4];                                                   // pooled k-steps j = wave, wave + 8, ... (clamped: every load is issued)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = min(wave + 8 * k, a.pooled - 1);
    int64_t off;
    if (KS9 > 0 && j < KS9) {                                      // tap 8 of a 9-tap segment / the only tap of a GroupNorm'd 1-tap one
      const int ch = j * 32;
      const bool s1 = ch >= a.c9_0;
      const int rel = ch - (s1 ? a.c9_0 : 0);
      off = (s1 ? a.wb9[1] : a.wb9[0]) + (a.taps9 == 9 ? 8 : 0) * (s1 ? a.wtap9[1] : a.wtap9[0]) + (int64_t)(rel >> 6) * a.wchunk + ((rel >> 5) & 1) * 512;
    } else {                                                       // a raw 1-tap segment's k-step
      const int ch = (j - KS9) * 32;
      const bool s1 = ch >= a.c1_0;
      const int rel = ch - (s1 ? a.c1_0 : 0);
      off = (s1 ? a.wb1[1] : a.wb1[0]) + (int64_t)(rel >> 6) * a.wchunk + ((rel >> 5) & 1) * 512;
    }
    wpool[k] = ld16(wl + off);
  }
  // the staged rows: thread tid keeps item column tid % IPP (8 channels) of staged pixels tid / IPP + PPASS j
  const int icol = tid % IPP;
  const bool seg1 = icol * 8 >= a.c9_0;                             // this thread's staged segment
  [[maybe_unused]] uint4 raw[NRAW];
  if constexpr (KS9 > 0) {
    const T* src = reinterpret_cast<const T*>(seg1 ? a.src9[1] : a.src9[0]) + img * (seg1 ? a.sC9[1] : a.sC9[0]) + (icol * 8 - (seg1 ? a.c9_0 : 0));
    const int sC = seg1 ? a.sC9[1] : a.sC9[0];
#pragma unroll
    for (int j = 0; j < NRAW; ++j) {
      const int sp = tid / IPP + PPASS * j;                         // staged pixel: row r0 + sp / W, column sp % W
      const int row = min(max(r0 + sp / W, 0), H - 1), col = sp % W; // (clamped: items outside the image / the band are masked below)
      raw[j] = ld16(src + (int64_t)(row * W + col) * sC);
    }
  }
  // raw 1-tap sources: this workgroup's PG pixels (<= 64 x 64 items per 16 pixels: 2 MT per thread; clamped)
  uint4 raw1[2 * MT];
  {
    const int ipp1 = max(a.c1 / 8, 1);
#pragma unroll
    for (int j = 0; j < 2 * MT; ++j) {
      const int q = min(tid + SM_NTHR * j, PG * ipp1 - 1);
      const int px = pgroup * PG + q / ipp1, ch = (q % ipp1) * 8;
      const bool s1 = ch >= a.c1_0;
      const T* base1 = reinterpret_cast<const T*>(a.c1 ? (s1 ? a.src1[1] : a.src1[0]) : a.w);   // (no raw segment: a harmless address)
      raw1[j] = ld16(base1 + (a.c1 ? (img + px) * (s1 ? a.sC1[1] : a.sC1[0]) + (ch - (s1 ? a.c1_0 : 0)) : 0));
    }
  }
  // GroupNorm parameters of this thread's 8 channels
  [[maybe_unused]] f32x4 g0, g1, b0, b1;
  if constexpr (GNP) {
    g0 = *reinterpret_cast<const f32x4*>(a.gamma + icol * 8); g1 = *reinterpret_cast<const f32x4*>(a.gamma + icol * 8 + 4);
    b0 = *reinterpret_cast<const f32x4*>(a.beta + icol * 8); b1 = *reinterpret_cast<const f32x4*>(a.beta + icol * 8 + 4);
  }
  // band form: the unit sums of this thread's group(s) - cpg 4: the two units of its item are two groups; 8 | 16: one group of
  // 2 | 4 units (the item's pair and, for 16, its neighbour's).  Up to 2 partial rows per launch here; all loads independent.
  [[maybe_unused]] double us[4][2];
  if constexpr (GNS == 2) {
    const double* ub = (seg1 ? a.useg[1] : a.useg[0]);
    const int nsp = seg1 ? a.unsp[1] : a.unsp[0], ucnt = seg1 ? a.ucnt[1] : a.ucnt[0];
    const int u_item = (icol * 8 - (seg1 ? a.c9_0 : 0)) >> 2;      // first unit of this item inside its segment
    const int upg = a.cpg >> 2;                                     // units per group (1 | 2 | 4)
    const int u0 = upg <= 2 ? u_item : (u_item & ~3);               // first unit to read: the item's own pair, or the group's four
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      us[q][0] = us[q][1] = 0.0;
      const int u = u0 + min(q, max(upg, 2) - 1);                   // (clamped repeats are not added: see the use)
      for (int sp = 0; sp < nsp; ++sp) {
        const double* r = ub + (((int64_t)b * nsp + sp) * ucnt + u) * 2;
        us[q][0] += r[0]; us[q][1] += r[1];
      }
    }
  }
  // the epilogue's operands (wave t < MT finishes column tile t; absent ones read a zero vector: no load behind a branch)
  const int etile = wave < MT ? wave : 0;
  const int pme_e = pgroup * PG + etile * 16 + frow;
  const int n_out = ntile * 16 + fk * 4;
  const bool live = n_out < a.Cout;
  const int n_ld = live ? n_out : 0;
  const f32x4 e_bias = *reinterpret_cast<const f32x4*>(a.bias ? a.bias + n_ld : g_conv_zero);
  const f32x4 e_tb = *reinterpret_cast<const f32x4*>(a.tbias ? a.tbias + (int64_t)b * a.tb_stride + n_ld : g_conv_zero);
  float e_res[4];
  OutVec<TO>::load(a.res ? reinterpret_cast<const TO*>(a.res) + (img + pme_e) * a.Cout + n_ld : reinterpret_cast<const TO*>(g_conv_zero), e_res);
  SSTAMP(1);

  // ---- 2. zero the padded rows and the output statistics (LDS only) ------------------------------------------------------------
  for (int i = tid; i < a.a9_bytes / 16; i += SM_NTHR) reinterpret_cast<uint4*>(s_a9)[i] = uint4{0u, 0u, 0u, 0u};
  if (stat_out)
    for (int i = tid; i < 128; i += SM_NTHR) s_ostat[i] = 0.0;
  SSTAMP(2);

  // ---- 3. GroupNorm mean / rstd of this thread's channels 0-3 and 4-7 -----------------------------------------------------------
  [[maybe_unused]] float mean_lo = 0.f, rstd_lo = 1.f, mean_hi = 0.f, rstd_hi = 1.f;
  if constexpr (GNS == 1 && KS9 > 0) {
    // by the workgroup itself (the whole map is staged), fixed order
    f32x4 ps = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NRAW; ++j) {
      if (tid / IPP + PPASS * j < HW) {
        const typename V16<T>::x8 e = *reinterpret_cast<const typename V16<T>::x8*>(&raw[j]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float x = (float)e[k], y = (float)e[4 + k];
          ps[0] += x; ps[1] += x * x; ps[2] += y; ps[3] += y * y;
        }
      }
    }
    s_part[tid] = ps;                                              // = [tid / IPP][icol]
    __syncthreads();
    // the PPASS partial sums of the thread's group, all reads in flight at once
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (a.cpg <= 8) {                                              // a group = one item (8 channels) or half an item (4)
      f32x4 v[PPASS];
#pragma unroll
      for (int q = 0; q < PPASS; ++q) v[q] = s_part[q * IPP + icol];
#pragma unroll
      for (int q = 0; q < PPASS; ++q) t += v[q];
      if (a.cpg == 8) { t[0] += t[2]; t[1] += t[3]; t[2] = t[0]; t[3] = t[1]; }
    } else {                                                       // a group = ipg whole items (16 channels: 2)
      const int ipg = a.cpg >> 3, i0 = (icol / ipg) * ipg;
      for (int i = 0; i < ipg; ++i) {
        f32x4 v[PPASS];
#pragma unroll
        for (int q = 0; q < PPASS; ++q) v[q] = s_part[q * IPP + i0 + i];
#pragma unroll
        for (int q = 0; q < PPASS; ++q) t += v[q];
      }
      t[0] += t[2]; t[1] += t[3]; t[2] = t[0]; t[3] = t[1];
    }
    mean_lo = t[0] * a.inv_count; mean_hi = t[2] * a.inv_count;
    rstd_lo = __builtin_amdgcn_rsqf(fmaxf(t[1] * a.inv_count - mean_lo * mean_lo, 0.f) + a.eps);
    rstd_hi = __builtin_amdgcn_rsqf(fmaxf(t[3] * a.inv_count - mean_hi * mean_hi, 0.f) + a.eps);
  } else {
    if constexpr (GNS == 2) {
      // from the producers' unit sums (fp64, like conv_gn_table): group = 1 | 2 | 4 units
      const int upg = a.cpg >> 2;
      double s_lo0 = us[0][0], s_lo1 = us[0][1], s_hi0 = us[1][0], s_hi1 = us[1][1];
      if (upg >= 2) { s_lo0 += us[1][0]; s_lo1 += us[1][1]; }
      if (upg == 4) { s_lo0 += us[2][0] + us[3][0]; s_lo1 += us[2][1] + us[3][1]; }
      if (upg >= 2) { s_hi0 = s_lo0; s_hi1 = s_lo1; }
      const double m0 = s_lo0 * a.inv_count_d, m1 = s_hi0 * a.inv_count_d;
      double v0 = s_lo1 * a.inv_count_d - m0 * m0, v1 = s_hi1 * a.inv_count_d - m1 * m1;
      v0 = v0 < 0.0 ? 0.0 : v0; v1 = v1 < 0.0 ? 0.0 : v1;
      const double x0 = v0 + (double)a.eps, x1 = v1 + (double)a.eps;
      double q0 = __builtin_amdgcn_rsq(x0), q1 = __builtin_amdgcn_rsq(x1);
      q0 = q0 * (1.5 - 0.5 * x0 * q0 * q0); q1 = q1 * (1.5 - 0.5 * x1 * q1 * q1);
      mean_lo = (float)m0; mean_hi = (float)m1; rstd_lo = (float)q0; rstd_hi = (float)q1;
    }
    __syncthreads();                                               // (the zeroing above precedes the interior writes)
  }
  SSTAMP(3);

  // ---- 4. normalise + SiLU in registers; only the rows this workgroup's pixels touch go to LDS ----------------------------------------
  if constexpr (KS9 > 0) {
    const int r_lo = (pgroup * PG) / W - 1, r_hi = (pgroup * PG + PG - 1) / W + 1;
#pragma unroll
    for (int j = 0; j < NRAW; ++j) {
      const int sp = tid / IPP + PPASS * j;
      const int br = sp / W, col = sp - br * W, row = r0 + br;
      if (br < nst && row >= 0 && row < H && row >= r_lo && row <= r_hi) {
        uint4 v = raw[j];
        if constexpr (GNP) {
          typename V16<T>::x8 e = *reinterpret_cast<typename V16<T>::x8*>(&v);
          float y[8];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            y[k] = ((float)e[k] - mean_lo) * rstd_lo * g0[k] + b0[k];
            y[4 + k] = ((float)e[4 + k] - mean_hi) * rstd_hi * g1[k] + b1[k];
          }
          if (a.silu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) y[k] = silu_f(y[k]);
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) e[k] = (T)y[k];
          v = *reinterpret_cast<uint4*>(&e);
        }
        *reinterpret_cast<uint4*>(s_a9 + ((br + roff) * PW + col + 1) * RS9 + icol * 8) = v;
      }
    }
  }
  if (a.c1) {
    const int ipp1 = a.c1 / 8;
#pragma unroll
    for (int j = 0; j < 2 * MT; ++j) {
      const int q = tid + SM_NTHR * j;
      if (q < PG * ipp1) *reinterpret_cast<uint4*>(s_a1 + (q / ipp1) * RS1 + (q % ipp1) * 8) = raw1[j];
    }
  }
  __syncthreads();
  SSTAMP(4);

  // ---- 5. MFMAs: 16 output channels (rows of the weight fragments) x MT tiles of 16 pixels (columns) ----------------------------------
  f32x4 acc[MT];
  int arow_base[MT];                                               // LDS row of (pixel's row - 1, its column - 1) in the padded band
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pme = pgroup * PG + t * 16 + frow;
    const int py = pme / W, pxc = pme - py * W;
    arow_base[t] = (py - 1 - r0 + roff) * PW + pxc;
  }
  if constexpr (KS9 > 0) {
    if (a.taps9 == 9) {
      const int dy = wave / 3, dx = wave - dy * 3;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const T* arow = s_a9 + (arow_base[t] + dy * PW + dx) * RS9 + fk * 8;
#pragma unroll
        for (int k = 0; k < KS9; ++k) Mfma<T>::run(wown[k], *reinterpret_cast<const uint4*>(arow + k * 32), acc[t]);
      }
    }
  }
  {
    // tap 8 of a 9-tap segment sits at (+2, +2) of (row - 1, column - 1), the only tap of a GroupNorm'd 1-tap segment at the centre
    const int sh = a.taps9 == 9 ? 2 : 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = wave + 8 * k;
      if (j < a.pooled) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const T* ap = (KS9 > 0 && j < KS9) ? s_a9 + (arow_base[t] + sh * PW + sh) * RS9 + fk * 8 + j * 32
                                             : s_a1 + (t * 16 + frow) * RS1 + fk * 8 + (j - KS9) * 32;
          Mfma<T>::run(wpool[k], *reinterpret_cast<const uint4*>(ap), acc[t]);
        }
      }
    }
  }
  SSTAMP(5);
#pragma unroll
  for (int t = 0; t < MT; ++t) s_red[(wave * MT + t) * 64 + lane] = acc[t];
  __syncthreads();
  SSTAMP(6);

  // ---- 6. sum of the 8 partial tiles + epilogue: wave t finishes column tile t (conv_epilogue4's arithmetic, prefetched operands) ------
  if (wave < MT) {
    f32x4 pv[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) pv[w] = s_red[(w * MT + wave) * 64 + lane];
    f32x4 s = pv[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) s += pv[w];
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = ((s[r] + e_bias[r]) + e_tb[r]) + e_res[r];
    if (a.res_lo) {
      // upsampled half-resolution residual (the pyramid heads; tap order and weights of conv_epilogue4 / resample2x_kernel).
      // Loaded here, not with the other operands: a few launches per forward use it, every launch would carry its registers.
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
    if (a.comb_pyr) {                                              // Combine('sum') with the input pyramid
      const f32x4 cq = *reinterpret_cast<const f32x4*>(a.comb_pyr + (img + pme_e) * 4);
      const f32x4 cb = *reinterpret_cast<const f32x4*>(a.comb_b + n_ld);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 cw = *reinterpret_cast<const f32x4*>(a.comb_w + (int64_t)(n_ld + r) * 4);
        v[r] += cb[r] + (((cw[0] * cq[0] + cw[1] * cq[1]) + cw[2] * cq[2]) + cw[3] * cq[3]);
      }
    }
    if (live) OutVec<TO>::store(out + (img + pme_e) * a.Cout + n_out, v);
    if constexpr (sizeof(TO) == 2) {                               // statistics are those of the STORED tensor
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (float)(TO)v[r];
    }
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
    if constexpr (MT > 1) __syncthreads();                          // (several waves added into s_ostat)
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
  if (wave == 0) SSTAMP(7);
}

template <typename T, typename TO, int GNS, int KS9, int NRAW, int MT>
static int launch_small_i(const ConvParams& p, const SmallArgs& a, int lds, hipStream_t st) {
  auto kern = &conv_small_kernel<T, TO, GNS, KS9, NRAW, MT>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024);
    if (e != hipSuccess) {
      fdbm_set_error("fdbm_conv_igemm(small): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr = true;
  }
  dim3 grid((unsigned)((p.Cout + 15) / 16), (unsigned)(p.H * p.W / (16 * MT)), (unsigned)p.B);
  kern<<<grid, SM_NTHR, lds, st>>>(a, reinterpret_cast<TO*>(p.out), p.stat_out, p.stat_G, p.stat_nsplit);
  FDBM_LAUNCH_CHECK("fdbm_conv_igemm(small)");
  return 0;
}

// staged items per thread -> the instantiated NRAW
template <typename T, typename TO, int GNS, int KS9, int MT>
static int launch_small_n(const ConvParams& p, const SmallArgs& a, int lds, hipStream_t st) {
  if constexpr (KS9 == 0) {
    return launch_small_i<T, TO, 0, 0, 1, MT>(p, a, lds, st);
  } else {
    const int items = (a.band ? (a.rpw + 2) * a.W : a.HW) * (a.c9 / 8);
    const int nraw = (items + SM_NTHR - 1) / SM_NTHR;
    if constexpr (MT == 1 && GNS != 2) {
      if (!a.band) {
        if (nraw <= 1) return launch_small_i<T, TO, GNS, KS9, 1, 1>(p, a, lds, st);
        if (nraw <= 2) return launch_small_i<T, TO, GNS, KS9, 2, 1>(p, a, lds, st);
      }
    }
    if (nraw <= 4) return launch_small_i<T, TO, GNS, KS9, 4, MT>(p, a, lds, st);
    return launch_small_i<T, TO, GNS, KS9, 8, MT>(p, a, lds, st);
  }
}

template <typename T>
static int launch_small(const ConvParams& p, const SmallArgs& a, int lds, int mt, hipStream_t st) {
  const bool gnp = p.gn_sums != nullptr;
  if (a.c9 == 0) return launch_small_n<T, T, 0, 0, 1>(p, a, lds, st);
  if (mt == 4) {            // band form, four column tiles (32 x 32 level): 256 staged channels only (plan)
    return gnp ? launch_small_n<T, T, 2, 8, 4>(p, a, lds, st) : launch_small_n<T, T, 0, 8, 4>(p, a, lds, st);
  }
  if (a.band) {
    if (a.c9 == 256) return gnp ? launch_small_n<T, T, 2, 8, 1>(p, a, lds, st) : launch_small_n<T, T, 0, 8, 1>(p, a, lds, st);
    return gnp ? launch_small_n<T, T, 2, 16, 1>(p, a, lds, st) : launch_small_n<T, T, 0, 16, 1>(p, a, lds, st);
  }
  if (a.c9 == 256) return gnp ? launch_small_n<T, T, 1, 8, 1>(p, a, lds, st) : launch_small_n<T, T, 0, 8, 1>(p, a, lds, st);
  return gnp ? launch_small_n<T, T, 1, 16, 1>(p, a, lds, st) : launch_small_n<T, T, 0, 16, 1>(p, a, lds, st);
}

// f32 output (the 4-channel pyramid heads): GroupNorm'd 256-channel sources only
template <typename T>
static int launch_small_f32(const ConvParams& p, const SmallArgs& a, int lds, int mt, hipStream_t st) {
  if (mt == 4) return launch_small_n<T, float, 2, 8, 4>(p, a, lds, st);
  if (a.band) return launch_small_n<T, float, 2, 8, 1>(p, a, lds, st);
  return launch_small_n<T, float, 1, 8, 1>(p, a, lds, st);
}

}  // namespace

// can this convolution run on the whole-map / band kernel?  (16-bit tensors, output of the input's type; p filled by
// fdbm_conv_igemm, p.w = fragment-major weights)
bool fdbm_conv_small_ok(const ConvParams& p, bool f32_out) {
  SmallArgs a;
  int lds, mt;
  if (!small_plan(p, &a, &lds, &mt)) return false;
  return !f32_out || (p.gn_sums != nullptr && a.c9 == 256 && a.taps9 == 9);
}

int fdbm_launch_conv_small(const ConvParams& p, int dt_in, int dt_out, hipStream_t st) {
  SmallArgs a;
  int lds, mt;
  if (!small_plan(p, &a, &lds, &mt) || !fdbm_conv_small_ok(p, dt_out == FDBM_F32)) {
    fdbm_set_error("fdbm_conv_igemm(small): shape not supported");
    return 1;
  }
  if (dt_in == FDBM_BF16 && dt_out == FDBM_F32) return launch_small_f32<bf16_t>(p, a, lds, mt, st);
  if (dt_in == FDBM_F16 && dt_out == FDBM_F32) return launch_small_f32<f16_t>(p, a, lds, mt, st);
  if (dt_in == FDBM_BF16) return launch_small<bf16_t>(p, a, lds, mt, st);
  if (dt_in == FDBM_F16) return launch_small<f16_t>(p, a, lds, mt, st);
  fdbm_set_error("fdbm_conv_igemm(small): 16-bit tensors only");
  return 1;
}
