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
//   * every weight fragment the workgroup will use is requested FIRST (they depend on nothing): wave w owns tap w of the
//     9-tap segments; tap 8 and the 1-tap (shortcut / NIN) segments are dealt out k-step by k-step over the 8 waves;
//   * the WHOLE map of the 9-tap sources is staged once (H*W x C items of 16 bytes), its GroupNorm statistics are summed
//     by the workgroup itself in a fixed order (thread -> LDS partials -> group totals: no statistics buffers read, no
//     atomics, one barrier), normalised + SiLU'd in registers and written to LDS with a zero border; 1-tap sources: the
//     workgroup's own 16 pixels, raw;
//   * one pass of MFMAs per wave straight from LDS (row stride C + 8 elements: conflict-free b128), partial accumulators
//     summed through LDS, shared epilogue (conv_epilogue4: bias, time bias, residual, scale, Combine, upsampled residual)
//     and the output's unit statistics for consumers that still read them (resampling, larger maps).
// Results: the same convolution as every other kernel of fdbm_conv_igemm; GroupNorm mean / variance from fp32 sums over
// H*W*cpg <= 1 024 values in a fixed order (the producers' fp64 unit sums are not read).
// Roofline: latency (these launches hold < 0.1 us of MFMA work per CU); selected by fdbm_conv_igemm for 16-bit tensors when
// the padded map fits the LDS (fdbm_conv_small_ok).
#include "conv_common.h"

#define SM_NTHR 512
#define SM_MAXW 24            // weight fragments (uint4) a wave may hold: 16 own-tap k-steps... see small_plan

// Diagnostic build only (-DFDBM_STAMPS, tools/small_timeline.py): workgroup (0, 0, 0) writes realtime-clock stamps (100 MHz)
// of its phases into the scratch the caller passed as acc_ws.  The product library has none of it.
#ifdef FDBM_STAMPS
#define SSTAMP(i)                                                                                         \
  do {                                                                                                    \
    if (p.partial && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)            \
      reinterpret_cast<unsigned long long*>(p.partial)[i] = __builtin_amdgcn_s_memrealtime();             \
  } while (0)
#else
#define SSTAMP(i)
#endif

namespace {

struct SmallPlan {
  int c9, c1;                 // channels of the 9-tap / 1-tap segments (sums)
  int n9, n1;                 // number of 9-tap / 1-tap segments
  int ks9, ks1;               // 32-channel k-steps of one tap of the 9-tap segments / of the 1-tap segments
  int pooled;                 // k-steps dealt out over the waves: tap 8's + the 1-tap segments'
  int lds_bytes;
};

static bool small_plan(const ConvParams& p, SmallPlan* sp) {
  SmallPlan s;
  memset(&s, 0, sizeof(s));
  bool seen1 = false;
  for (int i = 0; i < p.nseg; ++i) {
    if (p.seg[i].cin % 64 != 0 || p.seg[i].coff % 8 != 0 || p.seg[i].C % 8 != 0) return false;
    if (p.seg[i].taps == 9) { if (seen1) return false; s.c9 += p.seg[i].cin; ++s.n9; }
    else { seen1 = true; s.c1 += p.seg[i].cin; ++s.n1; }
  }
  if (s.c9 > 512 || s.c1 > 512 || s.n9 > 2 || s.n1 > 2) return false;
  const int HW = p.H * p.W;
  if (HW % 16 != 0 || HW > 128 || p.W < 2) return false;
  if (HW * (s.c9 / 8) > 8 * SM_NTHR) return false;                 // at most 8 staged items per thread (registers)
  if (s.c9 && (SM_NTHR % (s.c9 / 8)) != 0) return false;           // a thread keeps ONE item column: c9 = 64 | 128 | 256 | 512
  s.ks9 = s.c9 / 32;
  s.ks1 = s.c1 / 32;
  s.pooled = (s.c9 ? s.ks9 : 0) + s.ks1;
  // per wave: its own tap's k-steps + its share of the pooled ones
  if (s.ks9 + (s.pooled + 7) / 8 > SM_MAXW) return false;
  // GroupNorm prologue: groups made of whole 16-byte items, all flagged segments are 9-tap ones covering c9
  if (p.gn_sums) {
    if (p.gn_C != s.c9 || s.c9 == 0) return false;
    const int cpg = p.gn_C / p.gn_G;
    if ((cpg % 8 != 0 && cpg != 4) || p.gn_G > 128) return false;   // groups of whole 16-byte items, or of half an item
    for (int i = 0; i < p.nseg; ++i)
      if ((p.seg_gn[i] >= 0) != (p.seg[i].taps == 9)) return false;
  }
  const int act9 = s.c9 ? (p.H + 2) * (p.W + 2) * (s.c9 + 8) * 2 : 0;
  const int act1 = s.c1 ? 16 * (s.c1 + 8) * 2 : 0;
  s.lds_bytes = act9 + act1 + 8 * 64 * 16 /*partials*/ + SM_NTHR * 16 /*stat partials*/ + 64 * 8 * 2 /*out stats*/ + 64;
  if (s.lds_bytes > 150 * 1024) return false;
  *sp = s;
  return true;
}

// one 16-byte load of activations
template <typename T>
__device__ __forceinline__ uint4 ld16(const T* p) { return *reinterpret_cast<const uint4*>(p); }

template <typename T, typename TO, bool GNP>
__global__ void __launch_bounds__(SM_NTHR) conv_small_kernel(const ConvParams p, const SmallPlan sp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int H = p.H, W = p.W, HW = H * W, PW = W + 2;
  const int RS9 = sp.c9 + 8, RS1 = sp.c1 + 8;                      // LDS row strides (elements)
  T* s_a9 = reinterpret_cast<T*>(smem);                            // [(H+2)(W+2)][RS9] activated, zero border
  const int a9_bytes = sp.c9 ? (H + 2) * PW * RS9 * 2 : 0;
  T* s_a1 = reinterpret_cast<T*>(smem + a9_bytes);                 // [16][RS1] raw
  const int a1_bytes = sp.c1 ? 16 * RS1 * 2 : 0;
  f32x4* s_red = reinterpret_cast<f32x4*>(smem + a9_bytes + a1_bytes);          // [8 waves][64 lanes]
  f32x4* s_part = s_red + 8 * 64;                                                // [parts][item columns] = one per thread: (sum, sumsq) of channels 0-3 | 4-7
  double* s_ostat = reinterpret_cast<double*>(s_part + SM_NTHR);                 // [64][2] output unit statistics

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fk = lane >> 4;
  const int ntile = blockIdx.x, pgroup = blockIdx.y, b = blockIdx.z;
  const int64_t img = (int64_t)b * HW;
  const T* wbase = reinterpret_cast<const T*>(p.w);                // fragment-major [kidx][CoutPad/16][8 chunks][16 rows][8]

  SSTAMP(0);
  // ---- 1. every weight fragment of this wave, requested before anything else --------------------------------------------
  // k-step numbering: one tap of the 9-tap segments = ks9 steps of 32 channels (segment after segment); the 1-tap segments
  // ks1 steps.  Packed weights: kidx = kbase_seg + tap * nch_seg + chunk (64-channel chunks), a 32-channel step is k-half
  // (step & 1) of chunk (step >> 1): lane (frow, fk) reads 16-byte chunk 4 * khalf + fk of row frow.
  auto wfrag_ptr = [&](int seg, int tap, int step_in_seg) __attribute__((always_inline)) {
    int kb = 0;
    for (int i = 0; i < seg; ++i) kb += SEG_FIELD(p, i, taps) * (SEG_FIELD(p, i, cin) / 64);
    const int nch = SEG_FIELD(p, seg, cin) / 64;
    const int kidx = kb + tap * nch + (step_in_seg >> 1);
    return wbase + (((int64_t)kidx * (p.CoutPad / 16) + ntile) * 8 + (step_in_seg & 1) * 4 + fk) * 128 + frow * 8;
  };
  // segment of a k-step index within the 9-tap (or 1-tap) channel axis
  const int c9_0 = sp.n9 >= 1 ? p.seg[0].cin : 0;                  // channels of the first 9-tap segment
  const int c1_0 = sp.n1 >= 1 ? SEG_FIELD(p, sp.n9, cin) : 0;      // ... of the first 1-tap segment
  uint4 wown[16];                                                  // this wave's tap (taps 0..7), k-steps 0 .. ks9-1
  uint4 wpool[4];                                                  // pooled k-steps j = wave, wave + 8, ... (<= 32 of them)
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    wown[k] = uint4{0u, 0u, 0u, 0u};
    if (k < sp.ks9) {
      const int ch = k * 32;
      const int seg = ch < c9_0 ? 0 : 1;
      wown[k] = ld16(wfrag_ptr(seg, wave, (ch - (seg ? c9_0 : 0)) / 32));
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    wpool[k] = uint4{0u, 0u, 0u, 0u};
    const int j = wave + 8 * k;
    if (j < sp.pooled) {
      if (sp.c9 && j < sp.ks9) {                                   // tap 8 of the 9-tap segments
        const int ch = j * 32;
        const int seg = ch < c9_0 ? 0 : 1;
        wpool[k] = ld16(wfrag_ptr(seg, 8, (ch - (seg ? c9_0 : 0)) / 32));
      } else {                                                     // a 1-tap segment's k-step
        const int ch = (j - (sp.c9 ? sp.ks9 : 0)) * 32;
        const int s1 = ch < c1_0 ? 0 : 1;
        wpool[k] = ld16(wfrag_ptr(sp.n9 + s1, 0, (ch - (s1 ? c1_0 : 0)) / 32));
      }
    }
  }

  SSTAMP(1);
  // ---- 2. stage the map ----------------------------------------------------------------------------------------------------
  // items of 16 bytes: (pixel, 8 channels of the concatenated 9-tap channel axis); thread tid keeps item column tid % ipp
  const int ipp = sp.c9 ? sp.c9 / 8 : 64;                          // items per pixel (32 | 64; no 9-tap segment: unused)
  const int icol = tid % ipp, ppass = SM_NTHR / ipp;               // pixels per pass (16 | 8)
  uint4 raw[8];
  {
    const int ch = icol * 8;
    const int seg = ch < c9_0 ? 0 : 1;
    const T* src = reinterpret_cast<const T*>(SEG_FIELD(p, seg, src)) + img * SEG_FIELD(p, seg, C) + SEG_FIELD(p, seg, coff) + (ch - (seg ? c9_0 : 0));
    const int sC = SEG_FIELD(p, seg, C);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      raw[j] = uint4{0u, 0u, 0u, 0u};
      const int px = tid / ipp + ppass * j;
      if (sp.c9 && px < HW) raw[j] = ld16(src + (int64_t)px * sC);
    }
  }
  // 1-tap sources: this workgroup's 16 pixels, raw
  uint4 raw1[2];
  {
    const int ipp1 = sp.c1 / 8;                                    // <= 64 items per pixel: 16 x 64 = 1 024 items, 2 per thread
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      raw1[j] = uint4{0u, 0u, 0u, 0u};
      const int q = tid + SM_NTHR * j;
      if (sp.c1 && q < 16 * ipp1) {
        const int px = pgroup * 16 + q / ipp1, ch = (q % ipp1) * 8;
        const int s1 = ch < c1_0 ? 0 : 1;
        const int seg = sp.n9 + s1;
        raw1[j] = ld16(reinterpret_cast<const T*>(SEG_FIELD(p, seg, src)) + (img + px) * SEG_FIELD(p, seg, C) + SEG_FIELD(p, seg, coff) + (ch - (s1 ? c1_0 : 0)));
      }
    }
  }
  // GroupNorm parameters of this thread's 8 channels (its items all sit in item column icol)
  float gam[8], bet[8];
  if constexpr (GNP) {
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gn_gamma + icol * 8), g1 = *reinterpret_cast<const f32x4*>(p.gn_gamma + icol * 8 + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.gn_beta + icol * 8), b1 = *reinterpret_cast<const f32x4*>(p.gn_beta + icol * 8 + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) { gam[k] = g0[k]; gam[4 + k] = g1[k]; bet[k] = b0[k]; bet[4 + k] = b1[k]; }
  }
  // zero the padded map (the border stays zero: padding AFTER the activation) and the output statistics
  for (int i = tid; i < a9_bytes / 16; i += SM_NTHR) reinterpret_cast<uint4*>(s_a9)[i] = uint4{0u, 0u, 0u, 0u};
  if (p.stat_out)
    for (int i = tid; i < 128; i += SM_NTHR) s_ostat[i] = 0.0;

  SSTAMP(2);
  // ---- 3. GroupNorm statistics by the workgroup itself, fixed order --------------------------------------------------------
  float mean_lo = 0.f, rstd_lo = 1.f, mean_hi = 0.f, rstd_hi = 1.f;        // of the item's channels 0-3 / 4-7
  if constexpr (GNP) {
    f32x4 ps = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const typename V16<T>::x8 e = *reinterpret_cast<const typename V16<T>::x8*>(&raw[j]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {                                // (absent items are zero)
        const float a = (float)e[k], c = (float)e[4 + k];
        ps[0] += a; ps[1] += a * a; ps[2] += c; ps[3] += c * c;
      }
    }
    s_part[tid] = ps;                                              // (= [tid / ipp][icol])
    __syncthreads();
    const int cpg = p.gn_C / p.gn_G;
    if (cpg == 4) {                                                // a group = half an item
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      for (int q = 0; q < ppass; ++q) t += s_part[q * ipp + icol];
      const float inv = 1.0f / (float)(HW * 4);
      mean_lo = t[0] * inv; mean_hi = t[2] * inv;
      rstd_lo = __builtin_amdgcn_rsqf(fmaxf(t[1] * inv - mean_lo * mean_lo, 0.f) + p.gn_eps);
      rstd_hi = __builtin_amdgcn_rsqf(fmaxf(t[3] * inv - mean_hi * mean_hi, 0.f) + p.gn_eps);
    } else {                                                       // a group = ipg whole items
      const int ipg = cpg / 8;
      const int i0 = (icol / ipg) * ipg;
      float t1 = 0.f, t2 = 0.f;
      for (int q = 0; q < ppass; ++q)
        for (int i = 0; i < ipg; ++i) { const f32x4 v = s_part[q * ipp + i0 + i]; t1 += v[0] + v[2]; t2 += v[1] + v[3]; }
      const float inv = 1.0f / (float)(HW * cpg);
      mean_lo = mean_hi = t1 * inv;
      rstd_lo = rstd_hi = __builtin_amdgcn_rsqf(fmaxf(t2 * inv - mean_lo * mean_lo, 0.f) + p.gn_eps);
    }
  } else {
    __syncthreads();                                               // (the zeroing above precedes the interior writes)
  }
  SSTAMP(3);
  // ---- 4. normalise + SiLU in registers, write the interior ----------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int px = tid / ipp + ppass * j;
    if (sp.c9 && px < HW) {
      uint4 v = raw[j];
      if constexpr (GNP) {
        typename V16<T>::x8 e = *reinterpret_cast<typename V16<T>::x8*>(&v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float y = ((float)e[k] - (k < 4 ? mean_lo : mean_hi)) * (k < 4 ? rstd_lo : rstd_hi) * gam[k] + bet[k];
          if (p.gn_silu) y = silu_f(y);
          e[k] = (T)y;
        }
        v = *reinterpret_cast<uint4*>(&e);
      }
      const int py = px / W, pxx = px - py * W;
      *reinterpret_cast<uint4*>(s_a9 + ((py + 1) * PW + pxx + 1) * RS9 + icol * 8) = v;
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = tid + SM_NTHR * j;
    const int ipp1 = sp.c1 / 8;
    if (sp.c1 && q < 16 * ipp1) *reinterpret_cast<uint4*>(s_a1 + (q / ipp1) * RS1 + (q % ipp1) * 8) = raw1[j];
  }
  __syncthreads();

  SSTAMP(4);
  // ---- 5. MFMAs: 16 output channels (rows of the weight fragments) x this workgroup's 16 pixels (columns) -------------------
  const int pme = pgroup * 16 + frow, pyy = pme / W, pxx = pme - pyy * W;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (sp.c9) {
    const int dy = wave / 3, dx = wave - dy * 3;
    const T* arow = s_a9 + ((pyy + dy) * PW + pxx + dx) * RS9 + fk * 8;
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k < sp.ks9) Mfma<T>::run(wown[k], *reinterpret_cast<const uint4*>(arow + k * 32), acc);
  }
  {
    const T* arow8 = s_a9 + ((pyy + 2) * PW + pxx + 2) * RS9 + fk * 8;
    const T* arow1 = s_a1 + frow * RS1 + fk * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = wave + 8 * k;
      if (j < sp.pooled) {
        const bool t8 = sp.c9 && j < sp.ks9;
        const T* ap = t8 ? arow8 + j * 32 : arow1 + (j - (sp.c9 ? sp.ks9 : 0)) * 32;
        Mfma<T>::run(wpool[k], *reinterpret_cast<const uint4*>(ap), acc);
      }
    }
  }
  SSTAMP(5);
  s_red[wave * 64 + lane] = acc;
  __syncthreads();
  SSTAMP(6);

  // ---- 6. sum of the 8 partial tiles + epilogue, by wave 0 ------------------------------------------------------------------
  if (wave == 0) {
    f32x4 s = s_red[lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) s += s_red[w * 64 + lane];
    const int n = ntile * 16 + fk * 4;
    const int64_t m = img + pme;
    float v[4] = {s[0], s[1], s[2], s[3]};
    const bool live = n < p.Cout;
    if (live) conv_epilogue4<TO>(p, m, b, n, v);
    if (p.stat_out) {
      const int scpg = p.Cout / p.stat_G;
      const float q1 = live ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
      const float q2 = live ? (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]) : 0.f;
      const float r1 = row16_sum(q1), r2 = row16_sum(q2);
      if (frow == 0 && live) {
        // (4 consecutive channels = one unit when scpg == 4; larger groups: several lanes add into one slot)
        atomicAdd(&s_ostat[((n - ntile * 16) / scpg) * 2], (double)r1);
        atomicAdd(&s_ostat[((n - ntile * 16) / scpg) * 2 + 1], (double)r2);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the LDS atomics of this wave have landed
      const int g0 = (ntile * 16) / scpg;
      const int ng = min(p.stat_G - g0, (16 + scpg - 1) / scpg);
      if (lane < ng * 2) {
        const int k = lane & 1, g = g0 + (lane >> 1);
        atomicAdd(p.stat_out + (((int64_t)b * p.stat_nsplit + pgroup % p.stat_nsplit) * p.stat_G + g) * 2 + k, s_ostat[(g - g0) * 2 + k]);
      }
    }
    SSTAMP(7);
  }
}

template <typename T, typename TO>
static int launch_small(const ConvParams& p, const SmallPlan& sp, hipStream_t st) {
  const bool gnp = p.gn_sums != nullptr;
  auto kern = gnp ? &conv_small_kernel<T, TO, true> : &conv_small_kernel<T, TO, false>;
  static bool attr[2] = {false, false};
  if (!attr[gnp]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) {
      fdbm_set_error("fdbm_conv_igemm(small): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr[gnp] = true;
  }
  dim3 grid((unsigned)((p.Cout + 15) / 16), (unsigned)(p.H * p.W / 16), (unsigned)p.B);
  kern<<<grid, SM_NTHR, sp.lds_bytes, st>>>(p, sp);
  FDBM_LAUNCH_CHECK("fdbm_conv_igemm(small)");
  return 0;
}

}  // namespace

// can this convolution run on the whole-map kernel?  (16-bit tensors; p filled by fdbm_conv_igemm, p.w = fragment-major weights)
bool fdbm_conv_small_ok(const ConvParams& p) {
  SmallPlan sp;
  return small_plan(p, &sp);
}

int fdbm_launch_conv_small(const ConvParams& p, int dt_in, int dt_out, hipStream_t st) {
  SmallPlan sp;
  if (!small_plan(p, &sp)) {
    fdbm_set_error("fdbm_conv_igemm(small): shape not supported");
    return 1;
  }
  if (dt_in == FDBM_BF16 && dt_out == FDBM_BF16) return launch_small<bf16_t, bf16_t>(p, sp, st);
  if (dt_in == FDBM_BF16 && dt_out == FDBM_F32) return launch_small<bf16_t, float>(p, sp, st);
  if (dt_in == FDBM_F16 && dt_out == FDBM_F16) return launch_small<f16_t, f16_t>(p, sp, st);
  if (dt_in == FDBM_F16 && dt_out == FDBM_F32) return launch_small<f16_t, float>(p, sp, st);
  fdbm_set_error("fdbm_conv_igemm(small): 16-bit tensors only");
  return 1;
}
