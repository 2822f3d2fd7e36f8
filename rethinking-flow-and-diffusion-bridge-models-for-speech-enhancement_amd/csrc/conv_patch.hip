// conv_patch.hip - 3x3 convolution for the large feature maps: halo patch in LDS, 9 taps per
// staged patch.
//
// The tap-outer kernel (conv.hip) re-stages the activation tile once per tap: 9x the L2->LDS
// traffic and, with a fused GroupNorm prologue, 9x the transform work.  Here a workgroup owns
// a TH x 16 pixel tile of one image and all BN output channels; per 128-byte channel chunk it
// stages the (TH+2) x 18 halo patch ONCE (GroupNorm scale/shift + SiLU applied on the way,
// zero padding after the activation) and runs the 9 taps as 9 k-steps whose activation operand
// is the same LDS patch read at a shifted row; only the 16 KiB weight tile changes per k-step.
//   L2->LDS bytes per k-step: 16 KiB (W) + patch/9  vs  16 KiB + 16..32 KiB (tap-outer)
//   GroupNorm/SiLU work: (TH+2)*18 / (TH*16) = 1.27..1.4x the tensor vs 9x.
// Roofline: MFMA-bound.  Waves: (TH/4) x (BN/64); each wave 4 patch rows x 16 px x 64 channels
// = 4 x 4 MFMA tiles of 16x16 (acc in 64 registers); operands exactly as in conv.hip
// (weights = A operand, activations = B operand, K-contiguous 128-byte LDS rows, XOR-swizzled
// 16-byte chunks, one ds_read_b128 per fragment).
// Pipeline: weight tile of step t+1 and (at the first tap of a chunk) the whole next patch are
// loaded into registers before the MFMAs of step t and written to the other LDS buffers after
// them (the patch at the chunk's last tap); one barrier per k-step.
// 1-tap segments (the res-block's 1x1 shortcut riding in the same accumulator) reuse the patch
// path with the centre tap only.
#include "conv_common.h"

// SPL (T = float only): split-precision matrix mode - patch and weight rows hold [32 halves hi | 32 halves lo]
// (conv_common.h), every product is three f16 MFMAs: hi.hi + hi.lo + lo.hi.
template <typename T, typename TO, int TH, int BN, bool GNP, bool SPL = false>
__global__ void __launch_bounds__(64 * (TH / 4) * (BN >= 64 ? BN / 64 : 1))
conv_patch_kernel(const ConvParams p, int tiles_x, int tiles_y) {
  constexpr int KC = 128 / (int)sizeof(T);
  constexpr int VW = 16 / (int)sizeof(T);
  // (BN = 16: the 4-channel heads - one n-tile per wave instead of four, a sixteenth of the weight tile)
  constexpr int WM = TH / 4, WN = BN >= 64 ? BN / 64 : 1;
  constexpr int NTHR = 64 * WM * WN;
  constexpr int MT = 4, NT = BN >= 64 ? 4 : BN / 16;
  constexpr int PC = 18, PROWS = (TH + 2) * PC;
  constexpr int PBUF = PROWS * 128;
  constexpr int WBUF = BN * 128;
  constexpr int NPL = (PROWS * 8 + NTHR - 1) / NTHR;   // patch 16-byte items per thread
  constexpr int NWL = (BN * 8 + NTHR - 1) / NTHR;      // weight 16-byte items per thread
  constexpr bool F32 = sizeof(T) == 4 && !SPL;
  static_assert(!SPL || sizeof(T) == 4, "the split-precision mode stages f32 tensors");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_patch = smem;                          // 2 x PBUF
  unsigned char* s_w = smem + 2 * PBUF;                   // 2 x WBUF
  float* s_gn = reinterpret_cast<float*>(smem + 2 * PBUF + 2 * WBUF);                         // scale[gnpad] | shift[gnpad]
  double* s_stat = reinterpret_cast<double*>(smem + 2 * PBUF + 2 * WBUF + (GNP ? ((p.gn_C + 63) & ~63) * 8 : 0));  // [32][2] doubles (after the table)
  float* s_mr = reinterpret_cast<float*>(s_stat + 64);                                    // [32][2] mean, rstd

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 15, fk = lane >> 4;
  const int H = p.H, W = p.W;
  const int tile = blockIdx.x;
  const int tx = tile % tiles_x;
  const int ty = (tile / tiles_x) % tiles_y;
  const int b = tile / (tiles_x * tiles_y);
  const int y0 = ty * TH, x0 = tx * 16;
  const int n0 = blockIdx.y * BN;
  const int64_t img = (int64_t)b * H * W;

  // ---- per-thread patch items: which pixel, where in LDS, real or padding ---------------------
  int ppix[NPL], plds[NPL];
  unsigned pmask = 0;
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const int q = tid + NTHR * j;
    const int prow = min(q >> 3, PROWS - 1), pch = q & 7;
    const int pr = prow / PC, pc = prow - pr * PC;
    const int iy = y0 + pr - 1, ix = x0 + pc - 1;
    const bool ok = (q < PROWS * 8) && iy >= 0 && iy < H && ix >= 0 && ix < W;
    ppix[j] = min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 1);
    // swizzle key from the patch COLUMN (see a_base); split mode: the item's 4 channels are 8 bytes of the hi plane,
    // their lo halves sit 64 bytes on (address ^ 64)
    if constexpr (SPL) plds[j] = (q < PROWS * 8) ? prow * 128 + (((pch >> 1) ^ ((pc >> 1) & 7)) << 4) + (pch & 1) * 8 : -1;
    else plds[j] = (q < PROWS * 8) ? prow * 128 + ((pch ^ ((pc >> 1) & 7)) << 4) : -1;
    pmask |= ok ? (1u << j) : 0u;
  }
  const int pchunk = tid & 7;          // NTHR is a multiple of 8: the 16-byte chunk is fixed per thread

  if (p.stat_out)
    for (int i = tid; i < 64; i += NTHR) s_stat[i] = 0.0;

  auto build_gn_table = [&]() __attribute__((always_inline)) {
    // (scratch: the second patch buffer, idle until the first chunk's taps; the first weight tile is
    // being written to s_w meanwhile)
    if constexpr (GNP) conv_gn_table<NTHR>(p, b, 1, s_gn, (p.gn_C + 63) & ~63, s_mr, s_patch + PBUF);
  };

  // ---- chunk cursor -------------------------------------------------------------------------------
  // (cs, cc): segment and channel chunk; kbase: first packed k-step of the segment
  uint4 preg[NPL];
  int pgcb = -1;                        // GN channel of this thread's chunk for the patch in preg
  bool pcok = true;                     // channel chunk inside the segment

  auto seg_nch = [&](int s) __attribute__((always_inline)) { return (SEG_FIELD(p, s, cin) + KC - 1) / KC; };

  auto load_patch = [&](int s, int c) __attribute__((always_inline)) {
    const void* sg_src = SEG_FIELD(p, s, src);
    const int sg_C = SEG_FIELD(p, s, C), sg_coff = SEG_FIELD(p, s, coff), sg_cin = SEG_FIELD(p, s, cin);
    const int cvalid = min(KC, sg_cin - c * KC);
    pcok = pchunk * VW < cvalid;
    const T* src = reinterpret_cast<const T*>(sg_src) + img * sg_C + sg_coff + (pcok ? c * KC + pchunk * VW : 0);
#pragma unroll
    for (int j = 0; j < NPL; ++j) preg[j] = *reinterpret_cast<const uint4*>(src + (int64_t)ppix[j] * sg_C);
    if constexpr (GNP) {
      const int sgn = s == 0 ? p.seg_gn[0] : s == 1 ? p.seg_gn[1] : s == 2 ? p.seg_gn[2] : p.seg_gn[3];
      pgcb = sgn >= 0 ? sgn + c * KC + pchunk * VW : -1;
    }
  };

  // transform (GroupNorm scale/shift + SiLU) and store ONE patch item; branch-free so that the
  // scheduler can interleave it with the MFMAs of the tap it is issued in
  auto write_patch_item = [&](auto JJ, int buf) __attribute__((always_inline)) {
    constexpr int j = decltype(JJ)::value;
    if constexpr (j < NPL) {
      unsigned char* P = s_patch + buf * PBUF;
      uint4 v = preg[j];
      if constexpr (GNP) {
        if (pgcb >= 0)                     // wave-uniform: a property of the segment
        {
          if constexpr (SPL) {
            if (p.gn_silu == 2) v = gn_transform16<T, true>(v, s_gn + pgcb, s_gn + ((p.gn_C + 63) & ~63) + pgcb, true);
            else v = gn_transform16<T>(v, s_gn + pgcb, s_gn + ((p.gn_C + 63) & ~63) + pgcb, p.gn_silu != 0);
          } else {
            v = gn_transform16<T>(v, s_gn + pgcb, s_gn + ((p.gn_C + 63) & ~63) + pgcb, p.gn_silu != 0);
          }
        }
      }
      if (!(((pmask >> j) & 1u) && pcok)) v = uint4{0u, 0u, 0u, 0u};   // padding AFTER the activation
      if constexpr (SPL) {
        uint2 hi, lo;
        split_f16x4(*reinterpret_cast<const f32x4*>(&v), hi, lo);
        if (plds[j] >= 0) {
          *reinterpret_cast<uint2*>(P + plds[j]) = hi;
          *reinterpret_cast<uint2*>(P + (plds[j] ^ 64)) = lo;
        }
      } else {
        if (plds[j] >= 0) *reinterpret_cast<uint4*>(P + plds[j]) = v;
      }
    }
  };
  auto write_patch = [&](int buf) __attribute__((always_inline)) {
    write_patch_item(std::integral_constant<int, 0>{}, buf);
    write_patch_item(std::integral_constant<int, 1>{}, buf);
    write_patch_item(std::integral_constant<int, 2>{}, buf);
    write_patch_item(std::integral_constant<int, 3>{}, buf);
    write_patch_item(std::integral_constant<int, 4>{}, buf);
    write_patch_item(std::integral_constant<int, 5>{}, buf);
    write_patch_item(std::integral_constant<int, 6>{}, buf);
    write_patch_item(std::integral_constant<int, 7>{}, buf);
    write_patch_item(std::integral_constant<int, 8>{}, buf);
    write_patch_item(std::integral_constant<int, 9>{}, buf);
    write_patch_item(std::integral_constant<int, 10>{}, buf);
    write_patch_item(std::integral_constant<int, 11>{}, buf);
  };
  static_assert(NPL <= 12, "patch staging assumes at most 12 items per thread");

  // weight staging registers as four named scalars (an array here ends up in scratch memory)
  uint4 wr0 = {0u, 0u, 0u, 0u}, wr1 = wr0, wr2 = wr0, wr3 = wr0;
  static_assert(NWL <= 4, "weight staging assumes at most 4 items per thread");
#define W_LOAD(J)                                                                              \
  if constexpr (NWL > J) {                                                                     \
    const int q = min(tid + NTHR * J, BN * 8 - 1);                                             \
    wr##J = *reinterpret_cast<const uint4*>(wp + (int64_t)(q >> 3) * KC + (q & 7) * VW);       \
  }
#define W_STORE(J)                                                                             \
  if constexpr (NWL > J) {                                                                     \
    const int q = tid + NTHR * J;                                                              \
    const int row = q >> 3, ch = q & 7;                                                        \
    if (q < BN * 8) *reinterpret_cast<uint4*>(Wt + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = wr##J;       \
  }
  auto load_w = [&](int kidx) __attribute__((always_inline)) {
    const T* wp = reinterpret_cast<const T*>(p.w) + ((int64_t)kidx * p.CoutPad + n0) * KC;
    W_LOAD(0) W_LOAD(1) W_LOAD(2) W_LOAD(3)
  };
  auto write_w = [&](int buf) __attribute__((always_inline)) {
    unsigned char* Wt = s_w + buf * WBUF;
    W_STORE(0) W_STORE(1) W_STORE(2) W_STORE(3)
  };
#undef W_LOAD
#undef W_STORE

  f32x4 acc[NT][MT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Fragment addresses as (per-lane base) + (wave-uniform offset) + (immediate): the XOR swizzle key of a patch row
  // is taken from its COLUMN pc = frow + dx ((pc >> 1) & 7 - within one ds_read_b128 the 16 lanes read 16 consecutive
  // columns of one patch row, exactly the rows the key has to spread), so it depends on the lane and on dx only; the
  // key of a weight row (wn * 64 + j * 16 + frow) is (frow >> 1) & 7 whatever j and wn.  rocprofv3 counted 6.5 VALU
  // instructions per MFMA in this kernel with the addresses recomputed per fragment - more issue cycles than the MFMAs.
#define A_BASE(DX, KK) ((wm * 4 * PC + frow + (DX)) * 128 + ((((KK) * 4 + fk) ^ (((frow + (DX)) >> 1) & 7)) << 4))
  const int ab00 = A_BASE(0, 0), ab01 = A_BASE(0, 1), ab10 = A_BASE(1, 0), ab11 = A_BASE(1, 1), ab20 = A_BASE(2, 0), ab21 = A_BASE(2, 1);
#undef A_BASE
  const int wb0 = (wn * (NT * 16) + frow) * 128 + (((0 * 4 + fk) ^ ((frow >> 1) & 7)) << 4);
  const int wb1 = (wn * (NT * 16) + frow) * 128 + (((1 * 4 + fk) ^ ((frow >> 1) & 7)) << 4);
  // one half (kk = 0 | 1: 16 of the 32 k-values of a 128-byte row) of a k-step's MFMAs
  auto compute = [&](int pbuf, int wbuf, int dy, int dx, int kk) __attribute__((always_inline)) {
    const unsigned char* P = s_patch + pbuf * PBUF;
    const unsigned char* Wt = s_w + wbuf * WBUF;
    f32x4 part[NT][MT];
    if constexpr (F32) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) part[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    {
      uint4 wf[NT], af[MT];
      const unsigned char* wsrc = Wt + (kk == 0 ? wb0 : wb1);
      // (split mode: half 0 = w_hi . (a_hi, a_lo), half 1 = w_lo . a_hi)
      const unsigned char* asrc = P + dy * (PC * 128) +
                                  ((kk == 0 || SPL) ? (dx == 0 ? ab00 : dx == 1 ? ab10 : ab20) : (dx == 0 ? ab01 : dx == 1 ? ab11 : ab21));
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const uint4*>(wsrc + j * (16 * 128));
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const uint4*>(asrc + i * (PC * 128));
      if constexpr (SPL) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) Mfma<f16_t>::run(wf[j], af[i], acc[j][i]);
        if (kk == 0) {
          const unsigned char* lsrc = P + dy * (PC * 128) + (dx == 0 ? ab01 : dx == 1 ? ab11 : ab21);
#pragma unroll
          for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const uint4*>(lsrc + i * (PC * 128));
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) Mfma<f16_t>::run(wf[j], af[i], acc[j][i]);
        }
        __builtin_amdgcn_s_setprio(0);
      } else if constexpr (!F32) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) Mfma<T>::run(wf[j], af[i], acc[j][i]);
        __builtin_amdgcn_s_setprio(0);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i)
              part[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(reinterpret_cast<const float*>(&wf[j])[q],
                                                                reinterpret_cast<const float*>(&af[i])[q],
                                                                part[j][i], 0, 0, 0);
      }
    }
    if constexpr (F32) {          // two-level summation, as in conv.hip
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] += part[j][i];
    }
  };

  // ---- main loop over (segment, chunk), 9 or 1 k-steps each ------------------------------------------
  int cs = 0, cc = 0, kbase = 0;        // current chunk, first packed k-step of its segment
  load_patch(0, 0);
  {   // first weight tile straight to LDS (keeps wreg's only writer inside the loop)
    const T* wp = reinterpret_cast<const T*>(p.w) + (int64_t)n0 * KC;
#pragma unroll
    for (int j = 0; j < NWL; ++j) {
      const int q = tid + NTHR * j;
      const int row = q >> 3, ch = q & 7;
      if (q < BN * 8)
        *reinterpret_cast<uint4*>(s_w + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) =
            *reinterpret_cast<const uint4*>(wp + (int64_t)row * KC + ch * VW);
    }
  }
  build_gn_table();
  __syncthreads();
  write_patch(0);
  __syncthreads();
  int pb = 0, wb = 0;
  const bool lo_half = __builtin_amdgcn_readfirstlane(wave) < (WM * WN) / 2;
  while (cs < p.nseg) {
    const int nch = seg_nch(cs);
    const int ntaps = SEG_FIELD(p, cs, taps);
    // next chunk
    int ns = cs, nc = cc + 1, nkbase = kbase;
    if (nc == nch) { nc = 0; ns = cs + 1; nkbase = kbase + ntaps * nch; }
    const bool has_next = ns < p.nseg;
    const int nnch = has_next ? seg_nch(ns) : 1;
    (void)nnch;
    auto step = [&](auto KK) __attribute__((always_inline)) {
      constexpr int k = decltype(KK)::value;
      if (k < ntaps) {
        const bool last = (k == ntaps - 1);
        // Loads and LDS writes below are UNCONDITIONAL (on the very last chunk / step they fetch
        // the current data again into buffers nobody reads): conditionally defined register arrays
        // end up in scratch memory with this compiler.
        if constexpr (k == 0) load_patch(has_next ? ns : cs, has_next ? nc : cc);   // whole next patch -> registers
        // weight tile of the next k-step
        const int knext = last ? (has_next ? nkbase + nc : kbase + k * nch + cc) : kbase + (k + 1) * nch + cc;
        load_w(knext);
        const int dy = ntaps == 9 ? k / 3 : 1, dx = ntaps == 9 ? k % 3 : 1;
        // 9-tap chunk: patch item k-1 (loaded at tap 0) is transformed and stored during tap k,
        // next to that tap's MFMAs; 1-tap chunk: all items after the only tap
        // Waves w and w+4 (TH=16) / w and w+2 (TH=8) tend to share a SIMD and run in lockstep
        // between barriers: the first half does its VALU-heavy patch item BEFORE the first MFMA
        // half, the other half between the two MFMA halves, so one wave's VALU work lies beside
        // its partner's MFMAs instead of beside its VALU work.
        if constexpr (k >= 1) { if (lo_half) write_patch_item(std::integral_constant<int, k - 1>{}, pb ^ 1); }
        compute(pb, wb, dy, dx, 0);
        if constexpr (k >= 1) { if (!lo_half) write_patch_item(std::integral_constant<int, k - 1>{}, pb ^ 1); }
        compute(pb, wb, dy, dx, 1);
        if constexpr (k == 8) {            // (the two-wave tiles of the 4-channel heads: up to 12 items per thread)
          write_patch_item(std::integral_constant<int, 8>{}, pb ^ 1);
          write_patch_item(std::integral_constant<int, 9>{}, pb ^ 1);
          write_patch_item(std::integral_constant<int, 10>{}, pb ^ 1);
          write_patch_item(std::integral_constant<int, 11>{}, pb ^ 1);
        }
        if (ntaps != 9) write_patch(pb ^ 1);
        write_w(wb ^ 1);
        __syncthreads();
        wb ^= 1;
      }
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
    step(std::integral_constant<int, 8>{});
    pb ^= 1;
    cs = ns; cc = nc; kbase = nkbase;
  }

  // ---- epilogue -------------------------------------------------------------------------------------
  const int Cout = p.Cout;
  const bool do_stat = p.stat_out != nullptr;
  const int scpg = do_stat ? Cout / p.stat_G : 1;
  float a1[NT], a2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) a1[j] = a2[j] = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int y = y0 + wm * 4 + i, x = x0 + frow;
    const int64_t m = img + (int64_t)y * W + x;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * (NT * 16) + j * 16 + fk * 4;
      float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
      if constexpr (SPL) { v[0] *= p.acc_scale; v[1] *= p.acc_scale; v[2] *= p.acc_scale; v[3] *= p.acc_scale; }
      const bool live = n < Cout;
      if (live) conv_epilogue4<TO>(p, m, b, n, v);
      if (do_stat && live) {
        a1[j] += (v[0] + v[1]) + (v[2] + v[3]);
        a2[j] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      }
    }
  }
  if (do_stat) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * (NT * 16) + j * 16 + fk * 4;
      const float r1 = row16_sum(a1[j]), r2 = row16_sum(a2[j]);
      if (frow == 0 && n < Cout) {
        atomicAdd(&s_stat[((n - n0) / scpg) * 2], (double)r1);
        atomicAdd(&s_stat[((n - n0) / scpg) * 2 + 1], (double)r2);
      }
    }
    __syncthreads();
    const int g0 = n0 / scpg;
    const int ng = min(p.stat_G - g0, (BN + scpg - 1) / scpg);
    for (int i = tid; i < ng * 2; i += NTHR) {
      const int k = i & 1, g = g0 + (i >> 1);
      atomicAdd(p.stat_out + (((int64_t)b * p.stat_nsplit + tile % p.stat_nsplit) * p.stat_G + g) * 2 + k,
                s_stat[(g - g0) * 2 + k]);
    }
  }
}

template <typename T, typename TO, int TH, int BN, bool GNP, bool SPL = false>
static int launch_patch(const ConvParams& p, hipStream_t st) {
  constexpr int NTHR = 64 * (TH / 4) * (BN >= 64 ? BN / 64 : 1);
  constexpr int SMEM_MAX = 2 * (TH + 2) * 18 * 128 + 2 * BN * 128 + (GNP ? CONV_GN_MAXC * 8 : 0) + 64 * 8 + 64 * 4;
  // the GroupNorm table takes what the layer needs: with 8-row tiles and <= 256 normalised channels two
  // workgroups fit the 160 KiB of a CU
  const int SMEM = 2 * (TH + 2) * 18 * 128 + 2 * BN * 128 + (GNP ? ((p.gn_C + 63) & ~63) * 8 : 0) + 64 * 8 + 64 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_patch_kernel<T, TO, TH, BN, GNP, SPL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_MAX);
    if (e != hipSuccess) {
      fdbm_set_error("fdbm_conv_igemm(patch): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  const int tiles_x = p.W / 16, tiles_y = p.H / TH;
  dim3 grid((unsigned)(p.B * tiles_x * tiles_y), (unsigned)((p.Cout + BN - 1) / BN));
  conv_patch_kernel<T, TO, TH, BN, GNP, SPL><<<grid, NTHR, SMEM, st>>>(p, tiles_x, tiles_y);
  FDBM_LAUNCH_CHECK("fdbm_conv_igemm(patch)");
  return 0;
}

template <typename T, typename TO, bool SPL = false>
static int launch_patch_th(const ConvParams& p, int th, hipStream_t st) {
  const bool gnp = p.gn_sums != nullptr;
  if constexpr (sizeof(TO) == 4) {
    // the 4-channel heads (f32 output): 16-channel tiles, 8 rows (2 waves, 50 KiB of LDS: three workgroups share a CU
    // and the GroupNorm + SiLU of one's patch runs beside the others' MFMAs)
    if (p.Cout <= 16) return gnp ? launch_patch<T, TO, 8, 16, true, SPL>(p, st) : launch_patch<T, TO, 8, 16, false, SPL>(p, st);
  }
  if (th == 16) return gnp ? launch_patch<T, TO, 16, 128, true, SPL>(p, st) : launch_patch<T, TO, 16, 128, false, SPL>(p, st);
  return gnp ? launch_patch<T, TO, 8, 128, true, SPL>(p, st) : launch_patch<T, TO, 8, 128, false, SPL>(p, st);
}

// called from fdbm_conv_igemm (conv.hip) once it has validated the arguments and filled ConvParams
int fdbm_launch_conv_patch(const ConvParams& p, int dt_in, int dt_out, int th, hipStream_t st) {
  if (dt_in == FDBM_BF16 && dt_out == FDBM_BF16) return launch_patch_th<bf16_t, bf16_t>(p, th, st);
  if (dt_in == FDBM_BF16 && dt_out == FDBM_F32) return launch_patch_th<bf16_t, float>(p, th, st);
  if (dt_in == FDBM_F16 && dt_out == FDBM_F16) return launch_patch_th<f16_t, f16_t>(p, th, st);
  if (dt_in == FDBM_F16 && dt_out == FDBM_F32) return launch_patch_th<f16_t, float>(p, th, st);
  if (p.mma_split) return launch_patch_th<float, float, true>(p, th, st);
  return launch_patch_th<float, float>(p, th, st);
}
