// conv_tap.hip - 3x3 convolution for the small-batch, small-map layers: one wave per filter tap.
//
// At batch 1 the layers below the 256 x 256 level have too few output tiles to hide the serial
// chain "load k-step -> LDS -> MFMA" of an output-stationary kernel: the tap-outer kernel
// (conv.hip) spends 0.6-0.8 us per k-step on 36-72 k-steps with one workgroup per CU and needs a
// second launch for its split-K slabs.  Here the taps of a 3x3 kernel run CONCURRENTLY:
//   workgroup = 8 waves (2 per SIMD) = 16*MT pixels (one image) x 16*NT output channels
//   wave w    = tap w (dy = w/3, dx = w%3): for every 128-byte input-channel chunk it multiplies
//               the SAME halo patch in LDS (read at its own shift) with its own weight slice,
//               whose MFMA fragments it loads straight from global memory (no LDS, no barrier:
//               nobody else needs them);
//   tap 8     = the "shared" tap: its (m-tile, k-half) products are dealt out over the 8 waves
//               (wave w: m-tiles i with i%4 == w%4, k-half w/4) - any wave's accumulator will do;
//   the 8 partial accumulators are summed through LDS at the end (3 barriers), and the
//   epilogue (bias, time bias, residual, scale, Combine, output statistics) is spread over
//   the waves by output tile.
// The chain per workgroup is (#channel chunks) steps instead of 9 x that, every step carries
// 9 x the loads in flight, the patch is staged once per chunk (GroupNorm + SiLU prologue costs
// 1.4-2.3 x the tile instead of 9 x) and no slab / second launch is needed.
// 1-tap segments (the res-block's 1x1 shortcut in the same accumulator) follow the 9-tap ones
// and are handled like the shared tap (with the centre shift).
// Pipeline: D register sets (patch items + weight fragments of chunks c .. c+D-1; D = 4 on the one-n-tile tiles
// of the small maps, 2 on the wide tiles) behind ONE load cursor whose segment fields live in scalars; LDS patch
// double-buffered (chunk c+1 is transformed and written while chunk c computes); every global load in
// straight-line code so that hipcc counts its waits (vmcnt(13..15) in the loop, not vmcnt(0)).
// Tiles: TW = 16 | 8 | 4 pixels wide (16/TW image rows per 16-pixel MFMA m-tile), so that 8x8
// and 4x4 maps are one tile.  Roofline: latency-bound by design (these layers hold < 1 us of
// MFMA work per CU); it is selected only when the grid is small (fdbm_conv_plan_ex).
#include "conv_common.h"

#define TAP_NTHR 512
#define TAP_SLAB0 16384      // floats: the split-accumulation slabs start 64 KiB into the scratch (counters first)

// Diagnostic build only (-DFDBM_STAMPS, tools/tap_timeline.py): workgroup (0,0) writes shader-clock
// stamps of its phases into the (otherwise unused) workspace.  The product library has none of it.
#ifdef FDBM_STAMPS
#define STAMP(i)                                                                                   \
  do {                                                                                             \
    if (p.partial && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)                       \
      reinterpret_cast<unsigned long long*>(p.partial)[i] = __builtin_amdgcn_s_memtime();          \
  } while (0)
#define STAMP_RT(i)                                                                                \
  do {                                                                                             \
    if (p.partial && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)                       \
      reinterpret_cast<unsigned long long*>(p.partial)[i] = __builtin_amdgcn_s_memrealtime();      \
  } while (0)
#else
#define STAMP(i)
#define STAMP_RT(i)
#endif

// SPL (T = float only): split-precision matrix mode - patch and weight rows hold [32 halves hi | 32 halves lo]
// (conv_common.h), every product is three f16 MFMAs: hi.hi + hi.lo + lo.hi.
template <typename T, typename TO, int TW, int MT, int NT, bool GNP, bool SPL = false>
__global__ void __launch_bounds__(TAP_NTHR) conv_tap_kernel(const ConvParams p, int tiles_x, int tiles_y) {
  constexpr int KC = 128 / (int)sizeof(T);
  constexpr int VW = 16 / (int)sizeof(T);
  constexpr int RPM = 16 / TW;               // image rows per m-tile
  constexpr int TR = MT * RPM;               // tile rows
  constexpr int PCW = TW + 2;
  constexpr int PROWS = (TR + 2) * PCW;
  constexpr int PBUF = PROWS * 128;
  constexpr int NPL = (PROWS * 8 + TAP_NTHR - 1) / TAP_NTHR;
  constexpr int SLAB = MT * NT * 1024;       // one wave's accumulators: [MT*NT][64 lanes] f32x4
  constexpr int MAIN = (2 * PBUF > 4 * SLAB) ? 2 * PBUF : 4 * SLAB;
  constexpr bool F32 = sizeof(T) == 4 && !SPL;
  static_assert(!SPL || sizeof(T) == 4, "the split-precision mode stages f32 tensors");
  static_assert(NPL <= 4, "patch staging assumes at most 4 items per thread");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* s_patch = smem;                                         // 2 x PBUF (slabs alias it later)
  float* s_gn = reinterpret_cast<float*>(smem + MAIN);                   // scale[CONV_GN_MAXC] | shift[CONV_GN_MAXC]
  double* s_stat = reinterpret_cast<double*>(smem + MAIN + (GNP ? CONV_GN_MAXC * 8 : 0));  // [32][2] doubles
  float* s_mr = reinterpret_cast<float*>(s_stat + 64);                                    // [32][2] mean, rstd

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fk = lane >> 4;
  const int H = p.H, W = p.W;
  const int tile = blockIdx.x;
  const int tx = tile % tiles_x;
  const int ty = (tile / tiles_x) % tiles_y;
  const int b = tile / (tiles_x * tiles_y);
  const int y0 = ty * TR, x0 = tx * TW;
  const int n0 = blockIdx.y * (16 * NT);
  const int64_t img = (int64_t)b * H * W;

  // ---- per-thread patch items -------------------------------------------------------------------
  int ppix[NPL], plds[NPL];
  unsigned pmask = 0;
#pragma unroll
  for (int j = 0; j < NPL; ++j) {
    const int q = tid + TAP_NTHR * j;
    const int prow = min(q >> 3, PROWS - 1), pch = q & 7;
    const int pr = prow / PCW, pc = prow - pr * PCW;
    const int iy = y0 + pr - 1, ix = x0 + pc - 1;
    const bool ok = (q < PROWS * 8) && iy >= 0 && iy < H && ix >= 0 && ix < W;
    ppix[j] = min(max(iy, 0), H - 1) * W + min(max(ix, 0), W - 1);
    // (split mode: the item's 4 channels are 8 bytes of the hi plane; their lo halves sit 64 bytes on, address ^ 64)
    if constexpr (SPL) plds[j] = (q < PROWS * 8) ? prow * 128 + (((pch >> 1) ^ ((prow >> 1) & 7)) << 4) + (pch & 1) * 8 : -1;
    else plds[j] = (q < PROWS * 8) ? prow * 128 + ((pch ^ ((prow >> 1) & 7)) << 4) : -1;
    pmask |= ok ? (1u << j) : 0u;
  }
  const int pchunk = tid & 7;          // 512 % 8 == 0: the 16-byte chunk is fixed per thread

  // ---- this wave's activation rows in the patch: its own tap, and the centre tap -----------------
  const int dy = wave / 3, dx = wave - dy * 3;
  const int arow0 = (frow / TW + dy) * PCW + frow % TW + dx;     // m-tile i: + i * RPM * PCW
  const int crow0 = (frow / TW + 1) * PCW + frow % TW + 1;       // centre tap (1-tap segments)
  // (tap 8, the shared one, is PCW + 1 rows further: crow0 + PCW + 1)
  constexpr int MROW = RPM * PCW;

  if (p.stat_out)
    for (int i = tid; i < 64; i += TAP_NTHR) s_stat[i] = 0.0;

  auto build_gn_table = [&]() __attribute__((always_inline)) {
    if constexpr (GNP) conv_gn_table<TAP_NTHR>(p, b, 1, s_gn, CONV_GN_MAXC, s_mr, smem);   // (the patch buffers are still idle)
  };

  auto seg_nch = [&](int s) __attribute__((always_inline)) { return (SEG_FIELD(p, s, cin) + KC - 1) / KC; };

  // ---- register sets ----------------------------------------------------------------------------------
  // The loop is a chain of short chunks whose operands come from L2 / HBM (weights: read once per forward;
  // the patch: the previous kernel's output): a chunk can start no earlier than its loads land, so the chunk
  // time is (memory latency) / (chunks in flight).  D register sets hold the patch items and the weight
  // fragments of the chunks c .. c+D-1; set c % D is reloaded with chunk c+D as soon as chunk c has consumed
  // it.  D = 4 where a set is small (one n-tile: the small maps, where the chain is the whole run time),
  // D = 2 for the wide tiles, whose MFMA work per chunk covers the latency.
  // Named sets A-D selected by `if constexpr` (an array of sets would be indexed through scratch memory).
  constexpr int D = (NT == 1) ? 4 : 2;
#define SET_PUT(S, lhsA, lhsB, lhsC, lhsD, val)                                                   \
  do {                                                                                           \
    if constexpr ((S) == 0) lhsA = (val); else if constexpr ((S) == 1) lhsB = (val);              \
    else if constexpr ((S) == 2) lhsC = (val); else lhsD = (val);                                 \
  } while (0)
#define SET_GET(S, a, b, c, d) ((S) == 0 ? (a) : (S) == 1 ? (b) : (S) == 2 ? (c) : (d))
  [[maybe_unused]] uint4 pregA[NPL], pregB[NPL], pregC[NPL], pregD[NPL];
  [[maybe_unused]] int pgcbA = 0, pgcbB = 0, pgcbC = 0, pgcbD = 0;               // first GN table entry of the thread's 16 bytes
  [[maybe_unused]] bool pgnA = false, pgnB = false, pgnC = false, pgnD = false;  // wave-uniform: the segment is normalised
  [[maybe_unused]] bool pcokA = true, pcokB = true, pcokC = true, pcokD = true;  // channel chunk inside the segment
  [[maybe_unused]] bool t9A = true, t9B = true, t9C = true, t9D = true;          // wave-uniform: a 9-tap chunk
  // ---- load cursor: the next chunk to fetch ----------------------------------------------------------------
  // The fields of its segment live in scalars that change only when the cursor enters a new segment: p.seg[s]
  // with a run-time s is a chain of dependent scalar memory loads from the kernel argument (measured: ~0.5 us
  // of every chunk went into them).
  const void* L_src = p.seg[0].src;
  int L_C = p.seg[0].C, L_coff = p.seg[0].coff, L_cin = p.seg[0].cin, L_taps = p.seg[0].taps;
  int L_gn = GNP ? p.seg_gn[0] : -1;
  int L_nch = (L_cin + KC - 1) / KC;
  int L_s = 0, L_c = 0, L_kb = 0;
  auto enter_seg = [&](int sn) __attribute__((always_inline)) {
    L_src = SEG_FIELD(p, sn, src);
    L_C = SEG_FIELD(p, sn, C); L_coff = SEG_FIELD(p, sn, coff); L_cin = SEG_FIELD(p, sn, cin); L_taps = SEG_FIELD(p, sn, taps);
    if constexpr (GNP) L_gn = sn == 0 ? p.seg_gn[0] : sn == 1 ? p.seg_gn[1] : sn == 2 ? p.seg_gn[2] : p.seg_gn[3];
    L_nch = (L_cin + KC - 1) / KC;
  };
  auto advance = [&]() __attribute__((always_inline)) {          // (stays on the conv's last chunk)
    if (L_c + 1 < L_nch) { ++L_c; }
    else if (L_s + 1 < p.nseg) { L_kb += L_taps * L_nch; ++L_s; L_c = 0; enter_seg(L_s); }
  };
  auto load_patch = [&](auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const int cvalid = min(KC, L_cin - L_c * KC);
    const bool ok = pchunk * VW < cvalid;
    const T* src = reinterpret_cast<const T*>(L_src) + img * L_C + L_coff + (ok ? L_c * KC + pchunk * VW : 0);
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const uint4 v = *reinterpret_cast<const uint4*>(src + (int64_t)ppix[j] * L_C);
      SET_PUT(S, pregA[j], pregB[j], pregC[j], pregD[j], v);
    }
    SET_PUT(S, pcokA, pcokB, pcokC, pcokD, ok);
    SET_PUT(S, pgcbA, pgcbB, pgcbC, pgcbD, max(L_gn, 0) + L_c * KC + pchunk * VW);
    SET_PUT(S, pgnA, pgnB, pgnC, pgnD, L_gn >= 0);
    SET_PUT(S, t9A, t9B, t9C, t9D, L_taps == 9);
  };
  // transform (GroupNorm scale/shift + SiLU) and store ONE patch item: issued between the MFMA groups of
  // a chunk so that its VALU work runs beside them
  auto write_patch_item = [&](auto SET, auto JJ, int buf) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    constexpr int j = decltype(JJ)::value;
    if constexpr (j < NPL) {
      unsigned char* P = s_patch + buf * PBUF;
      const int pgcb = SET_GET(S, pgcbA, pgcbB, pgcbC, pgcbD);
      const bool pgn = SET_GET(S, pgnA, pgnB, pgnC, pgnD);
      const bool pcok = SET_GET(S, pcokA, pcokB, pcokC, pcokD);
      uint4 v = SET_GET(S, pregA[j], pregB[j], pregC[j], pregD[j]);
      if constexpr (GNP) {
        if (pgn)                           // wave-uniform: a property of the segment (scalar branch)
        {
          if constexpr (SPL) {
            if (p.gn_silu == 2) v = gn_transform16<T, true>(v, s_gn + pgcb, s_gn + CONV_GN_MAXC + pgcb, true);
            else v = gn_transform16<T>(v, s_gn + pgcb, s_gn + CONV_GN_MAXC + pgcb, p.gn_silu != 0);
          } else {
            v = gn_transform16<T>(v, s_gn + pgcb, s_gn + CONV_GN_MAXC + pgcb, p.gn_silu != 0);
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
  auto write_patch = [&](auto SET, int buf) __attribute__((always_inline)) {
    write_patch_item(SET, std::integral_constant<int, 0>{}, buf);
    write_patch_item(SET, std::integral_constant<int, 1>{}, buf);
    write_patch_item(SET, std::integral_constant<int, 2>{}, buf);
    write_patch_item(SET, std::integral_constant<int, 3>{}, buf);
  };

  // ---- weight fragments: straight from global memory into MFMA operand registers -------------------
  // fragment-major weights [kidx][CoutPad/16][8 chunks][16 rows][16 B]: lane (frow, fk) of n-tile j,
  // k-half kk reads (n-tile n0/16 + j, chunk 4 kk + fk, row frow) - the 64 lanes of a wave load
  // 1 KiB contiguous (with row-major [CoutPad][128 B] weights consecutive lanes would sit 128 B apart
  // and every lane would be its own memory transaction: measured 3.2 us per channel chunk).
  // w?0 / w?1: the two k-halves of this wave's tap; w?s: this wave's k-half of the shared tap
  // (tap 8, or the only tap of a 1-tap segment).
  [[maybe_unused]] uint4 wA0[NT], wA1[NT], wAs[NT], wB0[NT], wB1[NT], wBs[NT], wC0[NT], wC1[NT], wCs[NT], wD0[NT], wD1[NT], wDs[NT];
  const int kkw = (wave >> 2) & 1;          // shared tap: this wave's k-half ...
  const int miw = wave & 3;                 // ... and m-tiles i with (i & 3) == miw
  auto w_ptr = [&](bool shared) __attribute__((always_inline)) {          // the cursor chunk's fragments
    const int tap = L_taps == 9 ? (shared ? 8 : wave) : 0;
    const int kidx = L_kb + tap * L_nch + L_c;
    return reinterpret_cast<const T*>(p.w) + ((int64_t)kidx * p.CoutPad + n0) * KC +
           (((shared ? kkw * 4 : 0) + fk) * 16 + frow) * VW;
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto mma = [&](const uint4& wf, const uint4& af, f32x4& c) __attribute__((always_inline)) {
    if constexpr (SPL) {
      Mfma<f16_t>::run(wf, af, c);
    } else if constexpr (!F32) {
      Mfma<T>::run(wf, af, c);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(reinterpret_cast<const float*>(&wf)[q],
                                                 reinterpret_cast<const float*>(&af)[q], c, 0, 0, 0);
    }
  };
  auto a_frag = [&](const unsigned char* P, int row, int kk) __attribute__((always_inline)) {
    const int cidx = kk * 4 + fk;
    return *reinterpret_cast<const uint4*>(P + row * 128 + ((cidx ^ ((row >> 1) & 7)) << 4));
  };

  // ---- main loop over (segment, channel chunk) ----------------------------------------------------------
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;
  auto load_w = [&](auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const T* wp = w_ptr(false);
    const T* wq = w_ptr(true);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const uint4 v0 = *reinterpret_cast<const uint4*>(wp + (int64_t)j * 16 * KC);
      const uint4 v1 = *reinterpret_cast<const uint4*>(wp + (int64_t)j * 16 * KC + 64 * VW);
      const uint4 vs = *reinterpret_cast<const uint4*>(wq + (int64_t)j * 16 * KC);
      SET_PUT(S, wA0[j], wB0[j], wC0[j], wD0[j], v0);
      SET_PUT(S, wA1[j], wB1[j], wC1[j], wD1[j], v1);
      SET_PUT(S, wAs[j], wBs[j], wCs[j], wDs[j], vs);
    }
  };
  // split over workgroups (p.ksplit > 1, grid.z): this workgroup owns chunks [zb, ze) of the conv and
  // writes its partial tile into slab blockIdx.z of the scratch p.partial; the workgroup that finishes
  // last sums the slabs in slice order and applies the epilogue.  For maps so small that the tiles
  // alone occupy a fraction of the chip and the chain over the chunks is the whole run time.
  int ntotal = 0;
  for (int s_ = 0; s_ < p.nseg; ++s_) ntotal += seg_nch(s_);
  const int zper = (ntotal + p.ksplit - 1) / p.ksplit;
  const int zraw = (int)blockIdx.z * zper;
  const int nmine = max(min(ntotal, zraw + zper) - zraw, 0);   // (a trailing workgroup may own nothing: it adds zeros)
  const int zb = min(zraw, ntotal - 1);
  for (int i = 0; i < zb; ++i) advance();
  // The cursor stops on this workgroup's last chunk: loads are unconditional (a load under a branch would make
  // every later wait in the loop conservative), the surplus ones fetch that chunk again - cache hits whose
  // results nobody consumes.
  int lidx = 0;
  auto next_load = [&]() __attribute__((always_inline)) {
    if (lidx + 1 < nmine) { advance(); ++lidx; }
  };
  STAMP_RT(60);
  STAMP(0);
  load_patch(S0{}); load_w(S0{}); next_load();
  load_patch(S1{}); load_w(S1{}); next_load();
  if constexpr (D == 4) {
    load_patch(S2{}); load_w(S2{}); next_load();
    load_patch(S3{}); load_w(S3{}); next_load();
  }
  STAMP(1);
  build_gn_table();
  __syncthreads();
  STAMP(2);
  write_patch(S0{}, 0);
  __syncthreads();
  STAMP(3);
  int nstamp = 4;
  (void)nstamp;

  // one chunk: SET = its register set (chunk index mod D); LDS buffer SET & 1 holds its patch.
  // Every global load of the loop sits in straight-line code: a load inside one arm of an if / else makes the
  // compiler wait for ALL outstanding loads (s_waitcnt vmcnt(0)) wherever the other arm reuses its destination
  // registers.  Only LDS reads, MFMAs and the GroupNorm arithmetic are conditional.
  auto chunk = [&](auto SET) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    using NEXT = std::integral_constant<int, (S + 1) % D>;
    constexpr int LB = S & 1;
    const bool t9 = SET_GET(S, t9A, t9B, t9C, t9D);
    // this set's next tenant: chunk c+D (the patch registers are free since the last barrier; the weight
    // fragments are requested right behind the MFMAs that consumed the register)
    const T* wp = w_ptr(false);
    const T* wq = w_ptr(true);
    load_patch(SET);
    const unsigned char* P = s_patch + LB * PBUF;
    if constexpr (NT == 1) {
      // all fragment reads of the chunk first (one LDS round trip instead of three), then its MFMAs, then the
      // reloads, then the next patch's items
      const int srow0 = crow0 + (t9 ? PCW + 1 : 0);       // shared tap: tap 8 of a 9-tap chunk, or the only tap of a 1-tap chunk
      // (split mode: the k-half-0 waves multiply w_hi by a_hi and a_lo, the k-half-1 waves w_lo by a_hi)
      const uint4 afs = a_frag(P, srow0 + (MT >= 4 ? miw : 0) * MROW, SPL ? 0 : kkw);
      [[maybe_unused]] uint4 afs2 = afs;
      if constexpr (SPL) afs2 = a_frag(P, srow0 + (MT >= 4 ? miw : 0) * MROW, 1);
      if (t9) {
        uint4 af0[MT < 2 ? 2 : MT], af1[MT < 2 ? 2 : MT];         // (not [1]: one-element arrays end up in scratch memory)
#pragma unroll
        for (int i = 0; i < MT; ++i) { af0[i] = a_frag(P, arow0 + i * MROW, 0); af1[i] = a_frag(P, arow0 + i * MROW, 1); }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) mma(SET_GET(S, wA0[j], wB0[j], wC0[j], wD0[j]), af0[i], acc[j][i]);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            if constexpr (SPL) {       // hi.hi above; w_hi.a_lo and w_lo.a_hi here
              mma(SET_GET(S, wA0[j], wB0[j], wC0[j], wD0[j]), af1[i], acc[j][i]);
              mma(SET_GET(S, wA1[j], wB1[j], wC1[j], wD1[j]), af0[i], acc[j][i]);
            } else {
              mma(SET_GET(S, wA1[j], wB1[j], wC1[j], wD1[j]), af1[i], acc[j][i]);
            }
          }
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if ((i & 3) == miw) {
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            mma(SET_GET(S, wAs[j], wBs[j], wCs[j], wDs[j]), afs, acc[j][i]);
            if constexpr (SPL) { if (kkw == 0) mma(SET_GET(S, wAs[j], wBs[j], wCs[j], wDs[j]), afs2, acc[j][i]); }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {          // (1-tap chunks do not use the own-tap registers; they stay loaded)
        const uint4 v0 = *reinterpret_cast<const uint4*>(wp + (int64_t)j * 16 * KC);
        const uint4 v1 = *reinterpret_cast<const uint4*>(wp + (int64_t)j * 16 * KC + 64 * VW);
        const uint4 vs = *reinterpret_cast<const uint4*>(wq + (int64_t)j * 16 * KC);
        SET_PUT(S, wA0[j], wB0[j], wC0[j], wD0[j], v0);
        SET_PUT(S, wA1[j], wB1[j], wC1[j], wD1[j], v1);
        SET_PUT(S, wAs[j], wBs[j], wCs[j], wDs[j], vs);
      }
      STAMP(nstamp + 32);
      write_patch_item(NEXT{}, std::integral_constant<int, 0>{}, 1 - LB);    // patch of chunk c+1
      write_patch_item(NEXT{}, std::integral_constant<int, 1>{}, 1 - LB);
      write_patch_item(NEXT{}, std::integral_constant<int, 2>{}, 1 - LB);
      write_patch_item(NEXT{}, std::integral_constant<int, 3>{}, 1 - LB);
    } else {
      // wide tiles: the next patch's items between the MFMA groups, whose length covers the VALU work of the
      // partner wave on the SIMD
      if (t9) {
        uint4 af[MT < 2 ? 2 : MT];         // (not [1]: one-element arrays end up in scratch memory)
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = a_frag(P, arow0 + i * MROW, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) mma(SET_GET(S, wA0[j], wB0[j], wC0[j], wD0[j]), af[i], acc[j][i]);
        if constexpr (SPL) {               // w_hi . a_lo while w_hi is still in its registers
#pragma unroll
          for (int i = 0; i < MT; ++i) af[i] = a_frag(P, arow0 + i * MROW, 1);
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) mma(SET_GET(S, wA0[j], wB0[j], wC0[j], wD0[j]), af[i], acc[j][i]);
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {          // (1-tap chunks do not use the own-tap registers; they stay loaded)
        const uint4 v = *reinterpret_cast<const uint4*>(wp + (int64_t)j * 16 * KC);
        SET_PUT(S, wA0[j], wB0[j], wC0[j], wD0[j], v);
      }
      write_patch_item(NEXT{}, std::integral_constant<int, 0>{}, 1 - LB);    // patch of chunk c+1, item 0
      if (t9) {
        uint4 af[MT < 2 ? 2 : MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = a_frag(P, arow0 + i * MROW, SPL ? 0 : 1);      // (split mode: w_lo . a_hi)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) mma(SET_GET(S, wA1[j], wB1[j], wC1[j], wD1[j]), af[i], acc[j][i]);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const uint4 v = *reinterpret_cast<const uint4*>(wp + (int64_t)j * 16 * KC + 64 * VW);
        SET_PUT(S, wA1[j], wB1[j], wC1[j], wD1[j], v);
      }
      write_patch_item(NEXT{}, std::integral_constant<int, 1>{}, 1 - LB);
      {   // shared tap: tap 8 of a 9-tap chunk, or the only tap of a 1-tap chunk
        const int srow0 = crow0 + (t9 ? PCW + 1 : 0);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if ((i & 3) == miw) {
            const uint4 af = a_frag(P, srow0 + i * MROW, SPL ? 0 : kkw);
#pragma unroll
            for (int j = 0; j < NT; ++j) mma(SET_GET(S, wAs[j], wBs[j], wCs[j], wDs[j]), af, acc[j][i]);
            if constexpr (SPL) {
              if (kkw == 0) {
                const uint4 af2 = a_frag(P, srow0 + i * MROW, 1);
#pragma unroll
                for (int j = 0; j < NT; ++j) mma(SET_GET(S, wAs[j], wBs[j], wCs[j], wDs[j]), af2, acc[j][i]);
              }
            }
          }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const uint4 v = *reinterpret_cast<const uint4*>(wq + (int64_t)j * 16 * KC);
          SET_PUT(S, wAs[j], wBs[j], wCs[j], wDs[j], v);
        }
      }
      STAMP(nstamp + 32);
      write_patch_item(NEXT{}, std::integral_constant<int, 2>{}, 1 - LB);
      write_patch_item(NEXT{}, std::integral_constant<int, 3>{}, 1 - LB);
    }
    next_load();
    __syncthreads();
    STAMP(nstamp);
    ++nstamp;
  };
  // whole groups of D chunks run unconditionally (waits counted exactly across the loop), the last 1 .. D-1 behind
  {
    int c = 0;
    for (; c + D <= nmine; c += D) {
      chunk(S0{});
      chunk(S1{});
      if constexpr (D == 4) { chunk(S2{}); chunk(S3{}); }
    }
    const int rem = nmine - c;
    if (rem > 0) chunk(S0{});
    if constexpr (D == 4) {
      if (rem > 1) chunk(S1{});
      if (rem > 2) chunk(S2{});
    }
  }
#undef SET_PUT
#undef SET_GET

  // ---- sum the 8 waves: 4..7 -> 0..3 through slabs, then slabs 0..3 are summed by output tile -------
  f32x4* red = reinterpret_cast<f32x4*>(smem);
  constexpr int SL4 = SLAB / 16;            // f32x4 per slab
  if (wave >= 4) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i) red[(wave - 4) * SL4 + (j * MT + i) * 64 + lane] = acc[j][i];
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[j][i] += red[wave * SL4 + (j * MT + i) * 64 + lane];
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i) red[wave * SL4 + (j * MT + i) * 64 + lane] = acc[j][i];
  }
  __syncthreads();

  STAMP(28);
  const int Cout = p.Cout;
  const bool split = p.ksplit > 1;
  const int64_t slab = (int64_t)p.B * H * W * Cout;          // floats per workgroup-slice slab
  if (split) {
    // partial tile -> this slice's slab in the scratch, then count this workgroup in.  Writes and reads of the
    // slabs are atomic read-modify-writes (exchange / add 0): those are performed AT the device coherence
    // point, whereas plain (even sc1) loads hit stale lines of an earlier launch in this XCD's L2 unless an
    // agent-scope acquire invalidates it (measured: garbage) - and a release would write the whole L2 back.
    float* mine = p.partial + TAP_SLAB0 + (int64_t)blockIdx.z * slab;
    for (int t = wave; t < MT * NT; t += 8) {
      const int j = t / MT, i = t - j * MT;
      const f32x4 s = (red[t * 64 + lane] + red[SL4 + t * 64 + lane]) + (red[2 * SL4 + t * 64 + lane] + red[3 * SL4 + t * 64 + lane]);
      const int y = y0 + i * RPM + frow / TW, x = x0 + frow % TW;
      const int64_t m = img + (int64_t)y * W + x;
      const int n = n0 + j * 16 + fk * 4;
      if (n < Cout) {
        float* dst = mine + m * Cout + n;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = __hip_atomic_exchange(dst + r, s[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));     // returning: completion is what the wait below sees
      }
    }
    // No agent-scope fence here: a fence would write back this XCD's whole L2 (the previous kernels' output,
    // tens of microseconds).  But the exchanges of THIS workgroup must all have completed before it
    // counts itself in: an explicit s_waitcnt on every counter (a workgroup-scope fence / __syncthreads does
    // not wait for vmcnt outside threadgroup-split mode - without this wait the last workgroup can read a slab
    // before its owner's stores have landed, seen only when the GPU is shared between processes).
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    // (arrival counters live in the first 64 KiB of the scratch, which no launch ever uses for slabs: they
    // are zero whenever no launch is in flight)
    int* counter = reinterpret_cast<int*>(p.partial) + blockIdx.x * gridDim.y + blockIdx.y;
    int* s_flag = reinterpret_cast<int*>(s_mr + 64);
    if (tid == 0) {
      const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *s_flag = (old == p.ksplit - 1);
      if (old == p.ksplit - 1) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    if (!*s_flag) return;                   // (whole workgroup)
  }
  // ---- epilogue, output tile t = j * MT + i handled by wave t mod 8 -----------------------------------------
  const bool do_stat = p.stat_out != nullptr;
  const int scpg = do_stat ? Cout / p.stat_G : 1;
  for (int t = wave; t < MT * NT; t += 8) {
    const int j = t / MT, i = t - j * MT;
    f32x4 s = (red[t * 64 + lane] + red[SL4 + t * 64 + lane]) + (red[2 * SL4 + t * 64 + lane] + red[3 * SL4 + t * 64 + lane]);
    const int y = y0 + i * RPM + frow / TW, x = x0 + frow % TW;
    const int64_t m = img + (int64_t)y * W + x;
    const int n = n0 + j * 16 + fk * 4;
    if (split && n < Cout) {                // all slices' partial tiles, summed in slice order (run-to-run identical)
      const float* src = p.partial + TAP_SLAB0 + m * Cout + n;
      s = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int z = 0; z < p.ksplit; ++z)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          s[r] += __hip_atomic_fetch_add(const_cast<float*>(src) + (int64_t)z * slab + r, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (SPL) s *= p.acc_scale;
    float v[4] = {s[0], s[1], s[2], s[3]};
    const bool live = n < Cout;
    if (live) conv_epilogue4<TO>(p, m, b, n, v);
    if (do_stat) {
      const float s1 = live ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
      const float s2 = live ? (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]) : 0.f;
      const float r1 = row16_sum(s1), r2 = row16_sum(s2);
      if (frow == 0 && live) {
        atomicAdd(&s_stat[((n - n0) / scpg) * 2], (double)r1);
        atomicAdd(&s_stat[((n - n0) / scpg) * 2 + 1], (double)r2);
      }
    }
  }
  STAMP(29);
  STAMP_RT(61);
  if (do_stat) {
    __syncthreads();
    const int g0 = n0 / scpg;
    const int ng = min(p.stat_G - g0, (16 * NT + scpg - 1) / scpg);
    for (int i = tid; i < ng * 2; i += TAP_NTHR) {
      const int k = i & 1, g = g0 + (i >> 1);
      atomicAdd(p.stat_out + (((int64_t)b * p.stat_nsplit + tile % p.stat_nsplit) * p.stat_G + g) * 2 + k,
                s_stat[(g - g0) * 2 + k]);
    }
  }
}

template <typename T, typename TO, int TW, int MT, int NT, bool GNP, bool SPL = false>
static int launch_tap(const ConvParams& p, hipStream_t st) {
  constexpr int RPM = 16 / TW, TR = MT * RPM;
  constexpr int PBUF = (TR + 2) * (TW + 2) * 128;
  constexpr int SLAB = MT * NT * 1024;
  constexpr int MAIN = (2 * PBUF > 4 * SLAB) ? 2 * PBUF : 4 * SLAB;
  constexpr int SMEM = MAIN + (GNP ? CONV_GN_MAXC * 8 : 0) + 64 * 8 + 64 * 4 + 16;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tap_kernel<T, TO, TW, MT, NT, GNP, SPL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) {
      fdbm_set_error("fdbm_conv_igemm(tap): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  const int tiles_x = p.W / TW, tiles_y = p.H / TR;
  dim3 grid((unsigned)(p.B * tiles_x * tiles_y), (unsigned)((p.Cout + 16 * NT - 1) / (16 * NT)), (unsigned)p.ksplit);
  conv_tap_kernel<T, TO, TW, MT, NT, GNP, SPL><<<grid, TAP_NTHR, SMEM, st>>>(p, tiles_x, tiles_y);
  FDBM_LAUNCH_CHECK("fdbm_conv_igemm(tap)");
  return 0;
}

template <typename T, typename TO, int TW, int MT, bool SPL>
static int launch_tap_nt(const ConvParams& p, int nt, hipStream_t st) {
  const bool gnp = p.gn_sums != nullptr;
  if (nt == 4) return gnp ? launch_tap<T, TO, TW, MT, 4, true, SPL>(p, st) : launch_tap<T, TO, TW, MT, 4, false, SPL>(p, st);
  if (nt == 2) return gnp ? launch_tap<T, TO, TW, MT, 2, true, SPL>(p, st) : launch_tap<T, TO, TW, MT, 2, false, SPL>(p, st);
  return gnp ? launch_tap<T, TO, TW, MT, 1, true, SPL>(p, st) : launch_tap<T, TO, TW, MT, 1, false, SPL>(p, st);
}

template <typename T, typename TO, bool SPL = false>
static int launch_tap_shape(const ConvParams& p, int tw, int nt, hipStream_t st) {
  if (tw == 16) return launch_tap_nt<T, TO, 16, 4, SPL>(p, nt, st);
  if (tw == 8) return launch_tap_nt<T, TO, 8, 4, SPL>(p, nt, st);
  return launch_tap_nt<T, TO, 4, 1, SPL>(p, nt, st);
}

// called from fdbm_conv_igemm (conv.hip) with validated arguments; tw = tile width, nt = n-tiles
int fdbm_launch_conv_tap(const ConvParams& p, int dt_in, int dt_out, int tw, int nt, hipStream_t st) {
  if (dt_in == FDBM_BF16 && dt_out == FDBM_BF16) return launch_tap_shape<bf16_t, bf16_t>(p, tw, nt, st);
  if (dt_in == FDBM_BF16 && dt_out == FDBM_F32) return launch_tap_shape<bf16_t, float>(p, tw, nt, st);
  if (dt_in == FDBM_F16 && dt_out == FDBM_F16) return launch_tap_shape<f16_t, f16_t>(p, tw, nt, st);
  if (dt_in == FDBM_F16 && dt_out == FDBM_F32) return launch_tap_shape<f16_t, float>(p, tw, nt, st);
  if (p.mma_split) return launch_tap_shape<float, float, true>(p, tw, nt, st);
  return launch_tap_shape<float, float>(p, tw, nt, st);
}
