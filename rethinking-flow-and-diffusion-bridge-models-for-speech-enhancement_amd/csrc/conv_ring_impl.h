#pragma once
// conv_ring_impl.h - 3x3 convolution for the large feature maps, producer / consumer waves around an LDS ring
// (the kernel template; conv_ring.hip instantiates the 16-row tile, conv_ring8.hip the 8-row tile).
//
// Same arithmetic as conv_patch.hip (a 16 x 16 pixel tile of one image x 128 output channels per workgroup, the
// (16+2) x 18 halo patch of a 64-channel chunk staged ONCE in LDS and read at 9 shifted positions, weights = MFMA A
// operand, activations = B operand, 128-byte K-contiguous LDS rows with XOR-swizzled 16-byte chunks), but the
// workgroup's 8 waves have fixed roles:
//   waves 0-3  CONSUMERS, one per SIMD: 8 patch rows x 16 px x 64 channels each (8 x 4 MFMA tiles, 128 accumulator
//              registers).  Their loop holds nothing but ds_read_b128 and v_mfma: the fragments of the next half
//              k-step are read while the 32 MFMAs of the current one issue, so the matrix pipe of a SIMD is fed
//              by ONE wave without gaps.
//   waves 4-7  PRODUCERS, one per SIMD: stream the 16 KiB weight tile of every k-step into a 3-slot LDS ring (requested
//              into registers 3 k-steps before it is written) and stage the next chunk's halo patch into the other
//              patch buffer, GroupNorm scale/shift + SiLU applied on the way (VALU work that runs beside the
//              consumer's MFMAs on the same SIMD: the two pipes are separate).
// One raw s_barrier per k-step ("tick", in the MIDDLE of the consumers' step) is the only synchronisation:
//   tick(s): producers have written (lgkmcnt(0)) the weights of step s+1 and, at a chunk's last step, the next
//            patch; consumers have completed every LDS read of the chunk at its last step.
//   after tick(s) consumers read the first half of step s+1, producers overwrite ring slot (s+2) % 3 = (s-1) % 3
//   and, at a chunk boundary, the patch buffer of the chunk that has just ended.
// Roofline: MFMA-bound; 2*M*Cout*K flop per launch, K = 9*Cin (+ Cin of 1-tap shortcut segments).
// Segment order expected by this kernel: 9-tap segments first, then 1-tap ones; GroupNorm only on 9-tap segments
// (fdbm_conv_igemm falls back to conv_patch.hip otherwise).
#include <stdlib.h>

#include "conv_common.h"

// cache policy of the epilogue's output stores (raw buffer store `aux`: 0 = default write-back, 2 = nt, 16 = sc1
// write-through).  Experiments: tools/build_store_variants.sh.
#ifndef FDBM_RING_STORE_AUX
#define FDBM_RING_STORE_AUX 0
#endif

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// Tile geometry: R image rows x 16 pixels x 128 output channels per workgroup.  R = 16 is the throughput shape (one
// tile per CU at 256 x 256 and batch 1, many at larger batches); R = 8 halves a workgroup's work for the maps that offer
// fewer than a tile per CU at R = 16 (128 x 128 and 64 x 64 at batch 1): twice the workgroups, half the patch to stage
// before the first MFMA, half the k-loop - these launches are latency-bound, not MFMA-bound.
template <int R>
struct RingGeom {
  static constexpr int PC = 18;                          // patch columns (16 + halo)
  static constexpr int PROWS = (R + 2) * PC;             // patch pixels
  static constexpr int PB = ((PROWS + 7) / 8 * 8) * 128; // bytes of one patch buffer
  static constexpr int WB = 128 * 128;                   // bytes of one weight tile (128 output channels x 128 bytes)
  static constexpr int NSLOT = 3;
  static constexpr int WOFF = 2 * PB;
  static constexpr int GOFF = WOFF + NSLOT * WB;
  static constexpr int NIT = (PROWS * 8 + 255) / 256;    // patch items (16 bytes) per producer thread
  static constexpr int NI0 = (PROWS * 8 + 511) / 512;    // ... per thread when all 512 stage patch 0
  static constexpr int RW = R / 2;                       // image rows per consumer wave
  static constexpr int NTP = R * 16 * 8 / 256;           // interior items per producer thread (1-tap chunks)
  static constexpr int L7 = NIT < 8 ? NIT : 8;           // items requested in interval 7 (the rest in interval 8)
  // cumulative item schedule of a 9-tap chunk: XS(t) items have been transformed before interval t
  // (11 items: 2,1,1,2,1,1,2,1 over intervals 0..7; up to 8: one per interval)
  static constexpr int XS(int t) { return t >= 8 ? NIT : NIT == 11 ? (t * 11 + 5) / 8 : (t < NIT ? t : NIT); }
};

// f(integral_constant<int, A>) ... f(integral_constant<int, B - 1>)
template <int A, int B, typename F>
__device__ __forceinline__ void ring_static_for(F&& f) {
  if constexpr (A < B) {
    f(std::integral_constant<int, A>{});
    ring_static_for<A + 1, B>(f);
  }
}


// Diagnostic build only (-DFDBM_STAMPS, tools/ring_timeline.py): workgroup (0,0) writes shader-clock stamps of its
// phases into the workspace (consumer wave 0: slots 0.., producer wave 4: slots 32..).  The product library has none.
#ifdef FDBM_STAMPS
#define RSTAMP(i, T0)                                                                              \
  do {                                                                                             \
    if (p.partial && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == (T0))                    \
      reinterpret_cast<unsigned long long*>(p.partial)[i] = __builtin_amdgcn_s_memtime();          \
  } while (0)
#define RSTAMP_RT(i, T0)                                                                           \
  do {                                                                                             \
    if (p.partial && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == (T0))                    \
      reinterpret_cast<unsigned long long*>(p.partial)[i] = __builtin_amdgcn_s_memrealtime();      \
  } while (0)
#define RDBG(bit) ((p.ksplit >> (bit)) & 1)      // experiments: p.ksplit carries FDBM_RING_DBG (1: no MFMA, 2: no transform, 4: no setprio)
#else
#define RSTAMP(i, T0)
#define RSTAMP_RT(i, T0)
#define RDBG(bit) 0
#endif

template <typename T, typename TO, bool GNP, int R>
__global__ void __launch_bounds__(512) conv_ring_kernel(const ConvParams p, int tiles_x, int tiles_y, int wgs_per_image) {
  using G_ = RingGeom<R>;
  constexpr int PC = G_::PC, PROWS = G_::PROWS, PB = G_::PB, WB = G_::WB, NSLOT = G_::NSLOT, WOFF = G_::WOFF, GOFF = G_::GOFF;
  constexpr int NIT = G_::NIT, NI0 = G_::NI0, RW = G_::RW, NTP = G_::NTP, L7 = G_::L7;
  constexpr int KC = 64, VW = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int gnpad = GNP ? ((p.gn_C + 63) & ~63) : 0;
  float* s_gn = reinterpret_cast<float*>(smem + GOFF);                          // scale[gnpad] | shift[gnpad]
  double* s_stat = reinterpret_cast<double*>(smem + GOFF + gnpad * 8);          // [32][2]
  float* s_mr = reinterpret_cast<float*>(s_stat + 64);                          // [32][2] mean, rstd
  float* s_cbt = s_mr + 64;                                                     // [128] bias + time-embedding bias of this (image, channel block)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W;
  // PERSISTENT over the tiles of ONE image: workgroup (b, wslot) walks tiles wslot, wslot + wgs_per_image, ... of image b.
  // The GroupNorm table (a property of the image) is built once; from the second tile on the prologue of a tile - its
  // first patch, its first weight tiles - is staged by the producers while the consumers still multiply the tile before,
  // and the tile's stores drain while the next one is computed.
  const int tiles_per_image = tiles_x * tiles_y;
  const int b = blockIdx.x / wgs_per_image;
  const int wslot = blockIdx.x - b * wgs_per_image;
  const int tile0 = wslot;
  const int n0 = blockIdx.y * 128;
  const int64_t img = (int64_t)b * H * W;
  const int y0 = (tile0 / tiles_x) * R, x0 = (tile0 % tiles_x) * 16;        // the first tile (patch 0 by all threads)


  // segment table in scalar registers, read from the kernel argument ONCE: a scalar load inside the k-loop shares
  // the LDS reads' counter (lgkmcnt) and, returning out of order, turns every counted wait of the loop into a
  // wait for everything
  int sg_t0 = p.seg[0].taps, sg_t1 = p.seg[1].taps, sg_t2 = p.seg[2].taps, sg_t3 = p.seg[3].taps;
  int sg_n0 = (p.seg[0].cin + KC - 1) / KC, sg_n1 = (p.seg[1].cin + KC - 1) / KC, sg_n2 = (p.seg[2].cin + KC - 1) / KC,
      sg_n3 = (p.seg[3].cin + KC - 1) / KC;
  int nseg = p.nseg, nsteps = p.nk;
  asm volatile("" : "+s"(sg_t0), "+s"(sg_t1), "+s"(sg_t2), "+s"(sg_t3), "+s"(sg_n0), "+s"(sg_n1), "+s"(sg_n2), "+s"(sg_n3),
               "+s"(nseg), "+s"(nsteps));
  auto seg_nch = [&](int s) __attribute__((always_inline)) { return s == 0 ? sg_n0 : s == 1 ? sg_n1 : s == 2 ? sg_n2 : sg_n3; };
  auto seg_taps = [&](int s) __attribute__((always_inline)) { return s == 0 ? sg_t0 : s == 1 ? sg_t1 : s == 2 ? sg_t2 : sg_t3; };

  // ---- patch 0 through registers by ALL 512 threads: item j = patch row (tid / 8 + 64 j), 16-byte chunk tid & 7.
  // The loads are requested here, ahead of the GroupNorm table they may need, and transformed behind it - with 8
  // waves this part of the launch's critical path takes half the time.
  uint4 p0reg[NI0];
  int p0lds[NI0];
  unsigned p0ok = 0;
  const bool reg0 = GNP && p.seg_gn[0] >= 0;
  const bool p0cok = (tid & 7) * VW < min(KC, (int)p.seg[0].cin);
  {
    const T* src = reinterpret_cast<const T*>(p.seg[0].src) + img * p.seg[0].C + p.seg[0].coff + (p0cok ? (tid & 7) : 0) * VW;
#pragma unroll
    for (int j = 0; j < NI0; ++j) {
      const int row = (tid >> 3) + 64 * j;
      const int rr = min(row, PROWS - 1);
      const int pr = rr / PC, pc = rr - pr * PC;
      const int iy = y0 + pr - 1, ix = x0 + pc - 1;
      const bool ok = row < PROWS && iy >= 0 && iy < H && ix >= 0 && ix < W;
      p0ok |= (ok && p0cok) ? (1u << j) : 0u;
      p0lds[j] = row < PROWS ? row * 128 + (((tid & 7) ^ ((pc >> 1) & 7)) << 4) : -1;
      p0reg[j] = *reinterpret_cast<const uint4*>(src + (int64_t)(ok ? iy * W + ix : 0) * p.seg[0].C);
    }
  }
  auto p0_store = [&]() __attribute__((always_inline)) {
    float sc[8], sh[8];
    if constexpr (GNP) {
      if (reg0) {
        const int gcb = p.seg_gn[0] + (p0cok ? (tid & 7) : 0) * VW;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(s_gn + gcb), a1 = *reinterpret_cast<const f32x4*>(s_gn + gcb + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(s_gn + gnpad + gcb), b1 = *reinterpret_cast<const f32x4*>(s_gn + gnpad + gcb + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { sc[q] = a0[q]; sc[4 + q] = a1[q]; sh[q] = b0[q]; sh[4 + q] = b1[q]; }
      }
    }
#pragma unroll
    for (int j = 0; j < NI0; ++j) {
      uint4 v = p0reg[j];
      if constexpr (GNP) { if (reg0) v = gn_transform16<T>(v, sc, sh, p.gn_silu != 0); }
      if (!((p0ok >> j) & 1u)) v = uint4{0u, 0u, 0u, 0u};          // padding AFTER the activation
      if (p0lds[j] >= 0) *reinterpret_cast<uint4*>(smem + p0lds[j]) = v;
    }
  };

  if (wave >= 4) {
    // =========================================== PRODUCERS ===========================================
    // Everything is staged through registers with ordinary (compiler-counted) loads.  An LDS-DMA version of this
    // role was measured first: one global_load_lds piece cost its wave ~230 cycles of issue beside the consumers' LDS
    // reads, 4 pieces per wave and k-step took the producers' whole interval (tools/ring_timeline.py).
    RSTAMP(32, 256);
    const int ptid = tid - 256;
    const int pchunk = ptid & 7;
    // ---- weights: the 16 KiB tile of k-step k lives in register set k % 3 (4 x 16 bytes per thread: rows ptid/8 + 32 i,
    // chunk pchunk) and goes to ring slot k % 3; requested 3 intervals before it is written
    const int wg_off = (ptid >> 3) * 128 + pchunk * 16;
    const int wl_off = WOFF + (ptid >> 3) * 128 + ((pchunk ^ ((ptid >> 4) & 7)) << 4);
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.w) + (int64_t)n0 * 128 + wg_off;
    const int64_t wstep = (int64_t)p.CoutPad * 128;
    // (twelve named registers: as arrays handed to lambdas these sets were placed in scratch memory by hipcc)
    uint4 wa0, wa1, wa2, wa3, wb0_, wb1_, wb2_, wb3_, wc0, wc1, wc2, wc3;
    int w_seg = 0, w_c = 0, w_tap = 0, w_kbase = 0, w_step = 0;
    // k index of the next k-step to request; advances the cursor, which WRAPS after the tile's last k-step: every tile
    // of the layer streams the same weight tiles, so the requests of a tile's last three intervals already are the
    // next tile's k-steps 0, 1, 2
    auto next_kidx = [&]() __attribute__((always_inline)) {
      const int nch = seg_nch(w_seg), ntaps = seg_taps(w_seg);
      const int kidx = w_kbase + w_tap * nch + w_c;
      if (++w_step == nsteps) {
        w_step = 0; w_seg = 0; w_c = 0; w_tap = 0; w_kbase = 0;
      } else if (++w_tap == ntaps) {
        w_tap = 0;
        if (++w_c == nch) { w_c = 0; w_kbase += ntaps * nch; ++w_seg; }
      }
      return kidx;
    };
#define RING_LOAD_W(A, B, C, D)                                             \
  do {                                                                      \
    const unsigned char* g_ = wbase + next_kidx() * wstep;                  \
    A = *reinterpret_cast<const uint4*>(g_);                                \
    B = *reinterpret_cast<const uint4*>(g_ + 4096);                         \
    C = *reinterpret_cast<const uint4*>(g_ + 8192);                         \
    D = *reinterpret_cast<const uint4*>(g_ + 12288);                        \
  } while (0)
#define RING_WRITE_W(A, B, C, D, SLOT)                                      \
  do {                                                                      \
    unsigned char* l_ = smem + wl_off + (SLOT) * WB;                        \
    *reinterpret_cast<uint4*>(l_) = A;                                      \
    *reinterpret_cast<uint4*>(l_ + 4096) = B;                               \
    *reinterpret_cast<uint4*>(l_ + 8192) = C;                               \
    *reinterpret_cast<uint4*>(l_ + 12288) = D;                              \
  } while (0)
    // interval of k-step s (s % 3 == SL): weights(s + 1) go to their slot, weights(s + 4) are requested into the set
    // that has just been written out
    auto weights_interval = [&](auto SL, int s) __attribute__((always_inline)) {
      constexpr int sl = (decltype(SL)::value + 1) % 3;
#ifdef RING_X_NOWEIGHTS
      return;
#endif
      // UNCONDITIONAL, every interval the same four writes and four requests: with a guard around them hipcc can no
      // longer count the requests in flight and waits for ALL of them (vmcnt(0)) before every write - the producers'
      // interval then lasts a load latency.  Past the tile's last k-step the cursor has wrapped: the surplus request is
      // the next tile's k-step (a workgroup walks several tiles only when the layer has 9-tap chunks alone, so
      // nsteps % 3 == 0 and k-step 0 of the next tile lands in slot 0 by itself); behind the last tile it is never read.
      (void)s;
      if constexpr (sl == 0) { RING_WRITE_W(wa0, wa1, wa2, wa3, 0); RING_LOAD_W(wa0, wa1, wa2, wa3); }
      else if constexpr (sl == 1) { RING_WRITE_W(wb0_, wb1_, wb2_, wb3_, 1); RING_LOAD_W(wb0_, wb1_, wb2_, wb3_); }
      else { RING_WRITE_W(wc0, wc1, wc2, wc3, 2); RING_LOAD_W(wc0, wc1, wc2, wc3); }
    };

    // weights 0..2 are requested FIRST: their latency (the launch's first touch of the layer's weights) passes behind
    // the integer work of the item tables below instead of in front of tick(-1)
    RING_LOAD_W(wa0, wa1, wa2, wa3);
    RING_LOAD_W(wb0_, wb1_, wb2_, wb3_);
    RING_LOAD_W(wc0, wc1, wc2, wc3);
    // item j of this thread = patch row (ptid / 8 + 32 j), source chunk pchunk, LDS position pchunk ^ key(row).
    // Pixel offsets of the CURRENT tile (ppix) and of the NEXT one (ppixn: the patches staged / requested across a tile
    // boundary belong to it), -1: padding; the LDS position is the same for every tile.
    int ppix[NIT], ppixn[NIT];
    int plds[NIT];          // row * 128 + ((pchunk ^ key) << 4)
    // (a value-returning helper, filled in by unrolled loops: handed to a lambda by reference the tables were placed in
    // scratch memory)
    auto ppix_of = [&](int j, int ti) __attribute__((always_inline)) {
      const int ty_ = ti / tiles_x, tx_ = ti - ty_ * tiles_x;
      // (the lane index made opaque per call site: the tile-independent halves of these expressions, 22 values, were
      // otherwise hoisted out of the tile loop, spilled, and reloaded behind a vmcnt(0) each)
      int ptid_o = ptid;
      asm volatile("" : "+v"(ptid_o));
      const int row = (ptid_o >> 3) + 32 * j;
      const int rr = min(row, PROWS - 1);
      const int pr = rr / PC, pc = rr - pr * PC;
      const int iy = ty_ * R + pr - 1, ix = tx_ * 16 + pc - 1;
      const bool ok = row < PROWS && iy >= 0 && iy < H && ix >= 0 && ix < W;
      return ok ? iy * W + ix : -1;
    };
#pragma unroll
    for (int j = 0; j < NIT; ++j) { ppix[j] = ppix_of(j, tile0); ppixn[j] = ppix[j]; }
    // (plds is filled in behind the first patch request, below: integer work that then covers the request's latency)
    int cy0 = y0, cx0 = x0;      // the current tile's origin (the tail's interior patches)

    // ---- patch staging of chunk (s, c) through registers
    uint4 preg[NIT];
    float tsc[8], tsh[8];
    bool pcok = true;
    auto patch_src = [&](int s, int c, int chunk16) __attribute__((always_inline)) {
      const int sg_C = SEG_FIELD(p, s, C), sg_coff = SEG_FIELD(p, s, coff);
      return reinterpret_cast<const T*>(SEG_FIELD(p, s, src)) + img * sg_C + sg_coff + c * KC + chunk16 * VW;
    };
    // pcok: this thread's 16-byte chunk lies inside the segment - for the patch being transformed / being requested
    bool pcok_next = true;
    const T* psrc_next = nullptr;
    int pC_next = 0;
    auto patch_request_begin = [&](int s, int c) __attribute__((always_inline)) {
      pC_next = SEG_FIELD(p, s, C);
      const int cvalid = min(KC, SEG_FIELD(p, s, cin) - c * KC);
      pcok_next = pchunk * VW < cvalid;
      psrc_next = patch_src(s, c, pcok_next ? pchunk : 0);
    };
    // (nt: the patch belongs to the NEXT tile of this workgroup)
    auto load_item = [&](auto JJ, bool nt) __attribute__((always_inline)) {
      constexpr int j = decltype(JJ)::value;
      // (two loads of VALUES, then a select: `nt ? ppixn[j] : ppix[j]` selects between two addresses and keeps both
      // tables in scratch memory)
      const int pa_ = ppix[j], pn_ = ppixn[j];
      preg[j] = *reinterpret_cast<const uint4*>(psrc_next + (int64_t)max(nt ? pn_ : pa_, 0) * pC_next);
    };
    auto load_scale_shift = [&](int s, int c) __attribute__((always_inline)) {
      if constexpr (GNP) {
        const int sgn = s == 0 ? p.seg_gn[0] : s == 1 ? p.seg_gn[1] : s == 2 ? p.seg_gn[2] : p.seg_gn[3];
        const int gcb = max(sgn, 0) + c * KC + (pcok ? pchunk : 0) * VW;   // (pcok: already this patch's)
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(s_gn + gcb), a1 = *reinterpret_cast<const f32x4*>(s_gn + gcb + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(s_gn + gnpad + gcb), b1 = *reinterpret_cast<const f32x4*>(s_gn + gnpad + gcb + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { tsc[q] = a0[q]; tsc[4 + q] = a1[q]; tsh[q] = b0[q]; tsh[4 + q] = b1[q]; }
      }
    };
    // GroupNorm + SiLU of item j in place (registers only: this is the part of an interval that runs AFTER its LDS
    // writes have been issued), and its write one interval later
    auto xform_item = [&](auto JJ, bool xform, bool nt) __attribute__((always_inline)) {
      constexpr int j = decltype(JJ)::value;
      uint4 v = preg[j];
#ifndef RING_X_NOXFORM
      if constexpr (GNP) { if (xform) v = gn_transform16<T>(v, tsc, tsh, p.gn_silu != 0); }
#endif
      const int pa_ = ppix[j], pn_ = ppixn[j];
      if (!((nt ? pn_ : pa_) >= 0 && pcok)) v = uint4{0u, 0u, 0u, 0u};            // padding AFTER the activation
      preg[j] = v;
    };
    auto write_item = [&](auto JJ, int buf) __attribute__((always_inline)) {
      constexpr int j = decltype(JJ)::value;
      if (j < NIT - 1 || (ptid >> 3) < PROWS - 32 * (NIT - 1))
        *reinterpret_cast<uint4*>(smem + buf * PB + plds[j]) = preg[j];
    };
    auto is_reg = [&](int s) __attribute__((always_inline)) {
      if constexpr (GNP) return (s == 0 ? p.seg_gn[0] : s == 1 ? p.seg_gn[1] : s == 2 ? p.seg_gn[2] : p.seg_gn[3]) >= 0;
      else return false;
    };
    // 1-tap chunks (raw shortcut segments) read the patch's interior only: item j = interior pixel (ptid / 8 + 32 j),
    // 8 per thread, in THREE register sets - the patch of tail chunk e is requested two intervals and written one
    // interval before the chunk's own (single) interval
    uint4 tp0[NTP], tp1[NTP], tp2[NTP];
    bool tcok0 = true, tcok1 = true, tcok2 = true;
    auto tail_load = [&](uint4 (&tp)[NTP], bool& cok, int s, int c) __attribute__((always_inline)) {
      const int sg_C = SEG_FIELD(p, s, C);
      const int cvalid = min(KC, SEG_FIELD(p, s, cin) - c * KC);
      cok = pchunk * VW < cvalid;
      const T* src = patch_src(s, c, cok ? pchunk : 0);
#pragma unroll
      for (int j = 0; j < NTP; ++j) {
        const int r = (ptid >> 3) + 32 * j;                    // interior pixel: row r / 16, column r % 16 (always inside the image)
        tp[j] = *reinterpret_cast<const uint4*>(src + (int64_t)((cy0 + (r >> 4)) * W + cx0 + (r & 15)) * sg_C);
      }
    };
    auto tail_store = [&](const uint4 (&tp)[NTP], bool cok, int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NTP; ++j) {
        const int r = (ptid >> 3) + 32 * j;
        const int pc = (r & 15) + 1;
        const int row = ((r >> 4) + 1) * PC + pc;
        *reinterpret_cast<uint4*>(smem + buf * PB + row * 128 + ((pchunk ^ ((pc >> 1) & 7)) << 4)) = cok ? tp[j] : uint4{0u, 0u, 0u, 0u};
      }
    };

    // ---- prologue: weights 0..3 requested, weights 0 written; patch 0 by all threads -----------------------------
    RSTAMP(33, 256);
    {
      // the patch of chunk 1 (clamped: of chunk 0 again where there is none) is requested HERE, a whole chunk ahead and in
      // front of the GroupNorm table and patch 0: requested behind them, its items 0 and 1 - transformed in the first
      // interval - were a full memory latency away (1.3 - 2 us of every launch, tools/ring_timeline.py)
      int s1 = 0, c1 = 1;
      if (c1 == sg_n0) { c1 = 0; s1 = 1; }
      if (s1 >= nseg) { s1 = 0; c1 = 0; }
      patch_request_begin(s1, c1);
      ring_static_for<0, NIT>([&](auto JJ) __attribute__((always_inline)) { load_item(JJ, false); });
    }
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int row = (ptid >> 3) + 32 * j;
      const int rr = min(row, PROWS - 1);
      const int pc = rr % PC;
      plds[j] = row * 128 + ((pchunk ^ ((pc >> 1) & 7)) << 4);
    }
    RSTAMP(34, 256);
    if constexpr (GNP) conv_gn_table<512>(p, b, 1, s_gn, gnpad, s_mr, smem + PB);
    RSTAMP(35, 256);
    p0_store();
    RING_WRITE_W(wa0, wa1, wa2, wa3, 0);
    RING_LOAD_W(wa0, wa1, wa2, wa3);     // weights(3)
    RSTAMP(36, 256);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();       // tick(-1)
    RSTAMP(37, 256);
    int pstamp = 38;
    (void)pstamp;

#ifdef RING_X_PRODPRIO
    __builtin_amdgcn_s_setprio(2);
#endif
    // ---- 9-tap chunks --------------------------------------------------------------------------------------------
    int n9 = 0, n1 = 0;
    n9 += sg_t0 == 9 ? sg_n0 : 0; n1 += sg_t0 == 1 ? sg_n0 : 0;
    if (nseg > 1) { n9 += sg_t1 == 9 ? sg_n1 : 0; n1 += sg_t1 == 1 ? sg_n1 : 0; }
    if (nseg > 2) { n9 += sg_t2 == 9 ? sg_n2 : 0; n1 += sg_t2 == 1 ? sg_n2 : 0; }
    if (nseg > 3) { n9 += sg_t3 == 9 ? sg_n3 : 0; n1 += sg_t3 == 1 ? sg_n3 : 0; }
    // cumulative item schedule of a 9-tap chunk: XS(t) items have been transformed before interval t
    static_assert(NIT == 11 || NIT <= 8, "item schedule");
    static_assert(G_::XS(8) == NIT && G_::XS(7) >= L7, "item schedule");
    int pb = 0;
    // (ns, nc): the chunk after (s_, c_); wraps to the next tile's chunk (0, 0) behind the tile's last chunk
    auto chunk_after = [&](int s_, int c_, int& ns_, int& nc_, bool& wrapped) __attribute__((always_inline)) {
      ns_ = s_; nc_ = c_ + 1;
      if (nc_ == seg_nch(s_)) { nc_ = 0; ns_ = s_ + 1; }
      if (ns_ == nseg) { ns_ = 0; wrapped = true; }
    };
    for (int ti = tile0; ti < tiles_per_image; ti += wgs_per_image) {
      const bool last_tile = ti + wgs_per_image >= tiles_per_image;
      int cs = 0, cc = 0, sbase = 0;
      if (!last_tile) {
        // (the next tile's table: first used by the patch requests of this tile's last chunks)
#pragma unroll
        for (int j = 0; j < NIT; ++j) ppixn[j] = ppix_of(j, ti + wgs_per_image);
      }
      // An interval = [LDS writes of what was prepared before: weights(s + 1), the patch items transformed in the previous
      // interval] -> [requests: weights(s + 4), at t = 7 / 8 the patch after the next one] -> [GroupNorm + SiLU of this
      // interval's items, registers only] -> tick.  The writes come FIRST so that they have completed by the time the
      // arithmetic is done: with the write at the end of the interval its completion latency (behind the consumers' LDS
      // reads) sat in front of every tick.  Items: transformed in intervals 0..7 (2,1,1,2,1,1,2,1), written one later.
      for (int ci = 0; ci < n9; ++ci) {
        int ns, nc, s2, c2;
        bool w1 = false, w2 = false;
        chunk_after(cs, cc, ns, nc, w1);
        w2 = w1;
        chunk_after(ns, nc, s2, c2, w2);
        // the chunk after this one: of this tile, or (no tail) the next tile's first - unless this is the last tile
        const bool has_next = !(w1 && last_tile);
        const bool nreg = has_next && is_reg(ns);
        // the patch requested during this chunk's last two intervals: the chunk after the next one.  Nothing to
        // request (this chunk's again, dropped): behind the last tile, or where the 1-tap tail requests its own patches
        const bool req = !(w2 && last_tile) && !(n1 > 0 && ci == n9 - 1);
        const bool req_nt = w2;
        pcok = pcok_next;
        auto interval = [&](auto TT) __attribute__((always_inline)) {
          constexpr int t = decltype(TT)::value;
          weights_interval(std::integral_constant<int, t % 3>{}, sbase + t);
          // items transformed in interval t: [XS(t), XS(t + 1)), written in interval t + 1; requested again (for the
          // chunk after the next) in interval 7 (items below L7, all written by then) and 8
          constexpr int x0_ = G_::XS(t), x1_ = G_::XS(t + 1), w0_ = t > 0 ? G_::XS(t - 1) : 0, w1_ = t > 0 ? G_::XS(t) : 0;
          if (has_next) ring_static_for<w0_, w1_>([&](auto JJ) __attribute__((always_inline)) { write_item(JJ, pb ^ 1); });
          // (unconditional loads)
          if constexpr (t == 7) {
            patch_request_begin(req ? s2 : cs, req ? c2 : cc);
            const bool nt = req && req_nt;
            ring_static_for<0, L7>([&](auto JJ) __attribute__((always_inline)) { load_item(JJ, nt); });
          }
          if constexpr (t == 8) {
            const bool nt = req && req_nt;
            ring_static_for<L7, NIT>([&](auto JJ) __attribute__((always_inline)) { load_item(JJ, nt); });
          }
          if constexpr (t == 0) { if (nreg) load_scale_shift(ns, nc); }
          ring_static_for<x0_, x1_>([&](auto JJ) __attribute__((always_inline)) { xform_item(JJ, nreg, w1); });
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef FDBM_STAMPS
          if (ci < 2 && ti == tile0) RSTAMP(64 + 2 * (ci * 9 + t), 256);      // interval's own work done (before the tick)
#endif
          __builtin_amdgcn_s_barrier();   // tick
#ifdef FDBM_STAMPS
          if (ci < 2 && ti == tile0) RSTAMP(65 + 2 * (ci * 9 + t), 256);      // tick passed
#endif
        };
        interval(std::integral_constant<int, 0>{}); interval(std::integral_constant<int, 1>{});
        interval(std::integral_constant<int, 2>{}); interval(std::integral_constant<int, 3>{});
        interval(std::integral_constant<int, 4>{}); interval(std::integral_constant<int, 5>{});
        interval(std::integral_constant<int, 6>{}); interval(std::integral_constant<int, 7>{});
        interval(std::integral_constant<int, 8>{});
        pb ^= 1;
        cs = ns; cc = nc;
        sbase += 9;
        RSTAMP(pstamp, 256);
        if (pstamp < 60) ++pstamp;
      }
      // ---- 1-tap chunks: tail chunk e runs in interval e; its interval writes the patch of chunk e + 1 (set (e+1) % 3,
      // requested one interval earlier) into the other buffer and requests the patch of chunk e + 2 (set (e+2) % 3).
      // The patch of tail chunk 1 is requested here (one exposed load latency per tile, instead of three more register
      // sets alive across the whole 9-tap loop); every request is unconditional, clamped to an existing chunk.
      if (n1 > 0) {
        int s1 = cs, c1 = cc;
        bool wdummy = false;
        if (n1 >= 2) chunk_after(cs, cc, s1, c1, wdummy);
        tail_load(tp1, tcok1, s1, c1);
        tail_load(tp2, tcok2, s1, c1);
        tail_load(tp0, tcok0, s1, c1);
      }
      for (int e0 = 0; e0 < n1; e0 += 3) {
        auto tail_interval = [&](auto EE) __attribute__((always_inline)) {
          constexpr int em = decltype(EE)::value;        // e % 3
          const int e = e0 + em;
          if (e < n1) {
            weights_interval(std::integral_constant<int, em>{}, sbase + e);
            int s1, c1, s2, c2;
            bool wd = false;
            chunk_after(cs, cc, s1, c1, wd);
            chunk_after(s1, c1, s2, c2, wd);
            if (e + 1 < n1) {
              if constexpr (em == 0) tail_store(tp1, tcok1, pb ^ 1); else if constexpr (em == 1) tail_store(tp2, tcok2, pb ^ 1); else tail_store(tp0, tcok0, pb ^ 1);
            }
            {
              const bool more = e + 2 < n1;
              const int sl = more ? s2 : cs, cl = more ? c2 : cc;
              if constexpr (em == 0) tail_load(tp2, tcok2, sl, cl); else if constexpr (em == 1) tail_load(tp0, tcok0, sl, cl); else tail_load(tp1, tcok1, sl, cl);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // tick
            pb ^= 1;
            cs = s1; cc = c1;
          }
        };
        tail_interval(std::integral_constant<int, 0>{});
        tail_interval(std::integral_constant<int, 1>{});
        tail_interval(std::integral_constant<int, 2>{});
      }
      // ---- tile boundary (the consumers are in the tile's second-to-last half / its epilogue) ---------------------------
      if (!last_tile) {
        // (weights: nothing to do - the ring runs on across the boundary, see weights_interval)
        // the next tile becomes the current one
        const int tn = ti + wgs_per_image;
#pragma unroll
        for (int j = 0; j < NIT; ++j) ppix[j] = ppixn[j];
        cy0 = (tn / tiles_x) * R; cx0 = (tn % tiles_x) * 16;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();       // tile boundary (also the consumers' statistics exchange)
    }
    return;
  }

  // ============================================ CONSUMERS ============================================
  RSTAMP_RT(30, 0);
  RSTAMP(0, 0);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fk = lane >> 4;
  f32x4 acc[4][RW];

  // per-lane fragment bases: (patch row wm*8 [+ i + dy], column frow + dx, k-half kk) and (weight row wn*64 [+ 16 j] + frow)
#define A_BASE(DX, KK) ((wm * RW * PC + frow + (DX)) * 128 + ((((KK) * 4 + fk) ^ (((frow + (DX)) >> 1) & 7)) << 4))
  int ab00 = A_BASE(0, 0), ab01 = A_BASE(0, 1), ab10 = A_BASE(1, 0), ab11 = A_BASE(1, 1), ab20 = A_BASE(2, 0), ab21 = A_BASE(2, 1);
#undef A_BASE
  int wb0 = WOFF + (wn * 64 + frow) * 128 + (((0 * 4 + fk) ^ ((frow >> 1) & 7)) << 4);
  int wb1 = WOFF + (wn * 64 + frow) * 128 + (((1 * 4 + fk) ^ ((frow >> 1) & 7)) << 4);

  // Fragment registers: the weights of a half k-step in TWO sets (the next half's are requested at the top of the
  // current one), the 8 activation rows in ONE set - row i is refilled with the next half's row i behind the MFMAs
  // that consumed it.  128 (acc) + 32 + 32 registers (+ what the scheduler renames).
  uint4 fa0[4], fa1[4], fb[RW];
  auto rd_w = [&](uint4 (&fa)[4], int base) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) fa[j] = *reinterpret_cast<const uint4*>(smem + base + j * (16 * 128));
  };
  // one half k-step: 32 MFMAs on (fa, fb); requests the next half's weights (wnext) into fan and rows (pnext) into fb.
  auto half = [&](const uint4 (&fa)[4], uint4 (&fan)[4], int wnext, int pnext) __attribute__((always_inline)) {
    rd_w(fan, wnext);
#pragma unroll
    for (int i = 0; i < RW; ++i) {
#pragma unroll
#ifdef RING_X_NOMFMA
      asm volatile("" :: "v"(fa[0].x), "v"(fa[1].x), "v"(fa[2].x), "v"(fa[3].x), "v"(fb[i].x), "v"(fb[i].w));
#else
      for (int j = 0; j < 4; ++j) Mfma<T>::run(fa[j], fb[i], acc[j][i]);
#endif
      fb[i] = *reinterpret_cast<const uint4*>(smem + pnext + i * (PC * 128));
    }
    // pinned issue order = the source order above: the 4 weight reads beside row 0's MFMAs, then every row's refill
    // right behind its own 4 MFMAs - the refill lands in the registers it has just freed (an earlier read would need
    // a second register set: 255 registers and spills), and still has 28 MFMAs (448 cycles) until its next use
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read (weights)
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA (row 0)
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // row 0 refill
#pragma unroll
    for (int i = 1; i < RW; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
  };
  // tick: LAST = the chunk's last k-step - the producers overwrite this chunk's patch buffer next, so every read of
  // it has to have completed (the reads of the second half were all issued during the first)
  auto tick = [&](auto LAST) __attribute__((always_inline)) {
    if constexpr (decltype(LAST)::value) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  RSTAMP(1, 0);
  if (tid < 128) {
    // per-channel additive constants of the epilogue, once per workgroup (published by tick(-1))
    const int n = n0 + tid;
    float v = 0.f;
    if (n < p.Cout) {
      if (p.bias) v = p.bias[n];
      if (p.tbias) v += p.tbias[(int64_t)b * p.tbias_stride + n];
    }
    s_cbt[tid] = v;
  }
  if constexpr (GNP) conv_gn_table<512>(p, b, 1, s_gn, gnpad, s_mr, smem + PB);
  RSTAMP(2, 0);
  p0_store();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();           // tick(-1): patch 0 and weight tile 0 are in LDS
  RSTAMP(3, 0);
  __builtin_amdgcn_sched_barrier(0);
#if defined(RING_X_NOPRIO) || defined(RING_X_PRODPRIO)
#else
  __builtin_amdgcn_s_setprio(2);
#endif

  // The 9-tap chunks come first (fdbm_conv_ring_ok), then the 1-tap ones: two loops one after the other, so the 128
  // accumulator registers are carried through straight-line loop bodies and stay in place.  In the 9-tap loop every
  // offset is an immediate: tap k reads ring slot k % 3 (a chunk is 9 k-steps: each starts at slot 0).
  int n9 = 0, n1 = 0;
  n9 += sg_t0 == 9 ? sg_n0 : 0; n1 += sg_t0 == 1 ? sg_n0 : 0;
  if (nseg > 1) { n9 += sg_t1 == 9 ? sg_n1 : 0; n1 += sg_t1 == 1 ? sg_n1 : 0; }
  if (nseg > 2) { n9 += sg_t2 == 9 ? sg_n2 : 0; n1 += sg_t2 == 1 ? sg_n2 : 0; }
  if (nseg > 3) { n9 += sg_t3 == 9 ? sg_n3 : 0; n1 += sg_t3 == 1 ? sg_n3 : 0; }
  for (int ti = tile0; ti < tiles_per_image; ti += wgs_per_image) {
  const int tile = ti;
  const int y0 = (ti / tiles_x) * R, x0 = (ti % tiles_x) * 16;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < RW; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    // (behind tick(-1) / the tile boundary: the tile's first patch and weight tile 0 are in LDS)
    rd_w(fa0, wb0);
#pragma unroll
    for (int i = 0; i < RW; ++i) fb[i] = *reinterpret_cast<const uint4*>(smem + ab00 + i * (PC * 128));
  }
  for (int c = 0; c < n9; ++c) {
    const int dflip = (ab00 >= PB) ? -PB : PB;      // to the other patch buffer (the same for every lane)
    // first k-half of the chunk after this one: tap 0 of a 9-tap chunk, or the centre tap of a 1-tap chunk
    // (past the end: an address inside LDS whose data is never used)
    const int pnext = (c + 1 < n9 ? ab00 : ab10 + PC * 128) + dflip;
    auto step = [&](auto KK) __attribute__((always_inline)) {
      constexpr int k = decltype(KK)::value;
      constexpr int dy = k / 3, dx = k % 3, sl = k % 3, sl2 = (k + 1) % 3;
      // first half (k-half 0, fa0); requests k-half 1 of this step
      half(fa0, fa1, wb1 + sl * WB, (dx == 0 ? ab01 : dx == 1 ? ab11 : ab21) + dy * (PC * 128));
      tick(std::integral_constant<bool, k == 8>{});
      // second half (fa1); requests k-half 0 of the next step
      if constexpr (k < 8) {
        constexpr int dy2 = (k + 1) / 3, dx2 = (k + 1) % 3;
        half(fa1, fa0, wb0 + sl2 * WB, (dx2 == 0 ? ab00 : dx2 == 1 ? ab10 : ab20) + dy2 * (PC * 128));
      } else {
        half(fa1, fa0, wb0 + sl2 * WB, pnext);
      }
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
    step(std::integral_constant<int, 8>{});
    ab00 += dflip; ab01 += dflip; ab10 += dflip; ab11 += dflip; ab20 += dflip; ab21 += dflip;
    RSTAMP(4 + (c < 15 ? c : 15), 0);
  }
  {
    int slot = 0;
    for (int c = 0; c < n1; ++c) {
      const int dflip = (ab00 >= PB) ? -PB : PB;
      half(fa0, fa1, wb1 + slot * WB, ab11 + PC * 128);
      tick(std::integral_constant<bool, true>{});
      slot = slot == NSLOT - 1 ? 0 : slot + 1;
      half(fa1, fa0, wb0 + slot * WB, ab10 + PC * 128 + dflip);
      ab00 += dflip; ab01 += dflip; ab10 += dflip; ab11 += dflip; ab20 += dflip; ab21 += dflip;
    }
  }
  RSTAMP(20, 0);

  // ---- epilogue -------------------------------------------------------------------------------------------------
  // Lane (frow_e, fk_e) holds channels n_j .. n_j+3 (n_j = n0 + wn*64 + 16 j + 4 fk_e) of pixel (y0 + wm*8 + i, x0 + frow_e).
  // No memory operation sits behind a per-element branch (a conditional load makes hipcc wait for each one
  // separately: 32 serialised round trips, 13 us of a 30 us launch): per-channel constants are fetched once, the
  // residual in batches of 16 unconditional loads, and with bf16 output the lanes of a DPP row pair exchange halves
  // (v_permlane16_swap) so that every lane owns 8 consecutive channels: 16-byte loads / stores, half the instructions.
  // (lane coordinates made opaque per tile: with them loop-invariant hipcc hoists some forty 64-bit addresses of the
  // epilogue out of the tile loop and spills them around the k-loop)
  int frow_e = frow, fk_e = fk;
  asm volatile("" : "+v"(frow_e), "+v"(fk_e));
  const int Cout = p.Cout;
  const bool do_stat = p.stat_out != nullptr;
  const int scpg = do_stat ? Cout / p.stat_G : 1;
  float a1[4], a2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a1[j] = a2[j] = 0.f;
  static_assert(sizeof(TO) == 2, "the ring kernel stores 16-bit tensors (fdbm_conv_ring_ok)");
  {
    // ONE form (fdbm_conv_ring_ok: 16-bit output, Cout a multiple of 8, no pyramid / Combine extras): bias, time bias,
    // residual, scale, statistics.  Kept short, because nothing overlaps it - the workgroup's producers wait at the tile
    // boundary: packed f32 arithmetic, the additive constants from LDS, buffer addressing (ONE 32-bit offset register
    // for the 16 stores and the 16 residual loads: row = scalar offset, channel pair = immediate), and the lanes of a
    // DPP row pair exchange halves (v_permlane16_swap) so that every lane owns 8 consecutive channels: 16-byte loads /
    // stores.  No memory operation sits behind a per-element branch (hipcc waits for each such load separately).
    using X4 = typename V16<TO>::x4;
    const uint32_t rowB = (uint32_t)W * (uint32_t)Cout * 2u;
    const bool has_r = p.res != nullptr;
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<TO*>(p.out) + img * Cout, 0, (int)((uint32_t)H * rowB), 0x00020000);
    __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<TO*>(reinterpret_cast<const TO*>(has_r ? p.res : p.out)) + img * Cout, 0, (int)((uint32_t)H * rowB), 0x00020000);
    // after the exchange the lane in DPP row fk_e owns channels [16 (2 q + (fk_e & 1)) + 8 (fk_e >> 1), + 8) of pair q
    const int nq = n0 + wn * 64 + (fk_e & 1) * 16 + (fk_e >> 1) * 8;
    const uint32_t voff = (uint32_t)((y0 + wm * RW) * W + x0 + frow_e) * (uint32_t)Cout * 2u + (uint32_t)nq * 2u;
    const bool okq0 = nq < Cout, okq1 = nq + 32 < Cout;
    f32x2 cl[4], ch[4], s1[4], s2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 c = *reinterpret_cast<const f32x4*>(s_cbt + wn * 64 + j * 16 + fk_e * 4);
      cl[j] = f32x2{c[0], c[1]}; ch[j] = f32x2{c[2], c[3]};
      s1[j] = s2[j] = f32x2{0.f, 0.f};
    }
    const f32x2 scale2 = {p.scale, p.scale};
#pragma unroll
    for (int ih = 0; ih < RW / 4; ++ih) {
      u32x4_t rw[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) rw[i][q] = u32x4_t{0u, 0u, 0u, 0u};
      if (has_r) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            rw[i][q] = __builtin_amdgcn_raw_buffer_load_b128(rr, voff + q * 64, (ih * 4 + i) * rowB, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ii = ih * 4 + i;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const auto x0_ = __builtin_amdgcn_permlane16_swap(rw[i][q][0], rw[i][q][2], false, false);
          const auto x1_ = __builtin_amdgcn_permlane16_swap(rw[i][q][1], rw[i][q][3], false, false);
          const uint2 ta = {x0_[0], x1_[0]}, tb = {x0_[1], x1_[1]};
          const X4 ea = *reinterpret_cast<const X4*>(&ta), eb = *reinterpret_cast<const X4*>(&tb);
          uint2 pk[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int j = 2 * q + h;
            const X4 e = h == 0 ? ea : eb;
            f32x2 lo = {acc[j][ii][0], acc[j][ii][1]}, hi = {acc[j][ii][2], acc[j][ii][3]};
            lo += cl[j]; hi += ch[j];
            lo += f32x2{(float)e[0], (float)e[1]}; hi += f32x2{(float)e[2], (float)e[3]};
            lo *= scale2; hi *= scale2;
            const X4 t = {(TO)lo[0], (TO)lo[1], (TO)hi[0], (TO)hi[1]};
            pk[h] = *reinterpret_cast<const uint2*>(&t);
            // statistics are those of the STORED (rounded) tensor
            const f32x2 bl = {(float)t[0], (float)t[1]}, bh = {(float)t[2], (float)t[3]};
            s1[j] += bl + bh;
            s2[j] += bl * bl + bh * bh;
          }
          const auto y0_ = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
          const auto y1_ = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
          if (q == 0 ? okq0 : okq1)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{y0_[0], y1_[0], y0_[1], y1_[1]}, ro, voff + q * 64, ii * rowB, FDBM_RING_STORE_AUX);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { a1[j] = s1[j][0] + s1[j][1]; a2[j] = s2[j][0] + s2[j][1]; }
  }
  RSTAMP(21, 0);
  if (do_stat) {
    // straight to the global unit statistics, one fp64 atomic pair per (wave, 4-channel unit): through an LDS stage and
    // the boundary barrier the adds were issued as the very last thing of the launch and their round trip (~2 us) sat
    // behind every workgroup's last store
    double* srow = p.stat_out + ((int64_t)b * p.stat_nsplit + tile % p.stat_nsplit) * p.stat_G * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fk_e * 4;
      const float r1 = row16_sum(a1[j]), r2 = row16_sum(a2[j]);
      if (frow_e == 0 && n < Cout) {
        atomicAdd(srow + (n / scpg) * 2, (double)r1);
        atomicAdd(srow + (n / scpg) * 2 + 1, (double)r2);
      }
    }
  }
  // tile boundary: the producers have put the next tile's first patch and weight tile 0 into LDS
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  }     // tiles
  __builtin_amdgcn_s_setprio(0);
  RSTAMP(22, 0);
  RSTAMP_RT(31, 0);
}

template <typename T, typename TO, bool GNP, int R>
static int launch_ring(const ConvParams& p, hipStream_t st) {
  constexpr int GOFF = RingGeom<R>::GOFF;
  constexpr int SMEM_MAX = GOFF + (GNP ? CONV_GN_MAXC * 8 : 0) + 64 * 8 + 64 * 4 + 128 * 4;
  const int SMEM = GOFF + (GNP ? ((p.gn_C + 63) & ~63) * 8 : 0) + 64 * 8 + 64 * 4 + 128 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ring_kernel<T, TO, GNP, R>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_MAX);
    if (e != hipSuccess) {
      fdbm_set_error("fdbm_conv_igemm(ring): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  const int tiles_x = p.W / 16, tiles_y = p.H / R;
  const int tpi = tiles_x * tiles_y, ny = (p.Cout + 127) / 128;
  // workgroups per image: enough to put one on every CU of the device, each then walks tpi / wpi tiles of its image; a
  // layer of a single channel chunk is not walked (its patch pipeline would reach two tiles ahead)
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
               ? prop.multiProcessorCount : 256;
  }
  int wpi = (n_cu + p.B * ny - 1) / (p.B * ny);
  if (wpi > tpi) wpi = tpi;
  if (wpi < 1) wpi = 1;
  { static const char* e = getenv("FDBM_RING_WGS_PER_IMAGE"); if (e && atoi(e) > 0) wpi = atoi(e) < tpi ? atoi(e) : tpi; }   // experiments / tests
  // The guards come AFTER the override: a workgroup may walk several tiles only where the 3-slot weight ring
  // stays in phase across a tile boundary - k-steps per tile a multiple of 3, which 9-tap chunks alone guarantee - and
  // where the patch pipeline's two-chunk look-ahead stays inside the tile (more than one channel chunk)
  bool one_tap = false;
  for (int sg = 0; sg < p.nseg; ++sg)
    if (p.seg[sg].taps == 1) one_tap = true;
  if (p.nk <= 9 || one_tap) wpi = tpi;
  FDBM_CHECK(wpi == tpi || (p.nk % 3 == 0 && !one_tap && p.nk > 9),
             "fdbm_conv_igemm(ring): %d workgroups per image of %d tiles with %d k-steps: the weight ring would lose its phase", wpi, tpi, p.nk);
  dim3 grid((unsigned)(p.B * wpi), (unsigned)ny);
#ifdef FDBM_STAMPS
  ConvParams pd = p;
  { const char* e = getenv("FDBM_RING_DBG"); pd.ksplit = e ? atoi(e) : 0; }
  conv_ring_kernel<T, TO, GNP, R><<<grid, 512, SMEM, st>>>(pd, tiles_x, tiles_y, wpi);
#else
  conv_ring_kernel<T, TO, GNP, R><<<grid, 512, SMEM, st>>>(p, tiles_x, tiles_y, wpi);
#endif
  FDBM_LAUNCH_CHECK("fdbm_conv_igemm(ring)");
  return 0;
}

