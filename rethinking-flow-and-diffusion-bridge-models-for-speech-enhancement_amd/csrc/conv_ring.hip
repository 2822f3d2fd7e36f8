// conv_ring.hip - 3x3 convolution for the large feature maps, producer / consumer waves around an LDS ring.
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

namespace ring {
constexpr int PC = 18;                 // patch columns (16 + halo)
constexpr int PROWS = 18 * 18;         // patch pixels
constexpr int PB = 328 * 128;          // bytes of one patch buffer (rows padded to a multiple of 8: whole DMA pieces)
constexpr int WB = 128 * 128;          // bytes of one weight tile (128 output channels x 128 bytes)
constexpr int NSLOT = 3;
constexpr int WOFF = 2 * PB;
constexpr int GOFF = WOFF + NSLOT * WB;
constexpr int NIT = 11;                // patch items (16 bytes) per producer thread: 324 rows x 8 / 256 threads
}  // namespace ring


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

template <typename T, typename TO, bool GNP>
__global__ void __launch_bounds__(512) conv_ring_kernel(const ConvParams p, int tiles_x, int tiles_y) {
  using namespace ring;
  constexpr int KC = 64, VW = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int gnpad = GNP ? ((p.gn_C + 63) & ~63) : 0;
  float* s_gn = reinterpret_cast<float*>(smem + GOFF);                          // scale[gnpad] | shift[gnpad]
  double* s_stat = reinterpret_cast<double*>(smem + GOFF + gnpad * 8);          // [32][2]
  float* s_mr = reinterpret_cast<float*>(s_stat + 64);                          // [32][2] mean, rstd

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W;
  // workgroups that share an XCD (blockIdx % 8) take neighbouring tiles: their halos meet in that XCD's L2
  int tile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tx = tile % tiles_x;
  const int ty = (tile / tiles_x) % tiles_y;
  const int b = tile / (tiles_x * tiles_y);
  const int y0 = ty * 16, x0 = tx * 16;
  const int n0 = blockIdx.y * 128;
  const int64_t img = (int64_t)b * H * W;

  if (p.stat_out && tid < 64) s_stat[tid] = 0.0;

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
  constexpr int NI0 = 6;
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
    // item j of this thread = patch row (ptid / 8 + 32 j), source chunk pchunk, LDS position pchunk ^ key(row)
    int ppix[NIT];          // pixel offset inside the image, -1: padding
    int plds[NIT];          // row * 128 + ((pchunk ^ key) << 4)
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int row = (ptid >> 3) + 32 * j;
      const int rr = min(row, PROWS - 1);
      const int pr = rr / PC, pc = rr - pr * PC;
      const int iy = y0 + pr - 1, ix = x0 + pc - 1;
      const bool ok = row < PROWS && iy >= 0 && iy < H && ix >= 0 && ix < W;
      ppix[j] = ok ? iy * W + ix : -1;
      plds[j] = row * 128 + ((pchunk ^ ((pc >> 1) & 7)) << 4);
    }

    // ---- weights: the 16 KiB tile of k-step k lives in register set k % 3 (4 x 16 bytes per thread: rows ptid/8 + 32 i,
    // chunk pchunk) and goes to ring slot k % 3; requested 3 intervals before it is written
    const int wg_off = (ptid >> 3) * 128 + pchunk * 16;
    const int wl_off = WOFF + (ptid >> 3) * 128 + ((pchunk ^ ((ptid >> 4) & 7)) << 4);
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.w) + (int64_t)n0 * 128 + wg_off;
    const int64_t wstep = (int64_t)p.CoutPad * 128;
    // (twelve named registers: as arrays handed to lambdas these sets were placed in scratch memory by hipcc)
    uint4 wa0, wa1, wa2, wa3, wb0_, wb1_, wb2_, wb3_, wc0, wc1, wc2, wc3;
    int w_seg = 0, w_c = 0, w_tap = 0, w_kbase = 0, w_step = 0;
    // k index of the next k-step to request; advances the cursor.  Past the last k-step it keeps returning the last
    // one: the loads below stay UNCONDITIONAL (a conditionally defined register set is placed in scratch memory by
    // this compiler - every load then waited for on its own and stored to scratch), the surplus tiles are never written
    auto next_kidx = [&]() __attribute__((always_inline)) {
      const int nch = seg_nch(w_seg), ntaps = seg_taps(w_seg);
      const int kidx = w_kbase + w_tap * nch + w_c;
      if (w_step + 1 < nsteps) {
        ++w_step;
        if (++w_tap == ntaps) {
          w_tap = 0;
          if (++w_c == nch) { w_c = 0; w_kbase += ntaps * nch; ++w_seg; }
        }
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
      const bool wr = s + 1 < nsteps;
      if constexpr (sl == 0) { if (wr) RING_WRITE_W(wa0, wa1, wa2, wa3, 0); RING_LOAD_W(wa0, wa1, wa2, wa3); }
      else if constexpr (sl == 1) { if (wr) RING_WRITE_W(wb0_, wb1_, wb2_, wb3_, 1); RING_LOAD_W(wb0_, wb1_, wb2_, wb3_); }
      else { if (wr) RING_WRITE_W(wc0, wc1, wc2, wc3, 2); RING_LOAD_W(wc0, wc1, wc2, wc3); }
    };

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
    auto load_item = [&](auto JJ) __attribute__((always_inline)) {
      constexpr int j = decltype(JJ)::value;
      preg[j] = *reinterpret_cast<const uint4*>(psrc_next + (int64_t)max(ppix[j], 0) * pC_next);
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
    auto xform_item = [&](auto JJ, bool xform) __attribute__((always_inline)) {
      constexpr int j = decltype(JJ)::value;
      uint4 v = preg[j];
#ifndef RING_X_NOXFORM
      if constexpr (GNP) { if (xform) v = gn_transform16<T>(v, tsc, tsh, p.gn_silu != 0); }
#endif
      if (!(ppix[j] >= 0 && pcok)) v = uint4{0u, 0u, 0u, 0u};            // padding AFTER the activation
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
    uint4 tp0[8], tp1[8], tp2[8];
    bool tcok0 = true, tcok1 = true, tcok2 = true;
    auto tail_load = [&](uint4 (&tp)[8], bool& cok, int s, int c) __attribute__((always_inline)) {
      const int sg_C = SEG_FIELD(p, s, C);
      const int cvalid = min(KC, SEG_FIELD(p, s, cin) - c * KC);
      cok = pchunk * VW < cvalid;
      const T* src = patch_src(s, c, cok ? pchunk : 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = (ptid >> 3) + 32 * j;                    // interior pixel: row r / 16, column r % 16 (always inside the image)
        tp[j] = *reinterpret_cast<const uint4*>(src + (int64_t)((y0 + (r >> 4)) * W + x0 + (r & 15)) * sg_C);
      }
    };
    auto tail_store = [&](const uint4 (&tp)[8], bool cok, int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = (ptid >> 3) + 32 * j;
        const int pc = (r & 15) + 1;
        const int row = ((r >> 4) + 1) * PC + pc;
        *reinterpret_cast<uint4*>(smem + buf * PB + row * 128 + ((pchunk ^ ((pc >> 1) & 7)) << 4)) = cok ? tp[j] : uint4{0u, 0u, 0u, 0u};
      }
    };

    // ---- prologue: weights 0..3 requested, weights 0 written; patch 0 by all threads -----------------------------
    RSTAMP(33, 256);
    RING_LOAD_W(wa0, wa1, wa2, wa3);
    RING_LOAD_W(wb0_, wb1_, wb2_, wb3_);
    RING_LOAD_W(wc0, wc1, wc2, wc3);
    RSTAMP(34, 256);
    if constexpr (GNP) conv_gn_table<512>(p, b, 1, s_gn, gnpad, s_mr, smem + PB);
    RSTAMP(35, 256);
    p0_store();
    RING_WRITE_W(wa0, wa1, wa2, wa3, 0);
    RING_LOAD_W(wa0, wa1, wa2, wa3);     // weights(3)
    {
      // the patch of chunk 1 (clamped: of chunk 0 again where there is none) is requested now, a whole chunk ahead
      int s1 = 0, c1 = 1;
      if (c1 == sg_n0) { c1 = 0; s1 = 1; }
      if (s1 >= nseg) { s1 = 0; c1 = 0; }
      patch_request_begin(s1, c1);
      load_item(std::integral_constant<int, 0>{}); load_item(std::integral_constant<int, 1>{}); load_item(std::integral_constant<int, 2>{});
      load_item(std::integral_constant<int, 3>{}); load_item(std::integral_constant<int, 4>{}); load_item(std::integral_constant<int, 5>{});
      load_item(std::integral_constant<int, 6>{}); load_item(std::integral_constant<int, 7>{}); load_item(std::integral_constant<int, 8>{});
      load_item(std::integral_constant<int, 9>{}); load_item(std::integral_constant<int, 10>{});
    }
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
    int cs = 0, cc = 0, pb = 0, sbase = 0;
    // (ns, nc): the chunk after (cs, cc)
    auto chunk_after = [&](int s_, int c_, int& ns_, int& nc_) __attribute__((always_inline)) {
      ns_ = s_; nc_ = c_ + 1;
      if (nc_ == seg_nch(s_)) { nc_ = 0; ns_ = s_ + 1; }
    };
    // An interval = [LDS writes of what was prepared before: weights(s + 1), the patch items transformed in the previous
    // interval] -> [requests: weights(s + 4), at t = 7 / 8 the patch after the next one] -> [GroupNorm + SiLU of this
    // interval's items, registers only] -> tick.  The writes come FIRST so that they have completed by the time the
    // arithmetic is done: with the write at the end of the interval its completion latency (behind the consumers' LDS
    // reads) sat in front of every tick.  Items: transformed in intervals 0..7 (2,1,1,2,1,1,2,1), written one later.
    for (int ci = 0; ci < n9; ++ci) {
      int ns, nc, s2, c2;
      chunk_after(cs, cc, ns, nc);
      chunk_after(ns, nc, s2, c2);
      const bool has_next = ns < nseg;
      const bool nreg = has_next && is_reg(ns);
      // the patch requested during this chunk's last two intervals: the chunk after the next one (none left: this
      // chunk's again, dropped)
      const bool req = s2 < nseg;
      pcok = pcok_next;
      auto interval = [&](auto TT) __attribute__((always_inline)) {
        constexpr int t = decltype(TT)::value;
        weights_interval(std::integral_constant<int, t % 3>{}, sbase + t);
        if (has_next) {
          if constexpr (t == 1) { write_item(std::integral_constant<int, 0>{}, pb ^ 1); write_item(std::integral_constant<int, 1>{}, pb ^ 1); }
          if constexpr (t == 2) write_item(std::integral_constant<int, 2>{}, pb ^ 1);
          if constexpr (t == 3) write_item(std::integral_constant<int, 3>{}, pb ^ 1);
          if constexpr (t == 4) { write_item(std::integral_constant<int, 4>{}, pb ^ 1); write_item(std::integral_constant<int, 5>{}, pb ^ 1); }
          if constexpr (t == 5) write_item(std::integral_constant<int, 6>{}, pb ^ 1);
          if constexpr (t == 6) write_item(std::integral_constant<int, 7>{}, pb ^ 1);
          if constexpr (t == 7) { write_item(std::integral_constant<int, 8>{}, pb ^ 1); write_item(std::integral_constant<int, 9>{}, pb ^ 1); }
          if constexpr (t == 8) write_item(std::integral_constant<int, 10>{}, pb ^ 1);
        }
        // (unconditional loads, see next_kidx)
        if constexpr (t == 7) {
          patch_request_begin(req ? s2 : cs, req ? c2 : cc);
          load_item(std::integral_constant<int, 0>{}); load_item(std::integral_constant<int, 1>{}); load_item(std::integral_constant<int, 2>{});
          load_item(std::integral_constant<int, 3>{}); load_item(std::integral_constant<int, 4>{}); load_item(std::integral_constant<int, 5>{});
          load_item(std::integral_constant<int, 6>{}); load_item(std::integral_constant<int, 7>{});
        }
        if constexpr (t == 8) {
          load_item(std::integral_constant<int, 8>{}); load_item(std::integral_constant<int, 9>{}); load_item(std::integral_constant<int, 10>{});
        }
        if constexpr (t == 0) { if (nreg) load_scale_shift(ns, nc); }
        if constexpr (t == 0) { xform_item(std::integral_constant<int, 0>{}, nreg); xform_item(std::integral_constant<int, 1>{}, nreg); }
        if constexpr (t == 1) xform_item(std::integral_constant<int, 2>{}, nreg);
        if constexpr (t == 2) xform_item(std::integral_constant<int, 3>{}, nreg);
        if constexpr (t == 3) { xform_item(std::integral_constant<int, 4>{}, nreg); xform_item(std::integral_constant<int, 5>{}, nreg); }
        if constexpr (t == 4) xform_item(std::integral_constant<int, 6>{}, nreg);
        if constexpr (t == 5) xform_item(std::integral_constant<int, 7>{}, nreg);
        if constexpr (t == 6) { xform_item(std::integral_constant<int, 8>{}, nreg); xform_item(std::integral_constant<int, 9>{}, nreg); }
        if constexpr (t == 7) xform_item(std::integral_constant<int, 10>{}, nreg);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // tick
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
    // The patch of tail chunk 1 is requested here (one exposed load latency per launch, instead of three more register
    // sets alive across the whole 9-tap loop); every request is unconditional, clamped to an existing chunk.
    {
      int s1 = cs, c1 = cc;
      if (n1 >= 2) chunk_after(cs, cc, s1, c1);
      tail_load(tp1, tcok1, min(s1, nseg - 1), s1 < nseg ? c1 : 0);
      tail_load(tp2, tcok2, min(s1, nseg - 1), s1 < nseg ? c1 : 0);
      tail_load(tp0, tcok0, min(s1, nseg - 1), s1 < nseg ? c1 : 0);
    }
    for (int e0 = 0; e0 < n1; e0 += 3) {
      auto tail_interval = [&](auto EE) __attribute__((always_inline)) {
        constexpr int em = decltype(EE)::value;        // e % 3
        const int e = e0 + em;
        if (e < n1) {
          weights_interval(std::integral_constant<int, em>{}, sbase + e);
          int s1, c1, s2, c2;
          chunk_after(cs, cc, s1, c1);
          chunk_after(s1, c1, s2, c2);
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
    if (p.stat_out) __builtin_amdgcn_s_barrier();     // the consumers' statistics exchange
    return;
  }

  // ============================================ CONSUMERS ============================================
  RSTAMP_RT(30, 0);
  RSTAMP(0, 0);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fk = lane >> 4;
  f32x4 acc[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane fragment bases: (patch row wm*8 [+ i + dy], column frow + dx, k-half kk) and (weight row wn*64 [+ 16 j] + frow)
#define A_BASE(DX, KK) ((wm * 8 * PC + frow + (DX)) * 128 + ((((KK) * 4 + fk) ^ (((frow + (DX)) >> 1) & 7)) << 4))
  int ab00 = A_BASE(0, 0), ab01 = A_BASE(0, 1), ab10 = A_BASE(1, 0), ab11 = A_BASE(1, 1), ab20 = A_BASE(2, 0), ab21 = A_BASE(2, 1);
#undef A_BASE
  int wb0 = WOFF + (wn * 64 + frow) * 128 + (((0 * 4 + fk) ^ ((frow >> 1) & 7)) << 4);
  int wb1 = WOFF + (wn * 64 + frow) * 128 + (((1 * 4 + fk) ^ ((frow >> 1) & 7)) << 4);

  // Fragment registers: the weights of a half k-step in TWO sets (the next half's are requested at the top of the
  // current one), the 8 activation rows in ONE set - row i is refilled with the next half's row i behind the MFMAs
  // that consumed it.  128 (acc) + 32 + 32 registers (+ what the scheduler renames).
  uint4 fa0[4], fa1[4], fb[8];
  auto rd_w = [&](uint4 (&fa)[4], int base) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) fa[j] = *reinterpret_cast<const uint4*>(smem + base + j * (16 * 128));
  };
  // one half k-step: 32 MFMAs on (fa, fb); requests the next half's weights (wnext) into fan and rows (pnext) into fb.
  auto half = [&](const uint4 (&fa)[4], uint4 (&fan)[4], int wnext, int pnext) __attribute__((always_inline)) {
    rd_w(fan, wnext);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
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
    for (int i = 1; i < 8; ++i) {
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
  {
    rd_w(fa0, wb0);
#pragma unroll
    for (int i = 0; i < 8; ++i) fb[i] = *reinterpret_cast<const uint4*>(smem + ab00 + i * (PC * 128));
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
  __builtin_amdgcn_s_setprio(0);
  RSTAMP(20, 0);

  // ---- epilogue -------------------------------------------------------------------------------------------------
  // Lane (frow, fk) holds channels n_j .. n_j+3 (n_j = n0 + wn*64 + 16 j + 4 fk) of pixel (y0 + wm*8 + i, x0 + frow).
  // No memory operation sits behind a per-element branch (a conditional load makes hipcc wait for each one
  // separately: 32 serialised round trips, 13 us of a 30 us launch): per-channel constants are fetched once, the
  // residual in batches of 16 unconditional loads, and with bf16 output the lanes of a DPP row pair exchange halves
  // (v_permlane16_swap) so that every lane owns 8 consecutive channels: 16-byte loads / stores, half the instructions.
  const int Cout = p.Cout;
  const bool do_stat = p.stat_out != nullptr;
  const int scpg = do_stat ? Cout / p.stat_G : 1;
  float a1[4], a2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a1[j] = a2[j] = 0.f;
  if (p.res_lo || p.comb_pyr) {
    // rare forms (pyramid heads, Combine): the shared per-element epilogue
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int y = y0 + wm * 8 + i, x = x0 + frow;
      const int64_t m = img + (int64_t)y * W + x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + fk * 4;
        float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
        const bool live = n < Cout;
        if (live) conv_epilogue4<TO>(p, m, b, n, v);
        if (do_stat && live) {
          a1[j] += (v[0] + v[1]) + (v[2] + v[3]);
          a2[j] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
      }
    }
  } else {
    f32x4 cb[4], ct[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = min(n0 + wn * 64 + j * 16 + fk * 4, Cout - 4);
      cb[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      ct[j] = p.tbias ? *reinterpret_cast<const f32x4*>(p.tbias + (int64_t)b * p.tbias_stride + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool has_b = p.bias != nullptr, has_t = p.tbias != nullptr, has_r = p.res != nullptr;
    const float scale = p.scale;
    constexpr bool WIDE = sizeof(TO) == 2;
    const bool wide = WIDE && (Cout & 7) == 0;
    // wide form: after the exchange the lane in DPP row fk owns channels [16 (2 q + (fk & 1)) + 8 (fk >> 1), + 8) of
    // pair q = tiles (2q, 2q+1)
    const int64_t pix0 = img + (int64_t)(y0 + wm * 8) * W + x0 + frow;
    constexpr int RB = WIDE ? 4 : 2;       // rows per batch of residual loads (f32: 16-byte vectors, 2 x 4 of them)
#pragma unroll
    for (int ih = 0; ih < 8 / RB; ++ih) {
      uint4 rw[RB][2];         // bf16: [row][pair] 16 bytes
      f32x4 rf[RB][4];         // f32:  [row][tile]
      if (has_r) {
        if constexpr (WIDE) {
          if (wide) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                const int n = min(n0 + wn * 64 + (2 * q + (fk & 1)) * 16 + (fk >> 1) * 8, Cout - 8);
                rw[i][q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const TO*>(p.res) + (pix0 + (int64_t)(ih * RB + i) * W) * Cout + n);
              }
          } else {
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int n = min(n0 + wn * 64 + j * 16 + fk * 4, Cout - 4);
                const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const TO*>(p.res) + (pix0 + (int64_t)(ih * RB + i) * W) * Cout + n);
                if (j & 1) { rw[i][j >> 1].z = t.x; rw[i][j >> 1].w = t.y; } else { rw[i][j >> 1].x = t.x; rw[i][j >> 1].y = t.y; }
              }
          }
        } else {
#pragma unroll
          for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int n = min(n0 + wn * 64 + j * 16 + fk * 4, Cout - 4);
              rf[i][j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + (pix0 + (int64_t)(ih * RB + i) * W) * Cout + n);
            }
        }
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int ii = ih * RB + i;
        const int64_t m = pix0 + (int64_t)ii * W;
        uint2 pk[4];           // bf16 output: the 4 packed channels of tile j
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          // residual of tiles 2q, 2q+1 in this lane's accumulator layout
          float r0[4] = {0.f, 0.f, 0.f, 0.f}, r1[4] = {0.f, 0.f, 0.f, 0.f};
          if (has_r) {
            if constexpr (WIDE) {
              uint2 ta, tb;
              if (wide) {
                const auto s0 = __builtin_amdgcn_permlane16_swap(rw[i][q].x, rw[i][q].z, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(rw[i][q].y, rw[i][q].w, false, false);
                ta = uint2{s0[0], s1[0]}; tb = uint2{s0[1], s1[1]};
              } else {
                ta = uint2{rw[i][q].x, rw[i][q].y}; tb = uint2{rw[i][q].z, rw[i][q].w};
              }
              const typename V16<TO>::x4 ea = *reinterpret_cast<const typename V16<TO>::x4*>(&ta), eb = *reinterpret_cast<const typename V16<TO>::x4*>(&tb);
#pragma unroll
              for (int r = 0; r < 4; ++r) { r0[r] = (float)ea[r]; r1[r] = (float)eb[r]; }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) { r0[r] = rf[i][2 * q][r]; r1[r] = rf[i][2 * q + 1][r]; }
            }
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int j = 2 * q + h;
            const int n = n0 + wn * 64 + j * 16 + fk * 4;
            float v[4] = {acc[j][ii][0], acc[j][ii][1], acc[j][ii][2], acc[j][ii][3]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (has_b) v[r] += cb[j][r];
              if (has_t) v[r] += ct[j][r];
              if (has_r) v[r] += h == 0 ? r0[r] : r1[r];
              v[r] *= scale;
            }
            if constexpr (WIDE) {
              typename V16<TO>::x4 t = {(TO)v[0], (TO)v[1], (TO)v[2], (TO)v[3]};
              pk[j] = *reinterpret_cast<uint2*>(&t);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = (float)t[r];            // statistics are those of the STORED tensor
              if (!wide && n < Cout) *reinterpret_cast<uint2*>(reinterpret_cast<TO*>(p.out) + m * Cout + n) = pk[j];
            } else {
              if (n < Cout) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + m * Cout + n) = f32x4{v[0], v[1], v[2], v[3]};
            }
            if (do_stat && n < Cout) {
              a1[j] += (v[0] + v[1]) + (v[2] + v[3]);
              a2[j] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
          }
          if constexpr (WIDE) {
            if (wide) {
              const auto s0 = __builtin_amdgcn_permlane16_swap(pk[2 * q].x, pk[2 * q + 1].x, false, false);
              const auto s1 = __builtin_amdgcn_permlane16_swap(pk[2 * q].y, pk[2 * q + 1].y, false, false);
              const int n = n0 + wn * 64 + (2 * q + (fk & 1)) * 16 + (fk >> 1) * 8;
              if (n < Cout)
                *reinterpret_cast<uint4*>(reinterpret_cast<TO*>(p.out) + m * Cout + n) = uint4{s0[0], s1[0], s0[1], s1[1]};
            }
          }
        }
      }
    }
  }
  RSTAMP(21, 0);
  if (do_stat) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fk * 4;
      const float r1 = row16_sum(a1[j]), r2 = row16_sum(a2[j]);
      if (frow == 0 && n < Cout) {
        atomicAdd(&s_stat[((n - n0) / scpg) * 2], (double)r1);
        atomicAdd(&s_stat[((n - n0) / scpg) * 2 + 1], (double)r2);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int g0 = n0 / scpg;
    const int ng = min(p.stat_G - g0, (128 + scpg - 1) / scpg);
    for (int i = tid; i < ng * 2; i += 256) {
      const int k = i & 1, g = g0 + (i >> 1);
      atomicAdd(p.stat_out + (((int64_t)b * p.stat_nsplit + tile % p.stat_nsplit) * p.stat_G + g) * 2 + k,
                s_stat[(g - g0) * 2 + k]);
    }
  }
  RSTAMP(22, 0);
  RSTAMP_RT(31, 0);
}

template <typename T, typename TO, bool GNP>
static int launch_ring(const ConvParams& p, hipStream_t st) {
  using namespace ring;
  constexpr int SMEM_MAX = GOFF + (GNP ? CONV_GN_MAXC * 8 : 0) + 64 * 8 + 64 * 4;
  const int SMEM = GOFF + (GNP ? ((p.gn_C + 63) & ~63) * 8 : 0) + 64 * 8 + 64 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ring_kernel<T, TO, GNP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_MAX);
    if (e != hipSuccess) {
      fdbm_set_error("fdbm_conv_igemm(ring): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  const int tiles_x = p.W / 16, tiles_y = p.H / 16;
  dim3 grid((unsigned)(p.B * tiles_x * tiles_y), (unsigned)((p.Cout + 127) / 128));
#ifdef FDBM_STAMPS
  ConvParams pd = p;
  { const char* e = getenv("FDBM_RING_DBG"); pd.ksplit = e ? atoi(e) : 0; }
  conv_ring_kernel<T, TO, GNP><<<grid, 512, SMEM, st>>>(pd, tiles_x, tiles_y);
#else
  conv_ring_kernel<T, TO, GNP><<<grid, 512, SMEM, st>>>(p, tiles_x, tiles_y);
#endif
  FDBM_LAUNCH_CHECK("fdbm_conv_igemm(ring)");
  return 0;
}

// Can this conv run on the ring kernel?  (shape / segment layout only; the caller checks the dtype)
bool fdbm_conv_ring_ok(const ConvParams& p) {
  if (p.H % 16 || p.W % 16 || p.nseg < 1 || p.seg[0].taps != 9) return false;
  bool seen1 = false;
  for (int s = 0; s < p.nseg; ++s) {
    if (p.seg[s].taps == 1) {
      seen1 = true;
      if (p.seg_gn[s] >= 0) return false;        // GroupNorm'd patches are staged over 8 intervals of a 9-tap chunk
    } else if (seen1) {
      return false;
    }
  }
  return true;
}

// called from fdbm_conv_igemm (conv.hip) once it has validated the arguments and filled ConvParams
int fdbm_launch_conv_ring(const ConvParams& p, int dt_in, int dt_out, hipStream_t st) {
  const bool gnp = p.gn_sums != nullptr;
  if (dt_in == FDBM_BF16 && dt_out == FDBM_BF16)
    return gnp ? launch_ring<bf16_t, bf16_t, true>(p, st) : launch_ring<bf16_t, bf16_t, false>(p, st);
  if (dt_in == FDBM_BF16 && dt_out == FDBM_F32)
    return gnp ? launch_ring<bf16_t, float, true>(p, st) : launch_ring<bf16_t, float, false>(p, st);
  if (dt_in == FDBM_F16 && dt_out == FDBM_F16)
    return gnp ? launch_ring<f16_t, f16_t, true>(p, st) : launch_ring<f16_t, f16_t, false>(p, st);
  if (dt_in == FDBM_F16 && dt_out == FDBM_F32)
    return gnp ? launch_ring<f16_t, float, true>(p, st) : launch_ring<f16_t, float, false>(p, st);
  fdbm_set_error("fdbm_conv_igemm(ring): unsupported dtypes %d -> %d", dt_in, dt_out);
  return 1;
}
