// conv.hip - 3x3 / 1x1 convolution and NIN as ONE implicit GEMM on the matrix cores.
//
//   out[m][n] = sum over k-steps  A_k[m][0..KC) . W_k[n][0..KC)
//     m = flattened (b, y, x) pixel of an NHWC tensor, n = output channel,
//     k-step = (segment, tap, channel chunk): a 128-byte slice of one source tensor's
//     channel axis at pixel (y+dy, x+dx), zero outside the image.
//
// Roofline: MFMA-bound (2*M*N*K flop against (M + N)*K*sizeof(T) operand bytes per tile).
// Tile: 128 pixels x BN channels per 256-thread workgroup (4 waves as 2 x 2, each wave
// 64 pixels x BN/2 channels = 4 x (BN/32) MFMA tiles of 16x16).  The WEIGHT tile is the
// MFMA A operand (rows = n) and the ACTIVATION tile the B operand (cols = m), so each
// lane ends up with 4 consecutive output channels of one pixel: 8/16-byte NHWC stores.
// Both operands are K-contiguous 128-byte rows in LDS; a lane's fragment is one
// ds_read_b128 (8 bf16 / 4 f32 along k).  16-byte chunk index is XOR-swizzled with
// (row>>1)&7 so the 16 lanes of every ds_read_b128 group hit 16 distinct bank slots.
// f32 path: v_mfma_f32_16x16x4_f32 consumes the 4 floats of the same b128 read as four
// successive k-groups {j, 4+j, 8+j, 12+j} (identical permutation on both operands).
// Pipeline: global loads run two k-steps ahead of the MFMAs (two register sets), are written
// to the other LDS buffer after the MFMAs of the step before them; one barrier per k-step.
// Small-M layers (the 4x4 ... 32x32 levels at batch 1) split the k-steps over grid.z and a
// second kernel sums the fp32 slabs and applies the epilogue (weights are streamed once).
#include <stdlib.h>

#include "conv_common.h"

// KG = k-groups: the workgroup has KG x 4 waves; group g runs the same 4-wave tile code on its
// own slice of the k-steps with its own LDS buffers, and the groups' accumulators are summed
// through LDS at the end (split-K inside the workgroup: KG x the loads in flight and MFMA issue
// per output tile, no slab traffic) - for layers with too few output tiles to fill the chip.
template <typename T, typename TO, int BM, int BN, bool GNP, int KG>
__global__ void __launch_bounds__(256 * KG) conv_igemm_kernel(const ConvParams p) {
  constexpr int KC = 128 / (int)sizeof(T);   // elements per k-step row
  constexpr int VW = 16 / (int)sizeof(T);    // elements per 16-byte chunk
  constexpr int WTM = BM / 2, WTN = BN / 2;  // wave tile (4 waves as 2 x 2)
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int AROWS = BM / 32, WROWS = BN / 32;   // 16-byte loads per thread per k-step
  constexpr int BUF = (BM + BN) * 128;
  constexpr bool F32 = sizeof(T) == 4;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  constexpr int NTHR = 256 * KG;
  const int grp = KG > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;   // wave-uniform
  const int tid = threadIdx.x & 255;            // thread within its k-group
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int H = p.H, W = p.W;
  const int HW = H * W;
  const int64_t M = (int64_t)p.B * HW;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  unsigned char* gsm = smem + grp * (2 * BUF);  // this group's two LDS buffers

  // ---- split-K: the block owns k-steps [zb, ze), this group [kbeg, kend) of them --------------
  const int kper = (p.nk + p.ksplit - 1) / p.ksplit;
  const int zb = blockIdx.z * kper;
  const int ze = min(p.nk, zb + kper);
  const int gper = (max(ze - zb, 0) + KG - 1) / KG;      // loop length, the same for every group
  const int kbeg = min(ze, zb + grp * gper);
  const int kend = min(ze, kbeg + gper);
  const int nloc = kend - kbeg;

  // ---- loader mapping: thread -> (16-byte chunk, rows lrow + 32 i) ------------------
  const int lchunk = tid & 7;
  const int lrow = tid >> 3;
  int py[AROWS], px[AROWS];
  int64_t pbase[AROWS];
  bool pval[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int64_t m = m0 + lrow + 32 * i;
    pval[i] = m < M;
    const int64_t mm = pval[i] ? m : 0;
    const int64_t b = mm / HW;
    const int rem = (int)(mm - b * HW);
    py[i] = rem / W;
    px[i] = rem - py[i] * W;
    pbase[i] = b * HW;
  }

  // ---- GroupNorm prologue table + output-statistics scratch (after the two LDS buffers) ----
  float* s_gn = reinterpret_cast<float*>(smem + KG * 2 * BUF);               // scale[nb][gn_C] | shift[nb][gn_C] (second half)
  double* s_stat = reinterpret_cast<double*>(smem + KG * 2 * BUF + (GNP ? CONV_MAX_NB * CONV_GN_MAXC * 8 : 0));  // [nb][32][2] doubles
  float* s_mr = reinterpret_cast<float*>(s_stat + CONV_MAX_NB * 64);                                        // [nb][32][2] mean, rstd
  const int b0 = (int)(m0 / HW);
  int pbl[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) pbl[i] = (int)(pbase[i] / HW) - b0;
  if (p.stat_out) {
    for (int i = threadIdx.x; i < CONV_MAX_NB * 64; i += NTHR) s_stat[i] = 0.0;
  }
  __syncthreads();

  uint4 areg0[AROWS], wreg0[WROWS], areg1[AROWS], wreg1[WROWS];   // two named register sets
  unsigned okm0 = 0, okm1 = 0;      // per set: which rows are real (in-image, in-range) data
  int gcb0 = -1, gcb1 = -1;         // per set: GN channel of this thread's chunk, or -1
  int ks = 0, ktap = 0, kc = 0;     // segment / tap / chunk of the NEXT k-step to load
  for (int i = 0; i < kbeg; ++i) {  // advance the cursor to this block's first k-step
    const int nchunks = (SEG_FIELD(p, ks, cin) + KC - 1) / KC;
    if (++kc == nchunks) {
      kc = 0;
      if (++ktap == SEG_FIELD(p, ks, taps)) { ktap = 0; ++ks; }
    }
  }

  // Loads are unconditional (clamped address, then a select): a branch around each load makes
  // hipcc wait for every load separately - one L2 round trip per row instead of one per k-step.
  auto load_regs = [&](auto SET, int kidx, bool adv) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const void* sg_src = SEG_FIELD(p, ks, src);
    const int sg_C = SEG_FIELD(p, ks, C), sg_coff = SEG_FIELD(p, ks, coff);
    const int sg_cin = SEG_FIELD(p, ks, cin), sg_taps = SEG_FIELD(p, ks, taps);
    const int dy = sg_taps == 9 ? ktap / 3 - 1 : 0;
    const int dx = sg_taps == 9 ? ktap % 3 - 1 : 0;
    const int cvalid = min(KC, sg_cin - kc * KC);
    const bool cok = lchunk * VW < cvalid;
    const T* src = reinterpret_cast<const T*>(sg_src) + sg_coff + (cok ? kc * KC + lchunk * VW : 0);
    unsigned okm = 0;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int iy = py[i] + dy, ix = px[i] + dx;
      const bool ok = cok && pval[i] && iy >= 0 && iy < H && ix >= 0 && ix < W;
      const int iyc = min(max(iy, 0), H - 1), ixc = min(max(ix, 0), W - 1);
      const uint4 v = *reinterpret_cast<const uint4*>(src + (pbase[i] + (int64_t)iyc * W + ixc) * sg_C);
      if constexpr (S == 0) areg0[i] = v; else areg1[i] = v;
      okm |= ok ? (1u << i) : 0u;
    }
    if constexpr (S == 0) okm0 = okm; else okm1 = okm;
    if constexpr (GNP) {
      const int sgn = ks == 0 ? p.seg_gn[0] : ks == 1 ? p.seg_gn[1] : ks == 2 ? p.seg_gn[2] : p.seg_gn[3];
      const int cb = sgn >= 0 ? sgn + kc * KC + lchunk * VW : -1;
      if constexpr (S == 0) gcb0 = cb; else gcb1 = cb;
    }
    const T* wp = reinterpret_cast<const T*>(p.w) + ((int64_t)kidx * p.CoutPad + n0 + lrow) * KC + lchunk * VW;
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
      const uint4 v = *reinterpret_cast<const uint4*>(wp + (int64_t)32 * i * KC);
      if constexpr (S == 0) wreg0[i] = v; else wreg1[i] = v;
    }
    if (adv) {
      const int nchunks = (sg_cin + KC - 1) / KC;
      if (++kc == nchunks) {
        kc = 0;
        if (++ktap == sg_taps) { ktap = 0; ++ks; }
      }
    }
  };

  auto write_lds = [&](auto SET, int buf) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    unsigned char* A = gsm + buf * BUF;
    unsigned char* Wt = A + BM * 128;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int row = lrow + 32 * i;
      uint4 v = (S == 0 ? areg0[i] : areg1[i]);
      const unsigned okm = (S == 0 ? okm0 : okm1);
      if constexpr (GNP) {
        const int cb = (S == 0 ? gcb0 : gcb1);
        if (cb >= 0) {             // wave-uniform per k-step (a property of the segment)
          const float* tsc = s_gn + pbl[i] * p.gn_C + cb;
          v = gn_transform16<T>(v, tsc, tsc + CONV_MAX_NB * CONV_GN_MAXC, p.gn_silu != 0);
        }
      }
      if (!((okm >> i) & 1u)) v = uint4{0u, 0u, 0u, 0u};     // zero padding AFTER the activation
      *reinterpret_cast<uint4*>(A + row * 128 + ((lchunk ^ ((row >> 1) & 7)) << 4)) = v;
    }
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
      const int row = lrow + 32 * i;
      *reinterpret_cast<uint4*>(Wt + row * 128 + ((lchunk ^ ((row >> 1) & 7)) << 4)) = (S == 0 ? wreg0[i] : wreg1[i]);
    }
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fk = lane >> 4;

  auto compute = [&](int buf) __attribute__((always_inline)) {
    const unsigned char* A = gsm + buf * BUF;
    const unsigned char* Wt = A + BM * 128;
    if constexpr (!F32) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int cidx = kk * 4 + fk;
        uint4 wf[NT], af[MT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int row = wn * WTN + j * 16 + frow;
          wf[j] = *reinterpret_cast<const uint4*>(Wt + row * 128 + ((cidx ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row = wm * WTM + i * 16 + frow;
          af[i] = *reinterpret_cast<const uint4*>(A + row * 128 + ((cidx ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) Mfma<T>::run(wf[j], af[i], acc[j][i]);
      }
    } else {
      // fp32 parity mode: two-level summation.  The 32 channels of this k-step are summed in a
      // fresh accumulator (an exact-fma chain of length 32 inside the MFMAs), which is then
      // added to the running sum: the rounding error grows like sqrt(32) + sqrt(#k-steps)
      // instead of sqrt(K) for one chain over K = 1152 ... 4608.
      f32x4 part[NT][MT];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) part[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int cidx = kk * 4 + fk;
        uint4 wf[NT], af[MT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int row = wn * WTN + j * 16 + frow;
          wf[j] = *reinterpret_cast<const uint4*>(Wt + row * 128 + ((cidx ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row = wm * WTM + i * 16 + frow;
          af[i] = *reinterpret_cast<const uint4*>(A + row * 128 + ((cidx ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              const float a = reinterpret_cast<const float*>(&wf[j])[q];
              const float b = reinterpret_cast<const float*>(&af[i])[q];
              part[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, part[j][i], 0, 0, 0);
            }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] += part[j][i];
    }
  };

  auto build_gn_table = [&]() __attribute__((always_inline)) {
    if constexpr (GNP) {
      const int64_t mlast = min(M, m0 + BM) - 1;
      const int nb = (int)(mlast / HW) - b0 + 1;
      // (scratch: the LDS tile buffers, not written before the barrier that follows)
      conv_gn_table<NTHR>(p, b0, nb, s_gn, CONV_MAX_NB * CONV_GN_MAXC, s_mr, smem);
    }
  };

  // ---- main loop: loads run two k-steps ahead of the MFMAs ------------------------------
  //   iteration t:  issue loads of step t+2 -> regs[t&1] ; MFMAs on LDS buf[t&1] ;
  //                 regs[(t+1)&1] (issued one iteration ago) -> LDS buf[(t+1)&1] ; barrier
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // every k-group executes the same number of barriers (gper steps); a group with fewer (or no)
  // k-steps of its own just skips the work between them
  // Loads are issued UNCONDITIONALLY every half-iteration (past the group's last k-step they
  // re-read that last step, and nobody consumes them): with loads under a run-time condition the
  // compiler cannot count what is in flight and waits vmcnt(0), which kills the 2-deep prefetch.
  const int klast = max(kend - 1, 0);               // (a group without k-steps reads step 0 of the conv)
  const int kfirst = min(kbeg, klast);
  {   // cursor -> kfirst (it was advanced to kbeg, which may be one past the end for an idle group)
    if (kbeg > klast && kbeg > 0) { ks = 0; ktap = 0; kc = 0; for (int i = 0; i < kfirst; ++i) {
      const int nchunks = (SEG_FIELD(p, ks, cin) + KC - 1) / KC;
      if (++kc == nchunks) { kc = 0; if (++ktap == SEG_FIELD(p, ks, taps)) { ktap = 0; ++ks; } } } }
  }
  load_regs(S0{}, kfirst, kfirst < klast);
  load_regs(S1{}, min(kfirst + 1, klast), kfirst + 1 < klast);
  build_gn_table();            // its global reads overlap the first two k-steps' loads
  __syncthreads();
  if (nloc > 0) write_lds(S0{}, 0);
  __syncthreads();
  for (int t = 0; t < gper; t += 2) {
    load_regs(S0{}, min(kbeg + t + 2, klast), kbeg + t + 2 < klast);
    if (t < nloc) compute(0);
    if (t + 1 < nloc) write_lds(S1{}, 1);
    __syncthreads();
    load_regs(S1{}, min(kbeg + t + 3, klast), kbeg + t + 3 < klast);
    if (t + 1 < nloc) compute(1);
    if (t + 2 < nloc) write_lds(S0{}, 0);
    __syncthreads();
  }

  // ---- sum the k-groups' accumulators through LDS (each group's own buffers, now idle) --------
  if constexpr (KG > 1) {
    f32x4* red = reinterpret_cast<f32x4*>(gsm);          // [NT*MT][256] f32x4 = 2*BUF bytes exactly
    if (grp > 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) red[(j * MT + i) * 256 + tid] = acc[j][i];
    }
    __syncthreads();
    if (grp == 0)                                         // group 0 owns the epilogue
#pragma unroll
    for (int g = 1; g < KG; ++g) {
      const f32x4* rg = reinterpret_cast<const f32x4*>(smem + g * (2 * BUF));
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] += rg[(j * MT + i) * 256 + tid];
    }
  }

  // ---- epilogue (k-group 0 only; the other groups just keep the barriers company) -----------
  const bool owner = grp == 0;
  const int Cout = p.Cout;
  if (p.ksplit > 1) {
    if (!owner) return;
    // raw fp32 partial sums -> slab [z][M][Cout]; fdbm's reduce kernel applies the epilogue
    float* slab = p.partial + (int64_t)blockIdx.z * M * Cout;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int64_t m = m0 + wm * WTM + i * 16 + frow;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * WTN + j * 16 + fk * 4;
        if (n >= Cout) continue;
        *reinterpret_cast<f32x4*>(slab + m * Cout + n) = acc[j][i];
      }
    }
    return;
  }
  const bool do_stat = p.stat_out != nullptr;
  const int scpg = do_stat ? Cout / p.stat_G : 1;
  // all m-tiles of this wave in one image (H*W a multiple of the wave tile): sum over them first
  const bool one_img = (HW % WTM) == 0;
  float a1[NT], a2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) a1[j] = a2[j] = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int64_t mt0 = m0 + wm * WTM + i * 16;          // first pixel of this 16-pixel m-tile
    const int64_t m = mt0 + frow;
    const int64_t b = (m < M ? m : M - 1) / HW;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * WTN + j * 16 + fk * 4;
      float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
      const bool live = owner && m < M && n < Cout;
      if (live) conv_epilogue4<TO>(p, m, b, n, v);
      if (do_stat) {
        const float s1 = live ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
        const float s2 = live ? (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]) : 0.f;
        if (one_img) {
          a1[j] += s1;
          a2[j] += s2;
        } else {
          // the 16 pixels of an m-tile lie in one image (H*W % 16 == 0)
          const float r1 = row16_sum(s1), r2 = row16_sum(s2);
          if (owner && frow == 0 && mt0 < M && n < Cout) {
            const int bl = (int)(mt0 / HW) - b0;
            atomicAdd(&s_stat[(bl * 32 + (n - n0) / scpg) * 2], (double)r1);
            atomicAdd(&s_stat[(bl * 32 + (n - n0) / scpg) * 2 + 1], (double)r2);
          }
        }
      }
    }
  }
  if (do_stat && one_img) {
    const int64_t mw = m0 + wm * WTM;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * WTN + j * 16 + fk * 4;
      const float r1 = row16_sum(a1[j]), r2 = row16_sum(a2[j]);
      if (owner && frow == 0 && mw < M && n < Cout) {
        const int bl = (int)(mw / HW) - b0;
        atomicAdd(&s_stat[(bl * 32 + (n - n0) / scpg) * 2], (double)r1);
        atomicAdd(&s_stat[(bl * 32 + (n - n0) / scpg) * 2 + 1], (double)r2);
      }
    }
  }
  if (do_stat) {
    __syncthreads();
    const int64_t mlast = min(M, m0 + BM) - 1;
    const int nb = (int)(mlast / HW) - b0 + 1;
    const int g0 = n0 / scpg;
    const int ng = min(p.stat_G - g0, (BN + scpg - 1) / scpg);
    if (owner)
    for (int i = tid; i < nb * ng * 2; i += 256) {
      const int k = i & 1, gl = (i >> 1) % ng, g = g0 + gl, bl = (i >> 1) / ng;
      // spread over stat_nsplit rows: hundreds of blocks adding into ONE row serialise at memory
      atomicAdd(p.stat_out + (((int64_t)(b0 + bl) * p.stat_nsplit + blockIdx.x % p.stat_nsplit) * p.stat_G + g) * 2 + k,
                s_stat[(bl * 32 + gl) * 2 + k]);
    }
  }
}

// split-K reduction + epilogue: out = epilogue(sum_z slab[z]); one image per blockIdx.y so the
// output statistics of a block belong to one (image, group) row.
template <typename TO>
__global__ void __launch_bounds__(256) conv_splitk_reduce_kernel(const ConvParams p) {
  __shared__ double s_stat[128];         // up to 64 groups (Cout / 4 with Cout <= 256)
  const int Cout = p.Cout;
  const int HW = p.H * p.W;
  const int64_t M = (int64_t)p.B * HW;
  const int nv = Cout / 4;
  const int b = blockIdx.y;
  const bool do_stat = p.stat_out != nullptr;
  const int scpg = do_stat ? Cout / p.stat_G : 1;
  if (do_stat) {
    if (threadIdx.x < 128) s_stat[threadIdx.x] = 0.0;
    __syncthreads();
  }
  const int total = HW * nv;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int n = (i % nv) * 4;
    const int64_t m = (int64_t)b * HW + i / nv;
    // four independent partial sums: the slab loads of one output are all in flight together
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    const float* src = p.partial + m * Cout + n;
    const int64_t zs = M * Cout;
    int z = 0;
    for (; z + 3 < p.ksplit; z += 4) {
      a0 += *reinterpret_cast<const f32x4*>(src + (z + 0) * zs);
      a1 += *reinterpret_cast<const f32x4*>(src + (z + 1) * zs);
      a2 += *reinterpret_cast<const f32x4*>(src + (z + 2) * zs);
      a3 += *reinterpret_cast<const f32x4*>(src + (z + 3) * zs);
    }
    for (; z < p.ksplit; ++z) a0 += *reinterpret_cast<const f32x4*>(src + z * zs);
    const f32x4 a = (a0 + a1) + (a2 + a3);
    float v[4] = {a[0], a[1], a[2], a[3]};
    conv_epilogue4<TO>(p, m, b, n, v);
    if (do_stat) {
      const int g = n / scpg;
      atomicAdd(&s_stat[g * 2], (double)((v[0] + v[1]) + (v[2] + v[3])));
      atomicAdd(&s_stat[g * 2 + 1], (double)((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])));
    }
  }
  if (do_stat) {
    __syncthreads();
    if (threadIdx.x < p.stat_G * 2)
      atomicAdd(p.stat_out + ((int64_t)b * p.stat_nsplit + blockIdx.x % p.stat_nsplit) * p.stat_G * 2 + threadIdx.x,
                s_stat[threadIdx.x]);
  }
}

template <typename T, typename TO, int BM, int BN, bool GNP, int KG>
static int launch_conv(const ConvParams& p, hipStream_t st) {
  constexpr int SMEM = KG * 2 * (BM + BN) * 128 + (GNP ? CONV_MAX_NB * CONV_GN_MAXC * 8 : 0) + CONV_MAX_NB * 64 * 8 + CONV_MAX_NB * 64 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, TO, BM, BN, GNP, KG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) {
      fdbm_set_error("fdbm_conv_igemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  const int64_t M = (int64_t)p.B * p.H * p.W;
  dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((p.Cout + BN - 1) / BN), (unsigned)p.ksplit);
  conv_igemm_kernel<T, TO, BM, BN, GNP, KG><<<grid, 256 * KG, SMEM, st>>>(p);
  FDBM_LAUNCH_CHECK("fdbm_conv_igemm");
  if (p.ksplit > 1) {
    const int total = p.H * p.W * (p.Cout / 4);
    int g = (total + 255) / 256;
    if (g > 256) g = 256;
    conv_splitk_reduce_kernel<TO><<<dim3(g, p.B), 256, 0, st>>>(p);
    FDBM_LAUNCH_CHECK("fdbm_conv_igemm/reduce");
  }
  return 0;
}

template <typename T, typename TO, bool GNP>
static int launch_conv_gnp(const ConvParams& p, int bm, int bn, int kg, hipStream_t st) {
  if (kg == 4) return launch_conv<T, TO, 64, 64, GNP, 4>(p, st);            // only the 64 x 64 tile has 4 k-groups
  if (bm == 128 && bn == 128) return launch_conv<T, TO, 128, 128, GNP, 1>(p, st);
  if (bm == 128 && bn == 64) return launch_conv<T, TO, 128, 64, GNP, 1>(p, st);
  if (bm == 64 && bn == 128) return launch_conv<T, TO, 64, 128, GNP, 1>(p, st);
  return launch_conv<T, TO, 64, 64, GNP, 1>(p, st);
}

template <typename T, typename TO>
static int launch_conv_tile(const ConvParams& p, int bm, int bn, int kg, hipStream_t st) {
  return p.gn_sums ? launch_conv_gnp<T, TO, true>(p, bm, bn, kg, st) : launch_conv_gnp<T, TO, false>(p, bm, bn, kg, st);
}

extern "C" int fdbm_conv_kc(int dtype) { return dtype != FDBM_F32 ? 64 : 32; }

// Tile / split-K plan for a conv of M pixels, Cout channels, nk k-steps (host side, also used
// by the caller to size the split-K workspace): fills bm, bn, ksplit.
// k-groups inside the workgroup (see conv_igemm_kernel): 4 for the 64 x 64 tile when the layer has
// too few tiles to give every CU two workgroups and enough k-steps to share out (opt-in).
// Kernel-selection policy: bit 0 = halo-patch kernel allowed, bit 1 = wave-per-tap kernel allowed,
// bit 2 = k-groups in the tap-outer kernel, bit 3 = producer/consumer ring kernel (conv_ring.hip) in place of the
// halo-patch kernel where it applies, bit 4 = 8-row ring tiles wherever they fit (tests), bit 5 = whole-map kernel
// (conv_small.hip) in place of the wave-per-tap kernel on the smallest maps.  Default 43, or from the environment
// (experiments): FDBM_CONV_PATCH=0, FDBM_CONV_TAP=0, FDBM_CONV_RING=0, FDBM_CONV_SMALL=0 clear a bit, FDBM_CONV_KG=4 sets bit 2.
static int g_policy = -1;
static int g_last_kind = -1;      // kernel family of the most recent fdbm_conv_igemm launch (see fdbm_conv_last_kind)
static int conv_policy() {
  if (g_policy < 0) {
    int m = 43;
    const char* e;
    if ((e = getenv("FDBM_CONV_SMALL")) && e[0] == '0') m &= ~32;
    if ((e = getenv("FDBM_CONV_PATCH")) && e[0] == '0') m &= ~1;
    if ((e = getenv("FDBM_CONV_RING")) && e[0] == '0') m &= ~8;
    if ((e = getenv("FDBM_CONV_TAP")) && e[0] == '0') m &= ~2;
    if ((e = getenv("FDBM_CONV_KG")) && e[0] == '4') m |= 4;
    g_policy = m;
  }
  return g_policy;
}
// 0 = tap-outer implicit GEMM, 1 = halo-patch, 2 = wave-per-tap, 3 = producer/consumer ring on 16 x 16 pixel tiles,
// 4 = the same on 8 x 16 pixel tiles, 5 = the 4-channel head kernel, 6 = whole-map kernel of the smallest maps; -1 before
// the first call.
// For measurement harnesses (bench.py prices each kernel family against its roofline).
extern "C" int fdbm_conv_last_kind(void) { return g_last_kind; }

extern "C" int fdbm_conv_policy(int mask) {
  const int old = conv_policy();
  if (mask >= 0) g_policy = mask & 63;
  return old;
}

static int plan_kgroups(int64_t nblocks, int bm, int bn, int nk) {
  // Measured on MI355X (B=1, ncsnpp_v2): 1.3-1.6x on isolated 64x64-map layers, but opt-in (bit 2):
  // the wave-per-tap kernel covers those layers by default.
  if (!(conv_policy() & 4)) return 1;
  // only where the tiles alone already cover half the CUs: below that split-K over MORE workgroups wins
  return (bm == 64 && bn == 64 && nblocks >= 128 && nblocks <= 256 && nk >= 8) ? 4 : 1;
}

extern "C" int fdbm_conv_plan(int64_t M, int Cout, int nk, int* bm, int* bn, int* ksplit) {
  const int target = 512;                         // blocks wanted in flight (2 per CU)
  int BMs = 128, BNs = Cout <= 64 ? 64 : 128;
  auto blocks = [&](int a, int b) { return ((M + a - 1) / a) * ((Cout + b - 1) / b); };
  if (blocks(BMs, BNs) < target && BNs == 128 && Cout > 64) BNs = 64;
  if (blocks(BMs, BNs) < target) BMs = 64;
  int ks = 1;
  const int64_t nb = blocks(BMs, BNs);
  const int kg = plan_kgroups(nb, BMs, BNs, nk);
  if (nb * kg < 256 && nk >= 4 * kg) {
    ks = (int)(target / (nb * kg));
    if (ks > nk / (2 * kg)) ks = nk / (2 * kg);
    // keep the fp32 slabs small: ks * M * Cout * 4 bytes <= 8 MiB
    const int64_t per = M * Cout * 4;
    while (ks > 1 && ks * per > (8 << 20)) --ks;
    if (ks < 1) ks = 1;
  }
  *bm = BMs; *bn = BNs; *ksplit = ks;
  return 0;
}

int fdbm_launch_conv_patch(const ConvParams& p, int dt_in, int dt_out, int th, hipStream_t st);   // conv_patch.hip
int fdbm_launch_conv_tap(const ConvParams& p, int dt_in, int dt_out, int tw, int nt, hipStream_t st);   // conv_tap.hip
int fdbm_launch_conv_ring(const ConvParams& p, int dt_in, int dt_out, hipStream_t st);                  // conv_ring.hip
bool fdbm_conv_ring_ok(const ConvParams& p, int rows);
int fdbm_launch_conv_ring8(const ConvParams& p, int dt_in, int dt_out, hipStream_t st);                 // conv_ring8.hip
int fdbm_launch_conv_head(const ConvParams& p, int dt_in, hipStream_t st);                               // conv_head.hip
int fdbm_launch_conv_small(const ConvParams& p, int dt_in, int dt_out, hipStream_t st);                  // conv_small.hip
bool fdbm_conv_small_ok(const ConvParams& p, bool f32_out);
int fdbm_launch_conv_mid(const ConvParams& p, int dt_in, int dt_out, hipStream_t st);                    // conv_mid.hip
bool fdbm_conv_mid_ok(const ConvParams& p);
int fdbm_launch_conv_small_split(const ConvParams& p, hipStream_t st);                                   // conv_small_split.hip
bool fdbm_conv_small_split_ok(const ConvParams& p);
bool fdbm_conv_head_ok(const ConvParams& p);

// Which kernel runs a conv of this shape: kind 1 = halo-patch 3x3 kernel (conv_patch.hip, tile
// th x 16 pixels x 128 channels), kind 2 = wave-per-tap 3x3 kernel for small grids (conv_tap.hip,
// tile 16 pixels x 1|4 m-tiles, th = tile width 16|8|4, bn = 16 x n-tiles), kind 0 = tap-outer
// implicit GEMM (this file).
extern "C" int fdbm_conv_plan_ex(int B, int H, int W, int Cout, int nk, int first_taps, int* kind,
                                 int* th, int* bm, int* bn, int* ksplit) {
  const int64_t M = (int64_t)B * H * W;
  fdbm_conv_plan(M, Cout, nk, bm, bn, ksplit);
  *kind = 0;
  *th = 0;
  const int64_t tiles16 = (int64_t)B * (H / 16) * (W / 16) * ((Cout + 127) / 128);
  // the halo-patch kernel wants >= 1 tile per CU (1-2 workgroups fit a CU); below that the
  // tap-outer kernel's smaller tiles fill the chip better
  static const char* pmin = getenv("FDBM_PATCH_MIN_TILES");  // experiments
  const int64_t min_tiles = pmin ? atoi(pmin) : 128;
  const bool patch_ok = (conv_policy() & 1) && first_taps == 9 && H % 8 == 0 && W % 16 == 0 && *ksplit == 1 && tiles16 >= min_tiles;
  // wave-per-tap kernel: its tile shape and grid for this layer
  int tw = 0, tr = 0, tap_nt = 0;
  if ((first_taps == 9 || first_taps == 1) && (conv_policy() & 2)) {   // (1x1: the shared-tap path alone)
    if (W % 16 == 0 && H % 4 == 0) { tw = 16; tr = 4; }
    else if (W % 8 == 0 && H % 8 == 0) { tw = 8; tr = 8; }
    else if (W % 4 == 0 && H % 4 == 0) { tw = 4; tr = 4; }
    if (tw) {
      const int64_t tiles = (int64_t)B * (H / tr) * (W / tw);
      auto grid = [&](int nt) { return tiles * ((Cout + 16 * nt - 1) / (16 * nt)); };
      // latency-bound regime only: with more workgroups than that the weight fragments, which every
      // workgroup reads from L2 by itself, cost more than the tap-outer kernel's shared LDS tile
      static const char* tmax = getenv("FDBM_TAP_MAX_GRID");   // experiments
      if (grid(4) <= (tmax ? atoi(tmax) : 1024)) {
        static const char* fnt = getenv("FDBM_TAP_NT");         // experiments: force the n-tiles of the small grids
        tap_nt = (Cout >= 64 && grid(4) >= 192) ? 4 : (Cout >= 32 && grid(2) >= 192) ? 2 : 1;
        if (fnt && tap_nt == 1 && Cout >= 64) tap_nt = atoi(fnt);
      }
    }
  }
  // (a conv with <= 16 output channels and f32 output - the 4-channel heads - runs the halo-patch kernel on 16-channel
  // tiles: fdbm_launch_conv_patch)
  static const char* headtap = getenv("FDBM_HEAD_TAP");          // experiments: "1" = the previous choice
  if (patch_ok && !(Cout <= 16 && tap_nt && headtap && headtap[0] == '1')) {
    *kind = 1;
    *th = (H % 16 == 0 && tiles16 >= 256) ? 16 : 8;
    return 0;
  }
  if (tap_nt) { *kind = 2; *th = tw; *bm = 16; *bn = 16 * tap_nt; *ksplit = 1; }
  return 0;
}

extern "C" int fdbm_conv_igemm(const fdbm_conv_args* a, void* stream) {
  FDBM_CHECK(a, "fdbm_conv_igemm: null args");
  FDBM_CHECK(a->nseg >= 1 && a->nseg <= FDBM_MAX_SEG, "fdbm_conv_igemm: nseg=%d out of range", a->nseg);
  FDBM_CHECK(a->w && a->out, "fdbm_conv_igemm: null weight/output pointer");
  FDBM_CHECK(a->dt_in == FDBM_F32 || a->dt_in == FDBM_BF16 || a->dt_in == FDBM_F16, "fdbm_conv_igemm: bad dt_in %d", a->dt_in);
  FDBM_CHECK(a->dt_out == FDBM_F32 || a->dt_out == a->dt_in, "fdbm_conv_igemm: bad dt_out %d", a->dt_out);
  FDBM_CHECK(!(a->dt_in == FDBM_F32 && a->dt_out != FDBM_F32), "fdbm_conv_igemm: f32 in / bf16 out is not built");
  FDBM_CHECK(a->mma_mode == 0 || (a->mma_mode == 1 && a->dt_in == FDBM_F32 && a->dt_out == FDBM_F32 && a->acc_scale > 0.f),
             "fdbm_conv_igemm: mma_mode %d needs f32 tensors and a positive acc_scale", a->mma_mode);
  FDBM_CHECK(a->Cout > 0 && a->Cout % 4 == 0, "fdbm_conv_igemm: Cout=%d must be a positive multiple of 4", a->Cout);
  FDBM_CHECK(a->CoutPad >= a->Cout && a->CoutPad % 128 == 0, "fdbm_conv_igemm: CoutPad=%d must be a multiple of 128 >= Cout", a->CoutPad);
  FDBM_CHECK(a->B > 0 && a->H > 0 && a->W > 0, "fdbm_conv_igemm: bad shape B=%d H=%d W=%d", a->B, a->H, a->W);
  const int kc = fdbm_conv_kc(a->dt_in);
  const int vw = a->dt_in != FDBM_F32 ? 8 : 4;
  ConvParams p;
  memset(&p, 0, sizeof(p));
  int nk = 0;
  for (int s = 0; s < a->nseg; ++s) {
    const fdbm_conv_seg& sg = a->seg[s];
    FDBM_CHECK(sg.src, "fdbm_conv_igemm: segment %d has a null source", s);
    FDBM_CHECK(sg.taps == 1 || sg.taps == 9, "fdbm_conv_igemm: segment %d taps=%d (must be 1 or 9)", s, sg.taps);
    FDBM_CHECK(sg.cin > 0 && sg.cin % vw == 0 && sg.coff % vw == 0 && sg.C % vw == 0 && sg.coff + sg.cin <= sg.C,
               "fdbm_conv_igemm: segment %d channel slice (C=%d coff=%d cin=%d) must be %d-aligned and inside the tensor",
               s, sg.C, sg.coff, sg.cin, vw);
    p.seg[s] = sg;
    nk += sg.taps * ((sg.cin + kc - 1) / kc);
  }
  p.nseg = a->nseg;
  p.w = a->w; p.bias = a->bias; p.tbias = a->tbias; p.tbias_stride = a->tbias_stride;
  p.res = a->res; p.res_lo = a->res_up2x; p.scale = a->scale; p.out = a->out;
  FDBM_CHECK(!a->res_up2x || (a->H % 2 == 0 && a->W % 2 == 0), "fdbm_conv_igemm: res_up2x needs even H, W (got %d x %d)", a->H, a->W);
  p.B = a->B; p.H = a->H; p.W = a->W; p.Cout = a->Cout; p.CoutPad = a->CoutPad;
  p.nk = nk;
  p.mma_split = a->mma_mode == 1;
  p.acc_scale = a->mma_mode == 1 ? a->acc_scale : 1.0f;
  int bm, bn, ks, kind, th;
  const int64_t M = (int64_t)a->B * a->H * a->W;
  fdbm_conv_plan_ex(a->B, a->H, a->W, a->Cout, nk, a->seg[0].taps, &kind, &th, &bm, &bn, &ks);
  if (kind == 2) {                 // wave-per-tap kernel: 9-tap segments first, then 1-tap ones
    bool seen1 = false, ordered = a->w_frag != nullptr;       // ... and fragment-major weights
    for (int s = 0; s < a->nseg; ++s) {
      if (a->seg[s].taps == 1) seen1 = true;
      else if (seen1) ordered = false;
    }
    if (!ordered) {
      kind = 0;
      fdbm_conv_plan(M, a->Cout, nk, &bm, &bn, &ks);
    }
  }
  FDBM_CHECK(!(p.mma_split && kind == 0), "fdbm_conv_igemm: the split-precision mode (mma_mode 1) is built for plan kinds 1 and 2 only "
             "(this shape takes the tap-outer kernel: pass plain f32 weights with mma_mode 0)");
  if (kind != 0) bm = 16;          // a patch / tap tile always lies inside one image
  if (!a->workspace || a->workspace_bytes <= 0) ks = 1;
  while (ks > 1 && (int64_t)ks * M * a->Cout * 4 > a->workspace_bytes) --ks;
  p.ksplit = ks;
  p.partial = reinterpret_cast<float*>(a->workspace);
  const int HW = a->H * a->W;
  for (int s = 0; s < FDBM_MAX_SEG; ++s) p.seg_gn[s] = -1;
  const bool gn_units = a->gn_seg_sums[0] != nullptr;
  FDBM_CHECK(!(gn_units && a->gn_sums), "fdbm_conv_igemm: give gn_sums OR gn_seg_sums, not both");
  if (a->gn_sums || gn_units) {
    FDBM_CHECK(a->gn_gamma && a->gn_beta && a->gn_G > 0 && a->gn_G <= 32 && a->gn_C > 0 && a->gn_C <= CONV_GN_MAXC &&
               a->gn_C % a->gn_G == 0 && (gn_units || a->gn_nsplit != 0) && a->gn_count > 0,
               "fdbm_conv_igemm: bad GroupNorm prologue arguments (G=%d C=%d nsplit=%d)", a->gn_G, a->gn_C, a->gn_nsplit);
    FDBM_CHECK(HW % 16 == 0 && (HW % bm == 0 || (bm % HW == 0 && bm / HW <= CONV_MAX_NB)),
               "fdbm_conv_igemm: GroupNorm prologue needs H*W (%d) to tile the %d-pixel M tile", HW, bm);
    int coff = 0;
    for (int s = 0; s < a->nseg; ++s)
      if (a->seg_gn_mask & (1u << s)) { p.seg_gn[s] = coff; coff += a->seg[s].cin; }
    FDBM_CHECK(coff == a->gn_C, "fdbm_conv_igemm: GroupNorm channels %d != flagged segment channels %d", a->gn_C, coff);
    if (gn_units) {
      FDBM_CHECK((a->seg_gn_mask & (a->seg_gn_mask + 1)) == 0, "fdbm_conv_igemm: unit statistics need the flagged segments to be 0 .. n-1");
      FDBM_CHECK((a->gn_C / a->gn_G) % 4 == 0, "fdbm_conv_igemm: unit statistics need gn_C/gn_G (%d/%d) to be a multiple of 4",
                 a->gn_C, a->gn_G);
      int uoff = 0;
      for (int s = 0; s < a->nseg; ++s) {
        if (!(a->seg_gn_mask & (1u << s))) continue;
        FDBM_CHECK(a->gn_seg_sums[s] && a->gn_seg_nsplit[s] >= 1 && a->seg[s].cin % 4 == 0,
                   "fdbm_conv_igemm: segment %d needs unit statistics (pointer, nsplit >= 1, cin %% 4 == 0)", s);
        p.gn_useg[s] = a->gn_seg_sums[s]; p.gn_unsp[s] = a->gn_seg_nsplit[s];
        p.gn_uoff[s] = uoff; p.gn_ucnt[s] = a->seg[s].cin / 4;
        uoff += a->seg[s].cin / 4;
      }
      p.gn_unit = 1;
      p.gn_sums = reinterpret_cast<const float*>(a->gn_seg_sums[0]);   // non-null = "prologue on" for the launchers
    } else {
      p.gn_sums = a->gn_sums;
    }
    p.gn_gamma = a->gn_gamma; p.gn_beta = a->gn_beta;
    p.gn_nsplit = a->gn_nsplit; p.gn_G = a->gn_G; p.gn_C = a->gn_C; p.gn_silu = a->gn_silu;
    {
      // split-precision mode: SiLU from the hardware exp2 / rcp (experiments: FDBM_SPLIT_SILU=precise = expf + division)
      static const char* ss = getenv("FDBM_SPLIT_SILU");
      if (p.mma_split && p.gn_silu && !(ss && ss[0] == 'p')) p.gn_silu = 2;
    }
    p.gn_inv_count = 1.0 / (double)a->gn_count; p.gn_eps = a->gn_eps;
  }
  if (a->comb_pyr) {
    FDBM_CHECK(a->comb_w && a->comb_b, "fdbm_conv_igemm: Combine epilogue needs comb_w and comb_b");
    p.comb_pyr = a->comb_pyr; p.comb_w = a->comb_w; p.comb_b = a->comb_b;
  }
  if (a->stat_out) {
    FDBM_CHECK(a->stat_G > 0 && a->stat_G <= 64 && a->Cout % a->stat_G == 0 && (a->Cout / a->stat_G) % 4 == 0,
               "fdbm_conv_igemm: output statistics need Cout/G (%d/%d) to be a multiple of 4", a->Cout, a->stat_G);
    FDBM_CHECK(HW % 16 == 0 && (HW % bm == 0 || (bm % HW == 0 && bm / HW <= CONV_MAX_NB)),
               "fdbm_conv_igemm: output statistics need H*W (%d) to tile the %d-pixel M tile", HW, bm);
    FDBM_CHECK(a->stat_nsplit >= 1, "fdbm_conv_igemm: stat_nsplit must be >= 1");
    p.stat_out = a->stat_out; p.stat_G = a->stat_G; p.stat_nsplit = a->stat_nsplit;
  }
  hipStream_t st = (hipStream_t)stream;
  if ((conv_policy() & 8) && (conv_policy() & 1) && a->dt_in != FDBM_F32 && a->dt_out == a->dt_in) {
    // producer / consumer ring kernel: one 512-thread workgroup per CU on a 16 x 16 pixel x 128 channel tile; wants
    // (nearly) a tile per CU.  Below that the same kernel on 8 x 16 pixel tiles: twice the workgroups, half the serial
    // work in each (these launches are latency-bound); below THAT the wave-per-tap kernel's small tiles fill the chip better.
    // (policy bit 16: 8-row tiles wherever they fit - tests)
    static const char* rmin = getenv("FDBM_RING_MIN_TILES");    // experiments
    static const char* rmin8 = getenv("FDBM_RING8_MIN_TILES");
    const int64_t nb = (int64_t)a->B * (a->W / 16) * ((a->Cout + 127) / 128);
    const bool force8 = (conv_policy() & 16) != 0;
    if (!force8 && fdbm_conv_ring_ok(p, 16) && nb * (a->H / 16) >= (rmin ? atoi(rmin) : 200)) {
      g_last_kind = 3;
      return fdbm_launch_conv_ring(p, a->dt_in, a->dt_out, st);
    }
    if (fdbm_conv_ring_ok(p, 8) && (force8 || nb * (a->H / 8) >= (rmin8 ? atoi(rmin8) : 128))) {
      g_last_kind = 4;
      return fdbm_launch_conv_ring8(p, a->dt_in, a->dt_out, st);
    }
  }
  if (kind == 1 && a->dt_in != FDBM_F32 && a->dt_out == FDBM_F32 && fdbm_conv_head_ok(p)) {
    // the 4-channel f32 heads: all weights resident in LDS, workgroups walk tiles (conv_head.hip)
    static const char* hoff = getenv("FDBM_CONV_HEAD");          // experiments: "0" = the 16-channel halo-patch tiles
    if (!(hoff && hoff[0] == '0')) { g_last_kind = 5; return fdbm_launch_conv_head(p, a->dt_in, st); }
  }
  g_last_kind = kind;
  if (kind == 1) {
    // 8-row tiles let two workgroups share a CU (their load / epilogue phases then overlap the other's
    // MFMAs) as long as the GroupNorm table stays within 2 KiB: measured +2...11 % over 16-row tiles
    static const char* fth = getenv("FDBM_PATCH_TH");          // experiments: "16" keeps the plan's choice
    if (!(fth && fth[0] == '1') && (!a->gn_sums && !gn_units ? true : a->gn_C <= 256)) th = 8;
    return fdbm_launch_conv_patch(p, a->dt_in, a->dt_out, th, st);
  }
  if (kind == 2 && (conv_policy() & 32) && p.mma_split && a->w_frag) {
    // the same maps in the split-precision parity mode: f32 tensors, (hi, lo) f16 operand pairs (conv_small_split.hip)
    ConvParams ps = p;
    ps.w = a->w_frag;
    ps.ksplit = 1;
    if (fdbm_conv_small_split_ok(ps)) { g_last_kind = 6; return fdbm_launch_conv_small_split(ps, st); }
  }
  if (kind == 2 && (conv_policy() & 32) && a->dt_in != FDBM_F32) {
    // the smallest maps (4 x 4, 8 x 8 at batch 1): whole map per workgroup, GroupNorm statistics by the consumer (conv_small.hip)
    ConvParams ps = p;
    ps.w = a->w_frag;
    ps.ksplit = 1;
    ps.partial = reinterpret_cast<float*>(a->acc_ws);          // (diagnostic stamps only)
    if (fdbm_conv_small_ok(ps, a->dt_out == FDBM_F32)) { g_last_kind = 6; return fdbm_launch_conv_small(ps, a->dt_in, a->dt_out, st); }
    // the 64 x 64 level at batch 1: the wave-per-tap kernel's 4 x 16 pixel x 64 channel tile with that launch body (conv_mid.hip)
    static const char* moff = getenv("FDBM_CONV_MID");          // experiments: "0" = the wave-per-tap kernel
    if (!(moff && moff[0] == '0') && a->dt_out == a->dt_in && fdbm_conv_mid_ok(ps)) { g_last_kind = 7; return fdbm_launch_conv_mid(ps, a->dt_in, a->dt_out, st); }
  }
  if (kind == 2) {
    p.w = a->w_frag;
    p.ksplit = 1;
    p.partial = reinterpret_cast<float*>(a->workspace);        // (ksplit 1: diagnostic stamps only)
    // split the channel chunks over workgroups when the tiles alone leave most of the chip idle
    int nchunks = 0;
    for (int s = 0; s < a->nseg; ++s) nchunks += (a->seg[s].cin + kc - 1) / kc;
    const int tw = th, tr = tw == 16 ? 4 : tw == 8 ? 8 : 4;
    const int64_t blocks = (int64_t)a->B * (a->H / tr) * (a->W / tw) * ((a->Cout + bn - 1) / bn);
    static const char* smin = getenv("FDBM_TAP_SPLIT_MIN");     // experiments
    if (a->acc_ws && blocks <= 128 && nchunks >= (smin ? atoi(smin) : 8)) {
      int ks = (int)(256 / blocks);
      if (ks > nchunks / 2) ks = nchunks / 2;
      if (ks > 8) ks = 8;
      if (blocks * 4 > 65536) ks = 1;
      while (ks > 1 && (int64_t)ks * M * a->Cout * 4 + 65536 > a->acc_ws_bytes) --ks;   // 64 KiB of counters, then one fp32 slab per slice
      if (ks > 1) { p.ksplit = ks; p.partial = reinterpret_cast<float*>(a->acc_ws); }
    }
    return fdbm_launch_conv_tap(p, a->dt_in, a->dt_out, th, bn / 16, st);
  }
  const int kg = plan_kgroups(((M + bm - 1) / bm) * ((a->Cout + bn - 1) / bn), bm, bn, nk);
  if (a->dt_in == FDBM_BF16 && a->dt_out == FDBM_BF16) return launch_conv_tile<bf16_t, bf16_t>(p, bm, bn, kg, st);
  if (a->dt_in == FDBM_BF16 && a->dt_out == FDBM_F32) return launch_conv_tile<bf16_t, float>(p, bm, bn, kg, st);
  if (a->dt_in == FDBM_F16 && a->dt_out == FDBM_F16) return launch_conv_tile<f16_t, f16_t>(p, bm, bn, kg, st);
  if (a->dt_in == FDBM_F16 && a->dt_out == FDBM_F32) return launch_conv_tile<f16_t, float>(p, bm, bn, kg, st);
  return launch_conv_tile<float, float>(p, bm, bn, kg, st);
}
