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
#include <type_traits>

#include "common.h"

struct ConvParams {
  fdbm_conv_seg seg[FDBM_MAX_SEG];
  int nseg;
  const void* w;
  const float* bias;
  const float* tbias;
  int tbias_stride;
  const void* res;
  float scale;
  void* out;
  int B, H, W, Cout, CoutPad;
  int nk;
  int ksplit;       // grid.z: k-steps are split over this many workgroups
  float* partial;   // fp32 slabs [ksplit][M][Cout] when ksplit > 1
};

template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
  __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                  *reinterpret_cast<const bf16x8*>(&b), acc, 0, 0, 0);
  }
};

template <typename TO> struct OutVec;
template <> struct OutVec<float> {
  __device__ static __forceinline__ void load(const float* p, float* v) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  __device__ static __forceinline__ void store(float* p, const float* v) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
};
template <> struct OutVec<bf16_t> {
  __device__ static __forceinline__ void load(const bf16_t* p, float* v) {
    bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float* v) {
    bf16x4 t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *reinterpret_cast<bf16x4*>(p) = t;
  }
};

// p.seg[ks] with a run-time ks makes the compiler copy the whole kernel argument to scratch
// and index it there; chains of wave-uniform selects on constant indices stay in SGPRs.
#define SEG_FIELD(p, ks, f) ((ks) == 0 ? (p).seg[0].f : (ks) == 1 ? (p).seg[1].f : (ks) == 2 ? (p).seg[2].f : (p).seg[3].f)

template <typename T, typename TO, int BM, int BN>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const ConvParams p) {
  constexpr int KC = 128 / (int)sizeof(T);   // elements per k-step row
  constexpr int VW = 16 / (int)sizeof(T);    // elements per 16-byte chunk
  constexpr int WTM = BM / 2, WTN = BN / 2;  // wave tile (4 waves as 2 x 2)
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int AROWS = BM / 32, WROWS = BN / 32;   // 16-byte loads per thread per k-step
  constexpr int BUF = (BM + BN) * 128;
  constexpr bool F32 = sizeof(T) == 4;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int H = p.H, W = p.W;
  const int HW = H * W;
  const int64_t M = (int64_t)p.B * HW;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // ---- split-K: this block owns k-steps [kbeg, kend) ------------------------------------
  const int kper = (p.nk + p.ksplit - 1) / p.ksplit;
  const int kbeg = blockIdx.z * kper;
  const int kend = min(p.nk, kbeg + kper);
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

  uint4 areg0[AROWS], wreg0[WROWS], areg1[AROWS], wreg1[WROWS];   // two named register sets
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
  auto load_regs = [&](auto SET, int kidx) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    const void* sg_src = SEG_FIELD(p, ks, src);
    const int sg_C = SEG_FIELD(p, ks, C), sg_coff = SEG_FIELD(p, ks, coff);
    const int sg_cin = SEG_FIELD(p, ks, cin), sg_taps = SEG_FIELD(p, ks, taps);
    const int dy = sg_taps == 9 ? ktap / 3 - 1 : 0;
    const int dx = sg_taps == 9 ? ktap % 3 - 1 : 0;
    const int cvalid = min(KC, sg_cin - kc * KC);
    const bool cok = lchunk * VW < cvalid;
    const T* src = reinterpret_cast<const T*>(sg_src) + sg_coff + (cok ? kc * KC + lchunk * VW : 0);
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int iy = py[i] + dy, ix = px[i] + dx;
      const bool ok = cok && pval[i] && iy >= 0 && iy < H && ix >= 0 && ix < W;
      const int iyc = min(max(iy, 0), H - 1), ixc = min(max(ix, 0), W - 1);
      uint4 v = *reinterpret_cast<const uint4*>(src + (pbase[i] + (int64_t)iyc * W + ixc) * sg_C);
      if (!ok) v = uint4{0u, 0u, 0u, 0u};
      if constexpr (S == 0) areg0[i] = v; else areg1[i] = v;
    }
    const T* wp = reinterpret_cast<const T*>(p.w) + ((int64_t)kidx * p.CoutPad + n0 + lrow) * KC + lchunk * VW;
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
      const uint4 v = *reinterpret_cast<const uint4*>(wp + (int64_t)32 * i * KC);
      if constexpr (S == 0) wreg0[i] = v; else wreg1[i] = v;
    }
    const int nchunks = (sg_cin + KC - 1) / KC;
    if (++kc == nchunks) {
      kc = 0;
      if (++ktap == sg_taps) { ktap = 0; ++ks; }
    }
  };

  auto write_lds = [&](auto SET, int buf) __attribute__((always_inline)) {
    constexpr int S = decltype(SET)::value;
    unsigned char* A = smem + buf * BUF;
    unsigned char* Wt = A + BM * 128;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int row = lrow + 32 * i;
      *reinterpret_cast<uint4*>(A + row * 128 + ((lchunk ^ ((row >> 1) & 7)) << 4)) = (S == 0 ? areg0[i] : areg1[i]);
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
    const unsigned char* A = smem + buf * BUF;
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
          for (int i = 0; i < MT; ++i) Mfma<bf16_t>::run(wf[j], af[i], acc[j][i]);
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

  // ---- main loop: loads run two k-steps ahead of the MFMAs ------------------------------
  //   iteration t:  issue loads of step t+2 -> regs[t&1] ; MFMAs on LDS buf[t&1] ;
  //                 regs[(t+1)&1] (issued one iteration ago) -> LDS buf[(t+1)&1] ; barrier
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  if (nloc > 0) {
    load_regs(S0{}, kbeg);
    if (nloc > 1) load_regs(S1{}, kbeg + 1);
    write_lds(S0{}, 0);
    __syncthreads();
    int t = 0;
    for (; t + 1 < nloc; t += 2) {
      if (t + 2 < nloc) load_regs(S0{}, kbeg + t + 2);
      compute(0);
      write_lds(S1{}, 1);
      __syncthreads();
      if (t + 3 < nloc) load_regs(S1{}, kbeg + t + 3);
      compute(1);
      if (t + 2 < nloc) write_lds(S0{}, 0);
      __syncthreads();
    }
    if (t < nloc) compute(0);
  }

  // ---- epilogue --------------------------------------------------------------------------
  const int Cout = p.Cout;
  if (p.ksplit > 1) {
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
  TO* out = reinterpret_cast<TO*>(p.out);
  const TO* res = reinterpret_cast<const TO*>(p.res);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int64_t m = m0 + wm * WTM + i * 16 + frow;
    if (m >= M) continue;
    const int64_t b = m / HW;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * WTN + j * 16 + fk * 4;
      if (n >= Cout) continue;
      float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
      if (p.bias) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
        v[0] += bv[0]; v[1] += bv[1]; v[2] += bv[2]; v[3] += bv[3];
      }
      if (p.tbias) {
        const f32x4 tv = *reinterpret_cast<const f32x4*>(p.tbias + b * p.tbias_stride + n);
        v[0] += tv[0]; v[1] += tv[1]; v[2] += tv[2]; v[3] += tv[3];
      }
      if (res) {
        float r[4];
        OutVec<TO>::load(res + m * Cout + n, r);
        v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
      }
      v[0] *= p.scale; v[1] *= p.scale; v[2] *= p.scale; v[3] *= p.scale;
      OutVec<TO>::store(out + m * Cout + n, v);
    }
  }
}

// split-K reduction + epilogue: out = (sum_z slab[z] + bias + tbias + res) * scale
template <typename TO>
__global__ void __launch_bounds__(256) conv_splitk_reduce_kernel(const ConvParams p) {
  const int Cout = p.Cout;
  const int HW = p.H * p.W;
  const int64_t M = (int64_t)p.B * HW;
  const int nv = Cout / 4;
  const int64_t total = M * nv;
  TO* out = reinterpret_cast<TO*>(p.out);
  const TO* res = reinterpret_cast<const TO*>(p.res);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % nv) * 4;
    const int64_t m = i / nv;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < p.ksplit; ++z) a += *reinterpret_cast<const f32x4*>(p.partial + ((int64_t)z * M + m) * Cout + n);
    float v[4] = {a[0], a[1], a[2], a[3]};
    if (p.bias) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
      v[0] += bv[0]; v[1] += bv[1]; v[2] += bv[2]; v[3] += bv[3];
    }
    if (p.tbias) {
      const f32x4 tv = *reinterpret_cast<const f32x4*>(p.tbias + (m / HW) * p.tbias_stride + n);
      v[0] += tv[0]; v[1] += tv[1]; v[2] += tv[2]; v[3] += tv[3];
    }
    if (res) {
      float r[4];
      OutVec<TO>::load(res + m * Cout + n, r);
      v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
    }
    v[0] *= p.scale; v[1] *= p.scale; v[2] *= p.scale; v[3] *= p.scale;
    OutVec<TO>::store(out + m * Cout + n, v);
  }
}

template <typename T, typename TO, int BM, int BN>
static int launch_conv(const ConvParams& p, hipStream_t st) {
  constexpr int SMEM = 2 * (BM + BN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, TO, BM, BN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) {
      fdbm_set_error("fdbm_conv_igemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = true;
  }
  const int64_t M = (int64_t)p.B * p.H * p.W;
  dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((p.Cout + BN - 1) / BN), (unsigned)p.ksplit);
  conv_igemm_kernel<T, TO, BM, BN><<<grid, 256, SMEM, st>>>(p);
  FDBM_LAUNCH_CHECK("fdbm_conv_igemm");
  if (p.ksplit > 1) {
    const int64_t total = M * (p.Cout / 4);
    int g = (int)((total + 255) / 256);
    if (g > 2048) g = 2048;
    conv_splitk_reduce_kernel<TO><<<g, 256, 0, st>>>(p);
    FDBM_LAUNCH_CHECK("fdbm_conv_igemm/reduce");
  }
  return 0;
}

template <typename T, typename TO>
static int launch_conv_tile(const ConvParams& p, int bm, int bn, hipStream_t st) {
  if (bm == 128 && bn == 128) return launch_conv<T, TO, 128, 128>(p, st);
  if (bm == 128 && bn == 64) return launch_conv<T, TO, 128, 64>(p, st);
  if (bm == 64 && bn == 128) return launch_conv<T, TO, 64, 128>(p, st);
  return launch_conv<T, TO, 64, 64>(p, st);
}

extern "C" int fdbm_conv_kc(int dtype) { return dtype == FDBM_BF16 ? 64 : 32; }

// Tile / split-K plan for a conv of M pixels, Cout channels, nk k-steps (host side, also used
// by the caller to size the split-K workspace): fills bm, bn, ksplit.
extern "C" int fdbm_conv_plan(int64_t M, int Cout, int nk, int* bm, int* bn, int* ksplit) {
  const int target = 512;                         // blocks wanted in flight (2 per CU)
  int BMs = 128, BNs = Cout <= 64 ? 64 : 128;
  auto blocks = [&](int a, int b) { return ((M + a - 1) / a) * ((Cout + b - 1) / b); };
  if (blocks(BMs, BNs) < target && BNs == 128 && Cout > 64) BNs = 64;
  if (blocks(BMs, BNs) < target) BMs = 64;
  int ks = 1;
  const int64_t nb = blocks(BMs, BNs);
  if (nb < 256 && nk >= 4) {
    ks = (int)(target / nb);
    if (ks > nk / 2) ks = nk / 2;
    // keep the fp32 slabs small: ks * M * Cout * 4 bytes <= 8 MiB
    const int64_t per = M * Cout * 4;
    while (ks > 1 && ks * per > (8 << 20)) --ks;
    if (ks < 1) ks = 1;
  }
  *bm = BMs; *bn = BNs; *ksplit = ks;
  return 0;
}

extern "C" int fdbm_conv_igemm(const fdbm_conv_args* a, void* stream) {
  FDBM_CHECK(a, "fdbm_conv_igemm: null args");
  FDBM_CHECK(a->nseg >= 1 && a->nseg <= FDBM_MAX_SEG, "fdbm_conv_igemm: nseg=%d out of range", a->nseg);
  FDBM_CHECK(a->w && a->out, "fdbm_conv_igemm: null weight/output pointer");
  FDBM_CHECK(a->dt_in == FDBM_F32 || a->dt_in == FDBM_BF16, "fdbm_conv_igemm: bad dt_in %d", a->dt_in);
  FDBM_CHECK(a->dt_out == FDBM_F32 || a->dt_out == FDBM_BF16, "fdbm_conv_igemm: bad dt_out %d", a->dt_out);
  FDBM_CHECK(!(a->dt_in == FDBM_F32 && a->dt_out == FDBM_BF16), "fdbm_conv_igemm: f32 in / bf16 out is not built");
  FDBM_CHECK(a->Cout > 0 && a->Cout % 4 == 0, "fdbm_conv_igemm: Cout=%d must be a positive multiple of 4", a->Cout);
  FDBM_CHECK(a->CoutPad >= a->Cout && a->CoutPad % 128 == 0, "fdbm_conv_igemm: CoutPad=%d must be a multiple of 128 >= Cout", a->CoutPad);
  FDBM_CHECK(a->B > 0 && a->H > 0 && a->W > 0, "fdbm_conv_igemm: bad shape B=%d H=%d W=%d", a->B, a->H, a->W);
  const int kc = fdbm_conv_kc(a->dt_in);
  const int vw = a->dt_in == FDBM_BF16 ? 8 : 4;
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
  p.res = a->res; p.scale = a->scale; p.out = a->out;
  p.B = a->B; p.H = a->H; p.W = a->W; p.Cout = a->Cout; p.CoutPad = a->CoutPad;
  p.nk = nk;
  int bm, bn, ks;
  const int64_t M = (int64_t)a->B * a->H * a->W;
  fdbm_conv_plan(M, a->Cout, nk, &bm, &bn, &ks);
  if (!a->workspace || a->workspace_bytes <= 0) ks = 1;
  while (ks > 1 && (int64_t)ks * M * a->Cout * 4 > a->workspace_bytes) --ks;
  p.ksplit = ks;
  p.partial = reinterpret_cast<float*>(a->workspace);
  hipStream_t st = (hipStream_t)stream;
  if (a->dt_in == FDBM_BF16 && a->dt_out == FDBM_BF16) return launch_conv_tile<bf16_t, bf16_t>(p, bm, bn, st);
  if (a->dt_in == FDBM_BF16 && a->dt_out == FDBM_F32) return launch_conv_tile<bf16_t, float>(p, bm, bn, st);
  return launch_conv_tile<float, float>(p, bm, bn, st);
}
