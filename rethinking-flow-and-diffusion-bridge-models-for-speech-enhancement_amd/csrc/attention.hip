// attention.hip - single-head global attention of AttnBlockpp:
//   out[b][i][:] = sum_j softmax_j(q_i . k_j * C^-0.5) v_j      over all N = H*W tokens.
//
// 0.04 % of the network's flops and latency-bound (N = T tokens at the 16-row level, 4*T/64
// at the bottleneck).  The kernels:
//  * attention_kernel: fp32 flash-style on the vector ALU (any N, C; the parity mode's kernel):
//    16 queries per workgroup, keys/values streamed through LDS in tiles of 32, online softmax
//    with 16-lane shuffle reductions, fp32 accumulation whatever the storage dtype.
//  * attention_mfma_kernel: bf16 storage, N % 64 == 0, C % 64 == 0: the same 16 queries per
//    workgroup, but Q.K^T and P.V on the matrix cores (4 waves: 16 keys each of a 64-key tile for
//    the scores, C/4 output channels each for P.V), scores of all N keys kept in LDS (two-pass
//    softmax, no running rescale), V staged TRANSPOSED in LDS so that its MFMA operand is
//    key-contiguous.  ~6x shorter than the VALU kernel at N = C = 256, batch 1.
#include "conv_common.h"

#define ATT_QB 16
#define ATT_KB 32

template <typename T>
__global__ void __launch_bounds__(256) attention_kernel(T* __restrict__ out, const T* __restrict__ qkv,
                                                        int N, int C, float scale) {
  constexpr int VW = DT<T>::vecw;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int CP = C + 4;                       // padded row (16-byte shift per row: conflict-free b128)
  float* q_s = sm;                            // [QB][CP]
  float* k_s = q_s + ATT_QB * CP;             // [KB][CP]
  float* v_s = k_s + ATT_KB * CP;             // [KB][CP]
  float* p_s = v_s + ATT_KB * CP;             // [QB][KB]
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * ATT_QB;
  const int tid = threadIdx.x;
  const int tq = tid >> 4;                    // query within the block
  const int tj = tid & 15;
  const T* base = qkv + (int64_t)b * N * 3 * C;
  const int nvec = C / VW;

  for (int i = tid; i < ATT_QB * nvec; i += 256) {
    const int r = i / nvec, v = i % nvec;
    float x[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) x[k] = 0.f;
    if (q0 + r < N) Vec16<T>::load(base + (int64_t)(q0 + r) * 3 * C + v * VW, x);
#pragma unroll
    for (int k = 0; k < VW; ++k) q_s[r * CP + v * VW + k] = x[k];
  }

  float m_run = -INFINITY, l_run = 0.f;
  float o[16];                                // channels tj*4.. within each 64-channel slab
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  const int nslab = (C + 63) / 64;            // each thread: 4 channels per 64-channel slab

  for (int k0 = 0; k0 < N; k0 += ATT_KB) {
    __syncthreads();
    for (int i = tid; i < ATT_KB * nvec; i += 256) {
      const int r = i / nvec, v = i % nvec;
      float kx[VW], vx[VW];
#pragma unroll
      for (int k = 0; k < VW; ++k) kx[k] = vx[k] = 0.f;
      if (k0 + r < N) {
        const T* row = base + (int64_t)(k0 + r) * 3 * C + v * VW;
        Vec16<T>::load(row + C, kx);
        Vec16<T>::load(row + 2 * C, vx);
      }
#pragma unroll
      for (int k = 0; k < VW; ++k) {
        k_s[r * CP + v * VW + k] = kx[k];
        v_s[r * CP + v * VW + k] = vx[k];
      }
    }
    __syncthreads();
    // scores for keys tj and tj+16
    float s0 = 0.f, s1 = 0.f;
    const float* qr = q_s + tq * CP;
    const float* kr0 = k_s + tj * CP;
    const float* kr1 = k_s + (tj + 16) * CP;
    for (int c = 0; c < C; c += 4) {
      const f32x4 qv = *reinterpret_cast<const f32x4*>(qr + c);
      const f32x4 a = *reinterpret_cast<const f32x4*>(kr0 + c);
      const f32x4 bq = *reinterpret_cast<const f32x4*>(kr1 + c);
      s0 += qv[0] * a[0] + qv[1] * a[1] + qv[2] * a[2] + qv[3] * a[3];
      s1 += qv[0] * bq[0] + qv[1] * bq[1] + qv[2] * bq[2] + qv[3] * bq[3];
    }
    s0 = (k0 + tj < N) ? s0 * scale : -INFINITY;
    s1 = (k0 + tj + 16 < N) ? s1 * scale : -INFINITY;
    float tmax = fmaxf(s0, s1);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, off));
    const float m_new = fmaxf(m_run, tmax);
    const float p0 = __expf(s0 - m_new), p1 = __expf(s1 - m_new);
    float psum = p0 + p1;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) psum += __shfl_xor(psum, off);
    const float alpha = __expf(m_run - m_new);     // exp(-inf) = 0 on the first tile
    l_run = l_run * alpha + psum;
    m_run = m_new;
    p_s[tq * ATT_KB + tj] = p0;
    p_s[tq * ATT_KB + tj + 16] = p1;
    __syncthreads();
    // O[tq][channels of this thread] = O*alpha + sum_key p * v
    const float* pr = p_s + tq * ATT_KB;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      if (sl < nslab && sl * 64 + tj * 4 < C) {
        f32x4 accv = {o[sl * 4] * alpha, o[sl * 4 + 1] * alpha, o[sl * 4 + 2] * alpha, o[sl * 4 + 3] * alpha};
        const float* vp = v_s + sl * 64 + tj * 4;
        for (int key = 0; key < ATT_KB; ++key) {
          const float pw = pr[key];
          const f32x4 vv = *reinterpret_cast<const f32x4*>(vp + key * CP);
          accv[0] += pw * vv[0]; accv[1] += pw * vv[1]; accv[2] += pw * vv[2]; accv[3] += pw * vv[3];
        }
        o[sl * 4] = accv[0]; o[sl * 4 + 1] = accv[1]; o[sl * 4 + 2] = accv[2]; o[sl * 4 + 3] = accv[3];
      }
    }
  }
  if (q0 + tq < N) {
    const float inv = 1.0f / l_run;
    T* dst = out + ((int64_t)b * N + q0 + tq) * C;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      if (sl < nslab && sl * 64 + tj * 4 < C) {
        const float r4[4] = {o[sl * 4] * inv, o[sl * 4 + 1] * inv, o[sl * 4 + 2] * inv, o[sl * 4 + 3] * inv};
        if constexpr (sizeof(T) == 2) {
          typename V16<T>::x4 t = {(T)r4[0], (T)r4[1], (T)r4[2], (T)r4[3]};
          *reinterpret_cast<typename V16<T>::x4*>(dst + sl * 64 + tj * 4) = t;
        } else {
          *reinterpret_cast<f32x4*>(dst + sl * 64 + tj * 4) = f32x4{r4[0], r4[1], r4[2], r4[3]};
        }
      }
    }
  }
}

// LDS rows: K tile [64 keys][C bf16 + 16 B pad]; V^T tile [C channels][64 keys bf16 + 16 B pad], its
// 16-byte slots (8 keys) XOR-swizzled with (channel >> 3) & 7; scores [16][N] f32; P [16][N bf16 + 16 B].
template <typename H>
__global__ void __launch_bounds__(256) attention_mfma_kernel(H* __restrict__ out, const H* __restrict__ qkv,
                                                             int N, int C, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_a[];
  const int KRS = C * 2 + 16;                 // K-tile row stride (bytes)
  constexpr int VRS = 64 * 2 + 16;            // V^T-tile row stride
  const int PRS = N * 2 + 16;                 // P row stride
  const int kv_bytes = max(64 * KRS, C * VRS);
  unsigned char* s_kv = smem_a;
  float* s_S = reinterpret_cast<float*>(smem_a + kv_bytes);                 // [16][N]
  unsigned char* s_P = smem_a + kv_bytes + 16 * N * 4;                      // [16][PRS]
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fk = lane >> 4;
  const H* base = qkv + (int64_t)b * N * 3 * C;
  const int nch = C / 8;                      // 16-byte chunks per row
  const int nks = C / 32;                     // MFMA k-steps over the channels (<= 8)

  // ---- Q tile -> LDS -> operand registers (B operand: column = query) ------------------------------
  for (int i = tid; i < 16 * nch; i += 256) {
    const int r = i / nch, ch = i - r * nch;
    *reinterpret_cast<uint4*>(s_kv + r * KRS + ch * 16) =
        *reinterpret_cast<const uint4*>(base + (int64_t)(q0 + r) * 3 * C + ch * 8);
  }
  __syncthreads();
  uint4 qf[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    qf[ks] = ks < nks ? *reinterpret_cast<const uint4*>(s_kv + frow * KRS + (ks * 4 + fk) * 16) : uint4{0u, 0u, 0u, 0u};
  __syncthreads();

  // ---- scores: S[query][key] = scale * q . k, 64 keys per tile, 16 per wave ---------------------------
  for (int k0 = 0; k0 < N; k0 += 64) {
    for (int i = tid; i < 64 * nch; i += 256) {
      const int r = i / nch, ch = i - r * nch;
      *reinterpret_cast<uint4*>(s_kv + r * KRS + ch * 16) =
          *reinterpret_cast<const uint4*>(base + (int64_t)(k0 + r) * 3 * C + C + ch * 8);
    }
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      if (ks < nks) {
        const uint4 kf = *reinterpret_cast<const uint4*>(s_kv + (wave * 16 + frow) * KRS + (ks * 4 + fk) * 16);
        Mfma<H>::run(kf, qf[ks], acc);      // rows = keys, columns = queries
      }
    // lane (query frow, fk) holds keys k0 + 16 wave + 4 fk + 0..3
    *reinterpret_cast<f32x4*>(s_S + frow * N + k0 + wave * 16 + fk * 4) = acc * scale;
    __syncthreads();
  }

  // ---- softmax over the N keys of each query: 16 threads per query ------------------------------------
  {
    const int q = tid >> 4, part = tid & 15;
    const int per = N / 16;
    const float* row = s_S + q * N + part * per;
    float m = -INFINITY;
    for (int i = 0; i < per; ++i) m = fmaxf(m, row[i]);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    float l = 0.f;
    for (int i = 0; i < per; ++i) l += __expf(row[i] - m);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) l += __shfl_xor(l, off);
    const float inv = 1.0f / l;
    H* prow = reinterpret_cast<H*>(s_P + q * PRS) + part * per;
    for (int i = 0; i < per; ++i) prow[i] = (H)(__expf(row[i] - m) * inv);
  }
  __syncthreads();

  // ---- O = P . V: wave w owns channels [w C/4, (w+1) C/4), 64 keys per tile -------------------------
  const int ntw = C / 64;                     // 16-channel n-tiles per wave (<= 4)
  f32x4 oacc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) oacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < N; k0 += 64) {
    // V tile transposed into LDS: a thread takes 8 channels of TWO consecutive keys and writes 8 dwords
    // {v[2kp][c], v[2kp+1][c]}; within a wave 8 chunks x 8 key pairs -> 64 distinct banks
    for (int i = tid; i < 32 * nch; i += 256) {
      const int ch = (i & 7) | ((i >> 6) % (nch / 8)) << 3;
      const int kp = ((i >> 3) & 7) | ((i >> 6) / (nch / 8)) << 3;
      const H* src = base + (int64_t)(k0 + 2 * kp) * 3 * C + 2 * C + ch * 8;
      const uint4 lo = *reinterpret_cast<const uint4*>(src);
      const uint4 hi = *reinterpret_cast<const uint4*>(src + 3 * C);
      const unsigned short* l16 = reinterpret_cast<const unsigned short*>(&lo);
      const unsigned short* h16 = reinterpret_cast<const unsigned short*>(&hi);
      const int slot = (kp >> 2) ^ (ch & 7);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        *reinterpret_cast<unsigned*>(s_kv + (ch * 8 + e) * VRS + slot * 16 + (kp & 3) * 4) =
            (unsigned)l16[e] | ((unsigned)h16[e] << 16);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 pf = *reinterpret_cast<const uint4*>(s_P + frow * PRS + (k0 + ks * 32 + fk * 8) * 2);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < ntw) {
          const int c = (wave * ntw + j) * 16 + frow;
          const uint4 vf = *reinterpret_cast<const uint4*>(s_kv + c * VRS + (((ks * 4 + fk) ^ ((c >> 3) & 7)) << 4));
          Mfma<H>::run(vf, pf, oacc[j]);    // rows = channels, columns = queries
        }
    }
    __syncthreads();
  }
  H* dst = out + ((int64_t)b * N + q0 + frow) * C;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < ntw) {
      const int c = (wave * ntw + j) * 16 + fk * 4;
      const typename V16<H>::x4 t = {(H)oacc[j][0], (H)oacc[j][1], (H)oacc[j][2], (H)oacc[j][3]};
      *reinterpret_cast<typename V16<H>::x4*>(dst + c) = t;
    }
}

// The benchmark shape (N = 256 tokens, C = 256 channels: the 16 x 16 level of a 4 s clip) as a fully unrolled
// variant: the kernel above pays one exposed memory round trip per 64-key tile (9 of them); here every load of
// Q, K and V is requested up front (33 x 16 B per thread, 512 threads), so the 16 workgroups pay ONE round trip
// and then only LDS phases.  8 waves: 16 keys each of a 128-key tile for the scores, 32 channels each for P.V.
template <typename H>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) attention_mfma256_kernel(H* __restrict__ out, const H* __restrict__ qkv,
                                                                float scale) {
  constexpr int N = 256, C = 256;
  constexpr int KRS = C * 2 + 16;             // K-tile / Q-tile row stride (bytes)
  constexpr int VRS = 64 * 2 + 16;            // V^T-tile row stride
  constexpr int PRS = N * 2 + 16;             // P row stride
  constexpr int KV_BYTES = 128 * KRS;         // 67,584 (>= C * VRS = 36,864)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  unsigned char* s_kv = smem_b;
  float* s_S = reinterpret_cast<float*>(smem_b + KV_BYTES);                 // [16][N]
  unsigned char* s_P = smem_b + KV_BYTES + 16 * N * 4;                      // [16][PRS]
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fk = lane >> 4;
  const H* base = qkv + (int64_t)b * N * 3 * C;
  const int r32 = tid >> 5, ch = tid & 31;    // this thread's row-within-16 and 16-byte chunk for K / V items

  // ---- every global load of the kernel --------------------------------------------------------------------
  const uint4 qreg = *reinterpret_cast<const uint4*>(base + (int64_t)(q0 + r32) * 3 * C + ch * 8);       // 16 queries x 32 chunks
  // (named scalars, not arrays: a 64-dword array is left in scratch memory by this compiler even with constant
  // indices; key r32 + 16 j for kreg_j, key pair r32 + 16 j - keys 2 kp, 2 kp + 1 - for vlo_j / vhi_j)
#define KLOAD(j) const uint4 kreg_##j = *reinterpret_cast<const uint4*>(base + (int64_t)(r32 + 16 * (j)) * 3 * C + C + ch * 8);
  KLOAD(0) KLOAD(1) KLOAD(2) KLOAD(3) KLOAD(4) KLOAD(5) KLOAD(6) KLOAD(7)
  KLOAD(8) KLOAD(9) KLOAD(10) KLOAD(11) KLOAD(12) KLOAD(13) KLOAD(14) KLOAD(15)
#undef KLOAD
#define VLOAD(j)                                                                                                  \
  const uint4 vlo_##j = *reinterpret_cast<const uint4*>(base + (int64_t)(2 * (r32 + 16 * (j))) * 3 * C + 2 * C + ch * 8);      \
  const uint4 vhi_##j = *reinterpret_cast<const uint4*>(base + (int64_t)(2 * (r32 + 16 * (j)) + 1) * 3 * C + 2 * C + ch * 8);
  VLOAD(0) VLOAD(1) VLOAD(2) VLOAD(3) VLOAD(4) VLOAD(5) VLOAD(6) VLOAD(7)
#undef VLOAD

  // ---- Q tile -> LDS -> operand registers ------------------------------------------------------------------
  *reinterpret_cast<uint4*>(s_kv + r32 * KRS + ch * 16) = qreg;
  __syncthreads();
  uint4 qf[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(s_kv + frow * KRS + (ks * 4 + fk) * 16);
  __syncthreads();

  // ---- scores, two tiles of 128 keys -------------------------------------------------------------------------
  // (tile bodies as macros with literal tile indices: with a run-time index, or captured by a lambda, the
  // register arrays end up in scratch memory)
#define KSTORE(jj, reg) *reinterpret_cast<uint4*>(s_kv + (r32 + 16 * (jj)) * KRS + ch * 16) = reg;
#define SCORE_TILE(kt, K0, K1, K2, K3, K4, K5, K6, K7)                                                                                             \
  {                                                                                                                \
    KSTORE(0, K0) KSTORE(1, K1) KSTORE(2, K2) KSTORE(3, K3) KSTORE(4, K4) KSTORE(5, K5) KSTORE(6, K6) KSTORE(7, K7)     \
    __syncthreads();                                                                                               \
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};                                                                              \
    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) {                                                             \
      const uint4 kf = *reinterpret_cast<const uint4*>(s_kv + (wave * 16 + frow) * KRS + (ks * 4 + fk) * 16);      \
      Mfma<H>::run(kf, qf[ks], acc);                                                                          \
    }                                                                                                              \
    *reinterpret_cast<f32x4*>(s_S + frow * N + (kt) * 128 + wave * 16 + fk * 4) = acc * scale;                     \
    __syncthreads();                                                                                               \
  }
  SCORE_TILE(0, kreg_0, kreg_1, kreg_2, kreg_3, kreg_4, kreg_5, kreg_6, kreg_7)
  SCORE_TILE(1, kreg_8, kreg_9, kreg_10, kreg_11, kreg_12, kreg_13, kreg_14, kreg_15)
#undef SCORE_TILE
#undef KSTORE

  // ---- softmax: 32 threads per query, 8 keys each -------------------------------------------------------------
  {
    const int q = tid >> 5, part = tid & 31;
    const float* row = s_S + q * N + part * 8;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = row[i];
    float m = v[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) m = fmaxf(m, v[i]);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = __expf(v[i] - m); l += v[i]; }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) l += __shfl_xor(l, off);
    const float inv = 1.0f / l;
    typename V16<H>::x8 pv;
#pragma unroll
    for (int i = 0; i < 8; ++i) pv[i] = (H)(v[i] * inv);
    *reinterpret_cast<typename V16<H>::x8*>(s_P + q * PRS + part * 16) = pv;
  }
  __syncthreads();

  // ---- O = P . V: wave w owns channels [32 w, 32 w + 32), four tiles of 64 keys -------------------------------
  f32x4 oacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#define VT_ITEM(jj, LO, HI)                                                                                        \
  {                                                                                                                \
    const int kp = r32 + 16 * (jj);                                                                                \
    const unsigned lw[4] = {LO.x, LO.y, LO.z, LO.w}, hw[4] = {HI.x, HI.y, HI.z, HI.w};                             \
    const int slot = (kp >> 2) ^ (ch & 7);                                                                         \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                \
      const unsigned le = (e & 1) ? (lw[e >> 1] >> 16) : (lw[e >> 1] & 0xffffu);                                   \
      const unsigned he = (e & 1) ? (hw[e >> 1] & 0xffff0000u) : (hw[e >> 1] << 16);                               \
      *reinterpret_cast<unsigned*>(s_kv + (ch * 8 + e) * VRS + slot * 16 + (kp & 3) * 4) = le | he;                \
    }                                                                                                              \
  }
#define PV_TILE(kt, L0, H0, L1, H1)                                                                                \
  {                                                                                                                \
    VT_ITEM(0, L0, H0)                                                                                             \
    VT_ITEM(1, L1, H1)                                                                                             \
    __syncthreads();                                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                             \
      const uint4 pf = *reinterpret_cast<const uint4*>(s_P + frow * PRS + ((kt) * 64 + ks * 32 + fk * 8) * 2);     \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                              \
        const int c = (wave * 2 + j) * 16 + frow;                                                                  \
        const uint4 vf = *reinterpret_cast<const uint4*>(s_kv + c * VRS + (((ks * 4 + fk) ^ ((c >> 3) & 7)) << 4)); \
        Mfma<H>::run(vf, pf, oacc[j]);                                                                        \
      }                                                                                                            \
    }                                                                                                              \
    __syncthreads();                                                                                               \
  }
  PV_TILE(0, vlo_0, vhi_0, vlo_1, vhi_1) PV_TILE(1, vlo_2, vhi_2, vlo_3, vhi_3)
  PV_TILE(2, vlo_4, vhi_4, vlo_5, vhi_5) PV_TILE(3, vlo_6, vhi_6, vlo_7, vhi_7)
#undef PV_TILE
#undef VT_ITEM
  H* dst = out + ((int64_t)b * N + q0 + frow) * C;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = (wave * 2 + j) * 16 + fk * 4;
    const typename V16<H>::x4 t = {(H)oacc[j][0], (H)oacc[j][1], (H)oacc[j][2], (H)oacc[j][3]};
    *reinterpret_cast<typename V16<H>::x4*>(dst + c) = t;
  }
}

// The parity modes' kernel where the shape allows (f32 storage, N % 64 == 0, N <= 512, C % 64 == 0): the structure of
// attention_mfma_kernel on the f32 matrix instruction (v_mfma_f32_16x16x4_f32: exact f32 products and sums).  A b128
// fragment is consumed as four k-groups - element q of lane (frow, fk) is channel / key 16 ks + 4 fk + q for BOTH
// operands, so the same permutation of the reduction index on either side.  Scores, P and the V tile stay f32.
// LDS rows: K tile [64 keys][C f32 + 16 B]; V^T tile [C channels][64 keys f32 + 16 B]; S [16][N] f32; P [16][N f32 + 16 B].
__global__ void __launch_bounds__(256) attention_mfma_f32_kernel(float* __restrict__ out, const float* __restrict__ qkv,
                                                                 int N, int C, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
  const int KRS = C * 4 + 16;                 // K-tile row stride (bytes)
  constexpr int VRS = 64 * 4 + 16;            // V^T-tile row stride
  const int PRS = N * 4 + 16;                 // P row stride
  const int kv_bytes = max(64 * KRS, C * VRS);
  unsigned char* s_kv = smem_f;
  float* s_S = reinterpret_cast<float*>(smem_f + kv_bytes);                 // [16][N]
  unsigned char* s_P = smem_f + kv_bytes + 16 * N * 4;                      // [16][PRS]
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fk = lane >> 4;
  const float* base = qkv + (int64_t)b * N * 3 * C;
  const int nch = C / 4;                      // 16-byte chunks per row
  const int nks = C / 16;                     // groups of four MFMA k-steps over the channels (<= 16)

  // ---- Q tile -> LDS -> operand registers (B operand: column = query) ------------------------------
  for (int i = tid; i < 16 * nch; i += 256) {
    const int r = i / nch, ch = i - r * nch;
    *reinterpret_cast<uint4*>(s_kv + r * KRS + ch * 16) =
        *reinterpret_cast<const uint4*>(base + (int64_t)(q0 + r) * 3 * C + ch * 4);
  }
  __syncthreads();
  f32x4 qf[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks)
    qf[ks] = ks < nks ? *reinterpret_cast<const f32x4*>(s_kv + frow * KRS + (ks * 4 + fk) * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  // ---- scores: S[query][key] = scale * q . k, 64 keys per tile, 16 per wave ---------------------------
  for (int k0 = 0; k0 < N; k0 += 64) {
    for (int i = tid; i < 64 * nch; i += 256) {
      const int r = i / nch, ch = i - r * nch;
      *reinterpret_cast<uint4*>(s_kv + r * KRS + ch * 16) =
          *reinterpret_cast<const uint4*>(base + (int64_t)(k0 + r) * 3 * C + C + ch * 4);
    }
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      if (ks < nks) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(s_kv + (wave * 16 + frow) * KRS + (ks * 4 + fk) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[q], qf[ks][q], acc, 0, 0, 0);   // rows = keys, columns = queries
      }
    *reinterpret_cast<f32x4*>(s_S + frow * N + k0 + wave * 16 + fk * 4) = acc * scale;
    __syncthreads();
  }

  // ---- softmax over the N keys of each query: 16 threads per query ------------------------------------
  {
    const int q = tid >> 4, part = tid & 15;
    const int per = N / 16;
    const float* row = s_S + q * N + part * per;
    float m = -INFINITY;
    for (int i = 0; i < per; ++i) m = fmaxf(m, row[i]);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    float l = 0.f;
    for (int i = 0; i < per; ++i) l += __expf(row[i] - m);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) l += __shfl_xor(l, off);
    const float inv = 1.0f / l;
    float* prow = reinterpret_cast<float*>(s_P + q * PRS) + part * per;
    for (int i = 0; i < per; ++i) prow[i] = __expf(row[i] - m) * inv;
  }
  __syncthreads();

  // ---- O = P . V: wave w owns channels [w C/4, (w+1) C/4), 64 keys per tile -------------------------
  const int ntw = C / 64;                     // 16-channel n-tiles per wave (<= 4)
  f32x4 oacc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) oacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < N; k0 += 64) {
    // V tile transposed into LDS: a thread takes 4 channels of one key and writes them to 4 channel rows
    for (int i = tid; i < 64 * nch; i += 256) {
      const int ch = i % nch, key = i / nch;
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + (int64_t)(k0 + key) * 3 * C + 2 * C + ch * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) *reinterpret_cast<float*>(s_kv + (ch * 4 + e) * VRS + key * 4) = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 pf = *reinterpret_cast<const f32x4*>(s_P + frow * PRS + (k0 + ks * 16 + fk * 4) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < ntw) {
          const int c = (wave * ntw + j) * 16 + frow;
          const f32x4 vf = *reinterpret_cast<const f32x4*>(s_kv + c * VRS + (ks * 4 + fk) * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) oacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[q], pf[q], oacc[j], 0, 0, 0);   // rows = channels, columns = queries
        }
    }
    __syncthreads();
  }
  float* dst = out + ((int64_t)b * N + q0 + frow) * C;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < ntw) *reinterpret_cast<f32x4*>(dst + (wave * ntw + j) * 16 + fk * 4) = oacc[j];
}

extern "C" int fdbm_attention(void* out, const void* qkv, int B, int N, int C, int dtype, void* stream) {
  FDBM_CHECK(out && qkv, "fdbm_attention: null pointer");
  FDBM_CHECK(C % 8 == 0 && C <= 256, "fdbm_attention: C=%d must be a multiple of 8, <= 256", C);
  FDBM_CHECK(B > 0 && N > 0, "fdbm_attention: bad shape");
  const size_t smem = ((size_t)(ATT_QB + 2 * ATT_KB) * (C + 4) + ATT_QB * ATT_KB) * sizeof(float);
  dim3 grid(cdiv(N, ATT_QB), B);
  const float scale = 1.0f / sqrtf((float)C);
  hipStream_t st = (hipStream_t)stream;
#define ATT_16(HT)                                                                                                       \
  do {                                                                                                                   \
    if (N == 256 && C == 256) {                                                                                          \
      const size_t sm = (size_t)128 * (256 * 2 + 16) + (size_t)16 * 256 * 4 + (size_t)16 * (256 * 2 + 16);              \
      static bool set256 = false;                                                                                        \
      if (!set256) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_mfma256_kernel<HT>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); set256 = true; } \
      attention_mfma256_kernel<HT><<<dim3(N / 16, B), 512, sm, st>>>((HT*)out, (const HT*)qkv, scale);                   \
    } else if (N % 64 == 0 && N <= 1024 && C % 64 == 0) {                                                                \
      const int krs = C * 2 + 16;                                                                                        \
      const size_t kv = (size_t)(64 * krs > C * 144 ? 64 * krs : C * 144);                                               \
      const size_t sm = kv + (size_t)16 * N * 4 + (size_t)16 * (N * 2 + 16);                                             \
      static bool setm = false;                                                                                          \
      if (!setm) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_mfma_kernel<HT>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024); setm = true; } \
      attention_mfma_kernel<HT><<<dim3(N / 16, B), 256, sm, st>>>((HT*)out, (const HT*)qkv, N, C, scale);                \
    } else {                                                                                                             \
      static bool set = false;                                                                                           \
      if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<HT>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); set = true; } \
      attention_kernel<HT><<<grid, 256, smem, st>>>((HT*)out, (const HT*)qkv, N, C, scale);                              \
    }                                                                                                                    \
  } while (0)
  if (dtype == FDBM_BF16) {
    ATT_16(bf16_t);
  } else if (dtype == FDBM_F16) {
    ATT_16(f16_t);
  } else if (dtype == FDBM_F32 && N % 64 == 0 && N <= 512 && C % 64 == 0) {
    const int krs = C * 4 + 16;
    const size_t kv = (size_t)(64 * krs > C * 272 ? 64 * krs : C * 272);
    const size_t sm = kv + (size_t)16 * N * 4 + (size_t)16 * (N * 4 + 16);
    static bool setf = false;
    if (!setf) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_mfma_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024); setf = true; }
    attention_mfma_f32_kernel<<<dim3(N / 16, B), 256, sm, st>>>((float*)out, (const float*)qkv, N, C, scale);
  } else if (dtype == FDBM_F32) {
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); set = true; }
    attention_kernel<float><<<grid, 256, smem, st>>>((float*)out, (const float*)qkv, N, C, scale);
  } else {
    FDBM_CHECK(false, "fdbm_attention: bad dtype %d", dtype);
  }
#undef ATT_16
  FDBM_LAUNCH_CHECK("fdbm_attention");
  return 0;
}
