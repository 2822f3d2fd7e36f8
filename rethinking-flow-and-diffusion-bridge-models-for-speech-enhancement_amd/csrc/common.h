// common.h - shared helpers for the gfx950 kernels (device + host glue).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/fdbm_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16_t;                      // FDBM_F16 storage mode (BASELINE configs[4]); same kernels, T = f16_t
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// vector types of a 16-bit storage type
template <typename T> struct V16;
template <> struct V16<bf16_t> { typedef bf16x8 x8; typedef bf16x4 x4; };
template <> struct V16<f16_t> { typedef f16x8 x8; typedef f16x4 x4; };
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void fdbm_set_error(const char* fmt, ...);

#define FDBM_CHECK(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      fdbm_set_error(__VA_ARGS__);       \
      return 1;                          \
    }                                    \
  } while (0)

#define FDBM_LAUNCH_CHECK(name)                                            \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      fdbm_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return 2;                                                            \
    }                                                                      \
  } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- device helpers ---------------------------------------------------------
template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int code = FDBM_F32;
  static constexpr int vecw = 4;   // elements per 16 bytes
};
template <> struct DT<bf16_t> {
  static constexpr int code = FDBM_BF16;
  static constexpr int vecw = 8;
};
template <> struct DT<f16_t> {
  static constexpr int code = FDBM_F16;
  static constexpr int vecw = 8;
};

// throughput mode (bf16 storage): one v_exp_f32 and one v_rcp_f32, no denormal/IEEE fix-up
// sequences (__expf / __frcp_rn expand to ~15 instructions on this target)
__device__ __forceinline__ float silu_f(float x) {
  const float t = __builtin_amdgcn_exp2f(-1.4426950408889634f * x);
  return x * __builtin_amdgcn_rcpf(1.0f + t);
}
// parity mode (f32 storage): correctly rounded-ish exp and a true division, like the host libm
__device__ __forceinline__ float silu_precise(float x) { return x / (1.0f + expf(-x)); }
template <typename T> __device__ __forceinline__ float silu_t(float x) {
  if constexpr (sizeof(T) == 4) return silu_precise(x); else return silu_f(x);
}

// GroupNorm statistics -> per-channel scale/shift in LDS (s_ss[0..C) scale, s_ss[C..2C) shift).
// stats: nsplit == 0: final [B][G][2] (mean, rstd);  nsplit > 0: float partial (sum, sumsq)
// [B][nsplit][G][2];  nsplit < 0: the same with |nsplit| splits stored as doubles.  Reduced
// here in fp64 in a fixed order.  G <= 32, blockDim.x == 256.  upg >= 1 (fp64 partials): the rows
// hold G * upg entries, upg consecutive ones per group (unit statistics, see fdbm_conv_args).
__device__ __forceinline__ void gn_scale_shift(float* s_ss, double* s_red /*[8][32][2]*/,
                                               const float* __restrict__ stats, int nsplit,
                                               double inv_count, float eps, int b, int C, int G,
                                               const float* __restrict__ gamma,
                                               const float* __restrict__ beta, int upg = 1) {
  __shared__ float s_mr[64];
  const int tid = threadIdx.x;
  if (nsplit == 0) {
    if (tid < 2 * G) s_mr[tid] = stats[(int64_t)b * G * 2 + tid];
  } else {
    const int g = tid & 31, part = tid >> 5;
    double a0 = 0.0, a1 = 0.0;
    if (g < G) {
      if (nsplit < 0) {          // fp64 partials (f32 parity mode)
        const double* sd = reinterpret_cast<const double*>(stats);
        for (int sp = part; sp < -nsplit; sp += 8) {
          const double* q = sd + (((int64_t)b * (-nsplit) + sp) * G + g) * upg * 2;
          for (int k = 0; k < upg; ++k) {
            a0 += q[2 * k];
            a1 += q[2 * k + 1];
          }
        }
      } else {
        for (int sp = part; sp < nsplit; sp += 8) {
          const float* q = stats + (((int64_t)b * nsplit + sp) * G + g) * 2;
          a0 += (double)q[0];
          a1 += (double)q[1];
        }
      }
    }
    s_red[(part * 32 + g) * 2] = a0;
    s_red[(part * 32 + g) * 2 + 1] = a1;
    __syncthreads();
    if (tid < G) {
      double t0 = 0.0, t1 = 0.0;
      for (int q = 0; q < 8; ++q) { t0 += s_red[(q * 32 + tid) * 2]; t1 += s_red[(q * 32 + tid) * 2 + 1]; }
      const double mean = t0 * inv_count;
      double var = t1 * inv_count - mean * mean;
      if (var < 0.0) var = 0.0;
      s_mr[2 * tid] = (float)mean;
      s_mr[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  const int cpg = C / G;
  for (int c = tid; c < C; c += blockDim.x) {
    const int g = c / cpg;
    const float sc = s_mr[2 * g + 1] * gamma[c];
    s_ss[c] = sc;
    s_ss[C + c] = beta[c] - s_mr[2 * g] * sc;
  }
  __syncthreads();
}

// 16-byte vector <-> float[vecw]
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  __device__ static __forceinline__ void load(const float* p, float* v) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  __device__ static __forceinline__ void store(float* p, const float* v) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
};
template <typename H> struct Vec16_16 {
  static constexpr int N = 8;
  __device__ static __forceinline__ void load(const H* p, float* v) {
    typename V16<H>::x8 t = *reinterpret_cast<const typename V16<H>::x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  }
  __device__ static __forceinline__ void store(H* p, const float* v) {
    typename V16<H>::x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (H)v[i];
    *reinterpret_cast<typename V16<H>::x8*>(p) = t;
  }
};
template <> struct Vec16<bf16_t> : Vec16_16<bf16_t> {};
template <> struct Vec16<f16_t> : Vec16_16<f16_t> {};
