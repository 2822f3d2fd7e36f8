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
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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

__device__ __forceinline__ float silu_f(float x) { return x * __frcp_rn(1.0f + __expf(-x)); }

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
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void load(const bf16_t* p, float* v) {
    bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float* v) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = t;
  }
};
