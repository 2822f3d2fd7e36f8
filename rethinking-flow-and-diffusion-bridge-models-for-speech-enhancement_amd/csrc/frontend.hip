// frontend.hip - STFT / iSTFT with the spectral compression and the time padding fused in.
//
// Once per utterance, ~3 MB of traffic and < 0.3 GFLOP against ~16 TFLOP of network work, so
// this favours exactness and generality over speed: a direct DFT per frame with fp64
// accumulation and an exact-argument fp64 twiddle table in LDS (works for any even n_fft,
// e.g. the reference's 510 as well as 512), one workgroup per frame.
//   stft : frame f of sample b: reflect-padded (centre=True) samples * window -> one-sided DFT
//          -> |X|^e e^{j arg X} * factor (or log / none) -> column f of spec[b][bins][Tpad];
//          columns >= frames are the time padding of pad_spec (zero or reflection).
//   istft: inverse compression -> one-sided inverse DFT * window -> frames_ws; then
//          overlap-add, divide by the window-square envelope, trim n_fft/2, cut to L.
#include "common.h"

#define FE_MAX_NFFT 1024

__device__ __forceinline__ void fill_twiddles(double* tw, int n_fft) {
  for (int i = threadIdx.x; i < n_fft; i += blockDim.x) {
    double s, c;
    sincospi(2.0 * (double)i / (double)n_fft, &s, &c);
    tw[2 * i] = c;
    tw[2 * i + 1] = s;
  }
}

__global__ void __launch_bounds__(256) stft_kernel(f32x2* __restrict__ spec,
                                                   const float* __restrict__ wave,
                                                   const float* __restrict__ window, int L, int n_fft,
                                                   int hop, int frames, int Tpad, int pad_mode,
                                                   int transform, float factor, float exponent) {
  __shared__ double tw[2 * FE_MAX_NFFT];
  __shared__ float xs[FE_MAX_NFFT];
  const int col = blockIdx.x, b = blockIdx.y;
  const int bins = n_fft / 2 + 1;
  int f = col;
  bool zero = false;
  if (col >= frames) {
    if (pad_mode == 1) f = frames - 2 - (col - frames);
    else zero = true;
  }
  f32x2* dst = spec + (int64_t)b * bins * Tpad + col;
  if (zero) {
    for (int k = threadIdx.x; k < bins; k += blockDim.x) dst[(int64_t)k * Tpad] = f32x2{0.f, 0.f};
    return;
  }
  fill_twiddles(tw, n_fft);
  const float* w = wave + (int64_t)b * L;
  for (int n = threadIdx.x; n < n_fft; n += blockDim.x) {
    int idx = f * hop + n - n_fft / 2;
    if (idx < 0) idx = -idx;
    if (idx >= L) idx = 2 * (L - 1) - idx;
    xs[n] = w[idx] * window[n];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < bins; k += blockDim.x) {
    double re = 0.0, im = 0.0;
    int ti = 0;
    for (int n = 0; n < n_fft; ++n) {
      const double x = (double)xs[n];
      re += x * tw[2 * ti];
      im -= x * tw[2 * ti + 1];
      ti += k;
      if (ti >= n_fft) ti -= n_fft;
    }
    float fr = (float)re, fi = (float)im;
    if (transform != 0) {
      const float mag = sqrtf(fr * fr + fi * fi);
      float g = 0.f;
      if (mag > 0.f) {
        float m2;
        if (transform == 1) m2 = (exponent == 0.5f) ? sqrtf(mag) : (exponent == 1.0f ? mag : powf(mag, exponent));
        else m2 = log1pf(mag);
        g = m2 * factor / mag;
      }
      fr *= g;
      fi *= g;
    }
    dst[(int64_t)k * Tpad] = f32x2{fr, fi};
  }
}

extern "C" int fdbm_stft(void* spec, const float* wave, const float* window, int B, int L, int n_fft,
                         int hop, int frames, int Tpad, int pad_mode, int transform, float factor,
                         float exponent, void* stream) {
  FDBM_CHECK(spec && wave && window, "fdbm_stft: null pointer");
  FDBM_CHECK(n_fft >= 2 && n_fft % 2 == 0 && n_fft <= FE_MAX_NFFT, "fdbm_stft: n_fft=%d must be even and <= %d", n_fft, FE_MAX_NFFT);
  FDBM_CHECK(L > n_fft / 2, "fdbm_stft: signal of %d samples is too short for reflect padding of %d", L, n_fft / 2);
  FDBM_CHECK(frames == 1 + L / hop, "fdbm_stft: frames=%d but 1 + L/hop = %d", frames, 1 + L / hop);
  FDBM_CHECK(Tpad >= frames && (pad_mode == 0 || Tpad - frames <= frames - 1), "fdbm_stft: bad Tpad=%d for %d frames", Tpad, frames);
  FDBM_CHECK(transform >= 0 && transform <= 2, "fdbm_stft: bad transform %d", transform);
  stft_kernel<<<dim3(Tpad, B), 256, 0, (hipStream_t)stream>>>((f32x2*)spec, wave, window, L, n_fft, hop,
                                                            frames, Tpad, pad_mode, transform, factor, exponent);
  FDBM_LAUNCH_CHECK("fdbm_stft");
  return 0;
}

__global__ void __launch_bounds__(256) istft_frames_kernel(float* __restrict__ frames_ws,
                                                           const f32x2* __restrict__ spec,
                                                           const float* __restrict__ window, int n_fft,
                                                           int frames, int Tpad, int transform,
                                                           float factor, float exponent) {
  __shared__ double tw[2 * FE_MAX_NFFT];
  __shared__ float sre[FE_MAX_NFFT / 2 + 1], sim[FE_MAX_NFFT / 2 + 1];
  const int f = blockIdx.x, b = blockIdx.y;
  const int bins = n_fft / 2 + 1;
  fill_twiddles(tw, n_fft);
  const f32x2* src = spec + (int64_t)b * bins * Tpad + f;
  for (int k = threadIdx.x; k < bins; k += blockDim.x) {
    f32x2 v = src[(int64_t)k * Tpad];
    float fr = v[0], fi = v[1];
    if (transform != 0) {
      fr /= factor;
      fi /= factor;
      const float mag = sqrtf(fr * fr + fi * fi);
      float g = 0.f;
      if (mag > 0.f) {
        float m2;
        if (transform == 1) m2 = (exponent == 0.5f) ? mag * mag : (exponent == 1.0f ? mag : powf(mag, 1.0f / exponent));
        else m2 = expm1f(mag);
        g = m2 / mag;
      }
      fr *= g;
      fi *= g;
    }
    sre[k] = fr;
    sim[k] = fi;
  }
  __syncthreads();
  const int half = n_fft / 2;
  for (int n = threadIdx.x; n < n_fft; n += blockDim.x) {
    // x[n] = (X0 + (-1)^n X_half + 2 sum_{k=1}^{half-1} (Re X_k cos - Im X_k sin)) / n_fft
    double acc = (double)sre[0] + ((n & 1) ? -(double)sre[half] : (double)sre[half]);
    int ti = n % n_fft;   // k = 1
    for (int k = 1; k < half; ++k) {
      acc += 2.0 * ((double)sre[k] * tw[2 * ti] - (double)sim[k] * tw[2 * ti + 1]);
      ti += n;
      if (ti >= n_fft) ti -= n_fft;
    }
    frames_ws[((int64_t)b * frames + f) * n_fft + n] = (float)(acc / (double)n_fft) * window[n];
  }
}

__global__ void __launch_bounds__(256) istft_ola_kernel(float* __restrict__ wave,
                                                        const float* __restrict__ frames_ws,
                                                        const float* __restrict__ window, int L,
                                                        int n_fft, int hop, int frames) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  const int pos = i + n_fft / 2;                 // position in the un-trimmed signal
  int k_hi = pos / hop;
  if (k_hi > frames - 1) k_hi = frames - 1;
  int k_lo = 0;                                  // smallest k with k*hop + n_fft > pos
  if (pos - n_fft + 1 > 0) k_lo = (pos - n_fft + 1 + hop - 1) / hop;
  float acc = 0.f, env = 0.f;
  for (int k = k_lo; k <= k_hi; ++k) {
    const int n = pos - k * hop;
    acc += frames_ws[((int64_t)b * frames + k) * n_fft + n];
    const float w = window[n];
    env += w * w;
  }
  wave[(int64_t)b * L + i] = acc / env;
}

extern "C" int fdbm_istft(float* wave, const void* spec, const float* window, float* frames_ws, int B,
                          int L, int n_fft, int hop, int frames, int Tpad, int transform, float factor,
                          float exponent, void* stream) {
  FDBM_CHECK(wave && spec && window && frames_ws, "fdbm_istft: null pointer");
  FDBM_CHECK(n_fft >= 2 && n_fft % 2 == 0 && n_fft <= FE_MAX_NFFT, "fdbm_istft: n_fft=%d must be even and <= %d", n_fft, FE_MAX_NFFT);
  FDBM_CHECK(frames >= 1 && frames <= Tpad, "fdbm_istft: frames=%d must be in [1, Tpad=%d]", frames, Tpad);
  FDBM_CHECK(L > 0 && L + n_fft / 2 <= n_fft + hop * (frames - 1), "fdbm_istft: length %d exceeds what %d frames cover", L, frames);
  FDBM_CHECK(transform >= 0 && transform <= 2, "fdbm_istft: bad transform %d", transform);
  hipStream_t st = (hipStream_t)stream;
  istft_frames_kernel<<<dim3(frames, B), 256, 0, st>>>(frames_ws, (const f32x2*)spec, window, n_fft, frames,
                                                      Tpad, transform, factor, exponent);
  FDBM_LAUNCH_CHECK("fdbm_istft/frames");
  istft_ola_kernel<<<dim3(cdiv(L, 256), B), 256, 0, st>>>(wave, frames_ws, window, L, n_fft, hop, frames);
  FDBM_LAUNCH_CHECK("fdbm_istft/ola");
  return 0;
}

// ---------------------------------------------------------------------------------
// stand-alone spec_fwd / spec_back (data_module.py:173-199) and pad_spec (other.py:76-90)
// for callers that hold a spectrogram already; the drivers use the fused forms above.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) spec_transform_kernel(f32x2* __restrict__ out,
                                                             const f32x2* __restrict__ in, int64_t n,
                                                             int transform, float factor,
                                                             float exponent, int inverse) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    f32x2 v = in[i];
    float fr = v[0], fi = v[1];
    if (transform != 0) {
      if (inverse) { fr /= factor; fi /= factor; }
      const float mag = sqrtf(fr * fr + fi * fi);
      float g = 0.f;
      if (mag > 0.f) {
        float m2;
        if (transform == 1) {
          const float e = inverse ? 1.0f / exponent : exponent;
          m2 = (e == 0.5f) ? sqrtf(mag) : (e == 2.0f ? mag * mag : (e == 1.0f ? mag : powf(mag, e)));
        } else {
          m2 = inverse ? expm1f(mag) : log1pf(mag);
        }
        g = m2 / mag;
      }
      if (!inverse) g *= factor;
      fr *= g;
      fi *= g;
    }
    out[i] = f32x2{fr, fi};
  }
}

extern "C" int fdbm_spec_transform(void* out, const void* in, int64_t n_complex, int transform,
                                   float factor, float exponent, int inverse, void* stream) {
  FDBM_CHECK(out && in, "fdbm_spec_transform: null pointer");
  FDBM_CHECK(transform >= 0 && transform <= 2, "fdbm_spec_transform: bad transform %d", transform);
  int g = (int)((n_complex + 255) / 256);
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  spec_transform_kernel<<<g, 256, 0, (hipStream_t)stream>>>((f32x2*)out, (const f32x2*)in, n_complex,
                                                           transform, factor, exponent, inverse);
  FDBM_LAUNCH_CHECK("fdbm_spec_transform");
  return 0;
}

__global__ void __launch_bounds__(256) pad_spec_kernel(f32x2* __restrict__ out,
                                                       const f32x2* __restrict__ in, int64_t rows,
                                                       int T, int Tpad, int mode) {
  const int64_t total = rows * Tpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % Tpad);
    const int64_t r = i / Tpad;
    f32x2 v = {0.f, 0.f};
    if (t < T) v = in[r * T + t];
    else if (mode == 1) v = in[r * T + (T - 2 - (t - T))];
    else if (mode == 2) v = in[r * T + (T - 1)];
    out[i] = v;
  }
}

extern "C" int fdbm_pad_spec(void* out, const void* in, int64_t rows, int T, int Tpad, int mode,
                             void* stream) {
  FDBM_CHECK(out && in, "fdbm_pad_spec: null pointer");
  FDBM_CHECK(Tpad >= T && mode >= 0 && mode <= 2 && (mode != 1 || Tpad - T <= T - 1), "fdbm_pad_spec: bad arguments");
  int g = (int)((rows * Tpad + 255) / 256);
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  pad_spec_kernel<<<g, 256, 0, (hipStream_t)stream>>>((f32x2*)out, (const f32x2*)in, rows, T, Tpad, mode);
  FDBM_LAUNCH_CHECK("fdbm_pad_spec");
  return 0;
}

// ---------------------------------------------------------------------------------
// Waveform normalisation either side of the path (infer_folder.py:102-107,118-121; infer_single.py:79-84,97-99;
// model.py:391-397,403): nf = max |y| ('noisy') or std(y) ('std'); y / nf goes into the STFT; after the iSTFT
// x_hat * nf and, if max |x_hat| > 1, x_hat / max |x_hat| * clip (0.95 folder driver, 0.5 single-file driver).
// Fused into the front-end kernels: the division happens where the frame is windowed, the multiplication and the
// peak search where the overlap-add writes its sample; one tiny kernel per side does the reduction / the rescale.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) wave_norm_kernel(float* __restrict__ nf, const float* __restrict__ wave, int L, int mode) {
  __shared__ double red[3][256];
  const int b = blockIdx.x;
  const float* w = wave + (int64_t)b * L;
  double mx = 0.0, s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < L; i += 256) {
    const double v = (double)w[i];
    mx = fmax(mx, fabs(v));
    s1 += v;
    s2 += v * v;
  }
  red[0][threadIdx.x] = mx; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      red[0][threadIdx.x] = fmax(red[0][threadIdx.x], red[0][threadIdx.x + st]);
      red[1][threadIdx.x] += red[1][threadIdx.x + st];
      red[2][threadIdx.x] += red[2][threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (mode == 0) {
      nf[b] = (float)red[0][0];
    } else {                              // torch.std: unbiased
      const double mean = red[1][0] / (double)L;
      double var = (red[2][0] - (double)L * mean * mean) / (double)(L > 1 ? L - 1 : 1);
      nf[b] = (float)sqrt(var > 0.0 ? var : 0.0);
    }
  }
}

extern "C" int fdbm_wave_norm_factor(float* nf, const float* wave, int B, int L, int mode, void* stream) {
  FDBM_CHECK(nf && wave && B > 0 && L > 0 && (mode == 0 || mode == 1), "fdbm_wave_norm_factor: bad arguments");
  wave_norm_kernel<<<B, 256, 0, (hipStream_t)stream>>>(nf, wave, L, mode);
  FDBM_LAUNCH_CHECK("fdbm_wave_norm_factor");
  return 0;
}

// stft_kernel with the sample divided by its clip's normalisation factor first (the same two roundings as
// `y = y / nf` followed by torch.stft's window product)
__global__ void __launch_bounds__(256) stft_norm_kernel(f32x2* __restrict__ spec, const float* __restrict__ wave,
                                                        const float* __restrict__ window, const float* __restrict__ norm,
                                                        int L, int n_fft, int hop, int frames, int Tpad, int pad_mode,
                                                        int transform, float factor, float exponent) {
  __shared__ double tw[2 * FE_MAX_NFFT];
  __shared__ float xs[FE_MAX_NFFT];
  const int col = blockIdx.x, b = blockIdx.y;
  const int bins = n_fft / 2 + 1;
  int f = col;
  bool zero = false;
  if (col >= frames) {
    if (pad_mode == 1) f = frames - 2 - (col - frames);
    else zero = true;
  }
  f32x2* dst = spec + (int64_t)b * bins * Tpad + col;
  if (zero) {
    for (int k = threadIdx.x; k < bins; k += blockDim.x) dst[(int64_t)k * Tpad] = f32x2{0.f, 0.f};
    return;
  }
  fill_twiddles(tw, n_fft);
  const float* w = wave + (int64_t)b * L;
  const float nf = norm[b];
  for (int n = threadIdx.x; n < n_fft; n += blockDim.x) {
    int idx = f * hop + n - n_fft / 2;
    if (idx < 0) idx = -idx;
    if (idx >= L) idx = 2 * (L - 1) - idx;
    xs[n] = (w[idx] / nf) * window[n];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < bins; k += blockDim.x) {
    double re = 0.0, im = 0.0;
    int ti = 0;
    for (int n = 0; n < n_fft; ++n) {
      const double x = (double)xs[n];
      re += x * tw[2 * ti];
      im -= x * tw[2 * ti + 1];
      ti += k;
      if (ti >= n_fft) ti -= n_fft;
    }
    float fr = (float)re, fi = (float)im;
    if (transform != 0) {
      const float mag = sqrtf(fr * fr + fi * fi);
      float g = 0.f;
      if (mag > 0.f) {
        float m2;
        if (transform == 1) m2 = (exponent == 0.5f) ? sqrtf(mag) : (exponent == 1.0f ? mag : powf(mag, exponent));
        else m2 = log1pf(mag);
        g = m2 * factor / mag;
      }
      fr *= g;
      fi *= g;
    }
    dst[(int64_t)k * Tpad] = f32x2{fr, fi};
  }
}

extern "C" int fdbm_stft_norm(void* spec, const float* wave, const float* window, const float* norm, int B, int L,
                              int n_fft, int hop, int frames, int Tpad, int pad_mode, int transform, float factor,
                              float exponent, void* stream) {
  FDBM_CHECK(spec && wave && window && norm, "fdbm_stft_norm: null pointer");
  FDBM_CHECK(n_fft >= 2 && n_fft % 2 == 0 && n_fft <= FE_MAX_NFFT, "fdbm_stft_norm: n_fft=%d must be even and <= %d", n_fft, FE_MAX_NFFT);
  FDBM_CHECK(L > n_fft / 2, "fdbm_stft_norm: signal of %d samples is too short for reflect padding of %d", L, n_fft / 2);
  FDBM_CHECK(frames == 1 + L / hop, "fdbm_stft_norm: frames=%d but 1 + L/hop = %d", frames, 1 + L / hop);
  FDBM_CHECK(Tpad >= frames && (pad_mode == 0 || Tpad - frames <= frames - 1), "fdbm_stft_norm: bad Tpad=%d for %d frames", Tpad, frames);
  FDBM_CHECK(transform >= 0 && transform <= 2, "fdbm_stft_norm: bad transform %d", transform);
  stft_norm_kernel<<<dim3(Tpad, B), 256, 0, (hipStream_t)stream>>>((f32x2*)spec, wave, window, norm, L, n_fft, hop,
                                                                 frames, Tpad, pad_mode, transform, factor, exponent);
  FDBM_LAUNCH_CHECK("fdbm_stft_norm");
  return 0;
}

// overlap-add with the renormalisation: writes x_hat * nf and leaves max |x_hat * nf| per clip (as the bit pattern of
// a non-negative float, which orders like the float) in peak[b]
__global__ void __launch_bounds__(256) istft_ola_renorm_kernel(float* __restrict__ wave, const float* __restrict__ frames_ws,
                                                               const float* __restrict__ window, const float* __restrict__ norm,
                                                               unsigned* __restrict__ peak, int L, int n_fft, int hop, int frames) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = 0.f;
  if (i < L) {
    const int pos = i + n_fft / 2;
    int k_hi = pos / hop;
    if (k_hi > frames - 1) k_hi = frames - 1;
    int k_lo = 0;
    if (pos - n_fft + 1 > 0) k_lo = (pos - n_fft + 1 + hop - 1) / hop;
    float acc = 0.f, env = 0.f;
    for (int k = k_lo; k <= k_hi; ++k) {
      const int n = pos - k * hop;
      acc += frames_ws[((int64_t)b * frames + k) * n_fft + n];
      const float w = window[n];
      env += w * w;
    }
    v = (acc / env) * norm[b];
    wave[(int64_t)b * L + i] = v;
  }
  // wave maximum by DPP-free shuffles, one atomic per wave
  float m = fabsf(v);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(peak + b, __float_as_uint(m));
}

__global__ void __launch_bounds__(256) wave_clip_kernel(float* __restrict__ wave, const unsigned* __restrict__ peak, int L, float clip) {
  const int b = blockIdx.y;
  const float pk = __uint_as_float(peak[b]);
  if (!(pk > 1.0f)) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < L) {
    float* p = wave + (int64_t)b * L + i;
    *p = *p / pk * clip;
  }
}

__global__ void zero_u32_kernel(unsigned* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

extern "C" int fdbm_istft_renorm(float* wave, const void* spec, const float* window, float* frames_ws, const float* norm,
                                 void* peak_ws, float clip, int B, int L, int n_fft, int hop, int frames, int Tpad,
                                 int transform, float factor, float exponent, void* stream) {
  FDBM_CHECK(wave && spec && window && frames_ws && norm && peak_ws, "fdbm_istft_renorm: null pointer");
  FDBM_CHECK(n_fft >= 2 && n_fft % 2 == 0 && n_fft <= FE_MAX_NFFT, "fdbm_istft_renorm: n_fft=%d must be even and <= %d", n_fft, FE_MAX_NFFT);
  FDBM_CHECK(frames >= 1 && frames <= Tpad, "fdbm_istft_renorm: frames=%d must be in [1, Tpad=%d]", frames, Tpad);
  FDBM_CHECK(L > 0 && L + n_fft / 2 <= n_fft + hop * (frames - 1), "fdbm_istft_renorm: length %d exceeds what %d frames cover", L, frames);
  FDBM_CHECK(transform >= 0 && transform <= 2, "fdbm_istft_renorm: bad transform %d", transform);
  FDBM_CHECK(clip >= 0.f, "fdbm_istft_renorm: clip must be >= 0 (0: no clip rule)");
  hipStream_t st = (hipStream_t)stream;
  zero_u32_kernel<<<cdiv(B, 256), 256, 0, st>>>((unsigned*)peak_ws, B);
  FDBM_LAUNCH_CHECK("fdbm_istft_renorm/zero");
  istft_frames_kernel<<<dim3(frames, B), 256, 0, st>>>(frames_ws, (const f32x2*)spec, window, n_fft, frames,
                                                      Tpad, transform, factor, exponent);
  FDBM_LAUNCH_CHECK("fdbm_istft_renorm/frames");
  istft_ola_renorm_kernel<<<dim3(cdiv(L, 256), B), 256, 0, st>>>(wave, frames_ws, window, norm, (unsigned*)peak_ws, L, n_fft, hop, frames);
  FDBM_LAUNCH_CHECK("fdbm_istft_renorm/ola");
  if (clip > 0.f) {
    wave_clip_kernel<<<dim3(cdiv(L, 256), B), 256, 0, st>>>(wave, (const unsigned*)peak_ws, L, clip);
    FDBM_LAUNCH_CHECK("fdbm_istft_renorm/clip");
  }
  return 0;
}
