/*
 * fdbm_hip.h - C ABI of libfdbm_hip.so, the MI355X (gfx950) kernels behind the
 * reverse-sampling path of flow / diffusion-bridge speech enhancement.
 *
 * Conventions:
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - the caller owns every buffer (inputs, outputs, workspaces); the library never
 *     allocates, never frees, never synchronises; every call only enqueues kernels on
 *     `stream` (a hipStream_t passed as void*), so calls are HIP-graph capturable.
 *   - return value: 0 = ok, non-zero = error; fdbm_last_error() gives the message of the
 *     last failure on the calling thread.  No C++ exception crosses the boundary.
 *   - activations are NHWC ("[B][H=freq][W=time][C]", C contiguous); dtype codes below.
 *
 * Each entry point names the reference interface it stands in for (file:line under
 * the reference checkout).
 */
#ifndef FDBM_HIP_H
#define FDBM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDBM_F32 0
#define FDBM_BF16 1
#define FDBM_F16 2      /* IEEE half storage, fp32 accumulation (BASELINE configs[4]); every entry that takes FDBM_BF16 takes it */

#define FDBM_MAX_SEG 4

const char* fdbm_last_error(void);
int fdbm_version(void);

/* ------------------------------------------------------------------ sampler state update
 * out = (wa[b]*a + wb[b]*b) + wc[b]*c over complex64 [B][n_complex] viewed as floats; each
 * product and sum rounded separately (no FMA contraction).  c / wc may be NULL (2 terms).
 * Replaces the 5 elementwise kernels of `xt = w_xt*xt + w_s*s + w_y*y` (fdbm/bridge.py:83),
 * `... + w_z*z` (:109) and prior_sampling's `y*b + z*sigma` (:48).  out may alias a. */
int fdbm_bridge_update(void* out, const void* a, const void* b, const void* c,
                       const float* wa, const float* wb, const float* wc,
                       int B, int64_t n_complex, void* stream);

/* Step boundary of the ei samplers inside a replayed graph (fdbm/bridge.py:73-85, :96-111): in ONE launch, elementwise
 * over [B][F][T]:  s = out-conv(pyramid) (ncsnpp_v2.py:391-399; rows f >= Fn are 0);  x = (wa*x + wb*s) + wc*third
 * (fdbm_bridge_update's arithmetic);  packed[b][f < Fn][t] = (x.re, x.im, y.re, y.im) = the next evaluation's network
 * input (ncsnpp_v2.py:246-251);  zero_bytes at zero_ptr set to 0 (the statistics arena);  dense_n floats copied (the next
 * evaluation's time-embedding rows).  pyramid NULL: no update (the boundary in front of the first evaluation);
 * packed NULL: no packing (behind the last one).  zero_ptr / dense_* may be NULL / 0. */
int fdbm_step_boundary(void* x, const void* y, const void* third, const float* pyramid, const float* out_w,
                       const float* out_b, const float* wa, const float* wb, const float* wc, float* packed,
                       void* zero_ptr, int64_t zero_bytes, float* dense_dst, const float* dense_src,
                       int64_t dense_n, int B, int F, int Fn, int T, void* stream);

/* Euler-Maruyama predictor move (fdbm/util/predictors.py:44-51 + ProbabilityPathSB.sde,
 * fdbm/bridge.py:294-306): drift = (wx*x + ws*s) + wy*y; x_mean = x + drift*dt;
 * x_new = x_mean + (gd*sqrt(-dt))*z.  Weights are [B] device floats; dt a host scalar. */
int fdbm_pc_predictor(void* x_new, void* x_mean, const void* x, const void* s, const void* y,
                      const void* z, const float* wx, const float* ws, const float* wy,
                      const float* gd, float dt, int B, int64_t n_complex, void* stream);

/* Langevin corrector move (fdbm/util/correctors.py:44-53,72-79 + Bridge.score_fn,
 * fdbm/bridge.py:51-54): score = -(x - (a*s + b*y))/den; x_mean = x + step*score;
 * x_new = x_mean + noise*noise_scale.  a, b, den, step, noise_scale: [B] device floats. */
int fdbm_pc_corrector(void* x_new, void* x_mean, const void* x, const void* s, const void* y,
                      const void* noise, const float* a, const float* b, const float* den,
                      const float* step, const float* noise_scale, int B, int64_t n_complex,
                      void* stream);

/* Langevin corrector's step size on the device (fdbm/util/correctors.py:46-51): step[b] = (snr * mean_b ||noise_b|| /
 * (mean_b ||score_b|| + 1e-8))^2 * 2, noise_scale[b] = sqrt(step * 2), score as in fdbm_pc_corrector.  Two launches,
 * no host round trip (graph-capturable); scratch: B * 128 doubles. */
int fdbm_langevin_step(float* step /*[B]*/, float* noise_scale /*[B]*/, void* scratch, const void* x, const void* s,
                       const void* y, const void* noise, const float* a, const float* b, const float* den,
                       float snr, int B, int64_t n_complex, void* stream);

/* ------------------------------------------------------------------ Gaussian noise on the device
 * The per-step draws of the stochastic samplers (torch.randn_like of the complex state: fdbm/bridge.py:47,108,
 * fdbm/util/predictors.py:46, fdbm/util/correctors.py:48,76) from a COUNTER-BASED generator, so that the kernel that
 * consumes a draw generates it in registers: no host generator, no noise tensors in HBM.
 * Complex element e (flat index into the [B][1][F][T] state) of draw d under seed s:
 *     (x0, x1, ., .) = Philox4x32-10(counter = (e_lo, e_hi, d, 0x46444d42), key = (s_lo, s_hi))
 *     u1 = ((x0 >> 8) + 0.5) 2^-24,  u2 = ((x1 >> 8) + 0.5) 2^-24
 *     r = sqrt(-2 ln u1);  re = r cos(2 pi u2) sqrt(1/2);  im = r sin(2 pi u2) sqrt(1/2)        (Re, Im ~ N(0, 1/2))
 * (restated in numpy, with Random123's known-answer vectors: oracle/rng.py).  rng = three DEVICE words {s_lo, s_hi, d_base};
 * the draw used is d_base + `draw` (a sampler numbers its draws in call order: 0 = prior, 1.. = the steps'), so a
 * captured graph keeps its literal `draw` arguments while the host moves d_base / the seed between replays.
 * fdbm_randn_complex materialises one draw; the *_rng forms of the sampler kernels below take (rng, draw) in place of
 * their noise pointer and are bit-identical to feeding them fdbm_randn_complex's tensor.  Parity with the reference's
 * own generator is a matter of injected noise tensors (hard part 3), not of this stream. */
int fdbm_randn_complex(void* out /*c64 [n_complex]*/, int64_t n_complex, const uint32_t* rng, uint32_t draw, void* stream);
int fdbm_step_boundary_rng(void* x, const void* y, const uint32_t* rng, uint32_t draw, const float* pyramid,
                           const float* out_w, const float* out_b, const float* wa, const float* wb, const float* wc,
                           float* packed, void* zero_ptr, int64_t zero_bytes, float* dense_dst, const float* dense_src,
                           int64_t dense_n, int B, int F, int Fn, int T, void* stream);
int fdbm_pc_predictor_rng(void* x_new, void* x_mean, const void* x, const void* s, const void* y,
                          const uint32_t* rng, uint32_t draw, const float* wx, const float* ws, const float* wy,
                          const float* gd, float dt, int B, int64_t n_complex, void* stream);
int fdbm_pc_corrector_rng(void* x_new, void* x_mean, const void* x, const void* s, const void* y,
                          const uint32_t* rng, uint32_t draw, const float* a, const float* b, const float* den,
                          const float* step, const float* noise_scale, int B, int64_t n_complex, void* stream);

/* ------------------------------------------------------------------ ode_sampler_int on the device
 * The stage arithmetic of one Dormand-Prince RK45 step of `ode_sampler_int` (fdbm/bridge.py:115-140, which calls
 * scipy.integrate.solve_ivp(method="RK45") on the host: scipy/integrate/_ivp/rk.py rk_step / _step_impl) over the
 * complex128 state [n_complex], resident on the device (fdbm_amd/odeint.py drives the step-size control):
 *   fdbm_rk45_lincomb  out = y + scale * (c_0 K_0 + c_1 K_1 + ... + c_{nk-1} K_{nk-1}), terms added in index order, zero
 *                      coefficients skipped (their K_j may be NULL): the stage inputs and y_new.  K: HOST array of nk
 *                      device pointers, coef: HOST array of nk doubles (nk <= 7).  out may alias nothing it reads.
 *   fdbm_rk45_error    partial[0..255] = per-block sums of |(sum_j e_j K_j) h / (atol + rtol max(|y|, |y_new|))|^2; the error
 *                      norm of the step is sqrt(sum(partial) / n_complex).  K, e: HOST arrays of 7. */
int fdbm_rk45_lincomb(void* out, const void* y, const void* const* K, const double* coef, int nk, double scale,
                      int64_t n_complex, void* stream);
int fdbm_rk45_error(double* partial /*device [256]*/, const void* const* K, const double* e, const void* y, const void* y_new,
                    double h, double atol, double rtol, int64_t n_complex, void* stream);

/* ------------------------------------------------------------------ network input / output
 * cat(x.re, x.im, y.re, y.im) with the Nyquist row dropped when F == 257
 * (fdbm/backbones/ncsnpp_v2.py:247-250): complex64 [B][1][F][T] x2 -> f32 [B][Fn][T][4]. */
int fdbm_pack_input(float* out, const void* x, const void* y, int B, int F, int Fn, int T,
                    void* stream);

/* output_layer 1x1 conv 4->2, NHWC->complex, zero Nyquist row re-appended
 * (ncsnpp_v2.py:392-399): f32 [B][Fn][T][4] -> complex64 [B][1][F][T].  w: [2][4], b: [2]. */
int fdbm_unpack_output(void* out, const float* pyr, const float* w, const float* b,
                       int B, int F, int Fn, int T, void* stream);

/* ------------------------------------------------------------------ time embedding
 * GaussianFourierProjection(log t) -> Linear -> SiLU -> Linear, then SiLU again (every
 * consumer applies act(temb) first): out f32 [B][4nf] = silu(temb).  The Fourier argument
 * is formed as ((log t * W) * 2) * pi in fp32 (layerspp.py:40; ncsnpp_v2.py:252-270).
 * log_t is log(t) evaluated by the HOST: arguments reach thousands of radians, so a 1-ulp
 * difference between two logf moves sin/cos by up to 1e-4 (hard part 7).  Give the
 * CORRECTLY ROUNDED float32 logarithm (evaluate in float64, round once - fdbm_amd.hip.log_time):
 * a float32 logf may be one unit off near a rounding tie and two hosts' vector libms disagree
 * there (observed at t = 0.10009 of the 30-step grids, 0.4988 ulp from the tie). */
int fdbm_temb(float* out_act, const float* log_t, const float* fourier_w, const float* w1,
              const float* b1, const float* w2, const float* b2, float* scratch, int B, int nf,
              void* stream);

/* All res-blocks' Dense_0(act(temb)) at once (layerspp.py:262-263): out[b][r] =
 * bias[r] + sum_k w[r][k]*act[b][k];  w: [R][K] f32 (all blocks' rows concatenated). */
int fdbm_dense_rows(float* out, const float* act, const float* w, const float* bias, int B, int R,
                    int K, void* stream);

/* ------------------------------------------------------------------ stem (4 -> nf, 3x3)
 * modules[3] conv3x3 (ncsnpp_v2.py:154,278): in f32 [B][H][W][4], w f32 [nf][3][3][4]
 * (o, ky, kx, c), out dtype `dt_out` [B][H][W][nf]. */
int fdbm_conv_stem(void* out, const float* in, const float* w, const float* bias, int B, int H,
                   int W, int nf, int dt_out, void* stream);
/* the same, also accumulating the UNIT statistics of the stored output ((sum, sumsq) per image and
 * 4 channels, double [B][stat_nsplit][nf/4][2], atomics into a zeroed buffer - the layout of
 * fdbm_conv_args.stat_out with stat_G = nf/4) for the GroupNorms that read the stem's output;
 * needs nf/8 to divide 256.  stat_out NULL = fdbm_conv_stem. */
int fdbm_conv_stem_stats(void* out, const float* in, const float* w, const float* bias, int B, int H,
                         int W, int nf, int dt_out, double* stat_out, int stat_nsplit, void* stream);

/* ------------------------------------------------------------------ GroupNorm
 * nn.GroupNorm(min(C/4,32), C, eps=1e-6) (layerspp.py:67,219,231; ncsnpp_v2.py:205,217)
 * over a VIRTUAL CONCAT of up to two NHWC tensors (torch.cat([h, hs.pop()], 1),
 * ncsnpp_v2.py:330).  Stage 1: per-(b, split, group) partial sums; stage 2: mean/rstd.
 * bf16 tensors: float partials.  f32 tensors (parity mode): sums AND partials are fp64
 * ([B][nsplit][G][2] doubles) because the one-pass variance cancels when |mean| >> std;
 * every consumer takes such a buffer with nsplit NEGATED. */
int fdbm_gn_stats(float* partial /*[B][nsplit][G][2]*/, const void* src0, int C0,
                  const void* src1, int C1, int B, int HW, int G, int nsplit, int dtype,
                  void* stream);
int fdbm_gn_finalize(float* mean_rstd /*[B][G][2]*/, const float* partial, int B, int nsplit,
                     int G, int64_t count, float eps, void* stream);
/* out = act(gn(cat(src0, src1))) written as one NHWC tensor of C0+C1 channels.
 * stats: nsplit == 0 -> final mean_rstd [B][G][2]; nsplit > 0 -> the PARTIAL sums of
 * fdbm_gn_stats, reduced in-kernel (fp64, fixed order) with `count` elements per group, which
 * saves the finalize launch. */
int fdbm_gn_apply(void* out, const void* src0, int C0, const void* src1, int C1,
                  const float* stats, int nsplit, int64_t count, float eps, const float* gamma,
                  const float* beta, int B, int HW, int G, int silu, int dtype, void* stream);

/* ------------------------------------------------------------------ FIR resampling
 * Generic upfirdn2d, the reference's own native-op boundary
 * (ncsnpp_utils/op/upfirdn2d.cpp:4-23, upfirdn2d_kernel.cu:107-207,209-368):
 * input f32 [major][in_h][in_w][minor], kernel f32 [kh][kw], out f32
 * [major][out_h][out_w][minor]. */
int fdbm_upfirdn2d(float* out, const float* in, const float* kernel, int major, int in_h, int in_w,
                   int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                   int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);

/* Fast path for the only configuration the network uses: FIR [1,3,3,1], factor 2
 * (upsample_2d / downsample_2d, up_or_down_sampling.py:195-257), NHWC, dtype in {f32,bf16}.
 * One read of x produces up to two outputs: out_plain = resample(x) and, when stats is
 * non-NULL (same convention as fdbm_gn_apply), out_act = resample(silu(gn(x))) (ResnetBlockBigGANpp.forward,
 * layerspp.py:243-258).  Either output pointer may be NULL.  up != 0: upsample. */
int fdbm_resample2x(void* out_plain, void* out_act, const void* in, const float* stats, int nsplit,
                    int64_t count, float eps, const float* gamma, const float* beta, int B, int H,
                    int W, int C, int G, int up, int dtype, void* stream);
/* the same with UNIT statistics: `stats` points to DOUBLES, nsplit = -(number of partial rows), rows
 * hold G * stat_units (sum, sumsq) entries, stat_units consecutive ones per group - what a
 * convolution's stat_out leaves with stat_G = C/4 (stat_units = C/G/4). */
int fdbm_resample2x_units(void* out_plain, void* out_act, const void* in, const float* stats,
                          int nsplit, int stat_units, int64_t count, float eps, const float* gamma,
                          const float* beta, int B, int H, int W, int C, int G, int up, int dtype,
                          void* stream);
/* the progressive-input pyramid (ncsnpp_v2.py:296-305) in one launch: outs[l] = downsample_2d applied l+1
 * times to `in` (f32 [B][H][W][4]); outs = HOST array of `levels` device pointers, level l is
 * [B][H>>(l+1)][W>>(l+1)][4].  Same arithmetic as `levels` chained fdbm_resample2x calls. */
int fdbm_pyramid_down_chain(const float* in, void* const* outs, int levels, int B, int H, int W, void* stream);

/* ------------------------------------------------------------------ convolution (implicit GEMM on MFMA)
 * nn.Conv2d 3x3 pad 1 / 1x1 (layers.py:100-105,118-124) and NIN (layers.py:546-555) as one
 * GEMM  out[m][n] = sum_k A[m][k] W[n][k]  with m = (b, y, x) pixels and the K axis a list
 * of SEGMENTS, each an NHWC source tensor slice with 1 or 9 taps: a virtual concat is two
 * 9-tap segments, the res-block's 1x1 shortcut (Conv_2) is an extra 1-tap segment summed in
 * the same accumulator.  Weights are pre-packed by the host as [kstep][CoutPad][KC]
 * (KC = 64 bf16 / 32 f32 elements = 128 bytes), zero padded.
 * Epilogue: v = acc + bias[n] + tbias[b][n];  if res: v = (v + res[m][n]);  out = v*scale. */
typedef struct {
  const void* src;   /* NHWC tensor [B][H][W][C]                                   */
  int32_t C;         /* its channel count (pixel stride in elements)               */
  int32_t coff;      /* first channel of the slice                                 */
  int32_t cin;       /* channels in the slice                                      */
  int32_t taps;      /* 1 (1x1 / NIN / shortcut) or 9 (3x3, pad 1)                 */
} fdbm_conv_seg;

typedef struct {
  fdbm_conv_seg seg[FDBM_MAX_SEG];
  int32_t nseg;
  const void* w;        /* packed weights, dtype = dt_in                            */
  const float* bias;    /* [Cout] or NULL                                           */
  const float* tbias;   /* [B][tbias_stride] or NULL (Dense_0(act(temb)) rows)      */
  int32_t tbias_stride;
  const void* res;      /* [M][Cout] dtype dt_out or NULL                           */
  float scale;          /* 1/sqrt(2) with skip_rescale, else 1                      */
  void* out;            /* [M][Cout] dtype dt_out                                   */
  int32_t B, H, W, Cout, CoutPad;
  int32_t dt_in;        /* FDBM_F32 | FDBM_BF16 | FDBM_F16 : sources and packed weights */
  int32_t dt_out;       /* FDBM_F32, or dt_in (f32 out with 16-bit in is allowed)   */
  void* workspace;      /* optional split-K scratch (fp32 slabs) or NULL            */
  int64_t workspace_bytes;
  /* optional GroupNorm(+SiLU) prologue: act(gn(x)) is applied to the segments selected by
   * seg_gn_mask (in order they form the normalised, virtually concatenated tensor of gn_C
   * channels) while they are staged, zero padding after the activation - i.e.
   * conv(act(GroupNorm(cat(...)))) of layerspp.py:243-266 in one kernel.  gn_sums holds
   * (sum, sumsq) per (image, split, group): [B][gn_nsplit][gn_G][2]. */
  const float* gn_sums;
  const float* gn_gamma;
  const float* gn_beta;
  int32_t gn_nsplit, gn_G, gn_C, gn_silu;
  int64_t gn_count;     /* elements per (image, group)                              */
  float gn_eps;
  uint32_t seg_gn_mask;
  /* optional Combine('sum') epilogue (layerspp.py:52-57): out += comb_b[n] + comb_w[n][0..3] .
   * comb_pyr[m][0..3]  (f32 [M][4], f32 [Cout][4], f32 [Cout]) */
  const float* comb_pyr;
  const float* comb_w;
  const float* comb_b;
  /* optional: accumulate (sum, sumsq) of the STORED output per (image, group of Cout/stat_G
   * channels) into stat_out[B][stat_nsplit][stat_G][2] (stat_G <= 64) with atomics (caller zeroes it; a block
   * adds into row blockIdx.x % stat_nsplit so the atomics do not all hit one address) - the
   * statistics the consuming GroupNorm needs, without another pass over the tensor; same layout
   * as fdbm_gn_stats' partial sums. */
  double* stat_out;     /* fp64: the order of the atomics then does not show in the statistics */
  int32_t stat_G;
  int32_t stat_nsplit;
  /* optional: the same packed weights in MFMA-fragment-major order
   * [kstep][CoutPad/16][8 chunks][16 rows][16 bytes] (row r, chunk c of `w` moves to
   * [r/16][c][r%16]); the wave-per-tap kernel (plan kind 2) and the kernels that take its place on
   * the small launches - whole-map / band (csrc/conv_small.hip, conv_small_split.hip), 64-channel
   * block (csrc/conv_mid.hip) - load their operand fragments straight from it with 1 KiB-contiguous
   * wave loads.  NULL: kind 2 is never selected. */
  const void* w_frag;
  /* optional, instead of gn_sums: UNIT statistics of the GroupNorm input, one buffer per flagged
   * segment s: gn_seg_sums[s] = double [B][gn_seg_nsplit[s]][cin_s/4][2], (sum, sumsq) over units of
   * 4 channels - exactly what a convolution writing that tensor leaves in stat_out with
   * stat_G = Cout/4.  Any GroupNorm whose group size is a multiple of 4 (all of NCSN++'s) is then
   * normalised from its producers' epilogues, also across a channel concatenation whose group
   * boundaries straddle the sources (e.g. 256 + 128 channels, 32 groups of 12).  The flagged
   * segments must be segments 0 .. n-1 (seg_gn_mask = 2^n - 1). */
  const double* gn_seg_sums[FDBM_MAX_SEG];
  int32_t gn_seg_nsplit[FDBM_MAX_SEG];
  /* optional: ZERO-INITIALISED scratch of at least 65536 + 2*M*Cout*4 bytes (65536 + 8*M*Cout*4 lets it use
   * up to 8 slices).  When given, the wave-per-tap kernel may split the input-channel chunks of a small-map
   * convolution over several workgroups; each writes its partial tile into its own fp32 slab and the one
   * that finishes last sums the slabs in slice order (results are run-to-run identical) and applies the
   * epilogue.  Only the arrival counters (the first 64 KiB) need the zero initialisation; they are left
   * zeroed for the next launch.  NULL: never split. */
  void* acc_ws;
  int64_t acc_ws_bytes;
  /* optional residual given at HALF resolution, f32 [B][H/2][W/2][Cout]: upsample_2d ([1,3,3,1] FIR,
   * up_or_down_sampling.py:181-216) of it is added before `scale`, evaluated in the epilogue with the tap
   * order of fdbm_resample2x - the progressive-output pyramid (ncsnpp_v2.py:379-383) without its own
   * upsampling launch.  H and W even. */
  const float* res_up2x;
  /* optional SPLIT-PRECISION matrix mode for FDBM_F32 tensors (mma_mode 1; 0 = exact f32 MFMA, the default):
   * f32 storage, f32 GroupNorm / SiLU / epilogue exactly as in the f32 mode, but the products run on the 16-bit
   * matrix pipe with both operands carried as two IEEE halves, x ~ hi + lo (22 significant bits):
   *   a.w ~ a_hi.w_hi + a_hi.w_lo + a_lo.w_hi   (three v_mfma_f32_16x16x32_f16 per 32 input channels, f32 sums;
   *   the dropped a_lo.w_lo term is 2^-22 of the product) - 5.3x less matrix-pipe time than v_mfma_f32_16x16x4_f32.
   * Activations are split where they are staged (hi = half(16 a), lo = half(16 a - hi), |16 a| clamped to 65504).
   * `w` / `w_frag` then hold PRE-SPLIT weights: per k-step and output channel a 128-byte row of 32 input channels as
   * [32 halves hi | 32 halves lo] of  s_w * w  with s_w a power of two chosen by the packer (max |s_w w| < 2^15);
   * acc_scale = 1 / (16 s_w) is applied to the f32 sums before the bias.  Halo-patch and wave-per-tap kernels only
   * (plan kinds 1 and 2); fdbm_conv_igemm fails for a kind-0 shape with mma_mode 1. */
  int32_t mma_mode;
  float acc_scale;
} fdbm_conv_args;

int fdbm_conv_igemm(const fdbm_conv_args* a, void* stream);
/* elements per k-step for a dtype (64 for bf16, 32 for f32) */
int fdbm_conv_kc(int dtype);
/* tile / split-K plan the launcher will use for M pixels, Cout channels, nk k-steps
 * (the caller sizes `workspace` as ksplit * M * Cout * 4 bytes when ksplit > 1) */
int fdbm_conv_plan(int64_t M, int Cout, int nk, int* bm, int* bn, int* ksplit);
/* full plan: kind 1 = halo-patch 3x3 kernel (tile th x 16 pixels, always inside one image),
 * kind 2 = wave-per-tap 3x3 kernel for small grids (th = tile width 16|8|4, bn = 16 x n-tiles,
 * tile inside one image, no workspace), kind 0 = tap-outer implicit GEMM with the (bm, bn,
 * ksplit) of fdbm_conv_plan */
int fdbm_conv_plan_ex(int B, int H, int W, int Cout, int nk, int first_taps, int* kind, int* th,
                      int* bm, int* bn, int* ksplit);
/* kernel-selection policy used by fdbm_conv_plan_ex / fdbm_conv_igemm: bit 0 allows kind 1, bit 1
 * allows kind 2, bit 2 turns on k-groups inside the kind-0 kernel, bit 3 lets the producer / consumer
 * ring kernel (csrc/conv_ring_impl.h: 16-bit tensors; 16 x 16 pixel tiles when the launch has >= 200 of them,
 * else 8 x 16 pixel tiles when it has >= 128 of those) take the place of kinds 1 and 2, bit 4 makes it use
 * 8 x 16 pixel tiles wherever they fit (tests), bit 5 lets the lean kernels of the small launches take kind 2's place
 * where their shapes allow (csrc/conv_small.hip: whole-map / band, <= 512 workgroups, maps up to 32 x 32;
 * csrc/conv_small_split.hip: the same in mma_mode 1; csrc/conv_mid.hip: 192-512 tiles of 4 x 16 pixels x 64 channels,
 * the 64 x 64 level at batch 1); mask < 0 only queries.  Returns the previous mask (default 43 = bits 0, 1, 3, 5).
 * Results are the same convolution under every policy. */
int fdbm_conv_policy(int mask);
/* kernel family of the most recent fdbm_conv_igemm launch of this process: 0 tap-outer implicit GEMM,
 * 1 halo-patch, 2 wave-per-tap, 3 ring (16 x 16 pixel tiles), 4 ring (8 x 16 pixel tiles), 5 head (<= 16 f32 output
 * channels, weights resident in LDS), 6 whole-map / band kernel of the <= 32 x 32 maps, 7 64-channel-block kernel of the
 * 64 x 64 level; -1 before the first call (for measurement harnesses) */
int fdbm_conv_last_kind(void);

/* Combine('sum'): out = conv1x1(4->C)(pyr) + h (layerspp.py:52-57; ncsnpp_v2.py:302-305).
 * pyr f32 [M][4], w f32 [C][4], bias f32 [C], h/out dtype [M][C]; out may alias h. */
int fdbm_combine(void* out, const void* h, const float* pyr, const float* w, const float* bias,
                 int64_t M, int C, int dtype, void* stream);

/* ------------------------------------------------------------------ attention
 * AttnBlockpp core (layerspp.py:82-86): softmax(q k^T * C^-0.5) v over all H*W tokens, one
 * head.  qkv: [B][N][3C] (q | k | v per token, dtype), out [B][N][C] dtype. */
int fdbm_attention(void* out, const void* qkv, int B, int N, int C, int dtype, void* stream);

/* ------------------------------------------------------------------ front-end
 * SpecsDataModule.stft / istft (fdbm/data_module.py:201-229; torch.stft centre, reflect pad,
 * one-sided), spec_fwd / spec_back (:173-199) fused in, pad_spec (fdbm/util/other.py:76-90).
 * n_fft must be even and <= 1024; window f32 [n_fft].
 * transform: 0 none, 1 exponent (|X|^e e^{j arg X} * f), 2 log. */
int fdbm_stft(void* spec /*c64 [B][n_fft/2+1][Tpad]*/, const float* wave /*[B][L]*/,
              const float* window, int B, int L, int n_fft, int hop, int frames, int Tpad,
              int pad_mode /*0 zero, 1 reflection*/, int transform, float factor, float exponent,
              void* stream);
int fdbm_istft(float* wave /*[B][L]*/, const void* spec /*c64 [B][bins][Tpad]*/,
               const float* window, float* frames_ws /*[B][frames][n_fft]*/, int B, int L,
               int n_fft, int hop, int frames, int Tpad, int transform, float factor,
               float exponent, void* stream);

/* Waveform normalisation either side of the path (infer_folder.py:102-107,118-121; infer_single.py:79-84,97-99;
 * model.py:391-397,403), fused into the front-end:
 *   fdbm_wave_norm_factor  nf[b] = max |y_b| (mode 0, normalize == "noisy") or torch.std(y_b) (mode 1, unbiased)
 *   fdbm_stft_norm         fdbm_stft of y / nf (the division where the frame is windowed: same two roundings)
 *   fdbm_istft_renorm      fdbm_istft, then x_hat * nf and, if max |x_hat| > 1, x_hat / max |x_hat| * clip
 *                          (clip 0.95: infer_folder.py:121, 0.5: infer_single.py:99, 0: no clip rule);
 *                          peak_ws: B x 4 bytes of scratch (holds max |x_hat * nf| per clip afterwards, as a float) */
int fdbm_wave_norm_factor(float* nf /*[B]*/, const float* wave /*[B][L]*/, int B, int L, int mode, void* stream);
int fdbm_stft_norm(void* spec, const float* wave, const float* window, const float* norm /*[B]*/, int B, int L,
                   int n_fft, int hop, int frames, int Tpad, int pad_mode, int transform, float factor,
                   float exponent, void* stream);
int fdbm_istft_renorm(float* wave, const void* spec, const float* window, float* frames_ws, const float* norm /*[B]*/,
                      void* peak_ws /*[B] x 4 bytes*/, float clip, int B, int L, int n_fft, int hop, int frames,
                      int Tpad, int transform, float factor, float exponent, void* stream);

/* Stand-alone spec_fwd (inverse=0) / spec_back (inverse=1) on complex64 data
 * (fdbm/data_module.py:173-199) and pad_spec (fdbm/util/other.py:76-90; mode 0 zero_pad,
 * 1 reflection, 2 replication; in [rows][T] -> out [rows][Tpad] complex64). */
int fdbm_spec_transform(void* out, const void* in, int64_t n_complex, int transform, float factor,
                        float exponent, int inverse, void* stream);
int fdbm_pad_spec(void* out, const void* in, int64_t rows, int T, int Tpad, int mode, void* stream);

/* ------------------------------------------------------------------ recorded programs
 * A backbone forward is a fixed list of the calls above.  The host records it once
 * (opcode + argument block per op) and replays it with one call (see fdbm_ncsnpp_forward). */
typedef struct {
  int32_t opcode;        /* FDBM_OP_* */
  int32_t lane;          /* 0 = the caller's stream, 1 = the library's side stream (see FDBM_OP_FORK) */
  int64_t iarg[24];      /* integer / pointer arguments, meaning per opcode (program.cpp) */
  float farg[4];
} fdbm_op;

#define FDBM_OP_CONV 1        /* iarg[0] = (const fdbm_conv_args*) host pointer */
#define FDBM_OP_GN_STATS 2
#define FDBM_OP_GN_FINALIZE 3
#define FDBM_OP_GN_APPLY 4
#define FDBM_OP_RESAMPLE 5
#define FDBM_OP_COMBINE 6
#define FDBM_OP_ATTENTION 7
#define FDBM_OP_STEM 8
#define FDBM_OP_PACK 9
#define FDBM_OP_UNPACK 10
#define FDBM_OP_TEMB 11
#define FDBM_OP_DENSE 12
#define FDBM_OP_UPDATE 13
#define FDBM_OP_MEMSET 14      /* iarg: ptr, bytes (hipMemsetAsync to 0) */
/* Off-critical-path chains (the 4-channel pyramids, the time-embedding MLP) may run on a side stream
 * beside the main chain; the dependencies are explicit ops, all capturable into a HIP graph:
 *   FORK e : everything enqueued so far on the caller's stream happens before the side-lane ops that follow
 *   MARK e : (side lane) a point in the side stream ...
 *   JOIN e : ... that the caller's stream waits for before going on.
 * Unless fdbm_runtime_init_side() has created the side stream (it does so only with FDBM_SIDE_STREAM=1 in
 * the environment: inside a HIP graph the cross-stream edges measured slower than what they save), every op
 * runs on the caller's stream in program order and these three are no-ops - the results are identical. */
#define FDBM_OP_FORK 15        /* iarg: event id */
#define FDBM_OP_MARK 16        /* iarg: event id */
#define FDBM_OP_JOIN 17        /* iarg: event id */
#define FDBM_OP_PYRDOWN 18     /* iarg: in, levels, B, H, W, out[0..7] */
#define FDBM_MAX_EVENTS 64

/* creates the side stream and FDBM_MAX_EVENTS events once per process (call it outside graph capture;
 * the only objects the library ever creates) */
int fdbm_runtime_init_side(void);

int fdbm_memset_zero(void* ptr, int64_t bytes, void* stream);   /* a kernel, not hipMemsetAsync (16-byte aligned ptr) */
int fdbm_copy_f32(float* dst, const float* src, int64_t n, void* stream);   /* a kernel, not hipMemcpyAsync */
int fdbm_run_program(const fdbm_op* ops_host, int n_ops, void* stream);

/* ------------------------------------------------------------------ composed backbone
 * s = dnn(x_t, y, t) (BridgeModel.forward, fdbm/model.py:356-357 -> NCSNpp_v2.forward,
 * ncsnpp_v2.py:241-401) as ONE call: a context wraps a recorded program (copied) and the static
 * device buffers it reads / writes.  x, y, out: complex64 [B][1][F][T]; log_t: f32 [B] (host-
 * evaluated log of the model time, see fdbm_temb).  Passing the context's own buffers skips the
 * copies.  One context per device per process; calls on a context are not re-entrant. */
typedef struct fdbm_ncsnpp_ctx fdbm_ncsnpp_ctx;
fdbm_ncsnpp_ctx* fdbm_ncsnpp_create(const fdbm_op* ops_host, int n_ops, void* x_in, void* y_in,
                                    float* logt_in, void* s_out, int64_t n_complex, int B);
void fdbm_ncsnpp_destroy(fdbm_ncsnpp_ctx* ctx);
int fdbm_ncsnpp_forward(fdbm_ncsnpp_ctx* ctx, const void* x, const void* y, const float* log_t,
                        void* out, void* stream);

/* ------------------------------------------------------------------ contexts without Python
 * The architecture walk of NCSNpp_v2 (fdbm/backbones/ncsnpp_v2.py:95-401) is recorded once per (variant, B, F, T,
 * dtype) and SERIALISED (python -m tools.export_program, fdbm_amd/export.py): a "program" (this header, the fdbm_op
 * array, the fdbm_conv_args table; every pointer stored as region + offset) and a flat weight blob in the kernels'
 * device formats.  A host in any language then needs: the two files, ONE device buffer for the weights (upload the
 * blob), ONE caller-owned workspace of fdbm_program_workspace_bytes() bytes, and
 *     ctx = fdbm_ncsnpp_create_from_program(program, n, weights_dev, workspace_dev, workspace_bytes);
 *     fdbm_ncsnpp_forward(ctx, x, y, log_t, out, stream);       // never allocates, never synchronises
 * (examples/host_cpp/run_program.cpp is such a host).  Pointers inside the program: bit 62 set, bits 60-61 = region
 * (0 workspace, 1 weights), low bits = byte offset. */
typedef struct fdbm_program_header {
  uint64_t magic;                 /* 0x474F5250424D4446 */
  uint32_t version, n_ops, n_conv, conv_args_size;
  int64_t workspace_bytes, weights_bytes;
  int32_t B, F, T, dtype;
  int64_t x_in, y_in, t_in, s_out;    /* relocatable pointers of the static I/O buffers */
  int64_t n_complex;
  uint8_t reserved[32];
} fdbm_program_header;             /* 128 bytes, followed by fdbm_op[n_ops], fdbm_conv_args[n_conv] */
int64_t fdbm_program_workspace_bytes(const void* program, int64_t nbytes);    /* < 0: not a program */
int64_t fdbm_program_weights_bytes(const void* program, int64_t nbytes);
/* the workspace is zeroed here (synchronously: context creation is not on the hot path) */
fdbm_ncsnpp_ctx* fdbm_ncsnpp_create_from_program(const void* program, int64_t nbytes, void* weights_dev,
                                                 void* workspace_dev, int64_t workspace_bytes);

/* ---- TF-GridNet backbone (fdbm/backbones/tfgridnet.py:83-510; BackboneRegistry "tfgridnet_5l32c100", "tfgridnet_4l32c80") ----
 * The context the survey asks for at this boundary: created from an architecture descriptor + ONE flat f32 weight blob
 * on the device, evaluated on caller-owned buffers (f32 throughout).  Blob order (fdbm_amd/tfgridnet.py:pack_state):
 *   stem conv w [C][3][3][in_ch], b [C]; GroupNorm gamma, beta [C];
 *   per block, for r in (intra, inter): LayerNorm gamma, beta [C]; W_in [8H][ks*C] (rows: forward i f g o | reverse i f g o;
 *     column i*C + c = weight_ih[.., c*ks + i]), b_in [8H] = bias_ih + bias_hh; W_hh forward [4H][H], reverse [4H][H];
 *     W_dec [C][ks*2H] (column j*2H + h = ConvTranspose1d weight[h][c][ks-1-j]), b_dec [C];
 *   then W_qkv [2 nh E + C][C] (attn_conv_Q | K | V), b_qkv; PReLU slopes [3][nh]; head-norm gamma, beta [2 nh E + C];
 *     W_proj [C][C], b_proj [C]; PReLU slope [1]; LayerNormalization gamma, beta [C];
 *   head conv (ConvTranspose2d as a convolution) w [out_ch][3][3][C] (= deconv.weight[c][o][2-ky][2-kx]), b [out_ch];
 *   Fourier W [C]; time_emb_fc.0 w [4C][2C], b; time_emb_fc.2 w [4C][4C], b; time_emb_blocks w [n_layers][C][4C], b [n_layers][C]. */
typedef struct fdbm_tfgridnet_desc {
  int32_t n_layers, emb_dim, hidden, emb_ks, n_head, qk_channels, in_ch, out_ch;   /* in_ch = 4, out_ch = 2 */
  float eps;
} fdbm_tfgridnet_desc;
typedef struct fdbm_tfgridnet_ctx fdbm_tfgridnet_ctx;
int64_t fdbm_tfgridnet_weights_count(const fdbm_tfgridnet_desc* d);                 /* floats in the blob; < 0: unsupported */
int64_t fdbm_tfgridnet_workspace_bytes(const fdbm_tfgridnet_desc* d, int B, int F, int T);
fdbm_tfgridnet_ctx* fdbm_tfgridnet_create(const fdbm_tfgridnet_desc* d, const float* weights_dev, int64_t n_weights);
void fdbm_tfgridnet_destroy(fdbm_tfgridnet_ctx* ctx);
/* model(x, y, t) of tfgridnet.py:194-232: x, y, out complex64 [B][1][F][T]; log_t f32 [B] = log(t) evaluated on the host;
 * block_out (may be NULL): f32 [n_layers][B][T][F][C], every block's output (parity tests). */
int fdbm_tfgridnet_forward(fdbm_tfgridnet_ctx* ctx, const void* x, const void* y, const float* log_t, void* out,
                           int B, int F, int T, void* workspace, int64_t workspace_bytes, float* block_out, void* stream);
/* The same evaluation entered at block `first_block` (1 .. n_layers - 1) with block_in f32 [B][T][F][C] as that block's
 * input (= the previous GridNetV3Block's output, tfgridnet.py:224-225); the stem is skipped, everything behind runs as in
 * fdbm_tfgridnet_forward (block_out rows < first_block are left untouched).  For teacher-forced per-block parity tests:
 * recurrent blocks amplify fp32 rounding, so a free-running comparison only bounds the first block tightly. */
int fdbm_tfgridnet_forward_from(fdbm_tfgridnet_ctx* ctx, const float* block_in, int first_block, const float* log_t, void* out,
                                int B, int F, int T, void* workspace, int64_t workspace_bytes, float* block_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FDBM_HIP_H */
