// chain.hip - MEASUREMENT PROTOTYPE (not part of the library): a chain of L small-map 3x3 convolutions
//   x_{l+1} = conv3x3_{256 -> 256}( silu( GroupNorm_32( x_l ) ) ) + bias        on an S x S map (S = 4 | 8), bf16 NHWC
// evaluated (A) as L kernel launches replayed from ONE HIP graph and (B) by ONE persistent kernel whose workgroups
// hand each layer's output to the next through write-through (sc1) stores + an arrival counter, prefetching the next
// layer's weight fragments BEFORE they wait.  Same workgroup body in both, bit-identical outputs.
// Answers (with a timeline, not a price list - VERDICT r2, item 1b): what does a layer of the <= 8 x 8 levels cost as a
// launch and as a phase of a persistent kernel, for a minimal kernel body?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/persist_proto/chain.hip -o /tmp/chain && /tmp/chain [S] [L] [WGS]
// Workgroup = 8 waves = taps 0..7 (wave 0 also tap 8) of 16 output channels x all S*S pixels; C/16 = 16 workgroups per
// layer (x 2 | 4 pixel groups of 16 at S = 8).  GroupNorm statistics are computed BY THE CONSUMER from the whole map it
// stages anyway (the map is tiny): no statistics buffers, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int C = 256;          // channels in = out
constexpr int G = 32;           // GroupNorm groups (8 channels each)
constexpr int NTHR = 512;

struct LayerW {                 // one layer's parameters
  const bf16_t* w;              // fragment-major [9 taps][C/16 n-tiles][C/32 k-steps][4 k-groups][16 rows][8] bf16
  const float* bias;            // [C]
  const float* gamma;           // [C]
  const float* beta;            // [C]
};

// ---- the workgroup body: one (n-tile, pixel group) of one layer ----------------------------------------------------
// in / out: [S*S][C] bf16.  SC1: loads of `in` bypass the L1 (sc1) and stores of `out` are write-through (sc1) - the
// persistent variant's hand-off; plain in the launch variant.
template <int S, bool SC1>
__device__ __forceinline__ void layer_body(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, const LayerW lw, int ntile, int pgroup,
                                           unsigned char* smem, uint4 (&wf)[9]) {
  // wf: this wave's weight fragments, ALREADY LOADED by the caller (the persistent kernel requests them before it waits)
  constexpr int HW = S * S;
  constexpr int PW = S + 2;                       // padded map
  constexpr int PROWS = PW * PW;
  bf16_t* s_act = reinterpret_cast<bf16_t*>(smem);                  // [PROWS][C + 8] activated, zero border (row stride 528 B: conflict-free b128)
  constexpr int RS = C + 8;
  float* s_red = reinterpret_cast<float*>(smem + PROWS * RS * 2);   // [8 waves][64 lanes][4] partial accumulators
  float* s_stat = s_red + 8 * 64 * 4;                               // [16 parts][G][2] partial sums (fixed summation order)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // this thread's GroupNorm parameters (its items all belong to group tid % 32): requested before anything else
  const int g_own = tid % (C / 8);
  const f32x4 ga0 = *reinterpret_cast<const f32x4*>(lw.gamma + g_own * 8), ga1 = *reinterpret_cast<const f32x4*>(lw.gamma + g_own * 8 + 4);
  const f32x4 be0 = *reinterpret_cast<const f32x4*>(lw.beta + g_own * 8), be1 = *reinterpret_cast<const f32x4*>(lw.beta + g_own * 8 + 4);
  const float gam[8] = {ga0[0], ga0[1], ga0[2], ga0[3], ga1[0], ga1[1], ga1[2], ga1[3]};
  const float bet[8] = {be0[0], be0[1], be0[2], be0[3], be1[0], be1[1], be1[2], be1[3]};
  // ---- stage the whole map: 16-byte items (pixel, group of 8 channels) ------------------------------------------------
  constexpr int NITEM = HW * (C / 8);              // 512 (S = 4) | 2048 (S = 8)
  constexpr int NPT = NITEM / NTHR;                // 1 | 4
  // (named registers, not an array: inline asm cannot tie array elements, and arrays behind asm end up in scratch memory)
  u32x4 raw0 = {0u, 0u, 0u, 0u}, raw1 = raw0, raw2 = raw0, raw3 = raw0;
#define STAGE_LOAD(J, R)                                                                         \
  if constexpr (NPT > J) {                                                                       \
    const int q = tid + NTHR * J;                                                                \
    const bf16_t* src = in + (q / (C / 8)) * C + (q % (C / 8)) * 8;                              \
    if constexpr (SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(R) : "v"(src) : "memory"); \
    else R = *reinterpret_cast<const u32x4*>(src);                                               \
  }
  STAGE_LOAD(0, raw0) STAGE_LOAD(1, raw1) STAGE_LOAD(2, raw2) STAGE_LOAD(3, raw3)
#undef STAGE_LOAD
  for (int i = tid; i < PROWS * RS / 8; i += NTHR) reinterpret_cast<uint4*>(s_act)[i] = uint4{0u, 0u, 0u, 0u};
  if constexpr (SC1) {      // (the asm loads are invisible to the compiler's wait insertion: wait here, and tie the registers to the wait)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw0), "+v"(raw1), "+v"(raw2), "+v"(raw3)::"memory");
  }
  const u32x4 raw[4] = {raw0, raw1, raw2, raw3};
  // GroupNorm statistics of the whole map: item = one group at one pixel (8 channels); thread tid owns group tid % 32 at
  // pixels tid / 32 + 16 j: its own sum in registers, then the 16 partial sums of a group in a fixed order (no atomics)
  float v[NPT][8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const bf16x8 e = *reinterpret_cast<const bf16x8*>(&raw[j]);
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[j][k] = (float)e[k]; s1 += v[j][k]; s2 += v[j][k] * v[j][k]; }
  }
  s_stat[((tid >> 5) * G + g_own) * 2] = s1;
  s_stat[((tid >> 5) * G + g_own) * 2 + 1] = s2;
  __syncthreads();
  float t1 = 0.f, t2 = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) { t1 += s_stat[(q * G + g_own) * 2]; t2 += s_stat[(q * G + g_own) * 2 + 1]; }
  const float mean = t1 * (1.f / (HW * 8)), var = fmaxf(t2 * (1.f / (HW * 8)) - mean * mean, 0.f);
  const float rstd = __builtin_amdgcn_rsqf(var + 1e-6f);
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const int q = tid + NTHR * j, px = q / (C / 8), g = g_own;
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float y = (v[j][k] - mean) * rstd * gam[k] + bet[k];
      o[k] = (bf16_t)(y * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * y)));
    }
    const int py = px / S + 1, pxx = px % S + 1;
    *reinterpret_cast<bf16x8*>(s_act + (py * PW + pxx) * RS + g * 8) = o;
  }
  __syncthreads();
  // ---- MFMA: wave w = tap w (wave 0 also tap 8); 16 pixels of this pixel group x 16 output channels ---------------------
  const int frow = lane & 15, fk = lane >> 4;
  const int p = pgroup * 16 + frow, py = p / S, pxx = p % S;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  auto tap_mma = [&](int tap, const uint4* wfr) __attribute__((always_inline)) {
    const int dy = tap / 3, dx = tap % 3;
    const bf16_t* arow = s_act + ((py + dy) * PW + pxx + dx) * RS + fk * 8;
#pragma unroll
    for (int ks = 0; ks < C / 32; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + ks * 32);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wfr[ks]), a, acc, 0, 0, 0);
    }
  };
  // (weights: the caller loaded this wave's tap, 8 k-steps = 8 uint4, + for wave 0 tap 8's in wf8)
  tap_mma(wave, wf);
  *reinterpret_cast<f32x4*>(s_red + (wave * 64 + lane) * 4) = acc;
  (void)ntile;
}

// weight fragments of (layer, n-tile, tap): 8 k-steps x 1 uint4 per lane
__device__ __forceinline__ void load_wfrag(const LayerW lw, int ntile, int tap, int lane, uint4 (&wf)[9]) {
  const bf16_t* base = lw.w + (((int64_t)tap * (C / 16) + ntile) * (C / 32)) * 512 + lane * 8;     // [k-step][64 lanes][8]
#pragma unroll
  for (int ks = 0; ks < C / 32; ++ks) wf[ks] = *reinterpret_cast<const uint4*>(base + ks * 512);
}

template <int S, bool SC1>
__device__ __forceinline__ void layer_finish(bf16_t* __restrict__ out, const LayerW lw, int ntile, int pgroup, unsigned char* smem,
                                             const uint4 (&wf8)[9], const bf16_t* s_act_dummy) {
  (void)s_act_dummy;
  constexpr int PW = S + 2, PROWS = PW * PW, RS = C + 8;
  bf16_t* s_act = reinterpret_cast<bf16_t*>(smem);
  float* s_red = reinterpret_cast<float*>(smem + PROWS * RS * 2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fk = lane >> 4;
  __syncthreads();
  if (wave == 0) {
    // tap 8 + the sum over the 8 waves + bias + store
    const int p = pgroup * 16 + frow, py = p / S, pxx = p % S;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bf16_t* arow = s_act + ((py + 2) * PW + pxx + 2) * RS + fk * 8;
#pragma unroll
    for (int ks = 0; ks < C / 32; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + ks * 32);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&wf8[ks]), a, acc, 0, 0, 0);
    }
#pragma unroll
    for (int w = 0; w < 8; ++w) acc += *reinterpret_cast<const f32x4*>(s_red + (w * 64 + lane) * 4);
    // lane (pixel frow, fk) holds output channels ntile*16 + fk*4 .. +3 of pixel p
    const int n = ntile * 16 + fk * 4;
    const f32x4 b = *reinterpret_cast<const f32x4*>(lw.bias + n);
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 o = {(bf16_t)(acc[0] + b[0]), (bf16_t)(acc[1] + b[1]), (bf16_t)(acc[2] + b[2]), (bf16_t)(acc[3] + b[3])};
    bf16_t* dst = out + p * C + n;
    if constexpr (SC1) {
      const uint2 ov = *reinterpret_cast<const uint2*>(&o);
      asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(ov) : "memory");
    } else {
      *reinterpret_cast<bf16x4*>(dst) = o;
    }
  }
}

// ---- (A) one launch per layer --------------------------------------------------------------------------------------
template <int S>
__global__ void __launch_bounds__(NTHR) layer_kernel(const bf16_t* in, bf16_t* out, const LayerW lw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ntile = blockIdx.x % (C / 16), pgroup = blockIdx.x / (C / 16);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4 wf[9], wf8[9];
  load_wfrag(lw, ntile, wave, lane, wf);
  if (wave == 0) load_wfrag(lw, ntile, 8, lane, wf8);
  layer_body<S, false>(in, out, lw, ntile, pgroup, smem, wf);
  layer_finish<S, false>(out, lw, ntile, pgroup, smem, wf8, nullptr);
}

// ---- (B) persistent: all layers in one launch ------------------------------------------------------------------------
// counter[l] counts the workgroups that have PUBLISHED layer l's output (write-through stores drained, then one agent-scope
// atomic add per workgroup); a consumer polls it with sc1 loads.  Two ping-pong activation buffers.
struct ChainArgs {
  const LayerW* layers;      // device array [L]
  bf16_t* buf[2];
  unsigned* counter;         // [L + 1], zeroed before the launch
  int L, nwg;
  unsigned long long* stamps;   // optional [L][2] realtime stamps of workgroup 0 (layer start, wait done)
};

template <int S>
__global__ void __launch_bounds__(NTHR) chain_kernel(const ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ntile = blockIdx.x % (C / 16), pgroup = blockIdx.x / (C / 16);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4 wf[9], wf8[9];
  LayerW lw = a.layers[0];
  load_wfrag(lw, ntile, wave, lane, wf);
  if (wave == 0) load_wfrag(lw, ntile, 8, lane, wf8);
  for (int l = 0; l < a.L; ++l) {
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[l * 2] = __builtin_amdgcn_s_memrealtime();
    if (l > 0) {
      // wait until every workgroup has published layer l - 1 (bounded spin: a lost workgroup must not hang the GPU)
      if (threadIdx.x == 0) {
        unsigned seen = 0;
        for (int spin = 0; spin < (1 << 22); ++spin) {
          asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(a.counter + l) : "memory");
          if (seen >= (unsigned)a.nwg) break;
          __builtin_amdgcn_s_sleep(2);
        }
      }
      __syncthreads();
    }
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[l * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    const bf16_t* in = a.buf[l & 1];
    bf16_t* out = a.buf[(l + 1) & 1];
    layer_body<S, true>(in, out, lw, ntile, pgroup, smem, wf);
    // next layer's parameters and weight fragments: requested BEFORE this layer's hand-off (they depend on nothing)
    uint4 nwf[9], nwf8[9];
    LayerW nlw = lw;
    if (l + 1 < a.L) {
      nlw = a.layers[l + 1];
      load_wfrag(nlw, ntile, wave, lane, nwf);
      if (wave == 0) load_wfrag(nlw, ntile, 8, lane, nwf8);
    }
    layer_finish<S, true>(out, lw, ntile, pgroup, smem, wf8, nullptr);
    // publish: every storing wave drains its write-through stores, the workgroup meets, ONE lane counts the workgroup in
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(a.counter + l + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (l + 1 < a.L) {
      lw = nlw;
#pragma unroll
      for (int k = 0; k < 9; ++k) { wf[k] = nwf[k]; wf8[k] = nwf8[k]; }
    }
  }
}

// ---- host ------------------------------------------------------------------------------------------------------------
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float frand(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((s >> 33) & 0xFFFFFF) / 8388608.0f - 1.0f; }

template <int S>
static int run(int L, int reps) {
  constexpr int HW = S * S, NWG = (C / 16) * (HW / 16);
  constexpr size_t SMEM = (size_t)(S + 2) * (S + 2) * (C + 8) * 2 + 8 * 64 * 4 * 4 + 16 * G * 2 * 4;
  uint64_t seed = 12345;
  std::vector<LayerW> hl(L);
  for (int l = 0; l < L; ++l) {
    std::vector<uint16_t> w((size_t)9 * C * C);
    for (auto& x : w) x = f2bf(frand(seed) * 0.036f);              // ~ N(0, 1 / fan_in)-sized
    std::vector<float> b(C), ga(C), be(C);
    for (int i = 0; i < C; ++i) { b[i] = 0.1f * frand(seed); ga[i] = 1.f + 0.1f * frand(seed); be[i] = 0.1f * frand(seed); }
    void *dw, *db, *dg, *dbe;
    CK(hipMalloc(&dw, w.size() * 2)); CK(hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&db, C * 4)); CK(hipMemcpy(db, b.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dg, C * 4)); CK(hipMemcpy(dg, ga.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dbe, C * 4)); CK(hipMemcpy(dbe, be.data(), C * 4, hipMemcpyHostToDevice));
    hl[l] = LayerW{(const bf16_t*)dw, (const float*)db, (const float*)dg, (const float*)dbe};
  }
  LayerW* dl;
  CK(hipMalloc(&dl, sizeof(LayerW) * L)); CK(hipMemcpy(dl, hl.data(), sizeof(LayerW) * L, hipMemcpyHostToDevice));
  std::vector<uint16_t> x0((size_t)HW * C);
  for (auto& x : x0) x = f2bf(frand(seed));
  bf16_t *bufA[2], *bufB[2];
  for (int i = 0; i < 2; ++i) { CK(hipMalloc(&bufA[i], x0.size() * 2)); CK(hipMalloc(&bufB[i], x0.size() * 2)); }
  unsigned* counter; CK(hipMalloc(&counter, (L + 1) * 4));
  unsigned long long* stamps; CK(hipMalloc(&stamps, (size_t)L * 2 * 8));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  // (A) graph of L launches
  hipGraph_t graph; hipGraphExec_t gexec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int l = 0; l < L; ++l) layer_kernel<S><<<NWG, NTHR, SMEM, st>>>(bufA[l & 1], bufA[(l + 1) & 1], hl[l]);
  CK(hipStreamEndCapture(st, &graph));
  CK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
  std::vector<float> tA, tB;
  for (int r = 0; r < reps + 2; ++r) {
    CK(hipMemcpyAsync(bufA[0], x0.data(), x0.size() * 2, hipMemcpyHostToDevice, st));
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(gexec, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) tA.push_back(ms * 1e3f);
  }
  // (B) one persistent launch (NWG <= 64 workgroups on a 256-CU device: all co-resident)
  ChainArgs ca{dl, {bufB[0], bufB[1]}, counter, L, NWG, stamps};
  for (int r = 0; r < reps + 2; ++r) {
    CK(hipMemcpyAsync(bufB[0], x0.data(), x0.size() * 2, hipMemcpyHostToDevice, st));
    CK(hipMemsetAsync(counter, 0, (L + 1) * 4, st));
    CK(hipEventRecord(e0, st)); chain_kernel<S><<<NWG, NTHR, SMEM, st>>>(ca); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 2) tB.push_back(ms * 1e3f);
  }
  std::vector<uint16_t> oa(x0.size()), ob(x0.size());
  CK(hipMemcpy(oa.data(), bufA[L & 1], oa.size() * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(ob.data(), bufB[L & 1], ob.size() * 2, hipMemcpyDeviceToHost));
  const bool same = memcmp(oa.data(), ob.data(), oa.size() * 2) == 0;
  int nonfinite = 0; double asum = 0;
  for (auto v : oa) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); if (!(f == f) || f > 1e30f || f < -1e30f) ++nonfinite; else asum += fabs(f); }
  std::sort(tA.begin(), tA.end()); std::sort(tB.begin(), tB.end());
  std::vector<unsigned long long> hs((size_t)L * 2);
  CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
  printf("map %dx%d, %d layers of conv3x3 256->256 (GroupNorm + SiLU prologue), %d workgroups per layer\n", S, S, L, NWG);
  printf("  (A) %d launches in one HIP graph : median %.1f us = %.2f us per layer (min %.1f)\n", L, tA[tA.size() / 2], tA[tA.size() / 2] / L, tA[0]);
  printf("  (B) one persistent launch        : median %.1f us = %.2f us per layer (min %.1f)\n", tB[tB.size() / 2], tB[tB.size() / 2] / L, tB[0]);
  printf("  outputs bit-identical: %s; non-finite: %d; mean |x_L| %.3f\n", same ? "yes" : "NO", nonfinite, asum / oa.size());
  printf("  persistent, workgroup 0 (100 MHz realtime clock): per layer  wait-for-previous us | body+publish us\n   ");
  for (int l = 1; l < L && l < 12; ++l)
    printf(" %.2f|%.2f", (hs[l * 2 + 1] - hs[l * 2]) / 100.0, (hs[(l + 1 < L ? (l + 1) * 2 : l * 2 + 1)] - hs[l * 2 + 1]) / 100.0);
  printf("\n");
  return same && !nonfinite ? 0 : 1;
}

int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 4, L = argc > 2 ? atoi(argv[2]) : 16, reps = argc > 3 ? atoi(argv[3]) : 20;
  if (L < 1 || L > 64) { fprintf(stderr, "L in 1..64\n"); return 2; }
  if (S == 4) return run<4>(L, reps);
  if (S == 8) return run<8>(L, reps);
  fprintf(stderr, "S must be 4 or 8\n");
  return 2;
}
