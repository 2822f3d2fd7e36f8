// conv_head.hip - 3x3 convolution with at most 16 output channels and f32 output: the 4-channel pyramid heads
// (GroupNorm -> SiLU -> Conv3x3(C -> 4), ncsnpp_v2.py:372-389) on maps of at least one 16 x 16 tile per CU.
//
// With 4 live output channels the 128-channel tile kernels spend a full conv's MFMAs (and the 16-channel halo-patch
// variant a weight tile's load -> LDS -> barrier chain per k-step) on 3 % useful work.  Here:
//   * ALL the layer's weights for the 16-channel block (nk x 16 rows x 128 bytes: 36 KiB at C = 128) are staged into LDS
//     ONCE per workgroup, which then WALKS tiles of its image - no weight traffic, no barrier inside a chunk;
//   * per 64-channel chunk the (16+2) x 18 halo patch is staged once (GroupNorm scale/shift + SiLU applied in
//     registers), the next chunk's loads requested before the current chunk's 9 taps are multiplied;
//   * 256 threads: wave w owns image rows 4w .. 4w+3 of the tile (4 MFMA tiles of 16 px x 16 channels).
// LDS = weights + one patch buffer + GroupNorm table = 78.5 KiB at C = 128: two workgroups share a CU, so one's patch
// arithmetic runs beside the other's loads.  Bound: the patch transform (VALU) and the input read (HBM), not the MFMAs.
#include <stdlib.h>

#include "conv_common.h"

namespace {
constexpr int HPC = 18, HPROWS = 18 * 18, HPB = 328 * 128;   // patch columns / pixels / bytes (rows padded to 8)
constexpr int HNPL = (HPROWS * 8 + 255) / 256;                // 16-byte patch items per thread (11)
}  // namespace

template <typename T, bool GNP>
__global__ void __launch_bounds__(256, 2) conv_head_kernel(const ConvParams p, int tiles_x, int tiles_y, int wgs_per_image) {
  constexpr int KC = 64, VW = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nk = p.nk;
  unsigned char* s_w = smem;                                    // [nk][16 rows][128 B], swizzled
  unsigned char* s_patch = smem + nk * 2048;                    // [HPROWS][128 B], swizzled
  const int gnpad = GNP ? ((p.gn_C + 63) & ~63) : 0;
  float* s_gn = reinterpret_cast<float*>(s_patch + HPB);        // scale[gnpad] | shift[gnpad]
  float* s_mr = s_gn + 2 * gnpad;                               // [32][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fk = lane >> 4;
  const int H = p.H, W = p.W;
  const int tpi = tiles_x * tiles_y;
  const int b = blockIdx.x / wgs_per_image;
  const int wslot = blockIdx.x - b * wgs_per_image;
  const int64_t img = (int64_t)b * H * W;
  const int C = p.seg[0].C, cin = p.seg[0].cin, coff = p.seg[0].coff;
  const int nch = (cin + KC - 1) / KC;
  const bool reg = GNP && p.seg_gn[0] >= 0;

  // ---- weights of every k-step, once: item i = (k-step i / 128, row (i / 8) % 16, chunk i % 8)
  for (int i = tid; i < nk * 128; i += 256) {
    const int ks = i >> 7, row = (i >> 3) & 15, ch = i & 7;
    const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p.w) + ((int64_t)ks * p.CoutPad + row) * 128 + ch * 16);
    *reinterpret_cast<uint4*>(s_w + ks * 2048 + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = v;
  }
  if constexpr (GNP) conv_gn_table<256>(p, b, 1, s_gn, gnpad, s_mr, s_patch);
  else __syncthreads();

  // ---- per-thread patch items: row = (tid / 8 + 32 j), chunk tid % 8
  const int pchunk = tid & 7;
  int plds[HNPL];
#pragma unroll
  for (int j = 0; j < HNPL; ++j) {
    const int row = (tid >> 3) + 32 * j;
    const int rr = min(row, HPROWS - 1);
    const int pc = rr % HPC;
    plds[j] = row < HPROWS ? row * 128 + ((pchunk ^ ((pc >> 1) & 7)) << 4) : -1;
  }
  uint4 preg[HNPL];
  unsigned pmask = 0;
  bool pcok = true;
  auto load_patch = [&](int ti, int c) __attribute__((always_inline)) {
    const int ty = ti / tiles_x, tx = ti - ty * tiles_x;
    const int cvalid = min(KC, cin - c * KC);
    pcok = pchunk * VW < cvalid;
    const T* src = reinterpret_cast<const T*>(p.seg[0].src) + img * C + coff + c * KC + (pcok ? pchunk : 0) * VW;
    pmask = 0;
#pragma unroll
    for (int j = 0; j < HNPL; ++j) {
      const int row = (tid >> 3) + 32 * j;
      const int rr = min(row, HPROWS - 1);
      const int pr = rr / HPC, pc = rr - pr * HPC;
      const int iy = ty * 16 + pr - 1, ix = tx * 16 + pc - 1;
      const bool ok = row < HPROWS && iy >= 0 && iy < H && ix >= 0 && ix < W;
      pmask |= ok ? (1u << j) : 0u;
      preg[j] = *reinterpret_cast<const uint4*>(src + (int64_t)(ok ? iy * W + ix : 0) * C);
    }
  };
  auto write_patch = [&](int c) __attribute__((always_inline)) {
    float sc[8], sh[8];
    if constexpr (GNP) {
      if (reg) {
        const int gcb = p.seg_gn[0] + c * KC + (pcok ? pchunk : 0) * VW;
#pragma unroll
        for (int q = 0; q < 8; ++q) { sc[q] = s_gn[gcb + q]; sh[q] = s_gn[gnpad + gcb + q]; }
      }
    }
#pragma unroll
    for (int j = 0; j < HNPL; ++j) {
      uint4 v = preg[j];
      if constexpr (GNP) { if (reg) v = gn_transform16<T>(v, sc, sh, p.gn_silu != 0); }
      if (!(((pmask >> j) & 1u) && pcok)) v = uint4{0u, 0u, 0u, 0u};          // padding AFTER the activation
      if (plds[j] >= 0) *reinterpret_cast<uint4*>(s_patch + plds[j]) = v;
    }
  };

  // fragment bases (as in conv_patch.hip): patch row (4 wave + i + dy), column frow + dx; weight row frow
#define A_BASE(DX, KK) ((wave * 4 * HPC + frow + (DX)) * 128 + ((((KK) * 4 + fk) ^ (((frow + (DX)) >> 1) & 7)) << 4))
  const int ab00 = A_BASE(0, 0), ab01 = A_BASE(0, 1), ab10 = A_BASE(1, 0), ab11 = A_BASE(1, 1), ab20 = A_BASE(2, 0), ab21 = A_BASE(2, 1);
#undef A_BASE
  const int wb0 = frow * 128 + (((0 * 4 + fk) ^ ((frow >> 1) & 7)) << 4);
  const int wb1 = frow * 128 + (((1 * 4 + fk) ^ ((frow >> 1) & 7)) << 4);

  if (wslot < tpi) load_patch(wslot, 0);
  for (int ti = wslot; ti < tpi; ti += wgs_per_image) {
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nch; ++c) {
      __syncthreads();                       // every wave is done reading the patch of the chunk before
      write_patch(c);
      __syncthreads();
      // the next patch (next chunk, or the next tile's first chunk) is requested before this chunk is multiplied
      if (c + 1 < nch) load_patch(ti, c + 1);
      else if (ti + wgs_per_image < tpi) load_patch(ti + wgs_per_image, 0);
      // (one filter row at a time: fully unrolled, hipcc hoisted all 72 fragment reads of a chunk in front of its MFMAs - 256
      // registers and 39-51 spills, the patch transform's operands among them)
#pragma unroll 1
      for (int k3 = 0; k3 < 3; ++k3) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int k = k3 * 3 + kx;
        const int dy = k3, dx = kx;
        const int ks = k * nch + c;          // packed k-step order of a 9-tap segment: tap-major, chunk-minor
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const uint4 wf = *reinterpret_cast<const uint4*>(s_w + ks * 2048 + (kk == 0 ? wb0 : wb1));
          const unsigned char* asrc = s_patch + dy * (HPC * 128) +
                                      (kk == 0 ? (dx == 0 ? ab00 : dx == 1 ? ab10 : ab20) : (dx == 0 ? ab01 : dx == 1 ? ab11 : ab21));
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint4 af = *reinterpret_cast<const uint4*>(asrc + i * (HPC * 128));
            Mfma<T>::run(wf, af, acc[i]);
          }
        }
      }
      }
    }
    // epilogue: lane (frow, fk) holds channels 4 fk .. 4 fk + 3 of pixel (row 4 wave + i, column frow)
    const int y0 = (ti / tiles_x) * 16, x0 = (ti % tiles_x) * 16;
    const int n = fk * 4;
    if (n < p.Cout) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t m = img + (int64_t)(y0 + wave * 4 + i) * W + x0 + frow;
        float v[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        conv_epilogue4<float>(p, m, b, n, v);
      }
    }
  }
}

template <typename T, bool GNP>
static int launch_head(const ConvParams& p, hipStream_t st) {
  const int SMEM = p.nk * 2048 + HPB + (GNP ? ((p.gn_C + 63) & ~63) * 8 : 0) + 64 * 4;
  static int attr_set = 0;
  if (attr_set < SMEM) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_head_kernel<T, GNP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      fdbm_set_error("fdbm_conv_igemm(head): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 2;
    }
    attr_set = 160 * 1024;
  }
  const int tiles_x = p.W / 16, tiles_y = p.H / 16, tpi = tiles_x * tiles_y;
  // workgroups per image: two per CU over the batch (LDS permitting), each walks tpi / wpi tiles
  int wpi = (512 + p.B - 1) / p.B;
  if (wpi > tpi) wpi = tpi;
  if (wpi < 1) wpi = 1;
  conv_head_kernel<T, GNP><<<dim3((unsigned)(p.B * wpi)), 256, SMEM, st>>>(p, tiles_x, tiles_y, wpi);
  FDBM_LAUNCH_CHECK("fdbm_conv_igemm(head)");
  return 0;
}

// Can this conv run on the head kernel?  (shape / segment layout; the caller checks dtypes: 16-bit input, f32 output)
bool fdbm_conv_head_ok(const ConvParams& p) {
  if (p.Cout > 16 || (p.Cout & 3) || p.H % 16 || p.W % 16 || p.nseg != 1 || p.seg[0].taps != 9) return false;
  if (p.stat_out || p.comb_pyr || p.tbias) return false;
  const int nch = (p.seg[0].cin + 63) / 64;
  if (p.nk != 9 * nch) return false;
  const int smem = p.nk * 2048 + HPB + ((p.gn_C + 63) & ~63) * 8 + 256;
  return smem <= 150 * 1024;
}

int fdbm_launch_conv_head(const ConvParams& p, int dt_in, hipStream_t st) {
  const bool gnp = p.gn_sums != nullptr || p.gn_unit;
  if (dt_in == FDBM_BF16) return gnp ? launch_head<bf16_t, true>(p, st) : launch_head<bf16_t, false>(p, st);
  if (dt_in == FDBM_F16) return gnp ? launch_head<f16_t, true>(p, st) : launch_head<f16_t, false>(p, st);
  fdbm_set_error("fdbm_conv_igemm(head): unsupported input dtype %d", dt_in);
  return 1;
}
