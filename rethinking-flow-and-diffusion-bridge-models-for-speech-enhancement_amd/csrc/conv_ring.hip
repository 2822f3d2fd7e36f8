// conv_ring.hip - the producer / consumer ring conv kernel (conv_ring_impl.h) on 16 x 16 pixel tiles, and the shape test
// both tile heights share.
#include "conv_ring_impl.h"

// Can this conv run on the ring kernel with tiles of `rows` (16 or 8) image rows?  (shape / segment layout only; the
// caller checks the dtype)
bool fdbm_conv_ring_ok(const ConvParams& p, int rows) {
  if (p.H % rows || p.W % 16 || p.nseg < 1 || p.seg[0].taps != 9) return false;
  // the one epilogue of the kernel: 16-byte stores of 8 channels, no pyramid / Combine extras, 32-bit offsets per image
  if ((p.Cout & 7) || p.res_lo || p.comb_pyr || (int64_t)p.H * p.W * p.Cout * 2 >= (int64_t)1 << 31) return false;
  // the producers stream whole 128-row weight tiles: the packed weights must hold them (CoutPad a multiple of 128)
  if (p.CoutPad % 128) return false;
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
    return gnp ? launch_ring<bf16_t, bf16_t, true, 16>(p, st) : launch_ring<bf16_t, bf16_t, false, 16>(p, st);
  if (dt_in == FDBM_F16 && dt_out == FDBM_F16)
    return gnp ? launch_ring<f16_t, f16_t, true, 16>(p, st) : launch_ring<f16_t, f16_t, false, 16>(p, st);
  fdbm_set_error("fdbm_conv_igemm(ring): unsupported dtypes %d -> %d", dt_in, dt_out);
  return 1;
}
