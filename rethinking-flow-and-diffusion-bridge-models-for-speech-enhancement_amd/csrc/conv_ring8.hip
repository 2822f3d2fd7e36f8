// conv_ring8.hip - the producer / consumer ring conv kernel (conv_ring_impl.h) on 8 x 16 pixel tiles: the maps that
// offer fewer than one 16 x 16 tile per CU (a translation unit of its own: the template takes ~20 s per instantiation).
#include "conv_ring_impl.h"

// called from fdbm_conv_igemm (conv.hip) once it has validated the arguments and filled ConvParams
int fdbm_launch_conv_ring8(const ConvParams& p, int dt_in, int dt_out, hipStream_t st) {
  const bool gnp = p.gn_sums != nullptr;
  if (dt_in == FDBM_BF16 && dt_out == FDBM_BF16)
    return gnp ? launch_ring<bf16_t, bf16_t, true, 8>(p, st) : launch_ring<bf16_t, bf16_t, false, 8>(p, st);
  if (dt_in == FDBM_F16 && dt_out == FDBM_F16)
    return gnp ? launch_ring<f16_t, f16_t, true, 8>(p, st) : launch_ring<f16_t, f16_t, false, 8>(p, st);
  fdbm_set_error("fdbm_conv_igemm(ring8): unsupported dtypes %d -> %d", dt_in, dt_out);
  return 1;
}
