// bf_fused16_k4p16.hip -- the deep (three / four k-step) instantiations of fused16_kernel for antenna class kAntK4P16 (bf_fused16.hpp):
// 8-wave workgroups, general kernel with two output slots per wave, conjugate-pair kernel with four (beams in groups of 512) or two.
#include "bf_fused16.hpp"

namespace dsabf {
FusedVariant fused16_variant_k4p16(int n_ipo, int mode, bool paired, int ns) { return fused16_variant_deep<kAntK4P16>(n_ipo, mode, paired, ns); }
}  // namespace dsabf
