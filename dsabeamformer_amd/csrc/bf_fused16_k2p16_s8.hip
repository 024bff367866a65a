// bf_fused16_k2p16_s8.hip -- the conjugate-pair kernel with 8 output slots per wave for antenna class k2p16 (bf_fused16.hpp); its own
// translation unit so that it compiles beside the others.
#include "bf_fused16.hpp"

namespace dsabf {
FusedVariant fused16_variant_k2p16_s8(int n_ipo, int mode) { return fused16_variant_s8<kAntK2P16>(n_ipo, mode); }
}  // namespace dsabf
