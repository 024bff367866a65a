// bf_fused16_k2p4_w8.hip -- the 8-wave-workgroup instantiations of antenna class k2p4 (bf_fused16.hpp); their own
// translation unit so that they compile beside the 4-wave ones.
#include "bf_fused16.hpp"

namespace dsabf {
FusedVariant fused16_variant_k2p4_w8(int n_ipo, int mode, bool paired) { return fused16_variant_w8<kAntK2P4>(n_ipo, mode, paired); }
}  // namespace dsabf
