// bf_fused16_k2p4_wide.hip -- the wide launches of antenna class k2p4 (8-wave workgroups; 8 output slots per wave for the
// conjugate-pair kernel: bf_fused16.hpp); their own translation unit so that they compile beside the others.
#include "bf_fused16.hpp"

namespace dsabf {
FusedVariant fused16_variant_k2p4_wide(int n_ipo, int mode, bool paired, bool ns8)
{
    return fused16_variant_wide<kAntK2P4>(n_ipo, mode, paired, ns8);
}
}  // namespace dsabf
