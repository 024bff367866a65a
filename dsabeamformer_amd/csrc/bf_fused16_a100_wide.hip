// bf_fused16_a100_wide.hip -- the wide launches of antenna class a100 (8-wave workgroups; 8 output slots per wave for the
// conjugate-pair kernel: bf_fused16.hpp); their own translation unit so that they compile beside the others.
#include "bf_fused16.hpp"

namespace dsabf {
FusedVariant fused16_variant_a100_wide(int n_ipo, int mode, bool paired, bool ns8)
{
    return fused16_variant_wide<100>(n_ipo, mode, paired, ns8);
}
}  // namespace dsabf
