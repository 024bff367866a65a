// bf_fused16_a64.hip -- the fused16_kernel instantiations of antenna class 64 (bf_fused16.hpp); one class per
// translation unit so that the classes compile in parallel.
#include "bf_fused16.hpp"

namespace dsabf {
FusedVariant fused16_variant_a64(int n_ipo, bool write_c, int mode, bool paired)
{
    return fused16_variant<64>(n_ipo, write_c, mode, paired);
}
}  // namespace dsabf
