// bf_fused16_a128.hip -- the fused16_kernel instantiations of antenna class 128 (bf_fused16.hpp); one class per
// translation unit so that the classes compile in parallel.
#include "bf_fused16.hpp"

namespace dsabf {
FusedVariant fused16_variant_a128(int n_ipo, bool write_c, int mode, bool paired)
{
    return fused16_variant<128>(n_ipo, write_c, mode, paired);
}
}  // namespace dsabf
