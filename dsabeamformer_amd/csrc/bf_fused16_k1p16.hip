// bf_fused16_k1p16.hip -- the fused16_kernel instantiations of antenna class kAntK1P16 (bf_fused16.hpp); one class per
// translation unit so that the classes compile in parallel.
#include "bf_fused16.hpp"

namespace dsabf {
FusedVariant fused16_variant_k1p16(int n_ipo, bool write_c, int mode, bool paired)
{
    return fused16_variant<kAntK1P16>(n_ipo, write_c, mode, paired);
}
}  // namespace dsabf
