// bf_fused16_a128_w8.hip -- the GENERAL kernel on 8-wave workgroups for antenna class a128 (bf_fused16.hpp).  Its own translation
// unit: it is compiled with the iterative-maxocc scheduling strategy (build.py; -4...-6 % against max-ilp, which the others keep).
#include "bf_fused16.hpp"

namespace dsabf {
FusedVariant fused16_variant_a128_w8(int n_ipo, int mode) { return fused16_variant_w8<128, false>(n_ipo, mode); }
}  // namespace dsabf
