// bf_fused16_a128_w8.hip -- the 8-wave-workgroup instantiations of antenna class a128 (bf_fused16.hpp).  Their own translation
// unit: they are compiled with the iterative-ilp scheduling strategy (build.py; -2...-3.6 % against max-ilp, which the others keep).
#include "bf_fused16.hpp"

namespace dsabf {
FusedVariant fused16_variant_a128_w8(int n_ipo, int mode, bool paired) { return fused16_variant_w8<128>(n_ipo, mode, paired); }
}  // namespace dsabf
