// bf_fused16_a128_w8p.hip -- the conjugate-pair kernel on 8-wave workgroups for antenna class a128 (bf_fused16.hpp).  Its own
// translation unit: it is compiled with the iterative-ilp scheduling strategy (build.py; -4 % against max-ilp).
#include "bf_fused16.hpp"

namespace dsabf {
FusedVariant fused16_variant_a128_w8p(int n_ipo, int mode) { return fused16_variant_w8<128, true>(n_ipo, mode); }
}  // namespace dsabf
