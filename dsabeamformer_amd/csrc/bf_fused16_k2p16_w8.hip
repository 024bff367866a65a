// bf_fused16_k2p16_w8.hip -- the 8-wave-workgroup instantiations of antenna class k2p16 (bf_fused16.hpp).  Their own translation
// unit: they are compiled with the iterative-ilp scheduling strategy (build.py; -2...-3.6 % against max-ilp, which the others keep).
#include "bf_fused16.hpp"

namespace dsabf {
FusedVariant fused16_variant_k2p16_w8(int n_ipo, int mode, bool paired) { return fused16_variant_w8<kAntK2P16>(n_ipo, mode, paired); }
}  // namespace dsabf
