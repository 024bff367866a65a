// bf_fused16_k3p4.hip -- the deep (three / four k-step) instantiations of fused16_kernel for antenna class kAntK3P4 (bf_fused16.hpp):
// antenna counts that are multiples of 4 but not of 16 (132, 140, ... : rows only dword-aligned, staged in 4-byte pieces).
#include "bf_fused16.hpp"

namespace dsabf {
FusedVariant fused16_variant_k3p4(int n_ipo, int mode, bool paired, int ns) { return fused16_variant_deep<kAntK3P4>(n_ipo, mode, paired, ns); }
}  // namespace dsabf
