// Instantiations of the MFMA conv kernel: float, 1x1 (Linear layers of the GroupMix block, lens-shading MLP).
#include "conv_kernel.hpp"
namespace rc {
int dispatch_conv_f32_k1(int ck, int nt, const ConvArgs& a, hipStream_t s) {
#define RC_CASE(CK, NT) if (ck == CK && nt == NT) return launch_conv<ConvCfg<float, CK, NT, 1>>(a, s);
    RC_CASE(4, 1)
    RC_CASE(4, 3)
    RC_CASE(4, 4)
    RC_CASE(4, 5)
    RC_CASE(16, 1)
    RC_CASE(16, 3)
    RC_CASE(16, 4)
    RC_CASE(16, 5)
#undef RC_CASE
    return fail(RC_ERR_UNSUPPORTED, "conv: no kernel instantiation for this (ck, nt)");
}
}  // namespace rc
