// Instantiations of the MFMA conv kernel: bf16_t, 1x1 (Linear layers of the GroupMix block, lens-shading MLP).
#include "conv_kernel.hpp"
namespace rc {
int dispatch_conv_bf16_k1(int ck, int nt, const ConvArgs& a, hipStream_t s) {
#define RC_CASE(CK, NT) if (ck == CK && nt == NT) return launch_conv<ConvCfg<bf16_t, CK, NT, 1>>(a, s);
    RC_CASE(8, 1)
    RC_CASE(8, 3)
    RC_CASE(8, 4)
    RC_CASE(8, 5)
    RC_CASE(16, 1)
    RC_CASE(16, 3)
    RC_CASE(16, 4)
    RC_CASE(16, 5)
    RC_CASE(48, 1)
    RC_CASE(48, 3)
    RC_CASE(48, 4)
    RC_CASE(48, 5)
    RC_CASE(64, 1)
    RC_CASE(64, 3)
    RC_CASE(64, 4)
    RC_CASE(64, 5)
    RC_CASE(80, 1)
    RC_CASE(80, 3)
    RC_CASE(80, 4)
    RC_CASE(80, 5)
#undef RC_CASE
    return fail(RC_ERR_UNSUPPORTED, "conv: no kernel instantiation for this (ck, nt)");
}
}  // namespace rc
