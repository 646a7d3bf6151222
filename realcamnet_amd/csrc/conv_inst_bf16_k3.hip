// Instantiations of the MFMA conv kernel: bf16_t, 3x3.
#include "conv_kernel.hpp"
namespace rc {
int dispatch_conv_bf16_k3(int ck, int nt, const ConvArgs& a, hipStream_t s) {
#define RC_CASE(CK, NT) if (ck == CK && nt == NT) return launch_conv<ConvCfg<bf16_t, CK, NT, 3>>(a, s);
    RC_CASE(8, 1)
    RC_CASE(8, 3)
    RC_CASE(8, 4)
    RC_CASE(16, 1)
    RC_CASE(16, 3)
    RC_CASE(16, 4)
    RC_CASE(48, 1)
    RC_CASE(48, 3)
    RC_CASE(48, 4)
    RC_CASE(64, 1)
    RC_CASE(64, 3)
    RC_CASE(64, 4)
#undef RC_CASE
    return fail(RC_ERR_UNSUPPORTED, "conv: no kernel instantiation for this (ck, nt)");
}
}  // namespace rc
