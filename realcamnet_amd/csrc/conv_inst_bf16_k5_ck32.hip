// Instantiation of the MFMA conv kernels: bf16_t, 5x5, 32-channel Cin chunks, one 16-wide cout tile -- the folded tail of the 32- / 64-channel nets
// (ISPUNet family, LiteISPNet; rc_tail_fold_weights).
#include "conv_kernel.hpp"
namespace rc {
int conv_bf16_k5_ck32(int nt, const ConvArgs& a, hipStream_t s) {
    if (nt == 1) return launch_conv<ConvCfg<bf16_t, 32, 1, 5>>(a, s);
    return fail(RC_ERR_UNSUPPORTED, "conv: the 5x5 form has one 16-wide cout tile");
}
}  // namespace rc
