// Instantiation of the MFMA conv kernels: float, 5x5, 16-channel Cin chunks, one 16-wide cout tile -- the folded tail in fp32 (cfg2: LiteISPNet at 1080p).
#include "conv_kernel.hpp"
namespace rc {
int conv_f32_k5_ck16(int nt, const ConvArgs& a, hipStream_t s) {
    if (nt == 1) return launch_conv<ConvCfg<float, 16, 1, 5>>(a, s);
    return fail(RC_ERR_UNSUPPORTED, "conv: the 5x5 form has one 16-wide cout tile");
}
}  // namespace rc
