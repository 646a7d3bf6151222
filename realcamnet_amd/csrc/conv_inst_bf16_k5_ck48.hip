// Instantiation of the MFMA conv kernels: bf16_t, 5x5, one 48-channel Cin chunk, one 16-wide cout tile -- the folded tail
// (conv 3x3 C -> 4C, PixelShuffle(2), conv 3x3 C -> 3 composed into ONE 5x5 convolution C -> 12, rc_tail_fold_weights).
#include "conv_kernel.hpp"
namespace rc {
int conv_bf16_k5_ck48(int nt, const ConvArgs& a, hipStream_t s) {
    if (nt == 1) return launch_conv<ConvCfg<bf16_t, 48, 1, 5>>(a, s);
    return fail(RC_ERR_UNSUPPORTED, "conv: the 5x5 form has one 16-wide cout tile");
}
}  // namespace rc
