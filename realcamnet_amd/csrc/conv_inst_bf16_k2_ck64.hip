// Instantiations of the MFMA conv kernels: bf16_t, 2x2 window at offsets {-1, 0}^2 (the non-zero taps of a stride-2 3x3 convolution
// over its space-to-depth map, see rc_conv2d), 64-channel Cin chunks.
#include "conv_kernel.hpp"
namespace rc {
int conv_bf16_k2_ck64(int nt, const ConvArgs& a, hipStream_t s) {
    if (nt == 1) return launch_conv<ConvCfg<bf16_t, 64, 1, 2>>(a, s);
    if (nt == 3) return launch_conv<ConvCfg<bf16_t, 64, 3, 2>>(a, s);
    if (nt == 4) return launch_conv<ConvCfg<bf16_t, 64, 4, 2>>(a, s);
    return fail(RC_ERR_UNSUPPORTED, "conv: no kernel instantiation for this cout tile width");
}
}  // namespace rc
