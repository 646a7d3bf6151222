// Instantiations of the MFMA conv kernels: bf16_t, 3x3, 32-channel Cin chunks: the multi-chunk producer/consumer form, and (NT = 2) the single-chunk
// persistent form of the 32 -> 32 layers (the ISPUNet family's level 0: three blocks per CU).
#include "conv_kernel.hpp"
namespace rc {
int conv_bf16_k3_ck32(int nt, const ConvArgs& a, hipStream_t s) {
    if (nt == 1) return launch_conv<ConvCfg<bf16_t, 32, 1, 3>>(a, s);
    if (nt == 2) return launch_conv<ConvCfg<bf16_t, 32, 2, 3>>(a, s);
    if (nt == 3) return launch_conv<ConvCfg<bf16_t, 32, 3, 3>>(a, s);
    if (nt == 4) return launch_conv<ConvCfg<bf16_t, 32, 4, 3>>(a, s);
    return fail(RC_ERR_UNSUPPORTED, "conv: no kernel instantiation for this cout tile width");
}
}  // namespace rc
