// Instantiations of the MFMA conv kernels: bf16_t, 1x1, 48-channel Cin chunks (one file per chunk width so they build in parallel).
#include "conv_kernel.hpp"
namespace rc {
int conv_bf16_k1_ck48(int nt, const ConvArgs& a, hipStream_t s) {
    if (nt == 1) return launch_conv<ConvCfg<bf16_t, 48, 1, 1>>(a, s);
    if (nt == 3) return launch_conv<ConvCfg<bf16_t, 48, 3, 1>>(a, s);
    if (nt == 4) return launch_conv<ConvCfg<bf16_t, 48, 4, 1>>(a, s);
    if (nt == 5) return launch_conv<ConvCfg<bf16_t, 48, 5, 1>>(a, s);
    return fail(RC_ERR_UNSUPPORTED, "conv: no kernel instantiation for this cout tile width");
}
}  // namespace rc
