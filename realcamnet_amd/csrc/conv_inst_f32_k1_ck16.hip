// Instantiations of the MFMA conv kernels: float, 1x1, 16-channel Cin chunks (one file per chunk width so they build in parallel).
#include "conv_kernel.hpp"
namespace rc {
int conv_f32_k1_ck16(int nt, const ConvArgs& a, hipStream_t s) {
    if (nt == 1) return launch_conv<ConvCfg<float, 16, 1, 1>>(a, s);
    if (nt == 3) return launch_conv<ConvCfg<float, 16, 3, 1>>(a, s);
    if (nt == 4) return launch_conv<ConvCfg<float, 16, 4, 1>>(a, s);
    if (nt == 5) return launch_conv<ConvCfg<float, 16, 5, 1>>(a, s);
    return fail(RC_ERR_UNSUPPORTED, "conv: no kernel instantiation for this cout tile width");
}
}  // namespace rc
