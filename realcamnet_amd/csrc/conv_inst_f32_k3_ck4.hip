// Instantiations of the MFMA conv kernels: float, 3x3, 4-channel Cin chunks (one file per chunk width so they build in parallel).
#include "conv_kernel.hpp"
namespace rc {
int conv_f32_k3_ck4(int nt, const ConvArgs& a, hipStream_t s) {
    if (nt == 1) return launch_conv<ConvCfg<float, 4, 1, 3>>(a, s);
    if (nt == 3) return launch_conv<ConvCfg<float, 4, 3, 3>>(a, s);
    if (nt == 4) return launch_conv<ConvCfg<float, 4, 4, 3>>(a, s);
    return fail(RC_ERR_UNSUPPORTED, "conv: no kernel instantiation for this cout tile width");
}
}  // namespace rc
