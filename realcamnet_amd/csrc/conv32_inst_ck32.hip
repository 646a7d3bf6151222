// Instantiations of the 32x32x16 multi-chunk conv: 32-channel Cin chunks, 64-wide cout tiles, 16 x 32 pixel tile.
#include "conv32_kernel.hpp"
namespace rc {
int conv32_ck32(int variant, const ConvArgs& a, hipStream_t s) {
    if (variant == 1) return launch_conv32<C32Cfg<32, 16, 8, 2>>(a, s);    // 8 compute waves x (64 px x 64 couts)
    return launch_conv32<C32Cfg<32, 16, 4, 2>>(a, s);                       // 4 compute waves x (128 px x 64 couts)
}
}  // namespace rc
