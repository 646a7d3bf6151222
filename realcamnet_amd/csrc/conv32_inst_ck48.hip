// Instantiations of the 32x32x16 conv for one 48-channel Cin chunk and 96-wide cout tiles (48 -> 192: the tail's PixelShuffle conv, up2's last conv).
#include "conv32_kernel.hpp"
namespace rc {
int conv32_ck48(int variant, const ConvArgs& a, hipStream_t s) {
    return launch_conv32<C32Cfg<48, 8, 4, 3>>(a, s);                        // 8 x 32 pixel tile, 4 compute waves x (64 px x 96 couts)
}
}  // namespace rc
