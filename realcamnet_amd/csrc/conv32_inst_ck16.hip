// Instantiations of the staged-output 32x32x16 multi-chunk conv (conv32s: 16-channel Cin chunks, 64-wide cout tiles, 16 x 32 pixel tile,
// 4 compute waves x (128 px x 64 couts), the tile's output handed to the loader waves through LDS).
#include "conv32_kernel.hpp"
namespace rc {
int conv32_ck16(int variant, const ConvArgs& a, hipStream_t s) {
    return launch_conv32s<C32SCfg<4>>(a, s);
}
}  // namespace rc
