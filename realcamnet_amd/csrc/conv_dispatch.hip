// Conv kernel dispatch: (dtype, ksize, Cin chunk width) -> the translation unit that instantiates it.
#include "conv_kernel.hpp"
namespace rc {
int conv_bf16_k3_ck8(int nt, const ConvArgs& a, hipStream_t s);
int conv_bf16_k3_ck16(int nt, const ConvArgs& a, hipStream_t s);
int conv_bf16_k3_ck32(int nt, const ConvArgs& a, hipStream_t s);
int conv_bf16_k3_ck48(int nt, const ConvArgs& a, hipStream_t s);
int conv_bf16_k3_ck64(int nt, const ConvArgs& a, hipStream_t s);
int conv_bf16_k5_ck48(int nt, const ConvArgs& a, hipStream_t s);
int conv_bf16_k5_ck32(int nt, const ConvArgs& a, hipStream_t s);
int conv_f32_k5_ck16(int nt, const ConvArgs& a, hipStream_t s);
int conv_bf16_k2_ck16(int nt, const ConvArgs& a, hipStream_t s);
int conv_bf16_k2_ck32(int nt, const ConvArgs& a, hipStream_t s);
int conv_bf16_k2_ck64(int nt, const ConvArgs& a, hipStream_t s);
int conv_bf16_k1_ck8(int nt, const ConvArgs& a, hipStream_t s);
int conv_bf16_k1_ck16(int nt, const ConvArgs& a, hipStream_t s);
int conv_bf16_k1_ck48(int nt, const ConvArgs& a, hipStream_t s);
int conv_bf16_k1_ck64(int nt, const ConvArgs& a, hipStream_t s);
int conv_bf16_k1_ck80(int nt, const ConvArgs& a, hipStream_t s);
int conv_f32_k3_ck4(int nt, const ConvArgs& a, hipStream_t s);
int conv_f32_k3_ck16(int nt, const ConvArgs& a, hipStream_t s);
int conv_f32_k1_ck4(int nt, const ConvArgs& a, hipStream_t s);
int conv_f32_k1_ck16(int nt, const ConvArgs& a, hipStream_t s);

int dispatch_conv(bool bf16, int ksize, int ck, int nt, const ConvArgs& a, hipStream_t s) {
    if (bf16 && ksize == 3 && ck == 8) return conv_bf16_k3_ck8(nt, a, s);
    if (bf16 && ksize == 3 && ck == 16) return conv_bf16_k3_ck16(nt, a, s);
    if (bf16 && ksize == 3 && ck == 32) return conv_bf16_k3_ck32(nt, a, s);
    if (bf16 && ksize == 3 && ck == 48) return conv_bf16_k3_ck48(nt, a, s);
    if (bf16 && ksize == 3 && ck == 64) return conv_bf16_k3_ck64(nt, a, s);
    if (bf16 && ksize == 5 && ck == 48) return conv_bf16_k5_ck48(nt, a, s);
    if (bf16 && ksize == 5 && ck == 32) return conv_bf16_k5_ck32(nt, a, s);
    if (!bf16 && ksize == 5 && ck == 16) return conv_f32_k5_ck16(nt, a, s);
    if (bf16 && ksize == 2 && ck == 16) return conv_bf16_k2_ck16(nt, a, s);
    if (bf16 && ksize == 2 && ck == 32) return conv_bf16_k2_ck32(nt, a, s);
    if (bf16 && ksize == 2 && ck == 64) return conv_bf16_k2_ck64(nt, a, s);
    if (bf16 && ksize == 1 && ck == 8) return conv_bf16_k1_ck8(nt, a, s);
    if (bf16 && ksize == 1 && ck == 16) return conv_bf16_k1_ck16(nt, a, s);
    if (bf16 && ksize == 1 && ck == 48) return conv_bf16_k1_ck48(nt, a, s);
    if (bf16 && ksize == 1 && ck == 64) return conv_bf16_k1_ck64(nt, a, s);
    if (bf16 && ksize == 1 && ck == 80) return conv_bf16_k1_ck80(nt, a, s);
    if (!bf16 && ksize == 3 && ck == 4) return conv_f32_k3_ck4(nt, a, s);
    if (!bf16 && ksize == 3 && ck == 16) return conv_f32_k3_ck16(nt, a, s);
    if (!bf16 && ksize == 1 && ck == 4) return conv_f32_k1_ck4(nt, a, s);
    if (!bf16 && ksize == 1 && ck == 16) return conv_f32_k1_ck16(nt, a, s);
    return fail(RC_ERR_UNSUPPORTED, "conv: no kernel instantiation for this (dtype, ksize, chunk width)");
}
}  // namespace rc
