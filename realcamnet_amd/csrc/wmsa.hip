// Window / shifted-window multi-head self-attention core (models/tcm.py:179-206):
//   for every ws x ws window of the (cyclically shifted) NHWC map and every head:
//       out = softmax(q k^T * hd^-0.5 + relpos[h, dy, dx]  (+ -inf across the wrap in the last window row / column)) v
// The two nn.Linear layers around it (embedding_layer C -> 3C, linear C -> C) run through rc_conv2d as 1x1 convolutions
// over the un-shifted map (a point-wise op commutes with torch.roll), so the shift is pure index arithmetic here:
// window token (w1, w2, p1, p2) is pixel ((w1*ws + p1 + s) mod H, (w2*ws + p2 + s) mod W), and the result goes back to
// that same pixel (the reference rolls the output back by +s).
//
// Mapping: one lane = one query token; a wave covers 64 / ws^2 windows of ONE head (1 window at ws = 8, 4 at ws = 4).
// K and V of the wave's windows sit in LDS as fp32 ([token][hd + 1], conflict-free column reads), the head's
// (2ws-1)^2 bias table beside them; every lane computes its ws^2 scores (hd FMAs each), the max-subtracted softmax
// exactly as torch does, then the weighted sum of V.  The work is tiny next to the block's Linear layers
// (2 * ws^2 * C FMAs per token vs 12 * C^2 MACs), so plain FMAs are fine.
#include "common.hpp"

namespace rc {

constexpr int kWmsaWaves = 4;
template <typename T, int HD, int WS>
__global__ __launch_bounds__(kWmsaWaves * 64) void wmsa_kernel(const T* __restrict__ qkv, const float* __restrict__ relpos, T* __restrict__ out,
                                                               int batch, int H, int W, int C, int shift) {
    constexpr int TPW = WS * WS;                    // tokens per window
    constexpr int WPW = 64 / TPW;                   // windows per wave
    constexpr int RP = 2 * WS - 1;
    constexpr int KS = HD + 1;                      // LDS row stride (floats)
    extern __shared__ float wm_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* s_k = wm_smem + wave * (2 * 64 * KS + RP * RP);
    float* s_v = s_k + 64 * KS;
    float* s_bias = s_v + 64 * KS;

    const int nh = C / HD, hw = H / WS, ww = W / WS;
    const long long wgroups = ((long long)batch * hw * ww + WPW - 1) / WPW;     // groups of WPW windows
    const long long job = (long long)blockIdx.x * kWmsaWaves + wave;            // job = (window group, head)
    if (job >= wgroups * nh) return;
    const int h = (int)(job % nh);
    const long long wg = job / nh;
    const int wl = lane / TPW, p = lane % TPW;                                  // window inside the wave, token inside it
    const long long win = wg * WPW + wl;
    const bool live = win < (long long)batch * hw * ww;
    const int b = live ? (int)(win / (hw * ww)) : 0;
    const int w1 = live ? (int)((win / ww) % hw) : 0, w2 = live ? (int)(win % ww) : 0;
    const int p1 = p / WS, p2 = p % WS;
    int y = w1 * WS + p1 + shift, x = w2 * WS + p2 + shift;                    // un-shifted pixel of this token
    if (y >= H) y -= H;
    if (x >= W) x -= W;
    const T* px = qkv + (((size_t)b * H + y) * W + x) * 3 * C + h * HD;

    for (int i = lane; i < RP * RP; i += 64) s_bias[i] = relpos[(size_t)h * RP * RP + i];
    float q[HD];
    const float scale = rsqrtf((float)HD);
#pragma unroll
    for (int c = 0; c < HD; ++c) {
        q[c] = live ? to_f32(px[c]) * scale : 0.f;
        s_k[lane * KS + c] = live ? to_f32(px[C + c]) : 0.f;
        s_v[lane * KS + c] = live ? to_f32(px[2 * C + c]) : 0.f;
    }
    __builtin_amdgcn_wave_barrier();            // one wave's LDS operations complete in order

    // after the cyclic shift the last window row / column holds pixels from both borders: a query sees its own side only
    const bool last_r = shift > 0 && w1 == hw - 1, last_c = shift > 0 && w2 == ww - 1;
    const bool side_y = p1 >= WS - shift, side_x = p2 >= WS - shift;
    float sim[TPW];
    float mx = -__builtin_inff();
    const float* kk = s_k + wl * TPW * KS;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) s = __builtin_fmaf(q[c], kk[j * KS + c], s);
        const int j1 = j / WS, j2 = j % WS;
        s += s_bias[(p1 - j1 + WS - 1) * RP + (p2 - j2 + WS - 1)];
        const bool masked = (last_r && ((j1 >= WS - shift) != side_y)) || (last_c && ((j2 >= WS - shift) != side_x));
        s = masked ? -__builtin_inff() : s;
        sim[j] = s;
        mx = fmaxf(mx, s);
    }
    float den = 0.f, acc[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = 0.f;
    const float* vv = s_v + wl * TPW * KS;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const float e = expf(sim[j] - mx);
        den += e;
#pragma unroll
        for (int c = 0; c < HD; ++c) acc[c] = __builtin_fmaf(e, vv[j * KS + c], acc[c]);
    }
    if (!live) return;
    const float inv = 1.f / den;
    T* o = out + (((size_t)b * H + y) * W + x) * C + h * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = from_f32<T>(acc[c] * inv);
}

}  // namespace rc

using namespace rc;

extern "C" {

int rc_window_attention(const void* d_qkv, const float* d_relpos, void* d_out, int dtype, int batch, int H, int W, int C,
                        int head_dim, int window, int shift, void* stream) {
    RC_REQUIRE(d_qkv && d_relpos && d_out, "rc_window_attention: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_window_attention: bad dtype");
    RC_REQUIRE(window == 4 || window == 8, "rc_window_attention: window size must be 4 or 8 (as models/tcm.py uses)");
    RC_REQUIRE(head_dim == 8 || head_dim == 16 || head_dim == 32, "rc_window_attention: head_dim must be 8, 16 or 32");
    RC_REQUIRE(batch >= 1 && H >= window && W >= window && H % window == 0 && W % window == 0, "rc_window_attention: H, W must be multiples of the window size");
    RC_REQUIRE(C >= head_dim && C % head_dim == 0, "rc_window_attention: C must be a multiple of head_dim");
    RC_REQUIRE(shift == 0 || shift == window / 2, "rc_window_attention: shift must be 0 (W-MSA) or window/2 (SW-MSA)");
    const int nh = C / head_dim, wpw = 64 / (window * window);
    const long long jobs = (((long long)batch * (H / window) * (W / window) + wpw - 1) / wpw) * nh;
    const long long blocks = (jobs + kWmsaWaves - 1) / kWmsaWaves;
    RC_REQUIRE(blocks < (1LL << 31), "rc_window_attention: too many windows");
    const size_t lds = (size_t)kWmsaWaves * (2 * 64 * (head_dim + 1) + (2 * window - 1) * (2 * window - 1)) * sizeof(float);
#define RC_WM(TT, HD, WS)                                                                                                 \
    do {                                                                                                                  \
        static PerDeviceFlag attr;                                                                                         \
        if (!attr.test_and_set()) {                                                                                                      \
            RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wmsa_kernel<TT, HD, WS>),                     \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));                     \
        }                                                                                                                 \
        hipLaunchKernelGGL((wmsa_kernel<TT, HD, WS>), dim3((unsigned)blocks), dim3(kWmsaWaves * 64), lds, as_stream(stream), \
                           static_cast<const TT*>(d_qkv), d_relpos, static_cast<TT*>(d_out), batch, H, W, C, shift);     \
    } while (0)
#define RC_WM_HD(TT, WS) do { if (head_dim == 8) RC_WM(TT, 8, WS); else if (head_dim == 16) RC_WM(TT, 16, WS); else RC_WM(TT, 32, WS); } while (0)
#define RC_WM_WS(TT) do { if (window == 8) RC_WM_HD(TT, 8); else RC_WM_HD(TT, 4); } while (0)
    if (dtype == RC_F32) RC_WM_WS(float); else RC_WM_WS(bf16_t);
#undef RC_WM_WS
#undef RC_WM_HD
#undef RC_WM
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

}  // extern "C"
