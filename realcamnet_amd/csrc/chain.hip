// Fused chain of point-wise (1x1) convolutions on 48-channel bf16 pixels: the lens-shading MLP
//   coord (B,H,W,cin0<=4) -> Conv1x1(cin0,48) -> LeakyReLU -> [Conv1x1(48,48) -> LeakyReLU] x (n_mid-1) -> Conv1x1(48,48)
// (models/LiteISP.py:363-378).  As separate rc_conv2d launches every layer writes and re-reads a full-resolution
// 48-channel map (1.6 GB each way at 4K x 8); a 1x1 conv has no halo, so a wave can carry its 64 pixels through ALL
// layers with the activations parked in a wave-private 6 KB LDS slab: HBM sees the 4-byte coordinates in and the final
// map out.  Layer 0 (K = cin0) is plain FMAs; the 48 -> 48 layers are v_mfma_f32_16x16x32_bf16 with the weight
// fragment layout of rc_conv_pack_weights(48, 48, 1, RC_BF16, RC_OUT_NHWC) and the unit map of conv_kernel.hpp
// (step 0: channel units 0-3, step 1: units 4,5 + zero padding).  Activations are rounded to bf16 between layers,
// exactly where the layer-by-layer path rounds them.
#include "conv_kernel.hpp"

namespace rc {
namespace chain {

constexpr int C = 48, NT = 3, NV = 12, SPIX = 96, STEPS = 2, MAX_MID = 4, MAX_CIN0 = 8;
constexpr int W0_BYTES = NT * 1024;                   // layer 0: Cin <= 8 = one 16-byte unit = one (zero-padded) MFMA step
constexpr int W_BYTES = STEPS * NT * 1024;            // one packed 48x48 matrix
constexpr int SLAB = 64 * SPIX;                        // one wave's 64 pixels

struct Args {
    const bf16_t* x; int cin0; const void* wp0; const float* bp0;     // layer 0: rc_conv_pack_weights(cin0, 48, 1) (one K-padded step), packed bias
    const void* wp[MAX_MID]; const float* bp[MAX_MID]; int n_mid;     // packed 48x48 layers
    float slope; bf16_t* out; long long pixels;
};

__global__ __launch_bounds__(256) void pointwise_chain48_kernel(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, n = lane & 15;
    char* s_w = smem;                                               // [n_mid][W_BYTES], then layer 0's [W0_BYTES]
    char* s_w0 = smem + MAX_MID * W_BYTES;
    float* s_b = reinterpret_cast<float*>(smem + MAX_MID * W_BYTES + W0_BYTES);  // [1 + n_mid][48]: layer-0 bias, then the others
    char* slab = smem + MAX_MID * W_BYTES + W0_BYTES + (1 + MAX_MID) * C * 4 + wave * SLAB;

    for (int m = 0; m < a.n_mid; ++m)
        for (int kb = wave; kb < STEPS * NT; kb += 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(static_cast<const char*>(a.wp[m]) + kb * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(s_w + m * W_BYTES + kb * 1024), 16, 0, 0);
    if (wave < NT)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(static_cast<const char*>(a.wp0) + wave * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(s_w0 + wave * 1024), 16, 0, 0);
    for (int i = tid; i < C; i += 256) s_b[i] = a.bp0 ? a.bp0[i] : 0.f;
    for (int m = 0; m < a.n_mid; ++m)
        for (int i = tid; i < C; i += 256) s_b[(1 + m) * C + i] = a.bp[m] ? a.bp[m][i] : 0.f;
    __syncthreads();

    const float inf = __builtin_inff();
    const long long groups = (a.pixels + 63) / 64;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < groups; g += (long long)gridDim.x * 4) {
        const long long p0 = g * 64;
        // ---- layer 0 (Cin <= 8): the coordinates go to the slab as ONE zero-padded 16-byte unit per pixel and the layer is a
        //      single MFMA step (K = 32, only lane group 0 carries data) -- same instruction, same rounding as rc_conv2d
        {
            const long long p = p0 + lane;
            float xin[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) xin[i] = (p < a.pixels && i < a.cin0) ? to_f32(a.x[p * a.cin0 + i]) : 0.f;
            *reinterpret_cast<uint4*>(slab + lane * SPIX) = Vec16<bf16_t>::pack(xin);
        }
        __builtin_amdgcn_wave_barrier();                              // one wave's LDS ops complete in order
        {
            f32x4 acc[4][NT];
            uint4 wf[NT], xf[4];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 t4 = *reinterpret_cast<const float4*>(s_b + q * NV + nt * 4);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[pt][nt] = f32x4{t4.x, t4.y, t4.z, t4.w};
                wf[nt] = *reinterpret_cast<const uint4*>(s_w0 + nt * 1024 + lane * 16);
            }
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                xf[pt] = *reinterpret_cast<const uint4*>(slab + (16 * pt + n) * SPIX);
                if (q > 0) xf[pt] = make_uint4(0u, 0u, 0u, 0u);       // K slots 8..31 are padding
            }
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) Mma<bf16_t>::run(wf[nt], xf[pt], acc[pt][nt]);
            __builtin_amdgcn_wave_barrier();                          // every lane has read the coordinates before they are overwritten
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                unsigned wd[NV / 2];
#pragma unroll
                for (int e = 0; e < NV / 2; ++e) {
                    const float v0 = acc[pt][(2 * e) / 4][(2 * e) % 4], v1 = acc[pt][(2 * e + 1) / 4][(2 * e + 1) % 4];
                    wd[e] = pack_bf16x2(__builtin_amdgcn_fmed3f(v0, v0 * a.slope, inf), __builtin_amdgcn_fmed3f(v1, v1 * a.slope, inf));
                }
                char* dst = slab + (16 * pt + n) * SPIX + q * (NV * 2);
#pragma unroll
                for (int e = 0; e < NV / 2; e += 2) *reinterpret_cast<uint2*>(dst + 4 * e) = make_uint2(wd[e], wd[e + 1]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- 48 -> 48 layers on MFMA; lane (q, n) of pixel tile pt holds channels 12q..12q+11 of pixel 16*pt + n
        for (int m = 0; m < a.n_mid; ++m) {
            f32x4 acc[4][NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 t4 = *reinterpret_cast<const float4*>(s_b + (1 + m) * C + q * NV + nt * 4);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[pt][nt] = f32x4{t4.x, t4.y, t4.z, t4.w};
            }
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                uint4 wf[NT], xf[4];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const uint4*>(s_w + m * W_BYTES + (s * NT + nt) * 1024 + lane * 16);
                const int uoff = s == 0 ? q * 16 : (4 + (q & 1)) * 16;     // step 1: units 4,5 (q = 0,1); q >= 2 is padding
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    xf[pt] = *reinterpret_cast<const uint4*>(slab + (16 * pt + n) * SPIX + uoff);
                    if (s == 1 && q >= 2) xf[pt] = make_uint4(0u, 0u, 0u, 0u);
                }
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) Mma<bf16_t>::run(wf[nt], xf[pt], acc[pt][nt]);
            }
            const bool last = m + 1 == a.n_mid;
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                float v[NV];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[pt][nt][r];
                if (!last) {
#pragma unroll
                    for (int e = 0; e < NV; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], v[e] * a.slope, inf);
                }
                unsigned wd[NV / 2];
#pragma unroll
                for (int e = 0; e < NV / 2; ++e) wd[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                if (!last) {
                    char* dst = slab + (16 * pt + n) * SPIX + q * (NV * 2);
#pragma unroll
                    for (int e = 0; e < NV / 2; e += 2) *reinterpret_cast<uint2*>(dst + 4 * e) = make_uint2(wd[e], wd[e + 1]);
                } else {
                    const long long p = p0 + 16 * pt + n;
                    if (p < a.pixels) {
                        bf16_t* dst = a.out + p * C + q * NV;
                        *reinterpret_cast<uint4*>(dst) = make_uint4(wd[0], wd[1], wd[2], wd[3]);
                        *reinterpret_cast<uint2*>(dst + 8) = make_uint2(wd[4], wd[5]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

constexpr int LDS_BYTES = MAX_MID * W_BYTES + W0_BYTES + (1 + MAX_MID) * C * 4 + 4 * SLAB;

}  // namespace chain
}  // namespace rc

using namespace rc;

extern "C" {

int rc_pointwise_chain48(const void* d_x, int cin0, const void* d_w0packed, const float* d_b0, const void* const* d_wpacked,
                         const float* const* d_bias, int n_mid, float slope, void* d_out, int dtype, long long pixels,
                         void* stream) {
    RC_REQUIRE(d_x && d_w0packed && d_out && d_wpacked && d_bias, "rc_pointwise_chain48: null pointer");
    RC_REQUIRE(dtype == RC_BF16, "rc_pointwise_chain48: bf16 only (use rc_conv2d per layer otherwise)");
    RC_REQUIRE(cin0 >= 1 && cin0 <= chain::MAX_CIN0 && n_mid >= 1 && n_mid <= chain::MAX_MID && pixels >= 1,
               "rc_pointwise_chain48: 1 <= cin0 <= 8, 1 <= n_mid <= 4");
    RC_REQUIRE(slope >= 0.f && slope <= 1.f, "rc_pointwise_chain48: slope must be in [0, 1]");
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_out) % 16 == 0, "rc_pointwise_chain48: out must be 16-byte aligned");
    chain::Args a{};
    a.x = static_cast<const bf16_t*>(d_x); a.cin0 = cin0; a.wp0 = d_w0packed; a.bp0 = d_b0;
    for (int m = 0; m < n_mid; ++m) {
        RC_REQUIRE(d_wpacked[m] != nullptr, "rc_pointwise_chain48: null packed weights");
        a.wp[m] = d_wpacked[m]; a.bp[m] = d_bias[m];
    }
    a.n_mid = n_mid; a.slope = slope; a.out = static_cast<bf16_t*>(d_out); a.pixels = pixels;
    const int num_cus = device_cu_count();          // per device (common.hpp)
    const long long groups = (pixels + 63) / 64;
    long long grid = (groups + 3) / 4;
    if (grid > 3LL * num_cus) grid = 3LL * num_cus;                   // 3 blocks per CU fit the LDS budget
    hipLaunchKernelGGL(chain::pointwise_chain48_kernel, dim3((unsigned)grid), dim3(256), chain::LDS_BYTES, as_stream(stream), a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

}  // extern "C"
