// Likelihood path of the codec's entropy models: what TCM.forward (upstream models/tcm.py:437-486)
// asks of CompressAI's EntropyBottleneck and GaussianConditional in eval mode, as element-wise kernels.  The CompressAI
// classes are not in the upstream tree; the arithmetic follows their published definitions (parity unpinned).
#include "../../include/realcam_hip.h"
#include "common.hpp"

namespace rc {

constexpr int kEbParams = 58;   // per channel: softplus(matrix), bias, tanh(factor) of the 1-3-3-3-3-1 cumulative-logit network

template <typename T> __device__ __forceinline__ float ld1(const T* p, size_t i) { return to_f32(p[i]); }
template <typename T> __device__ __forceinline__ void st1(T* p, size_t i, float v) { p[i] = from_f32<T>(v); }

// logits of the cumulative density at x for one channel (EntropyBottleneck._logits_cumulative)
__device__ __forceinline__ float eb_logits(const float* __restrict__ P, float x) {
    float h[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float v = P[j] * x + P[3 + j];
        h[j] = v + P[6 + j] * tanhf(v);
    }
    const float* Q = P + 9;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        float g[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float v = Q[3 * j] * h[0] + Q[3 * j + 1] * h[1] + Q[3 * j + 2] * h[2] + Q[9 + j];
            g[j] = v + Q[12 + j] * tanhf(v);
        }
        h[0] = g[0]; h[1] = g[1]; h[2] = g[2];
        Q += 15;
    }
    return Q[0] * h[0] + Q[1] * h[1] + Q[2] * h[2] + Q[3];
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }
// torch.round(t) - t.detach() + t  (ste_round, models/tcm.py:36-37), evaluated in that order
__device__ __forceinline__ float ste_round(float t) { return (rintf(t) - t) + t; }

template <typename T>
__global__ void entropy_bottleneck_kernel(const T* __restrict__ z, const float* __restrict__ params, const float* __restrict__ medians,
                                          T* __restrict__ z_hat, float* __restrict__ lik, size_t total, int C, float bound) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float x = ld1(z, i), med = medians[c];
        const float* P = params + (size_t)c * kEbParams;
        const float t = x - med;
        const float out = rintf(t) + med;                     // quantize(..., "dequantize", medians)
        const float lower = eb_logits(P, out - 0.5f), upper = eb_logits(P, out + 0.5f);
        const float s = lower + upper;
        const float sign = s > 0.f ? -1.f : (s < 0.f ? 1.f : 0.f);
        float l = fabsf(sigmoidf_(sign * upper) - sigmoidf_(sign * lower));
        lik[i] = l > bound ? l : bound;
        st1(z_hat, i, ste_round(t) + med);
    }
}

template <typename T>
__global__ void gaussian_conditional_kernel(const T* __restrict__ y, const T* __restrict__ scale, const T* __restrict__ mu,
                                            T* __restrict__ y_hat, float* __restrict__ lik, size_t total, float scale_bound, float bound) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float v = ld1(y, i), m = ld1(mu, i);
        float s = ld1(scale, i);
        s = s > scale_bound ? s : scale_bound;
        const float t = v - m;
        const float out = rintf(t) + m;
        const float a = fabsf(out - m);
        const float k = -0.70710678118654752f;
        const float upper = 0.5f * erfcf(k * ((0.5f - a) / s)), lower = 0.5f * erfcf(k * ((-0.5f - a) / s));
        const float l = upper - lower;
        lik[i] = l > bound ? l : bound;
        st1(y_hat, i, ste_round(t) + m);
    }
}

template <typename T>
__global__ void tanh_half_add_kernel(const T* __restrict__ a, const T* __restrict__ lrp, T* __restrict__ out, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        st1(out, i, ld1(a, i) + 0.5f * tanhf(ld1(lrp, i)));
}

static inline int eb_grid(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace rc

using namespace rc;

extern "C" {

int rc_entropy_bottleneck(const void* d_z, const float* d_params, const float* d_medians, void* d_z_hat, float* d_likelihood,
                          int dtype, long long n_pix, int channels, float likelihood_bound, void* stream) {
    RC_REQUIRE(d_z && d_params && d_medians && d_z_hat && d_likelihood, "rc_entropy_bottleneck: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_entropy_bottleneck: bad dtype");
    RC_REQUIRE(n_pix >= 1 && channels >= 1, "rc_entropy_bottleneck: bad shape");
    const size_t total = (size_t)n_pix * channels;
    if (dtype == RC_F32)
        hipLaunchKernelGGL(entropy_bottleneck_kernel<float>, dim3(eb_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const float*>(d_z), d_params, d_medians, static_cast<float*>(d_z_hat), d_likelihood, total, channels, likelihood_bound);
    else
        hipLaunchKernelGGL(entropy_bottleneck_kernel<bf16_t>, dim3(eb_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const bf16_t*>(d_z), d_params, d_medians, static_cast<bf16_t*>(d_z_hat), d_likelihood, total, channels, likelihood_bound);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_gaussian_conditional(const void* d_y, const void* d_scale, const void* d_mu, void* d_y_hat, float* d_likelihood, int dtype,
                            long long n_elems, float scale_bound, float likelihood_bound, void* stream) {
    RC_REQUIRE(d_y && d_scale && d_mu && d_y_hat && d_likelihood, "rc_gaussian_conditional: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_gaussian_conditional: bad dtype");
    RC_REQUIRE(n_elems >= 1 && scale_bound > 0.f, "rc_gaussian_conditional: bad arguments");
    const size_t total = (size_t)n_elems;
    if (dtype == RC_F32)
        hipLaunchKernelGGL(gaussian_conditional_kernel<float>, dim3(eb_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const float*>(d_y), static_cast<const float*>(d_scale), static_cast<const float*>(d_mu),
                           static_cast<float*>(d_y_hat), d_likelihood, total, scale_bound, likelihood_bound);
    else
        hipLaunchKernelGGL(gaussian_conditional_kernel<bf16_t>, dim3(eb_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const bf16_t*>(d_y), static_cast<const bf16_t*>(d_scale), static_cast<const bf16_t*>(d_mu),
                           static_cast<bf16_t*>(d_y_hat), d_likelihood, total, scale_bound, likelihood_bound);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_tanh_half_add(const void* d_a, const void* d_lrp, void* d_out, int dtype, long long n_elems, void* stream) {
    RC_REQUIRE(d_a && d_lrp && d_out, "rc_tanh_half_add: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_tanh_half_add: bad dtype");
    RC_REQUIRE(n_elems >= 1, "rc_tanh_half_add: bad shape");
    const size_t total = (size_t)n_elems;
    if (dtype == RC_F32)
        hipLaunchKernelGGL(tanh_half_add_kernel<float>, dim3(eb_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const float*>(d_a), static_cast<const float*>(d_lrp), static_cast<float*>(d_out), total);
    else
        hipLaunchKernelGGL(tanh_half_add_kernel<bf16_t>, dim3(eb_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const bf16_t*>(d_a), static_cast<const bf16_t*>(d_lrp), static_cast<bf16_t*>(d_out), total);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

}  // extern "C"
