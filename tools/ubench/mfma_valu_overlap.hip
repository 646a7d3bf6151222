// Do fp32 MFMAs (v_mfma_f32_16x16x4_f32) and plain VALU work of ANOTHER wave on the same SIMD overlap?  (bf16 MFMA for comparison.)
// 256 blocks x 512 threads: waves 0-3 (one per SIMD) play role A, waves 4-7 role B.  hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mvo && ./mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int ROLE_A, int ROLE_B>   // 0 idle, 1 fp32 MFMA, 2 VALU fma, 3 bf16 MFMA, 4 VALU add (no fma), 5 LDS reads
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? ROLE_A : ROLE_B;
    float r = 0.f;
    if (role == 1) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 16; ++i) r += acc[i].x;
    } else if (role == 3) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(threadIdx.x * 2e-3f - i); }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 16; ++i) r += acc[i].x;
    } else if (role == 2 || role == 4) {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
        const float m = 1.0001f, c = 1e-4f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (role == 2) v[i] = __builtin_fmaf(v[i], m, c);
                    else { v[i] = v[i] + c; asm volatile("" : "+v"(v[i])); }
                }
        for (int i = 0; i < 16; ++i) r += v[i];
    } else if (role == 5) {
        for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
        __builtin_amdgcn_s_waitcnt(0);
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 16; ++rep) {
                const f32x4 t = *reinterpret_cast<volatile f32x4*>(&lds[((threadIdx.x & 63) * 4 + rep * 256) & 4095]);
                s += t;
            }
        r = s.x + s.y + s.z + s.w;
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int A, int B>
static float run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    float* d; hipMalloc(&d, 4096);
    const int it = 2000;   // per wave: 32 000 MFMAs, or 256 000 VALU ops
    printf("fp32 MFMA alone (1 wave/SIMD)        : %8.1f us\n", run<1, 0>(d, it));
    printf("fp32 MFMA x2 waves/SIMD              : %8.1f us\n", run<1, 1>(d, it));
    printf("VALU fma alone                       : %8.1f us\n", run<0, 2>(d, it));
    printf("VALU add alone                       : %8.1f us\n", run<0, 4>(d, it));
    printf("fp32 MFMA + VALU fma (other wave)    : %8.1f us\n", run<1, 2>(d, it));
    printf("fp32 MFMA + VALU add (other wave)    : %8.1f us\n", run<1, 4>(d, it));
    printf("bf16 MFMA alone                      : %8.1f us\n", run<3, 0>(d, it));
    printf("bf16 MFMA + VALU fma (other wave)    : %8.1f us\n", run<3, 2>(d, it));
    printf("LDS b128 reads alone                 : %8.1f us\n", run<0, 5>(d, it));
    printf("fp32 MFMA + LDS reads (other wave)   : %8.1f us\n", run<1, 5>(d, it));
    return 0;
}
