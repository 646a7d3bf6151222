// Companion of mfma_valu_overlap.hip: the fillers are in the SAME wave's instruction stream, F independent v_fma_f32 after every MFMA
// (one wave per SIMD, 16 rotating accumulators, inline asm so that hipcc keeps the order).  If a SIMD hid VALU issue under a running
// MFMA pass, time per MFMA would stay flat until F fills the pass (32 cycles for 16x16x4 f32, 16 for 16x16x32 bf16) and grow after.
// hipcc --offload-arch=gfx950 -O3 mfma_valu_samewave.hip -o mvs && ./mvs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND /* 0 fp32 16x16x4, 1 bf16 16x16x32, 2 no MFMA */, int F, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a32 = threadIdx.x * 1e-3f, b32 = threadIdx.x * 2e-3f;
    bf16x8 a16, b16;
    for (int i = 0; i < 8; ++i) { a16[i] = (__bf16)(threadIdx.x * 1e-3f + i); b16[i] = (__bf16)(threadIdx.x * 2e-3f - i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const float m = 1.0001f, c = 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a32), "v"(b32));
            if (KIND == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a16), "v"(b16));
#pragma unroll
            for (int f = 0; f < F; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[f & 7]) : "v"(m), "v"(c));
        }
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc[i].x;
    for (int i = 0; i < 8; ++i) r += v[i];
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int KIND, int F, int WAVES>
static float run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<KIND, F, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, d, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<KIND, F, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

template <int KIND, int WAVES>
static void sweep(const char* name, float* d, int it) {
    const float t[9] = {run<KIND, 0, WAVES>(d, it), run<KIND, 1, WAVES>(d, it), run<KIND, 2, WAVES>(d, it), run<KIND, 3, WAVES>(d, it),
                        run<KIND, 4, WAVES>(d, it), run<KIND, 6, WAVES>(d, it), run<KIND, 8, WAVES>(d, it), run<KIND, 12, WAVES>(d, it),
                        run<KIND, 16, WAVES>(d, it)};
    printf("%-44s F = 0 1 2 3 4 6 8 12 16 :", name);
    for (int i = 0; i < 9; ++i) printf(" %7.1f", t[i]);
    printf("  us\n");
}

int main() {
    float* d; hipMalloc(&d, 4096);
    const int it = 2000;   // 32 000 MFMAs per wave, 32 000 F fillers
    sweep<2, 4>("fillers only, 1 wave / SIMD", d, it);
    sweep<0, 4>("fp32 16x16x4 + F fillers, 1 wave / SIMD", d, it);
    sweep<1, 4>("bf16 16x16x32 + F fillers, 1 wave / SIMD", d, it);
    sweep<0, 8>("fp32 16x16x4 + F fillers, 2 waves / SIMD", d, it);
    sweep<1, 8>("bf16 16x16x32 + F fillers, 2 waves / SIMD", d, it);
    return 0;
}
