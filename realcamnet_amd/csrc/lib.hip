// Library-level entry points: ABI version, error text, device query.
#include <string.h>

#include "common.hpp"

namespace rc {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) { g_err = msg; return code; }
}  // namespace rc

namespace rc {
// Matrix-pipe calibration: every wave issues nothing but independent v_mfma_f32_16x16x32_bf16 (16 accumulators,
// operands in registers).  What this sustains is the clock-and-power-limited MFMA rate of THIS part, the number a
// conv kernel's MFMA utilisation should be read against (the datasheet peak assumes the maximum engine clock).
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* sink, int iters, long long* cycles) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float seed = (float)threadIdx.x * 1e-3f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) sink[0] = s;                       // keep the chain alive
    if (blockIdx.x == 0 && threadIdx.x == 0 && cycles) cycles[0] = t1 - t0;
}
__global__ __launch_bounds__(256) void mfma_peak32_kernel(float* sink, int iters, long long* cycles) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const float seed = (float)threadIdx.x * 1e-3f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][5] + acc[i][10] + acc[i][15];
    if (s == 123.456f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0 && cycles) cycles[0] = t1 - t0;
}
// The same two loops with RANDOM operands that change from MFMA to MFMA (4 A and 4 B register sets per wave, random sign / exponent / mantissa bits):
// what the matrix pipe sustains -- and draws -- on data that toggles like real activations and weights.  With constant operands (above) the part runs
// at its maximum clock; on random data it sits at the board's power cap (tools/power_probe.py), which is the ceiling a conv kernel sees.
__device__ __forceinline__ unsigned rnd_hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ unsigned rnd_bf16_pair(unsigned h) {          // two bf16 values +-[0.5, 4): sign, 2 exponent bits, 7 mantissa bits each from h
    const unsigned lo = ((h & 1u) << 15) | ((126u + ((h >> 1) & 3u)) << 7) | ((h >> 3) & 127u);
    const unsigned hi = (((h >> 10) & 1u) << 15) | ((126u + ((h >> 11) & 3u)) << 7) | ((h >> 13) & 127u);
    return lo | (hi << 16);
}
template <int KIND>
__global__ __launch_bounds__(256) void mfma_power_kernel(float* sink, int iters) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    bf16x8 a[4], b[4];
    unsigned h = rnd_hash(blockIdx.x * 256u + threadIdx.x + 1u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32x4 ua, ub;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h = rnd_hash(h + e + 17u * i); ua[e] = rnd_bf16_pair(h); h = rnd_hash(h); ub[e] = rnd_bf16_pair(h); }
        a[i] = __builtin_bit_cast(bf16x8, ua); b[i] = __builtin_bit_cast(bf16x8, ub);
    }
    float s = 0.f;
    if constexpr (KIND == 16) {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][5] + acc[i][10] + acc[i][15];
    }
    if (s == 123.456f) sink[0] = s;
}
}  // namespace rc


namespace rc {
// HBM probe kernels (tools/hbm_probe.py): what does this part sustain for a streaming copy, read or write?
typedef float f4_t __attribute__((ext_vector_type(4)));
template <int MODE, int NT>    // MODE 0 copy, 1 read-only, 2 write-only;  NT: non-temporal accesses
__global__ void __launch_bounds__(256) hbm_probe_kernel(const f4_t* __restrict__ src, f4_t* __restrict__ dst, size_t n16,
                                                         int contiguous, float* sink) {
    const size_t nthreads = (size_t)gridDim.x * 256;
    size_t i, step, end;
    if (contiguous) {                   // each block owns one contiguous range
        const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
        i = blockIdx.x * per + threadIdx.x; step = 256; end = (blockIdx.x + 1) * per < n16 ? (blockIdx.x + 1) * per : n16;
    } else { i = (size_t)blockIdx.x * 256 + threadIdx.x; step = nthreads; end = n16; }
    f4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + 3 * step < end; i += 4 * step) {
        f4_t v[4];
        if (MODE != 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * step) : src[i + u * step];
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = f4_t{1.f, 2.f, 3.f, (float)u};
        }
        if (MODE != 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * step); else dst[i + u * step] = v[u]; }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u];
        }
    }
    for (; i < end; i += step) {
        f4_t v = MODE != 2 ? src[i] : f4_t{1.f, 2.f, 3.f, 4.f};
        if (MODE != 1) dst[i] = v; else acc += v;
    }
    if (MODE == 1 && acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}
}  // namespace rc

namespace rc {
// fills the whole LDS allocation of every CU with a signalling pattern (bf16 / fp32 NaNs): a kernel that reads LDS it has not written -- or
// before the write has landed -- then shows up in the parity tests whatever ran before it
__global__ __launch_bounds__(256) void lds_poison_kernel(unsigned pattern, int words) {
    extern __shared__ unsigned pz[];
    for (int i = threadIdx.x; i < words; i += 256) pz[i] = pattern;
    __syncthreads();
    if (pz[(threadIdx.x * 97) % words] != pattern) __builtin_trap();      // keeps the stores
}
}  // namespace rc

extern "C" {

int rc_debug_mfma_peak(int waves_per_simd, int iters, double* tflops, double* memtime_ticks_per_mfma) {
    const bool random_ops = waves_per_simd > 10;             // 11 / 12: random operands that change from MFMA to MFMA (power, not clock, limits those)
    if (random_ops) waves_per_simd -= 10;
    RC_REQUIRE(tflops != nullptr && iters >= 1 && waves_per_simd >= 1 && waves_per_simd <= 2, "rc_debug_mfma_peak: bad arguments");
    int dev = 0, cus = 0;
    RC_HIP_CHECK(hipGetDevice(&dev));
    RC_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    float* sink = nullptr; long long* cyc = nullptr;
    RC_HIP_CHECK(hipMalloc(&sink, 4)); RC_HIP_CHECK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1;
    RC_HIP_CHECK(hipEventCreate(&e0)); RC_HIP_CHECK(hipEventCreate(&e1));
    const int grid = cus * waves_per_simd;                // 256 threads = 4 waves = one per SIMD
    for (int rep = 0; rep < 3; ++rep) {                   // first passes bring the clocks up
        RC_HIP_CHECK(hipEventRecord(e0, nullptr));
        if (random_ops) hipLaunchKernelGGL(rc::mfma_power_kernel<16>, dim3(grid), dim3(256), 0, nullptr, sink, iters);
        else hipLaunchKernelGGL(rc::mfma_peak_kernel, dim3(grid), dim3(256), 0, nullptr, sink, iters, cyc);
        RC_HIP_CHECK(hipEventRecord(e1, nullptr));
        RC_HIP_CHECK(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    RC_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long h = 0;
    RC_HIP_CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    *tflops = (double)grid * 4 * iters * 16 * 16384.0 / (ms * 1e-3) / 1e12;
    if (memtime_ticks_per_mfma) *memtime_ticks_per_mfma = (double)h / ((double)iters * 16 * waves_per_simd);
    (void)hipFree(sink); (void)hipFree(cyc); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return RC_OK;
}

int rc_debug_poison_lds(unsigned pattern, void* stream) {
    constexpr int kBytes = 160 * 1024;
    static rc::PerDeviceFlag attr;
    if (!attr.test_and_set())
        RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rc::lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kBytes));
    hipLaunchKernelGGL(rc::lds_poison_kernel, dim3((unsigned)(rc::device_cu_count() * 4)), dim3(256), kBytes, static_cast<hipStream_t>(stream), pattern, kBytes / 4);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_debug_mfma_peak32(int waves_per_simd, int iters, double* tflops, double* memtime_ticks_per_mfma) {
    const bool random_ops = waves_per_simd > 10;
    if (random_ops) waves_per_simd -= 10;
    RC_REQUIRE(tflops != nullptr && iters >= 1 && waves_per_simd >= 1 && waves_per_simd <= 2, "rc_debug_mfma_peak32: bad arguments");
    const int cus = rc::device_cu_count();
    float* sink = nullptr; long long* cyc = nullptr;
    RC_HIP_CHECK(hipMalloc(&sink, 4)); RC_HIP_CHECK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1;
    RC_HIP_CHECK(hipEventCreate(&e0)); RC_HIP_CHECK(hipEventCreate(&e1));
    const int grid = cus * waves_per_simd;
    for (int rep = 0; rep < 3; ++rep) {
        RC_HIP_CHECK(hipEventRecord(e0, nullptr));
        if (random_ops) hipLaunchKernelGGL(rc::mfma_power_kernel<32>, dim3(grid), dim3(256), 0, nullptr, sink, iters);
        else hipLaunchKernelGGL(rc::mfma_peak32_kernel, dim3(grid), dim3(256), 0, nullptr, sink, iters, cyc);
        RC_HIP_CHECK(hipEventRecord(e1, nullptr));
        RC_HIP_CHECK(hipEventSynchronize(e1));
    }
    float ms = 0.f;
    RC_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long h = 0;
    RC_HIP_CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    *tflops = (double)grid * 4 * iters * 8 * 32768.0 / (ms * 1e-3) / 1e12;
    if (memtime_ticks_per_mfma) *memtime_ticks_per_mfma = (double)h / ((double)iters * 8 * waves_per_simd);
    (void)hipFree(sink); (void)hipFree(cyc); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return RC_OK;
}

int rc_debug_hbm_probe(const void* src, void* dst, size_t bytes, int mode, int nt, int contiguous, int blocks, int iters,
                       double* ms_per_iter) {
    RC_REQUIRE(src && dst && ms_per_iter && bytes % 16 == 0 && mode >= 0 && mode <= 2 && iters >= 1 && blocks >= 0,
               "rc_debug_hbm_probe: bad arguments");
    const size_t n16 = bytes / 16;
    const int grid = blocks > 0 ? blocks : (int)((n16 + 1023) / 1024);
    float* sink = nullptr;
    RC_HIP_CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    RC_HIP_CHECK(hipEventCreate(&e0)); RC_HIP_CHECK(hipEventCreate(&e1));
    auto go = [&]() {
#define RC_PROBE(M, N) hipLaunchKernelGGL((rc::hbm_probe_kernel<M, N>), dim3(grid), dim3(256), 0, nullptr, (const rc::f4_t*)src, (rc::f4_t*)dst, n16, contiguous, sink)
        if (mode == 0) { if (nt) RC_PROBE(0, 1); else RC_PROBE(0, 0); }
        else if (mode == 1) { if (nt) RC_PROBE(1, 1); else RC_PROBE(1, 0); }
        else { if (nt) RC_PROBE(2, 1); else RC_PROBE(2, 0); }
#undef RC_PROBE
    };
    for (int w = 0; w < 20; ++w) go();
    RC_HIP_CHECK(hipEventRecord(e0, nullptr));
    for (int w = 0; w < iters; ++w) go();
    RC_HIP_CHECK(hipEventRecord(e1, nullptr));
    RC_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    RC_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_iter = ms / iters;
    (void)hipFree(sink); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return RC_OK;
}

int rc_debug_stream_create_masked(int kind, void** stream_out) {
    RC_REQUIRE(stream_out != nullptr && kind >= 0 && kind <= 3, "rc_debug_stream_create_masked: kind 0..3");
    // experiment: a stream confined to half of the chip's CUs.  kind 0/1: lower / upper 128 mask bits; kind 2/3: lower /
    // upper 16 bits of every 32-bit word
    uint32_t mask[8];
    for (int i = 0; i < 8; ++i)
        mask[i] = kind == 0 ? (i < 4 ? 0xffffffffu : 0u) : kind == 1 ? (i < 4 ? 0u : 0xffffffffu) : kind == 2 ? 0x0000ffffu : 0xffff0000u;
    hipStream_t st = nullptr;
    RC_HIP_CHECK(hipExtStreamCreateWithCUMask(&st, 8, mask));
    *stream_out = st;
    return RC_OK;
}

int rc_abi_version(void) { return RC_ABI_VERSION; }

const char* rc_last_error(void) { return rc::g_err.c_str(); }

const char* rc_build_info(void) { return "librealcam_hip gfx950 (" __VERSION__ ")"; }

int rc_device_arch(char* buf, size_t buflen) {
    RC_REQUIRE(buf != nullptr && buflen > 0, "rc_device_arch: null buffer");
    int dev = 0;
    RC_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    RC_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    strncpy(buf, prop.gcnArchName, buflen - 1);
    buf[buflen - 1] = '\0';
    return RC_OK;
}

}  // extern "C"
