// Shared host/device helpers for librealcam_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/realcam_hip.h"

namespace rc {

// ---- error plumbing ---------------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define RC_HIP_CHECK(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return ::rc::fail(RC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define RC_REQUIRE(cond, msg)                                             \
    do {                                                                  \
        if (!(cond)) return ::rc::fail(RC_ERR_INVALID, std::string(msg)); \
    } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- element types ------------------------------------------------------------------------------
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr int value = RC_F32; };
template <> struct dtype_of<bf16_t> { static constexpr int value = RC_BF16; };

__host__ __device__ inline size_t dtype_size(int dt) { return dt == RC_F32 ? 4 : 2; }

template <typename T> __device__ __forceinline__ float to_f32(T v) { return static_cast<float>(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return static_cast<T>(v); }

// 16-byte vector of T viewed as floats and back (UNIT = 16/sizeof(T) elements).
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void unpack(const uint4& raw, float* f) {
        f[0] = __uint_as_float(raw.x); f[1] = __uint_as_float(raw.y);
        f[2] = __uint_as_float(raw.z); f[3] = __uint_as_float(raw.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void unpack(const uint4& raw, float* f) {
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ uint32_t rne(float v) {  // fp32 -> bf16 bits, round-nearest-even
        return static_cast<uint32_t>(__builtin_bit_cast(uint16_t, static_cast<bf16_t>(v)));
    }
    // two values per v_cvt_pk_bf16_f32 (round-nearest-even, the same instruction the scalar cast selects): converting them one
    // at a time costs two conversions and a v_perm per pair.  Written as an instruction because the vector-typed
    // __builtin_convertvector form keeps the callers' unrolled value arrays from being promoted to registers (scratch).
    // HAZARD: inline asm is opaque to the compiler's hazard recogniser -- never pass an MFMA accumulator straight in (no wait states
    // are inserted for the MFMA result latency); route it through a real VALU op first (the epilogues' bias add / activation do).
    // The same holds for a transcendental result (v_exp_f32, v_rcp_f32, v_rsq_f32 ...: one wait state before a VALU read, which the
    // compiler inserts for its own instructions only) -- see wm_pk in wmsa.hip.
    __device__ static __forceinline__ uint32_t rne2(float lo, float hi) {
        uint32_t r;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
        return r;
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = rne2(f[2 * i], f[2 * i + 1]);
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};

__host__ inline uint16_t host_f32_to_bf16(float f) {  // round-nearest-even, NaN preserved
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return static_cast<uint16_t>(u >> 16);
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == RC_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == RC_ACT_LEAKY) return v > 0.f ? v : v * slope;
    return v;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Function attributes (dynamic-LDS limit) and device properties are PER DEVICE in HIP: per-process `static bool` flags would leave a
// second GPU touched by the same process without them.  Small per-device tables instead (index = hipGetDevice(), < 64).
inline int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    return dev;
}
struct PerDeviceFlag {
    bool done[64] = {};
    bool test_and_set() { const int d = current_device(); const bool was = done[d]; done[d] = true; return was; }
};
inline int device_cu_count() {
    static int cus[64] = {};
    const int d = current_device();
    if (!cus[d]) (void)hipDeviceGetAttribute(&cus[d], hipDeviceAttributeMultiprocessorCount, d);
    return cus[d] > 0 ? cus[d] : 256;
}

// exact-erf GELU, 0.5 v (1 + erf(v / sqrt 2)) (nn.GELU() of groupmix.Mlp / tcm.Block), with erf as ONE odd polynomial: w = clamp(v, +-5), t = 2 w^2 / 25 - 1 in [-1, 1],
// erf(w / sqrt 2) = w Q(t), Q of degree 12 (a Remez-like weighted least-squares fit, tools/gelu_fit.py).  |error| <= 6.7e-7 as evaluated in fp32 (Horner in t is well
// conditioned) -- the Abramowitz-Stegun 7.1.28 form this replaces, 1 - (1 + a1 z + .. + a6 z^6)^-16, reaches 1.9e-6 in fp32 because its ^16 amplifies the rounding, and costs
// 13.5 VALU issue slots per value in packed code (a quarter-rate reciprocal, |v|, a sign transfer) against 9.5 here: GELU was 63 % of gma_tail's vector instructions.
// Saturates exactly: 5 Q(1) == 1.0f, so GELU(v <= -5) == 0 and GELU(v >= 5) == v.  No transcendental, no libm call (ocml's erff is ~40 instructions).
constexpr float kGeluC[13] = {0.28272763f, -0.14059256f, 0.103041045f, -0.08088522f, 0.062884346f, -0.046537306f, 0.032804348f, -0.02232868f,
                              0.012821025f, -0.0053711478f, 0.0034455948f, -0.0032265312f, 0.0012174561f};     // Q(t) = sum kGeluC[i] t^i
__device__ __forceinline__ float gelu_erf_f32(float v) {
    const float w = __builtin_amdgcn_fmed3f(v, -5.f, 5.f);
    const float t = __builtin_fmaf(w * w, 0.08f, -1.f);
    float q = kGeluC[12];
#pragma unroll
    for (int i = 11; i >= 0; --i) q = __builtin_fmaf(q, t, kGeluC[i]);
    const float hv = 0.5f * v;
    return __builtin_fmaf(hv, w * q, hv);
}

}  // namespace rc
