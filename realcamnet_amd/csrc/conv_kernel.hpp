// Implicit-GEMM KxK convolution on MFMA for gfx950 (wave64).  See DESIGN.md section 4.
//
// GEMM view:  D[cout][pixel] = sum_k  W[cout][k] * X[k][pixel],   k = (tap, cin)
//   A operand = packed weights  (MFMA rows    = 16 couts of one cout tile)
//   B operand = activations     (MFMA columns = 16 consecutive pixels of one image row)
// so each lane ends up holding 4 consecutive couts of ONE pixel per accumulator tile and the
// NHWC store is contiguous per lane.
//
// One 16-byte "unit" = UNIT consecutive input channels of one tap at one pixel (8 bf16 / 4 fp32).
// One "step" = 4 units (one per 16-lane group q = lane>>4):
//   bf16: 1 x v_mfma_f32_16x16x32_bf16 (K = 32 channels-of-taps)
//   fp32: 4 x v_mfma_f32_16x16x4_f32   (element j of each unit in MFMA j; K = 4 each)
// Units are numbered tap-major, so a step may span two taps (Cin = 48 -> 6 units per tap); each
// lane group simply reads its own (tap, channel) address from a per-lane table.
//
// Block = 256 threads = 4 waves; block tile = 8 rows x 32 cols of pixels x (16*NT) couts;
// wave w owns rows 2w, 2w+1 (4 pixel tiles of 16) x NT cout tiles -> 4*NT accumulator tiles.
// The input halo tile (10 x 34 pixels x CK channels) is staged once per Cin chunk in LDS with a
// pixel stride == 32 (mod 64) bytes, which makes the 16-lane-group ds_read_b128 conflict-free.
// Packed weights are streamed through LDS G steps at a time.
#pragma once
#include "common.hpp"

namespace rc {

struct ConvArgs {
    int batch, H, W, cin, cout;
    int n_chunks, n_ct, tiles_x, tiles_y;
    int cin_vec_ok;  // cin % UNIT == 0 and base pointers 16-B aligned -> vector staging
    const void* in0; const void* in1; const float* in_gate; void* in_store;
    const void* wpacked; const float* bias;
    const float* film_scale; const float* film_shift;
    int act; float act_slope;
    const void* mul_plus1; const void* residual;
    void* out; int out_mode; int out_dtype; int out_h, out_w;
    float* chan_sums; int cout_packed;
};

constexpr int kTH = 8, kTW = 32, kThreads = 256;

constexpr int pix_stride_bytes(int ck_bytes) {
    // smallest multiple of 16 >= ck_bytes that is == 32 (mod 64)
    int s = (ck_bytes + 15) / 16 * 16;
    while (s % 64 != 32) s += 16;
    return s;
}

template <typename T, int CK_, int NT_, int KS_>
struct ConvCfg {
    using elem = T;
    static constexpr int CK = CK_, NT = NT_, KS = KS_;
    static constexpr int UNIT = 16 / (int)sizeof(T);
    static_assert(CK % UNIT == 0, "CK must be a whole number of 16-byte units");
    static constexpr int UPT = CK / UNIT;         // units per tap
    static constexpr int TAPS = KS * KS;
    static constexpr int NU = TAPS * UPT;         // units per Cin chunk
    static constexpr int STEPS = (NU + 3) / 4;    // MFMA steps per Cin chunk
    static constexpr int HALO = KS / 2;
    static constexpr int THH = kTH + 2 * HALO, TWH = kTW + 2 * HALO;
    static constexpr int SPIX = pix_stride_bytes(CK * (int)sizeof(T));
    static constexpr int IN_BYTES = THH * TWH * SPIX;
    static constexpr int G_RAW = (80 * 1024 - IN_BYTES) / (NT * 1024);
    static constexpr int G_CAP = G_RAW < 1 ? 1 : (G_RAW > STEPS ? STEPS : G_RAW);
    static constexpr int NSUB = (STEPS + G_CAP - 1) / G_CAP;
    static constexpr int G = (STEPS + NSUB - 1) / NSUB;  // steps of weights resident in LDS
    static constexpr int W_BYTES = G * NT * 1024;
    static constexpr int RED_BYTES = 4 * 16 * NT * 4;     // per-wave channel partial sums
    static constexpr int LDS_BYTES = IN_BYTES + (W_BYTES > RED_BYTES ? W_BYTES : RED_BYTES);
    static constexpr int COUT_TILE = 16 * NT;
    static constexpr size_t CHUNK_W_BYTES = (size_t)STEPS * NT * 1024;  // packed weights per (ct, chunk)
};

// ---- MFMA wrappers -------------------------------------------------------------------------------
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(const uint4& w, const uint4& x, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    __device__ static __forceinline__ void run(const uint4& w, const uint4& x, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(x.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(x.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(x.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(x.w), acc, 0, 0, 0);
    }
};

// Store NV consecutive elements (given as floats) of type TO at dst (alignment ALIGN bytes).
template <typename TO, int NV>
__device__ __forceinline__ void store_row(TO* dst, const float* v) {
    constexpr int BYTES = NV * (int)sizeof(TO);
    if constexpr (sizeof(TO) == 4) {
        static_assert(NV % 4 == 0, "");
#pragma unroll
        for (int i = 0; i < NV; i += 4)
            *reinterpret_cast<float4*>(dst + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
    } else {
        if constexpr (BYTES % 16 == 0) {
#pragma unroll
            for (int i = 0; i < NV; i += 8) *reinterpret_cast<uint4*>(dst + i) = Vec16<bf16_t>::pack(v + i);
        } else {  // 8-byte pieces (NT = 3 or 1 with bf16: 24 / 8 bytes per lane)
#pragma unroll
            for (int i = 0; i < NV; i += 4) {
                uint2 p;
                p.x = Vec16<bf16_t>::rne(v[i]) | (Vec16<bf16_t>::rne(v[i + 1]) << 16);
                p.y = Vec16<bf16_t>::rne(v[i + 2]) | (Vec16<bf16_t>::rne(v[i + 3]) << 16);
                *reinterpret_cast<uint2*>(dst + i) = p;
            }
        }
    }
}

template <typename TI, int NV>
__device__ __forceinline__ void load_row(const TI* src, float* v) {
    if constexpr (sizeof(TI) == 4) {
#pragma unroll
        for (int i = 0; i < NV; i += 4) {
            const float4 t = *reinterpret_cast<const float4*>(src + i);
            v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; i += 4) {
            const uint2 p = *reinterpret_cast<const uint2*>(src + i);
            v[i] = __uint_as_float(p.x << 16); v[i + 1] = __uint_as_float(p.x & 0xffff0000u);
            v[i + 2] = __uint_as_float(p.y << 16); v[i + 3] = __uint_as_float(p.y & 0xffff0000u);
        }
    }
}

template <class Cfg>
__global__ __launch_bounds__(kThreads) void conv_mfma_kernel(const ConvArgs a) {
    using T = typename Cfg::elem;
    constexpr int CK = Cfg::CK, NT = Cfg::NT, KS = Cfg::KS, UNIT = Cfg::UNIT, UPT = Cfg::UPT;
    constexpr int NU = Cfg::NU, STEPS = Cfg::STEPS, HALO = Cfg::HALO, THH = Cfg::THH, TWH = Cfg::TWH;
    constexpr int SPIX = Cfg::SPIX, G = Cfg::G, NSUB = Cfg::NSUB, NV = 4 * NT;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_in = smem;
    char* s_w = smem + Cfg::IN_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, n = lane & 15;

    const int ct = blockIdx.x % a.n_ct;
    const int sp = blockIdx.x / a.n_ct;
    const int tx = sp % a.tiles_x, ty = sp / a.tiles_x;
    const int b = blockIdx.y;
    const int y0 = ty * kTH, x0 = tx * kTW;

    // per-lane LDS byte offset of this lane group's unit for every step of a chunk
    int uoff[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        int u = 4 * s + q;
        if (u >= NU) u = NU - 1;  // padded unit: any valid address; operand is zeroed below
        const int tap = u / UPT, cu = u % UPT;
        uoff[s] = ((tap / KS) * TWH + (tap % KS)) * SPIX + cu * 16;
    }
    const int lane_x = ((2 * wave) * TWH + n) * SPIX;  // + pixel-tile immediates below
    const int lane_w = lane * 16;

    f32x4 acc[4][NT];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[pt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const T* in0 = static_cast<const T*>(a.in0);
    const T* in1 = static_cast<const T*>(a.in1);
    T* in_store = (ct == 0) ? static_cast<T*>(a.in_store) : nullptr;
    const size_t img_base = (size_t)b * a.H * a.W;
    const char* wbase = static_cast<const char*>(a.wpacked) + (size_t)ct * a.n_chunks * Cfg::CHUNK_W_BYTES;

    for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
        if (chunk > 0) __syncthreads();  // all waves done reading s_in / s_w of the previous chunk

        // ---- stage the input halo tile of this Cin chunk -------------------------------------
        {
            constexpr int VPP = UPT;  // 16-byte vectors per pixel
            constexpr int TOTAL = THH * TWH * VPP;
            for (int i = tid; i < TOTAL; i += kThreads) {
                const int pix = i / VPP, v = i - pix * VPP;
                const int py = pix / TWH, px = pix - py * TWH;
                const int gy = y0 + py - HALO, gx = x0 + px - HALO;
                const int c0 = chunk * CK + v * UNIT;
                uint4 raw = make_uint4(0u, 0u, 0u, 0u);
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && c0 < a.cin) {
                    const size_t off = (img_base + (size_t)gy * a.W + gx) * a.cin + c0;
                    if (a.cin_vec_ok) {
                        raw = *reinterpret_cast<const uint4*>(in0 + off);
                        if (a.in_gate != nullptr) {
                            float f0[UNIT], f1[UNIT];
                            Vec16<T>::unpack(raw, f0);
                            Vec16<T>::unpack(*reinterpret_cast<const uint4*>(in1 + off), f1);
                            const float* g = a.in_gate + (size_t)b * a.cin + c0;
#pragma unroll
                            for (int e = 0; e < UNIT; ++e) f0[e] = f0[e] * g[e] + f1[e];
                            raw = Vec16<T>::pack(f0);
                            if (in_store != nullptr && py >= HALO && py < HALO + kTH && px >= HALO && px < HALO + kTW)
                                *reinterpret_cast<uint4*>(in_store + off) = raw;
                        }
                    } else {  // tiny / odd Cin (head 4->C, lens-shading 2->C): element loads
                        float f0[UNIT];
#pragma unroll
                        for (int e = 0; e < UNIT; ++e) {
                            float val = 0.f;
                            if (c0 + e < a.cin) {
                                val = to_f32(in0[off + e]);
                                if (a.in_gate != nullptr)
                                    val = val * a.in_gate[(size_t)b * a.cin + c0 + e] + to_f32(in1[off + e]);
                            }
                            f0[e] = val;
                        }
                        raw = Vec16<T>::pack(f0);
                        if (a.in_gate != nullptr && in_store != nullptr && py >= HALO && py < HALO + kTH &&
                            px >= HALO && px < HALO + kTW) {
#pragma unroll
                            for (int e = 0; e < UNIT; ++e)
                                if (c0 + e < a.cin) in_store[off + e] = from_f32<T>(f0[e]);
                        }
                    }
                }
                *reinterpret_cast<uint4*>(s_in + pix * SPIX + v * 16) = raw;
            }
        }

        const char* wchunk = wbase + (size_t)chunk * Cfg::CHUNK_W_BYTES;
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int s_begin = sub * G;
            const int s_count = (STEPS - s_begin) < G ? (STEPS - s_begin) : G;
            if (sub > 0) __syncthreads();  // previous sub-stage's weights fully consumed
            {
                const uint4* src = reinterpret_cast<const uint4*>(wchunk + (size_t)s_begin * NT * 1024);
                uint4* dst = reinterpret_cast<uint4*>(s_w);
                const int nvec = s_count * NT * 64;
                for (int i = tid; i < nvec; i += kThreads) dst[i] = src[i];
            }
            __syncthreads();

#pragma unroll
            for (int sl = 0; sl < G; ++sl) {
                const int s = s_begin + sl;
                if (s < STEPS) {
                    uint4 wf[NT], xf[4];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        wf[nt] = *reinterpret_cast<const uint4*>(s_w + (sl * NT + nt) * 1024 + lane_w);
                    const char* xp = s_in + lane_x + uoff[s];
#pragma unroll
                    for (int pt = 0; pt < 4; ++pt)
                        xf[pt] = *reinterpret_cast<const uint4*>(xp + ((pt >> 1) * TWH + (pt & 1) * 16) * SPIX);
                    if constexpr (NU % 4 != 0) {
                        if (s == STEPS - 1 && 4 * s + q >= NU) {
#pragma unroll
                            for (int pt = 0; pt < 4; ++pt) xf[pt] = make_uint4(0u, 0u, 0u, 0u);
                        }
                    }
#pragma unroll
                    for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) Mma<T>::run(wf[nt], xf[pt], acc[pt][nt]);
                }
            }
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    // lane (q, n) holds, per pixel tile pt, packed couts jbase .. jbase+NV-1 of pixel (row, col0+n)
    const int jbase = ct * Cfg::COUT_TILE + q * NV;
    float bias_v[NV], fs[NV], ft[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        bias_v[e] = a.bias ? a.bias[jbase + e] : 0.f;
        fs[e] = 0.f; ft[e] = 0.f;
    }
    if (a.film_scale != nullptr) {
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            if (jbase + e < a.cout) {
                fs[e] = a.film_scale[(size_t)b * a.cout + jbase + e];
                ft[e] = a.film_shift[(size_t)b * a.cout + jbase + e];
            }
        }
    }
    float csum[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) csum[e] = 0.f;

#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        const int gy = y0 + 2 * wave + (pt >> 1);
        const int gx = x0 + (pt & 1) * 16 + n;
        const bool valid = gy < a.H && gx < a.W;
        float v[NV];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[pt][nt][r] + bias_v[nt * 4 + r];
        if (a.film_scale != nullptr) {
#pragma unroll
            for (int e = 0; e < NV; ++e) v[e] = v[e] * fs[e] + ft[e] + v[e];
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) v[e] = apply_act(v[e], a.act, a.act_slope);
        if (!valid) continue;
        const size_t pix = img_base + (size_t)gy * a.W + gx;
        if (a.mul_plus1 != nullptr) {
            float m[NV];
            load_row<T, NV>(static_cast<const T*>(a.mul_plus1) + pix * a.cout + jbase, m);
#pragma unroll
            for (int e = 0; e < NV; ++e) v[e] = v[e] * (m[e] + 1.f);
        }
        if (a.residual != nullptr) {
            float m[NV];
            load_row<T, NV>(static_cast<const T*>(a.residual) + pix * a.cout + jbase, m);
#pragma unroll
            for (int e = 0; e < NV; ++e) v[e] += m[e];
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) csum[e] += v[e];

        if (a.out_mode == RC_OUT_NHWC) {
            if (jbase + NV <= a.cout) {
                store_row<T, NV>(static_cast<T*>(a.out) + pix * a.cout + jbase, v);
            } else {
#pragma unroll
                for (int e = 0; e < NV; ++e)
                    if (jbase + e < a.cout) static_cast<T*>(a.out)[pix * a.cout + jbase + e] = from_f32<T>(v[e]);
            }
        } else if (a.out_mode == RC_OUT_PIXEL_SHUFFLE2) {
            // packed cout tile `ct` holds out channels ct*NV .. ct*NV+NV-1 for sub-pixel q
            const int cps = a.cout >> 2;
            const size_t opix = ((size_t)b * (2 * a.H) + (2 * gy + (q >> 1))) * (2 * a.W) + (2 * gx + (q & 1));
            store_row<T, NV>(static_cast<T*>(a.out) + opix * cps + ct * NV, v);
        } else {  // RC_OUT_NCHW, cropped
            if (gy < a.out_h && gx < a.out_w) {
#pragma unroll
                for (int e = 0; e < NV; ++e) {
                    const int co = jbase + e;
                    if (co < a.cout) {
                        const size_t o = (((size_t)b * a.cout + co) * a.out_h + gy) * a.out_w + gx;
                        if (a.out_dtype == RC_F32) static_cast<float*>(a.out)[o] = v[e];
                        else static_cast<bf16_t*>(a.out)[o] = from_f32<bf16_t>(v[e]);
                    }
                }
            }
        }
    }

    if (a.chan_sums != nullptr) {  // uniform branch
        // reduce over the 16 pixels of the lane group (lanes sharing q), then over the 4 waves
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            float s = csum[e];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
            csum[e] = s;
        }
        __syncthreads();  // s_w no longer read by any wave
        float* red = reinterpret_cast<float*>(s_w);
        if (n == 0) {
#pragma unroll
            for (int e = 0; e < NV; ++e) red[wave * Cfg::COUT_TILE + q * NV + e] = csum[e];
        }
        __syncthreads();
        if (tid < Cfg::COUT_TILE) {
            const float s = ((red[tid] + red[Cfg::COUT_TILE + tid]) + red[2 * Cfg::COUT_TILE + tid]) + red[3 * Cfg::COUT_TILE + tid];
            const int co = ct * Cfg::COUT_TILE + tid;
            if (co < a.cout)
                a.chan_sums[((size_t)b * (a.tiles_x * a.tiles_y) + sp) * a.cout + co] = s;
        }
    }
}

// ---- host side: per-instantiation launcher ----------------------------------------------------------
template <class Cfg>
int launch_conv(const ConvArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<Cfg>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr_set = true;
    }
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.n_ct), (unsigned)a.batch, 1);
    hipLaunchKernelGGL(conv_mfma_kernel<Cfg>, grid, dim3(kThreads), Cfg::LDS_BYTES, stream, a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

// One dispatcher per (dtype, ksize) translation unit; defined in conv_inst_*.hip
int dispatch_conv_bf16_k3(int ck, int nt, const ConvArgs& a, hipStream_t s);
int dispatch_conv_bf16_k1(int ck, int nt, const ConvArgs& a, hipStream_t s);
int dispatch_conv_f32_k3(int ck, int nt, const ConvArgs& a, hipStream_t s);
int dispatch_conv_f32_k1(int ck, int nt, const ConvArgs& a, hipStream_t s);

}  // namespace rc
