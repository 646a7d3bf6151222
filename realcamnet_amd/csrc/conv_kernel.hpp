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
    int num_cus; int persist_ok;
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

// ==================================================================================================
// Device pieces shared by the two kernels below.
// ==================================================================================================
template <class Cfg>
struct ConvDev {
    using T = typename Cfg::elem;
    static constexpr int CK = Cfg::CK, NT = Cfg::NT, KS = Cfg::KS, UNIT = Cfg::UNIT, UPT = Cfg::UPT;
    static constexpr int NU = Cfg::NU, STEPS = Cfg::STEPS, HALO = Cfg::HALO, THH = Cfg::THH, TWH = Cfg::TWH;
    static constexpr int SPIX = Cfg::SPIX, NV = 4 * NT;
    static constexpr int VPP = UPT;                       // 16-byte vectors per pixel
    static constexpr int NPIX = THH * TWH;                // pixels in one halo tile
    static constexpr int TOTAL = NPIX * VPP;              // vectors in one halo tile
    // staging map: thread t always owns channel group v = t % VPP and walks pixels t/VPP + k*PPP, so
    // per-channel data (the CALayer gate) is loaded once per tile, and consecutive threads touch
    // consecutive 16-byte pieces (coalesced global loads, conflict-free LDS writes).
    static constexpr int PPP = kThreads / VPP;            // pixels per pass
    static constexpr int ACTIVE = PPP * VPP;              // threads that take part in staging
    static constexpr int NI = (NPIX + PPP - 1) / PPP;     // passes

    // per-lane LDS byte offset of this lane group's unit for every step of a chunk
    __device__ static __forceinline__ void unit_offsets(int q, int (&uoff)[STEPS]) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            int u = 4 * s + q;
            if (u >= NU) u = NU - 1;  // padded unit: any valid address; the operand is zeroed in mma_steps
            const int tap = u / UPT, cu = u % UPT;
            uoff[s] = ((tap / KS) * TWH + (tap % KS)) * SPIX + cu * 16;
        }
    }

    // element offset of (halo pixel pix, channel group v) in the NHWC tensor, or -1 (zero padding / idle)
    __device__ static __forceinline__ long tile_vec_offset(const ConvArgs& a, size_t img_base, int y0, int x0,
                                                           int chunk, int pix, int v, bool live, bool& center) {
        const int py = pix / TWH, px = pix - py * TWH;
        const int gy = y0 + py - HALO, gx = x0 + px - HALO;
        const int c0 = chunk * CK + v * UNIT;
        center = py >= HALO && py < HALO + kTH && px >= HALO && px < HALO + kTW;
        const bool ok = live && pix < NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && c0 < a.cin;
        return ok ? (long)((img_base + (size_t)gy * a.W + gx) * a.cin + c0) : -1;
    }

    // Issue every global load of one halo tile back-to-back into registers (vector path only).
    template <bool GATED>
    __device__ static __forceinline__ void load_tile(const ConvArgs& a, size_t img_base, int b, int y0, int x0,
                                                     int chunk, int tid, uint4 (&r0)[NI], uint4 (&r1)[GATED ? NI : 1],
                                                     float (&gv)[GATED ? UNIT : 1]) {
        const T* in0 = static_cast<const T*>(a.in0);
        const T* in1 = static_cast<const T*>(a.in1);
        const int v = tid % VPP, p0 = tid / VPP;
        const bool live = tid < ACTIVE;
        if constexpr (GATED) {
            const int c0 = chunk * CK + v * UNIT;
#pragma unroll
            for (int e = 0; e < UNIT; ++e) gv[e] = (live && c0 + e < a.cin) ? a.in_gate[(size_t)b * a.cin + c0 + e] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            bool center;
            const long off = tile_vec_offset(a, img_base, y0, x0, chunk, p0 + k * PPP, v, live, center);
            r0[k] = make_uint4(0u, 0u, 0u, 0u);
            if constexpr (GATED) r1[k] = make_uint4(0u, 0u, 0u, 0u);
            if (off >= 0) {
                r0[k] = *reinterpret_cast<const uint4*>(in0 + off);
                if constexpr (GATED) r1[k] = *reinterpret_cast<const uint4*>(in1 + off);
            }
        }
    }

    // Combine (CALayer gate + skip), optionally materialise, and write the tile to LDS.
    template <bool GATED>
    __device__ static __forceinline__ void commit_tile(const ConvArgs& a, size_t img_base, int y0, int x0, int chunk,
                                                       int tid, const uint4 (&r0)[NI], const uint4 (&r1)[GATED ? NI : 1],
                                                       const float (&gv)[GATED ? UNIT : 1], char* s_in, T* in_store) {
        const int v = tid % VPP, p0 = tid / VPP;
        const bool live = tid < ACTIVE;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int pix = p0 + k * PPP;
            if (live && pix < NPIX) {
                uint4 raw = r0[k];
                if constexpr (GATED) {
                    bool center;
                    const long off = tile_vec_offset(a, img_base, y0, x0, chunk, pix, v, live, center);
                    if (off >= 0) {
                        float f0[UNIT], f1[UNIT];
                        Vec16<T>::unpack(r0[k], f0);
                        Vec16<T>::unpack(r1[k], f1);
#pragma unroll
                        for (int e = 0; e < UNIT; ++e) f0[e] = f0[e] * gv[e] + f1[e];
                        raw = Vec16<T>::pack(f0);
                        if (in_store != nullptr && center) *reinterpret_cast<uint4*>(in_store + off) = raw;
                    }
                }
                *reinterpret_cast<uint4*>(s_in + pix * SPIX + v * 16) = raw;
            }
        }
    }

    // tiny / odd Cin (head 4->C, lens-shading 2->C): element loads straight to LDS
    __device__ static __forceinline__ void stage_tile_scalar(const ConvArgs& a, size_t img_base, int b, int y0, int x0,
                                                             int chunk, int tid, char* s_in, T* in_store) {
        const T* in0 = static_cast<const T*>(a.in0);
        const T* in1 = static_cast<const T*>(a.in1);
        for (int i = tid; i < TOTAL; i += kThreads) {
            const int pix = i / VPP, v = i - pix * VPP;
            bool center;
            const long off = tile_vec_offset(a, img_base, y0, x0, chunk, pix, v, true, center);
            const int c0 = chunk * CK + v * UNIT;
            uint4 raw = make_uint4(0u, 0u, 0u, 0u);
            if (off >= 0) {
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
                if (a.in_gate != nullptr && in_store != nullptr && center) {
#pragma unroll
                    for (int e = 0; e < UNIT; ++e)
                        if (c0 + e < a.cin) in_store[off + e] = from_f32<T>(f0[e]);
                }
            }
            *reinterpret_cast<uint4*>(s_in + pix * SPIX + v * 16) = raw;
        }
    }

    // packed weights: linear global -> LDS copy by LDS-DMA (no VGPR round trip), 1 KiB per wave-instruction.
    // The next __syncthreads() drains it (its release carries vmcnt(0)) before any wave reads them.
    __device__ static __forceinline__ void dma_weights(const char* src, char* dst, int nkb, int wave, int lane_w) {
        for (int kb = wave; kb < nkb; kb += kThreads / 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + kb * 1024 + lane_w),
                                             (__attribute__((address_space(3))) void*)(dst + kb * 1024), 16, 0, 0);
    }

    // MFMA steps [S0, S0+COUNT) of a chunk; weights for step s live at s_w + (s - W0)*NT KiB.
    template <int S0, int COUNT, int W0>
    __device__ static __forceinline__ void mma_steps(const char* s_in, const char* s_w, int lane_x, int lane_w, int q,
                                                     const int (&uoff)[STEPS], f32x4 (&acc)[4][NT]) {
#pragma unroll
        for (int sl = 0; sl < COUNT; ++sl) {
            const int s = S0 + sl;
            if (s < STEPS) {
                uint4 wf[NT], xf[4];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    wf[nt] = *reinterpret_cast<const uint4*>(s_w + ((s - W0) * NT + nt) * 1024 + lane_w);
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

    // Epilogue operands that can be fetched before the MFMA loop (so their latency hides under it).
    struct EpiPre {
        float bias_v[NV];
    };
    __device__ static __forceinline__ void epilogue_prefetch(const ConvArgs& a, int b, int jbase, EpiPre& e) {
#pragma unroll
        for (int i = 0; i < NV; ++i) e.bias_v[i] = a.bias ? a.bias[jbase + i] : 0.f;
    }

    // lane (q, n) holds, per pixel tile pt, packed couts jbase .. jbase+NV-1 of pixel (row, col0+n).
    // red: LDS scratch of 4*16*NT floats, not aliased with anything still being read.
    __device__ static __forceinline__ void epilogue(const ConvArgs& a, int b, int y0, int x0, int sp, int ct, int tid,
                                                    const EpiPre& pre, f32x4 (&acc)[4][NT], float* red) {
        const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, n = lane & 15;
        const int jbase = ct * Cfg::COUT_TILE + q * NV;
        const size_t img_base = (size_t)b * a.H * a.W;
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
                for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[pt][nt][r] + pre.bias_v[nt * 4 + r];
            if (a.film_scale != nullptr) {  // Res_GFM: (B,cout) vectors, L2-resident
#pragma unroll
                for (int e = 0; e < NV; ++e) {
                    if (jbase + e < a.cout) {
                        const float fs = a.film_scale[(size_t)b * a.cout + jbase + e];
                        const float ft = a.film_shift[(size_t)b * a.cout + jbase + e];
                        v[e] = v[e] * fs + ft + v[e];
                    }
                }
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
};

// ==================================================================================================
// Kernel 1: general form.  One block = one (spatial tile, cout tile, image); loops over Cin chunks,
// streaming packed weights through LDS G steps at a time.
// ==================================================================================================
template <class Cfg, bool GATED>
__global__ __launch_bounds__(kThreads, 2) void conv_mfma_kernel(const ConvArgs a) {
    using D = ConvDev<Cfg>;
    using T = typename Cfg::elem;
    constexpr int NT = Cfg::NT, STEPS = Cfg::STEPS, G = Cfg::G, NSUB = Cfg::NSUB, NV = 4 * NT;

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

    int uoff[STEPS];
    D::unit_offsets(q, uoff);
    const int lane_x = ((2 * wave) * D::TWH + n) * D::SPIX;  // + pixel-tile immediates in mma_steps
    const int lane_w = lane * 16;

    f32x4 acc[4][NT];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[pt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    T* in_store = (ct == 0) ? static_cast<T*>(a.in_store) : nullptr;
    const size_t img_base = (size_t)b * a.H * a.W;
    const char* wbase = static_cast<const char*>(a.wpacked) + (size_t)ct * a.n_chunks * Cfg::CHUNK_W_BYTES;

    typename D::EpiPre pre;
    D::epilogue_prefetch(a, b, ct * Cfg::COUT_TILE + q * NV, pre);

    for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
        if (chunk > 0) __syncthreads();  // all waves done reading s_in / s_w of the previous chunk
        const char* wchunk = wbase + (size_t)chunk * Cfg::CHUNK_W_BYTES;
        D::dma_weights(wchunk, s_w, (STEPS < G ? STEPS : G) * NT, wave, lane_w);  // async, overlaps the tile loads
        if (a.cin_vec_ok) {
            uint4 r0[D::NI], r1[GATED ? D::NI : 1];
            float gv[GATED ? D::UNIT : 1];
            D::template load_tile<GATED>(a, img_base, b, y0, x0, chunk, tid, r0, r1, gv);
            D::template commit_tile<GATED>(a, img_base, y0, x0, chunk, tid, r0, r1, gv, s_in, in_store);
        } else {
            D::stage_tile_scalar(a, img_base, b, y0, x0, chunk, tid, s_in, in_store);
        }
        __syncthreads();
        D::template mma_steps<0, G, 0>(s_in, s_w, lane_x, lane_w, q, uoff, acc);
        if constexpr (NSUB > 1) {
            __syncthreads();
            D::dma_weights(wchunk + (size_t)G * NT * 1024, s_w, ((STEPS - G) < G ? (STEPS - G) : G) * NT, wave, lane_w);
            __syncthreads();
            D::template mma_steps<G, G, G>(s_in, s_w, lane_x, lane_w, q, uoff, acc);
        }
        if constexpr (NSUB > 2) {
            __syncthreads();
            D::dma_weights(wchunk + (size_t)2 * G * NT * 1024, s_w, ((STEPS - 2 * G) < G ? (STEPS - 2 * G) : G) * NT, wave, lane_w);
            __syncthreads();
            D::template mma_steps<2 * G, G, 2 * G>(s_in, s_w, lane_x, lane_w, q, uoff, acc);
        }
        static_assert(NSUB <= 3, "add another weight sub-stage");
    }
    if (a.chan_sums != nullptr) __syncthreads();  // s_w (reused as reduction scratch) no longer read
    D::epilogue(a, b, y0, x0, sp, ct, tid, pre, acc, reinterpret_cast<float*>(s_w));
}

// ==================================================================================================
// Kernel 2: persistent form for single-chunk, single-cout-tile layers (Cin == CK, Cout <= 16*NT) -- the
// 48->48 convolutions that dominate the flagship net.  The whole packed weight matrix stays in LDS for
// the block's lifetime; the block walks a strided list of tiles and issues the NEXT tile's halo loads
// into registers before the MFMA loop, so HBM latency hides under compute.  Two such blocks share a CU
// and drift out of phase (one in MFMA while the other stores / stages).
// ==================================================================================================
template <class Cfg, bool GATED>
__global__ __launch_bounds__(kThreads, 2) void conv_mfma_persist_kernel(const ConvArgs a) {
    using D = ConvDev<Cfg>;
    using T = typename Cfg::elem;
    constexpr int NT = Cfg::NT, STEPS = Cfg::STEPS, NV = 4 * NT;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_in = smem;
    char* s_w = smem + Cfg::IN_BYTES;
    float* s_red = reinterpret_cast<float*>(smem + Cfg::IN_BYTES + Cfg::CHUNK_W_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, n = lane & 15;
    int uoff[STEPS];
    D::unit_offsets(q, uoff);
    const int lane_x = ((2 * wave) * D::TWH + n) * D::SPIX;
    const int lane_w = lane * 16;

    const int sp_total = a.tiles_x * a.tiles_y;
    const int n_tiles = sp_total * a.batch;
    // Tile order: at step k the grid covers the window [k*G, (k+1)*G) of consecutive tiles (same image
    // neighbourhood -> addresses spread over all HBM channels); inside the window XCD x (blocks with
    // blockIdx % 8 == x, observed placement -- speed only) takes a run of G/8 consecutive tiles, so
    // x-neighbouring halos are served by one L2.  (Giving each XCD its own image instead put all eight
    // XCDs exactly one image stride apart and serialised them on the same HBM channels: 1.5x slower.)
    const int slots = gridDim.x >> 3;                  // gridDim.x is a multiple of 8
    const int pos = (blockIdx.x & 7) * slots + (blockIdx.x >> 3);
    auto tile_of = [&](int k) -> int {                 // k-th tile of this block, or -1
        const int t = k * (int)gridDim.x + pos;
        return t < n_tiles ? t : -1;
    };

    D::dma_weights(static_cast<const char*>(a.wpacked), s_w, STEPS * NT, wave, lane_w);

    T* in_store = static_cast<T*>(a.in_store);
    uint4 r0[D::NI], r1[GATED ? D::NI : 1];
    float gv[GATED ? D::UNIT : 1];
    int k = 0;
    int tile = tile_of(0);
    if (tile >= 0 && a.cin_vec_ok) {
        const int b = tile / sp_total, sp = tile - b * sp_total;
        D::template load_tile<GATED>(a, (size_t)b * a.H * a.W, b, (sp / a.tiles_x) * kTH, (sp % a.tiles_x) * kTW, 0, tid, r0, r1, gv);
    }
    while (tile >= 0) {
        const int b = tile / sp_total, sp = tile - b * sp_total;
        const int y0 = (sp / a.tiles_x) * kTH, x0 = (sp % a.tiles_x) * kTW;
        const size_t img_base = (size_t)b * a.H * a.W;
        __syncthreads();                               // every wave finished reading s_in (previous tile)
        if (a.cin_vec_ok) D::template commit_tile<GATED>(a, img_base, y0, x0, 0, tid, r0, r1, gv, s_in, in_store);
        else D::stage_tile_scalar(a, img_base, b, y0, x0, 0, tid, s_in, in_store);
        __syncthreads();                               // tile (and, first time, the weights) visible

        const int next = tile_of(++k);
        if (next >= 0 && a.cin_vec_ok) {               // prefetch: in flight during the MFMA loop
            const int nb = next / sp_total, nsp = next - nb * sp_total;
            D::template load_tile<GATED>(a, (size_t)nb * a.H * a.W, nb, (nsp / a.tiles_x) * kTH, (nsp % a.tiles_x) * kTW, 0, tid, r0, r1, gv);
        }

        f32x4 acc[4][NT];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[pt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        D::template mma_steps<0, STEPS, 0>(s_in, s_w, lane_x, lane_w, q, uoff, acc);
        typename D::EpiPre pre;                        // bias: 48 floats, L1/L2-resident; not worth 12 VGPRs across the loop
        D::epilogue_prefetch(a, b, q * NV, pre);
        D::epilogue(a, b, y0, x0, sp, 0, tid, pre, acc, s_red);
        tile = next;
    }
}

// ---- host side: per-instantiation launcher ----------------------------------------------------------
template <class Cfg>
constexpr int persist_lds_bytes() { return Cfg::IN_BYTES + (int)Cfg::CHUNK_W_BYTES + Cfg::RED_BYTES; }

template <class Cfg, bool GATED>
int launch_conv_g(const ConvArgs& a, hipStream_t stream) {
    constexpr int P_LDS = persist_lds_bytes<Cfg>();
    constexpr bool P_OK = P_LDS <= 80 * 1024;          // two persistent blocks per CU
    if (P_OK && a.n_chunks == 1 && a.n_ct == 1 && a.persist_ok) {
        static bool attr_set = false;
        if (!attr_set) {
            RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_persist_kernel<Cfg, GATED>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS));
            attr_set = true;
        }
        const int n_tiles = a.tiles_x * a.tiles_y * a.batch;
        int grid = 2 * a.num_cus;
        if (grid > n_tiles) grid = n_tiles;
        grid = (grid + 7) / 8 * 8;
        hipLaunchKernelGGL((conv_mfma_persist_kernel<Cfg, GATED>), dim3((unsigned)grid), dim3(kThreads), P_LDS, stream, a);
        RC_HIP_CHECK(hipGetLastError());
        return RC_OK;
    }
    static bool attr_set = false;
    if (!attr_set) {
        RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<Cfg, GATED>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr_set = true;
    }
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.n_ct), (unsigned)a.batch, 1);
    hipLaunchKernelGGL((conv_mfma_kernel<Cfg, GATED>), grid, dim3(kThreads), Cfg::LDS_BYTES, stream, a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

template <class Cfg>
int launch_conv(const ConvArgs& a, hipStream_t stream) {
    // the gated form (x = in0*gate + in1) only occurs on vectorisable layers inside RCAGroups
    if (a.in_gate != nullptr && a.cin_vec_ok) return launch_conv_g<Cfg, true>(a, stream);
    return launch_conv_g<Cfg, false>(a, stream);
}

// One dispatcher per (dtype, ksize) translation unit; defined in conv_inst_*.hip
int dispatch_conv_bf16_k3(int ck, int nt, const ConvArgs& a, hipStream_t s);
int dispatch_conv_bf16_k1(int ck, int nt, const ConvArgs& a, hipStream_t s);
int dispatch_conv_f32_k3(int ck, int nt, const ConvArgs& a, hipStream_t s);
int dispatch_conv_f32_k1(int ck, int nt, const ConvArgs& a, hipStream_t s);

}  // namespace rc
