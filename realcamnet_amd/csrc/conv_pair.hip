// Fused pair of 3x3 convolutions, 48 -> 48 -> 48 channels, bf16 storage / fp32 accumulate (gfx950).
//
//     mid = act1(conv1(x))            act1 = ReLU (RCAB, networks.py:296-311) or FiLM + LeakyReLU (Res_GFM, LiteISP.py:537-559)
//     out = conv2(mid) [+ residual]   (+ per-wave channel sums for CALayer's global mean)
//
// Unfused, `mid` costs one HBM write and one HBM read of a full-resolution 48-channel map, and the 48->48 layers
// of the flagship net already run AT the measured HBM copy rate, so the only way to make
// them faster is to move fewer bytes.  Here a block keeps `mid` for its tile in LDS:
//
//   input halo tile 12 x 36 px  --conv1 (MFMA)-->  mid tile 10 x 34 px (LDS, bf16)  --conv2 (MFMA)-->  8 x 32 px out
//
// conv1 is recomputed on the 1-pixel ring (340 instead of 256 pixels: 22 instead of 16 MFMA pixel tiles), i.e. the
// pair costs 1.19x the MFMA work of the two separate layers and ~half their HBM traffic.  mid pixels outside the
// image are forced to zero (conv2 zero-pads ITS input, it does not see conv1 of the padding).
//
// One 768-thread block per CU (LDS: both packed weight matrices 84 KB + input tile 40.5 KB + mid tile 32 KB): 8 compute
// waves (two per SIMD: one wave alone only reaches 59 % of the MFMA rate) + 4 loader waves, persistent over a band-major
// XCD-aware tile list like the single-layer kernels.  Per tile:
//   phase 1  compute: conv1 on 22 linear 16-pixel tiles of the mid grid (3,3,3,3,3,3,2,2 per wave) -> LDS
//            loaders: tile k+1's registers -> r*gate + skip, skip-tensor store
//   barrier  (mid complete, input tile free)
//   phase 2  compute: conv2, wave w = output row w (2 pixel tiles) -> epilogue -> HBM
//            loaders: combined tile k+1 -> LDS; issue tile k+2's loads
//   barrier  (input tile k+1 visible, mid free)
// The MFMA / LDS-read interleave, weight fragment layout (rc_conv_pack_weights, CK = 48, NT = 3) and unit map are
// those of conv_kernel.hpp.
#include "conv_kernel.hpp"

#include <mutex>
#include <string>

namespace rc {
namespace pair {

constexpr int C = 48, NT = 3, NV = 12, SPIX = 96, STEPS = 14, ES = 2;
constexpr int OTH = 8, OTW = 32;                  // output tile
constexpr int MH = OTH + 2, MW = OTW + 2;         // mid tile (conv2's halo tile)
constexpr int IH = OTH + 4, IW = OTW + 4;         // input tile (conv1's halo tile of the mid tile)
constexpr int NMID = MH * MW, NIN = IH * IW;      // 340, 432 pixels
constexpr int N_MID_TILES = (NMID + 15) / 16;     // 22 MFMA pixel tiles
constexpr int W_BYTES = STEPS * NT * 1024;        // one packed 48x(9*48) weight matrix
constexpr int OFF_W1 = 0, OFF_W2 = W_BYTES, OFF_BIAS = 2 * W_BYTES, OFF_IN = OFF_BIAS + 2 * C * 4;
constexpr int OFF_MID = OFF_IN + NIN * SPIX, LDS_BYTES = OFF_MID + NMID * SPIX;
static_assert(LDS_BYTES <= 160 * 1024, "pair kernel LDS budget");
constexpr int WAVES = 8, LOADERS = 256, THREADS = WAVES * 64 + LOADERS;   // 8 compute waves (two per SIMD) + 4 loader waves
constexpr int VPP = 6, PPP = LOADERS / VPP, ACTIVE = PPP * VPP, NI = (NIN + PPP - 1) / PPP;   // staging map over the loader threads, as ConvDev

enum { E1_RELU = 0, E1_FILM_LEAKY = 1 };
enum { E2_PLAIN = 0, E2_SUMS = 1, E2_RES = 2 };

struct PairArgs {
    int batch, H, W;
    const bf16_t* in0; const bf16_t* in1; const float* in_gate; bf16_t* in_store;
    const void* w1; const float* b1; const float* film_scale; const float* film_shift; float slope;
    const void* w2; const float* b2;
    const bf16_t* residual; bf16_t* out; float* chan_sums;
    int tiles_x, tiles_y;
    TileDecode td;
    long long* dbg;                // optional phase-timing buffer (rc_debug_set_ptr("conv_phase_timing")), normally NULL
};

// ---- MFMA over one 3x3x48 chunk for NPT pixel tiles whose LDS pixel rows are P pixels apart ------------
template <int P, int NPT>
struct PairMma {
    static constexpr int FR = NT + NPT, FM = NT * NPT;
    struct Lane { int a, b, c; };
    __device__ static __forceinline__ Lane lane_consts(int q) {
        Lane l;
        l.a = q * 16;
        l.b = (q >> 1) * SPIX + (4 + (q & 1)) * 16;
        l.c = (q >> 1) * (P - 2) * SPIX + (4 + (q & 1)) * 16;
        return l;
    }
    __device__ static constexpr int tap_off(int tap) { return ((tap / 3) * P + (tap % 3)) * SPIX; }
    __device__ static __forceinline__ int step_off(int s, const Lane& l) {   // s is a constant after inlining
        if (s < 9) return l.a + tap_off(s);
        const int t0 = 2 * (s - 9);
        const bool wrap = (t0 + 1 < 9) && ((t0 + 1) % 3 == 0);
        return (wrap ? l.c : l.b) + tap_off(t0);
    }
    template <int I>
    __device__ static __forceinline__ void load_item(int s, const char* smem, const int (&xb)[NPT], int wb, const Lane& l,
                                                     uint4 (&wf)[NT], uint4 (&xf)[NPT]) {
        if constexpr (I < NT) wf[I] = *reinterpret_cast<const uint4*>(smem + wb + (s * NT + I) * 1024);
        else xf[I - NT] = *reinterpret_cast<const uint4*>(smem + xb[I - NT] + step_off(s, l));
    }
    template <int I>
    __device__ static __forceinline__ void load_all(int s, const char* smem, const int (&xb)[NPT], int wb, const Lane& l,
                                                    uint4 (&wf)[NT], uint4 (&xf)[NPT]) {
        if constexpr (I < FR) {
            load_item<I>(s, smem, xb, wb, l, wf, xf);
            load_all<I + 1>(s, smem, xb, wb, l, wf, xf);
        }
    }
    __device__ static __forceinline__ void zero_pad(int s, int q, uint4 (&xf)[NPT]) {
        if (s == STEPS - 1) {                    // last paired step: tap 9 does not exist
            if (q >= 2) {
#pragma unroll
                for (int i = 0; i < NPT; ++i) xf[i] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    template <int I, bool NEXT>
    __device__ static __forceinline__ void step(int s, const char* smem, const int (&xb)[NPT], int wb, const Lane& l,
                                                const uint4 (&wf)[NT], const uint4 (&xf)[NPT], uint4 (&wfn)[NT], uint4 (&xfn)[NPT],
                                                f32x4 (&acc)[NPT][NT]) {
        if constexpr (I < FR) {
            if constexpr (NEXT) load_item<I>(s + 1, smem, xb, wb, l, wfn, xfn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = (I * FM) / FR; j < ((I + 1) * FM) / FR; ++j) Mma<bf16_t>::run(wf[j % NT], xf[j / NT], acc[j / NT][j % NT]);
            __builtin_amdgcn_sched_barrier(0);
            step<I + 1, NEXT>(s, smem, xb, wb, l, wf, xf, wfn, xfn, acc);
        }
    }
    __device__ static __forceinline__ void run(const char* smem, const int (&xb)[NPT], int wb, int q, f32x4 (&acc)[NPT][NT]) {
        const Lane l = lane_consts(q);
        uint4 wfa[NT], xfa[NPT], wfb[NT], xfb[NPT];
        load_all<0>(0, smem, xb, wb, l, wfa, xfa);
#pragma unroll
        for (int s = 0; s < STEPS; s += 2) {     // STEPS is even
            step<0, true>(s, smem, xb, wb, l, wfa, xfa, wfb, xfb, acc);
            zero_pad(s + 1, q, xfb);
            if (s + 2 < STEPS) {
                step<0, true>(s + 1, smem, xb, wb, l, wfb, xfb, wfa, xfa, acc);
                zero_pad(s + 2, q, xfa);
            } else {
                step<0, false>(s + 1, smem, xb, wb, l, wfb, xfb, wfa, xfa, acc);
            }
        }
    }
};

// ---- input staging (the ConvDev scheme with a 12 x 36 halo tile and 512 threads) -----------------------
struct TileSrc {
    __amdgpu_buffer_rsrc_t r0, r1, rst;
    int gy0, gx0, soff;
    bool interior;
};
__device__ __forceinline__ TileSrc tile_src(const PairArgs& a, int b, int y0, int x0) {
    const size_t img = (size_t)a.H * a.W * C;
    const unsigned bytes = (unsigned)(img * ES);
    TileSrc t;
    t.r0 = make_rsrc(a.in0 + (size_t)b * img, bytes);
    t.r1 = make_rsrc(a.in1 ? a.in1 + (size_t)b * img : nullptr, a.in1 ? bytes : 0u);
    t.rst = make_rsrc(a.in_store ? a.in_store + (size_t)b * img : nullptr, a.in_store ? bytes : 0u);
    t.gy0 = y0 - 2; t.gx0 = x0 - 2;
    t.soff = (t.gy0 * a.W + t.gx0) * C * ES;
    t.interior = t.gy0 >= 0 && t.gx0 >= 0 && t.gy0 + IH <= a.H && t.gx0 + IW <= a.W;
    return t;
}
struct TileOffs { int o[NI]; int ctr[NI]; };
__device__ __forceinline__ void tile_offsets(const PairArgs& a, int tid, TileOffs& t) {
    const int v = tid % VPP, p0 = tid / VPP;
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int pix = p0 + k * PPP;
        const int py = pix / IW, px = pix - py * IW;
        const bool center = py >= 2 && py < 2 + OTH && px >= 2 && px < 2 + OTW;
        t.o[k] = (tid < ACTIVE && pix < NIN) ? ((py * a.W + px) * C + v * 8) * ES : kOOB;
        t.ctr[k] = center ? t.o[k] : kOOB;
    }
}
__device__ __forceinline__ int border_off(const PairArgs& a, const TileSrc& t, int tid, int k, bool& center) {
    const int v = tid % VPP, pix = tid / VPP + k * PPP;
    const int py = pix / IW, px = pix - py * IW;
    const int gy = t.gy0 + py, gx = t.gx0 + px;
    center = py >= 2 && py < 2 + OTH && px >= 2 && px < 2 + OTW;
    const bool ok = tid < ACTIVE && pix < NIN && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    return ok ? ((gy * a.W + gx) * C + v * 8) * ES : kOOB;
}
template <bool GATED>
__device__ __forceinline__ void load_interior(const TileSrc& t, const TileOffs& to, uint4 (&r0)[NI], uint4 (&r1)[GATED ? NI : 1]) {
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        r0[k] = buf_load16(t.r0, to.o[k], t.soff);
        if constexpr (GATED) r1[k] = buf_load16(t.r1, to.o[k], t.soff);
    }
}
template <bool GATED>
__device__ __forceinline__ void load_border(const PairArgs& a, const TileSrc& t, int tid, uint4 (&r0)[NI], uint4 (&r1)[GATED ? NI : 1]) {
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        bool center;
        const int off = border_off(a, t, tid, k, center);
        r0[k] = buf_load16(t.r0, off, 0);
        if constexpr (GATED) r1[k] = buf_load16(t.r1, off, 0);
    }
}
template <bool GATED>
__device__ __forceinline__ void load_gate(const PairArgs& a, int b, int tid, float (&gv)[GATED ? 8 : 1]) {
    if constexpr (GATED) {
        const int c0 = (tid % VPP) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[e] = tid < ACTIVE ? a.in_gate[(size_t)b * C + c0 + e] : 0.f;
    }
}
// registers -> combined tile (r*gate + skip, materialised for the centre pixels); leaves the 16-byte pieces in r0
template <bool GATED>
__device__ __forceinline__ void combine(const PairArgs& a, const TileSrc& t, const TileOffs& to, int tid, uint4 (&r0)[NI],
                                        const uint4 (&r1)[GATED ? NI : 1], const float (&gv)[GATED ? 8 : 1]) {
    if constexpr (GATED) {
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            float f0[8], f1[8];
            Vec16<bf16_t>::unpack(r0[k], f0);
            Vec16<bf16_t>::unpack(r1[k], f1);
#pragma unroll
            for (int e = 0; e < 8; ++e) f0[e] = f0[e] * gv[e] + f1[e];       // zero-filled lanes stay 0
            r0[k] = Vec16<bf16_t>::pack(f0);
            if (t.interior) {
                buf_store16(t.rst, to.ctr[k], t.soff, r0[k]);                 // rst has 0 records if in_store == NULL
            } else {
                bool center;
                const int off = border_off(a, t, tid, k, center);
                buf_store16(t.rst, center ? off : kOOB, 0, r0[k]);
            }
        }
    }
}
__device__ __forceinline__ void write_tile(int tid, const uint4 (&r0)[NI], char* smem) {
    const int v = tid % VPP, p0 = tid / VPP;
    char* dst = smem + OFF_IN + p0 * SPIX + v * 16;
#pragma unroll
    for (int k = 0; k < NI; ++k)
        if (tid < ACTIVE && p0 + k * PPP < NIN) *reinterpret_cast<uint4*>(dst + k * PPP * SPIX) = r0[k];
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float s) {
    return s + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_sum16(float s) {
    s = dpp_add<0xB1>(s); s = dpp_add<0x4E>(s); s = dpp_add<0x141>(s); s = dpp_add<0x140>(s);
    return s;
}

// ---- the kernel ---------------------------------------------------------------------------------------
template <bool GATED, int E1, int E2>
__global__ __launch_bounds__(THREADS) void conv_pair_kernel(const PairArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_bias = reinterpret_cast<float*>(smem + OFF_BIAS);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, n = lane & 15;

    // both weight matrices + biases, once per block (1 KiB per wave-instruction by LDS-DMA)
    for (int kb = w; kb < 2 * STEPS * NT; kb += THREADS / 64) {
        const char* src = kb < STEPS * NT ? static_cast<const char*>(a.w1) + kb * 1024 : static_cast<const char*>(a.w2) + (kb - STEPS * NT) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 16),
                                         (__attribute__((address_space(3))) void*)(smem + kb * 1024), 16, 0, 0);
    }
    if (tid < 2 * C) s_bias[tid] = tid < C ? (a.b1 ? a.b1[tid] : 0.f) : (a.b2 ? a.b2[tid - C] : 0.f);

    // phase-1 geometry: this wave's mid pixel tiles t = w, w+8, w+16 (linear 16-pixel runs of the 10 x 34 grid)
    const int npt1 = w + 16 < N_MID_TILES ? 3 : 2;                 // wave-uniform
    int xb1[3], mid_dst[3], my[3], mx[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = 16 * (w + 8 * i) + n;
        const int jc = j < NMID ? j : NMID - 1;                    // dummy lanes read a valid pixel, write nothing
        my[i] = jc / MW; mx[i] = jc - my[i] * MW;
        xb1[i] = OFF_IN + (my[i] * IW + mx[i]) * SPIX;
        mid_dst[i] = j < NMID ? OFF_MID + j * SPIX + q * (NV * ES) : -1;
    }
    int xb2[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) xb2[h] = OFF_MID + (w * MW + 16 * h + n) * SPIX;
    const int wb1 = OFF_W1 + lane * 16, wb2 = OFF_W2 + lane * 16;

    const int sp_total = a.tiles_x * a.tiles_y;
    const int n_tiles = sp_total * a.batch;
    const int slots = gridDim.x >> 3;                              // gridDim.x is a multiple of 8
    const int pos = (blockIdx.x & 7) * slots + (blockIdx.x >> 3);  // XCD x takes a run of consecutive (band-major) tiles
    const int stride = (int)gridDim.x;
    const int my_tiles = pos < n_tiles ? (n_tiles - pos + stride - 1) / stride : 0;

    auto decode = [&](int tile, int& b, int& sp, int& y0, int& x0) {
        b = magic_div(tile, a.td.sp_total);
        int ty, tx;
        band_decode(tile - b * sp_total, a.tiles_x, a.tiles_y, a.td, ty, tx);
        sp = ty * a.tiles_x + tx; y0 = ty * OTH; x0 = tx * OTW;
    };

    if (w >= WAVES) {
        // ---------------------------------------------------------------- loader waves 8-11
        // tile k+1's input sits in registers while tile k is computed: the gate math (r*g + skip) and the skip-tensor store
        // run beside phase 1, the LDS write and the next tile's loads beside phase 2 -- none of it on the compute waves
        const int ltid = tid - WAVES * 64;
        TileOffs to;
        tile_offsets(a, ltid, to);
        uint4 r0[NI], r1[GATED ? NI : 1];
        float gv[GATED ? 8 : 1];
        TileSrc ts;
        int b, sp, y0, x0;
        auto fetch = [&](int tile, bool sync_border) {              // loads of one tile -> registers (border tiles only when asked)
            decode(tile, b, sp, y0, x0);
            ts = tile_src(a, b, y0, x0);
            load_gate<GATED>(a, b, ltid, gv);
            if (ts.interior) load_interior<GATED>(ts, to, r0, r1);
            else if (sync_border) load_border<GATED>(a, ts, ltid, r0, r1);
        };
        if (my_tiles > 0) {
            fetch(pos, true);
            combine<GATED>(a, ts, to, ltid, r0, r1, gv);
            write_tile(ltid, r0, smem);
        }
        if (my_tiles > 1) fetch(pos + stride, false);
        __syncthreads();                                           // weights, biases, tile 0 visible
        for (int k = 0; k < my_tiles; ++k) {
            if (k + 1 < my_tiles) {                                // beside phase 1 of tile k
                if (!ts.interior) load_border<GATED>(a, ts, ltid, r0, r1);          // border tiles are not prefetched
                combine<GATED>(a, ts, to, ltid, r0, r1, gv);
            }
            __syncthreads();                                       // A: the input tile is free
            if (k + 1 < my_tiles) {                                // beside phase 2 of tile k
                write_tile(ltid, r0, smem);
                if (k + 2 < my_tiles) fetch(pos + (k + 2) * stride, false);
            }
            __syncthreads();                                       // B: input tile k+1 visible
        }
        return;
    }

    // -------------------------------------------------------------------- compute waves 0-7 (two per SIMD)
    int cb = 0, csp = 0, cy0 = 0, cx0 = 0;                         // the tile being computed
    if (my_tiles > 0) decode(pos, cb, csp, cy0, cx0);
    __syncthreads();                                               // weights, biases, tile 0 visible

    const float inf = __builtin_inff();
    const bool rec = a.dbg != nullptr && blockIdx.x == 8 && w == 0 && lane == 0;
    for (int k = 0; k < my_tiles; ++k) {
        const long long t0 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
        long long t1 = 0, t2 = 0, t3 = 0, t5 = 0, t6 = 0;
        // ------------------------------------------------ phase 1: mid = act1(conv1(x)) on the 10 x 34 ring tile
        {
            f32x4 acc[3][NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 t4 = *reinterpret_cast<const float4*>(s_bias + q * NV + nt * 4);
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[i][nt] = f32x4{t4.x, t4.y, t4.z, t4.w};
            }
            if (npt1 == 3) {
                PairMma<IW, 3>::run(smem, xb1, wb1, q, acc);
            } else {
                int xb[2] = {xb1[0], xb1[1]};
                f32x4 acc2[2][NT];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc2[i][nt] = acc[i][nt];
                PairMma<IW, 2>::run(smem, xb, wb1, q, acc2);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[i][nt] = acc2[i][nt];
            }
            t1 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            float fs[NV], ft[NV];
            if constexpr (E1 == E1_FILM_LEAKY) {
#pragma unroll
                for (int e = 0; e < NV; e += 4) {
                    const float4 s4 = *reinterpret_cast<const float4*>(a.film_scale + (size_t)cb * C + q * NV + e);
                    const float4 t4 = *reinterpret_cast<const float4*>(a.film_shift + (size_t)cb * C + q * NV + e);
                    fs[e] = s4.x; fs[e + 1] = s4.y; fs[e + 2] = s4.z; fs[e + 3] = s4.w;
                    ft[e] = t4.x; ft[e + 1] = t4.y; ft[e + 2] = t4.z; ft[e + 3] = t4.w;
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (i < npt1) {                                    // wave-uniform
                    float v[NV];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[i][nt][r];
                    if constexpr (E1 == E1_FILM_LEAKY) {
#pragma unroll
                        for (int e = 0; e < NV; ++e) {
                            v[e] = v[e] * fs[e] + ft[e] + v[e];
                            v[e] = __builtin_amdgcn_fmed3f(v[e], v[e] * a.slope, inf);   // 0 <= slope <= 1 (host)
                        }
                    }
                    // conv2 zero-pads its input: mid pixels outside the image are 0, not conv1 of the padding
                    const int gy = cy0 - 1 + my[i], gx = cx0 - 1 + mx[i];
                    const bool inside = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                    unsigned wd[NV / 2];
#pragma unroll
                    for (int e = 0; e < NV / 2; ++e) {
                        wd[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                        if constexpr (E1 == E1_RELU) {
                            typedef short s16x2 __attribute__((ext_vector_type(2)));
                            const s16x2 z = {0, 0};
                            wd[e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, wd[e]), z));
                        }
                        wd[e] = inside ? wd[e] : 0u;
                    }
                    if (mid_dst[i] >= 0) {
#pragma unroll
                        for (int e = 0; e < NV / 2; e += 2) *reinterpret_cast<uint2*>(smem + mid_dst[i] + 4 * e) = make_uint2(wd[e], wd[e + 1]);
                    }
                }
            }
        }
        t2 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
        __syncthreads();                                           // A: mid tile complete
        t3 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
        // ------------------------------------------------ phase 2: out = conv2(mid) (+ residual, + channel sums)
        {
            f32x4 acc[2][NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 t4 = *reinterpret_cast<const float4*>(s_bias + C + q * NV + nt * 4);
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[h][nt] = f32x4{t4.x, t4.y, t4.z, t4.w};
            }
            PairMma<MW, 2>::run(smem, xb2, wb2, q, acc);
            t5 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;

            const size_t img = (size_t)a.H * a.W * C;
            const unsigned img_bytes = (unsigned)(img * ES);
            const __amdgpu_buffer_rsrc_t r_out = make_rsrc(a.out + (size_t)cb * img, img_bytes);
            const int gy = cy0 + w;
            float csum[NV];
#pragma unroll
            for (int e = 0; e < NV; ++e) csum[e] = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int gx = cx0 + 16 * h + n;
                const bool valid = gy < a.H && gx < a.W;
                const int off = valid ? ((gy * a.W + gx) * C + q * NV) * ES : kOOB;
                float v[NV];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[h][nt][r];
                if constexpr (E2 == E2_RES) {
                    float m[NV];
                    buf_load_row<bf16_t, NV>(make_rsrc(a.residual + (size_t)cb * img, img_bytes), off, m);
#pragma unroll
                    for (int e = 0; e < NV; ++e) v[e] += m[e];
                }
                if constexpr (E2 == E2_SUMS) {
#pragma unroll
                    for (int e = 0; e < NV; ++e) csum[e] += valid ? v[e] : 0.f;
                }
                buf_store_row<bf16_t, NV, false>(r_out, off, v);
            }
            if constexpr (E2 == E2_SUMS) {
                // one partial per wave (slot = 8*tile + wave); rc_ca_gate folds them in fixed order
#pragma unroll
                for (int e = 0; e < NV; ++e) csum[e] = row_sum16(csum[e]);
                if (n == 0) {
                    float* dst = a.chan_sums + (((size_t)cb * sp_total + csp) * WAVES + w) * C + q * NV;
#pragma unroll
                    for (int e = 0; e < NV; ++e) dst[e] = csum[e];
                }
            }
        }
        if (k + 1 < my_tiles) decode(pos + (k + 1) * stride, cb, csp, cy0, cx0);
        t6 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
        __syncthreads();                                           // B: input tile k+1 visible; mid tile free
        if (rec && k < 60) {
            long long* d = a.dbg + 8 * k;
            d[0] = t1 - t0; d[1] = t2 - t1; d[2] = t3 - t2; d[3] = 0; d[4] = t5 - t3; d[5] = t6 - t5;
            d[6] = (long long)__builtin_amdgcn_s_memtime() - t6;
        }
    }
}

template <bool GATED, int E1, int E2>
int launch(const PairArgs& a, int num_cus, hipStream_t stream) {
    static PerDeviceFlag attr_set;
    if (!attr_set.test_and_set()) {
        RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pair_kernel<GATED, E1, E2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    }
    const int n_tiles = a.tiles_x * a.tiles_y * a.batch;
    int grid = num_cus < n_tiles ? num_cus : n_tiles;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL((conv_pair_kernel<GATED, E1, E2>), dim3((unsigned)grid), dim3(THREADS), LDS_BYTES, stream, a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

}  // namespace pair

// ==================================================================================================================================
// pair2: the same fused pair, re-cut after round 3's measurements:
//   * weights live in REGISTERS (one packed 48 x 432 matrix = 42 x 16 bytes per lane = 168 VGPRs), so the LDS only carries activations:
//     1/3 ds_read_b128 per MFMA instead of 0.58 -- the LDS-bound phases of the first pair kernel were reading both weight matrices
//     through the LDS for every 2-3 pixel tiles
//   * the two convolutions are two TEAMS of four waves, one wave of each per SIMD (a lone wave issuing v_mfma_f32_16x16x32_bf16 only
//     reaches 59 % of the matrix rate, two reach 91 %): team A runs conv1 of tile k+1 while team B runs conv2 + the epilogue of tile k,
//     so B's store issue (~250-650 cycles per buffer_store per wave, tools/ubench/store_issue.hip) sits beside A's MFMAs
//   * no loader waves (8 waves = 256 registers each; with 12 the cap is 168): all eight waves stage the next tile in two batches of three
//     16-byte pieces per thread, loaded half a step before they are combined (gate * r + skip), materialised and written to LDS
//   * input and mid tiles double-buffered, ONE barrier per tile.
// Arithmetic, rounding points and the order of every accumulation are those of two rc_conv2d launches: bit-identical (tests).
// ==================================================================================================================================
namespace pair2 {
using pair::PairArgs;
using pair::TileSrc;
using pair::tile_src;
using pair::row_sum16;
using pair::E1_RELU; using pair::E1_FILM_LEAKY; using pair::E2_PLAIN; using pair::E2_SUMS; using pair::E2_RES;

constexpr int C = 48, NT = 3, NV = 12, SPIX = 96, STEPS = 14, ES = 2;
constexpr int OTH = 8, OTW = 32, MH = OTH + 2, MW = OTW + 2, IH = OTH + 4, IW = OTW + 4;
constexpr int NMID = MH * MW, NIN = IH * IW, N_MID_TILES = (NMID + 15) / 16;
constexpr int IN_BYTES = NIN * SPIX, MID_BYTES = NMID * SPIX;
constexpr int kMaxBatchGates = 48;                               // every image's CALayer gate sits in LDS (48 x 192 B): a commit reads it from there
constexpr int OFF_IN = 0, OFF_MID = 2 * IN_BYTES, OFF_BIAS = OFF_MID + 2 * MID_BYTES, OFF_GATE = OFF_BIAS + 2 * C * 4, LDS_BYTES = OFF_GATE + kMaxBatchGates * C * 4;
static_assert(LDS_BYTES <= 160 * 1024, "pair2 LDS budget");
constexpr int THREADS = 512, VPP = 6, PPP = THREADS / VPP, ACTIVE = PPP * VPP, NI = (NIN + PPP - 1) / PPP, NB = NI / 2;
static_assert(NI == 6 && NB == 3, "staging map");

// byte offset of this thread's piece k of the staged tile inside the image, or kOOB; center = a pixel the tile's output covers
__device__ __forceinline__ int piece_off(const PairArgs& a, const TileSrc& t, int tid, int k, bool& center, bool& live) {
    const int v = tid % VPP, pix = tid / VPP + k * PPP;
    const int py = pix / IW, px = pix - py * IW;
    const int gy = t.gy0 + py, gx = t.gx0 + px;
    live = tid < ACTIVE && pix < NIN;
    center = py >= 2 && py < 2 + OTH && px >= 2 && px < 2 + OTW;
    const bool ok = live && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    return ok ? ((gy * a.W + gx) * C + v * 8) * ES : kOOB;
}

// one 3x3x48 chunk for TWO pixel tiles with the packed weights in registers; xb0 / xb1 = LDS byte address of each tile's top-left tap
template <int P>
struct Mma2 {
    using Base = pair::PairMma<P, 2>;
    __device__ static __forceinline__ void run(const char* smem, int xb0, int xb1, const uint4 (&w)[STEPS * NT], int q, f32x4 (&acc)[2][NT]) {
        const typename Base::Lane l = Base::lane_consts(q);
        uint4 xa[2], xb[2];
        xa[0] = *reinterpret_cast<const uint4*>(smem + xb0 + Base::step_off(0, l));
        xa[1] = *reinterpret_cast<const uint4*>(smem + xb1 + Base::step_off(0, l));
#pragma unroll
        for (int s = 0; s < STEPS; s += 2) {
            one<true>(s, smem, xb0, xb1, l, w, q, xa, xb, acc);
            if (s + 2 < STEPS) one<true>(s + 1, smem, xb0, xb1, l, w, q, xb, xa, acc);
            else one<false>(s + 1, smem, xb0, xb1, l, w, q, xb, xa, acc);
        }
    }
    template <bool NEXT>
    __device__ static __forceinline__ void one(int s, const char* smem, int xb0, int xb1, const typename Base::Lane& l, const uint4 (&w)[STEPS * NT], int q,
                                               uint4 (&x)[2], uint4 (&xn)[2], f32x4 (&acc)[2][NT]) {
        if (s == STEPS - 1 && q >= 2) { x[0] = make_uint4(0u, 0u, 0u, 0u); x[1] = x[0]; }      // last paired step: tap 9 does not exist
        if constexpr (NEXT) xn[0] = *reinterpret_cast<const uint4*>(smem + xb0 + Base::step_off(s + 1, l));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Mma<bf16_t>::run(w[s * NT + nt], x[0], acc[0][nt]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NEXT) xn[1] = *reinterpret_cast<const uint4*>(smem + xb1 + Base::step_off(s + 1, l));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Mma<bf16_t>::run(w[s * NT + nt], x[1], acc[1][nt]);
        __builtin_amdgcn_sched_barrier(0);
    }
};

// Staging of one 12 x 36 halo tile by all eight waves: the tile is 2592 16-byte pieces, piece P = (halo pixel P / 6, channel group P % 6),
// dense in LDS (a pixel is exactly six pieces), so DMA instruction i covers pieces [64 i, 64 i + 64) and lands them with no register and no
// ds_write (buffer_load ... lds; out-of-bounds lanes write zeros = the convolution's zero padding).  Instructions i = w, w + 8, ... belong
// to wave w (both teams stage: team B's store-heavy epilogue and team A's larger MFMA share leave them about the same slack).
constexpr int N_PIECES = NIN * VPP, N_DMA = (N_PIECES + 63) / 64, kDmaPerWave = (N_DMA + 7) / 8, kRounds = 3;   // per wave: 6 instructions = 2 batches of 3
static_assert(N_PIECES == 2592 && N_DMA == 41 && kDmaPerWave == 6, "staging map");

__device__ __forceinline__ int piece_goff(const PairArgs& a, const TileSrc& t, int P, bool& center) {
    const int pix = P / VPP, v = P - pix * VPP;
    const int py = pix / IW, px = pix - py * IW;
    const int gy = t.gy0 + py, gx = t.gx0 + px;
    center = py >= 2 && py < 2 + OTH && px >= 2 && px < 2 + OTW;
    const bool ok = P < N_PIECES && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    return ok ? ((gy * a.W + gx) * C + v * 8) * ES : kOOB;
}

template <bool GATED, int E1, int E2>
__global__ __launch_bounds__(THREADS) void conv_pair2_kernel(const PairArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_bias = reinterpret_cast<float*>(smem + OFF_BIAS);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool teamB = w8 >= 4;
    const int wi = w8 & 3;                                           // wave index inside the team
    const int q = lane >> 4, n = lane & 15;

    // this team's packed weight matrix -> registers (once per block)
    uint4 wreg[STEPS * NT];
    {
        const char* src = static_cast<const char*>(teamB ? a.w2 : a.w1) + lane * 16;
#pragma unroll
        for (int i = 0; i < STEPS * NT; ++i) wreg[i] = *reinterpret_cast<const uint4*>(src + i * 1024);
    }
    if (tid < 2 * C) s_bias[tid] = tid < C ? (a.b1 ? a.b1[tid] : 0.f) : (a.b2 ? a.b2[tid - C] : 0.f);
    if constexpr (GATED) {
        float* s_gate = reinterpret_cast<float*>(smem + OFF_GATE);
        for (int i = tid; i < a.batch * C; i += THREADS) s_gate[i] = a.in_gate[i];
    }
    __syncthreads();                                                 // biases and gates visible (the prologue's combine reads the gates)

    const int sp_total = a.tiles_x * a.tiles_y;
    const int n_tiles = sp_total * a.batch;
    const int slots = gridDim.x >> 3;
    const int pos = (blockIdx.x & 7) * slots + (blockIdx.x >> 3);
    const int stride = (int)gridDim.x;
    const int my_tiles = pos < n_tiles ? (n_tiles - pos + stride - 1) / stride : 0;

    auto decode = [&](int tile, int& b, int& sp, int& y0, int& x0) {
        b = magic_div(tile, a.td.sp_total);
        int ty, tx;
        band_decode(tile - b * sp_total, a.tiles_x, a.tiles_y, a.td, ty, tx);
        sp = ty * a.tiles_x + tx; y0 = ty * OTH; x0 = tx * OTW;
    };

    // ---- staging (all eight waves).  Rounds [R0, R0 + 3) of this wave's DMA instructions for the tile described by ts / st_b into input buffer
    // `buf`.  Gated: the r pieces go by DMA, the skip pieces into registers; stage_finish waits for both, combines r * gate + skip in place and
    // materialises the centre pixels (a store instruction none of whose pieces is a centre pixel is skipped: a dropped store still costs
    // its ~250 issue cycles).  Three rounds (12 registers) are in flight at a time.
    TileSrc ts;
    int st_b = 0;
    uint4 sk[GATED ? kRounds : 1];
    auto stage_begin = [&](int tile_idx) {
        int sp, y0, x0;
        decode(pos + tile_idx * stride, st_b, sp, y0, x0);
        ts = tile_src(a, st_b, y0, x0);
    };
    // interior tiles (95 % at 4K): a piece's offset is a per-lane LAUNCH constant + the tile's scalar offset (in the instruction's soffset), so a
    // DMA costs ~4 instructions instead of the ~40 of piece_goff (measured 400-650 cycles per DMA instruction on the general path: with
    // six per wave and tile that was as much issue time as the wave's MFMAs)
    int lc[kDmaPerWave];
    unsigned center_mask = 0;
#pragma unroll
    for (int r = 0; r < kDmaPerWave; ++r) {
        const int P = (w8 + 8 * r) * 64 + lane;
        const int pix = P / VPP, v = P - pix * VPP;
        const int py = pix / IW, px = pix - py * IW;
        lc[r] = P < N_PIECES ? ((py * a.W + px) * C + v * 8) * ES : kOOB;
        if (P < N_PIECES && py >= 2 && py < 2 + OTH && px >= 2 && px < 2 + OTW) center_mask |= 1u << r;
    }
    auto stage_issue = [&](int r0, int buf) {
        if (ts.interior) {
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
                const int inst = w8 + 8 * (r0 + r);                  // wave-uniform
                if (inst < N_DMA) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ts.r0, (__attribute__((address_space(3))) void*)(smem + OFF_IN + buf * IN_BYTES + inst * 1024),
                                                             16, lc[r0 + r], ts.soff, 0, 0);
                    if constexpr (GATED) sk[r] = buf_load16(ts.r1, lc[r0 + r], ts.soff);
                }
            }
            return;
        }
        // border tiles: bounds-checked geometry.  (It is per-lane loop-invariant: hoisted out of the tile loop it was SPILLED, and every DMA
        // then waited vmcnt(0) for its scratch reload -- i.e. for the previous DMA to land.  Laundering the lane id keeps it where it is used.)
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int inst = w8 + 8 * (r0 + r);
            if (inst < N_DMA) {
                bool center;
                const int off = piece_goff(a, ts, inst * 64 + ln, center);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ts.r0, (__attribute__((address_space(3))) void*)(smem + OFF_IN + buf * IN_BYTES + inst * 1024),
                                                         16, off, 0, 0, 0);
                if constexpr (GATED) sk[r] = buf_load16(ts.r1, off, 0);
            }
        }
    };
    auto stage_finish = [&](int r0, int buf) {                       // gated only: combine what stage_issue(r0) fetched
        if constexpr (GATED) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): this wave's DMA pieces have landed (lgkm / exp untouched)
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
                const int inst = w8 + 8 * (r0 + r);
                if (inst < N_DMA) {
                    const int P = inst * 64 + ln;
                    bool center;
                    int off, soff = 0;
                    if (ts.interior) { off = lc[r0 + r]; soff = ts.soff; center = (center_mask >> (r0 + r)) & 1u; }
                    else off = piece_goff(a, ts, P, center);
                    char* lp = smem + OFF_IN + buf * IN_BYTES + P * 16;
                    const float4* g4 = reinterpret_cast<const float4*>(smem + OFF_GATE + (st_b * C + (P % VPP) * 8) * 4);
                    const float4 ga = g4[0], gb = g4[1];
                    const float gv[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
                    float f0[8], f1[8];
                    Vec16<bf16_t>::unpack(*reinterpret_cast<const uint4*>(lp), f0);
                    Vec16<bf16_t>::unpack(sk[r], f1);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f0[e] = f0[e] * gv[e] + f1[e];
                    const uint4 v = Vec16<bf16_t>::pack(f0);
                    if (P < N_PIECES) *reinterpret_cast<uint4*>(lp) = v;
                    // pieces [64 inst, 64 inst + 64) = halo pixels [10.67 inst, ...): the first centre pixel is 2*36 + 2 = 74, the last 9*36 + 33 = 357
                    const bool any_center = inst * 64 + 63 >= 74 * VPP && inst * 64 < 358 * VPP;     // wave-uniform
                    if (any_center) buf_store16(ts.rst, center ? off : kOOB, soff, v);                // rst has 0 records if in_store == NULL
                }
            }
        }
    };

    // prologue: tile 0
    if (my_tiles > 0) {
        stage_begin(0);
        stage_issue(0, 0); stage_finish(0, 0);
        stage_issue(kRounds, 0); stage_finish(kRounds, 0);
    }
    __syncthreads();                                                 // tile 0 visible (a barrier on a path that carries LDS-DMA waits vmcnt(0))

    const float inf = __builtin_inff();
    const bool rec = a.dbg != nullptr && blockIdx.x == 8 && lane == 0 && (w8 == 0 || w8 == 4);
    // step k:  A: stage tile k+1 -> in[(k+1)&1]; conv1(tile k): in[k&1] -> mid[k&1]        B: conv2(tile k-1) from mid[(k-1)&1] -> HBM
    if (!teamB) {
        for (int k = 0; k <= my_tiles; ++k) {
            const bool staging = k + 1 < my_tiles;
            const int sbuf = (k + 1) & 1;
            const long long t0 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            long long tp[3] = {0, 0, 0};
            if (staging) { stage_begin(k + 1); stage_issue(0, sbuf); }
            if (k < my_tiles) {
                int cb, csp, cy0, cx0;
                decode(pos + k * stride, cb, csp, cy0, cx0);
                const char* in = smem + OFF_IN + (k & 1) * IN_BYTES;
                char* mid = smem + OFF_MID + (k & 1) * MID_BYTES;
#pragma unroll 1
                for (int p = 0; p < 3; ++p) {                        // passes of two 16-pixel tiles: t = wi + 8p, wi + 8p + 4
                    auto mid_pixel = [&](int i, int& j, int& my, int& mx) {   // lanes / tiles past the grid read a valid pixel, write nothing
                        j = 16 * (wi + 8 * p + 4 * i) + n;
                        const int jc = j < NMID ? j : NMID - 1;
                        my = jc / MW; mx = jc - my * MW;
                    };
                    int xb[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        int j, my, mx;
                        mid_pixel(i, j, my, mx);
                        xb[i] = (my * IW + mx) * SPIX;
                    }
                    f32x4 acc[2][NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float4 t4 = *reinterpret_cast<const float4*>(s_bias + q * NV + nt * 4);
                        acc[0][nt] = f32x4{t4.x, t4.y, t4.z, t4.w}; acc[1][nt] = acc[0][nt];
                    }
                    Mma2<IW>::run(in, xb[0], xb[1], wreg, q, acc);
                    float fs[NV], ft[NV];
                    if constexpr (E1 == E1_FILM_LEAKY) {             // per pass from L2 (24 registers would not survive the MFMA pass)
#pragma unroll
                        for (int e = 0; e < NV; e += 4) {
                            const float4 s4 = *reinterpret_cast<const float4*>(a.film_scale + (size_t)cb * C + q * NV + e);
                            const float4 t4 = *reinterpret_cast<const float4*>(a.film_shift + (size_t)cb * C + q * NV + e);
                            fs[e] = s4.x; fs[e + 1] = s4.y; fs[e + 2] = s4.z; fs[e + 3] = s4.w;
                            ft[e] = t4.x; ft[e + 1] = t4.y; ft[e + 2] = t4.z; ft[e + 3] = t4.w;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float v[NV];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[i][nt][r];
                        if constexpr (E1 == E1_FILM_LEAKY) {
#pragma unroll
                            for (int e = 0; e < NV; ++e) {
                                v[e] = v[e] * fs[e] + ft[e] + v[e];
                                v[e] = __builtin_amdgcn_fmed3f(v[e], v[e] * a.slope, inf);
                            }
                        }
                        int j, my, mx;                                       // recomputed here rather than kept live across the MFMA pass
                        mid_pixel(i, j, my, mx);
                        const int md = j < NMID ? j * SPIX + q * (NV * ES) : -1;
                        const int gy = cy0 - 1 + my, gx = cx0 - 1 + mx;        // conv2 zero-pads its input: mid outside the image is 0
                        const bool inside = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                        unsigned wd[NV / 2];
#pragma unroll
                        for (int e = 0; e < NV / 2; ++e) {
                            wd[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                            if constexpr (E1 == E1_RELU) {
                                typedef short s16x2 __attribute__((ext_vector_type(2)));
                                const s16x2 z = {0, 0};
                                wd[e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, wd[e]), z));
                            }
                            wd[e] = inside ? wd[e] : 0u;
                        }
                        if (md >= 0) {
#pragma unroll
                            for (int e = 0; e < NV / 2; e += 2) *reinterpret_cast<uint2*>(mid + md + 4 * e) = make_uint2(wd[e], wd[e + 1]);
                        }
                    }
                    tp[p] = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
                    if (staging) {                                   // the rounds issued before this pass have had it to land
                        if (p == 0) { stage_finish(0, sbuf); stage_issue(kRounds, sbuf); }
                        if (p == 1) stage_finish(kRounds, sbuf);
                    }
                }
            } else if (staging) {
                stage_finish(0, sbuf); stage_issue(kRounds, sbuf); stage_finish(kRounds, sbuf);
            }
            const long long t3 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            __syncthreads();                                         // (this path carries LDS-DMA: hipcc waits vmcnt(0) here -- wanted, the tile must have landed)
            if (rec && k < 60) {
                long long* d = a.dbg + 8 * k;
                d[0] = tp[0] - t0; d[1] = tp[1] - tp[0]; d[2] = tp[2] - tp[1]; d[3] = t3 - tp[2]; d[4] = (long long)__builtin_amdgcn_s_memtime() - t3;
            }
        }
    } else {
        for (int k = 0; k <= my_tiles; ++k) {
            const long long t0 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            long long t1 = 0, t2 = 0;
            const bool staging = k + 1 < my_tiles;
            const int sbuf = (k + 1) & 1;
            if (staging) { stage_begin(k + 1); stage_issue(0, sbuf); }
            if (k >= 1) {
                int cb, csp, cy0, cx0;
                decode(pos + (k - 1) * stride, cb, csp, cy0, cx0);
                const char* mid = smem + OFF_MID + ((k - 1) & 1) * MID_BYTES;
                const size_t img = (size_t)a.H * a.W * C;
                const unsigned img_bytes = (unsigned)(img * ES);
                const __amdgpu_buffer_rsrc_t r_out = make_rsrc(a.out + (size_t)cb * img, img_bytes);
#pragma unroll 1
                for (int p = 0; p < 2; ++p) {                        // one output row per pass
                    const int row = 2 * wi + p;
                    const int xb0 = (row * MW + n) * SPIX, xb1 = (row * MW + 16 + n) * SPIX;
                    f32x4 acc[2][NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float4 t4 = *reinterpret_cast<const float4*>(s_bias + C + q * NV + nt * 4);
                        acc[0][nt] = f32x4{t4.x, t4.y, t4.z, t4.w}; acc[1][nt] = acc[0][nt];
                    }
                    Mma2<MW>::run(mid, xb0, xb1, wreg, q, acc);
                    if (p == 0) t1 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
                    if (staging) {                                   // before this row's stores: the finish's vmcnt(0) would wait for them too
                        if (p == 0) { stage_finish(0, sbuf); stage_issue(kRounds, sbuf); }
                        else stage_finish(kRounds, sbuf);
                    }
                    const int gy = cy0 + row;
                    float csum[NV];
#pragma unroll
                    for (int e = 0; e < NV; ++e) csum[e] = 0.f;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int gx = cx0 + 16 * h + n;
                        const bool valid = gy < a.H && gx < a.W;
                        const int off = valid ? ((gy * a.W + gx) * C + q * NV) * ES : kOOB;
                        float v[NV];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[h][nt][r];
                        if constexpr (E2 == E2_RES) {
                            float m[NV];
                            buf_load_row<bf16_t, NV>(make_rsrc(a.residual + (size_t)cb * img, img_bytes), off, m);
#pragma unroll
                            for (int e = 0; e < NV; ++e) v[e] += m[e];
                        }
                        if constexpr (E2 == E2_SUMS) {
#pragma unroll
                            for (int e = 0; e < NV; ++e) csum[e] += valid ? v[e] : 0.f;
                        }
                        buf_store_row<bf16_t, NV, false>(r_out, off, v);
                    }
                    if constexpr (E2 == E2_SUMS) {                   // one partial per output row (slot = 8 * tile + row), as the first pair kernel
#pragma unroll
                        for (int e = 0; e < NV; ++e) csum[e] = row_sum16(csum[e]);
                        if (n == 0) {
                            float* dst = a.chan_sums + (((size_t)cb * sp_total + csp) * 8 + row) * C + q * NV;
#pragma unroll
                            for (int e = 0; e < NV; ++e) dst[e] = csum[e];
                        }
                    }
                    if (p == 0) t2 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
                }
            } else if (staging) {
                stage_finish(0, sbuf); stage_issue(kRounds, sbuf); stage_finish(kRounds, sbuf);
            }
            const long long t3 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            __syncthreads();                                         // (waits vmcnt(0): this wave's DMA pieces, and the acks of its stores)
            if (rec && k < 60) {
                long long* d = a.dbg + 512 + 8 * k;
                d[0] = t1 - t0; d[1] = t2 - t1; d[2] = t3 - t2; d[3] = (long long)__builtin_amdgcn_s_memtime() - t3;
            }
        }
    }
}

template <bool GATED, int E1, int E2>
int launch(const PairArgs& a, int num_cus, hipStream_t stream) {
    static PerDeviceFlag attr_set;
    if (!attr_set.test_and_set()) {
        RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pair2_kernel<GATED, E1, E2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    }
    const int n_tiles = a.tiles_x * a.tiles_y * a.batch;
    int grid = num_cus < n_tiles ? num_cus : n_tiles;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL((conv_pair2_kernel<GATED, E1, E2>), dim3((unsigned)grid), dim3(THREADS), LDS_BYTES, stream, a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

}  // namespace pair2
long long* conv_dbg_ptr();
int conv_pair_impl();
void conv_prof_begin(double flops, hipStream_t stream, void** token, double bytes, int cin, int cout, int ksize);
void conv_prof_end(void* token, hipStream_t stream);
}  // namespace rc

using namespace rc;

extern "C" {

size_t rc_conv_pair_desc_size(void) { return sizeof(rc_conv_pair_desc); }

int rc_conv_pair_sum_slots(int height, int width) { return pair::WAVES * ceil_div(height, pair::OTH) * ceil_div(width, pair::OTW); }

int rc_conv_pair(const rc_conv_pair_desc* d, void* stream_) {
    RC_REQUIRE(d != nullptr, "rc_conv_pair: null desc");
    RC_REQUIRE(d->channels == pair::C && d->dtype == RC_BF16, "rc_conv_pair: built for 48 channels, bf16 (use two rc_conv2d calls otherwise)");
    RC_REQUIRE(d->batch >= 1 && d->height >= 1 && d->width >= 1, "rc_conv_pair: empty tensor");
    RC_REQUIRE(d->in0 && d->w1 && d->w2 && d->out, "rc_conv_pair: null in0/w1/w2/out");
    RC_REQUIRE(d->act1 == RC_ACT_RELU || d->act1 == RC_ACT_LEAKY, "rc_conv_pair: act1 must be ReLU, or LeakyReLU with FiLM");
    if (d->act1 == RC_ACT_LEAKY) {
        RC_REQUIRE(d->film_scale && d->film_shift, "rc_conv_pair: LeakyReLU form needs film_scale/film_shift (Res_GFM)");
        RC_REQUIRE(d->act1_slope >= 0.f && d->act1_slope <= 1.f, "rc_conv_pair: slope must be in [0, 1]");
        RC_REQUIRE(d->in_gate == nullptr && d->chan_sums == nullptr, "rc_conv_pair: FiLM form takes no gate / sums");
    } else {
        RC_REQUIRE(!d->film_scale && !d->film_shift, "rc_conv_pair: FiLM needs act1 = LeakyReLU");
        RC_REQUIRE(d->residual == nullptr, "rc_conv_pair: the ReLU form takes no residual");
    }
    if (d->in_gate) RC_REQUIRE(d->in1 != nullptr, "rc_conv_pair: in_gate needs in1 (the skip tensor)");
    auto aligned16 = [](const void* q) { return q == nullptr || reinterpret_cast<uintptr_t>(q) % 16 == 0; };
    RC_REQUIRE(aligned16(d->in0) && aligned16(d->in1) && aligned16(d->in_store) && aligned16(d->out) && aligned16(d->residual) &&
               aligned16(d->film_scale) && aligned16(d->film_shift), "rc_conv_pair: tensors must be 16-byte aligned");
    RC_REQUIRE((double)d->height * d->width * pair::C * 4.0 < 2147483647.0, "rc_conv_pair: one image must be < 2 GiB");
    RC_REQUIRE(d->batch * (double)ceil_div(d->height, pair::OTH) * ceil_div(d->width, pair::OTW) < (double)(1 << 24), "rc_conv_pair: too many tiles");
    if (conv_pair_impl() == 2 && d->in_gate) RC_REQUIRE(d->batch <= pair2::kMaxBatchGates, "rc_conv_pair: the gated form keeps every image's gate in LDS: batch <= 48");

    pair::PairArgs a{};
    a.batch = d->batch; a.H = d->height; a.W = d->width;
    a.in0 = static_cast<const bf16_t*>(d->in0); a.in1 = static_cast<const bf16_t*>(d->in1);
    a.in_gate = d->in_gate; a.in_store = static_cast<bf16_t*>(d->in_store);
    a.w1 = d->w1; a.b1 = d->b1; a.film_scale = d->film_scale; a.film_shift = d->film_shift; a.slope = d->act1_slope;
    a.w2 = d->w2; a.b2 = d->b2;
    a.residual = static_cast<const bf16_t*>(d->residual); a.out = static_cast<bf16_t*>(d->out); a.chan_sums = d->chan_sums;
    a.tiles_x = ceil_div(d->width, pair::OTW); a.tiles_y = ceil_div(d->height, pair::OTH);
    a.td = make_tile_decode(a.tiles_x, a.tiles_y, kBandRows);
    a.dbg = conv_dbg_ptr();
    const int num_cus = device_cu_count();          // per device (common.hpp)
    hipStream_t stream = as_stream(stream_);
    void* tok = nullptr;
    conv_prof_begin(2.0 * 2.0 * d->batch * d->height * d->width * 9.0 * pair::C * pair::C, stream, &tok, 2.0 * 2.0 * d->batch * d->height * d->width * pair::C, pair::C, pair::C, 3);
    int rcode;
    const bool gated = d->in_gate != nullptr;
#define RC_PAIR_DISPATCH(NS)                                                                                                                        \
    if (d->act1 == RC_ACT_LEAKY)                                                                                                                    \
        rcode = d->residual ? NS::launch<false, pair::E1_FILM_LEAKY, pair::E2_RES>(a, num_cus, stream)                                              \
                            : NS::launch<false, pair::E1_FILM_LEAKY, pair::E2_PLAIN>(a, num_cus, stream);                                           \
    else if (d->chan_sums)                                                                                                                          \
        rcode = gated ? NS::launch<true, pair::E1_RELU, pair::E2_SUMS>(a, num_cus, stream) : NS::launch<false, pair::E1_RELU, pair::E2_SUMS>(a, num_cus, stream); \
    else                                                                                                                                            \
        rcode = gated ? NS::launch<true, pair::E1_RELU, pair::E2_PLAIN>(a, num_cus, stream) : NS::launch<false, pair::E1_RELU, pair::E2_PLAIN>(a, num_cus, stream);
    const int impl = conv_pair_impl() != 0 ? conv_pair_impl() : (gated ? 1 : 2);      // measured r3: gated 1.94 vs 2.6 ms, sums 1.64 vs 1.53 ms
    if (impl == 1) { RC_PAIR_DISPATCH(pair) } else { RC_PAIR_DISPATCH(pair2) }
#undef RC_PAIR_DISPATCH
    conv_prof_end(tok, stream);
    return rcode;
}

}  // extern "C"
