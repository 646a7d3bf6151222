// 3x3 convolution on v_mfma_f32_32x32x16_bf16 for the layers whose weights do not fit LDS at once (Cin = 128/192/512 in
// 32-channel chunks; 48 -> 192 in one 48-channel chunk).  See DESIGN.md section 4.8.
//
// Same GEMM view as conv_kernel.hpp (A = packed weights, B = activations, D[cout][pixel]) and the same producer/consumer
// stage protocol as its conv_mfma_wsm_kernel (weights single-buffered by halves, input tile double-buffered, two barriers per
// stage), but the compute side is built around the 32x32 tile:
//   * a K16 step = one tap x 16 input channels: lane (n = lane&31, h = lane>>5) supplies pixel n / cout row n, channels 8h..8h+7,
//     i.e. ONE 16-byte LDS read per operand fragment, and no padded steps for any Cin that is a multiple of 16
//   * a compute wave owns PT image rows x 32 pixels x (32*NT32) couts = PT*NT32 accumulator tiles of 16 VGPRs.  With PT = 4 and
//     NT32 = 2 (128 px x 64 couts, 128 accumulator registers) a step is 6 ds_read_b128 for 8 MFMAs of 32 cycles: 0.75 reads per
//     32 matrix-pipe cycles where the 16x16x32 kernel's 64 px x 64 cout tile needs 1.0 -- the instruction shape itself does not
//     change the LDS bytes per FLOP (those depend on the wave tile only), the larger tile it makes affordable does: a 32x32x16
//     MFMA occupies the pipe for 32 cycles, so ONE wave per SIMD can keep it busy (<= 5 other instructions per MFMA), which frees
//     the registers of the second compute wave for the larger tile.
//   * D layout: lane (n, h) holds rows (i&3) + 8*(i>>2) + 4h, i = 0..15, of column n.  The packer permutes the cout rows so that
//     these are 16 CONSECUTIVE channels (32 bytes of one pixel's NHWC record, or of one sub-pixel's record under PixelShuffle).
//   * LDS pixel stride = odd multiple of 16 bytes (80 for 32 channels, 112 for 48): the lane groups a ds_read_b128 is served in
//     ({0-3,12-15,20-27}, ...) then cover all 64 banks exactly once for 32 consecutive pixels at one channel offset.
#pragma once
#include "conv_kernel.hpp"

namespace rc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int spix32_bytes(int ck_bytes) {
    int s = (ck_bytes + 15) / 16 * 16;
    while ((s / 16) % 2 == 0) s += 16;
    return s;
}

// staging-side view: the member names ConvDev<> reads, so its halo-tile loaders (interior / border, gated, materialising) are reused
template <int CK_, int TH_>
struct C32Stage {
    using elem = bf16_t;
    static constexpr int CK = CK_, NT = 1, KS = 3, TH = TH_;
    static constexpr int UNIT = 8, UPT = CK / 8, TAPS = 9, STEPS = unit_map_steps(UPT, TAPS), HALO = 1;
    static constexpr int THH = TH + 2, TWH = kTW + 2;
    static constexpr int SPIX = spix32_bytes(CK * 2);
    static constexpr int IN_BYTES = THH * TWH * SPIX;
    static constexpr int COUT_TILE = 16;
};

template <int CK_, int TH_, int NCW_, int NT32_>
struct C32Cfg {
    using Stage = C32Stage<CK_, TH_>;
    static constexpr int CK = CK_, TH = TH_, NCW = NCW_, NT32 = NT32_;
    static_assert(CK % 16 == 0 && TH % NCW == 0, "conv32: chunk width / wave split");
    static constexpr int SPT = CK / 16;                   // K16 steps per tap
    static constexpr int STEPS = 9 * SPT;
    static constexpr int SA = (STEPS + 1) / 2;            // steps in the first weight half
    static constexpr int PT = TH / NCW;                   // image rows (32-pixel tiles) per compute wave
    static constexpr int COUT_TILE = 32 * NT32;
    static constexpr int W_BYTES = STEPS * NT32 * 1024;   // packed weights per (cout tile, chunk)
    static constexpr int WA = SA * NT32 * 1024, WB = W_BYTES - WA;
    static constexpr int COMPUTE = NCW * 64, THREADS = COMPUTE + kThreads;
    static constexpr int LDS_BYTES = W_BYTES + kPersistMaxCout * 4 + 2 * Stage::IN_BYTES;
    static constexpr int FR = NT32 + PT, FM = NT32 * PT;  // fragment reads / MFMAs per step
    static constexpr int NPEND = PT * NT32 * 2;           // 16-byte stores of one wave tile
    static_assert(LDS_BYTES <= 160 * 1024, "conv32: LDS budget");
};

// packed cout index -> conv output channel for the 32-row tiles (shared with the host packer)
//   NHWC:          identity
//   PixelShuffle2: cout tile ct = (out-channel block ct>>1, sub-row i = ct&1); inside it row tile t, lane half h, register e:
//                  out channel (ct>>1)*16*NT32 + 16 t + e of sub-pixel (i, h)  ->  conv channel 4*oc + 2 i + h
__host__ __device__ inline int c32_packed_to_cout(int nt32, int out_mode, int j) {
    if (out_mode != RC_OUT_PIXEL_SHUFFLE2) return j;
    const int tile = 32 * nt32;
    const int ct = j / tile, within = j % tile;
    const int t = within / 32, c = within % 32, h = c / 16, e = c % 16;
    const int oc = (ct >> 1) * 16 * nt32 + 16 * t + e;
    return 4 * oc + 2 * (ct & 1) + h;
}

template <class K>
struct C32Dev {
    using St = typename K::Stage;
    using SD = ConvDev<St>;
    static constexpr int PT = K::PT, NT32 = K::NT32, FR = K::FR, FM = K::FM, SPT = K::SPT;
    static constexpr int TWH = St::TWH, SPIX = St::SPIX;

    struct Pend {                                   // one wave tile's packed output, stored during the NEXT stage's MFMA loop
        __amdgpu_buffer_rsrc_t r;
        int off[PT];
        int tstride;
        uint4 v[K::NPEND];
        bool has;
    };

    __device__ static constexpr int b_imm(int s, int r) {
        const int tap = s / SPT, j = s % SPT;
        return ((r + tap / 3) * TWH + tap % 3) * SPIX + 2 * j * 16;
    }
    template <int I>
    __device__ static __forceinline__ void load_item(int s, const char* in_lane, const char* w_lane, uint4 (&wf)[NT32], uint4 (&xf)[PT]) {
        if constexpr (I < NT32) wf[I] = *reinterpret_cast<const uint4*>(w_lane + (s * NT32 + I) * 1024);
        else xf[I - NT32] = *reinterpret_cast<const uint4*>(in_lane + b_imm(s, I - NT32));
    }
    __device__ static __forceinline__ void load_all(int s, const char* in_lane, const char* w_lane, uint4 (&wf)[NT32], uint4 (&xf)[PT]) {
#pragma unroll
        for (int t = 0; t < NT32; ++t) wf[t] = *reinterpret_cast<const uint4*>(w_lane + (s * NT32 + t) * 1024);
#pragma unroll
        for (int r = 0; r < PT; ++r) xf[r] = *reinterpret_cast<const uint4*>(in_lane + b_imm(s, r));
    }
    __device__ static __forceinline__ void mma(int m, const uint4 (&wf)[NT32], const uint4 (&xf)[PT], f32x16 (&acc)[PT][NT32]) {
        const int r = m / NT32, t = m % NT32;
        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[t]), __builtin_bit_cast(bf16x8, xf[r]), acc[r][t], 0, 0, 0);
    }
    template <int I, bool NEXT>
    __device__ static __forceinline__ void step_il(int s, const char* in_lane, const char* w_lane, const uint4 (&wf)[NT32], const uint4 (&xf)[PT],
                                                   uint4 (&wfn)[NT32], uint4 (&xfn)[PT], f32x16 (&acc)[PT][NT32]) {
        if constexpr (I < FR) {
            if constexpr (NEXT) load_item<I>(s + 1, in_lane, w_lane, wfn, xfn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = (I * FM) / FR; m < ((I + 1) * FM) / FR; ++m) mma(m, wf, xf, acc);
            __builtin_amdgcn_sched_barrier(0);
            step_il<I + 1, NEXT>(s, in_lane, w_lane, wf, xf, wfn, xfn, acc);
        }
    }
    __device__ static __forceinline__ void pend_store(Pend& p, int k) {
        const int r = k / (NT32 * 2), rem = k % (NT32 * 2);
        buf_store16(p.r, p.off[r] + (rem & 1) * 16, (rem >> 1) * p.tstride, p.v[k]);
    }
    // MFMA steps [S0, S1) of a chunk; with DEFER the previous stage's stores are issued one per step from step 0 on
    template <int S0, int S1, bool DEFER>
    __device__ static __forceinline__ void mma_steps(const char* in_lane, const char* w_lane, f32x16 (&acc)[PT][NT32], Pend& pend) {
        if constexpr (S0 < S1) {
            uint4 wfa[NT32], xfa[PT], wfb[NT32], xfb[PT];
            load_all(S0, in_lane, w_lane, wfa, xfa);
#pragma unroll
            for (int s = S0; s < S1; s += 2) {
                if (s + 1 < S1) step_il<0, true>(s, in_lane, w_lane, wfa, xfa, wfb, xfb, acc);
                else step_il<0, false>(s, in_lane, w_lane, wfa, xfa, wfb, xfb, acc);
                if constexpr (DEFER) {
                    if (S0 == 0 && s < K::NPEND && pend.has) pend_store(pend, s);
                }
                if (s + 1 < S1) {
                    if (s + 2 < S1) step_il<0, true>(s + 1, in_lane, w_lane, wfb, xfb, wfa, xfa, acc);
                    else step_il<0, false>(s + 1, in_lane, w_lane, wfb, xfb, wfa, xfa, acc);
                    if constexpr (DEFER) {
                        if (S0 == 0 && s + 1 < K::NPEND && pend.has) pend_store(pend, s + 1);
                    }
                }
            }
        }
    }

    // ---- epilogue: every optional operand is a uniform run-time branch (these layers run hundreds of MFMAs per output value) ----
    // lane (n, h) of compute wave `wave` holds, for row r and row tile t, channels cbase(t) + [0,16) of pixel (y0 + wave*PT + r, x0 + n)
    template <bool DEFER>
    __device__ static __forceinline__ void epilogue(const ConvArgs& a, int b, int y0, int x0, int ct, int wave, int lane,
                                                    f32x16 (&acc)[PT][NT32], Pend& pend, bool last_stage) {
        constexpr int ES = 2;
        const int n = lane & 31, h = lane >> 5;
        const size_t img_out = (size_t)a.H * a.W * a.cout;
        const unsigned img_bytes = (unsigned)(img_out * ES);
        const __amdgpu_buffer_rsrc_t r_out = make_rsrc(static_cast<bf16_t*>(a.out) + (size_t)b * img_out, img_bytes);
        const bool ps = a.out_mode == RC_OUT_PIXEL_SHUFFLE2;
        const int gy0 = y0 + wave * PT, gx = x0 + n;
        const int cb0 = ct * K::COUT_TILE + 16 * h;                 // NHWC channel of (t = 0, e = 0)
        const int cps = a.cout >> 2;
        // byte offset of (row r, t = 0): NHWC  ((gy*W + gx)*cout + cb0)*2 ; PixelShuffle (((2gy+i)*2W + 2gx+h)*cps + ocblk*16*NT32)*2
        const int row_b = ps ? 4 * a.W * cps * ES : a.W * a.cout * ES;
        const int off00 = ps ? (((2 * gy0 + (ct & 1)) * (2 * a.W) + (2 * gx + h)) * cps + (ct >> 1) * 16 * NT32) * ES
                             : ((gy0 * a.W + gx) * a.cout + cb0) * ES;
        const int tstride = ps ? 32 : 64;
        const int in_off00 = ((gy0 * a.W + gx) * a.cout + cb0) * ES;          // NHWC-shaped operands (residual, mul_plus1)
        const int in_row_b = a.W * a.cout * ES;
        const float inf = __builtin_inff();
        const bool film = a.film_scale != nullptr, sums = a.chan_sums != nullptr;
        const bool nostore = (a.dbg_flags & 1) != 0;
        // the packs below are inline asm (invisible to the hazard recogniser): let the last MFMA's result land first
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DEFER) {
            pend.r = r_out; pend.tstride = tstride; pend.has = !last_stage;
        }
#pragma unroll
        for (int t = 0; t < NT32; ++t) {
            const int cpk = ct * K::COUT_TILE + 32 * t + 16 * h;             // packed index of e = 0 (film / sums / bias use packed order == NHWC channel unless ps)
            float fs[16], ft[16];
            if (film) {                                                       // NHWC only (host checks)
#pragma unroll
                for (int e = 0; e < 16; e += 4) {
                    const float4 s4 = *reinterpret_cast<const float4*>(a.film_scale + (size_t)b * a.cout + cpk + e);
                    const float4 t4 = *reinterpret_cast<const float4*>(a.film_shift + (size_t)b * a.cout + cpk + e);
                    fs[e] = s4.x; fs[e + 1] = s4.y; fs[e + 2] = s4.z; fs[e + 3] = s4.w;
                    ft[e] = t4.x; ft[e + 1] = t4.y; ft[e + 2] = t4.z; ft[e + 3] = t4.w;
                }
            }
            float csum[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) csum[e] = 0.f;
#pragma unroll
            for (int r = 0; r < PT; ++r) {
                const bool valid = gy0 + r < a.H && gx < a.W;
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = acc[r][t][e];
                if (film) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = v[e] * fs[e] + ft[e] + v[e];
                }
                if (a.act == RC_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.f, inf);
                } else if (a.act == RC_ACT_LEAKY) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.act_slope;
                } else if (a.act == RC_ACT_GELU) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = gelu_erf_f32(v[e]);
                }
                const int po = valid ? in_off00 + r * in_row_b + t * 64 : kOOB;
                if (a.mul_plus1 != nullptr) {
                    float m[16];
                    buf_load_row<bf16_t, 16>(make_rsrc(static_cast<const bf16_t*>(a.mul_plus1) + (size_t)b * img_out, img_bytes), po, m);
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = v[e] * (m[e] + 1.f);
                }
                if (a.residual != nullptr) {
                    float m[16];
                    buf_load_row<bf16_t, 16>(make_rsrc(static_cast<const bf16_t*>(a.residual) + (size_t)b * img_out, img_bytes), po, m);
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] += m[e];
                    if (a.act == RC_ACT_RELU_POST) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.f, inf);
                    }
                }
                if (sums) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) csum[e] += valid ? v[e] : 0.f;
                }
                unsigned w[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                const int oo = (valid && !nostore) ? off00 + r * row_b : kOOB;
                if (DEFER && !last_stage) {
                    pend.off[r] = oo;
                    pend.v[(r * NT32 + t) * 2] = make_uint4(w[0], w[1], w[2], w[3]);
                    pend.v[(r * NT32 + t) * 2 + 1] = make_uint4(w[4], w[5], w[6], w[7]);
                } else {
                    buf_store16(r_out, oo, t * tstride, make_uint4(w[0], w[1], w[2], w[3]));
                    buf_store16(r_out, oo + 16, t * tstride, make_uint4(w[4], w[5], w[6], w[7]));
                }
            }
            if (sums) {
                // partial sums per 8-row tile in the slots rc_ca_gate folds ([image][8x32 tile][4 slots][cout], rc_conv_sum_tiles):
                //   PT = 4: two waves per 8-row tile, each writes the totals of its two 16-pixel halves (lanes 0 / 16 of each 32-lane half)
                //   PT = 2: four waves per 8-row tile, halves combined by one swizzle, one slot per wave
#pragma unroll
                for (int e = 0; e < 16; ++e) csum[e] = SD::row_sum16(csum[e]);
                if constexpr (PT == 2) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        csum[e] += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(csum[e]), 0x401F));   // xor lane 16 inside each 32-lane half
                }
                const int row8 = (gy0 >> 3);                         // 8-row tile this wave's rows belong to (PT divides 8)
                const int slot = PT == 4 ? ((wave & 1) * 2 + ((lane >> 4) & 1)) : (wave & 3);
                const bool writer = PT == 4 ? (lane & 15) == 0 : (lane & 31) == 0;
                if (writer && row8 < a.tiles_y) {
                    float* dst = a.chan_sums + (((size_t)b * (a.tiles_x * a.tiles_y) + (size_t)row8 * a.tiles_x + (x0 / kTW)) * 4 + slot) * a.cout + cpk;
#pragma unroll
                    for (int e = 0; e < 16; ++e) dst[e] = csum[e];
                }
            }
        }
    }

    // Staged epilogue: same math as epilogue<>, but the packed bf16 tile goes to LDS ([row][pixel][8 slots of 16 B], slot ^= pixel & 7)
    // for the loader waves to store.  NHWC only.
    __device__ static __forceinline__ void epilogue_staged(const ConvArgs& a, int b, int y0, int x0, int ct, int wave, int lane,
                                                           f32x16 (&acc)[PT][NT32], char* s_out) {
        constexpr int ES = 2;
        // launder the lane id: everything below is loop-invariant per lane, and hoisted out of the unit loop it would sit in ~40 VGPRs
        // across the MFMA loop (the first build of this kernel spilled 250 registers that way)
        asm volatile("" : "+v"(lane));
        const int n = lane & 31, h = lane >> 5;
        const size_t img_out = (size_t)a.H * a.W * a.cout;
        const unsigned img_bytes = (unsigned)(img_out * ES);
        const int gy0 = y0 + wave * PT, gx = x0 + n;
        const int cb0 = ct * K::COUT_TILE + 16 * h;
        const int in_off00 = ((gy0 * a.W + gx) * a.cout + cb0) * ES;
        const int in_row_b = a.W * a.cout * ES;
        const float inf = __builtin_inff();
        const bool film = a.film_scale != nullptr, sums = a.chan_sums != nullptr;
        char* s_lane = s_out + ((wave * PT) * 32 + n) * 128;
        const int hm = (2 * h) ^ (n & 7);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT32; ++t) {
            const int cpk = ct * K::COUT_TILE + 32 * t + 16 * h;
            float fs[16], ft[16];
            if (film) {
#pragma unroll
                for (int e = 0; e < 16; e += 4) {
                    const float4 s4 = *reinterpret_cast<const float4*>(a.film_scale + (size_t)b * a.cout + cpk + e);
                    const float4 t4 = *reinterpret_cast<const float4*>(a.film_shift + (size_t)b * a.cout + cpk + e);
                    fs[e] = s4.x; fs[e + 1] = s4.y; fs[e + 2] = s4.z; fs[e + 3] = s4.w;
                    ft[e] = t4.x; ft[e + 1] = t4.y; ft[e + 2] = t4.z; ft[e + 3] = t4.w;
                }
            }
            float csum[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) csum[e] = 0.f;
#pragma unroll
            for (int r = 0; r < PT; ++r) {
                const bool valid = gy0 + r < a.H && gx < a.W;
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = acc[r][t][e];
                if (film) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = v[e] * fs[e] + ft[e] + v[e];
                }
                if (a.act == RC_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.f, inf);
                } else if (a.act == RC_ACT_LEAKY) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.act_slope;
                } else if (a.act == RC_ACT_GELU) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = gelu_erf_f32(v[e]);
                }
                const int po = valid ? in_off00 + r * in_row_b + t * 64 : kOOB;
                if (a.mul_plus1 != nullptr) {
                    float m[16];
                    buf_load_row<bf16_t, 16>(make_rsrc(static_cast<const bf16_t*>(a.mul_plus1) + (size_t)b * img_out, img_bytes), po, m);
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = v[e] * (m[e] + 1.f);
                }
                if (a.residual != nullptr) {
                    float m[16];
                    buf_load_row<bf16_t, 16>(make_rsrc(static_cast<const bf16_t*>(a.residual) + (size_t)b * img_out, img_bytes), po, m);
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] += m[e];
                    if (a.act == RC_ACT_RELU_POST) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.f, inf);
                    }
                }
                if (sums) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) csum[e] += valid ? v[e] : 0.f;
                }
                unsigned w[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
                // slot (4t + 2h + j) ^ (n & 7) = (4t + j) ^ hm with hm = 2h ^ (n & 7): t and j touch bits 2 and 0, h bit 1
                char* dst = s_lane + r * 4096;
                *reinterpret_cast<uint4*>(dst + (((4 * t) ^ hm) << 4)) = make_uint4(w[0], w[1], w[2], w[3]);
                *reinterpret_cast<uint4*>(dst + (((4 * t + 1) ^ hm) << 4)) = make_uint4(w[4], w[5], w[6], w[7]);
                __builtin_amdgcn_sched_barrier(0);       // keep one row's operand loads from being clustered with the next rows' (register pressure)
            }
            if (sums) {
#pragma unroll
                for (int e = 0; e < 16; ++e) csum[e] = SD::row_sum16(csum[e]);
                if constexpr (PT == 2) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        csum[e] += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(csum[e]), 0x401F));
                }
                const int row8 = (gy0 >> 3);
                const int slot = PT == 4 ? ((wave & 1) * 2 + ((lane >> 4) & 1)) : (wave & 3);
                const bool writer = PT == 4 ? (lane & 15) == 0 : (lane & 31) == 0;
                if (writer && row8 < a.tiles_y) {
                    float* dst = a.chan_sums + (((size_t)b * (a.tiles_x * a.tiles_y) + (size_t)row8 * a.tiles_x + (x0 / kTW)) * 4 + slot) * a.cout + cpk;
#pragma unroll
                    for (int e = 0; e < 16; ++e) dst[e] = csum[e];
                }
            }
        }
    }
};

template <class K, bool GATED, bool DEFER>
__global__ __launch_bounds__(K::THREADS) void conv32_kernel(const ConvArgs a) {
    using St = typename K::Stage;
    using D = ConvDev<St>;                                           // halo-tile staging on the TH-row tile
    using C = C32Dev<K>;
    constexpr int NT32 = K::NT32, STEPS = K::STEPS, SA = K::SA, PT = K::PT;
    constexpr int WA = K::WA, WB = K::WB, WALL = K::W_BYTES;
    constexpr int NWA = (WA / 16 + kThreads - 1) / kThreads, NWB = (WB / 16 + kThreads - 1) / kThreads;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;
    float* s_bias = reinterpret_cast<float*>(smem + WALL);
    char* s_in0 = smem + WALL + kPersistMaxCout * 4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wv >= K::NCW;

    const int tiles_y = (a.H + K::TH - 1) / K::TH;
    const TileDecode& td = K::TH == kTH ? a.td : a.td_wsm;
    const int sp_total = a.tiles_x * tiles_y;
    const int n_tiles = sp_total * a.batch;
    const int n_chunks = a.n_chunks, n_ct = a.n_ct;
    // work units as in conv_mfma_wsm_kernel: one chunk -> unit = tile (staged once, all cout tiles); several -> unit = (tile, cout tile)
    const bool one_chunk = n_chunks == 1;
    const int n_units = one_chunk ? n_tiles : n_tiles * n_ct;
    const int cts_per_unit = one_chunk ? n_ct : 1;
    const int slots = gridDim.x >> 3;
    const int pos = (blockIdx.x & 7) * slots + (blockIdx.x >> 3);
    const int stride = (int)gridDim.x;
    const int my_units = pos < n_units ? (n_units - pos + stride - 1) / stride : 0;
    const int my_stages = my_units * cts_per_unit * n_chunks;

    for (int i = tid; i < a.cout_packed; i += K::THREADS) s_bias[i] = a.bias ? a.bias[i] : 0.f;

    auto decode = [&](int unit, int& b, int& ty, int& tx, int& ct0) {
        int tile = unit;
        ct0 = 0;
        if (!one_chunk) { tile = magic_div(unit, a.div_n_ct); ct0 = unit - tile * n_ct; }
        b = magic_div(tile, td.sp_total);
        band_decode(tile - b * sp_total, a.tiles_x, tiles_y, td, ty, tx);
    };

    if (loader) {
        // ---------------------------------------------------------------- producer waves (protocol of conv_mfma_wsm_kernel)
        const int rtid = tid - K::COMPUTE;                           // 0..255
        const int lw = wv - K::NCW;                                  // 0..3
        constexpr bool WDMA = GATED && K::NCW > 4;                   // 12-wave form (168 registers), gated: two tile operands fill the registers, weight halves go by LDS-DMA
        uint4 r0[D::NI], r1[GATED ? D::NI : 1], wra[WDMA ? 1 : NWA], wrb[WDMA ? 1 : NWB];
        float gv[GATED ? D::UNIT : 1];
        typename D::TileSrc ts;
        typename D::TileOffs to;
        D::tile_offsets(a, rtid, to);
        int woa[WDMA ? 1 : NWA], wob[WDMA ? 1 : NWB];
        if constexpr (!WDMA) {
#pragma unroll
            for (int k = 0; k < NWA; ++k) woa[k] = (k * kThreads + rtid) * 16 < WA ? (k * kThreads + rtid) * 16 : kOOB;
#pragma unroll
            for (int k = 0; k < NWB; ++k) wob[k] = (k * kThreads + rtid) * 16 < WB ? WA + (k * kThreads + rtid) * 16 : kOOB;
        }
        auto dma = [&](int lds_off, int bytes, int goff) {
            for (int kb = lw; kb < bytes / 1024; kb += 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(static_cast<const char*>(a.wpacked) + goff + lds_off + kb * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(s_w + lds_off + kb * 1024), 16, 0, 0);
        };
        const __amdgpu_buffer_rsrc_t r_w = make_rsrc(a.wpacked, (unsigned)((size_t)n_ct * n_chunks * WALL));
        ConvArgs aa = a;

        int k_unit = 0, cti = 0, ct = 0, chunk = 0, gi = 0;
        int b = 0, ty = 0, tx = 0, ct0 = 0;
        int wsoff = 0, c_chunk = 0, c_buf = 0;
        bool c_tile = false;
        auto issue_a = [&]() {
            if (chunk == 0) {
                if (cti == 0) decode(pos + k_unit * stride, b, ty, tx, ct0);
                ct = ct0 + cti;
                aa.in_store = ct == 0 ? a.in_store : nullptr;
                ts = D::tile_src(aa, b, ty * K::TH, tx * kTW);
            }
            c_tile = !one_chunk || cti == 0;
            c_buf = one_chunk ? (k_unit & 1) : (gi & 1);
            c_chunk = chunk;
            if (c_tile) D::template load_tile<GATED>(aa, ts, to, b, chunk, rtid, r0, r1, gv);
            wsoff = (ct * n_chunks + chunk) * WALL;
            if constexpr (!WDMA) {
#pragma unroll
                for (int k = 0; k < NWA; ++k) wra[k] = buf_load16(r_w, woa[k], wsoff);
            }
            ++gi;
            if (++chunk == n_chunks) { chunk = 0; if (++cti == cts_per_unit) { cti = 0; ++k_unit; } }
        };
        auto issue_b = [&]() {
            if constexpr (!WDMA) {
#pragma unroll
                for (int k = 0; k < NWB; ++k) wrb[k] = buf_load16(r_w, wob[k], wsoff);
            }
        };
        auto commit_a = [&]() {
            if (c_tile) D::template commit_tile<GATED>(aa, ts, to, c_chunk, rtid, r0, r1, gv, s_in0 + c_buf * St::IN_BYTES);
            if constexpr (!WDMA) {
#pragma unroll
                for (int k = 0; k < NWA; ++k)
                    if (woa[k] != kOOB) *reinterpret_cast<uint4*>(s_w + woa[k]) = wra[k];
            }
        };
        auto commit_b = [&]() {
            if constexpr (!WDMA) {
#pragma unroll
                for (int k = 0; k < NWB; ++k)
                    if (wob[k] != kOOB) *reinterpret_cast<uint4*>(s_w + wob[k]) = wrb[k];
            }
        };

        if (my_stages > 0) {
            issue_a(); commit_a(); issue_b();
            if constexpr (WDMA) dma(0, WA, wsoff);
        }
        __syncthreads();                                             // barrier 0: bias, Wa(0), tile(0) visible
        const bool rec = a.dbg != nullptr && blockIdx.x == 8 && lw == 0 && lane == 0;
        for (int g = 0; g < my_stages; ++g) {
            const long long t0 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            // commit BEFORE issue in both halves: the commit's wait then covers only loads that have had a whole half-stage to land.
            // (In the other order hipcc cannot count the conditionally issued younger loads and waits vmcnt(0): every half then
            // sat out the latency of the loads it had just issued -- measured 2650 + 1660 loader cycles per stage.)
            const int wsoff_g = wsoff;
            if constexpr (WDMA) dma(WA, WB, wsoff_g);
            commit_b();
            if (g + 1 < my_stages) issue_a();
            const long long t1 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            __syncthreads();
            const long long t2 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            if (g + 1 < my_stages) {
                if constexpr (WDMA) dma(0, WA, wsoff);
                commit_a(); issue_b();
            }
            const long long t3 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            __syncthreads();
            if (rec && g < 60) {
                long long* d = a.dbg + 512 + 4 * g;
                d[0] = t1 - t0; d[1] = t2 - t1; d[2] = t3 - t2; d[3] = (long long)__builtin_amdgcn_s_memtime() - t3;
            }
        }
    } else {
        // ---------------------------------------------------------------- consumer waves (one per SIMD when NCW == 4)
        const int n = lane & 31, h = lane >> 5;
        const int lane_in = ((wv * PT) * St::TWH + n) * St::SPIX + h * 16;
        const char* w_lane = s_w + lane * 16;
        typename C::Pend pend;
        pend.has = false;
        __syncthreads();                                             // barrier 0
        const bool rec = a.dbg != nullptr && blockIdx.x == 8 && wv == 0 && lane == 0;    // phase stamps (tools/conv32_phases.py)
        long long tl = 0;
        int g = 0;
        for (int k = 0; k < my_units; ++k) {
            int b, ty, tx, ct0;
            decode(pos + k * stride, b, ty, tx, ct0);
            for (int ct = ct0; ct < ct0 + cts_per_unit; ++ct) {
                f32x16 acc[PT][NT32];                                // initial C operand = bias (packed order)
#pragma unroll
                for (int t = 0; t < NT32; ++t) {
                    const float* bp = s_bias + ct * K::COUT_TILE + 32 * t + 16 * h;
                    f32x16 bv;
#pragma unroll
                    for (int e = 0; e < 16; ++e) bv[e] = bp[e];
#pragma unroll
                    for (int r = 0; r < PT; ++r) acc[r][t] = bv;
                }
                for (int c = 0; c < n_chunks; ++c, ++g) {
                    const char* in_lane = s_in0 + (one_chunk ? (k & 1) : (g & 1)) * St::IN_BYTES + lane_in;
                    const long long t0 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
                    if (!(a.dbg_flags & 2)) C::template mma_steps<0, SA, DEFER>(in_lane, w_lane, acc, pend);
                    if constexpr (DEFER) pend.has = false;
                    const long long t1 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
                    __syncthreads();
                    const long long t2 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
                    if (!(a.dbg_flags & 2)) C::template mma_steps<SA, STEPS, false>(in_lane, w_lane, acc, pend);
                    const long long t3 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
                    if (c + 1 < n_chunks) __syncthreads();
                    if (rec && g < 60) {
                        long long* d = a.dbg + 8 * g;
                        d[0] = t1 - t0; d[1] = t2 - t1; d[2] = t3 - t2; d[3] = c + 1 < n_chunks ? (long long)__builtin_amdgcn_s_memtime() - t3 : 0;
                        d[4] = 0; d[5] = 0;
                    }
                    tl = t3;
                }
                C::template epilogue<DEFER>(a, b, ty * K::TH, tx * kTW, ct, wv, lane, acc, pend, g == my_stages);
                const long long t4 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
                __syncthreads();
                if (rec && g - 1 < 60) { long long* d = a.dbg + 8 * (g - 1); d[4] = t4 - tl; d[5] = (long long)__builtin_amdgcn_s_memtime() - t4; }
            }
        }
    }
}

template <class K, bool GATED>
int launch_conv32_g(const ConvArgs& a, hipStream_t stream) {
    const int tiles_y = (a.H + K::TH - 1) / K::TH;
    const int n_items = a.tiles_x * tiles_y * a.batch * (a.n_chunks > 1 ? a.n_ct : 1);
    int grid = a.num_cus;
    if (grid > n_items) grid = n_items;
    grid = (grid + 7) / 8 * 8;
    // single-chunk layers (the tail) end every stage with an epilogue: its stores are issued under the next stage's MFMAs
    constexpr bool CAN_DEFER = K::NPEND <= K::SA && K::CK == 48;
    bool deferred = false;
    if constexpr (CAN_DEFER) {
        if ((a.dbg_flags & 8) == 0 && a.n_chunks == 1) {
            static PerDeviceFlag attr_set;
            if (!attr_set.test_and_set())
                RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv32_kernel<K, GATED, true>), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
            hipLaunchKernelGGL((conv32_kernel<K, GATED, true>), dim3((unsigned)grid), dim3(K::THREADS), K::LDS_BYTES, stream, a);
            deferred = true;
        }
    }
    if (!deferred) {
        static PerDeviceFlag attr_set;
        if (!attr_set.test_and_set())
            RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv32_kernel<K, GATED, false>), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
        hipLaunchKernelGGL((conv32_kernel<K, GATED, false>), dim3((unsigned)grid), dim3(K::THREADS), K::LDS_BYTES, stream, a);
    }
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

template <class K>
int launch_conv32(const ConvArgs& a, hipStream_t stream) {
    if (!a.cin_vec_ok || !a.cin_chunk_ok) return fail(RC_ERR_INVALID, "conv32: tensors must be 16-byte aligned and Cin a whole number of chunks");
    if (a.tiles_x * ((a.H + K::TH - 1) / K::TH) * a.batch >= (1 << 24)) return fail(RC_ERR_INVALID, "conv32: too many tiles");
    return a.in_gate != nullptr ? launch_conv32_g<K, true>(a, stream) : launch_conv32_g<K, false>(a, stream);
}

// ==================================================================================================================================
// Staged-output form ("conv32s") for the multi-chunk layers.  What bounded the form above was not the matrix pipe (its MFMA phases run at
// 35 of 32 cycles per MFMA) but what sits between them (profiles/r03_conv32_phases.md):
//   * a wave needs ~225-650 cycles per buffer_store_dwordx4 (tools/ubench/store_issue.hip: 4 waves storing row-per-lane pieces reach one
//     store instruction per 158 cycles per CU, 8 waves writing whole 128-byte runs one per 43), so the four compute waves sat ~4 600 cycles
//     in every tile's epilogue with the matrix pipe idle
//   * two barriers per stage (weights single-buffered by halves).
// Here a stage is ONE tap-major pass over a 16-channel chunk (9 K16 steps), input tile AND packed weights are double-buffered (one
// barrier per stage), and the LDS this frees holds the tile's packed bf16 output (64 KB): the compute waves' epilogue is cvt + 16
// ds_write_b128, and the four loader waves -- who have the slack -- drain it to HBM over the next stages with 8-pixel x 128-byte
// store instructions (every 16-byte slot of a pixel's 64 couts is XOR-swizzled with the pixel index: conflict-free on both sides).
// ==================================================================================================================================
template <int NCW_>
struct C32SCfg {
    using Stage = C32Stage<16, 16>;
    static constexpr int CK = 16, TH = 16, NCW = NCW_, NT32 = 2;
    static constexpr int SPT = 1, STEPS = 9, SA = 9;
    static constexpr int PT = TH / NCW;
    static constexpr int COUT_TILE = 64;
    static constexpr int W_BYTES = STEPS * NT32 * 1024;            // 18 KB per (cout tile, chunk)
    static constexpr int COMPUTE = NCW * 64, THREADS = COMPUTE + kThreads;
    static constexpr int OUT_BYTES = TH * kTW * 128;               // 16 x 32 pixels x 64 couts bf16
    static constexpr int LDS_BYTES = 2 * W_BYTES + kPersistMaxCout * 4 + 2 * Stage::IN_BYTES + OUT_BYTES;
    static constexpr int FR = NT32 + PT, FM = NT32 * PT;
    static constexpr int NPEND = PT * NT32 * 2;
    static constexpr int NWR = (W_BYTES / 16 + kThreads - 1) / kThreads;   // weight pieces per loader thread
    static constexpr int DRAIN = OUT_BYTES / 16 / kThreads;                // store instructions per loader wave per tile (16)
    static constexpr int DRAIN_PER_STAGE = 4;
    static_assert(LDS_BYTES <= 160 * 1024, "conv32s: LDS budget");
};

template <class K, bool GATED>
__global__ __launch_bounds__(K::THREADS) void conv32s_kernel(const ConvArgs a) {
    using St = typename K::Stage;
    using D = ConvDev<St>;
    using C = C32Dev<K>;
    constexpr int NT32 = K::NT32, PT = K::PT, WALL = K::W_BYTES, NWR = K::NWR;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w0 = smem;                                               // 2 x packed weights
    float* s_bias = reinterpret_cast<float*>(smem + 2 * WALL);
    char* s_in0 = smem + 2 * WALL + kPersistMaxCout * 4;             // 2 x halo tile
    char* s_out = s_in0 + 2 * St::IN_BYTES;                          // packed output tile [row][pixel][8 swizzled 16-byte slots]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wv >= K::NCW;

    const int tiles_y = (a.H + K::TH - 1) / K::TH;
    const TileDecode& td = a.td_wsm;
    const int sp_total = a.tiles_x * tiles_y;
    const int n_tiles = sp_total * a.batch;
    const int n_chunks = a.n_chunks, n_ct = a.n_ct;                  // n_chunks >= 2: unit = (tile, cout tile), cout tile fastest
    const int n_units = n_tiles * n_ct;
    const int slots = gridDim.x >> 3;
    const int pos = (blockIdx.x & 7) * slots + (blockIdx.x >> 3);
    const int stride = (int)gridDim.x;
    const int my_units = pos < n_units ? (n_units - pos + stride - 1) / stride : 0;
    const int my_stages = my_units * n_chunks;
    constexpr bool staged = true;                                    // NHWC only: PixelShuffle layers take the two-barrier form above (make_plan)

    for (int i = tid; i < a.cout_packed; i += K::THREADS) s_bias[i] = a.bias ? a.bias[i] : 0.f;

    auto decode = [&](int unit, int& b, int& ty, int& tx, int& ct) {
        const int tile = magic_div(unit, a.div_n_ct);
        ct = unit - tile * n_ct;
        b = magic_div(tile, td.sp_total);
        band_decode(tile - b * sp_total, a.tiles_x, tiles_y, td, ty, tx);
    };

    if (loader) {
        // ---------------------------------------------------------------- producer waves
        const int rtid = tid - K::COMPUTE;
        const int lw = wv - K::NCW;
        uint4 r0[D::NI], r1[GATED ? D::NI : 1], wr[NWR];
        float gv[GATED ? D::UNIT : 1];
        typename D::TileSrc ts;
        typename D::TileOffs to;
        D::tile_offsets(a, rtid, to);
        int wo[NWR];
#pragma unroll
        for (int k = 0; k < NWR; ++k) wo[k] = (k * kThreads + rtid) * 16 < WALL ? (k * kThreads + rtid) * 16 : kOOB;
        const __amdgpu_buffer_rsrc_t r_w = make_rsrc(a.wpacked, (unsigned)((size_t)n_ct * n_chunks * WALL));
        ConvArgs aa = a;

        // stage whose loads are issued next
        int i_unit = 0, i_chunk = 0, i_g = 0;
        int b = 0, ty = 0, tx = 0, ct = 0;
        int c_chunk = 0, c_buf = 0;                                   // of the stage held in registers
        auto issue = [&]() {
            if (i_chunk == 0) {
                decode(pos + i_unit * stride, b, ty, tx, ct);
                aa.in_store = ct == 0 ? a.in_store : nullptr;        // only one cout tile materialises a gated input
                ts = D::tile_src(aa, b, ty * K::TH, tx * kTW);
            }
            c_buf = i_g & 1; c_chunk = i_chunk;
            D::template load_tile<GATED>(aa, ts, to, b, i_chunk, rtid, r0, r1, gv);
            const int wsoff = (ct * n_chunks + i_chunk) * WALL;
#pragma unroll
            for (int k = 0; k < NWR; ++k) wr[k] = buf_load16(r_w, wo[k], wsoff);
            ++i_g;
            if (++i_chunk == n_chunks) { i_chunk = 0; ++i_unit; }
        };
        auto commit = [&]() {
            D::template commit_tile<GATED>(aa, ts, to, c_chunk, rtid, r0, r1, gv, s_in0 + c_buf * St::IN_BYTES);
            char* wdst = s_w0 + c_buf * WALL;
#pragma unroll
            for (int k = 0; k < NWR; ++k)
                if (wo[k] != kOOB) *reinterpret_cast<uint4*>(wdst + wo[k]) = wr[k];
        };
        // drain of the staged output tile: store instruction i of this wave covers pieces ((i*4 + lw)*64 + lane) of the tile's 4096
        // 16-byte pieces, piece = (row*32 + px)*8 + slot: one instruction = 8 pixels x 128 contiguous bytes
        __amdgpu_buffer_rsrc_t d_r = make_rsrc(nullptr, 0u);
        int d_y0 = 0, d_x0 = 0, d_ct = 0, d_left = 0;
        auto drain = [&](int count) {
            for (int q = 0; q < count && d_left > 0; ++q, --d_left) {
                const int i = K::DRAIN - d_left;
                const int piece = (i * 4 + lw) * 64 + lane;
                const int row = piece >> 8, px = (piece >> 3) & 31, slot = piece & 7;
                const uint4 v = *reinterpret_cast<const uint4*>(s_out + (row * 32 + px) * 128 + ((slot ^ (px & 7)) << 4));
                const int gy = d_y0 + row, gx = d_x0 + px;
                const int off = (gy < a.H && gx < a.W && !(a.dbg_flags & 1)) ? ((gy * a.W + gx) * a.cout + d_ct * 64) * 2 + slot * 16 : kOOB;
                buf_store16(d_r, off, 0, v);
            }
        };

        if (my_stages > 0) { issue(); commit(); }
        if (my_stages > 1) issue();
        __syncthreads();                                             // barrier 0: bias, stage 0 visible
        const bool rec = a.dbg != nullptr && blockIdx.x == 8 && lw == 0 && lane == 0;
        int g_chunk = 0, g_unit = 0;                                 // the stage the computers are working on
        for (int g = 0; g < my_stages; ++g) {
            const long long t0 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            if (g + 1 < my_stages) commit();                         // stage g+1 (loaded during stage g-1) -> the buffers stage g-1 used
            if (g + 2 < my_stages) issue();
            const long long t1 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            if (staged) drain(K::DRAIN_PER_STAGE);
            const long long t2 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            __syncthreads();
            if (rec && g < 60) {
                long long* d = a.dbg + 512 + 4 * g;
                d[0] = t1 - t0; d[1] = t2 - t1; d[2] = (long long)__builtin_amdgcn_s_memtime() - t2; d[3] = 0;
            }
            if (++g_chunk == n_chunks) {                             // the computers have just staged unit g_unit's output
                g_chunk = 0;
                if (staged) {
                    int db, dty, dtx;
                    decode(pos + g_unit * stride, db, dty, dtx, d_ct);
                    d_y0 = dty * K::TH; d_x0 = dtx * kTW;
                    const size_t img_out = (size_t)a.H * a.W * a.cout;
                    d_r = make_rsrc(static_cast<bf16_t*>(a.out) + (size_t)db * img_out, (unsigned)(img_out * 2));
                    d_left = K::DRAIN;
                }
                ++g_unit;
            }
        }
        if (staged) drain(K::DRAIN);
    } else {
        // ---------------------------------------------------------------- consumer waves
        const int n = lane & 31, h = lane >> 5;
        const int lane_in = ((wv * PT) * St::TWH + n) * St::SPIX + h * 16;
        typename C::Pend pend;
        pend.has = false;
        __syncthreads();                                             // barrier 0
        const bool rec = a.dbg != nullptr && blockIdx.x == 8 && wv == 0 && lane == 0;
        int g = 0;
        for (int k = 0; k < my_units; ++k) {
            int b, ty, tx, ct;
            decode(pos + k * stride, b, ty, tx, ct);
            f32x16 acc[PT][NT32];
#pragma unroll
            for (int t = 0; t < NT32; ++t) {
                const float* bp = s_bias + ct * K::COUT_TILE + 32 * t + 16 * h;
                f32x16 bv;
#pragma unroll
                for (int e = 0; e < 16; ++e) bv[e] = bp[e];
#pragma unroll
                for (int r = 0; r < PT; ++r) acc[r][t] = bv;
            }
            // (the epilogue sits AFTER the chunk loop: placed conditionally inside it hipcc kept a second copy of the 128 accumulator registers)
            long long t0 = 0, t1 = 0;
            int c = 0;
            do {                                                     // (do-while: with a possibly-zero-trip for loop hipcc merges the bias-initialised and the accumulated tiles through copies and spills ~150 registers)
                const char* in_lane = s_in0 + (g & 1) * St::IN_BYTES + lane_in;
                const char* w_lane = s_w0 + (g & 1) * WALL + lane * 16;
                t0 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
                if (!(a.dbg_flags & 2)) C::template mma_steps<0, K::STEPS, false>(in_lane, w_lane, acc, pend);
                t1 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
                if (c + 1 < n_chunks) {
                    __syncthreads();
                    if (rec && g < 60) {
                        long long* d = a.dbg + 8 * g;
                        d[0] = t1 - t0; d[1] = 0; d[2] = 0; d[3] = 0; d[4] = 0; d[5] = (long long)__builtin_amdgcn_s_memtime() - t1;
                    }
                }
                ++c; ++g;
            } while (c < n_chunks);
            C::epilogue_staged(a, b, ty * K::TH, tx * kTW, ct, wv, lane, acc, s_out);
            const long long t2 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            __syncthreads();
            if (rec && g - 1 < 60) {
                long long* d = a.dbg + 8 * (g - 1);
                d[0] = t1 - t0; d[1] = 0; d[2] = 0; d[3] = 0; d[4] = t2 - t1; d[5] = (long long)__builtin_amdgcn_s_memtime() - t2;
            }
        }
    }
}

template <class K, bool GATED>
int launch_conv32s_g(const ConvArgs& a, hipStream_t stream) {
    const int tiles_y = (a.H + K::TH - 1) / K::TH;
    const int n_items = a.tiles_x * tiles_y * a.batch * a.n_ct;
    int grid = a.num_cus;
    if (grid > n_items) grid = n_items;
    grid = (grid + 7) / 8 * 8;
    static PerDeviceFlag attr_set;
    if (!attr_set.test_and_set())
        RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv32s_kernel<K, GATED>), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
    hipLaunchKernelGGL((conv32s_kernel<K, GATED>), dim3((unsigned)grid), dim3(K::THREADS), K::LDS_BYTES, stream, a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

template <class K>
int launch_conv32s(const ConvArgs& a, hipStream_t stream) {
    if (!a.cin_vec_ok || !a.cin_chunk_ok) return fail(RC_ERR_INVALID, "conv32: tensors must be 16-byte aligned and Cin a whole number of chunks");
    if (a.tiles_x * ((a.H + K::TH - 1) / K::TH) * a.batch >= (1 << 24)) return fail(RC_ERR_INVALID, "conv32: too many tiles");
    // the loaders drain a staged tile DRAIN_PER_STAGE instructions per stage and finish the rest when the next tile is staged
    // (4 stages); the computers overwrite the staging buffer in a unit's LAST stage, so a unit must have more than 5 stages
    if (a.n_chunks < 6) return fail(RC_ERR_INVALID, "conv32s: needs at least six 16-channel Cin chunks");
    if (a.out_mode != RC_OUT_NHWC) return fail(RC_ERR_INVALID, "conv32s: NHWC store only");
    return a.in_gate != nullptr ? launch_conv32s_g<K, true>(a, stream) : launch_conv32s_g<K, false>(a, stream);
}

// defined in conv32_inst_*.hip
int conv32_ck32(int variant, const ConvArgs& a, hipStream_t s);
int conv32_ck16(int variant, const ConvArgs& a, hipStream_t s);
int conv32_ck48(int variant, const ConvArgs& a, hipStream_t s);

}  // namespace rc
