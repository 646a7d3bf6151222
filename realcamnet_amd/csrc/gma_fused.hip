// GroupMix block, fused per-token stages for the cfg3 shape (dim 80 = 5 x 16, 8 heads, bf16) -- upstream models/groupmix.py:159-299.
//
// Every stage of GMA_Block that is a per-token map (LayerNorm, the Linear layers, the attention read-out
// q.(softmax(k)^T v) + q*crpe(v), GELU, residual adds) runs here as ONE pass with the token's activations held in registers:
//   gma_ln_qkv_kernel : x -> LayerNorm1 -> qkv Linear (80 -> 240)                                   (groupmix.py:178, 293)
//   gma_tail_kernel   : [q, convv, loc, k^T v] -> attention read-out -> proj + x -> LayerNorm2 -> fc1 -> GELU -> fc2 + .
//                       [-> Conv1x1 80 -> Cout + residual: the cfg3 net's gma_out]                 (groupmix.py:189-199, 296-298)
// The layer-by-layer form (rc_gma_apply, rc_layernorm, 1x1 rc_conv2d x4) moved 17 GB per step for these stages (the 320-channel
// MLP hidden map alone 5.3 GB); fused, each token's bytes cross HBM once: q/convv/loc/x (+ d1) in, the result out.
//
// Register-resident chain of MFMA layers.  A wave owns 64 consecutive tokens = 4 column tiles of v_mfma_f32_16x16x32_bf16
// (B operand = activations: lane (n = lane & 15, g = lane >> 4) holds channels 32 s + 8 g + 0..7 of token n for K-step s;
// a trailing 16 channels use the lower half of a K-step: channels C0 + 4 g + 0..3, upper half zero).  A layer's weights are packed so that
// MFMA row R = 4 g + j of output tile m is channel 32 (m >> 1) + 8 g + 4 (m & 1) + j: the C/D fragments of output tiles
// (2 p, 2 p + 1) in lane (n, g) are then exactly channels 32 p + 8 g + 0..7 of token n -- the B fragment of the NEXT layer's
// K-step p after fp32 -> bf16 packing, with no cross-lane movement (an unpaired last tile keeps natural order 16 m + R and feeds
// the K = 16 form).  HBM loads / stores use the same map: 16 bytes per lane, 64 contiguous bytes per token and K-step.
// Weights (packed by rc_chain_pack_weights) sit in LDS for the whole launch; each A fragment read (ds_read_b128) feeds 4 MFMAs.
// Rounding points are those of the layer-by-layer path (every tensor that path stored in bf16 is rounded to bf16 here too);
// accumulation is fp32.  LayerNorm statistics: per-lane partial sums + two xor-shuffles (the 4 lanes of a token).
#include "common.hpp"

namespace rc {
namespace gf {

typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int kNT = 4;                       // column tiles (16 tokens each) per wave
constexpr int kC = 80, kCT = 64, kSEG = 16, kHID = 320;

__host__ __device__ constexpr int tile_bytes(int cin) { return (cin / 32) * 1024 + ((cin % 32) ? 512 : 0); }
__host__ __device__ constexpr int n_mtiles(int cout) { return (cout + 15) / 16; }
__host__ __device__ constexpr int row_channel(int m, int R, int mt) {
    return (m < (mt & ~1)) ? 32 * (m >> 1) + 8 * (R >> 2) + 4 * (m & 1) + (R & 3) : 16 * m + R;
}

template <int C> struct Act {                // one token column tile's activations in B-operand layout
    static constexpr int KS = C / 32;
    static constexpr bool TAIL = (C % 32) != 0;
    uint4 f[KS > 0 ? KS : 1];
    uint2 t;
};

__device__ __forceinline__ void mma32(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// K = 16 step (4 values per lane: k = 4 q + i) as a K = 32 MFMA with the upper 4 k-slots of every lane zero in both operands.
// NOT v_mfma_f32_16x16x16_bf16: on gfx950 / ROCm 7.2 results of that (legacy) opcode read by the VALU shortly afterwards came out
// stale for random lanes -- hipcc pads its result latency with `s_nop 4` (a 4-pass model) and that is not enough; the K = 32 opcode's
// wait states are right (the conv kernels live on them).  Costs nothing here: the chains are HBM / VALU bound.
__device__ __forceinline__ void mma16(const uint2& a, const uint2& b, f32x4& c) {
    mma32(make_uint4(a.x, a.y, 0u, 0u), make_uint4(b.x, b.y, 0u, 0u), c);
}
// fp32 pair -> packed bf16: Vec16::rne2, ONE v_cvt_pk_bf16_f32 written as inline asm (the vector-typed __builtin_convertvector form
// of the same instruction cost this kernel ~120 spilled VGPRs).  HAZARD RULE: an inline-asm instruction is opaque to the compiler's
// hazard recogniser, so it must never read an MFMA accumulator directly -- the first build of this file did (accumulators seeded
// with bias + residual, packed straight after the MFMA chain): the conversion issued inside the MFMA's result latency and packed
// stale register contents (random tokens came out as garbage / NaN, and only for some instruction schedules).  Every pack below
// therefore takes the result of a real VALU instruction (the bias add is done AFTER the chain, on purpose).
__device__ __forceinline__ uint32_t pk(float lo, float hi) { return Vec16<bf16_t>::rne2(lo, hi); }
__device__ __forceinline__ uint4 pack_pair(const f32x4& lo, const f32x4& hi) {
    return make_uint4(pk(lo[0], lo[1]), pk(lo[2], lo[3]), pk(hi[0], hi[1]), pk(hi[2], hi[3]));
}
__device__ __forceinline__ uint2 pack_tail(const f32x4& v) { return make_uint2(pk(v[0], v[1]), pk(v[2], v[3])); }
// fp32 quad -> packed bf16 with the conversion visible to the compiler (the same v_cvt_pk_bf16_f32, round-to-nearest-even): in rc_gma_qkv_aggregate the
// packed values are MFMA B operands a few instructions later and their registers are recycled between the point-wise MFMAs; with the inline-asm
// form (pack_tail) hipcc's hazard recogniser did not see those writes -- it rewrote an in-flight MFMA's B registers in the slot right behind
// it and fed the next MFMA one instruction after the conversion, and the third pixel of every 7x7 patch came out wrong, differently run to run.
typedef __bf16 qa_bf16x2 __attribute__((ext_vector_type(2)));
typedef float qa_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t qa_pk(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(qa_f32x2{lo, hi}, qa_bf16x2));
}
__device__ __forceinline__ uint2 qa_pack(const f32x4& v) { return make_uint2(qa_pk(v[0], v[1]), qa_pk(v[2], v[3])); }

// bf16 pairs -> fp32, as vectors (no float arrays: with the vector-typed conversion above they would not be promoted to registers)
__device__ __forceinline__ f32x4 up_lo(const uint4& r) {
    return f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
}
__device__ __forceinline__ f32x4 up_hi(const uint4& r) {
    return f32x4{__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u), __uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u)};
}
__device__ __forceinline__ f32x4 up_tail(const uint2& r) {
    return f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
}
__device__ __forceinline__ f32x4 ld4(const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    return f32x4{t.x, t.y, t.z, t.w};
}

// acc[g][nt] += W[tiles m0 .. m0+MG) . in   (weights of a layer with CIN inputs at LDS address w, fragment order)
template <int CIN, int MG>
__device__ __forceinline__ void gemm_tiles(const char* w, int m0, int lane, const Act<CIN> (&in)[kNT], f32x4 (&acc)[MG][kNT]) {
    constexpr int KS = CIN / 32, TB = tile_bytes(CIN);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        uint4 a[MG];
#pragma unroll
        for (int g = 0; g < MG; ++g) a[g] = *reinterpret_cast<const uint4*>(w + (m0 + g) * TB + s * 1024 + lane * 16);
#pragma unroll
        for (int g = 0; g < MG; ++g)
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) mma32(a[g], in[nt].f[s], acc[g][nt]);
    }
    if constexpr ((CIN % 32) != 0) {
        uint2 a[MG];
#pragma unroll
        for (int g = 0; g < MG; ++g) a[g] = *reinterpret_cast<const uint2*>(w + (m0 + g) * TB + KS * 1024 + lane * 8);
#pragma unroll
        for (int g = 0; g < MG; ++g)
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) mma16(a[g], in[nt].t, acc[g][nt]);
    }
}

template <int MG> __device__ __forceinline__ void zero(f32x4 (&acc)[MG][kNT]) {
#pragma unroll
    for (int g = 0; g < MG; ++g)
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) acc[g][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// bias of this lane's 4 rows of output tile m (packed by rc_chain_pack_bias: [tile][row])
__device__ __forceinline__ f32x4 bias4(const float* b, int m, int g) { return ld4(b + m * 16 + 4 * g); }

// 16-byte loads of a token row in B-operand layout (pointer p = token's first channel)
template <int C> __device__ __forceinline__ void load_act(const bf16_t* p, int g, Act<C>& a) {
#pragma unroll
    for (int s = 0; s < Act<C>::KS; ++s) a.f[s] = *reinterpret_cast<const uint4*>(p + 32 * s + 8 * g);
    if constexpr (Act<C>::TAIL) a.t = *reinterpret_cast<const uint2*>(p + 32 * Act<C>::KS + 4 * g);
}
// the same fragments from a tensor laid out planar by 16-channel segment ([C / 16][tokens][16], `plane` elements per segment):
// K-step s, lane group g -> half (g & 1) of segment 2 s + (g >> 1): a wave instruction reads two contiguous 512-byte runs
template <int C> __device__ __forceinline__ void load_act_planar(const bf16_t* base, size_t plane, size_t tok, int g, Act<C>& a) {
#pragma unroll
    for (int s = 0; s < Act<C>::KS; ++s) a.f[s] = *reinterpret_cast<const uint4*>(base + (size_t)(2 * s + (g >> 1)) * plane + tok * 16 + 8 * (g & 1));
    if constexpr (Act<C>::TAIL) a.t = *reinterpret_cast<const uint2*>(base + (size_t)(2 * Act<C>::KS) * plane + tok * 16 + 4 * g);
}
template <int C> __device__ __forceinline__ void store_act(bf16_t* p, int g, const Act<C>& a) {
#pragma unroll
    for (int s = 0; s < Act<C>::KS; ++s) *reinterpret_cast<uint4*>(p + 32 * s + 8 * g) = a.f[s];
    if constexpr (Act<C>::TAIL) *reinterpret_cast<uint2*>(p + 32 * Act<C>::KS + 4 * g) = a.t;
}

// s + the values of the three other lane groups (lanes n, n + 16, n + 32, n + 48): the two __shfl_xor steps of layernorm80 as gfx950's row / half swaps.
// Operand order differs from own + partner only by commutation, so the sums are the same bits; no lane-index registers (hipcc kept the bpermute
// addresses of __shfl_xor alive across the whole tile loop: three spilled VGPRs) and no LDS crossbar traffic.  Inline asm because the builtin's
// second result is mis-lowered by this hipcc (both results come back as the first); s_nop on both sides: the swap is opaque to the hazard recogniser.
__device__ __forceinline__ float sum_lane_groups(float s) {
    float a = s, b = s;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    s = a + b; a = s; b = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

// nn.LayerNorm over the 80 channels of each token (two-pass like rc_layernorm: mean, then centred second moment);
// gamma / beta in natural channel order at LDS address gb (gamma[80] | beta[80])
__device__ __forceinline__ void layernorm80(const Act<kC> (&in)[kNT], Act<kC> (&out)[kNT], const float* gb, int g, float eps) {
    // this lane's 20 channels: 8 g + 0..7, 32 + 8 g + 0..7, 64 + 4 g + 0..3
    const f32x4 g0 = ld4(gb + 8 * g), g1 = ld4(gb + 8 * g + 4), g2 = ld4(gb + 32 + 8 * g), g3 = ld4(gb + 32 + 8 * g + 4), g4 = ld4(gb + 64 + 4 * g);
    const float* bb = gb + kC;
    const f32x4 e0 = ld4(bb + 8 * g), e1 = ld4(bb + 8 * g + 4), e2 = ld4(bb + 32 + 8 * g), e3 = ld4(bb + 32 + 8 * g + 4), e4 = ld4(bb + 64 + 4 * g);
#pragma unroll
    for (int nt = 0; nt < kNT; ++nt) {
        const f32x4 v0 = up_lo(in[nt].f[0]), v1 = up_hi(in[nt].f[0]), v2 = up_lo(in[nt].f[1]), v3 = up_hi(in[nt].f[1]), v4 = up_tail(in[nt].t);
        const f32x4 sv = ((v0 + v1) + (v2 + v3)) + v4;
        float s = (sv[0] + sv[1]) + (sv[2] + sv[3]);
        s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);          // (sum_lane_groups here costs gma_tail<192> three spilled registers: the swaps' operands are tied pairs)
        const float mean = s / (float)kC;
        const f32x4 d0 = v0 - mean, d1 = v1 - mean, d2 = v2 - mean, d3 = v3 - mean, d4 = v4 - mean;
        const f32x4 qv = ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + d4 * d4;
        float q = (qv[0] + qv[1]) + (qv[2] + qv[3]);
        q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
        const float rstd = 1.f / sqrtf(q / (float)kC + eps);
        out[nt].f[0] = pack_pair(d0 * rstd * g0 + e0, d1 * rstd * g1 + e1);
        out[nt].f[1] = pack_pair(d2 * rstd * g2 + e2, d3 * rstd * g3 + e3);
        out[nt].t = pack_tail(d4 * rstd * g4 + e4);
    }
}

// exact-erf GELU, 0.5 v (1 + erf(v / sqrt 2)), on four values at once, written on vectors so that it compiles to packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32, two
// values per instruction): common.hpp's gelu_erf_f32 -- erf(w / sqrt 2) = w Q(2 w^2 / 25 - 1), w = clamp(v, +-5), Q of degree 12, |error| <= 6.7e-7 -- as one clamp per
// value and 8.5 packed instructions per pair: 9.5 issue slots per value.  The form it replaces (Abramowitz-Stegun 7.1.28: 6 fma, 4 squarings, a quarter-rate reciprocal,
// |v| and a sign transfer per value) took 13.5 and was 63 % of the tail kernel's vector instructions (0.6 of its 1.9 ms in the scalar form before that).
__device__ __forceinline__ f32x4 gelu_erf4(const f32x4 v) {
    const f32x4 w = f32x4{__builtin_amdgcn_fmed3f(v[0], -5.f, 5.f), __builtin_amdgcn_fmed3f(v[1], -5.f, 5.f), __builtin_amdgcn_fmed3f(v[2], -5.f, 5.f),
                          __builtin_amdgcn_fmed3f(v[3], -5.f, 5.f)};
    auto bc = [](float c) { return f32x4{c, c, c, c}; };
    const f32x4 t = __builtin_elementwise_fma(w * w, bc(0.08f), bc(-1.f));
    f32x4 q = bc(kGeluC[12]);
#pragma unroll
    for (int i = 11; i >= 0; --i) q = __builtin_elementwise_fma(q, t, bc(kGeluC[i]));
    const f32x4 hv = v * 0.5f;
    return __builtin_elementwise_fma(hv, w * q, hv);
}

// ---- LayerNorm1 + qkv ------------------------------------------------------------------------------------------------------------
// Output layout: PLANAR BY 16-CHANNEL SEGMENT, qkv[15][tokens][16] (segment = 5 which + group).  The token-major (tokens, 240) form
// made this kernel's stores 64-byte pieces at a 480-byte stride (0.52 of its 0.90 ms) and the aggregator's loads 32-byte pieces of
// 480-byte records; segment planes turn both into contiguous 512-byte runs per instruction.
constexpr int kQkvThreads = 256;
__global__ __launch_bounds__(kQkvThreads) void gma_ln_qkv_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ qkv, size_t tokens,
                                                                   const void* __restrict__ w_qkv, const float* __restrict__ b_qkv,
                                                                   const float* __restrict__ ln_g, const float* __restrict__ ln_b, float eps) {
    constexpr int MT = n_mtiles(3 * kC), TB = tile_bytes(kC);          // 15 tiles of 2560 bytes
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* s_w = lds;
    float* s_bias = reinterpret_cast<float*>(lds + MT * TB);            // [MT*16]
    float* s_gb = s_bias + MT * 16;                                     // gamma[80] | beta[80]
    const int tid = threadIdx.x;
    for (int i = tid; i < MT * TB / 16; i += kQkvThreads) reinterpret_cast<uint4*>(s_w)[i] = reinterpret_cast<const uint4*>(w_qkv)[i];
    for (int i = tid; i < MT * 16; i += kQkvThreads) s_bias[i] = b_qkv ? b_qkv[i] : 0.f;
    for (int i = tid; i < kC; i += kQkvThreads) { s_gb[i] = ln_g[i]; s_gb[kC + i] = ln_b[i]; }
    __syncthreads();
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const size_t n_tiles = (tokens + 63) / 64, wave = (size_t)blockIdx.x * (kQkvThreads / 64) + (tid >> 6), n_waves = (size_t)gridDim.x * (kQkvThreads / 64);
    // software pipeline: the next tile's x is in flight while this tile's LayerNorm + 240 MFMAs run (a wave that loads, waits,
    // computes and stores in turn left the launch at 3 TB/s)
    Act<kC> xnext[kNT];
    if (wave < n_tiles) {
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            const size_t t = wave * 64 + 16 * nt + n;
            load_act<kC>(x + (t < tokens ? t : tokens - 1) * kC, g, xnext[nt]);
        }
    }
    for (size_t tile = wave; tile < n_tiles; tile += n_waves) {
        Act<kC> xin[kNT], n1[kNT];
        size_t tok[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            const size_t t = tile * 64 + 16 * nt + n;
            tok[nt] = t < tokens ? t : tokens - 1;
            xin[nt] = xnext[nt];
        }
        if (tile + n_waves < n_tiles) {
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                const size_t t = (tile + n_waves) * 64 + 16 * nt + n;
                load_act<kC>(x + (t < tokens ? t : tokens - 1) * kC, g, xnext[nt]);
            }
        }
        layernorm80(xin, n1, s_gb, g, eps);
#pragma unroll
        for (int p = 0; p < MT / 2; ++p) {
            f32x4 acc[2][kNT];
            zero<2>(acc);
            gemm_tiles<kC, 2>(s_w, 2 * p, lane, n1, acc);
            const f32x4 b0 = bias4(s_bias, 2 * p, g), b1 = bias4(s_bias, 2 * p + 1, g);
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt)
                if (tile * 64 + 16 * nt + n < tokens)      // channels 32 p + 8 g .. + 8 = half (g & 1) of segment 2 p + (g >> 1)
                    *reinterpret_cast<uint4*>(qkv + ((size_t)(2 * p + (g >> 1)) * tokens + tok[nt]) * kSEG + 8 * (g & 1)) =
                        pack_pair(acc[0][nt] + b0, acc[1][nt] + b1);
        }
        if constexpr (MT & 1) {
            f32x4 acc[1][kNT];
            zero<1>(acc);
            gemm_tiles<kC, 1>(s_w, MT - 1, lane, n1, acc);
            const f32x4 b0 = bias4(s_bias, MT - 1, g);
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt)
                if (tile * 64 + 16 * nt + n < tokens)
                    *reinterpret_cast<uint2*>(qkv + ((size_t)(MT - 1) * tokens + tok[nt]) * kSEG + 4 * g) = pack_tail(acc[0][nt] + b0);
        }
    }
}

// ---- k^T v as MFMA A fragments: [b][out tile m (4)][K-step s (2)][lane][8 bf16], block-diagonal 64 x 64 --------------------------
__global__ void gma_ktv_pack_kernel(const float* __restrict__ ktv /* (B, 8, 8, 8) [h][i][j] */, uint4* __restrict__ frags) {
    const int b = blockIdx.x, frag = threadIdx.x >> 6, lane = threadIdx.x & 63;   // 512 threads: 8 fragments
    const int m = frag >> 1, s = frag & 1, R = lane & 15, q = lane >> 4;
    const int jo = row_channel(m, R, 4);                                          // output channel h*8 + j
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ki = 32 * s + 8 * q + i;                                        // input channel h*8 + i
        v[i] = (ki >> 3) == (jo >> 3) ? ktv[(((size_t)b * 8 + (jo >> 3)) * 8 + (ki & 7)) * 8 + (jo & 7)] : 0.f;
    }
    frags[((size_t)b * 8 + frag) * 64 + lane] = make_uint4(pk(v[0], v[1]), pk(v[2], v[3]), pk(v[4], v[5]), pk(v[6], v[7]));
}

// ---- attention read-out + proj + LN2 + MLP [+ output conv] ---------------------------------------------------------------------------
struct TailArgs {
    const bf16_t* qkvp; size_t plane;        // segment-planar [12][tokens][16]: q = segments 0..3; `plane` = tokens * 16 elements
    const bf16_t* convv;                      // segment-planar [4][tokens][16]
    const bf16_t* loc; const bf16_t* x;       // (tokens, 16), (tokens, 80)
    const bf16_t* res;                        // COUT > 0: residual of the output conv (tokens, COUT)
    bf16_t* out;                              // (tokens, COUT > 0 ? COUT : 80)
    const uint4* ktv_frags;
    const void* w_proj; const void* w_fc1; const void* w_fc2; const void* w_out;
    const float* b_proj; const float* b_fc1; const float* b_fc2; const float* b_out;
    const float* ln_g; const float* ln_b; float eps;
    int n_tok; int batch;
};

constexpr int kTailWaves = 8, kTailThreads = 64 * kTailWaves;
template <int COUT>
__host__ __device__ constexpr int tail_lds_bytes() {
    return n_mtiles(kC) * tile_bytes(kC) + n_mtiles(kHID) * tile_bytes(kC) + n_mtiles(kC) * tile_bytes(kHID) + n_mtiles(COUT) * tile_bytes(kC) +
           4 * (n_mtiles(kC) * 16 * 2 + n_mtiles(kHID) * 16 + n_mtiles(COUT) * 16 + 2 * kC);
}

template <int COUT>
__global__ __launch_bounds__(kTailThreads, 2) void gma_tail_kernel(TailArgs a) {
    constexpr int MT80 = n_mtiles(kC), MTH = n_mtiles(kHID), MTO = n_mtiles(COUT);
    constexpr int TB80 = tile_bytes(kC), TBH = tile_bytes(kHID);
    static_assert(COUT % 32 == 0, "output width must be a whole number of tile pairs");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* s_proj = lds;
    char* s_fc1 = s_proj + MT80 * TB80;
    char* s_fc2 = s_fc1 + MTH * TB80;
    char* s_out = s_fc2 + MT80 * TBH;
    float* s_bproj = reinterpret_cast<float*>(s_out + MTO * TB80);
    float* s_bfc1 = s_bproj + MT80 * 16;
    float* s_bfc2 = s_bfc1 + MTH * 16;
    float* s_bout = s_bfc2 + MT80 * 16;
    float* s_gb = s_bout + MTO * 16;
    const int tid = threadIdx.x;
    auto stage = [&](char* dst, const void* src, int bytes) {
        for (int i = tid; i < bytes / 16; i += kTailThreads) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    };
    stage(s_proj, a.w_proj, MT80 * TB80); stage(s_fc1, a.w_fc1, MTH * TB80); stage(s_fc2, a.w_fc2, MT80 * TBH);
    if constexpr (COUT > 0) stage(s_out, a.w_out, MTO * TB80);
    for (int i = tid; i < MT80 * 16; i += kTailThreads) { s_bproj[i] = a.b_proj[i]; s_bfc2[i] = a.b_fc2[i]; }
    for (int i = tid; i < MTH * 16; i += kTailThreads) s_bfc1[i] = a.b_fc1[i];
    if constexpr (COUT > 0) for (int i = tid; i < MTO * 16; i += kTailThreads) s_bout[i] = a.b_out[i];
    for (int i = tid; i < kC; i += kTailThreads) { s_gb[i] = a.ln_g[i]; s_gb[kC + i] = a.ln_b[i]; }
    __syncthreads();

    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int tpi = (a.n_tok + 63) / 64;                                  // wave tiles per image (a tile never straddles images)
    const int n_tiles = tpi * a.batch;
    for (int tile = blockIdx.x * kTailWaves + (tid >> 6); tile < n_tiles; tile += gridDim.x * kTailWaves) {
        const int b = tile / tpi, t0 = (tile - b * tpi) * 64;
        size_t tok[kNT];
        bool ok[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            const int t = t0 + 16 * nt + n;
            ok[nt] = t < a.n_tok;
            tok[nt] = (size_t)b * a.n_tok + (ok[nt] ? t : a.n_tok - 1);
        }
        Act<kC> x2[kNT];                                                  // starts as x, becomes x2 = proj(y) + x (bf16-rounded)
        Act<kC> y[kNT];
        {   // ---- attention read-out: y[h*8+j] = sum_i q[h*8+i] ktv[h][i][j] + q[h*8+j] convv[h*8+j]; y[64..80) = loc --------------
            Act<kCT> q[kNT], cv[kNT];
            Act<kSEG> lc[kNT];
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                load_act_planar<kCT>(a.qkvp, a.plane, tok[nt], g, q[nt]);
                load_act_planar<kCT>(a.convv, a.plane, tok[nt], g, cv[nt]);
                load_act<kSEG>(a.loc + tok[nt] * kSEG, g, lc[nt]);
                load_act<kC>(a.x + tok[nt] * kC, g, x2[nt]);
            }
            f32x4 att[4][kNT];
            zero<4>(att);
            const uint4* kf = a.ktv_frags + (size_t)b * 8 * 64 + lane;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint4 af = kf[(m * 2 + s) * 64];
#pragma unroll
                    for (int nt = 0; nt < kNT; ++nt) mma32(af, q[nt].f[s], att[m][nt]);
                }
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    y[nt].f[p] = pack_pair(att[2 * p][nt] + up_lo(q[nt].f[p]) * up_lo(cv[nt].f[p]),
                                           att[2 * p + 1][nt] + up_hi(q[nt].f[p]) * up_hi(cv[nt].f[p]));
                y[nt].t = lc[nt].t;
            }
        }
        Act<kC> n2[kNT];
        f32x4 oa[2][kNT], ob[2][kNT], oc[1][kNT];                         // fc2 accumulators, seeded with the residual x2 below
        {   // ---- proj + bias + x -> x2 (rounded to bf16 where the layer-by-layer path stored it); accumulators start at the residual x ----
            f32x4 pa[2][kNT], pb[2][kNT], pc[1][kNT];
            const f32x4 b0 = bias4(s_bproj, 0, g), b1 = bias4(s_bproj, 1, g), b2 = bias4(s_bproj, 2, g), b3 = bias4(s_bproj, 3, g),
                        b4 = bias4(s_bproj, 4, g);
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                pa[0][nt] = up_lo(x2[nt].f[0]); pa[1][nt] = up_hi(x2[nt].f[0]); pb[0][nt] = up_lo(x2[nt].f[1]);
                pb[1][nt] = up_hi(x2[nt].f[1]); pc[0][nt] = up_tail(x2[nt].t);
            }
            gemm_tiles<kC, 2>(s_proj, 0, lane, y, pa);
            gemm_tiles<kC, 2>(s_proj, 2, lane, y, pb);
            gemm_tiles<kC, 1>(s_proj, 4, lane, y, pc);
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                x2[nt].f[0] = pack_pair(pa[0][nt] + b0, pa[1][nt] + b1); x2[nt].f[1] = pack_pair(pb[0][nt] + b2, pb[1][nt] + b3);
                x2[nt].t = pack_tail(pc[0][nt] + b4);
                // the bf16-rounded x2 is what LayerNorm2 and the MLP residual see
                oa[0][nt] = up_lo(x2[nt].f[0]); oa[1][nt] = up_hi(x2[nt].f[0]); ob[0][nt] = up_lo(x2[nt].f[1]);
                ob[1][nt] = up_hi(x2[nt].f[1]); oc[0][nt] = up_tail(x2[nt].t);
            }
            layernorm80(x2, n2, s_gb, g, a.eps);
        }
        // ---- MLP: x2 + fc2(gelu(fc1(n2))) accumulated over 10 chunks of 32 hidden channels (= one fc2 K-step each) ----------------------
#pragma unroll 1
        for (int hc = 0; hc < kHID / 32; ++hc) {
            f32x4 h[2][kNT];
            const f32x4 c0 = bias4(s_bfc1, 2 * hc, g), c1 = bias4(s_bfc1, 2 * hc + 1, g);
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) { h[0][nt] = c0; h[1][nt] = c1; }
            gemm_tiles<kC, 2>(s_fc1, 2 * hc, lane, n2, h);
            uint4 hb[kNT];
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                const f32x4 u = h[0][nt], w = h[1][nt];
                hb[nt] = pack_pair(gelu_erf4(u), gelu_erf4(w));
            }
            const char* w2 = s_fc2 + hc * 1024 + lane * 16;
            const uint4 a0 = *reinterpret_cast<const uint4*>(w2), a1 = *reinterpret_cast<const uint4*>(w2 + TBH),
                        a2 = *reinterpret_cast<const uint4*>(w2 + 2 * TBH), a3 = *reinterpret_cast<const uint4*>(w2 + 3 * TBH),
                        a4 = *reinterpret_cast<const uint4*>(w2 + 4 * TBH);
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                mma32(a0, hb[nt], oa[0][nt]); mma32(a1, hb[nt], oa[1][nt]); mma32(a2, hb[nt], ob[0][nt]);
                mma32(a3, hb[nt], ob[1][nt]); mma32(a4, hb[nt], oc[0][nt]);
            }
        }
        Act<kC> x3[kNT];
        {
            const f32x4 c0 = bias4(s_bfc2, 0, g), c1 = bias4(s_bfc2, 1, g), c2 = bias4(s_bfc2, 2, g), c3 = bias4(s_bfc2, 3, g),
                        c4 = bias4(s_bfc2, 4, g);
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                x3[nt].f[0] = pack_pair(oa[0][nt] + c0, oa[1][nt] + c1); x3[nt].f[1] = pack_pair(ob[0][nt] + c2, ob[1][nt] + c3);
                x3[nt].t = pack_tail(oc[0][nt] + c4);
            }
        }
        if constexpr (COUT == 0) {
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt)
                if (ok[nt]) store_act<kC>(a.out + tok[nt] * kC, g, x3[nt]);
        } else {   // ---- output conv 80 -> COUT + bias + residual, one tile pair (32 channels) at a time; residual loads run one pair ahead
            uint4 rcur[kNT], rnext[kNT];
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) rcur[nt] = *reinterpret_cast<const uint4*>(a.res + tok[nt] * COUT + 8 * g);
#pragma unroll
            for (int p = 0; p < COUT / 32; ++p) {
                if (p + 1 < COUT / 32) {
#pragma unroll
                    for (int nt = 0; nt < kNT; ++nt) rnext[nt] = *reinterpret_cast<const uint4*>(a.res + tok[nt] * COUT + 32 * (p + 1) + 8 * g);
                }
                f32x4 o[2][kNT];
                const f32x4 b0 = bias4(s_bout, 2 * p, g), b1 = bias4(s_bout, 2 * p + 1, g);
#pragma unroll
                for (int nt = 0; nt < kNT; ++nt) {
                    o[0][nt] = up_lo(rcur[nt]); o[1][nt] = up_hi(rcur[nt]);
                }
                gemm_tiles<kC, 2>(s_out, 2 * p, lane, x3, o);
#pragma unroll
                for (int nt = 0; nt < kNT; ++nt) {
                    if (ok[nt]) *reinterpret_cast<uint4*>(a.out + tok[nt] * COUT + 32 * p + 8 * g) = pack_pair(o[0][nt] + b0, o[1][nt] + b1);
                    rcur[nt] = rnext[nt];
                }
            }
        }
    }
}

}  // namespace gf
}  // namespace rc

using namespace rc;
using namespace rc::gf;

extern "C" {

size_t rc_chain_packed_bytes(int cin, int cout) {
    if (cin < 16 || cin % 16 || cout < 1) return 0;
    return (size_t)n_mtiles(cout) * tile_bytes(cin);
}

int rc_chain_packed_rows(int cout) { return cout >= 1 ? n_mtiles(cout) * 16 : 0; }

int rc_chain_pack_weights(const float* w, int cin, int cout, void* dst) {
    RC_REQUIRE(w && dst, "rc_chain_pack_weights: null pointer");
    RC_REQUIRE(cin >= 16 && cin % 16 == 0 && cout >= 1, "rc_chain_pack_weights: cin must be a multiple of 16");
    const int mt = n_mtiles(cout), ks = cin / 32, tb = tile_bytes(cin);
    uint16_t* out = static_cast<uint16_t*>(dst);
    for (int m = 0; m < mt; ++m) {
        uint16_t* tile = out + (size_t)m * tb / 2;
        for (int s = 0; s < ks; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int ch = row_channel(m, lane & 15, mt), q = lane >> 4;
                for (int i = 0; i < 8; ++i)
                    tile[(s * 64 + lane) * 8 + i] = ch < cout ? host_f32_to_bf16(w[(size_t)ch * cin + 32 * s + 8 * q + i]) : 0;
            }
        if (cin % 32)
            for (int lane = 0; lane < 64; ++lane) {
                const int ch = row_channel(m, lane & 15, mt), q = lane >> 4;
                for (int i = 0; i < 4; ++i)
                    tile[ks * 512 + lane * 4 + i] = ch < cout ? host_f32_to_bf16(w[(size_t)ch * cin + 32 * ks + 4 * q + i]) : 0;
            }
    }
    return RC_OK;
}

int rc_chain_pack_bias(const float* b, int cout, float* dst) {
    RC_REQUIRE(dst && cout >= 1, "rc_chain_pack_bias: bad arguments");
    const int mt = n_mtiles(cout);
    for (int m = 0; m < mt; ++m)
        for (int R = 0; R < 16; ++R) {
            const int ch = row_channel(m, R, mt);
            dst[m * 16 + R] = (b && ch < cout) ? b[ch] : 0.f;
        }
    return RC_OK;
}

int rc_gma_ln_qkv(const void* d_x, void* d_qkv, long long tokens, const void* d_wpacked, const float* d_bias_packed,
                  const float* d_ln_gamma, const float* d_ln_beta, float eps, void* stream) {
    RC_REQUIRE(d_x && d_qkv && d_wpacked && d_ln_gamma && d_ln_beta, "rc_gma_ln_qkv: null pointer");
    RC_REQUIRE(tokens >= 1, "rc_gma_ln_qkv: no tokens");
    constexpr int MT = n_mtiles(3 * kC);
    const size_t lds = (size_t)MT * tile_bytes(kC) + 4 * (MT * 16 + 2 * kC);
    int dev = 0;
    RC_HIP_CHECK(hipGetDevice(&dev));
    static bool attr[64] = {};                                            // function attributes are per device
    if (dev >= 0 && dev < 64 && !attr[dev]) {
        RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gma_ln_qkv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        attr[dev] = true;
    }
    const size_t n_tiles = ((size_t)tokens + 63) / 64;
    size_t blocks = (n_tiles + 3) / 4;
    if (blocks > 256 * 3) blocks = 256 * 3;
    hipLaunchKernelGGL(gma_ln_qkv_kernel, dim3((unsigned)blocks), dim3(kQkvThreads), lds, as_stream(stream), static_cast<const bf16_t*>(d_x),
                       static_cast<bf16_t*>(d_qkv), (size_t)tokens, d_wpacked, d_bias_packed, d_ln_gamma, d_ln_beta, eps);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_gma_tail(const void* d_qkvp, const void* d_convv, const void* d_loc, const void* d_x, const float* d_ktv, void* d_ktv_frags,
                int batch, int n_tok, const void* d_w_proj, const float* d_b_proj, const float* d_ln_gamma, const float* d_ln_beta,
                float eps, const void* d_w_fc1, const float* d_b_fc1, const void* d_w_fc2, const float* d_b_fc2, const void* d_res,
                const void* d_w_out, const float* d_b_out, int cout, void* d_out, void* stream) {
    RC_REQUIRE(d_qkvp && d_convv && d_loc && d_x && d_ktv && d_ktv_frags && d_w_proj && d_b_proj && d_ln_gamma && d_ln_beta && d_w_fc1 &&
               d_b_fc1 && d_w_fc2 && d_b_fc2 && d_out, "rc_gma_tail: null pointer");
    RC_REQUIRE(batch >= 1 && n_tok >= 1, "rc_gma_tail: bad shape");
    RC_REQUIRE(cout == 0 || cout == 192, "rc_gma_tail: the output conv is built for 0 (none) or 192 channels");
    if (cout) RC_REQUIRE(d_res && d_w_out && d_b_out, "rc_gma_tail: output conv needs residual, weights and bias");
    TailArgs a;
    a.qkvp = static_cast<const bf16_t*>(d_qkvp); a.plane = (size_t)batch * n_tok * kSEG;
    a.convv = static_cast<const bf16_t*>(d_convv); a.loc = static_cast<const bf16_t*>(d_loc); a.x = static_cast<const bf16_t*>(d_x);
    a.res = static_cast<const bf16_t*>(d_res); a.out = static_cast<bf16_t*>(d_out); a.ktv_frags = static_cast<const uint4*>(d_ktv_frags);
    a.w_proj = d_w_proj; a.w_fc1 = d_w_fc1; a.w_fc2 = d_w_fc2; a.w_out = d_w_out;
    a.b_proj = d_b_proj; a.b_fc1 = d_b_fc1; a.b_fc2 = d_b_fc2; a.b_out = d_b_out;
    a.ln_g = d_ln_gamma; a.ln_b = d_ln_beta; a.eps = eps; a.n_tok = n_tok; a.batch = batch;
    hipLaunchKernelGGL(gma_ktv_pack_kernel, dim3(batch), dim3(512), 0, as_stream(stream), d_ktv, static_cast<uint4*>(d_ktv_frags));
    const long long n_tiles = (long long)((n_tok + 63) / 64) * batch;
    int dev = 0;
    RC_HIP_CHECK(hipGetDevice(&dev));
    RC_REQUIRE(dev >= 0 && dev < 64, "rc_gma_tail: device index out of range");
    static int cus[64] = {};
    if (!cus[dev]) RC_HIP_CHECK(hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev));
    const int num_cus = cus[dev];
    long long blocks = (n_tiles + kTailWaves - 1) / kTailWaves;
    if (blocks > num_cus) blocks = num_cus;                               // one 8-wave block per CU (154 KB of LDS), grid-stride over tiles
#define RC_TAIL_LAUNCH(CO)                                                                                                          \
    do {                                                                                                                            \
        static bool attr[64] = {};                                                                                                  \
        if (!attr[dev]) {                                                                                                           \
            RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gma_tail_kernel<CO>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                             160 * 1024));                                                                          \
            attr[dev] = true;                                                                                                       \
        }                                                                                                                           \
        hipLaunchKernelGGL((gma_tail_kernel<CO>), dim3((unsigned)blocks), dim3(kTailThreads), tail_lds_bytes<CO>(), as_stream(stream), a); \
    } while (0)
    if (cout == 0) RC_TAIL_LAUNCH(0); else RC_TAIL_LAUNCH(192);
#undef RC_TAIL_LAUNCH
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

}  // extern "C"

// ======================================================================================================================================
// Aggregator (upstream models/groupmix.py:56-105) as ONE kernel: depth-wise KxK -> point-wise 16x16 (MFMA) -> BatchNorm(eval) ->
// Hardswish for the three conv groups of q, k and v, the pass-through group, and the local branch (dw 3x3 on 3x16 channels ->
// 48 -> 16 point-wise -> LayerNorm(16) -> Hardswish).  The layer-by-layer form wrote the depth-wise results (1.6 GB at cfg3) and read
// them back, staged every channel with a 7x7 halo, and spent 2.0 + 1.6 ms; here a block owns (16 x 32 pixel tile, one of q / k / v /
// local), stages one 16-channel segment at a time with ITS window's halo, and the depth-wise results never leave registers:
// lane (n, q) owns channels 4 q .. 4 q + 3 of a 2 x 4 pixel patch, its packed bf16 results are exactly the B fragments of
// a half-filled K-step (K = the segment's 16 channels), one MFMA per patch pixel, and the C fragment (4 output channels of the
// same pixel) goes through BN + Hardswish to an 8-byte store.  A lane's 16-byte LDS reads are reused by 4 x K taps.
// Depth-wise accumulation order (dy, dx ascending, fmaf) and the bf16 rounding point are those of rc_dwconv2d.
namespace rc {
namespace gf {

constexpr int AG_TH = 16, AG_TW = 32, AG_THREADS = 256;
constexpr int AG_PS = 40;                                                  // LDS pixel stride in bytes (32 of data): 4 pixels = 40 dwords,
                                                                           // so the 8 column groups of a wave hit 8 different bank octets
constexpr int AG_RPAD = 8;                                                 // + 8 bytes per halo row: the two row pairs of a wave (lanes n, n + 8)
                                                                           // then sit 4 dwords apart modulo 8 instead of on the same banks
constexpr int AG_LDS = (AG_TH + 6) * ((AG_TW + 6) * AG_PS + AG_RPAD);

// float max through the integer atomics (any finite / -inf start value): non-negative floats order like ints, negative ones like
// reversed unsigneds.  A maximum does not depend on the order of its operands, so the result is deterministic.
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}

struct AggArgs {
    const bf16_t* qkv; bf16_t* qkvp; bf16_t* loc;
    float* kmax;                 // optional (B, 64): per-channel maximum of the aggregated k over the image (the softmax_N(k) shift of rc_gma_kv)
    int batch, H, W, tiles_x, tiles_y;
    size_t plane;                // elements per segment plane = batch * H * W * 16
    const float* dw[3];          // groups 1..3: tap-major [K*K][16], K = 3, 5, 7
    const float* dwl;            // local branch: [3 (q,k,v)][9][16]
    const float* pw;             // [3][16 out][16 in]
    const float* pwl;            // [16 out][48 in]
    const float* bn_scale; const float* bn_shift;   // [4][16]
    const float* ln_g; const float* ln_b;           // [16]
};

__device__ __forceinline__ float hswish(float x) {
    const float r = fminf(fmaxf(x + 3.f, 0.f), 6.f);
    return x * r * (1.f / 6.f);
}

// stage channels [c0, c0 + 16) of the (TH + 2R) x (TW + 2R) halo tile (pixel stride `cs` channels), zero outside the image
template <int R>
__device__ __forceinline__ void agg_stage(char* s_x, const bf16_t* img, int cs, int c0, int y0, int x0, int H, int W, int tid) {
    constexpr int THH = AG_TH + 2 * R, TWH = AG_TW + 2 * R, NPIECE = THH * TWH * 2, NLD = (NPIECE + AG_THREADS - 1) / AG_THREADS;
    uint4 raw[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = tid + k * AG_THREADS, pix = i >> 1, half = i & 1;
        const int py = pix / TWH, px = pix - py * TWH;
        const int gy = y0 + py - R, gx = x0 + px - R;
        raw[k] = make_uint4(0u, 0u, 0u, 0u);
        if (i < NPIECE && gy >= 0 && gy < H && gx >= 0 && gx < W)
            raw[k] = *reinterpret_cast<const uint4*>(img + ((size_t)gy * W + gx) * cs + c0 + 8 * half);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = tid + k * AG_THREADS;
        if (i < NPIECE) {
            const int pix = i >> 1, py = pix / TWH;
            *reinterpret_cast<uint4*>(s_x + pix * AG_PS + py * AG_RPAD + (i & 1) * 16) = raw[k];
        }
    }
}

// depth-wise KxK of this lane's 2 x 4 patch, channels 4 q .. 4 q + 3: acc[o][c] (fp32, taps in (dy, dx) order)
template <int K>
__device__ __forceinline__ void agg_dw(const char* s_x, const float* s_w, int prow, int pcol, int q, f32x4 (&acc)[2][4]) {
    constexpr int R = K / 2, TWH = AG_TW + 2 * R, RS = TWH * AG_PS + AG_RPAD;
    const char* base = s_x + prow * RS + pcol * AG_PS + q * 8;
#pragma unroll 1                                                           // rolled: fully unrolled, all 4 K^2 tap vectors were hoisted (256+ VGPRs)
    for (int iy = 0; iy < K + 1; ++iy) {
        f32x4 xin[K + 3];
#pragma unroll
        for (int c = 0; c < K + 3; ++c) xin[c] = up_tail(*reinterpret_cast<const uint2*>(base + iy * RS + c * AG_PS));
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int dy = iy - o;
            if (dy < 0 || dy >= K) continue;
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
                const f32x4 w = ld4(s_w + (dy * K + dx) * 16 + 4 * q);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[o][c][e] = __builtin_fmaf(w[e], xin[c + dx][e], acc[o][c][e]);
            }
        }
    }
}

// A fragment (16 out x 16 in, lower half of a K-step) of an fp32 row-major matrix with row stride ld, starting at column col0
__device__ __forceinline__ uint2 agg_afrag(const float* w, int ld, int col0, int lane) {
    const float* p = w + (lane & 15) * ld + col0 + 4 * (lane >> 4);
    return make_uint2(pk(p[0] + 0.f, p[1] + 0.f), pk(p[2] + 0.f, p[3] + 0.f));
}

struct AggGeom { int y0, x0, prow, pcol, q, b; size_t pix0; };

// block-wide maximum of 16 channels (lane group q holds channels 4 q .. 4 q + 3) -> one atomic per channel
__device__ __forceinline__ void agg_block_max(float* dst, f32x4 m, float* s_red, int tid, int lane) {
#pragma unroll
    for (int sh = 1; sh < 16; sh <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], __shfl_xor(m[e], sh));
    __syncthreads();                                                       // s_red aliases the tap table: every wave is done with it
    if ((lane & 15) == 0) *reinterpret_cast<float4*>(s_red + (tid >> 6) * 16 + 4 * (lane >> 4)) = make_float4(m[0], m[1], m[2], m[3]);
    __syncthreads();
    if (tid < 16) atomic_max_f32(dst + tid, fmaxf(fmaxf(s_red[tid], s_red[16 + tid]), fmaxf(s_red[32 + tid], s_red[48 + tid])));
}

// one conv group: stage -> depth-wise K x K -> point-wise (MFMA) -> BN + Hardswish -> qkvp[.., which, 16 g + ..]
template <int K>
__device__ __forceinline__ void agg_conv_job(const AggArgs& a, char* s_x, float* s_w, const bf16_t* img, const AggGeom& t, int which, int g,
                                             int tid, int lane) {
    constexpr int R = K / 2;
    for (int i = tid; i < K * K * 16; i += AG_THREADS) s_w[i] = a.dw[g - 1][i];
    agg_stage<R>(s_x, img + (size_t)(5 * which + g) * a.plane, kSEG, 0, t.y0, t.x0, a.H, a.W, tid);
    const uint2 apw = agg_afrag(a.pw + (g - 1) * 256, 16, 0, lane);
    const f32x4 sc = ld4(a.bn_scale + 16 * g + 4 * t.q), sh = ld4(a.bn_shift + 16 * g + 4 * t.q);
    __syncthreads();
    f32x4 acc[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[o][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    agg_dw<K>(s_x, s_w, t.prow, t.pcol, t.q, acc);
    bf16_t* outp = a.qkvp + (size_t)(4 * which + g) * a.plane + 4 * t.q;   // segment-planar [12][tokens][16]
    f32x4 vmax = f32x4{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
            mma16(apw, qa_pack(acc[o][c]), d);
            f32x4 v = d * sc + sh;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = hswish(v[e]);
            if (t.y0 + t.prow + o < a.H && t.x0 + t.pcol + c < a.W) {
                const uint2 pkd = pack_tail(v);
                *reinterpret_cast<uint2*>(outp + (t.pix0 + (size_t)o * a.W + c) * kSEG) = pkd;
                const f32x4 r = up_tail(pkd);                                  // the stored (bf16) values are what rc_gma_kv sees
#pragma unroll
                for (int e = 0; e < 4; ++e) vmax[e] = fmaxf(vmax[e], r[e]);
            }
        }
    if (which == 1 && a.kmax != nullptr) agg_block_max(a.kmax + (size_t)t.b * 64 + 16 * g, vmax, s_w, tid, lane);
}

// Blocks = tiles x 10 jobs: (which, group 3 / 2 / 1) x 3, the group-1 job also doing the pass-through group 0, and the local branch.
// One short job per block: with 4 blocks per CU in different phases the staging loads of one hide under the FMAs of another
// (one block walking all of a tile's groups, barrier to barrier, ran at a sixth of its FMA time).
__global__ __launch_bounds__(AG_THREADS) void gma_agg_kernel(AggArgs a) {
    __shared__ __attribute__((aligned(16))) char s_x[AG_LDS];
    __shared__ __attribute__((aligned(16))) float s_w[49 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15;
    // block -> (tile, job): a tile's 10 jobs are placed 8 blocks apart so that they land on ONE XCD (block b runs on XCD b % 8) close in
    // time: their outputs interleave in the same 384-byte qkvp pixel records, which that XCD's L2 then completes line by line
    int blk = blockIdx.x;
    const int lane8 = blk & 7; blk >>= 3;
    const int job = blk % 10; blk = (blk / 10) * 8 + lane8;
    if (blk >= a.tiles_x * a.tiles_y * a.batch) return;
    const int tx = blk % a.tiles_x; blk /= a.tiles_x;
    const int ty = blk % a.tiles_y;
    const int b = blk / a.tiles_y;
    AggGeom t;
    t.y0 = ty * AG_TH; t.x0 = tx * AG_TW; t.q = lane >> 4;
    t.prow = 4 * wave + 2 * (n >> 3); t.pcol = 4 * (n & 7);               // this lane's patch: 2 rows x 4 columns of the tile
    t.pix0 = ((size_t)b * a.H + t.y0 + t.prow) * a.W + t.x0 + t.pcol; t.b = b;
    const bf16_t* img = a.qkv + (size_t)b * a.H * a.W * kSEG;             // + segment * plane
    const int q = t.q;

    if (job < 9) {
        const int which = job / 3, g = 3 - job % 3;
        if (g == 3) agg_conv_job<7>(a, s_x, s_w, img, t, which, g, tid, lane);
        else if (g == 2) agg_conv_job<5>(a, s_x, s_w, img, t, which, g, tid, lane);
        else {
            agg_conv_job<3>(a, s_x, s_w, img, t, which, g, tid, lane);
            // pass-through group 0: BatchNorm + Hardswish of the segment itself
            const f32x4 sc = ld4(a.bn_scale + 4 * q), sh = ld4(a.bn_shift + 4 * q);
            f32x4 vmax = f32x4{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (t.y0 + t.prow + o < a.H && t.x0 + t.pcol + c < a.W) {
                        const size_t p = t.pix0 + (size_t)o * a.W + c;
                        f32x4 v = up_tail(*reinterpret_cast<const uint2*>(a.qkv + (size_t)(5 * which) * a.plane + p * kSEG + 4 * q)) * sc + sh;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = hswish(v[e]);
                        const uint2 pkd = pack_tail(v);
                        *reinterpret_cast<uint2*>(a.qkvp + (size_t)(4 * which) * a.plane + p * kSEG + 4 * q) = pkd;
                        const f32x4 r = up_tail(pkd);
#pragma unroll
                        for (int e = 0; e < 4; ++e) vmax[e] = fmaxf(vmax[e], r[e]);
                    }
            if (which == 1 && a.kmax != nullptr) agg_block_max(a.kmax + (size_t)b * 64, vmax, s_w, tid, lane);
        }
    } else {
        f32x4 d[2][4];
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int c = 0; c < 4; ++c) d[o][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int wh = 0; wh < 3; ++wh) {
            __syncthreads();
            for (int i = tid; i < 9 * 16; i += AG_THREADS) s_w[i] = a.dwl[wh * 144 + i];
            agg_stage<1>(s_x, img + (size_t)(5 * wh + 4) * a.plane, kSEG, 0, t.y0, t.x0, a.H, a.W, tid);
            const uint2 apw = agg_afrag(a.pwl, 48, 16 * wh, lane);
            __syncthreads();
            f32x4 acc[2][4];
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[o][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            agg_dw<3>(s_x, s_w, t.prow, t.pcol, q, acc);
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int c = 0; c < 4; ++c) mma16(apw, qa_pack(acc[o][c]), d[o][c]);
        }
        const f32x4 lg = ld4(a.ln_g + 4 * q), lb = ld4(a.ln_b + 4 * q);
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 tt = d[o][c] + 0.f;
                float s = (tt[0] + tt[1]) + (tt[2] + tt[3]);
                s = sum_lane_groups(s);
                const float mean = s / 16.f;
                const f32x4 dd = tt - mean;
                const f32x4 d2 = dd * dd;
                float var = (d2[0] + d2[1]) + (d2[2] + d2[3]);
                var = sum_lane_groups(var);
                const float rstd = 1.f / sqrtf(var / 16.f + 1e-5f);
                f32x4 v = dd * rstd * lg + lb;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = hswish(v[e]);
                if (t.y0 + t.prow + o < a.H && t.x0 + t.pcol + c < a.W)
                    *reinterpret_cast<uint2*>(a.loc + (t.pix0 + (size_t)o * a.W + c) * kSEG + 4 * q) = pack_tail(v);
            }
    }
}

// ---- ConvRelPosEnc's depth-wise conv of v (groupmix.py:108-156): 64 channels in four 16-channel segments with windows 3, 5, 7, 7
// (segment 2 mixes heads of window 5 and 7: its taps are zero-padded to 7 x 7), + bias.  Same tiles, staging and FMA core. ------------
struct CrpeArgs {
    const bf16_t* qkvp; bf16_t* convv; int batch, H, W, tiles_x, tiles_y; size_t plane;
    const float* taps[4];        // tap-major [K*K][16] per segment, K = 3, 5, 7, 7
    const float* bias;           // [64]
};

template <int K>
__device__ __forceinline__ void crpe_job(const CrpeArgs& a, char* s_x, float* s_w, const bf16_t* img, const AggGeom& t, int seg, int tid) {
    constexpr int R = K / 2;
    for (int i = tid; i < K * K * 16; i += AG_THREADS) s_w[i] = a.taps[seg][i];
    agg_stage<R>(s_x, img + (size_t)(8 + seg) * a.plane, kSEG, 0, t.y0, t.x0, a.H, a.W, tid);          // v = segments 8..11
    const f32x4 bv = ld4(a.bias + 16 * seg + 4 * t.q);
    __syncthreads();
    f32x4 acc[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[o][c] = bv;
    agg_dw<K>(s_x, s_w, t.prow, t.pcol, t.q, acc);
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (t.y0 + t.prow + o < a.H && t.x0 + t.pcol + c < a.W)
                *reinterpret_cast<uint2*>(a.convv + (size_t)seg * a.plane + (t.pix0 + (size_t)o * a.W + c) * kSEG + 4 * t.q) = pack_tail(acc[o][c] + 0.f);
}

__global__ __launch_bounds__(AG_THREADS) void gma_crpe_kernel(CrpeArgs a) {
    __shared__ __attribute__((aligned(16))) char s_x[AG_LDS];
    __shared__ __attribute__((aligned(16))) float s_w[49 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15;
    int blk = blockIdx.x;                                                   // a tile's 4 segments on one XCD (see gma_agg_kernel)
    const int lane8 = blk & 7; blk >>= 3;
    const int seg = 3 - (blk & 3); blk = (blk >> 2) * 8 + lane8;            // heavy segments first
    if (blk >= a.tiles_x * a.tiles_y * a.batch) return;
    const int tx = blk % a.tiles_x; blk /= a.tiles_x;
    const int ty = blk % a.tiles_y;
    const int b = blk / a.tiles_y;
    AggGeom t;
    t.y0 = ty * AG_TH; t.x0 = tx * AG_TW; t.q = lane >> 4;
    t.prow = 4 * wave + 2 * (n >> 3); t.pcol = 4 * (n & 7);
    t.pix0 = ((size_t)b * a.H + t.y0 + t.prow) * a.W + t.x0 + t.pcol;
    const bf16_t* img = a.qkvp + (size_t)b * a.H * a.W * kSEG;
    if (seg == 0) crpe_job<3>(a, s_x, s_w, img, t, seg, tid);
    else if (seg == 1) crpe_job<5>(a, s_x, s_w, img, t, seg, tid);
    else crpe_job<7>(a, s_x, s_w, img, t, seg, tid);
}

}  // namespace gf
}  // namespace rc

extern "C" int rc_gma_aggregate(const void* d_qkv, void* d_qkvp, void* d_loc, int batch, int H, int W, const float* d_dw3, const float* d_dw5,
                                const float* d_dw7, const float* d_dwl, const float* d_pw, const float* d_pwl, const float* d_bn_scale,
                                const float* d_bn_shift, const float* d_ln_g, const float* d_ln_b, float* d_kmax, void* stream) {
    using namespace rc;
    using namespace rc::gf;
    RC_REQUIRE(d_qkv && d_qkvp && d_loc && d_dw3 && d_dw5 && d_dw7 && d_dwl && d_pw && d_pwl && d_bn_scale && d_bn_shift && d_ln_g && d_ln_b,
               "rc_gma_aggregate: null pointer");
    RC_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "rc_gma_aggregate: bad shape");
    AggArgs a;
    a.qkv = static_cast<const bf16_t*>(d_qkv); a.qkvp = static_cast<bf16_t*>(d_qkvp); a.loc = static_cast<bf16_t*>(d_loc);
    a.batch = batch; a.H = H; a.W = W; a.tiles_x = ceil_div(W, AG_TW); a.tiles_y = ceil_div(H, AG_TH);
    a.plane = (size_t)batch * H * W * kSEG;
    a.dw[0] = d_dw3; a.dw[1] = d_dw5; a.dw[2] = d_dw7; a.dwl = d_dwl; a.pw = d_pw; a.pwl = d_pwl;
    a.bn_scale = d_bn_scale; a.bn_shift = d_bn_shift; a.ln_g = d_ln_g; a.ln_b = d_ln_b;
    a.kmax = d_kmax;
    if (d_kmax) RC_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_kmax), (int)0xff800000u /* -inf */, (size_t)batch * 64, as_stream(stream)));
    const size_t blocks = (((size_t)a.tiles_x * a.tiles_y * batch + 7) / 8) * 8 * 10;
    RC_REQUIRE(blocks < (1ull << 31), "rc_gma_aggregate: too many tiles");
    hipLaunchKernelGGL(gma_agg_kernel, dim3((unsigned)blocks), dim3(AG_THREADS), 0, as_stream(stream), a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

extern "C" int rc_gma_crpe(const void* d_qkvp, void* d_convv, int batch, int H, int W, const float* d_taps0, const float* d_taps1,
                           const float* d_taps2, const float* d_taps3, const float* d_bias, void* stream) {
    using namespace rc;
    using namespace rc::gf;
    RC_REQUIRE(d_qkvp && d_convv && d_taps0 && d_taps1 && d_taps2 && d_taps3 && d_bias, "rc_gma_crpe: null pointer");
    RC_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "rc_gma_crpe: bad shape");
    CrpeArgs a;
    a.qkvp = static_cast<const bf16_t*>(d_qkvp); a.convv = static_cast<bf16_t*>(d_convv);
    a.batch = batch; a.H = H; a.W = W; a.tiles_x = ceil_div(W, AG_TW); a.tiles_y = ceil_div(H, AG_TH);
    a.plane = (size_t)batch * H * W * kSEG;
    a.taps[0] = d_taps0; a.taps[1] = d_taps1; a.taps[2] = d_taps2; a.taps[3] = d_taps3; a.bias = d_bias;
    const size_t blocks = (((size_t)a.tiles_x * a.tiles_y * batch + 7) / 8) * 8 * 4;
    RC_REQUIRE(blocks < (1ull << 31), "rc_gma_crpe: too many tiles");
    hipLaunchKernelGGL(gma_crpe_kernel, dim3((unsigned)blocks), dim3(AG_THREADS), 0, as_stream(stream), a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

// ======================================================================================================================================
// LayerNorm1 + qkv Linear + Aggregator as ONE launch (rc_gma_qkv_aggregate; groupmix.py:178 after :293, then :56-105), with the
// depth-wise convolutions ON THE MATRIX CORES.
// rc_gma_ln_qkv wrote the 240-channel qkv map (2.0 GB at cfg3) and rc_gma_aggregate read it back with halos: 6.4 GB for the pair, and the
// aggregator's K x K windows were 0.5 G VALU instructions (its SIMDs 64 % VALU-busy).  Here a block of 16 waves owns a 16 x 32 pixel tile:
//   L   x (80 channels) of the tile + halo, 22 rows x 40 pixel slots, ONCE -> LayerNorm1 -> MFMA B fragments in registers (a lane's 4 column tiles
//       are 4 consecutive pixels of one row: 40 VGPRs);
//   then per 16-channel segment (15 = 5 x {q, k, v}; the local branch's three first):
//   P1  a 16-row slice of the qkv GEMM (3 K-steps per column tile) -> bias -> bf16 -> LDS tile S, CHANNEL-PLANAR [16][22 rows][40 slots] (zero
//       outside the image = the convolutions' zero padding); a lane packs 4 pixels of a channel into one 8-byte write;
//   P2  depth-wise K x K as MFMAs: for channel c and kernel row dy, D[ox][y] += T(c, dy)[ox][slot] . S[c][y + dy][slot] with T the banded
//       Toeplitz matrix of the 7 (5, 3) taps of that row (16 outputs x 32 input slots; rc_gma_toeplitz_pack builds the fragments on the host,
//       a wave streams its channel's K fragments from L2 one segment ahead) and the B fragment one ds_read_b128 (8 consecutive slots of a
//       row): K MFMAs per (channel, 16-pixel group) instead of 16 K^2 packed FMAs + unpacking; products are exact (bf16 x bf16), sums fp32;
//       D (4 pixels of a channel per lane) -> bf16 -> LDS tile R, PIXEL-MAJOR;
//   P3  point-wise 16 x 16 (MFMA, B = 8 bytes of R) -> BatchNorm -> Hardswish -> 8-byte stores (512 contiguous bytes per wave instruction);
//       k's per-channel maximum by DPP row reductions.  The pass-through group skips P2 (P1 writes R), the local branch accumulates its
//       48 -> 16 point-wise product over q / k / v in registers and finishes with LayerNorm(16) + Hardswish.
// qkv never exists in HBM: x in (0.67 GB + halo re-reads through L2), qkv' + loc out (1.7 GB).  Rounding points are those of the two-launch
// path; the depth-wise sums are formed in the matrix pipe's order instead of (dy, dx) fmaf order (fp32 either way).
namespace rc {
namespace gf {

constexpr int QA_WAVES = 16, QA_THREADS = 64 * QA_WAVES;                                        // 4 waves per SIMD, <= 128 VGPRs each
constexpr int QT_ROWS = AG_TH + 6, QT_SLOTS = 40, QT_SR = QT_SLOTS * 2, QT_SP = QT_ROWS * QT_SR;   // slot s of a row = pixel x0 - 4 + s; 80-byte rows, 1760-byte channel planes
constexpr int QT_S = 16 * QT_SP + 64;                                                           // + zeroed tail (the last channel's last-row window reads 16 bytes past)
constexpr int QT_RR = AG_TW * AG_PS + 8, QT_R = AG_TH * QT_RR;                                  // pixel-major tile: 40-byte pixels, 1288-byte rows
constexpr int QT_QUADS = QT_ROWS * (QT_SLOTS / 4), QT_QPW = (QT_QUADS + QA_WAVES - 1) / QA_WAVES;   // 220 quads of 4 slots, 14 per wave (one per lane n)
constexpr int QA_WSEG = tile_bytes(kC);                                                         // 2560 bytes of A fragments per segment
constexpr int QA_OFF_R = 2 * QT_S, QA_OFF_W = QA_OFF_R + 2 * QT_R, QA_OFF_BQ = QA_OFF_W + 15 * QA_WSEG, QA_OFF_GB = QA_OFF_BQ + 240 * 4,
              QA_OFF_BN = QA_OFF_GB + 160 * 4, QA_OFF_PW = QA_OFF_BN + 160 * 4, QA_OFF_RED = QA_OFF_PW + 6 * 64 * 8,
              QA_LDS = QA_OFF_RED + 4 * QA_WAVES * 16 * 4;
static_assert(QT_S % 16 == 0 && QT_R % 16 == 0 && QA_OFF_W % 16 == 0 && QA_OFF_PW % 8 == 0 && QT_QPW <= 16 && QA_WAVES == AG_TH && QA_LDS <= 160 * 1024,
              "qkv + aggregate LDS layout");
// Toeplitz fragment table (rc_gma_toeplitz_pack): [group][dy][channel][lane] 16 bytes; groups: K = 3, 5, 7, then the local branch's q / k / v (K = 3)
constexpr int QT_TAB_K3 = 0, QT_TAB_K5 = QT_TAB_K3 + 3 * 16 * 1024, QT_TAB_K7 = QT_TAB_K5 + 5 * 16 * 1024, QT_TAB_LOC = QT_TAB_K7 + 7 * 16 * 1024,
              QT_TAB_BYTES = QT_TAB_LOC + 3 * 3 * 16 * 1024;

struct QaArgs {
    const bf16_t* x; bf16_t* qkvp; bf16_t* loc; float* kmax;
    int batch, H, W, tiles_x, tiles_y, n_tiles, tiles_per_block;
    size_t plane;
    const void* wq; const float* bq;                 // rc_chain_pack_weights_natural(80 -> 240) fragments; bias [240] or NULL
    const float* ln1_g; const float* ln1_b; float eps;
    const char* toep;                                // rc_gma_toeplitz_pack
    const float* pw; const float* pwl;
    const float* bn_scale; const float* bn_shift; const float* ln_g; const float* ln_b;
};

// one token column tile of layernorm80 (same expressions, same order; gamma / beta fetched where they are used: 128 VGPRs to live in)
__device__ __forceinline__ void ln80_one(const Act<kC>& in, Act<kC>& out, const float* gb, int g, float eps) {
    const float* bb = gb + kC;
    const f32x4 v0 = up_lo(in.f[0]), v1 = up_hi(in.f[0]), v2 = up_lo(in.f[1]), v3 = up_hi(in.f[1]), v4 = up_tail(in.t);
    const f32x4 sv = ((v0 + v1) + (v2 + v3)) + v4;
    const float s = sum_lane_groups((sv[0] + sv[1]) + (sv[2] + sv[3]));
    const float mean = s / (float)kC;
    const f32x4 d0 = v0 - mean, d1 = v1 - mean, d2 = v2 - mean, d3 = v3 - mean, d4 = v4 - mean;
    const f32x4 qv = ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + d4 * d4;
    const float q = sum_lane_groups((qv[0] + qv[1]) + (qv[2] + qv[3]));
    const float rstd = 1.f / sqrtf(q / (float)kC + eps);
    const uint2 p0 = qa_pack(d0 * rstd * ld4(gb + 8 * g) + ld4(bb + 8 * g)), p1 = qa_pack(d1 * rstd * ld4(gb + 8 * g + 4) + ld4(bb + 8 * g + 4));
    out.f[0] = make_uint4(p0.x, p0.y, p1.x, p1.y);
    const uint2 p2 = qa_pack(d2 * rstd * ld4(gb + 32 + 8 * g) + ld4(bb + 32 + 8 * g)), p3 = qa_pack(d3 * rstd * ld4(gb + 32 + 8 * g + 4) + ld4(bb + 32 + 8 * g + 4));
    out.f[1] = make_uint4(p2.x, p2.y, p3.x, p3.y);
    out.t = qa_pack(d4 * rstd * ld4(gb + 64 + 4 * g) + ld4(bb + 64 + 4 * g));
}

// maximum over the 16 lanes of a DPP row (lanes with the same channel quad): xor 1, xor 2 inside quads, then the mirrored half and the mirrored row
__device__ __forceinline__ float row_max16(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)));    // quad_perm [1,0,3,2]
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true)));    // quad_perm [2,3,0,1]
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true)));   // row_half_mirror
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true)));   // row_mirror
    return v;
}

// this wave's Toeplitz fragments of a K x K group (channel = wave): T[dy].  Buffer loads: ONE 32-bit lane offset (wave * 1 KiB + lane * 16) and a scalar
// offset per fragment -- with flat pointers hipcc kept a 64-bit address per (group, dy) alive across the tile loop
typedef unsigned int qt_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int qt_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void qt_load(__amdgpu_buffer_rsrc_t tab, int group_off, int K, int voff, uint4 (&T)[7]) {
#pragma unroll
    for (int dy = 0; dy < 7; ++dy) {     // always seven loads (rows past K - 1 repeat the last one: an L1 hit): no K-dependent control flow, so T[dy] is never a copy of its old value
        const qt_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(tab, voff, group_off + (dy < K ? dy : K - 1) * 16 * 1024, 0);
        T[dy] = make_uint4(t[0], t[1], t[2], t[3]);
    }
}

// P2: depth-wise K x K of this wave's channel over the 16 x 32 tile: S (channel-planar) -> R (pixel-major, bf16).  src = S + c plane + n rows + 16 g bytes,
// dst = R + n rows + 4 g pixels + 2 c bytes (lane constants of the caller)
template <int K>
__device__ __forceinline__ void qt_dw(const char* src0, char* dst0, const uint4 (&T)[7]) {
    constexpr int Rk = K / 2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* src = src0 + (3 - Rk) * QT_SR + 32 * h;
#pragma unroll
        for (int dy = 0; dy < K; ++dy) mma32(T[dy], *reinterpret_cast<const uint4*>(src + dy * QT_SR), d);
        const uint32_t p01 = qa_pk(d[0], d[1]), p23 = qa_pk(d[2], d[3]);
        char* dst = dst0 + 16 * h * AG_PS;
        *reinterpret_cast<uint16_t*>(dst) = (uint16_t)(p01 & 0xffffu);
        *reinterpret_cast<uint16_t*>(dst + AG_PS) = (uint16_t)(p01 >> 16);
        *reinterpret_cast<uint16_t*>(dst + 2 * AG_PS) = (uint16_t)(p23 & 0xffffu);
        *reinterpret_cast<uint16_t*>(dst + 3 * AG_PS) = (uint16_t)(p23 >> 16);
    }
}

__global__ __launch_bounds__(QA_THREADS) void gma_qkv_agg_kernel(QaArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* s_wq = lds + QA_OFF_W;
    float* s_bq = reinterpret_cast<float*>(lds + QA_OFF_BQ);
    float* s_gb = reinterpret_cast<float*>(lds + QA_OFF_GB);
    float* s_bn = reinterpret_cast<float*>(lds + QA_OFF_BN);                // scale [64] | shift [64] | ln_g [16] | ln_b [16]
    uint2* s_pw = reinterpret_cast<uint2*>(lds + QA_OFF_PW);                // A fragments: groups 1..3, local q / k / v
    float* s_red = reinterpret_cast<float*>(lds + QA_OFF_RED);              // [4 groups][16 waves][16 channels]: row maxima of the aggregated k
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // the wave index as a scalar: everything derived from it stays in SGPRs
    const int lane = tid & 63, g0 = lane >> 4;
    int n = lane & 15;
    for (int i = tid; i < 15 * QA_WSEG / 16; i += QA_THREADS) reinterpret_cast<uint4*>(s_wq)[i] = reinterpret_cast<const uint4*>(a.wq)[i];
    for (int i = tid; i < 240; i += QA_THREADS) s_bq[i] = a.bq ? a.bq[i] : 0.f;
    for (int i = tid; i < kC; i += QA_THREADS) { s_gb[i] = a.ln1_g[i]; s_gb[kC + i] = a.ln1_b[i]; }
    for (int i = tid; i < 64; i += QA_THREADS) { s_bn[i] = a.bn_scale[i]; s_bn[64 + i] = a.bn_shift[i]; }
    if (tid < 16) { s_bn[128 + tid] = a.ln_g[tid]; s_bn[144 + tid] = a.ln_b[tid]; }
    for (int i = tid; i < 6 * 64; i += QA_THREADS) {
        const int f = i >> 6;
        s_pw[i] = f < 3 ? agg_afrag(a.pw + f * 256, 16, 0, i & 63) : agg_afrag(a.pwl, 48, 16 * (f - 3), i & 63);
    }
    for (int i = tid; i < (2 * QT_S + 2 * QT_R) / 16; i += QA_THREADS) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);   // finite everywhere: zero-weighted slots are still multiplied
    __syncthreads();

    const __amdgpu_buffer_rsrc_t r_toep = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.toep), 0, QT_TAB_BYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_loc = __builtin_amdgcn_make_buffer_rsrc(a.loc, 0, (int)(a.plane * 2), 0x00020000);
    // this lane's quad of the halo tile: 4 consecutive slots of one row (the same for all four g: they hold different channels of the same pixels)
    int qrc, idm = 0;                                                        // ((row << 8) | quad column) << 4 | which of its 4 pixels lie in the tile proper; -1 if this lane has no quad
    {
        const int quad = wave * QT_QPW + n;
        const bool has_quad = n < QT_QPW && quad < QT_QUADS;
        const int qrow = quad / (QT_SLOTS / 4), qcol = quad - qrow * (QT_SLOTS / 4);
        if (has_quad && qrow >= 3 && qrow < 3 + AG_TH)
            for (int i = 0; i < 4; ++i) idm |= (4 * qcol + i - 4 >= 0 && 4 * qcol + i - 4 < AG_TW) ? 1 << i : 0;
        qrc = has_quad ? (((qrow << 8) | qcol) << 4) | idm : -1;
    }
    const int qrow = qrc >> 12, qcol = (qrc >> 4) & 255;                    // (only for the lane constants below; the loops re-derive them from qrc)
    // Lane constants of the segment loops, ONE register each.  hipcc otherwise hoists every address it can form from lane / n / g out of those loops (dozens of
    // VGPRs, spilled, and each reload waits for ALL outstanding memory operations); QT_KEEP makes the loops see them as opaque values: adds stay inside.
    int o_p1s = (4 * g0) * QT_SP + qrow * QT_SR + qcol * 8;                  // P1 -> S: + j planes
    int o_p2s = wave * QT_SP + n * QT_SR + 16 * g0;                         // P2 <- S
    int o_p2d = QA_OFF_R + n * QT_RR + 4 * g0 * AG_PS + 2 * wave;            // P2 -> R
    int o_p3 = QA_OFF_R + wave * QT_RR + n * AG_PS + 8 * g0;                // P3 <- R: + 16 nt pixels
    int o_l16 = lane * 16, o_g16 = g0 * 16;                                  // A fragments; bias / BatchNorm quads
#define QT_KEEP() asm volatile("" : "+v"(o_p1s), "+v"(o_p2s), "+v"(o_p2d), "+v"(o_p3), "+v"(o_l16), "+v"(o_g16), "+v"(n), "+v"(qrc))

    const int t_begin = blockIdx.x * a.tiles_per_block;
    const int t_end = t_begin + a.tiles_per_block < a.n_tiles ? t_begin + a.tiles_per_block : a.n_tiles;
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        int r_ = tile;
        const int tx = r_ % a.tiles_x; r_ /= a.tiles_x;
        const int ty = r_ % a.tiles_y;
        const int b = r_ / a.tiles_y;
        const int y0 = ty * AG_TH, x0 = tx * AG_TW;
        QT_KEEP();

        // ---- L: x of the halo tile -> LayerNorm1 -> B fragments in registers
        Act<kC> n1[4];
        unsigned tfl = 0;                                                     // bits 0..3: which of this lane's 4 pixels lie in the image; bits 4, 5: which of its two P3 pixels do
        {
            const int gy = y0 - 3 + (qrc >> 12), gx0 = x0 - 4 + 4 * ((qrc >> 4) & 255), g = o_g16 >> 4;
            const bool row_ok = qrc >= 0 && gy >= 0 && gy < a.H;
            // per-image buffer descriptor, 32-bit byte offsets, out-of-image pixels at an offset past the image: the hardware returns zeros, no branches, no 64-bit address math
            const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x) + (size_t)b * a.H * a.W * kC, 0, a.H * a.W * kC * 2, 0x00020000);
            Act<kC> xin[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gx = gx0 + i;
                const bool ok = row_ok && gx >= 0 && gx < a.W;
                const int vo = ok ? (gy * a.W + gx) * (kC * 2) + 16 * g : (int)0x80000000;
                const qt_u32x4 f0 = __builtin_amdgcn_raw_buffer_load_b128(r_x, vo, 0, 0), f1 = __builtin_amdgcn_raw_buffer_load_b128(r_x, vo, 64, 0);
                const qt_u32x2 ft = __builtin_amdgcn_raw_buffer_load_b64(r_x, ok ? vo - 8 * g : vo, 128, 0);
                xin[i].f[0] = make_uint4(f0[0], f0[1], f0[2], f0[3]); xin[i].f[1] = make_uint4(f1[0], f1[1], f1[2], f1[3]); xin[i].t = make_uint2(ft[0], ft[1]);
                if (ok) tfl |= 1u << i;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ln80_one(xin[i], n1[i], s_gb, g, a.eps);
                __builtin_amdgcn_sched_barrier(0);                             // one pixel at a time: interleaved, the four LayerNorms' temporaries do not fit 128 VGPRs
            }
        }
        // this wave's 32 pixels of P3: tile row `wave`, column tile nt = pixels 16 nt + n; the lane's first output pixel (an offset in 16-channel records)
        // (byte offsets of its two pixel records inside a 16-channel plane; past the plane where the pixel is outside the image: such stores are dropped)
        const bool p3_row = y0 + wave < a.H;
        tfl |= (p3_row && x0 + n < a.W ? 16u : 0u) | (p3_row && x0 + 16 + n < a.W ? 32u : 0u);
        int p3_off = ((b * a.H + y0 + wave) * a.W + x0 + n) * kSEG * 2 + (o_g16 >> 1);

        // P1: segment `seg` of the qkv GEMM for this lane's 4 pixels -> S[sb] (channel-planar; zero outside the image)
        auto p1 = [&](int seg, int sb) {
            const char* w = s_wq + seg * QA_WSEG;
            const uint4 a0 = *reinterpret_cast<const uint4*>(w + o_l16), a1 = *reinterpret_cast<const uint4*>(w + 1024 + o_l16);
            const uint2 at = *reinterpret_cast<const uint2*>(w + 2048 + (o_l16 >> 1));
            const f32x4 bias = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(s_bq) + 64 * seg + o_g16);
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                mma32(a0, n1[i].f[0], acc); mma32(a1, n1[i].f[1], acc); mma16(at, n1[i].t, acc);
                v[i] = acc + bias;
            }
            if (qrc >= 0) {                                                   // halves of a bf16 pair that lie outside the image are zeroed: the convolutions' zero padding
                const uint32_t pm0 = ((tfl & 1u) ? 0x0000ffffu : 0u) | ((tfl & 2u) ? 0xffff0000u : 0u), pm1 = ((tfl & 4u) ? 0x0000ffffu : 0u) | ((tfl & 8u) ? 0xffff0000u : 0u);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<uint2*>(lds + sb * QT_S + o_p1s + j * QT_SP) = make_uint2(qa_pk(v[0][j], v[1][j]) & pm0, qa_pk(v[2][j], v[3][j]) & pm1);
            }
        };
        // P2 of segment `seg`: S[sb] -> R[sb]; the pass-through group is a plain transposition (4 pixels of this wave's channel per lane)
        uint4 T[7];
        auto p2 = [&](int seg, int sb) {
            const char* src = lds + sb * QT_S + o_p2s;
            char* dst = lds + sb * QT_R + o_p2d;
            const int g5 = seg % 5;
            if (g5 == 3) qt_dw<7>(src, dst, T);
            else if (g5 == 2) qt_dw<5>(src, dst, T);
            else if (g5 != 0) qt_dw<3>(src, dst, T);
            else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint2 c4 = *reinterpret_cast<const uint2*>(src - (o_g16 >> 1) + 3 * QT_SR + 8 + 32 * h);
                    char* d = dst + 16 * h * AG_PS;
                    *reinterpret_cast<uint16_t*>(d) = (uint16_t)(c4.x & 0xffffu);
                    *reinterpret_cast<uint16_t*>(d + AG_PS) = (uint16_t)(c4.x >> 16);
                    *reinterpret_cast<uint16_t*>(d + 2 * AG_PS) = (uint16_t)(c4.y & 0xffffu);
                    *reinterpret_cast<uint16_t*>(d + 3 * AG_PS) = (uint16_t)(c4.y >> 16);
                }
            }
        };
        // the Toeplitz fragments of segment `seg` (none for a pass-through segment)
        auto tload = [&](int seg) {
            const int which = seg / 5, g5 = seg - 5 * which;
            if (g5 == 4) qt_load(r_toep, QT_TAB_LOC + which * 3 * 16 * 1024, 3, o_l16 + wave * 1024, T);
            else if (g5 != 0) qt_load(r_toep, g5 == 3 ? QT_TAB_K7 : g5 == 2 ? QT_TAB_K5 : QT_TAB_K3, 2 * g5 + 1, o_l16 + wave * 1024, T);
        };
        // segment order: the local branch's three segments first (its accumulators are then dead), then q, k, v
        auto seg_at = [](int k) { return k < 3 ? 5 * k + 4 : (k - 3) + (k - 3) / 4; };

        // P3 of segment `seg` <- R[rb]: point-wise (identity for the pass-through group) -> BatchNorm -> Hardswish -> qkvp (+ k's per-channel maximum)
        auto p3 = [&](int seg, int rb) {
            const int which = seg / 5, g5 = seg - 5 * which;
            const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(a.qkvp + (size_t)(4 * which + g5) * a.plane, 0, (int)(a.plane * 2), 0x00020000);
            const f32x4 sc = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(s_bn) + 64 * g5 + o_g16);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(s_bn) + 256 + 64 * g5 + o_g16);
            const uint2 apw = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(s_pw) + (g5 > 0 ? g5 - 1 : 0) * 512 + (o_l16 >> 1));
            const bool want_max = which == 1 && a.kmax != nullptr;
            f32x4 vmax = f32x4{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const uint2 in = *reinterpret_cast<const uint2*>(lds + o_p3 + rb * QT_R + 16 * nt * AG_PS);
                f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
                if (g5 != 0) mma16(apw, in, d);
                else d = up_tail(in);
                f32x4 v = d * sc + sh;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = hswish(v[e]);
                if ((tfl >> (4 + nt)) & 1u) {
                    const uint2 pkd = qa_pack(v);
                    __builtin_amdgcn_raw_buffer_store_b64(qt_u32x2{pkd.x, pkd.y}, r_out, p3_off + 16 * nt * kSEG * 2, 0, 0);
                    if (want_max) {
                        const f32x4 rr = up_tail(pkd);                         // the stored (bf16) values are what rc_gma_kv sees
#pragma unroll
                        for (int e = 0; e < 4; ++e) vmax[e] = fmaxf(vmax[e], rr[e]);
                    }
                }
            }
            if (want_max) {                                                   // rows of 16 lanes share a channel quad: row maximum -> LDS
#pragma unroll
                for (int e = 0; e < 4; ++e) vmax[e] = row_max16(vmax[e]);
                if (n == 0) *reinterpret_cast<float4*>(reinterpret_cast<char*>(s_red) + (g5 * QA_WAVES + wave) * 64 + o_g16) = make_float4(vmax[0], vmax[1], vmax[2], vmax[3]);
            }
        };

        // Software pipeline, ONE barrier per segment: interval k = { P2(k): S[k & 1] -> R[k & 1] | P1(k + 1) -> S[(k + 1) & 1], its fragments' loads | P3(k - 1) <- R[(k - 1) & 1] }.
        // The three are independent inside an interval, and the waves of a SIMD (w, w + 4, w + 8, w + 12) take them in two different orders: while two of
        // them run the LDS / MFMA-bound P2 the other two run the VALU-bound P3 (in one order for all, the SIMD's units took turns: 6.6 k cycles an interval)
        const bool p3_first = (wave & 4) != 0;
        tload(seg_at(0));
        p1(seg_at(0), 0);
        __syncthreads();
        {   // ---- k = 0..3: the local branch, dw 3x3 of q4 / k4 / v4 -> 48 -> 16 (accumulated over the three) -> LayerNorm(16) -> Hardswish
            f32x4 dl[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                QT_KEEP(); asm volatile("" : "+v"(tfl), "+v"(p3_off));
                p2(seg_at(k), k & 1);
                p1(seg_at(k + 1), (k + 1) & 1);
                tload(seg_at(k + 1));
                if (k >= 1) {
                    const uint2 apw = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(s_pw) + (3 + k - 1) * 512 + (o_l16 >> 1));
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) mma16(apw, *reinterpret_cast<const uint2*>(lds + o_p3 + ((k - 1) & 1) * QT_R + 16 * nt * AG_PS), dl[nt]);
                }
                __syncthreads();
            }
            const f32x4 lg = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(s_bn) + 512 + o_g16), lb = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(s_bn) + 576 + o_g16);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x4 tt = dl[nt] + 0.f;
                const float s = sum_lane_groups((tt[0] + tt[1]) + (tt[2] + tt[3]));
                const float mean = s / 16.f;
                const f32x4 dd = tt - mean;
                const f32x4 d2 = dd * dd;
                const float var = sum_lane_groups((d2[0] + d2[1]) + (d2[2] + d2[3]));
                const float rstd = 1.f / sqrtf(var / 16.f + 1e-5f);
                f32x4 v = dd * rstd * lg + lb;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = hswish(v[e]);
                const uint2 pkd = qa_pack(v);
                __builtin_amdgcn_raw_buffer_store_b64(qt_u32x2{pkd.x, pkd.y}, r_loc, ((tfl >> (4 + nt)) & 1u) ? p3_off + 16 * nt * kSEG * 2 : (int)0x80000000, 0, 0);
            }
        }
#pragma unroll 1
        for (int k = 4; k < 16; ++k) {   // ---- the four groups of q, k, v
            QT_KEEP(); asm volatile("" : "+v"(tfl), "+v"(p3_off));
            if (p3_first) p3(seg_at(k - 1), (k - 1) & 1);
            if (k < 15) {
                p2(seg_at(k), k & 1);
                if (k + 1 < 15) {
                    p1(seg_at(k + 1), (k + 1) & 1);
                    tload(seg_at(k + 1));
                }
            }
            if (!p3_first) p3(seg_at(k - 1), (k - 1) & 1);
            if (k == 12 && a.kmax != nullptr && wave == 0) {                  // all four k segments' maxima are in LDS (the last one, segment 8 = P3 of interval 11, a barrier ago)
                const int ch = o_l16 >> 4;                                    // = lane (re-derived here: as `tid` the address was hoisted out of the tile loop and spilled)
                const float* p = s_red + (ch >> 4) * QA_WAVES * 16 + (ch & 15);
                float m = p[0];
#pragma unroll
                for (int w = 1; w < QA_WAVES; ++w) m = fmaxf(m, p[16 * w]);
                atomic_max_f32(a.kmax + (size_t)b * 64 + ch, m);
            }
            __syncthreads();
        }
    }
}
#undef QT_KEEP

}  // namespace gf
}  // namespace rc

extern "C" int rc_chain_pack_weights_natural(const float* w, int cin, int cout, void* dst) {
    using namespace rc;
    using namespace rc::gf;
    RC_REQUIRE(w && dst, "rc_chain_pack_weights_natural: null pointer");
    RC_REQUIRE(cin >= 16 && cin % 16 == 0 && cout >= 16 && cout % 16 == 0, "rc_chain_pack_weights_natural: cin and cout must be multiples of 16");
    const int mt = cout / 16, ks = cin / 32, tb = tile_bytes(cin);
    uint16_t* out = static_cast<uint16_t*>(dst);
    for (int m = 0; m < mt; ++m) {
        uint16_t* tile = out + (size_t)m * tb / 2;
        for (int s = 0; s < ks; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i)
                    tile[(s * 64 + lane) * 8 + i] = host_f32_to_bf16(w[(size_t)(16 * m + (lane & 15)) * cin + 32 * s + 8 * (lane >> 4) + i]);
        if (cin % 32)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i)
                    tile[ks * 512 + lane * 4 + i] = host_f32_to_bf16(w[(size_t)(16 * m + (lane & 15)) * cin + 32 * ks + 4 * (lane >> 4) + i]);
    }
    return RC_OK;
}

extern "C" size_t rc_gma_toeplitz_bytes(void) { return (size_t)rc::gf::QT_TAB_BYTES; }

// Banded Toeplitz A fragments of the aggregator's depth-wise kernels: for kernel row dy of channel c (tap-major fp32 taps [K*K][16]),
// T[ox][slot] = tap[dy][dx] with dx = slot - 4 - ox + K / 2 (slot s of a 32-slot window = input pixel s - 4 relative to the window's first
// output), zero outside 0 <= dx < K; lane (ox = lane & 15, q = lane >> 4) holds slots 8 q .. 8 q + 7.  Taps must be bf16-representable
// (they are: the module's parameters are bf16 on this path); they are rounded to nearest even otherwise.
extern "C" int rc_gma_toeplitz_pack(const float* dw3, const float* dw5, const float* dw7, const float* dwl, void* dst) {
    using namespace rc;
    using namespace rc::gf;
    RC_REQUIRE(dw3 && dw5 && dw7 && dwl && dst, "rc_gma_toeplitz_pack: null pointer");
    uint16_t* out = static_cast<uint16_t*>(dst);
    auto group = [&](const float* taps, int K, int off_bytes) {
        for (int dy = 0; dy < K; ++dy)
            for (int c = 0; c < 16; ++c)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 8; ++i) {
                        const int ox = lane & 15, slot = 8 * (lane >> 4) + i, dx = slot - 4 - ox + K / 2;
                        out[off_bytes / 2 + ((size_t)(dy * 16 + c) * 64 + lane) * 8 + i] =
                            (dx >= 0 && dx < K) ? host_f32_to_bf16(taps[(dy * K + dx) * 16 + c]) : (uint16_t)0;
                    }
    };
    group(dw3, 3, QT_TAB_K3); group(dw5, 5, QT_TAB_K5); group(dw7, 7, QT_TAB_K7);
    for (int which = 0; which < 3; ++which) group(dwl + which * 144, 3, QT_TAB_LOC + which * 3 * 16 * 1024);
    return RC_OK;
}

extern "C" int rc_gma_qkv_aggregate(const void* d_x, const void* d_wq_natural, const float* d_bq, const float* d_ln1_g, const float* d_ln1_b, float eps,
                                    void* d_qkvp, void* d_loc, int batch, int H, int W, const void* d_toeplitz, const float* d_pw, const float* d_pwl,
                                    const float* d_bn_scale, const float* d_bn_shift, const float* d_ln_g, const float* d_ln_b, float* d_kmax, void* stream) {
    using namespace rc;
    using namespace rc::gf;
    RC_REQUIRE(d_x && d_wq_natural && d_ln1_g && d_ln1_b && d_qkvp && d_loc && d_toeplitz && d_pw && d_pwl && d_bn_scale && d_bn_shift && d_ln_g && d_ln_b,
               "rc_gma_qkv_aggregate: null pointer");
    RC_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "rc_gma_qkv_aggregate: bad shape");
    QaArgs a;
    a.x = static_cast<const bf16_t*>(d_x); a.qkvp = static_cast<bf16_t*>(d_qkvp); a.loc = static_cast<bf16_t*>(d_loc); a.kmax = d_kmax;
    a.batch = batch; a.H = H; a.W = W; a.tiles_x = ceil_div(W, AG_TW); a.tiles_y = ceil_div(H, AG_TH);
    const long long n_tiles = (long long)a.tiles_x * a.tiles_y * batch;
    RC_REQUIRE(n_tiles < (1ll << 31), "rc_gma_qkv_aggregate: too many tiles");
    RC_REQUIRE((long long)batch * H * W * kSEG * 2 < (1ll << 31) && (long long)H * W * kC * 2 < (1ll << 31),
               "rc_gma_qkv_aggregate: a 16-channel plane of the batch and an 80-channel image must stay below 2 GiB (32-bit buffer offsets)");
    a.n_tiles = (int)n_tiles;
    a.plane = (size_t)batch * H * W * kSEG;
    a.wq = d_wq_natural; a.bq = d_bq; a.ln1_g = d_ln1_g; a.ln1_b = d_ln1_b; a.eps = eps;
    a.toep = static_cast<const char*>(d_toeplitz); a.pw = d_pw; a.pwl = d_pwl;
    a.bn_scale = d_bn_scale; a.bn_shift = d_bn_shift; a.ln_g = d_ln_g; a.ln_b = d_ln_b;
    int dev = 0;
    RC_HIP_CHECK(hipGetDevice(&dev));
    RC_REQUIRE(dev >= 0 && dev < 64, "rc_gma_qkv_aggregate: device index out of range");
    static int cus[64] = {};
    static bool attr[64] = {};
    if (!cus[dev]) RC_HIP_CHECK(hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev));
    if (!attr[dev]) {
        RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gma_qkv_agg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr[dev] = true;
    }
    if (d_kmax) RC_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_kmax), (int)0xff800000u /* -inf */, (size_t)batch * 64, as_stream(stream)));
    int blocks = a.n_tiles < cus[dev] ? a.n_tiles : cus[dev];                // one 8-wave block per CU, a contiguous run of tiles each
    a.tiles_per_block = (a.n_tiles + blocks - 1) / blocks;
    blocks = (a.n_tiles + a.tiles_per_block - 1) / a.tiles_per_block;
    hipLaunchKernelGGL(gma_qkv_agg_kernel, dim3((unsigned)blocks), dim3(QA_THREADS), QA_LDS, as_stream(stream), a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

// ---- depth-wise 3x3 (+ bias, + identity) of NHWC maps whose channel count is a multiple of 16: rc_dwconv2d's bf16 3x3 single-rep case
// (ConvPosEnc, upstream groupmix.py:33-53: x = t + dw3x3(t) + b).  The general rc_dwconv2d kernel (gma.hip) stages 40-channel pixel slots with an
// odd 16-byte stride and ran at 2.6 TB/s with 47 % of its LDS cycles in bank conflicts (profiles/r03_pmc_mfma_lds.md); this is the aggregator's
// core instead -- one (16 x 32 tile, 16-channel segment) per block, conflict-free 40-byte pixel slots, six blocks per CU -- with the same
// accumulation order (bias; taps in (dy, dx) order by fmaf; the identity after the centre row), so the two give the same bits. ----------------
namespace rc {
namespace gf {

struct Dw3Args {
    const bf16_t* x; bf16_t* y; int xs, x_c0, ys, y_c0;     // channel strides / first channels (elements)
    int batch, H, W, n_seg, tiles_x, tiles_y;
    const float* wT; int n_w;                               // tap-major [9][n_w]
    const float* bias; int add_identity;
};

constexpr int DW3_LDS = (AG_TH + 2) * ((AG_TW + 2) * AG_PS + AG_RPAD);

__global__ __launch_bounds__(AG_THREADS) void dw3x3_seg16_kernel(Dw3Args a) {
    __shared__ __attribute__((aligned(16))) char s_x[DW3_LDS];
    __shared__ __attribute__((aligned(16))) float s_w[9 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
    int blk = blockIdx.x;                                   // a tile's segments on one XCD, close in time (see gma_agg_kernel)
    const int lane8 = blk & 7; blk >>= 3;
    const int seg = blk % a.n_seg; blk = (blk / a.n_seg) * 8 + lane8;
    if (blk >= a.tiles_x * a.tiles_y * a.batch) return;
    const int tx = blk % a.tiles_x; blk /= a.tiles_x;
    const int ty = blk % a.tiles_y;
    const int b = blk / a.tiles_y;
    const int y0 = ty * AG_TH, x0 = tx * AG_TW;
    const int prow = 4 * wave + 2 * (n >> 3), pcol = 4 * (n & 7);      // this lane's 2 x 4 patch, channels 4 q .. 4 q + 3 of the segment
    for (int i = tid; i < 9 * 16; i += AG_THREADS) s_w[i] = a.wT[(i >> 4) * a.n_w + 16 * seg + (i & 15)];
    agg_stage<1>(s_x, a.x + (size_t)b * a.H * a.W * a.xs, a.xs, a.x_c0 + 16 * seg, y0, x0, a.H, a.W, tid);
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv = ld4(a.bias + 16 * seg + 4 * q);
    __syncthreads();
    f32x4 acc[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[o][c] = bv;
    constexpr int RS = (AG_TW + 2) * AG_PS + AG_RPAD;
    const char* base = s_x + prow * RS + pcol * AG_PS + q * 8;
#pragma unroll
    for (int iy = 0; iy < 4; ++iy) {
        f32x4 xin[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) xin[c] = up_tail(*reinterpret_cast<const uint2*>(base + iy * RS + c * AG_PS));
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int dy = iy - o;
            if (dy < 0 || dy >= 3) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const f32x4 w = ld4(s_w + (dy * 3 + dx) * 16 + 4 * q);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[o][c][e] = __builtin_fmaf(w[e], xin[c + dx][e], acc[o][c][e]);
            }
            if (a.add_identity && dy == 1) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[o][c] = acc[o][c] + xin[c + 1];
            }
        }
    }
    bf16_t* yb = a.y + (size_t)b * a.H * a.W * a.ys + a.y_c0 + 16 * seg + 4 * q;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (y0 + prow + o < a.H && x0 + pcol + c < a.W)
                *reinterpret_cast<uint2*>(yb + ((size_t)(y0 + prow + o) * a.W + x0 + pcol + c) * a.ys) = pack_tail(acc[o][c] + 0.f);
}

// called by rc_dwconv2d (gma.hip) for bf16, 3x3, one rep, no per-vector windows, 16 | n_ch
int launch_dw3x3_seg16(const void* x, int xs, int x_c0, void* y, int ys, int y_c0, int batch, int H, int W, int n_ch, const float* wT, int n_w,
                       const float* bias, int add_identity, hipStream_t stream) {
    Dw3Args a;
    a.x = static_cast<const bf16_t*>(x); a.y = static_cast<bf16_t*>(y); a.xs = xs; a.x_c0 = x_c0; a.ys = ys; a.y_c0 = y_c0;
    a.batch = batch; a.H = H; a.W = W; a.n_seg = n_ch / 16; a.tiles_x = ceil_div(W, AG_TW); a.tiles_y = ceil_div(H, AG_TH);
    a.wT = wT; a.n_w = n_w; a.bias = bias; a.add_identity = add_identity;
    const size_t blocks = (((size_t)a.tiles_x * a.tiles_y * batch + 7) / 8) * 8 * a.n_seg;
    RC_REQUIRE(blocks < (1ull << 31), "rc_dwconv2d: too many tiles");
    hipLaunchKernelGGL(dw3x3_seg16_kernel, dim3((unsigned)blocks), dim3(AG_THREADS), 0, stream, a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

}  // namespace gf
}  // namespace rc


// =====================================================================================================================================
// The block's entry in ONE launch (rc_gma_in_cpe): x = a + dw3x3(a) + b_cpe with a = Conv1x1(d1 (192 -> 80)) + b_in -- the cfg3 net's gma_in followed by ConvPosEnc
// (realcamnet_amd/LiteISP.py `gma_in`; upstream groupmix.py:203-217).  Written in round 6's second session, measured 0.91 -> 0.79 ms and NOT shipped because its sums
// differed from run to run; the third session found why (the comment at the MFMA loop below; tools/ubench/gi_experiment.hip, tools/dbg/gi_stress.py) and ships it.
namespace rc {
namespace gf {

constexpr int GI_WAVES = 16, GI_THREADS = 64 * GI_WAVES, GI_TH = 8, GI_TW = 32, GI_CIN = 192;
constexpr int GI_ROWS = GI_TH + 2, GI_SLOTS = 40, GI_SR = GI_SLOTS * 2, GI_SP = GI_ROWS * GI_SR;   // slot s of a row = pixel x0 - 4 + s; 800-byte channel planes
constexpr int GI_S = kC * GI_SP + 64;
constexpr int GI_PS = 176, GI_RR = GI_TW * GI_PS + 16, GI_R = GI_TH * GI_RR;                    // pixel-major tile: 176-byte pixel slots (160 of data)
constexpr int GI_PAIRS = GI_ROWS * (GI_SLOTS / 2), GI_PPW = (GI_PAIRS + GI_WAVES - 1) / GI_WAVES;   // 200 pixel pairs, 13 per wave
constexpr int GI_OFF_R = GI_S, GI_OFF_W = GI_OFF_R + GI_R, GI_OFF_B = GI_OFF_W + 5 * (GI_CIN / 32) * 1024, GI_LDS = GI_OFF_B + 2 * kC * 4;
static_assert(GI_S % 16 == 0 && GI_R % 16 == 0 && GI_PPW <= 16 && GI_LDS <= 160 * 1024, "gma_in + cpe LDS layout");

struct GiArgs {
    const bf16_t* d1; bf16_t* x;
    int batch, H, W, tiles_x, tiles_y, n_tiles, tiles_per_block;
    const void* w_in; const float* b_in;             // rc_chain_pack_weights_natural(192 -> 80); bias [80] or NULL
    const char* toep; const float* b_cpe;            // rc_dw_toeplitz_pack(taps (9, 80), K = 3, 80 channels); bias [80] or NULL
};

__global__ __launch_bounds__(GI_THREADS) void gma_in_cpe_kernel(GiArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* s_w = lds + GI_OFF_W;
    float* s_b = reinterpret_cast<float*>(lds + GI_OFF_B);                  // b_in [80] | b_cpe [80]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    for (int i = tid; i < 5 * (GI_CIN / 32) * 64; i += GI_THREADS) reinterpret_cast<uint4*>(s_w)[i] = reinterpret_cast<const uint4*>(a.w_in)[i];
    for (int i = tid; i < kC; i += GI_THREADS) { s_b[i] = a.b_in ? a.b_in[i] : 0.f; s_b[kC + i] = a.b_cpe ? a.b_cpe[i] : 0.f; }
    for (int i = tid; i < (GI_S + GI_R) / 16; i += GI_THREADS) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    const __amdgpu_buffer_rsrc_t r_toep = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.toep), 0, 3 * kC * 1024, 0x00020000);
    // this lane's pixel pair of the halo tile (the same for all four g: they hold different K-slices / output channels of the same pixels)
    const int pid = wave * GI_PPW + n;
    const bool has_pair = n < GI_PPW && pid < GI_PAIRS;
    const int prow = pid / (GI_SLOTS / 2), pcol = pid - prow * (GI_SLOTS / 2);
    int o_s = (4 * g) * GI_SP + prow * GI_SR + pcol * 4;                                        // G -> S: + (16 m + j) planes
    int o_ds = n * 0 + (n & 7) * GI_SR + (n >> 3) * 32 + 16 * g;                                // D <- S: + channel plane + kernel row
    int o_di = ((n & 7) + 1) * GI_SR + 8 + (n >> 3) * 32 + 8 * g;                               // D <- S, the identity: the lane's own 4 pixels
    int o_dr = GI_OFF_R + (n & 7) * GI_RR + ((n >> 3) * 16 + 4 * g) * GI_PS;                    // D -> R: + 2 c
    int o_l16 = lane * 16, o_g16 = g * 16;
#define GI_KEEP() asm volatile("" : "+v"(o_s), "+v"(o_ds), "+v"(o_di), "+v"(o_dr), "+v"(o_l16), "+v"(o_g16))

    const int t_begin = blockIdx.x * a.tiles_per_block;
    const int t_end = t_begin + a.tiles_per_block < a.n_tiles ? t_begin + a.tiles_per_block : a.n_tiles;
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        int r_ = tile;
        const int tx = r_ % a.tiles_x; r_ /= a.tiles_x;
        const int ty = r_ % a.tiles_y;
        const int b = r_ / a.tiles_y;
        const int y0 = ty * GI_TH, x0 = tx * GI_TW;
        GI_KEEP();
        {   // ---- G: a = W_in . d1 + b_in on the halo tile -> S
            const __amdgpu_buffer_rsrc_t r_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.d1) + (size_t)b * a.H * a.W * GI_CIN, 0, a.H * a.W * GI_CIN * 2, 0x00020000);
            const int gy = y0 - 1 + prow, gx = x0 - 4 + 2 * pcol;
            const bool row_ok = has_pair && gy >= 0 && gy < a.H;
            const bool ok0 = row_ok && gx >= 0 && gx < a.W, ok1 = row_ok && gx + 1 >= 0 && gx + 1 < a.W;
            const int vo0 = ok0 ? (gy * a.W + gx) * (GI_CIN * 2) + o_g16 : (int)0x80000000, vo1 = ok1 ? (gy * a.W + gx + 1) * (GI_CIN * 2) + o_g16 : (int)0x80000000;
            qt_u32x4 bx[2][GI_CIN / 32];
#pragma unroll
            for (int s = 0; s < GI_CIN / 32; ++s) {
                bx[0][s] = __builtin_amdgcn_raw_buffer_load_b128(r_d, vo0, 64 * s, 0);
                bx[1][s] = __builtin_amdgcn_raw_buffer_load_b128(r_d, vo1, 64 * s, 0);
            }
            // One 16-channel row tile at a time, its six A fragments in registers, the NEXT tile's six loaded meanwhile into the other set.
            const uint32_t pm = (ok0 ? 0x0000ffffu : 0u) | (ok1 ? 0xffff0000u : 0u);     // outside the image a is ZERO (the depth-wise conv's padding), not b_in
            uint4 af[2][GI_CIN / 32];
#pragma unroll
            for (int s = 0; s < GI_CIN / 32; ++s) af[0][s] = *reinterpret_cast<const uint4*>(s_w + s * 1024 + o_l16);
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                if (m + 1 < 5) {
#pragma unroll
                    for (int s = 0; s < GI_CIN / 32; ++s) af[(m + 1) & 1][s] = *reinterpret_cast<const uint4*>(s_w + ((m + 1) * (GI_CIN / 32) + s) * 1024 + o_l16);
                }
                f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                // EACH accumulator's six K-steps back to back (hardware-interlocked), nothing scheduled between: with the two accumulators ALTERNATING -- each revisited
                // one MFMA later, whatever was or was not scheduled in between (tools/ubench/gi_experiment.hip orders 0 / 2 / 3) -- ~1 of 10^6 outputs came out
                // with a stale partial sum, differently from run to run (40 of 40 launches differed; this order: 0 of 80, and 8 % faster).
                for (int s = 0; s < GI_CIN / 32; ++s) { __builtin_amdgcn_sched_barrier(0); mma32(af[m & 1][s], make_uint4(bx[0][s][0], bx[0][s][1], bx[0][s][2], bx[0][s][3]), acc0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int s = 0; s < GI_CIN / 32; ++s) { mma32(af[m & 1][s], make_uint4(bx[1][s][0], bx[1][s][1], bx[1][s][2], bx[1][s][3]), acc1); __builtin_amdgcn_sched_barrier(0); }
                const f32x4 bias = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(s_b) + 64 * m + o_g16);
                const f32x4 v0 = acc0 + bias, v1 = acc1 + bias;
                if (has_pair) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<uint32_t*>(lds + o_s + (16 * m + j) * GI_SP) = qa_pk(v0[j], v1[j]) & pm;
                }
            }
        }
        __syncthreads();
        GI_KEEP();
#pragma unroll 1
        for (int i = 0; i < kC / GI_WAVES; ++i) {   // ---- D: x = a + dw3x3(a) + b_cpe for channels wave, wave + 16, ..
            const int c = wave + GI_WAVES * i;
            uint4 T[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const qt_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r_toep, o_l16, (dy * kC + c) * 1024, 0);
                T[dy] = make_uint4(t[0], t[1], t[2], t[3]);
            }
            const char* src = lds + c * GI_SP + o_ds;
            f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) mma32(T[dy], *reinterpret_cast<const uint4*>(src + dy * GI_SR), d);
            const f32x4 idn = up_tail(*reinterpret_cast<const uint2*>(lds + c * GI_SP + o_di));
            const float bc = s_b[kC + c];
            const f32x4 v = (d + bc) + idn;
            const uint32_t p01 = qa_pk(v[0], v[1]), p23 = qa_pk(v[2], v[3]);
            char* dst = lds + o_dr + 2 * c;
            *reinterpret_cast<uint16_t*>(dst) = (uint16_t)(p01 & 0xffffu);
            *reinterpret_cast<uint16_t*>(dst + GI_PS) = (uint16_t)(p01 >> 16);
            *reinterpret_cast<uint16_t*>(dst + 2 * GI_PS) = (uint16_t)(p23 & 0xffffu);
            *reinterpret_cast<uint16_t*>(dst + 3 * GI_PS) = (uint16_t)(p23 >> 16);
        }
        __syncthreads();
        {   // ---- St: 256 pixels x 160 bytes, 16 bytes per lane in address order
            const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(a.x + (size_t)b * a.H * a.W * kC, 0, a.H * a.W * kC * 2, 0x00020000);
#pragma unroll
            for (int k = 0; k < (GI_TH * GI_TW * 10 + GI_THREADS - 1) / GI_THREADS; ++k) {
                const int chunk = tid + k * GI_THREADS;
                const int px = chunk / 10, part = chunk - 10 * px, row = px >> 5, col = px & 31;
                const bool ok = chunk < GI_TH * GI_TW * 10 && y0 + row < a.H && x0 + col < a.W;
                const uint4 v = *reinterpret_cast<const uint4*>(lds + GI_OFF_R + (chunk < GI_TH * GI_TW * 10 ? row * GI_RR + col * GI_PS + part * 16 : 0));
                __builtin_amdgcn_raw_buffer_store_b128(qt_u32x4{v.x, v.y, v.z, v.w}, r_x, ok ? ((y0 + row) * a.W + x0 + col) * (kC * 2) + part * 16 : (int)0x80000000, 0, 0);
            }
        }
    }
}
#undef GI_KEEP

}  // namespace gf
}  // namespace rc

// Banded Toeplitz A fragments of a depth-wise K x K kernel over n_ch channels (tap-major fp32 taps [K*K][n_ch]): dst[dy][channel][lane] 16 bytes,
// T[ox][slot] = tap[dy][dx], dx = slot - 4 - ox + K / 2 (see rc_gma_toeplitz_pack).  K * n_ch KiB.
extern "C" int rc_dw_toeplitz_pack(const float* taps, int K, int n_ch, void* dst) {
    using namespace rc;
    RC_REQUIRE(taps && dst, "rc_dw_toeplitz_pack: null pointer");
    RC_REQUIRE((K == 3 || K == 5 || K == 7) && n_ch >= 1, "rc_dw_toeplitz_pack: K = 3, 5 or 7");
    uint16_t* out = static_cast<uint16_t*>(dst);
    for (int dy = 0; dy < K; ++dy)
        for (int c = 0; c < n_ch; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    const int ox = lane & 15, slot = 8 * (lane >> 4) + i, dx = slot - 4 - ox + K / 2;
                    out[((size_t)(dy * n_ch + c) * 64 + lane) * 8 + i] = (dx >= 0 && dx < K) ? host_f32_to_bf16(taps[(size_t)(dy * K + dx) * n_ch + c]) : (uint16_t)0;
                }
    return RC_OK;
}

extern "C" int rc_gma_in_cpe(const void* d_d1, const void* d_w_in_natural, const float* d_b_in, const void* d_toeplitz3, const float* d_b_cpe, void* d_x,
                             int batch, int H, int W, void* stream) {
    using namespace rc;
    using namespace rc::gf;
    RC_REQUIRE(d_d1 && d_w_in_natural && d_toeplitz3 && d_x, "rc_gma_in_cpe: null pointer");
    RC_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "rc_gma_in_cpe: bad shape");
    RC_REQUIRE((long long)H * W * GI_CIN * 2 < (1ll << 31), "rc_gma_in_cpe: a 192-channel image must stay below 2 GiB (32-bit buffer offsets)");
    GiArgs a;
    a.d1 = static_cast<const bf16_t*>(d_d1); a.x = static_cast<bf16_t*>(d_x);
    a.batch = batch; a.H = H; a.W = W; a.tiles_x = ceil_div(W, GI_TW); a.tiles_y = ceil_div(H, GI_TH);
    const long long n_tiles = (long long)a.tiles_x * a.tiles_y * batch;
    RC_REQUIRE(n_tiles < (1ll << 31), "rc_gma_in_cpe: too many tiles");
    a.n_tiles = (int)n_tiles;
    a.w_in = d_w_in_natural; a.b_in = d_b_in; a.toep = static_cast<const char*>(d_toeplitz3); a.b_cpe = d_b_cpe;
    int dev = 0;
    RC_HIP_CHECK(hipGetDevice(&dev));
    RC_REQUIRE(dev >= 0 && dev < 64, "rc_gma_in_cpe: device index out of range");
    static int cus[64] = {};
    static bool attr[64] = {};
    if (!cus[dev]) RC_HIP_CHECK(hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev));
    if (!attr[dev]) {
        RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gma_in_cpe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr[dev] = true;
    }
    int blocks = a.n_tiles < cus[dev] ? a.n_tiles : cus[dev];
    a.tiles_per_block = (a.n_tiles + blocks - 1) / blocks;
    blocks = (a.n_tiles + a.tiles_per_block - 1) / a.tiles_per_block;
    hipLaunchKernelGGL(gma_in_cpe_kernel, dim3((unsigned)blocks), dim3(GI_THREADS), GI_LDS, as_stream(stream), a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

// =====================================================================================================================================
// Lens_Shading_Correction as a register-resident chain, optionally with the convolution it modulates
//   coord (B,H,W,cin0 <= 4) -> Conv1x1(cin0, C) -> LeakyReLU -> [Conv1x1(C, C) -> LeakyReLU] x (n_mid - 1) -> Conv1x1(C, C) = lsc
//   HEAD:  out = (conv3x3(raw (B,H,W,raw_c <= 4)) + bias) * (lsc + 1)          (models/LiteISP.py:363-378, 2012-2014; ISPUNet :1352-1355)
// The LDS-slab form of this chain (csrc/chain.hip) is VALU-bound on its per-layer LDS round trips (0.69 ms at 4K x 8, 1.26 ms with the
// head folded in).  Here a wave's 64 pixels stay in MFMA fragments through every layer (pair-packed rows, see the top of this file):
// layer 0 is one half-filled K-step (k = 4 q + i < cin0), the head's 9 taps x 4 channels are one full K-step (lane group g: taps 2g, 2g+1)
// plus a half-filled one (tap 8), and the lsc map is rounded to bf16 exactly where the two-launch path stores it.
namespace rc {
namespace gf {

struct LscArgs {
    const bf16_t* x; int cin0;
    const char* blob; int n_mid; float slope;      // rc_lsc_pack: [layer 0 | mid layers | head] bf16 fragments, then the packed fp32 biases
    const bf16_t* raw; int raw_c;                  // HEAD only
    bf16_t* out; long long pixels; int H, W;
};

template <int C> __host__ __device__ constexpr int lsc_weight_bytes(int n_mid, bool head) {
    return (C / 16) * 512 + n_mid * (C / 16) * tile_bytes(C) + (head ? (C / 16) * 1536 : 0);
}

// acc[m][nt] = W[tile m] . in with a literal-zero C operand on the first K-step (no accumulator initialisation instructions) -- or, with `bias` (the fp32 bias rows
// of tiles 0 .. MT-1 for lane group g), bias + W . in: the bias as the MFMA chain's initial C operand, as rc_conv2d has it, instead of a packed add per pair of values
// afterwards (the lens-shading chain is VALU-bound: 2.5 issues per value and layer were its 1.1 ms)
template <int CIN, int MT>
__device__ __forceinline__ void gemm_fresh(const char* w, int lane, const Act<CIN> (&in)[kNT], f32x4 (&acc)[MT][kNT], const float* bias = nullptr, int g = 0) {
    constexpr int KS = CIN / 32, TB = tile_bytes(CIN);
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const f32x4 b = bias4(bias, m, g);
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) acc[m][nt] = b;
        }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        uint4 a[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const uint4*>(w + m * TB + s * 1024 + lane * 16);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                if (s == 0 && bias == nullptr) { acc[m][nt] = z; }
                mma32(a[m], in[nt].f[s], acc[m][nt]);
            }
    }
    if constexpr ((CIN % 32) != 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const uint2 a = *reinterpret_cast<const uint2*>(w + m * TB + KS * 1024 + lane * 8);
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                if (KS == 0 && bias == nullptr) { acc[m][nt] = z; }
                mma16(a, in[nt].t, acc[m][nt]);
            }
        }
    }
}

// LeakyReLU with 0 <= slope <= 1 (host-checked) as med3(v, slope v, FLT_MAX) = max(v, slope v): ONE VALU op per value behind the packed multiply.  fmaxf() is two:
// without fast-math hipcc quiets each operand first (v_max_f32 v, v, v), and an MFMA accumulator is not known to be quiet -- 96 v_max per 48 values and layer --
// and med3 against +inf is folded back into exactly that maxnum.  (v = +inf stays +inf: the median of {inf, inf, FLT_MAX}.)
__device__ __forceinline__ f32x4 leaky4(const f32x4& v, float slope) {
    const f32x4 s = v * slope;
    const float inf = 3.402823466e+38f;
    return f32x4{__builtin_amdgcn_fmed3f(v[0], s[0], inf), __builtin_amdgcn_fmed3f(v[1], s[1], inf), __builtin_amdgcn_fmed3f(v[2], s[2], inf), __builtin_amdgcn_fmed3f(v[3], s[3], inf)};
}

// fp32 accumulator tiles (+ bias [, LeakyReLU]) -> the next layer's B fragments
template <int C, bool ACT, bool BIASED = false>
__device__ __forceinline__ void lsc_pack(const f32x4 (&acc)[C / 16][kNT], const float* bias, int g, float slope, Act<C> (&out)[kNT]) {
    constexpr int MT = C / 16;
    auto fin = [&](const f32x4& v, const f32x4& b) {
        f32x4 r = BIASED ? v : v + b;                  // BIASED: the accumulators started from the bias (gemm_fresh)
        if constexpr (ACT) r = leaky4(r, slope);
        return r;
    };
    // BIASED && !ACT: the value packed is the bare MFMA accumulator -- never through the inline-asm conversion (HAZARD RULE at pk()): the compiler-visible one
    constexpr bool RAW_ACC = BIASED && !ACT;
#pragma unroll
    for (int p = 0; p < MT / 2; ++p) {
        const f32x4 b0 = bias4(bias, 2 * p, g), b1 = bias4(bias, 2 * p + 1, g);
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            if constexpr (RAW_ACC) {
                const uint2 lo = qa_pack(acc[2 * p][nt]), hi = qa_pack(acc[2 * p + 1][nt]);
                out[nt].f[p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            } else {
                out[nt].f[p] = pack_pair(fin(acc[2 * p][nt], b0), fin(acc[2 * p + 1][nt], b1));
            }
        }
    }
    if constexpr (MT & 1) {
        const f32x4 b0 = bias4(bias, MT - 1, g);
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            if constexpr (RAW_ACC) out[nt].t = qa_pack(acc[MT - 1][nt]);
            else out[nt].t = pack_tail(fin(acc[MT - 1][nt], b0));
        }
    }
}

constexpr int kOOBoff = (int)0x80000000;            // buffer offset past any image: the load returns 0

struct LscIn {                                       // one wave tile's inputs in flight
    uint2 x[kNT];                                    // layer-0 B operand (k = 4 g + i: the coordinates in lane group 0)
    uint4 rf[kNT]; uint2 rt[kNT];                    // head B operands: taps (2g, 2g+1) x 4 channels; tap 8 in lane group 0
};

template <bool HEAD>
__device__ __forceinline__ void lsc_load(const LscArgs& a, long long p0, int n, int g, __amdgpu_buffer_rsrc_t r_x, __amdgpu_buffer_rsrc_t r_raw, LscIn& in) {
    const unsigned uW = (unsigned)a.W, uH = (unsigned)a.H;
    unsigned row0 = 0, x0 = 0, y0 = 0;
    if constexpr (HEAD) {
        row0 = __builtin_amdgcn_readfirstlane((unsigned)p0 / uW);
        x0 = (unsigned)p0 - row0 * uW;
        y0 = __builtin_amdgcn_readfirstlane(row0 % uH);
    }
#pragma unroll
    for (int nt = 0; nt < kNT; ++nt) {
        const long long p = p0 + 16 * nt + n;
        const bool live = p < a.pixels;
        {   // coordinates: cin0 bf16 values of lane group 0
            const int off = (live && g == 0) ? (int)p * a.cin0 * 2 : kOOBoff;
            unsigned lo = 0, hi = 0;
            if (a.cin0 == 2) lo = __builtin_amdgcn_raw_buffer_load_b32(r_x, off, 0, 0);
            else if (a.cin0 == 4) { const auto v = __builtin_amdgcn_raw_buffer_load_b64(r_x, off, 0, 0); lo = v[0]; hi = v[1]; }
            else {
                unsigned short e[4] = {0, 0, 0, 0};
                for (int i = 0; i < a.cin0; ++i) e[i] = __builtin_amdgcn_raw_buffer_load_b16(r_x, off == kOOBoff ? kOOBoff : off + 2 * i, 0, 0);
                lo = e[0] | ((unsigned)e[1] << 16); hi = e[2] | ((unsigned)e[3] << 16);
            }
            in.x[nt] = make_uint2(lo, hi);
        }
        if constexpr (HEAD) {
            unsigned x = x0 + 16 * nt + n, y = y0;
            while (x >= uW) { x -= uW; ++y; }
            while (y >= uH) y -= uH;
            auto tap = [&](int t) -> uint2 {                                   // 8 bytes of channels of tap t, zero outside the image
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                const bool ok = live && (unsigned)((int)y + dy) < uH && (unsigned)((int)x + dx) < uW;
                const int off = ok ? ((int)p + dy * a.W + dx) * a.raw_c * 2 : kOOBoff;
                if (a.raw_c == 4) { const auto v = __builtin_amdgcn_raw_buffer_load_b64(r_raw, off, 0, 0); return make_uint2(v[0], v[1]); }
                unsigned short e[4] = {0, 0, 0, 0};
                for (int i = 0; i < a.raw_c; ++i) e[i] = __builtin_amdgcn_raw_buffer_load_b16(r_raw, ok ? off + 2 * i : kOOBoff, 0, 0);
                return make_uint2(e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16));
            };
            const uint2 t0 = tap(2 * g), t1 = tap(2 * g + 1), t8 = tap(8);
            in.rf[nt] = make_uint4(t0.x, t0.y, t1.x, t1.y);
            in.rt[nt] = g == 0 ? t8 : make_uint2(0u, 0u);
        }
    }
}

// The same loads for the shapes the nets have (2 coordinates; a 4-channel packed RAW; rows of at least 64 pixels), written for instruction count: the generic
// form above re-derives (row, column) of every pixel with loops and decides every tap's bounds from scratch -- ~450 VALU, ~400 SALU and ~250 branches per
// 64-pixel tile, half of what the kernel issued (1.08 ms at cfg3 against 0.27 ms for its bytes).  Here the tile's first pixel (tx0, ty0) is WAVE-UNIFORM state
// walked by the caller, a lane's (x, y) is one conditional wrap away from it (W >= 64), the two taps of lane group g and their offsets are lane constants
// (LscTaps), and a tap is {2 adds, 2 unsigned compares, 1 select} + its load.  Same addresses, same zero padding.
struct LscTaps { int dy0, dx0, dy1, dx1, d0, d1, d8; };          // taps 2 g and 2 g + 1 of this lane group; byte offsets relative to the pixel's own record
__device__ __forceinline__ LscTaps lsc_taps(int g, int W) {
    LscTaps t;
    t.dy0 = (2 * g) / 3 - 1; t.dx0 = (2 * g) % 3 - 1; t.dy1 = (2 * g + 1) / 3 - 1; t.dx1 = (2 * g + 1) % 3 - 1;
    t.d0 = (t.dy0 * W + t.dx0) * 8; t.d1 = (t.dy1 * W + t.dx1) * 8; t.d8 = (W + 1) * 8;
    return t;
}
template <bool HEAD>
__device__ __forceinline__ void lsc_load_fast(const LscArgs& a, int p0, unsigned tx0, unsigned ty0, int n, int g, const LscTaps& tp, __amdgpu_buffer_rsrc_t r_x,
                                              __amdgpu_buffer_rsrc_t r_raw, LscIn& in) {
    const unsigned uW = (unsigned)a.W, uH = (unsigned)a.H;
    const int npix = (int)a.pixels;
#pragma unroll
    for (int nt = 0; nt < kNT; ++nt) {
        const int p = p0 + 16 * nt + n;
        const bool live = p < npix;
        in.x[nt] = make_uint2(__builtin_amdgcn_raw_buffer_load_b32(r_x, (live && g == 0) ? p * 4 : kOOBoff, 0, 0), 0u);
        if constexpr (HEAD) {
            unsigned x = tx0 + 16 * nt + n, y = ty0;
            const bool wrap = x >= uW;                                           // W >= 64: at most one row further
            x = wrap ? x - uW : x; y = wrap ? y + 1 : y; y = y == uH ? 0u : y;
            const int pb = p * 8;
            const bool ok0 = live && (unsigned)((int)y + tp.dy0) < uH && (unsigned)((int)x + tp.dx0) < uW;
            const bool ok1 = live && (unsigned)((int)y + tp.dy1) < uH && (unsigned)((int)x + tp.dx1) < uW;
            const bool ok8 = live && g == 0 && y + 1 < uH && x + 1 < uW;
            const auto t0 = __builtin_amdgcn_raw_buffer_load_b64(r_raw, ok0 ? pb + tp.d0 : kOOBoff, 0, 0);
            const auto t1 = __builtin_amdgcn_raw_buffer_load_b64(r_raw, ok1 ? pb + tp.d1 : kOOBoff, 0, 0);
            const auto t8 = __builtin_amdgcn_raw_buffer_load_b64(r_raw, ok8 ? pb + tp.d8 : kOOBoff, 0, 0);
            in.rf[nt] = make_uint4(t0[0], t0[1], t1[0], t1[1]);
            in.rt[nt] = make_uint2(t8[0], t8[1]);
        }
    }
}

template <int C> constexpr int lsc_threads() { return C > 64 ? 512 : 256; }   // wide chains keep ~100 KB of weights in LDS: one 8-wave block per CU
template <int C, bool HEAD>
__global__ __launch_bounds__(lsc_threads<C>(), 2) void lsc_chain_kernel(const LscArgs a) {    // <= 256 registers: VGPR-form MFMA, no AGPR copies
    constexpr int kLscThreads = lsc_threads<C>();
    constexpr int MT = C / 16, TBM = tile_bytes(C);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int w_bytes = lsc_weight_bytes<C>(a.n_mid, HEAD);
    const int total = w_bytes + (1 + a.n_mid + (HEAD ? 1 : 0)) * MT * 16 * 4;
    for (int i = tid; i < total / 16; i += kLscThreads) reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(a.blob)[i];
    __syncthreads();
    constexpr bool STAGE = C <= 64;                      // narrow chains: per-wave output staging behind the weights (the wide form's LDS is full)
    [[maybe_unused]] char* s_stage = lds + ((total + 15) & ~15);
    const char* s_l0 = lds;
    const char* s_mid = lds + MT * 512;
    const char* s_head = s_mid + a.n_mid * MT * TBM;
    const float* s_b = reinterpret_cast<const float*>(lds + w_bytes);

    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)(a.pixels * a.cin0 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_raw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(HEAD ? a.raw : a.x), 0,
                                                                           (int)(a.pixels * (HEAD ? a.raw_c : a.cin0) * 2), 0x00020000);
    const long long n_tiles = (a.pixels + 63) / 64, n_waves = (long long)gridDim.x * (kLscThreads / 64);
    long long tile = (long long)blockIdx.x * (kLscThreads / 64) + __builtin_amdgcn_readfirstlane(tid >> 6);
    // fast loads (lsc_load_fast): 2 coordinates, a 4-channel RAW, rows of >= 64 pixels, everything inside 32-bit pixel indexes (the host checks pixels * 8 < 2^31)
    const bool fast = a.cin0 == 2 && (!HEAD || a.raw_c == 4) && a.W >= 64;                                  // uniform
    const LscTaps tp = lsc_taps(g, a.W);
    const unsigned uW = (unsigned)a.W, uH = (unsigned)a.H;
    const unsigned step = (unsigned)(64 * n_waves), sx = step % uW, sy = (step / uW) % uH;                  // the walk of (tx0, ty0): uniform, no division in the loop
    unsigned tx0 = 0, ty0 = 0;                                                                               // first pixel of the tile whose loads are issued NEXT
    if (tile < n_tiles) { const unsigned p0 = (unsigned)(tile * 64), row0 = p0 / uW; tx0 = p0 - row0 * uW; ty0 = row0 % uH; }
    auto load = [&](long long t, LscIn& dst) {
        if (fast) {
            lsc_load_fast<HEAD>(a, (int)(t * 64), tx0, ty0, n, g, tp, r_x, r_raw, dst);
            tx0 += sx; ty0 += sy;
            if (tx0 >= uW) { tx0 -= uW; ++ty0; }
            if (ty0 >= uH) ty0 -= uH;
        } else {
            lsc_load<HEAD>(a, t * 64, n, g, r_x, r_raw, dst);
        }
    };
    LscIn nxt;
    if (tile < n_tiles) load(tile, nxt);
    for (; tile < n_tiles; tile += n_waves) {
        const LscIn in = nxt;
        if (tile + n_waves < n_tiles) load(tile + n_waves, nxt);     // next tile's operands in flight
        Act<C> cur[kNT];
        {   // layer 0, one output tile pair at a time (the accumulators of all MT tiles are never live together)
            auto leaky = [&](f32x4 v) { return leaky4(v, a.slope); };
#pragma unroll
            for (int p = 0; p < MT / 2; ++p) {
                const uint2 w0 = *reinterpret_cast<const uint2*>(s_l0 + (2 * p) * 512 + lane * 8), w1 = *reinterpret_cast<const uint2*>(s_l0 + (2 * p + 1) * 512 + lane * 8);
                const f32x4 b0 = bias4(s_b, 2 * p, g), b1 = bias4(s_b, 2 * p + 1, g);
#pragma unroll
                for (int nt = 0; nt < kNT; ++nt) {
                    f32x4 a0 = b0, a1 = b1;                         // bias = the initial C operand
                    mma16(w0, in.x[nt], a0); mma16(w1, in.x[nt], a1);
                    cur[nt].f[p] = pack_pair(leaky(a0), leaky(a1));
                }
            }
            if constexpr (MT & 1) {
                const uint2 w0 = *reinterpret_cast<const uint2*>(s_l0 + (MT - 1) * 512 + lane * 8);
                const f32x4 b0 = bias4(s_b, MT - 1, g);
#pragma unroll
                for (int nt = 0; nt < kNT; ++nt) {
                    f32x4 a0 = b0;
                    mma16(w0, in.x[nt], a0);
                    cur[nt].t = pack_tail(leaky(a0));
                }
            }
        }
#pragma unroll 1
        for (int l = 0; l < a.n_mid; ++l) {
            if constexpr (MT <= 4) {
                f32x4 acc[MT][kNT];
                gemm_fresh<C, MT>(s_mid + l * MT * TBM, lane, cur, acc, s_b + (1 + l) * MT * 16, g);
                if (l + 1 < a.n_mid) lsc_pack<C, true, true>(acc, s_b + (1 + l) * MT * 16, g, a.slope, cur);
                else lsc_pack<C, false, true>(acc, s_b + (1 + l) * MT * 16, g, a.slope, cur);
            } else {      // wide chains: 4 output tiles at a time (all MT accumulators + both activation sets do not fit 256 registers)
                static_assert(MT % 4 == 0, "wide chain: whole groups of 4 output tiles");
                Act<C> nxt[kNT];
#pragma unroll
                for (int h = 0; h < MT / 4; ++h) {
                    f32x4 acc[4][kNT];
                    const float* bias = s_b + (1 + l) * MT * 16 + 4 * h * 16;
                    gemm_fresh<C, 4>(s_mid + (l * MT + 4 * h) * TBM, lane, cur, acc, bias, g);
                    const bool act = l + 1 < a.n_mid;
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
#pragma unroll
                        for (int nt = 0; nt < kNT; ++nt) {
                            f32x4 v0 = acc[2 * p][nt], v1 = acc[2 * p + 1][nt];
                            if (act) {
                                nxt[nt].f[2 * h + p] = pack_pair(leaky4(v0, a.slope), leaky4(v1, a.slope));
                            } else {                                     // the bare accumulators: the compiler-visible conversion (HAZARD RULE at pk())
                                const uint2 lo = qa_pack(v0), hi = qa_pack(v1);
                                nxt[nt].f[2 * h + p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                            }
                        }
                    }
                }
#pragma unroll
                for (int nt = 0; nt < kNT; ++nt) cur[nt] = nxt[nt];
            }
        }
        if constexpr (HEAD) {   // out = (conv3x3(raw) + bias) * (lsc + 1), lsc = cur as the two-launch path would re-read it (bf16); pair by pair
            const float* hb = s_b + (1 + a.n_mid) * MT * 16;
            auto head_tile = [&](int m, int nt, const f32x4& b) {        // conv3x3(raw) + bias: the bias is the initial C operand
                f32x4 acc = b;
                mma32(*reinterpret_cast<const uint4*>(s_head + m * 1536 + lane * 16), in.rf[nt], acc);
                mma16(*reinterpret_cast<const uint2*>(s_head + m * 1536 + 1024 + lane * 8), in.rt[nt], acc);
                return acc;
            };
            auto mulp1 = [](const f32x4& h, const f32x4& l) { return __builtin_elementwise_fma(h, l, h); };     // h (l + 1) as one packed fma per pair of values
#pragma unroll
            for (int p = 0; p < MT / 2; ++p) {
                const f32x4 b0 = bias4(hb, 2 * p, g), b1 = bias4(hb, 2 * p + 1, g);
#pragma unroll
                for (int nt = 0; nt < kNT; ++nt)
                    cur[nt].f[p] = pack_pair(mulp1(head_tile(2 * p, nt, b0), up_lo(cur[nt].f[p])), mulp1(head_tile(2 * p + 1, nt, b1), up_hi(cur[nt].f[p])));
            }
            if constexpr (MT & 1) {
                const f32x4 b0 = bias4(hb, MT - 1, g);
#pragma unroll
                for (int nt = 0; nt < kNT; ++nt) cur[nt].t = pack_tail(mulp1(head_tile(MT - 1, nt, b0), up_tail(cur[nt].t)));
            }
        }
        if constexpr (STAGE) {
            // The tile's 64 x C output leaves as whole kilobytes: a lane's pieces (16 + 16 [+ 8] bytes of one token) go through this wave's
            // 6 KB of LDS and come back as 16-byte units in address order.  Stored straight from the B-operand layout they were 64-byte runs at a
            // C * 2-byte pitch, 2-3 instructions per 16 tokens, and those instructions' issue (tools/ubench/store_issue.hip: ~2x the cycles of a
            // whole-kilobyte store, CU-wide) was what the kernel waited for: 1.12 -> see DESIGN.md 4.9 at cfg3.  Same wave writes and reads: the LDS
            // queue is in order, no barrier.
            char* st = s_stage + (tid >> 6) * (64 * C * 2);
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) store_act<C>(reinterpret_cast<bf16_t*>(st) + (16 * nt + n) * C, g, cur[nt]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                         // other lanes' writes are read back below: keep the compiler from
            __builtin_amdgcn_wave_barrier();                                               // moving the reads above them (the LDS queue itself is in order)
            const long long left = (a.pixels - tile * 64) * (C * 2);                       // bytes of this tile inside the tensor
            const __amdgpu_buffer_rsrc_t r_o = __builtin_amdgcn_make_buffer_rsrc(a.out + tile * 64 * C, 0, (int)(left < 64 * C * 2 ? left : 64 * C * 2), 0x00020000);
#pragma unroll
            for (int j = 0; j < (64 * C * 2) / 1024; ++j) {
                const uint4 v = *reinterpret_cast<const uint4*>(st + (lane + 64 * j) * 16);
                typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
                __builtin_amdgcn_raw_buffer_store_b128(u4_t{v.x, v.y, v.z, v.w}, r_o, (lane + 64 * j) * 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                const long long p = tile * 64 + 16 * nt + n;
                if (p < a.pixels) store_act<C>(a.out + p * C, g, cur[nt]);
            }
        }
    }
}

}  // namespace gf
}  // namespace rc

extern "C" size_t rc_lsc_packed_bytes(int c, int n_mid, int has_head) {
    if (!(c == 32 || c == 48 || c == 64 || c == 128) || n_mid < 1) return 0;
    const bool hd = has_head != 0;
    const int w = c == 32 ? lsc_weight_bytes<32>(n_mid, hd) : c == 48 ? lsc_weight_bytes<48>(n_mid, hd) : c == 64 ? lsc_weight_bytes<64>(n_mid, hd) : lsc_weight_bytes<128>(n_mid, hd);
    return (size_t)w + (size_t)(1 + n_mid + (has_head ? 1 : 0)) * (c / 16) * 16 * 4;
}

// Host-side packer: w0 (c, cin0), wmid[l] (c, c), whead (c, raw_c, 3, 3) fp32 row-major as in the state_dict; biases (c) or NULL.
extern "C" int rc_lsc_pack(const float* w0, const float* b0, int cin0, const float* const* wmid, const float* const* bmid, int n_mid,
                           const float* whead, const float* bhead, int raw_c, int c, void* dst) {
    RC_REQUIRE(w0 && wmid && bmid && dst, "rc_lsc_pack: null pointer");
    RC_REQUIRE((c == 32 || c == 48 || c == 64 || c == 128) && n_mid >= 1 && n_mid <= 4 && cin0 >= 1 && cin0 <= 4, "rc_lsc_pack: width 32, 48, 64 or 128, 1..4 mid layers, cin0 <= 4");
    RC_REQUIRE(!whead || (raw_c >= 1 && raw_c <= 4), "rc_lsc_pack: the head's input has 1..4 channels");
    const int mt = c / 16, tbm = tile_bytes(c);
    char* p = static_cast<char*>(dst);
    for (int m = 0; m < mt; ++m)                                         // layer 0: half-filled K-step, k = 4 q + i
        for (int lane = 0; lane < 64; ++lane) {
            const int ch = row_channel(m, lane & 15, mt), q = lane >> 4;
            uint16_t* o = reinterpret_cast<uint16_t*>(p + m * 512 + lane * 8);
            for (int i = 0; i < 4; ++i) o[i] = (4 * q + i < cin0) ? host_f32_to_bf16(w0[(size_t)ch * cin0 + 4 * q + i]) : 0;
        }
    p += mt * 512;
    for (int l = 0; l < n_mid; ++l) {
        RC_REQUIRE(wmid[l] != nullptr, "rc_lsc_pack: null layer");
        const int rc = rc_chain_pack_weights(wmid[l], c, c, p);
        if (rc != RC_OK) return rc;
        p += mt * tbm;
    }
    if (whead) {
        for (int m = 0; m < mt; ++m)
            for (int lane = 0; lane < 64; ++lane) {
                const int ch = row_channel(m, lane & 15, mt), q = lane >> 4;
                uint16_t* o4 = reinterpret_cast<uint16_t*>(p + m * 1536 + lane * 16);
                uint16_t* o2 = reinterpret_cast<uint16_t*>(p + m * 1536 + 1024 + lane * 8);
                auto wv = [&](int tap, int ci) { return ci < raw_c ? host_f32_to_bf16(whead[((size_t)ch * raw_c + ci) * 9 + tap]) : (uint16_t)0; };
                for (int i = 0; i < 8; ++i) o4[i] = wv(2 * q + (i >> 2), i & 3);                       // k = 8 q + i -> tap 2 q + i / 4, channel i % 4
                for (int i = 0; i < 4; ++i) o2[i] = q == 0 ? wv(8, i) : (uint16_t)0;                   // k = 32 + 4 q + i -> tap 8 in lane group 0
            }
        p += mt * 1536;
    }
    float* b = reinterpret_cast<float*>(p);
    rc_chain_pack_bias(b0, c, b);
    for (int l = 0; l < n_mid; ++l) rc_chain_pack_bias(bmid[l], c, b + (1 + l) * mt * 16);
    if (whead) rc_chain_pack_bias(bhead, c, b + (1 + n_mid) * mt * 16);
    return RC_OK;
}

extern "C" int rc_lsc_chain(const void* d_x, int cin0, const void* d_blob, int c, int n_mid, float slope, const void* d_raw, int raw_c,
                            void* d_out, int batch, int H, int W, void* stream) {
    RC_REQUIRE(d_x && d_blob && d_out, "rc_lsc_chain: null pointer");
    RC_REQUIRE((c == 32 || c == 48 || c == 64 || c == 128) && n_mid >= 1 && n_mid <= 4 && cin0 >= 1 && cin0 <= 4, "rc_lsc_chain: width 32, 48, 64 or 128, 1..4 mid layers, cin0 <= 4");
    RC_REQUIRE(!d_raw || (raw_c >= 1 && raw_c <= 4), "rc_lsc_chain: the head's input has 1..4 channels");
    RC_REQUIRE(!(d_raw && c == 128), "rc_lsc_chain: the fused head exists at widths 32 / 48 / 64 (at 128 it spilled 29 registers and the codec needs the lens-shading map itself)");
    RC_REQUIRE(batch >= 1 && H >= 1 && W >= 1 && (long long)batch * H * W * 8 < (1LL << 31), "rc_lsc_chain: bad shape (inputs must stay below 2 GiB)");
    RC_REQUIRE(slope >= 0.f && slope <= 1.f, "rc_lsc_chain: slope must be in [0, 1]");
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_out) % 16 == 0 && reinterpret_cast<uintptr_t>(d_blob) % 16 == 0, "rc_lsc_chain: misaligned pointer");
    LscArgs a{};
    a.x = static_cast<const bf16_t*>(d_x); a.cin0 = cin0; a.blob = static_cast<const char*>(d_blob); a.n_mid = n_mid; a.slope = slope;
    a.raw = static_cast<const bf16_t*>(d_raw); a.raw_c = raw_c; a.out = static_cast<bf16_t*>(d_out);
    a.pixels = (long long)batch * H * W; a.H = H; a.W = W;
    size_t lds = rc_lsc_packed_bytes(c, n_mid, d_raw != nullptr);
    if (c <= 64) lds = ((lds + 15) & ~(size_t)15) + 4 * 64 * (size_t)c * 2;            // + the four waves' output staging (lsc_chain_kernel)
    RC_REQUIRE(lds <= 150 * 1024, "rc_lsc_chain: weights (+ output staging) do not fit LDS");
    const long long tiles = (a.pixels + 63) / 64;
    const int wpb = (c > 64 ? 512 : 256) / 64;
    long long grid = (tiles + wpb - 1) / wpb;
    const long long cap = (long long)device_cu_count() * (c > 64 ? 1 : 3);      // resident blocks per CU (LDS for the wide form, VGPRs for the 48-wide)
    if (grid > cap) grid = cap;
#define RC_LSC(CC, HH)                                                                                                                    \
    do {                                                                                                                                  \
        static PerDeviceFlag attr;                                                                                                        \
        if (!attr.test_and_set())                                                                                                         \
            RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lsc_chain_kernel<CC, HH>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
        hipLaunchKernelGGL((lsc_chain_kernel<CC, HH>), dim3((unsigned)grid), dim3(lsc_threads<CC>()), lds, as_stream(stream), a);              \
    } while (0)
    if (c == 32) { if (d_raw) RC_LSC(32, true); else RC_LSC(32, false); }            // the ISPUNet family's width
    else if (c == 48) { if (d_raw) RC_LSC(48, true); else RC_LSC(48, false); }
    else if (c == 64) { if (d_raw) RC_LSC(64, true); else RC_LSC(64, false); }
    else RC_LSC(128, false);
#undef RC_LSC
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}


// =====================================================================================================================================
// softmax_N(k)^T v on the matrix cores for the dim-80 block (8 heads x 8 channels, segment-planar bf16 qkv'): given the per-channel
// maxima M (folded into rc_gma_aggregate), one pass accumulates  Z[i] = sum_t exp(k[t][i] - M[i])  and  S[i][j] = sum_t exp(..) v[t][j]
// for the 4 diagonal 16 x 16 tiles of the 64 x 64 channel product (a head's 8 x 8 block lies inside one of them).  Tokens are the GEMM's
// K dimension: a 128-token tile of exp(k - M) and of v is transposed through LDS ([channel][token] bf16, 272-byte rows: conflict-free
// ds_read_b128), wave w owns diagonal tile w: 4 MFMAs per tile.  The VALU kernel it replaces (gma_kvsum_kernel) spent 0.60 ms on 64 FMAs
// per token and head; this one is bound by reading k and v once.  Partial sums per block, merged in fixed order by gma_kv_merge_kernel.
namespace rc {
namespace gf {

constexpr int KV_TILE = 128, KV_ROW = KV_TILE + 8, KV_THREADS = 256;
__global__ __launch_bounds__(KV_THREADS) void gma_kvsum_mfma_kernel(const bf16_t* __restrict__ qkvp, size_t plane, const float* __restrict__ kmax,
                                                                     float* __restrict__ part, int n_tok, int L) {
    __shared__ __attribute__((aligned(16))) unsigned short s_e[64 * KV_ROW];     // exp(k - M), [channel][token]
    __shared__ __attribute__((aligned(16))) unsigned short s_v[64 * KV_ROW];     // v, [channel][token]
    __shared__ float s_m[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const int blk = blockIdx.x, b = blockIdx.y, nblk = gridDim.x;
    const float kLog2e = 1.4426950408889634f;
    if (tid < 64) s_m[tid] = kmax[(size_t)b * 64 + tid] * kLog2e;
    const int t0 = blk * L, t1 = (t0 + L) < n_tok ? (t0 + L) : n_tok;
    const int half = tid & 1, tl = tid >> 1;                                    // this thread stages token tl of a tile, channels 16 r + 8 half + 0..7
    const bf16_t* kbase = qkvp + (size_t)4 * plane + (size_t)b * n_tok * kSEG + 8 * half;     // k = segments 4..7, v = 8..11
    const bf16_t* vbase = qkvp + (size_t)8 * plane + (size_t)b * n_tok * kSEG + 8 * half;
    uint4 kr[4], vr[4];
    auto fetch = [&](int tt) {
        const int t = tt + tl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            kr[r] = vr[r] = make_uint4(0u, 0u, 0u, 0u);
            if (t < t1) {
                kr[r] = *reinterpret_cast<const uint4*>(kbase + (size_t)r * plane + (size_t)t * kSEG);
                vr[r] = *reinterpret_cast<const uint4*>(vbase + (size_t)r * plane + (size_t)t * kSEG);
            }
        }
    };
    float z[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) z[r][e] = 0.f;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t0 < t1) fetch(t0);
    __syncthreads();
    for (int tt = t0; tt < t1; tt += KV_TILE) {
        const bool live = tt + tl < t1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned kw[4] = {kr[r].x, kr[r].y, kr[r].z, kr[r].w}, vw[4] = {vr[r].x, vr[r].y, vr[r].z, vr[r].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = 16 * r + 8 * half + e;
                const float kf = __uint_as_float((e & 1) ? (kw[e >> 1] & 0xffff0000u) : (kw[e >> 1] << 16));
                const float ex = live ? __builtin_amdgcn_exp2f(__builtin_fmaf(kf, kLog2e, -s_m[c])) : 0.f;
                const unsigned short eb = (unsigned short)Vec16<bf16_t>::rne(ex);          // compiler-visible conversion (ex is a v_exp result)
                z[r][e] += __uint_as_float((unsigned)eb << 16);
                s_e[c * KV_ROW + tl] = eb;
                s_v[c * KV_ROW + tl] = (unsigned short)(vw[e >> 1] >> (16 * (e & 1)));
            }
        }
        if (tt + KV_TILE < t1) fetch(tt + KV_TILE);                                // next tile in flight under the MFMA phase
        __syncthreads();
#pragma unroll
        for (int st = 0; st < KV_TILE / 32; ++st) {
            const uint4 af = *reinterpret_cast<const uint4*>(s_e + (16 * wave + n) * KV_ROW + 32 * st + 8 * g);
            const uint4 bf = *reinterpret_cast<const uint4*>(s_v + (16 * wave + n) * KV_ROW + 32 * st + 8 * g);
            mma32(af, bf, acc);
        }
        __syncthreads();
    }
    // partial record [Z: 64][S: (h*8 + i)*8 + j], as gma_kv_merge_kernel reads it
    float* rec = part + ((size_t)b * nblk + blk) * (64 + 512);
    {   // D[i][j]: lane (n = j, g), rows i = 4 g + r of diagonal tile `wave`; keep the entries inside a head's 8 x 8 block
        const f32x4 d = acc + 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * g + r;
            if ((i >> 3) == (n >> 3)) rec[64 + (16 * wave + i) * 8 + (n & 7)] = d[r];
        }
    }
    float* s_z = reinterpret_cast<float*>(s_e);                                    // [64 channels][128 contributors] = 32 KB (s_e + s_v are 34 KB)
    static_assert(sizeof(s_e) + sizeof(s_v) >= 64 * 128 * 4, "Z reduction buffer");
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) s_z[(16 * r + 8 * half + e) * 128 + tl] = z[r][e];
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
        for (int k = 0; k < 128; ++k) s += s_z[tid * 128 + k];                   // fixed order
        rec[tid] = s;
    }
}

}  // namespace gf
}  // namespace rc

namespace rc { void gma_kv_merge_launch(const float* part, float* ktv, int nblk, int batch, float scale, void* stream); }

extern "C" int rc_gma_kv_mfma_blocks(int n_tok) {
    int nblk = (n_tok + 2047) / 2048;          // >= 2048 tokens per block
    if (nblk > 512) nblk = 512;
    return nblk < 1 ? 1 : nblk;
}

extern "C" size_t rc_gma_kv_mfma_scratch_bytes(int batch, int n_tok) {
    return (size_t)batch * rc_gma_kv_mfma_blocks(n_tok) * (64 + 512) * sizeof(float);
}

extern "C" int rc_gma_kv_mfma(const void* d_qkvp, int batch, int n_tok, float scale, const float* d_kmax, float* d_scratch, float* d_ktv, void* stream) {
    using namespace rc;
    using namespace rc::gf;
    RC_REQUIRE(d_qkvp && d_kmax && d_scratch && d_ktv, "rc_gma_kv_mfma: null pointer");
    RC_REQUIRE(batch >= 1 && batch <= 65535 && n_tok >= 1, "rc_gma_kv_mfma: bad shape");
    const int nblk = rc_gma_kv_mfma_blocks(n_tok);
    const int L = ((n_tok + nblk - 1) / nblk + KV_TILE - 1) / KV_TILE * KV_TILE;
    hipLaunchKernelGGL(gma_kvsum_mfma_kernel, dim3(nblk, batch), dim3(KV_THREADS), 0, as_stream(stream), static_cast<const bf16_t*>(d_qkvp),
                       (size_t)batch * n_tok * kSEG, d_kmax, d_scratch, n_tok, L);
    gma_kv_merge_launch(d_scratch, d_ktv, nblk, batch, scale, stream);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

// =====================================================================================================================================
// Transformer-block MLP of the codecs as one register-resident chain:  out = x + fc2(GELU(fc1(LayerNorm(x))))   (models/tcm.py:234-235)
// for token width C = 32 or 64 (hidden 4 C), bf16.  Layer by layer this is rc_layernorm + two 1x1 rc_conv2d launches that write and re-read
// the normalised map and the 4C-wide hidden map (2.8 GB per block at the codec's 576 x 960 x 4 stage); here a wave's 64 tokens stay in MFMA
// fragments from the load of x to the store of the result (same machinery as gma_tail_kernel: pair-packed weight rows, hidden channels
// consumed 32 at a time as one fc2 K-step).  Rounding points = the layer-by-layer path's (LayerNorm output and GELU output to bf16).
namespace rc {
namespace gf {

template <int C>
__device__ __forceinline__ void layernorm_c(const Act<C> (&in)[kNT], Act<C> (&out)[kNT], const float* gb, int g, float eps) {
    constexpr int KS = C / 32;
    f32x4 gam[2 * KS], bet[2 * KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        gam[2 * s] = ld4(gb + 32 * s + 8 * g); gam[2 * s + 1] = ld4(gb + 32 * s + 8 * g + 4);
        bet[2 * s] = ld4(gb + C + 32 * s + 8 * g); bet[2 * s + 1] = ld4(gb + C + 32 * s + 8 * g + 4);
    }
#pragma unroll
    for (int nt = 0; nt < kNT; ++nt) {
        f32x4 v[2 * KS];
        f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) { v[2 * s] = up_lo(in[nt].f[s]); v[2 * s + 1] = up_hi(in[nt].f[s]); sv += v[2 * s] + v[2 * s + 1]; }
        float sm = (sv[0] + sv[1]) + (sv[2] + sv[3]);
        sm = sum_lane_groups(sm);
        const float mean = sm / (float)C;
        f32x4 qv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2 * KS; ++i) { v[i] = v[i] - mean; qv += v[i] * v[i]; }
        float q = (qv[0] + qv[1]) + (qv[2] + qv[3]);
        q = sum_lane_groups(q);
        const float rstd = 1.f / sqrtf(q / (float)C + eps);
#pragma unroll
        for (int s = 0; s < KS; ++s) out[nt].f[s] = pack_pair(v[2 * s] * rstd * gam[2 * s] + bet[2 * s], v[2 * s + 1] * rstd * gam[2 * s + 1] + bet[2 * s + 1]);
    }
}

struct MlpArgs {
    const bf16_t* x; bf16_t* out; size_t tokens;
    const void* w_fc1; const float* b_fc1; const void* w_fc2; const float* b_fc2;     // rc_chain_pack_weights / rc_chain_pack_bias
    const float* ln_g; const float* ln_b; float eps;
};

constexpr int kMlpThreads = 256;
template <int C>
__global__ __launch_bounds__(kMlpThreads, 2) void ln_mlp_kernel(const MlpArgs a) {
    constexpr int HID = 4 * C, MT = C / 16, KS = C / 32, TB1 = tile_bytes(C), TB2 = tile_bytes(HID);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* s_fc1 = lds;                                                  // [HID / 16 tiles][TB1]
    char* s_fc2 = s_fc1 + (HID / 16) * TB1;                             // [MT tiles][TB2]
    float* s_b1 = reinterpret_cast<float*>(s_fc2 + MT * TB2);           // [HID]
    float* s_b2 = s_b1 + HID;                                           // [C]
    float* s_gb = s_b2 + C;                                             // gamma[C] | beta[C]
    const int tid = threadIdx.x;
    for (int i = tid; i < (HID / 16) * TB1 / 16; i += kMlpThreads) reinterpret_cast<uint4*>(s_fc1)[i] = reinterpret_cast<const uint4*>(a.w_fc1)[i];
    for (int i = tid; i < MT * TB2 / 16; i += kMlpThreads) reinterpret_cast<uint4*>(s_fc2)[i] = reinterpret_cast<const uint4*>(a.w_fc2)[i];
    for (int i = tid; i < HID; i += kMlpThreads) s_b1[i] = a.b_fc1 ? a.b_fc1[i] : 0.f;
    for (int i = tid; i < C; i += kMlpThreads) { s_b2[i] = a.b_fc2 ? a.b_fc2[i] : 0.f; s_gb[i] = a.ln_g[i]; s_gb[C + i] = a.ln_b[i]; }
    __syncthreads();
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const size_t n_tiles = (a.tokens + 63) / 64, n_waves = (size_t)gridDim.x * (kMlpThreads / 64);
    size_t tile = (size_t)blockIdx.x * (kMlpThreads / 64) + (tid >> 6);
    Act<C> xnext[kNT];
    auto fetch = [&](size_t tl) {
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            const size_t t = tl * 64 + 16 * nt + n;
            load_act<C>(a.x + (t < a.tokens ? t : a.tokens - 1) * C, g, xnext[nt]);
        }
    };
    if (tile < n_tiles) fetch(tile);
    for (; tile < n_tiles; tile += n_waves) {
        Act<C> xin[kNT], n2[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) xin[nt] = xnext[nt];
        if (tile + n_waves < n_tiles) fetch(tile + n_waves);                 // next tile in flight
        layernorm_c<C>(xin, n2, s_gb, g, a.eps);
        f32x4 acc[MT][kNT];                                                   // fc2 accumulators, seeded with the residual x
#pragma unroll
        for (int p = 0; p < KS; ++p)
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) { acc[2 * p][nt] = up_lo(xin[nt].f[p]); acc[2 * p + 1][nt] = up_hi(xin[nt].f[p]); }
#pragma unroll 1
        for (int hc = 0; hc < HID / 32; ++hc) {
            f32x4 h[2][kNT];
            const f32x4 c0 = bias4(s_b1, 2 * hc, g), c1 = bias4(s_b1, 2 * hc + 1, g);
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) { h[0][nt] = c0; h[1][nt] = c1; }
            gemm_tiles<C, 2>(s_fc1, 2 * hc, lane, n2, h);
            uint4 hb[kNT];
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) hb[nt] = pack_pair(gelu_erf4(h[0][nt]), gelu_erf4(h[1][nt]));
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const uint4 w2 = *reinterpret_cast<const uint4*>(s_fc2 + m * TB2 + hc * 1024 + lane * 16);
#pragma unroll
                for (int nt = 0; nt < kNT; ++nt) mma32(w2, hb[nt], acc[m][nt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            Act<C> o;
#pragma unroll
            for (int p = 0; p < KS; ++p) o.f[p] = pack_pair(acc[2 * p][nt] + bias4(s_b2, 2 * p, g), acc[2 * p + 1][nt] + bias4(s_b2, 2 * p + 1, g));
            const size_t t = tile * 64 + 16 * nt + n;
            if (t < a.tokens) store_act<C>(a.out + t * C, g, o);
        }
    }
}

}  // namespace gf
}  // namespace rc

extern "C" int rc_ln_mlp(const void* d_x, void* d_out, long long tokens, int c, const void* d_w_fc1, const float* d_b_fc1, const void* d_w_fc2,
                         const float* d_b_fc2, const float* d_ln_gamma, const float* d_ln_beta, float eps, void* stream) {
    using namespace rc;
    using namespace rc::gf;
    RC_REQUIRE(d_x && d_out && d_w_fc1 && d_w_fc2 && d_ln_gamma && d_ln_beta, "rc_ln_mlp: null pointer");
    RC_REQUIRE(tokens >= 1 && (c == 32 || c == 64), "rc_ln_mlp: token width 32 or 64");
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_x) % 16 == 0 && reinterpret_cast<uintptr_t>(d_out) % 16 == 0, "rc_ln_mlp: misaligned tensor");
    MlpArgs a{static_cast<const bf16_t*>(d_x), static_cast<bf16_t*>(d_out), (size_t)tokens, d_w_fc1, d_b_fc1, d_w_fc2, d_b_fc2, d_ln_gamma, d_ln_beta, eps};
    const int hid = 4 * c;
    const size_t lds = (size_t)(hid / 16) * tile_bytes(c) + (size_t)(c / 16) * tile_bytes(hid) + (size_t)(hid + 3 * c) * 4;
    const long long tiles = (tokens + 63) / 64;
    long long grid = (tiles + 3) / 4;
    const long long cap = (long long)device_cu_count() * 2;
    if (grid > cap) grid = cap;
#define RC_MLP(CC)                                                                                                                         \
    do {                                                                                                                                   \
        static PerDeviceFlag attr;                                                                                                         \
        if (!attr.test_and_set())                                                                                                          \
            RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_mlp_kernel<CC>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)); \
        hipLaunchKernelGGL((ln_mlp_kernel<CC>), dim3((unsigned)grid), dim3(kMlpThreads), lds, as_stream(stream), a);                       \
    } while (0)
    if (c == 32) RC_MLP(32); else RC_MLP(64);
#undef RC_MLP
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

// =====================================================================================================================================
// LayerNorm + Linear as one launch:  out = Linear(LayerNorm(x)),  C = 32 / 64 -> COUT (a multiple of 32), bf16 token-major in and out
// (the W-MSA embedding layer of the codecs' transformer blocks, models/tcm.py:179-181 after :232's ln1).  Same chain machinery; saves the
// normalised map's round trip and a launch.
namespace rc {
namespace gf {

struct LnLinArgs { const bf16_t* x; bf16_t* out; size_t tokens; int cout; const void* w; const float* b; const float* ln_g; const float* ln_b; float eps;
                   int planar8; };   // planar8: out as [cout / 8 segments][tokens][8 channels] (rc_window_attention_planar8's q / k / v layout) instead of (tokens, cout)

template <int C>
__global__ __launch_bounds__(kMlpThreads, 2) void ln_linear_kernel(const LnLinArgs a) {
    constexpr int TB = tile_bytes(C);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int mt = a.cout / 16;
    char* s_w = lds;
    float* s_b = reinterpret_cast<float*>(lds + mt * TB);
    float* s_gb = s_b + a.cout;
    const int tid = threadIdx.x;
    for (int i = tid; i < mt * TB / 16; i += kMlpThreads) reinterpret_cast<uint4*>(s_w)[i] = reinterpret_cast<const uint4*>(a.w)[i];
    for (int i = tid; i < a.cout; i += kMlpThreads) s_b[i] = a.b ? a.b[i] : 0.f;
    for (int i = tid; i < C; i += kMlpThreads) { s_gb[i] = a.ln_g[i]; s_gb[C + i] = a.ln_b[i]; }
    __syncthreads();
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const size_t n_tiles = (a.tokens + 63) / 64, n_waves = (size_t)gridDim.x * (kMlpThreads / 64);
    size_t tile = (size_t)blockIdx.x * (kMlpThreads / 64) + (tid >> 6);
    Act<C> xnext[kNT];
    auto fetch = [&](size_t tl) {
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            const size_t t = tl * 64 + 16 * nt + n;
            load_act<C>(a.x + (t < a.tokens ? t : a.tokens - 1) * C, g, xnext[nt]);
        }
    };
    if (tile < n_tiles) fetch(tile);
    for (; tile < n_tiles; tile += n_waves) {
        Act<C> xin[kNT], n1[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) xin[nt] = xnext[nt];
        if (tile + n_waves < n_tiles) fetch(tile + n_waves);
        layernorm_c<C>(xin, n1, s_gb, g, a.eps);
#pragma unroll 1
        for (int p = 0; p < mt / 2; ++p) {                                   // output channels 32 p .. 32 p + 31: this lane's 8 g .. 8 g + 7 of them
            f32x4 acc[2][kNT];
            zero<2>(acc);
            gemm_tiles<C, 2>(s_w, 2 * p, lane, n1, acc);
            const f32x4 b0 = bias4(s_b, 2 * p, g), b1 = bias4(s_b, 2 * p + 1, g);
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                const size_t t = tile * 64 + 16 * nt + n;
                if (t < a.tokens) {
                    bf16_t* dst = a.planar8 ? a.out + ((size_t)(4 * p + g) * a.tokens + t) * 8 : a.out + t * a.cout + 32 * p + 8 * g;      // planar: 16 tokens x 16 B = 256-byte runs
                    *reinterpret_cast<uint4*>(dst) = pack_pair(acc[0][nt] + b0, acc[1][nt] + b1);
                }
            }
        }
    }
}

}  // namespace gf
}  // namespace rc

static int ln_linear_launch(const void* d_x, void* d_out, long long tokens, int c, int cout, const void* d_w, const float* d_b, const float* d_ln_gamma,
                            const float* d_ln_beta, float eps, int planar8, void* stream);
extern "C" int rc_ln_linear(const void* d_x, void* d_out, long long tokens, int c, int cout, const void* d_w, const float* d_b, const float* d_ln_gamma,
                            const float* d_ln_beta, float eps, void* stream) {
    return ln_linear_launch(d_x, d_out, tokens, c, cout, d_w, d_b, d_ln_gamma, d_ln_beta, eps, 0, stream);
}
extern "C" int rc_ln_linear_planar8(const void* d_x, void* d_out, long long tokens, int c, int cout, const void* d_w, const float* d_b, const float* d_ln_gamma,
                                    const float* d_ln_beta, float eps, void* stream) {
    return ln_linear_launch(d_x, d_out, tokens, c, cout, d_w, d_b, d_ln_gamma, d_ln_beta, eps, 1, stream);
}
static int ln_linear_launch(const void* d_x, void* d_out, long long tokens, int c, int cout, const void* d_w, const float* d_b, const float* d_ln_gamma,
                            const float* d_ln_beta, float eps, int planar8, void* stream) {
    using namespace rc;
    using namespace rc::gf;
    RC_REQUIRE(d_x && d_out && d_w && d_ln_gamma && d_ln_beta, "rc_ln_linear: null pointer");
    RC_REQUIRE(tokens >= 1 && (c == 32 || c == 64) && cout >= 32 && cout % 32 == 0 && cout <= 512, "rc_ln_linear: width 32 / 64 -> a multiple of 32 (<= 512)");
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_x) % 16 == 0 && reinterpret_cast<uintptr_t>(d_out) % 16 == 0, "rc_ln_linear: misaligned tensor");
    LnLinArgs a{static_cast<const bf16_t*>(d_x), static_cast<bf16_t*>(d_out), (size_t)tokens, cout, d_w, d_b, d_ln_gamma, d_ln_beta, eps, planar8};
    const size_t lds = (size_t)(cout / 16) * tile_bytes(c) + (size_t)(cout + 2 * c) * 4;
    const long long tiles = (tokens + 63) / 64;
    long long grid = (tiles + 3) / 4;
    const long long cap = (long long)device_cu_count() * 2;
    if (grid > cap) grid = cap;
#define RC_LL(CC)                                                                                                                          \
    do {                                                                                                                                   \
        static PerDeviceFlag attr;                                                                                                         \
        if (!attr.test_and_set())                                                                                                          \
            RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_linear_kernel<CC>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)); \
        hipLaunchKernelGGL((ln_linear_kernel<CC>), dim3((unsigned)grid), dim3(kMlpThreads), lds, as_stream(stream), a);                    \
    } while (0)
    if (c == 32) RC_LL(32); else RC_LL(64);
#undef RC_LL
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

// =====================================================================================================================================
// GDN / inverse GDN as one per-token chain:  y = x * rsqrt(beta + gamma . x^2)  (inverse: * sqrt(..))  [+ identity]
// (compressai.layers.GDN inside ResidualBlockWithStride / ResidualBlockUpsample, call sites models/tcm.py:336-364).  Layer by layer this
// is rc_square -> rc_conv2d (1x1, gamma / beta) -> rc_gdn_apply with the squared map and the norm map written and re-read; here the
// token's channels go x^2 -> MFMA with gamma -> rsqrt -> scale in registers.  Rounding points as in the three-launch form (x^2 and the
// norm are rounded to bf16 where that form stores them).  bf16, C = 64 or 128.
namespace rc {
namespace gf {

struct GdnArgs { const bf16_t* x; const bf16_t* idn; bf16_t* out; size_t tokens; const void* w; const float* b; int inverse; };

template <int C>
__global__ __launch_bounds__(kMlpThreads, 2) void gdn_chain_kernel(const GdnArgs a) {
    constexpr int MT = C / 16, KS = C / 32, TB = tile_bytes(C);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* s_w = lds;
    float* s_b = reinterpret_cast<float*>(lds + MT * TB);
    const int tid = threadIdx.x;
    for (int i = tid; i < MT * TB / 16; i += kMlpThreads) reinterpret_cast<uint4*>(s_w)[i] = reinterpret_cast<const uint4*>(a.w)[i];
    for (int i = tid; i < C; i += kMlpThreads) s_b[i] = a.b ? a.b[i] : 0.f;
    __syncthreads();
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const size_t n_tiles = (a.tokens + 63) / 64, n_waves = (size_t)gridDim.x * (kMlpThreads / 64);
    for (size_t tile = (size_t)blockIdx.x * (kMlpThreads / 64) + (tid >> 6); tile < n_tiles; tile += n_waves) {
        Act<C> sq[kNT];                                                      // x^2 as the three-launch form stores it (bf16)
        size_t tok[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            const size_t t = tile * 64 + 16 * nt + n;
            tok[nt] = t < a.tokens ? t : a.tokens - 1;
            load_act<C>(a.x + tok[nt] * C, g, sq[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const f32x4 lo = up_lo(sq[nt].f[s]), hi = up_hi(sq[nt].f[s]);
                sq[nt].f[s] = pack_pair(lo * lo, hi * hi);
            }
#pragma unroll 1
        for (int p = 0; p < KS; ++p) {
            // this step's 8 channels of x (32 p + 8 g ..) and of the identity: re-read (cache-hot) under the MFMAs instead of held in registers
            uint4 xv[kNT], iv[kNT];
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                xv[nt] = *reinterpret_cast<const uint4*>(a.x + tok[nt] * C + 32 * p + 8 * g);
                iv[nt] = a.idn != nullptr ? *reinterpret_cast<const uint4*>(a.idn + tok[nt] * C + 32 * p + 8 * g) : make_uint4(0u, 0u, 0u, 0u);
            }
            f32x4 acc[2][kNT];
            zero<2>(acc);
            gemm_tiles<C, 2>(s_w, 2 * p, lane, sq, acc);
            const f32x4 b0 = bias4(s_b, 2 * p, g), b1 = bias4(s_b, 2 * p + 1, g);
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                const uint4 nb = pack_pair(acc[0][nt] + b0, acc[1][nt] + b1);        // the norm as the three-launch form stores it
                const f32x4 n0 = up_lo(nb), n1 = up_hi(nb);
                f32x4 r0, r1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {                                    // v_rsq_f32 / v_sqrt_f32 (1 ulp in fp32, then bf16): the IEEE
                    r0[e] = a.inverse ? __builtin_amdgcn_sqrtf(n0[e]) : __builtin_amdgcn_rsqf(n0[e]);   // division + square root sequences cost ~25
                    r1[e] = a.inverse ? __builtin_amdgcn_sqrtf(n1[e]) : __builtin_amdgcn_rsqf(n1[e]);   // instructions per value
                }
                const f32x4 y0 = up_lo(xv[nt]) * r0 + up_lo(iv[nt]), y1 = up_hi(xv[nt]) * r1 + up_hi(iv[nt]);
                if (tile * 64 + 16 * nt + n < a.tokens) *reinterpret_cast<uint4*>(a.out + tok[nt] * C + 32 * p + 8 * g) = pack_pair(y0, y1);
            }
        }
    }
}

}  // namespace gf
}  // namespace rc

extern "C" int rc_gdn_chain(const void* d_x, const void* d_identity, void* d_out, long long tokens, int c, const void* d_gamma_packed,
                            const float* d_beta_packed, int inverse, void* stream) {
    using namespace rc;
    using namespace rc::gf;
    RC_REQUIRE(d_x && d_out && d_gamma_packed, "rc_gdn_chain: null pointer");
    RC_REQUIRE(tokens >= 1 && (c == 64 || c == 128), "rc_gdn_chain: 64 or 128 channels");
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_x) % 16 == 0 && reinterpret_cast<uintptr_t>(d_out) % 16 == 0 &&
               (d_identity == nullptr || reinterpret_cast<uintptr_t>(d_identity) % 16 == 0), "rc_gdn_chain: misaligned tensor");
    GdnArgs a{static_cast<const bf16_t*>(d_x), static_cast<const bf16_t*>(d_identity), static_cast<bf16_t*>(d_out), (size_t)tokens, d_gamma_packed,
              d_beta_packed, inverse};
    const size_t lds = (size_t)(c / 16) * tile_bytes(c) + (size_t)c * 4;
    const long long tiles = (tokens + 63) / 64;
    long long grid = (tiles + 3) / 4;
    const long long cap = (long long)device_cu_count() * 4;
    if (grid > cap) grid = cap;
    if (c == 64) hipLaunchKernelGGL((gdn_chain_kernel<64>), dim3((unsigned)grid), dim3(kMlpThreads), lds, as_stream(stream), a);
    else hipLaunchKernelGGL((gdn_chain_kernel<128>), dim3((unsigned)grid), dim3(kMlpThreads), lds, as_stream(stream), a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

// =====================================================================================================================================
// Linear over the channel concatenation of two token maps, + residual:  out = res + W . [a ; b] + bias   (the closing
// `conv1_2(torch.cat((conv_x, trans_x), dim=1)) + x` of ConvTransBlock, models/tcm.py:265-267 / raw2bit.py:324-327).  The two halves are read
// straight into the K-steps of one activation fragment set, so the concatenated map is never written; the first half may be given as a
// sum a + a2 (ConvTransBlock's `conv_block(conv_x) + conv_x`, tcm.py:262).  bf16, C = 64 or 128 (halves C/2).
namespace rc {
namespace gf {

struct CatLinArgs { const bf16_t* a; const bf16_t* a2; const bf16_t* b; const bf16_t* res; bf16_t* out; size_t tokens; const void* w; const float* bias; };

template <int C>
__global__ __launch_bounds__(kMlpThreads, 2) void cat_linear_kernel(const CatLinArgs a) {
    constexpr int MT = C / 16, KS = C / 32, H = C / 2, TB = tile_bytes(C);
    static_assert(H % 32 == 0, "each half is a whole number of K-steps");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* s_w = lds;
    float* s_b = reinterpret_cast<float*>(lds + MT * TB);
    const int tid = threadIdx.x;
    for (int i = tid; i < MT * TB / 16; i += kMlpThreads) reinterpret_cast<uint4*>(s_w)[i] = reinterpret_cast<const uint4*>(a.w)[i];
    for (int i = tid; i < C; i += kMlpThreads) s_b[i] = a.bias ? a.bias[i] : 0.f;
    __syncthreads();
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    const size_t n_tiles = (a.tokens + 63) / 64, n_waves = (size_t)gridDim.x * (kMlpThreads / 64);
    for (size_t tile = (size_t)blockIdx.x * (kMlpThreads / 64) + (tid >> 6); tile < n_tiles; tile += n_waves) {
        Act<C> in[kNT];
        size_t tok[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            const size_t t = tile * 64 + 16 * nt + n;
            tok[nt] = t < a.tokens ? t : a.tokens - 1;
#pragma unroll
            for (int s = 0; s < KS / 2; ++s) {
                uint4 av = *reinterpret_cast<const uint4*>(a.a + tok[nt] * H + 32 * s + 8 * g);
                if (a.a2 != nullptr) {                        // first half = a + a2, rounded to bf16 like the separate add launch it replaces
                    const uint4 a2v = *reinterpret_cast<const uint4*>(a.a2 + tok[nt] * H + 32 * s + 8 * g);
                    av = pack_pair(up_lo(av) + up_lo(a2v), up_hi(av) + up_hi(a2v));
                }
                in[nt].f[s] = av;
                in[nt].f[KS / 2 + s] = *reinterpret_cast<const uint4*>(a.b + tok[nt] * H + 32 * s + 8 * g);
            }
        }
#pragma unroll 1
        for (int p = 0; p < KS; ++p) {
            uint4 rv[kNT];
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt)
                rv[nt] = a.res != nullptr ? *reinterpret_cast<const uint4*>(a.res + tok[nt] * C + 32 * p + 8 * g) : make_uint4(0u, 0u, 0u, 0u);
            f32x4 acc[2][kNT];
            zero<2>(acc);
            gemm_tiles<C, 2>(s_w, 2 * p, lane, in, acc);
            const f32x4 b0 = bias4(s_b, 2 * p, g), b1 = bias4(s_b, 2 * p + 1, g);
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt)
                if (tile * 64 + 16 * nt + n < a.tokens)
                    *reinterpret_cast<uint4*>(a.out + tok[nt] * C + 32 * p + 8 * g) = pack_pair(acc[0][nt] + b0 + up_lo(rv[nt]), acc[1][nt] + b1 + up_hi(rv[nt]));
        }
    }
}

}  // namespace gf
}  // namespace rc

extern "C" int rc_cat_linear(const void* d_a, const void* d_a_add, const void* d_b, const void* d_residual, void* d_out, long long tokens, int c,
                             const void* d_w, const float* d_bias, void* stream) {
    using namespace rc;
    using namespace rc::gf;
    RC_REQUIRE(d_a && d_b && d_out && d_w, "rc_cat_linear: null pointer");
    RC_REQUIRE(tokens >= 1 && (c == 64 || c == 128), "rc_cat_linear: concatenated width 64 or 128");
    for (const void* q : {d_a, d_a_add, d_b, d_residual, static_cast<const void*>(d_out)})
        RC_REQUIRE(q == nullptr || reinterpret_cast<uintptr_t>(q) % 16 == 0, "rc_cat_linear: misaligned tensor");
    CatLinArgs a{static_cast<const bf16_t*>(d_a), static_cast<const bf16_t*>(d_a_add), static_cast<const bf16_t*>(d_b), static_cast<const bf16_t*>(d_residual), static_cast<bf16_t*>(d_out),
                 (size_t)tokens, d_w, d_bias};
    const size_t lds = (size_t)(c / 16) * tile_bytes(c) + (size_t)c * 4;
    const long long tiles = (tokens + 63) / 64;
    long long grid = (tiles + 3) / 4;
    const long long cap = (long long)device_cu_count() * 4;
    if (grid > cap) grid = cap;
    if (c == 64) hipLaunchKernelGGL((cat_linear_kernel<64>), dim3((unsigned)grid), dim3(kMlpThreads), lds, as_stream(stream), a);
    else hipLaunchKernelGGL((cat_linear_kernel<128>), dim3((unsigned)grid), dim3(kMlpThreads), lds, as_stream(stream), a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}
