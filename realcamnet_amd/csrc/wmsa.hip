// Window / shifted-window multi-head self-attention core (models/tcm.py:179-206):
//   for every ws x ws window of the (cyclically shifted) NHWC map and every head:
//       out = softmax(q k^T * hd^-0.5 + relpos[h, dy, dx]  (+ -inf across the wrap in the last window row / column)) v
// The two nn.Linear layers around it (embedding_layer C -> 3C, linear C -> C) run through rc_conv2d as 1x1 convolutions
// over the un-shifted map (a point-wise op commutes with torch.roll), so the shift is pure index arithmetic here:
// window token (w1, w2, p1, p2) is pixel ((w1*ws + p1 + s) mod H, (w2*ws + p2 + s) mod W), and the result goes back to
// that same pixel (the reference rolls the output back by +s).
//
// Mapping: one lane = one query token; a wave covers 64 / ws^2 windows of ONE head (1 window at ws = 8, 4 at ws = 4).
// K and V of the wave's windows sit in LDS as fp32 ([token][hd + 1], conflict-free column reads), the head's
// (2ws-1)^2 bias table beside them; every lane computes its ws^2 scores (hd FMAs each), the max-subtracted softmax
// exactly as torch does, then the weighted sum of V.  The work is tiny next to the block's Linear layers
// (2 * ws^2 * C FMAs per token vs 12 * C^2 MACs), so plain FMAs are fine.
#include <type_traits>
#include "common.hpp"

namespace rc {

constexpr int kWmsaWaves = 4;
template <typename T, int HD, int WS>
__global__ __launch_bounds__(kWmsaWaves * 64) void wmsa_kernel(const T* __restrict__ qkv, const float* __restrict__ relpos, T* __restrict__ out,
                                                               int batch, int H, int W, int C, int shift) {
    constexpr int TPW = WS * WS;                    // tokens per window
    constexpr int WPW = 64 / TPW;                   // windows per wave
    constexpr int RP = 2 * WS - 1;
    constexpr int KS = HD + 1;                      // LDS row stride (floats)
    extern __shared__ float wm_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* s_k = wm_smem + wave * (2 * 64 * KS + RP * RP);
    float* s_v = s_k + 64 * KS;
    float* s_bias = s_v + 64 * KS;

    const int nh = C / HD, hw = H / WS, ww = W / WS;
    const long long wgroups = ((long long)batch * hw * ww + WPW - 1) / WPW;     // groups of WPW windows
    const long long job = (long long)blockIdx.x * kWmsaWaves + wave;            // job = (window group, head)
    if (job >= wgroups * nh) return;
    const int h = (int)(job % nh);
    const long long wg = job / nh;
    const int wl = lane / TPW, p = lane % TPW;                                  // window inside the wave, token inside it
    const long long win = wg * WPW + wl;
    const bool live = win < (long long)batch * hw * ww;
    const int b = live ? (int)(win / (hw * ww)) : 0;
    const int w1 = live ? (int)((win / ww) % hw) : 0, w2 = live ? (int)(win % ww) : 0;
    const int p1 = p / WS, p2 = p % WS;
    int y = w1 * WS + p1 + shift, x = w2 * WS + p2 + shift;                    // un-shifted pixel of this token
    if (y >= H) y -= H;
    if (x >= W) x -= W;
    const T* px = qkv + (((size_t)b * H + y) * W + x) * 3 * C + h * HD;

    for (int i = lane; i < RP * RP; i += 64) s_bias[i] = relpos[(size_t)h * RP * RP + i];
    float q[HD];
    const float scale = rsqrtf((float)HD);
#pragma unroll
    for (int c = 0; c < HD; ++c) {
        q[c] = live ? to_f32(px[c]) * scale : 0.f;
        s_k[lane * KS + c] = live ? to_f32(px[C + c]) : 0.f;
        s_v[lane * KS + c] = live ? to_f32(px[2 * C + c]) : 0.f;
    }
    __builtin_amdgcn_wave_barrier();            // one wave's LDS operations complete in order

    // after the cyclic shift the last window row / column holds pixels from both borders: a query sees its own side only
    const bool last_r = shift > 0 && w1 == hw - 1, last_c = shift > 0 && w2 == ww - 1;
    const bool side_y = p1 >= WS - shift, side_x = p2 >= WS - shift;
    float sim[TPW];
    float mx = -__builtin_inff();
    const float* kk = s_k + wl * TPW * KS;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) s = __builtin_fmaf(q[c], kk[j * KS + c], s);
        const int j1 = j / WS, j2 = j % WS;
        s += s_bias[(p1 - j1 + WS - 1) * RP + (p2 - j2 + WS - 1)];
        const bool masked = (last_r && ((j1 >= WS - shift) != side_y)) || (last_c && ((j2 >= WS - shift) != side_x));
        s = masked ? -__builtin_inff() : s;
        sim[j] = s;
        mx = fmaxf(mx, s);
    }
    float den = 0.f, acc[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = 0.f;
    const float* vv = s_v + wl * TPW * KS;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const float e = expf(sim[j] - mx);
        den += e;
#pragma unroll
        for (int c = 0; c < HD; ++c) acc[c] = __builtin_fmaf(e, vv[j * KS + c], acc[c]);
    }
    if (!live) return;
    const float inv = 1.f / den;
    T* o = out + (((size_t)b * H + y) * W + x) * C + h * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = from_f32<T>(acc[c] * inv);
}

// ---- bf16, 8 x 8 windows: the same attention on the matrix cores --------------------------------------------------------------------
// One wave = one (window, head): 64 queries x 64 keys.
//   S^T = K Q^T   : 16 v_mfma_f32_16x16x32_bf16 (A = K rows, B = Q rows; the head_dim <= 32 channels are ONE K-step, zero-padded),
//   softmax over keys: a lane (n, g) holds 16 of query n's 64 scores per query tile (D layout: rows 4g + j of 4 key tiles); max / sum
//                   finish with two xor-shuffles across the 4 lanes of a query,
//   O^T = V^T P^T : the P fragments come straight out of the S accumulators -- the key tiles are loaded in the pair-packed row order
//                   (tile pair (2s, 2s+1), row 4g + j  <->  key 32s + 8g + 4 (tile & 1) + j), so after fp32 -> bf16 packing lane (n, g)
//                   already holds keys 32s + 8g + 0..7 of query n = the B operand of K-step s; V^T (A operand) is transposed through a
//                   wave-private LDS slab.
// Relative-position bias: a per-head table expanded once per block to [query][key in fragment order] (x log2 e, rows padded to 68
// floats: conflict-free ds_read_b128), so the softmax runs on v_exp_f32 directly.  The wrap mask of shifted windows is tile-uniform in
// the row direction and one compare per lane in the column direction, and only the last window row / column takes that path.
// The one-lane-per-query kernel above spent 1.47 ms per call on the codec's 576 x 960 x 4 map with 8 heads of 8 channels
// (VALU FMAs + scalar 2-byte loads); this form is bound by the softmax's VALU work (~5 instructions per score).
constexpr int kWmWaves = 4, kWmBiasRow = 68;
// fp32 pair -> packed bf16 through the compiler-visible conversion, NOT Vec16::rne2: that one is inline asm, invisible to the hazard
// recogniser, and both packs of this kernel read registers with pending-result hazards -- the probabilities come straight from v_exp_f32
// (a transcendental result needs a wait state before a VALU read: the first build packed stale values for a few queries of the first tile
// when head_dim = 8, and only for that schedule) and the outputs come from MFMA accumulators.
typedef float wm_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 wm_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t wm_pk(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(wm_f32x2{lo, hi}, wm_bf16x2));
}
__device__ __forceinline__ void wm_mma(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// max / sum over the four lane groups of a query (lanes n, n + 16, n + 32, n + 48) by gfx950's row / half swaps instead of two __shfl_xor steps each: a
// ds_bpermute is an LDS round trip the wave sits out (16 of them per window and head, in a dependent chain with three waves per SIMD to cover it).  Operand order differs
// from own-op-partner only by commutation: the same bits.  Inline asm with s_nop on both sides: this hipcc mis-lowers the builtin's second result and the swap is opaque
// to the hazard recogniser (csrc/gma_fused.hip sum_lane_groups).
__device__ __forceinline__ float wm_sum_groups(float s) {
    float a = s, b = s;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    s = a + b; a = s; b = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float wm_max_groups(float s) {
    float a = s, b = s;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    s = fmaxf(a, b); a = s; b = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

template <int HD>
__global__ __launch_bounds__(kWmWaves * 64, 2) void wmsa_mfma_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ relpos,
                                                                        bf16_t* __restrict__ out, int batch, int H, int W, int C, int shift,
                                                                        int windows_per_block, int planar8) {
    constexpr int WS = 8, RP = 15, DT = (HD + 15) / 16;                 // DT: 16-row tiles of O^T
    constexpr int VROW = 64 + 8;                                        // V^T slab row (bf16 elements): 144 B, 16-byte aligned, spreads banks
    extern __shared__ __attribute__((aligned(16))) char wm_lds[];
    float* s_bias = reinterpret_cast<float*>(wm_lds);                   // [64 queries][kWmBiasRow]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16_t* s_vt = reinterpret_cast<bf16_t*>(wm_lds + 64 * kWmBiasRow * 4) + wave * (DT * 16 * VROW);
    const int n = lane & 15, g = lane >> 4;
    const int nh = C / HD, hw = H / WS, ww = W / WS;
    // block -> (window chunk, head): the nh blocks that read the same pixels (one per head, 16 of a pixel record's 384 bytes each when
    // head_dim = 8) are 8 apart in launch order, i.e. on the same XCD (block b runs on XCD b mod 8), so the chunk's cache lines are
    // filled into ONE L2 once instead of into eight (head-major order made the head_dim 8 call fabric-bound: 6.8 GB of line fills
    // for 0.85 GB of data, 1.14 ms)
    const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
    const int h = r % nh;
    const long long chunk = (long long)(r / nh) * 8 + xcd;
    const long long n_win = (long long)batch * hw * ww;
    const long long w_begin = chunk * windows_per_block;
    if (w_begin >= n_win) return;
    const long long w_end = (w_begin + windows_per_block) < n_win ? (w_begin + windows_per_block) : n_win;
    const float kLog2e = 1.4426950408889634f;
    // expanded bias: entry [tq][16 mt + 4 gg + j] = relpos[h][p1 - j1 + 7][p2 - j2 + 7] * log2(e), key = 32 (mt >> 1) + 8 gg + 4 (mt & 1) + j
    for (int i = threadIdx.x; i < 64 * 64; i += kWmWaves * 64) {
        const int tq = i >> 6, e = i & 63, mt = e >> 4, gg = (e >> 2) & 3, j = e & 3;
        const int key = 32 * (mt >> 1) + 8 * gg + 4 * (mt & 1) + j;
        const int p1 = tq >> 3, p2 = tq & 7, j1 = key >> 3, j2 = key & 7;
        s_bias[tq * kWmBiasRow + e] = relpos[(size_t)h * RP * RP + (p1 - j1 + WS - 1) * RP + (p2 - j2 + WS - 1)] * kLog2e;
    }
    // zero the padding rows of the V^T slab once (rows HD .. 16 DT - 1 stay zero; rows < HD are rewritten per window)
    for (int i = lane; i < DT * 16 * VROW; i += 64) s_vt[i] = from_f32<bf16_t>(0.f);
    __syncthreads();
    const float scale = rsqrtf((float)HD) * kLog2e;
    const int rec = 3 * C;                                              // elements per pixel record
    // this lane's token roles: A operand of key tile mt: key(mt, R = n); B operand of query tile nt: query 16 nt + n; V staging: token = lane.
    // Addressing is written for instruction count (the head_dim 8 call spends as much on it as on its softmax): the window walk (b, w1, w2) is WAVE-UNIFORM
    // state advanced without divisions (it was three 64-bit divisions per window, in vector registers: `wave` was not known uniform), a role's pixel is the
    // window's first pixel + a lane constant unless the window wraps (last row / column of a shifted call: uniform), offsets are 32-bit inside one image
    // (host: H W 3C < 2^31).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int role_k[4], role_q[4];                                           // (row, column) of the role's token inside the window, as row * W + column
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int key = 32 * (t >> 1) + 8 * (n >> 2) + 4 * (t & 1) + (n & 3), tq = 16 * t + n;
        role_k[t] = (key >> 3) * W + (key & 7); role_q[t] = (tq >> 3) * W + (tq & 7);
    }
    const int role_v = (lane >> 3) * W + (lane & 7);
    int win = (int)w_begin + wave_u;
    int b = win / (hw * ww), w1 = (win - b * (hw * ww)) / ww, w2 = win - b * (hw * ww) - w1 * ww;
    for (; win < (int)w_end; win += kWmWaves) {
        const int cur_w1 = w1, cur_w2 = w2;                             // (the walk below moves w1 / w2 on: everything about THIS window uses these)
        const bool wraps = shift > 0 && (cur_w1 == hw - 1 || cur_w2 == ww - 1);         // uniform
        const int base = (cur_w1 * WS + shift) * W + cur_w2 * WS + shift;
        auto pixel = [&](int role, int t) -> int {                      // window token t (role = its lane constant) -> pixel index inside image b
            if (!wraps) return base + role;
            int y = cur_w1 * WS + (t >> 3) + shift, x = cur_w2 * WS + (t & 7) + shift;
            if (y >= H) y -= H;
            if (x >= W) x -= W;
            return y * W + x;
        };
        // element offset of 8 channels starting at channel c0 of pixel px of image b.  Interleaved (B, H, W, 3C): inside the image's records.  planar8
        // ([3C / 8 segments][B H W pixels][8]; rc_ln_linear_planar8 writes it): a head_dim-8 lane reads 16 of a record's 384 bytes, so the interleaved form pulls a
        // 64-byte sector per lane (6.6 GB of sectors per call for 1.7 GB of q, k, v: the L2 -> L1 path was what the call waited for); in the plane of its segment
        // the 8 pixels of a window row are 128 contiguous bytes.
        const unsigned plane = (unsigned)batch * (unsigned)(H * W);
        const bf16_t* img_in = planar8 ? qkv + (size_t)b * H * W * 8 : qkv + (size_t)b * H * W * rec;           // uniform
        auto at = [&](int px, int c0) -> unsigned {
            return planar8 ? ((unsigned)(c0 >> 3) * plane + (unsigned)px) * 8u : (unsigned)(px * rec + c0);
        };
        bf16_t* img_out = out + (size_t)b * H * W * C;
        w2 += kWmWaves;                                                 // the walk: the next window of this wave
        while (w2 >= ww) { w2 -= ww; ++w1; }
        while (w1 >= hw) { w1 -= hw; ++b; }
        const bool chan = 8 * g < HD;                                   // this lane group carries channels 8 g .. 8 g + 7 of the head
        uint4 qf[4], kf[4];
        int q_px[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int key = 32 * (t >> 1) + 8 * (n >> 2) + 4 * (t & 1) + (n & 3);
            q_px[t] = pixel(role_q[t], 16 * t + n);
            const bf16_t* kp = img_in + at(pixel(role_k[t], key), C + h * HD + 8 * g);
            const bf16_t* qp = img_in + at(q_px[t], h * HD + 8 * g);
            kf[t] = chan ? *reinterpret_cast<const uint4*>(kp) : make_uint4(0u, 0u, 0u, 0u);
            qf[t] = chan ? *reinterpret_cast<const uint4*>(qp) : make_uint4(0u, 0u, 0u, 0u);
        }
        {   // V^T slab: lane = token, rows = channels
            const int v_px = pixel(role_v, lane);
#pragma unroll
            for (int c8 = 0; c8 < HD / 8; ++c8) {
                const uint4 v = *reinterpret_cast<const uint4*>(img_in + at(v_px, 2 * C + h * HD + 8 * c8));
                const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    reinterpret_cast<unsigned short*>(s_vt)[(8 * c8 + e) * VROW + lane] = (unsigned short)(wv[e >> 1] >> (16 * (e & 1)));
            }
        }
        // ---- scores: S^T[key][query], scaled + biased in the log2 domain
        const bool last_r = shift > 0 && cur_w1 == hw - 1, last_c = shift > 0 && cur_w2 == ww - 1;
        const bool q_side_c = (n & 7) >= WS - shift;                    // query column side (queries 16 nt + n: p2 = n & 7)
        uint4 pf[4][2];                                                 // P^T fragments: [query tile][K-step]
        float inv_den[4];
        // Two copies behind ONE uniform branch: left as a condition inside the loop, the wrap mask was if-converted into 64 selects (and 48 operand-quieting v_max
        // in front of the row maximum) per (window, head) for the 99 % of windows that do not wrap -- a sixth of this kernel's vector instructions.
        auto softmax = [&](auto masked_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                f32x4 sc[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) { sc[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; wm_mma(kf[mt], qf[nt], sc[mt]); }
                const float* brow = s_bias + (16 * nt + n) * kWmBiasRow + 4 * g;
                float mx = -__builtin_inff();
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const float4 bb = *reinterpret_cast<const float4*>(brow + 16 * mt);
                    f32x4 v = sc[mt] * scale + f32x4{bb.x, bb.y, bb.z, bb.w};
                    if constexpr (MASKED) {
                        // rows: key j1 = 4 (mt >> 1) + g >= 4  <=>  mt >= 2;  query p1 = 2 nt + (n >> 3) >= 4  <=>  nt >= 2   (shift = 4)
                        const bool mr = last_r && ((mt >> 1) != (nt >> 1));
                        // columns: key j2 = 4 (mt & 1) + j >= 4  <=>  mt odd
                        const bool mc = last_c && (((mt & 1) != 0) != q_side_c);
                        if (mr || mc) v = f32x4{-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
                    }
                    sc[mt] = v;
                    mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
                }
                mx = wm_max_groups(mx);
                float den = 0.f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    f32x4 e;
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(sc[mt][j] - mx);
                    den += (e[0] + e[1]) + (e[2] + e[3]);
                    sc[mt] = e;
                }
                den = wm_sum_groups(den);
                inv_den[nt] = 1.f / den;
#pragma unroll
                for (int st = 0; st < 2; ++st)
                    pf[nt][st] = make_uint4(wm_pk(sc[2 * st][0], sc[2 * st][1]), wm_pk(sc[2 * st][2], sc[2 * st][3]),
                                            wm_pk(sc[2 * st + 1][0], sc[2 * st + 1][1]), wm_pk(sc[2 * st + 1][2], sc[2 * st + 1][3]));
            }
        };
        if (last_r | last_c) softmax(std::true_type{});                 // wave-uniform
        else softmax(std::false_type{});
        __builtin_amdgcn_wave_barrier();                                // the slab writes above are complete (one wave's LDS ops are in order)
        // ---- O^T[d][query] = sum_key V^T[d][key] P^T[key][query]
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            uint4 vf[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) vf[st] = *reinterpret_cast<const uint4*>(s_vt + (16 * dt + n) * VROW + 32 * st + 8 * g);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
                wm_mma(vf[0], pf[nt][0], o);
                wm_mma(vf[1], pf[nt][1], o);
                o = o * inv_den[nt];
                if (16 * dt + 4 * g < HD) {                              // rows 4 g .. 4 g + 3 of this tile are real channels
                    bf16_t* op = img_out + (unsigned)(q_px[nt] * C + h * HD + 16 * dt + 4 * g);
                    *reinterpret_cast<uint2*>(op) = make_uint2(wm_pk(o[0], o[1]), wm_pk(o[2], o[3]));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                                // slab reads done before the next window overwrites it
    }
}

}  // namespace rc

using namespace rc;

extern "C" {

static int window_attention_impl(const void* d_qkv, const float* d_relpos, void* d_out, int dtype, int batch, int H, int W, int C,
                                 int head_dim, int window, int shift, int planar8, void* stream);
int rc_window_attention(const void* d_qkv, const float* d_relpos, void* d_out, int dtype, int batch, int H, int W, int C,
                        int head_dim, int window, int shift, void* stream) {
    return window_attention_impl(d_qkv, d_relpos, d_out, dtype, batch, H, W, C, head_dim, window, shift, 0, stream);
}
int rc_window_attention_planar8(const void* d_qkv, const float* d_relpos, void* d_out, int dtype, int batch, int H, int W, int C,
                                int head_dim, int window, int shift, void* stream) {
    return window_attention_impl(d_qkv, d_relpos, d_out, dtype, batch, H, W, C, head_dim, window, shift, 1, stream);
}
int rc_window_attention_planar8_ok(int dtype, int batch, int H, int W, int C, int window) {
    return dtype == RC_BF16 && window == 8 && batch >= 1 && H >= 8 && W >= 8 && H % 8 == 0 && W % 8 == 0 && (long long)batch * (H / 8) * (W / 8) < (1LL << 30) &&
           (long long)batch * H * W * 3 * C < (1LL << 31);
}
}  // extern "C"
static int window_attention_impl(const void* d_qkv, const float* d_relpos, void* d_out, int dtype, int batch, int H, int W, int C,
                                 int head_dim, int window, int shift, int planar8, void* stream) {
    RC_REQUIRE(d_qkv && d_relpos && d_out, "rc_window_attention: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_window_attention: bad dtype");
    RC_REQUIRE(window == 4 || window == 8, "rc_window_attention: window size must be 4 or 8 (as models/tcm.py uses)");
    RC_REQUIRE(head_dim == 8 || head_dim == 16 || head_dim == 32, "rc_window_attention: head_dim must be 8, 16 or 32");
    RC_REQUIRE(batch >= 1 && H >= window && W >= window && H % window == 0 && W % window == 0, "rc_window_attention: H, W must be multiples of the window size");
    RC_REQUIRE(C >= head_dim && C % head_dim == 0, "rc_window_attention: C must be a multiple of head_dim");
    RC_REQUIRE(shift == 0 || shift == window / 2, "rc_window_attention: shift must be 0 (W-MSA) or window/2 (SW-MSA)");
    if (planar8 && !rc_window_attention_planar8_ok(dtype, batch, H, W, C, window))
        return fail(RC_ERR_UNSUPPORTED, "rc_window_attention_planar8: the segment-planar q / k / v layout exists for the bf16 8 x 8-window matrix-core form with B H W 3C < 2^31");
    // matrix-core form: 32-bit window counts and in-image element offsets (larger maps take the one-lane-per-query kernel below)
    if (dtype == RC_BF16 && window == 8 && (long long)batch * (H / 8) * (W / 8) < (1LL << 30) && (long long)H * W * 3 * C < (1LL << 31)) {
        const int nh = C / head_dim;
        const long long n_win = (long long)batch * (H / 8) * (W / 8);
        const int dtiles = (head_dim + 15) / 16;
        const size_t lds = (size_t)64 * kWmBiasRow * 4 + (size_t)kWmWaves * dtiles * 16 * (64 + 8) * 2;
        // windows per wave: 8 where that still leaves >= 4 blocks per CU, else 4, else 2.  Every block expands its head's bias table first (4 096
        // entries with an integer divide each): at 2 windows per wave that was ~15 % of the big head_dim-8 call (700 -> 590 us at 576 x 960 x 4);
        // the small latent-resolution maps need the blocks more than the amortisation (30 us at 2, 37 at 4).  The nh head-blocks of a chunk stay 8
        // apart in launch order (one XCD), so a chunk's pixels are still filled into one L2
        long long per_wave = 8;
        while (per_wave > 2 && ((n_win + per_wave * kWmWaves - 1) / (per_wave * kWmWaves)) * nh < 4LL * device_cu_count()) per_wave >>= 1;
        long long per_block = per_wave * kWmWaves;
        if (per_block > n_win) per_block = n_win;
        const long long chunks = ((n_win + per_block - 1) / per_block + 7) / 8 * 8;      // whole groups of 8 (one chunk per XCD)
        RC_REQUIRE(chunks * nh < (1LL << 31), "rc_window_attention: too many windows");
#define RC_WMM(HD)                                                                                                                          \
        hipLaunchKernelGGL((wmsa_mfma_kernel<HD>), dim3((unsigned)(chunks * nh)), dim3(kWmWaves * 64), lds, as_stream(stream),              \
                           static_cast<const bf16_t*>(d_qkv), d_relpos, static_cast<bf16_t*>(d_out), batch, H, W, C, shift, (int)per_block, planar8)
        if (head_dim == 8) RC_WMM(8); else if (head_dim == 16) RC_WMM(16); else RC_WMM(32);
#undef RC_WMM
        RC_HIP_CHECK(hipGetLastError());
        return RC_OK;
    }
    const int nh = C / head_dim, wpw = 64 / (window * window);
    const long long jobs = (((long long)batch * (H / window) * (W / window) + wpw - 1) / wpw) * nh;
    const long long blocks = (jobs + kWmsaWaves - 1) / kWmsaWaves;
    RC_REQUIRE(blocks < (1LL << 31), "rc_window_attention: too many windows");
    const size_t lds = (size_t)kWmsaWaves * (2 * 64 * (head_dim + 1) + (2 * window - 1) * (2 * window - 1)) * sizeof(float);
#define RC_WM(TT, HD, WS)                                                                                                 \
    do {                                                                                                                  \
        static PerDeviceFlag attr;                                                                                         \
        if (!attr.test_and_set()) {                                                                                                      \
            RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wmsa_kernel<TT, HD, WS>),                     \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));                     \
        }                                                                                                                 \
        hipLaunchKernelGGL((wmsa_kernel<TT, HD, WS>), dim3((unsigned)blocks), dim3(kWmsaWaves * 64), lds, as_stream(stream), \
                           static_cast<const TT*>(d_qkv), d_relpos, static_cast<TT*>(d_out), batch, H, W, C, shift);     \
    } while (0)
#define RC_WM_HD(TT, WS) do { if (head_dim == 8) RC_WM(TT, 8, WS); else if (head_dim == 16) RC_WM(TT, 16, WS); else RC_WM(TT, 32, WS); } while (0)
#define RC_WM_WS(TT) do { if (window == 8) RC_WM_HD(TT, 8); else RC_WM_HD(TT, 4); } while (0)
    if (dtype == RC_F32) RC_WM_WS(float); else RC_WM_WS(bf16_t);
#undef RC_WM_WS
#undef RC_WM_HD
#undef RC_WM
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}
