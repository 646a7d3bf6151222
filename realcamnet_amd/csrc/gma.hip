// GroupMix attention (GMA_Block) kernels -- upstream models/groupmix.py:159-299.
//
// Tokens are NHWC pixels: a (B,N,C) token tensor IS the (B,H,W,C) activation layout of the conv path, so the
// four Linear layers (qkv, proj, fc1, fc2 = 12 C^2 of the ~13 C^2 MAC/token) run on the MFMA conv kernel as
// 1x1 convolutions.  What is left here is HBM-bound streaming work:
//   dwconv2d        depth-wise KxK (ConvPosEnc, the aggregators' depth-wise stage, ConvRelPosEnc)
//   layernorm       per-token LayerNorm over C
//   gma_pointwise   aggregator tail: per-group seg x seg point-wise conv + BatchNorm(eval) + Hardswish, and
//                   the local branch (3seg -> seg point-wise, LayerNorm(seg), Hardswish)
//   gma_kmax / gma_kvsum / gma_kv_merge   softmax over the N tokens fused with the k^T v contraction: per-channel
//                   max, then exp-sums and k^T v partials per block, folded in fixed order (no (N x C) softmax tensor)
//   gma_apply       out = scale * q (softmax(k)^T v) + q * dwconv(v), concatenated with the local branch
// All arithmetic is fp32 on bf16/fp32 storage; reductions have a fixed order (bitwise reproducible).
#include "common.hpp"

namespace rc {

constexpr int kGThreads = 256;

__device__ __forceinline__ float hardswish(float x) {
    const float r = fminf(fmaxf(x + 3.f, 0.f), 6.f);
    return x * r * (1.f / 6.f);
}

// ---- depth-wise KxK conv on channel sub-ranges of NHWC tensors ---------------------------------------
// y[b,p, y_c0 + r*y_rep + c] = bias[r*w_rep + c] + sum_t wT[t][r*w_rep + c] * x[b, p+t, x_c0 + r*x_rep + c] (+ x[..] itself)
// wT is tap-major (K*K, n_w).  kvec (optional, one int per weight vector) gives the true window of that vector when
// taps are zero-padded to a common K (ConvRelPosEnc mixes 3/5/7 windows): padded taps are skipped, never multiplied.
//
// LDS-tiled: a block owns a 16 x 32 pixel tile of VB consecutive 16-byte channel vectors (one "slot group"); the
// halo tile is staged once (every input vector is fetched from L2/HBM once per tile instead of ~K times per
// output through the vector L1), each wave owns ONE channel vector (so the kvec window test is wave-uniform),
// each lane a 2-row x 4-column output patch: an input row's K+3 vectors are unpacked once and feed both output
// rows and all taps.  The loop is FMA-bound (K*K*UNIT fused multiply-adds per output vector).
constexpr int DW_TH = 16, DW_TW = 32, DW_PX = 4, DW_PY = 2;
template <typename T, int K>
__global__ __launch_bounds__(64 * 5) void dwconv2d_kernel(const T* __restrict__ x, int xs, int x_c0, T* __restrict__ y, int ys,
                                                          int y_c0, int batch, int H, int W, int n_ch,
                                                          const float* __restrict__ wT, int n_w, const float* __restrict__ bias,
                                                          int n_rep, int x_rep, int y_rep, int w_rep, int add_identity,
                                                          const int* __restrict__ kvec, int vb, int tiles_x, int tiles_y) {
    constexpr int U = Vec16<T>::N, R = K / 2, NCOL = K + DW_PX - 1, THH = DW_TH + 2 * R, TWH = DW_TW + 2 * R;
    extern __shared__ __attribute__((aligned(16))) char dw_smem[];
    const int ps = (vb | 1) * 16;                        // pixel stride: an odd number of 16-byte slots
    float* s_w = reinterpret_cast<float*>(dw_smem);      // [K*K][vb*U]
    char* s_x = dw_smem + K * K * vb * U * 4;            // [THH*TWH][ps]

    const int vpc = n_ch / U, groups = n_rep * vpc / vb;
    int blk = blockIdx.x;
    const int g = blk % groups; blk /= groups;
    const int tx = blk % tiles_x; blk /= tiles_x;
    const int ty = blk % tiles_y;
    const int b = blk / tiles_y;
    const int r = (g * vb) / vpc, v0 = (g * vb) % vpc;   // vb divides vpc: a group never straddles two reps
    const int y0 = ty * DW_TH, x0 = tx * DW_TW;
    const int xc = x_c0 + r * x_rep + v0 * U, cw0 = r * w_rep + v0 * U;

    for (int i = threadIdx.x; i < K * K * vb * U; i += blockDim.x) s_w[i] = wT[(i / (vb * U)) * n_w + cw0 + i % (vb * U)];
    {   // halo tile: every thread owns channel vector tid % vb of pixels tid / vb + 64 k; all its loads are issued before the
        // first LDS write (a load -> write loop pays the memory latency once per pixel: 14 round trips per 7 x 7 tile)
        constexpr int NLD = (THH * TWH + 63) / 64;
        const int sv = threadIdx.x % vb, sp0 = threadIdx.x / vb;
        uint4 raw[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int pix = sp0 + 64 * k;
            const int py = pix / TWH, px = pix - py * TWH;
            const int gy = y0 + py - R, gx = x0 + px - R;
            raw[k] = make_uint4(0u, 0u, 0u, 0u);
            if (pix < THH * TWH && gy >= 0 && gy < H && gx >= 0 && gx < W)
                raw[k] = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + gy) * W + gx) * xs + xc + sv * U);
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int pix = sp0 + 64 * k;
            if (pix < THH * TWH) *reinterpret_cast<uint4*>(s_x + pix * ps + sv * 16) = raw[k];
        }
    }
    __syncthreads();

    const int vv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave = channel vector
    const int lane = threadIdx.x & 63, rp = lane >> 3, cq = lane & 7;
    const int cw = cw0 + vv * U;
    const int Rv = kvec ? kvec[cw / U] / 2 : R;          // this vector's true half-window (wave-uniform)
    float acc[DW_PY][DW_PX][U];
#pragma unroll
    for (int e = 0; e < U; ++e) {
        const float bv = bias ? bias[cw + e] : 0.f;
#pragma unroll
        for (int o = 0; o < DW_PY; ++o)
#pragma unroll
            for (int c = 0; c < DW_PX; ++c) acc[o][c][e] = bv;
    }
    const char* base = s_x + ((DW_PY * rp) * TWH + DW_PX * cq) * ps + vv * 16;
    const float* wbase = s_w + vv * U;
#pragma unroll
    for (int iy = 0; iy < K + DW_PY - 1; ++iy) {         // input row iy of this lane's patch feeds output row o with dy = iy - o
        if (iy + 1 < R - Rv || iy > R + Rv + DW_PY - 1) continue;          // outside every output row's window (uniform)
        float f[NCOL][U];
#pragma unroll
        for (int c = 0; c < NCOL; ++c) Vec16<T>::unpack(*reinterpret_cast<const uint4*>(base + (iy * TWH + c) * ps), f[c]);
#pragma unroll
        for (int o = 0; o < DW_PY; ++o) {
            const int dy = iy - o;
            if (dy < 0 || dy >= K) continue;                               // compile-time after unrolling
            if (dy < R - Rv || dy > R + Rv) continue;                      // uniform
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
                if (dx < R - Rv || dx > R + Rv) continue;                  // uniform
                float wv[U];
                const float* wp = wbase + (dy * K + dx) * vb * U;
#pragma unroll
                for (int e = 0; e < U; e += 4) {
                    const float4 t4 = *reinterpret_cast<const float4*>(wp + e);
                    wv[e] = t4.x; wv[e + 1] = t4.y; wv[e + 2] = t4.z; wv[e + 3] = t4.w;
                }
#pragma unroll
                for (int c = 0; c < DW_PX; ++c)
#pragma unroll
                    for (int e = 0; e < U; ++e) acc[o][c][e] = __builtin_fmaf(wv[e], f[c + dx][e], acc[o][c][e]);
            }
            if (add_identity && dy == R) {
#pragma unroll
                for (int c = 0; c < DW_PX; ++c)
#pragma unroll
                    for (int e = 0; e < U; ++e) acc[o][c][e] += f[c + R][e];
            }
        }
    }
#pragma unroll
    for (int o = 0; o < DW_PY; ++o)
#pragma unroll
        for (int c = 0; c < DW_PX; ++c) {
            const int gy = y0 + DW_PY * rp + o, gx = x0 + DW_PX * cq + c;
            if (gy < H && gx < W)
                *reinterpret_cast<uint4*>(y + (((size_t)b * H + gy) * W + gx) * ys + y_c0 + r * y_rep + (v0 + vv) * U) = Vec16<T>::pack(acc[o][c]);
        }
}

// ---- LayerNorm over the channel dim of (tokens, C); 16 lanes per token -------------------------------
template <typename T>
__global__ void layernorm_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, size_t tokens, int c, float eps) {
    constexpr int U = Vec16<T>::N, MAXV = 4;            // up to 16*4 vectors: C <= 512 (bf16) / 256 (fp32)
    const int nvec = c / U;
    const int sub = threadIdx.x & 15;
    const size_t tok0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const size_t stride = ((size_t)gridDim.x * blockDim.x) >> 4;
    for (size_t t = tok0; t < tokens + 0; t += stride) {   // whole 16-lane groups stay together (t uniform per group)
        float f[MAXV][U];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int v = sub + 16 * k;
            if (v < nvec) {
                Vec16<T>::unpack(*reinterpret_cast<const uint4*>(x + t * c + v * U), f[k]);
#pragma unroll
                for (int e = 0; e < U; ++e) s += f[k][e];
            }
        }
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
        const float mean = s / (float)c;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int v = sub + 16 * k;
            if (v < nvec) {
#pragma unroll
                for (int e = 0; e < U; ++e) { const float d = f[k][e] - mean; q += d * d; }
            }
        }
        q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4); q += __shfl_xor(q, 8);
        const float rstd = 1.f / sqrtf(q / (float)c + eps);
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int v = sub + 16 * k;
            if (v < nvec) {
                float o[U];
#pragma unroll
                for (int e = 0; e < U; ++e) o[e] = (f[k][e] - mean) * rstd * gamma[v * U + e] + beta[v * U + e];
                *reinterpret_cast<uint4*>(y + t * c + v * U) = Vec16<T>::pack(o);
            }
        }
    }
}

// ---- aggregator tail ---------------------------------------------------------------------------------
// job = which*4 + g (g = 0..3): in = (g == 0 ? qkv[.., which*C + 0..seg) : dw[.., which, (g-1)*seg ..]);
//   out qkvp[.., which, g*seg + s] = hardswish(bn_scale[g][s] * (g == 0 ? in[s] : sum_j pw[g-1][s][j] in[j]) + bn_shift[g][s])
// job = 12: local: in = dwl[.., 0..3seg); t[s] = sum_j pwl[s][j] in[j]; out loc[.., s] = hardswish(LN_seg(t)[s])
// Block = 13 waves (832 threads): wave j runs job j for a tile of TOK tokens (lane = token), so a wave never
// diverges.  The local branch's 16 x 48 product is three times a point-wise job, so it is split by input segment over the
// three g = 0 waves (whose own job has no product); wave 12 adds the partial sums and applies LayerNorm + Hardswish.  The tile's input slices (g0 of q,k,v from qkv; dw; dwl = 15 segments per token) are staged into LDS
// with coalesced 16-byte loads, results (qkvp rows + loc = 13 segments per token) are assembled in LDS and written
// back coalesced.  LDS rows are padded to an odd number of 16-byte slots -> conflict-free per-lane ds_read_b128.
// The job's weight matrix is wave-uniform read-only data: indexed straight from global it goes through the scalar
// cache into SGPRs (an LDS copy would cost one ds_read per FMA).
constexpr int kPwWaves = 13;
template <typename T, int SEG, int TOK>
__global__ __launch_bounds__(kPwWaves * 64) void gma_pointwise_kernel(
    const T* __restrict__ qkv, const T* __restrict__ dw, const T* __restrict__ dwl, int dw_ts, int dw_rs, int dwl_ts, int dwl_rs,
    T* __restrict__ qkvp, T* __restrict__ loc,
    size_t tokens, int c, const float* __restrict__ pw /*3,seg,seg*/, const float* __restrict__ bn_scale /*4,seg*/,
    const float* __restrict__ bn_shift, const float* __restrict__ pwl /*seg,3seg*/, const float* __restrict__ ln_g,
    const float* __restrict__ ln_b) {
    constexpr int U = Vec16<T>::N, NV = SEG / U;           // vectors per segment
    static_assert(SEG % U == 0, "segment must be whole 16-byte vectors");
    constexpr int IN_V = 15 * NV, OUT_V = 13 * NV;          // vectors per token: staged inputs / outputs
    constexpr int IN_S = IN_V | 1, OUT_S = OUT_V | 1;       // row strides in 16-byte slots (odd)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint4* s_in = reinterpret_cast<uint4*>(lds);            // [TOK][IN_S]
    uint4* s_out = s_in + TOK * IN_S;                       // [TOK][OUT_S]
    float* s_part = reinterpret_cast<float*>(s_out + TOK * OUT_S);   // [TOK][3][SEG] partial sums of the local branch
    const int tid = threadIdx.x, lane = tid & 63;
    const int job = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform -> scalar weight loads
    const int which = job >> 2, g = job & 3;
    const float* __restrict__ sw = job == 12 ? pwl : pw + (g > 0 ? (g - 1) * SEG * SEG : 0);
    const size_t n_tiles = (tokens + TOK - 1) / TOK;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t t0 = tile * TOK;
        const int nt = (int)((tokens - t0) < (size_t)TOK ? (tokens - t0) : (size_t)TOK);
        // ---- stage in: slot layout per token = [q.g0 | k.g0 | v.g0 | dw (9 seg) | dwl (3 seg)] -----------------
        for (int i = tid; i < TOK * IN_V; i += kPwWaves * 64) {
            const int t = i / IN_V, v = i - t * IN_V;
            uint4 raw = make_uint4(0u, 0u, 0u, 0u);
            if (t < nt) {
                const size_t tok = t0 + t;
                const int sg = v / NV, vv = v - sg * NV;     // segment index 0..14
                const T* src = sg < 3 ? qkv + tok * 3 * c + (size_t)sg * c + vv * U
                             : sg < 12 ? dw + tok * dw_ts + ((sg - 3) / 3) * dw_rs + ((sg - 3) % 3) * SEG + vv * U
                                       : dwl + tok * dwl_ts + (sg - 12) * dwl_rs + vv * U;
                raw = *reinterpret_cast<const uint4*>(src);
            }
            s_in[t * IN_S + v] = raw;
        }
        __syncthreads();
        // ---- compute: wave = job, lane = token ------------------------------------------------------------------
        if (lane < TOK) {
            float out[SEG];
            const uint4* row = s_in + lane * IN_S;
            if (job < 12) {
                float in[SEG];
                const int sg = g == 0 ? which : 3 + which * 3 + (g - 1);
#pragma unroll
                for (int v = 0; v < NV; ++v) Vec16<T>::unpack(row[sg * NV + v], in + v * U);
                if (g == 0) {
#pragma unroll
                    for (int s2 = 0; s2 < SEG; ++s2) out[s2] = in[s2];
                } else {
#pragma unroll
                    for (int s2 = 0; s2 < SEG; ++s2) {
                        float a = 0.f;
#pragma unroll
                        for (int j = 0; j < SEG; ++j) a += sw[s2 * SEG + j] * in[j];
                        out[s2] = a;
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < SEG; ++s2) out[s2] = hardswish(out[s2] * bn_scale[g * SEG + s2] + bn_shift[g * SEG + s2]);
                uint4* dst = s_out + lane * OUT_S + (which * 4 + g) * NV;      // qkvp row: [which][g][seg]
#pragma unroll
                for (int v = 0; v < NV; ++v) dst[v] = Vec16<T>::pack(out + v * U);
                if (g == 0) {
                    // one third of the local branch's 16 x 48 product (this wave's own job is only BN + Hardswish): the
                    // `which`-th input segment of dwl against the matching 16 columns of pwl; summed by wave 12 below
                    float lin[SEG];
#pragma unroll
                    for (int v = 0; v < NV; ++v) Vec16<T>::unpack(row[(12 + which) * NV + v], lin + v * U);
                    float* part = s_part + (lane * 3 + which) * SEG;
#pragma unroll
                    for (int s2 = 0; s2 < SEG; ++s2) {
                        float a = 0.f;
#pragma unroll
                        for (int j = 0; j < SEG; ++j) a += pwl[s2 * 3 * SEG + which * SEG + j] * lin[j];
                        part[s2] = a;
                    }
                }
            }
        }
        __syncthreads();
        if (job == 12 && lane < TOK) {
            float out[SEG];
            const float* part = s_part + lane * 3 * SEG;
            float mean = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < SEG; ++s2) { out[s2] = (part[s2] + part[SEG + s2]) + part[2 * SEG + s2]; mean += out[s2]; }
            mean /= (float)SEG;
            float var = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < SEG; ++s2) { const float d = out[s2] - mean; var += d * d; }
            const float rstd = 1.f / sqrtf(var / (float)SEG + 1e-5f);
#pragma unroll
            for (int s2 = 0; s2 < SEG; ++s2) out[s2] = hardswish((out[s2] - mean) * rstd * ln_g[s2] + ln_b[s2]);
            uint4* dst = s_out + lane * OUT_S + 12 * NV;
#pragma unroll
            for (int v = 0; v < NV; ++v) dst[v] = Vec16<T>::pack(out + v * U);
        }
        __syncthreads();
        // ---- stage out: qkvp rows are 12 segments contiguous per token, loc 1 segment ---------------------------
        for (int i = tid; i < TOK * OUT_V; i += kPwWaves * 64) {
            const int t = i / OUT_V, v = i - t * OUT_V;
            if (t < nt) {
                const size_t tok = t0 + t;
                T* dst = v < 12 * NV ? qkvp + tok * 12 * SEG + v * U : loc + tok * SEG + (v - 12 * NV) * U;
                *reinterpret_cast<uint4*>(dst) = s_out[t * OUT_S + v];
            }
        }
        // no barrier needed here: the next iteration's stage-in only writes s_in, which every wave finished reading
        // before the barrier above; s_out is rewritten only after the next iteration's first barrier.
    }
}

// ---- softmax over N fused with k^T v (no (N x C) softmax tensor is ever written) ---------------------------
// qkvp (B,N,3,ct): k = [..,1,:], v = [..,2,:]; ct = heads*ch.
//  pass A  gma_kmax:  per-block partial max of k per channel              -> pmax[b][blkA][ct]
//  pass B  gma_kvsum: M = max over pmax (block prologue), then per tile p = exp(k - M):
//                     Z[c] += p,  KTV[h][i][j] += p[h,i] * v[h,j]         -> part[b][blkB] = { Z[ct], KTV[nacc] }
//  merge              fixed-order sum over blocks, ktv = scale * KTV / Z
// Address of channel vector v (U channels) of q/k/v `which` at token t of image b.  plane == 0: token-major (B,N,3,ct);
// plane > 0: planar by 16-channel segment, [3 * ct / 16][B * N][16] with `plane` elements per segment (bf16: a vector is half a segment).
template <typename T>
__device__ __forceinline__ const T* qkvp_vec(const T* qkvp, size_t plane, int b, int n_tok, int ct, int which, int t, int v) {
    constexpr int U = Vec16<T>::N;
    const size_t tok = (size_t)b * n_tok + t;
    if (plane == 0) return qkvp + tok * 3 * ct + which * ct + v * U;
    const int c = v * U;
    return qkvp + (size_t)(which * (ct / 16) + c / 16) * plane + tok * 16 + (c & 15);
}

template <typename T>
__global__ __launch_bounds__(kGThreads) void gma_kmax_kernel(const T* __restrict__ qkvp, size_t plane, float* __restrict__ pmax, int n_tok, int L, int ct) {
    constexpr int U = Vec16<T>::N;
    extern __shared__ float sm[];             // [groups][ct]
    const int vpt = ct / U, groups = kGThreads / vpt;
    const int tid = threadIdx.x, blk = blockIdx.x, b = blockIdx.y;
    const int v = tid % vpt, grp = tid / vpt;
    const int t0 = blk * L, t1 = (t0 + L) < n_tok ? (t0 + L) : n_tok;
    float m[U];
#pragma unroll
    for (int e = 0; e < U; ++e) m[e] = -INFINITY;
    if (grp < groups) {
        for (int t = t0 + grp; t < t1; t += groups) {
            float f[U];
            Vec16<T>::unpack(*reinterpret_cast<const uint4*>(qkvp_vec(qkvp, plane, b, n_tok, ct, 1, t, v)), f);
#pragma unroll
            for (int e = 0; e < U; ++e) m[e] = fmaxf(m[e], f[e]);
        }
#pragma unroll
        for (int e = 0; e < U; ++e) sm[grp * ct + v * U + e] = m[e];
    }
    __syncthreads();
    for (int c = tid; c < ct; c += kGThreads) {
        float r = -INFINITY;
        for (int g = 0; g < groups; ++g) r = fmaxf(r, sm[g * ct + c]);
        pmax[((size_t)b * gridDim.x + blk) * ct + c] = r;
    }
}

template <typename T, int MAXA>   // MAXA = accumulators per thread >= ceil(heads*ch*ch / 256)
__global__ __launch_bounds__(kGThreads, 2) void gma_kvsum_kernel(const T* __restrict__ qkvp, size_t plane, const float* __restrict__ pmax, int nblk_a,
                                                              float* __restrict__ part, int n_tok, int L, int heads, int ch, int tile) {
    constexpr int U = Vec16<T>::N;
    const int ct = heads * ch, nacc = heads * ch * ch, vpt = ct / U;
    extern __shared__ float sm[];
    float* s_p = sm;                          // [tile][ct]  exp(k - M)
    float* s_v = s_p + tile * ct;             // [tile][ct]
    float* s_m = s_v + tile * ct;             // [ct]
    const int tid = threadIdx.x, blk = blockIdx.x, b = blockIdx.y;
    {   // M[c] = max over pass A's partial maxima: all 256 threads, (channel, part) split, combined through LDS
        const int parts = kGThreads / ct > 0 ? kGThreads / ct : 1;
        for (int c0 = 0; c0 < ct; c0 += kGThreads) {
            const int c = c0 + tid % (ct < kGThreads ? ct : kGThreads), part = tid / (ct < kGThreads ? ct : kGThreads);
            float r = -INFINITY;
            if (c < ct && part < parts)
                for (int k = part; k < nblk_a; k += parts) r = fmaxf(r, pmax[((size_t)b * nblk_a + k) * ct + c]);
            s_p[tid] = r;
            __syncthreads();
            if (tid < ct - c0 && tid < kGThreads) {
                float m = -INFINITY;
                for (int q2 = 0; q2 < parts; ++q2) m = fmaxf(m, s_p[q2 * (ct < kGThreads ? ct : kGThreads) + tid]);
                s_m[c0 + tid] = m;
            }
            __syncthreads();
        }
    }
    float acc[MAXA];
#pragma unroll
    for (int a = 0; a < MAXA; ++a) acc[a] = 0.f;
    float z = 0.f;                            // thread c < ct owns Z[c]
    __syncthreads();
    const int t0 = blk * L, t1 = (t0 + L) < n_tok ? (t0 + L) : n_tok;
    const int nvec_tile = tile * vpt;
    for (int tt = t0; tt < t1; tt += tile) {
        const int nt = (t1 - tt) < tile ? (t1 - tt) : tile;
        for (int i = tid; i < nvec_tile; i += kGThreads) {
            const int t = i / vpt, v = i - t * vpt;
            float fk[U], fv[U];
            if (t < nt) {
                Vec16<T>::unpack(*reinterpret_cast<const uint4*>(qkvp_vec(qkvp, plane, b, n_tok, ct, 1, tt + t, v)), fk);
                Vec16<T>::unpack(*reinterpret_cast<const uint4*>(qkvp_vec(qkvp, plane, b, n_tok, ct, 2, tt + t, v)), fv);
#pragma unroll
                for (int e = 0; e < U; ++e) fk[e] = expf(fk[e] - s_m[v * U + e]);
            } else {
#pragma unroll
                for (int e = 0; e < U; ++e) { fk[e] = 0.f; fv[e] = 0.f; }
            }
#pragma unroll
            for (int e = 0; e < U; e += 4) {
                *reinterpret_cast<float4*>(s_p + t * ct + v * U + e) = make_float4(fk[e], fk[e + 1], fk[e + 2], fk[e + 3]);
                *reinterpret_cast<float4*>(s_v + t * ct + v * U + e) = make_float4(fv[e], fv[e + 1], fv[e + 2], fv[e + 3]);
            }
        }
        __syncthreads();
        if (tid < ct) {
            for (int t = 0; t < tile; ++t) z += s_p[t * ct + tid];
        }
#pragma unroll
        for (int a = 0; a < MAXA; ++a) {
            const int o = tid + a * kGThreads;                // o = (h*ch + i)*ch + j
            if (o < nacc) {
                const int j = o % ch, hi = o / ch, h = hi / ch;
                float sacc = acc[a];
                const float* pk = s_p + hi;
                const float* pv = s_v + h * ch + j;
                for (int t = 0; t < tile; ++t) sacc += pk[t * ct] * pv[t * ct];
                acc[a] = sacc;
            }
        }
        __syncthreads();
    }
    float* out = part + ((size_t)b * gridDim.x + blk) * (ct + nacc);
    if (tid < ct) out[tid] = z;
#pragma unroll
    for (int a = 0; a < MAXA; ++a) {
        const int o = tid + a * kGThreads;
        if (o < nacc) out[ct + o] = acc[a];
    }
}

// fixed-order sum of the partials; ktv[b][h][i][j] = scale * softmax-normalised k^T v.
// Block (chunk, b): 64 outputs x 4 parts; each part sums every 4th block, parts are combined in a fixed order.
__global__ __launch_bounds__(kGThreads) void gma_kv_merge_kernel(const float* __restrict__ part, float* __restrict__ ktv,
                                                                 int nblk, int heads, int ch, float scale) {
    __shared__ float s_s[kGThreads], s_z[kGThreads];
    const int ct = heads * ch, nacc = heads * ch * ch, rec = ct + nacc;
    const int b = blockIdx.y, o = blockIdx.x * 64 + (threadIdx.x & 63), prt = threadIdx.x >> 6;
    const float* p = part + (size_t)b * nblk * rec;
    float z = 0.f, sacc = 0.f;
    if (o < nacc) {
        const int hi = o / ch;
        for (int k = prt; k < nblk; k += 4) { z += p[(size_t)k * rec + hi]; sacc += p[(size_t)k * rec + ct + o]; }
    }
    s_s[threadIdx.x] = sacc; s_z[threadIdx.x] = z;
    __syncthreads();
    if (prt == 0 && o < nacc) {
        const int l = threadIdx.x;
        const float zt = ((s_z[l] + s_z[l + 64]) + s_z[l + 128]) + s_z[l + 192];
        const float st = ((s_s[l] + s_s[l + 64]) + s_s[l + 128]) + s_s[l + 192];
        ktv[(size_t)b * nacc + o] = scale * st / zt;
    }
}

// ---- out[b,n, h*ch + j] = sum_i q[h,i] * ktv[h][i][j] + q[h,j] * convv[h*ch + j];  out[.., ct + s] = loc[s] ----
template <typename T, int CH>   // CH > 0: compile-time head width; 0: runtime (<= 32)
__global__ void gma_apply_kernel(const T* __restrict__ qkvp, const T* __restrict__ convv, const T* __restrict__ loc,
                                 const float* __restrict__ ktv, T* __restrict__ out, int n_tok, int heads, int ch_rt, int seg) {
    const int ch = CH > 0 ? CH : ch_rt;
    extern __shared__ float s_ktv[];          // this image's [heads][ch*ch + 1]: adjacent lanes work on different heads of one token,
                                              // and with a head stride of ch*ch = 64 floats they would all hit one LDS bank
                                              // (PMC: 0.87 conflict cycles per LDS cycle, the kernel was LDS-bound)
    const int ct = heads * ch, c = ct + seg, nacc = heads * ch * ch, hs = ch * ch + 1, b = blockIdx.y;
    for (int i = threadIdx.x; i < nacc; i += blockDim.x) s_ktv[(i / (ch * ch)) * hs + i % (ch * ch)] = ktv[(size_t)b * nacc + i];
    __syncthreads();
    const size_t jobs = (size_t)n_tok * (heads + 1);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < jobs; i += (size_t)gridDim.x * blockDim.x) {
        const int h = (int)(i % (heads + 1));
        const size_t t = (size_t)b * n_tok + i / (heads + 1);
        T* dst = out + t * c;
        if (h == heads) {
            for (int s = 0; s < seg; ++s) dst[ct + s] = loc[t * seg + s];
            continue;
        }
        constexpr int MAXCH = CH > 0 ? CH : 32;
        float q[MAXCH];
        const T* qp = qkvp + t * 3 * ct + h * ch;
#pragma unroll
        for (int k = 0; k < ch; ++k) q[k] = to_f32(qp[k]);
        const float* m = s_ktv + h * hs;
#pragma unroll
        for (int j = 0; j < ch; ++j) {
            float a = q[j] * to_f32(convv[t * ct + h * ch + j]);
#pragma unroll
            for (int k = 0; k < ch; ++k) a += q[k] * m[k * ch + j];
            dst[h * ch + j] = from_f32<T>(a);
        }
    }
}

// softmax-normalise + scale the block partials of the 8-head x 8-channel product (also used by csrc/gma_fused.hip's MFMA pass)
void gma_kv_merge_launch(const float* part, float* ktv, int nblk, int batch, float scale, void* stream) {
    hipLaunchKernelGGL(gma_kv_merge_kernel, dim3((8 * 8 * 8 + 63) / 64, batch), dim3(kGThreads), 0, as_stream(stream), part, ktv, nblk, 8, 8, scale);
}

static inline int pw_grid(size_t n) {
    size_t g = (n + kGThreads - 1) / kGThreads;
    if (g < 1) g = 1;
    if (g > 256 * 32) g = 256 * 32;
    return (int)g;
}

}  // namespace rc

namespace rc {
int g_dw3_seg16 = 1;     // rc_debug_set("dw3_seg16", v): 1 (default) bf16 3x3 single-rep calls of rc_dwconv2d take the 16-channel-segment kernel (gma_fused.hip); 0: the general one
namespace gf {
int launch_dw3x3_seg16(const void* x, int xs, int x_c0, void* y, int ys, int y_c0, int batch, int H, int W, int n_ch, const float* wT, int n_w,
                       const float* bias, int add_identity, hipStream_t stream);
}
}
using namespace rc;

extern "C" {

int rc_dwconv2d(const void* d_x, int x_stride_c, int x_c0, void* d_y, int y_stride_c, int y_c0, int dtype,
                int batch, int H, int W, int n_ch, int ksize, const float* d_wT, int n_w, const float* d_bias,
                int n_rep, int x_rep_stride, int y_rep_stride, int w_rep_stride, int add_identity, const int* d_kvec,
                void* stream) {
    RC_REQUIRE(d_x && d_y && d_wT, "rc_dwconv2d: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_dwconv2d: bad dtype");
    const int U = dtype == RC_F32 ? 4 : 8;
    RC_REQUIRE(batch >= 1 && H >= 1 && W >= 1 && n_ch >= U && n_rep >= 1, "rc_dwconv2d: bad shape");
    RC_REQUIRE(ksize == 3 || ksize == 5 || ksize == 7, "rc_dwconv2d: kernel size must be 3, 5 or 7");
    RC_REQUIRE(n_ch % U == 0 && x_stride_c % U == 0 && y_stride_c % U == 0 && x_c0 % U == 0 && y_c0 % U == 0 &&
               x_rep_stride % U == 0 && y_rep_stride % U == 0 && w_rep_stride % U == 0,
               "rc_dwconv2d: channel counts/offsets must be multiples of 16 bytes");
    RC_REQUIRE(x_c0 + (n_rep - 1) * x_rep_stride + n_ch <= x_stride_c && y_c0 + (n_rep - 1) * y_rep_stride + n_ch <= y_stride_c &&
               (n_rep - 1) * w_rep_stride + n_ch <= n_w, "rc_dwconv2d: channel range exceeds tensor");
    // bf16 3x3 on 16-channel segments without reps / per-vector windows (ConvPosEnc): the aggregator's conflict-free core, same bits (gma_fused.hip)
    if (dtype == RC_BF16 && ksize == 3 && n_rep == 1 && d_kvec == nullptr && n_ch % 16 == 0 && g_dw3_seg16)
        return gf::launch_dw3x3_seg16(d_x, x_stride_c, x_c0, d_y, y_stride_c, y_c0, batch, H, W, n_ch, d_wT, n_w, d_bias, add_identity, as_stream(stream));
    const int vpc = n_ch / U;
    int vb = 1;                                           // channel vectors (= waves) per block: largest divisor of vpc <= 5
    for (int c = 5; c >= 1; --c) if (vpc % c == 0) { vb = c; break; }   // (narrower blocks for mixed kvec windows measured slower:
                                                                         //  32-byte pixel pieces fetch badly)
    const int tiles_x = ceil_div(W, DW_TW), tiles_y = ceil_div(H, DW_TH);
    const int R = ksize / 2;
    const size_t lds = (size_t)ksize * ksize * vb * U * 4 + (size_t)(DW_TH + 2 * R) * (DW_TW + 2 * R) * (vb | 1) * 16;
    const size_t blocks = (size_t)tiles_x * tiles_y * batch * (n_rep * vpc / vb);
    RC_REQUIRE(blocks < (1ull << 31), "rc_dwconv2d: too many tiles");
    RC_REQUIRE(n_w % 4 == 0, "rc_dwconv2d: n_w must be a multiple of 4");
#define RC_DW_LAUNCH(TT, KK)                                                                                              \
    do {                                                                                                                  \
        static PerDeviceFlag attr_set;                                                                                     \
        if (!attr_set.test_and_set()) {                                                                                                  \
            RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv2d_kernel<TT, KK>),                     \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                    \
        }                                                                                                                 \
        hipLaunchKernelGGL((dwconv2d_kernel<TT, KK>), dim3((unsigned)blocks), dim3(64 * vb), lds, as_stream(stream),      \
                           static_cast<const TT*>(d_x), x_stride_c, x_c0, static_cast<TT*>(d_y), y_stride_c, y_c0, batch, \
                           H, W, n_ch, d_wT, n_w, d_bias, n_rep, x_rep_stride, y_rep_stride, w_rep_stride, add_identity,  \
                           d_kvec, vb, tiles_x, tiles_y);                                                                 \
    } while (0)
#define RC_DW_K(TT) if (ksize == 3) RC_DW_LAUNCH(TT, 3); else if (ksize == 5) RC_DW_LAUNCH(TT, 5); else RC_DW_LAUNCH(TT, 7);
    if (dtype == RC_F32) { RC_DW_K(float) } else { RC_DW_K(bf16_t) }
#undef RC_DW_K
#undef RC_DW_LAUNCH
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_layernorm(const void* d_x, void* d_y, int dtype, long long tokens, int c, const float* d_gamma,
                 const float* d_beta, float eps, void* stream) {
    RC_REQUIRE(d_x && d_y && d_gamma && d_beta, "rc_layernorm: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_layernorm: bad dtype");
    const int U = dtype == RC_F32 ? 4 : 8;
    RC_REQUIRE(tokens >= 1 && c >= U && c % U == 0 && c / U <= 64, "rc_layernorm: C must be a multiple of 16 bytes, at most 64 vectors");
    const size_t threads = (size_t)tokens * 16;
    if (dtype == RC_F32)
        hipLaunchKernelGGL(layernorm_kernel<float>, dim3(pw_grid(threads)), dim3(kGThreads), 0, as_stream(stream),
                           static_cast<const float*>(d_x), static_cast<float*>(d_y), d_gamma, d_beta, (size_t)tokens, c, eps);
    else
        hipLaunchKernelGGL(layernorm_kernel<bf16_t>, dim3(pw_grid(threads)), dim3(kGThreads), 0, as_stream(stream),
                           static_cast<const bf16_t*>(d_x), static_cast<bf16_t*>(d_y), d_gamma, d_beta, (size_t)tokens, c, eps);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_gma_pointwise(const void* d_qkv, const void* d_dw, const void* d_dwl, int dw_tok_stride, int dw_rep_stride,
                     int dwl_tok_stride, int dwl_rep_stride, void* d_qkvp, void* d_loc, int dtype, long long tokens, int c,
                     const float* d_pw, const float* d_bn_scale, const float* d_bn_shift, const float* d_pwl, const float* d_ln_g, const float* d_ln_b, void* stream) {
    RC_REQUIRE(d_qkv && d_dw && d_dwl && d_qkvp && d_loc && d_pw && d_bn_scale && d_bn_shift && d_pwl && d_ln_g && d_ln_b,
               "rc_gma_pointwise: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_gma_pointwise: bad dtype");
    RC_REQUIRE(tokens >= 1 && c >= 10 && c % 5 == 0, "rc_gma_pointwise: C must be a multiple of 5");
    const int seg = c / 5;
    RC_REQUIRE(seg == 16 || seg == 40 || seg == 8 || seg == 24 || seg == 32, "rc_gma_pointwise: C/5 must be one of 8, 16, 24, 32, 40 (dims 40..200)");
    const int U = dtype == RC_F32 ? 4 : 8;
    RC_REQUIRE(dw_tok_stride >= 9 * seg && dw_rep_stride >= 3 * seg && dwl_tok_stride >= 3 * seg && dwl_rep_stride >= seg &&
               dw_tok_stride % U == 0 && dw_rep_stride % U == 0 && dwl_tok_stride % U == 0 && dwl_rep_stride % U == 0,
               "rc_gma_pointwise: bad dw/dwl strides");
    const int nv = seg / U;
    const int per_tok = ((15 * nv | 1) + (13 * nv | 1)) * 16 + 3 * seg * 4;            // staged in + out rows, local-branch partial sums
    const int tok = per_tok * 64 <= 150 * 1024 ? 64 : 32;                               // tokens per tile that fit LDS
    const size_t lds = (size_t)per_tok * tok;
    RC_REQUIRE(lds <= 160 * 1024, "rc_gma_pointwise: tile does not fit LDS");
    size_t n_tiles = ((size_t)tokens + tok - 1) / tok;
    const unsigned gx = (unsigned)(n_tiles < 1024 ? n_tiles : 1024);
#define RC_PW_LAUNCH(TT, SG, TK)                                                                                        \
    do {                                                                                                                \
        static PerDeviceFlag attr;                                                                                       \
        if (!attr.test_and_set()) {                                                                                                    \
            RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gma_pointwise_kernel<TT, SG, TK>),          \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                  \
        }                                                                                                               \
        hipLaunchKernelGGL((gma_pointwise_kernel<TT, SG, TK>), dim3(gx), dim3(kPwWaves * 64), lds, as_stream(stream),   \
                           static_cast<const TT*>(d_qkv), static_cast<const TT*>(d_dw), static_cast<const TT*>(d_dwl),  \
                           dw_tok_stride, dw_rep_stride, dwl_tok_stride, dwl_rep_stride, static_cast<TT*>(d_qkvp), static_cast<TT*>(d_loc), (size_t)tokens, c, d_pw, d_bn_scale,      \
                           d_bn_shift, d_pwl, d_ln_g, d_ln_b);                                                          \
    } while (0)
#define RC_PW_TOK(TT, SG) do { if (tok == 64) RC_PW_LAUNCH(TT, SG, 64); else RC_PW_LAUNCH(TT, SG, 32); } while (0)
#define RC_PW_SEG(TT)                                                                                                   \
    switch (seg) { case 8: RC_PW_TOK(TT, 8); break; case 16: RC_PW_TOK(TT, 16); break; case 24: RC_PW_TOK(TT, 24); break; \
                   case 32: RC_PW_TOK(TT, 32); break; default: RC_PW_TOK(TT, 40); break; }
    if (dtype == RC_F32) { RC_PW_SEG(float); } else { RC_PW_SEG(bf16_t); }
#undef RC_PW_SEG
#undef RC_PW_TOK
#undef RC_PW_LAUNCH
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_gma_kv_blocks(int n_tok) {
    int nblk = (n_tok + 1023) / 1024;          // >= 1024 tokens per block, at most 256 blocks per image
    if (nblk > 256) nblk = 256;
    if (nblk < 1) nblk = 1;
    return nblk;
}

size_t rc_gma_kv_scratch_bytes(int batch, int n_tok, int heads, int ch) {
    // [ partial maxima: nblk x ct ] [ partial sums: nblk x (ct + heads*ch*ch) ] per image
    return (size_t)batch * rc_gma_kv_blocks(n_tok) * (2 * heads * ch + heads * ch * ch) * sizeof(float);
}

static int gma_kv_impl(const void* d_qkvp, size_t plane, int dtype, int batch, int n_tok, int heads, int ch, float scale, float* d_scratch,
                       float* d_ktv, void* stream) {
    RC_REQUIRE(d_qkvp && d_scratch && d_ktv, "rc_gma_kv: null pointer");
    RC_REQUIRE(plane == 0 || (dtype == RC_BF16 && (heads * ch) % 16 == 0), "rc_gma_kv_planar: bf16 with whole 16-channel segments only");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_gma_kv: bad dtype");
    const int U = dtype == RC_F32 ? 4 : 8;
    RC_REQUIRE(batch >= 1 && batch <= 65535 && n_tok >= 1 && heads >= 1 && ch >= 1 && ch <= 32 && heads * ch * ch <= 256 * 16 && heads * ch <= kGThreads /* one thread owns Z[c] */ &&
               (heads * ch) % U == 0 && (heads * ch) / U <= kGThreads, "rc_gma_kv: unsupported head geometry");
    const int ct = heads * ch;
    const int nblk = rc_gma_kv_blocks(n_tok);
    const int L = (n_tok + nblk - 1) / nblk;
    const int nblk_a = nblk < 32 ? nblk : 32;     // the max pass is light: few blocks keep pass B's prologue short
    const int L_a = (n_tok + nblk_a - 1) / nblk_a;
    float* pmax = d_scratch;
    float* part = d_scratch + (size_t)batch * nblk * ct;
    int tile = 6144 / ct;                      // two fp32 tiles of `tile` tokens in <= 48 KiB of LDS
    tile = tile > 64 ? 64 : tile / 8 * 8;
    RC_REQUIRE(tile >= 8, "rc_gma_kv: too many attention channels");
    const size_t lds_a = (size_t)(kGThreads / (ct / U)) * ct * sizeof(float);
    const size_t lds_b = ((size_t)2 * tile * ct + ct) * sizeof(float);
    RC_REQUIRE(lds_a <= 64 * 1024 && lds_b <= 64 * 1024, "rc_gma_kv: too many attention channels");
    const int nacc = heads * ch * ch;
#define RC_KV_LAUNCH(TT, MA)                                                                                              \
    hipLaunchKernelGGL((gma_kvsum_kernel<TT, MA>), dim3(nblk, batch), dim3(kGThreads), lds_b, as_stream(stream),          \
                       static_cast<const TT*>(d_qkvp), plane, pmax, nblk_a, part, n_tok, L, heads, ch, tile)
#define RC_KV_BOTH(TT)                                                                                                    \
    hipLaunchKernelGGL(gma_kmax_kernel<TT>, dim3(nblk_a, batch), dim3(kGThreads), lds_a, as_stream(stream),                \
                       static_cast<const TT*>(d_qkvp), plane, pmax, n_tok, L_a, ct);                                             \
    if (nacc <= 2 * kGThreads) RC_KV_LAUNCH(TT, 2); else if (nacc <= 4 * kGThreads) RC_KV_LAUNCH(TT, 4); else RC_KV_LAUNCH(TT, 16);
    if (dtype == RC_F32) { RC_KV_BOTH(float) } else { RC_KV_BOTH(bf16_t) }
#undef RC_KV_BOTH
#undef RC_KV_LAUNCH
    hipLaunchKernelGGL(gma_kv_merge_kernel, dim3((heads * ch * ch + 63) / 64, batch), dim3(kGThreads), 0, as_stream(stream), part, d_ktv, nblk, heads,
                       ch, scale);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_gma_kv(const void* d_qkvp, int dtype, int batch, int n_tok, int heads, int ch, float scale, float* d_scratch,
              float* d_ktv, void* stream) {
    return gma_kv_impl(d_qkvp, 0, dtype, batch, n_tok, heads, ch, scale, d_scratch, d_ktv, stream);
}

int rc_gma_kv_planar(const void* d_qkvp, int batch, int n_tok, int heads, int ch, float scale, float* d_scratch, float* d_ktv, void* stream) {
    return gma_kv_impl(d_qkvp, (size_t)batch * n_tok * 16, RC_BF16, batch, n_tok, heads, ch, scale, d_scratch, d_ktv, stream);
}

int rc_gma_apply(const void* d_qkvp, const void* d_convv, const void* d_loc, const float* d_ktv, void* d_out, int dtype,
                 int batch, int n_tok, int heads, int ch, int seg, void* stream) {
    RC_REQUIRE(d_qkvp && d_convv && d_loc && d_ktv && d_out, "rc_gma_apply: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_gma_apply: bad dtype");
    RC_REQUIRE(batch >= 1 && batch <= 65535 && n_tok >= 1 && heads >= 1 && ch >= 1 && ch <= 32 && seg >= 1, "rc_gma_apply: bad shape");
    const size_t lds = (size_t)heads * (ch * ch + 1) * sizeof(float);
    RC_REQUIRE(lds <= 64 * 1024, "rc_gma_apply: too many attention channels");
    size_t g = ((size_t)n_tok * (heads + 1) + kGThreads - 1) / kGThreads;
    if (g > 4096) g = 4096;
#define RC_AP_LAUNCH(TT, CC)                                                                                            \
    hipLaunchKernelGGL((gma_apply_kernel<TT, CC>), dim3((unsigned)g, batch), dim3(kGThreads), lds, as_stream(stream),   \
                       static_cast<const TT*>(d_qkvp), static_cast<const TT*>(d_convv), static_cast<const TT*>(d_loc), \
                       d_ktv, static_cast<TT*>(d_out), n_tok, heads, ch, seg)
    if (dtype == RC_F32) {
        if (ch == 8) RC_AP_LAUNCH(float, 8); else if (ch == 20) RC_AP_LAUNCH(float, 20); else RC_AP_LAUNCH(float, 0);
    } else {
        if (ch == 8) RC_AP_LAUNCH(bf16_t, 8); else if (ch == 20) RC_AP_LAUNCH(bf16_t, 20); else RC_AP_LAUNCH(bf16_t, 0);
    }
#undef RC_AP_LAUNCH
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

}  // extern "C"
