// GroupMix attention (GMA_Block) kernels -- upstream models/groupmix.py:159-299.
//
// Tokens are NHWC pixels: a (B,N,C) token tensor IS the (B,H,W,C) activation layout of the conv path, so the
// four Linear layers (qkv, proj, fc1, fc2 = 12 C^2 of the ~13 C^2 MAC/token) run on the MFMA conv kernel as
// 1x1 convolutions.  What is left here is HBM-bound streaming work:
//   dwconv2d        depth-wise KxK (ConvPosEnc, the aggregators' depth-wise stage, ConvRelPosEnc)
//   layernorm       per-token LayerNorm over C
//   gma_pointwise   aggregator tail: per-group seg x seg point-wise conv + BatchNorm(eval) + Hardswish, and
//                   the local branch (3seg -> seg point-wise, LayerNorm(seg), Hardswish)
//   gma_kv_reduce   ONE streaming pass over k and v: softmax over the N tokens (online max/sum) fused with the
//                   k^T v contraction -> per-block partials;  gma_kv_merge folds them in fixed order
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
// wT is tap-major (K*K, n_w) so a thread's UNIT weights per tap are one contiguous load.
template <typename T>
__global__ void dwconv2d_kernel(const T* __restrict__ x, int xs, int x_c0, T* __restrict__ y, int ys, int y_c0,
                                int batch, int H, int W, int n_ch, int K, const float* __restrict__ wT, int n_w,
                                const float* __restrict__ bias, int n_rep, int x_rep, int y_rep, int w_rep,
                                int add_identity) {
    constexpr int U = Vec16<T>::N;
    const int vpc = n_ch / U, R = K / 2;
    const size_t total = (size_t)batch * H * W * n_rep * vpc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vpc);
        const int r = (int)((i / vpc) % n_rep);
        const size_t p = i / ((size_t)vpc * n_rep);
        const int px = (int)(p % W), py = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
        const int cw = r * w_rep + v * U;
        float acc[U];
#pragma unroll
        for (int e = 0; e < U; ++e) acc[e] = bias ? bias[cw + e] : 0.f;
        const int xc = x_c0 + r * x_rep + v * U;
        for (int dy = 0; dy < K; ++dy) {
            const int gy = py + dy - R;
            if (gy < 0 || gy >= H) continue;
            for (int dx = 0; dx < K; ++dx) {
                const int gx = px + dx - R;
                if (gx < 0 || gx >= W) continue;
                float f[U];
                Vec16<T>::unpack(*reinterpret_cast<const uint4*>(x + (((size_t)b * H + gy) * W + gx) * xs + xc), f);
                const float* wp = wT + (size_t)(dy * K + dx) * n_w + cw;
#pragma unroll
                for (int e = 0; e < U; ++e) acc[e] += wp[e] * f[e];
                if (add_identity && dy == R && dx == R) {
#pragma unroll
                    for (int e = 0; e < U; ++e) acc[e] += f[e];
                }
            }
        }
        *reinterpret_cast<uint4*>(y + p * ys + y_c0 + r * y_rep + v * U) = Vec16<T>::pack(acc);
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
template <typename T, int SEG>   // SEG > 0: compile-time segment width (arrays stay in registers); 0: runtime (<= 48)
__global__ void gma_pointwise_kernel(const T* __restrict__ qkv, const T* __restrict__ dw, const T* __restrict__ dwl,
                                     T* __restrict__ qkvp, T* __restrict__ loc, size_t tokens, int c, int seg_rt,
                                     const float* __restrict__ pw /*3,seg,seg*/, const float* __restrict__ bn_scale /*4,seg*/,
                                     const float* __restrict__ bn_shift, const float* __restrict__ pwl /*seg,3seg*/,
                                     const float* __restrict__ ln_g, const float* __restrict__ ln_b) {
    const int seg = SEG > 0 ? SEG : seg_rt;
    extern __shared__ float sw[];   // pw | pwl
    float* s_pw = sw;
    float* s_pwl = sw + 3 * seg * seg;
    for (int i = threadIdx.x; i < 3 * seg * seg; i += blockDim.x) s_pw[i] = pw[i];
    for (int i = threadIdx.x; i < 3 * seg * seg; i += blockDim.x) s_pwl[i] = pwl[i];
    __syncthreads();
    constexpr int MAXSEG = SEG > 0 ? SEG : 48;
    const size_t total = tokens * 13;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int job = (int)(i % 13);
        const size_t t = i / 13;
        float out[MAXSEG];
        if (job < 12) {
            const int which = job >> 2, g = job & 3;
            if (g == 0) {
                const T* src = qkv + t * 3 * c + (size_t)which * c;
#pragma unroll
                for (int s = 0; s < seg; ++s) out[s] = to_f32(src[s]);
            } else {
                const T* src = dw + (t * 3 + which) * 3 * seg + (g - 1) * seg;
                float in[MAXSEG];
#pragma unroll
                for (int s = 0; s < seg; ++s) in[s] = to_f32(src[s]);
                const float* wm = s_pw + (g - 1) * seg * seg;
#pragma unroll
                for (int s = 0; s < seg; ++s) {
                    float a = 0.f;
#pragma unroll
                    for (int j = 0; j < seg; ++j) a += wm[s * seg + j] * in[j];
                    out[s] = a;
                }
            }
            T* dst = qkvp + (t * 3 + which) * 4 * seg + g * seg;
#pragma unroll
            for (int s = 0; s < seg; ++s)
                dst[s] = from_f32<T>(hardswish(out[s] * bn_scale[g * seg + s] + bn_shift[g * seg + s]));
        } else {
            const T* src = dwl + t * 3 * seg;
            float mean = 0.f;
            float lin[3 * MAXSEG];
#pragma unroll
            for (int j = 0; j < 3 * seg; ++j) lin[j] = to_f32(src[j]);
#pragma unroll
            for (int s = 0; s < seg; ++s) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < 3 * seg; ++j) a += s_pwl[s * 3 * seg + j] * lin[j];
                out[s] = a; mean += a;
            }
            mean /= (float)seg;
            float var = 0.f;
#pragma unroll
            for (int s = 0; s < seg; ++s) { const float d = out[s] - mean; var += d * d; }
            const float rstd = 1.f / sqrtf(var / (float)seg + 1e-5f);
            T* dst = loc + t * seg;
#pragma unroll
            for (int s = 0; s < seg; ++s) dst[s] = from_f32<T>(hardswish((out[s] - mean) * rstd * ln_g[s] + ln_b[s]));
        }
    }
}

// ---- softmax over N fused with k^T v: one streaming pass with online rescaling ------------------------
// qkvp (B,N,3,ct): k = [..,1,:], v = [..,2,:]; ct = heads*ch.  Block (blk, b) folds tokens [blk*L, (blk+1)*L) in
// tiles of TT tokens: running per-channel max M and sum Z, and KTV[h][i][j] = sum_n exp(k[n][h,i]-M[h,i]) v[n][h,j].
// Partials: part[(b*nblk + blk)] = { M[ct], Z[ct], KTV[heads*ch*ch] }.
constexpr int kKvTile = 32;
template <typename T>
__global__ __launch_bounds__(kGThreads) void gma_kv_reduce_kernel(const T* __restrict__ qkvp, float* __restrict__ part,
                                                                  int n_tok, int L, int heads, int ch) {
    const int ct = heads * ch, nacc = heads * ch * ch;
    extern __shared__ float sm[];
    float* s_k = sm;                          // [TT][ct] -> exp(k - M) in place
    float* s_v = s_k + kKvTile * ct;          // [TT][ct]
    float* s_m = s_v + kKvTile * ct;          // [ct] running max
    float* s_f = s_m + ct;                    // [ct] rescale factor of this tile
    float* s_z = s_f + ct;                    // [ct] running sum
    const int tid = threadIdx.x, blk = blockIdx.x, b = blockIdx.y;
    constexpr int MAXA = 16;                  // accumulators per thread: nacc <= 256*16
    float acc[MAXA];
#pragma unroll
    for (int a = 0; a < MAXA; ++a) acc[a] = 0.f;
    for (int c = tid; c < ct; c += kGThreads) { s_m[c] = -INFINITY; s_z[c] = 0.f; }
    __syncthreads();
    const int t0 = blk * L, t1 = (t0 + L) < n_tok ? (t0 + L) : n_tok;
    const T* base = qkvp + (size_t)b * n_tok * 3 * ct;
    for (int tt = t0; tt < t1; tt += kKvTile) {
        const int nt = (t1 - tt) < kKvTile ? (t1 - tt) : kKvTile;
        for (int i = tid; i < kKvTile * ct; i += kGThreads) {
            const int t = i / ct, c = i - t * ct;
            float kv = -INFINITY, vv = 0.f;
            if (t < nt) {
                const T* tok = base + (size_t)(tt + t) * 3 * ct;
                kv = to_f32(tok[ct + c]); vv = to_f32(tok[2 * ct + c]);
            }
            s_k[i] = kv; s_v[i] = vv;
        }
        __syncthreads();
        for (int c = tid; c < ct; c += kGThreads) {           // new running max + rescale factor
            float m = s_m[c];
            const float m_old = m;
            for (int t = 0; t < nt; ++t) m = fmaxf(m, s_k[t * ct + c]);
            s_f[c] = (m_old == -INFINITY) ? 0.f : expf(m_old - m);
            s_m[c] = m;
        }
        __syncthreads();
        for (int i = tid; i < kKvTile * ct; i += kGThreads) { // p = exp(k - M)  (padding tokens: exp(-inf) = 0)
            const int c = i % ct;
            s_k[i] = expf(s_k[i] - s_m[c]);
        }
        __syncthreads();
        for (int c = tid; c < ct; c += kGThreads) {
            float z = s_z[c] * s_f[c];
            for (int t = 0; t < nt; ++t) z += s_k[t * ct + c];
            s_z[c] = z;
        }
#pragma unroll
        for (int a = 0; a < MAXA; ++a) {
            const int o = tid + a * kGThreads;                // o = (h*ch + i)*ch + j
            if (o < nacc) {
                const int j = o % ch, hi = o / ch, h = hi / ch;
                float s = acc[a] * s_f[hi];
                const float* pk = s_k + hi;
                const float* pv = s_v + h * ch + j;
                for (int t = 0; t < nt; ++t) s += pk[t * ct] * pv[t * ct];
                acc[a] = s;
            }
        }
        __syncthreads();
    }
    float* out = part + ((size_t)b * gridDim.x + blk) * (2 * ct + nacc);
    for (int c = tid; c < ct; c += kGThreads) { out[c] = s_m[c]; out[ct + c] = s_z[c]; }
#pragma unroll
    for (int a = 0; a < MAXA; ++a) {
        const int o = tid + a * kGThreads;
        if (o < nacc) out[2 * ct + o] = acc[a];
    }
}

// merge partials in fixed order; ktv[b][h][i][j] = scale * softmax-normalised k^T v
__global__ __launch_bounds__(kGThreads) void gma_kv_merge_kernel(const float* __restrict__ part, float* __restrict__ ktv,
                                                                 int nblk, int heads, int ch, float scale) {
    const int ct = heads * ch, nacc = heads * ch * ch, rec = 2 * ct + nacc;
    const int b = blockIdx.x;
    const float* p = part + (size_t)b * nblk * rec;
    for (int o = threadIdx.x; o < nacc; o += kGThreads) {
        const int hi = o / ch;
        float m = -INFINITY;
        for (int k = 0; k < nblk; ++k) m = fmaxf(m, p[(size_t)k * rec + hi]);
        float z = 0.f, s = 0.f;
        for (int k = 0; k < nblk; ++k) {
            const float mk = p[(size_t)k * rec + hi];
            const float f = (mk == -INFINITY) ? 0.f : expf(mk - m);
            z += p[(size_t)k * rec + ct + hi] * f;
            s += p[(size_t)k * rec + 2 * ct + o] * f;
        }
        ktv[(size_t)b * nacc + o] = scale * s / z;
    }
}

// ---- out[b,n, h*ch + j] = sum_i q[h,i] * ktv[h][i][j] + q[h,j] * convv[h*ch + j];  out[.., ct + s] = loc[s] ----
template <typename T, int CH>   // CH > 0: compile-time head width; 0: runtime (<= 32)
__global__ void gma_apply_kernel(const T* __restrict__ qkvp, const T* __restrict__ convv, const T* __restrict__ loc,
                                 const float* __restrict__ ktv, T* __restrict__ out, int n_tok, int heads, int ch_rt, int seg) {
    const int ch = CH > 0 ? CH : ch_rt;
    extern __shared__ float s_ktv[];          // this image's [heads][ch][ch]
    const int ct = heads * ch, c = ct + seg, nacc = heads * ch * ch, b = blockIdx.y;
    for (int i = threadIdx.x; i < nacc; i += blockDim.x) s_ktv[i] = ktv[(size_t)b * nacc + i];
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
        const float* m = s_ktv + h * ch * ch;
#pragma unroll
        for (int j = 0; j < ch; ++j) {
            float a = q[j] * to_f32(convv[t * ct + h * ch + j]);
#pragma unroll
            for (int k = 0; k < ch; ++k) a += q[k] * m[k * ch + j];
            dst[h * ch + j] = from_f32<T>(a);
        }
    }
}

static inline int pw_grid(size_t n) {
    size_t g = (n + kGThreads - 1) / kGThreads;
    if (g < 1) g = 1;
    if (g > 256 * 32) g = 256 * 32;
    return (int)g;
}

}  // namespace rc

using namespace rc;

extern "C" {

int rc_dwconv2d(const void* d_x, int x_stride_c, int x_c0, void* d_y, int y_stride_c, int y_c0, int dtype,
                int batch, int H, int W, int n_ch, int ksize, const float* d_wT, int n_w, const float* d_bias,
                int n_rep, int x_rep_stride, int y_rep_stride, int w_rep_stride, int add_identity, void* stream) {
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
    const size_t total = (size_t)batch * H * W * n_rep * (n_ch / U);
    if (dtype == RC_F32)
        hipLaunchKernelGGL(dwconv2d_kernel<float>, dim3(pw_grid(total)), dim3(kGThreads), 0, as_stream(stream),
                           static_cast<const float*>(d_x), x_stride_c, x_c0, static_cast<float*>(d_y), y_stride_c, y_c0, batch, H, W,
                           n_ch, ksize, d_wT, n_w, d_bias, n_rep, x_rep_stride, y_rep_stride, w_rep_stride, add_identity);
    else
        hipLaunchKernelGGL(dwconv2d_kernel<bf16_t>, dim3(pw_grid(total)), dim3(kGThreads), 0, as_stream(stream),
                           static_cast<const bf16_t*>(d_x), x_stride_c, x_c0, static_cast<bf16_t*>(d_y), y_stride_c, y_c0, batch, H, W,
                           n_ch, ksize, d_wT, n_w, d_bias, n_rep, x_rep_stride, y_rep_stride, w_rep_stride, add_identity);
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

int rc_gma_pointwise(const void* d_qkv, const void* d_dw, const void* d_dwl, void* d_qkvp, void* d_loc, int dtype,
                     long long tokens, int c, const float* d_pw, const float* d_bn_scale, const float* d_bn_shift,
                     const float* d_pwl, const float* d_ln_g, const float* d_ln_b, void* stream) {
    RC_REQUIRE(d_qkv && d_dw && d_dwl && d_qkvp && d_loc && d_pw && d_bn_scale && d_bn_shift && d_pwl && d_ln_g && d_ln_b,
               "rc_gma_pointwise: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_gma_pointwise: bad dtype");
    RC_REQUIRE(tokens >= 1 && c >= 10 && c % 5 == 0 && c / 5 <= 48, "rc_gma_pointwise: C must be a multiple of 5 with C/5 <= 48");
    const int seg = c / 5;
    const size_t lds = (size_t)6 * seg * seg * sizeof(float);
    const size_t total = (size_t)tokens * 13;
#define RC_PW_LAUNCH(TT, SG)                                                                                          \
    hipLaunchKernelGGL((gma_pointwise_kernel<TT, SG>), dim3(pw_grid(total)), dim3(kGThreads), lds, as_stream(stream), \
                       static_cast<const TT*>(d_qkv), static_cast<const TT*>(d_dw), static_cast<const TT*>(d_dwl),    \
                       static_cast<TT*>(d_qkvp), static_cast<TT*>(d_loc), (size_t)tokens, c, seg, d_pw, d_bn_scale,    \
                       d_bn_shift, d_pwl, d_ln_g, d_ln_b)
    if (dtype == RC_F32) {
        if (seg == 16) RC_PW_LAUNCH(float, 16); else if (seg == 40) RC_PW_LAUNCH(float, 40); else RC_PW_LAUNCH(float, 0);
    } else {
        if (seg == 16) RC_PW_LAUNCH(bf16_t, 16); else if (seg == 40) RC_PW_LAUNCH(bf16_t, 40); else RC_PW_LAUNCH(bf16_t, 0);
    }
#undef RC_PW_LAUNCH
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_gma_kv_blocks(int n_tok) {
    int nblk = (n_tok + 4095) / 4096;          // >= 4096 tokens per block, at most 256 blocks per image
    if (nblk > 256) nblk = 256;
    if (nblk < 1) nblk = 1;
    return nblk;
}

size_t rc_gma_kv_scratch_bytes(int batch, int n_tok, int heads, int ch) {
    return (size_t)batch * rc_gma_kv_blocks(n_tok) * (2 * heads * ch + heads * ch * ch) * sizeof(float);
}

int rc_gma_kv(const void* d_qkvp, int dtype, int batch, int n_tok, int heads, int ch, float scale, float* d_scratch,
              float* d_ktv, void* stream) {
    RC_REQUIRE(d_qkvp && d_scratch && d_ktv, "rc_gma_kv: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_gma_kv: bad dtype");
    RC_REQUIRE(batch >= 1 && batch <= 65535 && n_tok >= 1 && heads >= 1 && ch >= 1 && ch <= 32 && heads * ch * ch <= 256 * 16,
               "rc_gma_kv: unsupported head geometry");
    const int ct = heads * ch;
    const int nblk = rc_gma_kv_blocks(n_tok);
    const int L = (n_tok + nblk - 1) / nblk;
    const size_t lds = ((size_t)2 * kKvTile * ct + 3 * ct) * sizeof(float);
    RC_REQUIRE(lds <= 64 * 1024, "rc_gma_kv: too many attention channels");
    if (dtype == RC_F32)
        hipLaunchKernelGGL(gma_kv_reduce_kernel<float>, dim3(nblk, batch), dim3(kGThreads), lds, as_stream(stream),
                           static_cast<const float*>(d_qkvp), d_scratch, n_tok, L, heads, ch);
    else
        hipLaunchKernelGGL(gma_kv_reduce_kernel<bf16_t>, dim3(nblk, batch), dim3(kGThreads), lds, as_stream(stream),
                           static_cast<const bf16_t*>(d_qkvp), d_scratch, n_tok, L, heads, ch);
    hipLaunchKernelGGL(gma_kv_merge_kernel, dim3(batch), dim3(kGThreads), 0, as_stream(stream), d_scratch, d_ktv, nblk, heads, ch, scale);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_gma_apply(const void* d_qkvp, const void* d_convv, const void* d_loc, const float* d_ktv, void* d_out, int dtype,
                 int batch, int n_tok, int heads, int ch, int seg, void* stream) {
    RC_REQUIRE(d_qkvp && d_convv && d_loc && d_ktv && d_out, "rc_gma_apply: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_gma_apply: bad dtype");
    RC_REQUIRE(batch >= 1 && batch <= 65535 && n_tok >= 1 && heads >= 1 && ch >= 1 && ch <= 32 && seg >= 1, "rc_gma_apply: bad shape");
    const size_t lds = (size_t)heads * ch * ch * sizeof(float);
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
