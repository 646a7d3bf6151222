// Small conditioning kernels: CALayer gate, Color_Condition_GFM (global colour prior), GFM vector
// MLPs.  None of them is on the FLOP or byte budget (a few hundred KB per frame); what matters is
// that every reduction uses a fixed order (no float atomics) so results are run-to-run bitwise
// stable, and that they stay on the launch stream without host syncs.
#include "common.hpp"

namespace rc {

// ---- CALayer gate: fixed-order reduction of the conv's per-tile channel sums + 2-layer MLP -------
// Stage 1 (large images): block (k, b) folds tiles [k*L, (k+1)*L) of image b into slot k*L, in place
// (a block only ever writes inside its own range, so there is no cross-block hazard).  Fixed order.
__device__ __forceinline__ void ca_reduce_body(float* __restrict__ sums, int n_tiles, int c, int L, int k, int b, int tid, float* part) {
    float* s = sums + (size_t)b * n_tiles * c;
    const int t0 = k * L, t1 = (t0 + L) < n_tiles ? (t0 + L) : n_tiles;
    for (int c0 = 0; c0 < c; c0 += 256) {
        const int cw = (c - c0) < 256 ? (c - c0) : 256;
        const int nparts = 256 / cw;
        const int ch = tid % cw, pt = tid / cw;
        float acc = 0.f;
        if (pt < nparts)
            for (int t = t0 + pt; t < t1; t += nparts) acc += s[(size_t)t * c + c0 + ch];
        part[tid] = acc;
        __syncthreads();
        if (tid < cw) {
            float tot = 0.f;
            for (int p = 0; p < nparts; ++p) tot += part[p * cw + tid];
            s[(size_t)t0 * c + c0 + tid] = tot;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void ca_reduce_kernel(float* __restrict__ sums, int n_tiles, int c, int L) {
    __shared__ float part[256];
    ca_reduce_body(sums, n_tiles, c, L, blockIdx.x, blockIdx.y, threadIdx.x, part);
}

// Stage 2: sum `n_tiles` slots spaced `tile_stride` tiles apart, then the 2-layer gate MLP.  One block per image on the critical path
// between two convolutions, so it is built for latency: 1024 threads, every thread's few slot loads independent (4 accumulators, fixed order).
constexpr int kGateThreads = 1024;
__global__ __launch_bounds__(kGateThreads) void ca_gate_kernel(const float* __restrict__ sums, int n_tiles, int tile_stride,
                                                               size_t image_stride, int c, int cr,
                                                               float inv_hw, const float* __restrict__ w0,
                                                               const float* __restrict__ b0, const float* __restrict__ w1,
                                                               const float* __restrict__ b1, float* __restrict__ gate) {
    extern __shared__ float sm[];          // [kGateThreads] partials | [c] mean | [cr] hidden
    float* part = sm;
    float* mean = sm + kGateThreads;
    float* hid = mean + c;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* s = sums + (size_t)b * image_stride;
    for (int c0 = 0; c0 < c; c0 += kGateThreads) {
        const int cw = (c - c0) < kGateThreads ? (c - c0) : kGateThreads;   // channels in this pass
        const int nparts = kGateThreads / cw;                                // >= 1
        const int ch = tid % cw, pt = tid / cw;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (pt < nparts) {
            const size_t step = (size_t)tile_stride * c;
            const float* q = s + c0 + ch;
            int t = pt;
            for (; t + 3 * nparts < n_tiles; t += 4 * nparts) {
                const float v0 = q[(size_t)t * step], v1 = q[(size_t)(t + nparts) * step];
                const float v2 = q[(size_t)(t + 2 * nparts) * step], v3 = q[(size_t)(t + 3 * nparts) * step];
                a0 += v0; a1 += v1; a2 += v2; a3 += v3;
            }
            for (; t < n_tiles; t += nparts) a0 += q[(size_t)t * step];
        }
        part[tid] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (tid < cw) {
            float tot = 0.f;
            for (int p = 0; p < nparts; ++p) tot += part[p * cw + tid];
            mean[c0 + tid] = tot * inv_hw;
        }
        __syncthreads();
    }
    for (int j = tid; j < cr; j += kGateThreads) {
        float h = b0[j];
        for (int k = 0; k < c; ++k) h += w0[(size_t)j * c + k] * mean[k];
        hid[j] = h > 0.f ? h : 0.f;
    }
    __syncthreads();
    for (int k = tid; k < c; k += kGateThreads) {
        float z = b1[k];
        for (int j = 0; j < cr; ++j) z += w1[(size_t)k * cr + j] * hid[j];
        gate[(size_t)b * c + k] = 1.f / (1.f + expf(-z));
    }
}

// ---- CALayer gate AHEAD of the convolution that feeds it (rc_ca_gate_ahead) --------------------------------------------------
// RCABlock is x + CA(conv2(t)), t = relu(conv1(x)) (models/networks.py:296-311); CA's gate needs mean_HW(conv2(t)), and the mean of a
// convolution is linear in its input: mean_HW(conv2(t))[o] = b2[o] + 1/HW * sum_{c,tap} W2[o][c][tap] * S_tap[c], where S_tap[c] is the sum of
// t[c] over the pixels tap (dy, dx) reaches inside the image = the total minus the row / column the tap's zero padding cuts off, plus the
// doubly-subtracted corner.  So the gate follows from conv1's channel sums (its epilogue emits them) and t's four border lines -- BEFORE conv2
// runs, whose epilogue can then write x_new = conv2(t) * gate + x directly (rc_conv_desc.out_scale + residual).
constexpr int kEdgeSegs = 8;
// grid (4 * kEdgeSegs, B): fixed-order partial sums of row 0 / row H-1 / column 0 / column W-1 of an NHWC map, and its four corner pixels
template <typename T>
__device__ __forceinline__ void ca_border_body(const T* __restrict__ t, float* __restrict__ edge, float* __restrict__ corner,
                                               int H, int W, int c, int idx, int b, int tid, float* part) {
    const int e = idx / kEdgeSegs, seg = idx % kEdgeSegs;
    const T* img = t + (size_t)b * H * W * c;
    const int n = e < 2 ? W : H, per = (n + kEdgeSegs - 1) / kEdgeSegs;
    const int i0 = seg * per, i1 = (i0 + per) < n ? (i0 + per) : n;
    const size_t base = e == 0 ? 0 : e == 1 ? (size_t)(H - 1) * W * c : e == 2 ? 0 : (size_t)(W - 1) * c;
    const size_t step = e < 2 ? (size_t)c : (size_t)W * c;
    for (int c0 = 0; c0 < c; c0 += 64) {
        const int ch = c0 + (tid & 63), sub = tid >> 6;
        float acc = 0.f;
        if (ch < c) {
#pragma unroll 4
            for (int i = i0 + sub; i < i1; i += 4) acc += to_f32(img[base + (size_t)i * step + ch]);
        }
        part[tid] = acc;
        __syncthreads();
        if (tid < 64 && ch < c) edge[(((size_t)b * 4 + e) * kEdgeSegs + seg) * c + ch] = (part[tid] + part[tid + 64]) + (part[tid + 128] + part[tid + 192]);
        __syncthreads();
    }
    if (idx == 0)
        for (int ch = tid; ch < c; ch += 256) {
            corner[((size_t)b * 4 + 0) * c + ch] = to_f32(img[ch]);
            corner[((size_t)b * 4 + 1) * c + ch] = to_f32(img[(size_t)(W - 1) * c + ch]);
            corner[((size_t)b * 4 + 2) * c + ch] = to_f32(img[(size_t)(H - 1) * W * c + ch]);
            corner[((size_t)b * 4 + 3) * c + ch] = to_f32(img[((size_t)(H - 1) * W + (W - 1)) * c + ch]);
        }
}

// ca_reduce_kernel's slot folding and the border lines in ONE launch (they are independent): blocks [0, n_red) fold, the next 4 * kEdgeSegs take the borders
template <typename T>
__global__ __launch_bounds__(256) void ca_reduce_border_kernel(float* __restrict__ sums, int n_tiles, int c, int L, int n_red, const T* __restrict__ t,
                                                               float* __restrict__ edge, float* __restrict__ corner, int H, int W) {
    __shared__ float part[256];
    if ((int)blockIdx.x < n_red) ca_reduce_body(sums, n_tiles, c, L, blockIdx.x, blockIdx.y, threadIdx.x, part);
    else ca_border_body<T>(t, edge, corner, H, W, c, (int)blockIdx.x - n_red, blockIdx.y, threadIdx.x, part);
}

// sm: [kGateThreads] partials | S[c] | E[4][c] | K[4][c] | mean[c] | hid[cr]; one 1024-thread block per image
__device__ __forceinline__ void ca_gate_ahead_body(const float* __restrict__ sums, int n_tiles, int tile_stride, size_t image_stride,
                                                   const float* __restrict__ edge, const float* __restrict__ corner,
                                                   const float* __restrict__ w2t, const float* __restrict__ b2,
                                                   int c, int cr, float inv_hw, const float* __restrict__ w0,
                                                   const float* __restrict__ b0, const float* __restrict__ w1,
                                                   const float* __restrict__ b1, float* __restrict__ gate, int b, int tid, float* sm) {
    float* part = sm;
    float* S = sm + kGateThreads;
    float* E = S + c;
    float* K = E + 4 * c;
    float* mean = K + 4 * c;
    float* hid = mean + c;
    const float* s = sums + (size_t)b * image_stride;
    for (int c0 = 0; c0 < c; c0 += kGateThreads) {             // total of t per channel: the same fixed-order fold as ca_gate_kernel
        const int cw = (c - c0) < kGateThreads ? (c - c0) : kGateThreads;
        const int nparts = kGateThreads / cw;
        const int ch = tid % cw, pt = tid / cw;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (pt < nparts) {
            const size_t step = (size_t)tile_stride * c;
            const float* q = s + c0 + ch;
            int t = pt;
            for (; t + 3 * nparts < n_tiles; t += 4 * nparts) {
                const float v0 = q[(size_t)t * step], v1 = q[(size_t)(t + nparts) * step];
                const float v2 = q[(size_t)(t + 2 * nparts) * step], v3 = q[(size_t)(t + 3 * nparts) * step];
                a0 += v0; a1 += v1; a2 += v2; a3 += v3;
            }
            for (; t < n_tiles; t += nparts) a0 += q[(size_t)t * step];
        }
        part[tid] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (tid < cw) {
            float tot = 0.f;
            for (int p = 0; p < nparts; ++p) tot += part[p * cw + tid];
            S[c0 + tid] = tot;
        }
        __syncthreads();
    }
    for (int i = tid; i < 4 * c; i += kGateThreads) {          // border lines: fold the segments in order; corners as they are
        const int e = i / c, ch = i - e * c;
        float tot = 0.f;
        for (int g = 0; g < kEdgeSegs; ++g) tot += edge[(((size_t)b * 4 + e) * kEdgeSegs + g) * c + ch];
        E[i] = tot;
        K[i] = corner[(size_t)b * 4 * c + i];
    }
    __syncthreads();
    // mean_HW(conv2(t))[o]: thread (o, pt) takes every nparts-th input channel; fixed-order fold over pt
    for (int o0 = 0; o0 < c; o0 += kGateThreads) {
        const int cw = (c - o0) < kGateThreads ? (c - o0) : kGateThreads;
        const int nparts = kGateThreads / cw;
        const int o = o0 + tid % cw, pt = tid / cw;
        float acc = 0.f;
        if (pt < nparts) {
            for (int cc = pt; cc < c; cc += nparts) {
                const float* wr = w2t + (size_t)cc * 9 * c + o;              // [cin][tap][cout]: the threads of a wave read consecutive couts
                const float tot = S[cc], r0 = E[cc], r1 = E[c + cc], q0 = E[2 * c + cc], q1 = E[3 * c + cc];
                // tap (dy, dx) reads t at (y + dy - 1, x + dx - 1): dy = 2 never reaches row 0, dy = 0 never row H-1 (likewise columns)
                acc += wr[0] * (tot - r1 - q1 + K[3 * c + cc]) + wr[c] * (tot - r1) + wr[2 * c] * (tot - r1 - q0 + K[2 * c + cc]);
                acc += wr[3 * c] * (tot - q1) + wr[4 * c] * tot + wr[5 * c] * (tot - q0);
                acc += wr[6 * c] * (tot - r0 - q1 + K[c + cc]) + wr[7 * c] * (tot - r0) + wr[8 * c] * (tot - r0 - q0 + K[cc]);
            }
        }
        part[tid] = acc;
        __syncthreads();
        if (tid < cw) {
            float tot = 0.f;
            for (int p = 0; p < nparts; ++p) tot += part[p * cw + tid];
            mean[o0 + tid] = (b2 ? b2[o0 + tid] : 0.f) + tot * inv_hw;
        }
        __syncthreads();
    }
    for (int j = tid; j < cr; j += kGateThreads) {
        float h = b0[j];
        for (int k = 0; k < c; ++k) h += w0[(size_t)j * c + k] * mean[k];
        hid[j] = h > 0.f ? h : 0.f;
    }
    __syncthreads();
    for (int k = tid; k < c; k += kGateThreads) {
        float z = b1[k];
        for (int j = 0; j < cr; ++j) z += w1[(size_t)k * cr + j] * hid[j];
        gate[(size_t)b * c + k] = 1.f / (1.f + expf(-z));
    }
}

__global__ __launch_bounds__(kGateThreads) void ca_gate_ahead_kernel(const float* __restrict__ sums, int n_tiles, int tile_stride, size_t image_stride,
                                                                     const float* __restrict__ edge, const float* __restrict__ corner,
                                                                     const float* __restrict__ w2t, const float* __restrict__ b2,
                                                                     int c, int cr, float inv_hw, const float* __restrict__ w0,
                                                                     const float* __restrict__ b0, const float* __restrict__ w1,
                                                                     const float* __restrict__ b1, float* __restrict__ gate) {
    extern __shared__ float sm[];
    ca_gate_ahead_body(sums, n_tiles, tile_stride, image_stride, edge, corner, w2t, b2, c, cr, inv_hw, w0, b0, w1, b1, gate, blockIdx.x, threadIdx.x, sm);
}

// (Round 5 tried rc_ca_gate_ahead as ONE launch: the fold / border blocks publish their results, the block that arrives last at a per-image counter runs
// the gate stage -- bit-identical to the two launches.  With agent-scope release / acquire fences it took 85 us per call at level 0 against 35 us for
// the two launches (every block's release writes back an L2 full of the conv's output); with write-through `sc0 sc1` stores and cache-bypassing loads 52 us;
// the whole step 52.4 vs 51.9 ms.  A dependent launch boundary (~2 us) is cheaper than any cross-CU hand-off inside a launch here.  Removed.)

// ---- color_block: conv1x1 -> avgpool(3, s2, p1, count_include_pad) -> LeakyReLU(0.2) -------------
// x NCHW (B,cin,h,w), optionally instance-normalised on load; y fp32 NCHW (B,cout,ho,wo).
// Both stages are linear, so the pool runs FIRST, on the cin input channels (9x fewer multiply-adds than pooling
// cout conv outputs, and each input value is fetched once per block instead of once per output channel):
//   y = lrelu( W . (sum_valid x) / 9 + bias * n_valid / 9 )       (padded taps contribute 0, bias included)
// A block owns 64 output pixels of one image: the pooled (cin x 64) panel is built in LDS by all 256 threads, then
// thread (pixel, quarter) produces a quarter of the output channels.
constexpr int CB_PX = 64;
template <typename TI>
__global__ __launch_bounds__(256) void color_block_kernel(const TI* __restrict__ x, float* __restrict__ y, int batch, int cin, int cout,
                                                          int h, int w, int ho, int wo, const float* __restrict__ wgt,
                                                          const float* __restrict__ bias, const float* __restrict__ in_mean,
                                                          const float* __restrict__ in_rstd, const float* __restrict__ in_gamma,
                                                          const float* __restrict__ in_beta) {
    extern __shared__ float cb_pooled[];                 // [cin][CB_PX] sums over the valid taps, then [CB_PX] tap counts
    float* s_cnt = cb_pooled + (size_t)cin * CB_PX;
    const int npx = ho * wo;
    const int tiles = (npx + CB_PX - 1) / CB_PX;
    // gridDim.y = cout slices: the late blocks of the prior are a handful of pixel tiles (8 x 8 outputs per image, 128 -> 128), and ONE block per tile left a thread
    // 32 outputs x 128 serial multiply-adds on scalar weight loads (296 us for 8 blocks on 256 CUs); a slice re-pools its tile (cheap) and does cout / slices of them
    const int per_slice = (cout + (int)gridDim.y - 1) / (int)gridDim.y, co_begin = (int)blockIdx.y * per_slice;
    const int co_end = co_begin + per_slice < cout ? co_begin + per_slice : cout;
    const int b = blockIdx.x / tiles, p0 = (blockIdx.x % tiles) * CB_PX;
    const int tid = threadIdx.x;
    for (int i = tid; i < cin * CB_PX; i += 256) {
        const int ci = i / CB_PX, px = i - ci * CB_PX, p = p0 + px;
        float sum = 0.f; int cnt = 0;
        if (p < npx) {
            const int oy = p / wo, ox = p - oy * wo;
            float sc = 1.f, sh = 0.f;
            if (in_mean != nullptr) {                    // InstanceNorm(affine) of the previous block, applied on load
                const size_t k = (size_t)b * cin + ci;
                sc = in_rstd[k] * in_gamma[ci]; sh = in_beta[ci] - in_mean[k] * sc;
            }
            const TI* plane = x + ((size_t)b * cin + ci) * h * w;
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = 2 * oy + dy;
                if (yy < 0 || yy >= h) continue;
                for (int dx = -1; dx <= 1; ++dx) {
                    const int xx = 2 * ox + dx;
                    if (xx < 0 || xx >= w) continue;
                    sum += to_f32(plane[(size_t)yy * w + xx]) * sc + sh;
                    ++cnt;
                }
            }
        }
        cb_pooled[i] = sum;
        if (ci == 0) s_cnt[px] = (float)cnt;
    }
    __syncthreads();
    const int px = tid & (CB_PX - 1), quarter = tid / CB_PX, p = p0 + px;
    if (p >= npx) return;
    const float cnt = s_cnt[px];
    for (int co = co_begin + quarter; co < co_end; co += 256 / CB_PX) {
        const float* wr = wgt + (size_t)co * cin;        // uniform over the wave: scalar loads
        float acc = 0.f;
        for (int ci = 0; ci < cin; ++ci) acc += wr[ci] * cb_pooled[ci * CB_PX + px];
        const float pooled = (acc + bias[co] * cnt) * (1.f / 9.f);
        y[((size_t)b * cout + co) * npx + p] = pooled > 0.f ? pooled : 0.2f * pooled;
    }
}

// ---- InstanceNorm statistics: per (b,c) mean and rstd (biased variance), two-pass, fixed order ----
__global__ __launch_bounds__(256) void instance_stats_kernel(const float* __restrict__ x, float* __restrict__ mean,
                                                             float* __restrict__ rstd, int hw, float eps) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const float* p = x + (size_t)blockIdx.x * hw;
    float s = 0.f;
    for (int i = tid; i < hw; i += 256) s += p[i];
    red[tid] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (tid < k) red[tid] += red[tid + k]; __syncthreads(); }
    const float m = red[0] / (float)hw;
    __syncthreads();
    float v = 0.f;
    for (int i = tid; i < hw; i += 256) { const float d = p[i] - m; v += d * d; }
    red[tid] = v;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (tid < k) red[tid] += red[tid + k]; __syncthreads(); }
    if (tid == 0) {
        mean[blockIdx.x] = m;
        rstd[blockIdx.x] = 1.f / sqrtf(red[0] / (float)hw + eps);
    }
}

// ---- InstanceNorm2d(affine) applied with given statistics: y = (x - mean) * rstd * gamma + beta, fp32 NCHW ----
__global__ __launch_bounds__(256) void instance_norm_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int c, int hw) {
    const int plane = blockIdx.x;                    // b * c + ch
    const float a = rstd[plane] * gamma[plane % c], o = beta[plane % c] - mean[plane] * a;
    const float* p = x + (size_t)plane * hw;
    float* q = y + (size_t)plane * hw;
    for (int i = threadIdx.x; i < hw; i += 256) q[i] = __builtin_fmaf(p[i], a, o);
}

// ---- closing Conv1x1 + AdaptiveAvgPool2d(1): vec[b][o] = mean_p (sum_ci w[o][ci] x[b][ci][p] + b[o]) ----
__global__ __launch_bounds__(256) void color_head_kernel(const float* __restrict__ x, float* __restrict__ vec, int cin,
                                                         int cout, int hw, const float* __restrict__ wgt,
                                                         const float* __restrict__ bias) {
    __shared__ float red[256];
    const int b = blockIdx.x / cout, o = blockIdx.x % cout, tid = threadIdx.x;
    float s = 0.f;
    for (int p = tid; p < hw; p += 256) {
        float acc = bias[o];
        for (int ci = 0; ci < cin; ++ci) acc += wgt[(size_t)o * cin + ci] * x[((size_t)b * cin + ci) * hw + p];
        s += acc;
    }
    red[tid] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (tid < k) red[tid] += red[tid + k]; __syncthreads(); }
    if (tid == 0) vec[(size_t)b * cout + o] = red[0] / (float)hw;
}

// ---- GFM vector MLP: out[b] = W1 * leaky_relu(W0 * vec[b] + b0, 0.1) + b1 ------------------------
__global__ __launch_bounds__(256) void gfm_vector_kernel(const float* __restrict__ vec, int cond_c, int nf, int c,
                                                         const float* __restrict__ w0, const float* __restrict__ b0,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         float* __restrict__ out) {
    extern __shared__ float sm[];  // [cond_c] vec | [nf] hidden
    float* v = sm;
    float* hid = sm + cond_c;
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < cond_c; k += 256) v[k] = vec[(size_t)b * cond_c + k];
    __syncthreads();
    for (int j = tid; j < nf; j += 256) {
        float h = b0[j];
        for (int k = 0; k < cond_c; ++k) h += w0[(size_t)j * cond_c + k] * v[k];
        hid[j] = h > 0.f ? h : 0.1f * h;
    }
    __syncthreads();
    for (int k = tid; k < c; k += 256) {
        float z = b1[k];
        for (int j = 0; j < nf; ++j) z += w1[(size_t)k * nf + j] * hid[j];
        out[(size_t)b * c + k] = z;
    }
}


// ---- per-channel partial sums of an NHWC map (standalone CALayer / AdaptiveAvgPool2d(1), networks.py:259,268) ----
// Block (slot, b) folds pixels [slot*L, (slot+1)*L) of image b: thread = (channel vector v, part p), parts are combined
// through LDS in a fixed order, so the result is run-to-run bitwise stable.  Output (B, n_slots, C) feeds rc_ca_gate.
template <typename T>
__global__ __launch_bounds__(256) void channel_sums_kernel(const T* __restrict__ x, float* __restrict__ sums, int n_pix, int c, int L) {
    constexpr int U = Vec16<T>::N;
    extern __shared__ float cs_part[];                   // [parts][c]
    const int vpp = c / U, parts = 256 / vpp;
    const int slot = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int v = tid % vpp, p = tid / vpp;
    const int p0 = slot * L, p1 = (p0 + L) < n_pix ? (p0 + L) : n_pix;
    float acc[U];
#pragma unroll
    for (int e = 0; e < U; ++e) acc[e] = 0.f;
    if (p < parts) {
        const T* base = x + (size_t)b * n_pix * c + v * U;
        for (int i = p0 + p; i < p1; i += parts) {
            float f[U];
            Vec16<T>::unpack(*reinterpret_cast<const uint4*>(base + (size_t)i * c), f);
#pragma unroll
            for (int e = 0; e < U; ++e) acc[e] += f[e];
        }
#pragma unroll
        for (int e = 0; e < U; ++e) cs_part[p * c + v * U + e] = acc[e];
    }
    __syncthreads();
    for (int ch = tid; ch < c; ch += 256) {
        float tot = 0.f;
        for (int q = 0; q < parts; ++q) tot += cs_part[q * c + ch];
        sums[((size_t)b * gridDim.x + slot) * c + ch] = tot;
    }
}

}  // namespace rc

using namespace rc;

extern "C" {

int rc_channel_sums_slots(int n_pix) {
    int s = (n_pix + 4095) / 4096;          // >= 4096 pixels per slot, at most 256 slots per image
    return s < 1 ? 1 : (s > 256 ? 256 : s);
}

int rc_channel_sums(const void* d_x, int dtype, int batch, int n_pix, int c, float* d_sums, void* stream) {
    RC_REQUIRE(d_x && d_sums, "rc_channel_sums: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_channel_sums: bad dtype");
    const int U = dtype == RC_F32 ? 4 : 8;
    RC_REQUIRE(batch >= 1 && batch <= 65535 && n_pix >= 1 && c >= U && c % U == 0 && c / U <= 256,
               "rc_channel_sums: C must be a multiple of 16 bytes, at most 256 vectors");
    const int slots = rc_channel_sums_slots(n_pix);
    const int L = ceil_div(n_pix, slots);
    const size_t lds = (size_t)(256 / (c / U)) * c * sizeof(float);
    RC_REQUIRE(lds <= 64 * 1024, "rc_channel_sums: too many channels");
    if (dtype == RC_F32)
        hipLaunchKernelGGL(channel_sums_kernel<float>, dim3(slots, batch), dim3(256), lds, as_stream(stream),
                           static_cast<const float*>(d_x), d_sums, n_pix, c, L);
    else
        hipLaunchKernelGGL(channel_sums_kernel<bf16_t>, dim3(slots, batch), dim3(256), lds, as_stream(stream),
                           static_cast<const bf16_t*>(d_x), d_sums, n_pix, c, L);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_ca_gate(float* d_sums, int batch, int n_tiles, int c, int cr, float inv_hw,
               const float* d_w0, const float* d_b0, const float* d_w1, const float* d_b1,
               float* d_gate, void* stream) {
    RC_REQUIRE(d_sums && d_w0 && d_b0 && d_w1 && d_b1 && d_gate, "rc_ca_gate: null pointer");
    RC_REQUIRE(batch >= 1 && n_tiles >= 1 && c >= 1 && cr >= 1, "rc_ca_gate: bad shape");
    const size_t lds = (kGateThreads + (size_t)c + cr) * sizeof(float);
    RC_REQUIRE(lds <= 64 * 1024, "rc_ca_gate: too many channels");
    RC_REQUIRE(batch <= 65535, "rc_ca_gate: batch > 65535");
    int slots = n_tiles, stride = 1;
    if (n_tiles > 128) {  // two-stage: ~256 slices per image folded in place first (short serial chains in both stages)
        const int L = ceil_div(n_tiles, 256);
        slots = ceil_div(n_tiles, L);
        stride = L;
        hipLaunchKernelGGL(ca_reduce_kernel, dim3(slots, batch), dim3(256), 0, as_stream(stream), d_sums, n_tiles, c, L);
    }
    hipLaunchKernelGGL(ca_gate_kernel, dim3(batch), dim3(kGateThreads), lds, as_stream(stream), d_sums, slots, stride,
                       (size_t)n_tiles * c, c, cr, inv_hw, d_w0, d_b0, d_w1, d_b1, d_gate);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

size_t rc_ca_gate_ahead_scratch_floats(int batch, int c) { return (size_t)(batch > 0 ? batch : 0) * (4 * kEdgeSegs + 4) * (c > 0 ? c : 0); }

int rc_ca_gate_ahead(float* d_sums, int batch, int n_tiles, int c, int cr, const void* d_t, int dtype, int H, int W,
                     const float* d_w2t, const float* d_b2, const float* d_w0, const float* d_b0, const float* d_w1, const float* d_b1,
                     float* d_scratch, float* d_gate, void* stream) {
    RC_REQUIRE(d_sums && d_t && d_w2t && d_w0 && d_b0 && d_w1 && d_b1 && d_scratch && d_gate, "rc_ca_gate_ahead: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_ca_gate_ahead: bad dtype");
    RC_REQUIRE(batch >= 1 && batch <= 65535 && n_tiles >= 1 && c >= 1 && cr >= 1 && H >= 1 && W >= 1, "rc_ca_gate_ahead: bad shape");
    const size_t lds = (kGateThreads + 10 * (size_t)c + cr) * sizeof(float);
    RC_REQUIRE(lds <= 64 * 1024, "rc_ca_gate_ahead: too many channels");
    float* edge = d_scratch;
    float* corner = d_scratch + (size_t)batch * 4 * kEdgeSegs * c;
    int slots = n_tiles, stride = 1, L = 1, n_red = 0;
    if (n_tiles > 128) {      // two-stage fold as in rc_ca_gate
        L = ceil_div(n_tiles, 256);
        slots = ceil_div(n_tiles, L);
        stride = L;
        n_red = slots;
    }
    if (dtype == RC_F32)
        hipLaunchKernelGGL(ca_reduce_border_kernel<float>, dim3(n_red + 4 * kEdgeSegs, batch), dim3(256), 0, as_stream(stream), d_sums, n_tiles, c, L, n_red,
                           static_cast<const float*>(d_t), edge, corner, H, W);
    else
        hipLaunchKernelGGL(ca_reduce_border_kernel<bf16_t>, dim3(n_red + 4 * kEdgeSegs, batch), dim3(256), 0, as_stream(stream), d_sums, n_tiles, c, L, n_red,
                           static_cast<const bf16_t*>(d_t), edge, corner, H, W);
    hipLaunchKernelGGL(ca_gate_ahead_kernel, dim3(batch), dim3(kGateThreads), lds, as_stream(stream), d_sums, slots, stride, (size_t)n_tiles * c,
                       edge, corner, d_w2t, d_b2, c, cr, 1.0f / ((float)H * (float)W), d_w0, d_b0, d_w1, d_b1, d_gate);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_color_block(const void* d_x, int x_dtype, float* d_y, int batch, int cin, int cout, int h, int w,
                   const float* d_w, const float* d_b, const float* d_in_mean, const float* d_in_rstd,
                   const float* d_in_gamma, const float* d_in_beta, void* stream) {
    RC_REQUIRE(d_x && d_y && d_w && d_b, "rc_color_block: null pointer");
    RC_REQUIRE(batch >= 1 && cin >= 1 && cout >= 1 && h >= 1 && w >= 1, "rc_color_block: bad shape");
    RC_REQUIRE(x_dtype == RC_F32 || x_dtype == RC_BF16, "rc_color_block: bad dtype");
    if (d_in_mean) RC_REQUIRE(d_in_rstd && d_in_gamma && d_in_beta, "rc_color_block: incomplete InstanceNorm arguments");
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
    const size_t lds = ((size_t)cin + 1) * CB_PX * sizeof(float);
    RC_REQUIRE(lds <= 144 * 1024, "rc_color_block: cin too large");       // cin <= 575 (the LFM colour branch reaches 256)
    static PerDeviceFlag attr;
    if (!attr.test_and_set()) {
        RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&color_block_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
        RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&color_block_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    }
    const size_t g = (size_t)batch * (((size_t)ho * wo + CB_PX - 1) / CB_PX);
    RC_REQUIRE(g < (1ull << 31), "rc_color_block: too many tiles");
    // cout slices (gridDim.y): enough blocks for the chip on the small late maps, never fewer than 4 outputs per thread and slice (same sums, same order: bit-identical)
    int slices = 1;
    while (g * slices < 2 * (size_t)device_cu_count() && cout / (2 * slices) >= 16) slices *= 2;
    if (x_dtype == RC_F32)
        hipLaunchKernelGGL(color_block_kernel<float>, dim3((unsigned)g, (unsigned)slices), dim3(256), lds, as_stream(stream),
                           static_cast<const float*>(d_x), d_y, batch, cin, cout, h, w, ho, wo, d_w, d_b,
                           d_in_mean, d_in_rstd, d_in_gamma, d_in_beta);
    else
        hipLaunchKernelGGL(color_block_kernel<bf16_t>, dim3((unsigned)g, (unsigned)slices), dim3(256), lds, as_stream(stream),
                           static_cast<const bf16_t*>(d_x), d_y, batch, cin, cout, h, w, ho, wo, d_w, d_b,
                           d_in_mean, d_in_rstd, d_in_gamma, d_in_beta);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_instance_stats(const float* d_x, float* d_mean, float* d_rstd, int batch, int c, int hw, float eps, void* stream) {
    RC_REQUIRE(d_x && d_mean && d_rstd, "rc_instance_stats: null pointer");
    RC_REQUIRE(batch >= 1 && c >= 1 && hw >= 1, "rc_instance_stats: bad shape");
    hipLaunchKernelGGL(instance_stats_kernel, dim3(batch * c), dim3(256), 0, as_stream(stream), d_x, d_mean, d_rstd, hw, eps);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_instance_norm(const float* d_x, float* d_y, const float* d_mean, const float* d_rstd, const float* d_gamma, const float* d_beta,
                     int batch, int c, int hw, void* stream) {
    RC_REQUIRE(d_x && d_y && d_mean && d_rstd && d_gamma && d_beta, "rc_instance_norm: null pointer");
    RC_REQUIRE(batch >= 1 && c >= 1 && hw >= 1, "rc_instance_norm: bad shape");
    hipLaunchKernelGGL(instance_norm_kernel, dim3(batch * c), dim3(256), 0, as_stream(stream), d_x, d_y, d_mean, d_rstd, d_gamma, d_beta, c, hw);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_color_head(const float* d_x, float* d_vec, int batch, int cin, int cout, int hw,
                  const float* d_w, const float* d_b, void* stream) {
    RC_REQUIRE(d_x && d_vec && d_w && d_b, "rc_color_head: null pointer");
    RC_REQUIRE(batch >= 1 && cin >= 1 && cout >= 1 && hw >= 1, "rc_color_head: bad shape");
    hipLaunchKernelGGL(color_head_kernel, dim3(batch * cout), dim3(256), 0, as_stream(stream), d_x, d_vec, cin, cout, hw, d_w, d_b);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_gfm_vector(const float* d_vec, int batch, int cond_c, int nf, int c, const float* d_w0, const float* d_b0,
                  const float* d_w1, const float* d_b1, float* d_out, void* stream) {
    RC_REQUIRE(d_vec && d_w0 && d_b0 && d_w1 && d_b1 && d_out, "rc_gfm_vector: null pointer");
    RC_REQUIRE(batch >= 1 && cond_c >= 1 && nf >= 1 && c >= 1, "rc_gfm_vector: bad shape");
    const size_t lds = ((size_t)cond_c + nf) * sizeof(float);
    RC_REQUIRE(lds <= 64 * 1024, "rc_gfm_vector: vector too long");
    hipLaunchKernelGGL(gfm_vector_kernel, dim3(batch), dim3(256), lds, as_stream(stream), d_vec, cond_c, nf, c, d_w0, d_b0,
                       d_w1, d_b1, d_out);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

}  // extern "C"
