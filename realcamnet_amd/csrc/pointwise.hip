// HBM-bound kernels of the RAW->sRGB path: Bayer unshuffle, layout plumbing, Haar DWT/IDWT,
// CA gate application.  All are pure streaming passes; the design rule is 16-byte accesses per lane
// on the NHWC side and row-contiguous (coalesced along x) accesses on the planar side.
#include "common.hpp"

namespace rc {

constexpr int kPwThreads = 256;

// One-shot grids: this part streams fastest when every thread handles one 16-byte item and the hardware scheduler orders the blocks
// (tools/hbm_probe.py: copy 5.9 TB/s one-shot vs 4.8-5.0 TB/s from a 4096-block grid-stride loop); the loops remain for sizes past the cap.
static inline int grid_for(size_t n_items, int cap = 1 << 22) {
    size_t g = (n_items + kPwThreads - 1) / kPwThreads;
    if (g < 1) g = 1;
    if (g > (size_t)cap) g = cap;
    return (int)g;
}

// ---- Bayer unshuffle + pad: mosaic (B,2h,2w) -> NHWC (B,hp,wp,4) ---------------------------------
// One thread per packed pixel: reads two 2-element row segments (8 B fp32 / 4 B bf16, neighbours
// coalesce into full lines), writes one 4-channel pixel (16 B fp32 / 8 B bf16).
template <typename TI, typename TO>
__global__ void bayer_unshuffle_kernel(const TI* __restrict__ mosaic, TO* __restrict__ packed,
                                       int batch, int h, int w, int hp, int wp) {
    const size_t total = (size_t)batch * hp * wp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % wp);
        const int y = (int)((i / wp) % hp);
        const int b = (int)(i / ((size_t)wp * hp));
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (y < h && x < w) {
            const TI* r0 = mosaic + ((size_t)b * 2 * h + 2 * y) * (2 * (size_t)w) + 2 * x;
            const TI* r1 = r0 + 2 * (size_t)w;
            v[0] = to_f32(r0[0]); v[1] = to_f32(r0[1]); v[2] = to_f32(r1[0]); v[3] = to_f32(r1[1]);
        }
        TO* o = packed + i * 4;
        if constexpr (sizeof(TO) == 4) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            uint2 p;
            p.x = Vec16<bf16_t>::rne2(v[0], v[1]);
            p.y = Vec16<bf16_t>::rne2(v[2], v[3]);
            *reinterpret_cast<uint2*>(o) = p;
        }
    }
}

// ---- RAW ingest: black/white-level normalisation + Bayer unshuffle + pad, and the bilinear "Resize" of the packed RAW
// to the colour-prior's cond image, one launch (the two boxes in front of the net in
// assets/networkarch.png).  Items [0, B*hp*wp): one packed pixel each (as bayer_unshuffle_kernel);
// items [B*hp*wp, + B*ch*cw): one cond pixel each, all 4 planes: F.interpolate(mode='bilinear', align_corners=False)
// semantics (src = scale*(dst+0.5)-0.5 clamped at 0, second tap clamped to the last row/column), read straight from
// the mosaic (packed channel 2i+j at (y,x) = mosaic (2y+i, 2x+j)) and normalised before interpolation.
template <typename TI, typename TO>
__global__ void raw_ingest_kernel(const TI* __restrict__ mosaic, TO* __restrict__ packed, TO* __restrict__ cond,
                                  int batch, int h, int w, int hp, int wp, int ch, int cw, float black, float inv_range,
                                  float scale_y, float scale_x) {
    const size_t n_packed = (size_t)batch * hp * wp, n_cond = (size_t)batch * ch * cw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_packed + n_cond; i += (size_t)gridDim.x * blockDim.x) {
        if (i < n_packed) {
            const int x = (int)(i % wp);
            const int y = (int)((i / wp) % hp);
            const int b = (int)(i / ((size_t)wp * hp));
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (y < h && x < w) {
                const TI* r0 = mosaic + ((size_t)b * 2 * h + 2 * y) * (2 * (size_t)w) + 2 * x;
                const TI* r1 = r0 + 2 * (size_t)w;
                v[0] = (to_f32(r0[0]) - black) * inv_range; v[1] = (to_f32(r0[1]) - black) * inv_range;
                v[2] = (to_f32(r1[0]) - black) * inv_range; v[3] = (to_f32(r1[1]) - black) * inv_range;
            }
            TO* o = packed + i * 4;
            if constexpr (sizeof(TO) == 4) {
                *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                uint2 p;
                p.x = Vec16<bf16_t>::rne2(v[0], v[1]);
                p.y = Vec16<bf16_t>::rne2(v[2], v[3]);
                *reinterpret_cast<uint2*>(o) = p;
            }
        } else {
            const size_t j = i - n_packed;
            const int ox = (int)(j % cw);
            const int oy = (int)((j / cw) % ch);
            const int b = (int)(j / ((size_t)cw * ch));
            float sy = scale_y * ((float)oy + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
            float sx = scale_x * ((float)ox + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
            int y0 = (int)sy, x0 = (int)sx;
            y0 = y0 < h - 1 ? y0 : h - 1; x0 = x0 < w - 1 ? x0 : w - 1;
            const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
            const float ly1 = sy - (float)y0, ly0 = 1.f - ly1, lx1 = sx - (float)x0, lx0 = 1.f - lx1;
            const TI* img = mosaic + (size_t)b * 4 * h * w;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int di = k >> 1, dj = k & 1;
                const float v00 = (to_f32(img[(size_t)(2 * y0 + di) * 2 * w + 2 * x0 + dj]) - black) * inv_range;
                const float v01 = (to_f32(img[(size_t)(2 * y0 + di) * 2 * w + 2 * x1 + dj]) - black) * inv_range;
                const float v10 = (to_f32(img[(size_t)(2 * y1 + di) * 2 * w + 2 * x0 + dj]) - black) * inv_range;
                const float v11 = (to_f32(img[(size_t)(2 * y1 + di) * 2 * w + 2 * x1 + dj]) - black) * inv_range;
                const float r = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
                cond[(((size_t)b * 4 + k) * ch + oy) * cw + ox] = from_f32<TO>(r);
            }
        }
    }
}

// ---- NCHW -> NHWC for C <= 4 (the coordinate map, packed RAW): one thread per output pixel, plane reads and the
// pixel-record writes are both coalesced; the 32x32 transpose tile below would launch a block per 32 pixels
template <typename TI, typename TO>
__global__ void nchw_to_nhwc_small_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int batch, int c, int h, int w,
                                          int hp, int wp) {
    const size_t total = (size_t)batch * hp * wp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % wp), y = (int)((i / wp) % hp), b = (int)(i / ((size_t)wp * hp));
        const bool in = y < h && x < w;
        for (int ch = 0; ch < c; ++ch)
            dst[i * c + ch] = from_f32<TO>(in ? to_f32(src[(((size_t)b * c + ch) * h + y) * w + x]) : 0.f);
    }
}

// ---- NCHW <-> NHWC through an LDS transpose tile (32 pixels x 32 channels) -------------------------
template <typename TI, typename TO>
__global__ void nchw_to_nhwc_kernel(const TI* __restrict__ src, TO* __restrict__ dst,
                                    int c, int h, int w, int hp, int wp) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int row = blockIdx.y;                 // y in [0,hp)
    const int xblk = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, tyy = threadIdx.x >> 5;  // 256 threads: 32 x 8
    for (int c0 = 0; c0 < c; c0 += 32) {
        for (int k = tyy; k < 32; k += 8) {     // k = channel within block; tx = pixel
            const int ch = c0 + k, x = xblk + tx;
            float v = 0.f;
            if (ch < c && row < h && x < w) v = to_f32(src[(((size_t)b * c + ch) * h + row) * w + x]);
            tile[k][tx] = v;
        }
        __syncthreads();
        for (int k = tyy; k < 32; k += 8) {     // k = pixel; tx = channel
            const int ch = c0 + tx, x = xblk + k;
            if (ch < c && x < wp) dst[(((size_t)b * hp + row) * wp + x) * c + ch] = from_f32<TO>(tile[tx][k]);
        }
        __syncthreads();
    }
}

template <typename TI, typename TO>
__global__ void nhwc_to_nchw_kernel(const TI* __restrict__ src, TO* __restrict__ dst,
                                    int c, int H, int W, int h, int w) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int row = blockIdx.y;                 // y in [0,h)
    const int xblk = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, tyy = threadIdx.x >> 5;
    for (int c0 = 0; c0 < c; c0 += 32) {
        for (int k = tyy; k < 32; k += 8) {     // k = pixel; tx = channel
            const int ch = c0 + tx, x = xblk + k;
            float v = 0.f;
            if (ch < c && x < w) v = to_f32(src[(((size_t)b * H + row) * W + x) * c + ch]);
            tile[k][tx] = v;
        }
        __syncthreads();
        for (int k = tyy; k < 32; k += 8) {     // k = channel; tx = pixel
            const int ch = c0 + k, x = xblk + tx;
            if (ch < c && x < w) dst[(((size_t)b * c + ch) * h + row) * w + x] = from_f32<TO>(tile[tx][k]);
        }
        __syncthreads();
    }
}

// ---- y = r*gate[b][c] + x   (x == NULL: y = r*gate, the standalone CALayer scale) --------------------
template <typename T>
__global__ void gate_residual_kernel(const T* __restrict__ r, const float* __restrict__ gate,
                                     const T* __restrict__ x, T* __restrict__ y,
                                     int batch, size_t n_pix, int c) {
    constexpr int U = Vec16<T>::N;
    const int vpp = c / U;                      // vectors per pixel
    const size_t total = (size_t)batch * n_pix * vpp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vpp);
        const int b = (int)(i / ((size_t)vpp * n_pix));
        float fr[U], fx[U];
        Vec16<T>::unpack(reinterpret_cast<const uint4*>(r)[i], fr);
        if (x) {
            Vec16<T>::unpack(reinterpret_cast<const uint4*>(x)[i], fx);
        } else {
#pragma unroll
            for (int e = 0; e < U; ++e) fx[e] = 0.f;
        }
        const float* g = gate + (size_t)b * c + v * U;
#pragma unroll
        for (int e = 0; e < U; ++e) fr[e] = fr[e] * g[e] + fx[e];
        reinterpret_cast<uint4*>(y)[i] = Vec16<T>::pack(fr);
    }
}

// ---- y = x*scale[b][c] + shift[b][c] + x : GFMLayer on a feature map (models/LiteISP.py:308-321) -----------------------------
template <typename T>
__global__ void film_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                  T* __restrict__ y, int batch, size_t n_pix, int c) {
    constexpr int U = Vec16<T>::N;
    const int vpp = c / U;
    const size_t total = (size_t)batch * n_pix * vpp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vpp);
        const int b = (int)(i / ((size_t)vpp * n_pix));
        float fx[U];
        Vec16<T>::unpack(reinterpret_cast<const uint4*>(x)[i], fx);
        const float* s = scale + (size_t)b * c + v * U;
        const float* t = shift + (size_t)b * c + v * U;
#pragma unroll
        for (int e = 0; e < U; ++e) fx[e] = (fx[e] * s[e] + t[e]) + fx[e];
        reinterpret_cast<uint4*>(y)[i] = Vec16<T>::pack(fx);
    }
}

// ---- y = a * sigmoid(b) + identity ----------------------------------------------------------------------
template <typename T>
__global__ void sigmoid_gate_add_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ idn, T* __restrict__ y,
                                        size_t total) {
    constexpr int U = Vec16<T>::N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float fa[U], fb[U], fi[U];
        Vec16<T>::unpack(reinterpret_cast<const uint4*>(a)[i], fa);
        Vec16<T>::unpack(reinterpret_cast<const uint4*>(b)[i], fb);
        Vec16<T>::unpack(reinterpret_cast<const uint4*>(idn)[i], fi);
#pragma unroll
        for (int e = 0; e < U; ++e) fa[e] = fa[e] * (1.f / (1.f + expf(-fb[e]))) + fi[e];
        reinterpret_cast<uint4*>(y)[i] = Vec16<T>::pack(fa);
    }
}

// ---- dst[b][y][x] = src[b][2y][2x] ----------------------------------------------------------------------
template <typename T>
__global__ void subsample2_kernel(const T* __restrict__ src, T* __restrict__ dst, int batch, int H, int W, int c) {
    constexpr int U = Vec16<T>::N;
    const int vpp = c / U, oh = (H + 1) / 2, ow = (W + 1) / 2;
    const size_t total = (size_t)batch * oh * ow * vpp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vpp);
        const size_t p = i / vpp;
        const int x = (int)(p % ow), y = (int)((p / ow) % oh), b = (int)(p / ((size_t)ow * oh));
        reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[(((size_t)b * H + 2 * y) * W + 2 * x) * vpp + v];
    }
}

// ---- bilinear x2 upsampling, align_corners=True (torch's area_pixel_compute_source_index with align_corners) ------------
template <typename T>
__global__ void upsample_bilinear2_kernel(const T* __restrict__ src, T* __restrict__ dst, int batch, int H, int W, int c) {
    constexpr int U = Vec16<T>::N;
    const int vpp = c / U, oh = 2 * H, ow = 2 * W;
    const float ry = oh > 1 ? (float)(H - 1) / (float)(oh - 1) : 0.f, rx = ow > 1 ? (float)(W - 1) / (float)(ow - 1) : 0.f;
    const size_t total = (size_t)batch * oh * ow * vpp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vpp);
        size_t p = i / vpp;
        const int x = (int)(p % ow); p /= ow;
        const int y = (int)(p % oh);
        const int b = (int)(p / oh);
        const float fy = ry * (float)y, fx = rx * (float)x;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const uint4* s4 = reinterpret_cast<const uint4*>(src) + (size_t)b * H * W * vpp + v;
        float a[U], bq[U], cq[U], d[U], o[U];
        Vec16<T>::unpack(s4[((size_t)y0 * W + x0) * vpp], a);
        Vec16<T>::unpack(s4[((size_t)y0 * W + x1) * vpp], bq);
        Vec16<T>::unpack(s4[((size_t)y1 * W + x0) * vpp], cq);
        Vec16<T>::unpack(s4[((size_t)y1 * W + x1) * vpp], d);
#pragma unroll
        for (int e = 0; e < U; ++e) o[e] = hy * (hx * a[e] + lx * bq[e]) + ly * (hx * cq[e] + lx * d[e]);
        reinterpret_cast<uint4*>(dst)[i] = Vec16<T>::pack(o);
    }
}

// ---- y = x*scale + shift + x (+ identity) ----------------------------------------------------------------
template <typename T>
__global__ void sft_apply_kernel(const T* __restrict__ x, const T* __restrict__ scale, const T* __restrict__ shift, const T* __restrict__ idn,
                                 T* __restrict__ y, size_t total) {
    constexpr int U = Vec16<T>::N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float fx[U], fs[U], ft[U];
        Vec16<T>::unpack(reinterpret_cast<const uint4*>(x)[i], fx);
        Vec16<T>::unpack(reinterpret_cast<const uint4*>(scale)[i], fs);
        Vec16<T>::unpack(reinterpret_cast<const uint4*>(shift)[i], ft);
#pragma unroll
        for (int e = 0; e < U; ++e) fs[e] = (fx[e] * fs[e] + ft[e]) + fx[e];
        if (idn != nullptr) {
            Vec16<T>::unpack(reinterpret_cast<const uint4*>(idn)[i], ft);
#pragma unroll
            for (int e = 0; e < U; ++e) fs[e] += ft[e];
        }
        reinterpret_cast<uint4*>(y)[i] = Vec16<T>::pack(fs);
    }
}

// ---- space-to-depth by 2, element-wise: dst[y][x][(2i+j)*c + k] = src[2y+i][2x+j][k] ----------------------
template <typename T>
__global__ void space_to_depth2_kernel(const T* __restrict__ src, T* __restrict__ dst, int batch, int H, int W, int c) {
    const int oh = (H + 1) / 2, ow = (W + 1) / 2;
    const size_t total = (size_t)batch * oh * ow * 4 * c;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % c);
        size_t p = i / c;
        const int ph = (int)(p % 4); p /= 4;
        const int x = (int)(p % ow); p /= ow;
        const int y = (int)(p % oh);
        const int b = (int)(p / oh);
        const int sy = 2 * y + (ph >> 1), sx = 2 * x + (ph & 1);
        dst[i] = (sy < H && sx < W) ? src[(((size_t)b * H + sy) * W + sx) * c + k] : from_f32<T>(0.f);
    }
}

template <typename T>     // the same in 16-byte vectors, c % (16 / sizeof(T)) == 0
__global__ void space_to_depth2_vec_kernel(const T* __restrict__ src, T* __restrict__ dst, int batch, int H, int W, int c) {
    constexpr int U = Vec16<T>::N;
    const int vpp = c / U, oh = (H + 1) / 2, ow = (W + 1) / 2;
    const size_t total = (size_t)batch * oh * ow * 4 * vpp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vpp);
        size_t p = i / vpp;
        const int ph = (int)(p % 4); p /= 4;
        const int x = (int)(p % ow); p /= ow;
        const int y = (int)(p % oh);
        const int b = (int)(p / oh);
        const int sy = 2 * y + (ph >> 1), sx = 2 * x + (ph & 1);
        uint4 val = make_uint4(0, 0, 0, 0);
        if (sy < H && sx < W) val = reinterpret_cast<const uint4*>(src)[(((size_t)b * H + sy) * W + sx) * vpp + v];
        reinterpret_cast<uint4*>(dst)[i] = val;
    }
}

// ---- nn.PixelShuffle(2), element-wise (narrow maps) ------------------------------------------------------
template <typename T>
__global__ void pixel_shuffle2_kernel(const T* __restrict__ src, T* __restrict__ dst, int batch, int H, int W, int c) {
    const size_t total = (size_t)batch * 2 * H * 2 * W * c;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % c);
        size_t p = i / c;
        const int ox = (int)(p % (2 * W)); p /= (2 * W);
        const int oy = (int)(p % (2 * H));
        const int b = (int)(p / (2 * H));
        dst[i] = src[(((size_t)b * H + (oy >> 1)) * W + (ox >> 1)) * (4 * c) + 4 * k + 2 * (oy & 1) + (ox & 1)];
    }
}

// nn.PixelShuffle(2) straight into NCHW (the codecs' x_hat: subpel_conv3x3(2N, 3, 2) at the end of g_s, models/tcm.py:364): one thread per
// output element in NCHW order, so stores are contiguous; the 4c-channel source pixel is re-read from cache by its 4c consumers.
template <typename T>
__global__ void pixel_shuffle2_nchw_kernel(const T* __restrict__ src, T* __restrict__ dst, int batch, int H, int W, int c) {
    const size_t total = (size_t)batch * c * 2 * H * 2 * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t p = i;
        const int ox = (int)(p % (2 * W)); p /= (2 * W);
        const int oy = (int)(p % (2 * H)); p /= (2 * H);
        const int k = (int)(p % c);
        const int b = (int)(p / c);
        dst[i] = src[(((size_t)b * H + (oy >> 1)) * W + (ox >> 1)) * (4 * c) + 4 * k + 2 * (oy & 1) + (ox & 1)];
    }
}

// ---- GDN pieces: y = x*x;  y = x * rsqrt(norm) | x * sqrt(norm)  (+ identity) ----------------------------------
template <typename T>
__global__ void square_kernel(const T* __restrict__ x, T* __restrict__ y, size_t total) {
    constexpr int U = Vec16<T>::N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float f[U];
        Vec16<T>::unpack(reinterpret_cast<const uint4*>(x)[i], f);
#pragma unroll
        for (int e = 0; e < U; ++e) f[e] = f[e] * f[e];
        reinterpret_cast<uint4*>(y)[i] = Vec16<T>::pack(f);
    }
}
template <typename T, bool INVERSE>
__global__ void gdn_apply_kernel(const T* __restrict__ x, const T* __restrict__ norm, const T* __restrict__ idn, T* __restrict__ y,
                                 size_t total) {
    constexpr int U = Vec16<T>::N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float fx[U], fn[U];
        Vec16<T>::unpack(reinterpret_cast<const uint4*>(x)[i], fx);
        Vec16<T>::unpack(reinterpret_cast<const uint4*>(norm)[i], fn);
#pragma unroll
        for (int e = 0; e < U; ++e) fx[e] = fx[e] * (INVERSE ? sqrtf(fn[e]) : 1.f / sqrtf(fn[e]));
        if (idn != nullptr) {
            float fi[U];
            Vec16<T>::unpack(reinterpret_cast<const uint4*>(idn)[i], fi);
#pragma unroll
            for (int e = 0; e < U; ++e) fx[e] += fi[e];
        }
        reinterpret_cast<uint4*>(y)[i] = Vec16<T>::pack(fx);
    }
}

// ---- dst[p, dst_c0 + c] = src[p, src_c0 + c], c < n_ch: channel split / concat of NHWC tensors in 16-byte vectors ------
template <typename T>
__global__ void channel_copy_kernel(const T* __restrict__ src, int src_stride, int src_c0, T* __restrict__ dst, int dst_stride,
                                    int dst_c0, int n_ch, size_t pixels) {
    constexpr int U = Vec16<T>::N;
    const int vpp = n_ch / U;
    const size_t total = pixels * vpp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / vpp;
        const int v = (int)(i - p * vpp);
        *reinterpret_cast<uint4*>(dst + p * dst_stride + dst_c0 + v * U) = *reinterpret_cast<const uint4*>(src + p * src_stride + src_c0 + v * U);
    }
}

// ---- torch.cat along the channel dim of up to 8 NHWC tensors in ONE launch (the slice loop concatenates 2 .. 6 maps, 22 times per forward) ----
struct ConcatArgs { const uint4* src[8]; int vec[8]; int vec0[8]; int n; int dst_vec; };
__global__ void channel_concat_kernel(ConcatArgs a, uint4* __restrict__ dst, size_t pixels) {
    const size_t total = pixels * a.dst_vec;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / a.dst_vec;
        const int v = (int)(i - p * a.dst_vec);
        int k = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) k += (j < a.n && v >= a.vec0[j]) ? 1 : 0;
        dst[i] = a.src[k][p * a.vec[k] + (v - a.vec0[k])];
    }
}

// ---- Haar DWT as the reference's frozen grouped conv (taps read from the state_dict tensor) ------
// forward: x (B,H,W,C) -> y (B,H/2,W/2,4C): y[.., 4c+k] = sum_{i,j} taps[4c+k][i][j] * x[2y+i][2x+j][c]
// One thread per (output pixel, 16-byte group of input channels).
template <typename T, bool UNIFORM>
__global__ void dwt_forward_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ taps,
                                   int batch, int H, int W, int c) {
    constexpr int U = Vec16<T>::N;
    float ut[16];
    if constexpr (UNIFORM) {
#pragma unroll
        for (int k = 0; k < 16; ++k) ut[k] = taps[k];   // wave-uniform -> scalar registers
    }
    const int h = H / 2, w = W / 2, vpp = c / U;
    const size_t total = (size_t)batch * h * w * vpp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vpp);
        const size_t p = i / vpp;
        const int ox = (int)(p % w), oy = (int)((p / w) % h), b = (int)(p / ((size_t)w * h));
        float in[4][U];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const size_t src = (((size_t)b * H + 2 * oy + (t >> 1)) * W + 2 * ox + (t & 1)) * c + v * U;
            Vec16<T>::unpack(*reinterpret_cast<const uint4*>(x + src), in[t]);
        }
        float out[4 * U];
#pragma unroll
        for (int e = 0; e < U; ++e) {
            const float* tp = UNIFORM ? ut : taps + (size_t)(4 * (v * U + e)) * 4;   // (4C,1,2,2): 4 floats per out channel
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float s = in[0][e] * tp[k * 4 + 0];
                s += in[1][e] * tp[k * 4 + 1];
                s += in[2][e] * tp[k * 4 + 2];
                s += in[3][e] * tp[k * 4 + 3];
                out[4 * e + k] = s;
            }
        }
        T* dst = y + p * (4 * (size_t)c) + 4 * v * U;
#pragma unroll
        for (int j = 0; j < 4; ++j) reinterpret_cast<uint4*>(dst)[j] = Vec16<T>::pack(out + j * U);
    }
}

// inverse: x (B,h,w,4C) -> y (B,2h,2w,C): y[2y+i][2x+j][c] = sum_k taps[4c+k][i][j] * x[y][x][4c+k]
// One thread per (input pixel, 16-byte group of OUTPUT channels) = 4 input vectors -> 4 output vectors.
template <typename T, bool UNIFORM>
__global__ __launch_bounds__(kPwThreads) void dwt_inverse_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ taps,
                                   int batch, int h, int w, int c4) {
    constexpr int U = Vec16<T>::N;
    float ut[16];
    if constexpr (UNIFORM) {
#pragma unroll
        for (int k = 0; k < 16; ++k) ut[k] = taps[k];
    }
    const int c = c4 / 4, vpp = c / U;
    const size_t total = (size_t)batch * h * w * vpp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int v = (int)(i % vpp);
        const size_t p = i / vpp;
        const int ix = (int)(p % w), iy = (int)((p / w) % h), b = (int)(p / ((size_t)w * h));
        float in[4 * U];
        const T* src = x + p * (size_t)c4 + 4 * v * U;
#pragma unroll
        for (int j = 0; j < 4; ++j) Vec16<T>::unpack(reinterpret_cast<const uint4*>(src)[j], in + j * U);
        float out[4][U];
#pragma unroll
        for (int e = 0; e < U; ++e) {
            const float* tp = UNIFORM ? ut : taps + (size_t)(4 * (v * U + e)) * 4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float s = in[4 * e + 0] * tp[0 * 4 + t];
                s += in[4 * e + 1] * tp[1 * 4 + t];
                s += in[4 * e + 2] * tp[2 * 4 + t];
                s += in[4 * e + 3] * tp[3 * 4 + t];
                out[t][e] = s;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const size_t dst = (((size_t)b * 2 * h + 2 * iy + (t >> 1)) * (2 * (size_t)w) + 2 * ix + (t & 1)) * c + v * U;
            *reinterpret_cast<uint4*>(y + dst) = Vec16<T>::pack(out[t]);
        }
    }
}

// ---- folded tail, border ring (rc_tail_ring_gather / rc_tail_ring_scatter) -----------------------------------------
// The tail conv 3x3 C -> 4C, PixelShuffle(2), conv 3x3 C -> 3 (models/LiteISP.py:1996-2000) has no activation in between and runs as ONE 5x5
// convolution (rc_tail_fold_weights).  The two differ on the outermost ring of output pixels only: the second conv zero-pads the SHUFFLED map,
// the composition sees conv1 evaluated outside the image there.  The ring is recomputed by the two original convolutions on four thin strips of
// the input (2 rows / 2 columns: output row 0 of the top strip, row 3 of the bottom strip, column 0 / 3 of the side strips are exact).
//   gather : x (B,H,W,C) -> rows (2B,2,W,C) = [x[:, 0:2], x[:, H-2:H]],  cols (2B,2,H,C) = [x[:, :, 0:2], x[:, :, W-2:W]] TRANSPOSED (pixel (r, y) =
//            x[b, y, r]): a 2-pixel-wide image wastes 15/16 of every 8 x 32 conv tile, so the side strips run as 2 x H images through the same two
//            convolutions with ky <-> kx swapped (and conv1's sub-pixel order 2i + j <-> 2j + i)
//   scatter: strip results rows_out (2B,Co,4,2W), cols_out (2B,Co,4,2H) (transposed) -> ring of out (B,Co,out_h,out_w) (the parts inside the crop)
__global__ void tail_ring_gather_kernel(const uint4* __restrict__ x, uint4* __restrict__ rows, uint4* __restrict__ cols,
                                        int batch, int H, int W, int pix16) {
    const size_t row16 = (size_t)W * pix16, n_rows = (size_t)2 * batch * 2 * row16, n_cols = (size_t)2 * batch * H * 2 * pix16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows + n_cols; i += (size_t)gridDim.x * blockDim.x) {
        if (i < n_rows) {
            const size_t off = i % row16; size_t p = i / row16;
            const int r = (int)(p % 2); const int img = (int)(p / 2);
            const int b = img % batch, y = img < batch ? r : H - 2 + r;
            rows[i] = x[((size_t)b * H + y) * row16 + off];
        } else {
            size_t p = i - n_rows;
            const int part = (int)(p % pix16); p /= pix16;
            const int y = (int)(p % H); p /= H;
            const int cx = (int)(p % 2); const int img = (int)(p / 2);
            const int b = img % batch, xx = img < batch ? cx : W - 2 + cx;
            cols[i - n_rows] = x[(((size_t)b * H + y) * W + xx) * pix16 + part];
        }
    }
}
template <typename T>
__global__ void tail_ring_scatter_kernel(const T* __restrict__ rows_out, const T* __restrict__ cols_out, T* __restrict__ out,
                                         int batch, int co, int H, int W, int out_h, int out_w) {
    const int per = 2 * out_w + 2 * out_h;
    const size_t total = (size_t)batch * co * per;
    const bool bottom = out_h == 2 * H, right = out_w == 2 * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int k = (int)(i % per); const size_t pc = i / per;
        const int c = (int)(pc % co), b = (int)(pc / co);
        T* o = out + ((size_t)b * co + c) * out_h * out_w;
        if (k < 2 * out_w) {
            const int bot = k >= out_w, X = bot ? k - out_w : k;
            if (bot && !bottom) continue;
            const T* src = rows_out + (((size_t)(bot ? batch + b : b) * co + c) * 4 + (bot ? 3 : 0)) * (2 * W);
            o[(size_t)(bot ? out_h - 1 : 0) * out_w + X] = src[X];
        } else {
            k -= 2 * out_w;
            const int rgt = k >= out_h, Y = rgt ? k - out_h : k;
            if (rgt && !right) continue;
            if (Y == 0 || (bottom && Y == out_h - 1)) continue;      // corners belong to the row strips: the transposed strips sum their taps in another
                                                                     // order, and two writers of one pixel would make the result depend on a race
            const T* src = cols_out + (((size_t)(rgt ? batch + b : b) * co + c) * 4 + (rgt ? 3 : 0)) * (2 * H);     // transposed strip: row = output column
            o[(size_t)Y * out_w + (rgt ? out_w - 1 : 0)] = src[Y];
        }
    }
}

}  // namespace rc

using namespace rc;

#define RC_DISPATCH_2(in_dt, out_dt, CALL)                                              \
    do {                                                                                \
        if (in_dt == RC_F32 && out_dt == RC_F32) { CALL(float, float); }                \
        else if (in_dt == RC_F32 && out_dt == RC_BF16) { CALL(float, bf16_t); }         \
        else if (in_dt == RC_BF16 && out_dt == RC_F32) { CALL(bf16_t, float); }         \
        else if (in_dt == RC_BF16 && out_dt == RC_BF16) { CALL(bf16_t, bf16_t); }       \
        else return fail(RC_ERR_INVALID, "bad dtype");                                  \
    } while (0)

extern "C" {

int rc_bayer_unshuffle(const void* d_mosaic, int in_dtype, void* d_packed, int out_dtype,
                       int batch, int h, int w, int hp, int wp, void* stream) {
    RC_REQUIRE(d_mosaic && d_packed, "rc_bayer_unshuffle: null pointer");
    RC_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && hp >= h && wp >= w, "rc_bayer_unshuffle: bad shape");
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_packed) % 16 == 0, "rc_bayer_unshuffle: packed must be 16-byte aligned");
    const size_t total = (size_t)batch * hp * wp;
#define CALL(TI, TO)                                                                                           \
    hipLaunchKernelGGL((bayer_unshuffle_kernel<TI, TO>), dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream), \
                       static_cast<const TI*>(d_mosaic), static_cast<TO*>(d_packed), batch, h, w, hp, wp)
    RC_DISPATCH_2(in_dtype, out_dtype, CALL);
#undef CALL
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_raw_ingest(const void* d_mosaic, int in_dtype, void* d_packed, void* d_cond, int out_dtype, int batch, int h, int w,
                  int hp, int wp, int cond_h, int cond_w, float black_level, float white_level, void* stream) {
    RC_REQUIRE(d_mosaic && d_packed && d_cond, "rc_raw_ingest: null pointer");
    RC_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && hp >= h && wp >= w && cond_h >= 1 && cond_w >= 1, "rc_raw_ingest: bad shape");
    RC_REQUIRE(white_level > black_level, "rc_raw_ingest: white_level must exceed black_level");
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_packed) % 16 == 0, "rc_raw_ingest: packed must be 16-byte aligned");
    const size_t total = (size_t)batch * hp * wp + (size_t)batch * cond_h * cond_w;
    const float inv = 1.f / (white_level - black_level);
    const float sy = (float)h / (float)cond_h, sx = (float)w / (float)cond_w;
#define CALL(TI, TO)                                                                                                        \
    hipLaunchKernelGGL((raw_ingest_kernel<TI, TO>), dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),          \
                       static_cast<const TI*>(d_mosaic), static_cast<TO*>(d_packed), static_cast<TO*>(d_cond), batch, h, w, \
                       hp, wp, cond_h, cond_w, black_level, inv, sy, sx)
    if (in_dtype == RC_U16) {
        if (out_dtype == RC_F32) { CALL(uint16_t, float); } else if (out_dtype == RC_BF16) { CALL(uint16_t, bf16_t); } else return fail(RC_ERR_INVALID, "bad dtype");
    } else {
        RC_DISPATCH_2(in_dtype, out_dtype, CALL);
    }
#undef CALL
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_nchw_to_nhwc(const void* d_src, int src_dtype, void* d_dst, int dst_dtype,
                    int batch, int c, int h, int w, int hp, int wp, void* stream) {
    RC_REQUIRE(d_src && d_dst, "rc_nchw_to_nhwc: null pointer");
    RC_REQUIRE(batch >= 1 && c >= 1 && h >= 1 && w >= 1 && hp >= h && wp >= w, "rc_nchw_to_nhwc: bad shape");
    RC_REQUIRE(hp <= 65535 && batch <= 65535, "rc_nchw_to_nhwc: dimension too large");
    if (c <= 4) {
        const size_t total = (size_t)batch * hp * wp;
#define CALL(TI, TO)                                                                                                  \
    hipLaunchKernelGGL((nchw_to_nhwc_small_kernel<TI, TO>), dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream), \
                       static_cast<const TI*>(d_src), static_cast<TO*>(d_dst), batch, c, h, w, hp, wp)
        RC_DISPATCH_2(src_dtype, dst_dtype, CALL);
#undef CALL
        RC_HIP_CHECK(hipGetLastError());
        return RC_OK;
    }
    dim3 grid(ceil_div(wp, 32), hp, batch);
#define CALL(TI, TO)                                                                               \
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<TI, TO>), grid, dim3(256), 0, as_stream(stream),       \
                       static_cast<const TI*>(d_src), static_cast<TO*>(d_dst), c, h, w, hp, wp)
    RC_DISPATCH_2(src_dtype, dst_dtype, CALL);
#undef CALL
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_nhwc_to_nchw(const void* d_src, int src_dtype, void* d_dst, int dst_dtype,
                    int batch, int c, int H, int W, int h, int w, void* stream) {
    RC_REQUIRE(d_src && d_dst, "rc_nhwc_to_nchw: null pointer");
    RC_REQUIRE(batch >= 1 && c >= 1 && h >= 1 && w >= 1 && h <= H && w <= W, "rc_nhwc_to_nchw: bad shape");
    RC_REQUIRE(h <= 65535 && batch <= 65535, "rc_nhwc_to_nchw: dimension too large");
    dim3 grid(ceil_div(w, 32), h, batch);
#define CALL(TI, TO)                                                                               \
    hipLaunchKernelGGL((nhwc_to_nchw_kernel<TI, TO>), grid, dim3(256), 0, as_stream(stream),       \
                       static_cast<const TI*>(d_src), static_cast<TO*>(d_dst), c, H, W, h, w)
    RC_DISPATCH_2(src_dtype, dst_dtype, CALL);
#undef CALL
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_gate_residual(const void* d_r, const float* d_gate, const void* d_x, void* d_y, int dtype,
                     int batch, int n_pix, int c, void* stream) {
    RC_REQUIRE(d_r && d_gate && d_y, "rc_gate_residual: null pointer");
    const int U = dtype == RC_F32 ? 4 : 8;
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_gate_residual: bad dtype");
    RC_REQUIRE(batch >= 1 && n_pix >= 1 && c >= U && c % U == 0, "rc_gate_residual: c must be a multiple of 16 bytes");
    const size_t total = (size_t)batch * n_pix * (c / U);
    if (dtype == RC_F32)
        hipLaunchKernelGGL(gate_residual_kernel<float>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const float*>(d_r), d_gate, static_cast<const float*>(d_x), static_cast<float*>(d_y), batch, (size_t)n_pix, c);
    else
        hipLaunchKernelGGL(gate_residual_kernel<bf16_t>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const bf16_t*>(d_r), d_gate, static_cast<const bf16_t*>(d_x), static_cast<bf16_t*>(d_y), batch, (size_t)n_pix, c);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_film_apply(const void* d_x, const float* d_scale, const float* d_shift, void* d_y, int dtype, int batch, int n_pix, int c, void* stream) {
    RC_REQUIRE(d_x && d_scale && d_shift && d_y, "rc_film_apply: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_film_apply: bad dtype");
    const int U = dtype == RC_F32 ? 4 : 8;
    RC_REQUIRE(batch >= 1 && n_pix >= 1 && c >= U && c % U == 0, "rc_film_apply: c must be a multiple of 16 bytes");
    const size_t total = (size_t)batch * n_pix * (c / U);
    if (dtype == RC_F32)
        hipLaunchKernelGGL(film_apply_kernel<float>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream), static_cast<const float*>(d_x),
                           d_scale, d_shift, static_cast<float*>(d_y), batch, (size_t)n_pix, c);
    else
        hipLaunchKernelGGL(film_apply_kernel<bf16_t>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream), static_cast<const bf16_t*>(d_x),
                           d_scale, d_shift, static_cast<bf16_t*>(d_y), batch, (size_t)n_pix, c);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_sigmoid_gate_add(const void* d_a, const void* d_b, const void* d_identity, void* d_y, int dtype, long long n_elems,
                        void* stream) {
    RC_REQUIRE(d_a && d_b && d_identity && d_y, "rc_sigmoid_gate_add: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_sigmoid_gate_add: bad dtype");
    const int U = dtype == RC_F32 ? 4 : 8;
    RC_REQUIRE(n_elems >= U && n_elems % U == 0, "rc_sigmoid_gate_add: element count must be a whole number of 16-byte vectors");
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_a) % 16 == 0 && reinterpret_cast<uintptr_t>(d_b) % 16 == 0 &&
               reinterpret_cast<uintptr_t>(d_identity) % 16 == 0 && reinterpret_cast<uintptr_t>(d_y) % 16 == 0, "rc_sigmoid_gate_add: 16-byte alignment");
    const size_t total = (size_t)n_elems / U;
    if (dtype == RC_F32)
        hipLaunchKernelGGL(sigmoid_gate_add_kernel<float>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const float*>(d_a), static_cast<const float*>(d_b), static_cast<const float*>(d_identity), static_cast<float*>(d_y), total);
    else
        hipLaunchKernelGGL(sigmoid_gate_add_kernel<bf16_t>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const bf16_t*>(d_a), static_cast<const bf16_t*>(d_b), static_cast<const bf16_t*>(d_identity), static_cast<bf16_t*>(d_y), total);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_subsample2(const void* d_src, void* d_dst, int dtype, int batch, int H, int W, int c, void* stream) {
    RC_REQUIRE(d_src && d_dst, "rc_subsample2: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_subsample2: bad dtype");
    const int U = dtype == RC_F32 ? 4 : 8;
    RC_REQUIRE(batch >= 1 && H >= 1 && W >= 1 && c >= U && c % U == 0, "rc_subsample2: channels must be a whole number of 16-byte vectors");
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_src) % 16 == 0 && reinterpret_cast<uintptr_t>(d_dst) % 16 == 0, "rc_subsample2: 16-byte alignment");
    const size_t total = (size_t)batch * ((H + 1) / 2) * ((W + 1) / 2) * (c / U);
    if (dtype == RC_F32)
        hipLaunchKernelGGL(subsample2_kernel<float>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const float*>(d_src), static_cast<float*>(d_dst), batch, H, W, c);
    else
        hipLaunchKernelGGL(subsample2_kernel<bf16_t>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const bf16_t*>(d_src), static_cast<bf16_t*>(d_dst), batch, H, W, c);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_upsample_bilinear2(const void* d_src, void* d_dst, int dtype, int batch, int H, int W, int c, void* stream) {
    RC_REQUIRE(d_src && d_dst, "rc_upsample_bilinear2: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_upsample_bilinear2: bad dtype");
    const int U = dtype == RC_F32 ? 4 : 8;
    RC_REQUIRE(batch >= 1 && H >= 1 && W >= 1 && c >= U && c % U == 0, "rc_upsample_bilinear2: channels must be a whole number of 16-byte vectors");
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_src) % 16 == 0 && reinterpret_cast<uintptr_t>(d_dst) % 16 == 0, "rc_upsample_bilinear2: 16-byte alignment");
    const size_t total = (size_t)batch * 4 * H * W * (c / U);
    if (dtype == RC_F32)
        hipLaunchKernelGGL(upsample_bilinear2_kernel<float>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const float*>(d_src), static_cast<float*>(d_dst), batch, H, W, c);
    else
        hipLaunchKernelGGL(upsample_bilinear2_kernel<bf16_t>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const bf16_t*>(d_src), static_cast<bf16_t*>(d_dst), batch, H, W, c);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_sft_apply(const void* d_x, const void* d_scale, const void* d_shift, const void* d_identity, void* d_y, int dtype,
                 long long n_elems, void* stream) {
    RC_REQUIRE(d_x && d_scale && d_shift && d_y, "rc_sft_apply: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_sft_apply: bad dtype");
    const int U = dtype == RC_F32 ? 4 : 8;
    RC_REQUIRE(n_elems >= U && n_elems % U == 0, "rc_sft_apply: element count must be a whole number of 16-byte vectors");
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_x) % 16 == 0 && reinterpret_cast<uintptr_t>(d_scale) % 16 == 0 && reinterpret_cast<uintptr_t>(d_shift) % 16 == 0 &&
               reinterpret_cast<uintptr_t>(d_identity) % 16 == 0 && reinterpret_cast<uintptr_t>(d_y) % 16 == 0, "rc_sft_apply: 16-byte alignment");
    const size_t total = (size_t)n_elems / U;
    if (dtype == RC_F32)
        hipLaunchKernelGGL(sft_apply_kernel<float>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const float*>(d_x), static_cast<const float*>(d_scale), static_cast<const float*>(d_shift),
                           static_cast<const float*>(d_identity), static_cast<float*>(d_y), total);
    else
        hipLaunchKernelGGL(sft_apply_kernel<bf16_t>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const bf16_t*>(d_x), static_cast<const bf16_t*>(d_scale), static_cast<const bf16_t*>(d_shift),
                           static_cast<const bf16_t*>(d_identity), static_cast<bf16_t*>(d_y), total);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_space_to_depth2(const void* d_src, void* d_dst, int dtype, int batch, int H, int W, int c, void* stream) {
    RC_REQUIRE(d_src && d_dst, "rc_space_to_depth2: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_space_to_depth2: bad dtype");
    RC_REQUIRE(batch >= 1 && H >= 1 && W >= 1 && c >= 1, "rc_space_to_depth2: bad shape");
    const int U = dtype == RC_F32 ? 4 : 8;
    if (c % U == 0 && reinterpret_cast<uintptr_t>(d_src) % 16 == 0 && reinterpret_cast<uintptr_t>(d_dst) % 16 == 0) {
        const size_t tv = (size_t)batch * ((H + 1) / 2) * ((W + 1) / 2) * 4 * (c / U);
        if (dtype == RC_F32)
            hipLaunchKernelGGL(space_to_depth2_vec_kernel<float>, dim3(grid_for(tv)), dim3(kPwThreads), 0, as_stream(stream),
                               static_cast<const float*>(d_src), static_cast<float*>(d_dst), batch, H, W, c);
        else
            hipLaunchKernelGGL(space_to_depth2_vec_kernel<bf16_t>, dim3(grid_for(tv)), dim3(kPwThreads), 0, as_stream(stream),
                               static_cast<const bf16_t*>(d_src), static_cast<bf16_t*>(d_dst), batch, H, W, c);
        RC_HIP_CHECK(hipGetLastError());
        return RC_OK;
    }
    const size_t total = (size_t)batch * ((H + 1) / 2) * ((W + 1) / 2) * 4 * c;
    if (dtype == RC_F32)
        hipLaunchKernelGGL(space_to_depth2_kernel<float>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const float*>(d_src), static_cast<float*>(d_dst), batch, H, W, c);
    else
        hipLaunchKernelGGL(space_to_depth2_kernel<bf16_t>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const bf16_t*>(d_src), static_cast<bf16_t*>(d_dst), batch, H, W, c);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_pixel_shuffle2(const void* d_src, void* d_dst, int dtype, int batch, int H, int W, int c_out, void* stream) {
    RC_REQUIRE(d_src && d_dst, "rc_pixel_shuffle2: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_pixel_shuffle2: bad dtype");
    RC_REQUIRE(batch >= 1 && H >= 1 && W >= 1 && c_out >= 1, "rc_pixel_shuffle2: bad shape");
    const size_t total = (size_t)batch * 4 * H * W * c_out;
    if (dtype == RC_F32)
        hipLaunchKernelGGL(pixel_shuffle2_kernel<float>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const float*>(d_src), static_cast<float*>(d_dst), batch, H, W, c_out);
    else
        hipLaunchKernelGGL(pixel_shuffle2_kernel<bf16_t>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const bf16_t*>(d_src), static_cast<bf16_t*>(d_dst), batch, H, W, c_out);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_pixel_shuffle2_nchw(const void* d_src, void* d_dst, int dtype, int batch, int H, int W, int c_out, void* stream) {
    RC_REQUIRE(d_src && d_dst, "rc_pixel_shuffle2_nchw: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_pixel_shuffle2_nchw: bad dtype");
    RC_REQUIRE(batch >= 1 && H >= 1 && W >= 1 && c_out >= 1, "rc_pixel_shuffle2_nchw: bad shape");
    const size_t total = (size_t)batch * 4 * H * W * c_out;
    if (dtype == RC_F32)
        hipLaunchKernelGGL(pixel_shuffle2_nchw_kernel<float>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const float*>(d_src), static_cast<float*>(d_dst), batch, H, W, c_out);
    else
        hipLaunchKernelGGL(pixel_shuffle2_nchw_kernel<bf16_t>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const bf16_t*>(d_src), static_cast<bf16_t*>(d_dst), batch, H, W, c_out);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_square(const void* d_x, void* d_y, int dtype, long long n_elems, void* stream) {
    RC_REQUIRE(d_x && d_y, "rc_square: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_square: bad dtype");
    const int U = dtype == RC_F32 ? 4 : 8;
    RC_REQUIRE(n_elems >= U && n_elems % U == 0, "rc_square: element count must be a whole number of 16-byte vectors");
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_x) % 16 == 0 && reinterpret_cast<uintptr_t>(d_y) % 16 == 0, "rc_square: 16-byte alignment");
    const size_t total = (size_t)n_elems / U;
    if (dtype == RC_F32)
        hipLaunchKernelGGL(square_kernel<float>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const float*>(d_x), static_cast<float*>(d_y), total);
    else
        hipLaunchKernelGGL(square_kernel<bf16_t>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const bf16_t*>(d_x), static_cast<bf16_t*>(d_y), total);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_gdn_apply(const void* d_x, const void* d_norm, const void* d_identity, void* d_y, int dtype, int inverse, long long n_elems,
                 void* stream) {
    RC_REQUIRE(d_x && d_norm && d_y, "rc_gdn_apply: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_gdn_apply: bad dtype");
    const int U = dtype == RC_F32 ? 4 : 8;
    RC_REQUIRE(n_elems >= U && n_elems % U == 0, "rc_gdn_apply: element count must be a whole number of 16-byte vectors");
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_x) % 16 == 0 && reinterpret_cast<uintptr_t>(d_norm) % 16 == 0 &&
               reinterpret_cast<uintptr_t>(d_identity) % 16 == 0 && reinterpret_cast<uintptr_t>(d_y) % 16 == 0, "rc_gdn_apply: 16-byte alignment");
    const size_t total = (size_t)n_elems / U;
#define RC_GDN(T, INV) hipLaunchKernelGGL((gdn_apply_kernel<T, INV>), dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream), \
        static_cast<const T*>(d_x), static_cast<const T*>(d_norm), static_cast<const T*>(d_identity), static_cast<T*>(d_y), total)
    if (dtype == RC_F32) { if (inverse) RC_GDN(float, true); else RC_GDN(float, false); }
    else { if (inverse) RC_GDN(bf16_t, true); else RC_GDN(bf16_t, false); }
#undef RC_GDN
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_channel_copy(const void* d_src, int src_stride_c, int src_c0, void* d_dst, int dst_stride_c, int dst_c0, int n_ch,
                    long long pixels, int dtype, void* stream) {
    RC_REQUIRE(d_src && d_dst, "rc_channel_copy: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_channel_copy: bad dtype");
    const int U = dtype == RC_F32 ? 4 : 8;
    RC_REQUIRE(pixels >= 1 && n_ch >= U && n_ch % U == 0 && src_stride_c % U == 0 && dst_stride_c % U == 0 && src_c0 % U == 0 && dst_c0 % U == 0 &&
               src_c0 >= 0 && dst_c0 >= 0 && src_c0 + n_ch <= src_stride_c && dst_c0 + n_ch <= dst_stride_c,
               "rc_channel_copy: channel ranges must be whole 16-byte vectors inside both tensors");
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_src) % 16 == 0 && reinterpret_cast<uintptr_t>(d_dst) % 16 == 0, "rc_channel_copy: 16-byte alignment");
    const size_t total = (size_t)pixels * (n_ch / U);
    if (dtype == RC_F32)
        hipLaunchKernelGGL(channel_copy_kernel<float>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const float*>(d_src), src_stride_c, src_c0, static_cast<float*>(d_dst), dst_stride_c, dst_c0, n_ch, (size_t)pixels);
    else
        hipLaunchKernelGGL(channel_copy_kernel<bf16_t>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const bf16_t*>(d_src), src_stride_c, src_c0, static_cast<bf16_t*>(d_dst), dst_stride_c, dst_c0, n_ch, (size_t)pixels);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_channel_concat(const void* const* d_parts, const int* widths, int n_parts, void* d_dst, long long pixels, int dtype, void* stream) {
    RC_REQUIRE(d_parts && widths && d_dst && n_parts >= 1 && n_parts <= 8 && pixels >= 1, "rc_channel_concat: 1..8 parts");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_channel_concat: bad dtype");
    const int U = dtype == RC_F32 ? 4 : 8;
    ConcatArgs a{};
    int v0 = 0;
    for (int k = 0; k < n_parts; ++k) {
        RC_REQUIRE(d_parts[k] && widths[k] >= U && widths[k] % U == 0 && reinterpret_cast<uintptr_t>(d_parts[k]) % 16 == 0,
                   "rc_channel_concat: every part a whole number of 16-byte vectors, 16-byte aligned");
        a.src[k] = static_cast<const uint4*>(d_parts[k]); a.vec[k] = widths[k] / U; a.vec0[k] = v0; v0 += widths[k] / U;
    }
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_dst) % 16 == 0, "rc_channel_concat: 16-byte alignment");
    a.n = n_parts; a.dst_vec = v0;
    hipLaunchKernelGGL(channel_concat_kernel, dim3(grid_for((size_t)pixels * v0)), dim3(kPwThreads), 0, as_stream(stream), a, static_cast<uint4*>(d_dst), (size_t)pixels);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_dwt_forward(const void* d_x, void* d_y, const float* d_taps, int taps_uniform, int dtype,
                   int batch, int H, int W, int c, void* stream) {
    RC_REQUIRE(d_x && d_y && d_taps, "rc_dwt_forward: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_dwt_forward: bad dtype");
    const int U = dtype == RC_F32 ? 4 : 8;
    RC_REQUIRE(batch >= 1 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "rc_dwt_forward: H and W must be even");
    RC_REQUIRE(c % U == 0, "rc_dwt_forward: channels must be a multiple of 16 bytes");
    const size_t total = (size_t)batch * (H / 2) * (W / 2) * (c / U);
    if (dtype == RC_F32) {
        if (taps_uniform) hipLaunchKernelGGL((dwt_forward_kernel<float, true>), dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const float*>(d_x), static_cast<float*>(d_y), d_taps, batch, H, W, c);
        else hipLaunchKernelGGL((dwt_forward_kernel<float, false>), dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const float*>(d_x), static_cast<float*>(d_y), d_taps, batch, H, W, c);
    } else {
        if (taps_uniform) hipLaunchKernelGGL((dwt_forward_kernel<bf16_t, true>), dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const bf16_t*>(d_x), static_cast<bf16_t*>(d_y), d_taps, batch, H, W, c);
        else hipLaunchKernelGGL((dwt_forward_kernel<bf16_t, false>), dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const bf16_t*>(d_x), static_cast<bf16_t*>(d_y), d_taps, batch, H, W, c);
    }
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_dwt_inverse(const void* d_x, void* d_y, const float* d_taps, int taps_uniform, int dtype,
                   int batch, int h, int w, int c4, void* stream) {
    RC_REQUIRE(d_x && d_y && d_taps, "rc_dwt_inverse: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_dwt_inverse: bad dtype");
    const int U = dtype == RC_F32 ? 4 : 8;
    RC_REQUIRE(batch >= 1 && h >= 1 && w >= 1 && c4 % 4 == 0 && (c4 / 4) % U == 0,
               "rc_dwt_inverse: out channels (c4/4) must be a multiple of 16 bytes");
    const size_t total = (size_t)batch * h * w * (c4 / 4 / U);
    if (dtype == RC_F32) {
        if (taps_uniform) hipLaunchKernelGGL((dwt_inverse_kernel<float, true>), dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const float*>(d_x), static_cast<float*>(d_y), d_taps, batch, h, w, c4);
        else hipLaunchKernelGGL((dwt_inverse_kernel<float, false>), dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const float*>(d_x), static_cast<float*>(d_y), d_taps, batch, h, w, c4);
    } else {
        if (taps_uniform) hipLaunchKernelGGL((dwt_inverse_kernel<bf16_t, true>), dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const bf16_t*>(d_x), static_cast<bf16_t*>(d_y), d_taps, batch, h, w, c4);
        else hipLaunchKernelGGL((dwt_inverse_kernel<bf16_t, false>), dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream),
                           static_cast<const bf16_t*>(d_x), static_cast<bf16_t*>(d_y), d_taps, batch, h, w, c4);
    }
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_tail_ring_gather(const void* d_x, void* d_rows, void* d_cols, int dtype, int batch, int H, int W, int c, void* stream) {
    RC_REQUIRE(d_x && d_rows && d_cols, "rc_tail_ring_gather: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_tail_ring_gather: bad dtype");
    RC_REQUIRE(batch >= 1 && H >= 2 && W >= 2 && c >= 1 && (c * dtype_size(dtype)) % 16 == 0, "rc_tail_ring_gather: H, W >= 2 and a pixel record of whole 16-byte vectors");
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_x) % 16 == 0 && reinterpret_cast<uintptr_t>(d_rows) % 16 == 0 && reinterpret_cast<uintptr_t>(d_cols) % 16 == 0,
               "rc_tail_ring_gather: 16-byte alignment");
    const int pix16 = (int)(c * dtype_size(dtype) / 16);
    const size_t total = (size_t)4 * batch * pix16 * ((size_t)W + H);
    hipLaunchKernelGGL(tail_ring_gather_kernel, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream), static_cast<const uint4*>(d_x),
                       static_cast<uint4*>(d_rows), static_cast<uint4*>(d_cols), batch, H, W, pix16);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_tail_ring_scatter(const void* d_rows_out, const void* d_cols_out, void* d_out, int dtype, int batch, int c_out, int H, int W,
                         int out_h, int out_w, void* stream) {
    RC_REQUIRE(d_rows_out && d_cols_out && d_out, "rc_tail_ring_scatter: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_tail_ring_scatter: bad dtype");
    RC_REQUIRE(batch >= 1 && c_out >= 1 && H >= 2 && W >= 2 && out_h >= 1 && out_h <= 2 * H && out_w >= 1 && out_w <= 2 * W, "rc_tail_ring_scatter: bad shape");
    const size_t total = (size_t)batch * c_out * (2 * (size_t)out_w + 2 * (size_t)out_h);
    if (dtype == RC_F32)
        hipLaunchKernelGGL(tail_ring_scatter_kernel<float>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream), static_cast<const float*>(d_rows_out),
                           static_cast<const float*>(d_cols_out), static_cast<float*>(d_out), batch, c_out, H, W, out_h, out_w);
    else
        hipLaunchKernelGGL(tail_ring_scatter_kernel<bf16_t>, dim3(grid_for(total)), dim3(kPwThreads), 0, as_stream(stream), static_cast<const bf16_t*>(d_rows_out),
                           static_cast<const bf16_t*>(d_cols_out), static_cast<bf16_t*>(d_out), batch, c_out, H, W, out_h, out_w);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

}  // extern "C"
