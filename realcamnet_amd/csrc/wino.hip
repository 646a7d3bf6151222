// Winograd F(2x2, 3x3) for the 3x3 stride-1 "same" convolutions (networks.conv mode 'C', models/networks.py:146-160), fp32 on
// v_mfma_f32_16x16x4_f32.  DESIGN.md section 4.13.
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A          per 2x2 output tile, d = its 4x4 input patch
//
// 16 MACs per (cin, cout) and 4 outputs instead of 36: 2.25x fewer MFMAs than the implicit GEMM of conv_kernel.hpp.  As GEMMs: 16
// independent products M_xi[cout][tile] = sum_cin U_xi[cout][cin] V_xi[cin][tile], xi = 4 * row + col of the transformed 4x4 tile.
//   A operand = U_xi (packed by rc_wino_pack_weights: G g G^T in double, rounded once)      rows  = 16 couts
//   B operand = V_xi, computed IN THE LANE THAT FEEDS IT: B-operand lane (n = lane & 15, k = lane >> 4) supplies tile n, channel k --
//               that lane reads its tile's 4x4 patch of its channel from the LDS halo tile and runs B^T d B (32 adds) in registers;
//               the 16 results ARE the 16 xi's B fragments.  The transformed tile never exists in memory.
//   D          = lane (n, g) holds couts 4g..4g+3 of tile n for all 16 xi in the same register slot, so A^T M A is in-lane too (24 adds
//               per cout) and a lane ends with 4 consecutive couts of 4 pixels: 16-byte NHWC stores.
//
// Block = 2 * NCW waves: wave (cw, th) owns cout tile cw (16 couts) and tile rows 2 th, 2 th + 1 of an 8 x 32-pixel region (4 x 16 tiles),
// i.e. 2 n-tiles x 16 xi = 32 accumulator tiles (128 registers).  Every wave of a cout tile re-derives V for its tiles (the fp32 MFMA
// takes 32 cycles: 64 VALU adds per 32 MFMAs are noise); nothing but the raw halo tile and the packed U chunk goes through LDS.
// Persistent: a block walks items (image, region, cout group) x 8-channel stages as ONE flat list, the next stage's input pieces and
// U chunk are fetched into registers during the current stage's MFMAs and written to the other LDS buffer before the stage's single
// barrier.  Zero padding, ragged regions and odd sizes are buffer bounds checks (kOOB), as in conv_kernel.hpp.
#include "conv_kernel.hpp"

namespace rc {

void conv_prof_begin(double flops, hipStream_t stream, void** token, double bytes, int cin, int cout, int ksize);
void conv_prof_end(void* token, hipStream_t stream);

struct WinoArgs {
    int batch, H, W, cin, cout;
    int rx, ry, n_cg, n_items, n_chunks;
    MagicDiv d_cg, d_img, d_rx, d_chunks;      // scalar divisions by n_cg, rx * ry, rx, n_chunks
    const void* in0; const void* wpacked; const float* bias;
    const float* film_scale; const float* film_shift;
    int act; float act_slope;
    const float* out_scale; const void* residual;
    void* out; float* chan_sums; int sum_slots;
    int dbg_flags;                     // rc_debug_set("conv_flags") knock-outs (timing experiments): 1 no epilogue, 2 no MFMA phase, 4 no loads, 8 no transform
    long long* dbg;                    // rc_debug_set_ptr("conv_phase_timing"): s_memtime stamps of block 0, waves 0 and 1: [wave][stage < 32][8]
};

constexpr int kWRH = 4;                            // output rows of one item: two Winograd tile rows
constexpr int kWHH = kWRH + 2;                     // halo rows
constexpr int kWCK = 8;                            // input channels per stage (two K = 4 MFMA steps)

// NNT = n-tiles (16 Winograd tiles) per wave.  2: the item is a 4 x 32-pixel region (2 x 16 tiles, n-tile = one tile row); 1: a 4 x 16-pixel region
// (2 x 8 tiles = ONE n-tile) for maps with too few 4 x 32 regions to fill two blocks per CU (cfg2's 128-channel levels at 136 x 240 and 68 x 120).
// Raw halo tile in LDS: channel-PLANAR fp32 [channel][row][col].  The transform thread of (channel, tile row tr, tile column tc) reads its 4 x 4 patch as
// ds_read_b64 of columns (2 tc, 2 tc + 1) and (2 tc + 2, 2 tc + 3).  NNT 2: a 32-lane group is (tr 0..1) x (tc 0..15) of one channel: 48 dwords per row put
// tile row tr + 1 on the other 32 banks (2 * 48 == 32 mod 64).  NNT 1: a group is 2 channels x (tr 0..1) x (tc 0..7): 24 dwords per row (2 * 24 = 48) and a plane
// pitch == 32 mod 64 give the four (channel, tr) quarters four disjoint runs of 16 banks.  Conflict-free as b64 and as the read2_b64 pairs hipcc merges them into.
template <int NCW, int NNT>
struct WinoCfg {
    static constexpr int WAVES = NCW, THREADS = 64 * WAVES;
    static constexpr int TC = 8 * NNT;                                 // tile columns of a region
    static constexpr int RW = 2 * TC, HW = RW + 2;                     // region / halo width in pixels
    static constexpr int NPIX = kWHH * HW;                             // halo pixels
    static constexpr int RP = NNT == 2 ? 48 : 24;                      // dwords per halo row in LDS
    static constexpr int PLANE = NNT == 2 ? 384 : 224;                 // dwords per channel plane: 6 halo rows + pad rows (where pieces past the tile are parked)
    static constexpr int RAW = kWCK * PLANE * 4;                       // bytes per raw buffer
    static constexpr int V = 2 * NNT * 4 * 1024;                       // transformed tile V of one stage: [ks][n-tile][q][lane (n, k)][4 xi] floats
    static constexpr int U_STAGE = 2 * NCW * 4 * 1024;                 // [ks][cw][q][lane][4 xi] floats of one stage (global memory only)
    static constexpr int NIP = (2 * NPIX + THREADS - 1) / THREADS;     // 16-byte input pieces per thread and stage (a halo pixel = 2 pieces)
    static constexpr int NPAIR = 16 * NNT * kWCK;                      // (tile, channel) pairs of a stage
    static constexpr int NTR = (NPAIR + THREADS - 1) / THREADS;        // transforms per thread and stage (1 with 4 waves)
    static constexpr int LDS_BYTES = 2 * V + 2 * RAW;                  // 56 KB (NNT 2) / 30 KB: two blocks per CU
    static_assert(PLANE >= (kWHH + 1) * RP, "plane holds the halo tile and a pad row");
};

// Block = NCW waves, wave cw owns cout tile cw (16 couts) x the region's 32 tiles x 16 xi (128 accumulator registers); two blocks share a CU and drift freely,
// so one block's loads / transform / epilogue run under the other's MFMAs.  Iteration g of a block (one barrier each):
//   top      raw halo pieces of stage g + 2 -> registers
//   transform stage g + 1:  raw[(g + 1) & 1] -> V[(g + 1) & 1]       (each thread ONE (tile, channel): 8 ds_read_b64, 32 adds, 4 ds_write_b128)
//   MFMA     stage g:       B = V[g & 1] (16 ds_read_b128), A = this wave's U fragments straight from global memory / L2 into registers, fetched
//                           half a stage ahead (8 x 1 KB per stage: every block reads the same 16 * cin * cout floats, they live in L2)
//   epilogue if stage g closes an item
//   commit   registers -> raw[g & 1] (stage g's raw tile was consumed an iteration ago);  barrier
// DBG: knock-out flags (rc_debug_set("conv_flags")) and s_memtime stamps of block 0 (rc_debug_set_ptr("conv_phase_timing")) -- experiments only.
template <int NCW, int NNT, bool DBG>
__global__ __launch_bounds__(64 * NCW, (NCW == 4 ? 2 : 1)) void wino_f32_kernel(const WinoArgs a) {
    using C = WinoCfg<NCW, NNT>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const vbuf = lds;                                  // 2 x C::V
    char* const rbuf = vbuf + 2 * C::V;                      // 2 x C::RAW
    const int tid = threadIdx.x, lane = tid & 63, cw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int flags = DBG ? a.dbg_flags : 0;

    const int my_items = (a.n_items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nst = my_items * a.n_chunks;
    if (nst <= 0) return;

    // ---- item decode (uniform, multiply-high divisions) ----
    struct Item { int b, y0, x0, cg, reg; };
    auto decode = [&](int k) {
        Item it;
        int idx = (int)blockIdx.x + k * (int)gridDim.x;
        const int q = magic_div(idx, a.d_cg);
        it.cg = idx - q * a.n_cg;
        it.b = magic_div(q, a.d_img);
        it.reg = q - it.b * (a.rx * a.ry);
        const int ty = magic_div(it.reg, a.d_rx);
        it.y0 = ty * kWRH; it.x0 = (it.reg - ty * a.rx) * C::RW;
        return it;
    };

    // ---- loaders.  On this part a SIMD issues EITHER an MFMA pass OR another instruction (tools/ubench/mfma_valu_overlap.hip: an fp32-MFMA wave and a VALU /
    // LDS wave on one SIMD take the SUM of their times), so the loop below is written for instruction COUNT: no conditionals in the steady state (stages past
    // the end of the list fetch out-of-range pieces and transform a dead buffer), lane addresses as one register + immediates, packed fp32 adds.
    const int lane16 = lane * 16;
    // LDS byte offset of raw piece p (plane 4 q, halo pixel); pieces past the tile park in the pad rows
    auto piece_lds = [](int p) {
        const int pix = p >> 1, hy = pix / C::HW;
        return p < 2 * C::NPIX ? ((4 * (p & 1)) * C::PLANE + hy * C::RP + pix - hy * C::HW) * 4 : ((4 * (p & 1)) * C::PLANE + kWHH * C::RP + (p & 15)) * 4;
    };
    constexpr bool KEEP_PLDS = NCW == 4;                     // narrower blocks carry more pieces per thread: there the offsets are recomputed at the commit (from a
    int p_lds[KEEP_PLDS ? C::NIP : 1];                       // laundered thread id, or hipcc hoists them back out of the loop and spills)
    if constexpr (KEEP_PLDS) {
#pragma unroll
        for (int i = 0; i < C::NIP; ++i) p_lds[i] = piece_lds(tid + i * C::THREADS);
    }
    uint4 in_r[C::NIP];
    int voff[C::NIP];                                        // byte offset of the pieces inside the image being loaded (kOOB = zero padding)
    __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in0, 0u);
    const size_t img_in = (size_t)a.H * a.W * a.cin;
    int kr = 0, cr = 0;                                      // (item, chunk) of the next raw stage to fetch
    auto issue_raw = [&]() {
        if (cr == 0) {                                       // a new item: its image descriptor and this thread's piece offsets
            const bool live = kr < my_items;
            const Item it = decode(live ? kr : 0);
            rs_in = make_rsrc(static_cast<const float*>(a.in0) + (size_t)it.b * img_in, live ? (unsigned)(img_in * 4) : 0u);
#pragma unroll
            for (int i = 0; i < C::NIP; ++i) {
                const int p = tid + i * C::THREADS, pix = p >> 1, hy = pix / C::HW;
                const int gy = it.y0 - 1 + hy, gx = it.x0 - 1 + pix - hy * C::HW;
                const bool ok = p < 2 * C::NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                voff[i] = ok ? ((gy * a.W + gx) * a.cin + 4 * (p & 1)) * 4 : kOOB;
            }
        }
#pragma unroll
        for (int i = 0; i < C::NIP; ++i) in_r[i] = buf_load16(rs_in, voff[i], cr * (kWCK * 4));
        if (++cr == a.n_chunks) { cr = 0; ++kr; }
    };
    auto commit_raw = [&](int buf) {                          // 16-byte piece = 4 channels of one halo pixel -> 4 planes
        int t0 = tid;
        if constexpr (!KEEP_PLDS) asm volatile("" : "+v"(t0));
#pragma unroll
        for (int i = 0; i < C::NIP; ++i) {
            float* dst = reinterpret_cast<float*>(rbuf + buf * C::RAW + (KEEP_PLDS ? p_lds[KEEP_PLDS ? i : 0] : piece_lds(t0 + i * C::THREADS)));
            dst[0] = __uint_as_float(in_r[i].x); dst[C::PLANE] = __uint_as_float(in_r[i].y);
            dst[2 * C::PLANE] = __uint_as_float(in_r[i].z); dst[3 * C::PLANE] = __uint_as_float(in_r[i].w);
        }
    };
    // this wave's A fragments, half-stage by half-stage: 4 x 1 KB [q][lane][4 xi] per half; a UNIFORM base pointer walks (chunk, ks) and re-bases per item
    int ka = 0, ca = 0;
    const char* ua = static_cast<const char*>(a.wpacked);
    auto load_a = [&](f32x4 (&dst)[4]) {                     // call order: (g, ks 0), (g, ks 1), (g + 1, ks 0), ...
        if (ca == 0) {
            int idx = (int)blockIdx.x + ka * (int)gridDim.x;
            const int cg = idx - magic_div(idx, a.d_cg) * a.n_cg;           // (past the end of the list: still a valid cout group -> a valid address)
            ua = static_cast<const char*>(a.wpacked) + (size_t)cg * a.n_chunks * C::U_STAGE + cw * 4096;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const f32x4*>(ua + lane16 + q * 1024);
        ua += NCW * 4096;
        if (++ca == 2 * a.n_chunks) { ca = 0; ++ka; }
    };
    // B^T d B of this thread's (tile, channel) pairs: raw[buf] -> V[buf].  Pair p: channel p / (16 NNT), tile row, tile column from the rest.
    int t_src[C::NTR], t_dst[C::NTR];
#pragma unroll
    for (int t = 0; t < C::NTR; ++t) {
        const int p = tid + t * C::THREADS, ch = (p / (16 * NNT)) & 7, tr = (p / C::TC) & 1, tc = p % C::TC;
        t_src[t] = (ch * C::PLANE + (2 * tr) * C::RP + 2 * tc) * 4;
        // V[ks = ch & 1][n-tile][q][lane (n, k = ch >> 1)][e]: the B-operand lane of the tile, channel 2 k + ks.  NNT 2: n-tile = tile row, n = tc; NNT 1: n = 8 tr + tc
        const int nti = NNT == 2 ? tr : 0, n_ = NNT == 2 ? tc : 8 * tr + tc;
        t_dst[t] = (((ch & 1) * NNT + nti) * 4) * 1024 + (n_ + 16 * (ch >> 1)) * 16;
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 tx[C::NTR][4][2];                                  // the patches between transform_read and transform_finish: [row][column pair]
    auto transform_read = [&](int buf) {
#pragma unroll
        for (int t = 0; t < C::NTR; ++t) {
            const char* src = rbuf + buf * C::RAW + t_src[t];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                tx[t][r][0] = *reinterpret_cast<const f32x2*>(src + r * (C::RP * 4));
                tx[t][r][1] = *reinterpret_cast<const f32x2*>(src + r * (C::RP * 4) + 8);
            }
        }
    };
    auto transform_finish = [&](int buf) {
#pragma unroll
        for (int t = 0; t < C::NTR; ++t) {
            if (C::NTR * C::THREADS == C::NPAIR || tid + t * C::THREADS < C::NPAIR) {
                f32x2 w[4][2];                                   // rows of B^T d, as column pairs (packed adds)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    w[0][h] = tx[t][0][h] - tx[t][2][h]; w[1][h] = tx[t][1][h] + tx[t][2][h];
                    w[2][h] = tx[t][2][h] - tx[t][1][h]; w[3][h] = tx[t][1][h] - tx[t][3][h];
                }
                char* dst = vbuf + buf * C::V + t_dst[t];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<f32x4*>(dst + r * 1024) = f32x4{w[r][0].x - w[r][1].x, w[r][0].y + w[r][1].x, w[r][1].x - w[r][0].y, w[r][0].y - w[r][1].y};
            }
        }
    };

    f32x4 acc[16][NNT];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int nt = 0; nt < NNT; ++nt) acc[x][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // one half-stage.  (Reading the 8 B-fragment quads up front behind a sched_barrier measured 3 % SLOWER than hipcc's own pairing of two reads with the
    // eight MFMAs they feed: 183 vs 178 us on the 64 -> 64 layer, same box, alternating.)
    auto mfma_half = [&](const char* vb, const f32x4 (&aq)[4]) {
#pragma unroll
        for (int nt = 0; nt < NNT; ++nt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 vq = *reinterpret_cast<const f32x4*>(vb + nt * 4096 + q * 1024);
                acc[4 * q + 0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q].x, vq.x, acc[4 * q + 0][nt], 0, 0, 0);
                acc[4 * q + 1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q].y, vq.y, acc[4 * q + 1][nt], 0, 0, 0);
                acc[4 * q + 2][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q].z, vq.z, acc[4 * q + 2][nt], 0, 0, 0);
                acc[4 * q + 3][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q].w, vq.w, acc[4 * q + 3][nt], 0, 0, 0);
            }
        }
    };
    long long* const stamp = DBG && a.dbg && blockIdx.x == 0 && cw < 2 ? a.dbg + cw * 256 : nullptr;
    auto mark = [&](int g, int i) {
        if constexpr (DBG) { if (stamp && g < 32 && lane == 0) stamp[g * 8 + i] = (long long)__builtin_amdgcn_s_memtime(); }
    };

    // ---- prologue: raw(0) -> LDS, raw(1) in registers, A(0, 0) in flight; transform(0) ----
    f32x4 a0[4], a1[4];
    issue_raw();
    commit_raw(0);
    issue_raw();
    load_a(a0);
    __syncthreads();
    transform_read(0);
    transform_finish(0);
    commit_raw(1);
    __syncthreads();

    int kc = 0, cc = 0;                                      // (item, chunk) of the stage being multiplied
    const int co_w = 16 * cw + 4 * kq;                       // this lane's 4 couts inside the cout group
    for (int g = 0; g < nst; ++g) {
        const int buf = g & 1;
        mark(g, 0);
        if (!(flags & 8)) transform_read(buf ^ 1);                         // the patch reads fly under the address work
        if (!(flags & 4)) issue_raw();
        load_a(a1);
        if (!(flags & 8)) transform_finish(buf ^ 1);
        mark(g, 1);
        const char* vb = vbuf + buf * C::V + lane16;
        if (!(flags & 2)) mfma_half(vb, a0);
        mark(g, 2);
        load_a(a0);
        mark(g, 3);
        if (!(flags & 2)) mfma_half(vb + NNT * 4096, a1);
        mark(g, 4);

        if (++cc == a.n_chunks && !(flags & 1)) {
            // ---- item finished: Y = A^T M A in-lane (the accumulators die here), then the epilogue as whole-array passes under uniform branches ----
            const Item it = decode(kc);
            const int co0 = it.cg * (16 * NCW) + co_w;
            const size_t img_out = (size_t)a.H * a.W * a.cout;
            const __amdgpu_buffer_rsrc_t r_out = make_rsrc(static_cast<float*>(a.out) + (size_t)it.b * img_out, (unsigned)(img_out * 4));
            f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f}, osc = f32x4{1.f, 1.f, 1.f, 1.f};
            if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + co0);              // (their latency runs under the 192 adds of A^T M A)
            if (a.out_scale) osc = *reinterpret_cast<const f32x4*>(a.out_scale + (size_t)it.b * a.cout + co0);
            // a lane's tile: NNT 2: (tile row nt, tile column n); NNT 1: (tile row n >> 3, tile column n & 7).  Below, index m = (nt, i, j) runs over the lane's 4 NNT pixels
            constexpr int NPX = 4 * NNT;
            f32x4 y[NNT][2][2];                                             // [nt][i][j]
#pragma unroll
            for (int nt = 0; nt < NNT; ++nt) {
                f32x4 t0[4], t1[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    t0[c] = acc[c][nt] + acc[4 + c][nt] + acc[8 + c][nt];
                    t1[c] = acc[4 + c][nt] - acc[8 + c][nt] - acc[12 + c][nt];
                }
                y[nt][0][0] = t0[0] + t0[1] + t0[2]; y[nt][0][1] = t0[1] - t0[2] - t0[3];
                y[nt][1][0] = t1[0] + t1[1] + t1[2]; y[nt][1][1] = t1[1] - t1[2] - t1[3];
            }
#pragma unroll
            for (int m = 0; m < NPX; ++m) y[m >> 2][(m >> 1) & 1][m & 1] = y[m >> 2][(m >> 1) & 1][m & 1] + bias4;
#pragma unroll
            for (int x = 0; x < 16; ++x)
#pragma unroll
                for (int nt = 0; nt < NNT; ++nt) acc[x][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            // pixel (nt, i, j) = (py + 2 nt + i, px + j)
            const int px = it.x0 + 2 * (NNT == 2 ? n : (n & 7)), py = it.y0 + (NNT == 2 ? 0 : 2 * (n >> 3));
            const int off00 = ((py * a.W + px) * a.cout + co0) * 4;
            const bool vx0 = px < a.W, vx1 = px + 1 < a.W;
            int off[NNT][2][2];
#pragma unroll
            for (int nt = 0; nt < NNT; ++nt)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bool vy = py + 2 * nt + i < a.H;
                    const int o = off00 + ((2 * nt + i) * a.W * a.cout) * 4;
                    off[nt][i][0] = (vy && vx0) ? o : kOOB;
                    off[nt][i][1] = (vy && vx1) ? o + a.cout * 4 : kOOB;
                }
            f32x4 res[NNT][2][2];
            if (a.residual) {
                const __amdgpu_buffer_rsrc_t r_res = make_rsrc(static_cast<const float*>(a.residual) + (size_t)it.b * img_out, (unsigned)(img_out * 4));
#pragma unroll
                for (int m = 0; m < NPX; ++m) {
                    const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(r_res, off[m >> 2][(m >> 1) & 1][m & 1], 0, 0);
                    res[m >> 2][(m >> 1) & 1][m & 1] = f32x4{__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w)};
                }
            }
            if (a.film_scale) {
                const f32x4 fs = *reinterpret_cast<const f32x4*>(a.film_scale + (size_t)it.b * a.cout + co0);
                const f32x4 ft = *reinterpret_cast<const f32x4*>(a.film_shift + (size_t)it.b * a.cout + co0);
#pragma unroll
                for (int m = 0; m < NPX; ++m) { f32x4& v = y[m >> 2][(m >> 1) & 1][m & 1]; v = v * fs + ft + v; }
            }
            if (a.act == RC_ACT_RELU) {
#pragma unroll
                for (int m = 0; m < NPX; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float v = y[m >> 2][(m >> 1) & 1][m & 1][e]; y[m >> 2][(m >> 1) & 1][m & 1][e] = v > 0.f ? v : 0.f; }
            } else if (a.act == RC_ACT_LEAKY) {
#pragma unroll
                for (int m = 0; m < NPX; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float v = y[m >> 2][(m >> 1) & 1][m & 1][e]; y[m >> 2][(m >> 1) & 1][m & 1][e] = v > 0.f ? v : v * a.act_slope; }
            }
            if (a.out_scale) {
#pragma unroll
                for (int m = 0; m < NPX; ++m) y[m >> 2][(m >> 1) & 1][m & 1] = y[m >> 2][(m >> 1) & 1][m & 1] * osc;
            }
            if (a.residual) {
#pragma unroll
                for (int m = 0; m < NPX; ++m) y[m >> 2][(m >> 1) & 1][m & 1] = y[m >> 2][(m >> 1) & 1][m & 1] + res[m >> 2][(m >> 1) & 1][m & 1];
                if (a.act == RC_ACT_RELU_POST) {
#pragma unroll
                    for (int m = 0; m < NPX; ++m)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float v = y[m >> 2][(m >> 1) & 1][m & 1][e]; y[m >> 2][(m >> 1) & 1][m & 1][e] = v > 0.f ? v : 0.f; }
                }
            }
#pragma unroll
            for (int m = 0; m < NPX; ++m) {
                const f32x4 v = y[m >> 2][(m >> 1) & 1][m & 1];
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)},
                                                       r_out, off[m >> 2][(m >> 1) & 1][m & 1], 0, 0);
            }
            if (a.chan_sums) {
                // pixels outside the image do not count; fixed-order butterfly over the 16 tiles of a lane row, then one slot per region
                f32x4 csum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < NPX; ++m) {
                    const f32x4 v = y[m >> 2][(m >> 1) & 1][m & 1];
                    const bool ok = off[m >> 2][(m >> 1) & 1][m & 1] != kOOB;
#pragma unroll
                    for (int e = 0; e < 4; ++e) csum[e] += ok ? v[e] : 0.f;
                }
#pragma unroll
                for (int m = 1; m < 16; m <<= 1)
#pragma unroll
                    for (int e = 0; e < 4; ++e) csum[e] += __shfl_xor(csum[e], m, 64);
                if (n == 0) {
                    float* dst = a.chan_sums + ((size_t)it.b * a.sum_slots + (size_t)it.reg) * a.cout + co0;
                    *reinterpret_cast<f32x4*>(dst) = csum;
                }
            }
        }
        if (cc == a.n_chunks) { cc = 0; ++kc; }
        mark(g, 5);
        if (!(flags & 4)) commit_raw(buf);
        mark(g, 6);
        __syncthreads();
        mark(g, 7);
    }
}

long long* conv_dbg_ptr();

template <int NCW, int NNT, bool DBG = false>
static int wino_launch_f32(const WinoArgs& a, int num_cus, hipStream_t stream) {
    using C = WinoCfg<NCW, NNT>;
    static PerDeviceFlag attr_set;
    if (!attr_set.test_and_set())
        RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_f32_kernel<NCW, NNT, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    const int per_cu = NNT == 1 && NCW == 4 ? 3 : 2;                        // resident blocks per CU (registers, LDS)
    const int grid = a.n_items < per_cu * num_cus ? a.n_items : per_cu * num_cus;
    hipLaunchKernelGGL((wino_f32_kernel<NCW, NNT, DBG>), dim3(grid), dim3(C::THREADS), C::LDS_BYTES, stream, a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

// cout tiles (of 16) per block for a layer: the widest of 4 / 3 / 2 / 1 that divides cout / 16
static int wino_ncw(int cout) {
    if (cout % 64 == 0) return 4;
    if (cout % 48 == 0) return 3;
    if (cout % 32 == 0) return 2;
    return cout % 16 == 0 ? 1 : 0;
}

// n-tiles per wave for a layer: 1 (4 x 16-pixel items) when ONE image's 4 x 32 regions quantise badly over the chip's 2 x CUs block slots -- cfg2's 128-channel
// levels: 544 items = 1.06 rounds at 136 x 240, 136 items = a quarter round at 68 x 120 (half-size items: 3 x 0.5 / 1 x 0.5 rounds instead of 2 / 1).  Decided per IMAGE,
// not per batch, so that frame i of a batch and frame i alone get the same channel-sum slot layout (tested slot for slot).
static int g_wino_nnt = 0;                              // rc_debug_set("wino_nnt", v): 0 automatic, 1 / 2 forced (tests, A/B)
void wino_set_nnt(int v) { g_wino_nnt = v < 0 || v > 2 ? 0 : v; }
int wino_get_nnt() { return g_wino_nnt; }
static int wino_nnt(int H, int W, int cout, int num_cus) {
    const int ncw = wino_ncw(cout);
    if (ncw != 4) return 2;
    if (g_wino_nnt) return g_wino_nnt;
    const long ry = ceil_div(H, kWRH), n_cg = cout / 64;
    const long i2 = ry * ceil_div(W, 32) * n_cg, i1 = ry * ceil_div(W, 16) * n_cg;
    const long s2 = 2L * num_cus, s1 = 3L * num_cus;    // block slots: the half-size form needs 155 registers and 30 KB of LDS -> three blocks per CU
    const long r2 = (i2 + s2 - 1) / s2, r1 = (i1 + s1 - 1) / s1;
    return 81 * r1 < 100 * r2 ? 1 : 2;                 // measured (tools/wino_probe.py, RC_DEBUG=wino_nnt=1|2): a round of half-size items on three blocks per CU takes 0.81 of a round of
                                                       // full ones on two (64 -> 64 at 544 x 960: 192 us / 10.6 rounds vs 178 us / 7.97) -- half items win only where they save rounds
}

int wino_sum_slots(int H, int W, int cout) { return ceil_div(H, kWRH) * ceil_div(W, 16 * wino_nnt(H, W, cout, device_cu_count())); }

bool wino_supported(const rc_conv_desc* d, std::string* why) {
    auto no = [&](const char* m) { if (why) *why = m; return false; };
    if (d->ksize != 3) return no("Winograd F(2x2,3x3) is a 3x3 form");
    if (d->dtype != RC_F32) return no("Winograd: fp32 only");
    if (d->cin % kWCK != 0 || d->cout % 16 != 0) return no("Winograd: cin % 8 == 0 and cout % 16 == 0");
    if (d->out_mode != RC_OUT_NHWC || d->out_dtype != d->dtype) return no("Winograd: plain NHWC store");
    if (d->in_gate || d->in1 || d->in_store || d->mul_plus1 || d->src_h || d->src_w || d->cout_tile) return no("Winograd: no gated input / mul_plus1 / src_h / cout_tile");
    if (d->act == RC_ACT_GELU) return no("Winograd: no GELU epilogue");
    return true;
}

int wino_conv(const rc_conv_desc* d, hipStream_t stream) {
    std::string why;
    RC_REQUIRE(d != nullptr, "rc_conv2d: null desc");
    if (!wino_supported(d, &why)) return fail(RC_ERR_UNSUPPORTED, "rc_conv2d (algo 1): " + why);
    RC_REQUIRE(d->batch >= 1 && d->height >= 1 && d->width >= 1 && d->batch <= 65535, "rc_conv2d: empty tensor / batch > 65535");
    RC_REQUIRE(d->in0 && d->wpacked && d->out, "rc_conv2d: null in0/wpacked/out");
    RC_REQUIRE(d->act >= RC_ACT_NONE && d->act <= RC_ACT_RELU_POST, "rc_conv2d: bad act");
    RC_REQUIRE(d->act != RC_ACT_RELU_POST || (d->residual != nullptr && d->film_scale == nullptr && d->chan_sums == nullptr),
               "rc_conv2d: RC_ACT_RELU_POST is relu(conv + residual): needs residual, excludes film / chan_sums");
    RC_REQUIRE((d->film_scale == nullptr) == (d->film_shift == nullptr), "rc_conv2d: film_scale/film_shift must come together");
    auto al16 = [](const void* q) { return q == nullptr || reinterpret_cast<uintptr_t>(q) % 16 == 0; };
    RC_REQUIRE(al16(d->in0) && al16(d->out) && al16(d->residual) && al16(d->bias) && al16(d->film_scale) && al16(d->film_shift) && al16(d->out_scale) &&
               al16(d->wpacked) && al16(d->chan_sums), "rc_conv2d (algo 1): every operand must be 16-byte aligned");
    RC_REQUIRE((double)d->height * d->width * d->cin * 4.0 < 2147483647.0 && (double)d->height * d->width * d->cout * 4.0 < 2147483647.0,
               "rc_conv2d: one image must be < 2 GiB");
    WinoArgs a{};
    a.batch = d->batch; a.H = d->height; a.W = d->width; a.cin = d->cin; a.cout = d->cout;
    const int ncw = wino_ncw(d->cout);
    const int cus = device_cu_count();
    const int nnt = wino_nnt(d->height, d->width, d->cout, cus);
    a.rx = ceil_div(d->width, 16 * nnt); a.ry = ceil_div(d->height, kWRH);
    a.n_cg = d->cout / (16 * ncw);
    a.n_chunks = d->cin / kWCK;
    a.d_cg = make_magic(a.n_cg); a.d_img = make_magic(a.rx * a.ry); a.d_rx = make_magic(a.rx); a.d_chunks = make_magic(a.n_chunks);
    const double items = (double)d->batch * a.rx * a.ry * a.n_cg;
    RC_REQUIRE(items < 2147483647.0 / 64, "rc_conv2d (algo 1): too many items");
    a.n_items = (int)items;
    a.in0 = d->in0; a.wpacked = d->wpacked; a.bias = d->bias;
    a.film_scale = d->film_scale; a.film_shift = d->film_shift; a.act = d->act; a.act_slope = d->act_slope;
    a.out_scale = d->out_scale; a.residual = d->residual; a.out = d->out; a.chan_sums = d->chan_sums;
    a.sum_slots = wino_sum_slots(d->height, d->width, d->cout);
    a.dbg_flags = rc_debug_get("conv_flags");
    a.dbg = conv_dbg_ptr();
    if (d->chan_sums) RC_REQUIRE(d->chan_sums_slots == a.sum_slots, "rc_conv2d (algo 1): chan_sums_slots must be rc_conv_sum_slots()");
    const double px = (double)d->batch * d->height * d->width;
    void* tok = nullptr;
    // executed MFMA FLOPs: 16 MACs per (cin, cout) and 2x2 tile (tiles counted over the regions actually computed)
    conv_prof_begin(2.0 * 4.0 * (double)d->cin * d->cout * ((double)d->batch * a.rx * a.ry * kWRH * 16 * nnt), stream, &tok,
                    px * 4.0 * (d->cin + d->cout * (d->residual ? 2.0 : 1.0)), d->cin, d->cout, 3);
    int rc_ = RC_OK;
    switch (ncw) {
#ifdef RC_WINO_DBG          // the instrumented instantiation (knock-out flags, s_memtime stamps) is an experiment build: python -m realcamnet_amd.build --variant dbg RC_WINO_DBG
        case 4: rc_ = nnt == 1 ? wino_launch_f32<4, 1>(a, cus, stream) : (a.dbg_flags || a.dbg) ? wino_launch_f32<4, 2, true>(a, cus, stream) : wino_launch_f32<4, 2>(a, cus, stream); break;
#else
        case 4: rc_ = nnt == 1 ? wino_launch_f32<4, 1>(a, cus, stream) : wino_launch_f32<4, 2>(a, cus, stream); break;
#endif
        case 3: rc_ = wino_launch_f32<3, 2>(a, cus, stream); break;
        case 2: rc_ = wino_launch_f32<2, 2>(a, cus, stream); break;
        default: rc_ = wino_launch_f32<1, 2>(a, cus, stream); break;
    }
    conv_prof_end(tok, stream);
    return rc_;
}

}  // namespace rc

using namespace rc;

extern "C" {

size_t rc_wino_packed_bytes(int cin, int cout, int dtype) {
    if (dtype != RC_F32 || cin < kWCK || cin % kWCK != 0 || cout < 16 || cout % 16 != 0) { set_error("rc_wino_packed_bytes: fp32, cin % 8 == 0, cout % 16 == 0"); return 0; }
    return (size_t)16 * cin * cout * 4;
}

int rc_wino_pack_weights(const float* w, int cin, int cout, int dtype, void* dst) {
    RC_REQUIRE(w && dst, "rc_wino_pack_weights: null pointer");
    RC_REQUIRE(rc_wino_packed_bytes(cin, cout, dtype) != 0, "rc_wino_pack_weights: fp32, cin % 8 == 0, cout % 16 == 0");
    const int ncw = wino_ncw(cout), n_cg = cout / (16 * ncw), n_chunks = cin / kWCK;
    static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    float* out = static_cast<float*>(dst);
    // layout: [cout group][stage][ks][cw][q][lane][e]:  xi = 4 q + e, cout = cg * 16 NCW + 16 cw + (lane & 15), cin = 8 stage + 2 (lane >> 4) + ks
    for (int cg = 0; cg < n_cg; ++cg)
        for (int s = 0; s < n_chunks; ++s)
            for (int ks = 0; ks < 2; ++ks)
                for (int cw = 0; cw < ncw; ++cw)
                    for (int q = 0; q < 4; ++q)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int co = cg * 16 * ncw + 16 * cw + (lane & 15), ci = kWCK * s + 2 * (lane >> 4) + ks;
                            const float* g = w + ((size_t)co * cin + ci) * 9;
                            for (int e = 0; e < 4; ++e) {
                                double u = 0.0;                      // (G g G^T)[q][e]
                                for (int i = 0; i < 3; ++i)
                                    for (int j = 0; j < 3; ++j) u += G[q][i] * (double)g[3 * i + j] * G[e][j];
                                *out++ = (float)u;
                            }
                        }
    return RC_OK;
}

}  // extern "C"
