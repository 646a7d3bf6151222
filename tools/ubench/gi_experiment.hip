// Experiment (not part of the library): the unshipped gma_in + ConvPosEnc fusion of profiles/attempts/r06_gi_exp_kernel.hip.txt, built as a variant library to test ONE
// hypothesis about its run-to-run differences: its 1x1 stage alternates two accumulators (each revisited ONE MFMA later); RC_GX_ORDER=1 runs each accumulator's six
// K-steps back to back instead.  tools/dbg/gi_stress.py drives it.   hipcc ... -DRC_GX_ORDER=0|1 -c tools/ubench/gi_experiment.hip, linked with the library's other objects.
#include "../../realcamnet_amd/csrc/gma_fused.hip"
// // The block's entry in ONE launch (rc_gi_exp): x = a + dw3x3(a) + b_cpe with a = Conv1x1(d1 (192 -> 80)) + b_in  -- the cfg3 net's gma_in followed by
// ConvPosEnc (realcamnet_amd/LiteISP.py `gma_in`; upstream groupmix.py:203-217).  Two launches wrote a (0.67 GB) and read it back with halos; here a block
// of 16 waves owns an 8 x 32 pixel tile: G -- the 1x1 convolution for the tile + 1-pixel halo (10 rows x 40 slots; a lane owns two neighbouring pixels,
// 6 K-steps x 5 row tiles of MFMAs, d1 through a per-image buffer descriptor) -> + bias -> bf16 -> LDS S, channel-planar, zero outside the image;
// D -- depth-wise 3x3 on the matrix cores (banded Toeplitz fragments as in rc_gma_qkv_aggregate; a wave takes channels w, w + 16, ..; one MFMA per kernel
// row covers 8 rows x 32 pixels) + bias + the identity (fp32, one rounding) -> LDS R, pixel-major; St -- R leaves as whole 160-byte token records, 16 bytes
// per lane in address order.  d1 in (1.6 GB + halo rows through L2), x out (0.67 GB).
namespace rc {
namespace gf {

constexpr int GX_WAVES = 16, GX_THREADS = 64 * GX_WAVES, GX_TH = 8, GX_TW = 32, GX_CIN = 192;
constexpr int GX_ROWS = GX_TH + 2, GX_SLOTS = 40, GX_SR = GX_SLOTS * 2, GX_SP = GX_ROWS * GX_SR;   // slot s of a row = pixel x0 - 4 + s; 800-byte channel planes
constexpr int GX_S = kC * GX_SP + 64;
constexpr int GX_PS = 176, GX_RR = GX_TW * GX_PS + 16, GX_R = GX_TH * GX_RR;                    // pixel-major tile: 176-byte pixel slots (160 of data)
constexpr int GX_PAIRS = GX_ROWS * (GX_SLOTS / 2), GX_PPW = (GX_PAIRS + GX_WAVES - 1) / GX_WAVES;   // 200 pixel pairs, 13 per wave
constexpr int GX_OFF_R = GX_S, GX_OFF_W = GX_OFF_R + GX_R, GX_OFF_B = GX_OFF_W + 5 * (GX_CIN / 32) * 1024, GX_LDS = GX_OFF_B + 2 * kC * 4;
static_assert(GX_S % 16 == 0 && GX_R % 16 == 0 && GX_PPW <= 16 && GX_LDS <= 160 * 1024, "gma_in + cpe LDS layout");

struct GxArgs {
    const bf16_t* d1; bf16_t* x;
    int batch, H, W, tiles_x, tiles_y, n_tiles, tiles_per_block;
    const void* w_in; const float* b_in;             // rc_chain_pack_weights_natural(192 -> 80); bias [80] or NULL
    const char* toep; const float* b_cpe;            // rc_dw_toeplitz_pack(taps (9, 80), K = 3, 80 channels); bias [80] or NULL
};

__global__ __launch_bounds__(GX_THREADS) void gi_exp_kernel(GxArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* s_w = lds + GX_OFF_W;
    float* s_b = reinterpret_cast<float*>(lds + GX_OFF_B);                  // b_in [80] | b_cpe [80]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, n = lane & 15, g = lane >> 4;
    for (int i = tid; i < 5 * (GX_CIN / 32) * 64; i += GX_THREADS) reinterpret_cast<uint4*>(s_w)[i] = reinterpret_cast<const uint4*>(a.w_in)[i];
    for (int i = tid; i < kC; i += GX_THREADS) { s_b[i] = a.b_in ? a.b_in[i] : 0.f; s_b[kC + i] = a.b_cpe ? a.b_cpe[i] : 0.f; }
    for (int i = tid; i < (GX_S + GX_R) / 16; i += GX_THREADS) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();

    const __amdgpu_buffer_rsrc_t r_toep = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.toep), 0, 3 * kC * 1024, 0x00020000);
    // this lane's pixel pair of the halo tile (the same for all four g: they hold different K-slices / output channels of the same pixels)
    const int pid = wave * GX_PPW + n;
    const bool has_pair = n < GX_PPW && pid < GX_PAIRS;
    const int prow = pid / (GX_SLOTS / 2), pcol = pid - prow * (GX_SLOTS / 2);
    int o_s = (4 * g) * GX_SP + prow * GX_SR + pcol * 4;                                        // G -> S: + (16 m + j) planes
    int o_ds = n * 0 + (n & 7) * GX_SR + (n >> 3) * 32 + 16 * g;                                // D <- S: + channel plane + kernel row
    int o_di = ((n & 7) + 1) * GX_SR + 8 + (n >> 3) * 32 + 8 * g;                               // D <- S, the identity: the lane's own 4 pixels
    int o_dr = GX_OFF_R + (n & 7) * GX_RR + ((n >> 3) * 16 + 4 * g) * GX_PS;                    // D -> R: + 2 c
    int o_l16 = lane * 16, o_g16 = g * 16;
#define GX_KEEP() asm volatile("" : "+v"(o_s), "+v"(o_ds), "+v"(o_di), "+v"(o_dr), "+v"(o_l16), "+v"(o_g16))

    const int t_begin = blockIdx.x * a.tiles_per_block;
    const int t_end = t_begin + a.tiles_per_block < a.n_tiles ? t_begin + a.tiles_per_block : a.n_tiles;
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        int r_ = tile;
        const int tx = r_ % a.tiles_x; r_ /= a.tiles_x;
        const int ty = r_ % a.tiles_y;
        const int b = r_ / a.tiles_y;
        const int y0 = ty * GX_TH, x0 = tx * GX_TW;
        GX_KEEP();
        {   // ---- G: a = W_in . d1 + b_in on the halo tile -> S
            const __amdgpu_buffer_rsrc_t r_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.d1) + (size_t)b * a.H * a.W * GX_CIN, 0, a.H * a.W * GX_CIN * 2, 0x00020000);
            const int gy = y0 - 1 + prow, gx = x0 - 4 + 2 * pcol;
            const bool row_ok = has_pair && gy >= 0 && gy < a.H;
            const bool ok0 = row_ok && gx >= 0 && gx < a.W, ok1 = row_ok && gx + 1 >= 0 && gx + 1 < a.W;
            const int vo0 = ok0 ? (gy * a.W + gx) * (GX_CIN * 2) + o_g16 : (int)0x80000000, vo1 = ok1 ? (gy * a.W + gx + 1) * (GX_CIN * 2) + o_g16 : (int)0x80000000;
            qt_u32x4 bx[2][GX_CIN / 32];
#pragma unroll
            for (int s = 0; s < GX_CIN / 32; ++s) {
                bx[0][s] = __builtin_amdgcn_raw_buffer_load_b128(r_d, vo0, 64 * s, 0);
                bx[1][s] = __builtin_amdgcn_raw_buffer_load_b128(r_d, vo1, 64 * s, 0);
            }
            // One 16-channel row tile at a time, its six A fragments in registers, the NEXT tile's six loaded meanwhile into the other set: a fragment register is
            // rewritten a whole tile (12 MFMAs) after its last reader was issued.  With one fragment set recycled every K-step (ds_read into the registers the
            // MFMAs just issued were reading) single products came out wrong, differently from run to run, whenever four waves contended for the SIMD's matrix pipe.
            const uint32_t pm = (ok0 ? 0x0000ffffu : 0u) | (ok1 ? 0xffff0000u : 0u);     // outside the image a is ZERO (the depth-wise conv's padding), not b_in
            uint4 af[2][GX_CIN / 32];
#pragma unroll
            for (int s = 0; s < GX_CIN / 32; ++s) af[0][s] = *reinterpret_cast<const uint4*>(s_w + s * 1024 + o_l16);
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                if (m + 1 < 5) {
#pragma unroll
                    for (int s = 0; s < GX_CIN / 32; ++s) af[(m + 1) & 1][s] = *reinterpret_cast<const uint4*>(s_w + ((m + 1) * (GX_CIN / 32) + s) * 1024 + o_l16);
                }
                f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#if RC_GX_ORDER == 3
                asm volatile("" : "+v"(acc0), "+v"(acc1));              // order 3: as order 2, but the chains start from zeroed REGISTERS, not from the inline-constant 0 SrcC form
#endif
#pragma unroll
#if RC_GX_ORDER == 0
                for (int s = 0; s < GX_CIN / 32; ++s) {                 // the two accumulators alternate: each is revisited one MFMA later
                    mma32(af[m & 1][s], make_uint4(bx[0][s][0], bx[0][s][1], bx[0][s][2], bx[0][s][3]), acc0);
                    mma32(af[m & 1][s], make_uint4(bx[1][s][0], bx[1][s][1], bx[1][s][2], bx[1][s][3]), acc1);
                }
#elif RC_GX_ORDER == 2 || RC_GX_ORDER == 3
                for (int s = 0; s < GX_CIN / 32; ++s) {                 // alternating accumulators as in order 0, but nothing may be scheduled between the MFMAs
                    __builtin_amdgcn_sched_barrier(0);
                    mma32(af[m & 1][s], make_uint4(bx[0][s][0], bx[0][s][1], bx[0][s][2], bx[0][s][3]), acc0);
                    __builtin_amdgcn_sched_barrier(0);
                    mma32(af[m & 1][s], make_uint4(bx[1][s][0], bx[1][s][1], bx[1][s][2], bx[1][s][3]), acc1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#else
                for (int s = 0; s < GX_CIN / 32; ++s) { __builtin_amdgcn_sched_barrier(0); mma32(af[m & 1][s], make_uint4(bx[0][s][0], bx[0][s][1], bx[0][s][2], bx[0][s][3]), acc0); __builtin_amdgcn_sched_barrier(0); }     // back to back on ONE accumulator
#pragma unroll
                for (int s = 0; s < GX_CIN / 32; ++s) { mma32(af[m & 1][s], make_uint4(bx[1][s][0], bx[1][s][1], bx[1][s][2], bx[1][s][3]), acc1); __builtin_amdgcn_sched_barrier(0); }
#endif
                const f32x4 bias = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(s_b) + 64 * m + o_g16);
                const f32x4 v0 = acc0 + bias, v1 = acc1 + bias;
                if (has_pair) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<uint32_t*>(lds + o_s + (16 * m + j) * GX_SP) = qa_pk(v0[j], v1[j]) & pm;
                }
            }
        }
        __syncthreads();
        GX_KEEP();
#pragma unroll 1
        for (int i = 0; i < kC / GX_WAVES; ++i) {   // ---- D: x = a + dw3x3(a) + b_cpe for channels wave, wave + 16, ..
            const int c = wave + GX_WAVES * i;
            uint4 T[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const qt_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r_toep, o_l16, (dy * kC + c) * 1024, 0);
                T[dy] = make_uint4(t[0], t[1], t[2], t[3]);
            }
            const char* src = lds + c * GX_SP + o_ds;
            f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) mma32(T[dy], *reinterpret_cast<const uint4*>(src + dy * GX_SR), d);
            const f32x4 idn = up_tail(*reinterpret_cast<const uint2*>(lds + c * GX_SP + o_di));
            const float bc = s_b[kC + c];
            const f32x4 v = (d + bc) + idn;
            const uint32_t p01 = qa_pk(v[0], v[1]), p23 = qa_pk(v[2], v[3]);
            char* dst = lds + o_dr + 2 * c;
            *reinterpret_cast<uint16_t*>(dst) = (uint16_t)(p01 & 0xffffu);
            *reinterpret_cast<uint16_t*>(dst + GX_PS) = (uint16_t)(p01 >> 16);
            *reinterpret_cast<uint16_t*>(dst + 2 * GX_PS) = (uint16_t)(p23 & 0xffffu);
            *reinterpret_cast<uint16_t*>(dst + 3 * GX_PS) = (uint16_t)(p23 >> 16);
        }
        __syncthreads();
        {   // ---- St: 256 pixels x 160 bytes, 16 bytes per lane in address order
            const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(a.x + (size_t)b * a.H * a.W * kC, 0, a.H * a.W * kC * 2, 0x00020000);
#pragma unroll
            for (int k = 0; k < (GX_TH * GX_TW * 10 + GX_THREADS - 1) / GX_THREADS; ++k) {
                const int chunk = tid + k * GX_THREADS;
                const int px = chunk / 10, part = chunk - 10 * px, row = px >> 5, col = px & 31;
                const bool ok = chunk < GX_TH * GX_TW * 10 && y0 + row < a.H && x0 + col < a.W;
                const uint4 v = *reinterpret_cast<const uint4*>(lds + GX_OFF_R + (chunk < GX_TH * GX_TW * 10 ? row * GX_RR + col * GX_PS + part * 16 : 0));
                __builtin_amdgcn_raw_buffer_store_b128(qt_u32x4{v.x, v.y, v.z, v.w}, r_x, ok ? ((y0 + row) * a.W + x0 + col) * (kC * 2) + part * 16 : (int)0x80000000, 0, 0);
            }
        }
    }
}
#undef GX_KEEP

}  // namespace gf
}  // namespace rc

extern "C" int rc_gi_exp(const void* d_d1, const void* d_w_in_natural, const float* d_b_in, const void* d_toeplitz3, const float* d_b_cpe, void* d_x,
                             int batch, int H, int W, void* stream) {
    using namespace rc;
    using namespace rc::gf;
    RC_REQUIRE(d_d1 && d_w_in_natural && d_toeplitz3 && d_x, "rc_gi_exp: null pointer");
    RC_REQUIRE(batch >= 1 && H >= 1 && W >= 1, "rc_gi_exp: bad shape");
    RC_REQUIRE((long long)H * W * GX_CIN * 2 < (1ll << 31), "rc_gi_exp: a 192-channel image must stay below 2 GiB (32-bit buffer offsets)");
    GxArgs a;
    a.d1 = static_cast<const bf16_t*>(d_d1); a.x = static_cast<bf16_t*>(d_x);
    a.batch = batch; a.H = H; a.W = W; a.tiles_x = ceil_div(W, GX_TW); a.tiles_y = ceil_div(H, GX_TH);
    const long long n_tiles = (long long)a.tiles_x * a.tiles_y * batch;
    RC_REQUIRE(n_tiles < (1ll << 31), "rc_gi_exp: too many tiles");
    a.n_tiles = (int)n_tiles;
    a.w_in = d_w_in_natural; a.b_in = d_b_in; a.toep = static_cast<const char*>(d_toeplitz3); a.b_cpe = d_b_cpe;
    int dev = 0;
    RC_HIP_CHECK(hipGetDevice(&dev));
    RC_REQUIRE(dev >= 0 && dev < 64, "rc_gi_exp: device index out of range");
    static int cus[64] = {};
    static bool attr[64] = {};
    if (!cus[dev]) RC_HIP_CHECK(hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev));
    if (!attr[dev]) {
        RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gi_exp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr[dev] = true;
    }
    int blocks = a.n_tiles < cus[dev] ? a.n_tiles : cus[dev];
    a.tiles_per_block = (a.n_tiles + blocks - 1) / blocks;
    blocks = (a.n_tiles + a.tiles_per_block - 1) / a.tiles_per_block;
    hipLaunchKernelGGL(gi_exp_kernel, dim3((unsigned)blocks), dim3(GX_THREADS), GX_LDS, as_stream(stream), a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

