// Host side of the MFMA convolution: config selection, weight/bias re-packing, launch, timing hooks.
#include <mutex>
#include <vector>

#include "conv32_kernel.hpp"

namespace rc {

struct ConvPlan {
    int ck = 0, nt = 0, unit = 0, upt = 0, nu = 0, steps = 0;
    int n_chunks = 0, n_ct = 0, cout_packed = 0;
    int m32 = 0;          // 32x32x16 form (conv32_kernel.hpp): nt counts 32-row tiles, steps counts K16 steps
};

static int g_persist_on = 1;       // rc_debug_set("persist", v): 0 general kernel only, 1 automatic (default), 2 producer/consumer wherever eligible, 3 persistent only
extern int g_dw3_seg16;            // gma.hip
extern int g_dec_lds;              // rans.hip
static int g_pss = 0;              // rc_debug_set("pss", v): 1: single-chunk pixel-shuffle layers (the tail 48 -> 192) take kernel 5 (output staged through LDS, stored by the
                                   // loader waves); 0 (default): kernel 4.  Measured on MI355X at 8 x 1088 x 1920: 3.16-3.29 vs 3.24-3.27 ms (conv_kernel.hpp, kernel 5)
static int g_poison = 0;           // rc_debug_set("lds_poison", 1): every rc_conv2d launch is preceded by rc_debug_poison_lds (bf16 NaNs in all LDS) -- test aid
static int g_thin = 2;             // rc_debug_set("thin", v): kernel 4b where kernel 4 would run -- 1: the layers with thin stages (per-chunk weights <= 20 KB: tiles by LDS-DMA two stages ahead); 2 (default): also the 3x3 layers with <= 36 KB of weights a chunk (one barrier per stage, tile through registers one stage ahead); 0: kernel 4
static int g_auto = 1;             // rc_debug_set("persist_auto", v): single-chunk, single-cout-tile bf16 3x3 layers (48 -> 48, 32 -> 32) on kernel 6 (wave-autonomous strips):
                                   // 0 never, 1 (default) the plain / ReLU / LeakyReLU / +sums forms (1-5 % faster than kernel 2; profiles/r05_power_wall.md), 2 also the residual forms (4-5 % slower)
static int g_sums_compact = 1;     // rc_debug_set("sums_compact", v): 0 = the carried-sums kernels keep the per-tile slot layout (A/B and tests)
static int g_conv32 = 0;           // rc_debug_set("conv32", v): which layers take the 32x32x16 forms (conv32_kernel.hpp).  0 (default) none: the one layer they were
                                   // faster on (48 -> 192 + residual at 544x960x8: 1.25 vs 1.31 ms) now runs on the persistent kernel with the residual prefetched
                                   // (1.13 ms); 4 = that layer family (one-chunk 48 -> 96k NHWC) only; 1 all eligible layers, multi-chunk
                                   // NHWC ones in the staged-output form; 2 / 3 multi-chunk layers in the two-barrier form with 4 / 8 compute waves (A/B experiments)

// Must mirror ConvCfg<> (static_asserts in check_plan_consistency below keep them in lock-step).
static bool make_plan(int cin, int cout, int ksize, int dtype, int out_mode, ConvPlan* p, int cout_tile = 0) {
    if (cin < 1 || cout < 1 || (ksize != 1 && ksize != 3 && ksize != 2 && ksize != 5)) return false;
    if (dtype != RC_F32 && dtype != RC_BF16) return false;
    // ksize 5: the folded tail (rc_tail_fold_weights): one 16-wide cout tile; bf16 with 48 k or 32 k input channels, fp32 with 16 k
    if (ksize == 5 && (cout > 16 || out_mode == RC_OUT_PIXEL_SHUFFLE2 || (dtype == RC_BF16 ? (cin % 48 != 0 && cin % 32 != 0) : cin % 16 != 0))) return false;
    if (out_mode == RC_OUT_PIXEL_SHUFFLE2_NCHW && cout % 4 != 0) return false;
    // ksize 2: the 2x2 window at pixel offsets {-1, 0}^2 = the non-zero taps of a stride-2 3x3 convolution over its space-to-depth map
    // (bf16, Cin a multiple of 16 that is not routed to the 8- / 48- / 80-wide chunk forms; plain NHWC store)
    if (ksize == 2 && (dtype != RC_BF16 || cin % 16 != 0 || out_mode != RC_OUT_NHWC || (cin % 48 == 0 && cin % 64 != 0))) return false;
    p->unit = dtype == RC_F32 ? 4 : 8;
    if (ksize == 5) p->ck = dtype == RC_BF16 ? (cin % 48 == 0 ? 48 : 32) : 16;      // 25 taps: chunks that keep halo tile + weights inside 80 KB of LDS
    else if (dtype == RC_BF16) {
        if (cin <= 8) p->ck = 8;
        else if (ksize == 1 && cin % 80 == 0 && cin % 64 != 0) p->ck = 80;   // GroupMix dims (80, 240, 320 -> 64)
        else if (ksize >= 2 && cin % 64 == 0 && cin > 64) p->ck = 32;         // multi-chunk layers: 32-channel chunks so two
                                                                               // input + two weight buffers fit one CU's LDS
        else if (cin % 64 == 0) p->ck = 64;
        else if (cin % 48 == 0) p->ck = 48;
        else if (ksize == 3 && cin == 32) p->ck = 32;                              // the 32-channel nets (ISPUNet family): ONE chunk -> persistent kernel
        else if (ksize == 3 && cin % 32 == 0 && cin > 64) p->ck = 32;              // 160, 224, 352 ...: half as many stages as 16-channel chunks (the codec's 224 -> 128 slice transforms)
        else p->ck = 16;
    } else {
        p->ck = cin <= 4 ? 4 : 16;
    }
    // cout tile: 64 or 48 channels per block.  48 -> 4x48 (the tail conv) stays on 48-wide tiles so the
    // persistent kernel's LDS footprint (input tile + one cout tile of weights) lets two blocks share a CU.
    if (out_mode == RC_OUT_PIXEL_SHUFFLE2) {
        // a cout tile = one sub-pixel x 16*NT consecutive OUT channels, so stores write whole pixel records
        if (cout % 4 != 0) return false;
        const int cps = cout / 4;
        if (cps % 64 == 0 && !(cps % 48 == 0 && dtype == RC_BF16 && cin == 48)) p->nt = 4;
        else if (cps % 48 == 0) p->nt = 3;
        else if (cps % 64 == 0) p->nt = 4;
        else if (cps % 16 == 0) p->nt = 1;
        else return false;
    } else if (cout % 48 == 0 && dtype == RC_BF16 && cin == 48) p->nt = 3;
    else if (dtype == RC_BF16 && ksize == 3 && cin == 32 && cout == 32) p->nt = 2;   // 32 -> 32: one 32-wide cout tile, three persistent blocks per CU
    else if (ksize == 1 && cout % 80 == 0) p->nt = 5;          // 80-wide cout tiles for the GroupMix Linears
    else if (cout % 64 == 0) p->nt = 4;
    else if (cout % 48 == 0) p->nt = 3;
    else if (dtype == RC_BF16 && ksize == 3 && p->ck == 32 && cout % 32 == 0) p->nt = 2;   // e.g. the codec's 320..640 -> 224 slice transforms: 7 exact 32-wide tiles instead of 14
                                                                                             // 16-wide ones (each re-staging the input): 286 -> 192 us at 576 -> 224, 8 x 72 x 120 (tools/cout_tile_probe.py)
    else p->nt = 1;
    // Caller-chosen cout tile width (rc_conv_desc.cout_tile, the *_ct packers): narrower tiles = more blocks for the general kernel on maps too small to fill the chip
    // with the automatic width (fp32 at 1080p, B = 1: 72-272 blocks of 64 couts for 256 CUs at the 128-channel levels).  Plain stores only; the packed order depends on
    // it, so weights, bias and launch must be given the same value (a width without a kernel instantiation fails at the launch, RC_ERR_UNSUPPORTED).
    if (cout_tile != 0) {
        if (cout_tile < 16 || cout_tile > 80 || cout_tile % 16 != 0 || ksize == 5 || ksize == 2 || out_mode == RC_OUT_PIXEL_SHUFFLE2 || out_mode == RC_OUT_PIXEL_SHUFFLE2_NCHW) return false;
        p->nt = cout_tile / 16;
    }
    // 32x32x16 form: 3x3 bf16 layers whose weights are streamed (several 32-channel chunks, or the one-chunk 48 -> 192 layers) and whose
    // couts fill whole 32-row tiles.  The packed order differs, so the choice depends on the shape and on the `conv32` knob ONLY (not on `persist`:
    // a 32x32x16 layer runs its own kernel in every persist mode) -- weights packed under one conv32 setting must not be used under another
    // (rc_debug_get("conv32") is part of the host mirror's pack-cache key).
    p->m32 = 0;
    if (cout_tile == 0 && dtype == RC_BF16 && ksize == 3 && g_conv32 != 0 && cout <= kPersistMaxCout && out_mode != RC_OUT_NCHW && out_mode != RC_OUT_PIXEL_SHUFFLE2_NCHW) {
        const int cw = out_mode == RC_OUT_PIXEL_SHUFFLE2 ? cout / 4 : cout;          // channels a lane's 16-value run must tile
        const bool all = g_conv32 != 4;
        if (all && p->ck == 32 && cin % 32 == 0 && (out_mode == RC_OUT_PIXEL_SHUFFLE2 ? cw % 32 == 0 : cw % 64 == 0)) { p->m32 = 1; p->nt = 2; p->ck = (g_conv32 == 1 && out_mode == RC_OUT_NHWC) ? 16 : 32; }
        else if (cin == 48 && (out_mode == RC_OUT_PIXEL_SHUFFLE2 ? (all && cw % 48 == 0) : cw % 96 == 0)) { p->m32 = 1; p->nt = 3; p->ck = 48; }
    }
    if (p->m32) {
        p->upt = p->ck / p->unit;
        p->nu = 9 * p->upt;
        p->steps = 9 * (p->ck / 16);
        p->n_chunks = cin / p->ck;
        p->n_ct = cout / (32 * p->nt);
        p->cout_packed = cout;
        return true;
    }
    p->upt = p->ck / p->unit;
    p->nu = ksize * ksize * p->upt;
    p->steps = unit_map_steps(p->upt, ksize * ksize);
    p->n_chunks = ceil_div(cin, p->ck);
    p->n_ct = ceil_div(cout, 16 * p->nt);
    p->cout_packed = p->n_ct * 16 * p->nt;
    return true;
}

// packed cout index j -> conv output channel (or -1 for padding rows)
static int packed_to_cout(const ConvPlan& p, int cout, int out_mode, int j) {
    if (p.m32) return c32_packed_to_cout(p.nt, out_mode, j);
    if (out_mode == RC_OUT_PIXEL_SHUFFLE2) {
        // cout tile ct = cb*4 + sub-pixel; inside it packed index = out channel offset within block cb
        const int tile = 16 * p.nt;
        const int ct = j / tile, within = j % tile;
        const int cb = ct >> 2, sub = ct & 3;            // sub = 2i + j of nn.PixelShuffle(2)
        return 4 * (cb * tile + within) + sub;            // conv channel 4c + sub
    }
    return j < cout ? j : -1;
}

// wino.hip: Winograd F(2x2, 3x3) (rc_conv_desc.algo == 1)
int wino_conv(const rc_conv_desc* d, hipStream_t stream);
int wino_sum_slots(int H, int W, int cout);
void wino_set_nnt(int v);
int wino_get_nnt();
bool wino_supported(const rc_conv_desc* d, std::string* why);

static int g_dbg_flags = 0;
static int g_pair_impl = 0;         // rc_debug_set("pair_impl", v): 0 = the faster one per form (gated: the first pair kernel; else pair2), 1 = the first pair kernel
                                   // (weights in LDS, 8 + 4 waves), 2 = pair2 (weights in registers, two teams of four waves)
static long long* g_dbg_ptr = nullptr;   // rc_debug_set_ptr("conv_phase_timing", device buffer of >= 512 int64)
static std::mutex g_prof_mu;
static bool g_prof_on = false;
struct ProfRec { hipEvent_t e0, e1; double flops, bytes; int cin, cout, ksize; };
static std::vector<ProfRec> g_prof;

long long* conv_dbg_ptr() { return g_dbg_ptr; }
int conv_pair_impl() { return g_pair_impl; }

// HIP-event bracket around one conv launch on its own stream (rc_prof_enable); shared with conv_pair.hip
void conv_prof_begin(double flops, hipStream_t stream, void** token, double bytes, int cin, int cout, int ksize) {
    *token = nullptr;
    { std::lock_guard<std::mutex> lk(g_prof_mu); if (!g_prof_on) return; }
    ProfRec* rec = new ProfRec{};
    rec->flops = flops; rec->bytes = bytes; rec->cin = cin; rec->cout = cout; rec->ksize = ksize;
    if (hipEventCreate(&rec->e0) != hipSuccess || hipEventCreate(&rec->e1) != hipSuccess) { delete rec; return; }
    (void)hipEventRecord(rec->e0, stream);
    *token = rec;
}
void conv_prof_end(void* token, hipStream_t stream) {
    if (!token) return;
    ProfRec* rec = static_cast<ProfRec*>(token);
    (void)hipEventRecord(rec->e1, stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(*rec);
    delete rec;
}

}  // namespace rc

using namespace rc;

extern "C" {

size_t rc_conv_packed_bytes_ct(int cin, int cout, int ksize, int dtype, int out_mode, int cout_tile) {
    ConvPlan p;
    if (!make_plan(cin, cout, ksize, dtype, out_mode, &p, cout_tile)) { set_error("rc_conv_packed_bytes: bad shape"); return 0; }
    return (size_t)p.n_ct * p.n_chunks * p.steps * p.nt * 1024;
}
size_t rc_conv_packed_bytes(int cin, int cout, int ksize, int dtype, int out_mode) { return rc_conv_packed_bytes_ct(cin, cout, ksize, dtype, out_mode, 0); }

int rc_conv_packed_cout_ct(int cin, int cout, int ksize, int dtype, int out_mode, int cout_tile) {
    ConvPlan p;
    if (!make_plan(cin, cout, ksize, dtype, out_mode, &p, cout_tile)) return fail(RC_ERR_INVALID, "rc_conv_packed_cout: bad shape");
    return p.cout_packed;
}
int rc_conv_packed_cout(int cin, int cout, int ksize, int dtype, int out_mode) { return rc_conv_packed_cout_ct(cin, cout, ksize, dtype, out_mode, 0); }

int rc_conv_pack_weights(const float* w, int cin, int cout, int ksize, int dtype, int out_mode, void* dst) {
    return rc_conv_pack_weights_ct(w, cin, cout, ksize, dtype, out_mode, 0, dst);
}
int rc_conv_pack_bias(const float* bias, int cin, int cout, int ksize, int dtype, int out_mode, float* dst) {
    return rc_conv_pack_bias_ct(bias, cin, cout, ksize, dtype, out_mode, 0, dst);
}

int rc_conv_pack_weights_ct(const float* w, int cin, int cout, int ksize, int dtype, int out_mode, int cout_tile, void* dst) {
    ConvPlan p;
    RC_REQUIRE(w && dst, "rc_conv_pack_weights: null pointer");
    RC_REQUIRE(make_plan(cin, cout, ksize, dtype, out_mode, &p, cout_tile), "rc_conv_pack_weights: bad shape");
    const int kk = ksize * ksize;
    char* out = static_cast<char*>(dst);
    if (p.m32) {
        // layout: [ct][chunk][K16 step][row tile t][lane 0..63][8 bf16].  Lane l supplies MFMA row m = l & 31, channels 8*(l >> 5) + e of
        // the step's 16; D row m comes out in lane half (m >> 2) & 1, register (m & 3) + 4 * (m >> 3), so row m carries packed channel
        // 16 * half + register: a lane's 16 registers are 16 consecutive channels.
        const int spt = p.ck / 16;
        for (int ct = 0; ct < p.n_ct; ++ct)
            for (int chunk = 0; chunk < p.n_chunks; ++chunk)
                for (int s = 0; s < p.steps; ++s)
                    for (int t = 0; t < p.nt; ++t)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int m = lane & 31, kg = lane >> 5;
                            const int c = 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3);
                            const int co = packed_to_cout(p, cout, out_mode, ct * 32 * p.nt + 32 * t + c);
                            const int tap = s / spt, ci0 = chunk * p.ck + (2 * (s % spt) + kg) * 8;
                            for (int e = 0; e < 8; ++e) {
                                *reinterpret_cast<uint16_t*>(out) = host_f32_to_bf16(w[((size_t)co * cin + ci0 + e) * kk + tap]);
                                out += 2;
                            }
                        }
        return RC_OK;
    }
    // layout: [ct][chunk][step][nt][lane 0..63][UNIT elements]   (16 bytes per lane)
    for (int ct = 0; ct < p.n_ct; ++ct)
        for (int chunk = 0; chunk < p.n_chunks; ++chunk)
            for (int s = 0; s < p.steps; ++s)
                for (int nt = 0; nt < p.nt; ++nt)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int m = lane & 15, qk = lane >> 4;
                        int tap = 0, cu = 0;
                        const bool live = unit_map(p.upt, kk, s, qk, tap, cu);
                        // MFMA row m of cout tile nt lands in lane group m>>2, register m&3
                        const int j = ct * 16 * p.nt + (m >> 2) * (4 * p.nt) + nt * 4 + (m & 3);
                        const int co = packed_to_cout(p, cout, out_mode, j);
                        for (int e = 0; e < p.unit; ++e) {
                            float val = 0.f;
                            if (live && co >= 0 && co < cout) {
                                const int ci = chunk * p.ck + cu * p.unit + e;
                                if (ci < cin) val = w[((size_t)co * cin + ci) * kk + tap];
                            }
                            if (dtype == RC_F32) {
                                *reinterpret_cast<float*>(out) = val; out += 4;
                            } else {
                                *reinterpret_cast<uint16_t*>(out) = host_f32_to_bf16(val); out += 2;
                            }
                        }
                    }
    return RC_OK;
}

int rc_conv_pack_bias_ct(const float* bias, int cin, int cout, int ksize, int dtype, int out_mode, int cout_tile, float* dst) {
    ConvPlan p;
    RC_REQUIRE(dst, "rc_conv_pack_bias: null pointer");
    RC_REQUIRE(make_plan(cin, cout, ksize, dtype, out_mode, &p, cout_tile), "rc_conv_pack_bias: bad shape");
    for (int j = 0; j < p.cout_packed; ++j) {
        const int co = packed_to_cout(p, cout, out_mode, j);
        dst[j] = (bias && co >= 0 && co < cout) ? bias[co] : 0.f;
    }
    return RC_OK;
}

size_t rc_conv_desc_size(void) { return sizeof(rc_conv_desc); }

int rc_tail_fold_weights(const float* w1, const float* b1, const float* w2, const float* b2, int c, int o, float* wc, float* bc) {
    RC_REQUIRE(w1 && w2 && wc && bc && c >= 1 && o >= 1, "rc_tail_fold_weights: bad arguments");
    // out[o][2y+i][2x+j] = b2[o] + sum_{cc,dy,dx} w2[o][cc][dy][dx] * T[cc][2y+i+dy-1][2x+j+dx-1],  T[cc][2y'+i'][2x'+j'] = t[4cc+2i'+j'][y'][x'],
    // t = conv3x3(x; w1, b1): substitute, collect the taps of x around (y, x).  With a = i + dy - 1: y' = y + floor(a / 2), i' = a mod 2.
    std::vector<double> acc((size_t)4 * o * c * 25, 0.0), bacc((size_t)4 * o, 0.0);
    for (int oo = 0; oo < o; ++oo)
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) {
                const int row = oo * 4 + 2 * i + j;
                bacc[row] = b2 ? (double)b2[oo] : 0.0;
                for (int dy = 0; dy < 3; ++dy)
                    for (int dx = 0; dx < 3; ++dx) {
                        const int a = i + dy - 1, b = j + dx - 1;
                        const int fy = a < 0 ? -1 : a >> 1, ii = a & 1, fx = b < 0 ? -1 : b >> 1, jj = b & 1;
                        for (int cc = 0; cc < c; ++cc) {
                            const int m = 4 * cc + 2 * ii + jj;
                            const double v2 = w2[(((size_t)oo * c + cc) * 3 + dy) * 3 + dx];
                            if (b1) bacc[row] += v2 * (double)b1[m];
                            for (int k = 0; k < c; ++k)
                                for (int ey = 0; ey < 3; ++ey)
                                    for (int ex = 0; ex < 3; ++ex)
                                        acc[(((size_t)row * c + k) * 5 + (fy + ey + 1)) * 5 + (fx + ex + 1)] += v2 * (double)w1[(((size_t)m * c + k) * 3 + ey) * 3 + ex];
                        }
                    }
            }
    for (size_t n = 0; n < acc.size(); ++n) wc[n] = (float)acc[n];
    for (size_t n = 0; n < bacc.size(); ++n) bc[n] = (float)bacc[n];
    return RC_OK;
}

int rc_conv_sum_tiles(int height, int width) { return 4 * ceil_div(height, kTH) * ceil_div(width, kTW); }  // one slot per wave

int rc_debug_set(const char* key, int value) {
    RC_REQUIRE(key != nullptr, "rc_debug_set: null key");
    if (std::string(key) == "persist") { g_persist_on = value < 0 ? 0 : (value > 3 ? 3 : value); return RC_OK; }
    if (std::string(key) == "conv_flags") { g_dbg_flags = value; return RC_OK; }
    if (std::string(key) == "pair_impl") { g_pair_impl = value < 0 || value > 2 ? 0 : value; return RC_OK; }
    if (std::string(key) == "dec_lds") { g_dec_lds = value != 0; return RC_OK; }
    if (std::string(key) == "dw3_seg16") { g_dw3_seg16 = value != 0; return RC_OK; }
    if (std::string(key) == "pss") { g_pss = value != 0; return RC_OK; }
    if (std::string(key) == "sums_compact") { g_sums_compact = value != 0; return RC_OK; }
    if (std::string(key) == "persist_auto") { g_auto = value < 0 ? 0 : (value > 2 ? 2 : value); return RC_OK; }
    if (std::string(key) == "thin") { g_thin = value < 0 ? 0 : (value > 2 ? 2 : value); return RC_OK; }
    if (std::string(key) == "lds_poison") { g_poison = value != 0; return RC_OK; }
    if (std::string(key) == "conv32") { g_conv32 = value < 0 ? 0 : (value > 4 ? 4 : value); return RC_OK; }
    if (std::string(key) == "wino_nnt") { wino_set_nnt(value); return RC_OK; }
    return fail(RC_ERR_INVALID, std::string("rc_debug_set: unknown key ") + key);
}

int rc_debug_get(const char* key) {
    if (key == nullptr) return -1;
    if (std::string(key) == "persist") return g_persist_on;
    if (std::string(key) == "conv32") return g_conv32;
    if (std::string(key) == "conv_flags") return g_dbg_flags;
    if (std::string(key) == "pss") return g_pss;
    if (std::string(key) == "sums_compact") return g_sums_compact;
    if (std::string(key) == "persist_auto") return g_auto;
    if (std::string(key) == "thin") return g_thin;
    if (std::string(key) == "lds_poison") return g_poison;
    if (std::string(key) == "wino_nnt") return wino_get_nnt();
    return -1;
}

int rc_debug_set_ptr(const char* key, void* p) {
    RC_REQUIRE(key != nullptr, "rc_debug_set_ptr: null key");
    if (std::string(key) == "conv_phase_timing") { g_dbg_ptr = static_cast<long long*>(p); return RC_OK; }
    return fail(RC_ERR_INVALID, std::string("rc_debug_set_ptr: unknown key ") + key);
}

int rc_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof.clear();
    g_prof_on = on != 0;
    return RC_OK;
}

int rc_prof_collect(int64_t* n_launches, double* total_ms, double* total_flops) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double ms = 0.0, fl = 0.0;
    for (auto& r : g_prof) {
        RC_HIP_CHECK(hipEventSynchronize(r.e1));
        float t = 0.f;
        RC_HIP_CHECK(hipEventElapsedTime(&t, r.e0, r.e1));
        ms += t; fl += r.flops;
    }
    if (n_launches) *n_launches = (int64_t)g_prof.size();
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    return RC_OK;
}

int rc_prof_collect_rows(rc_prof_row* rows, int max_rows, int* n_rows) {
    RC_REQUIRE(rows != nullptr && n_rows != nullptr && max_rows >= 1, "rc_prof_collect_rows: bad arguments");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    for (auto& r : g_prof) {
        RC_HIP_CHECK(hipEventSynchronize(r.e1));
        float t = 0.f;
        RC_HIP_CHECK(hipEventElapsedTime(&t, r.e0, r.e1));
        int i = 0;
        while (i < n && !(rows[i].cin == r.cin && rows[i].cout == r.cout && rows[i].ksize == r.ksize)) ++i;
        if (i == n) {
            if (n == max_rows) continue;                    // table full: the launch is still in rc_prof_collect's totals
            rows[n] = rc_prof_row{r.cin, r.cout, r.ksize, 0, 0.0, 0.0, 0.0};
            ++n;
        }
        rows[i].launches += 1; rows[i].ms += t; rows[i].flops += r.flops; rows[i].bytes += r.bytes;
    }
    *n_rows = n;
    return RC_OK;
}

// Validation + launch arguments of one rc_conv2d call; shared with rc_conv_sum_slots (which asks the launcher what it WOULD do with them).
static int conv_build_args(const rc_conv_desc* d, ConvPlan& p, ConvArgs& a, size_t& es_out) {
    RC_REQUIRE(d != nullptr, "rc_conv2d: null desc");
    const bool dwt = d->out_mode == RC_OUT_NHWC_DWT;      // conv -> Haar DWT: packed, planned and validated as the RC_OUT_NHWC layer it replaces
    RC_REQUIRE(make_plan(d->cin, d->cout, d->ksize, d->dtype, dwt ? (int)RC_OUT_NHWC : d->out_mode, &p, d->cout_tile), "rc_conv2d: unsupported cin/cout/ksize/dtype/cout_tile");
    RC_REQUIRE(d->batch >= 1 && d->height >= 1 && d->width >= 1, "rc_conv2d: empty tensor");
    RC_REQUIRE(d->batch <= 65535, "rc_conv2d: batch > 65535");
    RC_REQUIRE(d->in0 && d->wpacked && d->out, "rc_conv2d: null in0/wpacked/out");
    RC_REQUIRE(d->out_mode >= RC_OUT_NHWC && d->out_mode <= RC_OUT_NHWC_DWT, "rc_conv2d: bad out_mode");
    if (dwt) {
        RC_REQUIRE(d->height % 2 == 0 && d->width % 2 == 0, "rc_conv2d: RC_OUT_NHWC_DWT needs even height and width");
        RC_REQUIRE(d->out_dtype == d->dtype && reinterpret_cast<uintptr_t>(d->out) % 16 == 0, "rc_conv2d: RC_OUT_NHWC_DWT: out_dtype must equal dtype, out 16-byte aligned");
        RC_REQUIRE(!d->mul_plus1 && !d->film_scale && !d->out_scale && !d->chan_sums && !d->in_gate && !d->in1 && !d->in_store && !d->src_h && !d->src_w && !d->cout_tile,
                   "rc_conv2d: RC_OUT_NHWC_DWT excludes mul_plus1 / film / out_scale / chan_sums / gated input / src_h / cout_tile");
        if (d->dtype != RC_BF16 || d->ksize != 3 || p.m32 || !(d->act == RC_ACT_NONE || ((d->act == RC_ACT_RELU || d->act == RC_ACT_LEAKY) && !d->residual)))
            return fail(RC_ERR_UNSUPPORTED, "rc_conv2d: RC_OUT_NHWC_DWT is a bf16 3x3 form with act NONE / RELU / LEAKY, or act NONE + residual");
    }
    RC_REQUIRE(d->act >= RC_ACT_NONE && d->act <= RC_ACT_RELU_POST, "rc_conv2d: bad act");
    RC_REQUIRE(d->act != RC_ACT_RELU_POST || (d->residual != nullptr && d->mul_plus1 == nullptr && d->film_scale == nullptr && d->chan_sums == nullptr),
               "rc_conv2d: RC_ACT_RELU_POST is relu(conv + residual): needs residual, excludes film / mul_plus1 / chan_sums");
    if (d->in_gate) RC_REQUIRE(d->in1 != nullptr, "rc_conv2d: in_gate needs in1 (the skip tensor)");
    RC_REQUIRE((d->film_scale == nullptr) == (d->film_shift == nullptr), "rc_conv2d: film_scale/film_shift must come together");
    const bool full_tiles = d->cout == p.cout_packed;
    if (d->mul_plus1 || d->residual)   // a lane's 4-channel group must be wholly inside or outside [0, cout)
        RC_REQUIRE((full_tiles || d->cout % 4 == 0) && (d->out_mode == RC_OUT_NHWC || (dwt && !d->mul_plus1)), "rc_conv2d: mul_plus1/residual need cout % 4 == 0 and RC_OUT_NHWC (residual: or RC_OUT_NHWC_DWT)");
    if (d->chan_sums) RC_REQUIRE(d->out_mode == RC_OUT_NHWC, "rc_conv2d: chan_sums needs RC_OUT_NHWC");
    if (d->out_scale) {
        RC_REQUIRE(d->out_mode == RC_OUT_NHWC && !p.m32, "rc_conv2d: out_scale needs RC_OUT_NHWC and a layer outside the 32x32x16 forms");
        RC_REQUIRE(reinterpret_cast<uintptr_t>(d->out_scale) % 4 == 0, "rc_conv2d: out_scale must be 4-byte aligned");
    }
    const size_t es = dtype_size(d->dtype);
    if (dwt) {
        // validated above; the store is whole 16-byte pieces of (H / 2, W / 2, 4 cout) pixel records
    } else if (d->out_mode == RC_OUT_NHWC) {
        RC_REQUIRE(d->out_dtype == d->dtype, "rc_conv2d: out_dtype must equal dtype for RC_OUT_NHWC");
        RC_REQUIRE(!full_tiles || (d->cout * es) % 8 == 0, "rc_conv2d: cout*elem_size must be a multiple of 8 bytes for RC_OUT_NHWC");
        RC_REQUIRE(reinterpret_cast<uintptr_t>(d->out) % 16 == 0, "rc_conv2d: out must be 16-byte aligned");
        if (p.nt == 4 && d->dtype == RC_BF16) RC_REQUIRE(d->cout % 8 == 0, "rc_conv2d: cout % 8");
    } else if (d->out_mode == RC_OUT_PIXEL_SHUFFLE2) {
        RC_REQUIRE(d->out_dtype == d->dtype, "rc_conv2d: out_dtype must equal dtype for RC_OUT_PIXEL_SHUFFLE2");
        RC_REQUIRE(full_tiles && ((d->cout / 4) * es) % 8 == 0, "rc_conv2d: pixel-shuffle store needs (cout/4)*elem_size % 8 == 0");
        RC_REQUIRE(!d->film_scale, "rc_conv2d: film not supported with pixel-shuffle store");
        RC_REQUIRE(reinterpret_cast<uintptr_t>(d->out) % 16 == 0, "rc_conv2d: out must be 16-byte aligned");
    } else if (d->out_mode == RC_OUT_PIXEL_SHUFFLE2_NCHW) {
        RC_REQUIRE(d->out_h >= 1 && d->out_h <= 2 * d->height && d->out_w >= 1 && d->out_w <= 2 * d->width, "rc_conv2d: bad pixel-shuffle NCHW crop");
        RC_REQUIRE(d->out_dtype == RC_F32 || d->out_dtype == RC_BF16, "rc_conv2d: bad out_dtype");
        RC_REQUIRE(!d->chan_sums, "rc_conv2d: chan_sums needs RC_OUT_NHWC");
        RC_REQUIRE((double)d->out_h * d->out_w * (d->cout / 4) * 4.0 < 2147483647.0, "rc_conv2d: one output image must be < 2 GiB");
    } else {
        RC_REQUIRE(d->out_h >= 1 && d->out_h <= d->height && d->out_w >= 1 && d->out_w <= d->width, "rc_conv2d: bad NCHW crop");
        RC_REQUIRE(d->out_dtype == RC_F32 || d->out_dtype == RC_BF16, "rc_conv2d: bad out_dtype");
    }
    {   // buffer descriptors use signed 32-bit byte offsets inside one image
        const double lim = 2147483647.0;
        RC_REQUIRE((double)d->height * d->width * d->cin * es < lim, "rc_conv2d: one input image must be < 2 GiB");
        RC_REQUIRE((double)d->height * d->width * d->cout * 4.0 < lim, "rc_conv2d: one output image must be < 2 GiB");
    }
    const bool fold = d->src_h != 0 || d->src_w != 0;      // ksize 2 reading the stride-2 convolution's own input (no space-to-depth map)
    if (fold) {
        RC_REQUIRE(d->ksize == 2 && d->cin % 4 == 0 && (d->cin / 4) % p.ck == 0, "rc_conv2d: src_h/src_w need ksize 2 and cin / 4 a whole number of Cin chunks");
        RC_REQUIRE(d->src_h >= 1 && d->src_w >= 1 && d->height == (d->src_h + 1) / 2 && d->width == (d->src_w + 1) / 2,
                   "rc_conv2d: height/width must be ceil(src_h / 2), ceil(src_w / 2)");
        RC_REQUIRE(d->in_gate == nullptr && d->in1 == nullptr && d->in_store == nullptr, "rc_conv2d: src_h/src_w exclude the gated input");
        RC_REQUIRE(reinterpret_cast<uintptr_t>(d->in0) % 16 == 0, "rc_conv2d: src_h/src_w need a 16-byte aligned in0 (the scalar staging path does not gather)");
        RC_REQUIRE((double)d->src_h * d->src_w * (d->cin / 4) * es < 2147483647.0, "rc_conv2d: one input image must be < 2 GiB");
    }
    if (d->residual) RC_REQUIRE(reinterpret_cast<uintptr_t>(d->residual) % 16 == 0, "rc_conv2d: residual must be 16-byte aligned");
    if (d->mul_plus1) RC_REQUIRE(reinterpret_cast<uintptr_t>(d->mul_plus1) % 16 == 0, "rc_conv2d: mul_plus1 must be 16-byte aligned");

    a = ConvArgs{};
    a.batch = d->batch; a.H = d->height; a.W = d->width; a.cin = d->cin; a.cout = d->cout;
    a.n_chunks = p.n_chunks; a.n_ct = p.n_ct;
    a.tiles_x = ceil_div(d->width, kTW); a.tiles_y = ceil_div(d->height, kTH);
    if (p.m32) RC_REQUIRE(!d->film_scale || (reinterpret_cast<uintptr_t>(d->film_scale) % 16 == 0 && reinterpret_cast<uintptr_t>(d->film_shift) % 16 == 0),
                          "rc_conv2d: film vectors must be 16-byte aligned");
    auto aligned16 = [](const void* q) { return q == nullptr || reinterpret_cast<uintptr_t>(q) % 16 == 0; };
    a.cin_vec_ok = (d->cin % p.unit == 0) && aligned16(d->in0) && aligned16(d->in1) && aligned16(d->in_store) &&
                   (d->in_gate == nullptr || reinterpret_cast<uintptr_t>(d->in_gate) % 4 == 0);
    a.cin_chunk_ok = d->cin % p.ck == 0;
    a.fold2 = fold ? 1 : 0; a.src_H = d->src_h; a.src_W = d->src_w; a.cfold = d->cin / 4;
    a.in0 = d->in0; a.in1 = d->in1; a.in_gate = d->in_gate; a.in_store = d->in_store;
    a.wpacked = d->wpacked; a.bias = d->bias;
    a.film_scale = d->film_scale; a.film_shift = d->film_shift;
    a.act = d->act; a.act_slope = d->act_slope;
    a.mul_plus1 = d->mul_plus1; a.residual = d->residual; a.out_scale = d->out_scale;
    a.out = d->out; a.out_mode = d->out_mode; a.out_dtype = d->out_dtype; a.out_h = d->out_h; a.out_w = d->out_w;
    a.chan_sums = d->chan_sums; a.cout_packed = p.cout_packed;
    a.sum_slots = d->chan_sums_slots > 0 ? d->chan_sums_slots : 4 * ceil_div(d->height, kTH) * ceil_div(d->width, kTW);   // 0: the legacy count (rc_conv_sum_tiles)
    a.sums_compact = 0; a.query = nullptr; a.sums_compact_ok = g_sums_compact;
    {   // epilogue feature mask (ConvDev::EP_*); anything outside the compiled set takes the generic epilogue
        int key = (d->act == RC_ACT_RELU ? 1 : 0) | (d->act == RC_ACT_LEAKY ? 2 : 0) | (d->film_scale ? 4 : 0) |
                  (d->mul_plus1 ? 8 : 0) | (d->residual ? 16 : 0) | (d->chan_sums ? 32 : 0) | (d->out_scale ? 64 : 0);
        bool fast = full_tiles && d->out_mode != RC_OUT_NCHW && d->out_mode != RC_OUT_PIXEL_SHUFFLE2_NCHW && d->act != RC_ACT_GELU && d->act != RC_ACT_RELU_POST;
        if (d->act == RC_ACT_LEAKY) fast = fast && d->act_slope >= 0.f && d->act_slope <= 1.f;
        if (d->film_scale)
            fast = fast && d->cout % 4 == 0 && reinterpret_cast<uintptr_t>(d->film_scale) % 16 == 0 && reinterpret_cast<uintptr_t>(d->film_shift) % 16 == 0;
        if (d->out_scale) fast = fast && d->cout % 4 == 0 && reinterpret_cast<uintptr_t>(d->out_scale) % 16 == 0;
        switch (key) { case 0: case 1: case 2: case 16: case 32: case 8: case 6: case 33: case 34: case 80: break; default: fast = false; }
        a.ep_key = fast ? key : -1;
        // a cout tile of a single-chunk pixel-shuffle layer with cout = 4 cout tiles is one sub-pixel of every pixel: kernel 5
        a.pss = g_pss && d->out_mode == RC_OUT_PIXEL_SHUFFLE2 && !p.m32 && p.n_chunks == 1 && p.n_ct == 4 && d->cout == 4 * 16 * p.nt &&
                d->dtype == RC_BF16 && d->in_gate == nullptr && fast && (key == 0 || key == 1 || key == 2);
    }
    {
        const int num_cus = device_cu_count();          // per device (common.hpp)
        a.num_cus = num_cus;
        a.persist_ok = g_persist_on;
        a.auto_impl = g_auto;
        a.thin = g_thin;
        a.dbg = g_dbg_ptr;
        a.dbg_flags = g_dbg_flags;
        a.td = make_tile_decode(a.tiles_x, a.tiles_y, kBandRows);
        a.td_wsm = make_tile_decode(a.tiles_x, (a.H + kWsmTH - 1) / kWsmTH, kBandRows);
        a.div_n_ct = make_magic(a.n_ct);
    }
    es_out = es;
    return RC_OK;
}

int rc_conv2d(const rc_conv_desc* d, void* stream_) {
    RC_REQUIRE(d != nullptr, "rc_conv2d: null desc");
    RC_REQUIRE(d->algo == 0 || d->algo == 1, "rc_conv2d: algo must be 0 (implicit GEMM) or 1 (Winograd F(2x2,3x3))");
    if (d->algo == 1) {
        if (g_poison) { if (int e = rc_debug_poison_lds(0x7FC07FC0u, stream_)) return e; }
        return wino_conv(d, as_stream(stream_));
    }
    ConvPlan p;
    ConvArgs a;
    size_t es = 0;
    if (int e = conv_build_args(d, p, a, es)) return e;
    hipStream_t stream = as_stream(stream_);
    if (g_poison) { if (int e = rc_debug_poison_lds(0x7FC07FC0u, stream_)) return e; }
    void* tok = nullptr;
    // algorithmic FLOPs: a ksize-2 launch is a stride-2 3x3 convolution over its space-to-depth map -- 9 of its 16 (tap, phase) blocks are real
    const double taps = d->ksize == 2 ? 9.0 / 4.0 : (double)d->ksize * d->ksize;
    // algorithmic bytes: every map the launch must touch once (input, output, residual / x(lsc+1) operand, the gated form's skip and materialised input)
    const double px = (double)d->batch * d->height * d->width;
    const double osz = (d->out_mode == RC_OUT_NCHW || d->out_mode == RC_OUT_PIXEL_SHUFFLE2_NCHW) ? (d->out_dtype == RC_F32 ? 4.0 : 2.0) : (double)es;
    const double bytes = px * d->cin * es * (1.0 + (d->in1 ? 1.0 : 0.0) + (d->in_store ? 1.0 : 0.0)) + px * d->cout * osz +
                         px * d->cout * es * ((d->residual ? 1.0 : 0.0) + (d->mul_plus1 ? 1.0 : 0.0));
    conv_prof_begin(2.0 * px * (double)d->cin * d->cout * taps, stream, &tok, bytes, d->cin, d->cout, d->ksize);
    const int rcode = p.m32 ? (p.ck == 16 ? conv32_ck16(0, a, stream) : p.ck == 32 ? conv32_ck32(g_conv32 == 3 ? 1 : 0, a, stream) : conv32_ck48(0, a, stream))
                            : dispatch_conv(d->dtype == RC_BF16, d->ksize, p.ck, p.nt, a, stream);
    conv_prof_end(tok, stream);
    return rcode;
}

int rc_conv_sum_slots(const rc_conv_desc* d) {
    if (d != nullptr && d->algo == 1) return wino_supported(d, nullptr) ? wino_sum_slots(d->height, d->width, d->cout) : -1;
    ConvPlan p;
    ConvArgs a;
    size_t es = 0;
    if (conv_build_args(d, p, a, es) != RC_OK) return -1;
    const int legacy = 4 * ceil_div(d->height, kTH) * ceil_div(d->width, kTW);
    if (p.m32) return legacy;                                  // the 32x32x16 forms write the per-tile layout
    int q[2] = {0, legacy};
    a.query = q;                                               // the launcher reports the branch it would take and launches nothing
    if (dispatch_conv(d->dtype == RC_BF16, d->ksize, p.ck, p.nt, a, nullptr) != RC_OK) return -1;
    return q[1];
}

}  // extern "C"
