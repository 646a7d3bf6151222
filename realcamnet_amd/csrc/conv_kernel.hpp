// Implicit-GEMM KxK convolution on MFMA for gfx950 (wave64).  See DESIGN.md section 4.
//
// GEMM view:  D[cout][pixel] = sum_k  W[cout][k] * X[k][pixel],   k = (tap, cin)
//   A operand = packed weights  (MFMA rows    = 16 couts of one cout tile)
//   B operand = activations     (MFMA columns = 16 consecutive pixels of one image row)
// so each lane ends up holding 4 consecutive couts of ONE pixel per accumulator tile and the
// NHWC store is contiguous per lane.
//
// One 16-byte "unit" = UNIT consecutive input channels of one tap at one pixel (8 bf16 / 4 fp32).
// One "step" = 4 units (one per 16-lane group q = lane>>4):
//   bf16: 1 x v_mfma_f32_16x16x32_bf16 (K = 32 channels-of-taps)
//   fp32: 4 x v_mfma_f32_16x16x4_f32   (element j of each unit in MFMA j; K = 4 each)
// Which (tap, channel-unit) a lane group handles in a step is fixed by unit_map() below, shared with
// the host-side weight packer.  For Cin chunks of 6 units (48 bf16 channels -- the flagship width) the
// map keeps every LDS address "per-lane constant + compile-time immediate" (no address table).
//
// Block = 256 threads = 4 waves; block tile = 8 rows x 32 cols of pixels x (16*NT) couts;
// wave w owns rows 2w, 2w+1 (4 pixel tiles of 16) x NT cout tiles -> 4*NT accumulator tiles.
// The input halo tile (10 x 34 pixels x CK channels) is staged once per Cin chunk in LDS with a
// pixel stride == 32 (mod 64) bytes, which makes the 16-lane-group ds_read_b128 conflict-free.
//
// Memory instructions are buffer loads/stores against per-image descriptors with 32-bit offsets:
// zero padding, ragged edges and crops are hardware bounds checks (offset = kOOB -> load 0 / store
// dropped), so there are no divergent branches in the staging or store paths.
#pragma once
#include "common.hpp"

namespace rc {

// Unsigned division by a launch constant as multiply-high + shift: scalar-ALU only.  (The float-reciprocal form it
// replaces went SGPR -> VALU convert/multiply/convert -> readfirstlane three times per tile decode.)
// Exact for 0 <= n < 2^24 and 1 <= d < 2^24: m = ceil(2^(31+L) / d), L = floor(log2 d), q = mulhi(n, m) >> (L - 1).
struct MagicDiv { unsigned m, sh; };
inline MagicDiv make_magic(int d) {
    if (d <= 1) return MagicDiv{0u, 0u};
    int L = 0;
    while ((2 << L) <= d) ++L;
    const unsigned long long m = ((1ull << (31 + L)) + (unsigned long long)d - 1) / (unsigned long long)d;
    return MagicDiv{(unsigned)m, (unsigned)(L - 1)};
}
__device__ __forceinline__ int magic_div(int n, const MagicDiv& d) {
    return d.m ? (int)(__umulhi((unsigned)n, d.m) >> d.sh) : n;
}
struct TileDecode { MagicDiv sp_total, band, last_rows; };
inline TileDecode make_tile_decode(int tiles_x, int tiles_y, int band_rows) {
    return TileDecode{make_magic(tiles_x * tiles_y), make_magic(band_rows * tiles_x), make_magic(tiles_y % band_rows ? tiles_y % band_rows : 1)};
}

struct ConvArgs {
    int batch, H, W, cin, cout;
    int n_chunks, n_ct, tiles_x, tiles_y;
    int cin_vec_ok;  // cin % UNIT == 0 and base pointers 16-B aligned -> vector staging
    int cin_chunk_ok;  // cin is a whole number of CK-channel chunks (no ragged last chunk)
    const void* in0; const void* in1; const float* in_gate; void* in_store;
    const void* wpacked; const float* bias;
    const float* film_scale; const float* film_shift;
    int act; float act_slope;
    const void* mul_plus1; const void* residual;
    const float* out_scale;            // (B, cout) fp32 or NULL: v *= out_scale[b][c] after act / mul_plus1, before the residual (CALayer gate applied by the producing conv)
    void* out; int out_mode; int out_dtype; int out_h, out_w;
    float* chan_sums; int cout_packed;
    int sum_slots;                     // partial-sum slots per image in chan_sums as the caller allocated them (rc_conv_desc.chan_sums_slots; 0 there = 4 per 8x32 tile)
    int sums_compact_ok;               // rc_debug_set("sums_compact"): 0 keeps the per-tile layout everywhere
    int sums_compact;                  // set by the launcher: the carried-sums kernels (2 / 6 / 7) write ONE slot per (residue class of the tile walk, wave) -- see compact_slot()
    int* query;                        // rc_conv_sum_slots(): the launcher reports {kind, slots per image} of the branch it WOULD take and launches nothing
    int num_cus; int persist_ok;
    int ep_key;                        // epilogue_fast feature mask, or -1 for the generic epilogue (ep_key_for)
    TileDecode td, td_wsm;             // division constants of the persistent kernels' tile decode (8- and 16-row tiles)
    MagicDiv div_n_ct;
    int w_resident;                    // kernel 2 with several cout tiles whose packed weights ALL fit LDS beside the tile (set by the launcher): staged once per block, no per-tile DMA
    int pss;                           // single-chunk pixel-shuffle layer: kernel 5 (output staged through LDS)
    int auto_impl;                     // single-chunk, single-cout-tile bf16 3x3 layers: kernel 6 (wave-autonomous strips)
    int thin;                          // kernel 4b (one barrier per stage) where kernel 4 would run a layer with thin stages
    int fold2, src_H, src_W, cfold;    // ksize 2 over the UN-shuffled input: in0 is (B, src_H, src_W, cfold = cin / 4), see ConvDev::fold_*
    long long* dbg;                    // optional phase-timing buffer (rc_debug_set_ptr), normally NULL
    int dbg_flags;                     // knock-out experiments (rc_debug_set "conv_flags"): 1 no stores, 2 no MFMA, 4 no tile loads
};

constexpr int kTH = 8, kTW = 32, kThreads = 256;

constexpr int pix_stride_bytes(int ck_bytes) {
    // smallest multiple of 16 >= ck_bytes that is == 32 (mod 64)
    int s = (ck_bytes + 15) / 16 * 16;
    while (s % 64 != 32) s += 16;
    return s;
}

// ---- unit map: which (tap, channel unit) does lane group q handle in MFMA step s? ------------------
// Shared by the device kernels and the host packer (rc_conv_pack_weights).
//   upt % 4 == 0 : linear, a step never spans taps                       (64 bf16 / 16 fp32 channels)
//   upt == 6     : step t < taps  -> tap t, units 0..3;                  (48 bf16 channels)
//                  step taps + p  -> units 4,5 of taps 2p (q=0,1) and 2p+1 (q=2,3)
//   otherwise    : linear over (tap, unit); steps may span taps          (tiny Cin: head, lens shading)
__host__ __device__ constexpr int unit_map_steps(int upt, int taps) {
    return upt == 6 ? taps + (taps + 1) / 2 : (taps * upt + 3) / 4;
}
__host__ __device__ constexpr bool unit_map(int upt, int taps, int s, int q, int& tap, int& cu) {
    if (upt == 6) {
        if (s < taps) { tap = s; cu = q; return true; }
        tap = 2 * (s - taps) + (q >> 1); cu = 4 + (q & 1);
        return tap < taps;
    }
    const int u = 4 * s + q;
    tap = u / upt; cu = u % upt;
    return u < taps * upt;
}

template <typename T, int CK_, int NT_, int KS_, int TH_ = kTH>
struct ConvCfg {
    using elem = T;
    static constexpr int CK = CK_, NT = NT_, KS = KS_, TH = TH_;   // TH: pixel-tile rows staged per block (2 per compute wave)
    static constexpr int UNIT = 16 / (int)sizeof(T);
    static_assert(CK % UNIT == 0, "CK must be a whole number of 16-byte units");
    static constexpr int UPT = CK / UNIT;         // units per tap
    static constexpr int TAPS = KS * KS;
    static constexpr int STEPS = unit_map_steps(UPT, TAPS);  // MFMA steps per Cin chunk
    static constexpr int HALO = KS / 2;
    static constexpr int THH = TH + 2 * HALO, TWH = kTW + 2 * HALO;
    static constexpr int SPIX = pix_stride_bytes(CK * (int)sizeof(T));
    static constexpr int IN_BYTES = THH * TWH * SPIX;
    static constexpr int G_RAW = (80 * 1024 - IN_BYTES) / (NT * 1024);
    static constexpr int G_CAP = G_RAW < 1 ? 1 : (G_RAW > STEPS ? STEPS : G_RAW);
    static constexpr int NSUB = (STEPS + G_CAP - 1) / G_CAP;
    static constexpr int G = (STEPS + NSUB - 1) / NSUB;  // steps of weights resident in LDS (general kernel)
    static constexpr int W_BYTES = G * NT * 1024;
    static constexpr int COUT_TILE = 16 * NT;
    static constexpr int LDS_BYTES = IN_BYTES + W_BYTES;
    static constexpr size_t CHUNK_W_BYTES = (size_t)STEPS * NT * 1024;  // packed weights per (ct, chunk)
};

// Out-of-bounds sentinel for buffer offsets: stays >= num_records (< 2 GiB by contract, checked on the host)
// after a row's immediate offsets are added -- unlike ~0, which wraps back into the image.
constexpr int kOOB = (int)0x80000000;

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t r, int voff, int soff, const uint4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{v.x, v.y, v.z, v.w}, r, voff, soff, 0);
}

// ---- MFMA wrappers -------------------------------------------------------------------------------
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(const uint4& w, const uint4& x, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    __device__ static __forceinline__ void run(const uint4& w, const uint4& x, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(x.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(x.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(x.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(x.w), acc, 0, 0, 0);
    }
};

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return Vec16<bf16_t>::rne2(lo, hi);
}

// Store / load NV consecutive elements of type TT at byte offset voff of a buffer (OOB -> dropped / 0).
// bf16: values are converted in pairs (v_cvt_pk_bf16_f32) and written with the widest pieces that fit
// (24 bytes = 16 + 8).  PK_RELU applies max(x, 0) to the packed pairs: signed 16-bit max with 0 equals
// max(x, +0) on bf16 bit patterns, one VALU op per two values.
template <typename TT, int NV, bool PK_RELU>
__device__ __forceinline__ void buf_store_row(__amdgpu_buffer_rsrc_t r, int voff, const float* v) {
    if constexpr (sizeof(TT) == 4) {
        static_assert(!PK_RELU, "packed ReLU is a bf16 form");
#pragma unroll
        for (int i = 0; i < NV; i += 4)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{__float_as_uint(v[i]), __float_as_uint(v[i + 1]), __float_as_uint(v[i + 2]), __float_as_uint(v[i + 3])},
                                                   r, voff + 4 * i, 0, 0);
    } else {
        unsigned w[NV / 2];
#pragma unroll
        for (int i = 0; i < NV / 2; ++i) {
            w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
            if constexpr (PK_RELU) {
                typedef short s16x2 __attribute__((ext_vector_type(2)));
                const s16x2 z = {0, 0};
                w[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w[i]), z));
            }
        }
        constexpr int NQ = NV / 8;               // whole 16-byte pieces
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]}, r, voff + 16 * i, 0, 0);
        if constexpr (NV % 8 != 0)
            __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{w[4 * NQ], w[4 * NQ + 1]}, r, voff + 16 * NQ, 0, 0);
    }
}

template <typename TT, int NV>
__device__ __forceinline__ void buf_load_row(__amdgpu_buffer_rsrc_t r, int voff, float* v) {
    if constexpr (sizeof(TT) == 4) {
#pragma unroll
        for (int i = 0; i < NV; i += 4) {
            const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 4 * i, 0, 0);
            v[i] = __uint_as_float(t.x); v[i + 1] = __uint_as_float(t.y); v[i + 2] = __uint_as_float(t.z); v[i + 3] = __uint_as_float(t.w);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; i += 4) {
            const u32x2_t p = __builtin_amdgcn_raw_buffer_load_b64(r, voff + 2 * i, 0, 0);
            v[i] = __uint_as_float(p.x << 16); v[i + 1] = __uint_as_float(p.x & 0xffff0000u);
            v[i + 2] = __uint_as_float(p.y << 16); v[i + 3] = __uint_as_float(p.y & 0xffff0000u);
        }
    }
}

// Persistent-kernel tile enumeration inside one image: bands of kBandRows tile rows, column-major inside a
// band.  A run of 64 consecutive indices is then an 8x8 block of tiles, so the run each XCD takes from the
// current window shares its horizontal AND vertical halos through that XCD's L2 (row-major order left the
// vertical halos on different XCDs: measured FETCH_SIZE 1.27x the algorithmic input bytes).
constexpr int kBandRows = 8;
__device__ __forceinline__ void band_decode(int idx, int tiles_x, int tiles_y, const TileDecode& td, int& ty, int& tx) {
    const int band = magic_div(idx, td.band);
    const int rem = idx - band * (kBandRows * tiles_x);
    const int rows_left = tiles_y - band * kBandRows;
    const bool whole = rows_left >= kBandRows;         // only the image's last band can be shorter
    const int col = whole ? rem >> 3 : magic_div(rem, td.last_rows);
    ty = band * kBandRows + (rem - col * (whole ? kBandRows : rows_left));
    tx = col;
}

// ==================================================================================================
// Device pieces shared by the two kernels below.
// ==================================================================================================
template <class Cfg>
struct ConvDev {
    using T = typename Cfg::elem;
    static constexpr int CK = Cfg::CK, NT = Cfg::NT, KS = Cfg::KS, UNIT = Cfg::UNIT, UPT = Cfg::UPT, TAPS = Cfg::TAPS;
    static constexpr int STEPS = Cfg::STEPS, HALO = Cfg::HALO, THH = Cfg::THH, TWH = Cfg::TWH;
    static constexpr int SPIX = Cfg::SPIX, NV = 4 * NT, ES = (int)sizeof(T);
    static constexpr int VPP = UPT;                       // 16-byte vectors per pixel
    static constexpr int NPIX = THH * TWH;                // pixels in one halo tile
    // staging map: thread t always owns channel group v = t % VPP and walks pixels t/VPP + k*PPP, so
    // per-channel data (the CALayer gate) is loaded once per tile, consecutive threads touch consecutive
    // 16-byte pieces (coalesced loads, conflict-free LDS writes) and LDS addresses are lane-const + imm.
    static constexpr int PPP = kThreads / VPP;            // pixels per pass
    static constexpr int ACTIVE = PPP * VPP;              // threads that take part in staging
    static constexpr int NI = (NPIX + PPP - 1) / PPP;     // passes
    static constexpr bool TABLE = (UPT % 4 != 0) && (UPT != 6);   // generic map needs a per-lane address table

    __device__ static constexpr int tap_off(int tap) { return ((tap / KS) * TWH + (tap % KS)) * SPIX; }

    // per-lane operand-address state (bytes into the LDS input tile, relative to the pixel-tile origin)
    struct LaneOff {
        int a;   // q*16                                   : steps whose 4 units are consecutive in one tap
        int b;   // (q>>1)*SPIX          + (4+(q&1))*16    : paired step, second tap = next column
        int c;   // (q>>1)*(TWH-(KS-1))*SPIX + (4+(q&1))*16 : paired step, second tap = first column of next row
        int tab[TABLE ? STEPS : 1];
    };
    __device__ static __forceinline__ void lane_offsets(int q, LaneOff& lo) {
        lo.a = q * 16;
        lo.b = (q >> 1) * SPIX + (4 + (q & 1)) * 16;
        lo.c = (q >> 1) * (TWH - (KS - 1)) * SPIX + (4 + (q & 1)) * 16;
        if constexpr (TABLE) {
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                int tap = 0, cu = 0;
                if (!unit_map(UPT, TAPS, s, q, tap, cu)) { tap = TAPS - 1; cu = UPT - 1; }  // padded: any valid address
                lo.tab[s] = tap_off(tap) + cu * 16;
            }
        } else {
            lo.tab[0] = 0;
        }
    }
    // byte offset of this lane group's unit for step s (s is a compile-time constant after unrolling)
    __device__ static __forceinline__ int step_off(int s, const LaneOff& lo) {
        if constexpr (TABLE) {
            return lo.tab[s];
        } else if constexpr (UPT == 6) {
            if (s < TAPS) return lo.a + tap_off(s);
            const int t0 = 2 * (s - TAPS);
            const bool wrap = (t0 + 1 < TAPS) && ((t0 + 1) % KS == 0);   // second tap starts a new kernel row
            return (wrap ? lo.c : lo.b) + tap_off(t0);
        } else {
            const int u = 4 * s;
            return lo.a + tap_off(u / UPT) + (u % UPT) * 16;
        }
    }
    // does some lane group of step s carry a padding unit (operand must be zeroed)?
    __device__ static constexpr bool step_has_pad(int s) {
        int tap = 0, cu = 0;
        return !unit_map(UPT, TAPS, s, 3, tap, cu);
    }
    __device__ static __forceinline__ bool lane_is_pad(int s, int q) {
        if constexpr (UPT == 6) return q >= 2;            // only the odd tap of the last pair can be missing
        else return 4 * s + q >= TAPS * UPT;
    }

    // ---- input staging ----------------------------------------------------------------------------
    struct TileSrc {
        __amdgpu_buffer_rsrc_t r0, r1, rst;   // in0 / in1 (skip) / in_store, one image each
        int gy0, gx0;                         // global coords of halo pixel (0,0)
        int soff;                             // byte offset of halo pixel (0,0), channel 0 (interior tiles)
        bool interior;                        // whole halo tile inside the image and Cin a multiple of CK (uniform)
    };
    // ksize 2 with a.fold2 (a stride-2 3x3 convolution read straight from its input, no space-to-depth pass): in0 is the un-shuffled
    // (B, src_H, src_W, cfold) map; channel c0 of the (H, W, 4 cfold) map the MFMA loop sees is channel c0 % cfold of pixel
    // (2y + i, 2x + j), phase 2i + j = c0 / cfold (rc_space_to_depth2's order), zero beyond the source edge.  cfold % CK == 0 (host),
    // so a chunk lies inside one phase and an interior tile's address is still lane constant + scalar.
    __device__ static __forceinline__ bool folded(const ConvArgs& a) {
        if constexpr (Cfg::KS == 2) return a.fold2 != 0; else return false;
    }
    __device__ static __forceinline__ int fold_phase(const ConvArgs& a, int c0) { return (c0 >= a.cfold) + (c0 >= 2 * a.cfold) + (c0 >= 3 * a.cfold); }
    __device__ static __forceinline__ int chunk_soff(const ConvArgs& a, int chunk) {   // byte offset a Cin chunk adds to an interior tile's address
        if (folded(a)) {
            const int c0 = chunk * CK, ph = fold_phase(a, c0);
            return (((ph >> 1) * a.src_W + (ph & 1)) * a.cfold + (c0 - ph * a.cfold)) * ES;
        }
        return chunk * CK * ES;
    }
    __device__ static __forceinline__ TileSrc tile_src(const ConvArgs& a, int b, int y0, int x0) {
        const bool fold = folded(a);
        const size_t img = fold ? (size_t)a.src_H * a.src_W * a.cfold : (size_t)a.H * a.W * a.cin;
        const unsigned bytes = (unsigned)(img * ES);
        TileSrc t;
        t.r0 = make_rsrc(static_cast<const T*>(a.in0) + (size_t)b * img, bytes);
        t.r1 = make_rsrc(a.in1 ? static_cast<const T*>(a.in1) + (size_t)b * img : nullptr, a.in1 ? bytes : 0u);
        t.rst = make_rsrc(a.in_store ? static_cast<T*>(a.in_store) + (size_t)b * img : nullptr, a.in_store ? bytes : 0u);
        t.gy0 = y0 - HALO; t.gx0 = x0 - HALO;
        if (fold) {
            t.soff = (2 * t.gy0 * a.src_W + 2 * t.gx0) * a.cfold * ES;
            t.interior = t.gy0 >= 0 && t.gx0 >= 0 && 2 * (t.gy0 + THH) <= a.src_H && 2 * (t.gx0 + TWH) <= a.src_W;
        } else {
            t.soff = (t.gy0 * a.W + t.gx0) * a.cin * ES;
            t.interior = t.gy0 >= 0 && t.gx0 >= 0 && t.gy0 + THH <= a.H && t.gx0 + TWH <= a.W && a.cin_chunk_ok;
        }
        return t;
    }
    // Per-thread, per-launch constants: byte offset of (halo pixel k of this thread, its channel group) relative
    // to halo pixel (0,0), or kOOB for slots past the tile.  For interior tiles (95 % at 4K) a load address is
    // this constant + a scalar tile offset carried in the buffer instruction's soffset: no per-tile VALU at all.
    // ctr has the same offsets for the tile's centre pixels only (the ones a gated conv materialises).
    struct TileOffs { int o[NI]; int ctr[NI]; };
    __device__ static __forceinline__ void tile_offsets(const ConvArgs& a, int tid, TileOffs& t) {
        const int v = tid % VPP, p0 = tid / VPP;
        const bool live = tid < ACTIVE;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int pix = p0 + k * PPP;
            const int py = pix / TWH, px = pix - py * TWH;
            const bool center = py >= HALO && py < HALO + Cfg::TH && px >= HALO && px < HALO + kTW;
            const int o = folded(a) ? ((2 * py * a.src_W + 2 * px) * a.cfold + v * UNIT) * ES : ((py * a.W + px) * a.cin + v * UNIT) * ES;
            t.o[k] = (live && pix < NPIX) ? o : kOOB;
            t.ctr[k] = center ? t.o[k] : kOOB;
        }
    }
    // byte offset of (halo pixel pix, channel group v) inside the image, or kOOB (-> zero fill / dropped)
    __device__ static __forceinline__ int vec_off(const ConvArgs& a, const TileSrc& t, int chunk, int pix, int v,
                                                  bool live, bool& center) {
        const int py = pix / TWH, px = pix - py * TWH;
        const int gy = t.gy0 + py, gx = t.gx0 + px;
        const int c0 = chunk * CK + v * UNIT;
        center = py >= HALO && py < HALO + Cfg::TH && px >= HALO && px < HALO + kTW;
        if (folded(a)) {
            const int ph = fold_phase(a, c0), sy = 2 * gy + (ph >> 1), sx = 2 * gx + (ph & 1);       // gy, gx >= -1: negative stays negative
            const bool ok = live && pix < NPIX && (unsigned)sy < (unsigned)a.src_H && (unsigned)sx < (unsigned)a.src_W && c0 < a.cin;
            return ok ? ((sy * a.src_W + sx) * a.cfold + c0 - ph * a.cfold) * ES : kOOB;
        }
        const bool ok = live && pix < NPIX && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W && c0 < a.cin;
        return ok ? ((gy * a.W + gx) * a.cin + c0) * ES : kOOB;
    }

    // Issue every global load of one halo tile back-to-back into registers (vector path only).
    template <bool GATED>
    __device__ static __forceinline__ void load_tile(const ConvArgs& a, const TileSrc& t, const TileOffs& to, int b, int chunk, int tid,
                                                     uint4 (&r0)[NI], uint4 (&r1)[GATED ? NI : 1],
                                                     float (&gv)[GATED ? UNIT : 1]) {
        load_gate<GATED>(a, b, chunk, tid, gv);
        if (t.interior) load_tile_interior<GATED>(a, t, to, chunk, r0, r1);
        else load_tile_border<GATED>(a, t, chunk, tid, r0, r1);
    }
    template <bool GATED>
    __device__ static __forceinline__ void load_gate(const ConvArgs& a, int b, int chunk, int tid, float (&gv)[GATED ? UNIT : 1]) {
        if constexpr (GATED) {
            const int c0 = chunk * CK + (tid % VPP) * UNIT;
#pragma unroll
            for (int e = 0; e < UNIT; ++e) gv[e] = (tid < ACTIVE && c0 + e < a.cin) ? a.in_gate[(size_t)b * a.cin + c0 + e] : 0.f;
        }
    }
    template <bool GATED>
    __device__ static __forceinline__ void load_tile_interior(const ConvArgs& a, const TileSrc& t, const TileOffs& to, int chunk,
                                                              uint4 (&r0)[NI], uint4 (&r1)[GATED ? NI : 1]) {
        const int soff = t.soff + chunk_soff(a, chunk);
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            r0[k] = buf_load16(t.r0, to.o[k], soff);
            if constexpr (GATED) r1[k] = buf_load16(t.r1, to.o[k], soff);
        }
    }
    template <bool GATED>
    __device__ static __forceinline__ void load_tile_border(const ConvArgs& a, const TileSrc& t, int chunk, int tid,
                                                            uint4 (&r0)[NI], uint4 (&r1)[GATED ? NI : 1]) {
        const int v = tid % VPP, p0 = tid / VPP;
        const bool live = tid < ACTIVE;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            bool center;
            const int off = vec_off(a, t, chunk, p0 + k * PPP, v, live, center);
            r0[k] = buf_load16(t.r0, off, 0);
            if constexpr (GATED) r1[k] = buf_load16(t.r1, off, 0);
        }
    }

    // Combine (CALayer gate + skip), optionally materialise, and write the tile to LDS.
    template <bool GATED>
    __device__ static __forceinline__ void commit_tile(const ConvArgs& a, const TileSrc& t, const TileOffs& to, int chunk, int tid,
                                                       const uint4 (&r0)[NI], const uint4 (&r1)[GATED ? NI : 1],
                                                       const float (&gv)[GATED ? UNIT : 1], char* s_in) {
        const int v = tid % VPP, p0 = tid / VPP;
        const bool live = tid < ACTIVE;
        char* dst = s_in + p0 * SPIX + v * 16;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int pix = p0 + k * PPP;
            uint4 raw = r0[k];
            if constexpr (GATED) {
                float f0[UNIT], f1[UNIT];
                Vec16<T>::unpack(r0[k], f0);
                Vec16<T>::unpack(r1[k], f1);
#pragma unroll
                for (int e = 0; e < UNIT; ++e) f0[e] = f0[e] * gv[e] + f1[e];   // zero-filled lanes stay 0*g+0 = 0
                raw = Vec16<T>::pack(f0);
                if (t.interior) {                                                  // rst has 0 records if in_store == NULL
                    buf_store16(t.rst, to.ctr[k], t.soff + chunk * CK * ES, raw);
                } else {
                    bool center;
                    const int off = vec_off(a, t, chunk, pix, v, live, center);
                    buf_store16(t.rst, center ? off : kOOB, 0, raw);
                }
            }
            if (live && pix < NPIX) *reinterpret_cast<uint4*>(dst + k * PPP * SPIX) = raw;
        }
    }

    // tiny / odd Cin (head 4->C, lens-shading 2->C): element loads straight to LDS (never gated in practice,
    // but the gate is honoured for completeness)
    __device__ static __forceinline__ void stage_tile_scalar(const ConvArgs& a, int b, int y0, int x0, int chunk, int tid,
                                                             char* s_in, T* in_store) {
        const T* in0 = static_cast<const T*>(a.in0);
        const T* in1 = static_cast<const T*>(a.in1);
        const size_t img_base = (size_t)b * a.H * a.W;
        if constexpr (ES == 2 && UNIT == 8 && VPP == 1) {
            // a 4-channel bf16 map (the packed RAW in front of the codec's 4 -> 128 head, models/raw2bit.py:1791): a pixel is ONE 8-byte load, its unit the 8 bytes + 8 of zeros
            // (the element loop below -- 4 two-byte loads, 4 conversions and a re-pack per pixel -- was 2.2 ms of the codec step for a layer that writes 4.5 GB)
            if (a.cin == 4 && a.in_gate == nullptr && (reinterpret_cast<uintptr_t>(in0) & 7) == 0) {      // uniform
                for (int i = tid; i < NPIX; i += kThreads) {
                    const int py = i / TWH, px = i - py * TWH;
                    const int gy = y0 + py - HALO, gx = x0 + px - HALO;
                    uint2 w = make_uint2(0u, 0u);
                    if (chunk == 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                        w = *reinterpret_cast<const uint2*>(in0 + (img_base + (size_t)gy * a.W + gx) * 4);
                    *reinterpret_cast<uint4*>(s_in + i * SPIX) = make_uint4(w.x, w.y, 0u, 0u);
                }
                return;
            }
        }
        for (int i = tid; i < NPIX * VPP; i += kThreads) {
            const int pix = i / VPP, v = i - pix * VPP;
            const int py = pix / TWH, px = pix - py * TWH;
            const int gy = y0 + py - HALO, gx = x0 + px - HALO;
            const int c0 = chunk * CK + v * UNIT;
            const bool center = py >= HALO && py < HALO + Cfg::TH && px >= HALO && px < HALO + kTW;
            uint4 raw = make_uint4(0u, 0u, 0u, 0u);
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && c0 < a.cin) {
                const size_t off = (img_base + (size_t)gy * a.W + gx) * a.cin + c0;
                float f0[UNIT];
#pragma unroll
                for (int e = 0; e < UNIT; ++e) {
                    float val = 0.f;
                    if (c0 + e < a.cin) {
                        val = to_f32(in0[off + e]);
                        if (a.in_gate != nullptr)
                            val = val * a.in_gate[(size_t)b * a.cin + c0 + e] + to_f32(in1[off + e]);
                    }
                    f0[e] = val;
                }
                raw = Vec16<T>::pack(f0);
                if (a.in_gate != nullptr && in_store != nullptr && center) {
#pragma unroll
                    for (int e = 0; e < UNIT; ++e)
                        if (c0 + e < a.cin) in_store[off + e] = from_f32<T>(f0[e]);
                }
            }
            *reinterpret_cast<uint4*>(s_in + pix * SPIX + v * 16) = raw;
        }
    }

    // packed weights: linear global -> LDS copy by LDS-DMA (no VGPR round trip), 1 KiB per wave-instruction.
    // The next __syncthreads() drains it (its release carries vmcnt(0)) before any wave reads them.
    __device__ static __forceinline__ void dma_weights(const char* src, char* dst, int nkb, int wave, int lane_w) {
        for (int kb = wave; kb < nkb; kb += kThreads / 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + kb * 1024 + lane_w),
                                             (__attribute__((address_space(3))) void*)(dst + kb * 1024), 16, 0, 0);
    }

    // ---- MFMA steps [S0, S0+COUNT) of a chunk; weights of step s live at s_w + (s - W0)*NT KiB ------
    // Operand fragments are double-buffered in registers and the issue order is fixed by hand: while step s's
    // 4*NT MFMA items run, step s+1's NT+4 LDS fragment reads are issued one at a time between them, each fenced
    // with sched_barrier(0).  Left to itself the machine scheduler sinks every read to just before its first use
    // (to save registers) and the wave then sits out the LDS latency several times per step: measured 25 cycles
    // per MFMA instead of 16.
    static constexpr int FR = NT + 4;          // fragment reads per step: NT weight + 4 pixel-tile fragments
    static constexpr int FM = 4 * NT;          // MFMA items per step (an item = 1 bf16 MFMA or 4 fp32 MFMAs)
    template <int I>
    __device__ static __forceinline__ void load_frag_item(int s, int w0, const char* s_in, const char* s_w, int lane_x,
                                                          int lane_w, const LaneOff& lo, uint4 (&wf)[NT], uint4 (&xf)[4]) {
        if constexpr (I < NT) {
            wf[I] = *reinterpret_cast<const uint4*>(s_w + ((s - w0) * NT + I) * 1024 + lane_w);
        } else {
            constexpr int pt = I - NT;
            xf[pt] = *reinterpret_cast<const uint4*>(s_in + lane_x + step_off(s, lo) + ((pt >> 1) * TWH + (pt & 1) * 16) * SPIX);
        }
    }
    __device__ static __forceinline__ void zero_pad_frags(int s, int q, uint4 (&xf)[4]) {
        if (step_has_pad(s)) {                 // compile-time after unrolling
            if (lane_is_pad(s, q)) {
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) xf[pt] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    __device__ static __forceinline__ void load_frags(int s, int w0, const char* s_in, const char* s_w, int lane_x,
                                                      int lane_w, int q, const LaneOff& lo, uint4 (&wf)[NT], uint4 (&xf)[4]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            wf[nt] = *reinterpret_cast<const uint4*>(s_w + ((s - w0) * NT + nt) * 1024 + lane_w);
        const char* xp = s_in + lane_x + step_off(s, lo);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
            xf[pt] = *reinterpret_cast<const uint4*>(xp + ((pt >> 1) * TWH + (pt & 1) * 16) * SPIX);
        zero_pad_frags(s, q, xf);
    }
    __device__ static __forceinline__ void mma_frags(const uint4 (&wf)[NT], const uint4 (&xf)[4], f32x4 (&acc)[4][NT]) {
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) Mma<T>::run(wf[nt], xf[pt], acc[pt][nt]);
    }
    // MFMA slot m of a step.  bf16: slot = item (pt, nt).  fp32: an item is FOUR dependent 16x16x4 MFMAs on one
    // accumulator, so the slots run k-slice-major (all items' slice 0, then slice 1, ...): 4*NT independent MFMAs
    // sit between two that touch the same accumulator (item-major order stalled every MFMA on its predecessor and
    // cost the fp32 path a factor 2.3).
    static constexpr int SLOTS = FM * (ES == 4 ? 4 : 1);
    __device__ static __forceinline__ void mma_slot(int m, const uint4 (&wf)[NT], const uint4 (&xf)[4], f32x4 (&acc)[4][NT]) {
        if constexpr (ES == 4) {
            const int e = m / FM, j = m % FM, pt = j / NT, nt = j % NT;
            const unsigned w = e == 0 ? wf[nt].x : e == 1 ? wf[nt].y : e == 2 ? wf[nt].z : wf[nt].w;
            const unsigned x = e == 0 ? xf[pt].x : e == 1 ? xf[pt].y : e == 2 ? xf[pt].z : xf[pt].w;
            acc[pt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w), __uint_as_float(x), acc[pt][nt], 0, 0, 0);
        } else {
            Mma<T>::run(wf[m % NT], xf[m / NT], acc[m / NT][m % NT]);
        }
    }
    // one step with the next step's reads interleaved (I = read index; recursion keeps every index a constant)
    template <int I, bool NEXT>
    __device__ static __forceinline__ void step_interleaved(int s, int w0, const char* s_in, const char* s_w, int lane_x, int lane_w,
                                                            const LaneOff& lo, const uint4 (&wf)[NT], const uint4 (&xf)[4],
                                                            uint4 (&wfn)[NT], uint4 (&xfn)[4], f32x4 (&acc)[4][NT]) {
        if constexpr (I < FR) {
            if constexpr (NEXT) load_frag_item<I>(s + 1, w0, s_in, s_w, lane_x, lane_w, lo, wfn, xfn);
            if constexpr (ES == 2) __builtin_amdgcn_sched_barrier(0);   // fp32: 64 MFMAs per step give hipcc room; it schedules better unfenced
#pragma unroll
            for (int m = (I * SLOTS) / FR; m < ((I + 1) * SLOTS) / FR; ++m) mma_slot(m, wf, xf, acc);
            if constexpr (ES == 2) __builtin_amdgcn_sched_barrier(0);
            step_interleaved<I + 1, NEXT>(s, w0, s_in, s_w, lane_x, lane_w, lo, wf, xf, wfn, xfn, acc);
        }
    }
    template <int S0, int COUNT, int W0, bool DBUF = true>
    __device__ static __forceinline__ void mma_steps(const char* s_in, const char* s_w, int lane_x, int lane_w, int q,
                                                     const LaneOff& lo, f32x4 (&acc)[4][NT]) {
        constexpr int END = (S0 + COUNT) < STEPS ? (S0 + COUNT) : STEPS;
        if constexpr (S0 < END && !DBUF) {     // register-starved variants: one fragment set, compiler-scheduled
#pragma unroll
            for (int s = S0; s < END; ++s) {
                uint4 wf[NT], xf[4];
                load_frags(s, W0, s_in, s_w, lane_x, lane_w, q, lo, wf, xf);
                mma_frags(wf, xf, acc);
            }
        } else if constexpr (S0 < END && ES == 4) {
            // fp32: 16*NT MFMAs per step leave hipcc's own scheduler plenty of cover for the LDS reads; every
            // hand-ordered variant tried (item-major or k-slice-major, fenced or not) measured 20-55 % slower
            uint4 wfa[NT], xfa[4], wfb[NT], xfb[4];
            load_frags(S0, W0, s_in, s_w, lane_x, lane_w, q, lo, wfa, xfa);
#pragma unroll
            for (int s = S0; s < END; s += 2) {
                if (s + 1 < END) load_frags(s + 1, W0, s_in, s_w, lane_x, lane_w, q, lo, wfb, xfb);
                mma_frags(wfa, xfa, acc);
                if (s + 2 < END) load_frags(s + 2, W0, s_in, s_w, lane_x, lane_w, q, lo, wfa, xfa);
                if (s + 1 < END) mma_frags(wfb, xfb, acc);
            }
        } else if constexpr (S0 < END) {
            uint4 wfa[NT], xfa[4], wfb[NT], xfb[4];
            load_frags(S0, W0, s_in, s_w, lane_x, lane_w, q, lo, wfa, xfa);
#pragma unroll
            for (int s = S0; s < END; s += 2) {
                if (s + 1 < END) step_interleaved<0, true>(s, W0, s_in, s_w, lane_x, lane_w, lo, wfa, xfa, wfb, xfb, acc);
                else step_interleaved<0, false>(s, W0, s_in, s_w, lane_x, lane_w, lo, wfa, xfa, wfb, xfb, acc);
                if (s + 1 < END) {
                    zero_pad_frags(s + 1, q, xfb);
                    if (s + 2 < END) step_interleaved<0, true>(s + 1, W0, s_in, s_w, lane_x, lane_w, lo, wfb, xfb, wfa, xfa, acc);
                    else step_interleaved<0, false>(s + 1, W0, s_in, s_w, lane_x, lane_w, lo, wfb, xfb, wfa, xfa, acc);
                    if (s + 2 < END) zero_pad_frags(s + 2, q, xfa);
                }
            }
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------------
    // lane (q, n) holds, per pixel tile pt, packed couts jbase .. jbase+NV-1 of pixel (row, col0+n).
    // The accumulators already contain the bias (it is the MFMA chain's initial C operand).
    //
    // Two forms.  epilogue_fast<F> is compiled once per feature mask F that the big layers use (NHWC or
    // pixel-shuffle store of full cout tiles): every optional operand is a compile-time decision, so a tile's
    // epilogue is ~100 VALU instead of ~370 (a lone wave issues about one instruction per 4 cycles -- the
    // epilogue's instruction COUNT was costing as much time as the 168-MFMA loop).  epilogue_generic handles
    // everything (GELU, ragged cout, planar NCHW store, any operand mix) with run-time branches.
    enum : int { EP_RELU = 1, EP_LEAKY = 2, EP_FILM = 4, EP_MUL = 8, EP_RES = 16, EP_SUMS = 32, EP_GATE = 64 };

    template <bool FAST>
    __device__ static __forceinline__ void epilogue(const ConvArgs& a, int b, int y0, int x0, int sp, int ct, int tid,
                                                    f32x4 (&acc)[4][NT]) {
        // fp32 always takes the generic form: with the seven-way switch hipcc keeps the fp32 accumulators in scratch
        // (224 B/lane, measured 92 -> 74 TF/s), and its MFMA loop is 4x longer per tile anyway
        if constexpr (!FAST || ES == 4) return epilogue_generic(a, b, y0, x0, sp, ct, tid, acc);
        else switch (a.ep_key) {                  // uniform; set by the host, >= 0 in FAST kernels
            case 0: return epilogue_fast<0>(a, b, y0, x0, sp, ct, tid, acc);
            case EP_RELU: return epilogue_fast<EP_RELU>(a, b, y0, x0, sp, ct, tid, acc);
            case EP_LEAKY: return epilogue_fast<EP_LEAKY>(a, b, y0, x0, sp, ct, tid, acc);
            case EP_RES: return epilogue_fast<EP_RES>(a, b, y0, x0, sp, ct, tid, acc);
            case EP_SUMS: return epilogue_fast<EP_SUMS>(a, b, y0, x0, sp, ct, tid, acc);
            case EP_MUL: return epilogue_fast<EP_MUL>(a, b, y0, x0, sp, ct, tid, acc);
            case EP_FILM | EP_LEAKY: return epilogue_fast<EP_FILM | EP_LEAKY>(a, b, y0, x0, sp, ct, tid, acc);
            case EP_RELU | EP_SUMS: return epilogue_fast<EP_RELU | EP_SUMS>(a, b, y0, x0, sp, ct, tid, acc);       // RCAB conv1 of the early-gate schedule
            case EP_LEAKY | EP_SUMS: return epilogue_fast<EP_LEAKY | EP_SUMS>(a, b, y0, x0, sp, ct, tid, acc);     // ... of the codec's ResidualBlockWithCA
            case EP_GATE | EP_RES: return epilogue_fast<EP_GATE | EP_RES>(a, b, y0, x0, sp, ct, tid, acc);         // RCAB conv2: conv * gate + x
            default: return;                      // unreachable: the host launches the !FAST kernel for other masks
        }
    }

    // RUN form of the channel sums (persistent kernel): a tile whose sums are carried on to the block's next tile leaves ZEROS in its 4 slots.  One 16-byte store
    // per lane of wave 0 covers them (4 waves x cout floats) -- per-wave element stores cost 12 store instructions per wave and tile, more than the tile's own output.
    // Issued between the tile's two workgroup barriers, so a later flush store of any wave to the same slot is ordered behind it.
    __device__ static __forceinline__ void zero_sum_slots(const ConvArgs& a, int b, int sp, int tid) {
        zero_slots4(a, b, 4 * sp, tid);
    }
    // zeros into the 4 consecutive slots [slot0, slot0 + 4) of image b (4 * cout floats, 16-byte aligned: cout % 4 == 0 in FAST kernels), by a 256-thread block
    __device__ static __forceinline__ void zero_slots4(const ConvArgs& a, int b, int slot0, int tid) {
        float* dst = a.chan_sums + ((size_t)b * a.sum_slots + slot0) * a.cout;
        for (int i = tid; i < a.cout; i += kThreads) *reinterpret_cast<float4*>(dst + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // ... and into ONE slot, by one wave (kernels 6 / 7)
    __device__ static __forceinline__ void zero_slot1(const ConvArgs& a, int b, int slot, int lane) {
        float* dst = a.chan_sums + ((size_t)b * a.sum_slots + slot) * a.cout;
        if (4 * lane < a.cout) *reinterpret_cast<float4*>(dst + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // COMPACT layout of the carried sums (kernels 2 / 6 / 7, a.sums_compact): a block (wave group) walks units pos, pos + stride, ...; the units of image b it
    // visits are one residue class r = (pos - b * units_per_image) mod stride of the image's own unit index, so its run total for that image goes to slot
    // r * waves + wave -- `stride * waves` slots per image instead of 4 per tile (2 048 instead of 32 640 at 4K), no zero stores per tile, and a function of the
    // image's own tile indices only: frame i of a batch and frame i alone produce the same partial sums in the same slots (bitwise batch invariance).
    // A residue class without a unit in image b (small images, kernel 6's skipped strips) gets zeros: cover_until().
    __device__ static __forceinline__ int compact_residue(int pos, int b, int units_per_image, int stride) {
        int r = (pos - b * units_per_image) % stride;
        return r < 0 ? r + stride : r;
    }

    template <int CTRL>
    __device__ static __forceinline__ float dpp_add(float s) {   // s + (s of the lane CTRL selects inside the 16-lane row)
        return s + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), CTRL, 0xf, 0xf, false));
    }
    // sum over the 16 lanes of a row; every lane ends with the total (pairs, quads, halves, row: the same
    // tree as an xor butterfly, so the result does not depend on the lane)
    __device__ static __forceinline__ float row_sum16(float s) {
        s = dpp_add<0xB1>(s);    // quad_perm [1,0,3,2]
        s = dpp_add<0x4E>(s);    // quad_perm [2,3,0,1]
        s = dpp_add<0x141>(s);   // row_half_mirror
        s = dpp_add<0x140>(s);   // row_mirror
        return s;
    }
    __device__ static __forceinline__ void write_chan_sums(const ConvArgs& a, int b, int slot, int n, int jbase,
                                                           float (&csum)[NV], bool ragged) {
        // every wave writes its own partial (legacy layout: slot = 4*tile + wave); rc_ca_gate folds them in fixed order
#pragma unroll
        for (int e = 0; e < NV; ++e) csum[e] = row_sum16(csum[e]);
        if (n == 0) {
            float* dst = a.chan_sums + ((size_t)b * a.sum_slots + slot) * a.cout;
#pragma unroll
            for (int e = 0; e < NV; ++e)
                if (!ragged || jbase + e < a.cout) dst[jbase + e] = csum[e];
        }
    }

    template <int F>
    __device__ static __forceinline__ void epilogue_fast(const ConvArgs& a, int b, int y0, int x0, int sp, int ct, int tid,
                                                         f32x4 (&acc)[4][NT]) {
        float none[NV];
        epilogue_fast_impl<F, false>(a, b, y0, x0, sp, ct, tid, acc, none, true);
    }
    // RUN (CALayer sums in a persistent kernel): `run` carries this lane's channel sums from tile to tile; only when
    // `flush` is set (the block's next tile belongs to another image, or there is none) are they reduced over the 16-lane
    // rows and written, to the current tile's slot -- the other tiles' slots get zeros, so rc_ca_gate's fixed-order fold
    // over all slots is unchanged.  Per tile this leaves NV adds per pixel tile instead of a 4-step DPP reduction of NV values.
    // run_slot0 >= 0 (compact layout): the run total goes to slot run_slot0 + (tid >> 6) instead of 4 * sp + (tid >> 6)
    template <int F, bool RUN>
    __device__ static __forceinline__ void epilogue_fast_impl(const ConvArgs& a, int b, int y0, int x0, int sp, int ct, int tid,
                                                              f32x4 (&acc)[4][NT], float (&run)[NV], bool flush, int run_slot0 = -1) {
        const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, n = lane & 15;
        const int jbase = ct * Cfg::COUT_TILE + q * NV;
        const size_t img_out = (size_t)a.H * a.W * a.cout;    // output / residual / mul image (elements)
        const unsigned img_bytes = (unsigned)(img_out * ES);
        const __amdgpu_buffer_rsrc_t r_out = make_rsrc(static_cast<T*>(a.out) + (size_t)b * img_out, img_bytes);
        const int gy = y0 + 2 * wave, gx = x0 + n;
        // NHWC-shaped byte offsets: pixel tile pt adds (pt>>1) rows and (pt&1)*16 columns
        const int row_b = a.W * a.cout * ES, col_b = 16 * a.cout * ES;
        const int off0 = ((gy * a.W + gx) * a.cout + jbase) * ES;
        int o_off0 = off0, o_row = row_b, o_col = col_b;
        if (a.out_mode == RC_OUT_PIXEL_SHUFFLE2) {
            // cout tile ct = (out-channel block ct>>2, sub-pixel ct&3): this lane's NV values are consecutive
            // OUT channels of one output pixel -> the 4 lane groups write one contiguous 16*NT-channel run
            const int cps = a.cout >> 2, sub = ct & 3;
            o_off0 = (((2 * gy + (sub >> 1)) * (2 * a.W) + (2 * gx + (sub & 1))) * cps + (ct >> 2) * Cfg::COUT_TILE + q * NV) * ES;
            o_row = 4 * a.W * cps * ES; o_col = 32 * cps * ES;
        }
        const bool full = y0 + kTH <= a.H && x0 + kTW <= a.W;     // uniform
        float fs[NV], ft[NV];
        if constexpr ((F & EP_FILM) != 0) {                      // Res_GFM: (B,cout) vectors, L2-resident
#pragma unroll
            for (int e = 0; e < NV; e += 4) {
                const float4 s4 = *reinterpret_cast<const float4*>(a.film_scale + (size_t)b * a.cout + jbase + e);
                const float4 t4 = *reinterpret_cast<const float4*>(a.film_shift + (size_t)b * a.cout + jbase + e);
                fs[e] = s4.x; fs[e + 1] = s4.y; fs[e + 2] = s4.z; fs[e + 3] = s4.w;
                ft[e] = t4.x; ft[e + 1] = t4.y; ft[e + 2] = t4.z; ft[e + 3] = t4.w;
            }
        }
        float gs[NV];
        if constexpr ((F & EP_GATE) != 0) {                      // CALayer gate of THIS conv's output, known ahead (rc_ca_gate_ahead)
#pragma unroll
            for (int e = 0; e < NV; e += 4) {
                const float4 g4 = *reinterpret_cast<const float4*>(a.out_scale + (size_t)b * a.cout + jbase + e);
                gs[e] = g4.x; gs[e + 1] = g4.y; gs[e + 2] = g4.z; gs[e + 3] = g4.w;
            }
        }
        float csum[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) csum[e] = RUN ? run[e] : 0.f;
        const float inf = __builtin_inff();

#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int dy = pt >> 1, dx = pt & 1;
            const bool valid = gy + dy < a.H && gx + 16 * dx < a.W;
            float v[NV];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[pt][nt][r];
            if constexpr ((F & EP_FILM) != 0) {
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] = v[e] * fs[e] + ft[e] + v[e];
            }
            // ReLU with nothing after it on bf16: applied to the packed pairs below (one op per two values)
            constexpr bool PK_RELU = F == EP_RELU && ES == 2;
            if constexpr ((F & EP_RELU) != 0 && !PK_RELU) {
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.f, inf);
            }
            if constexpr ((F & EP_LEAKY) != 0) {                 // 0 <= slope <= 1 (host): leaky(v) = med3(v, slope*v, +inf)
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], v[e] * a.act_slope, inf);
            }
            if constexpr ((F & (EP_MUL | EP_RES)) != 0) {
                const int po = valid ? off0 + dy * row_b + dx * col_b : kOOB;
                if constexpr ((F & EP_MUL) != 0) {
                    float m[NV];
                    buf_load_row<T, NV>(make_rsrc(static_cast<const T*>(a.mul_plus1) + (size_t)b * img_out, img_bytes), po, m);
#pragma unroll
                    for (int e = 0; e < NV; ++e) v[e] = v[e] * (m[e] + 1.f);
                }
                if constexpr ((F & EP_GATE) != 0) {
#pragma unroll
                    for (int e = 0; e < NV; ++e) v[e] *= gs[e];
                }
                if constexpr ((F & EP_RES) != 0) {
                    float m[NV];
                    buf_load_row<T, NV>(make_rsrc(static_cast<const T*>(a.residual) + (size_t)b * img_out, img_bytes), po, m);
#pragma unroll
                    for (int e = 0; e < NV; ++e) v[e] += m[e];
                }
            }
            if constexpr ((F & EP_SUMS) != 0) {
                if (full) {
#pragma unroll
                    for (int e = 0; e < NV; ++e) csum[e] += v[e];
                } else {
#pragma unroll
                    for (int e = 0; e < NV; ++e) csum[e] += valid ? v[e] : 0.f;
                }
            }
            int oo = (valid && !(a.dbg_flags & 1)) ? o_off0 + dy * o_row + dx * o_col : kOOB;
            if ((a.dbg_flags & 8) && oo != kOOB) oo &= 0x3fffff;      // knock-out: every store lands in one 4 MB window (L2-resident: no HBM write stream)
            buf_store_row<T, NV, PK_RELU>(r_out, oo, v);
        }
        if constexpr ((F & EP_SUMS) != 0) {
            if (!RUN || flush) {
                write_chan_sums(a, b, (run_slot0 >= 0 ? run_slot0 : 4 * sp) + wave, n, jbase, csum, false);
                if constexpr (RUN) {
#pragma unroll
                    for (int e = 0; e < NV; ++e) run[e] = 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < NV; ++e) run[e] = csum[e];       // this tile's slots hold zeros: written by the kernel before the tile's MFMA loop (zero_sum_slots)
            }
        }
    }

    // ---- residual epilogue with the residual PREFETCHED (persistent kernel, bf16 NHWC, key == EP_RES) -------------------------------------
    // epilogue_fast<EP_RES> loads the residual when the MFMA loop is over: one exposed HBM round trip per tile (level-0 48 -> 48 + residual:
    // 1.14 ms for 4.8 GB where the plain layer moves 3.2 GB in 0.75, tools/conv48_knockouts.py).  Here its loads are issued BEFORE the loop
    // (the tile's output coordinates are known) and land under the MFMAs; NV / 2 more registers per pixel tile.  Same arithmetic, same bits.
    static constexpr int NRH = NV / 2;
    __device__ static __forceinline__ void res_prefetch(const ConvArgs& a, int b, int y0, int x0, int ct, int tid, unsigned (&rp)[4][NRH]) {
        const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, n = lane & 15;
        const int jbase = ct * Cfg::COUT_TILE + q * NV;
        const size_t img_out = (size_t)a.H * a.W * a.cout;
        const __amdgpu_buffer_rsrc_t r_res = make_rsrc(static_cast<const T*>(a.residual) + (size_t)b * img_out, (unsigned)(img_out * ES));
        const int gy = y0 + 2 * wave, gx = x0 + n;
        const int row_b = a.W * a.cout * ES, col_b = 16 * a.cout * ES;
        const int off0 = ((gy * a.W + gx) * a.cout + jbase) * ES;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int dy = pt >> 1, dx = pt & 1;
            const bool valid = gy + dy < a.H && gx + 16 * dx < a.W;
            const int po = valid ? off0 + dy * row_b + dx * col_b : kOOB;
#pragma unroll
            for (int i = 0; i < NRH / 4; ++i) {
                const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(r_res, po + 16 * i, 0, 0);
                rp[pt][4 * i] = t.x; rp[pt][4 * i + 1] = t.y; rp[pt][4 * i + 2] = t.z; rp[pt][4 * i + 3] = t.w;
            }
            if constexpr (NRH % 4 != 0) {
                const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(r_res, po + 16 * (NRH / 4), 0, 0);
                rp[pt][4 * (NRH / 4)] = t.x; rp[pt][4 * (NRH / 4) + 1] = t.y;
            }
        }
    }
    // gate_row: image b's row of out_scale (key EP_GATE | EP_RES), or NULL.  Kernel 6 passes a row of its LDS copy: a GLOBAL load issued here
    // is younger than the next tile's prefetch, and waiting for it (vmcnt counts in order) drains the prefetch.
    __device__ static __forceinline__ void epilogue_res_pre(const ConvArgs& a, int b, int y0, int x0, int ct, int tid, f32x4 (&acc)[4][NT],
                                                            const unsigned (&rp)[4][NRH], const float* gate_row) {
        const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, n = lane & 15;
        const int jbase = ct * Cfg::COUT_TILE + q * NV;
        // key EP_GATE | EP_RES (uniform): v * out_scale[b][c] first.  ONE copy of this epilogue with a uniform branch: a second instantiation
        // beside the plain one cost the kernel 9 spilled registers.  Without a gate the multipliers are 1.0f (exact: the same bits).
        float gs[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) gs[e] = 1.f;
        if (gate_row != nullptr) {
#pragma unroll
            for (int e = 0; e < NV; e += 4) {
                const float4 g4 = *reinterpret_cast<const float4*>(gate_row + jbase + e);
                gs[e] = g4.x; gs[e + 1] = g4.y; gs[e + 2] = g4.z; gs[e + 3] = g4.w;
            }
        }
        const size_t img_out = (size_t)a.H * a.W * a.cout;
        const __amdgpu_buffer_rsrc_t r_out = make_rsrc(static_cast<T*>(a.out) + (size_t)b * img_out, (unsigned)(img_out * ES));
        const int gy = y0 + 2 * wave, gx = x0 + n;
        const int row_b = a.W * a.cout * ES, col_b = 16 * a.cout * ES;
        const int off0 = ((gy * a.W + gx) * a.cout + jbase) * ES;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int dy = pt >> 1, dx = pt & 1;
            const bool valid = gy + dy < a.H && gx + 16 * dx < a.W;
            float v[NV];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[pt][nt][r];
#pragma unroll
            for (int e = 0; e < NV; ++e) v[e] *= gs[e];
#pragma unroll
            for (int i = 0; i < NRH; ++i) {
                v[2 * i] += __uint_as_float(rp[pt][i] << 16);
                v[2 * i + 1] += __uint_as_float(rp[pt][i] & 0xffff0000u);
            }
            const int oo = (valid && !(a.dbg_flags & 1)) ? off0 + dy * row_b + dx * col_b : kOOB;
            buf_store_row<T, NV, false>(r_out, oo, v);
        }
    }

    // ---- kernel 7: epilogue of a 2-row x 16-pixel sub-strip (lane (n, q): pixel (gy + p, gx0 + n), channels q NV .. q NV + NV - 1) ------------
    // NHWC, one cout tile, bf16.  F: EP_RELU / EP_LEAKY / EP_SUMS (carried: `run`, written to slot (sp, wv) when `flush`) / EP_RES (+ EP_GATE),
    // the residual prefetched into rp by sub_res_prefetch.  Same arithmetic, same order as epilogue_fast_impl / epilogue_res_pre.
    __device__ static __forceinline__ void sub_res_prefetch(const ConvArgs& a, int b, int gy, int gx0, int lane, unsigned (&rp)[2][NRH]) {
        const int q = lane >> 4, n = lane & 15;
        const size_t img_out = (size_t)a.H * a.W * a.cout;
        const __amdgpu_buffer_rsrc_t r_res = make_rsrc(static_cast<const T*>(a.residual) + (size_t)b * img_out, (unsigned)(img_out * ES));
        const int gx = gx0 + n;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int po = (gy + p < a.H && gx < a.W) ? (((gy + p) * a.W + gx) * a.cout + q * NV) * ES : kOOB;
#pragma unroll
            for (int i = 0; i < NRH / 4; ++i) {
                const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(r_res, po + 16 * i, 0, 0);
                rp[p][4 * i] = t.x; rp[p][4 * i + 1] = t.y; rp[p][4 * i + 2] = t.z; rp[p][4 * i + 3] = t.w;
            }
            if constexpr (NRH % 4 != 0) {
                const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(r_res, po + 16 * (NRH / 4), 0, 0);
                rp[p][4 * (NRH / 4)] = t.x; rp[p][4 * (NRH / 4) + 1] = t.y;
            }
        }
    }
    template <int F, int NRP>
    __device__ static __forceinline__ void epilogue_sub(const ConvArgs& a, int b, int gy, int gx0, int slot, int lane, f32x4 (&acc)[2][NT],
                                                        float (&run)[(F & EP_SUMS) ? NV : 1], bool flush, const unsigned (&rp)[NRP][(F & EP_RES) ? NRH : 1],
                                                        const float* gate_row) {
        const int q = lane >> 4, n = lane & 15, jbase = q * NV;
        const size_t img_out = (size_t)a.H * a.W * a.cout;
        const __amdgpu_buffer_rsrc_t r_out = make_rsrc(static_cast<T*>(a.out) + (size_t)b * img_out, (unsigned)(img_out * ES));
        const int gx = gx0 + n;
        float gs[(F & EP_GATE) ? NV : 1];
        if constexpr ((F & EP_GATE) != 0) {
#pragma unroll
            for (int e = 0; e < NV; e += 4) {
                const float4 g4 = *reinterpret_cast<const float4*>(gate_row + jbase + e);
                gs[e] = g4.x; gs[e + 1] = g4.y; gs[e + 2] = g4.z; gs[e + 3] = g4.w;
            }
        }
        const float inf = __builtin_inff();
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const bool valid = gy + p < a.H && gx < a.W;
            float v[NV];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[p][nt][r];
            constexpr bool PK_RELU = F == EP_RELU;
            if constexpr ((F & EP_RELU) != 0 && !PK_RELU) {
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.f, inf);
            }
            if constexpr ((F & EP_LEAKY) != 0) {
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], v[e] * a.act_slope, inf);
            }
            if constexpr ((F & EP_GATE) != 0) {
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] *= gs[e];
            }
            if constexpr ((F & EP_RES) != 0) {
#pragma unroll
                for (int i = 0; i < NRH; ++i) {
                    v[2 * i] += __uint_as_float(rp[p][i] << 16);
                    v[2 * i + 1] += __uint_as_float(rp[p][i] & 0xffff0000u);
                }
            }
            if constexpr ((F & EP_SUMS) != 0) {
#pragma unroll
                for (int e = 0; e < NV; ++e) run[e] += valid ? v[e] : 0.f;
            }
            const int oo = (valid && !(a.dbg_flags & 1)) ? (((gy + p) * a.W + gx) * a.cout + jbase) * ES : kOOB;
            buf_store_row<T, NV, PK_RELU>(r_out, oo, v);
        }
        if constexpr ((F & EP_SUMS) != 0) {
            if (flush) {
                write_chan_sums(a, b, slot, n, jbase, run, false);
#pragma unroll
                for (int e = 0; e < NV; ++e) run[e] = 0.f;
            }
        }
    }

    // ---- kernel 6, RC_OUT_NHWC_DWT: conv [+ ReLU / LeakyReLU] -> networks.DWTForward (models/networks.py:224-235) without the full-resolution map ------
    // The wave's strip (wave `tid >> 6` of an 8 x 32 tile: rows y0 + 2 w, + 1, columns x0 .. x0 + 31, one cout tile) is 16 whole 2 x 2 blocks per channel:
    // ONE row of 16 output pixels with 4 COUT_TILE channels.  The values are rounded to bf16 exactly as the NHWC store rounds them and parked, pixel-major, in
    // the wave's own LDS strip `st` (its halo strip, dead once the MFMA loop is over; the same wave writes and reads: the LDS queue is in order, no barrier);
    // lane (n, q) then reads the 2 x 2 block of output pixel n for its channel group q (the NV channels it computed) and forms, per channel, in the order of
    // dwt_forward_kernel's fma chain with the reference's frozen taps haar[k] = .5 {++++, ++--, +-+-, +--+} over (a b / c d):
    //     ((.5 a +- .5 b) +- .5 c) +- .5 d         (every product exact, so fma(in, tap, s) == s + in * tap)
    // as packed fp32 adds over channel pairs, and stores channels 4 (q NV) .. + 4 NV of the output pixel: 8 NV contiguous bytes.  Bit-identical to the two launches.
    // F == EP_RES: + the residual (an RCAGroup's closing conv + group skip, then DWT: LiteISP down2), prefetched into rp by res_prefetch as in epilogue_res_pre.
    template <int F>
    __device__ static __forceinline__ void epilogue_dwt(const ConvArgs& a, int b, int y0, int x0, int tid, f32x4 (&acc)[4][NT], char* st,
                                                        const unsigned (&rp)[(F & EP_RES) ? 4 : 1][(F & EP_RES) ? NV / 2 : 1]) {
        static_assert(ES == 2 && NV % 4 == 0 && (F == 0 || F == EP_RELU || F == EP_LEAKY || F == EP_RES), "bf16; plain / ReLU / LeakyReLU / + residual");
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, n = lane & 15;
        constexpr int REC = Cfg::COUT_TILE * 2 + 8, ROW = kTW * REC;      // bytes of a parked pixel (+ 8: the 2 x 2 reads of 16 lanes at a 2-pixel pitch spread over the banks) / strip row
        const float inf = __builtin_inff();
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            float v[NV];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[pt][nt][r];
            if constexpr ((F & EP_RELU) != 0) {
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.f, inf);
            }
            if constexpr ((F & EP_LEAKY) != 0) {
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], v[e] * a.act_slope, inf);
            }
            if constexpr ((F & EP_RES) != 0) {
#pragma unroll
                for (int i = 0; i < NV / 2; ++i) {
                    v[2 * i] += __uint_as_float(rp[pt][i] << 16);
                    v[2 * i + 1] += __uint_as_float(rp[pt][i] & 0xffff0000u);
                }
            }
            char* dst = st + (pt >> 1) * ROW + (16 * (pt & 1) + n) * REC + q * (NV * 2);
#pragma unroll
            for (int i = 0; i < NV / 4; ++i)
                *reinterpret_cast<uint2*>(dst + 8 * i) = make_uint2(pack_bf16x2(v[4 * i], v[4 * i + 1]), pack_bf16x2(v[4 * i + 2], v[4 * i + 3]));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");             // other lanes' writes are read back below: keep the compiler from moving the
        __builtin_amdgcn_wave_barrier();                                   // reads above them (the LDS queue itself is in order)
        f32x2 in[4][NV / 2];                                               // [2 i + j][channel pair], already times .5
        const char* src = st + (2 * n) * REC + q * (NV * 2);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < NV / 4; ++i) {
                const uint2 w = *reinterpret_cast<const uint2*>(src + (t >> 1) * ROW + (t & 1) * REC + 8 * i);
                in[t][2 * i] = f32x2{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u)} * 0.5f;
                in[t][2 * i + 1] = f32x2{__uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)} * 0.5f;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");             // ... and the next strip's commit (same buffer) below these reads
        __builtin_amdgcn_wave_barrier();
        unsigned o[2 * NV];                                                // channels 4 e + k of the output pixel, e = q NV + (0 .. NV - 1): bf16 pairs (k 0 1), (k 2 3)
#pragma unroll
        for (int p = 0; p < NV / 2; ++p) {
            const f32x2 sab = in[0][p] + in[1][p], dab = in[0][p] - in[1][p];
            const f32x2 k0 = (sab + in[2][p]) + in[3][p], k1 = (sab - in[2][p]) - in[3][p];
            const f32x2 k2 = (dab + in[2][p]) - in[3][p], k3 = (dab - in[2][p]) + in[3][p];
            o[4 * p] = pack_bf16x2(k0.x, k1.x); o[4 * p + 1] = pack_bf16x2(k2.x, k3.x);
            o[4 * p + 2] = pack_bf16x2(k0.y, k1.y); o[4 * p + 3] = pack_bf16x2(k2.y, k3.y);
        }
        const int Hh = a.H >> 1, Wh = a.W >> 1, oy = (y0 >> 1) + wave, ox = (x0 >> 1) + n;
        const size_t img_out = (size_t)a.H * a.W * a.cout;                  // (H / 2) (W / 2) (4 cout): the same bytes per image
        const __amdgpu_buffer_rsrc_t r_out = make_rsrc(static_cast<T*>(a.out) + (size_t)b * img_out, (unsigned)(img_out * ES));
        const int oo = (oy < Hh && ox < Wh && !(a.dbg_flags & 1)) ? ((oy * Wh + ox) * (4 * a.cout) + 4 * q * NV) * ES : kOOB;
#pragma unroll
        for (int i = 0; i < NV / 2; ++i)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]}, r_out, oo, 16 * i, 0);
    }

    __device__ static __forceinline__ void epilogue_generic(const ConvArgs& a, int b, int y0, int x0, int sp, int ct, int tid,
                                                            f32x4 (&acc)[4][NT]) {
        const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, n = lane & 15;
        const int jbase = ct * Cfg::COUT_TILE + q * NV;
        float fs[NV], ft[NV];
        const bool film = a.film_scale != nullptr;
        if (film) {
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                const bool in = jbase + e < a.cout;
                fs[e] = in ? a.film_scale[(size_t)b * a.cout + jbase + e] : 0.f;
                ft[e] = in ? a.film_shift[(size_t)b * a.cout + jbase + e] : 0.f;
            }
        }
        const int gy_w = y0 + 2 * wave, gx_l = x0 + n;
        float csum[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) csum[e] = 0.f;

        const size_t img_out = (size_t)a.H * a.W * a.cout;    // NHWC output / residual / mul image (elements)
        const unsigned img_bytes_out = (unsigned)(img_out * ES);
        __amdgpu_buffer_rsrc_t r_out;
        if (a.out_mode == RC_OUT_NCHW || a.out_mode == RC_OUT_PIXEL_SHUFFLE2_NCHW) {
            const size_t plane = (size_t)a.out_h * a.out_w;
            const int osz = a.out_dtype == RC_F32 ? 4 : 2;
            const int planes = a.out_mode == RC_OUT_NCHW ? a.cout : a.cout >> 2;
            r_out = make_rsrc(static_cast<char*>(a.out) + (size_t)b * planes * plane * osz, (unsigned)(planes * plane * osz));
        } else {  // NHWC (H,W,cout) or pixel-shuffled (2H,2W,cout/4): same bytes per image
            r_out = make_rsrc(static_cast<T*>(a.out) + (size_t)b * img_out, img_bytes_out);
        }
        const bool full = a.cout == a.cout_packed;

#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int gy = gy_w + (pt >> 1);
            const int gx = gx_l + (pt & 1) * 16;
            const bool valid = gy < a.H && gx < a.W;
            float v[NV];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[pt][nt][r];
            if (film) {
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] = v[e] * fs[e] + ft[e] + v[e];
            }
            if (a.act == RC_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.f, __builtin_inff());
            } else if (a.act == RC_ACT_LEAKY) {
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.act_slope;
            } else if (a.act == RC_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] = gelu_erf_f32(v[e]);
            }
            const int pix_off = valid ? ((gy * a.W + gx) * a.cout + jbase) * ES : kOOB;   // NHWC-shaped operands
            if (a.mul_plus1 != nullptr) {
                float m[NV];
                const __amdgpu_buffer_rsrc_t r_mul = make_rsrc(static_cast<const T*>(a.mul_plus1) + (size_t)b * img_out, img_bytes_out);
                buf_load_row<T, NV>(r_mul, pix_off, m);
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] = v[e] * (m[e] + 1.f);
            }
            if (a.out_scale != nullptr) {
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] *= (jbase + e < a.cout) ? a.out_scale[(size_t)b * a.cout + jbase + e] : 0.f;
            }
            if (a.residual != nullptr) {
                float m[NV];
                const __amdgpu_buffer_rsrc_t r_res = make_rsrc(static_cast<const T*>(a.residual) + (size_t)b * img_out, img_bytes_out);
                buf_load_row<T, NV>(r_res, pix_off, m);
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] += m[e];
                if (a.act == RC_ACT_RELU_POST) {           // relu(conv + residual): CompressAI ResidualUnit
#pragma unroll
                    for (int e = 0; e < NV; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], 0.f, __builtin_inff());
                }
            }
            if (a.chan_sums != nullptr) {
#pragma unroll
                for (int e = 0; e < NV; ++e) csum[e] += valid ? v[e] : 0.f;
            }

            if (a.out_mode == RC_OUT_NHWC) {
                if (full) {
                    buf_store_row<T, NV, false>(r_out, pix_off, v);
                } else {  // ragged cout (test sizes): element stores
#pragma unroll
                    for (int e = 0; e < NV; ++e) {
                        const int o = (valid && jbase + e < a.cout) ? pix_off + e * ES : kOOB;
                        if constexpr (ES == 4) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[e]), r_out, o, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b16((unsigned short)Vec16<bf16_t>::rne(v[e]), r_out, o, 0, 0);
                    }
                }
            } else if (a.out_mode == RC_OUT_PIXEL_SHUFFLE2) {
                const int cps = a.cout >> 2, sub = ct & 3;
                const int o = valid ? (((2 * gy + (sub >> 1)) * (2 * a.W) + (2 * gx + (sub & 1))) * cps + (ct >> 2) * Cfg::COUT_TILE + q * NV) * ES : kOOB;
                buf_store_row<T, NV, false>(r_out, o, v);
            } else if (a.out_mode == RC_OUT_PIXEL_SHUFFLE2_NCHW) {
                // nn.PixelShuffle(2) + planar store: conv channel 4c + 2i + j -> out[b][c][2 gy + i][2 gx + j], cropped to (out_h, out_w).
                // A lane's 4 values of cout tile nt are the 2x2 sub-pixels of ONE out channel: two 2-element row pieces.
                const bool pair_ok = (a.out_w & 1) == 0;                       // uniform: keeps the 2-element stores naturally aligned
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int co = jbase + 4 * nt, c = co >> 2;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int Y = 2 * gy + i, X = 2 * gx;
                        const bool in0 = valid && co < a.cout && Y < a.out_h && X < a.out_w, in1 = in0 && X + 1 < a.out_w;
                        const int idx = (c * a.out_h + Y) * a.out_w + X;
                        const float v0 = v[4 * nt + 2 * i], v1 = v[4 * nt + 2 * i + 1];
                        if (a.out_dtype == RC_F32) {
                            if (pair_ok) __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__float_as_uint(v0), __float_as_uint(v1)}, r_out, in1 ? idx * 4 : kOOB, 0, 0);
                            else {
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), r_out, in0 ? idx * 4 : kOOB, 0, 0);
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), r_out, in1 ? idx * 4 + 4 : kOOB, 0, 0);
                            }
                        } else {
                            if (pair_ok) __builtin_amdgcn_raw_buffer_store_b32(pack_bf16x2(v0, v1), r_out, in1 ? idx * 2 : kOOB, 0, 0);
                            else {
                                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)Vec16<bf16_t>::rne(v0), r_out, in0 ? idx * 2 : kOOB, 0, 0);
                                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)Vec16<bf16_t>::rne(v1), r_out, in1 ? idx * 2 + 2 : kOOB, 0, 0);
                            }
                        }
                    }
                }
            } else {  // RC_OUT_NCHW, cropped
                const bool inside = gy < a.out_h && gx < a.out_w;
#pragma unroll
                for (int e = 0; e < NV; ++e) {
                    const int co = jbase + e;
                    const int idx = (co * a.out_h + gy) * a.out_w + gx;
                    if (a.out_dtype == RC_F32)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[e]), r_out, (inside && co < a.cout) ? idx * 4 : kOOB, 0, 0);
                    else
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)Vec16<bf16_t>::rne(v[e]), r_out, (inside && co < a.cout) ? idx * 2 : kOOB, 0, 0);
                }
            }
        }
        if (a.chan_sums != nullptr) write_chan_sums(a, b, 4 * sp + wave, n, jbase, csum, true);
    }
};

// ==================================================================================================
// Kernel 1: general form.  One block = one (spatial tile, cout tile, image); loops over Cin chunks,
// streaming packed weights through LDS G steps at a time.
// ==================================================================================================
template <class Cfg, bool GATED, bool FAST>
__global__ __launch_bounds__(kThreads, 2) void conv_mfma_kernel(const ConvArgs a) {
    using D = ConvDev<Cfg>;
    using T = typename Cfg::elem;
    constexpr int NT = Cfg::NT, STEPS = Cfg::STEPS, G = Cfg::G, NSUB = Cfg::NSUB, NV = 4 * NT;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;                                  // weights first: their ds_read immediates stay below 64 KiB
    char* s_in = smem + Cfg::W_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int ct = blockIdx.x % a.n_ct;
    const int sp = blockIdx.x / a.n_ct;
    const int tx = sp % a.tiles_x, ty = sp / a.tiles_x;
    const int b = blockIdx.y;
    const int y0 = ty * kTH, x0 = tx * kTW;

    typename D::LaneOff lo;
    D::lane_offsets(q, lo);
    const int lane_x = ((2 * wave) * D::TWH + n) * D::SPIX;  // + pixel-tile immediates in load_frags
    const int lane_w = lane * 16;

    f32x4 acc[4][NT];                         // initial C operand = bias (packed order, padded to cout_packed)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
            const float4 t4 = *reinterpret_cast<const float4*>(a.bias + ct * Cfg::COUT_TILE + q * NV + nt * 4);
            bv = f32x4{t4.x, t4.y, t4.z, t4.w};
        }
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) acc[pt][nt] = bv;
    }

    ConvArgs aa = a;
    if (ct != 0) aa.in_store = nullptr;       // only one cout tile materialises the gated input
    const typename D::TileSrc ts = D::tile_src(aa, b, y0, x0);
    typename D::TileOffs to;
    D::tile_offsets(aa, tid, to);
    const char* wbase = static_cast<const char*>(a.wpacked) + (size_t)ct * a.n_chunks * Cfg::CHUNK_W_BYTES;

    for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
        if (chunk > 0) __syncthreads();  // all waves done reading s_in / s_w of the previous chunk
        const char* wchunk = wbase + (size_t)chunk * Cfg::CHUNK_W_BYTES;
        D::dma_weights(wchunk, s_w, (STEPS < G ? STEPS : G) * NT, wave, lane_w);  // async, overlaps the tile loads
        if (a.cin_vec_ok) {
            uint4 r0[D::NI], r1[GATED ? D::NI : 1];
            float gv[GATED ? D::UNIT : 1];
            D::template load_tile<GATED>(aa, ts, to, b, chunk, tid, r0, r1, gv);
            D::template commit_tile<GATED>(aa, ts, to, chunk, tid, r0, r1, gv, s_in);
        } else {
            D::stage_tile_scalar(aa, b, y0, x0, chunk, tid, s_in, static_cast<T*>(aa.in_store));
        }
        __syncthreads();
        D::template mma_steps<0, G, 0>(s_in, s_w, lane_x, lane_w, q, lo, acc);
        if constexpr (NSUB > 1) {
            __syncthreads();
            D::dma_weights(wchunk + (size_t)G * NT * 1024, s_w, ((STEPS - G) < G ? (STEPS - G) : G) * NT, wave, lane_w);
            __syncthreads();
            D::template mma_steps<G, G, G>(s_in, s_w, lane_x, lane_w, q, lo, acc);
        }
        if constexpr (NSUB > 2) {
            __syncthreads();
            D::dma_weights(wchunk + (size_t)2 * G * NT * 1024, s_w, ((STEPS - 2 * G) < G ? (STEPS - 2 * G) : G) * NT, wave, lane_w);
            __syncthreads();
            D::template mma_steps<2 * G, G, 2 * G>(s_in, s_w, lane_x, lane_w, q, lo, acc);
        }
        static_assert(NSUB <= 3, "add another weight sub-stage");
    }
    D::template epilogue<FAST>(a, b, y0, x0, sp, ct, tid, acc);
}

// ==================================================================================================
// Kernel 2: persistent form for single-chunk layers (Cin == CK): the 48->48 convolutions that dominate
// the flagship net, and 48->192 (+PixelShuffle).  A block walks a strided list of tiles; the NEXT tile's
// halo loads are issued into registers before the last MFMA loop of the current tile, so HBM latency hides
// under compute.  With one cout tile the whole packed weight matrix stays in LDS for the block's lifetime;
// with several, the input tile stays resident and the weights of each cout tile are DMA'd in turn.
// Two such blocks share a CU and drift out of phase (one in MFMA while the other stores / stages).
// ==================================================================================================
constexpr int kPersistMaxCout = 512;   // bias slots kept in LDS
// the 5x5 form (the folded tail, <= 16 couts): 64 slots, so that halo tile (12 x 36) + 38 KiB of weights still fit twice per CU
template <class Cfg>
constexpr int persist_bias_slots() { return Cfg::KS == 5 ? 64 : kPersistMaxCout; }
template <class Cfg>
constexpr int persist_lds_bytes_c() { return Cfg::IN_BYTES + (int)Cfg::CHUNK_W_BYTES + persist_bias_slots<Cfg>() * 4; }

// blocks per CU: layers with one 16-wide cout tile (the 48 -> 3 output conv) do almost no math per byte, so what
// matters is bytes in flight: three blocks (their accumulators are small enough for 168 VGPRs)
template <class Cfg>
constexpr int persist_blocks_per_cu() { return Cfg::NT <= 2 && persist_lds_bytes_c<Cfg>() * 3 <= 160 * 1024 ? 3 : 2; }

template <class Cfg, bool GATED, bool FAST>
__global__ __launch_bounds__(kThreads, persist_blocks_per_cu<Cfg>()) void conv_mfma_persist_kernel(const ConvArgs a) {
    using D = ConvDev<Cfg>;
    constexpr int NT = Cfg::NT, STEPS = Cfg::STEPS, NV = 4 * NT;

    // w_resident (uniform): the packed weights of ALL n_ct cout tiles stay in LDS for the block's lifetime (small-Cin layers: the codec's 4 -> 128 head is 2 x 12 KB).
    // Re-staging one tile's worth per (pixel tile, cout tile) -- an L2 round trip between two workgroup barriers, 135 x 2 times per block -- was what that layer
    // waited for: 2.17 ms to write 4.5 GB.
    const int w_copies = a.w_resident ? a.n_ct : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;                                  // weights first: their ds_read immediates stay below 64 KiB
    float* s_bias = reinterpret_cast<float*>(smem + w_copies * Cfg::CHUNK_W_BYTES);
    char* s_in = smem + w_copies * Cfg::CHUNK_W_BYTES + persist_bias_slots<Cfg>() * 4;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, n = lane & 15;
    typename D::LaneOff lo;
    D::lane_offsets(q, lo);
    const int lane_x = ((2 * wave) * D::TWH + n) * D::SPIX;
    const int lane_w = lane * 16;

    const int sp_total = a.tiles_x * a.tiles_y;
    const int n_tiles = sp_total * a.batch;
    const int n_ct = a.n_ct;
    // Tile order: at step k the grid covers the window [k*G, (k+1)*G) of consecutive (band-major) tile
    // indices (same image neighbourhood -> addresses spread over all HBM channels); inside the window XCD x
    // (blocks with blockIdx % 8 == x, observed placement -- speed only) takes a run of G/8 = 64 consecutive
    // indices = an 8x8 block of tiles, so its halos are served by one L2.
    const int slots = gridDim.x >> 3;                  // gridDim.x is a multiple of 8
    const int pos = (blockIdx.x & 7) * slots + (blockIdx.x >> 3);

    if (n_ct == 1 || a.w_resident) D::dma_weights(static_cast<const char*>(a.wpacked), s_w, STEPS * NT * w_copies, wave, lane_w);
    for (int i = tid; i < a.cout_packed; i += kThreads) s_bias[i] = a.bias ? a.bias[i] : 0.f;

    uint4 r0[D::NI], r1[GATED ? D::NI : 1];
    float gv[GATED ? D::UNIT : 1];
    typename D::TileOffs to;
    D::tile_offsets(a, tid, to);
    int tile = pos < n_tiles ? pos : -1;
    int b = 0, sp = 0, y0 = 0, x0 = 0;
    typename D::TileSrc ts;
    if (tile >= 0) {
        b = magic_div(tile, a.td.sp_total);
        int ty, tx;
        band_decode(tile - b * sp_total, a.tiles_x, a.tiles_y, a.td, ty, tx);
        sp = ty * a.tiles_x + tx; y0 = ty * kTH; x0 = tx * kTW;
        ts = D::tile_src(a, b, y0, x0);
        if (a.cin_vec_ok) {
            D::template load_gate<GATED>(a, b, 0, tid, gv);
            if (ts.interior) D::template load_tile_interior<GATED>(a, ts, to, 0, r0, r1);
        }
    }
    float run[NV];                                     // CALayer channel sums carried across this block's tiles
#pragma unroll
    for (int e = 0; e < NV; ++e) run[e] = 0.f;
    int covered = 0;                                   // compact sums: images [0, covered) have this block's 4 slots written (a run total or zeros)
    // the 3x3 kernels with 3-4 cout tiles (256-register budget) prefetch border tiles too; elsewhere the bounds-checked addressing spills
    constexpr bool BORDER_PRE = !GATED && Cfg::KS == 3 && (NT == 3 || NT == 4) && sizeof(typename Cfg::elem) == 2;
    bool first_tile = true;                            // the first tile's border loads are never prefetched
    while (tile >= 0) {
        __syncthreads();                               // every wave finished reading s_in / s_w (previous tile)
        // border tiles (5 % at 4K) were not prefetched: their bounds-checked addressing would otherwise sit,
        // as live masks and offsets, across the MFMA loop of every tile
        if (a.cin_vec_ok && !ts.interior && (!BORDER_PRE || first_tile || (a.dbg_flags & 16))) D::template load_tile_border<GATED>(a, ts, 0, tid, r0, r1);   // conv_flags 16: A/B
        first_tile = false;
        if (a.cin_vec_ok) D::template commit_tile<GATED>(a, ts, to, 0, tid, r0, r1, gv, s_in);
        else D::stage_tile_scalar(a, b, y0, x0, 0, tid, s_in, static_cast<typename Cfg::elem*>(a.in_store));
        constexpr bool RUN_SUMS = FAST && sizeof(typename Cfg::elem) == 2 && NT >= 3;   // sums carried across the block's tiles (narrow tiles: per-tile sums; their
                                                                                        // 168-register instantiations have no room for the carried values)
        const int cb = b, csp = sp, cy0 = y0, cx0 = x0;
        const int next = tile + (int)gridDim.x;
        tile = next < n_tiles ? next : -1;
        if constexpr (RUN_SUMS) {
            // RUN form below: a tile whose sums are CARRIED on leaves zeros in its slots.  A tile that flushes (the block's next tile is another image's, or
            // there is none) writes its slots itself in the epilogue: no zero store for it -- nothing then depends on a zero store and a flush store to the
            // same address being ordered (they were, through the fences of the two __syncthreads() between them, but only by that).
            const bool will_flush = tile < 0 || magic_div(tile, a.td.sp_total) != cb;                                             // uniform
            if ((a.ep_key == D::EP_SUMS || a.ep_key == (D::EP_RELU | D::EP_SUMS)) && n_ct == 1) {
                if (!a.sums_compact) { if (!will_flush) D::zero_sum_slots(a, cb, csp, tid); }
                else if (will_flush) {                 // compact layout: residue classes this block has no tile of, in the images it skipped, get zeros
                    for (int bb = covered; bb < cb; ++bb) D::zero_slots4(a, bb, 4 * D::compact_residue(pos, bb, sp_total, (int)gridDim.x), tid);
                    covered = cb + 1;
                }
            }
        }

        for (int ct = 0; ct < n_ct; ++ct) {
            if (n_ct > 1 && !a.w_resident) {
                if (ct > 0) __syncthreads();           // previous cout tile's weights fully consumed
                D::dma_weights(static_cast<const char*>(a.wpacked) + (size_t)ct * Cfg::CHUNK_W_BYTES, s_w, STEPS * NT, wave, lane_w);
            }
            if (ct == 0 || !a.w_resident) __syncthreads();   // input tile, weights (+ first time: bias) visible
            if (ct == n_ct - 1 && tile >= 0) {         // prefetch the next tile: in flight during the MFMA loop
                b = magic_div(tile, a.td.sp_total);
                int ty, tx;
                band_decode(tile - b * sp_total, a.tiles_x, a.tiles_y, a.td, ty, tx);
                sp = ty * a.tiles_x + tx; y0 = ty * kTH; x0 = tx * kTW;
                ts = D::tile_src(a, b, y0, x0);
                if (a.cin_vec_ok) {
                    D::template load_gate<GATED>(a, b, 0, tid, gv);
                    if (ts.interior) D::template load_tile_interior<GATED>(a, ts, to, 0, r0, r1);
                    else if constexpr (BORDER_PRE) { if (!(a.dbg_flags & 16)) D::template load_tile_border<GATED>(a, ts, 0, tid, r0, r1); }   // plain input: border tiles are prefetched too
                }                                                                                        // (their offsets die with the issue; 9-18 % of the tiles at levels 1-2)
            }
            f32x4 acc[4][NT];                          // initial C operand = bias
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 t4 = *reinterpret_cast<const float4*>(s_bias + ct * Cfg::COUT_TILE + q * NV + nt * 4);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[pt][nt] = f32x4{t4.x, t4.y, t4.z, t4.w};
            }
            constexpr bool RESPRE = FAST && !GATED && sizeof(typename Cfg::elem) == 2 && NT <= 3;   // NT 4: the 32 extra registers spill
            [[maybe_unused]] unsigned rpre[RESPRE ? 4 : 1][RESPRE ? D::NRH : 1];
            bool res_pre = false;
            if constexpr (RESPRE) {
                if ((a.ep_key == D::EP_RES || a.ep_key == (D::EP_GATE | D::EP_RES)) && a.out_mode == RC_OUT_NHWC) {   // uniform: the residual's loads go out before the MFMA loop
                    res_pre = true;
                    D::res_prefetch(a, cb, cy0, cx0, ct, tid, rpre);
                }
            }
            D::template mma_steps<0, STEPS, 0, (!GATED && NT < 5)>(s_in, a.w_resident ? s_w + ct * (int)Cfg::CHUNK_W_BYTES : s_w, lane_x, lane_w, q, lo, acc);
            if constexpr (FAST && sizeof(typename Cfg::elem) == 2) {
                const int rslot = a.sums_compact ? 4 * D::compact_residue(pos, cb, sp_total, (int)gridDim.x) : -1;                 // uniform
                if (RUN_SUMS && a.ep_key == D::EP_SUMS && n_ct == 1)    // uniform
                    D::template epilogue_fast_impl<D::EP_SUMS, RUN_SUMS>(a, cb, cy0, cx0, csp, ct, tid, acc, run, tile < 0 || b != cb, rslot);
                else if (RUN_SUMS && a.ep_key == (D::EP_RELU | D::EP_SUMS) && n_ct == 1)
                    D::template epilogue_fast_impl<D::EP_RELU | D::EP_SUMS, RUN_SUMS>(a, cb, cy0, cx0, csp, ct, tid, acc, run, tile < 0 || b != cb, rslot);
                else if (res_pre) {
                    if constexpr (RESPRE) {
                        D::epilogue_res_pre(a, cb, cy0, cx0, ct, tid, acc, rpre, a.ep_key == (D::EP_GATE | D::EP_RES) ? a.out_scale + (size_t)cb * a.cout : nullptr);
                    }
                } else D::template epilogue<FAST>(a, cb, cy0, cx0, csp, ct, tid, acc);
            } else D::template epilogue<FAST>(a, cb, cy0, cx0, csp, ct, tid, acc);
        }
    }
    if constexpr (FAST && sizeof(typename Cfg::elem) == 2 && NT >= 3) {
        // compact sums: the images after this block's last one (and all of them, for a block without tiles) still need their 4 slots
        if (a.sums_compact && (a.ep_key == D::EP_SUMS || a.ep_key == (D::EP_RELU | D::EP_SUMS)) && n_ct == 1)
            for (int bb = covered; bb < a.batch; ++bb) D::zero_slots4(a, bb, 4 * D::compact_residue(pos, bb, sp_total, (int)gridDim.x), tid);
    }
}

// ==================================================================================================
// Kernel 3: producer/consumer ("wave-specialised") persistent form for single-chunk, single-cout-tile
// layers.  One block of 8 waves per CU: waves 0-3 compute (MFMA loop + epilogue, nothing else), waves 4-7
// load (halo-tile loads two tiles ahead into registers, CALayer gate + skip combine, skip-tensor store, LDS
// writes).  The LDS input tile is double-buffered, the packed weights and bias stay resident, and the two
// roles meet at exactly ONE workgroup barrier per tile:
//     barrier k:   loaders have written tile k+1 into buf[(k+1)&1];  computers have finished tile k (buf[k&1])
// so staging VALU work, HBM latency and LDS writes all sit beside the matrix pipe instead of in front of it.
// Each SIMD hosts one compute wave and one loader wave.
// ==================================================================================================
constexpr int kWsThreads = 512;

template <class Cfg, bool GATED, bool FAST>
__global__ __launch_bounds__(kWsThreads) void conv_mfma_ws_kernel(const ConvArgs a) {
    using D = ConvDev<Cfg>;
    constexpr int NT = Cfg::NT, STEPS = Cfg::STEPS, NV = 4 * NT;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;                                  // weights first: their ds_read immediates stay below 64 KiB
    float* s_bias = reinterpret_cast<float*>(smem + Cfg::CHUNK_W_BYTES);
    char* s_buf0 = smem + Cfg::CHUNK_W_BYTES + 16 * Cfg::NT * 4;
    char* s_buf1 = s_buf0 + Cfg::IN_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform
    const bool loader = wave8 >= 4;
    const int rtid = tid & 255, wave = wave8 & 3;                    // thread / wave index inside the role
    const int q = lane >> 4, n = lane & 15;

    const int sp_total = a.tiles_x * a.tiles_y;
    const int n_tiles = sp_total * a.batch;
    const int slots = gridDim.x >> 3;                                // gridDim.x is a multiple of 8
    const int pos = (blockIdx.x & 7) * slots + (blockIdx.x >> 3);
    const int stride = (int)gridDim.x;
    const int my_tiles = pos < n_tiles ? (n_tiles - pos + stride - 1) / stride : 0;

    // weights + bias: every wave helps (1 KiB per wave-instruction); drained by the first barrier
    for (int kb = wave8; kb < STEPS * NT; kb += kWsThreads / 64)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(static_cast<const char*>(a.wpacked) + kb * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(s_w + kb * 1024), 16, 0, 0);
    for (int i = tid; i < a.cout_packed; i += kWsThreads) s_bias[i] = a.bias ? a.bias[i] : 0.f;

    auto decode = [&](int tile, int& b, int& sp, int& y0, int& x0) {
        b = magic_div(tile, a.td.sp_total);
        int ty, tx;
        band_decode(tile - b * sp_total, a.tiles_x, a.tiles_y, a.td, ty, tx);
        sp = ty * a.tiles_x + tx; y0 = ty * kTH; x0 = tx * kTW;
    };

    if (loader) {
        // ---------------------------------------------------------------- producer waves
        uint4 r0[D::NI], r1[GATED ? D::NI : 1];
        float gv[GATED ? D::UNIT : 1];
        typename D::TileSrc ts;
        typename D::TileOffs to;
        D::tile_offsets(a, rtid, to);
        int b, sp, y0, x0;
        if (my_tiles > 0) {                              // tile 0 -> buf0 (synchronously), then tile 1's loads in flight
            decode(pos, b, sp, y0, x0);
            ts = D::tile_src(a, b, y0, x0);
            D::template load_tile<GATED>(a, ts, to, b, 0, rtid, r0, r1, gv);
            D::template commit_tile<GATED>(a, ts, to, 0, rtid, r0, r1, gv, s_buf0);
        }
        if (my_tiles > 1) {
            decode(pos + stride, b, sp, y0, x0);
            ts = D::tile_src(a, b, y0, x0);
            D::template load_tile<GATED>(a, ts, to, b, 0, rtid, r0, r1, gv);
        }
        __syncthreads();                                 // barrier 0: weights, bias, tile 0 visible
        const bool rec = a.dbg != nullptr && blockIdx.x == 8 && wave8 == 4 && lane == 0;
        for (int k = 0; k < my_tiles; ++k) {
            const long long tl0 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            if (k + 1 < my_tiles) {                      // registers hold tile k+1 -> the buffer the computers are NOT reading
                D::template commit_tile<GATED>(a, ts, to, 0, rtid, r0, r1, gv, ((k + 1) & 1) ? s_buf1 : s_buf0);
                if (k + 2 < my_tiles) {                  // ... and immediately put tile k+2's loads in flight
                    decode(pos + (k + 2) * stride, b, sp, y0, x0);
                    ts = D::tile_src(a, b, y0, x0);
                    if (!(a.dbg_flags & 4)) D::template load_tile<GATED>(a, ts, to, b, 0, rtid, r0, r1, gv);
                }
            }
            const long long tl1 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            __syncthreads();                             // barrier k+1
            if (rec && k < 64) { a.dbg[256 + 2 * k] = tl1 - tl0; a.dbg[256 + 2 * k + 1] = (long long)__builtin_amdgcn_s_memtime() - tl1; }
        }
    } else {
        // ---------------------------------------------------------------- consumer waves
        typename D::LaneOff lo;
        D::lane_offsets(q, lo);
        const int lane_x = ((2 * wave) * D::TWH + n) * D::SPIX;
        const int lane_w = lane * 16;
        __syncthreads();                                 // barrier 0
        const bool rec = a.dbg != nullptr && blockIdx.x == 8 && wave8 == 0 && lane == 0;
        for (int k = 0; k < my_tiles; ++k) {
            const long long tc0 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            int b, sp, y0, x0;
            decode(pos + k * stride, b, sp, y0, x0);
            f32x4 acc[4][NT];                            // initial C operand = bias
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 t4 = *reinterpret_cast<const float4*>(s_bias + q * NV + nt * 4);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[pt][nt] = f32x4{t4.x, t4.y, t4.z, t4.w};
            }
            if (!(a.dbg_flags & 2)) D::template mma_steps<0, STEPS, 0, (NT < 5)>((k & 1) ? s_buf1 : s_buf0, s_w, lane_x, lane_w, q, lo, acc);
            const long long tc1 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            D::template epilogue<FAST>(a, b, y0, x0, sp, 0, rtid, acc);
            const long long tc2 = rec ? (long long)__builtin_amdgcn_s_memtime() : 0;
            __syncthreads();                             // barrier k+1
            if (rec && k < 64) { a.dbg[4 * k] = tc1 - tc0; a.dbg[4 * k + 1] = tc2 - tc1; a.dbg[4 * k + 2] = (long long)__builtin_amdgcn_s_memtime() - tc2; }
        }
    }
}

// ==================================================================================================
// Kernel 4: producer/consumer form for layers whose weights do not fit LDS at once: several Cin chunks (the
// 128/192/512-channel levels of the U-Net) and/or several cout tiles (48 -> 192 + PixelShuffle).  Calibration (rc_debug_mfma_peak): ONE wave per SIMD issuing nothing but independent
// MFMAs sustains 59 % of the matrix peak, two waves 91 % -- so this kernel runs 8 compute waves (two per SIMD) on
// a 16 x 32 pixel tile beside 4 loader waves, 12 waves = 768 threads, one block per CU.  The block walks a flat
// list of stages = (tile, cout tile, Cin chunk) -- a tile's cout tiles and chunks back to back, so its input comes
// from this XCD's L2 after the first touch; every stage has two halves separated by a barrier:
//
//            compute waves 0-7                          loader waves 8-11
//   half a   MFMA steps [0, SA) of stage g   (Wa, in[g&1])     LDS-DMA Wb(g) -> its half; issue the loads of tile(g+1)
//   barrier
//   half b   MFMA steps [SA, STEPS)          (Wb, in[g&1])     LDS-DMA Wa(g+1) -> its half; tile(g+1) registers -> LDS in[(g+1)&1]
//            (+ epilogue on an item's last chunk)
//   barrier
//
// so the packed weights are single-buffered BY HALVES (each half is rewritten while the other is being read) and
// only the input tile is double-buffered: 36 + 2 x 57 KB of LDS.  (Rounds 2-4 moved the weights through the loaders' registers,
// issued one half-stage ahead, because hipcc's barrier waits vmcnt(0) with an LDS-DMA in flight and then also drains the tile's loads;
// round 5 made them DMA everywhere -- at the power cap the register trip costs more than the wait, see the loader -- and gave the
// layers whose stages are too thin for this scheme kernel 4b.)  The general kernel (one block per (tile, cout tile), loads
// in front of the MFMAs, weights by LDS-DMA in sub-stages) left the matrix pipe at ~50 % on these layers.
// For the epilogue each half of the block is an ordinary 8 x 32 tile: same code, same CALayer partial-sum slots.
// ==================================================================================================
constexpr int kWsmThreads = 768, kWsmTH = 16, kWsmCompute = 512;
template <class Cfg>
using WsmCfg = ConvCfg<typename Cfg::elem, Cfg::CK, Cfg::NT, Cfg::KS, kWsmTH>;
template <class Cfg>
constexpr int wsm_lds_bytes() { return (int)Cfg::CHUNK_W_BYTES + kPersistMaxCout * 4 + 2 * WsmCfg<Cfg>::IN_BYTES; }

template <class Cfg8, bool GATED, bool FAST>
__global__ __launch_bounds__(kWsmThreads) void conv_mfma_wsm_kernel(const ConvArgs a) {
    using Cfg = WsmCfg<Cfg8>;
    using D = ConvDev<Cfg>;                                          // staging + MFMA pieces on the 16-row tile
    using D8 = ConvDev<Cfg8>;                                        // epilogue: each half is an 8-row tile
    constexpr int NT = Cfg::NT, STEPS = Cfg::STEPS, NV = 4 * NT;
    constexpr int SA = (STEPS + 1) / 2;                              // steps in half a
    constexpr int WA = SA * NT * 1024, WB = (STEPS - SA) * NT * 1024, WALL = (int)Cfg::CHUNK_W_BYTES;
    static_assert(STEPS >= 2 && WA + WB == WALL, "weight halves");
    constexpr bool FOLD_SKIP = Cfg::KS == 2 && Cfg::UPT == 4 && STEPS == 4 && sizeof(typename Cfg::elem) == 2;   // a tap = one MFMA step (CK = 32, the form cfold % 64 == 0 layers take)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;                                                // weights first: ds_read immediates < 64 KiB
    float* s_bias = reinterpret_cast<float*>(smem + WALL);
    char* s_in0 = smem + WALL + kPersistMaxCout * 4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave12 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave12 >= 8;
    const int q = lane >> 4, n = lane & 15;

    const int tiles_y = (a.H + kWsmTH - 1) / kWsmTH;                 // 16-row tiles (a.tiles_y counts 8-row tiles)
    const int sp_total = a.tiles_x * tiles_y;
    const int n_tiles = sp_total * a.batch;
    const int n_chunks = a.n_chunks, n_ct = a.n_ct;
    // Work units.  One Cin chunk: unit = tile, staged once and reused by all its cout tiles.  Several chunks: every
    // stage stages a fresh chunk anyway, so unit = (tile, cout tile), cout tile fastest -- finer grains balance the
    // small deep levels (72 tiles per image at 136 x 240) and the run of units one XCD takes shares tiles through L2.
    const bool one_chunk = n_chunks == 1;
    const int n_units = one_chunk ? n_tiles : n_tiles * n_ct;
    const int cts_per_unit = one_chunk ? n_ct : 1;
    const int slots = gridDim.x >> 3;                               // XCD x takes a run of consecutive units
    const int pos = (blockIdx.x & 7) * slots + (blockIdx.x >> 3);
    const int stride = (int)gridDim.x;
    const int my_units = pos < n_units ? (n_units - pos + stride - 1) / stride : 0;
    const int my_stages = my_units * cts_per_unit * n_chunks;

    for (int i = tid; i < a.cout_packed; i += kWsmThreads) s_bias[i] = a.bias ? a.bias[i] : 0.f;

    auto decode = [&](int unit, int& b, int& ty, int& tx, int& ct0) {
        int tile = unit;
        ct0 = 0;
        if (!one_chunk) { tile = magic_div(unit, a.div_n_ct); ct0 = unit - tile * n_ct; }
        b = magic_div(tile, a.td_wsm.sp_total);
        band_decode(tile - b * sp_total, a.tiles_x, tiles_y, a.td_wsm, ty, tx);
    };

    if (loader) {
        // ---------------------------------------------------------------- producer waves
        const int rtid = tid - kWsmCompute;                          // 0..255
        // The packed weights go global -> LDS by LDS-DMA in EVERY form (round 5; rounds 2-4: through the loaders' registers except in the gated form): a half's DMA is
        // issued at the start of the half-stage in which its LDS region is free and landed by the barrier that ends it.  Rounds 2-4 avoided that because the barrier's
        // vmcnt(0) then also waits for the tile loads of the same half -- a cycle cost -- but these kernels run at the board's power cap with cycles to spare,
        // and not moving 36 KB per stage through VGPRs and ds_write_b128 saves joules: the multi-chunk layers of cfg3 18.04 -> 17.83 ms (tools/wsm_probe.py, two runs each),
        // and the loaders' 48 weight registers (8-10 spilled in the one-chunk 48-channel form) are gone.
        uint4 r0[D::NI], r1[GATED ? D::NI : 1];
        float gv[GATED ? D::UNIT : 1];
        typename D::TileSrc ts;
        typename D::TileOffs to;
        D::tile_offsets(a, rtid, to);
        auto dma = [&](int lds_off, int bytes, int goff) {           // packed weights, global -> LDS, 1 KiB per wave-instruction
            for (int kb = wave12 - 8; kb < bytes / 1024; kb += 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(static_cast<const char*>(a.wpacked) + goff + lds_off + kb * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(s_w + lds_off + kb * 1024), 16, 0, 0);
        };
        ConvArgs aa = a;                                             // per-item view: only ct == 0 materialises a gated input

        int k_unit = 0, cti = 0, ct = 0, chunk = 0, gi = 0;          // the stage whose tile loads are issued next
        int b = 0, ty = 0, tx = 0, ct0 = 0;
        int wsoff = 0, c_chunk = 0, c_buf = 0;                       // of the stage held in registers
        bool c_tile = false;
        auto issue_a = [&]() {                                       // input tile of the next stage -> registers
            if (chunk == 0) {
                if (cti == 0) decode(pos + k_unit * stride, b, ty, tx, ct0);
                ct = ct0 + cti;
                aa.in_store = ct == 0 ? a.in_store : nullptr;
                ts = D::tile_src(aa, b, ty * kWsmTH, tx * kTW);
            }
            c_tile = !one_chunk || cti == 0;
            c_buf = one_chunk ? (k_unit & 1) : (gi & 1);
            c_chunk = chunk;
            if (c_tile) D::template load_tile<GATED>(aa, ts, to, b, chunk, rtid, r0, r1, gv);
            wsoff = (ct * n_chunks + chunk) * WALL;
            ++gi;
            if (++chunk == n_chunks) { chunk = 0; if (++cti == cts_per_unit) { cti = 0; ++k_unit; } }
        };
        auto commit_a = [&]() {
            if (c_tile) D::template commit_tile<GATED>(aa, ts, to, c_chunk, rtid, r0, r1, gv, s_in0 + c_buf * Cfg::IN_BYTES);
        };
        // folded stride-2 layers: the chunks of phases 0 and 1 (source row 2y) have no weights in half a = the taps of map row y - 1 (see the compute waves)
        auto need_wa = [&]() { if constexpr (FOLD_SKIP) return !(D::folded(a) && D::fold_phase(a, c_chunk * Cfg::CK) < 2); else return true; };

        if (my_stages > 0) {
            issue_a(); commit_a();
            if (need_wa()) dma(0, WA, wsoff);                        // Wa(0)
        }
        __syncthreads();                                             // barrier 0: bias, Wa(0), tile(0) visible
        for (int g = 0; g < my_stages; ++g) {
            // the tile's loads are issued in half a and written to LDS in half b; each half's weights land during the half before the one that reads them
            const int wsoff_g = wsoff;                               // stage g's weights (issue_a moves wsoff on to g+1)
            dma(WA, WB, wsoff_g);                                    // half a: Wb(g) straight into its (free) LDS half; the computers read Wa(g) ...
            if (g + 1 < my_stages) issue_a();                        //         ... and the loads of tile(g+1) go out
            __syncthreads();
            if (g + 1 < my_stages) {                                 // half b: Wa(g+1) into its half, tile(g+1) -> LDS
                if (need_wa()) dma(0, WA, wsoff);
                commit_a();
            }
            __syncthreads();
        }
    } else {
        // ---------------------------------------------------------------- consumer waves (two per SIMD)
        const int wave = wave12;                                     // 0..7: rows 2*wave, 2*wave+1 of the 16-row tile
        typename D::LaneOff lo;
        D::lane_offsets(q, lo);
        const int lane_x = ((2 * wave) * D::TWH + n) * D::SPIX;
        const int lane_w = lane * 16;
        __syncthreads();                                             // barrier 0
        int g = 0;
        for (int k = 0; k < my_units; ++k) {
            int b, ty, tx, ct0;
            decode(pos + k * stride, b, ty, tx, ct0);
            const int ty8 = 2 * ty + (wave >> 2);                    // each half of the block stores as an ordinary 8 x 32 tile
            for (int ct = ct0; ct < ct0 + cts_per_unit; ++ct) {
                f32x4 acc[4][NT];                                    // initial C operand = bias
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float4 t4 = *reinterpret_cast<const float4*>(s_bias + ct * Cfg::COUT_TILE + q * NV + nt * 4);
#pragma unroll
                    for (int pt = 0; pt < 4; ++pt) acc[pt][nt] = f32x4{t4.x, t4.y, t4.z, t4.w};
                }
                // chunks [c0, c1) of the item, all of fold phase PH (3 = every tap carries weights: the un-folded layers)
                auto run_chunks = [&](auto PH, int c0, int c1) {
                    constexpr int ph = decltype(PH)::value;
                    for (int c = c0; c < c1; ++c, ++g) {
                        const char* s_in = s_in0 + (one_chunk ? (k & 1) : (g & 1)) * Cfg::IN_BYTES;
                        if constexpr (ph == 3) { if (!(a.dbg_flags & 2)) D::template mma_steps<0, SA, 0, true>(s_in, s_w, lane_x, lane_w, q, lo, acc); }
                        else if constexpr (ph == 2) D::template mma_steps<1, 1, 0, false>(s_in, s_w, lane_x, lane_w, q, lo, acc);
                        __syncthreads();
                        if constexpr ((ph & 1) != 0) { if (!(a.dbg_flags & 2)) D::template mma_steps<SA, STEPS - SA, 0, true>(s_in, s_w, lane_x, lane_w, q, lo, acc); }
                        else D::template mma_steps<3, 1, 0, false>(s_in, s_w, lane_x, lane_w, q, lo, acc);
                        if (c + 1 < n_chunks) __syncthreads();
                    }
                };
                if constexpr (FOLD_SKIP) {
                    // Folded stride-2 3x3 layer (a.fold2): the chunks come phase by phase of the 2x2 un-shuffle (cfold % CK == 0), and of the {-1, 0}^2 window only the
                    // taps with (ty == 1 || i == 1) && (tx == 1 || j == 1) carry weights for phase (i, j) -- 1 / 2 / 2 / 4 of the 4 taps, 9 of 16 over a pixel; the others
                    // would multiply by the zeros _stride2_view packs, so they are not issued (a tap = one step here: half a = taps (0,0) (0,1), half b = (1,0) (1,1)).
                    // Four loops with a fixed instruction stream each, not one loop with a branch per chunk: with alternative MFMA streams under if / else inside the
                    // loop hipcc spilled 33-135 registers of the 168 this 768-thread kernel has.
                    const int cpp = D::folded(a) ? n_chunks >> 2 : 0;
                    run_chunks(std::integral_constant<int, 0>{}, 0, cpp);
                    run_chunks(std::integral_constant<int, 1>{}, cpp, 2 * cpp);
                    run_chunks(std::integral_constant<int, 2>{}, 2 * cpp, 3 * cpp);
                    run_chunks(std::integral_constant<int, 3>{}, 3 * cpp, n_chunks);
                } else {
                    run_chunks(std::integral_constant<int, 3>{}, 0, n_chunks);
                }
                if (ty8 < a.tiles_y)
                    D8::template epilogue<FAST>(a, b, ty8 * kTH, tx * kTW, ty8 * a.tiles_x + tx, ct, tid & 255, acc);
                __syncthreads();
            }
        }
    }
}

// ==================================================================================================
// Kernel 4b: kernel 4 with ONE barrier per stage.  Kernel 4's stage is two halves, each ended by a barrier that waits for an LDS-DMA of weights issued at its
// start, with the tile's loads issued at the start of half a and written to LDS in half b: a stage cannot be shorter than two memory latencies in series.  Where
// the packed weights of a chunk can be double-buffered WHOLE beside the tile buffers (and the pixels can sit dense in LDS: 2- and 4-unit chunks), everything stage
// g + 1 reads is requested at the start of stage g and waited for once.  Two forms (wst_form() below):
//   form 3 -- THIN stages, <= 20 KB of weights a chunk: the folded stride-2 layers (ksize 2, 32-channel chunks: 64 MFMAs per wave and stage; measured 2.3 us per
//             stage in kernel 4 where the MFMAs need 0.85), the 16/32-wide cout tiles (128 -> 12, 320 -> 224), the 16-channel chunks.  Three tile buffers, filled by
//             LDS-DMA TWO stages ahead; the scheme drawn below.
//   form 2 -- <= 36 KB of weights a chunk: the 64/48-wide cout tiles of the 3x3 layers over 32-channel chunks (cfg3's 128 / 192 / 512-channel levels).  Two tile
//             buffers; the loader waves fetch tile(g+1) into registers and W(g+1) by LDS-DMA back to back, wait, write the tile, meet the barrier.
// Here a stage is ONE barrier and nothing waits for a load younger than a stage:
//
//            compute waves 0-7                                          tile waves 8-11
//   stage g  MFMA steps of stage g   (W[g&1], in[g % 3])                LDS-DMA W(g+1) -> W[(g+1)&1], then tile(g+2) -> in[(g+2) % 3]  (NDW instructions a wave);
//            (+ epilogue on an item's last chunk)                       s_waitcnt vmcnt(NDW): everything older -- W(g+1), tile(g+1) -- has landed
//   barrier                                                             barrier (raw s_barrier: the newest tile stays in flight across it)
//
// (The compute waves issue no LDS-DMA: with one in flight hipcc puts vmcnt(0) in front of their next LDS read -- it cannot tell the regions apart.)
//
// The tile goes global -> LDS by `buffer_load ... lds` (out-of-range pieces land zeros = the zero padding), which writes 64 consecutive 16-byte
// pieces per instruction: the pixels sit DENSE in LDS here (WstCfg::SPIX = the chunk's bytes; conflict-free for the 2- and 4-unit chunks this
// kernel takes: the four lane groups of a fragment read the four pieces of one pixel or of two neighbouring ones).  No staging registers, no
// ds_write, and the wait is an instruction COUNT the code states itself (the pieces an interior tile's wave has just issued; zero after a border
// tile or a stage that fetches none) -- left to hipcc's own waitcnt insertion a two-deep register pipeline collapsed to vmcnt(0..1) at the first
// uncountable branch.  Work decomposition, tile order, accumulation order and epilogue are kernel 4's: same bits.
// ==================================================================================================
template <class Cfg8>
struct WstCfg : WsmCfg<Cfg8> {
    using B = WsmCfg<Cfg8>;
    static constexpr int SPIX = B::CK * (int)sizeof(typename B::elem);              // dense pixels
    static constexpr int N_PIECES = B::THH * B::TWH * B::UPT;                       // 16-byte pieces of a halo tile
    static constexpr int N_DMA = (N_PIECES + 63) / 64;                              // wave-instructions per tile
    static constexpr int IN_BYTES = N_DMA * 1024;
};
// form 3: weights <= 20 KB a chunk, THREE tile buffers filled by LDS-DMA two stages ahead;  form 2: weights <= 36 KB a chunk (the 64-wide cout tiles of the 3x3
// layers over 32-channel chunks: cfg3's 128 / 192 / 512-channel levels), TWO tile buffers, the tile through the loaders' registers one stage ahead;  0: kernel 4 only
template <class Cfg>
constexpr int wst_form() {
    if (!(sizeof(typename Cfg::elem) == 2 && (Cfg::KS == 2 || Cfg::KS == 3) && (Cfg::UPT == 2 || Cfg::UPT == 4) && Cfg::STEPS >= 2)) return 0;
    const int w = (int)Cfg::CHUNK_W_BYTES, in = WstCfg<Cfg>::IN_BYTES, fixed = 2 * w + kPersistMaxCout * 4;
    if (w <= 20 * 1024 && fixed + 3 * in <= 160 * 1024) return 3;
    if (w <= 36 * 1024 && fixed + 2 * in <= 160 * 1024) return 2;
    return 0;
}
template <class Cfg>
constexpr int wst_lds_bytes() { return 2 * (int)Cfg::CHUNK_W_BYTES + kPersistMaxCout * 4 + (wst_form<Cfg>() == 3 ? 3 : 2) * WstCfg<Cfg>::IN_BYTES; }
template <class Cfg>
constexpr bool wst_eligible() { return wst_form<Cfg>() != 0; }
constexpr int waitcnt_vm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }   // s_waitcnt vmcnt(n), lgkmcnt / expcnt untouched (gfx9 encoding)

template <class Cfg8, bool FAST>
__global__ __launch_bounds__(kWsmThreads) void conv_mfma_wst_kernel(const ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the stub: with the body visible to it, it silently dropped the stub of every instantiation (undefined symbol at load)
    using Cfg = WstCfg<Cfg8>;
    using D = ConvDev<Cfg>;
    using D8 = ConvDev<Cfg8>;
    constexpr int NT = Cfg::NT, STEPS = Cfg::STEPS, NV = 4 * NT, WALL = (int)Cfg::CHUNK_W_BYTES, UPT = Cfg::UPT, TWH = Cfg::TWH, ES = 2;
    constexpr int N_DMA = Cfg::N_DMA, NDW = (N_DMA + 3) / 4;         // tile pieces: wave-instructions per tile / per tile wave
    constexpr int NBUF = wst_form<Cfg8>() == 3 ? 3 : 2;              // tile buffers (form 3: by LDS-DMA two stages ahead; form 2: through registers one stage ahead)
    constexpr bool FOLD_SKIP = Cfg::KS == 2 && UPT == 4 && STEPS == 4;   // a tap = one MFMA step (see kernel 4)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;                                                // W[2]
    float* s_bias = reinterpret_cast<float*>(smem + 2 * WALL);
    char* s_in0 = smem + 2 * WALL + kPersistMaxCout * 4;             // in[NBUF]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave12 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave12 >= 8;
    const int q = lane >> 4, n = lane & 15;

    const int tiles_y = (a.H + kWsmTH - 1) / kWsmTH;
    const int sp_total = a.tiles_x * tiles_y;
    const int n_tiles = sp_total * a.batch;
    const int n_chunks = a.n_chunks, n_ct = a.n_ct;
    const bool one_chunk = n_chunks == 1;                            // unit = tile (fetched once for all its cout tiles); else unit = (tile, cout tile)
    const int n_units = one_chunk ? n_tiles : n_tiles * n_ct;
    const int cts_per_unit = one_chunk ? n_ct : 1;
    const int slots = gridDim.x >> 3;
    const int pos = (blockIdx.x & 7) * slots + (blockIdx.x >> 3);
    const int stride = (int)gridDim.x;
    const int my_units = pos < n_units ? (n_units - pos + stride - 1) / stride : 0;
    const int my_stages = my_units * cts_per_unit * n_chunks;

    for (int i = tid; i < a.cout_packed; i += kWsmThreads) s_bias[i] = a.bias ? a.bias[i] : 0.f;

    auto decode = [&](int unit, int& b, int& ty, int& tx, int& ct0) {
        int tile = unit;
        ct0 = 0;
        if (!one_chunk) { tile = magic_div(unit, a.div_n_ct); ct0 = unit - tile * n_ct; }
        b = magic_div(tile, a.td_wsm.sp_total);
        band_decode(tile - b * sp_total, a.tiles_x, tiles_y, a.td_wsm, ty, tx);
    };

    if (loader && NBUF == 2) {
        // ---------------------------------------------------------------- loader waves, form 2: during stage g, W(g+1) by LDS-DMA and tile(g+1) through registers,
        // both issued at the start of the stage and waited for ONCE (kernel 4 waits for the tile in half a and for a weight half in half b: two latencies in series)
        if constexpr (NBUF == 2) {
            const int rtid = tid - kWsmCompute, w4 = wave12 - 8;
            uint4 r0[D::NI], r1[1];
            float gv[1];
            typename D::TileSrc ts;
            typename D::TileOffs to;
            D::tile_offsets(a, rtid, to);
            int k_unit = 0, cti = 0, chunk = 0, gi = 0, ct = 0;
            int b = 0, ty = 0, tx = 0, ct0 = 0;
            auto fetch = [&]() {                                     // everything stage gi reads
                if (gi < my_stages) {
                    if (chunk == 0) {
                        if (cti == 0) decode(pos + k_unit * stride, b, ty, tx, ct0);
                        ct = ct0 + cti;
                        ts = D::tile_src(a, b, ty * kWsmTH, tx * kTW);
                    }
                    const int goff = (ct * n_chunks + chunk) * WALL;
                    for (int kb = w4; kb < WALL / 1024; kb += 4)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(static_cast<const char*>(a.wpacked) + goff + kb * 1024 + lane * 16),
                                                         (__attribute__((address_space(3))) void*)(s_w + (gi & 1) * WALL + kb * 1024), 16, 0, 0);
                    if (!one_chunk || cti == 0) {
                        D::template load_tile<false>(a, ts, to, b, chunk, rtid, r0, r1, gv);
                        D::template commit_tile<false>(a, ts, to, chunk, rtid, r0, r1, gv, s_in0 + (one_chunk ? (k_unit & 1) : (gi & 1)) * Cfg::IN_BYTES);
                    }
                }
                ++gi;
                if (++chunk == n_chunks) { chunk = 0; if (++cti == cts_per_unit) { cti = 0; ++k_unit; } }
            };
            fetch();                                                 // stage 0
            __syncthreads();                                         // barrier 0
            for (int g = 0; g < my_stages; ++g) {
                fetch();                                             // stage g + 1: its buffers were last read in stage g - 1
                __syncthreads();
            }
        }
    } else if (loader) {
        // ---------------------------------------------------------------- tile waves
        const int w4 = wave12 - 8;
        // piece r of this wave = wave-instruction w4 + 4 r of the tile: 16-byte piece P = 64 inst + lane = (halo pixel P / UPT, unit P % UPT).  Interior
        // tiles (95 % at 4K): its offset is this per-lane launch constant + the tile's scalar offset in the instruction's soffset
        int lc[NDW];
#pragma unroll
        for (int r = 0; r < NDW; ++r) {
            const int P = (w4 + 4 * r) * 64 + lane, pix = P / UPT, v = P - pix * UPT, py = pix / TWH, px = pix - py * TWH;
            const int o = D::folded(a) ? ((2 * py * a.src_W + 2 * px) * a.cfold + v * Cfg::UNIT) * ES : ((py * a.W + px) * a.cin + v * Cfg::UNIT) * ES;
            lc[r] = P < Cfg::N_PIECES ? o : kOOB;
        }
        typename D::TileSrc ts;
        int k_unit = 0, cti = 0, chunk = 0, gi = 0, gb3 = 0, kb3 = 0;   // the stage whose tile is fetched next (+ stage / unit index mod 3)
        int b = 0, ty = 0, tx = 0, ct0 = 0;
        const int n_mine = (N_DMA - w4 + 3) / 4;                     // pieces of a tile this wave issues: NDW or NDW - 1
        // Fetch the tile of the next stage in line; returns how many of its pieces may stay in flight across the coming barrier.  Only an INTERIOR tile's do:
        // the count wait needs the counter to retire in issue order, and an instruction whose lanes are all out of range (border tiles have them; a round-5
        // form that padded tile-less stages with zero-record pieces to keep the count constant) completes without a memory access -- the rare stale tile that
        // form produced on a cold GPU (1 launch in ~1 000, tools/thin_ab.py) is why border and tile-less stages now wait for everything.
        auto issue = [&]() -> int {
            const bool tile = gi < my_stages && (!one_chunk || cti == 0);
            char* dst = s_in0 + (one_chunk ? kb3 : gb3) * Cfg::IN_BYTES;
            int in_flight = 0;
            if (tile && chunk == 0) {
                decode(pos + k_unit * stride, b, ty, tx, ct0);
                ts = D::tile_src(a, b, ty * kWsmTH, tx * kTW);
            }
            if (tile && ts.interior) {
                const int soff = ts.soff + D::chunk_soff(a, chunk);
#pragma unroll
                for (int r = 0; r < NDW; ++r) {
                    const int inst = w4 + 4 * r;                     // wave-uniform
                    if (inst < N_DMA)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(ts.r0, (__attribute__((address_space(3))) void*)(dst + inst * 1024), 16, lc[r], soff, 0, 0);
                }
                in_flight = n_mine;
            } else if (tile) {                                       // border tiles: bounds-checked geometry per piece
                int ln = lane;
                asm volatile("" : "+v"(ln));                         // keep it here: hoisted out of the stage loop it would be spilled
#pragma unroll
                for (int r = 0; r < NDW; ++r) {
                    const int inst = w4 + 4 * r, P = inst * 64 + ln, pix = P / UPT, v = P - pix * UPT;
                    if (inst < N_DMA) {
                        bool center;
                        const int off = P < Cfg::N_PIECES ? D::vec_off(a, ts, chunk, pix, v, true, center) : kOOB;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(ts.r0, (__attribute__((address_space(3))) void*)(dst + inst * 1024), 16, off, 0, 0, 0);
                    }
                }
            }
            ++gi; gb3 = gb3 == 2 ? 0 : gb3 + 1;
            if (++chunk == n_chunks) { chunk = 0; if (++cti == cts_per_unit) { cti = 0; ++k_unit; kb3 = kb3 == 2 ? 0 : kb3 + 1; } }
            return in_flight;
        };
        // the packed weights of the stage after the current one, global -> W[stage & 1] (1 KiB per wave-instruction; only the steps -- taps -- the chunk's
        // fold phase multiplies by, see run_chunks).  Issued BEFORE the stage's tile pieces: older in the in-order counter, so the count wait lands them
        int wi = 0, wk = 0, wcti = 0, wchunk = 0, wct0 = 0;
        auto issue_w = [&]() {
            if (wi < my_stages) {
                if (wchunk == 0 && wcti == 0) { int b_, ty_, tx_; decode(pos + wk * stride, b_, ty_, tx_, wct0); }
                const int goff = ((wct0 + wcti) * n_chunks + wchunk) * WALL;
                int ph = 3;
                if constexpr (FOLD_SKIP) ph = D::folded(a) ? D::fold_phase(a, wchunk * Cfg::CK) : 3;
                for (int kb = w4; kb < WALL / 1024; kb += 4) {
                    if constexpr (FOLD_SKIP) {
                        const int st = kb / NT;
                        if (!(st == 3 || ph == 3 || (ph == 1 && st == 2) || (ph == 2 && st == 1))) continue;
                    }
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(static_cast<const char*>(a.wpacked) + goff + kb * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*)(s_w + (wi & 1) * WALL + kb * 1024), 16, 0, 0);
                }
            }
            ++wi;
            if (++wchunk == n_chunks) { wchunk = 0; if (++wcti == cts_per_unit) { wcti = 0; ++wk; } }
        };
        issue_w();                                                   // W(0)
        issue();                                                     // tile(0)
        issue();                                                     // tile(1)
        __syncthreads();                                             // barrier 0 (vmcnt(0): all three landed): bias, W(0), tile(0) visible
        for (int g = 0; g < my_stages; ++g) {
            issue_w();                                               // W(g+1): its buffer was last read in stage g-1
            const int nfl = issue();                                 // tile(g+2)
            // everything older than the pieces just issued -- W(g+1), tile(g+1) -- is in LDS (the count is an immediate: one branch per value)
            if (nfl == NDW) __builtin_amdgcn_s_waitcnt(waitcnt_vm(NDW));
            else if (NDW > 1 && nfl == NDW - 1) __builtin_amdgcn_s_waitcnt(waitcnt_vm(NDW > 1 ? NDW - 1 : 0));
            else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
    } else {
        // ---------------------------------------------------------------- compute waves (two per SIMD)
        const int wave = wave12;
        typename D::LaneOff lo;
        D::lane_offsets(q, lo);
        const int lane_x = ((2 * wave) * D::TWH + n) * D::SPIX;
        const int lane_w = lane * 16;
        __syncthreads();                                             // barrier 0
        int g = 0, gb3 = 0, kb3 = 0;
        for (int k = 0; k < my_units; ++k) {
            int b, ty, tx, ct0;
            decode(pos + k * stride, b, ty, tx, ct0);
            const int ty8 = 2 * ty + (wave >> 2);                    // each half of the block stores as an ordinary 8 x 32 tile
            for (int ct = ct0; ct < ct0 + cts_per_unit; ++ct) {
                f32x4 acc[4][NT];                                    // initial C operand = bias
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float4 t4 = *reinterpret_cast<const float4*>(s_bias + ct * Cfg::COUT_TILE + q * NV + nt * 4);
#pragma unroll
                    for (int pt = 0; pt < 4; ++pt) acc[pt][nt] = f32x4{t4.x, t4.y, t4.z, t4.w};
                }
                // chunks [c0, c1) of the item, all of fold phase PH (3 = every tap carries weights: the un-folded layers; see kernel 4)
                auto run_chunks = [&](auto PH, int c0, int c1) {
                    constexpr int ph = decltype(PH)::value;
                    for (int c = c0; c < c1; ++c, ++g) {
                        const char* s_in = s_in0 + (one_chunk ? kb3 : gb3) * Cfg::IN_BYTES;
                        const char* s_wg = s_w + (g & 1) * WALL;
                        if constexpr (ph == 3) D::template mma_steps<0, STEPS, 0, true>(s_in, s_wg, lane_x, lane_w, q, lo, acc);
                        else if constexpr (ph == 1) D::template mma_steps<2, 2, 0, true>(s_in, s_wg, lane_x, lane_w, q, lo, acc);
                        else {
                            if constexpr (ph == 2) D::template mma_steps<1, 1, 0, false>(s_in, s_wg, lane_x, lane_w, q, lo, acc);
                            D::template mma_steps<3, 1, 0, false>(s_in, s_wg, lane_x, lane_w, q, lo, acc);
                        }
                        gb3 = gb3 == NBUF - 1 ? 0 : gb3 + 1;
                        if (c + 1 < n_chunks) __syncthreads();
                    }
                };
                if constexpr (FOLD_SKIP) {
                    const int cpp = D::folded(a) ? n_chunks >> 2 : 0;
                    run_chunks(std::integral_constant<int, 0>{}, 0, cpp);
                    run_chunks(std::integral_constant<int, 1>{}, cpp, 2 * cpp);
                    run_chunks(std::integral_constant<int, 2>{}, 2 * cpp, 3 * cpp);
                    run_chunks(std::integral_constant<int, 3>{}, 3 * cpp, n_chunks);
                } else {
                    run_chunks(std::integral_constant<int, 3>{}, 0, n_chunks);
                }
                if (ty8 < a.tiles_y)
                    D8::template epilogue<FAST>(a, b, ty8 * kTH, tx * kTW, ty8 * a.tiles_x + tx, ct, tid & 255, acc);
                __syncthreads();
            }
            kb3 = kb3 == NBUF - 1 ? 0 : kb3 + 1;
        }
    }
#endif
}

// ==================================================================================================
// Kernel 5: kernel 4 for the single-chunk pixel-shuffle layers (the tail 48 -> 192 + PixelShuffle(2): 6.4 GB of output at cfg3) with the
// OUTPUT staged through LDS and stored by the loader waves.  OFF by default (rc_debug_set("pss", 1)): it ties with kernel 4.
// Knock-outs of kernel 4 on that layer (tools/tail_probe.py, tools/pss_flags.py; 8 x 1088 x 1920): 3.27 ms; 2.04 ms without its stores
// (MFMA floor 1.6 ms at the clock these kernels run at); 2.64 ms with every store redirected into one L2-resident 4 MB window -- so
// about half of the 1.2 ms the stores cost is issuing them (all 8 compute waves reach the epilogue together, each of their 8 store
// instructions holds the wave's issue port until the write path takes it) and half is the HBM write stream itself (96-byte sub-pixel
// pieces = half of every 128-byte line per cout tile; tools/ubench/store_issue.hip writes that pattern at 2.3 TB/s when the two halves of
// a line are more than ~1 MB per XCD apart, 5 TB/s when they leave together).  Tried on kernel 4 and dropped: the previous tile's stores
// issued one by one inside the next stage's MFMA stream (3.6 ms), and holding the even sub-pixel in registers to store complete lines with
// the odd one (3.31 ms).  Here the compute waves never touch the write path: they pack a cout tile (= one sub-pixel of every pixel) to
// bf16, drop it into an LDS out tile (3 ds_write_b64 per pixel tile) and go on; the 4 loader waves drain it in address order while the
// next stage's MFMAs run.  LDS: the out tile (48 KB) takes the place of the second input buffer -- a tile's 4 stages read one input tile,
// and the next one is committed from registers in the short phase in which the computers write the out tile:
//
//            compute waves 0-7                                   loader waves 8-11
//   phase 0  packed result of stage g-1 -> out tile              Wb(g) registers -> LDS; first stage of a tile: input tile registers -> LDS
//   barrier
//   half a   MFMA steps [0, SA) of stage g    (Wa)               issue loads Wa(g+1) (+ the next input tile during a tile's last stage);
//                                                                drain the out tile, first half  (LDS -> HBM, pixel-shuffled address)
//   barrier
//   half b   MFMA steps [SA, STEPS)           (Wb); pack         Wa(g+1) registers -> LDS; issue loads Wb(g+1); drain, second half
//   barrier
//
// Result: the exposed store time halves (3.16-3.29 ms with stores, 2.60 without) but the third barrier, the single input buffer and the
// out tile's LDS traffic cost what that gains (2.60 vs 2.04 ms without stores): bit-identical to kernel 4, not faster.
// ==================================================================================================
template <class Cfg>
constexpr int pss_lds_bytes() { return (int)Cfg::CHUNK_W_BYTES + kPersistMaxCout * 4 + WsmCfg<Cfg>::IN_BYTES + kWsmTH * kTW * Cfg::COUT_TILE * 2; }

template <class Cfg8>
__global__ __launch_bounds__(kWsmThreads) void conv_mfma_pss_kernel(const ConvArgs a) {
    using Cfg = WsmCfg<Cfg8>;
    using D = ConvDev<Cfg>;
    using T = typename Cfg::elem;
    static_assert(sizeof(T) == 2, "bf16 form");
    constexpr int NT = Cfg::NT, STEPS = Cfg::STEPS, NV = 4 * NT, NH = NV / 2;
    constexpr int SA = (STEPS + 1) / 2;
    constexpr int WA = SA * NT * 1024, WB = (STEPS - SA) * NT * 1024, WALL = (int)Cfg::CHUNK_W_BYTES;
    constexpr int NWA = (WA / 16 + kThreads - 1) / kThreads, NWB = (WB / 16 + kThreads - 1) / kThreads;
    constexpr int PXB = Cfg::COUT_TILE * 2, ROWB = kTW * PXB, OUTB = kWsmTH * ROWB;   // out tile: [16 rows][32 pixels][COUT_TILE bf16]
    constexpr int UPP = PXB / 16, UPR = kTW * UPP;                   // 16-byte units per pixel / per row
    constexpr int NDR = OUTB / 16 / kThreads;                        // units per loader thread
    static_assert(STEPS >= 2 && WA + WB == WALL && OUTB % (16 * kThreads) == 0 && NDR % 2 == 0 && PXB % 16 == 0, "pss shapes");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;
    float* s_bias = reinterpret_cast<float*>(smem + WALL);
    char* s_in = smem + WALL + kPersistMaxCout * 4;
    char* s_out = s_in + Cfg::IN_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave12 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave12 >= 8;
    const int q = lane >> 4, n = lane & 15;

    const int tiles_y = (a.H + kWsmTH - 1) / kWsmTH;
    const int sp_total = a.tiles_x * tiles_y;
    const int n_tiles = sp_total * a.batch;
    constexpr int n_ct = 4;                                          // host: cout = 4 cout tiles = the 4 sub-pixels
    const int slots = gridDim.x >> 3;                                // XCD x takes a run of consecutive tiles
    const int pos = (blockIdx.x & 7) * slots + (blockIdx.x >> 3);
    const int stride = (int)gridDim.x;
    const int my_units = pos < n_tiles ? (n_tiles - pos + stride - 1) / stride : 0;
    const int my_stages = my_units * n_ct;
    const int cps = a.cout >> 2;                                     // == COUT_TILE

    for (int i = tid; i < a.cout_packed; i += kWsmThreads) s_bias[i] = a.bias ? a.bias[i] : 0.f;

    auto decode = [&](int unit, int& b, int& ty, int& tx) {
        b = magic_div(unit, a.td_wsm.sp_total);
        band_decode(unit - b * sp_total, a.tiles_x, tiles_y, a.td_wsm, ty, tx);
    };

    if (loader) {
        const int rtid = tid - kWsmCompute;
        uint4 r0[D::NI], r1[1], wra[NWA], wrb[NWB];
        float gv[1];
        typename D::TileSrc ts;
        typename D::TileOffs to;
        D::tile_offsets(a, rtid, to);
        // this thread's 16-byte pieces of the two weight halves (recomputed where used: the loader's registers go to the tile and the weights)
        // (through `lt`, the thread index hidden from loop-invariant code motion: hoisted, these offsets and the drain's would be spilled)
        int lt = rtid;
        auto woa = [&](int k) { return (k * kThreads + lt) * 16 < WA ? (k * kThreads + lt) * 16 : kOOB; };
        auto wob = [&](int k) { return (k * kThreads + lt) * 16 < WB ? WA + (k * kThreads + lt) * 16 : kOOB; };
        const __amdgpu_buffer_rsrc_t r_w = make_rsrc(a.wpacked, (unsigned)((size_t)n_ct * WALL));
        const size_t img_out = (size_t)a.H * a.W * a.cout;

        int b = 0, ty = 0, tx = 0;                                   // tile whose loads were issued last
        int d_b = 0, d_y0 = 0, d_x0 = 0, d_ct = 0;                  // stage whose result sits in the out tile
        int wsoff = 0;
        auto issue_tile = [&](int k) {
            decode(pos + k * stride, b, ty, tx);
            ts = D::tile_src(a, b, ty * kWsmTH, tx * kTW);
            D::template load_tile<false>(a, ts, to, b, 0, rtid, r0, r1, gv);
        };
        auto issue_wa = [&](int ct) {
            wsoff = ct * WALL;
#pragma unroll
            for (int k = 0; k < NWA; ++k) wra[k] = buf_load16(r_w, woa(k), wsoff);
        };
        auto issue_wb = [&]() {
#pragma unroll
            for (int k = 0; k < NWB; ++k) wrb[k] = buf_load16(r_w, wob(k), wsoff);
        };
        auto commit_tile = [&]() { D::template commit_tile<false>(a, ts, to, 0, rtid, r0, r1, gv, s_in); };
        auto commit_wa = [&]() {
#pragma unroll
            for (int k = 0; k < NWA; ++k)
                if (woa(k) != kOOB) *reinterpret_cast<uint4*>(s_w + woa(k)) = wra[k];
        };
        auto commit_wb = [&]() {
#pragma unroll
            for (int k = 0; k < NWB; ++k)
                if (wob(k) != kOOB) *reinterpret_cast<uint4*>(s_w + wob(k)) = wrb[k];
        };
        auto drain = [&](int j0, bool live) {                        // NDR / 2 units: LDS -> the pixel-shuffled NHWC output (always issued: countable)
            const __amdgpu_buffer_rsrc_t r_out = make_rsrc(static_cast<T*>(a.out) + (size_t)d_b * img_out, (unsigned)(img_out * 2));
            const int base = (((2 * d_y0 + (d_ct >> 1)) * (2 * a.W) + 2 * d_x0 + (d_ct & 1)) * cps) * 2;
            const bool full = live && d_y0 + kWsmTH <= a.H && d_x0 + kTW <= a.W && !(a.dbg_flags & 1);     // uniform
            asm volatile("" : "+v"(lt));
            // unit u = rtid + 256 j of the out tile is (row, pixel, 16-byte part) -> pixel (2 row, 2 pixel) of the stage's sub-pixel plane
#pragma unroll
            for (int j = j0; j < j0 + NDR / 2; ++j) {
                const int u = lt + kThreads * j, row = u / UPR, rem = u - row * UPR, px = rem / UPP, part = rem - px * UPP;
                const uint4 v = *reinterpret_cast<const uint4*>(s_out + u * 16);
                int off = base + ((2 * row * (2 * a.W) + 2 * px) * cps) * 2 + part * 16;
                if (!full && (!live || d_y0 + row >= a.H || d_x0 + px >= a.W || (a.dbg_flags & 1))) off = kOOB;
                buf_store16(r_out, off, 0, v);
                if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // two units in flight: the registers belong to the tile and the weights
            }
        };

        if (my_stages > 0) {
            issue_tile(0); issue_wa(0);
            commit_tile(); commit_wa();
            issue_wb();
        }
        __syncthreads();                                             // barrier 0: bias, Wa(0), tile(0) visible
        for (int g = 0; g < my_stages; ++g) {
            const int k = g >> 2, ct = g & 3;
            asm volatile("" : "+v"(lt));
            // ---- phase 0: LDS writes only (the computers wait for it)
            commit_wb();                                             // Wb(g)
            if (ct == 0 && g > 0) commit_tile();                     // this tile's input (loaded during the previous tile's last stage)
            __syncthreads();
            // ---- half a: loads first, then the stores (a commit waits for everything older than its loads, see kernel 4)
            if (g + 1 < my_stages) issue_wa((g + 1) & 3);            // Wa(g+1)
            if (ct == 3 && k + 1 < my_units) issue_tile(k + 1);
            drain(0, g > 0);
            __syncthreads();
            // ---- half b
            if (g + 1 < my_stages) { commit_wa(); issue_wb(); }
            drain(NDR / 2, g > 0);
            // the stage the computers are packing now is drained during the next one
            {
                int cb, cty, ctx;
                decode(pos + k * stride, cb, cty, ctx);
                d_b = cb; d_y0 = cty * kWsmTH; d_x0 = ctx * kTW; d_ct = ct;
            }
            __syncthreads();
        }
        __syncthreads();                                             // the last stage's result is in the out tile
        drain(0, my_stages > 0); drain(NDR / 2, my_stages > 0);
    } else {
        const int wave = wave12;                                     // rows 2*wave, 2*wave+1 of the 16-row tile
        typename D::LaneOff lo;
        D::lane_offsets(q, lo);
        const int lane_x = ((2 * wave) * D::TWH + n) * D::SPIX;
        const int lane_w = lane * 16;
        const int o_lane = (2 * wave) * ROWB + n * PXB + q * NV * 2;  // this lane's slot of pixel tile 0 in the out tile
        unsigned pend[4][NH];
        auto put = [&]() {
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                char* dst = s_out + o_lane + (pt >> 1) * ROWB + (pt & 1) * 16 * PXB;
                if constexpr (NV * 2 % 16 == 0) {
#pragma unroll
                    for (int i = 0; i < NH; i += 4) *reinterpret_cast<uint4*>(dst + 4 * i) = make_uint4(pend[pt][i], pend[pt][i + 1], pend[pt][i + 2], pend[pt][i + 3]);
                } else {
#pragma unroll
                    for (int i = 0; i < NH; i += 2) *reinterpret_cast<uint2*>(dst + 4 * i) = make_uint2(pend[pt][i], pend[pt][i + 1]);
                }
            }
        };
        const float inf = __builtin_inff();
        __syncthreads();                                             // barrier 0
        for (int g = 0; g < my_stages; ++g) {
            const int ct = g & 3;
            if (g > 0) put();                                        // phase 0
            __syncthreads();
            f32x4 acc[4][NT];                                        // initial C operand = bias
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 t4 = *reinterpret_cast<const float4*>(s_bias + ct * Cfg::COUT_TILE + q * NV + nt * 4);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[pt][nt] = f32x4{t4.x, t4.y, t4.z, t4.w};
            }
            D::template mma_steps<0, SA, 0, true>(s_in, s_w, lane_x, lane_w, q, lo, acc);
            __syncthreads();
            D::template mma_steps<SA, STEPS - SA, 0, true>(s_in, s_w, lane_x, lane_w, q, lo, acc);
            const int key = a.ep_key;                                // 0 none, EP_RELU, EP_LEAKY (host: nothing else takes this kernel)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int i = 0; i < NH; ++i) {
                    float v0 = acc[pt][(2 * i) >> 2][(2 * i) & 3], v1 = acc[pt][(2 * i + 1) >> 2][(2 * i + 1) & 3];
                    if (key == D::EP_RELU) { v0 = __builtin_amdgcn_fmed3f(v0, 0.f, inf); v1 = __builtin_amdgcn_fmed3f(v1, 0.f, inf); }
                    else if (key == D::EP_LEAKY) { v0 = __builtin_amdgcn_fmed3f(v0, v0 * a.act_slope, inf); v1 = __builtin_amdgcn_fmed3f(v1, v1 * a.act_slope, inf); }
                    pend[pt][i] = pack_bf16x2(v0, v1);
                }
            __syncthreads();
        }
        if (my_stages > 0) put();
        __syncthreads();
    }
}

// ==================================================================================================
// Kernel 6: wave-AUTONOMOUS persistent form for single-chunk, single-cout-tile bf16 layers (the 48 -> 48 convolutions: half of the step).
// Kernel 2 keeps two 4-wave blocks per CU, each with its own copy of the packed weights (43 KB) and ONE halo tile; the waves of a block
// meet at two workgroup barriers per tile, so a CU has at most two tiles' loads in flight and every wave waits for its block's slowest.
// gfx950 has no sub-group barrier, so three tiles in flight cannot be three teams of one block.  Here a CU runs ONE block of 8 waves
// that share the weights and nothing else: wave w owns a 2-row x 32-pixel STRIP (the wave tile of every other form: 4 pixel tiles x NT
// cout tiles) with a PRIVATE 4 x 34 halo strip in LDS (13 KB; 8 x 13 + 43 KB of weights), loads it itself (13 x 16 B per lane,
// prefetched one strip ahead in registers), and never meets another wave after the block's first barrier: no barrier in the loop,
// 8 strips = 104 KB of loads in flight per CU instead of 65 KB, and the waves drift apart instead of marching in two groups.
// The price is halo traffic through L2 -> LDS: 136 halo pixels per 64 outputs (2.1x) instead of 340 per 256 (1.33x) -- and, because free-running
// waves drift apart by more than the L2 keeps a row (each XCD's 4 MB also holds the layer's write stream), a good part of those re-reads comes from
// BEYOND L2: FETCH_SIZE x 2 = 1.54x the input map in mode 0 and 1.39x in mode 1, where kernel 2 reads 1.00x (profiles/r05_pmc_bench.json, r06_kernel6_fetch.md).
// Round 6 measured what that costs: with the waves held together (one s_barrier per region, or a ticket that bounds the drift to one region) the
// ratio falls to 1.02-1.09x and the launch takes THE SAME time in mode 0 (772 vs 773 us) and 3 % MORE in mode 1 (817 vs 795 us): the 0.5-0.8 GB of
// extra HBM reads per launch are not what paces these layers.  Mode 0 keeps the barrier (free, fewer bytes), modes 1 / 2 run free.
// Same unit map, same MFMA chain, same epilogues (a strip IS wave j & 3 of an ordinary 8 x 32 tile) -> bit-identical to kernels 1-4.
// Every load of the loop is issued unconditionally (out-of-range strips load from offset kOOB = no memory access), so hipcc can count
// them; per-image vectors an epilogue needs (the CALayer gate of key EP_GATE | EP_RES) are copied to LDS once per block: a global load
// issued after the MFMA loop is younger than the next strip's prefetch and its wait (vmcnt counts in order) would drain it.
// MEASURED (profiles/r05_power_wall.md): 0.756-0.78 ms on the level-0 plain layer against kernel 2's 0.76-0.785 -- and the same again with
// whole-kilobyte stores staged through the strip buffer, and with half the wave tile at 3 or 4 waves per SIMD ("kernel 7", removed).
// All of them run the socket at its 1400 W cap with the shader clock throttled to 1.6-1.9 GHz: time = joules per launch / 1.1 kW, and
// none of this changes the joules.  Kept for the forms where it is 1-5 % faster (plain, ReLU, + channel sums: `persist_auto` 1);
// the residual forms stay on kernel 2.
// ==================================================================================================
constexpr int kAutoWaves = 8, kAutoThreads = 64 * kAutoWaves, kAutoBias = 64, kAutoGate = 1024;   // LDS floats: bias; the (batch, cout) CALayer gates of key EP_GATE | EP_RES
template <class Cfg>
using StripCfg = ConvCfg<typename Cfg::elem, Cfg::CK, Cfg::NT, Cfg::KS, 2>;
template <class Cfg>
constexpr int auto_lds_bytes() { return (int)Cfg::CHUNK_W_BYTES + (kAutoBias + kAutoGate) * 4 + kAutoWaves * StripCfg<Cfg>::IN_BYTES; }
template <class Cfg>
constexpr bool auto_eligible() {
    return sizeof(typename Cfg::elem) == 2 && Cfg::KS == 3 && Cfg::NT <= 3 && Cfg::CK >= 32 && Cfg::COUT_TILE <= kAutoBias && auto_lds_bytes<Cfg>() <= 160 * 1024;
}

// MODE: 0 = any fast epilogue without carried state, 1 = CALayer channel sums carried across the wave's strips (EP_SUMS, EP_RELU | EP_SUMS),
// 2 = residual prefetched before the MFMA loop (EP_RES, EP_GATE | EP_RES).  Three instantiations: the carried sums (12 registers) and the
// prefetched residual (24) are never live together, and one kernel holding both spilled 37 registers.
template <class Cfg8, int MODE>
__global__ __launch_bounds__(kAutoThreads) void conv_mfma_auto_kernel(const ConvArgs a) {
    using Cfg = StripCfg<Cfg8>;
    using D = ConvDev<Cfg>;                                          // MFMA pieces on the 2-row strip
    using D8 = ConvDev<Cfg8>;                                        // epilogue: the strip is wave (j & 3) of an ordinary 8 x 32 tile
    using T = typename Cfg::elem;
    constexpr int NT = Cfg::NT, STEPS = Cfg::STEPS, NV = 4 * NT, UPT = Cfg::UPT, UNIT = Cfg::UNIT, TWH = Cfg::TWH, SPIX = Cfg::SPIX, ES = 2;
    constexpr int NPIX = Cfg::THH * TWH, NU = NPIX * UPT, NI = (NU + 63) / 64;      // 16-byte units of a halo strip; per lane
    constexpr int RU = TWH * UPT;                                    // units per strip row (204: more than a wave, so the lanes of one load span <= 2 rows)
    constexpr bool LINEAR = SPIX == UPT * 16;                        // no padding between pixels: unit u sits at byte 16 u of the strip
    static_assert(Cfg::HALO == 1 && Cfg::THH == 4 && RU >= 64, "3x3, 2-row strips");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;                                                // weights first: ds_read immediates < 64 KiB
    float* s_bias = reinterpret_cast<float*>(smem + Cfg::CHUNK_W_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* s_gate = s_bias + kAutoBias;
    char* s_my = smem + Cfg::CHUNK_W_BYTES + (kAutoBias + kAutoGate) * 4 + wave * Cfg::IN_BYTES;
    const int q = lane >> 4, n = lane & 15;
    const int ftid = ((wave & 3) << 6) | lane;                       // this lane's thread index inside the 8 x 32 tile its strip belongs to

    for (int kb = wave; kb < STEPS * NT; kb += kAutoWaves)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(static_cast<const char*>(a.wpacked) + kb * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(s_w + kb * 1024), 16, 0, 0);
    for (int i = tid; i < kAutoBias; i += kAutoThreads) s_bias[i] = (a.bias && i < a.cout_packed) ? a.bias[i] : 0.f;
    [[maybe_unused]] const bool gated_out = a.ep_key == (D8::EP_GATE | D8::EP_RES);                                      // uniform
    if constexpr (MODE == 2) {
        if (gated_out)
            for (int i = tid; i < a.batch * a.cout; i += kAutoThreads) s_gate[i] = a.out_scale[i];                      // host: batch * cout <= kAutoGate
    }

    const int tiles_y16 = (a.H + kWsmTH - 1) / kWsmTH;               // the block walks 16 x 32 regions: wave w = rows 2w, 2w + 1
    const int sp_total = a.tiles_x * tiles_y16;
    const int n_units = sp_total * a.batch;
    const int slots = gridDim.x >> 3;                                // XCD x takes a run of consecutive regions
    const int pos = (blockIdx.x & 7) * slots + (blockIdx.x >> 3);
    const int stride = (int)gridDim.x;
    const size_t img = (size_t)a.H * a.W * a.cin;
    const unsigned img_bytes = (unsigned)(img * ES);

    // this wave's next strip at or after region `from`: regions whose 8-row tile ty8 = 2 ty + (wave >> 2) lies below the image are skipped
    // (they have no channel-sum slot either); returns -1 past the end
    auto next_strip = [&](int from, int& b, int& ty8, int& tx) -> int {
        for (int u = from; u < n_units; u += stride) {
            b = magic_div(u, a.td_wsm.sp_total);
            int ty;
            band_decode(u - b * sp_total, a.tiles_x, tiles_y16, a.td_wsm, ty, tx);
            ty8 = 2 * ty + (wave >> 2);
            if (ty8 < a.tiles_y) return u;
        }
        return -1;
    };

    // Load k of a strip covers units u = lane + 64 k: pixel u / UPT of the 4 x 34 halo strip, 16-byte piece u % UPT.  Interior strips
    // (cin == CK): a strip row is RU contiguous units in memory, so the byte offset relative to halo pixel (0, 0) is
    // 16 u + row(u) * (image row pitch - strip row bytes); the lanes of one load lie in at most two rows, so it is
    // 16 lane + (lane >= t_k ? delta : 0) with everything else in the instruction's scalar offset -- nothing per strip is kept in VGPRs.
    const int rdelta = (a.W * a.cin - RU * UNIT) * ES;               // uniform
    uint4 r[NI];
    // all NI loads of one strip, always issued (valid == false: every offset out of range, no memory access)
    auto issue = [&](bool valid, int b, int ty8, int tx) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(static_cast<const T*>(a.in0) + (size_t)b * img, img_bytes);
        const int gy0 = ty8 * kTH + 2 * (wave & 3) - 1, gx0 = tx * kTW - 1;
        if (a.dbg_flags & 4) valid = false;                          // knock-out: no tile loads (every offset out of range)
        const bool interior = valid && gy0 >= 0 && gx0 >= 0 && gy0 + Cfg::THH <= a.H && gx0 + TWH <= a.W && a.cin_chunk_ok;   // uniform
        if (interior) {
            const int soff = (gy0 * a.W + gx0) * a.cin * ES;
            const int l16 = lane * 16;
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int row0 = (64 * k) / RU, t = (row0 + 1) * RU - 64 * k;        // first row of this load; lanes >= t are in the next row
                int voff = l16;
                if (t < 64) voff = lane >= t ? l16 + rdelta : l16;
                if (64 * k + 63 >= NU) voff = lane + 64 * k < NU ? voff : kOOB;
                r[k] = buf_load16(rs, voff, soff + 1024 * k + row0 * rdelta);
            }
        } else {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int u = lane + 64 * k, pix = u / UPT, v = u - pix * UPT, py = pix / TWH, px = pix - py * TWH;
                const int gy = gy0 + py, gx = gx0 + px;
                const bool ok = valid && u < NU && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W && v * UNIT < a.cin;
                r[k] = buf_load16(rs, ok ? ((gy * a.W + gx) * a.cin + v * UNIT) * ES : kOOB, 0);
            }
        }
    };

    typename D::LaneOff lo;
    D::lane_offsets(q, lo);
    const int lane_x = n * SPIX, lane_w = lane * 16;

    int cb = 0, cty8 = 0, ctx = 0;
    int cu = next_strip(pos, cb, cty8, ctx);
    issue(cu >= 0, cb, cty8, ctx);
    __syncthreads();                                                 // weights + bias visible (hipcc's barrier also lands the DMA and the first strip's loads)

    [[maybe_unused]] float run[MODE == 1 ? NV : 1];                  // CALayer channel sums carried across this wave's strips
#pragma unroll
    for (int e = 0; e < (MODE == 1 ? NV : 1); ++e) run[e] = 0.f;
    [[maybe_unused]] int covered = 0;                                // compact sums: images [0, covered) have this wave's slot written (a run total or zeros)

    while (cu >= 0) {
        // MODE 0 only: the eight waves meet once per region, so the rows two neighbouring strips share are fetched within one region's time of each other and
        // the second fetch hits L2 (see the header: 1.54x -> 1.02x of the input from beyond L2, same time).  A wave that has run out of strips has ended and is
        // not waited for; a wave that skipped a region (bottom band) is one region ahead from then on -- ordering only, never correctness.
        if constexpr (MODE == 0) { if (!(a.dbg_flags & 32)) __builtin_amdgcn_s_barrier(); }   // conv_flags 32: A/B, free-running waves
        // ---- commit: the strip's units, registers -> this wave's LDS strip (the same wave reads them back: program order, no barrier)
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            if (64 * k + 63 < NU || lane + 64 * k < NU) {
                const int u = lane + 64 * k;
                if constexpr (LINEAR) *reinterpret_cast<uint4*>(s_my + u * 16) = r[k];
                else *reinterpret_cast<uint4*>(s_my + (u / UPT) * SPIX + (u % UPT) * 16) = r[k];
            }
        }
        const int y0 = cty8 * kTH, x0 = ctx * kTW, sp = cty8 * a.tiles_x + ctx;
        int nb = 0, nty8 = 0, ntx = 0;
        const int nu = next_strip(cu + stride, nb, nty8, ntx);
        // ---- the residual of THIS strip first, then the next strip's loads: the epilogue's wait then covers the residual only
        [[maybe_unused]] unsigned rpre[MODE == 2 ? 4 : 1][MODE == 2 ? D8::NRH : 1];
        if constexpr (MODE == 2) D8::res_prefetch(a, cb, y0, x0, 0, ftid, rpre);
        issue(nu >= 0, nb, nty8, ntx);

        f32x4 acc[4][NT];                                            // initial C operand = bias
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float4 t4 = *reinterpret_cast<const float4*>(s_bias + q * NV + nt * 4);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) acc[pt][nt] = f32x4{t4.x, t4.y, t4.z, t4.w};
        }
        if (!(a.dbg_flags & 2)) D::template mma_steps<0, STEPS, 0, true>(s_my, s_w, lane_x, lane_w, q, lo, acc);

        if constexpr (MODE == 1) {
            const bool flush = nu < 0 || nb != cb;                   // uniform: this wave's next strip belongs to another image (or there is none)
            int rslot = -1;
            if (!a.sums_compact) {
                if (!flush) D8::zero_slot1(a, cb, 4 * sp + (wave & 3), lane);   // legacy layout: a strip whose sums are carried on leaves ZEROS in its (tile, wave) slot
            } else if (flush) {                                      // compact layout: slot (residue class of the region walk) * 8 + wave; skipped images get zeros
                for (int bb = covered; bb < cb; ++bb) D8::zero_slot1(a, bb, 8 * D8::compact_residue(pos, bb, sp_total, stride) + wave, lane);
                covered = cb + 1;
                rslot = 8 * D8::compact_residue(pos, cb, sp_total, stride) + (wave & 4);        // + (ftid >> 6) = wave & 3 inside the epilogue
            }
            if (a.ep_key == D8::EP_SUMS) D8::template epilogue_fast_impl<D8::EP_SUMS, true>(a, cb, y0, x0, sp, 0, ftid, acc, run, flush, rslot);
            else D8::template epilogue_fast_impl<D8::EP_RELU | D8::EP_SUMS, true>(a, cb, y0, x0, sp, 0, ftid, acc, run, flush, rslot);
        } else if constexpr (MODE == 2) {
            if (a.out_mode == RC_OUT_NHWC_DWT) D8::template epilogue_dwt<D8::EP_RES>(a, cb, y0, x0, ftid, acc, s_my, rpre);      // uniform: (conv + skip) -> Haar DWT
            else D8::epilogue_res_pre(a, cb, y0, x0, 0, ftid, acc, rpre, gated_out ? s_gate + cb * a.cout : nullptr);
        } else {
            if (a.out_mode == RC_OUT_NHWC_DWT) {                      // uniform: conv -> Haar DWT, the strip's output staged through its own (now dead) halo strip
                const unsigned nor[1][1] = {{0u}};
                if (a.ep_key == D8::EP_RELU) D8::template epilogue_dwt<D8::EP_RELU>(a, cb, y0, x0, ftid, acc, s_my, nor);
                else if (a.ep_key == D8::EP_LEAKY) D8::template epilogue_dwt<D8::EP_LEAKY>(a, cb, y0, x0, ftid, acc, s_my, nor);
                else D8::template epilogue_dwt<0>(a, cb, y0, x0, ftid, acc, s_my, nor);
            } else {
                D8::template epilogue<true>(a, cb, y0, x0, sp, 0, ftid, acc);
            }
        }
        cu = nu; cb = nb; cty8 = nty8; ctx = ntx;
    }
    if constexpr (MODE == 1) {
        if (a.sums_compact)                                          // the images after this wave's last one (all of them for a wave without strips)
            for (int bb = covered; bb < a.batch; ++bb) D8::zero_slot1(a, bb, 8 * D8::compact_residue(pos, bb, sp_total, stride) + wave, lane);
    }
}

// ==================================================================================================
// Kernel 7: the wave-autonomous form for the 64-CHANNEL layers (64 -> 64: the trunk of LiteISPNet / LiteISPNet_GFM, the codec's ResidualBlocks).
// Those layers had no kernel configuration of their own: 72 KB of packed weights + a 54 KB halo tile do not fit twice per CU, so they ran on the
// general kernel, which re-stages the weights for every 8 x 32 tile in three sub-stages (8 GB of L2 -> LDS traffic per level-0 layer; 2.0 ms per
// layer at 1.34 kW, twice the energy of the 48-channel layers per FLOP).  Here the weights stay resident (one copy per CU) and 8 autonomous waves
// (kernel 6's scheme: private halo strips, loads prefetched one strip ahead, no barrier in the loop) each own 2-row x 32-pixel strips, processed as
// two 2 x 16 SUB-strips so that a private 4 x 18 halo sub-strip is 9 KB: 72 + 8 x 9 KB of LDS.  A wave tile is 2 pixel tiles x 4 cout tiles (32
// accumulator registers, 6 fragment reads per 8 MFMAs).
// LDS layout: a pixel is 8 units of 16 bytes = 128 bytes, which maps every second pixel onto the same banks; padding the slot to 160 bytes (the other
// kernels' remedy) would not fit, so unit u of the pixel in halo column c sits in slot u ^ (c & 7): with that XOR every 16-lane group of a
// ds_read_b128 covers 16 distinct 4-bank groups for all three kx and both channel halves (checked exhaustively: tools/swizzle_check.py).
// Same unit map, same step order, same MFMA chain per pixel as kernels 1-4 -> bit-identical results (tested).
// ==================================================================================================
template <class Cfg>
constexpr bool auto64_eligible() { return sizeof(typename Cfg::elem) == 2 && Cfg::KS == 3 && Cfg::CK == 64 && Cfg::NT == 4; }
constexpr int kA64TWH = 18, kA64PXB = 128, kA64ROWB = kA64TWH * kA64PXB, kA64STRIP = 4 * kA64ROWB;
template <class Cfg>
constexpr int auto64_lds_bytes() { return (int)Cfg::CHUNK_W_BYTES + (kAutoBias + kAutoGate) * 4 + kAutoWaves * kA64STRIP; }

// one MFMA step with the next step's fragment reads interleaved (kernel 7): 4 weight + 2 pixel fragments, 8 MFMAs
template <int I, bool NEXT, int NT>
__device__ __forceinline__ void a64_step(int s, const char* s_my, const char* s_w, int lane_w, const int (&xo)[3][2], const uint4 (&wf)[NT], const uint4 (&xf)[2],
                                         uint4 (&wfn)[NT], uint4 (&xfn)[2], f32x4 (&acc)[2][NT]) {
    constexpr int FR = NT + 2, FM = 2 * NT;
    if constexpr (I < FR) {
        if constexpr (NEXT) {
            const int sn = s + 1, tap = sn >> 1, ky = tap / 3, kx = tap - 3 * ky, h = sn & 1;
            if constexpr (I < NT) wfn[I] = *reinterpret_cast<const uint4*>(s_w + (sn * NT + I) * 1024 + lane_w);
            else xfn[I - NT] = *reinterpret_cast<const uint4*>(s_my + xo[kx][h] + (ky + (I - NT)) * kA64ROWB);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = (I * FM) / FR; m < ((I + 1) * FM) / FR; ++m) Mma<bf16_t>::run(wf[m % NT], xf[m / NT], acc[m / NT][m % NT]);
        __builtin_amdgcn_sched_barrier(0);
        a64_step<I + 1, NEXT, NT>(s, s_my, s_w, lane_w, xo, wf, xf, wfn, xfn, acc);
    }
}

template <class Cfg8, int MODE>
__global__ __launch_bounds__(kAutoThreads) void conv_mfma_auto64_kernel(const ConvArgs a) {
    using D = ConvDev<Cfg8>;
    using T = typename Cfg8::elem;
    constexpr int NT = Cfg8::NT, STEPS = Cfg8::STEPS, NV = 4 * NT, UPT = 8, UNIT = 8, ES = 2, TWH = kA64TWH;
    constexpr int NU = 4 * TWH * UPT, NI = NU / 64, RU = TWH * UPT;                 // 576 units = 9 loads per lane, 144 units per halo row
    static_assert(auto64_eligible<Cfg8>() && NU % 64 == 0 && STEPS == 18, "64-channel 3x3 bf16 form");
    constexpr int EP_RELU = D::EP_RELU, EP_LEAKY = D::EP_LEAKY, EP_RES = D::EP_RES, EP_SUMS = D::EP_SUMS, EP_GATE = D::EP_GATE;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_w = smem;
    float* s_bias = reinterpret_cast<float*>(smem + Cfg8::CHUNK_W_BYTES);
    float* s_gate = s_bias + kAutoBias;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* s_my = smem + Cfg8::CHUNK_W_BYTES + (kAutoBias + kAutoGate) * 4 + wave * kA64STRIP;
    const int q = lane >> 4, n = lane & 15;
    const int wv = wave & 3;                                         // strip of its group's 8 x 32 tile (= the wave index the other kernels' epilogues use)

    for (int kb = wave; kb < STEPS * NT; kb += kAutoWaves)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(static_cast<const char*>(a.wpacked) + kb * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(s_w + kb * 1024), 16, 0, 0);
    for (int i = tid; i < kAutoBias; i += kAutoThreads) s_bias[i] = (a.bias && i < a.cout_packed) ? a.bias[i] : 0.f;
    [[maybe_unused]] const bool gated_out = a.ep_key == (EP_GATE | EP_RES);
    if constexpr (MODE == 2) {
        if (gated_out)
            for (int i = tid; i < a.batch * a.cout; i += kAutoThreads) s_gate[i] = a.out_scale[i];
    }

    // a block is two groups of 4 waves, each walking its own list of 8 x 32 tiles (band-major); XCD x (blockIdx % 8) takes a run of consecutive tiles
    constexpr int NG = kAutoWaves / 4;
    const int sp_total = a.tiles_x * a.tiles_y;
    const int n_units = sp_total * a.batch;
    const int slots = (int)(gridDim.x >> 3) * NG;
    const int pos = (blockIdx.x & 7) * slots + (blockIdx.x >> 3) * NG + (wave >> 2);
    const int stride = (int)gridDim.x * NG;
    const size_t img = (size_t)a.H * a.W * a.cin;
    const unsigned img_bytes = (unsigned)(img * ES);

    auto decode = [&](int u, int& b, int& ty8, int& tx) -> int {
        if (u >= n_units) return -1;
        b = magic_div(u, a.td.sp_total);
        band_decode(u - b * sp_total, a.tiles_x, a.tiles_y, a.td, ty8, tx);
        return u;
    };

    // Load k of a sub-strip covers units u = lane + 64 k: halo pixel u / 8 (row-major in the 4 x 18 sub-strip), 16-byte piece u % 8 = lane & 7.
    const int rdelta = (a.W * a.cin - RU * UNIT) * ES;               // image row pitch - sub-strip row bytes (uniform)
    const int j8 = lane >> 3, cu = lane & 7;
    uint4 r[NI];
    auto issue = [&](bool valid, int b, int ty8, int tx, int half) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(static_cast<const T*>(a.in0) + (size_t)b * img, img_bytes);
        const int gy0 = ty8 * kTH + 2 * wv - 1, gx0 = tx * kTW + 16 * half - 1;
        if (a.dbg_flags & 4) valid = false;
        const bool interior = valid && gy0 >= 0 && gx0 >= 0 && gy0 + 4 <= a.H && gx0 + TWH <= a.W && a.cin_chunk_ok;   // uniform
        if (interior) {
            const int soff = (gy0 * a.W + gx0) * a.cin * ES;
            const int l16 = lane * 16;
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int row0 = (64 * k) / RU, t = (row0 + 1) * RU - 64 * k;        // lanes >= t of this load are in the next halo row
                const int voff = t < 64 ? (lane >= t ? l16 + rdelta : l16) : l16;
                r[k] = buf_load16(rs, voff, soff + 1024 * k + row0 * rdelta);
            }
        } else {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int pix = 8 * k + j8, py = pix / TWH, px = pix - py * TWH;
                const int gy = gy0 + py, gx = gx0 + px;
                const bool ok = valid && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W && cu * UNIT < a.cin;
                r[k] = buf_load16(rs, ok ? ((gy * a.W + gx) * a.cin + cu * UNIT) * ES : kOOB, 0);
            }
        }
    };

    // per-lane operand offsets: pixel column n + kx of a halo row, unit 4 h + q in its swizzled slot
    int xo[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int h = 0; h < 2; ++h) xo[kx][h] = (n + kx) * kA64PXB + (((4 * h + q) ^ ((n + kx) & 7)) * 16);
    const int lane_w = lane * 16;

    int cb = 0, cty8 = 0, ctx = 0, chalf = 0;
    int cu_ = decode(pos, cb, cty8, ctx);
    issue(cu_ >= 0, cb, cty8, ctx, 0);
    __syncthreads();

    float run[MODE == 1 ? NV : 1];
#pragma unroll
    for (int e = 0; e < (MODE == 1 ? NV : 1); ++e) run[e] = 0.f;
    [[maybe_unused]] int covered = 0;                                // compact sums: images [0, covered) have this wave's slot written

    while (cu_ >= 0) {
        // ---- commit: unit (pixel 8 k + lane / 8, piece lane % 8) -> slot piece ^ (column & 7) of the pixel's 128 bytes
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int row0 = (8 * k) / TWH, thr = (row0 + 1) * TWH - 8 * k;
            const int pix = 8 * k + j8;
            const int px = pix - TWH * (row0 + ((thr < 8 && j8 >= thr) ? 1 : 0));
            *reinterpret_cast<uint4*>(s_my + pix * kA64PXB + ((cu ^ (px & 7)) * 16)) = r[k];
        }
        const int gy = cty8 * kTH + 2 * wv, gx0 = ctx * kTW + 16 * chalf, sp = cty8 * a.tiles_x + ctx;
        int nb = cb, nty8 = cty8, ntx = ctx, nu = cu_;
        const int nhalf = chalf ^ 1;
        if (chalf == 1) nu = decode(cu_ + stride, nb, nty8, ntx);
        unsigned rpre[MODE == 2 ? 2 : 1][MODE == 2 ? D::NRH : 1];
        if constexpr (MODE == 2) D::sub_res_prefetch(a, cb, gy, gx0, lane, rpre);
        issue(nu >= 0, nb, nty8, ntx, nhalf);

        f32x4 acc[2][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float4 t4 = *reinterpret_cast<const float4*>(s_bias + q * NV + nt * 4);
            acc[0][nt] = f32x4{t4.x, t4.y, t4.z, t4.w};
            acc[1][nt] = f32x4{t4.x, t4.y, t4.z, t4.w};
        }
        if (!(a.dbg_flags & 2)) {
            uint4 wfa[NT], xfa[2], wfb[NT], xfb[2];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wfa[nt] = *reinterpret_cast<const uint4*>(s_w + nt * 1024 + lane_w);
            xfa[0] = *reinterpret_cast<const uint4*>(s_my + xo[0][0]);
            xfa[1] = *reinterpret_cast<const uint4*>(s_my + xo[0][0] + kA64ROWB);
#pragma unroll
            for (int s = 0; s < STEPS; s += 2) {
                a64_step<0, true, NT>(s, s_my, s_w, lane_w, xo, wfa, xfa, wfb, xfb, acc);
                if (s + 2 < STEPS) a64_step<0, true, NT>(s + 1, s_my, s_w, lane_w, xo, wfb, xfb, wfa, xfa, acc);
                else a64_step<0, false, NT>(s + 1, s_my, s_w, lane_w, xo, wfb, xfb, wfa, xfa, acc);
            }
        }

        if constexpr (MODE == 1) {
            const bool last = chalf == 1;
            const bool flush = last && (nu < 0 || nb != cb);         // uniform: the strip's sums leave with its right half, if the wave's next strip is another image's
            int slot = 4 * sp + wv;                                  // legacy layout: (tile, wave); a strip whose sums are carried on leaves zeros there
            if (!a.sums_compact) {
                if (last && !flush) D::zero_slot1(a, cb, slot, lane);
            } else if (flush) {                                      // compact layout: slot (residue class of this group's tile walk) * 4 + wave
                for (int bb = covered; bb < cb; ++bb) D::zero_slot1(a, bb, 4 * D::compact_residue(pos, bb, sp_total, stride) + wv, lane);
                covered = cb + 1;
                slot = 4 * D::compact_residue(pos, cb, sp_total, stride) + wv;
            }
            const unsigned none[1][1] = {{0u}};
            if (a.ep_key == EP_SUMS) D::template epilogue_sub<EP_SUMS, 1>(a, cb, gy, gx0, slot, lane, acc, run, flush, none, nullptr);
            else if (a.ep_key == (EP_RELU | EP_SUMS)) D::template epilogue_sub<EP_RELU | EP_SUMS, 1>(a, cb, gy, gx0, slot, lane, acc, run, flush, none, nullptr);
            else D::template epilogue_sub<EP_LEAKY | EP_SUMS, 1>(a, cb, gy, gx0, slot, lane, acc, run, flush, none, nullptr);
        } else if constexpr (MODE == 2) {
            float none[1] = {0.f};
            if (gated_out) D::template epilogue_sub<EP_GATE | EP_RES, 2>(a, cb, gy, gx0, 0, lane, acc, none, false, rpre, s_gate + cb * a.cout);
            else D::template epilogue_sub<EP_RES, 2>(a, cb, gy, gx0, 0, lane, acc, none, false, rpre, nullptr);
        } else {
            float none[1] = {0.f};
            const unsigned nonr[1][1] = {{0u}};
            if (a.ep_key == EP_RELU) D::template epilogue_sub<EP_RELU, 1>(a, cb, gy, gx0, 0, lane, acc, none, false, nonr, nullptr);
            else if (a.ep_key == EP_LEAKY) D::template epilogue_sub<EP_LEAKY, 1>(a, cb, gy, gx0, 0, lane, acc, none, false, nonr, nullptr);
            else D::template epilogue_sub<0, 1>(a, cb, gy, gx0, 0, lane, acc, none, false, nonr, nullptr);
        }
        cu_ = nu; cb = nb; cty8 = nty8; ctx = ntx; chalf = nhalf;
    }
    if constexpr (MODE == 1) {
        if (a.sums_compact)
            for (int bb = covered; bb < a.batch; ++bb) D::zero_slot1(a, bb, 4 * D::compact_residue(pos, bb, sp_total, stride) + wv, lane);
    }
}

template <class Cfg>
constexpr int ws_lds_bytes() { return 2 * Cfg::IN_BYTES + (int)Cfg::CHUNK_W_BYTES + 16 * Cfg::NT * 4; }

// ---- host side: per-instantiation launcher ----------------------------------------------------------
template <class Cfg>
constexpr int persist_lds_bytes() { return persist_lds_bytes_c<Cfg>(); }

template <class Cfg, bool FAST>
int launch_wst(const ConvArgs& a, int grid, hipStream_t stream) {
    constexpr int WST_LDS = wst_lds_bytes<Cfg>();
    static PerDeviceFlag attr_thin;                      // function attributes are per device (common.hpp)
    if (!attr_thin.test_and_set())
        RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_wst_kernel<Cfg, FAST>), hipFuncAttributeMaxDynamicSharedMemorySize, WST_LDS));
    hipLaunchKernelGGL((conv_mfma_wst_kernel<Cfg, FAST>), dim3((unsigned)grid), dim3(kWsmThreads), WST_LDS, stream, a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

template <class Cfg, bool GATED, bool FAST>
int launch_conv_g(const ConvArgs& a, hipStream_t stream) {
    constexpr int P_LDS = persist_lds_bytes<Cfg>();
    constexpr bool P_OK = P_LDS <= 80 * 1024;          // two persistent blocks per CU
    const int n_tiles = a.tiles_x * a.tiles_y * a.batch;
    // Channel-sum slots.  Every kernel writes the LEGACY layout (one slot per (8 x 32 tile, wave): 4 x tiles per image); the carried-sums kernels (2, 6, 7) also have the
    // COMPACT one (compact_residue(): grid-many residue classes x waves).  rc_conv_sum_slots() asks THIS function (a.query) which branch a launch will take and how many
    // slots it fills, so the host's allocation cannot drift from the dispatch; a launch whose chan_sums_slots is neither count is an error, never a guess.
    const int legacy_slots = 4 * a.tiles_x * a.tiles_y;
    auto report = [&](int kind, int slots) -> bool {         // query mode: {kind (0 legacy-only kernel, 2 / 6 / 7), slots per image}; nothing is launched
        if (a.query == nullptr) return false;
        a.query[0] = kind; a.query[1] = slots;
        return true;
    };
    auto sums_mode = [&](int compact_slots, int& compact) -> int {    // which layout did the caller allocate?
        compact = 0;
        if (a.chan_sums == nullptr || a.sum_slots == legacy_slots) return RC_OK;
        if (compact_slots > 0 && a.sum_slots == compact_slots) { compact = 1; return RC_OK; }
        return fail(RC_ERR_INVALID, "rc_conv2d: chan_sums_slots matches neither layout of the kernel this launch takes (ask rc_conv_sum_slots)");
    };
    if (a.out_mode == RC_OUT_NHWC_DWT) {
        // conv -> networks.DWTForward in one launch: a form of kernel 6 (mode 0) only, whatever the `persist` / `persist_auto` knobs say
        if constexpr (!GATED && FAST && auto_eligible<Cfg>()) {
            using DD = ConvDev<Cfg>;
            static_assert(StripCfg<Cfg>::IN_BYTES >= 2 * kTW * (Cfg::COUT_TILE * 2 + 8), "the strip's output fits its own halo strip");
            if ((a.ep_key == 0 || a.ep_key == DD::EP_RELU || a.ep_key == DD::EP_LEAKY || a.ep_key == DD::EP_RES) && a.n_chunks == 1 && a.n_ct == 1 && a.cout == Cfg::COUT_TILE &&
                a.cin_vec_ok && a.cin_chunk_ok && n_tiles < (1 << 24)) {
                if (report(0, legacy_slots)) return RC_OK;
                constexpr int A_LDS = auto_lds_bytes<Cfg>();
                const int n_items = a.tiles_x * ((a.H + kWsmTH - 1) / kWsmTH) * a.batch;
                int grid = a.num_cus;
                if (grid > n_items) grid = n_items;
                grid = (grid + 7) / 8 * 8;
                static PerDeviceFlag attr_set;
                if (!attr_set.test_and_set()) {
                    RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_auto_kernel<Cfg, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS));
                    RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_auto_kernel<Cfg, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS));
                }
                if (a.ep_key == DD::EP_RES) hipLaunchKernelGGL((conv_mfma_auto_kernel<Cfg, 2>), dim3((unsigned)grid), dim3(kAutoThreads), A_LDS, stream, a);      // residual prefetched
                else hipLaunchKernelGGL((conv_mfma_auto_kernel<Cfg, 0>), dim3((unsigned)grid), dim3(kAutoThreads), A_LDS, stream, a);
                RC_HIP_CHECK(hipGetLastError());
                return RC_OK;
            }
        }
        return fail(RC_ERR_UNSUPPORTED, "rc_conv2d: RC_OUT_NHWC_DWT needs a bf16 3x3 layer of one Cin chunk and one 32- or 48-wide cout tile (cin == cout == 32 or 48), 16-byte aligned operands");
    }
    constexpr int WS_LDS = ws_lds_bytes<Cfg>();
    if constexpr (WS_LDS <= 150 * 1024) {              // one 8-wave producer/consumer block per CU
        // persist_ok: 1 = automatic (producer/consumer form for the register-starved variants: gated input or 80-wide
        // cout tiles, whose prefetch registers would otherwise spill), 2 = wherever eligible, 3 = never
        const bool ws_auto = GATED || Cfg::NT == 5 || Cfg::CK == 80;
        if (a.n_chunks == 1 && a.n_ct == 1 && a.cin_vec_ok && (a.persist_ok == 2 || (a.persist_ok == 1 && ws_auto)) && n_tiles < (1 << 24)) {
            if (report(0, legacy_slots)) return RC_OK;
            { int c_; if (int e_ = sums_mode(0, c_)) return e_; }
            static PerDeviceFlag attr_set;                       // function attributes are per device (common.hpp)
            if (!attr_set.test_and_set()) {
                RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_ws_kernel<Cfg, GATED, FAST>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS));
            }
            int grid = a.num_cus;
            if (grid > n_tiles) grid = n_tiles;
            grid = (grid + 7) / 8 * 8;
            hipLaunchKernelGGL((conv_mfma_ws_kernel<Cfg, GATED, FAST>), dim3((unsigned)grid), dim3(kWsThreads), WS_LDS, stream, a);
            RC_HIP_CHECK(hipGetLastError());
            return RC_OK;
        }
    }
    if constexpr (!GATED && FAST && sizeof(typename Cfg::elem) == 2 && Cfg::KS == 3 && Cfg::STEPS >= 2 && pss_lds_bytes<Cfg>() <= 160 * 1024 &&
                  (kWsmTH * kTW * Cfg::COUT_TILE * 2) % (16 * kThreads * 2) == 0) {
        // single-chunk pixel-shuffle layers: output staged through LDS, stored by the loader waves (kernel 5)
        if (a.pss && a.n_chunks == 1 && a.n_ct == 4 && a.cin_vec_ok && a.cin_chunk_ok && a.persist_ok && n_tiles < (1 << 24)) {
            if (report(0, legacy_slots)) return RC_OK;
            constexpr int PSS_LDS = pss_lds_bytes<Cfg>();
            const int n_items = a.tiles_x * ((a.H + kWsmTH - 1) / kWsmTH) * a.batch;
            static PerDeviceFlag attr_set;
            if (!attr_set.test_and_set()) {
                RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_pss_kernel<Cfg>), hipFuncAttributeMaxDynamicSharedMemorySize, PSS_LDS));
            }
            int grid = a.num_cus;
            if (grid > n_items) grid = n_items;
            grid = (grid + 7) / 8 * 8;
            hipLaunchKernelGGL((conv_mfma_pss_kernel<Cfg>), dim3((unsigned)grid), dim3(kWsmThreads), PSS_LDS, stream, a);
            RC_HIP_CHECK(hipGetLastError());
            return RC_OK;
        }
    }
    constexpr int WSM_LDS = wsm_lds_bytes<Cfg>();
    // multi-chunk producer/consumer form
    // bf16 only: in fp32 the MFMAs are 4x longer, the layers are MFMA-bound either way and the general kernel's many
    // small blocks balance the B = 1 configurations better (measured 99 vs 88 TF/s on 64 -> 64 at 1080p)
    // (kernels 2 and 4 are instantiated for plain inputs only: their gated forms -- the round-3 schedule that folded the CALayer gate into the next conv's
    // staging -- spilled 5-77 registers each, 14 instantiations, and nothing on the default path has dispatched to them since the early gate.
    // A gated input takes kernel 3 (single chunk) or kernel 1 (several chunks): the same bits, tested in every persist mode.)
    if constexpr (!GATED && WSM_LDS <= 160 * 1024 && Cfg::KS >= 2 && Cfg::STEPS >= 2 && sizeof(typename Cfg::elem) == 2) {
        // one-chunk layers with several cout tiles AND a residual (the U-Net's 48 -> 192 + skip at level 1) stay on the persistent kernel: input tile
        // resident, weights per cout tile, the residual's loads issued before the MFMA loop (1.13 vs 1.31 ms here, 1.25 in the 32x32x16 form);
        // persist_ok 3 ("persistent only") sends every one-chunk layer there
        const bool res_pre_form = FAST && !GATED && sizeof(typename Cfg::elem) == 2 && Cfg::NT <= 3 && a.ep_key == ConvDev<Cfg>::EP_RES && a.out_mode == RC_OUT_NHWC;
        if ((a.n_chunks > 1 || (a.n_ct > 1 && a.persist_ok != 3 && !(P_OK && res_pre_form))) && a.cin_vec_ok && a.cin_chunk_ok && a.persist_ok && a.cout_packed <= kPersistMaxCout &&
            n_tiles < (1 << 24)) {
            if (report(0, legacy_slots)) return RC_OK;
            { int c_; if (int e_ = sums_mode(0, c_)) return e_; }
            const int n_items = a.tiles_x * ((a.H + kWsmTH - 1) / kWsmTH) * a.batch * (a.n_chunks > 1 ? a.n_ct : 1);
            int grid = a.num_cus;
            if (grid > n_items) grid = n_items;
            grid = (grid + 7) / 8 * 8;
            if constexpr ((FAST || Cfg::NT <= 3) && wst_eligible<Cfg>()) {   // (with the generic epilogue the 64-wide cout tiles spilled 1-22 registers: those stay on kernel 4)
                if (a.thin >= (wst_form<Cfg>() == 3 ? 1 : 2)) return launch_wst<Cfg, FAST>(a, grid, stream);   // thin stages: one barrier per stage (kernel 4b), rc_debug_set("thin", 0) for kernel 4
            }
            static PerDeviceFlag attr_set;                       // function attributes are per device (common.hpp)
            if (!attr_set.test_and_set()) {
                RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_wsm_kernel<Cfg, GATED, FAST>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, WSM_LDS));
            }
            hipLaunchKernelGGL((conv_mfma_wsm_kernel<Cfg, GATED, FAST>), dim3((unsigned)grid), dim3(kWsmThreads), WSM_LDS, stream, a);
            RC_HIP_CHECK(hipGetLastError());
            return RC_OK;
        }
    }
    if constexpr (!GATED && FAST && auto64_eligible<Cfg>()) {
        // the 64-channel wave-autonomous form (kernel 7): 64 -> 64, one cout tile, rc_debug_set("persist_auto", != 0)
        using DD = ConvDev<Cfg>;
        const int k7 = a.ep_key;
        const bool key_ok = k7 == 0 || k7 == DD::EP_RELU || k7 == DD::EP_LEAKY || k7 == DD::EP_SUMS || k7 == (DD::EP_RELU | DD::EP_SUMS) || k7 == (DD::EP_LEAKY | DD::EP_SUMS) ||
                            k7 == DD::EP_RES || (k7 == (DD::EP_GATE | DD::EP_RES) && a.batch * a.cout <= kAutoGate);
        if (a.auto_impl && key_ok && a.n_chunks == 1 && a.n_ct == 1 && a.cout == Cfg::COUT_TILE && a.cin_vec_ok && a.persist_ok && a.out_mode == RC_OUT_NHWC && n_tiles < (1 << 24)) {
            constexpr int A_LDS = auto64_lds_bytes<Cfg>();
            static_assert(A_LDS <= 160 * 1024, "kernel 7 LDS");
            int grid = a.num_cus;
            if (grid * 2 > n_tiles) grid = (n_tiles + 1) / 2;      // two groups of four waves per block, one 8 x 32 tile each
            grid = (grid + 7) / 8 * 8;
            const int cslots = ((k7 & DD::EP_SUMS) && a.sums_compact_ok && grid * 8 < legacy_slots) ? grid * 8 : 0;   // compact sums: (grid x 2 groups) residue classes x 4 waves (small images: the per-tile layout is smaller)
            if (report(7, cslots ? cslots : legacy_slots)) return RC_OK;
            ConvArgs aa = a;
            if (int e_ = sums_mode(cslots, aa.sums_compact)) return e_;
            static PerDeviceFlag attr_set;
            if (!attr_set.test_and_set()) {
                RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_auto64_kernel<Cfg, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS));
                RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_auto64_kernel<Cfg, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS));
                RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_auto64_kernel<Cfg, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS));
            }
            if (k7 & DD::EP_SUMS) hipLaunchKernelGGL((conv_mfma_auto64_kernel<Cfg, 1>), dim3((unsigned)grid), dim3(kAutoThreads), A_LDS, stream, aa);
            else if (k7 & DD::EP_RES) hipLaunchKernelGGL((conv_mfma_auto64_kernel<Cfg, 2>), dim3((unsigned)grid), dim3(kAutoThreads), A_LDS, stream, aa);
            else hipLaunchKernelGGL((conv_mfma_auto64_kernel<Cfg, 0>), dim3((unsigned)grid), dim3(kAutoThreads), A_LDS, stream, aa);
            RC_HIP_CHECK(hipGetLastError());
            return RC_OK;
        }
    }
    if constexpr (!GATED && FAST && auto_eligible<Cfg>()) {
        // wave-autonomous persistent form (kernel 6): rc_debug_set("persist_auto", 0 / 1 / 2)
        using DD = ConvDev<Cfg>;
        // keys whose epilogue would load per-image vectors or a second map from global memory after the MFMA loop (FiLM, x (lsc + 1)) stay on kernel 2
        const bool key_ok = a.ep_key >= 0 && (a.ep_key & (DD::EP_FILM | DD::EP_MUL)) == 0 && (a.ep_key != (DD::EP_GATE | DD::EP_RES) || a.batch * a.cout <= kAutoGate);
        const bool mode_ok = a.auto_impl == 2 || (a.ep_key & DD::EP_RES) == 0;      // persist_auto 1 (default): the residual forms stay on kernel 2 (4-5 % faster there); 2: every eligible form
        if (a.auto_impl && key_ok && mode_ok && a.n_chunks == 1 && a.n_ct == 1 && a.cin_vec_ok && a.persist_ok && a.out_mode == RC_OUT_NHWC && n_tiles < (1 << 24)) {
            constexpr int A_LDS = auto_lds_bytes<Cfg>();
            const int n_items = a.tiles_x * ((a.H + kWsmTH - 1) / kWsmTH) * a.batch;
            int grid = a.num_cus;
            if (grid > n_items) grid = n_items;
            grid = (grid + 7) / 8 * 8;
            const bool run_sums = a.ep_key == DD::EP_SUMS || a.ep_key == (DD::EP_RELU | DD::EP_SUMS);
            const int cslots = (run_sums && a.sums_compact_ok && grid * 8 < legacy_slots) ? grid * 8 : 0;             // compact sums: grid residue classes of the region walk x 8 waves
            if (report(6, cslots ? cslots : legacy_slots)) return RC_OK;
            ConvArgs aa = a;
            if (int e_ = sums_mode(cslots, aa.sums_compact)) return e_;
            static PerDeviceFlag attr_set;
            if (!attr_set.test_and_set()) {
                RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_auto_kernel<Cfg, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS));
                RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_auto_kernel<Cfg, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS));
                RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_auto_kernel<Cfg, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS));
            }
            if (run_sums)
                hipLaunchKernelGGL((conv_mfma_auto_kernel<Cfg, 1>), dim3((unsigned)grid), dim3(kAutoThreads), A_LDS, stream, aa);
            else if (a.ep_key == DD::EP_RES || a.ep_key == (DD::EP_GATE | DD::EP_RES))
                hipLaunchKernelGGL((conv_mfma_auto_kernel<Cfg, 2>), dim3((unsigned)grid), dim3(kAutoThreads), A_LDS, stream, aa);
            else
                hipLaunchKernelGGL((conv_mfma_auto_kernel<Cfg, 0>), dim3((unsigned)grid), dim3(kAutoThreads), A_LDS, stream, aa);
            RC_HIP_CHECK(hipGetLastError());
            return RC_OK;
        }
    }
    if constexpr (P_OK && !GATED) {
        if (a.n_chunks == 1 && a.cout_packed <= persist_bias_slots<Cfg>() && a.persist_ok && n_tiles < (1 << 24)) {
            int grid = persist_blocks_per_cu<Cfg>() * a.num_cus;
            const bool one_per_cu = (a.dbg_flags & 64) != 0 && P_LDS <= 80 * 1024;      // occupancy experiment: LDS padded so that ONE block fits a CU
            if (one_per_cu) grid = a.num_cus;
            if (grid > n_tiles) grid = n_tiles;
            grid = (grid + 7) / 8 * 8;
            // the carried-sums (RUN) form of this kernel: fast epilogue, bf16, >= 3 cout tiles per block, one cout tile per layer
            const bool run_sums = FAST && sizeof(typename Cfg::elem) == 2 && Cfg::NT >= 3 && a.n_ct == 1 &&
                                  (a.ep_key == ConvDev<Cfg>::EP_SUMS || a.ep_key == (ConvDev<Cfg>::EP_RELU | ConvDev<Cfg>::EP_SUMS));
            const int cslots = (run_sums && a.sums_compact_ok && grid * 4 < legacy_slots) ? grid * 4 : 0;             // compact sums: grid residue classes of the tile walk x 4 waves
            if (report(2, cslots ? cslots : legacy_slots)) return RC_OK;
            ConvArgs aa = a;
            if (int e_ = sums_mode(cslots, aa.sums_compact)) return e_;
            // several cout tiles whose packed weights all fit beside the tile with two blocks per CU: resident (the kernel's header comment)
            const int lds_res = P_LDS + (a.n_ct - 1) * (int)Cfg::CHUNK_W_BYTES;
            aa.w_resident = (a.n_ct > 1 && lds_res <= 80 * 1024 && !(a.dbg_flags & 128)) ? 1 : 0;       // conv_flags 128: A/B
            const int lds_bytes = one_per_cu ? 100 * 1024 : (aa.w_resident ? lds_res : P_LDS);
            static PerDeviceFlag attr_set;                       // function attributes are per device (common.hpp)
            if (!attr_set.test_and_set()) {
                RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_persist_kernel<Cfg, GATED, FAST>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS > 100 * 1024 ? P_LDS : 100 * 1024));
            }
            hipLaunchKernelGGL((conv_mfma_persist_kernel<Cfg, GATED, FAST>), dim3((unsigned)grid), dim3(kThreads), lds_bytes, stream, aa);
            RC_HIP_CHECK(hipGetLastError());
            return RC_OK;
        }
    }
    if (report(0, legacy_slots)) return RC_OK;
    { int c_; if (int e_ = sums_mode(0, c_)) return e_; }
    static PerDeviceFlag attr_set;                       // function attributes are per device (common.hpp)
    if (!attr_set.test_and_set()) {
        RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<Cfg, GATED, FAST>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
    }
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.n_ct), (unsigned)a.batch, 1);
    hipLaunchKernelGGL((conv_mfma_kernel<Cfg, GATED, FAST>), grid, dim3(kThreads), Cfg::LDS_BYTES, stream, a);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

template <class Cfg>
int launch_conv(const ConvArgs& a, hipStream_t stream) {
    // the gated form (x = in0*gate + in1) only occurs on vectorisable layers inside RCAGroups
    const bool gated = a.in_gate != nullptr && a.cin_vec_ok;
    if (a.ep_key >= 0) return gated ? launch_conv_g<Cfg, true, true>(a, stream) : launch_conv_g<Cfg, false, true>(a, stream);
    return gated ? launch_conv_g<Cfg, true, false>(a, stream) : launch_conv_g<Cfg, false, false>(a, stream);
}

// defined in conv_dispatch.hip; the instantiations live in conv_inst_*.hip
int dispatch_conv(bool bf16, int ksize, int ck, int nt, const ConvArgs& a, hipStream_t s);

}  // namespace rc
