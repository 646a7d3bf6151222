// On-device entropy coding for the codecs' compress() / decompress() (upstream models/tcm.py:511-570, 592-637;
// models/raw2bit.py:1876-1944, 1961-2027) -- SURVEY.md 8f rank 3.
//
// Upstream dumps every latent to the host (`.tolist()`, raw2bit.py:1943-1944) and feeds CompressAI's C++ rANS coder one Python
// list at a time.  Here the symbols never leave the GPU until they are bytes:
//   rc_gc_symbols / rc_eb_symbols   quantise (round(y - mean)), look up the CDF index of every element (scale table search /
//                                   channel), write y_hat -- one pass, int32 symbols + indexes in the coder's (c, h, w) order
//   rc_rans_encode_chunks           rANS, ONE LANE PER CHUNK of `chunk` consecutive symbols: every chunk is a complete stream in
//                                   CompressAI's BufferedRansEncoder layout (64-bit state, 32-bit words emitted back to front,
//                                   lower bound 2^31, 16-bit probabilities, 4-bit bypass escapes), so chunk = all symbols
//                                   reproduces CompressAI's single stream byte for byte, and smaller chunks are the parallel form
//                                   a GPU needs (a rANS state is a serial dependency; 2 M symbols per 4K frame on one lane would
//                                   take ~0.2 s).  A container (rc_rans_container_*) prefixes the chunk sizes.
//   rc_rans_decode_chunks           the inverse: binary search of the 16-bit cumulative frequency in the element's CDF row
//   rc_rans_encode_host / _decode_host   the same primitives compiled for the host: the single-stream CompressAI-layout form
//                                   (format="compressai") for interchange with a CompressAI decoder.
// All of it is integer arithmetic: bit-exact against the C oracle (oracle/rans_oracle.c), tests/test_bitstream.py.
// The coder's definition is CompressAI's (absent from /root/reference, unpinned upstream): restated, parity unpinned.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "common.hpp"

namespace rc {
namespace ans {

constexpr uint64_t kL = 1ull << 31;
constexpr uint32_t kPrec = 16, kBypassBits = 4, kMaxBypass = (1u << kBypassBits) - 1;

struct Tables { const int32_t* cdf; int stride; int n_cdfs; const int32_t* sizes; const int32_t* offsets; };

__host__ __device__ inline bool put(uint64_t& x, uint32_t*& ptr, uint32_t* begin, uint32_t start, uint32_t freq) {
    const uint64_t x_max = ((kL >> kPrec) << 32) * freq;
    if (x >= x_max) {
        if (ptr == begin) return false;
        *--ptr = (uint32_t)x;
        x >>= 32;
    }
    x = ((x / freq) << kPrec) + (x % freq) + start;
    return true;
}

__host__ __device__ inline bool put_bits(uint64_t& x, uint32_t*& ptr, uint32_t* begin, uint32_t val) {
    const uint32_t freq = 1u << (16 - kBypassBits);
    const uint64_t x_max = ((kL >> 16) << 32) * freq;
    if (x >= x_max) {
        if (ptr == begin) return false;
        *--ptr = (uint32_t)x;
        x >>= 32;
    }
    x = (x << kBypassBits) | val;
    return true;
}

// Encode symbols [i0, i1) as one complete stream whose last word is end[-1]; returns the first word, or NULL (bad index / no room).
// Symbols are coded LAST TO FIRST (the decoder pops them first to last); an escaped symbol's pieces are therefore emitted in the
// reverse of BufferedRansEncoder's push order: nibbles high to low, the length's base-15 digits (final digit, then the 15s), the
// escape symbol itself.
__host__ __device__ inline uint32_t* encode_range(const int32_t* sym, const int32_t* idx, long i0, long i1, const Tables& t, uint32_t* begin,
                                                  uint32_t* end) {
    uint64_t x = kL;
    uint32_t* ptr = end;
    for (long i = i1 - 1; i >= i0; --i) {
        const int32_t ci = idx[i];
        if (ci < 0 || ci >= t.n_cdfs) return nullptr;
        const int32_t* cdf = t.cdf + (long)ci * t.stride;
        const int32_t max_value = t.sizes[ci] - 2;
        int32_t value = sym[i] - t.offsets[ci];
        uint32_t raw = 0;
        if (value < 0) { raw = (uint32_t)(-2 * value - 1); value = max_value; }
        else if (value >= max_value) { raw = (uint32_t)(2 * (value - max_value)); value = max_value; }
        if (value == max_value) {
            int n_bypass = 0;
            while (n_bypass < 8 && (raw >> (n_bypass * kBypassBits)) != 0) ++n_bypass;
            for (int j = n_bypass - 1; j >= 0; --j)
                if (!put_bits(x, ptr, begin, (raw >> (j * kBypassBits)) & kMaxBypass)) return nullptr;
            if (!put_bits(x, ptr, begin, (uint32_t)n_bypass % kMaxBypass)) return nullptr;
            for (uint32_t k = 0; k < (uint32_t)n_bypass / kMaxBypass; ++k)
                if (!put_bits(x, ptr, begin, kMaxBypass)) return nullptr;
        }
        if (!put(x, ptr, begin, (uint32_t)cdf[value], (uint32_t)(cdf[value + 1] - cdf[value]))) return nullptr;
    }
    if (ptr - begin < 2) return nullptr;
    ptr -= 2;
    ptr[0] = (uint32_t)x;
    ptr[1] = (uint32_t)(x >> 32);
    return ptr;
}

// ---- division-free put (ryg rans64 "RansEncSymbol"): bit-identical to put(), the per-symbol reciprocal is computed where it is cheap ----
// q = floor(x / freq) = mulhi(x, rcp) >> shift for every x < 2^63, with shift = ceil(log2 freq) - 1 and rcp = ceil(2^(shift + 64) / freq);
// freq == 1 takes rcp = 2^64 - 1, shift 0 and the bias start + 2^16 - 1 (mulhi gives x - 1).  Then x' = x + bias + q * (2^16 - freq).
struct Rcp { uint64_t rcp; uint32_t shift; };
__host__ __device__ inline Rcp make_rcp(uint32_t freq) {
    Rcp r;
    if (freq < 2) { r.rcp = ~0ull; r.shift = 0; return r; }
    uint32_t shift = 0;
    while (freq > (1u << shift)) ++shift;
    // ((1 << (shift + 63)) + freq - 1) / freq as a 128 / 32-bit long division in two 64-bit halves
    uint64_t x0 = freq - 1;
    const uint64_t x1 = 1ull << (shift + 31);
    const uint64_t t1 = x1 / freq;
    x0 += (x1 % freq) << 32;
    const uint64_t t0 = x0 / freq;
    r.rcp = t0 + (t1 << 32);
    r.shift = shift - 1;
    return r;
}
__host__ __device__ inline uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
__host__ __device__ inline bool put_rcp(uint64_t& x, uint32_t*& ptr, uint32_t* begin, uint32_t start, uint32_t freq, const Rcp& r) {
    const uint64_t x_max = ((kL >> kPrec) << 32) * freq;
    if (x >= x_max) {
        if (ptr == begin) return false;
        *--ptr = (uint32_t)x;
        x >>= 32;
    }
    const uint64_t q = mulhi64(x, r.rcp) >> r.shift;
    const uint64_t bias = freq < 2 ? (uint64_t)start + (1u << kPrec) - 1 : (uint64_t)start;
    x = x + bias + q * ((1u << kPrec) - freq);
    return true;
}

// n_words: length of the stream in 32-bit words.  A renormalisation that would read past it sets `bad` (and reads nothing): a truncated or
// corrupt stream ends in an error, never in an out-of-bounds read.
struct DecState { uint64_t x; long pos; long n_words; bool bad; };

__host__ __device__ inline uint32_t next_word(DecState& st, const uint32_t* words) {
    if (st.pos >= st.n_words) { st.bad = true; return 0u; }
    return words[st.pos++];
}

__host__ __device__ inline uint32_t get_bits(DecState& st, const uint32_t* words) {
    uint64_t x = st.x;
    const uint32_t val = (uint32_t)(x & kMaxBypass);
    x >>= kBypassBits;
    if (x < kL) x = (x << 32) | next_word(st, words);
    st.x = x;
    return val;
}

// decode symbols [i0, i1) from a stream (state carried in st: call with x = words[0] | words[1] << 32, pos = 2 at its start)
__host__ __device__ inline bool decode_range(const uint32_t* words, DecState& st, const int32_t* idx, long i0, long i1, const Tables& t,
                                             int32_t* out) {
    for (long i = i0; i < i1; ++i) {
        const int32_t ci = idx[i];
        if (ci < 0 || ci >= t.n_cdfs) return false;
        const int32_t* cdf = t.cdf + (long)ci * t.stride;
        const int32_t size = t.sizes[ci], max_value = size - 2;
        const uint32_t cum = (uint32_t)(st.x & ((1u << kPrec) - 1));
        int lo = 0, hi = size;                                   // first entry > cum (the CDF is strictly increasing up to size)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((uint32_t)cdf[mid] <= cum) lo = mid + 1; else hi = mid;
        }
        const int32_t s = lo - 1;
        {
            const uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
            uint64_t x = (uint64_t)freq * (st.x >> kPrec) + (st.x & ((1u << kPrec) - 1)) - start;
            if (x < kL) x = (x << 32) | next_word(st, words);
            st.x = x;
        }
        int32_t value = s;
        if (value == max_value) {
            int32_t val = (int32_t)get_bits(st, words), n_bypass = val;
            while (val == (int32_t)kMaxBypass && !st.bad && n_bypass <= 8) { val = (int32_t)get_bits(st, words); n_bypass += val; }
            if (n_bypass > 8) st.bad = true;                     // a 32-bit escape has at most 8 nibbles: anything longer is a corrupt stream
            if (st.bad) return false;
            uint32_t raw = 0;
            for (int32_t j = 0; j < n_bypass; ++j) raw |= get_bits(st, words) << (j * kBypassBits);
            value = (int32_t)(raw >> 1);
            if (raw & 1) value = -value - 1; else value += max_value;
        }
        out[i] = value + t.offsets[ci];
        if (st.bad) return false;
    }
    return true;
}

// ---- kernels ------------------------------------------------------------------------------------------------------------------
// Encoding in two launches.  Everything about a symbol except the state update is independent of the rANS state: which CDF row, the
// escape decision, (start, freq) and the reciprocal of freq.  prepare_kernel does that for all symbols in parallel and writes the
// operations TRANSPOSED, op i of chunk c at [i][c], so the lanes of the serial kernel (one per chunk) read consecutive addresses; the
// serial kernel is then ~25 integer instructions per symbol with no dependent loads and no 64-bit division.  (As ONE kernel -- idx ->
// sizes/offsets -> cdf as three dependent global loads at an 8 KB lane stride, then a 64-bit division -- a 2048-symbol chunk took 1.5 ms:
// 6 calls = 9 of compress()'s 30 ms per 4K frame, profiles/r03_codec_stream.md.)
struct EncOps { uint64_t* rcp; uint32_t* fs; uint32_t* raw; uint16_t* meta; };   // fs = start | freq << 16; meta = rcp shift | n_bypass << 8 (0xFF: no escape)

__host__ __device__ inline EncOps enc_ops(void* scratch, long n_pad) {
    EncOps o;
    char* p = static_cast<char*>(scratch);
    o.rcp = reinterpret_cast<uint64_t*>(p); p += 8 * n_pad;
    o.fs = reinterpret_cast<uint32_t*>(p); p += 4 * n_pad;
    o.raw = reinterpret_cast<uint32_t*>(p); p += 4 * n_pad;
    o.meta = reinterpret_cast<uint16_t*>(p);
    return o;
}

__global__ void prepare_kernel(const int32_t* __restrict__ sym, const int32_t* __restrict__ idx, long n, int chunk, long n_chunks, Tables t, EncOps ops,
                               int32_t* __restrict__ err) {
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) {
        const long c = o / chunk, i = o - c * chunk;
        const long dst = i * n_chunks + c;
        const int32_t ci = idx[o];
        if (ci < 0 || ci >= t.n_cdfs) { atomicExch(err, 1); ops.fs[dst] = 1u << 16; ops.meta[dst] = 0xFF00; ops.rcp[dst] = ~0ull; continue; }
        const int32_t* cdf = t.cdf + (long)ci * t.stride;
        const int32_t max_value = t.sizes[ci] - 2;
        int32_t value = sym[o] - t.offsets[ci];
        uint32_t raw = 0;
        if (value < 0) { raw = (uint32_t)(-2 * value - 1); value = max_value; }
        else if (value >= max_value) { raw = (uint32_t)(2 * (value - max_value)); value = max_value; }
        uint32_t nb = 0xFF;
        if (value == max_value) {
            nb = 0;
            while (nb < 8 && (raw >> (nb * kBypassBits)) != 0) ++nb;
        }
        const uint32_t start = (uint32_t)cdf[value], freq = (uint32_t)(cdf[value + 1] - cdf[value]);
        const Rcp r = make_rcp(freq);
        ops.rcp[dst] = r.rcp;
        ops.fs[dst] = start | (freq << 16);
        ops.raw[dst] = raw;
        ops.meta[dst] = (uint16_t)(r.shift | (nb << 8));
    }
}

// One lane per chunk: the serial part.  words: [n_chunks][cap] scratch (each chunk's stream ends at its region's end); nbytes[chunk] = stream
// length, -1 on error.  Symbols are coded LAST TO FIRST (see encode_range for the order of an escape's pieces).
__global__ void encode_serial_kernel(long n, int chunk, long n_chunks, EncOps ops, uint32_t* words, int cap, int32_t* nbytes, const int32_t* err) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const long i0 = c * chunk, len = ((i0 + chunk) < n ? (i0 + chunk) : n) - i0;
    uint32_t* begin = words + c * cap;
    uint32_t* ptr = begin + cap;
    uint64_t x = kL;
    bool ok = *err == 0;
    // batches of kB symbols: their operations are loaded together (the addresses do not depend on the state), so one memory latency is
    // paid per batch instead of per symbol (one symbol per iteration measured ~1000 cycles per symbol: all of it load latency)
    constexpr int kB = 16;
    for (long hi = len; hi > 0 && ok; hi -= kB) {
        uint32_t fs[kB], meta[kB];
        uint64_t rcp[kB];
#pragma unroll
        for (int k = 0; k < kB; ++k) {
            const long i = hi - 1 - k;
            const long src = (i >= 0 ? i : 0) * n_chunks + c;
            fs[k] = ops.fs[src]; meta[k] = ops.meta[src]; rcp[k] = ops.rcp[src];
        }
#pragma unroll
        for (int k = 0; k < kB; ++k) {
            const long i = hi - 1 - k;
            if (i < 0 || !ok) continue;
            Rcp r; r.rcp = rcp[k]; r.shift = meta[k] & 0xFF;
            const uint32_t nb = meta[k] >> 8;
            if (nb != 0xFF) {
                const uint32_t raw = ops.raw[i * n_chunks + c];
                for (int j = (int)nb - 1; j >= 0; --j) ok = ok && put_bits(x, ptr, begin, (raw >> (j * kBypassBits)) & kMaxBypass);
                ok = ok && put_bits(x, ptr, begin, nb % kMaxBypass);
                for (uint32_t q = 0; q < nb / kMaxBypass; ++q) ok = ok && put_bits(x, ptr, begin, kMaxBypass);
            }
            ok = ok && put_rcp(x, ptr, begin, fs[k] & 0xFFFFu, fs[k] >> 16, r);
        }
    }
    if (ok && ptr - begin >= 2) {
        ptr -= 2;
        ptr[0] = (uint32_t)x;
        ptr[1] = (uint32_t)(x >> 32);
        nbytes[c] = (int32_t)((begin + cap - ptr) * 4);
    } else nbytes[c] = -1;
}

// gather the chunk streams into one contiguous buffer: block = chunk
__global__ void compact_kernel(const uint32_t* words, int cap, const int32_t* nbytes, const long long* offsets /* exclusive, bytes */,
                               uint8_t* out) {
    const long c = blockIdx.x;
    const int nw = nbytes[c] / 4;
    const uint32_t* src = words + c * cap + (cap - nw);
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + offsets[c]);
    for (int i = threadIdx.x; i < nw; i += blockDim.x) dst[i] = src[i];
}

__global__ void decode_chunks_kernel(const uint8_t* stream, long long stream_bytes, const long long* offsets, const int32_t* idx, long n, int chunk,
                                     Tables t, int32_t* out, int32_t* err, long n_chunks) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const long i0 = c * chunk, i1 = (i0 + chunk) < n ? (i0 + chunk) : n;
    const long long b0 = offsets[c], b1 = c + 1 < n_chunks ? offsets[c + 1] : stream_bytes;
    if (b0 < 0 || b1 > stream_bytes || b1 - b0 < 8 || ((b1 - b0) & 3) || (b0 & 3)) { atomicExch(err, 2); return; }   // a stream is >= one flushed state, whole words
    const uint32_t* w = reinterpret_cast<const uint32_t*>(stream + b0);
    DecState st;
    st.x = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    st.pos = 2; st.n_words = (long)((b1 - b0) >> 2); st.bad = false;
    if (!decode_range(w, st, idx, i0, i1, t, out)) atomicExch(err, st.bad ? 2 : 1);
}

// Decoding with the tables in LDS.  A lane's symbol costs one dependent chain: CDF index -> row size -> binary search over the row (8-12 probes) ->
// (start, freq) -> state -> (every other symbol) the next 32-bit word of ITS stream.  From global memory that was 1.8 ms per 2 048-symbol call
// (10.8 of decompress()'s 30 ms per 4K frame).  Here (a) every block first copies sizes, offsets and ALL rows, packed back to back as 16-bit entries
// (a quantised CDF fits: only a row's last entry is 2^16, and the search never needs to read it), into LDS -- 54 KB for the 64 Gaussian rows -- so
// the probes are LDS reads; (b) a lane always holds the NEXT word of its stream in a register, loaded when the previous one is consumed; (c) the CDF
// indexes of the next 8 symbols are loaded ahead (they do not depend on the state) and the symbols leave 16 bytes at a time.  Tables that do not fit
// the LDS given to the kernel fall back to probes in global memory.  Same arithmetic and error codes as decode_range.
// Measured: 1.80 -> 1.50 ms per call from (a), 1.42 with 3 instead of 64 chunks per wave.  What is left is ONE lane's dependent chain per symbol
// (row size/start -> 12 probes -> start/freq -> 64-bit multiply -> word: ~14 LDS round trips of ~80 cycles): a 2 048-symbol chunk per lane makes only
// 1 350 chains for a whole 4K latent.  (b) and (c) changed nothing; a 256-bucket table per row in front of the search (1.73 ms with 64 lanes per wave,
// 2.5 ms with 3: its construction per block) and an 8-ary search with 8 independent probes per level (1.69 ms) were slower -- fewer dependent probes,
// more instructions per level.  More chains means smaller chunks: 512-symbol chunks would be 4x faster and cost +20 % bytes (8 B flush + 4 B size per
// chunk), which is why 2 048 stays.
constexpr int kDecLdsEntries = 32 * 1024;                        // 16-bit CDF entries the kernel's dynamic LDS holds (64 KB: two blocks per CU)
__global__ __launch_bounds__(64) void decode_chunks_lds_kernel(const uint8_t* stream, long long stream_bytes, const long long* offsets, const int32_t* idx,
                                                               long n, int chunk, Tables t, int32_t* out, int32_t* err, long n_chunks, int lanes) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    int32_t* s_size = reinterpret_cast<int32_t*>(dsm);
    int32_t* s_off = s_size + t.n_cdfs;
    int32_t* s_start = s_off + t.n_cdfs;
    uint16_t* s_cdf = reinterpret_cast<uint16_t*>(s_start + t.n_cdfs + 2);
    __shared__ int s_total;
    const int tid = threadIdx.x;
    for (int r = tid; r < t.n_cdfs; r += 64) { s_size[r] = t.sizes[r]; s_off[r] = t.offsets[r]; }
    __syncthreads();
    if (tid == 0) {
        int acc = 0; bool ok = true;
        for (int r = 0; r < t.n_cdfs; ++r) {
            const int sz = s_size[r];
            if (sz < 2 || sz > t.stride) ok = false;
            s_start[r] = acc; acc += ok ? sz : 0;
            if (acc > kDecLdsEntries) ok = false;
        }
        s_total = ok ? acc : -1;
    }
    __syncthreads();
    // 16-bit LDS entries assume a well-formed row: row[0] == 0 and only the LAST entry equal to 65536 (stored as 0, never compared).  A row that
    // breaks this would make the LDS search differ from decode_range's: such tables take the global-memory path (s_total = -1) instead.
    __shared__ int s_bad;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    if (s_total >= 0)
        for (int r = 0; r < t.n_cdfs; ++r) {
            const int sz = s_size[r], st0 = s_start[r];
            const int32_t* row = t.cdf + (long)r * t.stride;
            for (int j = tid; j < sz; j += 64) {
                const int32_t v = row[j];
                if ((j == 0 && v != 0) || v < 0 || (j < sz - 1 ? v >= 65536 : v > 65536)) s_bad = 1;     // benign race: every writer stores 1
                s_cdf[st0 + j] = (uint16_t)v;
            }
        }
    __syncthreads();
    const bool in_lds = s_total >= 0 && s_bad == 0;

    // only `lanes` lanes of the wave decode (the others helped to fill the LDS)
    const long c = (long)blockIdx.x * lanes + tid;
    if (tid >= lanes || c >= n_chunks) return;
    const long i0 = c * chunk, i1 = (i0 + chunk) < n ? (i0 + chunk) : n;
    const long long b0 = offsets[c], b1 = c + 1 < n_chunks ? offsets[c + 1] : stream_bytes;
    if (b0 < 0 || b1 > stream_bytes || b1 - b0 < 8 || ((b1 - b0) & 3) || (b0 & 3)) { atomicExch(err, 2); return; }
    const uint32_t* w = reinterpret_cast<const uint32_t*>(stream + b0);
    const int n_words = (int)((b1 - b0) >> 2);
    int pos = 2;                                                   // next word to consume
    uint64_t x = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    uint32_t w_ahead = pos < n_words ? w[pos] : 0u;               // ... which is already on its way: loaded when the previous one was consumed
    bool bad = false;
    auto next_w = [&]() -> uint32_t {
        if (pos >= n_words) { bad = true; return 0u; }
        const uint32_t v = w_ahead;
        ++pos;
        w_ahead = pos < n_words ? w[pos] : 0u;
        return v;
    };
    auto bits = [&]() -> uint32_t {
        const uint32_t val = (uint32_t)(x & kMaxBypass);
        x >>= kBypassBits;
        if (x < kL) x = (x << 32) | next_w();
        return val;
    };
    const bool vec_out = (chunk & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    int32_t pend[4];
    constexpr int kAhead = 8;
    for (long ib = i0; ib < i1; ib += kAhead) {
        int32_t cis[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) cis[k] = ib + k < i1 ? idx[ib + k] : 0;
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            const long i = ib + k;
            if (i >= i1) break;
            const int32_t ci = cis[k];
            if (ci < 0 || ci >= t.n_cdfs) { atomicExch(err, 1); return; }
            const int32_t size = s_size[ci], max_value = size - 2;
            const uint32_t cum = (uint32_t)(x & ((1u << kPrec) - 1));
            uint32_t start, freq;
            int32_t s;
            if (in_lds) {
                const uint16_t* row = s_cdf + s_start[ci];
                int lo = 0, hi = size - 1;                           // the last entry (2^16) is above every cum
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if ((uint32_t)row[mid] <= cum) lo = mid + 1; else hi = mid;
                }
                s = lo - 1;
                start = row[s];
                freq = (s + 1 == size - 1 ? (1u << kPrec) : (uint32_t)row[s + 1]) - start;
            } else {
                const int32_t* row = t.cdf + (long)ci * t.stride;
                int lo = 0, hi = size;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if ((uint32_t)row[mid] <= cum) lo = mid + 1; else hi = mid;
                }
                s = lo - 1;
                start = (uint32_t)row[s]; freq = (uint32_t)(row[s + 1] - row[s]);
            }
            x = (uint64_t)freq * (x >> kPrec) + (x & ((1u << kPrec) - 1)) - start;
            if (x < kL) x = (x << 32) | next_w();
            int32_t value = s;
            if (value == max_value) {
                int32_t val = (int32_t)bits(), n_bypass = val;
                while (val == (int32_t)kMaxBypass && !bad && n_bypass <= 8) { val = (int32_t)bits(); n_bypass += val; }
                if (n_bypass > 8) bad = true;
                if (bad) { atomicExch(err, 2); return; }
                uint32_t raw = 0;
                for (int32_t j = 0; j < n_bypass; ++j) raw |= bits() << (j * kBypassBits);
                value = (int32_t)(raw >> 1);
                if (raw & 1) value = -value - 1; else value += max_value;
            }
            if (bad) { atomicExch(err, 2); return; }
            const int32_t sym = value + s_off[ci];
            if (vec_out) {
                pend[k & 3] = sym;
                if ((k & 3) == 3) *reinterpret_cast<int4*>(out + i - 3) = make_int4(pend[0], pend[1], pend[2], pend[3]);
                else if (i + 1 == i1) for (int e = 0; e <= (k & 3); ++e) out[i - (k & 3) + e] = pend[e];
            } else out[i] = sym;
        }
    }
}

// ---- symbol preparation ------------------------------------------------------------------------------------------------------------
// GaussianConditional: symbols = round(y - mu) (round-half-even, as torch.round), index = n_levels - 1 - #{ table[j] >= max(scale, bound),
// j < n_levels - 1 } (CompressAI build_indexes), y_hat = symbols + mu.  Inputs NHWC (B, hw, C); int outputs in (b, c, hw) order.
// y == NULL: indexes only (the decoder's side).
template <typename T>
__global__ void gc_symbols_kernel(const T* __restrict__ y, const T* __restrict__ mu, const T* __restrict__ scale, int batch, long hw, int C,
                                  const float* __restrict__ table, int n_levels, float scale_bound, int32_t* __restrict__ symbols,
                                  int32_t* __restrict__ indexes, T* __restrict__ y_hat) {
    const long total = (long)batch * C * hw;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const long p = o % hw;
        const int c = (int)((o / hw) % C);
        const long b = o / (hw * C);
        const long i = (b * hw + p) * C + c;
        float s = to_f32(scale[i]);
        s = s > scale_bound ? s : scale_bound;
        int ix = n_levels - 1;
        for (int j = 0; j < n_levels - 1; ++j) ix -= (s <= table[j]) ? 1 : 0;
        indexes[o] = ix;
        if (y) {
            const float m = to_f32(mu[i]);
            const float q = rintf(to_f32(y[i]) - m);
            symbols[o] = (int32_t)q;
            y_hat[i] = from_f32<T>(q + m);
        }
    }
}

// y_hat = symbols + mu (GaussianConditional.dequantize); symbols (b, c, hw), mu / y_hat NHWC
template <typename T>
__global__ void gc_dequant_kernel(const int32_t* __restrict__ symbols, const T* __restrict__ mu, int batch, long hw, int C, T* __restrict__ y_hat) {
    const long total = (long)batch * C * hw;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const long p = o % hw;
        const int c = (int)((o / hw) % C);
        const long b = o / (hw * C);
        const long i = (b * hw + p) * C + c;
        y_hat[i] = from_f32<T>((float)symbols[o] + to_f32(mu[i]));
    }
}

// EntropyBottleneck: symbols = round(z - median[c]), index = c, z_hat = symbols + median[c]; encode = 1: from z; 0: z_hat from symbols
template <typename T>
__global__ void eb_symbols_kernel(const T* __restrict__ z, const float* __restrict__ med, int batch, long hw, int C, int encode,
                                  int32_t* __restrict__ symbols, int32_t* __restrict__ indexes, T* __restrict__ z_hat) {
    const long total = (long)batch * C * hw;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const long p = o % hw;
        const int c = (int)((o / hw) % C);
        const long b = o / (hw * C);
        const long i = (b * hw + p) * C + c;
        indexes[o] = c;
        float q;
        if (encode) { q = rintf(to_f32(z[i]) - med[c]); symbols[o] = (int32_t)q; }
        else q = (float)symbols[o];
        z_hat[i] = from_f32<T>(q + med[c]);
    }
}

static inline int grid1d(long n) {
    long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g));
}

}  // namespace ans
}  // namespace rc

namespace rc { int g_dec_lds = 1; }     // rc_debug_set("dec_lds", v): 1 (default) rc_rans_decode_chunks keeps the CDF rows in LDS; 0: probes in global memory
using namespace rc;
using namespace rc::ans;

extern "C" {

// compressai._CXX.pmf_to_quantized_cdf (host): n probabilities -> n + 1 cumulative frequencies summing to 2^precision, none zero
int rc_pmf_to_quantized_cdf(const float* pmf, int n, int precision, int32_t* cdf) {
    RC_REQUIRE(pmf && cdf && n >= 1 && precision >= 1 && precision <= 24, "rc_pmf_to_quantized_cdf: bad arguments");
    for (int i = 0; i < n; ++i) RC_REQUIRE(pmf[i] >= 0.f && isfinite(pmf[i]), "rc_pmf_to_quantized_cdf: pmf must be finite and non-negative");
    uint32_t* c = reinterpret_cast<uint32_t*>(cdf);
    c[0] = 0;
    uint64_t total = 0;
    for (int i = 0; i < n; ++i) { c[i + 1] = (uint32_t)llroundf(pmf[i] * (float)(1 << precision)); total += c[i + 1]; }
    RC_REQUIRE(total != 0, "rc_pmf_to_quantized_cdf: pmf sums to zero");
    uint32_t run = 0;
    for (int i = 0; i <= n; ++i) { run += (uint32_t)((((uint64_t)1 << precision) * (uint64_t)c[i]) / total); c[i] = run; }
    c[n] = 1u << precision;
    for (int i = 0; i < n; ++i) {
        if (c[i] != c[i + 1]) continue;
        uint32_t best = ~0u;
        int steal = -1;
        for (int j = 0; j < n; ++j) {
            const uint32_t f = c[j + 1] - c[j];
            if (f > 1 && f < best) { best = f; steal = j; }
        }
        RC_REQUIRE(steal >= 0, "rc_pmf_to_quantized_cdf: no frequency left to redistribute");
        if (steal < i) { for (int j = steal + 1; j <= i; ++j) c[j]--; }
        else { for (int j = i + 1; j <= steal; ++j) c[j]++; }
    }
    return RC_OK;
}

int rc_gc_symbols(const void* d_y, const void* d_mu, const void* d_scale, int dtype, int batch, long long hw, int channels,
                  const float* d_scale_table, int n_levels, float scale_bound, int32_t* d_symbols, int32_t* d_indexes, void* d_y_hat,
                  void* stream) {
    RC_REQUIRE(d_scale && d_scale_table && d_indexes && n_levels >= 1, "rc_gc_symbols: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_gc_symbols: bad dtype");
    RC_REQUIRE(batch >= 1 && hw >= 1 && channels >= 1, "rc_gc_symbols: bad shape");
    if (d_y) RC_REQUIRE(d_mu && d_symbols && d_y_hat, "rc_gc_symbols: encoding needs mu, symbols and y_hat");
    const long total = (long)batch * channels * hw;
    if (dtype == RC_F32)
        hipLaunchKernelGGL(gc_symbols_kernel<float>, dim3(grid1d(total)), dim3(256), 0, as_stream(stream), static_cast<const float*>(d_y),
                           static_cast<const float*>(d_mu), static_cast<const float*>(d_scale), batch, (long)hw, channels, d_scale_table, n_levels,
                           scale_bound, d_symbols, d_indexes, static_cast<float*>(d_y_hat));
    else
        hipLaunchKernelGGL(gc_symbols_kernel<bf16_t>, dim3(grid1d(total)), dim3(256), 0, as_stream(stream), static_cast<const bf16_t*>(d_y),
                           static_cast<const bf16_t*>(d_mu), static_cast<const bf16_t*>(d_scale), batch, (long)hw, channels, d_scale_table,
                           n_levels, scale_bound, d_symbols, d_indexes, static_cast<bf16_t*>(d_y_hat));
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_gc_dequantize(const int32_t* d_symbols, const void* d_mu, int dtype, int batch, long long hw, int channels, void* d_y_hat, void* stream) {
    RC_REQUIRE(d_symbols && d_mu && d_y_hat, "rc_gc_dequantize: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_gc_dequantize: bad dtype");
    const long total = (long)batch * channels * hw;
    if (dtype == RC_F32)
        hipLaunchKernelGGL(gc_dequant_kernel<float>, dim3(grid1d(total)), dim3(256), 0, as_stream(stream), d_symbols, static_cast<const float*>(d_mu),
                           batch, (long)hw, channels, static_cast<float*>(d_y_hat));
    else
        hipLaunchKernelGGL(gc_dequant_kernel<bf16_t>, dim3(grid1d(total)), dim3(256), 0, as_stream(stream), d_symbols,
                           static_cast<const bf16_t*>(d_mu), batch, (long)hw, channels, static_cast<bf16_t*>(d_y_hat));
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_eb_symbols(const void* d_z, const float* d_medians, int dtype, int batch, long long hw, int channels, int encode, int32_t* d_symbols,
                  int32_t* d_indexes, void* d_z_hat, void* stream) {
    RC_REQUIRE(d_medians && d_symbols && d_indexes && d_z_hat && (d_z || !encode), "rc_eb_symbols: null pointer");
    RC_REQUIRE(dtype == RC_F32 || dtype == RC_BF16, "rc_eb_symbols: bad dtype");
    const long total = (long)batch * channels * hw;
    if (dtype == RC_F32)
        hipLaunchKernelGGL(eb_symbols_kernel<float>, dim3(grid1d(total)), dim3(256), 0, as_stream(stream), static_cast<const float*>(d_z), d_medians,
                           batch, (long)hw, channels, encode, d_symbols, d_indexes, static_cast<float*>(d_z_hat));
    else
        hipLaunchKernelGGL(eb_symbols_kernel<bf16_t>, dim3(grid1d(total)), dim3(256), 0, as_stream(stream), static_cast<const bf16_t*>(d_z),
                           d_medians, batch, (long)hw, channels, encode, d_symbols, d_indexes, static_cast<bf16_t*>(d_z_hat));
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

// words per chunk of scratch that can never overflow: an escaped symbol costs at most 16 + 4 + 8 x 4 = 52 bits < 2 words
int rc_rans_chunk_words(int chunk) { return chunk > 0 ? 2 * chunk + 8 : 0; }

// scratch of rc_rans_encode_chunks: the prepared operations (18 bytes per symbol slot, chunk-padded) + an error word
size_t rc_rans_encode_scratch_bytes(long long n, int chunk) {
    if (n < 1 || chunk < 1) return 0;
    const long long n_pad = (n + chunk - 1) / chunk * chunk;
    return (size_t)n_pad * 18 + 16;
}

int rc_rans_encode_chunks(const int32_t* d_symbols, const int32_t* d_indexes, long long n, int chunk, const int32_t* d_cdf, int cdf_stride,
                          int n_cdfs, const int32_t* d_cdf_sizes, const int32_t* d_offsets, uint32_t* d_words, int32_t* d_nbytes, void* d_scratch,
                          void* stream) {
    RC_REQUIRE(d_symbols && d_indexes && d_cdf && d_cdf_sizes && d_offsets && d_words && d_nbytes && d_scratch, "rc_rans_encode_chunks: null pointer");
    RC_REQUIRE(n >= 1 && chunk >= 1 && cdf_stride >= 2 && n_cdfs >= 1, "rc_rans_encode_chunks: bad shape");
    RC_REQUIRE(reinterpret_cast<uintptr_t>(d_scratch) % 8 == 0, "rc_rans_encode_chunks: scratch must be 8-byte aligned");
    const long n_chunks = (n + chunk - 1) / chunk;
    const long n_pad = n_chunks * chunk;
    const Tables t{d_cdf, cdf_stride, n_cdfs, d_cdf_sizes, d_offsets};
    const EncOps ops = enc_ops(d_scratch, n_pad);
    int32_t* d_err = reinterpret_cast<int32_t*>(static_cast<char*>(d_scratch) + (size_t)n_pad * 18 + ((8 - ((size_t)n_pad * 18) % 8) % 8));
    RC_HIP_CHECK(hipMemsetAsync(d_err, 0, 4, as_stream(stream)));
    hipLaunchKernelGGL(prepare_kernel, dim3(grid1d(n)), dim3(256), 0, as_stream(stream), d_symbols, d_indexes, (long)n, chunk, n_chunks, t, ops, d_err);
    hipLaunchKernelGGL(encode_serial_kernel, dim3((unsigned)((n_chunks + 63) / 64)), dim3(64), 0, as_stream(stream), (long)n, chunk, n_chunks, ops,
                       d_words, rc_rans_chunk_words(chunk), d_nbytes, d_err);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

// self-test of the division-free put against the dividing one on random (state, start, freq): returns the number of mismatches
long long rc_debug_rans_rcp_selftest(long long trials, unsigned long long seed) {
    long long bad = 0;
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    uint32_t buf[4];
    for (long long i = 0; i < trials; ++i) {
        uint32_t freq = 1 + (uint32_t)(next() % 65535u);
        if (i % 7 == 0) freq = 1u << (next() % 16);                       // powers of two and tiny frequencies are the edge cases
        if (i % 11 == 0) freq = 1 + (uint32_t)(next() % 3);
        const uint32_t start = (uint32_t)(next() % (65536u - freq + 1));
        uint64_t x = kL + next() % ((kL << 32) - kL);                       // any reachable state in [2^31, 2^63)
        if (i % 5 == 0) x = (i & 1) ? kL : (kL << 32) - 1;
        uint64_t xa = x, xb = x;
        uint32_t *pa = buf + 2, *pb = buf + 4;
        uint32_t keep[4] = {0, 0, 0, 0};
        put(xa, pa, buf, start, freq);
        keep[0] = buf[1]; const long na = (buf + 2) - pa;
        put_rcp(xb, pb, buf + 2, start, freq, make_rcp(freq));
        const long nb = (buf + 4) - pb;
        if (xa != xb || na != nb || (na == 1 && keep[0] != buf[3])) ++bad;
    }
    return bad;
}

int rc_rans_compact(const uint32_t* d_words, int chunk, const int32_t* d_nbytes, const long long* d_offsets, long long n_chunks, void* d_out,
                    void* stream) {
    RC_REQUIRE(d_words && d_nbytes && d_offsets && d_out && n_chunks >= 1, "rc_rans_compact: bad arguments");
    hipLaunchKernelGGL(compact_kernel, dim3((unsigned)n_chunks), dim3(64), 0, as_stream(stream), d_words, rc_rans_chunk_words(chunk), d_nbytes,
                       d_offsets, static_cast<uint8_t*>(d_out));
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

int rc_rans_decode_chunks(const void* d_stream, long long stream_bytes, const long long* d_offsets, const int32_t* d_indexes, long long n, int chunk,
                          const int32_t* d_cdf, int cdf_stride, int n_cdfs, const int32_t* d_cdf_sizes, const int32_t* d_cdf_offsets, int32_t* d_symbols,
                          int32_t* d_err, void* stream) {
    RC_REQUIRE(d_stream && d_offsets && d_indexes && d_cdf && d_cdf_sizes && d_cdf_offsets && d_symbols && d_err, "rc_rans_decode_chunks: null pointer");
    RC_REQUIRE(n >= 1 && chunk >= 1 && stream_bytes >= 8 && stream_bytes < (1ll << 31), "rc_rans_decode_chunks: bad shape (a container is < 2 GiB)");
    const long n_chunks = (n + chunk - 1) / chunk;
    const Tables t{d_cdf, cdf_stride, n_cdfs, d_cdf_sizes, d_cdf_offsets};
    const size_t lds = (size_t)(3 * n_cdfs + 2) * 4 + (size_t)kDecLdsEntries * 2;
    if (g_dec_lds && n_cdfs >= 1 && n_cdfs <= 1024 && cdf_stride >= 2) {       // tables in LDS (falls back to global probes inside the kernel if they do not fit)
        static PerDeviceFlag attr;
        if (!attr.test_and_set())
            RC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_chunks_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        // chunks per wave: as few as keeps every CU busy with two blocks -- a wave executes every branch any of its lanes takes, and the chunk count
        // (1 350 for a 4K latent) leaves most of the chip idle either way: 1.50 -> 1.42 ms
        long lanes = (n_chunks + 2 * device_cu_count() - 1) / (2 * device_cu_count());
        lanes = lanes < 1 ? 1 : (lanes > 64 ? 64 : lanes);
        hipLaunchKernelGGL(decode_chunks_lds_kernel, dim3((unsigned)((n_chunks + lanes - 1) / lanes)), dim3(64), lds, as_stream(stream), static_cast<const uint8_t*>(d_stream),
                           stream_bytes, d_offsets, d_indexes, (long)n, chunk, t, d_symbols, d_err, n_chunks, (int)lanes);
    } else
        hipLaunchKernelGGL(decode_chunks_kernel, dim3((unsigned)((n_chunks + 63) / 64)), dim3(64), 0, as_stream(stream), static_cast<const uint8_t*>(d_stream),
                           stream_bytes, d_offsets, d_indexes, (long)n, chunk, t, d_symbols, d_err, n_chunks);
    RC_HIP_CHECK(hipGetLastError());
    return RC_OK;
}

// ---- host forms: one stream in CompressAI's layout (format="compressai") -----------------------------------------------------------
long long rc_rans_encode_host(const int32_t* symbols, const int32_t* indexes, long long n, const int32_t* cdf, int cdf_stride, int n_cdfs,
                              const int32_t* cdf_sizes, const int32_t* offsets, void* out, long long out_cap) {
    if (!symbols || !indexes || !cdf || !cdf_sizes || !offsets || !out || n < 0) { rc::fail(RC_ERR_INVALID, "rc_rans_encode_host: bad arguments"); return -1; }
    const long cap = 2 * n + 8;
    uint32_t* words = static_cast<uint32_t*>(malloc((size_t)cap * 4));
    if (!words) { rc::fail(RC_ERR_INVALID, "rc_rans_encode_host: out of memory"); return -1; }
    const Tables t{cdf, cdf_stride, n_cdfs, cdf_sizes, offsets};
    uint32_t* p = encode_range(symbols, indexes, 0, n, t, words, words + cap);
    long long nbytes = -1;
    if (p) {
        nbytes = (long long)(words + cap - p) * 4;
        if (nbytes > out_cap) { rc::fail(RC_ERR_INVALID, "rc_rans_encode_host: output buffer too small"); nbytes = -1; }
        else memcpy(out, p, (size_t)nbytes);
    } else rc::fail(RC_ERR_INVALID, "rc_rans_encode_host: CDF index out of range");
    free(words);
    return nbytes;
}

int rc_rans_decode_host(const void* stream_bytes, long long n_bytes, unsigned long long* state /* [2]: x, next word; x == 0: start of stream */,
                        const int32_t* indexes, long long n, const int32_t* cdf, int cdf_stride, int n_cdfs, const int32_t* cdf_sizes,
                        const int32_t* offsets, int32_t* out) {
    RC_REQUIRE(stream_bytes && state && indexes && cdf && cdf_sizes && offsets && out && n >= 0, "rc_rans_decode_host: bad arguments");
    RC_REQUIRE(n_bytes >= 8 && n_bytes % 4 == 0, "rc_rans_decode_host: a stream is a whole number of 32-bit words, at least the flushed state");
    const uint32_t* w = static_cast<const uint32_t*>(stream_bytes);
    DecState st;
    st.n_words = (long)(n_bytes / 4); st.bad = false;
    if (state[0] == 0) { st.x = (uint64_t)w[0] | ((uint64_t)w[1] << 32); st.pos = 2; }
    else { st.x = state[0]; st.pos = (long)state[1]; RC_REQUIRE(st.pos >= 2 && st.pos <= st.n_words, "rc_rans_decode_host: bad decoder state"); }
    const Tables t{cdf, cdf_stride, n_cdfs, cdf_sizes, offsets};
    const bool ok = decode_range(w, st, indexes, 0, n, t, out);
    RC_REQUIRE(ok || !st.bad, "rc_rans_decode_host: truncated or corrupt stream (read past its end / escape longer than 8 nibbles)");
    RC_REQUIRE(ok, "rc_rans_decode_host: CDF index out of range");
    state[0] = st.x; state[1] = (unsigned long long)st.pos;
    return RC_OK;
}

}  // extern "C"
