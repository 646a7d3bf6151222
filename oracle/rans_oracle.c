/* rANS entropy coder + CDF quantiser: CPU oracle in plain C.            *** TEST INFRASTRUCTURE ***
 *
 * Restates the arithmetic that the reference's compress() / decompress() (models/tcm.py:511-570, 592-637;
 * models/raw2bit.py:1876-1944, 1961-2027) delegate to CompressAI: `compressai.ans.BufferedRansEncoder.encode_with_indexes /
 * flush`, `RansDecoder.set_stream / decode_stream` and `compressai._CXX.pmf_to_quantized_cdf`.  CompressAI is a PyPI dependency
 * that is NOT in /root/reference (no version is pinned upstream), so this is written from its published algorithm
 * (a 64-bit-state rANS after F. Giesen's ryg_rans `rans64.h`: 32-bit renormalisation words, lower bound 2^31, 16-bit probability
 * resolution, out-of-range symbols escaped through 4-bit "bypass" symbols) and is PARITY-UNPINNED against the real package: the
 * reference holds no golden bitstream.  What it pins is the build's own GPU coder (bit-exact, tests/test_bitstream.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's library (oracle/_build/).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RANS64_L (1ull << 31)
#define PRECISION 16
#define BYPASS_PRECISION 4
#define MAX_BYPASS_VAL ((1 << BYPASS_PRECISION) - 1)

/* ---- pmf -> quantised CDF (sum 2^precision, every symbol of the pmf keeps a non-zero frequency) ------------------------- */
/* returns n + 1 entries in cdf; < 0 on error */
int ro_pmf_to_quantized_cdf(const float* pmf, int n, int precision, uint32_t* cdf) {
    if (n < 1) return -1;
    for (int i = 0; i < n; ++i)
        if (!(pmf[i] >= 0.f) || !isfinite(pmf[i])) return -2;
    cdf[0] = 0;
    for (int i = 0; i < n; ++i) cdf[i + 1] = (uint32_t)llroundf(pmf[i] * (float)(1 << precision));
    uint64_t total = 0;
    for (int i = 0; i <= n; ++i) total += cdf[i];
    if (total == 0) return -3;
    for (int i = 0; i <= n; ++i) cdf[i] = (uint32_t)((((uint64_t)1 << precision) * (uint64_t)cdf[i]) / total);
    for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];                 /* partial sums */
    cdf[n] = 1u << precision;
    for (int i = 0; i < n; ++i) {
        if (cdf[i] == cdf[i + 1]) {                                    /* a zero-frequency symbol: steal from the rarest donor */
            uint32_t best_freq = ~0u;
            int best_steal = -1;
            for (int j = 0; j < n; ++j) {
                const uint32_t freq = cdf[j + 1] - cdf[j];
                if (freq > 1 && freq < best_freq) { best_freq = freq; best_steal = j; }
            }
            if (best_steal < 0) return -4;
            if (best_steal < i) { for (int j = best_steal + 1; j <= i; ++j) cdf[j]--; }
            else { for (int j = i + 1; j <= best_steal; ++j) cdf[j]++; }
        }
    }
    return n + 1;
}

/* ---- encoder: symbols are buffered, then coded LAST TO FIRST into a word buffer that fills from the back ------------------- */
typedef struct { uint16_t start, range; uint8_t bypass; } ro_sym;

static void enc_put(uint64_t* r, uint32_t** pptr, uint32_t start, uint32_t freq, uint32_t scale_bits) {
    uint64_t x = *r;
    const uint64_t x_max = ((RANS64_L >> scale_bits) << 32) * freq;
    if (x >= x_max) { *pptr -= 1; **pptr = (uint32_t)x; x >>= 32; }
    *r = ((x / freq) << scale_bits) + (x % freq) + start;
}

static void enc_put_bits(uint64_t* r, uint32_t** pptr, uint32_t val, uint32_t nbits) {
    uint64_t x = *r;
    const uint32_t freq = 1u << (16 - nbits);
    const uint64_t x_max = ((RANS64_L >> 16) << 32) * freq;
    if (x >= x_max) { *pptr -= 1; **pptr = (uint32_t)x; x >>= 32; }
    *r = (x << nbits) | val;
}

/* returns the stream length in bytes (written to out), or < 0: -1 bad index, -2 out too small, -3 allocation */
long ro_encode_with_indexes(const int32_t* symbols, const int32_t* indexes, long n, const int32_t* cdfs, int cdf_stride, int n_cdfs,
                            const int32_t* cdf_sizes, const int32_t* offsets, uint8_t* out, long out_cap) {
    long cap = n + 16, cnt = 0;
    ro_sym* syms = (ro_sym*)malloc((size_t)cap * sizeof(ro_sym));
    if (!syms) return -3;
#define PUSH(S, R, B)                                                                      \
    do {                                                                                   \
        if (cnt == cap) { cap = cap * 2; syms = (ro_sym*)realloc(syms, (size_t)cap * sizeof(ro_sym)); if (!syms) return -3; } \
        syms[cnt].start = (uint16_t)(S); syms[cnt].range = (uint16_t)(R); syms[cnt].bypass = (B); ++cnt; \
    } while (0)
    for (long i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        if (ci < 0 || ci >= n_cdfs) { free(syms); return -1; }
        const int32_t* cdf = cdfs + (long)ci * cdf_stride;
        const int32_t max_value = cdf_sizes[ci] - 2;
        int32_t value = symbols[i] - offsets[ci];
        uint32_t raw_val = 0;
        if (value < 0) { raw_val = (uint32_t)(-2 * value - 1); value = max_value; }
        else if (value >= max_value) { raw_val = (uint32_t)(2 * (value - max_value)); value = max_value; }
        PUSH(cdf[value], cdf[value + 1] - cdf[value], 0);
        if (value == max_value) {                                       /* escape: length in base-15 digits, then 4-bit nibbles, low first */
            int32_t n_bypass = 0;
            while ((raw_val >> (n_bypass * BYPASS_PRECISION)) != 0) ++n_bypass;
            int32_t val = n_bypass;
            while (val >= MAX_BYPASS_VAL) { PUSH(MAX_BYPASS_VAL, MAX_BYPASS_VAL + 1, 1); val -= MAX_BYPASS_VAL; }
            PUSH(val, val + 1, 1);
            for (int32_t j = 0; j < n_bypass; ++j) {
                const int32_t v = (raw_val >> (j * BYPASS_PRECISION)) & MAX_BYPASS_VAL;
                PUSH(v, v + 1, 1);
            }
        }
    }
#undef PUSH
    uint32_t* words = (uint32_t*)malloc((size_t)(cnt + 2) * sizeof(uint32_t));
    if (!words) { free(syms); return -3; }
    uint32_t* ptr = words + cnt + 2;
    uint64_t rans = RANS64_L;
    for (long k = cnt - 1; k >= 0; --k) {
        if (!syms[k].bypass) enc_put(&rans, &ptr, syms[k].start, syms[k].range, PRECISION);
        else enc_put_bits(&rans, &ptr, syms[k].start, BYPASS_PRECISION);
    }
    ptr -= 2;
    ptr[0] = (uint32_t)(rans >> 0);
    ptr[1] = (uint32_t)(rans >> 32);
    const long nbytes = (long)((words + cnt + 2) - ptr) * 4;
    long ret = nbytes;
    if (nbytes > out_cap) ret = -2; else memcpy(out, ptr, (size_t)nbytes);
    free(words); free(syms);
    return ret;
}

/* ---- decoder: decode_stream semantics -- the state persists across calls so that a stream can be consumed slice by slice ------ */
typedef struct { uint64_t x; long pos; } ro_dec_state;      /* pos: index of the next unread 32-bit word */

void ro_dec_init(const uint8_t* stream, ro_dec_state* st) {
    const uint32_t* p = (const uint32_t*)stream;
    st->x = (uint64_t)p[0] | ((uint64_t)p[1] << 32);
    st->pos = 2;
}

static uint32_t dec_get_bits(ro_dec_state* st, const uint32_t* words, uint32_t nbits) {
    uint64_t x = st->x;
    const uint32_t val = (uint32_t)(x & ((1u << nbits) - 1));
    x >>= nbits;
    if (x < RANS64_L) { x = (x << 32) | words[st->pos]; st->pos += 1; }
    st->x = x;
    return val;
}

/* returns 0, or -1 on a bad index */
int ro_decode_stream(const uint8_t* stream, ro_dec_state* st, const int32_t* indexes, long n, const int32_t* cdfs, int cdf_stride, int n_cdfs,
                     const int32_t* cdf_sizes, const int32_t* offsets, int32_t* out) {
    const uint32_t* words = (const uint32_t*)stream;
    for (long i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        if (ci < 0 || ci >= n_cdfs) return -1;
        const int32_t* cdf = cdfs + (long)ci * cdf_stride;
        const int32_t size = cdf_sizes[ci], max_value = size - 2;
        const uint32_t cum = (uint32_t)(st->x & ((1u << PRECISION) - 1));
        int32_t s = 0;
        while (s < size && (uint32_t)cdf[s] <= cum) ++s;                /* first entry > cum */
        s -= 1;
        {   /* advance */
            const uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
            uint64_t x = st->x;
            x = (uint64_t)freq * (x >> PRECISION) + (x & ((1u << PRECISION) - 1)) - start;
            if (x < RANS64_L) { x = (x << 32) | words[st->pos]; st->pos += 1; }
            st->x = x;
        }
        int32_t value = s;
        if (value == max_value) {
            int32_t val = (int32_t)dec_get_bits(st, words, BYPASS_PRECISION);
            int32_t n_bypass = val;
            while (val == MAX_BYPASS_VAL) { val = (int32_t)dec_get_bits(st, words, BYPASS_PRECISION); n_bypass += val; }
            uint32_t raw_val = 0;
            for (int32_t j = 0; j < n_bypass; ++j) raw_val |= dec_get_bits(st, words, BYPASS_PRECISION) << (j * BYPASS_PRECISION);
            value = (int32_t)(raw_val >> 1);
            if (raw_val & 1) value = -value - 1; else value += max_value;
        }
        out[i] = value + offsets[ci];
    }
    return 0;
}
