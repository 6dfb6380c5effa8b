// tic_entropy.cpp - host entropy stage of the codec (product code; independent of oracle/).
//
// Replaces the per-block Python loops of compress() (codec.py:133-164 of the reference): DC DPCM
// (codec.py:34-35), run-length coding (huffman.py:12-33), Huffman symbol emission (huffman.py:41-63) into an
// MSB-first bit stream that is zero-padded to a byte (bitbuffer.py:17-18), after the 16-byte header of
// make_header (codec.py:102-114).  And, for decompress() (codec.py:167-189): the header parse, the bit-serial
// Huffman decode (huffman.py:66-98) and run-length expansion (huffman.py:36-38), with the reference's
// "swallow any per-block exception" behaviour.
//
// Input is the device stage's layout: int16 [N][64], zig-zag order, element 0 = quantised DC before DPCM.
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/tinyimgcodec_hip.h"
#include "tic_entropy.h"
#include "tic_hooks.h"
#include "tic_tables.h"

#include <algorithm>
#include <chrono>
#include <stdio.h>
#include <atomic>
#include <thread>
#include <vector>

namespace tic {

namespace {

struct EncTables {
    uint32_t dc_code[16];
    uint8_t dc_len[16];
    uint32_t ac_code[256];
    uint8_t ac_len[256];
    // decode: for each code length 1..16 the first code value, count and index into the symbol list
    struct Dec {
        uint16_t first[17];
        uint16_t count[17];
        uint16_t base[17];
        uint8_t sym[256];
        // fast path: the next 16 stream bits -> (code length << 8) | symbol, 0 when no codeword is a prefix of them
        uint16_t lut[65536];
        uint16_t lut11[2048]; // same for codewords of at most 11 bits (L1-resident); 0 = look in lut
    } dcd, acd;
    EncTables() {
        build(kDcBits, kDcVals, dc_code, dc_len, 16, dcd);
        build(kAcBits, kAcVals, ac_code, ac_len, 256, acd);
        build_lut(dcd);
        build_lut(acd);
    }
    static void build_lut(Dec &d) {
        for (int i = 0; i < 65536; i++) d.lut[i] = 0;
        for (int l = 1; l <= 16; l++)
            for (int i = 0; i < d.count[l]; i++) {
                const unsigned code = (unsigned)d.first[l] + (unsigned)i;
                const unsigned lo = code << (16 - l), hi = lo + (1u << (16 - l));
                for (unsigned v = lo; v < hi; v++) d.lut[v] = (uint16_t)((l << 8) | d.sym[d.base[l] + i]);
            }
        for (int i = 0; i < 2048; i++) {
            const uint16_t e = d.lut[i << 5];
            d.lut11[i] = (e >> 8) <= 11 ? e : 0;
        }
    }
    static void build(const uint8_t bits[16], const uint8_t *vals, uint32_t *code_out, uint8_t *len_out, int nsym,
                      Dec &d) {
        for (int i = 0; i < nsym; i++) {
            code_out[i] = 0;
            len_out[i] = 0;
        }
        unsigned code = 0;
        int k = 0;
        for (int l = 1; l <= 16; l++) {
            d.first[l] = (uint16_t)code;
            d.count[l] = bits[l - 1];
            d.base[l] = (uint16_t)k;
            for (int i = 0; i < bits[l - 1]; i++) {
                code_out[vals[k]] = code;
                len_out[vals[k]] = (uint8_t)l;
                d.sym[k] = vals[k];
                code++;
                k++;
            }
            code <<= 1;
        }
    }
};

const EncTables &tables() {
    static const EncTables t;
    return t;
}

// MSB-first bit writer: bits collect at the low end of a 64-bit accumulator; whenever 32 or more are pending the
// top 32 are stored as one big-endian word.  The caller guarantees 8 spare bytes behind `end`.
struct BitWriter {
    uint8_t *p, *end;
    uint64_t acc = 0;
    int nacc = 0;
    bool overflow = false;
    BitWriter(uint8_t *b, uint8_t *e) : p(b), end(e) {}
    inline void put(uint32_t v, int n) { // n <= 32, v < 2^n
        acc = (acc << n) | v;
        nacc += n;
        if (nacc >= 32) {
            nacc -= 32;
            if (p + 4 > end) {
                overflow = true;
                return;
            }
            const uint32_t w = __builtin_bswap32((uint32_t)(acc >> nacc));
            memcpy(p, &w, 4);
            p += 4;
        }
    }
    inline void finish() {
        while (nacc >= 8) {
            if (p >= end) {
                overflow = true;
                return;
            }
            nacc -= 8;
            *p++ = (uint8_t)(acc >> nacc);
        }
        if (nacc > 0) {
            if (p >= end) {
                overflow = true;
                return;
            }
            *p++ = (uint8_t)((acc << (8 - nacc)) & 0xff); // zero padding (bitbuffer.py:17-18)
            nacc = 0;
        }
    }
};

inline int bit_length(uint32_t a) { return a ? 32 - __builtin_clz(a) : 0; } // utils.py:9-10

} // namespace

size_t num_blocks(int h, int w) {
    if (h <= 0 || w <= 0) return 0;
    return (size_t)((h + 7) / 8) * (size_t)((w + 7) / 8);
}

size_t compress_bound(int h, int w) {
    // per block: DC 9+11 bits, 63 x (16+10) AC bits, EOB 4 bits = 1662 bits < 208 bytes
    return 16 + num_blocks(h, w) * 208 + 8;
}

void write_header(uint8_t *out, int h, int w, int quality) {
    uint32_t v[3] = {(uint32_t)h, (uint32_t)w, (uint32_t)quality};
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 4; k++) out[i * 4 + k] = (uint8_t)(v[i] >> (8 * k)); // struct.pack("III") little-endian
    memset(out + 12, 0, 4);                                                         // flag 0: default tables
}

int entropy_encode(const int16_t *zz, int h, int w, int quality, uint8_t *out, size_t cap, size_t *out_len) {
    if (!out || !out_len || (!zz && num_blocks(h, w))) return TIC_E_ARG;
    if (h < 0 || w < 0) return TIC_E_ARG;
    if (quality < 1 || quality > 99) return TIC_E_QUALITY;
    if (cap < 16) return TIC_E_SPACE;
    const EncTables &T = tables();
    write_header(out, h, w, quality);
    BitWriter bw(out + 16, out + cap);
    const size_t n = num_blocks(h, w);
    int prev_dc = 0;
    for (size_t b = 0; b < n; b++) {
        const int16_t *c = zz + b * 64;
        // DC: difference to the previous block in raster order, first block raw (codec.py:34-35)
        int dc = c[0];
        int diff = b ? dc - prev_dc : dc;
        prev_dc = dc;
        {
            uint32_t a = (uint32_t)(diff < 0 ? -diff : diff);
            int size = bit_length(a);
            if (size > 11) return TIC_E_RANGE;
            bw.put(T.dc_code[size], T.dc_len[size]);
            if (size) bw.put((diff < 0 ? ~a : a) & ((1u << size) - 1u), size);
        }
        // AC: (run,size) symbols; ZRL per 16 zeros; EOB always (huffman.py:12-33)
        int last = 63;
        while (last > 0 && c[last] == 0) last--;
        int run = 0;
        for (int k = 1; k <= last; k++) {
            int v = c[k];
            if (v == 0) {
                run++;
                continue;
            }
            while (run >= 16) {
                bw.put(T.ac_code[0xF0], T.ac_len[0xF0]);
                run -= 16;
            }
            const uint32_t a = (uint32_t)(v < 0 ? -v : v);
            const int size = bit_length(a);
            if (size > 10) return TIC_E_RANGE;
            const int sym = (run << 4) | size;
            // codeword (<= 16 bits) and value bits (<= 10) in one append; v < 0 -> one's complement of |v|
            bw.put((T.ac_code[sym] << size) | ((uint32_t)(v + (v >> 31)) & ((1u << size) - 1u)), T.ac_len[sym] + size);
            run = 0;
        }
        bw.put(T.ac_code[0], T.ac_len[0]); // EOB
        if (bw.overflow) return TIC_E_SPACE;
    }
    bw.finish();
    if (bw.overflow) return TIC_E_SPACE;
    *out_len = (size_t)(bw.p - out);
    return TIC_OK;
}

int parse_header(const uint8_t *data, size_t len, int *h, int *w, int *quality, uint32_t *flag) {
    if (!data || len < 16) return TIC_E_STREAM;
    uint32_t v[4];
    for (int i = 0; i < 4; i++)
        v[i] = (uint32_t)data[4 * i] | ((uint32_t)data[4 * i + 1] << 8) | ((uint32_t)data[4 * i + 2] << 16) |
               ((uint32_t)data[4 * i + 3] << 24);
    if (h) *h = (int)v[0];
    if (w) *w = (int)v[1];
    if (quality) *quality = (int)v[2];
    if (flag) *flag = v[3];
    return TIC_OK;
}

namespace {

struct BitReader {
    const uint8_t *p;
    size_t nbits, pos;
    // A read past the end returns no bit but still advances (bitarray slicing semantics, bitbuffer.py:20-23).
    inline bool bit(int &b) {
        bool ok = pos < nbits;
        if (ok) b = (p[pos >> 3] >> (7 - (pos & 7))) & 1;
        pos++;
        return ok;
    }
};

// Fast path while at least 64 bits remain: the 32 stream bits starting at the read position (MSB first).
inline uint32_t peek32(const BitReader &r) {
    uint64_t w;
    memcpy(&w, r.p + (r.pos >> 3), 8); // one unaligned load; the caller guarantees 8 readable bytes
    return (uint32_t)((__builtin_bswap64(w) << (r.pos & 7)) >> 32);
}

// huffman.py:66-74: grow the prefix bit by bit; fail (ValueError) after 17 reads.
inline bool read_code(BitReader &r, const EncTables::Dec &d, int &sym) {
    if (r.pos + 64 <= r.nbits) { // table lookup; an invalid prefix falls through to the bit-serial walk and its quirks
        const uint16_t e = d.lut[peek32(r) >> 16];
        if (e) {
            sym = e & 0xff;
            r.pos += e >> 8;
            return true;
        }
    }
    unsigned code = 0;
    int len = 0;
    for (int i = 0; i <= 16; i++) {
        if (len > 0 && d.count[len] && code >= d.first[len] && code < (unsigned)d.first[len] + d.count[len]) {
            sym = d.sym[d.base[len] + (code - d.first[len])];
            return true;
        }
        if (i == 16) break;
        int b;
        if (r.bit(b)) {
            code = (code << 1) | (unsigned)b;
            len++;
        }
    }
    r.pos++;
    return false;
}

// bitbuffer.py:55-65
inline bool read_int(BitReader &r, int size, int &out) {
    if (size == 0) {
        out = 0;
        return true;
    }
    if (r.pos + 64 <= r.nbits) { // size <= 15 bits, all present
        const uint32_t v = peek32(r) >> (32 - size);
        r.pos += (size_t)size;
        out = (v >> (size - 1)) ? (int)v : -(int)((~v) & ((1u << size) - 1u));
        return true;
    }
    uint32_t v = 0;
    int first = -1, got = 0;
    for (int i = 0; i < size; i++) {
        int b;
        if (r.bit(b)) {
            if (first < 0) first = b;
            v = (v << 1) | (uint32_t)b;
            got++;
        }
    }
    if (!got) return false;
    if (first == 0)
        out = -(int)((~v) & ((1u << got) - 1u));
    else
        out = (int)v;
    return true;
}

inline int16_t sat16(int v) { return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

// One symbol and its value bits (huffman.py:77-98).  While 64 bits remain, a single 32-bit peek covers the codeword
// (<= 16 bits) and the value (<= 15 bits); otherwise, and for prefixes that are no codeword, the bit-serial readers
// above reproduce the reference's behaviour at the end of the stream.
inline bool read_symbol(BitReader &r, const EncTables::Dec &d, int &sym, int &val) {
    if (r.pos + 64 <= r.nbits) {
        const uint32_t pk = peek32(r);
        uint16_t e = d.lut11[pk >> 21];
        if (!e) e = d.lut[pk >> 16];
        if (e) {
            const int len = e >> 8, size = e & 15;
            sym = e & 0xff;
            if (size == 0) {
                val = 0;
            } else {
                const uint32_t v = (pk << len) >> (32 - size);
                val = (v >> (size - 1)) ? (int)v : -(int)((~v) & ((1u << size) - 1u));
            }
            r.pos += (size_t)(len + size);
            return true;
        }
    }
    return read_code(r, d, sym) && read_int(r, sym & 15, val);
}

} // namespace

namespace {

// One block on the table-driven fast path (the caller guarantees 2048 readable bits from pos0: any valid block is at most 64 x 27
// bits long).  STORE: coefficients c[1..63] are written (c must be zero on entry); otherwise the block is only measured.  Returns
// false on anything unusual - a prefix that is no codeword, more than 63 coefficients - with nothing consumed; the caller then
// takes the bit-serial path, which reproduces the reference's behaviour on malformed streams.
template <bool STORE>
inline bool block_fast(const EncTables &T, const uint8_t *p, size_t pos0, int16_t *c, int &dc_diff, size_t &used_out) {
    // bit buffer in a register: `cnt` valid bits at the top of `buf`; branch-free refill to >= 56 bits before every symbol (a
    // data-dependent refill branch mispredicts every few symbols and doubles the time)
    const uint8_t *bp = p + (pos0 >> 3);
    uint64_t buf;
    memcpy(&buf, bp, 8);
    buf = __builtin_bswap64(buf) << (pos0 & 7);
    int cnt = 64 - (int)(pos0 & 7);
    bp += 8;
    size_t used = 0;
    auto refill = [&]() {
        uint64_t w8;
        memcpy(&w8, bp, 8);
        buf |= cnt < 64 ? __builtin_bswap64(w8) >> cnt : 0;
        bp += (63 - (cnt > 63 ? 63 : cnt)) >> 3;
        cnt |= 56;
    };
    auto value = [](uint64_t bits, int len, int size) -> int { // value bits follow the codeword
        // branch-free (the sign bit is a coin flip): x with its top bit clear stands for x - (2^size - 1)
        const int x = (int)(((bits << len) >> 1) >> (63 - size)); // size = 0 gives 0
        const int half = (1 << size) >> 1;
        const int neg_mask = (x - half) >> 31; // all ones when the top bit is clear
        return x + (neg_mask & (1 - (1 << size)));
    };
    uint16_t e = T.dcd.lut11[buf >> 53];
    if (!e) return false; // DC categories are at most 9 bits long
    int len = e >> 8, size = e & 15;
    dc_diff = value(buf, len, size);
    buf <<= len + size;
    cnt -= len + size;
    used += (size_t)(len + size);
    int k = 1;
    for (;;) {
        refill();
        e = T.acd.lut11[buf >> 53];
        if (!e) e = T.acd.lut[buf >> 48];
        if (!e) return false;
        len = e >> 8;
        size = e & 15;
        const uint64_t bits = buf;
        buf <<= len + size;
        cnt -= len + size;
        used += (size_t)(len + size);
        if ((e & 0xff) == 0) break; // EOB
        k += (e >> 4) & 15;
        if (k > 63) return false;
        if (STORE) c[k] = (int16_t)value(bits, len, size);
        k++;
    }
    used_out = used;
    return true;
}

// Parallel decode of a long, well-formed stream (the format has no restart markers: the serial decoder is one dependent chain,
// 44 ms for a 4096^2 frame).  The stream is cut into T bit ranges.  (A) every thread measures blocks from the start of its range
// as if a block started there - a guess, except for the first - recording each block's first bit and DC difference; (B) a serial
// stitch follows the true chain: from the end of a range it measures on until it lands on a block start the next thread also
// recorded; the two decoders are then in the same state (a block start carries none), so the rest of that thread's list is the
// truth - Huffman streams re-synchronise within a few blocks; (C) with every block's first bit and the running DC known the
// threads decode disjoint block ranges straight into the output.  Anything unusual inside the TRUE chain (an invalid prefix,
// a block of more than 63 coefficients) makes the function give up, and the caller decodes serially as before: malformed
// streams keep the reference's quirks.  Returns the number of blocks decoded (0: not attempted / given up) and the read
// position and running DC behind them.
size_t decode_parallel(const EncTables &T, const uint8_t *data, size_t nbits, size_t n, int16_t *zz, size_t &pos_out, int &dc_out) {
    const size_t first_bit = 128;
    if (n < 16384 || nbits < first_bit + (1u << 21) || test_hook("TIC_DECODE_SERIAL")) return 0; // (tic_hooks.h: off unless TIC_TEST_HOOKS=1)
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)(hw ? hw / 2 : 4);
    nt = nt < 2 ? 2 : (nt > 16 ? 16 : nt);
    if (const char *e = test_hook("TIC_DECODE_THREADS")) nt = atoi(e) < 1 ? 1 : (atoi(e) > 64 ? 64 : atoi(e));
    const size_t fast_end = nbits - 2048; // a block may START on the fast path up to here
    const size_t span = (fast_end - first_bit + (size_t)nt - 1) / (size_t)nt;
    struct Trace {
        std::vector<uint64_t> start;
        std::vector<int32_t> dc;
        std::vector<size_t> breaks; // number of blocks recorded when a measurement failed and the thread moved on by one bit
        size_t end = 0;             // first bit behind the last measured block
    };
    std::vector<Trace> tr((size_t)nt);
#ifdef TIC_ABLATION // phase times on stderr (tools/native/dec_time.cpp)
    const bool trace = getenv("TIC_DECODE_TRACE") != nullptr;
#else
    const bool trace = false;
#endif
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(now() - t0).count(); };
    auto t_start = now();
    auto seg_lo = [&](int t) { return first_bit + span * (size_t)t; };
    auto seg_hi = [&](int t) { const size_t e = first_bit + span * (size_t)(t + 1); return e < fast_end ? e : fast_end; };
    {   // (A)
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                Trace &r = tr[(size_t)t];
                const size_t hi = seg_hi(t);
                r.start.reserve(n / (size_t)nt + n / 8);
                r.dc.reserve(n / (size_t)nt + n / 8);
                size_t pos = seg_lo(t);
                while (pos < hi) {
                    int d;
                    size_t used;
                    if (block_fast<false>(T, data, pos, nullptr, d, used)) {
                        r.start.push_back(pos);
                        r.dc.push_back(d);
                        pos += used;
                    } else {
                        if (t == 0) break; // the true chain: the stitch below sees the short trace and gives up
                        r.breaks.push_back(r.start.size());
                        pos++; // a guess that led nowhere (or, behind the point of synchronisation, a malformed stream): next bit
                    }
                }
                r.end = pos;
            });
        for (auto &x : th) x.join();
    }
    if (trace) {
        size_t tot = 0, brk = 0;
        for (auto &r : tr) { tot += r.start.size(); brk += r.breaks.size(); }
        fprintf(stderr, "decode_parallel: %d threads, phase A %.2f ms, %zu blocks measured, %zu failed guesses\n", nt, ms_since(t_start), tot, brk);
    }
    auto t_b = now();
    size_t by_hand = 0;
    // (B) the true chain: block starts and DC differences in order
    std::vector<uint64_t> start;
    std::vector<int32_t> dcd;
    start.reserve(n);
    dcd.reserve(n);
    size_t pos = first_bit;
    int t = 0;
    bool synced = true; // thread 0 starts on a true block start
    size_t idx = 0;     // next entry of thread t's trace on the true chain
    while (start.size() < n && pos < fast_end) {
        if (t < nt && synced) {
            const Trace &r = tr[(size_t)t];
            // copy the rest of this thread's trace (it ends at the first block start at or behind the range's end)
            const size_t take = r.start.size() - idx < n - start.size() ? r.start.size() - idx : n - start.size();
            start.insert(start.end(), r.start.begin() + (long)idx, r.start.begin() + (long)(idx + take));
            dcd.insert(dcd.end(), r.dc.begin() + (long)idx, r.dc.begin() + (long)(idx + take));
            if (idx + take < r.start.size()) break; // n blocks reached
            pos = r.end;
            if (t == 0 && pos < seg_hi(0)) return 0; // the first thread stopped at something unusual
            t++;
            synced = false;
            continue;
        }
        if (t < nt) { // is `pos` a block start the next thread recorded?  (its trace is sorted)
            const Trace &r = tr[(size_t)t];
            if (pos >= seg_hi(t) && !(t == nt - 1)) { // walked through the whole range without meeting its trace: next one
                t++;
                continue;
            }
            const auto it = std::lower_bound(r.start.begin(), r.start.end(), (uint64_t)pos);
            if (it != r.start.end() && *it == (uint64_t)pos) {
                idx = (size_t)(it - r.start.begin());
                // from here on the thread walked the true chain: a failed measurement behind this point is the stream's fault
                if (!r.breaks.empty() && r.breaks.back() > idx) return 0;
                synced = true;
                continue;
            }
        }
        // measure one block of the true chain by hand
        int d;
        size_t used;
        if (!block_fast<false>(T, data, pos, nullptr, d, used)) return 0; // unusual: the serial decoder takes the whole stream
        start.push_back(pos);
        dcd.push_back(d);
        pos += used;
        by_hand++;
    }
    const size_t m = start.size();
    if (trace) fprintf(stderr, "decode_parallel: phase B %.2f ms, %zu blocks on the chain, %zu measured by hand\n", ms_since(t_b), m, by_hand);
    auto t_c = now();
    if (m == 0) return 0;
    // running DC (np.cumsum(dc), codec.py:53) in front of every block
    std::vector<int32_t> run(m);
    {
        int acc = 0;
        for (size_t b = 0; b < m; b++) {
            acc += dcd[b];
            run[b] = acc;
        }
    }
    // (C)
    std::atomic<int> bad{0};
    size_t last_used = 0;
    {
        std::vector<std::thread> th;
        for (int j = 0; j < nt; j++)
            th.emplace_back([&, j]() {
                const size_t b0 = m * (size_t)j / (size_t)nt, b1 = m * (size_t)(j + 1) / (size_t)nt;
                memset(zz + b0 * 64, 0, (b1 - b0) * 64 * sizeof(int16_t)); // (the caller zeroes only what this function leaves)
                for (size_t b = b0; b < b1; b++) {
                    int16_t *c = zz + b * 64;
                    int d;
                    size_t used;
                    if (!block_fast<true>(T, data, (size_t)start[b], c, d, used)) {
                        bad.store(1);
                        return;
                    }
                    c[0] = sat16(run[b]);
                    if (b == m - 1) last_used = used;
                }
            });
        for (auto &x : th) x.join();
    }
    if (trace) fprintf(stderr, "decode_parallel: phase C %.2f ms\n", ms_since(t_c));
    if (bad.load()) return 0;
    pos_out = (size_t)start[m - 1] + last_used;
    dc_out = run[m - 1];
    return m;
}

} // namespace

namespace {
// The serial decoder from block b_first at read position r.pos with running DC `running_dc`: blocks [b_first, n) into zz (zeroed
// here), zz indexed from block b_first (zz points at block b_first).
void decode_serial_from(const EncTables &T, BitReader &r, int running_dc, size_t b_first, size_t n, int16_t *zz) {
    memset(zz, 0, (n - b_first) * 64 * sizeof(int16_t));
    for (size_t b = b_first; b < n; b++) {
        int16_t *c = zz + (b - b_first) * 64;
        int sym, v;
        // Fast path for a whole block while the stream is long enough for any valid block (<= 64 x 27 bits): table
        // look-ups, coefficients written in place.  Anything unusual - a prefix that is no codeword, more than 63
        // coefficients - rewinds to the block's first bit and takes the bit-serial path below, which reproduces the
        // reference's behaviour on malformed streams.
        if (r.pos + 2048 <= r.nbits) {
            int d;
            size_t used;
            if (block_fast<true>(T, r.p, r.pos, c, d, used)) {
                running_dc += d;
                c[0] = sat16(running_dc);
                r.pos += used;
                continue;
            }
            memset(c, 0, 64 * sizeof(int16_t));
        }
        bool have_dc = read_symbol(r, T.dcd, sym, v);
        if (have_dc) running_dc += v;
        c[0] = sat16(running_dc);
        if (!have_dc) continue; // exception before the AC loop: block stays zero (codec.py:185-186)
        int16_t tmp[1100];
        int m = 0;
        bool ok = true, too_long = false;
        for (;;) {
            if (!read_symbol(r, T.acd, sym, v)) {
                ok = false;
                break;
            }
            const int run = sym >> 4;
            // the reference (huffman.py:77-98) reads symbols until EOB or a decode error however long the list grows and
            // rejects a list longer than 63 only afterwards: keep consuming (the read position matters), stop storing
            if (m + run + 1 > 1090) too_long = true;
            if (!too_long) {
                for (int z = 0; z < run; z++) tmp[m++] = 0;
                tmp[m++] = sat16(v);
            }
            if (sym == 0) break;
        }
        if (!ok || too_long) continue;
        m -= 1; // decode_run_length drops the element produced by EOB (huffman.py:36-38)
        if (m > 63) continue;
        memcpy(c + 1, tmp, (size_t)m * sizeof(int16_t));
    }
}
} // namespace

int entropy_decode(const uint8_t *data, size_t len, int h, int w, int16_t *zz) {
    const EncTables &T = tables();
    const size_t n = num_blocks(h, w);
    BitReader r{data, len * 8, 128};
    int running_dc = 0; // np.cumsum(dc), codec.py:53
    size_t b_first = 0;
    {
        size_t pos = 0;
        int dc = 0;
        const size_t done = decode_parallel(T, data, len * 8, n, zz, pos, dc); // zeroes and fills blocks [0, done)
        if (done) {
            b_first = done;
            r.pos = pos;
            running_dc = dc;
        }
    }
    decode_serial_from(T, r, running_dc, b_first, n, zz + b_first * 64); // (everything, if the parallel attempt gave up)
    return TIC_OK;
}

// Blocks [first_block, n) from read position pos_bits with running DC `running_dc` (what the device decoder leaves to the host:
// the blocks that start in the last 2048 bits of the stream) into zz_tail, which holds n - first_block blocks.
int entropy_decode_tail(const uint8_t *data, size_t len, int h, int w, size_t first_block, size_t pos_bits, int running_dc, int16_t *zz_tail) {
    const size_t n = num_blocks(h, w);
    if (first_block >= n) return TIC_OK;
    BitReader r{data, len * 8, pos_bits};
    decode_serial_from(tables(), r, running_dc, first_block, n, zz_tail);
    return TIC_OK;
}

void dec_luts_fill(uint16_t *dc11, uint16_t *ac11, uint16_t *ac16) {
    const EncTables &T = tables();
    memcpy(dc11, T.dcd.lut11, sizeof T.dcd.lut11);
    memcpy(ac11, T.acd.lut11, sizeof T.acd.lut11);
    memcpy(ac16, T.acd.lut, sizeof T.acd.lut);
}

// Chain tables of the device decoder's measure walk: what a walk that only needs the END of every block consumes in one look-up.
// From a window of W stream bits: the first symbol (DC category or AC run/size; its codeword must be at most 11 bits, as in lut11 -
// the walk resolves longer AC codewords in a second step), then as many further AC symbols as have their CODEWORD inside the window
// (a symbol's length is its codeword's plus its size: the value bits themselves are not needed), stopping behind an EOB (the DC table
// comes next) and in front of anything that is not a codeword.  Symbol by symbol this is exactly the sequence of steps the
// one-symbol walk makes, so both walks stand on the same bits at every block end.
// An entry is everything the walk's next step needs: (bits consumed << 3) | (table of the next step << 1) | (the chain ended with EOB) - table 0: the DC
// category comes next (behind an EOB), 1: AC symbols, 2: the long AC codewords.  A window without a codeword is an entry like any other: with the DC category
// next, skip a bit and stay (a walk out of step, or a damaged stream); inside the AC symbols it is the prefix of a long codeword - consume nothing, go to
// table 2; in table 2 skip a bit and go back to the AC symbols.  (Until late in round 6 the entry was (bits << 1) | EOB, 0 for no codeword, and the walk
// derived the rest: eight instructions of a step of 49.)
static inline uint8_t chain_entry(int bits, int next_table, bool eob) { return (uint8_t)((bits << 3) | (next_table << 1) | (eob ? 1 : 0)); }
static void build_chain(const uint16_t *first_lut16, bool first_is_dc, const uint16_t *ac_lut16, int W, uint8_t *out) {
    for (unsigned w = 0; w < (1u << W); w++) {
        int pos = 0, total = 0;
        bool eob = false, first = true;
        while (pos < W) {
            const int vis = W - pos;
            const unsigned next16 = ((w << pos) & ((1u << W) - 1u)) << (16 - W); // the bits from pos on, zeros behind the window
            const uint16_t e = (first ? first_lut16 : ac_lut16)[next16];
            const int L = e >> 8;
            if (e == 0 || L > vis || (first && L > 11)) break; // no codeword, or not wholly inside the window
            const bool is_dc = first && first_is_dc;
            const int len = L + (e & 15);
            total += len;
            pos += len;
            first = false;
            if (!is_dc && (e & 0xff) == 0) {
                eob = true;
                break;
            }
        }
        out[w] = total ? chain_entry(total, eob ? 0 : 1, eob) : (first_is_dc ? chain_entry(1, 0, false) : chain_entry(0, 2, false)); // total <= 22
    }
}
void dec_chain_luts_fill(uint8_t *mdc, uint8_t *mac, uint8_t *mlong) {
    const EncTables &T = tables();
    build_chain(T.dcd.lut, true, T.acd.lut, 11, mdc);
    build_chain(T.acd.lut, false, T.acd.lut, 12, mac);
    memset(mlong, chain_entry(1, 1, false), 256); // (also the slot of an index out of range, behind the 192 codewords)
    for (int i = 0; i < 0x10000 - 0xff40; i++) {
        const uint16_t e = T.acd.lut[0xff40 + i];
        if (e) mlong[i] = chain_entry((e >> 8) + (e & 15), 1, false); // (at most 16 + 11 bits; never EOB: its codeword has 4 bits)
    }
}

// Pair table of the device decoder's fused kernel (a lane per block, values needed): what one look-up of the next 11 stream bits
// settles inside the AC symbols - the first symbol (codeword of at most 11 bits, as in lut11) and, when the first is not EOB and the
// CODEWORD of the symbol behind it lies inside the same 11 bits, that second symbol too (its value bits may lie behind the window).
// Entry: bits 0-12 the first symbol as in lut11 (size | run << 4 | codeword length << 8), bit 13 a second symbol follows, bits 14-26
// the second symbol in the same form, bits 27-31 the stream bits both consume together.  0: no codeword of at most 11 bits here.
// long32: the AC codewords of 12..16 bits (index: the next 16 bits - 0xff40), one symbol each, same form.
void dec_pair_luts_fill(uint32_t *ac2 /*[2048]*/, uint32_t *long32 /*[192]*/) {
    const EncTables &T = tables();
    for (unsigned w = 0; w < 2048u; w++) {
        const uint32_t e1 = T.acd.lut11[w];
        uint32_t e = 0;
        if (e1) {
            const unsigned used = (e1 >> 8) + (e1 & 15u);
            e = e1 | (uint32_t)used << 27;
            if ((e1 & 0xffu) != 0 && used < 11u) {
                const unsigned left = 11u - used;
                const uint32_t e2 = T.acd.lut11[(w << used) & 0x7ffu]; // (zeros behind the window: a codeword no longer than `left` does not see them)
                if (e2 && (e2 >> 8) <= left) {
                    const unsigned both = used + (e2 >> 8) + (e2 & 15u);
                    e = e1 | 1u << 13 | (uint32_t)e2 << 14 | (uint32_t)both << 27;
                }
            }
        }
        ac2[w] = e;
    }
    for (int i = 0; i < 0x10000 - 0xff40; i++) {
        const uint32_t e = T.acd.lut[0xff40 + i];
        long32[i] = e ? e | ((e >> 8) + (e & 15u)) << 27 : 0u;
    }
}

} // namespace tic
