// host_selftest.cpp - the product's host entropy coder (tic_entropy.cpp) against the oracle's (tic_oracle.c) on random and
// adversarial coefficient blocks, built with AddressSanitizer + UBSan (CPU only; tests/test_host_cpu.py runs it).
// Checks: identical streams, decode(encode(x)) == x, bounded writes (exact-size buffers), clean errors on truncated
// and corrupted streams.  The oracle is the checker here, as everywhere under tests/.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../tinyimgcodec_amd/csrc/tic_entropy.h"
extern "C" {
#include "../../oracle/tic_oracle.h"
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}

static int fail(const char *what, int h, int w, int mode) {
    fprintf(stderr, "FAIL %s (h=%d w=%d mode=%d)\n", what, h, w, mode);
    return 1;
}

int main() {
    setenv("TIC_TEST_HOOKS", "1", 1); // tic_hooks.h: TIC_DECODE_SERIAL below is a test hook
    static const int kZig[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                 41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    (void)kZig;
    int cases = 0;
    const int shapes[][2] = {{8, 8}, {1, 1}, {8, 72}, {40, 24}, {17, 33}, {64, 64}, {0, 8}};
    for (auto &sh : shapes) {
        const int h = sh[0], w = sh[1];
        const size_t n = tic::num_blocks(h, w);
        for (int mode = 0; mode < 6; mode++) {
            // zig-zag int16 coefficients with un-differenced DC, as the transform kernels produce them
            std::vector<int16_t> zz(n * 64 + 1);
            for (size_t b = 0; b < n; b++)
                for (int k = 0; k < 64; k++) {
                    int v = 0;
                    const uint32_t r = rnd();
                    switch (mode) {
                    case 0: v = (int)(r % 41) - 20; break;                                  // dense small values
                    case 1: v = (r % 7 == 0) ? (int)(r >> 8) % 2047 - 1023 : 0; break;      // sparse, full AC range
                    case 2: v = (k == 0) ? (int)(r % 2047) - 1023 : 0; break;               // DC only (EOB right away)
                    case 3: v = (k == 63 || k == 0) ? 1 : 0; break;                         // longest zero run + ZRLs
                    case 4: v = (k % 17 == 16) ? -1023 : 0; break;                          // runs of exactly 16
                    default: v = (int)(r % 2047) - 1023; break;                             // everything maximal
                    }
                    if (k == 0) v = (v > 1023) ? 1023 : (v < -1023 ? -1023 : v);            // keeps DC differences codable
                    zz[b * 64 + k] = (int16_t)v;
                }
            std::vector<int32_t> dc(n + 1), ac(n * 63 + 1);
            for (size_t b = 0; b < n; b++) {
                dc[b] = (int32_t)zz[b * 64] - (b ? (int32_t)zz[(b - 1) * 64] : 0);
                for (int k = 1; k < 64; k++) ac[b * 63 + k - 1] = zz[b * 64 + k];
            }
            const size_t bound = tic::compress_bound(h, w);
            std::vector<uint8_t> ref(bound + 16);
            size_t ref_len = 0;
            const int rc_ref = tico_entropy_encode(dc.data(), ac.data(), h, w, 50, ref.data(), ref.size(), &ref_len);
            size_t got_len = 0;
            std::vector<uint8_t> probe(bound);
            const int rc = tic::entropy_encode(zz.data(), h, w, 50, probe.data(), probe.size(), &got_len);
            if ((rc == 0) != (rc_ref == 0)) return fail("encoders disagree on codability", h, w, mode);
            cases++;
            if (rc != 0) continue; // |DC difference| > 2047 has no code in either coder
            if (got_len != ref_len || memcmp(probe.data(), ref.data(), ref_len) != 0) return fail("stream mismatch", h, w, mode);
            // exact-size output buffer (ASan guards its end), then one byte too small
            std::vector<uint8_t> exact(got_len);
            size_t l2 = 0;
            if (tic::entropy_encode(zz.data(), h, w, 50, exact.data(), exact.size(), &l2) != 0 || l2 != got_len)
                return fail("exact-size buffer rejected", h, w, mode);
            if (got_len > 16) {
                std::vector<uint8_t> small(got_len - 1);
                if (tic::entropy_encode(zz.data(), h, w, 50, small.data(), small.size(), &l2) == 0)
                    return fail("undersized buffer accepted", h, w, mode);
            }
            // decode
            std::vector<int16_t> back(n * 64 + 1, (int16_t)0x5A5A);
            if (tic::entropy_decode(exact.data(), exact.size(), h, w, back.data()) != 0) return fail("decode failed", h, w, mode);
            if (n && memcmp(back.data(), zz.data(), n * 128) != 0) return fail("round trip mismatch", h, w, mode);
            // truncated and corrupted streams must fail or decode within bounds, never touch memory outside
            for (size_t cut = 16; cut < exact.size(); cut += 1 + exact.size() / 23) {
                std::vector<uint8_t> t(exact.begin(), exact.begin() + cut);
                (void)tic::entropy_decode(t.data(), t.size(), h, w, back.data());
            }
            for (int k = 0; k < 32 && exact.size() > 16; k++) {
                std::vector<uint8_t> t(exact);
                t[16 + rnd() % (t.size() - 16)] ^= (uint8_t)(1u << (rnd() & 7));
                (void)tic::entropy_decode(t.data(), t.size(), h, w, back.data());
            }
        }
    }
    // ---- long streams: the parallel decoder (ranges measured speculatively, stitched, decoded in parallel) against the serial one
    //      (TIC_DECODE_SERIAL), on well-formed streams and on streams damaged in the middle, where the parallel path has to notice
    //      and hand the whole stream to the serial decoder (the reference's quirks on malformed data live there)
    for (int mode : {0, 1, 5, 6}) {
        const int h = 1024, w = 1536; // 24,576 blocks
        const size_t n = tic::num_blocks(h, w);
        std::vector<int16_t> zz(n * 64);
        for (size_t b = 0; b < n; b++)
            for (int k = 0; k < 64; k++) {
                const uint32_t r = rnd();
                int v;
                switch (mode) {
                case 0: v = (int)(r % 41) - 20; break;
                case 1: v = (r % 7 == 0) ? (int)(r >> 8) % 2047 - 1023 : 0; break;
                case 5: v = (int)(r % 2047) - 1023; break;
                default: v = (k < 6 + (int)(b % 5)) ? (int)(r % 255) - 127 : 0; break; // natural-image-like: short blocks of varying length
                }
                if (k == 0) v = (int)(r % 1500) - 750;
                zz[b * 64 + k] = (int16_t)v;
            }
        const size_t bound = tic::compress_bound(h, w);
        std::vector<uint8_t> bs(bound);
        size_t len = 0;
        if (tic::entropy_encode(zz.data(), h, w, 50, bs.data(), bs.size(), &len) != 0) return fail("long stream: encode", h, w, mode);
        bs.resize(len);
        std::vector<int16_t> par(n * 64), ser(n * 64);
        auto both = [&](const std::vector<uint8_t> &t) {
            unsetenv("TIC_DECODE_SERIAL");
            (void)tic::entropy_decode(t.data(), t.size(), h, w, par.data());
            setenv("TIC_DECODE_SERIAL", "1", 1);
            (void)tic::entropy_decode(t.data(), t.size(), h, w, ser.data());
            unsetenv("TIC_DECODE_SERIAL");
            return memcmp(par.data(), ser.data(), n * 128) == 0;
        };
        if (!both(bs)) return fail("long stream: parallel != serial", h, w, mode);
        if (memcmp(par.data(), zz.data(), n * 128) != 0) return fail("long stream: round trip mismatch", h, w, mode);
        cases++;
        for (int k = 0; k < 12; k++) { // one flipped bit somewhere: both decoders must agree on every coefficient
            std::vector<uint8_t> t(bs);
            t[16 + rnd() % (t.size() - 16)] ^= (uint8_t)(1u << (rnd() & 7));
            if (!both(t)) return fail("long stream, flipped bit: parallel != serial", h, w, mode * 100 + k);
            cases++;
        }
        for (int k = 0; k < 4; k++) { // truncated
            std::vector<uint8_t> t(bs.begin(), bs.begin() + (long)(bs.size() / 5 * (size_t)(k + 1)));
            if (!both(t)) return fail("long stream, truncated: parallel != serial", h, w, mode * 100 + k);
            cases++;
        }
    }
    // ---- the device decoder's chain tables (dec_chain_luts_fill): a walk that consumes a chain of symbols per look-up must stand on the
    //      same bits at every EOB as the walk that takes one symbol per look-up (dec_luts_fill's tables: what the device decoder's
    //      fused kernel and rounds 2-3's measure kernel use), from any bit of any stream, in step with the true symbols or not
    {
        std::vector<uint16_t> dc11(2048), ac11(2048), ac16(65536);
        std::vector<uint8_t> mdc(2048), mac(4096), mlong(256);
        tic::dec_luts_fill(dc11.data(), ac11.data(), ac16.data());
        tic::dec_chain_luts_fill(mdc.data(), mac.data(), mlong.data());
        for (int kind = 0; kind < 3; kind++) {
            // kind 0: a real stream of dense blocks; 1: of short blocks; 2: random bits
            const int h = 256, w = 512;
            const size_t n = tic::num_blocks(h, w);
            std::vector<uint8_t> bs(tic::compress_bound(h, w) + 64, 0);
            size_t len = bs.size() - 64;
            if (kind < 2) {
                std::vector<int16_t> zz(n * 64);
                for (size_t b = 0; b < n; b++)
                    for (int k = 0; k < 64; k++) {
                        const uint32_t r = rnd();
                        int v = kind == 0 ? (r % 5 == 0 ? 0 : (int)(r % 2047) - 1023) : (k < 4 + (int)(b % 7) ? (int)(r % 31) - 15 : 0);
                        if (kind == 0 && r % 97 == 0) v = 0;
                        zz[b * 64 + k] = (int16_t)v;
                    }
                if (tic::entropy_encode(zz.data(), h, w, 50, bs.data(), bs.size() - 64, &len) != 0) return fail("chain tables: encode", h, w, kind);
            } else {
                for (size_t i = 16; i < len; i++) bs[i] = (uint8_t)rnd();
            }
            const size_t nbits = len * 8;
            auto peek32 = [&](size_t pos) {
                uint64_t v = 0;
                for (int k = 0; k < 8; k++) v = (v << 8) | bs[(pos >> 3) + (size_t)k];
                return (uint32_t)((v << (pos & 7)) >> 32);
            };
            for (int start = 0; start < 400; start++) {
                const size_t from = 128 + (size_t)(rnd() % (uint32_t)(nbits - 128 - 4096));
                // one symbol per step (the rules of the device walk: an invalid prefix skips a bit; EOB only inside the AC symbols)
                std::vector<size_t> ends1, ends2;
                {
                    size_t pos = from;
                    bool at_dc = true;
                    while (pos < from + 3000) {
                        const uint32_t pk = peek32(pos);
                        uint32_t e = at_dc ? dc11[pk >> 21] : ac11[pk >> 21];
                        if (!e && !at_dc) e = ac16[pk >> 16];
                        if (!e) {
                            pos += 1;
                            continue;
                        }
                        pos += (e >> 8) + (e & 15);
                        const bool eob = !at_dc && (e & 0xff) == 0;
                        if (eob) ends1.push_back(pos);
                        at_dc = eob;
                    }
                }
                {
                    size_t pos = from;
                    bool at_dc = true, in_long = false;
                    while (pos < from + 3000) {
                        const uint32_t pk = peek32(pos);
                        const uint32_t li = (pk >> 16) - 0xff40u;
                        const uint32_t e = in_long ? mlong[li < 192u ? li : 192u] : (at_dc ? mdc[pk >> 21] : mac[pk >> 20]);
                        const bool eob = (e & 1u) != 0; // entry = (bits << 3) | (next table << 1) | EOB, also where no codeword is (tic_entropy.cpp chain_entry)
                        pos += e >> 3;
                        if (eob) ends2.push_back(pos);
                        at_dc = ((e >> 1) & 3u) == 0u;
                        in_long = ((e >> 1) & 3u) == 2u;
                    }
                }
                // the chain walk may overshoot the 3,000-bit horizon by one chain: compare the common prefix, which must be nearly all
                const size_t m = ends1.size() < ends2.size() ? ends1.size() : ends2.size();
                if (m + 2 < ends1.size() || m + 2 < ends2.size()) return fail("chain tables: different number of block ends", (int)from, start, kind);
                for (size_t k = 0; k < m; k++)
                    if (ends1[k] != ends2[k]) return fail("chain tables: a block end differs", (int)from, (int)k, kind);
                cases++;
            }
        }
    }
    // ---- the fused kernel's pair table (dec_pair_luts_fill): one look-up of the next 11 bits gives up to two AC symbols.  The kernel's
    //      loop (tic_entropy_dec_gpu.hip, phase 1), restated here step for step, must emit the same (position, value) pairs, end on the same
    //      bit and flag the same blocks as a loop that takes one symbol per look-up from dec_luts_fill's tables - from any bit of a real
    //      stream or of random bits
    {
        std::vector<uint16_t> dc11(2048), ac11(2048), ac16(65536);
        std::vector<uint32_t> ac2(2048), long32(196, 0u);
        tic::dec_luts_fill(dc11.data(), ac11.data(), ac16.data());
        tic::dec_pair_luts_fill(ac2.data(), long32.data());
        auto value_of = [](uint32_t pk, int len, int size) {
            if (size == 0) return 0;
            const uint32_t x = (pk << len) >> (32 - size);
            return (x >> (size - 1)) ? (int)x : (int)x - ((1 << size) - 1);
        };
        size_t pairs_seen = 0;
        for (int kind = 0; kind < 3; kind++) {
            const int h = 256, w = 512;
            const size_t n = tic::num_blocks(h, w);
            std::vector<uint8_t> bs(tic::compress_bound(h, w) + 64, 0);
            size_t len = bs.size() - 64;
            if (kind < 2) {
                std::vector<int16_t> zz(n * 64);
                for (size_t b = 0; b < n; b++)
                    for (int k = 0; k < 64; k++) {
                        const uint32_t r = rnd();
                        int v = kind == 0 ? (r % 5 == 0 ? 0 : (int)(r % 2047) - 1023) : (r % 3 == 0 ? 0 : (int)(r % 15) - 7);
                        zz[b * 64 + k] = (int16_t)v;
                    }
                if (tic::entropy_encode(zz.data(), h, w, 50, bs.data(), bs.size() - 64, &len) != 0) return fail("pair table: encode", h, w, kind);
            } else {
                for (size_t i = 16; i < len; i++) bs[i] = (uint8_t)rnd();
            }
            const size_t nbits = len * 8;
            auto peek32 = [&](size_t pos) {
                uint64_t v = 0;
                for (int k = 0; k < 8; k++) v = (v << 8) | bs[(pos >> 3) + (size_t)k];
                return (uint32_t)((v << (pos & 7)) >> 32);
            };
            struct Out { std::vector<int> at, val; size_t end; bool ok; };
            for (int start = 0; start < 4000; start++) {
                const size_t from = 128 + (size_t)(rnd() % (uint32_t)(nbits - 128 - 4096));
                Out a, b;
                { // one symbol per look-up
                    size_t pos = from;
                    int k = 1;
                    a.ok = true;
                    for (;;) {
                        const uint32_t pk = peek32(pos);
                        uint32_t e = ac11[pk >> 21];
                        if (!e) e = ac16[pk >> 16];
                        if (!e) { a.ok = false; break; }
                        const int l = (int)(e >> 8), sz = (int)(e & 15);
                        if ((e & 0xff) == 0) { pos += (size_t)l; break; }
                        const int k_at = k + (int)((e >> 4) & 15);
                        if (k_at > 63) { a.ok = false; break; }
                        a.at.push_back(k_at);
                        a.val.push_back(value_of(pk, l, sz));
                        pos += (size_t)(l + sz);
                        k = k_at + 1;
                    }
                    a.end = pos;
                }
                { // the kernel's loop
                    size_t pos = from;
                    int k = 1;
                    bool live = true, in_long = false;
                    b.ok = true;
                    while (live) {
                        const uint32_t pk = peek32(pos);
                        const uint32_t li = (pk >> 16) - 0xff40u;
                        const uint32_t e = in_long ? long32[li < 192u ? li : 192u] : ac2[pk >> 21];
                        const bool none = e == 0u, esc = none && !in_long, nocode = none && in_long;
                        const bool eob1 = !none && (e & 0xffu) == 0u;
                        const int len1 = (int)((e >> 8) & 31u), size1 = (int)(e & 15u);
                        const int k1 = k + (int)((e >> 4) & 15u);
                        const bool bad1 = nocode || (!none && !eob1 && k1 > 63);
                        if (!none && !eob1 && !bad1) { b.at.push_back(k1); b.val.push_back(value_of(pk, len1, size1)); }
                        const bool has2 = ((e >> 13) & 1u) != 0u;
                        const uint32_t e2 = e >> 14;
                        const bool eob2 = has2 && (e2 & 0xffu) == 0u;
                        const int len2 = (int)((e2 >> 8) & 31u), size2 = (int)(e2 & 15u);
                        const int k2 = k1 + 1 + (int)((e2 >> 4) & 15u);
                        const bool bad2 = has2 && !bad1 && !eob2 && k2 > 63;
                        if (has2 && !eob2 && !bad1 && !bad2) { b.at.push_back(k2); b.val.push_back(value_of(pk << (len1 + size1), len2, size2)); pairs_seen++; }
                        if (has2 && (eob1 || in_long)) return fail("pair table: a second symbol behind an EOB or a long codeword", (int)from, start, kind);
                        pos += none ? 0u : (size_t)(e >> 27);
                        k = none ? k : (has2 ? k2 + 1 : k1 + 1);
                        in_long = esc;
                        b.ok = b.ok && !bad1 && !bad2;
                        live = !eob1 && !eob2 && !bad1 && !bad2;
                    }
                    b.end = pos;
                }
                if (a.ok != b.ok) return fail("pair table: one walk flags the block, the other does not", (int)from, start, kind);
                if (a.at != b.at || a.val != b.val || (a.ok && a.end != b.end)) return fail("pair table: the walks differ", (int)from, start, kind);
                cases++;
            }
        }
        if (pairs_seen < 10000) return fail("pair table: hardly any look-up gave two symbols", (int)pairs_seen, 0, 0);
    }
    printf("host_selftest ok: %d cases\n", cases);
    return 0;
}
