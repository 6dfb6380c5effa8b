// dec_time.cpp - the host Huffman decoder alone on a 4096^2 stream: parallel (TIC_DECODE_THREADS, phase times with TIC_DECODE_TRACE)
// against serial.  g++ -O2 -std=c++17 -DTIC_ABLATION -o /tmp/dec_time tools/native/dec_time.cpp tinyimgcodec_amd/csrc/tic_entropy.cpp -lpthread
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../tinyimgcodec_amd/csrc/tic_entropy.h"
int main() {
    const int h = 4096, w = 4096;
    const size_t n = tic::num_blocks(h, w);
    std::vector<int16_t> zz(n * 64);
    uint64_t s = 88172645463325252ull;
    for (size_t b = 0; b < n; b++)
        for (int k = 0; k < 64; k++) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            int v = (k < 44) ? (int)((s >> 32) % 41) - 20 : 0;
            if (k == 0) v = (int)((s >> 32) % 1500) - 750;
            zz[b * 64 + k] = (int16_t)v;
        }
    std::vector<uint8_t> bs(tic::compress_bound(h, w));
    size_t len = 0;
    tic::entropy_encode(zz.data(), h, w, 50, bs.data(), bs.size(), &len);
    std::vector<int16_t> out(n * 64);
    for (int mode = 0; mode < 2; mode++) {
        if (mode) setenv("TIC_DECODE_SERIAL", "1", 1);
        for (int rep = 0; rep < 3; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            tic::entropy_decode(bs.data(), len, h, w, out.data());
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("%s: %.2f ms (%zu bytes) %s\n", mode ? "serial" : "parallel", ms, len, memcmp(out.data(), zz.data(), n * 128) ? "MISMATCH" : "ok");
        }
    }
}
