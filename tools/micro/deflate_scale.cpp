// how does zlib deflate (level 2) scale with threads on this host?  g++ -O2 -pthread deflate_scale.cpp -lz
#include <zlib.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>
int main(int argc, char **argv) {
    const size_t N = 32 << 20;
    std::vector<unsigned char> src(N);
    unsigned s = 12345;
    const char *al = "ACGT";
    for (size_t i = 0; i < N; ++i) { s = s * 1664525u + 1013904223u; src[i] = (i % 330) < 25 ? '@' + (i % 7) : ((i % 330) < 176 ? al[(s >> 16) & 3] : 33 + 30 + ((s >> 20) % 10)); }
    for (int t : {1, 8, 16, 32, 64, 128}) {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int k = 0; k < t; ++k) th.emplace_back([&] {
            std::vector<unsigned char> out(N);
            z_stream z; memset(&z, 0, sizeof z);
            deflateInit2(&z, 2, Z_DEFLATED, 31, 8, Z_DEFAULT_STRATEGY);
            z.next_in = src.data(); z.avail_in = N; z.next_out = out.data(); z.avail_out = N;
            deflate(&z, Z_FINISH); deflateEnd(&z);
        });
        for (auto &x : th) x.join();
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %3d: %.3f s  %.0f MB/s total  %.0f MB/s per thread\n", t, dt, t * (N / 1e6) / dt, (N / 1e6) / dt);
    }
}
