#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <time.h>
#include "../../soapnuke_amd/host/snk_crc32.h"
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main() {
    std::vector<uint8_t> b(64 << 20);
    unsigned s = 1;
    for (auto &x : b) { s = s * 1664525u + 1013904223u; x = (uint8_t)(s >> 24); }
    int bad = 0;
    for (int it = 0; it < 20000; ++it) {
        s = s * 1664525u + 1013904223u; size_t off = (s >> 8) % 4096;
        s = s * 1664525u + 1013904223u; size_t n = (s >> 8) % (it < 10000 ? 700 : 70000);
        s = s * 1664525u + 1013904223u; uint32_t c0 = it % 3 ? s : 0;
        if (snk::crc32_fast(c0, b.data() + off, n) != (uint32_t)crc32_z(c0, b.data() + off, n)) { if (bad++ < 5) printf("MISMATCH off %zu n %zu\n", off, n); }
    }
    double t0 = now(); uint32_t a = snk::crc32_fast(0, b.data(), b.size()); double t1 = now(); uint32_t z = (uint32_t)crc32_z(0, b.data(), b.size()); double t2 = now();
    printf("%s  clmul %.0f MB/s  zlib %.0f MB/s  have_clmul %d\n", (bad == 0 && a == z) ? "CRC_OK" : "CRC_BAD", b.size() / (t1 - t0) / 1e6, b.size() / (t2 - t1) / 1e6, (int)snk::crc32_have_clmul());
    return bad || a != z;
}
