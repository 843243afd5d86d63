// snk_deflate.h round trip + speed: ./deflate_test file [slice bytes]   (output checked with zlib's inflate AND snk_inflate.h)
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <vector>
#include <string>
#include "../../soapnuke_amd/host/snk_deflate.h"
#include "../../soapnuke_amd/host/snk_inflate.h"
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    std::vector<uint8_t> raw;
    { uint8_t b[1 << 16]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) raw.insert(raw.end(), b, b + n); }
    fclose(f);
    const size_t slice = argc > 2 ? (size_t)atol(argv[2]) : raw.size();
    snk::FastDeflate enc;
    std::string z;
    double t0 = now();
    for (size_t at = 0; at < raw.size() || at == 0; at += slice ? slice : 1) {
        const size_t n = std::min(slice ? slice : raw.size(), raw.size() - at);
        enc.gzip_member(raw.data() + at, n, z, getenv("FASTQ") != nullptr);
        if (raw.empty()) break;
    }
    double t1 = now();
    // zlib reference compression for ratio / speed
    std::string zr;
    {
        z_stream s; memset(&s, 0, sizeof s);
        deflateInit2(&s, 2, Z_DEFLATED, 31, 8, Z_DEFAULT_STRATEGY);
        zr.resize(deflateBound(&s, raw.size()) + 64);
        s.next_in = raw.data(); s.avail_in = (uInt)raw.size(); s.next_out = (Bytef *)&zr[0]; s.avail_out = (uInt)zr.size();
        deflate(&s, Z_FINISH); zr.resize(zr.size() - s.avail_out); deflateEnd(&s);
    }
    double t2 = now();
    // decode with zlib (multi-member)
    std::vector<uint8_t> back(raw.size() + 16);
    size_t got = 0;
    {
        size_t in_at = 0;
        while (in_at < z.size()) {
            z_stream s; memset(&s, 0, sizeof s);
            if (inflateInit2(&s, 31) != Z_OK) return 4;
            s.next_in = (Bytef *)&z[in_at]; s.avail_in = (uInt)(z.size() - in_at);
            s.next_out = back.data() + got; s.avail_out = (uInt)(back.size() - got);
            const int rc = inflate(&s, Z_FINISH);
            if (rc != Z_STREAM_END) { printf("ERROR zlib inflate rc %d (%s) at member offset %zu\n", rc, s.msg ? s.msg : "", in_at); return 2; }
            got = (size_t)(s.next_out - back.data());
            in_at = z.size() - s.avail_in;
            inflateEnd(&s);
        }
    }
    const bool same_zlib = got == raw.size() && memcmp(back.data(), raw.data(), raw.size()) == 0;
    // decode with our own decoder
    std::vector<uint8_t> back2(snk::GzipInflate::HIST + raw.size() + 16);
    snk::GzipInflate inf;
    inf.init((const uint8_t *)z.data(), z.size());
    size_t got2 = 0;
    for (;;) {
        const size_t k = inf.run(back2.data() + snk::GzipInflate::HIST + got2, raw.size() + 16 - got2);
        got2 += k;
        if (inf.error()) { printf("ERROR snk_inflate: %s\n", inf.error()); return 2; }
        if (k == 0) break;
    }
    const bool same_own = got2 == raw.size() && memcmp(back2.data() + snk::GzipInflate::HIST, raw.data(), raw.size()) == 0;
    printf("%s: %zu -> %zu bytes (zlib -2: %zu)  %s  ours %.3fs (%.0f MB/s)  zlib %.3fs (%.0f MB/s)\n", argv[1], raw.size(), z.size(), zr.size(),
           same_zlib && same_own ? "ROUNDTRIP_OK" : "DIFFERENT", t1 - t0, raw.size() / (t1 - t0) / 1e6, t2 - t1, raw.size() / (t2 - t1) / 1e6);
    return same_zlib && same_own ? 0 : 1;
}
