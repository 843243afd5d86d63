// Byte-for-byte check of soapnuke_amd/host/snk_inflate.h against zlib's gzread, plus speed.
//   g++ -O2 -std=c++17 -o inflate_test inflate_test.cpp -lz && ./inflate_test file.gz [chunk]
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <vector>
#include "../../soapnuke_amd/host/snk_inflate.h"
#include "../../soapnuke_amd/host/snk_pgunzip.h"
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv) {
    const char *path = argv[1];
    const size_t chunk = argc > 2 ? (size_t)atol(argv[2]) : (size_t)1 << 22;
    double t0 = now();
    std::vector<uint8_t> ref;
    {
        gzFile f = gzopen(path, "rb");
        gzbuffer(f, 1 << 20);
        std::vector<uint8_t> buf(1 << 22);
        int n;
        while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) ref.insert(ref.end(), buf.begin(), buf.begin() + n);
        gzclose(f);
    }
    double t1 = now();
    int fd = open(path, O_RDONLY);
    struct stat st; fstat(fd, &st);
    const uint8_t *in = (const uint8_t *)mmap(NULL, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (argc > 3 && !strcmp(argv[3], "par")) {           // ./inflate_test file.gz <out block> par <threads> <chunk bytes>
        const int threads = argc > 4 ? atoi(argv[4]) : 4;
        const size_t cb = argc > 5 ? (size_t)atol(argv[5]) : (size_t)4 << 20;
        double a = now();
        snk::ParallelGunzip pz(in, st.st_size, threads, cb);
        std::vector<uint8_t> got, blk(chunk);
        got.reserve(ref.size());
        for (;;) {
            const size_t n = pz.run(blk.data(), chunk);
            got.insert(got.end(), blk.begin(), blk.begin() + n);
            if (pz.error()) { printf("ERROR %s after %zu bytes\n", pz.error(), got.size()); return 2; }
            if (n == 0 && pz.done()) break;
        }
        double b = now();
        const bool same = got.size() == ref.size() && memcmp(got.data(), ref.data(), ref.size()) == 0;
        printf("%s: %zu -> %zu bytes  %s  zlib %.3fs (%.0f MB/s)  parallel(%d) %.3fs (%.0f MB/s)\n", path, (size_t)st.st_size, ref.size(),
               same ? "IDENTICAL" : "DIFFERENT", t1 - t0, ref.size() / (t1 - t0) / 1e6, threads, b - a, ref.size() / (b - a) / 1e6);
        return same ? 0 : 1;
    }
    snk::GzipInflate z;
    z.init(in, st.st_size);
    // output through a sliding buffer [HIST | chunk], as the reader does
    std::vector<uint8_t> buf(snk::GzipInflate::HIST + chunk), got;
    got.reserve(ref.size());
    double t2 = now();
    size_t hist = 0;
    for (;;) {
        size_t n = z.run(buf.data() + snk::GzipInflate::HIST, chunk);
        if (z.error()) { printf("ERROR %s after %zu bytes\n", z.error(), got.size()); return 2; }
        got.insert(got.end(), buf.begin() + snk::GzipInflate::HIST, buf.begin() + snk::GzipInflate::HIST + n);
        if (n == 0 && z.done()) break;
        // keep the last HIST bytes in front
        const size_t keep = n >= (size_t)snk::GzipInflate::HIST ? (size_t)snk::GzipInflate::HIST : n;
        if (n >= (size_t)snk::GzipInflate::HIST) memmove(buf.data(), buf.data() + n, snk::GzipInflate::HIST);
        else { memmove(buf.data(), buf.data() + n, snk::GzipInflate::HIST); }
        (void)keep; (void)hist;
    }
    double t3 = now();
    {   // decode-only timing (no copy out)
        snk::GzipInflate z2; z2.init(in, st.st_size); if (getenv("NOCRC")) z2.set_verify_crc(false);
        double a = now(); size_t tot = 0;
        for (;;) { size_t n = z2.run(buf.data() + snk::GzipInflate::HIST, chunk); tot += n; if (z2.error() || (n == 0 && z2.done())) break; memmove(buf.data(), buf.data() + n, snk::GzipInflate::HIST); }
        double b = now();
        printf("   decode-only: %.3fs (%.0f MB/s)\n", b - a, tot / (b - a) / 1e6);
    }
    const bool same = got.size() == ref.size() && memcmp(got.data(), ref.data(), ref.size()) == 0;
    printf("%s: %zu -> %zu bytes  %s  zlib %.3fs (%.0f MB/s)  ours %.3fs (%.0f MB/s)\n", path, (size_t)st.st_size, ref.size(), same ? "IDENTICAL" : "DIFFERENT",
           t1 - t0, ref.size() / (t1 - t0) / 1e6, t3 - t2, ref.size() / (t3 - t2) / 1e6);
    return same ? 0 : 1;
}
