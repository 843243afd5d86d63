// snk_crc32.h -- CRC-32 (IEEE 802.3, the gzip one) by carry-less multiplication.
//
// zlib's crc32_z() (table driven) costs about a tenth of this CLI's host CPU time with .gz input and output: every
// byte of FASTQ passes it once on the way in and once on the way out.  x86 has PCLMULQDQ: the classic folding scheme
// (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel 2009) keeps four
// 128-bit accumulators, folds 64 bytes per step with one pair of multiplications each, folds the four into one, and
// finishes with a Barrett reduction.  Same function as zlib's: snk::crc32_fast(crc, p, n) == crc32_z(crc, p, n) for every input
// (tests/test_deflate.py::test_crc32_matches_zlib); without PCLMUL / SSE4.1 at run time it IS zlib's.
#ifndef SNK_CRC32_H
#define SNK_CRC32_H
#include <stddef.h>
#include <stdint.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace snk {

#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_clmul(uint32_t crc, const uint8_t *buf, size_t len) {
    // len >= 64, a multiple of 16.  Constants for the reflected polynomial 0xEDB88320:
    //   k1 = x^(4*128+32) mod P, k2 = x^(4*128-32) mod P, k3 = x^(128+32) mod P, k4 = x^(128-32) mod P, k5 = x^64 mod P, mu, P
    const __m128i k1k2 = _mm_set_epi64x(0x00000001c6e41596, 0x0000000154442bd4);
    const __m128i k3k4 = _mm_set_epi64x(0x00000000ccaa009e, 0x00000001751997d0);
    const __m128i k5k0 = _mm_set_epi64x(0x0000000000000000, 0x0000000163cd6124);
    const __m128i poly = _mm_set_epi64x(0x00000001f7011641, 0x00000001db710641);
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = k1k2;
    buf += 64;
    len -= 64;
    while (len >= 64) {                                  // fold 64 bytes per step
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
        x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
        x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
        y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
        y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64;
        len -= 64;
    }
    x0 = k3k4;                                           // four accumulators into one
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {                                  // the remaining whole 16-byte blocks
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16;
        len -= 16;
    }
    // 128 -> 64 bits
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = k5k0;
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    // Barrett reduction 64 -> 32 bits
    x0 = poly;
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
inline bool crc32_have_clmul() {
    static const bool ok = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    return ok;
}
#endif

// crc32_z() of zlib, faster
inline uint32_t crc32_fast(uint32_t crc, const uint8_t *p, size_t n) {
#if defined(__x86_64__)
    if (n >= 256 && crc32_have_clmul()) {
        const size_t body = n & ~(size_t)15;             // whole 16-byte blocks, at least 64
        crc = ~crc32_clmul(~crc, p, body);
        p += body;
        n -= body;
    }
#endif
    return n ? (uint32_t)crc32_z(crc, p, n) : crc;
}

}  // namespace snk
#endif
