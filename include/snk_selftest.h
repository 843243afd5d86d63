/* snk_selftest.h -- device self-tests of library-internal primitives (test hooks, not part of the
 * drop-in boundary of include/snk_filter.h; the reference has no counterpart).  Test infrastructure
 * calls them through the same shared library so that the code under test is the code that ships. */
#ifndef SNK_SELFTEST_H
#define SNK_SELFTEST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 64 x 64 bit-matrix transpose across the lanes of a wave (csrc/snk_bittr.hip.h, the lane = position ->
 * lane = read hand-over of the tiled kernel).  in: n_matrices x 64 lanes x 2 words (host memory, lane p =
 * bits r); out: n_matrices x 64 x 2 words (lane r = bits p); out_lo: n_matrices x 64 words, the
 * half-work variant (bits p < 32).  Returns 0, or a negative SNK error code (snk_last_error()). */
int snk_selftest_bit_transpose(int device, const uint32_t *in, int n_matrices, uint32_t *out, uint32_t *out_lo);

#ifdef __cplusplus
}
#endif
#endif
