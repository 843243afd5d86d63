// snk_planes.hip.h -- the plane store of a batch of long reads (257..1024 positions): snk_long_prep_kernel (snk_long.hip) writes
// it, the decide kernel (snk_long.hip) and the block-wise contaminant kernel (snk_contam.hip) read it.  For every group of 64
// consecutive reads of a mate: nquads quads of 4 plane words (32 positions each; nquads = the capacity in 128-position units,
// at most 8) x 5 planes (A C G T N) x 64 reads x 16 bytes, so that the 64 lanes of a wavefront -- 64 consecutive reads -- fetch
// one quad of one plane as one contiguous kilobyte.
#pragma once
#include "snk_common.hip.h"

namespace snk {
namespace {

constexpr int PL_QUADS = 8, PL_PLANES = 5;                                  // (PL_QUADS: the most a store has)
constexpr int PL_QUAD_DWORDS = PL_PLANES * 64 * 4;                          // 5 KB per quad of a group
__host__ __device__ inline int plane_quads(int lcap) { return lcap <= 128 ? 1 : (lcap >= 128 * PL_QUADS ? PL_QUADS : (lcap + 127) / 128); }
constexpr int PL_BLK = 256;        // candidate offsets per block of a long read
constexpr int PL_NW = 10;          // plane words of a block: 256 offsets + 64 positions behind them
constexpr int PL_VLEN = 32 * PL_NW - 1;   // characters a non-final block shows to a search

__host__ __device__ inline long plane_store_dwords(long n, int mates, int nquads) { return (long)mates * ((n + 63) / 64) * nquads * PL_QUAD_DWORDS; }

typedef u32 pl_v4u32 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) pl_v4u32 *pl_gl_uint4_p;

// the 12 words [p0 / 32, p0 / 32 + 12) of the five planes of read r of a group (p0 a multiple of 256); quads past the store or
// wholly past vlen are not fetched (zeros)
__device__ __forceinline__ void plane_block_words(const u32 *grp, int nquads, int r, int p0, int vlen, u32 (&W)[PL_PLANES][12]) {
    const int q0 = p0 >> 7;
#pragma unroll
    for (int k = 0; k < PL_PLANES; ++k) {
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) {
            pl_v4u32 v = {0, 0, 0, 0};
            if (q0 + qq < nquads && 128 * qq < vlen) v = *(pl_gl_uint4_p)(grp + ((long)((q0 + qq) * PL_PLANES + k) * 64 + r) * 4);
            W[k][4 * qq] = v.x; W[k][4 * qq + 1] = v.y; W[k][4 * qq + 2] = v.z; W[k][4 * qq + 3] = v.w;
        }
    }
}

}  // namespace
}  // namespace snk
