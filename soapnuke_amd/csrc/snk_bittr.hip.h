// 64 x 64 bit-matrix transpose across the lanes of one wave (gfx950).
//
// In:  lane p holds 64 bits r = 0..63 (lo: r < 32, hi: r >= 32).
// Out: lane r holds 64 bits p = 0..63 (lo: p < 32, hi: p >= 32).
//
// Six butterfly stages swap bit k of the lane index with bit k of the bit index:
//   k = 32  v_permlane32_swap (lanes 32..63 of lo <-> lanes 0..31 of hi): one instruction
//   k = 16  v_permlane16_swap of the register with itself puts both lanes' words in both lanes,
//           one v_perm_b32 picks the halves
//   k = 8..1  partner word by DPP, rotated into place, merged with v_bfi
// 1 + 2*3 + 4*2*3 (+2 for the two-instruction lane^4 exchange) VALU for the whole matrix, where
// 64 ballots + 128 v_writelane did the same job before (tiled kernel, phase 1 -> phase 2 hand-over).
#pragma once
#include <hip/hip_runtime.h>

namespace snk {

// keep[K]: the bits a lane keeps of its own word in stage K (even lane: the bits whose index has bit K clear; odd lane: the others),
// rot[K]: the rotation that brings the partner's bits into the other places.  Lane constants: made once per transpose and opaque
// to the compiler -- which otherwise sees two selects of complementary constants for `keep` and `~keep`, does not know them for
// complements, and spends v_and + v_and + v_or where one v_bfi_b32 does it.
template <int K>
__device__ __forceinline__ unsigned bt_keep(int lane) {
    constexpr unsigned MK = K == 8 ? 0x00FF00FFu : K == 4 ? 0x0F0F0F0Fu : K == 2 ? 0x33333333u : 0x55555555u;
    unsigned keep = (lane & K) ? ~MK : MK;
    asm("" : "+v"(keep));            // (not volatile: the same lane gives the same mask, the compiler may keep one copy)
    return keep;
}
template <int K>
__device__ __forceinline__ unsigned bt_stage(unsigned x, int lane, unsigned keep) {
    const bool odd = (lane & K) != 0;
    int t;
    // (mov_dpp: every lane has a source in these patterns -- no `old` value to materialise in front of each of them)
    if (K == 8) t = __builtin_amdgcn_mov_dpp((int)x, 0x128, 0xF, 0xF, false);                  // row_ror:8
    else if (K == 4) {
        t = __builtin_amdgcn_mov_dpp((int)x, 0x104, 0xF, 0x5, false);                          // row_shl:4 -> banks 0,2 read lane+4
        t = __builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xF, 0xA, false);                    // row_shr:4 -> banks 1,3 read lane-4
    } else if (K == 2) t = __builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, false);            // quad_perm [2,3,0,1]
    else t = __builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, false);                          // quad_perm [1,0,3,2]
    // even lane: partner << K into the bits whose index has bit K set; odd lane: partner >> K into the others
    // (a rotate does both: the wrapped-around bits fall under the kept side of the mask)
    const unsigned y = __builtin_amdgcn_alignbit((unsigned)t, (unsigned)t, odd ? (unsigned)K : (unsigned)(32 - K));
    return (x & keep) | (y & ~keep);                                                          // v_bfi_b32
}

__device__ __forceinline__ unsigned bt_stage16(unsigned x, int lane) {
    auto ab = __builtin_amdgcn_permlane16_swap(x, x, false, false);   // [0]: word of the even-row lane, [1]: of the odd-row lane
    return __builtin_amdgcn_perm(ab[1], ab[0], (lane & 16) ? 0x07060302u : 0x05040100u);
}

__device__ __forceinline__ unsigned bt_low_stages(unsigned x, int lane) {
    x = bt_stage16(x, lane);
    x = bt_stage<8>(x, lane, bt_keep<8>(lane));
    x = bt_stage<4>(x, lane, bt_keep<4>(lane));
    x = bt_stage<2>(x, lane, bt_keep<2>(lane));
    return bt_stage<1>(x, lane, bt_keep<1>(lane));
}

__device__ __forceinline__ void bit_transpose64(unsigned &lo, unsigned &hi, int lane) {
    auto s = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
    lo = bt_low_stages(s[0], lane);
    hi = bt_low_stages(s[1], lane);
}

// only the first 32 input lanes matter (output bits p < 32): half the work
__device__ __forceinline__ unsigned bit_transpose64_lo(unsigned lo, unsigned hi, int lane) {
    auto s = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
    return bt_low_stages(s[0], lane);
}

}  // namespace snk
