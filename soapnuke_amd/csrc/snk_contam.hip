// snk_contam.hip -- contaminant screening of a batch: include_contam / include_global_contam of every
// read, i.e. the verdicts of hasContam() (src/read_filter.cpp:507-603) and global_contam_pos() (:961-1062; only
// ">= 0" is ever used, :189-248), written as one byte per pair for the wave-tiled kernel.
//
// Bit-parallel, lane = read: every lane turns its read into bit planes over the positions (X[k] bit p = read[p] ==
// "ACGT"[k], XN bit p = read[p] == 'N'), the contaminant is a set of uniform 64-bit letter masks, and all the
// alignments of a contaminant against a read are treated at once, one bit per alignment offset:
//
//   hasContam   A hit needs the mismatches among the first T-1 cells of the alignment (T = its run threshold) to stay
//               within its budget: a run of T matches cannot complete earlier, and without a run all mismatches count.
//               So the first `scr` cells are counted for every offset at once in unary counter planes (one funnel shift
//               and 2*NC logic ops per contaminant character and plane word); the few surviving offsets are decided
//               exactly, one per lane and trip: 64 match bits of the alignment, then a walk over its at most budget+1
//               first mismatches -- the run between two mismatches is a popcount of the match bits (an 'N' in the read
//               is neither match nor mismatch, so it drops out of both).
//   global      A hit of the reference's carried scoring window (snk_common.hip.h) implies min_match_len consecutive cells
//               of one lay with <= mismatch_number mismatches (or nearly that at the read end, see gcontam_bits).  The
//               mismatches of the last cells are kept as a bit-sliced binary counter per offset (biased, so that its top
//               plane is the verdict), sliding along the contaminant; the offsets with such a stretch are decided exactly by the window walk (gc_walk) on the 64
//               equality bits of their lay.
//
// What the bit paths do not cover (contaminants over 64 characters or with anything but ACGTN, reads shorter than the
// contaminant) goes to the sequential matchers of snk_common.hip.h, per lane.  Reads over 256 nt: the same bit paths block by
// block on the plane store of the long-read path (snk_long_contam_kernel below).
#include <hip/hip_runtime.h>
#include "snk_common.hip.h"
#include "snk_planes.hip.h"

using namespace snk;

#ifndef SNK_CWAVES
#define SNK_CWAVES 3         // waves per SIMD the register allocation aims at (168 VGPRs: 3 workgroups of 46 KB LDS per CU; 2 -> 3: 9.0 -> 7.65 ms)
                             // -- up to 160 positions; the 8-word instance (PE250) spills 293 registers at that cap and none at 248: two waves there
#endif
#ifndef SNK_CABL
#define SNK_CABL 0          // ablation builds (tools/ab_contam.sh): 1 no head section, 2 no middle/tail decisions, 3 no counting screen, 4 no planes
#endif

namespace {

// the descriptors as the kernels read them: through the constant address space, so that every field with a uniform index is a
// scalar load into SGPRs (through a generic reference the compiler took them for per-lane values: flat loads into VGPRs, 64-bit
// VALU shifts of the letter masks, a global round trip per trip of the head loop, loops run under EXEC masks)
typedef __attribute__((address_space(4))) DevContam CDevContam;
typedef __attribute__((address_space(4))) DevGContam CDevGContam;
typedef __attribute__((address_space(4))) DevParams CDevParams;

__device__ __forceinline__ u32 lowmask32(int n) { return n <= 0 ? 0u : (n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u)); }
__device__ __forceinline__ u64 lowmask64(int n) { return n <= 0 ? 0ull : (n >= 64 ? ~0ull : ((1ull << n) - 1ull)); }
__device__ __forceinline__ u64 cat64(u32 hi, u32 lo) { return ((u64)hi << 32) | lo; }

// 0x80 in every byte of x that is zero (exact)
__device__ __forceinline__ u32 zero_bytes(u32 x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); }
// bits 7, 15, 23, 31 -> bits 0..3
__device__ __forceinline__ u32 pack4(u32 z) { return ((z >> 7) | (z >> 14) | (z >> 21) | (z >> 28)) & 0xFu; }

// The planes of one read from its row (LDS, dword-aligned): X[k] = exactly "ACGT"[k], XN = exactly 'N'; zero past len.
template <int NW>
__device__ __forceinline__ void build_planes(const uint8_t *row, int pitch, int len, u32 (&X)[4][NW], u32 (&XN)[NW]) {
    const u32 *r32 = reinterpret_cast<const u32 *>(row);
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        u32 e = 0, c1 = 0, c2 = 0, nn = 0;
        if (32 * w < len) {
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const u32 v = (32 * w + 4 * d < pitch) ? r32[8 * w + d] : 0u;
                const u32 t = v & 0x06060606u;                                         // 2 * code: A 0, C 2, T 4, G 6
                const u32 ex = __builtin_amdgcn_perm(0x00470054u, 0x00430041u, t);     // the letter of that code
                e |= pack4(zero_bytes(ex ^ v)) << (4 * d);
                nn |= pack4(zero_bytes(v ^ 0x4E4E4E4Eu)) << (4 * d);
                c1 |= pack4((v << 6) & 0x80808080u) << (4 * d);
                c2 |= pack4((v << 5) & 0x80808080u) << (4 * d);
            }
        }
        const u32 in = lowmask32(len - 32 * w);
        e &= in;
        X[0][w] = e & ~c1 & ~c2;
        X[1][w] = e & c1 & ~c2;
        X[2][w] = e & c1 & c2;
        X[3][w] = e & ~c1 & c2;
        XN[w] = nn & in;
    }
}

template <int NW>
__device__ __forceinline__ bool any_bit(const u32 (&w)[NW]) {
    u32 o = 0;
#pragma unroll
    for (int j = 0; j < NW; ++j) o |= w[j];
    return o != 0;
}
template <int NW>
__device__ __forceinline__ int lowest_bit(const u32 (&w)[NW]) {
    int p = -1;
#pragma unroll
    for (int j = NW - 1; j >= 0; --j) p = w[j] ? 32 * j + __ffs((int)w[j]) - 1 : p;
    return p;
}
template <int NW>
__device__ __forceinline__ void clear_bit(u32 (&w)[NW], int p) {
#pragma unroll
    for (int j = 0; j < NW; ++j) w[j] &= ~(((p >> 5) == j) ? (1u << (p & 31)) : 0u);
}

// bits [p, p+64) of a plane (per-lane p >= 0); zero past the plane
template <int NW>
__device__ __forceinline__ u64 window64(const u32 (&X)[NW], int p) {
    const int q = p >> 5, sh = p & 31;
    u32 w0 = 0, w1 = 0, w2 = 0;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        w0 = (q == j) ? X[j] : w0;
        w1 = (q + 1 == j) ? X[j] : w1;
        w2 = (q + 2 == j) ? X[j] : w2;
    }
    return cat64(__builtin_amdgcn_alignbit(w2, w1, sh), __builtin_amdgcn_alignbit(w1, w0, sh));
}

// One counting step for every offset at once: x = the cells that do NOT match at contaminant position c = 32*CQ + cr,
// i.e. ~(plane >> c) with ones shifted in; unary counters C[k] = "more than k such cells so far".
template <int NW, int CQ, int NC>
__device__ __forceinline__ void count_step(const u32 (&Pl)[NW], int cr, u32 (&C)[NC][NW]) {
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const u32 lo = (j + CQ < NW) ? Pl[(j + CQ < NW) ? j + CQ : 0] : 0xFFFFFFFFu;
        const u32 hi = (j + CQ + 1 < NW) ? Pl[(j + CQ + 1 < NW) ? j + CQ + 1 : 0] : 0xFFFFFFFFu;
        const u32 x = ~__builtin_amdgcn_alignbit(hi, lo, cr);
#pragma unroll
        for (int k = NC - 1; k >= 1; --k) C[k][j] |= C[k - 1][j] & x;
        C[0][j] |= x;
    }
}
// the steps of the contaminant positions in m64 (uniform)
template <int NW, int NC>
__device__ __forceinline__ void count_steps(const u32 (&Pl)[NW], u64 m64, u32 (&C)[NC][NW]) {
    u32 m = __builtin_amdgcn_readfirstlane((u32)m64);
    while (m) {
        const int c = __ffs((int)m) - 1;
        m &= m - 1;
        count_step<NW, 0, NC>(Pl, c, C);
    }
    m = __builtin_amdgcn_readfirstlane((u32)(m64 >> 32));
    while (m) {
        const int c = __ffs((int)m) - 1;
        m &= m - 1;
        count_step<NW, 1, NC>(Pl, c, C);
    }
}

// Exact outcome of one hasContam() alignment (src/read_filter.cpp:536-547 and its two siblings): m = cells whose
// characters are equal, n = cells where the read has 'N' and the contaminant has not, over ncells cells.  Hit <=> a run
// of T matches completes before the (B+1)-th mismatch, or the alignment ends with at most B mismatches.
template <int NC>                      // NC - 1 = the largest budget there is (0: no bound known)
__device__ __forceinline__ bool contam_accept(u64 m, u64 n, int ncells, int T, int B) {
    const u64 cells = lowmask64(ncells);
    u64 mis = ~m & ~n & cells;
    m &= cells;
    T = max(T, 1);                       // the run is tested at a match only
    const int iters = max(B, 0) + 1;     // B < 0 (INT_MIN thresholds): the first mismatch ends it, and no mismatch is no hit either
    u64 seen = 0;                        // cells up to and including the previous mismatch
    bool hit = false, open = true;
    const int trips = NC > 0 ? NC : iters;           // NC == 0: budgets beyond the counter planes, as many trips as it takes
#pragma unroll
    for (int it = 0; it < trips; ++it) {
        if (open && it < iters) {
            const int z = mis ? __ffsll((long long)mis) - 1 : ncells;
            const u64 upto = lowmask64(z);
            if (__popcll(m & upto & ~seen) >= T) { hit = true; open = false; }
            else if (z >= ncells) { hit = B >= 0; open = false; }
            seen = lowmask64(z + 1);
            mis &= mis - 1;
        }
    }
    return hit;
}

// hasContam() verdict for the lanes with `active` (len >= contaminant length), bit paths only.
// C: the descriptor in global memory (uniform fields -> scalar loads), L: its LDS copy (per-lane table look-ups).
// The screen knows every alignment's own run threshold and budget: an alignment of the middle section is counted over
// its first S - 1 cells against adaMis, one hanging off the read end with r1 = k - edge over its first T(r1) - 1 cells
// against its budget(r1) -- through their monotone envelopes (rT, rk: DevContam), so that "cell c counts for this
// offset" and "this offset's budget is at least b" are prefix masks over the offsets.
template <int NW, int NC, bool BIG = false>      // BIG: budgets of 4 and more exist -- NC == 4 planes count to four, such offsets are never screened out
// do_head (uniform) / do_tail (per lane): the planes start at the read's first character / end at its last one.  A block in the
// middle of a long read (snk_long_contam_kernel) has neither: only the alignments of the middle section exist there.
__device__ bool has_contam_bits(const CDevContam &C, const DevContam &L, const u32 (&X)[4][NW], const u32 (&XN)[NW], int len, bool active,
                                bool do_head = true, bool do_tail = true) {
    const int cl = C.len, edge = C.edge, nC = C.nC;
    const u64 cm0 = C.cm[0], cm1 = C.cm[1], cm2 = C.cm[2], cm3 = C.cm[3], nm = C.nm;
    bool hit = false;
    // ---- head (:523-547): the last k = r1 + edge characters of the contaminant on read[0, k)
    if (nC > 0 && SNK_CABL != 1 && do_head) {
        const u64 x0 = cat64(X[0][1], X[0][0]), x1 = cat64(X[1][1], X[1][0]), x2 = cat64(X[2][1], X[2][0]), x3 = cat64(X[3][1], X[3][0]);
        const u64 xn = cat64(XN[1], XN[0]);
        u64 cand = 0;
        for (int k = edge; k < cl; ++k) {
            const int sh = cl - k, r1 = k - edge;
            const int scr = min(max(C.sm1[r1], 1) - 1, k), bud = max(C.mm[r1], 0);          // uniform
            const u64 ok = (x0 & (cm0 >> sh)) | (x1 & (cm1 >> sh)) | (x2 & (cm2 >> sh)) | (x3 & (cm3 >> sh)) | xn;
            if (__popcll(~ok & lowmask64(scr)) <= bud) cand |= 1ull << k;
        }
        if (!active) cand = 0;
        while (__any(cand != 0)) {
            if (cand) {
                const int k = __ffsll((long long)cand) - 1;
                cand &= cand - 1;
                const int sh = cl - k;
                const u64 m = (x0 & (cm0 >> sh)) | (x1 & (cm1 >> sh)) | (x2 & (cm2 >> sh)) | (x3 & (cm3 >> sh)) | (xn & (nm >> sh));
                if (contam_accept<BIG ? 0 : NC>(m, xn & ~(nm >> sh), k, L.sm1[k - edge], L.mm[k - edge])) { hit = true; cand = 0; }
            }
        }
    }
    // ---- middle (:549-573) and tail (:575-601): contaminant[0, ncells) on read[p, ...), p = 0 .. len - edge
    u32 alive[NW];
    {
        u32 Cn[NC][NW], S[NW], mid[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) {
#pragma unroll
            for (int k = 0; k < NC; ++k) Cn[k][j] = 0;
            mid[j] = lowmask32(len - cl + 1 - 32 * j);                 // the offsets of the middle section
        }
        const int tm1 = max(C.S, 1) - 1;                              // screened cells of a middle alignment
        const u64 sm = SNK_CABL == 3 ? 0ull : lowmask64(C.scr);       // cells screened for any offset at all
        const int tb = len - edge + 1;                                // tail offsets with r1 >= r: p < tb - r
        // ... as a thermometer over the offsets, made once: the mask of a cell is this plane shifted down by the cell's rT (uniform,
        // 0..64) -- one funnel shift per plane word instead of a per-lane mask construction (10 VALU) per word and cell
        u32 TB[NW + 3];
#pragma unroll
        for (int j = 0; j < NW + 3; ++j) TB[j] = j < NW ? lowmask32(tb - 32 * j) : 0u;
        auto steps = [&](u64 m64) {
            for (int h = 0; h < 2; ++h) {
                u32 m = __builtin_amdgcn_readfirstlane((u32)(m64 >> (32 * h)));
                while (m) {
                    const int cr = __ffs((int)m) - 1, c = 32 * h + cr;
                    m &= m - 1;
                    const int rt = __builtin_amdgcn_readfirstlane(L.rT[c]);   // the cell counts for tail alignments with r1 >= rt (LDS copy: no global round trip per cell)
                    const u32 im = c < tm1 ? 0xFFFFFFFFu : 0u;               // ... and for the middle alignments or not
                    const int ws = rt >> 5, bs = rt & 31;
                    u32 tl[NW];
                    if (ws == 0) {
#pragma unroll
                        for (int j = 0; j < NW; ++j) tl[j] = __builtin_amdgcn_alignbit(TB[j + 1], TB[j], bs);
                    } else if (ws == 1) {
#pragma unroll
                        for (int j = 0; j < NW; ++j) tl[j] = __builtin_amdgcn_alignbit(TB[j + 2], TB[j + 1], bs);
                    } else {
#pragma unroll
                        for (int j = 0; j < NW; ++j) tl[j] = TB[j + 2 < NW + 3 ? j + 2 : 0];      // rt == 64
                    }
#pragma unroll
                    for (int j = 0; j < NW; ++j) {
                        const int jl = j + h, jh = j + h + 1;
                        const u32 lo = jl < NW ? S[jl < NW ? jl : 0] : 0xFFFFFFFFu, hi = jh < NW ? S[jh < NW ? jh : 0] : 0xFFFFFFFFu;
                        const u32 msk = (tl[j] & ~mid[j]) | (mid[j] & im);
                        const u32 x = ~__builtin_amdgcn_alignbit(hi, lo, cr) & msk;
#pragma unroll
                        for (int k = NC - 1; k >= 1; --k) Cn[k][j] |= Cn[k - 1][j] & x;
                        Cn[0][j] |= x;
                    }
                }
            }
        };
        // the planes that do NOT count as a mismatch: the letter itself, a read 'N', anything past the read
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const u64 mk = C.cm[b] & sm;
            if (mk) {
#pragma unroll
                for (int j = 0; j < NW; ++j) S[j] = X[b][j] | XN[j] | ~lowmask32(len - 32 * j);
                steps(mk);
            }
        }
        if (nm & sm) {
#pragma unroll
            for (int j = 0; j < NW; ++j) S[j] = XN[j] | ~lowmask32(len - 32 * j);
            steps(nm & sm);
        }
        // an offset is out when its count exceeds its budget: budgets as thermometer planes [budget >= b]
        const int nvalid = ((nC > 0 && do_tail) ? len - edge : len - cl) + 1, mis = max(C.mis, 0);
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            u32 rej = Cn[NC - 1][j];
            if (BIG) rej &= ~((mis >= NC ? mid[j] : 0u) | (lowmask32(tb - C.rk[0] - 32 * j) & ~mid[j]));   // budget >= 4 here (rk[0]: that envelope)
#pragma unroll
            for (int b = 1; b < NC; ++b) {
                const u32 tbp = (mis >= b ? mid[j] : 0u) | (lowmask32(tb - C.rk[b] - 32 * j) & ~mid[j]);
                rej |= Cn[b - 1][j] & ~tbp;
            }
            alive[j] = (active && !hit) ? (lowmask32(nvalid - 32 * j) & ~rej) : 0u;
        }
    }
    while (SNK_CABL != 2 && __any(any_bit(alive))) {
        if (any_bit(alive)) {
            const int p = lowest_bit(alive);
            clear_bit(alive, p);
            const int k = len - p;
            const bool mid = k >= cl;
            const int r1 = mid ? 0 : k - edge;
            const int ncells = mid ? cl : k;
            const int B = mid ? C.mis : L.mm[r1], T = mid ? C.S : L.sm3[r1];
            const u64 n = window64(XN, p);
            const u64 m = (window64(X[0], p) & cm0) | (window64(X[1], p) & cm1) | (window64(X[2], p) & cm2) | (window64(X[3], p) & cm3) | (n & nm);
            if (contam_accept<BIG ? 0 : NC>(m, n & ~nm, ncells, T, B)) {
                hit = true;
#pragma unroll
                for (int j = 0; j < NW; ++j) alive[j] = 0;
            }
        }
    }
    return hit;
}

template <int NW>
__device__ __forceinline__ bool has_contam_bits_nc(const CDevContam &C, const DevContam &L, const u32 (&X)[4][NW], const u32 (&XN)[NW], int len, bool active,
                                   bool do_head = true, bool do_tail = true) {
    const int b = __builtin_amdgcn_readfirstlane(C.bmax);
    if (b <= 0) return has_contam_bits<NW, 1>(C, L, X, XN, len, active, do_head, do_tail);
    if (b == 1) return has_contam_bits<NW, 2>(C, L, X, XN, len, active, do_head, do_tail);
    if (b == 2) return has_contam_bits<NW, 3>(C, L, X, XN, len, active, do_head, do_tail);
    if (b == 3) return has_contam_bits<NW, 4>(C, L, X, XN, len, active, do_head, do_tail);
    return has_contam_bits<NW, 4, true>(C, L, X, XN, len, active, do_head, do_tail);
}

// e = plane >> c for a uniform c in [0, 64): the offsets whose cell at contaminant position c IS that letter; zeros shifted
// in, so anything outside the read counts as a mismatch
template <int NQ>
__device__ __forceinline__ void shifted(const u32 (&P)[NQ], int c, u32 (&e)[NQ]) {
    if (c < 32) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) e[j] = __builtin_amdgcn_alignbit(j + 1 < NQ ? P[j + 1 < NQ ? j + 1 : 0] : 0u, P[j], c);
    } else {
#pragma unroll
        for (int j = 0; j < NQ; ++j)
            e[j] = __builtin_amdgcn_alignbit(j + 2 < NQ ? P[j + 2 < NQ ? j + 2 : 0] : 0u, j + 1 < NQ ? P[j + 1 < NQ ? j + 1 : 0] : 0u, c - 32);
    }
}
template <int NQ>
__device__ __forceinline__ void match_plane(const u32 (&XP)[5][NQ], int letter, int c, u32 (&e)[NQ]) {
    switch (letter) {                                // uniform
    case 0: shifted<NQ>(XP[0], c, e); break;
    case 1: shifted<NQ>(XP[1], c, e); break;
    case 2: shifted<NQ>(XP[2], c, e); break;
    case 3: shifted<NQ>(XP[3], c, e); break;
    default: shifted<NQ>(XP[4], c, e); break;
    }
}

// global_contam_pos() verdict of one strand for the lanes with `active` (len >= contaminant length).
// What a hit takes (snk_common.hip.h has the walk): a window opens on a matching cell and is dead once it holds more than
// mismatch_number mismatches; in the first two sections it only opens with min_match_len cells of the lay in front of it,
// so it either hits when it spans exactly min_match_len cells or is dead by then -- every lay starts from a dead window,
// the lays are independent.  In the last section a window may open on one of the last cells, the lay is abandoned and the
// window goes on at cell 0 of the next lay.  Hence a hit implies
//   (W) min_match_len consecutive cells of one lay with at most mismatch_number mismatches, or
//   (J) the first min_match_len - 1 cells of a lay hanging off the read end with at most mismatch_number mismatches,
// and so it does with Ls = min(min_match_len, 16 + mismatch_number) cells instead.  Both are sliding counts along the lay, kept
// for every offset at once: bit q of the planes = offset p = q - PAD, the count a bit-sliced binary number that takes the
// cell entering the window and gives back the cell leaving it; cells outside the read count as mismatches.  The number kept is
// count + 15 - mismatch_number in five planes, so that "more than mismatch_number" IS the top plane (no comparison, and
// nothing that depends on the budget inside the loop): 14 three-input operations per plane word and cell while the window
// slides, 3 while it fills.  The offsets that pass are decided exactly with the window walk (gc_walk) on the 64 equality bits
// of their lay: one lay for an offset of the first two sections, the whole last section once one of its offsets passes.
template <int NW, int NQ>
// do_head (uniform) / do_tail (per lane) as in has_contam_bits: a block in the middle of a long read has the whole lays only
// (offsets p >= 0 that end inside the block), the lays hanging off the read's start belong to its first block, the last section to
// its final one.
__device__ bool gcontam_bits(const CDevGContam &G, int d, const u32 (&X)[4][NW], const u32 (&XN)[NW], int len, bool active,
                             bool do_head = true, bool do_tail = true) {
    constexpr int NP = NW + 2;                       // plane words: read positions + PAD
    const int cl = __builtin_amdgcn_readfirstlane(G.len), mml = __builtin_amdgcn_readfirstlane(G.min_match_len), mmn = __builtin_amdgcn_readfirstlane(G.mm), PAD = cl - mml;
    const int Ls = min(mml, 16 + mmn);
    const u64 cm0 = G.cm[d][0], cm1 = G.cm[d][1], cm2 = G.cm[d][2], cm3 = G.cm[d][3], nm = G.nm[d];
    auto letter = [&](int c) -> int { return ((cm0 >> c) & 1) ? 0 : ((cm1 >> c) & 1) ? 1 : ((cm2 >> c) & 1) ? 2 : ((cm3 >> c) & 1) ? 3 : 4; };
    u32 W[NQ], J[NQ];
    {
        // planes shifted up by PAD: the cell (offset bit q, contaminant position c) is bit q + c
        u32 XP[5][NP];
        const int wq = PAD >> 5, r = PAD & 31;      // uniform, PAD <= 60
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            u32 T[NP + 1];
            T[0] = 0;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int s = j - (wq ? 1 : 0);
                u32 v = 0;
#pragma unroll
                for (int jj = 0; jj < NW; ++jj) v = (s == jj) ? (b < 4 ? X[b < 4 ? b : 0][jj] : XN[jj]) : v;
                T[j + 1] = v;
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) XP[b][j] = r ? __builtin_amdgcn_alignbit(T[j + 1], T[j], 32 - r) : T[j + 1];
        }
        // count + 15 - mmn in five planes: bit 4 is set exactly while the window holds more than mmn mismatches
        // (Ls <= 16 + mmn, mmn <= 4: the number stays inside 0..31), so "within the budget" is a plane, not a comparison
        u32 cnt[5][NQ], Wn[NQ];
        const int bias = 15 - mmn;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            J[j] = 0;
            Wn[j] = 0xFFFFFFFFu;
#pragma unroll
            for (int b = 0; b < 5; ++b) cnt[b][j] = ((bias >> b) & 1) ? 0xFFFFFFFFu : 0u;
        }
        // one cell: ein = the offsets whose cell entering the window matches, eout = those whose cell leaving it did
        auto slide = [&](const u32 (&ein)[NP], const u32 (&eout)[NP]) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const u32 dec = ein[j] & ~eout[j];
                u32 carry = ein[j] ^ eout[j];        // the offsets whose count changes; up where dec is clear, down where set
#pragma unroll
                for (int b = 0; b < 5; ++b) {
                    const u32 t = cnt[b][j];
                    cnt[b][j] = t ^ carry;
                    carry &= t ^ dec;
                }
            }
        };
        {
            u32 ones[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) ones[j] = 0xFFFFFFFFu;
            for (int c = 0; c < Ls; ++c) {           // the window fills: nothing leaves
                u32 ein[NP];
                match_plane<NP>(XP, letter(c), c, ein);
                slide(ein, ones);
                if (c == Ls - 2) {
#pragma unroll
                    for (int j = 0; j < NQ; ++j) J[j] = ~cnt[4][j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NQ; ++j) Wn[j] = cnt[4][j];
        for (int c = Ls; c < cl; ++c) {
            u32 ein[NP], eout[NP];
            match_plane<NP>(XP, letter(c), c, ein);
            match_plane<NP>(XP, letter(c - Ls), c - Ls, eout);
            slide(ein, eout);
#pragma unroll
            for (int j = 0; j < NQ; ++j) Wn[j] &= cnt[4][j];
        }
#pragma unroll
        for (int j = 0; j < NQ; ++j) W[j] = ~Wn[j];
    }
    // offsets that exist: p = -PAD .. len - mml; the lays hanging off the end: p = len - cl + 1 .. len - mml
    bool tail = false;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const u32 all = active ? lowmask32(len - mml + PAD + 1 - 32 * j) : 0u, front = lowmask32(len - cl + PAD + 1 - 32 * j);
        tail |= do_tail && ((W[j] | J[j]) & all & ~front) != 0;
        W[j] &= all & front & (do_head ? 0xFFFFFFFFu : ~lowmask32(PAD - 32 * j));
    }
    const int tms = -200 * mmn, lower = (mml - mmn) + tms;
    bool hit = false;
    const u64 x0 = cat64(X[0][1], X[0][0]), x1 = cat64(X[1][1], X[1][0]), x2 = cat64(X[2][1], X[2][0]), x3 = cat64(X[3][1], X[3][0]);
    const u64 xn = cat64(XN[1], XN[0]);
    auto lay_bits = [&](int p) -> u64 {              // equality bits of the lay at read offset p >= 0: bit j = read[p + j] == contaminant[j]
        return (window64(X[0], p) & cm0) | (window64(X[1], p) & cm1) | (window64(X[2], p) & cm2) | (window64(X[3], p) & cm3) | (window64(XN, p) & nm);
    };
    while (__any(any_bit(W))) {                       // the first two sections: one lay per offset, from a dead window
        if (any_bit(W)) {
            const int q = lowest_bit(W);
            clear_bit(W, q);
            const int p = q - PAD;
            u64 e;
            int n = cl;
            if (p >= 0) e = lay_bits(p);
            else {                                    // contaminant[-p + j] on read[j]
                const int sh = -p;
                e = (x0 & (cm0 >> sh)) | (x1 & (cm1 >> sh)) | (x2 & (cm2 >> sh)) | (x3 & (cm3 >> sh)) | (xn & (nm >> sh));
                n = cl - sh;
            }
            GcWindow st = {-1000, 0};
            if (gc_walk(st, [&](int) { return e; }, n, mml, tms, lower, true)) {
                hit = true;
                tail = false;
#pragma unroll
                for (int j = 0; j < NQ; ++j) W[j] = 0;
            }
        }
    }
    if (__any(tail)) {                                // the last section: its lays in order, the window carried along
        GcWindow st = {-1000, 0};
        for (int i = 0; i <= PAD; ++i) {
            if (tail) {
                const u64 e = lay_bits(len - cl + i);
                if (gc_walk(st, [&](int) { return e; }, cl - i, mml, tms, lower, false)) { hit = true; tail = false; }
            }
        }
    }
    return hit;
}

template <int NW>
__device__ __forceinline__ bool gcontam_bits_nq(const CDevGContam &G, int d, const u32 (&X)[4][NW], const u32 (&XN)[NW], int len, int lcap, bool active,
                                bool do_head = true, bool do_tail = true) {
    const int need = __builtin_amdgcn_readfirstlane((lcap - 2 * G.min_match_len + G.len + 32) >> 5);   // words of offsets -PAD .. lcap - mml
    if (need <= NW) return gcontam_bits<NW, NW>(G, d, X, XN, len, active, do_head, do_tail);
    if (need == NW + 1) return gcontam_bits<NW, NW + 1>(G, d, X, XN, len, active, do_head, do_tail);
    return gcontam_bits<NW, NW + 2>(G, d, X, XN, len, active, do_head, do_tail);
}

// One work-item per pair.  The workgroup first copies its 256 rows (coalesced) and the contaminant tables into LDS --
// row stride an odd number of dwords, so the per-lane walks (plane building, the sequential matchers) are free of bank
// conflicts.  NW = plane words (32 positions each); NW == 0: sequential matchers only (reads over 256 nt).
template <int NW>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NW <= 5 ? SNK_CWAVES : 2, NW <= 5 ? SNK_CWAVES : 2))) snk_contam_kernel(const DevParams *Pp, DevBatch B, unsigned char *cf, int stride) {
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) uint8_t, sm)
    const CDevParams &P = *(const CDevParams *)(uintptr_t)Pp;
    const int pe = P.paired ? 1 : 0, tid = threadIdx.x;
    uint8_t *rows = sm;
    DevContam *lct = reinterpret_cast<DevContam *>(sm + (((size_t)256 * stride + 15) & ~(size_t)15));
    DevGContam *lg = reinterpret_cast<DevGContam *>(lct + max(P.n_ct[0], P.n_ct[1]));       // (the launcher sizes the tables by these counts)
    const int n_gct = P.n_gct;
    for (int k = tid; k < (int)(n_gct * sizeof(DevGContam) / 4); k += 256)
        reinterpret_cast<uint32_t *>(lg)[k] = reinterpret_cast<const uint32_t *>(P.gct)[k];
    const long nround = (B.n + 255) / 256 * 256;
    for (long base = (long)blockIdx.x * 256; base < nround; base += (long)gridDim.x * 256) {
        const long i = base + tid;
        const int rows_here = (int)min((long)256, B.n - base);
        int f = 0;
        for (int m = 0; m <= pe; ++m) {
            const int n_ct = P.n_ct[m];
            __syncthreads();
            for (int k = tid; k < (int)(n_ct * sizeof(DevContam) / 4); k += 256)
                reinterpret_cast<uint32_t *>(lct)[k] = reinterpret_cast<const uint32_t *>(P.ct + m * SNK_MAX_CONTAMS)[k];
            const uint8_t *src = B.seq[m] + base * (long)B.pitch;
            const int dwr = B.pitch >> 2;                          // pitch is a multiple of 4 (C ABI)
            {   // the rows are contiguous in global memory (row k / dwr, dword k % dwr of the block), kept without divisions
                int r = tid / dwr, c = tid - r * dwr;
                const int dr = 256 / dwr, dc = 256 - dr * dwr, total = rows_here * dwr;
                const uint32_t *src32 = reinterpret_cast<const uint32_t *>(src);
                for (int k = tid; k < total; k += 256) {
                    *reinterpret_cast<uint32_t *>(rows + (size_t)r * stride + 4 * c) = src32[k];
                    c += dc;
                    r += dr;
                    if (c >= dwr) { c -= dwr; ++r; }
                }
            }
            __syncthreads();
            const bool exists = i < B.n;
            const int len = exists ? min(B.len[m] ? (int)B.len[m][i] : B.fixed_len[m], P.lcap) : 0;
            const uint8_t *row = rows + (size_t)tid * stride;
            int fm = 0;
            if constexpr (NW > 0) {
                u32 X[4][NW], XN[NW];
                build_planes<NW>(row, B.pitch, len, X, XN);
                for (int c = 0; c < n_ct; ++c) {
                    const CDevContam &C = ((const CDevContam *)(uintptr_t)P.ct)[m * SNK_MAX_CONTAMS + c];
                    const bool want = exists && !(fm & 1);
                    const bool bits = C.bits_ok != 0, fast = bits && len >= C.len;
                    if (bits && __any(want && fast) && has_contam_bits_nc<NW>(C, lct[c], X, XN, len, want && fast)) fm |= 1;
                    if (want && !fast && has_contam_seq(row, len, lct[c]) >= 0) fm |= 1;
                }
                for (int c = 0; c < n_gct; ++c) {
                    const CDevGContam &G = ((const CDevGContam *)(uintptr_t)P.gct)[c];
                    for (int d = 0; d < 2; ++d) {
                        const bool want = exists && !(fm & 2);
                        const bool bits = G.bits_ok != 0, fast = bits && len >= G.len;
                        if (bits && __any(want && fast) && gcontam_bits_nq<NW>(G, d, X, XN, len, P.lcap, want && fast)) fm |= 2;
                        if (want && !fast && global_contam_hit(row, len, lg[c].seq[d], lg[c].len, lg[c].min_match_len, lg[c].mm)) fm |= 2;
                    }
                }
            } else {
                if (exists) fm = contam_flags(lct, n_ct, lg, n_gct, row, len);
            }
            f |= fm << (2 * m);
        }
        if (i < B.n) cf[i] = (unsigned char)f;
    }
}

// Reads of 257..1024 positions: the same bit paths on BLOCKS of a read, from the plane store the long-read prep kernel wrote
// (snk_planes.hip.h): planes of 320 positions serve the 256 alignment offsets of a block and the 64 positions a contaminant reaches
// past them.  Contaminant alignments that hang off the read's start belong to its first block, those hanging off its end to its
// final block (which is cut so that it ends with the read and holds at least 64 positions), whole alignments to the block of
// their offset: only the verdict is ever used, so a hit in any block is the hit.  One work-item per pair; reads shorter than a
// contaminant and contaminants outside the bit paths take the sequential matchers on the read's row, per lane.
// (allocated for two waves per SIMD: 264 registers uncapped, i.e. one wave per SIMD for a kernel that waits on plane loads; at 256 six
// registers are spilled -- the same trade as the 8-word instance of snk_contam_kernel; static figures, not timed)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) snk_long_contam_kernel(const DevParams *Pp, DevBatch B, unsigned char *cf, const u32 *planes, long ngroups, int nquads) {
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) uint8_t, sm)
    const CDevParams &P = *(const CDevParams *)(uintptr_t)Pp;
    const int pe = P.paired ? 1 : 0, tid = threadIdx.x;
    DevContam *lct = reinterpret_cast<DevContam *>(sm);
    DevGContam *lg = reinterpret_cast<DevGContam *>(lct + max(P.n_ct[0], P.n_ct[1]));
    const int n_gct = P.n_gct;
    for (int k = tid; k < (int)(n_gct * sizeof(DevGContam) / 4); k += 256)
        reinterpret_cast<uint32_t *>(lg)[k] = reinterpret_cast<const uint32_t *>(P.gct)[k];
    const long nround = (B.n + 255) / 256 * 256;
    for (long base = (long)blockIdx.x * 256; base < nround; base += (long)gridDim.x * 256) {
        const long i = base + tid;
        const bool exists = i < B.n;
        int f = 0;
        for (int m = 0; m <= pe; ++m) {
            const int n_ct = P.n_ct[m];
            __syncthreads();
            for (int k = tid; k < (int)(n_ct * sizeof(DevContam) / 4); k += 256)
                reinterpret_cast<uint32_t *>(lct)[k] = reinterpret_cast<const uint32_t *>(P.ct + m * SNK_MAX_CONTAMS)[k];
            __syncthreads();
            const int len = exists ? min(B.len[m] ? (int)B.len[m][i] : B.fixed_len[m], P.lcap) : 0;
            const uint8_t *row = B.seq[m] + (exists ? i : 0) * (long)B.pitch;
            const u32 *grp = planes + ((long)m * ngroups + (i >> 6)) * nquads * PL_QUAD_DWORDS;
            int fm = 0;
            // the sequential share first: reads shorter than 64 (no block decomposition), contaminants the bit paths do not cover
            const bool blocks = exists && len >= 64;
            for (int c = 0; c < n_ct; ++c)
                if (exists && !(fm & 1) && (!blocks || !lct[c].bits_ok) && has_contam_seq(row, len, lct[c]) >= 0) fm |= 1;
            for (int c = 0; c < n_gct; ++c)
                for (int d = 0; d < 2; ++d)
                    if (exists && !(fm & 2) && (!blocks || !lg[c].bits_ok) &&
                        global_contam_hit(row, len, lg[c].seq[d], lg[c].len, lg[c].min_match_len, lg[c].mm)) fm |= 2;
            bool through = false;
            for (int p0 = 0; __any(blocks && !through); p0 += PL_BLK) {
                const int rem = len - p0;
                const bool here = blocks && !through, final = rem <= PL_VLEN;
                const int vlen = here ? (final ? rem : PL_VLEN) : 0;
                u32 W[PL_PLANES][12], X[4][PL_NW], XN[PL_NW];
                plane_block_words(grp, nquads, (int)(i & 63), here ? p0 : 0, vlen, W);
#pragma unroll
                for (int w = 0; w < PL_NW; ++w) {
                    const u32 in = lowmask32(vlen - 32 * w);
                    X[0][w] = W[0][w] & in; X[1][w] = W[1][w] & in; X[2][w] = W[2][w] & in; X[3][w] = W[3][w] & in;
                    XN[w] = W[4][w] & in;
                }
                for (int c = 0; c < n_ct; ++c) {
                    const CDevContam &C = ((const CDevContam *)(uintptr_t)P.ct)[m * SNK_MAX_CONTAMS + c];
                    const bool want = here && !(fm & 1) && C.bits_ok != 0;
                    if (__any(want) && has_contam_bits_nc<PL_NW>(C, lct[c], X, XN, vlen, want, p0 == 0, final)) fm |= 1;
                }
                for (int c = 0; c < n_gct; ++c) {
                    const CDevGContam &G = ((const CDevGContam *)(uintptr_t)P.gct)[c];
                    for (int d = 0; d < 2; ++d) {
                        const bool want = here && !(fm & 2) && G.bits_ok != 0;
                        if (__any(want) && gcontam_bits_nq<PL_NW>(G, d, X, XN, vlen, PL_VLEN + 1, want, p0 == 0, final)) fm |= 2;
                    }
                }
                through |= here && final;
            }
            f |= fm << (2 * m);
        }
        if (exists) cf[i] = (unsigned char)f;
    }
}

}  // namespace

// the long-read variant: verdicts from the plane store (snk_long.hip wrote it on the same stream)
void snk_launch_long_contam(const DevParams *dp, const DevBatch &b, unsigned char *cf, int n_ct, int n_gct, const unsigned *planes, int nquads, void *stream) {
    if (b.n <= 0) return;
    long blocks = (b.n + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    const size_t shmem = (size_t)n_ct * sizeof(DevContam) + (size_t)n_gct * sizeof(DevGContam) + 16;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)snk_long_contam_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(snk_long_contam_kernel, dim3((unsigned)blocks), dim3(256), shmem, (hipStream_t)stream, dp, b, cf, (const u32 *)planes, (b.n + 63) / 64, nquads);
}

void snk_launch_contam(const DevParams *dp, const DevBatch &b, unsigned char *cf, int lcap, int n_ct, int n_gct, void *stream) {
    if (b.n <= 0) return;
    long blocks = (b.n + 255) / 256;
    int sd = (b.pitch + 3) / 4;
    if (!(sd & 1)) ++sd;                                            // odd dword stride
    const int stride = sd * 4;
    const size_t shmem = (((size_t)256 * stride + 15) & ~(size_t)15) + (size_t)n_ct * sizeof(DevContam) + (size_t)n_gct * sizeof(DevGContam);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)snk_contam_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)snk_contam_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (lcap <= 160) hipLaunchKernelGGL(snk_contam_kernel<5>, dim3((unsigned)blocks), dim3(256), shmem, (hipStream_t)stream, dp, b, cf, stride);
    else hipLaunchKernelGGL(snk_contam_kernel<8>, dim3((unsigned)blocks), dim3(256), shmem, (hipStream_t)stream, dp, b, cf, stride);
    // (the caller runs this pass for lcap <= 256 only, snk_filter.cpp: longer reads take snk_launch_long_contam on the plane store;
    //  the sequential-only instance <0> that used to stand here had been unreachable since round 3)
}
