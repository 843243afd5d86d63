// snk_adapter_bits.hip.h -- the bit-sliced adapter search (A2 adapter_pos, src/read_filter.cpp:707-790) on the bit planes of
// one read per lane: screening of all candidate offsets at once with unary mismatch counters, exact closed-form decision
// of the survivors in the reference's order.  Shared by the wave-tiled kernel (snk_tiled.hip: planes of a whole read,
// NW <= 8 words) and the long-read kernel (snk_long.hip: planes of a 320-position block of the read).
#pragma once
#include "snk_common.hip.h"

#ifndef SNK_ABL
#define SNK_ABL 0
#endif
#ifndef SNK_SCREEN_BIN
#define SNK_SCREEN_BIN 1     // 0: unary counter planes for every budget (A/B builds)
#endif

namespace snk {
namespace {

// a TileAdapter read through the constant address space (uniform address: scalar loads, hoisted like kernel arguments)
typedef __attribute__((address_space(4))) TileAdapter CTileAdapter;

// the n lowest bits (n <= 0: none, n >= 32: all): clamp, 64-bit shift, low word - 1 (1 << 32 leaves 0 there, 0 - 1 = all ones):
// v_med3_i32 + v_lshlrev_b64 + v_add_u32 where two compares, two selects, a shift and an add stood (round 5: the region masks of the
// adapter screen are ~35 of these per adapter and tile-mate)
__device__ __forceinline__ u32 lowmask32(int n) { return (u32)(1ull << min(max(n, 0), 32)) - 1u; }
__device__ __forceinline__ u64 lowmask64(int n) { return n <= 0 ? 0ull : (n >= 64 ? ~0ull : ((1ull << n) - 1ull)); }

// Exact outcome of one alignment of the reference's scan (src/read_filter.cpp:726-741 and
// its two siblings): m bit c = adapter/read characters equal at step c, n steps.
// accept <=> a run of S matches completes before the (budget+1)-th mismatch, or fewer than
// budget+1 mismatches occur at all.  budget < 0 (INT_MIN): only the run can accept.
__device__ inline bool accept_exact(u64 m, int n, int S, int budget) {
    const u64 nm = lowmask64(n);
    m &= nm;
    const u64 z = ~m & nm;
    if (budget < 0) return S <= n && (z & lowmask64(S)) == 0;
    if (__popcll(z) <= budget) return true;
    u64 t = z;
    for (int i = 0; i < budget; ++i) t &= t - 1;
    const int kz = __ffsll((long long)t) - 1;          // position of the (budget+1)-th mismatch
    if (S > kz) return false;
    u64 r = m & lowmask64(kz);
    int have = 1;
    while (have < S) {                                  // r bit i = ones at i..i+have-1
        const int st = min(have, S - have);
        r &= r >> st;
        have += st;
    }
    return r != 0;
}

// The same outcome for ONE alignment walked character by character (adapters of more than 64 characters, whose survivors of the
// screen do not fit the 64 match bits of accept_exact): adapter[a0 + c] against read[p + c], c = 0 .. n-1; outside the read: '\0'.
__device__ inline bool accept_seq(const uint8_t *ada, int a0, const uint8_t *s, int len, int p, int n, int S, int budget) {
    int mis = 0, run = 0;
    for (int c = 0; c < n; ++c) {
        if ((int)ada[a0 + c] == rdc(s, len, p + c)) { if (++run >= S) return true; }
        else { ++mis; run = 0; if (mis > budget) break; }
    }
    return mis <= budget;
}

template <int NW>
__device__ __forceinline__ int lowest_bit(const u32 (&w)[NW]) {
    int p = -1;
#pragma unroll
    for (int j = NW - 1; j >= 0; --j) p = w[j] ? 32 * j + __ffs((int)w[j]) - 1 : p;
    return p;
}
template <int NW>
__device__ __forceinline__ int highest_bit(const u32 (&w)[NW]) {
    int p = -1;
#pragma unroll
    for (int j = 0; j < NW; ++j) p = w[j] ? 32 * j + 31 - __clz((int)w[j]) : p;
    return p;
}
template <int NW>
__device__ __forceinline__ void clear_bit(u32 (&w)[NW], int p) {
#pragma unroll
    for (int j = 0; j < NW; ++j) w[j] &= ~(((p >> 5) == j) ? (1u << (p & 31)) : 0u);
}
template <int NW>
__device__ __forceinline__ bool any_bit(const u32 (&w)[NW]) {
    u32 o = 0;
#pragma unroll
    for (int j = 0; j < NW; ++j) o |= w[j];
    return o != 0;
}

// bits [p, p+64) of a plane (per-lane p); bits past the plane read as `fill`
template <int NW>
__device__ __forceinline__ u64 window64(const u32 (&X)[NW], int p, u32 fill) {
    const int q = p >> 5, sh = p & 31;
    u32 w0 = fill, w1 = fill, w2 = fill;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        w0 = (q == j) ? X[j] : w0;
        w1 = (q + 1 == j) ? X[j] : w1;
        w2 = (q + 2 == j) ? X[j] : w2;
    }
    const u32 lo = __builtin_amdgcn_alignbit(w1, w0, sh);
    const u32 hi = __builtin_amdgcn_alignbit(w2, w1, sh);
    return ((u64)hi << 32) | lo;
}

// number of consecutive set bits going DOWN from position len-1 (0 if bit len-1 is clear);
// returns len when every bit below len is set.
template <int NW>
__device__ __forceinline__ int run_down(const u32 (&X)[NW], int len) {
    int hz = -1;                                         // highest zero below len
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const u32 z = ~X[j] & lowmask32(len - 32 * j);
        hz = z ? 32 * j + 31 - __clz((int)z) : hz;
    }
    return len - 1 - hz;
}
// number of consecutive set bits going UP from position 0; 32*NW when all are set
template <int NW>
__device__ __forceinline__ int run_up(const u32 (&X)[NW]) {
    int p = 32 * NW;
#pragma unroll
    for (int j = NW - 1; j >= 0; --j) p = (~X[j]) ? 32 * j + __ffs((int)~X[j]) - 1 : p;
    return p;
}

// plane >> st (uniform st), zero fill
template <int NW>
__device__ __forceinline__ void shr_plane(u32 (&R)[NW], int st) {
    const int q = st >> 5, r = st & 31;
    u32 T[NW + 1];
#pragma unroll
    for (int j = 0; j <= NW; ++j) T[j] = 0;
#pragma unroll
    for (int qq = 0; qq < NW; ++qq)
        if (q == qq) {                                   // uniform branch, static indices inside
#pragma unroll
            for (int j = 0; j + qq < NW; ++j) T[j] = R[j + qq];
        }
#pragma unroll
    for (int j = 0; j < NW; ++j) R[j] = __builtin_amdgcn_alignbit(T[j + 1], T[j], r);
}

// one screening step: x = ~(plane >> c) (ones shifted in), C_k |= C_{k-1} & x  (NC unary counter planes).
// NC == 3 (budgets up to 2, the default adaMis): the count 0..3 is kept as a saturating BINARY counter in C[0] (low bit) and C[1]
// -- two 3-input operations per word and step instead of three; screen_planes() turns it back into the thermometer
template <int NW, int CQ, int NC>
__device__ __forceinline__ void screen_step(const u32 (&Pl)[NW], int cr, u32 (&C)[NC][NW]) {
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const u32 lo = (j + CQ < NW) ? Pl[(j + CQ < NW) ? j + CQ : 0] : 0xFFFFFFFFu;
        const u32 hi = (j + CQ + 1 < NW) ? Pl[(j + CQ + 1 < NW) ? j + CQ + 1 : 0] : 0xFFFFFFFFu;
        const u32 x = ~__builtin_amdgcn_alignbit(hi, lo, cr);
        if constexpr (NC == 3 && SNK_SCREEN_BIN) {
            const u32 c0 = C[0][j], c1 = C[1][j];
            C[0][j] = (c0 ^ x) | (c1 & c0);          // 00 -> 01 -> 10 -> 11 -> 11
            C[1][j] = c1 | (c0 & x);
        } else {
#pragma unroll
            for (int k = NC - 1; k >= 1; --k) C[k][j] |= C[k - 1][j] & x;
            C[0][j] |= x;
        }
    }
}

// two steps of the binary counter at once (NC == 3): the counter needs both of its old planes for either new one, so a single step
// in place costs a register copy per word (round 4: 22 instead of 15 instructions per step, profiles/r04_isa_budget.md); with two steps
// the intermediate pair lives in temporaries and the second step writes the planes back -- 6 instructions per word and two steps
template <int NW, int CQ>
__device__ __forceinline__ void screen_step2_bin(const u32 (&Pl)[NW], int cr1, int cr2, u32 (&C)[3][NW]) {
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const u32 lo = (j + CQ < NW) ? Pl[(j + CQ < NW) ? j + CQ : 0] : 0xFFFFFFFFu;
        const u32 hi = (j + CQ + 1 < NW) ? Pl[(j + CQ + 1 < NW) ? j + CQ + 1 : 0] : 0xFFFFFFFFu;
        const u32 x1 = ~__builtin_amdgcn_alignbit(hi, lo, cr1), x2 = ~__builtin_amdgcn_alignbit(hi, lo, cr2);
        const u32 c0 = C[0][j], c1 = C[1][j];
        const u32 t0 = (c0 ^ x1) | (c1 & c0), t1 = c1 | (c0 & x1);
        C[0][j] = (t0 ^ x2) | (t1 & t0);
        C[1][j] = t1 | (t0 & x2);
    }
}

// Bit-sliced screening of the candidates p = 0 .. len-edge of phases B and C over the first S-1 adapter
// characters with NC unary mismatch-counter planes.
// NC = largest budget + 1 counter planes, at most 4: offsets whose own budget is 4 or more are not screened out by the count.
template <int NW, bool FULL, int NC, class AD>
__device__ __forceinline__ void screen_planes(const AD &A, const u32 (&NX)[4][NW], const u32 (&NXN)[NW], int len, bool done,
                                              u32 (&aliveB)[NW], u32 (&aliveC)[NW], int nofs) {
    const int al = A.len, S = max(A.S, 1), edge = A.edge, mis = A.mis;
    u32 C[NC][NW], BY[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
#pragma unroll
        for (int k = 0; k < NC; ++k) C[k][j] = 0;
        BY[j] = ~lowmask32(len - 32 * j);                 // a character that matches nothing inside the read
    }
    const int steps = min(min(S - 1, al), 64);       // (the masks hold the adapter's first 64 characters: fewer cells is still a necessary condition)
    // The counters only count, so the order of the steps is free: one loop per plane over the adapter
    // positions holding that letter (a switch on the letter inside one loop over the positions costs
    // three times the instructions: register copies at every join of its arms)
    const u64 sm = lowmask64(steps);
    auto run = [&](const u32 (&Pl)[NW], u64 m64) {
        u32 m = __builtin_amdgcn_readfirstlane((u32)m64);
        if constexpr (NC == 3 && SNK_SCREEN_BIN) {
            while (m & (m - 1)) {                                // two positions at a time while there are two
                const int c = __ffs((int)m) - 1;
                m &= m - 1;
                const int c2 = __ffs((int)m) - 1;
                m &= m - 1;
                screen_step2_bin<NW, 0>(Pl, c, c2, C);
            }
        }
        while (m) {
            const int c = __ffs((int)m) - 1;
            m &= m - 1;
            screen_step<NW, 0, NC>(Pl, c, C);
        }
        m = __builtin_amdgcn_readfirstlane((u32)(m64 >> 32));
        if constexpr (NC == 3 && SNK_SCREEN_BIN) {
            while (m & (m - 1)) {
                const int c = __ffs((int)m) - 1;
                m &= m - 1;
                const int c2 = __ffs((int)m) - 1;
                m &= m - 1;
                screen_step2_bin<NW, 1>(Pl, c, c2, C);
            }
        }
        while (m) {
            const int c = __ffs((int)m) - 1;
            m &= m - 1;
            screen_step<NW, 1, NC>(Pl, c, C);
        }
    };
    run(NX[0], A.cmask[0] & sm);
    run(NX[1], A.cmask[1] & sm);
    run(NX[2], A.cmask[2] & sm);
    run(NX[3], A.cmask[3] & sm);
    const u64 other = sm & ~(A.cmask[0] | A.cmask[1] | A.cmask[2] | A.cmask[3]);   // N and anything else in the adapter
    if (other) {
        if (FULL) {
            run(NXN, other & A.nmask);
            run(BY, other & ~A.nmask);
        } else {
            run(BY, other);
        }
    }
    // budgets as thermometer planes T_k = [budget >= k]; reject = mis count > budget
    const int rk1 = A.rk[1], rk2 = A.rk[2], rk3 = A.rk[3], rk4 = A.rk[0];     // (rk[0]: budgets >= 4, nC when there is none)
    // (a scalar copy of this for waves whose live lanes all hold reads of one length -- the masks are functions of the length alone --
    // was tried in round 5: ~3 VALU per read less on such tiles, but the second copy cost the kernel 6 / 12 more spilled VGPRs; the
    // cheaper lowmask32() above halves the masks' cost for every tile instead)
    auto tail = [&](const auto L) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            u32 c0p = C[0][j], c1p = C[1][j], c2p = NC >= 3 ? C[NC >= 3 ? 2 : 0][j] : 0u;
            if constexpr (NC == 3 && SNK_SCREEN_BIN) {         // binary counter -> "at least 1 / 2 / 3 mismatches"
                const u32 c0 = c0p, c1 = c1p;
                c0p = c0 | c1;
                c1p = c1;
                c2p = c0 & c1;
            }
            // candidates: phase B offsets 0 .. len - al (budget adaMis), phase C offsets len - al + 1 .. len - edge (budget of
            // r1 = len - edge - p: at least k for r1 >= rk_k); adaEdge may exceed the adapter (then there is no phase C at all)
            const u32 bm = lowmask32(L - al + 1 - 32 * j);                    // phase B region
            // (nofs: the offsets this call owns -- a block in the middle of a long read owns its first 256, the planes behind them
            // only serve as the characters an alignment reaches)
            const u32 valid = (bm | lowmask32(L - edge + 1 - 32 * j)) & lowmask32(nofs - 32 * j);
            auto thermo = [&](const int rk, const int k) { return (mis >= k ? bm : 0u) | (~bm & lowmask32(L - edge - rk + 1 - 32 * j)); };
            const u32 t1 = thermo(rk1, 1);
            u32 rej = c0p & ~t1;
            if (NC >= 3) rej |= c1p & ~thermo(rk2, 2);
            if (NC >= 4) rej |= c2p & ~thermo(rk3, 3);
            if (NC >= 4) {
                // four or more mismatches: out, except where the budget itself is 4 or more -- the counters stop at four, those
                // offsets all go to the exact decision
                rej |= C[NC >= 4 ? 3 : 0][j] & ~thermo(rk4, 4);
            } else {
                rej |= NC == 3 ? c2p : c1p;                                    // more mismatches than any budget of this adapter
            }
            const u32 alive = done ? 0u : (valid & ~rej);
            aliveB[j] = alive & bm;
            aliveC[j] = alive & ~bm;
        }
    };
    tail(len);
}

// Adapter search for the lanes with `todo`; returns the position or -1.
// X[k] bit p = read[p] == "ACGT"[k] (exact), ones beyond the read; XN likewise for 'N'.
// doA (per lane): the planes start at the read's first character (phase A exists, and `len` is the whole read).
// Blocks of a long read (snk_long.hip): `len` is what is LEFT of the read from the block's first position on (it may reach far
// past the planes), `nofs` the number of offsets the block owns (256, all of them in the read's last block) -- which of them are
// phase B and which phase C follows from len as in a whole read; the screen looks at 64 characters at most and an exact decision
// either fits the 64 match bits (adapters up to 64 characters: 255 + 63 < the planes' 320 positions) or walks the row.
// *hit_c: the position returned is a phase C one (a later block's phase C hit takes precedence: descending offsets, :765-788).
template <int NW, bool FULL, class AD>
__device__ __forceinline__ int adapter_tile(const AD &A, const DevAdapter &AG, const u32 (&X)[4][NW], const u32 (&XN)[NW],
                            int len, bool todo, const uint8_t *sptr, bool doA = true, bool doC = true, bool seq_lane = false,
                            int nofs = 32 * NW, bool *hit_c = nullptr) {
    const int al = A.len, S = max(A.S, 1), edge = A.edge, mis = A.mis;
    int result = -1;
    bool done = !todo;
    // seq_lane: a read with characters the planes do not hold against an adapter with lower-case characters
    if (!done && ((len < al && doA) || seq_lane)) {  // a READ shorter than the adapter: negative offsets, rare -> sequential
        result = adapter_pos_seq(sptr, len, AG);
        done = true;
    }
    const u64 cm0 = A.cmask[0], cm1 = A.cmask[1], cm2 = A.cmask[2], cm3 = A.cmask[3];
    const u64 cmn = FULL ? A.nmask : 0ull;
    // ---------------- phase A (src/read_filter.cpp:720-742): adapter[r1..] on read[0..].
    // Quick screen only (mismatches inside the first S-1 steps); survivors are queued in `pa`.
    u32 pa = 0;
    {
        const u64 x0 = ((u64)X[0][1] << 32) | X[0][0], x1 = ((u64)X[1][1] << 32) | X[1][0];
        const u64 x2 = ((u64)X[2][1] << 32) | X[2][0], x3 = ((u64)X[3][1] << 32) | X[3][0];
        const u64 xn = FULL ? (((u64)XN[1] << 32) | XN[0]) : 0ull;
        for (int r1 = 1; r1 <= 5; ++r1) {
            const int n = al - r1, budget = A.budgetA[r1];
            const u64 m = ((x0 & (cm0 >> r1)) | (x1 & (cm1 >> r1)) | (x2 & (cm2 >> r1)) | (x3 & (cm3 >> r1)) |
                           (xn & (cmn >> r1)));
            const u64 zz = ~m & lowmask64(min(min(n, S - 1), 64 - r1));       // (the shifted masks hold 64 - r1 cells)
            if (__popcll(zz) <= max(budget, 0)) pa |= 1u << r1;
        }
        if (done || !doA) pa = 0;
    }
    // ---------------- phases B+C screening: candidates p = 0 .. len-edge, bit-sliced
    u32 aliveB[NW], aliveC[NW];
    if (SNK_ABL != 7 && __any(!done)) {
        if (A.maxb <= 1) screen_planes<NW, FULL, 2>(A, X, XN, len, done, aliveB, aliveC, nofs);
        else if (A.maxb == 2) screen_planes<NW, FULL, 3>(A, X, XN, len, done, aliveB, aliveC, nofs);
        else screen_planes<NW, FULL, 4>(A, X, XN, len, done, aliveB, aliveC, nofs);
    } else {
#pragma unroll
        for (int j = 0; j < NW; ++j) aliveB[j] = aliveC[j] = 0;
    }
    if (!doC) {
#pragma unroll
        for (int j = 0; j < NW; ++j) aliveC[j] = 0;
    }
    // ---------------- exact decision of the survivors, in the reference's order:
    // phase A r1 = 1..5 (-> 0), phase B ascending offset (:743-764), phase C ascending r1 ==
    // descending offset (:765-788).  One candidate per lane per trip; trips are rare.
    if (SNK_ABL == 6) {                   // keep the screening alive, skip the decisions
#pragma unroll
        for (int j = 0; j < NW; ++j) asm volatile("" ::"v"(aliveB[j]), "v"(aliveC[j]));
        asm volatile("" ::"v"(pa));
    }
    while (SNK_ABL != 6 && __any(!done && (pa != 0 || any_bit(aliveB) || any_bit(aliveC)))) {
        if (!done) {
            int p = 0, sh = 0, n = 0, budget = 0, res = 0;
            bool have = true, skip_eval = false, skip_ok = false, from_c = false;
            if (pa) {
                sh = __ffs((int)pa) - 1;
                pa &= pa - 1;
                n = al - sh;
                budget = sh == 1 ? A.budgetA[1] : sh == 2 ? A.budgetA[2] : sh == 3 ? A.budgetA[3]
                         : sh == 4 ? A.budgetA[4] : A.budgetA[5];
                res = 0;
            } else if (any_bit(aliveB)) {
                p = lowest_bit(aliveB);
                clear_bit(aliveB, p);
                n = al;
                budget = mis;
                res = p;
            } else if (any_bit(aliveC)) {
                p = highest_bit(aliveC);
                clear_bit(aliveC, p);
                n = len - p;                                         // compared length, edge <= n < al
                res = p;
                from_c = true;
                if (n < S && n <= 64 && A.maxb <= 3) { skip_eval = true; skip_ok = !A.negC; }  // no run possible and every cell screened: survived <=> mis <= budget (exact counters)
                else budget = AG.budgetC[n - edge];
            } else {
                have = false;
            }
            if (have) {
                bool ok = skip_ok;
                if (!skip_eval) {
                    if (al > 64) {
                        // more cells than the 64 match bits hold: this one alignment character by character (adapter cell sh + c
                        // against read[p + c]; phase A: sh = r1 at p = 0)
                        ok = accept_seq(AG.seq, sh, sptr, len, p, n, S, budget);
                    } else {
                        u64 m = (window64(X[0], p, ~0u) & (cm0 >> sh)) | (window64(X[1], p, ~0u) & (cm1 >> sh)) |
                                (window64(X[2], p, ~0u) & (cm2 >> sh)) | (window64(X[3], p, ~0u) & (cm3 >> sh));
                        if (FULL) m |= window64(XN, p, ~0u) & (cmn >> sh);
                        ok = accept_exact(m, n, S, budget);
                    }
                }
                if (ok) { result = res; done = true; if (hit_c) *hit_c = from_c; }
            }
        }
    }
    return result;
}

}  // namespace
}  // namespace snk
