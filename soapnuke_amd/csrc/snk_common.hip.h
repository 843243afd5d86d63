// snk_common.hip.h -- device code shared by the generic and the wave-tiled kernels:
// the per-read state, the sequential adapter matcher (also the in-kernel fallback
// for reads shorter than the adapter), the discard cascade and the trimming-
// position bookkeeping.  Reference rows: SURVEY.md 8(a) A2/A3/A6/A8.
#pragma once
#include <hip/hip_runtime.h>
#include "snk_device.h"

namespace snk {

typedef unsigned int u32;
typedef unsigned long long u64;

struct ReadState {
    int len;
    int n_a, n_n, lowq, sumq;
    int polyx;                         // contig_base >= polyX_num
    int inc_ada;                       // include_adapter_seq == 1
    int hd_h, lq_h, hd_t, lq_t, adacut;
    int start, clen;
};

__device__ __forceinline__ void rs_init(ReadState &r, int len) {
    r.len = len;
    r.n_a = r.n_n = r.lowq = r.sumq = 0;
    r.polyx = 0;
    r.inc_ada = 0;
    r.hd_h = r.lq_h = r.hd_t = r.lq_t = r.adacut = -1;   // C_fastq_init, src/peprocess.cpp:1674
    r.start = 0;
    r.clen = len;
}

__device__ __forceinline__ int rdc(const uint8_t *s, int len, int i) {
    return (i >= 0 && i < len) ? (int)s[i] : 0;   // outside the read: '\0' 
}

// A2 adapter_pos(), src/read_filter.cpp:707-790, one read, sequential.
__device__ inline int adapter_pos_seq(const uint8_t *s, int len, const DevAdapter &A) {
    const int al = A.len;
    if (al == 0) return -1;
    for (int r1 = 1; r1 <= 5; ++r1) {                       // phase A, :720-742
        int mis = 0, run = 0;
        const int budget = A.budgetA[r1];
        for (int c = 0; c < al - r1; ++c) {
            if ((int)A.seq[r1 + c] == rdc(s, len, c)) { if (++run >= A.S) return 0; }
            else { ++mis; run = 0; if (mis > budget) break; }
        }
        if (mis <= budget) return 0;
    }
    for (int r1 = 0; r1 <= len - al; ++r1) {                // phase B, :743-764
        int mis = 0, run = 0;
        for (int c = 0; c < al; ++c) {
            if (A.seq[c] == s[r1 + c]) { if (++run >= A.S) return r1; }
            else { ++mis; run = 0; if (mis > A.mis) break; }
        }
        if (mis <= A.mis) return r1;
    }
    for (int r1 = 0; r1 < A.nC; ++r1) {                     // phase C, :765-788
        int mis = 0, run = 0;
        const int budget = A.budgetC[r1];
        const int st = len - r1 - A.edge;
        for (int c = 0; c < r1 + A.edge; ++c) {
            if ((int)A.seq[c] == rdc(s, len, st + c)) { if (++run >= A.S) return st; }
            else { ++mis; run = 0; if (mis > budget) break; }
        }
        if (mis <= budget) return st;
    }
    return -1;
}

// hasContam(), src/read_filter.cpp:507-603: sequential; an 'N' in the read is neither match nor mismatch.
__device__ inline int has_contam_seq(const uint8_t *s, int len, const DevContam &C) {
    if (C.len == 0) return -1;
    for (int r1 = 0; r1 < C.nC; ++r1) {                              // head, :523-547
        int mis = 0, run = 0;
        const int budget = C.mm[r1], seg = C.sm1[r1];
        for (int c = 0; c < r1 + C.edge; ++c) {
            const int rc = rdc(s, len, c);
            if ((int)C.seq[C.len - r1 - C.edge + c] == rc) { if (++run >= seg) return 0; }
            else if (rc != 'N') { ++mis; run = 0; if (mis > budget) break; }
        }
        if (mis <= budget) return 0;
    }
    for (int r1 = 0; r1 <= len - C.len; ++r1) {                      // middle, :549-573
        int mis = 0, run = 0;
        for (int c = 0; c < C.len; ++c) {
            const int rc = rdc(s, len, r1 + c);
            if ((int)C.seq[c] == rc) { if (++run >= C.S) return r1; }
            else if (rc != 'N') { ++mis; run = 0; if (mis > C.mis) break; }
        }
        if (mis <= C.mis) return r1;
    }
    for (int r1 = 0; r1 < C.nC; ++r1) {                              // tail, :575-601
        int mis = 0, run = 0;
        const int budget = C.mm[r1], seg = C.sm3[r1], st = len - r1 - C.edge;
        for (int c = 0; c < r1 + C.edge; ++c) {
            const int rc = rdc(s, len, st + c);
            if ((int)C.seq[c] == rc) { if (++run >= seg) return st; }
            else if (rc != 'N') { ++mis; run = 0; if (mis > budget) break; }
        }
        if (mis <= budget) return st;
    }
    return -1;
}

// ---- global contaminants: the verdict of global_contam_pos() (src/read_filter.cpp:961-1062; only ">= 0" is ever used,
// src/read_filter.cpp:229-262).
//
// What the reference's three scans compute, in other words: the read and the contaminant are laid on each other at every
// offset of a section (contaminant hanging off the front / inside / hanging off the end); the cells of one lay are
// walked in order, the lays of a section one after the other WITHOUT resetting the state in between.  The state is a
// window: it opens on a matching cell (score 1, span 1) provided at least min_match_len cells of the lay remain (in the
// last section the window opens first and the lay is then abandoned), every later cell extends it (score +1 / -200),
// it is dead once the score is down to -200 * mismatches, and the verdict is "hit" as soon as a window spans
// min_match_len cells with its score at (min_match_len - mismatches) - 200 * mismatches or better.
//
// Here: one equality bit per cell of a lay (up to 4 words), and an event walk over them -- a live window jumps over a
// whole run of matches (count-trailing-zeros; the hit test inside the run is a closed form), a dead one jumps to the
// next match.  Valid for the parameter range snk_create() admits: 0 <= mismatches <= 4 < ... < min_match_len, for which
// a dead window can never pass the hit test.
struct GcWindow { int score, span; };

// the event walk over one lay whose equality bits are given by word(w) (64 cells each)
template <class WordFn>
__device__ inline bool gc_walk(GcWindow &st, WordFn word, int n, int mml, int tms, int lower, bool early_stop) {
    auto next = [&](int pos, bool ones) -> int {         // first cell >= pos that is a match (ones) / a mismatch, n if none
        while (pos < n) {
            const int w = pos >> 6;
            unsigned long long x = ones ? word(w) : ~word(w);
            x &= ~0ull << (pos & 63);
            if (x) { const int p = (w << 6) + __ffsll((long long)x) - 1; return p < n ? p : n; }
            pos = (w + 1) << 6;
        }
        return n;
    };
    const int last_ok = n - mml;                 // cells j <= last_ok still have mml cells of the lay in front of them
    int pos = 0;
    while (pos < n) {
        if (st.score > tms) {                    // live window
            const int z = next(pos, false);
            const int run = z - pos;
            if (run > 0) {
                const int need = max(max(lower - st.score, mml - st.span), 1);     // matches until the hit test passes
                if (need <= run) return true;
                st.score += run;
                st.span += run;
                pos = z;
                if (pos == n) break;
            }
            st.score -= 200;                     // the mismatch at pos
            st.span += 1;
            if (st.score >= lower && st.span >= mml) return true;
            ++pos;
        } else {                                 // dead: nothing happens until the next match
            const int o = next(pos, true);
            if (o >= n) break;
            if (early_stop) {
                if (o > last_ok) break;
            } else {
                if (max(pos, last_ok + 1) < o) break;        // a mismatch without room comes first
            }
            st.score = 1;
            st.span = 1;
            if (!early_stop && n - o < mml) break;          // opened, and the lay is abandoned (the window lives on)
            if (st.score >= lower && st.span >= mml) return true;
            pos = o + 1;
        }
    }
    return false;
}

// one lay of n cells: cell j compares a[j] with b[j].  early_stop: sections 1 and 2 (a dead window meeting a cell with
// fewer than mml cells left abandons the lay); otherwise section 3 (a mismatch there abandons it, a match opens the
// window first).  True = hit.
__device__ inline bool gc_lay(GcWindow &st, const uint8_t *a, const uint8_t *b, int n, int mml, int tms, int lower, bool early_stop) {
    if (n <= 0) return false;
    if (n <= 64) {                               // the usual case: one word, all in registers
        unsigned lo = 0, hi = 0;
        const int n0 = min(n, 32);
        for (int j = 0; j < n0; ++j) lo |= (unsigned)(a[j] == b[j]) << j;
        for (int j = 32; j < n; ++j) hi |= (unsigned)(a[j] == b[j]) << (j - 32);
        const unsigned long long e = ((unsigned long long)hi << 32) | lo;
        return gc_walk(st, [&](int) { return e; }, n, mml, tms, lower, early_stop);
    }
    unsigned long long e0 = 0, e1 = 0, e2 = 0, e3 = 0;       // contaminants of up to 255 characters
    for (int j = 0; j < n; ++j) {
        const unsigned long long bit = (unsigned long long)(a[j] == b[j]) << (j & 63);
        const int w = j >> 6;
        e0 |= w == 0 ? bit : 0ull; e1 |= w == 1 ? bit : 0ull; e2 |= w == 2 ? bit : 0ull; e3 |= w == 3 ? bit : 0ull;
    }
    return gc_walk(st, [&](int w) { return w == 0 ? e0 : (w == 1 ? e1 : (w == 2 ? e2 : e3)); }, n, mml, tms, lower, early_stop);
}

// The same lay cell by cell, with nothing assumed about the parameters: for settings outside the event walk's range (more
// than 4 mismatches, or a match length not above the mismatch number) a window that is "dead" can still pass the hit
// test, which is evaluated behind every cell that does not abandon the lay (src/read_filter.cpp:973-1060).
__device__ inline bool gc_lay_cells(GcWindow &st, const uint8_t *a, const uint8_t *b, int n, int mml, int tms, int lower, bool early_stop) {
    for (int j = 0; j < n; ++j) {
        const bool eq = a[j] == b[j], live = st.score > tms, room = n - j >= mml;
        if (live) { st.score += eq ? 1 : -200; st.span += 1; }
        else if (eq) {
            if (early_stop && !room) break;
            st.score = 1;
            st.span = 1;
            if (!early_stop && !room) break;
        } else if (!room) break;
        if (st.score >= lower && st.span >= mml) return true;
    }
    return false;
}

__device__ inline bool global_contam_hit(const uint8_t *ref, int rl, const uint8_t *gc, int cl, int mml, int mmn) {
    const int tms = -200 * mmn, lower = (mml - mmn) + tms;
    GcWindow st = {-1000, 0};
    if (!(mmn >= 0 && mmn <= 4 && mml > mmn)) {                            // outside the event walk's range: cell by cell
        for (int i = cl - mml; i >= 0; --i)
            if (gc_lay_cells(st, ref, gc + i, min(cl - i, rl), mml, tms, lower, true)) return true;
        st = GcWindow{-1000, 0};
        for (int i = 0; i <= rl - cl; ++i)
            if (gc_lay_cells(st, ref + i, gc, cl, mml, tms, lower, true)) return true;
        st = GcWindow{-1000, 0};
        for (int i = cl > rl ? cl - rl : 0; i <= cl - mml; ++i)
            if (gc_lay_cells(st, ref + rl - (cl - i), gc, cl - i, mml, tms, lower, false)) return true;
        return false;
    }
    for (int i = cl - mml; i >= 0; --i)                                    // contaminant hanging off the front, less and less
        if (gc_lay(st, ref, gc + i, min(cl - i, rl), mml, tms, lower, true)) return true;
    st = GcWindow{-1000, 0};
    for (int i = 0; i <= rl - cl; ++i)                                     // contaminant inside the read
        if (gc_lay(st, ref + i, gc, cl, mml, tms, lower, true)) return true;
    st = GcWindow{-1000, 0};
    for (int i = cl > rl ? cl - rl : 0; i <= cl - mml; ++i)                // hanging off the end, more and more
        if (gc_lay(st, ref + rl - (cl - i), gc, cl - i, mml, tms, lower, false)) return true;
    return false;
}

// include_contam (bit 0) / include_global_contam (bit 1) of one read, src/read_filter.cpp:189-248.
// ct: the n_ct hasContam() contaminants of this mate, gct: the n_gct global ones (global memory or LDS copies)
__device__ inline int contam_flags(const DevContam *ct, int n_ct, const DevGContam *gct, int n_gct, const uint8_t *s, int len) {
    int f = 0;
    for (int i = 0; i < n_ct && !(f & 1); ++i)
        if (has_contam_seq(s, len, ct[i]) >= 0) f |= 1;
    for (int i = 0; i < n_gct && !(f & 2); ++i)
        for (int d = 0; d < 2 && !(f & 2); ++d)
            if (global_contam_hit(s, len, gct[i].seq[d], gct[i].len, gct[i].min_match_len, gct[i].mm)) f |= 2;
    return f;
}

// A3 fastq_trim() arithmetic once the per-read scan results are known
// (src/read_filter.cpp:383-468).  lq_hix/lq_tix/polyg: the three run lengths.
template <class PT>       // PT: DevParams, or DevParams in the constant address space (scalar loads of every field: snk_long.hip)
__device__ __forceinline__ void trim_finish(const PT &P, int mate, ReadState &r, int lq_hix,
                                            int lq_tix, int polyg) {
    const int len = r.len;
    int head_cut = 0, tail_cut = 0;
    if (P.has_hard) {
        r.hd_h = P.hard[P.paired ? 2 * mate : 0];
        r.hd_t = P.hard[P.paired ? 2 * mate + 1 : 1];
        head_cut = r.hd_h;
        tail_cut = r.hd_t;
    }
    if (P.has_lq) {
        r.lq_h = lq_hix;
        r.lq_t = lq_tix;
        head_cut = max(head_cut, lq_hix);
        tail_cut = max(tail_cut, lq_tix);
    }
    if (P.ada_trim && r.adacut > 0) tail_cut = max(tail_cut, r.adacut);
    if (P.has_polyG && polyg >= P.polyG_thr && polyg > tail_cut) tail_cut = polyg;
    if ((u64)(long long)(head_cut + tail_cut) > (u64)len) {       // int sum compared as size_t, :462
        r.start = 0;
        r.clen = 0;
    } else {
        r.clen = len - head_cut - tail_cut;
        r.start = r.clen ? head_cut : 0;
    }
}

// A1 stat_read() and A3 fastq_trim() of one read, sequential (the generic kernel; the per-lane fallback of the long-read kernel)
template <class PT>
__device__ inline void stat_read_dev(const PT &P, int mate, const uint8_t *s, const uint8_t *q,
                              int len, ReadState &r, int &err) {
    rs_init(r, len);
    err = SNK_OK;
    int ada_pos = -1;
    for (int i = 0; i < P.n_ada[mate]; ++i) {               // :175-188
        ada_pos = adapter_pos_seq(s, len, P.ada[mate * P.ada_stride + i]);
        if (ada_pos >= 0) break;
    }
    if (ada_pos >= 0) { r.inc_ada = 1; r.adacut = len - ada_pos; }
    if (len == 0) { err = SNK_E_EMPTY_SEQ; return; }      // :250
    int last = 'Q', run = 0, maxrun = 1;
    for (int i = 0; i < len; ++i) {                          // :258-308
        const int c = s[i];
        if (c == last) { if (++run > maxrun) maxrun = run; } else run = 1;
        last = c;
        const int u = c & 0xDF;                              // fold case
        if (u == 'A') ++r.n_a;
        else if (u == 'N') ++r.n_n;
        else if (!(u == 'C' || u == 'G' || u == 'T')) { err = SNK_E_BAD_BASE; return; }
        const int bq = (int)q[i] - P.phred;
        r.sumq += bq;
        r.lowq += (bq <= P.low_qual);
    }
    r.polyx = (P.polyX_num != -1 && maxrun >= P.polyX_num) ? 1 : 0;
}

template <class PT>
__device__ inline void fastq_trim_dev(const PT &P, int mate, const uint8_t *s, const uint8_t *q,
                               ReadState &r) {
    if (!P.trim_on) return;                                  // src/read_filter.cpp:354
    const int len = r.len;
    int hix = 0, tix = 0, g = 0;
    if (P.has_lq) {                                          // :390-429
        for (int i = 0; i < P.lq_head_len; ++i) {
            if (rdc(q, len, i) - P.phred < P.lq_head_q) ++hix; else break;
        }
        for (int i = 0; i < P.lq_tail_len; ++i) {
            if (rdc(q, len, len - i - 1) - P.phred < P.lq_tail_q) ++tix; else break;
        }
    }
    if (P.has_polyG)                                         // :472-482
        for (int i = len - 1; i >= 0; --i) { if ((s[i] & 0xDF) == 'G') ++g; else break; }
    trim_finish(P, mate, r, hix, tix, g);
}

__device__ __forceinline__ int pe_dis(bool a, bool b) { return (a ? 1 : 0) + (b ? 2 : 0); }

// A6: the cascade (src/sequence.cpp:198-387 PE, :76-178 SE); returns the reason and
// the pe_dis() code, counters are the caller's business.
// cfa / cfb: contaminant verdicts of the mates, bit 0 = include_contam, bit 1 = include_global_contam
template <class PT>
__device__ inline int discard_reason(const PT &P, const ReadState &a, const ReadState &b, int dup,
                                     int &vout, int cfa = 0, int cfb = 0) {
    const bool pe = P.paired;
    int v;
    vout = 0;
#define SNK_TEST(COND_A, COND_B, REASON)                 \
    v = pe_dis((COND_A), pe && (COND_B));                \
    if (v > 0) { vout = pe ? v : 0; return REASON; }
    if (P.rmdup && (dup & 1)) return SNK_R_DUP;
    if (dup & 2) return SNK_R_TILE;                                  // src/sequence.cpp:213-231 (fq1's read name decides)
    if (dup & 4) return SNK_R_FOV;
    if (P.has_min) {
        SNK_TEST((u32)a.clen < P.min_len_u, (u32)b.clen < P.min_len_u, SNK_R_SHORT)
    } else if (pe && (a.clen == 0 || b.clen == 0)) {
        return SNK_R_EMPTY;
    }
    if (P.has_max) { SNK_TEST((u32)a.clen > P.max_len_u, (u32)b.clen > P.max_len_u, SNK_R_LONG) }
    if (P.contam_discard && (cfa | cfb)) {                          // src/sequence.cpp:264-290 (PE), :116-127 (SE)
        if (pe) {
            SNK_TEST(cfa & 2, cfb & 2, SNK_R_GCONTAM)
            SNK_TEST(cfa & 1, cfb & 1, SNK_R_CONTAM)
        } else {
            if (cfa & 1) return SNK_R_CONTAM;
            if (cfa & 2) return SNK_R_GCONTAM;
        }
    }
    if (P.has_n) { SNK_TEST(a.n_n >= P.thr_n[a.len], b.n_n >= P.thr_n[b.len], SNK_R_NRATE) }
    if (P.has_highA) { SNK_TEST(a.n_a >= P.thr_a[a.len], b.n_a >= P.thr_a[b.len], SNK_R_HIGHA) }
    if (P.polyX_num != -1) { SNK_TEST(a.polyx, b.polyx, SNK_R_POLYX) }
    if (P.has_lowq) { SNK_TEST(a.lowq >= P.thr_lowq[a.len], b.lowq >= P.thr_lowq[b.len], SNK_R_LOWQUAL) }
    if (P.has_meanq) { SNK_TEST(a.sumq < P.thr_meanq[a.len], b.sumq < P.thr_meanq[b.len], SNK_R_MEANQ) }
    if (!P.ada_trim) { SNK_TEST(a.inc_ada, b.inc_ada, SNK_R_ADAPTER) }
#undef SNK_TEST
    return SNK_KEEP;
}

__device__ __forceinline__ int reason_family(int reason) {
    switch (reason) {
    case SNK_R_SHORT: return SNK_FS_SHORT;
    case SNK_R_LONG: return SNK_FS_LONG;
    case SNK_R_GCONTAM: return SNK_FS_GCONTAM;
    case SNK_R_CONTAM: return SNK_FS_CONTAM;
    case SNK_R_NRATE: return SNK_FS_NRATE;
    case SNK_R_HIGHA: return SNK_FS_HIGHA;
    case SNK_R_POLYX: return SNK_FS_POLYX;
    case SNK_R_LOWQUAL: return SNK_FS_LOWQUAL;
    case SNK_R_MEANQ: return SNK_FS_MEANQ;
    case SNK_R_ADAPTER: return SNK_FS_ADAPTER;
    default: return -1;
    }
}

// ++*p for every lane that gets here (all of them with the same p), as one atomic
__device__ __forceinline__ void agg_inc(unsigned long long *p) {
    const u64 same = __ballot(1);
    if ((int)__lane_id() == __ffsll((long long)same) - 1) atomicAdd(p, (unsigned long long)__popcll(same));
}

// atomicMax of a "last read" word (index + 1 in the high bits, see DevStats::maxb) for every lane that gets here: the lanes of a
// wavefront walk consecutive reads, so the highest of them carries the largest word -- one atomic instead of up to 64
__device__ __forceinline__ void agg_max_last(unsigned long long *p, unsigned long long v) {
    const u64 same = __ballot(1);
    if ((int)__lane_id() == 63 - __clzll((long long)same)) atomicMax(p, v);
}

// the filter-statistics counters of one verdict (C_filter_stat, src/peprocess.cpp:1535-1600): the reason's family, and for a pair
// which mate(s) gave it.  One atomic per counter and wavefront: the lanes that get here agree on the counters present (a uniform
// loop over them, ballots for the counts).  The counters are a handful of words -- one atomic per read to the same few addresses
// serialises in L2: 4.8 ms per 1 M long pairs went there (rocprof: 79 % of the wave cycles waiting).
template <class CT>                    // CT: u64 (the stats block) or u32 (a workgroup's LDS copy of its first SNK_FS_N words)
__device__ inline void count_reason(CT *fs, bool pe, int reason, int v) {
    int idx = reason == SNK_R_DUP ? SNK_FS_DUP : reason == SNK_R_TILE ? SNK_FS_TILE : reason == SNK_R_FOV ? SNK_FS_FOV : reason_family(reason);
    const int vv = (pe && idx >= 0 && reason != SNK_R_DUP && reason != SNK_R_TILE && reason != SNK_R_FOV) ? v : 0;
    const int lane = (int)__lane_id();
    u64 todo = __ballot(idx >= 0);
    while (todo) {
        const int lead = __ffsll((long long)todo) - 1;
        const int f = __shfl(idx, lead, 64);
        const u64 same = __ballot(idx == f);
        const u64 s1 = __ballot(idx == f && (vv & 1)), s2 = __ballot(idx == f && (vv & 2)), s3 = __ballot(idx == f && vv == 3);
        if (lane == lead) {
            atomicAdd(&fs[f], (CT)__popcll(same));
            if (s1) atomicAdd(&fs[f + 1], (CT)__popcll(s1));
            if (s2) atomicAdd(&fs[f + 2], (CT)__popcll(s2));
            if (s3) atomicAdd(&fs[f + 3], (CT)__popcll(s3));
        }
        todo &= ~same;
    }
}

// CT = u64 (the stats block itself) or u32 (the tiled kernel's per-workgroup copy)
template <class CT>
__device__ __forceinline__ void ts_inc(CT *ts, long idx) {
    if (idx >= 0 && idx < SNK_TS_N) atomicAdd(&ts[idx], (CT)1);   // outside the struct the reference is UB
}

// src/peprocess.cpp:1107-1143 (fq1) / :1325-1360 (fq2) / src/seprocess.cpp:647-682
template <class CT>
__device__ inline void ts_update(CT *ts, int hd_h, int lq_h, int hd_t, int lq_t, int ada, long base_len,
                                 bool se) {
    if (hd_h > 0 || lq_h > 0) {
        if (hd_h >= lq_h) ts_inc(ts, SNK_TS_HT + hd_h); else ts_inc(ts, SNK_TS_HLQ + lq_h);
    }
    if (hd_t > 0 || lq_t > 0 || (se ? ada >= 0 : ada > 0)) {
        if (hd_t >= lq_t) {
            if (hd_t >= ada) ts_inc(ts, SNK_TS_TT + base_len - hd_t + 1);
            else ts_inc(ts, SNK_TS_TA + base_len - ada + 1);
        } else {
            if (lq_t >= ada) ts_inc(ts, SNK_TS_TLQ + base_len - lq_t + 1);
            else ts_inc(ts, SNK_TS_TA + base_len - ada + 1);
        }
    }
}

template <class ST>      // ST: DevStats, or DevStats in the constant address space
__device__ __forceinline__ void report_err(const ST &st, u64 index, int mate, int code) {
    // within one pair the reference meets stat_read() errors of either mate before the
    // quality-range check of the raw-stats pass: class bit 5 orders them that way
    atomicMin(st.err, (index << 8) | ((u64)(code == SNK_E_QUAL_RANGE) << 5) | ((u64)mate << 4) | (u64)code);
}

__device__ __forceinline__ void store_rec(snk_read_result *out, long i, const ReadState &r, int reason, int v) {
    snk_read_result o;
    o.head_hdcut = (int16_t)r.hd_h; o.head_lqcut = (int16_t)r.lq_h;
    o.tail_hdcut = (int16_t)r.hd_t; o.tail_lqcut = (int16_t)r.lq_t;
    o.adacut_pos = (int16_t)r.adacut;
    o.clean_start = (uint16_t)r.start; o.clean_len = (uint16_t)r.clen;
    o.reason = (uint8_t)reason; o.flags = (uint8_t)v;
    reinterpret_cast<uint4 *>(out)[i] = *reinterpret_cast<const uint4 *>(&o);
}

__device__ __forceinline__ long file_block(int lcap, int nq) {
    return SNK_GS_N + (long)lcap * 5 + (long)lcap * nq + SNK_TS_N;
}

}  // namespace snk
