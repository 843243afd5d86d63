// snk_generic.hip -- "generic" gfx950 kernel of the SOAPnuke-filter hot path:
// one work-item per read pair, any read length up to 1000, any adapter length up
// to 255, any parameter combination.  It is the correctness anchor and the
// fallback for shapes the wave-tiled kernel (snk_tiled.hip) does not cover
// (reads > 256 nt, adapters > 64 nt); it is NOT the fast path: its loads are
// strided by the batch pitch and its histograms are global atomics.
//
// Semantics follow the reference row by row:
//   A1 stat_read      src/read_filter.cpp:80-313
//   A2 adapter_pos    src/read_filter.cpp:707-790
//   A3 fastq_trim     src/read_filter.cpp:338-471
//   A4 polyG_number   src/read_filter.cpp:472-482
//   A6 pe/se_discard  src/sequence.cpp:76-178,198-387
//   A8 stat_*_fqs     src/peprocess.cpp:1076-1423, src/seprocess.cpp:632-869
#include <hip/hip_runtime.h>
#include "snk_device.h"

namespace {

struct ReadState {
    int len;
    int n_a, n_n, contig, lowq, sumq;
    int inc_ada;                       // include_adapter_seq == 1
    int hd_h, lq_h, hd_t, lq_t, adacut;
    int start, clen;
    int err;
};

__device__ __forceinline__ int rdc(const uint8_t *s, int len, int i) {
    return (i >= 0 && i < len) ? (int)s[i] : 0;   // out of range reads as '\0'
}

// A2: three phases, first accepted alignment wins; inside an alignment the scan
// stops at a run of S matches (accept) or at mismatch budget+1 (reject).
__device__ int adapter_pos_dev(const uint8_t *s, int len, const DevAdapter &A) {
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

__device__ void stat_read_dev(const DevParams &P, int mate, const uint8_t *s, const uint8_t *q,
                              int len, ReadState &r) {
    r.len = len;
    r.n_a = r.n_n = r.lowq = r.sumq = 0;
    r.contig = 1;
    r.inc_ada = 0;
    r.hd_h = r.lq_h = r.hd_t = r.lq_t = r.adacut = -1;
    r.start = 0;
    r.clen = len;
    r.err = SNK_OK;
    int ada_pos = -1;
    for (int i = 0; i < P.n_ada[mate]; ++i) {               // :175-188
        ada_pos = adapter_pos_dev(s, len, P.ada[mate * SNK_MAX_ADAPTERS + i]);
        if (ada_pos >= 0) break;
    }
    if (ada_pos >= 0) { r.inc_ada = 1; r.adacut = len - ada_pos; }
    if (len == 0) { r.err = SNK_E_EMPTY_SEQ; return; }      // :250
    int last = 'Q', run = 0, maxrun = 1;
    for (int i = 0; i < len; ++i) {                          // :258-308
        const int c = s[i];
        if (c == last) { if (++run > maxrun) maxrun = run; } else run = 1;
        last = c;
        const int u = c & 0xDF;                              // fold case
        if (u == 'A') ++r.n_a;
        else if (u == 'N') ++r.n_n;
        else if (!(u == 'C' || u == 'G' || u == 'T')) { r.err = SNK_E_BAD_BASE; return; }
        const int bq = (int)q[i] - P.phred;
        r.sumq += bq;
        r.lowq += (bq <= P.low_qual);
    }
    r.contig = maxrun;
}

__device__ void fastq_trim_dev(const DevParams &P, int mate, const uint8_t *s, const uint8_t *q,
                               ReadState &r) {
    if (!P.trim_on) return;                                  // :354
    const int len = r.len;
    int head_cut = 0, tail_cut = 0;
    if (P.has_hard) {                                        // :384-389
        r.hd_h = P.hard[P.paired ? 2 * mate : 0];
        r.hd_t = P.hard[P.paired ? 2 * mate + 1 : 1];
        head_cut = r.hd_h;
        tail_cut = r.hd_t;
    }
    if (P.has_lq) {                                          // :390-429
        int hix = 0, tix = 0;
        for (int i = 0; i < P.lq_head_len; ++i) {
            if (rdc(q, len, i) - P.phred < P.lq_head_q) ++hix; else break;
        }
        for (int i = 0; i < P.lq_tail_len; ++i) {
            if (rdc(q, len, len - i - 1) - P.phred < P.lq_tail_q) ++tix; else break;
        }
        r.lq_h = hix;
        r.lq_t = tix;
        head_cut = max(head_cut, hix);
        tail_cut = max(tail_cut, tix);
    }
    if (P.ada_trim && r.adacut > 0) tail_cut = max(tail_cut, r.adacut);   // :430-442
    if (P.has_polyG) {                                       // :454-461
        int g = 0;
        for (int i = len - 1; i >= 0; --i) { if ((s[i] & 0xDF) == 'G') ++g; else break; }
        if (g >= P.polyG_thr && g > tail_cut) tail_cut = g;
    }
    // :462-468  (int sum compared as size_t)
    if ((unsigned long long)(long long)(head_cut + tail_cut) > (unsigned long long)len) {
        r.start = 0; r.clen = 0;
    } else {
        r.clen = len - head_cut - tail_cut;
        r.start = r.clen ? head_cut : 0;
    }
}

__device__ __forceinline__ int pe_dis(bool a, bool b) { return (a ? 1 : 0) + (b ? 2 : 0); }

__device__ void fam_add(unsigned long long *fs, int base, int v) {
    if (v & 1) atomicAdd(&fs[base + 1], 1ull);
    if (v & 2) atomicAdd(&fs[base + 2], 1ull);
    if (v == 3) atomicAdd(&fs[base + 3], 1ull);
    atomicAdd(&fs[base], 1ull);
}

// the discard cascade, PE (src/sequence.cpp:198-387) and SE (:76-178) in one
__device__ int discard_dev(const DevParams &P, const ReadState &a, const ReadState &b, int dup,
                           unsigned long long *fs, int &vout) {
    const bool pe = P.paired;
    int v;
    vout = 0;
#define SNK_TEST(COND_A, COND_B, FAM, REASON)                         \
    v = pe_dis((COND_A), pe && (COND_B));                             \
    if (v > 0) {                                                      \
        if (pe) { fam_add(fs, FAM, v); vout = v; }                    \
        else atomicAdd(&fs[FAM], 1ull);                               \
        return REASON;                                                \
    }
    if (P.rmdup && dup) { atomicAdd(&fs[SNK_FS_DUP], 1ull); return SNK_R_DUP; }
    if (P.has_min) {
        SNK_TEST((uint32_t)a.clen < P.min_len_u, (uint32_t)b.clen < P.min_len_u, SNK_FS_SHORT, SNK_R_SHORT)
    } else if (pe && (a.clen == 0 || b.clen == 0)) {
        return SNK_R_EMPTY;
    }
    if (P.has_max) { SNK_TEST((uint32_t)a.clen > P.max_len_u, (uint32_t)b.clen > P.max_len_u, SNK_FS_LONG, SNK_R_LONG) }
    if (P.has_n) { SNK_TEST(a.n_n >= P.thr_n[a.len], b.n_n >= P.thr_n[b.len], SNK_FS_NRATE, SNK_R_NRATE) }
    if (P.has_highA) { SNK_TEST(a.n_a >= P.thr_a[a.len], b.n_a >= P.thr_a[b.len], SNK_FS_HIGHA, SNK_R_HIGHA) }
    if (P.polyX_num != -1) { SNK_TEST(a.contig >= P.polyX_num, b.contig >= P.polyX_num, SNK_FS_POLYX, SNK_R_POLYX) }
    if (P.has_lowq) { SNK_TEST(a.lowq >= P.thr_lowq[a.len], b.lowq >= P.thr_lowq[b.len], SNK_FS_LOWQUAL, SNK_R_LOWQUAL) }
    if (P.has_meanq) { SNK_TEST(a.sumq < P.thr_meanq[a.len], b.sumq < P.thr_meanq[b.len], SNK_FS_MEANQ, SNK_R_MEANQ) }
    if (!P.ada_trim) { SNK_TEST(a.inc_ada, b.inc_ada, SNK_FS_ADAPTER, SNK_R_ADAPTER) }
#undef SNK_TEST
    return SNK_KEEP;
}

__device__ __forceinline__ void ts_inc(unsigned long long *ts, long idx) {
    if (idx >= 0 && idx < SNK_TS_N) atomicAdd(&ts[idx], 1ull);
}

// src/peprocess.cpp:1107-1143 / :1325-1360 / src/seprocess.cpp:647-682
__device__ void ts_update(unsigned long long *ts, int hd_h, int lq_h, int hd_t, int lq_t, int ada,
                          long base_len, bool se) {
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

__device__ int hist_read(const DevParams &P, unsigned long long *file, int lcap, int nq,
                         const uint8_t *s, const uint8_t *q, int start, int n) {
    unsigned long long *bs = file + SNK_GS_N, *qs = file + SNK_GS_N + (long)lcap * 5;
    int rc = SNK_OK;
    for (int i = 0; i < n; ++i) {
        const int u = s[start + i] & 0xDF;
        const int b = u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : 4;
        atomicAdd(&bs[i * 5 + b], 1ull);
        const int bq = (int)q[start + i] - P.phred;
        if (bq < 0 || bq >= nq) { rc = SNK_E_QUAL_RANGE; continue; }
        atomicAdd(&qs[(long)i * nq + bq], 1ull);
    }
    atomicAdd(&file[SNK_GS_READS], 1ull);
    return rc;
}

__device__ __forceinline__ void report_err(const DevStats &st, unsigned long long index, int mate, int code) {
    atomicMin(st.err, (index << 8) | ((unsigned long long)mate << 4) | (unsigned long long)code);
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

__global__ void __launch_bounds__(256)
snk_generic_kernel(const DevParams *Pp, DevBatch B, DevStats st, int lcap, int nq) {
    const DevParams &P = *Pp;
    const long fb = SNK_GS_N + (long)lcap * 5 + (long)lcap * nq + SNK_TS_N;
    const long ts_off = SNK_GS_N + (long)lcap * 5 + (long)lcap * nq;
    const int pe = P.paired ? 1 : 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < B.n; i += (long)gridDim.x * blockDim.x) {
        ReadState r[2];
        const uint8_t *s[2], *q[2];
        const unsigned long long gidx = B.first_index + (unsigned long long)i;
        bool bad = false;
        for (int m = 0; m <= pe; ++m) {
            const int len = B.len[m] ? (int)B.len[m][i] : B.fixed_len[m];
            s[m] = B.seq[m] + i * (long)B.pitch;
            q[m] = B.qual[m] + i * (long)B.pitch;
            if (len > lcap) { report_err(st, gidx, m, SNK_E_TOO_LONG); bad = true; break; }
            stat_read_dev(P, m, s[m], q[m], len, r[m]);
            if (r[m].err) { report_err(st, gidx, m, r[m].err); bad = true; break; }
        }
        if (bad) continue;
        for (int m = 0; m <= pe; ++m) fastq_trim_dev(P, m, s[m], q[m], r[m]);
        int v = 0;
        const int reason = discard_dev(P, r[0], r[pe], B.dup ? B.dup[i] : 0, st.sum, v);
        store_rec(B.out[0], i, r[0], reason, v);
        if (pe) store_rec(B.out[1], i, r[1], reason, v);

        const unsigned long long key = (gidx + 1) << 16;
        for (int m = 0; m <= pe; ++m) {
            unsigned long long *file = st.sum + SNK_FS_N + m * fb;
            int hh = -1, lh = -1, ht = -1, lt = -1, ad = -1;
            if (P.copy_back) { hh = r[m].hd_h; lh = r[m].lq_h; ht = r[m].hd_t; lt = r[m].lq_t; ad = r[m].adacut; }
            ts_update(file + ts_off, hh, lh, ht, lt, ad, (pe && m == 1) ? r[m].len : 0, !pe);
            const int rc = hist_read(P, file, lcap, nq, s[m], q[m], 0, r[m].len);
            if (rc) report_err(st, gidx, m, rc);
            atomicMax(&st.maxb[m], key | (unsigned long long)r[m].len);
        }
        if (reason == SNK_KEEP) {
            for (int m = 0; m <= pe; ++m) {
                unsigned long long *file = st.sum + SNK_FS_N + (2 + m) * fb;
                ts_update(file + ts_off, r[m].hd_h, r[m].lq_h, r[m].hd_t, r[m].lq_t, r[m].adacut,
                          (pe && m == 1) ? r[m].clen : r[m].len, !pe);
                hist_read(P, file, lcap, nq, s[m], q[m], r[m].start, r[m].clen);
                atomicMax(&st.maxb[2 + m], key | (unsigned long long)r[m].clen);
            }
        }
    }
}

// gs[] = column sums of the histograms (a/c/g/t/n, q20, q30, bases); reads_number
// is accumulated by the kernels.  One workgroup per file block; idempotent.
__global__ void __launch_bounds__(256) snk_finalize_kernel(DevStats st, int lcap, int nq) {
    const long fb = SNK_GS_N + (long)lcap * 5 + (long)lcap * nq + SNK_TS_N;
    unsigned long long *file = st.sum + SNK_FS_N + blockIdx.x * fb;
    const unsigned long long *bs = file + SNK_GS_N, *qs = file + SNK_GS_N + (long)lcap * 5;
    __shared__ unsigned long long acc[7];
    if (threadIdx.x < 7) acc[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long loc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int p = threadIdx.x; p < lcap; p += blockDim.x) {
        for (int b = 0; b < 5; ++b) loc[b] += bs[p * 5 + b];
        for (int j = 20; j < nq; ++j) {
            const unsigned long long x = qs[(long)p * nq + j];
            loc[5] += x;
            if (j >= 30) loc[6] += x;
        }
    }
    for (int k = 0; k < 7; ++k) if (loc[k]) atomicAdd(&acc[k], loc[k]);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long bases = 0;
        for (int b = 0; b < 5; ++b) { file[SNK_GS_A + b] = acc[b]; bases += acc[b]; }
        file[SNK_GS_BASES] = bases;
        file[SNK_GS_Q20] = acc[5];
        file[SNK_GS_Q30] = acc[6];
    }
}

}  // namespace

void snk_launch_generic(const DevParams *dp, const DevBatch &b, const DevStats &st, int lcap,
                        int nq, void *stream) {
    if (b.n <= 0) return;
    long blocks = (b.n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(snk_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       dp, b, st, lcap, nq);
}

void snk_launch_finalize(const DevStats &st, int lcap, int nq, void *stream) {
    hipLaunchKernelGGL(snk_finalize_kernel, dim3(4), dim3(256), 0, (hipStream_t)stream, st, lcap, nq);
}
