// snk_generic.hip -- "generic" gfx950 kernel of the SOAPnuke-filter hot path:
// one work-item per read pair, any read length up to 1000, any adapter length up
// to 255, any parameter combination.  It is the correctness anchor and the
// fallback for what the fast paths (snk_tiled.hip up to 256 positions, snk_long.hip
// beyond) do not cover: adapters over 64 nt or in lower case, more than four per
// mate, rows that are not 16-byte aligned beyond 256 positions.  Its loads are
// strided by the batch pitch.  As a fallback it only decides (records, reason and
// trimming-position counters) and leaves the per-position histograms to the LDS
// histogram kernel of snk_long.hip; with own_hist it adds them itself, one global
// atomic per base and quality (the anchor, kernel = 3).
//
// Semantics follow the reference row by row:
//   A1 stat_read      src/read_filter.cpp:80-313
//   A2 adapter_pos    src/read_filter.cpp:707-790
//   A3 fastq_trim     src/read_filter.cpp:338-471
//   A4 polyG_number   src/read_filter.cpp:472-482
//   A6 pe/se_discard  src/sequence.cpp:76-178,198-387
//   A8 stat_*_fqs     src/peprocess.cpp:1076-1423, src/seprocess.cpp:632-869
#include <hip/hip_runtime.h>
#include "snk_common.hip.h"

using namespace snk;

namespace {

// own_hist == 0: the per-position histograms (and the quality-range check that comes with them) are left to
// snk_long_hist_kernel, which runs behind this kernel on the records it wrote -- one global atomic per base and quality was
// nine tenths of this kernel's time
template <class PT>
__device__ int hist_read(const PT &P, unsigned long long *file, int lcap, int nq,
                         const uint8_t *s, const uint8_t *q, int start, int n, int own_hist) {
    unsigned long long *bs = file + SNK_GS_N, *qs = file + SNK_GS_N + (long)lcap * 5;
    int rc = SNK_OK;
    for (int i = 0; i < (own_hist ? n : 0); ++i) {
        const int u = s[start + i] & 0xDF;
        const int b = u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : 4;
        atomicAdd(&bs[i * 5 + b], 1ull);
        const int bq = (int)q[start + i] - P.phred;
        if (bq < 0 || bq >= nq) { rc = SNK_E_QUAL_RANGE; continue; }
        atomicAdd(&qs[(long)i * nq + bq], 1ull);
    }
    agg_inc(&file[SNK_GS_READS]);
    return rc;
}

__global__ void __launch_bounds__(256)
snk_generic_kernel(const DevParams *Pp, DevBatch B, DevStats st, int lcap, int nq, int own_hist) {
    typedef __attribute__((address_space(4))) DevParams CDevParams;          // scalar loads of every field (see snk_long_decide_kernel)
    const CDevParams &P = *(const CDevParams *)(uintptr_t)Pp;
    const long fb = file_block(lcap, nq);
    const long ts_off = SNK_GS_N + (long)lcap * 5 + (long)lcap * nq;
    const int pe = P.paired ? 1 : 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < B.n; i += (long)gridDim.x * blockDim.x) {
        ReadState r[2];
        int cf[2] = {0, 0};
        const uint8_t *s[2], *q[2];
        const unsigned long long gidx = B.first_index + (unsigned long long)i;
        bool bad = false;
        for (int m = 0; m <= pe; ++m) {
            const int len = B.len[m] ? (int)B.len[m][i] : B.fixed_len[m];
            s[m] = B.seq[m] + i * (long)B.pitch;
            q[m] = B.qual[m] + i * (long)B.pitch;
            if (len > lcap) { report_err(st, gidx, m, SNK_E_TOO_LONG); bad = true; break; }
            int e;
            if (P.n_ct[m] | P.n_gct) cf[m] = contam_flags(P.ct + m * SNK_MAX_CONTAMS, P.n_ct[m], P.gct, P.n_gct, s[m], len);   // src/read_filter.cpp:189-248
            stat_read_dev(P, m, s[m], q[m], len, r[m], e);
            if (e) { report_err(st, gidx, m, e); bad = true; break; }
        }
        if (bad) {                                           // (a record no later pass uses)
            ReadState z;
            rs_init(z, 0);
            store_rec(B.out[0], i, z, 255, 0);
            if (pe) store_rec(B.out[1], i, z, 255, 0);
            continue;
        }
        for (int m = 0; m <= pe; ++m) fastq_trim_dev(P, m, s[m], q[m], r[m]);
        int v = 0;
        const int reason = discard_reason(P, r[0], r[pe], B.dup ? B.dup[i] : 0, v, cf[0], cf[pe]);
        count_reason(st.sum, pe, reason, v);
        store_rec(B.out[0], i, r[0], reason, v);
        if (pe) store_rec(B.out[1], i, r[1], reason, v);

        const unsigned long long key = (gidx + 1) << 16;
        for (int m = 0; m <= pe; ++m) {
            unsigned long long *file = st.sum + SNK_FS_N + m * fb;
            int hh = -1, lh = -1, ht = -1, lt = -1, ad = -1;
            if (P.copy_back) { hh = r[m].hd_h; lh = r[m].lq_h; ht = r[m].hd_t; lt = r[m].lq_t; ad = r[m].adacut; }
            ts_update(file + ts_off, hh, lh, ht, lt, ad, (pe && m == 1) ? r[m].len : 0, !pe);
            const int rc = hist_read(P, file, lcap, nq, s[m], q[m], 0, r[m].len, own_hist);
            if (rc) report_err(st, gidx, m, rc);
            agg_max_last(&st.maxb[m], key | (unsigned long long)r[m].len);
        }
        if (reason == SNK_KEEP) {
            for (int m = 0; m <= pe; ++m) {
                unsigned long long *file = st.sum + SNK_FS_N + (2 + m) * fb;
                ts_update(file + ts_off, r[m].hd_h, r[m].lq_h, r[m].hd_t, r[m].lq_t, r[m].adacut,
                          (pe && m == 1) ? r[m].clen : r[m].len, !pe);
                hist_read(P, file, lcap, nq, s[m], q[m], r[m].start, r[m].clen, own_hist);
                agg_max_last(&st.maxb[2 + m], key | (unsigned long long)r[m].clen);
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
                        int nq, int own_hist, void *stream) {
    if (b.n <= 0) return;
    long blocks = (b.n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(snk_generic_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       dp, b, st, lcap, nq, own_hist);
}

void snk_launch_finalize(const DevStats &st, int lcap, int nq, void *stream) {
    hipLaunchKernelGGL(snk_finalize_kernel, dim3(4), dim3(256), 0, (hipStream_t)stream, st, lcap, nq);
}
