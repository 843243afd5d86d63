/*
 * snk_filter.h -- C ABI of the MI355X-native `SOAPnuke filter` per-read hot path.
 *
 * This is the drop-in boundary.  The reference has no
 * FFI; the seam this library replaces is the pair of virtual batch methods on
 * its pipeline driver:
 *
 *   peProcess::filter_pe_fqs(PEcalOption*)          src/peprocess.h:61, src/peprocess.cpp:1424-1484
 *   peProcess::stat_pe_fqs(PEstatOption, string)    src/peprocess.h:60, src/peprocess.cpp:1076-1423
 *   seProcess::filter_se_fqs(SEcalOption)           src/seprocess.h:38, src/seprocess.cpp:871-917
 *   seProcess::stat_se_fqs(SEstatOption, string)    src/seprocess.h:40, src/seprocess.cpp:632-869
 *
 * both called per patch of reads from thread_process_reads()
 * (src/peprocess.cpp:1862-1992).  One snk_filter_batch() call == one
 * filter_*_fqs() + stat_*_fqs("raw") + stat_*_fqs("clean") on the same patch.
 *
 * Plain C, plain pointers and sizes; no C++/torch types.  All structs are POD
 * and versioned through `struct_size`.
 */
#ifndef SNK_FILTER_H
#define SNK_FILTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNK_ABI_VERSION 1

/* reference limits: READ_MAX_LEN src/global_variable.h:9 */
#define SNK_READ_MAX_LEN 1000
#define SNK_MAX_ADAPTERS 16   /* per mate in snk_params.adapters[]; longer lists travel through snk_params.adapter_list
                                 (the reference takes a list file of any length, src/process_argv.cpp:242-304,
                                 loop at src/read_filter.cpp:175-188) */
#define SNK_MAX_ADAPTER_LEN 255

/* ------------------------------------------------------------------ params
 * POD mirror of the hot-path fields of C_global_parameter
 * (src/global_parameter.h:8-190, defaults :20-83).  "-1 == off" follows the
 * reference's own convention (compared with != -1 in src/sequence.cpp).      */
typedef struct snk_params {
    int32_t struct_size;          /* = sizeof(snk_params) */
    int32_t paired;               /* 1: peProcess semantics, 0: seProcess semantics */
    int32_t quality_phred;        /* gp.qualityPhred        (33)  */
    int32_t output_quality_phred; /* gp.outputQualityPhred  (33)  */
    int32_t max_base_quality;     /* gp.maxBaseQuality      (42)  */
    int32_t low_qual;             /* gp.lowQual             (5)   -l */
    float   low_qual_ratio;       /* gp.lowQualityBaseRatio (0.5) -q */
    float   n_ratio;              /* gp.n_ratio             (0.05)-n */
    float   highA_ratio;          /* gp.highA_ratio         (-1)  -p */
    float   polyG_tail;           /* gp.polyG_tail          (-1)  -g */
    int32_t polyX_num;            /* gp.polyX_num           (-1)  -X */
    int32_t mean_quality;         /* gp.meanQuality         (-1)  -m */
    int32_t min_read_length;      /* gp.min_read_length     (30)  -4 */
    int32_t max_read_length;      /* gp.max_read_length     (-1)     */
    int32_t ada_trim;             /* gp.adapter_discard_or_trim=="trim" (-J) */
    int32_t contam_trim;          /* gp.contam_discard_or_trim=="trim": contaminant hits do not discard (the trimming
                                     itself is commented out in the reference); also feeds the "copy cut fields
                                     back" test, src/peprocess.cpp:1441 */
    int32_t has_hard_trim;        /* !gp.trim.empty() (-t) */
    int32_t hard_trim[4];         /* head1,tail1,head2,tail2 (SE: head,tail) */
    int32_t has_lq_trim;          /* trimBadHead or trimBadTail given (-x/-y) */
    int32_t lq_head_qual, lq_head_len; /* trimBadHead "qual,maxlen"; 0,0 when absent */
    int32_t lq_tail_qual, lq_tail_len; /* trimBadTail */
    int32_t ada_mis[2];           /* gp.adaMis , gp.adaMis2  (2)   */
    float   ada_mr[2];            /* gp.adaMR  , gp.adaMR2   (0.5) */
    int32_t ada_edge[2];          /* gp.adaEdge, gp.adaEdge2 (6)   */
    int32_t n_adapters[2];        /* gp.ada1s.size(), gp.ada2s.size() */
    const char *adapters[2][SNK_MAX_ADAPTERS]; /* NUL-terminated, compared as-is */
    int32_t rmdup;                /* gp.rmdup: honour snk_batch.dup */
    int32_t max_read_len;         /* capacity: longest read this context will see (<=1000) */
    /* contaminant screening (config keys contam1/contam2, ctMatchR, global_contams,
     * glob_cotm_mR, glob_cotm_mM).  Strings exactly as the reference holds them -- comma-separated
     * lists -- NULL or "" = off.  Only the verdicts matter downstream: with contam_trim == 0 a hit
     * discards the read/pair (src/sequence.cpp:116-127,264-290); the contam trimming itself is
     * commented out in the reference (src/read_filter.cpp:443-452).  Matchers use the mate's own
     * adaMis/adaEdge: ada_mis[1]/ada_edge[1] for mate 2 of a pair (gp2, src/sequence.cpp:182-189;
     * src/read_filter.cpp:513,611). */
    const char *contam[2];        /* gp.contam1_seq, gp.contam2_seq */
    const char *ct_match_r;       /* gp.ctMatchR ("0.2"; a list when contam is a list) */
    const char *global_contams;   /* gp.global_contams */
    const char *g_mrs, *g_mms;    /* gp.g_mrs, gp.g_mms (one value per global contaminant) */
    /* adapter lists of any length: when adapter_list[m] is non-NULL, mate m's n_adapters[m] adapters are
     * adapter_list[m][0 .. n_adapters[m]) and adapters[m][] is ignored (n_adapters[m] may exceed SNK_MAX_ADAPTERS) */
    const char *const *adapter_list[2];
} snk_params;

/* adapter i of mate m, whichever of the two forms carries it */
static inline const char *snk_adapter_at(const snk_params *p, int m, int i) {
    return p->adapter_list[m] ? p->adapter_list[m][i] : (i < SNK_MAX_ADAPTERS ? p->adapters[m][i] : (const char *)0);
}

/* fill with the reference defaults (src/global_parameter.h:20-83) */
void snk_params_default(snk_params *p);

/* ------------------------------------------------------------------- batch
 * Structure-of-arrays patch of reads.  Read i of mate m occupies
 * seq[m][i*pitch .. i*pitch+len) and qual[m][same]; pitch is a multiple of 4
 * and >= the longest read in the batch.  IDs and the '+' line never cross the
 * boundary.  For snk_filter_batch_device() every pointer is a device pointer;
 * for snk_filter_batch() every pointer is a host pointer.                    */
typedef struct snk_batch {
    int64_t n;                /* pairs (PE) or reads (SE) */
    int32_t pitch;            /* bytes between consecutive reads */
    int32_t fixed_len[2];     /* used when len[m]==NULL: all reads have this length */
    const uint8_t  *seq[2];   /* seq[1]/qual[1] ignored for SE */
    const uint8_t  *qual[2];
    const uint16_t *len[2];   /* per-read lengths or NULL */
    const uint8_t  *dup;      /* per-pair host verdicts or NULL: bit 0 duplicate (honoured when params.rmdup;
                                 from snk_rmdup_mark_device), bit 1 "in a filtered tile", bit 2 "in a filtered
                                 fov" (decided by the host from the read name, src/read_filter.cpp:14-150) */
    uint64_t first_index;     /* input-order index of pair 0 (for the "last read seen"
                                 semantics of gs.read_length, src/peprocess.cpp:1202) */
} snk_batch;

/* ---------------------------------------------------------- per-read result
 * 16 bytes per mate.  Cut fields are those of the filter's private copy of
 * the read after stat_read()+fastq_trim() (C_fastq::head_hdcut ...,
 * src/sequence.h:69); -1 == never set, as in C_fastq_init
 * (src/peprocess.cpp:1674-1689).                                              */
typedef struct snk_read_result {
    int16_t  head_hdcut, head_lqcut, tail_hdcut, tail_lqcut;
    int16_t  adacut_pos;      /* len - adapter position, or -1 */
    uint16_t clean_start;     /* first kept base  (0 when the read was emptied) */
    uint16_t clean_len;       /* kept length      (0 when the read was emptied) */
    uint8_t  reason;          /* SNK_KEEP or the discard reason of the pair/read */
    uint8_t  flags;           /* bits0-1: pe_dis() code (src/sequence.cpp:392) of the
                                 reason: 1=this is fq1-only,2=fq2-only,3=both; same
                                 value stored in both mates' records */
} snk_read_result;

enum snk_reason {             /* order == cascade order, src/sequence.cpp:198-387 */
    SNK_KEEP = 0,
    SNK_R_DUP = 1, SNK_R_TILE = 2, SNK_R_FOV = 3, SNK_R_SHORT = 4, SNK_R_LONG = 5,
    SNK_R_GCONTAM = 6, SNK_R_CONTAM = 7, SNK_R_NRATE = 8, SNK_R_HIGHA = 9,
    SNK_R_POLYX = 10, SNK_R_LOWQUAL = 11, SNK_R_MEANQ = 12, SNK_R_OVERLAP = 13,
    SNK_R_ADAPTER = 14,
    SNK_R_EMPTY = 15          /* PE, min_read_length==-1, a mate emptied: dropped
                                 without a counter, src/sequence.cpp:245-249 */
};

/* ------------------------------------------------------------------- stats
 * One flat block of uint64 that is a pure SUM over reads (this is what the
 * RCCL all-reduce carries) plus a tiny MAX block.
 *
 *   sum block  = fs[SNK_FS_N] | file[0] | file[1] | file[2] | file[3]
 *   file[k]    = gs[SNK_GS_N] | bs[lcap][5] | qs[lcap][nq] | ts[5][1000]
 *   k          : 0 raw fq1, 1 raw fq2, 2 clean fq1, 3 clean fq2
 *   nq         = max_base_quality + 1   (the report loops read j<=maxBaseQuality,
 *                src/peprocess.cpp:475)
 *   ts         : hlq,ht,ta,tlq,tt exactly in the member order of
 *                C_reads_trim_stat (src/global_variable.h:118-124) so that the
 *                reference's negative indices land where they do there.
 *   max block  = last_key[4]: ((first_index+i+1)<<16 | length) of the last read
 *                accumulated into file[k] -> gs.read_length of that virtual thread.
 */
#define SNK_FS_N 64
#define SNK_GS_N 16
#define SNK_TS_N 5000
#define SNK_MAX_N 8

enum snk_fs_index {           /* C_filter_stat, src/global_variable.h:13-87 */
    SNK_FS_DUP = 0, SNK_FS_TILE = 1, SNK_FS_FOV = 2, SNK_FS_OVERLAP = 3,
    /* families of 4: total, fq1, fq2, overlap (num, num1, num2, num_overlap) */
    SNK_FS_SHORT = 4, SNK_FS_LONG = 8, SNK_FS_GCONTAM = 12, SNK_FS_CONTAM = 16,
    SNK_FS_NRATE = 20, SNK_FS_HIGHA = 24, SNK_FS_POLYX = 28, SNK_FS_LOWQUAL = 32,
    SNK_FS_MEANQ = 36, SNK_FS_ADAPTER = 40
};
enum snk_gs_index {           /* C_general_stat, src/global_variable.h:88-100 */
    SNK_GS_READS = 0, SNK_GS_BASES = 1, SNK_GS_A = 2, SNK_GS_C = 3, SNK_GS_G = 4,
    SNK_GS_T = 5, SNK_GS_N_ = 6, SNK_GS_Q20 = 7, SNK_GS_Q30 = 8
};
enum snk_ts_index { SNK_TS_HLQ = 0, SNK_TS_HT = 1000, SNK_TS_TA = 2000,
                    SNK_TS_TLQ = 3000, SNK_TS_TT = 4000 };

static inline int64_t snk_file_block_u64(int lcap, int nq) {
    return (int64_t)SNK_GS_N + (int64_t)lcap * 5 + (int64_t)lcap * nq + SNK_TS_N;
}
static inline int64_t snk_stats_u64(int lcap, int nq) {
    return (int64_t)SNK_FS_N + 4 * snk_file_block_u64(lcap, nq);
}
static inline int64_t snk_file_off(int lcap, int nq, int k) {
    return (int64_t)SNK_FS_N + k * snk_file_block_u64(lcap, nq);
}
static inline int64_t snk_bs_off(int lcap, int nq) { (void)lcap; (void)nq; return SNK_GS_N; }
static inline int64_t snk_qs_off(int lcap, int nq) { (void)nq; return (int64_t)SNK_GS_N + (int64_t)lcap * 5; }
static inline int64_t snk_ts_off(int lcap, int nq) { return (int64_t)SNK_GS_N + (int64_t)lcap * 5 + (int64_t)lcap * nq; }

/* ------------------------------------------------------------------ errors
 * The reference prints "Error:<msg>" and exit(1)s from inside the path; the
 * library reports the same conditions as codes, the CLI prints and exits.   */
enum snk_error_code {
    SNK_OK = 0,
    SNK_E_BAD_BASE = 1,    /* "Error:unrecognized sequence,<seq>"  src/read_filter.cpp:283 */
    SNK_E_EMPTY_SEQ = 2,   /* "Error:empty sequence"               src/read_filter.cpp:251 */
    SNK_E_QUAL_RANGE = 3,  /* quality outside [0,max_base_quality]: UB in the reference
                              (src/peprocess.cpp:1196), refused here */
    SNK_E_TOO_LONG = 4,    /* read longer than params.max_read_len */
    SNK_E_PARAM = -1, SNK_E_HIP = -2, SNK_E_NOMEM = -3, SNK_E_UNSUPPORTED = -4
};
typedef struct snk_error {
    int32_t  code;          /* snk_error_code of the first (lowest-index) offending read */
    int32_t  mate;          /* 0 / 1 */
    uint64_t index;         /* first_index + i */
} snk_error;

/* --------------------------------------------------------------------- API */
typedef struct snk_ctx snk_ctx;

/* Create a context on HIP device `device` (params are copied; adapter strings
 * too).  Returns NULL on failure; see snk_last_error().                      */
snk_ctx *snk_create(const snk_params *params, int device);
void     snk_destroy(snk_ctx *ctx);
const char *snk_last_error(void);

/* geometry of the stats block of this context */
int snk_stats_geometry(const snk_ctx *ctx, int32_t *lcap, int32_t *nq, int64_t *sum_u64);

/* Bind caller-owned device memory as the accumulators (sum: snk_stats_u64()
 * uint64, max: SNK_MAX_N uint64).  Lets a host framework (e.g. a torch int64
 * tensor handed to torch.distributed/RCCL) own the all-reduce buffer.  Without
 * this call the context allocates its own.  Does not clear the memory.       */
int snk_bind_stats(snk_ctx *ctx, void *d_sum, void *d_max);
int snk_stats_clear(snk_ctx *ctx, void *hip_stream);

/* Run the hot path on one device-resident patch: per-read records into
 * out[m] (device, n*16 bytes each; out[1] ignored for SE), stats accumulated
 * into the bound block.  Asynchronous on `hip_stream` (hipStream_t, NULL =
 * default stream).  kernel: 0 = auto, 1 = generic decisions (any read length,
 * any adapter) + the LDS histogram kernel, 2 = fast paths only (wave-tiled
 * up to 256 positions, block-wise beyond; error if neither takes the
 * configuration), 3 = the generic kernel alone (the anchor: one global
 * atomic per base and quality).                                              */
int snk_filter_batch_device(snk_ctx *ctx, const snk_batch *batch,
                            snk_read_result *d_out1, snk_read_result *d_out2,
                            void *hip_stream, int kernel);

/* Same with host pointers: stages H2D, runs, copies records back, syncs.    */
int snk_filter_batch(snk_ctx *ctx, const snk_batch *batch,
                     snk_read_result *out1, snk_read_result *out2);

/* Finalise gs[] from the histograms (device side) and copy both blocks to the
 * host; synchronises `hip_stream`.  `err` receives the first data error.     */
int snk_stats_finalize(snk_ctx *ctx, void *hip_stream);
int snk_stats_fetch(snk_ctx *ctx, uint64_t *sum, uint64_t *max, snk_error *err,
                    void *hip_stream);

/* The context's error word (first data error so far: all-ones = none) copied asynchronously on
 * `hip_stream` into *host_word (pinned host memory): lets a host poll per patch and stop before
 * it writes anything, as the reference exits at the offending read (src/read_filter.cpp:251,283).
 * snk_error_decode() turns the word into a snk_error (code SNK_OK when none).               */
#define SNK_ERR_WORD_NONE 0xFFFFFFFFFFFFFFFFull
int  snk_error_peek_async(snk_ctx *ctx, uint64_t *host_word, void *hip_stream);
void snk_error_decode(uint64_t word, snk_error *err);

/* In-place sum/max all-reduce of the bound blocks over an RCCL communicator
 * (ncclComm_t) -- the only collective on this path.             */
int snk_stats_allreduce(snk_ctx *ctx, void *nccl_comm, void *hip_stream);

/* Sizes the per-stream scratch the launch path would otherwise grow on demand (a hipMalloc + a stream synchronisation in the middle
 * of a run: contaminant verdicts, the long-read plane store, the tiled kernel's histogram partials) for batches of up to max_pairs
 * pairs on up to n_streams launching streams (1..8).  Optional: without it the first large batch of a stream pays for the growth. */
int snk_reserve(snk_ctx *ctx, int64_t max_pairs, int n_streams);

/* ms spent in the hot-path kernels of the last snk_filter_batch_device() call
 * measured with hipEvents on the launch stream (0 when timing is disabled).  */
int snk_set_timing(snk_ctx *ctx, int enabled);
int snk_last_kernel_ms(snk_ctx *ctx, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* SNK_FILTER_H */
