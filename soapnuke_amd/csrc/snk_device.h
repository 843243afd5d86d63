// snk_device.h -- structures shared by the host side of the C ABI (snk_filter.cpp)
// and the gfx950 kernels (snk_generic.hip, snk_tiled.hip, snk_rmdup.hip).  Internal; the public surface is
// include/snk_filter.h.
#pragma once
#include <stdint.h>
#include "../../include/snk_filter.h"

// SNK_WAVE_SYNC(): lanes of one wave hand data to each other through LDS or memory at the marked places without a workgroup
// barrier.  A wave executes in lock-step and its memory operations complete in order, so on the device there is nothing to emit;
// the CPU test tier (tests/simt), which runs the lanes of a wave one after the other, defines it as the point where they wait
// for each other.
#ifndef SNK_WAVE_SYNC
#if defined(__HIP_DEVICE_COMPILE__)
// compiler-level ordering only: a wavefront-scope fence pair around a wave barrier emits no instruction on gfx950, but it stops
// LLVM from moving one lane's LDS / memory load across another lane's store at this point
#define SNK_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                             __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#else
#define SNK_WAVE_SYNC() ((void)0)
#endif
#endif

#define SNK_DEV_MAX_ADA_LEN 256

// One adapter with everything adapter_pos() (src/read_filter.cpp:707-790) derives
// from (adptLen, adaMis, adaMR, adaEdge) precomputed on the host with the
// reference's own int->float->int arithmetic, so the GPU does integer work only
// (SURVEY H1).
struct DevAdapter {
    int32_t len;            // adptLen
    int32_t S;              // segMatchThr = (int)ceil(adptLen*adaMR), src/read_filter.cpp:717
    int32_t mis;            // adaMis (phase B budget)
    int32_t edge;           // adaEdge
    int32_t nC;             // adptLen - adaEdge (phase C trip count, may be <= 0)
    int32_t budgetA[6];     // [r1] = (int)((adptLen-r1)/misGrad5), r1 = 1..5   (:724)
    int32_t budgetC[SNK_DEV_MAX_ADA_LEN]; // [r1] = (int)(r1/misGrad)           (:769)
    uint8_t seq[SNK_DEV_MAX_ADA_LEN];
    // ---- bit-parallel view used by the wave-tiled kernel (adapters <= 64 nt) ----
    uint64_t cmask[4];      // cmask[k] bit c = adapter[c] == "ACGT"[k] (exact, upper case)
    uint8_t  code[SNK_DEV_MAX_ADA_LEN]; // 0..3 = ACGT, 4 = can never match a valid read base
    int32_t  tile_ok;       // 1 when the tiled kernel can handle this adapter
    int32_t  maxBudget;     // max over all phases of max(budget,0)
    int32_t  rk[4];         // rk[k] = smallest phase-C r1 with budgetC[r1] >= k (nC if none), k = 1..3; rk[0]: k = 4
    int32_t  negC;          // phase-C budgets are INT_MIN (misGrad == 0): only a run of S accepts
    uint64_t nmask;         // bit c = adapter[c] == 'N' (matches a read 'N' exactly)
    int32_t  has_lower;     // lower-case characters: they match lower-case read characters only (src/read_filter.cpp:728 compares bytes), which the
                            // letter planes do not hold -- a read with anything but upper-case ACGT takes the sequential matcher for this adapter
    int32_t  long_ok;       // ... and the block-wise search of the long-read kernel (6..64 characters, adaEdge <= length)
};

// One hasContam() contaminant (src/read_filter.cpp:507-603): the per-r1 thresholds of its head and
// tail loops precomputed on the host with the reference's int->float->int arithmetic.
#define SNK_MAX_CONTAMS 8
struct DevContam {
    int32_t len;            // contamLen
    int32_t S;              // segMatchThr of the middle loop (:549-573)
    int32_t mis;            // gp.adaMis
    int32_t edge;           // gp.adaEdge
    int32_t nC;             // contamLen - adaEdge: trip count of the head (:523) and tail (:575) loops
    int32_t mm[SNK_DEV_MAX_ADA_LEN];   // misMatchTemp(r1) = (int)(r1/misGrad)
    int32_t sm1[SNK_DEV_MAX_ADA_LEN];  // segMatchTemp(r1) of the head loop (7 when segGrad == 0)
    int32_t sm3[SNK_DEV_MAX_ADA_LEN];  // segMatchTemp(r1) of the tail loop (no such guard there)
    uint8_t seq[SNK_DEV_MAX_ADA_LEN];
    // ---- bit-parallel view (snk_contam.hip; contaminants of 1..64 upper-case ACGTN characters, adaEdge >= 1)
    uint64_t cm[4], nm;     // cm[k] bit c = seq[c] == "ACGT"[k]; nm bit c = seq[c] == 'N'
    int32_t bits_ok;        // the bit-parallel matcher covers this contaminant
    int32_t scr;            // cells screened for some alignment: max over the alignments of max(segMatch threshold, 1) - 1, at most 63
    int32_t bmax;           // largest mismatch budget of any alignment (>= 0)
    int32_t pad_;
    // monotone envelopes of the tail loop's per-r1 thresholds (r1 = overlap - adaEdge; the tail alignment with r1 starts at
    // read offset len - adaEdge - r1): cell c is screened for the alignments with r1 >= rT[c], the budget is >= b for
    // r1 >= rk[b] (nC: none), b = 1..3; rk[0]: b = 4
    int32_t rT[64], rk[4];
};
// One global contaminant (src/read_filter.cpp:927-1062): forward and reverse-complement strand.
struct DevGContam {
    int32_t len, min_match_len, mm;
    int32_t bits_ok;        // the sliding-count screen of snk_contam.hip covers it (4 <= min_match_len <= len <= 64, < 64, ACGTN)
    int32_t g;              // (unused)
    int32_t pad_;
    uint64_t cm[2][4], nm[2];   // per strand, as DevContam::cm / nm
    uint8_t seq[2][SNK_DEV_MAX_ADA_LEN];
};

// Compact adapter descriptor for the wave-tiled kernel.  It travels BY VALUE in the kernel
// argument segment so that every field sits in SGPRs: fetching DevAdapter fields from global
// memory inside the screening loop cost a memory round trip per adapter character.
#define SNK_TILE_MAX_ADA 4
struct TileAdapter {
    uint64_t cmask[4], nmask;   // as DevAdapter
    uint64_t code4[4];          // 4 bits per adapter position (DevAdapter::code), 16 per word
    int32_t len, S, mis, edge, negC;
    int32_t budgetA[6];
    int32_t rk[4];
    int32_t maxb;               // DevAdapter::maxBudget over phases B and C (the screen needs maxb + 1 counter planes)
    int32_t has_lower;          // DevAdapter::has_lower
};
struct TileAdapters { TileAdapter a[2][SNK_TILE_MAX_ADA]; };

struct DevParams {
    int32_t paired, phred, nq, low_qual;
    int32_t polyX_num;                 // -1 off
    uint32_t min_len_u, max_len_u;     // unsigned compares as in src/sequence.cpp:233,251
    int32_t has_min, has_max, has_n, has_highA, has_lowq, has_meanq;
    int32_t ada_trim, copy_back, trim_on, has_hard, has_lq, has_polyG, rmdup;
    int32_t hard[4];
    int32_t lq_head_q, lq_head_len, lq_tail_q, lq_tail_len;
    int32_t polyG_thr;                 // min n with (float)n >= polyG_tail  (:456)
    int32_t lcap;                      // positions per histogram row block
    int32_t n_ada[2];
    int32_t ada_stride;                // adapters of mate m: ada[m * ada_stride + i], tile_ada likewise
    int32_t tile_ok;                   // every adapter can run in the wave-tiled kernel
    int32_t long_ok;                   // ... and in the long-read kernel (snk_long.hip)
    int32_t need_n;                    // some adapter contains 'N' (needs the N plane)
    // per-length integer thresholds replacing the fp32 ratio compares:
    //   discard iff count >= thr_x[len]            (n_ratio, highA, low-quality ratio)
    //   discard iff sumq  <  thr_meanq[len]        (mean quality)
    const int32_t *thr_n, *thr_a, *thr_lowq, *thr_meanq;
    const DevAdapter *ada;             // [2][ada_stride]
    const TileAdapter *tile_ada;       // [2][ada_stride]: the compact descriptors of all adapters (the first SNK_TILE_MAX_ADA of a
                                       // mate also travel in the kernel arguments)
    // contaminant screening (generic kernel only; tile_ok is 0 when any is configured)
    int32_t n_ct[2], n_gct, contam_discard;
    const DevContam *ct;               // [2][SNK_MAX_CONTAMS]
    const DevGContam *gct;             // [SNK_MAX_CONTAMS]
};

// device view of one patch (see snk_batch)
struct DevBatch {
    int64_t n;
    int32_t pitch;
    int32_t fixed_len[2];
    const uint8_t *seq[2];
    const uint8_t *qual[2];
    const uint16_t *len[2];
    const uint8_t *dup;
    const uint8_t *cf;          // tiled kernel: contaminant verdicts from snk_contam_kernel (bits 0-1 mate 1, bits 2-3 mate 2) or null
    uint64_t first_index;
    snk_read_result *out[2];
};

struct DevStats {
    unsigned long long *sum;   // snk_stats_u64(lcap,nq)
    unsigned long long *maxb;  // SNK_MAX_N
    unsigned long long *err;   // 1 word: min over offending reads of (index<<8 | mate<<4 | code)
    unsigned *tsw;             // tiled kernel: one private uint32 copy of the 4 x SNK_TS_N trimming-position
                               // counters per workgroup (n_cu of them), drained into `sum` at the end of a launch
    unsigned *part;            // tiled kernel: every workgroup's flushed histogram words (snk_tiled_part_bytes(); zero between launches),
                               // summed into `sum` by a small kernel behind it
};

#define SNK_ERR_NONE 0xFFFFFFFFFFFFFFFFull

// own_hist: 1 = the kernel adds its per-position histograms itself (the anchor), 0 = snk_launch_hist() follows
void snk_launch_generic(const DevParams *dp, const DevBatch &b, const DevStats &st, int lcap,
                        int nq, int own_hist, void *stream);
// per-position raw / clean histograms from the records of a decision kernel (snk_long.hip); returns 0 when it cannot run
int snk_launch_hist(const DevParams *dp, int paired, const DevBatch &b, const DevStats &st, int lcap, int nq, int n_cu, void *stream);
// returns 0 when the tiled kernel cannot run this configuration
int snk_launch_tiled(const DevParams &dp_host, const TileAdapters &ta, const DevBatch &b,
                     const DevStats &st, int lcap, int nq, int n_cu, void *stream);
size_t snk_tiled_part_bytes(int lcap, int nq, int n_cu);      // DevStats::part of a stream slot
void snk_launch_finalize(const DevStats &st, int lcap, int nq, void *stream);
// reads of 257..1024 positions (snk_long.hip); returns 0 when it cannot take the batch
// (`planes`: snk_long_scratch_bytes(n, paired, lcap) bytes of scratch the launch owns until it has run; `cf`: n bytes for the
// contaminant verdicts when contaminants are configured, else null)
int snk_launch_long(const DevParams *dp, const DevParams &hp, const TileAdapters &ta, const DevBatch &b, const DevStats &st, int lcap, int nq,
                    int n_cu, unsigned *planes, unsigned char *cf, void *stream);
size_t snk_long_scratch_bytes(long n, int paired, int lcap);
// contaminant verdicts of a batch (one work-item per pair) into cf[n], for the tiled kernel
void snk_launch_contam(const DevParams *dp, const DevBatch &b, unsigned char *cf, int lcap, int n_ct, int n_gct, void *stream);   // snk_contam.hip
// the same for reads of 257..1024 positions, block-wise from the plane store of snk_launch_long (which calls it)
void snk_launch_long_contam(const DevParams *dp, const DevBatch &b, unsigned char *cf, int n_ct, int n_gct, const unsigned *planes, int nquads, void *stream);

// rmdup pre-pass (snk_rmdup.hip); return 0 or a hipError_t
int snk_launch_hash(const uint8_t *const seq[2], const uint16_t *const len[2], const int fixed_len[2], int pitch, long n,
                    int paired, unsigned long long *out, int n_cu, void *stream);
int snk_launch_bucket_count(const unsigned long long *hash, long n, unsigned prime, const unsigned *flag,
                            unsigned long long *count, void *stream);
int snk_launch_bittr_selftest(const unsigned *d_in, int n_matrices, unsigned *d_out, unsigned *d_out_lo);   // snk_tiled.hip
int snk_launch_mark(const unsigned long long *hash, const unsigned *index, long n, unsigned prime, long bucket_total,
                    unsigned char *dup, void *stream);
