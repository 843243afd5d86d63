/* snk_rmdup.h -- C ABI of the duplicate-marking pre-pass of `SOAPnuke filter` (config key
 * `rmdup`, SURVEY 8(f) N1, BASELINE config 5), MI355X-native.
 *
 * What it replaces in the reference:
 *   - peProcess::sub_thread_rmdup_step1 (src/peprocess.cpp:3609-3807; SE: seProcess,
 *     src/seprocess.cpp:2480-2650): every raw pair is hashed with
 *     std::hash<std::string>(seq1 + seq2)  (src/peprocess.cpp:3665,3680) -- libstdc++'s
 *     _Hash_bytes, 64-bit, seed 0xc70f6907 -- into one uint64 per pair, in input order
 *     (src/peprocess.cpp:3093-3124);
 *   - rmdup::markDup (src/rmdup.cpp:14-149): dupFlag[i] = the hash of pair i occurred at an
 *     earlier index (hash equality IS the duplicate criterion; collisions are duplicates),
 *     plus its (uint64_t)-1 sentinel quirk (see snk_rmdup_mark_device);
 *   - the flags then enter the discard cascade as `dup` (snk_batch.dup, src/sequence.cpp:207).
 *
 * Both entry points work on device memory and are asynchronous on `stream`.  The hash array of a
 * whole run stays resident in HBM (8 B per pair: 1.6 GB for the 200 M pairs of config 5).
 * Multi-GPU: ranks hash their own shard, exchange (hash, global index) by owner = hash % world
 * (the one collective of this row, an all-to-all over RCCL, driven by soapnuke_amd/shard.py),
 * mark locally with explicit global indices and send the flags back.
 */
#ifndef SNK_RMDUP_H
#define SNK_RMDUP_H
#include "snk_filter.h"

#ifdef __cplusplus
extern "C" {
#endif

/* d_hash[i] = std::hash<std::string>(mate1_i ++ mate2_i)  (SE contexts: mate1_i alone) for the
 * batch->n pairs of a device-resident batch (same layout as snk_filter_batch_device; qualities
 * are not read).  Returns SNK_OK or a negative error (snk_last_error).                        */
int snk_rmdup_hash_device(snk_ctx *ctx, const snk_batch *batch, uint64_t *d_hash, void *stream);

/* Number of elements whose hash lies in the bucket of (uint64_t)-1, i.e.
 * hash % prime == (2^64-1) % prime with prime = rmdup::getPrime(total_n) (src/rmdup.cpp:150).
 * Written to *d_count (device, uint64).  Only needed to reproduce the sentinel quirk across
 * ranks; a single-GPU caller passes sentinel_bucket_total = -1 to snk_rmdup_mark_device.      */
int snk_rmdup_bucket_count_device(snk_ctx *ctx, const uint64_t *d_hash, int64_t n, uint64_t total_n,
                                  uint64_t *d_count, void *stream);

/* d_dup[i] = 1 iff some j has d_hash[j] == d_hash[i] and index_j < index_i, where index_k =
 * d_index[k] (global input-order indices, all distinct) or k when d_index is NULL.
 * Sentinel quirk of the reference (src/rmdup.cpp:100,116): elements whose hash is 2^64-1 are
 * flagged -- the first one included -- iff their bucket holds more than one element;
 * sentinel_bucket_total is that bucket's population over all ranks, or -1 to count it here
 * (single GPU).  total_n = pairs hashed in the whole run (all ranks); the reference refuses
 * more than 2^32-1 (src/peprocess.cpp:3094) and so does this call (SNK_E_PARAM).
 * The call allocates and frees its hash table (16 B x 2..4 n) on the device.                  */
int snk_rmdup_mark_device(snk_ctx *ctx, const uint64_t *d_hash, const uint32_t *d_index, int64_t n,
                          uint64_t total_n, int64_t sentinel_bucket_total, uint8_t *d_dup, void *stream);

/* ---- one pass (round 3).  "An equal hash at an earlier index" only needs the reads in front, so the marking can follow the
 * input batch by batch with a hash table that lives across the calls -- no pre-pass over the whole input, which the reference
 * needs for its sort-free bucket scheme (src/peprocess.cpp:3071-3152 reads and inflates everything twice).
 * snk_rmdup_stream_mark_device(): inserts the n hashes of a batch (global indices first_index + i) and writes
 * d_dup[i] = 1 iff the hash was seen at a smaller index (in an earlier call or in this one).  Calls must be issued in input
 * order; they are asynchronous on `stream`, and a call on another stream than the previous one waits for it on the device.
 * The table grows by itself (all hashes stay resident, 8 B per pair).  The reference's (uint64_t)-1 sentinel quirk depends
 * on the total number of reads: snk_rmdup_stream_stats() reports whether such a hash occurred (the caller then has to fall
 * back to the two-pass calls above) together with the number of duplicates marked so far.                                  */
typedef struct snk_rmdup_stream snk_rmdup_stream;
snk_rmdup_stream *snk_rmdup_stream_create(snk_ctx *ctx, uint64_t expected_pairs);
int snk_rmdup_stream_mark_device(snk_rmdup_stream *t, const uint64_t *d_hash, uint64_t first_index, int64_t n, uint8_t *d_dup, void *stream);
/* Single-end runs.  The reference's SE filter reads i of a FULL patch with the duplicate flag of read i - 1 (it records "reads so
 * far" before it counts the patch's last quality line, src/seprocess.cpp:1086,1159; PE: src/peprocess.cpp:2147); only the partial
 * patch at the end of the file is aligned (:1112).  This call marks like snk_rmdup_stream_mark_device and writes the flags the
 * reference's cascade sees: d_dup[i] = true flag of read i - 1 for i < n_shifted (the flag of the last read of the previous call
 * for i = 0; 0 in the first call), the true flag of read i behind that.  A caller that cuts its batches at multiples of the patch
 * size passes n_shifted = n for every batch but a last one that ends inside a patch (n / patch * patch there).              */
int snk_rmdup_stream_mark_se_device(snk_rmdup_stream *t, const uint64_t *d_hash, uint64_t first_index, int64_t n, int64_t n_shifted,
                                    uint8_t *d_dup, void *stream);
int snk_rmdup_stream_stats(snk_rmdup_stream *t, uint64_t *n_marked, int32_t *sentinel_seen);     /* synchronises */
void snk_rmdup_stream_destroy(snk_rmdup_stream *t);
/* device bytes a one-pass table for `pairs` pairs holds at its largest (resident hashes 8 B per pair + the open-addressing table,
 * 12 B per slot at load 0.25..0.5): lets the caller choose the memory-lean two-pass calls up front.  A call that cannot get its
 * memory returns SNK_E_NOMEM and leaves the table unusable (later calls repeat the code): fall back to the two passes then.  */
uint64_t snk_rmdup_stream_bytes(uint64_t pairs);

/* ---- multi-GPU exchange helpers (round 5; SURVEY 8e: "all-to-all keyed by hash % G").  The reference's rmdup::markDup needs the
 * GLOBAL input order (src/rmdup.cpp:70-123), so with the input sharded over G devices every hash travels to its owner
 * (hash % G) together with its global index, the owner marks with explicit indices (snk_rmdup_mark_device) and the flags travel
 * back.  snk_rmdup_partition_device() groups the n hashes of a shard by owner: d_send_hash / d_send_index hold the elements of
 * owner 0, then of owner 1, ... (the order inside a group is unspecified; the global index first_index + i rides along as
 * uint32 -- the reference's own limit), h_counts[world] (host memory) receives the group sizes and d_slot[i] the place of
 * element i in that order; the call synchronises `stream`.  snk_rmdup_flags_home_device(): d_dup[i] = d_back[d_slot[i]], where
 * d_back are the owners' flags in send order.  The exchange itself (RCCL ncclSend / ncclRecv groups, or a host wire) is the
 * caller's: soapnuke_amd/host/snk_wire.h for the CLI's shards, soapnuke_amd/shard.py over torch.distributed.              */
int snk_rmdup_partition_device(snk_ctx *ctx, const uint64_t *d_hash, int64_t n, uint64_t first_index, int32_t world,
                               uint64_t *d_send_hash, uint32_t *d_send_index, uint32_t *d_slot, uint64_t *h_counts, void *stream);
int snk_rmdup_flags_home_device(snk_ctx *ctx, const uint8_t *d_back, const uint32_t *d_slot, int64_t n, uint8_t *d_dup, void *stream);

/* rmdup::getPrime(n) (host helper; 0 for n == 0 where the reference exits with "code error") */
uint32_t snk_rmdup_prime(uint64_t n);

#ifdef __cplusplus
}
#endif
#endif
