/*
 * snk_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's `filter` per-read hot path, written
 * against the same SoA batch / params / stats layout as include/snk_filter.h so
 * that its outputs can be memcmp'd with the HIP path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Parity pin: every function here is checked against the compiled reference
 * itself (oracle/_ref/libsnkref.so built from /root/reference/src by
 * oracle/Makefile) in tests/test_oracle_vs_ref.py, and against the golden
 * vectors generated from it under tests/golden/.
 */
#ifndef SNK_ORACLE_H
#define SNK_ORACLE_H
#include "../include/snk_filter.h"
#ifdef __cplusplus
extern "C" {
#endif

/* adapter_pos(), src/read_filter.cpp:707-790 */
int snk_oracle_adapter_pos(const uint8_t *read, int read_len,
                           const char *adapter, int adapter_len,
                           int ada_mis, float ada_mr, int ada_edge);

/* hasContam(), src/read_filter.cpp:507-603 (and its 3-argument twin :604-700): position or -1.
 * seg_match_thr = (int)ceil(contamLen * ratio) is computed by the caller (float for list entries,
 * double for a single contaminant, as the two overloads do).                                    */
int snk_oracle_has_contam(const uint8_t *read, int read_len, const char *contam, int contam_len,
                          int seg_match_thr, int ada_mis, int ada_edge);
/* global_contam_pos(), src/read_filter.cpp:961-1062 */
int snk_oracle_global_contam_pos(const uint8_t *read, int read_len, const char *contam, int contam_len,
                                 float min_match_ratio, int mismatch_number);

/* polyG_number(), src/read_filter.cpp:472-482 */
int snk_oracle_polyG_number(const uint8_t *read, int read_len);

/* One patch: filter_pe_fqs/filter_se_fqs + stat("raw") + stat("clean").
 * sum (snk_stats_u64(max_read_len, max_base_quality+1) u64) and maxb
 * (SNK_MAX_N u64) are accumulated into, not cleared.  All pointers are host
 * pointers.  Returns SNK_OK or the first data error (also in *err).          */
int snk_oracle_filter_batch(const snk_params *P, const snk_batch *B,
                            snk_read_result *out1, snk_read_result *out2,
                            uint64_t *sum, uint64_t *maxb, snk_error *err);

/* ---- rmdup pre-pass -------------------------------------------
 * The reference hashes seq1+seq2 of every raw pair with std::hash<std::string>
 * (src/peprocess.cpp:3665-3681; SE: the read alone, src/seprocess.cpp:2545), i.e.
 * libstdc++'s _Hash_bytes (libstdc++-v3/libsupc++/hash_bytes.cc, 64-bit variant:
 * seed 0xc70f6907, mul 0xc6a4a7935bd1e995, shift_mix(v) = v ^ (v >> 47); not under
 * /root/reference -- pinned by GCC 11.4 here and by the known answers of SURVEY 8(c)),
 * then marks every later occurrence of a hash value (rmdup::markDup,
 * src/rmdup.cpp:14-149).                                                        */
uint64_t snk_oracle_hash_bytes(const void *p, uint64_t len);
/* hash of every pair (mate 1 ++ mate 2; SE: mate 1) of a host batch */
void snk_oracle_hash_batch(const snk_batch *B, int paired, uint64_t *out);
/* rmdup::getPrime (src/rmdup.cpp:150-185): n for n < 10, else the largest prime < n */
uint32_t snk_oracle_rmdup_prime(uint64_t n);
/* rmdup::markDup: dup[i] = 1 iff hash[i] occurred at an earlier index, plus the
 * reference's sentinel quirk: it overwrites later occurrences with (uint64_t)-1 in
 * its bucket copy, so a genuine hash of 2^64-1 is flagged -- first occurrence
 * included -- whenever its bucket (hash % prime) holds more than one element.   */
void snk_oracle_markdup(const uint64_t *hash, uint64_t n, uint8_t *dup);

/* derive gs a/c/g/t/n/q20/q30/bases from the histograms exactly like the
 * product's finalize kernel does (used to cross-check that derivation).      */
void snk_oracle_params_default(snk_params *p);

#ifdef __cplusplus
}
#endif
#endif
