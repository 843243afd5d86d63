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

/* polyG_number(), src/read_filter.cpp:472-482 */
int snk_oracle_polyG_number(const uint8_t *read, int read_len);

/* One patch: filter_pe_fqs/filter_se_fqs + stat("raw") + stat("clean").
 * sum (snk_stats_u64(max_read_len, max_base_quality+1) u64) and maxb
 * (SNK_MAX_N u64) are accumulated into, not cleared.  All pointers are host
 * pointers.  Returns SNK_OK or the first data error (also in *err).          */
int snk_oracle_filter_batch(const snk_params *P, const snk_batch *B,
                            snk_read_result *out1, snk_read_result *out2,
                            uint64_t *sum, uint64_t *maxb, snk_error *err);

/* derive gs a/c/g/t/n/q20/q30/bases from the histograms exactly like the
 * product's finalize kernel does (used to cross-check that derivation).      */
void snk_oracle_params_default(snk_params *p);

#ifdef __cplusplus
}
#endif
#endif
