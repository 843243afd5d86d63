// snk_report.h -- stats merge + report writers of `SOAPnuke filter` (host side, plain C++).
//
// Counterpart of merge_stat/update_stat/print_stat of the reference
// (src/peprocess.cpp:178-1075,1994-2005; src/seprocess.cpp:96-630; src/gc.cpp:68-119).
// Input: one stats block (include/snk_filter.h layout) per *virtual reference thread*: the
// reference's reports depend on how reads were dealt to its -T threads (SURVEY quirk Q3,
// appendix C), so the host keeps one accumulator per virtual thread and replays the fold.
#ifndef SNK_REPORT_H
#define SNK_REPORT_H
#include <stdint.h>
#include "../../include/snk_filter.h"

#ifdef __cplusplus
extern "C" {
#endif

// Writes the 10 (PE) / 6 (SE) report files into out_dir.  sums[t] / maxs[t]: the sum and max
// blocks of virtual thread t (t = 0..n_threads-1), geometry (params->max_read_len,
// params->max_base_quality+1).  Returns 0, or -1 with a message in err (cap bytes).
int snk_write_reports(const snk_params *params, int n_threads, const uint64_t *const *sums,
                      const uint64_t *const *maxs, const char *out_dir, char *err, int cap);

// Virtual reference thread of input-order pair index r for -T threads and the given patchSize
// (0 = default threads*2500): block = patchSize * (160/threads) pairs, owner = (r/block) % threads
// (src/peprocess.cpp:81,2063,2092; src/process_argv.cpp:541-544).
int64_t snk_vthread_block(int threads, int patch_size);

// -j / --streaming: the cumulative statistics of one (virtual) reference thread in the text form that
// peStreaming_stat / seStreaming_stat print behind every patch (src/peprocess.cpp:3485-3594,
// src/seprocess.cpp:2405-2462), appended to *out (a std::string passed as void*).
void snk_streaming_stat_text(const snk_params *params, const uint64_t *sum, const uint64_t *max, void *out_string);

// cal_quar_from_array (src/gc.cpp:68-119) on one histogram row of nq counters: mean, median, lower, upper,
// first10, last10 -- with its 32-bit positions.  Returns a bit mask of the fields the scan assigned
// (an unassigned field is uninitialised memory in the reference, 0 here).  Exported for the tests.
int snk_report_quartiles(const uint64_t *data, int nq, int len, float out[6]);

#ifdef __cplusplus
}
#endif
#endif
