/*
 * snk_fastq.h -- C ABI of the device-side FASTQ ingest / egress around the filter hot path
 * (the step either side of snk_filter_batch_device()).
 *
 * What it replaces in the reference:
 *   - the record loop of the reading thread: four gzgets() lines per read, the trailing
 *     `spaceNum` characters of every line dropped, fields copied into C_fastq objects
 *     (src/peprocess.cpp:2063-2113; SE: src/seprocess.cpp:1040-1112)
 *       -> snk_fastq_parse_device(): raw FASTQ text in HBM -> line index + the SoA planes of
 *          snk_batch (seq / qual / len), checked (four lines per record, |seq| == |qual|);
 *   - output_fastqs() + preOutput(): id, [pe_info suffix], bases of the kept range, "+", qualities
 *     of the kept range re-based to outQualSys, baseConvert applied
 *     (src/peprocess.cpp:3383-3484, 1617-1647)
 *       -> snk_fastq_format_device(): records of the filter kernels + the same text in HBM ->
 *          the clean FASTQ text of one mate, contiguous, kept reads in input order, plus the
 *          offset of every read's record in it (so the host can cut it into gzip members).
 *
 * Everything is device memory and asynchronous on `stream`; nothing here needs a snk_ctx.
 * Plain C, plain pointers and sizes.
 */
#ifndef SNK_FASTQ_H
#define SNK_FASTQ_H
#include "snk_filter.h"

#ifdef __cplusplus
extern "C" {
#endif

/* status words written by snk_fastq_parse_device (uint32_t[SNK_FQ_STATUS_N], zeroed by the call) */
#define SNK_FQ_STATUS_N 4
#define SNK_FQ_ST_FLAGS 0        /* OR of SNK_FQ_F_* */
#define SNK_FQ_ST_MAXLEN 1       /* longest sequence line of the batch (after the trailing characters were dropped) */
#define SNK_FQ_ST_LINES 2        /* '\n' characters found in the text */
#define SNK_FQ_ST_BADREC 3       /* smallest record index with a length mismatch (valid with SNK_FQ_F_LEN_MISMATCH) */
#define SNK_FQ_F_LEN_MISMATCH 1u /* a record's sequence and quality lines differ in length */
#define SNK_FQ_F_TOO_LONG 2u     /* a sequence line is longer than `lcap`: nothing of that record was copied */
#define SNK_FQ_F_TRUNCATED 4u    /* the text holds fewer than 4 * n_records lines */

/* scratch bytes snk_fastq_parse_device / snk_fastq_format_device need for a text of at most `max_bytes` bytes and
 * at most `max_records` records */
size_t snk_fastq_tmp_bytes(uint64_t max_bytes, int64_t max_records);

/* d_text[0, n_bytes): FASTQ text that starts at a record start and holds n_records whole records (the last line may
 * lack its '\n').  Every line loses its last `space_num` characters, terminator included (the reference measures them
 * on the first line of fq1: 1 for "\n", 2 for "\r\n", src/peprocess.cpp:2066-2077).
 *   d_line[4 * n_records + 1]  start offset of every line; line k ends at d_line[k + 1] - space_num
 *   d_seq / d_qual             read i at [i * pitch, i * pitch + len), as snk_batch wants them (pitch a multiple of 4)
 *   d_len[n_records]           sequence lengths
 *   d_status                   see SNK_FQ_ST_*
 * n_bytes < 2^32 - 64; d_text is 16-byte aligned and its allocation extends at least 16 bytes past n_bytes (the kernels
 * read whole dwords around unaligned line starts).  Returns SNK_OK or a negative snk_status (snk_last_error() has the text). */
int snk_fastq_parse_device(const uint8_t *d_text, uint64_t n_bytes, int64_t n_records, int32_t space_num,
                           int32_t pitch, int32_t lcap, uint8_t *d_seq, uint8_t *d_qual, uint16_t *d_len,
                           uint32_t *d_line, uint32_t *d_status, void *d_tmp, size_t tmp_bytes, void *stream);

typedef struct snk_fastq_format {
    int32_t struct_size;        /* = sizeof(snk_fastq_format) */
    int32_t space_num;          /* as given to snk_fastq_parse_device */
    int32_t qual_delta;         /* outputQualityPhred - qualityPhred, added to every quality character (src/peprocess.cpp:3398-3405) */
    int32_t id_suffix_times;    /* how often `id_suffix` is appended to the id line (pe_info: "/1" or "/2", once per
                                   preOutput() call: src/peprocess.cpp:1617-1628) */
    char    id_suffix[4];       /* NUL-terminated, at most 3 characters */
    uint8_t base_from, base_to; /* baseConvert: every base whose upper case is `base_from` becomes `base_to`; 0 = off
                                   (src/peprocess.cpp:1629-1646) */
    uint8_t select_reason;      /* the records written are those with d_keep[i].reason == select_reason (0 = SNK_KEEP: the clean text) */
    uint8_t whole_read;         /* non-zero: the whole sequence / quality lines, not the kept range (with select_reason =
                                   SNK_R_DUP: C_fastq::toString of the raw duplicates, src/peprocess.cpp:1541) */
} snk_fastq_format;

/* Clean text of one mate: for every record i with d_keep[i].reason == fmt->select_reason (SNK_KEEP; the pair verdict sits
 * in both mates' records; pass mate 1's),
 *      <id line><suffix...>\n<seq[clean_start, +clean_len)>\n+\n<qual[clean_start, +clean_len) + qual_delta>\n
 * with clean_start / clean_len from d_rec[i] (this mate's records), appended in input order.
 *   d_out_off[n + 1]  offset of record i's text in d_out (records that are not kept have length 0); d_out_off[n] =
 *                     total bytes
 * d_out must hold the text's upper bound: n_bytes of the input + n * id_suffix_times * strlen(id_suffix) + 1 (the newline a
 * last line without terminator did not have). */
int snk_fastq_format_device(const uint8_t *d_text, const uint32_t *d_line, const snk_read_result *d_keep,
                            const snk_read_result *d_rec, int64_t n, const snk_fastq_format *fmt, uint8_t *d_out,
                            uint32_t *d_out_off, void *d_tmp, size_t tmp_bytes, void *stream);

/* gzip members of a clean text, made on the device (the reference compresses its clean files with zlib level 2, one gzip
 * member per thread part: src/peprocess.cpp:1809, 2386; the compressed bytes are not part of the contract, the text is).
 * d_text / d_off[n + 1]: the text and record offsets snk_fastq_format_device() wrote (any text cut into records will do).
 * Every `records_per_member` records (a power of two, at most 1024) become one gzip member -- one dynamic-Huffman deflate block
 * whose code is built from the symbol counts of the whole call, CRC-32 and ISIZE in the trailer -- and the members are
 * concatenated in d_gz[0, d_info[0]).  d_info (uint32_t[4]): [0] total bytes, [1] members, [2] non-zero when gz_cap was too
 * small (nothing usable was written: compress on the host).  d_gz: 4-byte aligned, gz_cap < 2^32; it is zeroed by the call. */
size_t snk_fastq_deflate_tmp_bytes(int64_t max_records, int32_t records_per_member);
int snk_fastq_deflate_device(const uint8_t *d_text, const uint32_t *d_off, int64_t n, int32_t records_per_member, uint8_t *d_gz,
                             uint64_t gz_cap, uint32_t *d_info, void *d_tmp, size_t tmp_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif
