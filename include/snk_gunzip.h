/*
 * snk_gunzip.h -- C ABI of the device inflate ("parallel inflate ... the real speed lever").
 *
 * Replaces, for the FASTQ reader, the reference's gzgets() loop over one zlib inflate stream per file
 * (src/peprocess.cpp:2063-2113; gzopen at :2018-2026).  One gzip stream is decoded by thousands of wavefronts: the
 * compressed bytes of a WINDOW of the file are cut into chunks, a search kernel finds a deflate block start in every
 * chunk, a decode kernel turns every chunk into 16-bit symbols with an unknown 32 KiB window (markers), the host checks
 * that the chunks chain (every chunk ends where the next one starts), and a chain + resolve kernel pair replaces the
 * markers and leaves the window's text in HBM / copies it to the host.  host/snk_dgunzip.h drives the calls, verifies
 * every member's CRC-32 / ISIZE and falls back to the host decoder (host/snk_inflate.h) for whatever does not fit, so
 * the bytes are always zlib's bytes or an error.  The decoding itself is csrc/snk_inflate_core.hip.h, which also compiles
 * for the host (tests/test_inflate_emul.py: against zlib, no GPU needed).
 *
 * Plain C, plain pointers and sizes.  All calls are synchronous on the library's own stream.
 */
#ifndef SNK_GUNZIP_H
#define SNK_GUNZIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* one chunk of a window, as csrc/snk_inflate_core.hip.h snkinf::Chunk (same layout, 80 bytes) */
typedef struct snk_gunzip_chunk {
    uint64_t start_bit, stop_bit, out_off;
    uint32_t out_cap, first_of_member, n_syms, status;
    uint64_t end_bit;
    uint32_t known_from, n_ends, stream_end, ends_off, ends_cap, pad_[3];
} snk_gunzip_chunk;
typedef struct snk_gunzip_member { uint32_t sym_index, crc, isize, pad_; } snk_gunzip_member;

enum { SNK_GZ_OK = 0, SNK_GZ_FULL = 1, SNK_GZ_BAD = 2, SNK_GZ_TOO_MANY_MEMBERS = 3, SNK_GZ_NOT_STARTED = 4 };

typedef struct snk_gunzip snk_gunzip;

/* buffers for windows of up to max_window_bytes compressed bytes cut into chunks of chunk_bytes, syms_per_chunk symbol slots
 * and ends_per_chunk member-end slots each; NULL + snk_last_error() on failure */
snk_gunzip *snk_gunzip_create(int device, uint64_t max_window_bytes, uint32_t chunk_bytes, uint32_t syms_per_chunk, uint32_t ends_per_chunk);
void snk_gunzip_destroy(snk_gunzip *g);

/* Uploads the window (nbytes <= max_window_bytes compressed bytes from the host), finds a block start in every chunk but the
 * first (chunk c = bytes [c * chunk_bytes, (c + 1) * chunk_bytes); the first chunk starts at first_bit, a block header), decodes all
 * chunks and copies the chunk table (ceil(nbytes / chunk_bytes) entries) and the member ends back.  Chunk c's symbols stay in HBM. */
int snk_gunzip_decode(snk_gunzip *g, const uint8_t *h_comp, uint64_t nbytes, uint64_t first_bit, int first_of_member,
                      snk_gunzip_chunk *h_chunks, snk_gunzip_member *h_ends);

/* Resolves the chunks listed in order[0 .. k) (indices into the table of the last snk_gunzip_decode(); they must chain) against the
 * 32 KiB in front of the first of them (h_window_in, or NULL for an empty window) and copies their text, text_bytes = the sum of
 * their n_syms, to h_text (host) and the last 32 KiB of the stream so far to h_window_out. */
int snk_gunzip_resolve(snk_gunzip *g, const uint32_t *order, uint32_t k, const uint8_t *h_window_in, uint8_t *h_text, uint64_t text_bytes,
                       uint8_t *h_window_out);

#ifdef __cplusplus
}
#endif
#endif
