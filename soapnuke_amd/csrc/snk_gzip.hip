// snk_gzip.hip -- gzip members of the clean FASTQ text, made on the device (include/snk_fastq.h: snk_fastq_deflate_device;
// SURVEY.md 8f N2 "GPU deflate").  The reference writes its clean files with zlib level 2, one gzip member per thread part
// (src/peprocess.cpp:1809, 2386); the compressed bytes are not part of the contract, the decompressed text is.
//
// The text of a batch is already in HBM as records with known offsets (fq_format_kernel).  What repeats inside 32 KiB of
// FASTQ is the text of the name lines, at the same place one record earlier; bases and qualities are literals for any LZ77
// that does not search far (the host encoder snk_deflate.h is built on the same observation).  So:
//   tokens    per record (one lane = one record, the records of a member side by side): the name line is compared with the
//             record in front of it at the distance of the two record starts, runs of >= 4 equal bytes become matches,
//             everything else is literals.  The same walk runs three times with different sinks (count, measure, emit).
//   dfl_hist  symbol counts of the whole batch (LDS histograms, flushed with atomics)
//   dfl_build ONE wavefront builds the batch's Huffman codes: two-smallest merging with wave-wide argmin, lengths limited to
//             15 bits by halving the counts, canonical codes, and the dynamic-block header every member of the batch starts
//             with (every symbol gets a code, so any later token is encodable)
//   dfl_bits  bits of every record under that code, its CRC-32
//   dfl_member  per member (a fixed number of records): bit offsets of its records (scan), its size, CRC-32 (zlib's
//             crc32_combine as a tree over the records) and ISIZE
//   (scan over the member sizes -> byte offsets)
//   dfl_emit  gzip header, block header, every lane writes its records' codes at their bit offsets (whole dwords plainly, the
//             two dwords it shares with its neighbours by atomicOr into the zeroed buffer), end-of-block, trailer
// Integer / byte work; nothing here is bound by anything but the latency of the per-lane byte walks, and it runs beside the
// host's inflate, which is what bounds a .gz -> .gz run.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "snk_device.h"
#include "../../include/snk_fastq.h"

void snk_set_error(const char *msg);
void snk_fq_launch_scan(const unsigned *in, unsigned long long n, unsigned *out, unsigned *tmp, void *stream);   // snk_fastq.hip

namespace {

typedef unsigned int u32;
typedef unsigned long long u64;

constexpr int NLIT = 286, NDIST = 30, NCL = 19;
constexpr u32 NONE = 0xFFFFFFFFu;

struct DfCode {                                   // one per call, in the scratch area
    u32 lit_hist[NLIT], dist_hist[NDIST];
    u32 lit_code[NLIT], dist_code[NDIST];         // bit-reversed (LSB-first emission)
    u32 lit_len[NLIT], dist_len[NDIST];
    u32 hdr_bits;
    u32 hdr[96];                                  // the dynamic block header (BFINAL + BTYPE + code description), LSB first
};

__constant__ u32 c_crc_tab[256];
__constant__ u32 c_x2n[32];                       // x^(2^n) mod p, reflected (zlib's crc32_combine)

// ---- symbols
__device__ __forceinline__ void len_sym(u32 L, u32 &sym, u32 &xb, u32 &xv) {       // L in 3..258
    const u32 l = L - 3;
    if (L == 258) { sym = 285; xb = 0; xv = 0; return; }
    if (l < 8) { sym = 257 + l; xb = 0; xv = 0; return; }
    const u32 nb = 31u - (u32)__clz((int)l);
    xb = nb - 2;
    sym = 257 + 4 * (nb - 1) + ((l >> xb) & 3u);
    xv = l & ((1u << xb) - 1u);
}
__device__ __forceinline__ void dist_sym(u32 D, u32 &sym, u32 &xb, u32 &xv) {      // D in 1..32768
    const u32 d = D - 1;
    if (d < 4) { sym = d; xb = 0; xv = 0; return; }
    const u32 nb = 31u - (u32)__clz((int)d);
    xb = nb - 1;
    sym = 2 * nb + ((d >> xb) & 1u);
    xv = d & ((1u << xb) - 1u);
}

// ---- the token walk of one record: text[o, o + len), the kept record in front of it at prev (NONE: none / another member)
template <class Sink>
__device__ __forceinline__ void walk_record(const uint8_t *text, u32 o, u32 len, u32 prev, Sink &S) {
    u32 i = 0;
    if (prev != NONE && o - prev <= 32768u && o - prev >= 1u) {
        const u32 dist = o - prev;
        u32 run = 0;
        for (; i < len; ++i) {                    // the name line
            const uint8_t c = text[o + i];
            const bool eq = text[o + i - dist] == c;
            if (eq) { ++run; }
            else {
                if (run >= 4) S.match(run, dist, text + o + i - run);
                else for (u32 k = run; k > 0; --k) S.lit(text[o + i - k]);
                run = 0;
                S.lit(c);
            }
            if (c == '\n') { ++i; break; }
        }
        if (run >= 4) S.match(run, dist, text + o + i - run);
        else for (u32 k = run; k > 0; --k) S.lit(text[o + i - k]);
    }
    for (; i < len; ++i) S.lit(text[o + i]);
}

// a run of `run` matching bytes as matches of at most 258 (a rest of 1 or 2 bytes is not a legal match: it is taken from
// the piece before)
template <class F>
__device__ __forceinline__ void split_match(u32 run, F f) {
    while (run) {
        u32 piece = run < 258u ? run : 258u;
        if (run - piece > 0 && run - piece < 3) piece = run - 3;
        f(piece);
        run -= piece;
    }
}

struct HistSink {
    u32 *lh, *dh;
    __device__ __forceinline__ void lit(uint8_t c) { atomicAdd(&lh[c], 1u); }
    __device__ __forceinline__ void match(u32 run, u32 dist, const uint8_t *) {
        u32 ds, dxb, dxv;
        dist_sym(dist, ds, dxb, dxv);
        split_match(run, [&](u32 piece) {
            u32 ls, xb, xv;
            len_sym(piece, ls, xb, xv);
            atomicAdd(&lh[ls], 1u);
            atomicAdd(&dh[ds], 1u);
        });
    }
};
struct BitsSink {
    const u32 *ll, *dl;
    u32 bits, crc;
    __device__ __forceinline__ void upd(uint8_t c) { crc = c_crc_tab[(crc ^ c) & 0xFFu] ^ (crc >> 8); }
    __device__ __forceinline__ void lit(uint8_t c) { bits += ll[c]; upd(c); }
    __device__ __forceinline__ void match(u32 run, u32 dist, const uint8_t *src) {
        u32 ds, dxb, dxv;
        dist_sym(dist, ds, dxb, dxv);
        split_match(run, [&](u32 piece) {
            u32 ls, xb, xv;
            len_sym(piece, ls, xb, xv);
            bits += ll[ls] + xb + dl[ds] + dxb;
        });
        for (u32 k = 0; k < run; ++k) upd(src[k]);
    }
};
// LSB-first bit writer into 32-bit words of a zeroed buffer; the first and the last word may be shared with neighbours
struct EmitSink {
    const u32 *lc, *ll, *dc, *dl;
    u32 *out;                                     // word array of the whole output
    u64 word;                                     // index of the word being filled
    u64 acc;
    u32 nacc;                                     // bits in acc (the low `skip` bits of the first word belong to somebody else)
    bool first;
    __device__ __forceinline__ void put(u32 v, u32 n) {
        acc |= (u64)v << nacc;
        nacc += n;
        if (nacc >= 32) {
            if (first) { atomicOr(&out[word], (u32)acc); first = false; }
            else out[word] = (u32)acc;
            ++word;
            acc >>= 32;
            nacc -= 32;
        }
    }
    __device__ __forceinline__ void finish() { if (nacc) atomicOr(&out[word], (u32)acc); }
    __device__ __forceinline__ void lit(uint8_t c) { put(lc[c], ll[c]); }
    __device__ __forceinline__ void match(u32 run, u32 dist, const uint8_t *) {
        u32 ds, dxb, dxv;
        dist_sym(dist, ds, dxb, dxv);
        split_match(run, [&](u32 piece) {
            u32 ls, xb, xv;
            len_sym(piece, ls, xb, xv);
            put(lc[ls], ll[ls]);
            if (xb) put(xv, xb);
            put(dc[ds], dl[ds]);
            if (dxb) put(dxv, dxb);
        });
    }
};

// the kept record in front of record r inside its member (records that are not kept have no text)
__device__ __forceinline__ u32 prev_kept(const u32 *off, long r, long member_first) {
    for (long q = r - 1; q >= member_first; --q)
        if (off[q + 1] > off[q]) return off[q];
    return NONE;
}

__global__ void __launch_bounds__(256) dfl_hist_kernel(const uint8_t *text, const u32 *off, long n, int rpm, DfCode *C) {
    __shared__ u32 lh[NLIT], dh[NDIST];
    for (int k = threadIdx.x; k < NLIT; k += blockDim.x) lh[k] = 0;
    if (threadIdx.x < NDIST) dh[threadIdx.x] = 0;
    __syncthreads();
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) {
        const u32 o = off[r], len = off[r + 1] - o;
        if (len) {
            HistSink S{lh, dh};
            walk_record(text, o, len, prev_kept(off, r, r / rpm * rpm), S);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < NLIT; k += blockDim.x) if (lh[k]) atomicAdd(&C->lit_hist[k], lh[k]);
    if (threadIdx.x < NDIST && dh[threadIdx.x]) atomicAdd(&C->dist_hist[threadIdx.x], dh[threadIdx.x]);
}

// ---- Huffman code lengths of up to 288 symbols by one wavefront.  The leaves are sorted by rank counting (every lane counts the
// keys below its own: n * n / 64 compares), lane 0 merges with the two-queue method (leaves ascending, internal nodes in the order
// they were made: both queues are sorted, the two lightest nodes are at their fronts), the depths are read off in parallel.
// Lengths above `maxbits`: the counts are halved (rounding up) and the tree rebuilt; a floor of total / 2^(maxbits - 2) under the
// counts makes the first tree fit in practice (a FASTQ batch has a few symbols with 10^7 occurrences and 250 with none).
__device__ void wave_huff_lengths(const u32 *cnt, int n, int maxbits, u32 *len_out, u32 *w, int *par, u32 *scale_buf) {
    const int lane = threadIdx.x & 63;
    u32 *sorted = scale_buf + 288;                        // leaf indices by ascending weight
    {
        u64 tot = 0;
        for (int k = lane; k < n; k += 64) tot += cnt[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) tot += __shfl_xor(tot, o, 64);
        const u32 floor_ = (u32)(tot >> (maxbits - 2)) + 1u;
        for (int k = lane; k < n; k += 64) scale_buf[k] = cnt[k] > floor_ ? cnt[k] : floor_;
    }
    __syncthreads();
    for (;;) {
        for (int k = lane; k < n; k += 64) {
            const u64 key = ((u64)scale_buf[k] << 9) | (u32)k;
            int rank = 0;
            for (int j = 0; j < n; ++j) rank += ((((u64)scale_buf[j] << 9) | (u32)j) < key) ? 1 : 0;
            sorted[rank] = (u32)k;
        }
        for (int k = lane; k < 2 * n; k += 64) par[k] = -1;
        __syncthreads();
        if (lane == 0) {
            int lf = 0, in_f = n, in_b = n;               // leaf queue front; internal nodes [in_f, in_b) live in w[] / par[] at n..
            auto take = [&]() -> int {                    // the lighter of the two fronts (ties: the leaf)
                const bool have_leaf = lf < n, have_in = in_f < in_b;
                if (have_leaf && (!have_in || scale_buf[sorted[lf]] <= w[in_f])) return (int)sorted[lf++];
                return in_f++;
            };
            for (int it = 0; it < n - 1; ++it) {
                const int x = take(), y = take();
                w[in_b] = (x < n ? scale_buf[x] : w[x]) + (y < n ? scale_buf[y] : w[y]);
                par[x] = in_b;
                par[y] = in_b;
                ++in_b;
            }
        }
        __syncthreads();
        u32 mx = 0;
        for (int k = lane; k < n; k += 64) {
            u32 d = 0;
            for (int q = k; par[q] >= 0; q = par[q]) ++d;
            len_out[k] = d;
            mx = d > mx ? d : mx;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const u32 x = (u32)__shfl_xor((int)mx, o, 64); mx = x > mx ? x : mx; }
        __syncthreads();
        if ((int)mx <= maxbits) break;
        for (int k = lane; k < n; k += 64) scale_buf[k] = (scale_buf[k] + 1) >> 1;       // flatter counts, shorter codes
        __syncthreads();
    }
}
__device__ __forceinline__ u32 bitrev(u32 v, u32 n) { return __brev(v) >> (32 - n); }
// canonical codes (lane 0), bit-reversed
__device__ void canon_codes(const u32 *len, int n, u32 *code) {
    u32 cnt[16], next[16];
    for (int b = 0; b < 16; ++b) cnt[b] = 0;
    for (int k = 0; k < n; ++k) cnt[len[k]]++;
    cnt[0] = 0;
    u32 c = 0;
    for (int b = 1; b < 16; ++b) { c = (c + cnt[b - 1]) << 1; next[b] = c; }
    for (int k = 0; k < n; ++k) code[k] = len[k] ? bitrev(next[len[k]]++, len[k]) : 0u;
}

__global__ void __launch_bounds__(64) dfl_build_kernel(DfCode *C) {
    __shared__ u32 w[2 * 288], scale_buf[2 * 288], cnt[288], cl_len[NCL], cl_code[NCL];
    __shared__ int par[2 * 288];
    const int lane = threadIdx.x;
    // every symbol gets a code (a later batch position may need any of them): counts + 1
    for (int k = lane; k < NLIT; k += 64) cnt[k] = C->lit_hist[k] + 1u;
    __syncthreads();
    wave_huff_lengths(cnt, NLIT, 15, C->lit_len, w, par, scale_buf);
    __syncthreads();
    for (int k = lane; k < NDIST; k += 64) cnt[k] = C->dist_hist[k] + 1u;
    __syncthreads();
    wave_huff_lengths(cnt, NDIST, 15, C->dist_len, w, par, scale_buf);
    __syncthreads();
    // the code-length code over the lengths 1..15 that occur (no run-length symbols: 316 lengths cost ~160 bytes per member)
    if (lane == 0) {
        for (int k = 0; k < NCL; ++k) { cnt[k] = 0; cl_len[k] = 0; }
        for (int k = 0; k < NLIT; ++k) cnt[C->lit_len[k]]++;
        for (int k = 0; k < NDIST; ++k) cnt[C->dist_len[k]]++;
        int used = 0;
        for (int k = 0; k < 16; ++k) used += cnt[k] != 0;
        if (used < 2) cnt[cnt[1] ? 2 : 1] = 1;                        // a code needs two symbols
        u32 c2[16];
        int idx[16], m = 0;
        for (int k = 0; k < 16; ++k) if (cnt[k]) { c2[m] = cnt[k]; idx[m] = k; ++m; }
        // sequential Huffman over m <= 16 symbols, at most 7 bits (counts halved until it fits)
        for (;;) {
            u32 ww[32];
            int pp[32];
            for (int k = 0; k < m; ++k) { ww[k] = c2[k]; pp[k] = -1; }
            int hi = m;
            for (int it = 0; it < m - 1; ++it) {
                int x = -1, y = -1;
                for (int k = 0; k < hi; ++k) if (pp[k] < 0 && (x < 0 || ww[k] < ww[x])) x = k;
                for (int k = 0; k < hi; ++k) if (pp[k] < 0 && k != x && (y < 0 || ww[k] < ww[y])) y = k;
                ww[hi] = ww[x] + ww[y]; pp[hi] = -1; pp[x] = hi; pp[y] = hi; ++hi;
            }
            u32 mx = 0;
            for (int k = 0; k < m; ++k) { u32 d = 0; for (int q = k; pp[q] >= 0; q = pp[q]) ++d; cl_len[idx[k]] = d; mx = d > mx ? d : mx; }
            if (mx <= 7) break;
            for (int k = 0; k < m; ++k) c2[k] = (c2[k] + 1) >> 1;
        }
        canon_codes(C->lit_len, NLIT, C->lit_code);
        canon_codes(C->dist_len, NDIST, C->dist_code);
        canon_codes(cl_len, NCL, cl_code);
        // header bits: BFINAL 1, BTYPE 10, HLIT 29 (286), HDIST 29 (30), HCLEN 15 (19), the 19 code-length-code lengths in
        // their fixed order, then the 316 lengths
        for (int k = 0; k < 96; ++k) C->hdr[k] = 0;
        u32 nb = 0;
        auto put = [&](u32 v, u32 n) {
            for (u32 k = 0; k < n; ++k, ++nb) if ((v >> k) & 1u) C->hdr[nb >> 5] |= 1u << (nb & 31);
        };
        put(1, 1); put(2, 2); put(NLIT - 257, 5); put(NDIST - 1, 5); put(NCL - 4, 4);
        const int order[NCL] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (int k = 0; k < NCL; ++k) put(cl_len[order[k]], 3);
        for (int k = 0; k < NLIT; ++k) put(cl_code[C->lit_len[k]], cl_len[C->lit_len[k]]);
        for (int k = 0; k < NDIST; ++k) put(cl_code[C->dist_len[k]], cl_len[C->dist_len[k]]);
        C->hdr_bits = nb;
    }
}

// ---- per-record bits + CRC-32
__global__ void __launch_bounds__(256) dfl_bits_kernel(const uint8_t *text, const u32 *off, long n, int rpm, const DfCode *C, u32 *bits, u32 *crc) {
    __shared__ u32 ll[NLIT], dl[NDIST];
    for (int k = threadIdx.x; k < NLIT; k += blockDim.x) ll[k] = C->lit_len[k];
    if (threadIdx.x < NDIST) dl[threadIdx.x] = C->dist_len[threadIdx.x];
    __syncthreads();
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const u32 o = off[r], len = off[r + 1] - o;
    BitsSink S{ll, dl, 0u, 0xFFFFFFFFu};
    if (len) walk_record(text, o, len, prev_kept(off, r, r / rpm * rpm), S);
    bits[r] = S.bits;
    crc[r] = len ? ~S.crc : 0u;
}

// zlib's crc32_combine arithmetic (reflected polynomial 0xedb88320)
__device__ __forceinline__ u32 multmodp(u32 a, u32 b) {
    u32 m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xedb88320u : b >> 1;
    }
    return p;
}
__device__ __forceinline__ u32 x2nmodp(u32 n, u32 k) {
    u32 p = 1u << 31;
    while (n) {
        if (n & 1u) p = multmodp(c_x2n[k & 31], p);
        n >>= 1;
        ++k;
    }
    return p;
}

// per member: exclusive scan of the records' bits (in place), total bits, CRC-32 of the member's text, its byte size
constexpr int RPM_MAX = 1024;
__global__ void __launch_bounds__(256) dfl_member_kernel(const u32 *off, long n, int rpm, const DfCode *C, u32 *bits, const u32 *crc,
                                                         u32 *msize, u32 *mcrc, u32 *mbits) {
    __shared__ u32 sb[RPM_MAX], sc[RPM_MAX], sl[RPM_MAX];
    const long first = (long)blockIdx.x * rpm;
    const int cnt = (int)((n - first) < rpm ? (n - first) : rpm);
    for (int k = threadIdx.x; k < rpm; k += blockDim.x) {
        sb[k] = k < cnt ? bits[first + k] : 0u;
        sc[k] = k < cnt ? crc[first + k] : 0u;
        sl[k] = k < cnt ? off[first + k + 1] - off[first + k] : 0u;
    }
    __syncthreads();
    // bits: Hillis-Steele inclusive scan in LDS (rpm <= 1024)
    for (int d = 1; d < rpm; d <<= 1) {
        u32 v[RPM_MAX / 256];
        int c = 0;
        for (int k = threadIdx.x; k < rpm; k += blockDim.x) v[c++] = k >= d ? sb[k - d] : 0u;
        __syncthreads();
        c = 0;
        for (int k = threadIdx.x; k < rpm; k += blockDim.x) sb[k] += v[c++];
        __syncthreads();
    }
    const u32 total = sb[rpm - 1];
    for (int k = threadIdx.x; k < cnt; k += blockDim.x) bits[first + k] = k ? sb[k - 1] : 0u;     // exclusive
    // CRC: pairwise tree, left = earlier text
    for (int d = 1; d < rpm; d <<= 1) {
        for (int k = threadIdx.x * 2 * d; k + d < rpm; k += blockDim.x * 2 * d) {
            const u32 l2 = sl[k + d];
            if (l2) { sc[k] = (sl[k] ? multmodp(x2nmodp(l2, 3), sc[k]) : 0u) ^ sc[k + d]; sl[k] += l2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const u32 text_bytes = sl[0];
        u32 sz = 0;
        if (text_bytes) {
            const u64 b = (u64)C->hdr_bits + total + C->lit_len[256];
            sz = 10u + (u32)((b + 7) >> 3) + 8u;
        }
        msize[blockIdx.x] = sz;
        mcrc[blockIdx.x] = sc[0];
        mbits[blockIdx.x] = total;
    }
}

__global__ void __launch_bounds__(256) dfl_emit_kernel(const uint8_t *text, const u32 *off, long n, int rpm, const DfCode *C, const u32 *bits,
                                                       const u32 *moff, const u32 *mcrc, const u32 *mbits, u32 nmem, uint8_t *gz, u64 cap,
                                                       u32 *info) {
    __shared__ u32 lc[NLIT], ll[NLIT], dc[NDIST], dl[NDIST];
    for (int k = threadIdx.x; k < NLIT; k += blockDim.x) { lc[k] = C->lit_code[k]; ll[k] = C->lit_len[k]; }
    if (threadIdx.x < NDIST) { dc[threadIdx.x] = C->dist_code[threadIdx.x]; dl[threadIdx.x] = C->dist_len[threadIdx.x]; }
    __syncthreads();
    const u32 mem = blockIdx.x;
    const u32 mo = moff[mem], msz = moff[mem + 1] - mo;
    if (mem == 0 && threadIdx.x == 0) { info[0] = moff[nmem]; info[1] = nmem; }
    if (!msz) return;
    // whole 32-bit words are written: the capacity counts in words (a member that ends in the buffer's last, partial word is an
    // overflow); a member offset scan that wrapped (more than 4 GB of members) is one too
    if ((u64)mo + msz > (cap & ~3ull) || moff[mem + 1] < mo || moff[nmem] < mo) { if (threadIdx.x == 0) atomicOr(&info[2], 1u); return; }
    const long first = (long)mem * rpm;
    const int cnt = (int)((n - first) < rpm ? (n - first) : rpm);
    u32 *out = reinterpret_cast<u32 *>(gz);                    // (the buffer is 4-byte aligned and zeroed)
    const u64 base_bit = ((u64)mo + 10) * 8;                  // the deflate stream starts behind the 10-byte gzip header
    const u32 hb = C->hdr_bits;
    if (threadIdx.x == 0) {
        const uint8_t h[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 4, 3};
        for (int k = 0; k < 10; ++k) atomicOr(&out[(mo + k) >> 2], (u32)h[k] << (8 * ((mo + k) & 3)));
        // block header + end of block + trailer
        EmitSink E{lc, ll, dc, dl, out, base_bit >> 5, 0ull, (u32)(base_bit & 31), true};
        for (u32 k = 0; k < hb; k += 32) { const u32 nb = hb - k < 32 ? hb - k : 32; E.put(nb == 32 ? C->hdr[k >> 5] : (C->hdr[k >> 5] & ((1u << nb) - 1u)), nb); }
        E.finish();
        const u64 eob_bit = base_bit + hb + mbits[mem];
        EmitSink Z{lc, ll, dc, dl, out, eob_bit >> 5, 0ull, (u32)(eob_bit & 31), true};
        Z.put(lc[256], ll[256]);
        Z.finish();
        const u32 tr = mo + msz - 8;
        const u32 c = mcrc[mem];
        u32 isz = 0;
        for (int k = 0; k < cnt; ++k) isz += off[first + k + 1] - off[first + k];
        for (int k = 0; k < 4; ++k) {
            atomicOr(&out[(tr + k) >> 2], ((c >> (8 * k)) & 0xFFu) << (8 * ((tr + k) & 3)));
            atomicOr(&out[(tr + 4 + k) >> 2], ((isz >> (8 * k)) & 0xFFu) << (8 * ((tr + 4 + k) & 3)));
        }
    }
    for (int k = threadIdx.x; k < cnt; k += blockDim.x) {
        const long r = first + k;
        const u32 o = off[r], len = off[r + 1] - o;
        if (!len) continue;
        const u64 bit = base_bit + hb + bits[r];
        EmitSink E{lc, ll, dc, dl, out, bit >> 5, 0ull, (u32)(bit & 31), true};
        walk_record(text, o, len, prev_kept(off, r, first), E);
        E.finish();
    }
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
bool g_tables_done[16] = {false};

int upload_tables() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -1;
    if (g_tables_done[dev]) return 0;
    uint32_t tab[256], x2n[32];
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xedb88320u : c >> 1;
        tab[i] = c;
    }
    auto mult = [](uint32_t a, uint32_t b) {
        uint32_t m = 1u << 31, p = 0;
        for (;;) {
            if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
            m >>= 1;
            b = (b & 1u) ? (b >> 1) ^ 0xedb88320u : b >> 1;
        }
        return p;
    };
    uint32_t p = 1u << 30;                                     // x^1
    x2n[0] = p;
    for (int n = 1; n < 32; ++n) x2n[n] = p = mult(p, p);
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_crc_tab), tab, sizeof tab) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_x2n), x2n, sizeof x2n) != hipSuccess) return -1;
    g_tables_done[dev] = true;
    return 0;
}

}  // namespace

extern "C" {

size_t snk_fastq_deflate_tmp_bytes(int64_t max_records, int32_t records_per_member) {
    const size_t n = (size_t)(max_records > 0 ? max_records : 0), rpm = (size_t)(records_per_member > 0 ? records_per_member : 1);
    const size_t nmem = n / rpm + 2;
    return al256(sizeof(DfCode)) + 2 * al256((n + 1) * 4) + 4 * al256((nmem + 2) * 4) + al256((nmem / 1024 + 4) * 4) + 1024;
}

int snk_fastq_deflate_device(const uint8_t *d_text, const uint32_t *d_off, int64_t n, int32_t records_per_member, uint8_t *d_gz,
                             uint64_t gz_cap, uint32_t *d_info, void *d_tmp, size_t tmp_bytes, void *stream) {
    if (!d_text || !d_off || !d_gz || !d_info || !d_tmp || n < 0 || records_per_member < 1 || records_per_member > RPM_MAX ||
        (records_per_member & (records_per_member - 1)) || ((uintptr_t)d_gz & 3) || gz_cap >= 0xFFFFFF00ull) {
        snk_set_error("snk_fastq_deflate_device: bad argument (records_per_member: a power of two up to 1024; d_gz 4-byte aligned, below 4 GB)");
        return SNK_E_PARAM;
    }
    if (tmp_bytes < snk_fastq_deflate_tmp_bytes(n, records_per_member)) { snk_set_error("snk_fastq_deflate_device: scratch too small"); return SNK_E_PARAM; }
    if (upload_tables() != 0) { snk_set_error("snk_fastq_deflate_device: cannot upload the CRC tables"); return SNK_E_HIP; }
    hipStream_t s = (hipStream_t)stream;
    const int rpm = records_per_member;
    const u32 nmem = (u32)((n + rpm - 1) / rpm);
    uint8_t *t = (uint8_t *)d_tmp;
    DfCode *C = (DfCode *)t; t += al256(sizeof(DfCode));
    u32 *bits = (u32 *)t; t += al256(((size_t)n + 1) * 4);
    u32 *crc = (u32 *)t; t += al256(((size_t)n + 1) * 4);
    u32 *msize = (u32 *)t; t += al256(((size_t)nmem + 2) * 4);
    u32 *moff = (u32 *)t; t += al256(((size_t)nmem + 2) * 4);
    u32 *mcrc = (u32 *)t; t += al256(((size_t)nmem + 2) * 4);
    u32 *mbits = (u32 *)t; t += al256(((size_t)nmem + 2) * 4);
    u32 *scr = (u32 *)t;
    if (hipMemsetAsync(d_info, 0, 4 * sizeof(u32), s) != hipSuccess) { snk_set_error("snk_fastq_deflate_device: memset failed"); return SNK_E_HIP; }
    if (n == 0) return SNK_OK;
    (void)hipMemsetAsync(C, 0, sizeof(DfCode), s);
    (void)hipMemsetAsync(d_gz, 0, (size_t)gz_cap, s);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(dfl_hist_kernel, dim3(blocks), dim3(256), 0, s, d_text, (const u32 *)d_off, (long)n, rpm, C);
    hipLaunchKernelGGL(dfl_build_kernel, dim3(1), dim3(64), 0, s, C);
    hipLaunchKernelGGL(dfl_bits_kernel, dim3(blocks), dim3(256), 0, s, d_text, (const u32 *)d_off, (long)n, rpm, (const DfCode *)C, bits, crc);
    hipLaunchKernelGGL(dfl_member_kernel, dim3(nmem), dim3(256), 0, s, (const u32 *)d_off, (long)n, rpm, (const DfCode *)C, bits, (const u32 *)crc, msize, mcrc, mbits);
    snk_fq_launch_scan(msize, nmem, moff, scr, s);
    hipLaunchKernelGGL(dfl_emit_kernel, dim3(nmem), dim3(256), 0, s, d_text, (const u32 *)d_off, (long)n, rpm, (const DfCode *)C, (const u32 *)bits,
                       (const u32 *)moff, (const u32 *)mcrc, (const u32 *)mbits, nmem, d_gz, (u64)gz_cap, d_info);
    if (hipGetLastError() != hipSuccess) { snk_set_error("snk_fastq_deflate_device: launch failed"); return SNK_E_HIP; }
    return SNK_OK;
}

}  // extern "C"
