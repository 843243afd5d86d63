// snk_fastq.hip -- device-side FASTQ ingest / egress around the filter hot path (include/snk_fastq.h, gfx950).
//
// Ingest (reference: the gzgets record loop, src/peprocess.cpp:2063-2113).  The raw text of a batch sits in HBM:
//   1. fq_count_kernel    one workgroup per 4 KB: '\n' characters per block (16 bytes per lane, byte-parallel compare)
//   2. scan               exclusive scan of the block counts (three small kernels, reused for the output offsets)
//   3. fq_index_kernel    every '\n' writes the start offset of the line behind it: d_line[rank + 1] = position + 1
//   4. fq_scatter_kernel  one wavefront per record: sequence and quality lines -> the SoA planes of snk_batch, four
//                         bytes per lane through aligned dword loads + a funnel shift (the source is unaligned text),
//                         lengths checked
// Egress (reference: output_fastqs + preOutput, src/peprocess.cpp:3383-3484, 1617-1647):
//   5. fq_outlen_kernel   bytes of every kept record's clean text
//   6. scan               -> offset of every record in the clean text
//   7. fq_format_kernel   one wavefront per kept record gathers id / bases / "+" / qualities of the kept range
// Byte work at HBM speed: a few hundred MB per batch, microseconds next to the host's I/O.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include "snk_device.h"
#include "../../include/snk_fastq.h"

void snk_set_error(const char *msg);          // snk_filter.cpp: the thread-local text behind snk_last_error()

namespace {

typedef unsigned int u32;
typedef unsigned long long u64;

constexpr int FQ_BLK = 4096;                  // text bytes per workgroup of the count / index kernels (256 lanes x 16 B)
constexpr int SCAN_BLK = 1024;                // elements per workgroup of the scan (256 lanes x 4)

// bit 7 of every byte of x that equals '\n' (exact per byte, no carries between bytes)
__device__ __forceinline__ u32 nl_flags(u32 x) {
    const u32 m = x ^ 0x0A0A0A0Au;
    const u32 t = (m & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | m | 0x7F7F7F7Fu);
}

// the 16 text bytes of this lane (bytes at and past n read as 0)
__device__ __forceinline__ uint4 load16(const uint8_t *text, u64 off, u64 n) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (off + 16 <= n) v = *reinterpret_cast<const uint4 *>(text + off);
    else if (off < n) {
        u32 w[4] = {0, 0, 0, 0};
        for (u64 k = off; k < n; ++k) w[(k - off) >> 2] |= (u32)text[k] << (8 * ((k - off) & 3));
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return v;
}

__device__ __forceinline__ u32 wave_incl_scan(u32 v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 o = (u32)__shfl_up((int)v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan of one value per lane over a workgroup of 256; total in *tot
__device__ __forceinline__ u32 block_excl_scan256(u32 v, u32 *lds4, u32 *tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 inc = wave_incl_scan(v, lane);
    if (lane == 63) lds4[wave] = inc;
    __syncthreads();
    u32 base = 0;
    for (int w = 0; w < wave; ++w) base += lds4[w];
    if (tot) *tot = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    __syncthreads();
    return base + inc - v;
}

__global__ void __launch_bounds__(256) fq_count_kernel(const uint8_t *text, u64 n, u32 *cnt) {
    __shared__ u32 l4[4];
    const u64 off = (u64)blockIdx.x * FQ_BLK + (u64)threadIdx.x * 16;
    const uint4 v = load16(text, off, n);
    const u32 c = __popc(nl_flags(v.x)) + __popc(nl_flags(v.y)) + __popc(nl_flags(v.z)) + __popc(nl_flags(v.w));
    u32 tot;
    (void)block_excl_scan256(c, l4, &tot);
    if (threadIdx.x == 0) cnt[blockIdx.x] = tot;
}

// d_line[r + 1] = position behind the r-th '\n' (r counted over the whole text); d_line[0] = 0
__global__ void __launch_bounds__(256) fq_index_kernel(const uint8_t *text, u64 n, const u32 *blk_off, u32 nblk, u32 *line, u32 max_lines,
                                                        int space_num, u32 *status) {
    __shared__ u32 l4[4];
    const u64 off = (u64)blockIdx.x * FQ_BLK + (u64)threadIdx.x * 16;
    const uint4 v = load16(text, off, n);
    const u32 f[4] = {nl_flags(v.x), nl_flags(v.y), nl_flags(v.z), nl_flags(v.w)};
    const u32 c = __popc(f[0]) + __popc(f[1]) + __popc(f[2]) + __popc(f[3]);
    u32 r = blk_off[blockIdx.x] + block_excl_scan256(c, l4, nullptr);
    if (c) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u32 m = f[k];
            while (m) {
                const int b = __ffs((int)m) - 1;            // bit 8*j + 7
                m &= m - 1;
                if (r < max_lines) line[r + 1] = (u32)(off + 4 * k + (b >> 3) + 1);
                ++r;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        line[0] = 0;
        const u32 total = blk_off[nblk];
        status[SNK_FQ_ST_LINES] = total;
        // the last line may lack its '\n' (end of file): it ends at the end of the text and keeps all its characters
        if (total + 1 == max_lines && n > 0 && text[n - 1] != '\n') line[max_lines] = (u32)n + (u32)space_num;
        else if (total < max_lines) atomicOr(&status[SNK_FQ_ST_FLAGS], SNK_FQ_F_TRUNCATED);
    }
}

// ---- exclusive scan of u32 (n elements -> n + 1 offsets): block sums, scan of the sums (one workgroup), block scans
__global__ void __launch_bounds__(256) scan_sums_kernel(const u32 *in, u64 n, u32 *sums) {
    __shared__ u32 l4[4];
    const u64 i = (u64)blockIdx.x * SCAN_BLK + (u64)threadIdx.x * 4;
    u32 s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + k < n) s += in[i + k];
    u32 tot;
    (void)block_excl_scan256(s, l4, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(1024) scan_top_kernel(u32 *sums, u32 nb) {       // in place, exclusive; sums[nb] = total
    __shared__ u32 wsum[16];
    __shared__ u32 carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (u32 base = 0; base < nb; base += 1024) {
        const u32 i = base + threadIdx.x;
        const u32 v = i < nb ? sums[i] : 0;
        const u32 inc = wave_incl_scan(v, lane);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        u32 wb = 0;
        for (int w = 0; w < wave; ++w) wb += wsum[w];
        const u32 carry = carry_s;
        if (i < nb) sums[i] = carry + wb + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wb + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[nb] = carry_s;
}
__global__ void __launch_bounds__(256) scan_apply_kernel(const u32 *in, u64 n, const u32 *sums, u32 *out) {
    __shared__ u32 l4[4];
    const u64 i = (u64)blockIdx.x * SCAN_BLK + (u64)threadIdx.x * 4;
    u32 v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = i + k < n ? in[i + k] : 0; s += v[k]; }
    u32 e = sums[blockIdx.x] + block_excl_scan256(s, l4, nullptr);
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (i + k < n) out[i + k] = e; e += v[k]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = sums[gridDim.x];
}
// out[0 .. n] = exclusive scan of in[0 .. n); tmp: n / 1024 + 2 words; in == out is allowed
void launch_scan(const u32 *in, u64 n, u32 *out, u32 *tmp, hipStream_t s) {
    const u32 nb = (u32)((n + SCAN_BLK - 1) / SCAN_BLK);
    if (n == 0) { (void)hipMemsetAsync(out, 0, sizeof(u32), s); return; }
    hipLaunchKernelGGL(scan_sums_kernel, dim3(nb), dim3(256), 0, s, in, n, tmp);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, s, tmp, nb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(256), 0, s, in, n, (const u32 *)tmp, out);
}

// four text bytes starting at the (unaligned) offset `at`: two aligned dword loads + funnel shift.  Reads up to 7 bytes
// past at + 4 rounded: the text buffer carries that much slack (snk_fastq.h)
__device__ __forceinline__ u32 load4u(const uint8_t *text, u32 at) {
    const u32 *p = reinterpret_cast<const u32 *>(text + (at & ~3u));
    const u32 lo = p[0], hi = p[1];
    return __builtin_amdgcn_alignbit(hi, lo, (at & 3u) * 8u);
}

__global__ void __launch_bounds__(256) fq_scatter_kernel(const uint8_t *text, const u32 *line, long n, int space_num, int pitch, int lcap,
                                                          uint8_t *seq, uint8_t *qual, uint16_t *len, u32 *status) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    u32 flags = 0, maxlen = 0, bad = 0xFFFFFFFFu;
    for (long r = wave; r < n; r += nwaves) {
        const u32 l1 = line[4 * r + 1], l2 = line[4 * r + 2], l3 = line[4 * r + 3], l4 = line[4 * r + 4];
        const u32 se = l2 > l1 + (u32)space_num ? l2 - (u32)space_num : l1, qe = l4 > l3 + (u32)space_num ? l4 - (u32)space_num : l3;
        const u32 slen = se - l1, qlen = qe - l3;
        maxlen = max(maxlen, slen);
        if (slen != qlen) { flags |= SNK_FQ_F_LEN_MISMATCH; bad = min(bad, (u32)r); }
        if (slen > (u32)lcap) { flags |= SNK_FQ_F_TOO_LONG; if (lane == 0) len[r] = (uint16_t)min(slen, 65535u); continue; }
        if (lane == 0) len[r] = (uint16_t)slen;
        u32 *ds = reinterpret_cast<u32 *>(seq + r * (long)pitch), *dq = reinterpret_cast<u32 *>(qual + r * (long)pitch);
        for (u32 p = 4u * (u32)lane; p < slen; p += 256u) {
            ds[p >> 2] = load4u(text, l1 + p);
            if (p < qlen) dq[p >> 2] = load4u(text, l3 + p);
        }
    }
    if (lane == 0) {
        if (flags) atomicOr(&status[SNK_FQ_ST_FLAGS], flags);
        if (maxlen) atomicMax(&status[SNK_FQ_ST_MAXLEN], maxlen);
        if (bad != 0xFFFFFFFFu) atomicMin(&status[SNK_FQ_ST_BADREC], bad);
    }
}

struct FmtDev {
    int space_num, qual_delta, sfx_len, sfx_total;
    u32 sfx;                         // up to 3 suffix characters, byte k = character k
    u32 bc_from, bc_to;
    u32 select, whole;
};

// kept range of a record as the host writer cuts it (clean_start / clean_len clamped to the line)
__device__ __forceinline__ void kept_range(const snk_read_result &x, u32 slen, u32 whole, u32 &cs, u32 &cl) {
    cs = whole ? 0u : min((u32)x.clean_start, slen);
    cl = whole ? slen : min((u32)x.clean_len, slen - cs);
}

__global__ void __launch_bounds__(256) fq_outlen_kernel(const u32 *line, const snk_read_result *keep, const snk_read_result *rec, long n,
                                                         FmtDev F, u32 *out_len) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 L = 0;
    if (keep[i].reason == F.select) {
        const u32 l0 = line[4 * i], l1 = line[4 * i + 1], l2 = line[4 * i + 2];
        const u32 ide = l1 > l0 + (u32)F.space_num ? l1 - (u32)F.space_num : l0, se = l2 > l1 + (u32)F.space_num ? l2 - (u32)F.space_num : l1;
        u32 cs, cl;
        kept_range(rec[i], se - l1, F.whole, cs, cl);
        L = (ide - l0) + (u32)F.sfx_total + 1u + cl + 3u + cl + 1u;
    }
    out_len[i] = L;
}

__global__ void __launch_bounds__(256) fq_format_kernel(const uint8_t *text, const u32 *line, const snk_read_result *keep,
                                                         const snk_read_result *rec, long n, FmtDev F, const u32 *out_off, uint8_t *out) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long r = wave; r < n; r += nwaves) {
        if (keep[r].reason != F.select) continue;
        const u32 l0 = line[4 * r], l1 = line[4 * r + 1], l2 = line[4 * r + 2], l3 = line[4 * r + 3];
        const u32 sp = (u32)F.space_num;
        const u32 ide = l1 > l0 + sp ? l1 - sp : l0, se = l2 > l1 + sp ? l2 - sp : l1;
        const u32 idl = ide - l0;
        u32 cs, cl;
        kept_range(rec[r], se - l1, F.whole, cs, cl);
        const u32 a = idl + (u32)F.sfx_total;             // '\n' behind the id line
        const u32 b = a + 1 + cl;                         // "\n+\n"
        const u32 c = b + 3 + cl;                         // final '\n'
        uint8_t *o = out + out_off[r];
        for (u32 k = (u32)lane; k <= c; k += 64u) {
            u32 ch;
            if (k < idl) ch = text[l0 + k];
            else if (k < a) ch = (F.sfx >> (8u * ((k - idl) % (u32)F.sfx_len))) & 0xFFu;
            else if (k == a) ch = '\n';
            else if (k < b) {
                ch = text[l1 + cs + (k - a - 1)];
                if (F.bc_from && ((ch >= 'a' && ch <= 'z') ? ch - 32u : ch) == F.bc_from) ch = F.bc_to;      // toupper(c) == from
            } else if (k < b + 3) ch = k == b + 1 ? '+' : '\n';
            else if (k < c) ch = (text[l3 + cs + (k - b - 3)] + (u32)F.qual_delta) & 0xFFu;
            else ch = '\n';
            o[k] = (uint8_t)ch;
        }
    }
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

// the scan, for snk_gzip.hip
void snk_fq_launch_scan(const unsigned *in, unsigned long long n, unsigned *out, unsigned *tmp, void *stream) {
    launch_scan(in, n, out, tmp, (hipStream_t)stream);
}

extern "C" {

size_t snk_fastq_tmp_bytes(uint64_t max_bytes, int64_t max_records) {
    const size_t nblk = (size_t)(max_bytes / FQ_BLK) + 2;
    // block counts / offsets of the text (+ their scan scratch), output lengths scan scratch
    return al256((nblk + 2) * 4) + al256((nblk / SCAN_BLK + 4) * 4) + al256(((size_t)max_records / SCAN_BLK + 4) * 4) + 1024;
}

int snk_fastq_parse_device(const uint8_t *d_text, uint64_t n_bytes, int64_t n_records, int32_t space_num, int32_t pitch, int32_t lcap,
                           uint8_t *d_seq, uint8_t *d_qual, uint16_t *d_len, uint32_t *d_line, uint32_t *d_status, void *d_tmp,
                           size_t tmp_bytes, void *stream) {
    if (!d_text || !d_seq || !d_qual || !d_len || !d_line || !d_status || !d_tmp || n_records < 0 || space_num < 0 || pitch <= 0 || (pitch & 3) ||
        lcap <= 0 || lcap > pitch) {
        snk_set_error("snk_fastq_parse_device: bad argument");
        return SNK_E_PARAM;
    }
    if (n_bytes >= 0xFFFFFFC0ull || n_records > 0x3FFFFFF0ll) { snk_set_error("snk_fastq_parse_device: text of 4 GB or more"); return SNK_E_PARAM; }
    if (((uintptr_t)d_text & 15) || ((uintptr_t)d_seq & 3) || ((uintptr_t)d_qual & 3)) { snk_set_error("snk_fastq_parse_device: text must be 16-byte aligned, planes 4-byte aligned"); return SNK_E_PARAM; }
    if (tmp_bytes < snk_fastq_tmp_bytes(n_bytes, n_records)) { snk_set_error("snk_fastq_parse_device: scratch too small (snk_fastq_tmp_bytes)"); return SNK_E_PARAM; }
    hipStream_t s = (hipStream_t)stream;
    const u32 nblk = (u32)((n_bytes + FQ_BLK - 1) / FQ_BLK);
    u32 *cnt = (u32 *)d_tmp, *scr = (u32 *)((uint8_t *)d_tmp + al256(((size_t)nblk + 2) * 4));
    if (hipMemsetAsync(d_status, 0, 3 * sizeof(u32), s) != hipSuccess || hipMemsetAsync(d_status + SNK_FQ_ST_BADREC, 0xFF, sizeof(u32), s) != hipSuccess) {
        snk_set_error("snk_fastq_parse_device: status reset failed");
        return SNK_E_HIP;
    }
    if (n_records == 0) return SNK_OK;
    if (nblk) hipLaunchKernelGGL(fq_count_kernel, dim3(nblk), dim3(256), 0, s, d_text, (u64)n_bytes, cnt);
    launch_scan(cnt, nblk, cnt, scr, s);
    if (nblk) hipLaunchKernelGGL(fq_index_kernel, dim3(nblk), dim3(256), 0, s, d_text, (u64)n_bytes, (const u32 *)cnt, nblk, d_line, (u32)(4 * n_records),
                                 (int)space_num, d_status);
    const long waves = (n_records + 0) < 256 * 4 * 8 ? (long)((n_records + 3) / 4 * 4) : 256L * 4 * 8;
    hipLaunchKernelGGL(fq_scatter_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, d_text, (const u32 *)d_line, (long)n_records,
                       (int)space_num, (int)pitch, (int)lcap, d_seq, d_qual, d_len, d_status);
    if (hipGetLastError() != hipSuccess) { snk_set_error("snk_fastq_parse_device: launch failed"); return SNK_E_HIP; }
    return SNK_OK;
}

int snk_fastq_format_device(const uint8_t *d_text, const uint32_t *d_line, const snk_read_result *d_keep, const snk_read_result *d_rec,
                            int64_t n, const snk_fastq_format *fmt, uint8_t *d_out, uint32_t *d_out_off, void *d_tmp, size_t tmp_bytes,
                            void *stream) {
    if (!d_text || !d_line || !d_keep || !d_rec || !fmt || !d_out || !d_out_off || !d_tmp || n < 0 || fmt->struct_size != (int32_t)sizeof(snk_fastq_format)) {
        snk_set_error("snk_fastq_format_device: bad argument");
        return SNK_E_PARAM;
    }
    if (tmp_bytes < snk_fastq_tmp_bytes(0, n)) { snk_set_error("snk_fastq_format_device: scratch too small (snk_fastq_tmp_bytes)"); return SNK_E_PARAM; }
    FmtDev F;
    F.space_num = fmt->space_num;
    F.qual_delta = fmt->qual_delta;
    F.sfx_len = (int)strnlen(fmt->id_suffix, 3);
    F.sfx_total = fmt->id_suffix_times > 0 ? F.sfx_len * fmt->id_suffix_times : 0;
    if (F.sfx_len == 0) { F.sfx_len = 1; F.sfx_total = 0; }
    F.sfx = 0;
    for (int k = 0; k < 3 && fmt->id_suffix[k]; ++k) F.sfx |= (u32)(uint8_t)fmt->id_suffix[k] << (8 * k);
    F.bc_from = fmt->base_from;
    F.bc_to = fmt->base_to;
    F.select = fmt->select_reason;
    F.whole = fmt->whole_read;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) { (void)hipMemsetAsync(d_out_off, 0, sizeof(u32), s); return SNK_OK; }
    hipLaunchKernelGGL(fq_outlen_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const u32 *)d_line, d_keep, d_rec, (long)n, F, d_out_off);
    launch_scan(d_out_off, (u64)n, d_out_off, (u32 *)d_tmp, s);
    const long waves = n < 256 * 4 * 8 ? (long)((n + 3) / 4 * 4) : 256L * 4 * 8;
    hipLaunchKernelGGL(fq_format_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, d_text, (const u32 *)d_line, d_keep, d_rec, (long)n, F,
                       (const u32 *)d_out_off, d_out);
    if (hipGetLastError() != hipSuccess) { snk_set_error("snk_fastq_format_device: launch failed"); return SNK_E_HIP; }
    return SNK_OK;
}

}  // extern "C"
