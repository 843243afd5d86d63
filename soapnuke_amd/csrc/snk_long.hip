// snk_long.hip -- the fast path for reads of 257..1024 positions (reference limit READ_MAX_LEN 1000,
// src/global_variable.h:9).  The wave-tiled kernel keeps a read's bit planes in registers and its histograms in LDS, both
// sized for <= 256 positions; long reads take three kernels instead:
//
//   snk_long_prep_kernel     one workgroup per 64 reads of a mate, one wavefront per read and trip, 16 bytes per lane: rows
//       arrive as coalesced kilobytes.  It leaves the quality half of A1 stat_read (src/read_filter.cpp:80-313: low-quality
//       count, quality sum) in the read's record and the read's five bit planes (A C G T N over the positions) in the
//       batch's plane store, laid out so that 64 consecutive reads fetch a quad of plane words as one contiguous kilobyte.
//   snk_long_decide_kernel   lane = read (one work-item per pair), on the plane store.  The adapter search (A2,
//       src/read_filter.cpp:707-790) is the bit-sliced one of the tiled kernel (snk_adapter_bits.hip.h) on BLOCKS of the
//       read: planes of 320 positions (10 words) serve the 256 candidate offsets of a block plus the 64 positions an
//       adapter can reach past them; a block in the middle of a read has phase B offsets only, phase A belongs to the first
//       block, phase C to the last one, which ends with the read.  The A / N counts of A1 are popcounts of the same planes;
//       a position that is neither ACGT nor N sends the read to the sequential functions of snk_common.hip.h in its lane
//       (lower case, other letters; also reads shorter than 64), which walk the row; so does poly-X when it is asked
//       for.  Trimming (A3), the discard cascade (A6), the reason counters (one atomic per counter and wavefront:
//       agg_inc) and the trimming-position counters follow as in the generic kernel.
//   snk_long_hist_kernel     lane = four positions.  A workgroup owns 128 positions of one mate for a slice of the batch:
//       raw and clean per-position base / quality histograms (A8, src/peprocess.cpp:1182-1201 and the clean twin) in LDS
//       (u32, 74 KB: raw-only / both / clean-only -- a kept, untrimmed read is added once; 16 wavefronts per workgroup, two
//       workgroups per CU: the kernel lives on loads in flight); one dword load per lane covers
//       the 128 positions of two reads, clean counts come from the records of the decision kernel (the trimmed kept reads,
//       at the shifted positions: a step most wavefronts skip), the adds are branch-free.  Flushed once per
//       workgroup.  It also runs behind the generic kernel (any capacity), which then only decides.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "snk_common.hip.h"
#include "snk_adapter_bits.hip.h"
#include "snk_planes.hip.h"

using namespace snk;


namespace {

constexpr int LNW = PL_NW;         // plane words of a block: 256 candidate offsets + 64 positions behind them
constexpr int LBLK = PL_BLK;       // candidate offsets per block
constexpr int LVLEN = 32 * LNW - 1;   // characters a non-final block shows to the adapter search

__device__ __forceinline__ u32 zero_bytes(u32 x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); }   // 0x80 per zero byte
__device__ __forceinline__ u32 pack4(u32 z) { return ((z >> 7) | (z >> 14) | (z >> 21) | (z >> 28)) & 0xFu; }           // bits 7,15,23,31 -> 0..3
// 0x80 in the bytes k of a dword at position pos with pos + k < len
__device__ __forceinline__ u32 valid80(int len, int pos) {
    const int d = len - pos;
    return d >= 4 ? 0x80808080u : (d <= 0 ? 0u : (0x80808080u >> (8 * (4 - d))));
}

// the rows are global memory: said explicitly, or the loads through pointers that passed a select / an array become FLAT loads
typedef u32 v4u32 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4u32 *gl_uint4_p;

// planes of the block [p0, p0 + 320) of a read from the plane store: X[k] bit j = read[p0 + j] == "ACGT"[k], XN likewise for 'N';
// ones from vlen on.  Also the A1 base counts of a read of upper-case ACGTN only (src/read_filter.cpp:258-308), which are
// popcounts of its planes: cntA / cntN += the 'A' / 'N' of the block's own positions (256, or all of a final block), other |=
// positions that are neither ACGT nor N (lower case included: such a read takes the sequential path, where case is folded).
__device__ __forceinline__ void block_planes(const u32 *grp, int nquads, int r, int p0, int vlen, bool final, u32 (&X)[4][LNW], u32 (&XN)[LNW],
                                             int &cntA, int &cntN, u32 &other) {
    u32 W[PL_PLANES][12];
    plane_block_words(grp, nquads, r, p0, vlen, W);
#pragma unroll
    for (int w = 0; w < LNW; ++w) {
        const u32 in = lowmask32(vlen - 32 * w);
        const u32 x0 = W[0][w] & in, x1 = W[1][w] & in, x2 = W[2][w] & in, x3 = W[3][w] & in, xn = W[4][w] & in;
        if (w < 8 || final) {
            cntA += __popc(x0);
            cntN += __popc(xn);
            other |= in & ~(x0 | x1 | x2 | x3 | xn);
        }
        X[0][w] = x0 | ~in;
        X[1][w] = x1 | ~in;
        X[2][w] = x2 | ~in;
        X[3][w] = x3 | ~in;
        XN[w] = xn | ~in;
    }
}

__device__ __forceinline__ u64 wave_max64(u64 v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const u64 x = __shfl_xor(v, o, 64);
        v = x > v ? x : v;
    }
    return v;
}

// ---- ahead of the decisions, everything the decide kernel would otherwise walk its rows for: one workgroup per group of 64
// reads of a mate, one wavefront per read and trip, 16 bytes per lane -- a row arrives as one coalesced kilobyte instead of 64
// lanes walking 64 rows (a lane-per-read kernel asks L2 for a sector per 16 bytes: 5.1 ms of waiting per 1 M PE1000 pairs).
//   * the quality half of A1: low-quality count and quality sum, packed (count << 20 | sum: a row holds at most 1024 qualities
//     of at most 255) into the first word of the read's record, which the decide kernel rewrites at its end;
//   * the base planes: every lane turns its 16 characters into 16 bits of each of the 5 planes, two lanes make a word, the
//     workgroup collects the group's words in LDS (40 KB) in the plane store's order and copies them out as one contiguous block.
// SEG = lanes per read: 64, or 32 when the rows are at most 512 bytes -- two reads per wavefront and trip then.
template <int SEG>
__global__ void __launch_bounds__(256)
snk_long_prep_kernel(const DevParams *Pp, DevBatch B, int lcap, u32 *planes, long ngroups, int nquads) {
    HIP_DYNAMIC_SHARED(u32, lds)                                      // nquads * 5 KB: block (quad, plane) at (quad * 5 + plane) * 256, its cells XOR-skewed
                                                                     // by quad * 4: the 32 words of a read fall into the 32 banks (8 quads: 40 KB, 4 per CU)
    const DevParams &P = *Pp;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int mates = P.paired ? 2 : 1;
    const u32 KL = ((u32)min(max(P.phred + P.low_qual, 0), 127) * 0x01010101u) | 0x80808080u;
    constexpr int RPT = 64 / SEG;                                    // reads per wavefront and trip
    const int sub = lane / SEG, ll = lane % SEG;                     // which of them, and the lane's 16-byte piece of its row
    const int pitch = B.pitch, pos = 16 * ll;
    const bool in_row = pos + 16 <= pitch;
    constexpr int U = 4;
    for (long gi = blockIdx.x; gi < ngroups * mates; gi += gridDim.x) {
        const int m = (int)(gi / ngroups);
        const long g = gi - (long)m * ngroups;
        for (int t0 = 0; t0 < 16; t0 += U * RPT) {
            v4u32 qv[U], sv[U];
            int ln[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const long i = g * 64 + wv * 16 + t0 + k * RPT + sub;
                const bool have = i < B.n;
                int len = !have ? 0 : (B.len[m] ? (int)B.len[m][i] : B.fixed_len[m]);
                if (len > lcap) len = 0;                             // (too long: reported by the decide kernel)
                ln[k] = len;
                const long row = (have ? i : 0) * (long)pitch;
                const bool on = in_row && pos < len;
                qv[k] = on ? ((gl_uint4_p)(B.qual[m] + row))[ll] : v4u32{0, 0, 0, 0};
                sv[k] = on ? ((gl_uint4_p)(B.seq[m] + row))[ll] : v4u32{0, 0, 0, 0};
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int r = wv * 16 + t0 + k * RPT + sub;
                const long i = g * 64 + r;
                // qualities
                const u32 qd[4] = {qv[k].x, qv[k].y, qv[k].z, qv[k].w};
                u32 lq = 0, qsum = 0;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const u32 ok = valid80(ln[k], pos + 4 * d);
                    lq += (u32)__popc((KL - (qd[d] & 0x7F7F7F7Fu)) & ~qd[d] & ok);
                    qsum = __builtin_amdgcn_sad_u8(qd[d] & ((ok >> 7) * 0xFFu), 0u, qsum);
                }
                u32 v = (lq << 20) + qsum;
#pragma unroll
                for (int o = SEG / 2; o >= 1; o >>= 1) v += (u32)__shfl_xor((int)v, o, 64);
                if (ll == 0 && i < B.n) reinterpret_cast<u32 *>(B.out[m] + i)[0] = v;
                // bases: 16 bits of every plane
                const u32 sd[4] = {sv[k].x, sv[k].y, sv[k].z, sv[k].w};
                u32 e = 0, c1 = 0, c2 = 0, nn = 0;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const u32 x = sd[d];
                    const u32 t = x & 0x06060606u;
                    const u32 ex = __builtin_amdgcn_perm(0x00470054u, 0x00430041u, t);
                    e |= pack4(zero_bytes(ex ^ x)) << (4 * d);
                    nn |= pack4(zero_bytes(x ^ 0x4E4E4E4Eu)) << (4 * d);
                    c1 |= pack4((x << 6) & 0x80808080u) << (4 * d);
                    c2 |= pack4((x << 5) & 0x80808080u) << (4 * d);
                }
                const u32 in = lowmask32(ln[k] - pos) & 0xFFFFu;
                e &= in;
                const u32 a01 = (e & ~c1 & ~c2) | ((e & c1 & ~c2) << 16);      // plane 0 | plane 1 << 16
                const u32 a23 = (e & c1 & c2) | ((e & ~c1 & c2) << 16);
                const u32 a4 = nn & in;
                const u32 b01 = (u32)__shfl_xor((int)a01, 1, 64), b23 = (u32)__shfl_xor((int)a23, 1, 64), b4 = (u32)__shfl_xor((int)a4, 1, 64);
                // an even lane 2w and its odd neighbour hold positions [32w, 32w + 16) and [32w + 16, 32w + 32)
                const int w = ll >> 1, quad = w >> 2;
                const bool odd = ll & 1;
                u32 *cell = lds + (long)(quad < nquads ? quad : 0) * PL_PLANES * 256 + ((r * 4 + (w & 3)) ^ (quad * 4));
                const bool keep = quad < nquads;                     // (words past the capacity are not stored)
                const u32 wA = odd ? ((b01 >> 16) | (a01 & 0xFFFF0000u)) : ((a01 & 0xFFFFu) | (b01 << 16));      // plane 1 (odd) / plane 0 (even)
                const u32 wB = odd ? ((b23 >> 16) | (a23 & 0xFFFF0000u)) : ((a23 & 0xFFFFu) | (b23 << 16));      // plane 3 / plane 2
                if (keep) {
                    cell[(odd ? 1 : 0) * 256] = wA;
                    cell[(odd ? 3 : 2) * 256] = wB;
                    if (!odd) cell[4 * 256] = (a4 & 0xFFFFu) | (b4 << 16);
                }
            }
        }
        __syncthreads();
        u32 *dst = planes + ((long)m * ngroups + g) * nquads * PL_QUAD_DWORDS;
        for (int blk = wv; blk < nquads * PL_PLANES; blk += 4) {       // blocks of 1 KB
            const int quad = blk / PL_PLANES;
            const v4u32 x = *reinterpret_cast<const v4u32 *>(lds + (long)blk * 256 + ((lane * 4) ^ (quad * 4)));
            *reinterpret_cast<v4u32 *>(dst + (long)blk * 256 + lane * 4) = x;
        }
        __syncthreads();
    }
}

#ifndef SNK_LONG_WPE
#define SNK_LONG_WPE 2           // waves per SIMD the decide kernel's register allocation aims at
#endif
__global__ void __launch_bounds__(256, SNK_LONG_WPE)
snk_long_decide_kernel(const DevParams *Pp, const TileAdapters TA, DevBatch B, DevStats st, int lcap, int nq, const u32 *planes, long ngroups, int nquads) {
    // the parameter block through the constant address space: every field is a scalar load the compiler may keep across the loop
    // (through the generic reference they were vector loads with a uniform address -- some sixty of them per pair --, their values
    // VGPRs, and every branch on a parameter a branch under an EXEC mask)
    typedef __attribute__((address_space(4))) DevParams CDevParams;
    const CDevParams &P = *(const CDevParams *)(uintptr_t)Pp;
    const long fb = file_block(lcap, nq);
    const long ts_off = SNK_GS_N + (long)lcap * 5 + (long)lcap * nq;
    const int pe = P.paired ? 1 : 0;
    // the workgroup's share of the counters every pair touches -- filter statistics, reads numbers, "last read" words -- collects in
    // LDS and goes out once at the end: one global atomic per wavefront and trip on the same dozen words was most of this kernel's
    // time (same-address atomics run at some 25 M/s: 62 k trips of 4 M PE300 pairs = 2.5 of its 3.0 ms)
    __shared__ u32 wfs[SNK_FS_N], wreads[4];
    __shared__ unsigned long long wmax[4];
    if (threadIdx.x < SNK_FS_N) wfs[threadIdx.x] = 0;
    if (threadIdx.x < 4) { wreads[threadIdx.x] = 0; wmax[threadIdx.x] = 0; }
    __syncthreads();
    const long nround = (B.n + 255) / 256 * 256;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nround; i += (long)gridDim.x * blockDim.x) {
        const bool exists = i < B.n;
        ReadState r[2];
        int cf[2] = {0, 0};
        const uint8_t *s[2] = {nullptr, nullptr}, *q[2] = {nullptr, nullptr};
        const unsigned long long gidx = B.first_index + (unsigned long long)i;
        bool bad = !exists;
        // one copy of the mate's code per mate, the mate a compile-time constant: r[], s[], q[] indexed by constants stay in registers
        // and the rows keep their global address space (as a loop under `#pragma unroll` it stayed a loop -- its body is too large for
        // the unroller --, and the dynamically indexed ReadState pair lived in scratch memory: 60 scratch loads and 28 stores per pair)
        auto mate = [&](auto MC) {
            constexpr int m = decltype(MC)::value;
            int len = 0;
            if (exists) {
                len = B.len[m] ? (int)B.len[m][i] : B.fixed_len[m];
                s[m] = B.seq[m] + i * (long)B.pitch;
                q[m] = B.qual[m] + i * (long)B.pitch;
                if (len > lcap && !bad) { report_err(st, gidx, m, SNK_E_TOO_LONG); bad = true; }
            }
            const bool live = exists && !bad;
            rs_init(r[m], len);
            if (live && (P.n_ct[m] | P.n_gct))                       // verdicts of snk_long_contam_kernel (snk_contam.hip), bits 2m, 2m + 1
                cf[m] = B.cf ? ((int)B.cf[i] >> (2 * m)) & 3 : contam_flags(P.ct + m * SNK_MAX_CONTAMS, P.n_ct[m], P.gct, P.n_gct, s[m], len);
            // ---- the read in blocks of 256 positions (uniform control flow: every lane walks along): its planes give the base
            // counts and the adapter search; the qualities are a pass of their own; the poly-X run, when asked for, too
            const bool longish = live && len >= 64;                  // (shorter: the sequential functions)
            const int n_ada = P.n_ada[m];
            int cntA = 0, cntN = 0;
            int best_a = 0x7fffffff, best_pos = -1;                  // the adapter earliest in the list with a hit so far, and where
            bool best_c = false;                                     // ... a phase C hit: the same adapter's phase C hit in a later block comes first in the reference's order
            u32 other = 0;
            bool through = false;                                    // this lane has seen its final block
            for (int p0 = 0; __any(longish && !through); p0 += LBLK) {
                const int rem = len - p0;
                const bool here = longish && !through, final = rem <= LVLEN;
                const int vlen = here ? (final ? rem : LVLEN) : 0;
                u32 X[4][LNW], XN[LNW];
                block_planes(planes + ((long)m * ngroups + (i >> 6)) * nquads * PL_QUAD_DWORDS, nquads, (int)(i & 63), here ? p0 : 0, vlen, final, X, XN, cntA, cntN, other);
                if (n_ada > 0) {
                    // the first adapter of the list with a hit decides, whatever block its hit is in (src/read_filter.cpp:175-188):
                    // only adapters in front of the best one so far are still searched
                    // (an adapter of up to 64 characters has its phase C offsets in the read's last block only; a longer one may have
                    // them in the last TWO: phase C is walked by descending offset, src/read_filter.cpp:765-788, so a phase C hit
                    // of the best adapter in a later block replaces its phase C hit of the block before)
                    for (int a = 0; a < n_ada; ++a) {
                        const bool todo = here && (a < best_a || (a == best_a && best_c));
                        if (__any(todo)) {
                            const CTileAdapter &Acur = ((const CTileAdapter *)(uintptr_t)P.tile_ada)[m * P.ada_stride + a];
                            bool hc = false;
                            // the block as adapter_tile sees it: `rem` characters are left of the read (the planes show 320 of them), the
                            // block owns its first 256 offsets, the last block all of its own
                            const int rel = adapter_tile<LNW, true>(Acur, P.ada[m * P.ada_stride + a], X, XN, here ? rem : 0, todo, s[m] + p0, p0 == 0, true, false,
                                                                    final ? 32 * LNW : LBLK, &hc);
                            if (todo && rel >= 0) { best_a = a; best_pos = p0 + rel; best_c = hc; }
                        }
                    }
                }
                through |= here && final;
            }
            const bool fast = longish && other == 0;
            int e = SNK_OK;
            if (live && !fast) {                                   // anything unusual: the sequential restatement, in this lane
                stat_read_dev(P, m, s[m], q[m], len, r[m], e);
                if (e) { report_err(st, gidx, m, e); bad = true; }
            }
            if (fast) {
                r[m].n_a = cntA;
                r[m].n_n = cntN;
                {                                                   // the quality half: from snk_long_qstat_kernel
                    const u32 qs = __builtin_nontemporal_load(reinterpret_cast<const u32 *>(B.out[m] + i));
                    r[m].lowq = P.phred + P.low_qual >= 0 ? (int)(qs >> 20) : 0;
                    r[m].sumq = (int)(qs & 0xFFFFFu) - P.phred * len;
                }
                if (P.polyX_num != -1) {                            // contig_base (:262-268): one character at a time
                    int last = 'Q', run = 0, maxrun = 1;
                    const gl_uint4_p s4 = (gl_uint4_p)s[m];
                    for (int pos = 0; pos < len; pos += 16) {
                        const v4u32 sv = s4[pos >> 4];
                        const u32 sd[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            if (pos + k < len) {
                                const int c = (int)((sd[k >> 2] >> (8 * (k & 3))) & 0xFFu);
                                if (c == last) { if (++run > maxrun) maxrun = run; } else run = 1;
                                last = c;
                            }
                        }
                    }
                    r[m].polyx = maxrun >= P.polyX_num ? 1 : 0;
                }
                const int ada_pos = best_pos;
                if (ada_pos >= 0) { r[m].inc_ada = 1; r[m].adacut = len - ada_pos; }
            }
        };
        mate(std::integral_constant<int, 0>{});
        if (pe) mate(std::integral_constant<int, 1>{});
        const bool ok = exists && !bad;
        if (ok) {
            fastq_trim_dev(P, 0, s[0], q[0], r[0]);
            if (pe) fastq_trim_dev(P, 1, s[1], q[1], r[1]);
        }
        int v = 0, reason = 255;
        if (ok) {
            reason = pe ? discard_reason(P, r[0], r[1], B.dup ? B.dup[i] : 0, v, cf[0], cf[1])
                        : discard_reason(P, r[0], r[0], B.dup ? B.dup[i] : 0, v, cf[0], cf[0]);
            count_reason(wfs, pe, reason, v);
        }
        if (exists) {                                                // (a pair that raised an error gets a record no later pass uses)
            store_rec(B.out[0], i, r[0], reason, v);
            if (pe) store_rec(B.out[1], i, r[1], reason, v);
        }
        const unsigned long long key = (gidx + 1) << 16;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if (m > pe) continue;
            if (ok) {
                unsigned long long *file = st.sum + SNK_FS_N + m * fb;
                int hh = -1, lh = -1, ht = -1, lt = -1, ad = -1;
                if (P.copy_back) { hh = r[m].hd_h; lh = r[m].lq_h; ht = r[m].hd_t; lt = r[m].lq_t; ad = r[m].adacut; }
                ts_update(file + ts_off, hh, lh, ht, lt, ad, (pe && m == 1) ? r[m].len : 0, !pe);
            }
            // reads_number and the "last read" word of the raw / clean files: one atomic per wave
            const unsigned long long nraw = __popcll(__ballot(ok)), ncl = __popcll(__ballot(ok && reason == SNK_KEEP));
            const u64 kraw = wave_max64(ok ? (key | (u64)r[m].len) : 0ull);
            const u64 kcl = wave_max64((ok && reason == SNK_KEEP) ? (key | (u64)r[m].clen) : 0ull);
            if ((threadIdx.x & 63) == 0) {
                if (nraw) { atomicAdd(&wreads[m], (u32)nraw); atomicMax(&wmax[m], kraw); }
                if (ncl) { atomicAdd(&wreads[2 + m], (u32)ncl); atomicMax(&wmax[2 + m], kcl); }
            }
            if (ok && reason == SNK_KEEP) {
                unsigned long long *file = st.sum + SNK_FS_N + (2 + m) * fb;
                ts_update(file + ts_off, r[m].hd_h, r[m].lq_h, r[m].hd_t, r[m].lq_t, r[m].adacut,
                          (pe && m == 1) ? r[m].clen : r[m].len, !pe);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < SNK_FS_N && wfs[threadIdx.x]) atomicAdd(&st.sum[threadIdx.x], (unsigned long long)wfs[threadIdx.x]);
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;                                   // 0, 1: raw files of the mates; 2, 3: clean files
        if (wreads[k]) atomicAdd(&st.sum[SNK_FS_N + k * fb + SNK_GS_READS], (unsigned long long)wreads[k]);
        if (wmax[k]) atomicMax(&st.maxb[k], wmax[k]);
    }
}

// ---- per-position histograms.  blockIdx.x = (mate, position block of 128, slice of the batch)
constexpr int HPB = 128;

// LDS word of (row, slot); row 5 + nq is a bin nobody reads: lanes past a read's end and out-of-range qualities add there,
// so the adds need no branches
__device__ __forceinline__ void hist_add(u32 *hh, int slot, int trash, int nq, int phred, u32 c, u32 qc, bool on, bool &err) {
    const u32 u = c & 0xDFu, t = (u >> 1) & 3u;                     // A 0, C 1, T 2, G 3
    const u32 sel = t | 0x0C0C0C00u;                                // v_perm: byte t of the constant into byte 0, zeros above
    const u32 ex = __builtin_amdgcn_perm(0u, 0x47544341u, sel);     // the letter of that code ("ACTG")
    const u32 bt = __builtin_amdgcn_perm(0u, 0x02030100u, sel);     // its row: A C G T -> 0 1 2 3
    int b = u == ex ? (int)bt : 4;                                  // anything else (N): 4
    const int bq = (int)qc - phred;
    const bool qok = (unsigned)bq < (unsigned)nq;
    err |= on && !qok;
    b = on ? b : trash;
    const int qrow = (on && qok) ? 5 + bq : trash;
    atomicAdd(&hh[b * HPB + slot], 1u);
    atomicAdd(&hh[qrow * HPB + slot], 1u);
}

// The same for the four positions of a dword at once (columns 32 b + j, b = 0..3; the first nv of them exist): rows as bytes by
// SWAR -- v_perm picks letter and row of four codes at a time, the quality range test is two byte-wise subtractions that cannot
// borrow (needs phred + nq <= 128: the caller checks) -- then one shift-add per LDS add.  38 VALU per dword instead of 64.
__device__ __forceinline__ void hist_add4(u32 *hh, int j, int trash, int nq, int phred, u32 cw, u32 qw, int nv, bool &err) {
    const u32 validm = nv >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nv)) - 1u);
    const u32 u4 = cw & 0xDFDFDFDFu, t4 = (u4 >> 1) & 0x03030303u;       // A 0, C 1, T 2, G 3
    const u32 ex4 = __builtin_amdgcn_perm(0u, 0x47544341u, t4);           // the letters of those codes ("ACTG")
    const u32 bt4 = __builtin_amdgcn_perm(0u, 0x02030100u, t4);           // their rows: A C G T -> 0 1 2 3
    const u32 eqm = (zero_bytes(u4 ^ ex4) >> 7) * 0xFFu;
    const u32 q7 = qw & 0x7F7F7F7Fu;
    const u32 ge = (q7 | 0x80808080u) - (u32)phred * 0x01010101u;        // bit 7 of a byte: q7 >= phred; its low bits: q7 - phred then
    const u32 lt = (((u32)(phred + nq - 1) * 0x01010101u) | 0x80808080u) - q7;   // bit 7: q7 <= phred + nq - 1
    const u32 okm = ((ge & lt & ~qw & 0x80808080u) >> 7) * 0xFFu & validm;
    err |= (validm & ~okm) != 0;
    const u32 tr4 = (u32)trash * 0x01010101u;
    const u32 b4 = (((bt4 & eqm) | (0x04040404u & ~eqm)) & validm) | (tr4 & ~validm);
    const u32 q4 = (((ge & 0x7F7F7F7Fu) + 0x05050505u) & okm) | (tr4 & ~okm);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        atomicAdd(&hh[((b4 >> (8 * b)) & 0xFFu) * HPB + 32 * b + j], 1u);
        atomicAdd(&hh[((q4 >> (8 * b)) & 0xFFu) * HPB + 32 * b + j], 1u);
    }
}

// Every lane takes FOUR positions of a read as one dword load (lanes 0-31: the 128 positions of one read, lanes 32-63: of
// the next one) -- 256 bytes per load instruction instead of 64, which is what the kernel lives on: it is bound by the bytes
// in flight.  Position 4j + k sits in column 32k + j of a bin row, so the 32 lanes of a half hit 32 different banks.
#ifndef SNK_HIST_U
#define SNK_HIST_U 2
#endif
#ifndef SNK_HIST_T
#define SNK_HIST_T 1024
#endif
__global__ void __launch_bounds__(SNK_HIST_T)
snk_long_hist_kernel(const DevParams *Pp, DevBatch B, DevStats st, int lcap, int nq, int nblk, int slices) {
    // Three histograms: a kept read that nothing was cut from counts the same in the raw and the clean statistics -- it is added
    // once, to `both`; the other reads add their raw characters to `raw` and (kept, trimmed) their clean range to `clean`, in a
    // second step that most wavefronts skip.  raw = raw + both, clean = clean + both at the flush.
    HIP_DYNAMIC_SHARED(u32, h)                        // raw[(5 + nq + 1)][128] | both[...] | clean[...]
    const int rows = 5 + nq, words = (rows + 1) * HPB;
    u32 *hraw = h, *hcl = h + 2 * words;
    int id = blockIdx.x;
    const int slice = id % slices; id /= slices;
    const int pb = id % nblk, m = id / nblk;
    for (int k = threadIdx.x; k < 3 * words; k += blockDim.x) h[k] = 0;
    __syncthreads();
    const long per = (B.n + slices - 1) / slices;
    const long r0 = (long)slice * per, r1 = min(B.n, r0 + per);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = lane >> 5, j = lane & 31;
    const int base = pb * HPB, phred = __builtin_amdgcn_readfirstlane(Pp->phred);
    const bool swar = phred >= 0 && phred + nq <= 128 && rows < 256;   // (uniform) what hist_add4's byte arithmetic needs
    const uint8_t *seq = B.seq[m], *qual = B.qual[m];
    const snk_read_result *rec = B.out[m];
    const uint16_t *lens = B.len[m];
    const int fixed = B.fixed_len[m], pitch = B.pitch;
    const long fb = file_block(lcap, nq);
    constexpr int U = SNK_HIST_U;                                   // pairs of reads per trip: their loads go out together
    for (long rr = r0 + wave * 2 * U; rr < r1; rr += (SNK_HIST_T / 64) * 2 * U) {
        u32 cw[U], qw[U];
        int nv[U], nc[U], st0[U], sel[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const long rk = rr + 2 * k + half;
            const bool have = rk < r1;
            const long r = have ? rk : r1 - 1;
            int l = lens ? (int)lens[r] : fixed;
            if (l > lcap || !have) l = 0;                           // (too long: reported by the first kernel)
            const snk_read_result x = rec[r];
            const int start = x.clean_start;
            int cl = (have && x.reason == SNK_KEEP) ? (int)x.clean_len : 0;
            if (start + cl > l) cl = 0;                             // (never for a record of the decision kernels)
            const bool same = cl == l && start == 0;                 // (cl == l > 0: kept as it came)
            nv[k] = max(min(l - base - 4 * j, 4), 0);               // positions of this lane's dword the read has
            nc[k] = same ? 0 : max(min(cl - base - 4 * j, 4), 0);   // ... its clean range has, when that is not the same thing
            st0[k] = start;
            sel[k] = same ? words : 0;
            const uint8_t *ps = seq + r * (long)pitch, *pq = qual + r * (long)pitch;
            const int off = base + 4 * j;
            cw[k] = nv[k] ? *reinterpret_cast<const u32 *>(ps + off) : 0u;
            qw[k] = nv[k] ? *reinterpret_cast<const u32 *>(pq + off) : 0u;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            bool e = false, ec = false;
            u32 *hh = hraw + sel[k];
            if (swar) hist_add4(hh, j, rows, nq, phred, cw[k], qw[k], nv[k], e);
            else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    hist_add(hh, 32 * b + j, rows, nq, phred, (cw[k] >> (8 * b)) & 0xFFu, (qw[k] >> (8 * b)) & 0xFFu, b < nv[k], e);
            }
            // a quality outside [0, nq) of the RAW pass is the reference's heap corruption (src/peprocess.cpp:1196): first one reported
            const unsigned long long em = __ballot(e);
            if (em && lane == 0) {
                if (em & 0xFFFFFFFFull) report_err(st, B.first_index + (u64)(rr + 2 * k), m, SNK_E_QUAL_RANGE);
                if (em >> 32) report_err(st, B.first_index + (u64)(rr + 2 * k + 1), m, SNK_E_QUAL_RANGE);
            }
            if (__ballot(nc[k] != 0) == 0) continue;                // (uniform) nobody's clean range differs from the read here
            u32 ccw = cw[k], cqw = qw[k];
            if (st0[k] != 0 && nc[k]) {                             // head-trimmed: the clean read sits at shifted positions
                const long r = rr + 2 * k + half;
                const uint8_t *ps = seq + r * (long)pitch, *pq = qual + r * (long)pitch;
                const int so = st0[k] + base + 4 * j, al = so & ~3, sh = 8 * (so & 3);
                const u32 s0 = *reinterpret_cast<const u32 *>(ps + al), q0 = *reinterpret_cast<const u32 *>(pq + al);
                const u32 s1 = (sh && al + 8 <= pitch) ? *reinterpret_cast<const u32 *>(ps + al + 4) : 0u;
                const u32 q1 = (sh && al + 8 <= pitch) ? *reinterpret_cast<const u32 *>(pq + al + 4) : 0u;
                ccw = __builtin_amdgcn_alignbit(s1, s0, sh);
                cqw = __builtin_amdgcn_alignbit(q1, q0, sh);
            }
            if (swar) hist_add4(hcl, j, rows, nq, phred, ccw, cqw, nc[k], ec);
            else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    hist_add(hcl, 32 * b + j, rows, nq, phred, (ccw >> (8 * b)) & 0xFFu, (cqw >> (8 * b)) & 0xFFu, b < nc[k], ec);
            }
        }
    }
    __syncthreads();
    unsigned long long *fraw = st.sum + SNK_FS_N + m * fb + SNK_GS_N, *fcl = st.sum + SNK_FS_N + (2 + m) * fb + SNK_GS_N;
    for (int k = threadIdx.x; k < rows * HPB; k += blockDim.x) {
        const int row = k / HPB, col = k - row * HPB, p = base + 4 * (col & 31) + (col >> 5);
        if (p >= lcap) continue;
        const long off = row < 5 ? (long)p * 5 + row : (long)lcap * 5 + (long)p * nq + (row - 5);
        const u32 both = h[words + k], a = hraw[k] + both, b = hcl[k] + both;
        if (a) atomicAdd(&fraw[off], (unsigned long long)a);
        if (b) atomicAdd(&fcl[off], (unsigned long long)b);
    }
}

}  // namespace

// returns 0 when this path cannot take the batch (the caller falls back to the generic kernel)
size_t snk_long_scratch_bytes(long n, int paired, int lcap) { return (size_t)plane_store_dwords(n, paired ? 2 : 1, plane_quads(lcap)) * sizeof(u32); }

int snk_launch_long(const DevParams *dp, const DevParams &hp, const TileAdapters &ta, const DevBatch &b, const DevStats &st, int lcap, int nq,
                    int n_cu, unsigned *planes, unsigned char *cf, void *stream) {
    if (!hp.tile_ok || lcap <= 256 || lcap > 1024 || b.n <= 0 || !planes) return 0;
    if (b.pitch % 16 != 0 || b.pitch < ((lcap + 15) & ~15)) return 0;
    if ((((uintptr_t)b.seq[0] | (uintptr_t)b.qual[0] | (uintptr_t)b.seq[1] | (uintptr_t)b.qual[1]) % 16) != 0) return 0;
    long wgs = (b.n + 255) / 256;
    if (wgs > (long)n_cu * 8) wgs = (long)n_cu * 8;
    const long ngroups = (b.n + 63) / 64;
    const int nquads = plane_quads(lcap);
    {
        long pw = ngroups * (hp.paired ? 2 : 1);
        if (pw > (long)n_cu * 16) pw = (long)n_cu * 16;
        const size_t lds = (size_t)nquads * PL_QUAD_DWORDS * sizeof(u32);
        if (b.pitch <= 512) hipLaunchKernelGGL(snk_long_prep_kernel<32>, dim3((unsigned)pw), dim3(256), lds, (hipStream_t)stream, dp, b, lcap, planes, ngroups, nquads);
        else hipLaunchKernelGGL(snk_long_prep_kernel<64>, dim3((unsigned)pw), dim3(256), lds, (hipStream_t)stream, dp, b, lcap, planes, ngroups, nquads);
    }
    DevBatch bd = b;
    bd.cf = nullptr;
    if (cf && (hp.n_ct[0] | hp.n_ct[1] | hp.n_gct)) {
        snk_launch_long_contam(dp, b, cf, hp.n_ct[0] > hp.n_ct[1] ? hp.n_ct[0] : hp.n_ct[1], hp.n_gct, planes, nquads, stream);
        bd.cf = cf;
    }
    hipLaunchKernelGGL(snk_long_decide_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, dp, ta, bd, st, lcap, nq, (const u32 *)planes, ngroups, nquads);
    return snk_launch_hist(dp, hp.paired, b, st, lcap, nq, n_cu, stream);
}

// the histogram kernel on its own: behind the long-read decide kernel, and behind the generic kernel (any capacity)
int snk_launch_hist(const DevParams *dp, int paired, const DevBatch &b, const DevStats &st, int lcap, int nq, int n_cu, void *stream) {
    if (b.n <= 0 || lcap <= 0 || (b.pitch & 3)) return 0;
    if ((((uintptr_t)b.seq[0] | (uintptr_t)b.qual[0] | (uintptr_t)b.seq[1] | (uintptr_t)b.qual[1]) & 3) != 0) return 0;
    const int nblk = (lcap + HPB - 1) / HPB, mates = paired ? 2 : 1;
    const size_t shmem = (size_t)3 * (5 + nq + 1) * HPB * sizeof(u32);
    if (shmem > 150 * 1024) return 0;
    int slices = (int)((long)n_cu * 2 / (nblk * mates));
    if (slices < 1) slices = 1;
    while (slices > 1 && b.n / slices < 512) --slices;              // a flush per workgroup wants some reads behind it
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)snk_long_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(snk_long_hist_kernel, dim3((unsigned)(mates * nblk * slices)), dim3(SNK_HIST_T), shmem, (hipStream_t)stream, dp, b, st, lcap, nq,
                       nblk, slices);
    return 1;
}
