// snk_rmdup.hip -- the duplicate-marking pre-pass of `SOAPnuke filter` (config key rmdup) on gfx950.
//
//  hash   one uint64 per raw pair: std::hash<std::string>(seq1 + seq2) as the reference computes
//         it (src/peprocess.cpp:3665,3680), i.e. libstdc++ _Hash_bytes (64-bit: seed 0xc70f6907,
//         mul 0xc6a4a7935bd1e995, shift_mix(v) = v ^ (v >> 47)).  The chain over the 8-byte words of
//         a string is sequential, so one LANE owns one pair and walks its words; the wave first
//         pulls its 64 rows in with coalesced 16-byte loads and parks them in LDS with a padded
//         pitch (row-per-lane reads straight from HBM would touch 64 cache lines per instruction).
//         The seam between mate 1 and mate 2 (len1 % 8 != 0) is a per-lane funnel shift.
//         HBM-bound in theory: 2*L bytes in, 8 bytes out per pair.
//  mark   rmdup::markDup (src/rmdup.cpp:14-149) = "this hash value occurred at an earlier index":
//         an open-addressing table in HBM (uint64 key, uint32 smallest index), insert with
//         atomicCAS + atomicMin, then one lookup per pair.  Random-access HBM work.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "snk_device.h"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr u64 HMUL = 0xc6a4a7935bd1e995ull, HSEED = 0xc70f6907ull, EMPTY = ~0ull;

__device__ __forceinline__ u64 shift_mix(u64 v) { return v ^ (v >> 47); }

// One lane = one pair; the chain over the 8-byte words of mate 1 ++ mate 2 runs in two legs so that
// only one mate's rows have to be staged at a time.  LD(w) returns the w-th little-endian word of the
// lane's row of the mate currently staged.
struct Chain {
    u64 h, carry, bprev;
    int q, r8, nfull, t8;
};

__device__ __forceinline__ int wave_max(int v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ void mix_word(Chain &c, int k, u64 d) {
    if (k < c.nfull) {
        c.h ^= shift_mix(d * HMUL) * HMUL;
        c.h *= HMUL;
    } else if (k == c.nfull && c.t8) {                          // load_bytes: the last len % 8 bytes
        c.h ^= d & ((1ull << c.t8) - 1ull);
        c.h *= HMUL;
    }
}
__device__ __forceinline__ void chain_begin(Chain &c, int l1, int l2, bool valid) {
    const int total = l1 + l2;
    c.q = l1 >> 3;
    c.r8 = (l1 & 7) * 8;
    c.nfull = valid ? (total >> 3) : 0;
    c.t8 = valid ? (total & 7) * 8 : 0;
    c.h = HSEED ^ ((u64)total * HMUL);
    c.carry = c.bprev = 0;
}
// leg 1: the whole words of mate 1, and the bytes of its last partial word (the seam)
template <class LD>
__device__ __forceinline__ void chain_mate1(Chain &c, int wmax, bool valid, LD &&ld) {
    const int steps = wave_max(valid ? c.q : 0);
    for (int k = 0; k < steps; ++k) {
        const u64 a = ld(min(k, wmax));
        if (k < c.q) mix_word(c, k, a);
    }
    const u64 last = ld(min(c.q, wmax));
    c.carry = c.r8 ? (last & ((1ull << c.r8) - 1ull)) : 0ull;
}
// leg 2: mate 2 shifted behind the seam (PAIRED), or just the tail bytes of mate 1 (single end: ld unused)
template <bool PAIRED, class LD>
__device__ __forceinline__ u64 chain_mate2(Chain &c, int wmax, bool valid, LD &&ld) {
    const int steps = wave_max(valid ? c.nfull - c.q + (c.t8 ? 1 : 0) : 0);
    for (int j = 0; j < steps; ++j) {
        const u64 bcur = PAIRED ? ld(min(j, wmax)) : 0ull;
        u64 d;
        if (c.r8 == 0) d = bcur;
        else d = ((j == 0) ? c.carry : (c.bprev >> (64 - c.r8))) | (bcur << c.r8);
        c.bprev = bcur;
        mix_word(c, c.q + j, d);
    }
    const u64 h = shift_mix(c.h) * HMUL;
    return shift_mix(h);
}

struct HashArgs {
    const uint8_t *seq[2];
    const uint16_t *len[2];
    int fixed_len[2];
    int pitch;
    long n;
    u64 *out;
};

// rows staged through LDS one mate at a time (pitch a multiple of 16, 16-byte aligned planes)
template <bool PAIRED>
__global__ void __launch_bounds__(256) snk_hash_lds_kernel(const HashArgs A, const int p2, const long tiles) {
    HIP_DYNAMIC_SHARED(uint8_t, smem)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = blockDim.x >> 6;
    uint8_t *rows = smem + (size_t)wave * 64 * p2;              // [row][p2], mate 1 then mate 2
    const int upr = A.pitch >> 4;                               // 16-byte units per row
    const int rpp = 64 / upr;                                   // rows per pass of the wave
    const int rowoff = lane / upr, col = lane - rowoff * upr;
    const bool act = lane < rpp * upr;
    const int wmax = (A.pitch >> 3) - 1;
    const uint8_t *myrow = rows + (size_t)lane * p2;
    auto ld = [&](int w) { return *reinterpret_cast<const u64 *>(myrow + 8 * w); };
    for (long tile = (long)blockIdx.x * W + wave; tile < tiles; tile += (long)gridDim.x * W) {
        const long t0 = tile * 64;
        const long rem = A.n - t0;
        const int cnt = rem >= 64 ? 64 : (int)rem;
        auto stage = [&](int m) {                               // coalesced 16-byte loads -> padded LDS rows
            const uint8_t *base = A.seq[m] + t0 * (long)A.pitch;
            SNK_WAVE_SYNC();                                    // (the rows staged before have been read)
            for (int row0 = 0; row0 < cnt; row0 += rpp) {
                const int row = row0 + rowoff;
                if (act && row < cnt)
                    *reinterpret_cast<uint4 *>(rows + (size_t)row * p2 + col * 16) =
                        *reinterpret_cast<const uint4 *>(base + (long)row * A.pitch + col * 16);
            }
            SNK_WAVE_SYNC();
        };
        // (one wave reads only what it wrote itself: no barrier, LDS ops of a wave are ordered)
        const bool valid = lane < cnt;
        int l1 = 0, l2 = 0;
        if (valid) {
            l1 = A.len[0] ? (int)A.len[0][t0 + lane] : A.fixed_len[0];
            if (PAIRED) l2 = A.len[1] ? (int)A.len[1][t0 + lane] : A.fixed_len[1];
            l1 = min(l1, A.pitch);
            l2 = min(l2, A.pitch);
        }
        Chain c;
        chain_begin(c, l1, l2, valid);
        stage(0);
        chain_mate1(c, wmax, valid, ld);
        if (PAIRED) stage(1);
        const u64 h = chain_mate2<PAIRED>(c, wmax, valid, ld);
        if (valid) A.out[t0 + lane] = h;
    }
}

// any pitch / alignment: each lane assembles its words from global bytes (slow path)
template <bool PAIRED>
__global__ void __launch_bounds__(256) snk_hash_direct_kernel(const HashArgs A) {
    const int wmax = (A.pitch + 7) / 8 - 1;
    const long stride = (long)gridDim.x * blockDim.x;
    const long nround = (A.n + 63) / 64 * 64;                   // whole waves stay in the uniform loops
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nround; i += stride) {
        const bool valid = i < A.n;
        int l1 = 0, l2 = 0;
        if (valid) {
            l1 = min(A.len[0] ? (int)A.len[0][i] : A.fixed_len[0], A.pitch);
            if (PAIRED) l2 = min(A.len[1] ? (int)A.len[1][i] : A.fixed_len[1], A.pitch);
        }
        const long ii = valid ? i : 0;
        const int pitch = A.pitch;
        auto words_of = [&](const uint8_t *p) {
            return [=](int w) {
                u64 v = 0;
                for (int b = 7; b >= 0; --b) v = (v << 8) | (u64)((8 * w + b < pitch) ? p[8 * w + b] : 0);
                return v;
            };
        };
        Chain c;
        chain_begin(c, l1, l2, valid);
        chain_mate1(c, wmax, valid, words_of(A.seq[0] + ii * (long)pitch));
        const u64 h = chain_mate2<PAIRED>(c, wmax, valid, words_of((PAIRED ? A.seq[1] : A.seq[0]) + ii * (long)pitch));
        if (valid) A.out[i] = h;
    }
}

__device__ __forceinline__ u64 slot_of(u64 h, int shift) { return (h * 0x9E3779B97F4A7C15ull) >> shift; }

__global__ void __launch_bounds__(256) snk_mark_insert_kernel(const u64 *__restrict__ hash, const u32 *__restrict__ index, long n,
                                                              u64 *keys, u32 *minidx, u64 mask, int shift, u32 *flag) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 h = hash[i];
    const u32 my = index ? index[i] : (u32)i;
    if (h == EMPTY) { atomicOr(flag, 1u); return; }             // the reference's sentinel value: decided apart
    u64 s = slot_of(h, shift) & mask;
    for (;;) {
        const u64 k = atomicCAS(&keys[s], EMPTY, h);
        if (k == EMPTY || k == h) { atomicMin(&minidx[s], my); return; }
        s = (s + 1) & mask;
    }
}

// population of the bucket of 2^64-1 (hash % prime); only runs when that value occurs at all
__global__ void __launch_bounds__(256) snk_bucket_count_kernel(const u64 *__restrict__ hash, long n, u32 prime, const u32 *flag,
                                                               u64 *count) {
    if (flag && *flag == 0) return;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n && (hash[i] % prime) == (EMPTY % prime);
    const u64 b = __ballot(in);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (u64)__popcll(b));
}

__global__ void __launch_bounds__(256) snk_mark_lookup_kernel(const u64 *__restrict__ hash, const u32 *__restrict__ index, long n,
                                                              const u64 *__restrict__ keys, const u32 *__restrict__ minidx, u64 mask,
                                                              int shift, const u64 *bucket_count, long bucket_total, uint8_t *dup) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 h = hash[i];
    const u32 my = index ? index[i] : (u32)i;
    if (h == EMPTY) {                                           // src/rmdup.cpp:100,116
        const u64 pop = bucket_total >= 0 ? (u64)bucket_total : *bucket_count;
        dup[i] = pop > 1 ? 1 : 0;
        return;
    }
    u64 s = slot_of(h, shift) & mask;
    while (keys[s] != h) s = (s + 1) & mask;
    dup[i] = minidx[s] != my ? 1 : 0;
}

// ---- one pass: the table lives across the batches (include/snk_rmdup.h, snk_rmdup_stream_*)
__global__ void __launch_bounds__(256) snk_stream_insert_kernel(const u64 *__restrict__ hash, u32 base, long n, u64 *keys, u32 *minidx,
                                                                u64 mask, int shift, u32 *flag) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 h = hash[i];
    if (h == EMPTY) { atomicOr(flag, 1u); return; }             // the reference's sentinel value: the caller falls back to two passes
    u64 s = slot_of(h, shift) & mask;
    for (u64 probes = 0; probes <= mask; ++probes) {            // (the table is at most half full: bounded all the same, a wedged GPU helps nobody)
        const u64 k = atomicCAS(&keys[s], EMPTY, h);
        if (k == EMPTY || k == h) { atomicMin(&minidx[s], base + (u32)i); return; }
        s = (s + 1) & mask;
    }
    atomicOr(flag, 2u);
}
__global__ void __launch_bounds__(256) snk_stream_lookup_kernel(const u64 *__restrict__ hash, u32 base, long n, const u64 *__restrict__ keys,
                                                                const u32 *__restrict__ minidx, u64 mask, int shift, uint8_t *dup, u64 *marked,
                                                                u32 *flag) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    bool d = false;
    if (i < n) {
        const u64 h = hash[i];
        if (h != EMPTY) {
            u64 s = slot_of(h, shift) & mask, probes = 0;
            while (keys[s] != h && keys[s] != EMPTY && probes <= mask) { s = (s + 1) & mask; ++probes; }
            if (keys[s] == h) d = minidx[s] != base + (u32)i;
            else atomicOr(flag, 2u);                             // not there: the insert of this batch did not run before (never expected)
        }
        dup[i] = d ? 1 : 0;
    }
    const u64 b = __ballot(d);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(marked, (u64)__popcll(b));
}

// Single-end runs of the reference filter read i of a FULL patch with the duplicate flag of read i - 1 (seProcess records "reads
// so far" before it counts the patch's last quality line: src/seprocess.cpp:1086,1159 against src/peprocess.cpp:2147); only the
// partial patch at the end of the file is aligned (:1112).  out[i] = i < n_shifted ? (i ? flags[i - 1] : the flag in front of the
// batch) : flags[i]; the batch's last flag is handed to the next call.
__global__ void __launch_bounds__(256) snk_se_shift_kernel(const uint8_t *__restrict__ flags, long n, long n_shifted, const uint8_t *carry_in, uint8_t *carry_out,
                                                           uint8_t *__restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = i < n_shifted ? (i ? flags[i - 1] : *carry_in) : flags[i];
    if (i == n - 1) *carry_out = flags[i];
}


// ---- the multi-GPU exchange (all-to-all keyed by owner = hash % world, include/snk_rmdup.h).  Grouping a shard's
// hashes by owner: one pass counts (a ballot per owner value present in the wave, one atomic per wave and owner), the host turns
// the counts into group bases, the second pass hands out the places -- again one atomic per wave and owner, the lanes of a wave
// take consecutive places in lane order.  The order inside a group does not matter (the global index travels with the hash).
__global__ void __launch_bounds__(256) snk_owner_count_kernel(const u64 *__restrict__ hash, long n, u32 world, u64 *counts) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n;
    const u32 own = in ? (u32)(hash[i] % world) : 0u;
    u64 todo = __ballot(in);
    while (todo) {                                               // wave-uniform: one trip per owner value among the live lanes
        const int lead = __ffsll(todo) - 1;
        const u32 o = (u32)__shfl((int)own, lead);
        const u64 same = __ballot(in && own == o);
        if ((int)(threadIdx.x & 63) == lead) atomicAdd(&counts[o], (u64)__popcll(same));
        todo &= ~same;
    }
}
__global__ void __launch_bounds__(256) snk_owner_scatter_kernel(const u64 *__restrict__ hash, long n, u32 world, u64 first_index, u64 *cursor /* [world]: next free place per owner */,
                                                                u64 *__restrict__ send_hash, u32 *__restrict__ send_index, u32 *__restrict__ slot) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n;
    const u64 h = in ? hash[i] : 0ull;
    const u32 own = in ? (u32)(h % world) : 0u;
    const int lane = (int)(threadIdx.x & 63);
    u64 todo = __ballot(in);
    while (todo) {
        const int lead = __ffsll(todo) - 1;
        const u32 o = (u32)__shfl((int)own, lead);
        const u64 same = __ballot(in && own == o);
        u64 base = 0;
        if (lane == lead) base = atomicAdd(&cursor[o], (u64)__popcll(same));
        const u32 blo = (u32)__shfl((int)(u32)base, lead), bhi = (u32)__shfl((int)(u32)(base >> 32), lead);
        if (in && own == o) {
            const u64 at = (((u64)bhi << 32) | blo) + (u64)__popcll(same & ((1ull << lane) - 1ull));
            send_hash[at] = h;
            send_index[at] = (u32)(first_index + (u64)i);
            slot[i] = (u32)at;
        }
        todo &= ~same;
    }
}
__global__ void __launch_bounds__(256) snk_flags_home_kernel(const uint8_t *__restrict__ back, const u32 *__restrict__ slot, long n, uint8_t *__restrict__ dup) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dup[i] = back[slot[i]];
}

}  // namespace

#define RM_OK(call)                                      \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) return (int)e_;            \
    } while (0)

// returns 0 or a hipError_t
int snk_launch_hash(const uint8_t *const seq[2], const uint16_t *const len[2], const int fixed_len[2], int pitch, long n,
                    int paired, unsigned long long *out, int n_cu, void *stream) {
    if (n <= 0) return 0;
    HashArgs A;
    for (int m = 0; m < 2; ++m) { A.seq[m] = seq[m]; A.len[m] = len[m]; A.fixed_len[m] = fixed_len[m]; }
    A.pitch = pitch; A.n = n; A.out = out;
    hipStream_t st = (hipStream_t)stream;
    const bool aligned = pitch % 16 == 0 && pitch <= 1024 && ((uintptr_t)seq[0] % 16 == 0) && (!paired || (uintptr_t)seq[1] % 16 == 0);
    const int p2 = pitch + 16;                                   // padded LDS pitch: consecutive rows start 4 banks apart
    const int W = 4;
    const size_t shmem = (size_t)W * 64 * p2;                    // one mate's rows at a time
    if (aligned && shmem <= 150 * 1024) {
        static bool attr[2] = {false, false};
        const void *k = paired ? (const void *)snk_hash_lds_kernel<true> : (const void *)snk_hash_lds_kernel<false>;
        if (!attr[paired ? 1 : 0]) {
            RM_OK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr[paired ? 1 : 0] = true;
        }
        const long tiles = (n + 63) / 64;
        long wgs = (tiles + W - 1) / W;
        const long cap = (long)n_cu * (long)max((size_t)1, (size_t)(150 * 1024) / shmem);   // all resident at once
        if (wgs > cap) wgs = cap;
        if (paired) hipLaunchKernelGGL(snk_hash_lds_kernel<true>, dim3((unsigned)wgs), dim3(W * 64), shmem, st, A, p2, tiles);
        else hipLaunchKernelGGL(snk_hash_lds_kernel<false>, dim3((unsigned)wgs), dim3(W * 64), shmem, st, A, p2, tiles);
    } else {
        long wgs = (n + 255) / 256;
        if (wgs > (long)n_cu * 8) wgs = (long)n_cu * 8;
        if (paired) hipLaunchKernelGGL(snk_hash_direct_kernel<true>, dim3((unsigned)wgs), dim3(256), 0, st, A);
        else hipLaunchKernelGGL(snk_hash_direct_kernel<false>, dim3((unsigned)wgs), dim3(256), 0, st, A);
    }
    return (int)hipGetLastError();
}

int snk_launch_bucket_count(const unsigned long long *hash, long n, unsigned prime, const unsigned *flag,
                            unsigned long long *count, void *stream) {
    if (n <= 0 || prime == 0) return 0;
    hipLaunchKernelGGL(snk_bucket_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, hash, n, prime,
                       flag, count);
    return (int)hipGetLastError();
}

int snk_launch_mark(const unsigned long long *hash, const unsigned *index, long n, unsigned prime, long bucket_total,
                    unsigned char *dup, void *stream) {
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int lg = 4;
    while ((1ull << lg) < 2ull * (u64)n) ++lg;
    const u64 cap = 1ull << lg, mask = cap - 1;
    u64 *keys = nullptr, *count = nullptr;
    u32 *minidx = nullptr, *flag = nullptr;
    RM_OK(hipMallocAsync((void **)&keys, cap * sizeof(u64) + 16, st));
    RM_OK(hipMallocAsync((void **)&minidx, cap * sizeof(u32), st));
    count = keys + cap;                                          // [count u64][flag u32] behind the keys
    flag = reinterpret_cast<u32 *>(count + 1);
    RM_OK(hipMemsetAsync(keys, 0xFF, cap * sizeof(u64), st));
    RM_OK(hipMemsetAsync(count, 0, 16, st));
    RM_OK(hipMemsetAsync(minidx, 0xFF, cap * sizeof(u32), st));
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(snk_mark_insert_kernel, dim3(grid), dim3(256), 0, st, hash, index, n, keys, minidx, mask, 64 - lg, flag);
    if (bucket_total < 0 && prime)
        hipLaunchKernelGGL(snk_bucket_count_kernel, dim3(grid), dim3(256), 0, st, hash, n, prime, (const u32 *)flag, count);
    hipLaunchKernelGGL(snk_mark_lookup_kernel, dim3(grid), dim3(256), 0, st, hash, index, n, (const u64 *)keys, (const u32 *)minidx, mask,
                       64 - lg, (const u64 *)count, bucket_total, dup);
    RM_OK(hipGetLastError());
    RM_OK(hipFreeAsync(keys, st));
    RM_OK(hipFreeAsync(minidx, st));
    return 0;
}


// ---------------------------------------------------------------- one pass (snk_rmdup_stream_*)
#include <vector>
#include "../../include/snk_rmdup.h"
void snk_set_error(const char *msg);

struct snk_rmdup_stream {
    int device = 0;
    u64 *keys = nullptr;
    u32 *minidx = nullptr;
    u64 cap = 0;
    int lg = 0;
    u64 count = 0;                                   // hashes inserted so far
    struct Chunk { u64 *d; long n; u32 base; };
    std::vector<Chunk> chunks;                       // all hashes stay resident: a grown table is refilled from them
    // the resident hashes live in slabs of 2^23 words (64 MB) carved up batch by batch: no hipMalloc on the per-batch path
    // (a fresh slab every ~30 batches of the CLI), nothing to free on an error path
    std::vector<u64 *> slabs;
    size_t slab_used = 0, slab_words = 0;            // of the newest slab
    u64 *d_marked = nullptr;                         // [marked u64][flag u32]
    hipEvent_t ev = nullptr;
    hipStream_t last = nullptr;
    bool have_ev = false;
    bool dead = false;                               // a failed growth left no table: every later call reports SNK_E_NOMEM
    // single-end runs (snk_rmdup_stream_mark_se_device): the batch's true flags, and the flag of the read in front of the batch
    uint8_t *d_true = nullptr;
    size_t true_cap = 0;
    uint8_t *d_carry = nullptr;                      // [2]: read by call k at [k & 1], written at [(k + 1) & 1]
    u64 se_calls = 0;
};

static u64 *stream_hash_room(snk_rmdup_stream *t, size_t n) {
    if (t->slabs.empty() || t->slab_used + n > t->slab_words) {
        const size_t words = n > ((size_t)1 << 23) ? n : ((size_t)1 << 23);
        u64 *d = nullptr;
        if (hipMalloc((void **)&d, words * sizeof(u64)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        t->slabs.push_back(d);
        t->slab_used = 0;
        t->slab_words = words;
    }
    u64 *r = t->slabs.back() + t->slab_used;
    t->slab_used += n;
    return r;
}

static int stream_alloc_table(snk_rmdup_stream *t, u64 want_pairs, hipStream_t st) {
    int lg = 16;
    while ((1ull << lg) < 4ull * want_pairs) ++lg;   // load <= 0.25 when fresh, grown at 0.5
    t->lg = lg;
    t->cap = 1ull << lg;
    if (hipMalloc((void **)&t->keys, t->cap * sizeof(u64)) != hipSuccess || hipMalloc((void **)&t->minidx, t->cap * sizeof(u32)) != hipSuccess) {
        (void)hipGetLastError();
        if (t->keys) (void)hipFree(t->keys);
        t->keys = nullptr; t->minidx = nullptr;
        return (int)hipErrorOutOfMemory;
    }
    RM_OK(hipMemsetAsync(t->keys, 0xFF, t->cap * sizeof(u64), st));
    RM_OK(hipMemsetAsync(t->minidx, 0xFF, t->cap * sizeof(u32), st));
    return 0;
}

extern "C" {

snk_rmdup_stream *snk_rmdup_stream_create(snk_ctx *, uint64_t expected_pairs) {
    snk_rmdup_stream *t = new snk_rmdup_stream();
    if (hipGetDevice(&t->device) != hipSuccess || stream_alloc_table(t, expected_pairs > 1024 ? expected_pairs : 1024, nullptr) != 0 ||
        hipMalloc((void **)&t->d_marked, 16) != hipSuccess || hipMemset(t->d_marked, 0, 16) != hipSuccess ||
        hipEventCreateWithFlags(&t->ev, hipEventDisableTiming) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        snk_set_error("snk_rmdup_stream_create: device allocation failed");
        snk_rmdup_stream_destroy(t);
        return nullptr;
    }
    return t;
}

uint64_t snk_rmdup_stream_bytes(uint64_t pairs) {
    int lg = 16;
    while ((1ull << lg) < 4ull * pairs) ++lg;        // stream_alloc_table()
    return pairs * 8ull + (1ull << lg) * 12ull;
}

void snk_rmdup_stream_destroy(snk_rmdup_stream *t) {
    if (!t) return;
    (void)hipDeviceSynchronize();
    if (t->keys) (void)hipFree(t->keys);
    if (t->minidx) (void)hipFree(t->minidx);
    if (t->d_marked) (void)hipFree(t->d_marked);
    if (t->d_true) (void)hipFree(t->d_true);
    if (t->d_carry) (void)hipFree(t->d_carry);
    for (auto d : t->slabs) (void)hipFree(d);
    if (t->ev) (void)hipEventDestroy(t->ev);
    delete t;
}

// the marking itself; the event that orders the next call is recorded by the callers (behind whatever they add)
static int stream_mark(snk_rmdup_stream *t, const uint64_t *d_hash, uint64_t first_index, int64_t n, uint8_t *d_dup, void *stream) {
    if (!t || !d_hash || !d_dup || n < 0) { snk_set_error("snk_rmdup_stream_mark_device: bad argument"); return SNK_E_PARAM; }
    if (first_index + (uint64_t)n > 4294967295ull) {           // src/peprocess.cpp:3094
        snk_set_error("snk_rmdup_stream_mark_device: reads number is too large to do remove duplication (limit 2^32-1)");
        return SNK_E_PARAM;
    }
    if (n == 0) return SNK_OK;
    hipStream_t st = (hipStream_t)stream;
    auto fail = [](const char *what) { snk_set_error(what); return SNK_E_HIP; };
    if (t->have_ev && st != t->last && hipStreamWaitEvent(st, t->ev, 0) != hipSuccess) return fail("snk_rmdup_stream_mark_device: hipStreamWaitEvent failed");
    if (t->dead) { snk_set_error("snk_rmdup_stream_mark_device: out of device memory (table)"); return SNK_E_NOMEM; }
    // the batch's hashes stay resident
    snk_rmdup_stream::Chunk c{stream_hash_room(t, (size_t)n), (long)n, (u32)first_index};
    if (!c.d) { snk_set_error("snk_rmdup_stream_mark_device: out of device memory (resident hashes)"); return SNK_E_NOMEM; }
    if (hipMemcpyAsync(c.d, d_hash, (size_t)n * sizeof(u64), hipMemcpyDeviceToDevice, st) != hipSuccess) return fail("snk_rmdup_stream_mark_device: copy failed");
    u32 *flag = reinterpret_cast<u32 *>(t->d_marked + 1);
    if ((t->count + (u64)n) * 2 > t->cap) {                    // grow: a fresh table, refilled from the resident hashes
        if (hipStreamSynchronize(st) != hipSuccess) return fail("snk_rmdup_stream_mark_device: synchronise failed");
        (void)hipFree(t->keys);
        (void)hipFree(t->minidx);
        t->keys = nullptr; t->minidx = nullptr;
        if (stream_alloc_table(t, 2 * (t->count + (u64)n), st) != 0) {     // (the old table is gone: the caller falls back to two passes)
            t->dead = true;
            snk_set_error("snk_rmdup_stream_mark_device: out of device memory (table)");
            return SNK_E_NOMEM;
        }
        for (const auto &o : t->chunks)
            hipLaunchKernelGGL(snk_stream_insert_kernel, dim3((unsigned)((o.n + 255) / 256)), dim3(256), 0, st, (const u64 *)o.d, o.base, o.n, t->keys,
                               t->minidx, t->cap - 1, 64 - t->lg, flag);
    }
    t->chunks.push_back(c);
    t->count += (u64)n;
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(snk_stream_insert_kernel, dim3(grid), dim3(256), 0, st, (const u64 *)c.d, c.base, (long)n, t->keys, t->minidx, t->cap - 1, 64 - t->lg, flag);
    hipLaunchKernelGGL(snk_stream_lookup_kernel, dim3(grid), dim3(256), 0, st, (const u64 *)c.d, c.base, (long)n, (const u64 *)t->keys, (const u32 *)t->minidx,
                       t->cap - 1, 64 - t->lg, d_dup, t->d_marked, flag);
    if (hipGetLastError() != hipSuccess) return fail("snk_rmdup_stream_mark_device: launch failed");
    return SNK_OK;
}
static int stream_mark_done(snk_rmdup_stream *t, hipStream_t st) {
    if (hipEventRecord(t->ev, st) != hipSuccess) { snk_set_error("snk_rmdup_stream_mark_device: hipEventRecord failed"); return SNK_E_HIP; }
    t->have_ev = true;
    t->last = st;
    return SNK_OK;
}

int snk_rmdup_stream_mark_device(snk_rmdup_stream *t, const uint64_t *d_hash, uint64_t first_index, int64_t n, uint8_t *d_dup, void *stream) {
    const int rc = stream_mark(t, d_hash, first_index, n, d_dup, stream);
    if (rc != SNK_OK || n == 0) return rc;
    return stream_mark_done(t, (hipStream_t)stream);
}

int snk_rmdup_stream_mark_se_device(snk_rmdup_stream *t, const uint64_t *d_hash, uint64_t first_index, int64_t n, int64_t n_shifted, uint8_t *d_dup, void *stream) {
    if (!t || n < 0 || n_shifted < 0 || n_shifted > n) { snk_set_error("snk_rmdup_stream_mark_se_device: bad argument"); return SNK_E_PARAM; }
    if (n == 0) return SNK_OK;
    hipStream_t st = (hipStream_t)stream;
    if (!t->d_carry) {
        if (hipMalloc((void **)&t->d_carry, 16) != hipSuccess) { (void)hipGetLastError(); snk_set_error("snk_rmdup_stream_mark_se_device: out of device memory"); return SNK_E_NOMEM; }
        if (hipMemset(t->d_carry, 0, 16) != hipSuccess) { snk_set_error("snk_rmdup_stream_mark_se_device: memset failed"); return SNK_E_HIP; }     // (read 0 of the file: no duplicate in front of it)
    }
    if ((size_t)n > t->true_cap) {
        // (the launches that read the old buffer are ordered in front of everything this call issues only on their own stream)
        if (t->have_ev && hipEventSynchronize(t->ev) != hipSuccess) { snk_set_error("snk_rmdup_stream_mark_se_device: synchronise failed"); return SNK_E_HIP; }
        if (t->d_true) (void)hipFree(t->d_true);
        t->d_true = nullptr;
        t->true_cap = 0;
        const size_t cap = ((size_t)n + 65535) & ~(size_t)65535;
        if (hipMalloc((void **)&t->d_true, cap) != hipSuccess) { (void)hipGetLastError(); snk_set_error("snk_rmdup_stream_mark_se_device: out of device memory"); return SNK_E_NOMEM; }
        t->true_cap = cap;
    }
    const int rc = stream_mark(t, d_hash, first_index, n, t->d_true, stream);
    if (rc != SNK_OK) return rc;
    const int par = (int)(t->se_calls++ & 1);
    hipLaunchKernelGGL(snk_se_shift_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint8_t *)t->d_true, (long)n, (long)n_shifted,
                       (const uint8_t *)(t->d_carry + par), t->d_carry + (par ^ 1), d_dup);
    if (hipGetLastError() != hipSuccess) { snk_set_error("snk_rmdup_stream_mark_se_device: launch failed"); return SNK_E_HIP; }
    return stream_mark_done(t, st);
}

int snk_rmdup_stream_stats(snk_rmdup_stream *t, uint64_t *n_marked, int32_t *sentinel_seen) {
    if (!t) return SNK_E_PARAM;
    if (t->have_ev && hipEventSynchronize(t->ev) != hipSuccess) { snk_set_error("snk_rmdup_stream_stats: synchronise failed"); return SNK_E_HIP; }
    uint64_t h[2] = {0, 0};
    if (hipMemcpy(h, t->d_marked, 16, hipMemcpyDeviceToHost) != hipSuccess) { snk_set_error("snk_rmdup_stream_stats: copy failed"); return SNK_E_HIP; }
    if (n_marked) *n_marked = h[0];
    if (sentinel_seen) *sentinel_seen = (int32_t)(h[1] & 1u);
    if (h[1] & 2u) { snk_set_error("snk_rmdup_stream_stats: a hash was not found in the table (internal error)"); return SNK_E_HIP; }
    return SNK_OK;
}

// ---- multi-GPU exchange helpers (include/snk_rmdup.h)
int snk_rmdup_partition_device(snk_ctx *, const uint64_t *d_hash, int64_t n, uint64_t first_index, int32_t world, uint64_t *d_send_hash, uint32_t *d_send_index,
                               uint32_t *d_slot, uint64_t *h_counts, void *stream) {
    if (!d_hash || !d_send_hash || !d_send_index || !d_slot || !h_counts || n < 0 || world < 1 || world > 4096 || first_index + (uint64_t)n > 4294967296ull) {
        snk_set_error("snk_rmdup_partition_device: bad argument");
        return SNK_E_PARAM;
    }
    hipStream_t st = (hipStream_t)stream;
    for (int g = 0; g < world; ++g) h_counts[g] = 0;
    if (n == 0) return SNK_OK;
    u64 *d_cnt = nullptr;
    auto fail = [&](const char *what) { if (d_cnt) (void)hipFree(d_cnt); snk_set_error(what); return SNK_E_HIP; };
    if (hipMalloc((void **)&d_cnt, (size_t)world * 2 * sizeof(u64)) != hipSuccess) { (void)hipGetLastError(); snk_set_error("snk_rmdup_partition_device: out of device memory"); return SNK_E_NOMEM; }
    if (hipMemsetAsync(d_cnt, 0, (size_t)world * 2 * sizeof(u64), st) != hipSuccess) return fail("snk_rmdup_partition_device: memset failed");
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(snk_owner_count_kernel, dim3(grid), dim3(256), 0, st, (const u64 *)d_hash, (long)n, (u32)world, d_cnt);
    if (hipMemcpyAsync(h_counts, d_cnt, (size_t)world * sizeof(u64), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail("snk_rmdup_partition_device: count pass failed");
    std::vector<u64> base((size_t)world, 0);
    for (int g = 1; g < world; ++g) base[(size_t)g] = base[(size_t)g - 1] + h_counts[g - 1];
    if (hipMemcpyAsync(d_cnt + world, base.data(), (size_t)world * sizeof(u64), hipMemcpyHostToDevice, st) != hipSuccess) return fail("snk_rmdup_partition_device: copy failed");
    hipLaunchKernelGGL(snk_owner_scatter_kernel, dim3(grid), dim3(256), 0, st, (const u64 *)d_hash, (long)n, (u32)world, (u64)first_index, d_cnt + world,
                       (u64 *)d_send_hash, d_send_index, d_slot);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail("snk_rmdup_partition_device: scatter pass failed");
    (void)hipFree(d_cnt);
    return SNK_OK;
}

int snk_rmdup_flags_home_device(snk_ctx *, const uint8_t *d_back, const uint32_t *d_slot, int64_t n, uint8_t *d_dup, void *stream) {
    if (!d_back || !d_slot || !d_dup || n < 0) { snk_set_error("snk_rmdup_flags_home_device: bad argument"); return SNK_E_PARAM; }
    if (n == 0) return SNK_OK;
    hipLaunchKernelGGL(snk_flags_home_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_back, d_slot, (long)n, d_dup);
    if (hipGetLastError() != hipSuccess) { snk_set_error("snk_rmdup_flags_home_device: launch failed"); return SNK_E_HIP; }
    return SNK_OK;
}

}  // extern "C"
