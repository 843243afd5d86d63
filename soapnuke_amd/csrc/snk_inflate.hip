// snk_inflate.hip -- the device inflate (include/snk_gunzip.h): block-start search, marker-mode chunk decoding, window chain and
// marker resolution on gfx950.  The decoding is csrc/snk_inflate_core.hip.h (one thread per chunk: DEFLATE is a sequential bit
// stream; the parallelism is the thousands of chunks of a window -- one wavefront each, its Huffman tables in LDS, 11 per CU).
// Replaces the reference's gzgets() reading loop, src/peprocess.cpp:2063-2113.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "snk_inflate_core.hip.h"
#include "../../include/snk_gunzip.h"
#include "../../include/snk_filter.h"

void snk_set_error(const char *msg);

using namespace snkinf;
static_assert(sizeof(Chunk) == sizeof(snk_gunzip_chunk) && sizeof(Chunk) == 80, "snk_gunzip_chunk mirrors snkinf::Chunk");
static_assert(sizeof(MemberEnd) == sizeof(snk_gunzip_member), "snk_gunzip_member mirrors snkinf::MemberEnd");

namespace {

struct WaveSpace { Tables T; Scratch S; u32 cl_tab[128]; };

// One wavefront per chunk: lanes probe 64 consecutive bit offsets at a time with the register-only screen; the few offsets that
// pass are checked in full (tables in LDS) by lane 0, lowest offset first.  start[c] = the first block start in the chunk, or ~0.
__global__ void __launch_bounds__(64) inf_search_kernel(const u8 *comp, u64 nbytes, u32 chunk_bytes, u64 first_bit, u64 *start) {
    __shared__ WaveSpace W;
    const u32 c = blockIdx.x + 1;                           // (chunk 0 starts at first_bit)
    const int lane = threadIdx.x;
    u64 lo = (u64)c * chunk_bytes * 8, hi = min((u64)(c + 1) * chunk_bytes * 8, nbytes * 8);
    if (lo <= first_bit) lo = first_bit + 1;
    u64 found = ~0ull;
    for (u64 base = lo; base < hi && found == ~0ull; base += 64) {
        const u64 bit = base + (u64)lane;
        u64 cand = __ballot(bit < hi && probe_quick(comp, nbytes, bit));
        while (cand && found == ~0ull) {
            const int l = __ffsll((long long)cand) - 1;
            cand &= cand - 1;
            int ok = 0;
            if (lane == 0) ok = probe_header(comp, nbytes, base + (u64)l, W.T, W.S, W.cl_tab) ? 1 : 0;
            ok = __shfl(ok, 0, 64);
            if (ok) found = base + (u64)l;
        }
    }
    if (lane == 0) start[c] = found;
}

__global__ void __launch_bounds__(64) inf_decode_kernel(const u8 *comp, u64 nbytes, Chunk *chunks, u16 *syms, MemberEnd *ends) {
    __shared__ WaveSpace W;
    if (threadIdx.x == 0) {
        Chunk ck = chunks[blockIdx.x];
        decode_chunk(comp, nbytes, ck, syms, ends, W.T, W.S, W.cl_tab, nullptr, false);
        chunks[blockIdx.x] = ck;
    }
}

// The same with the whole wavefront at work: all 64 lanes run the decoder in lockstep on the same values (its tables, header
// workspace and decisions are uniform), and share the I/O -- compressed bytes through an LDS ring refilled 1 KB at a time by all
// lanes, matches queued and copied 64 at a time, one per lane (snk_inflate_core.hip.h, Coop).  A lane on its own pays a global-memory
// round trip per bit-buffer refill and per copied symbol: gzip turns the base lines of FASTQ into ~40 short matches per read.
__global__ void __launch_bounds__(64) inf_decode_coop_kernel(const u8 *comp, u64 nbytes, Chunk *chunks, u16 *syms, MemberEnd *ends) {
    SNK_WAVE_UNIFORM_SHARED WaveSpace W;
    __shared__ __attribute__((aligned(16))) u8 ring[2 * HALF];
    __shared__ u32 qdst[QCAP], qinfo[QCAP];
    __shared__ u16 hist[HS];
    Coop co;
    co.ring = ring; co.hist = hist; co.qdst = qdst; co.qinfo = qinfo; co.ring_lo = co.ring_end = 0; co.qn = 0; co.q_first = 0;
    Chunk ck = chunks[blockIdx.x];
    decode_chunk(comp, nbytes, ck, syms, ends, W.T, W.S, W.cl_tab, &co, true);
    if (threadIdx.x == 0) chunks[blockIdx.x] = ck;
}

// The windows in front of the chunks of a chain: wins[0] is given; wins[j + 1] = the last 32 KiB of (wins[j] ++ text of chunk j).
// One workgroup walks the chain (a chunk's window needs the one before it); 32 positions per thread and step.
__global__ void __launch_bounds__(1024) inf_chain_kernel(const Chunk *chunks, const u32 *order, u32 k, const u16 *syms, u8 *wins) {
    for (u32 j = 0; j < k; ++j) {
        const Chunk &ck = chunks[order[j]];
        const u8 *w = wins + (size_t)j * WIN;
        u8 *nw = wins + (size_t)(j + 1) * WIN;
        const u32 n = ck.n_syms;
        const u16 *s = syms + ck.out_off;
        for (u32 i = threadIdx.x; i < (u32)WIN; i += blockDim.x) nw[i] = chain_byte(n, s, w, i);
        __threadfence_block();
        __syncthreads();
    }
}

// text of chunk order[j] at text_off[j]: its symbols with the markers replaced from its window; four symbols per thread
__global__ void __launch_bounds__(256) inf_resolve_kernel(const Chunk *chunks, const u32 *order, const u64 *text_off, const u16 *syms, const u8 *wins,
                                                          u8 *text, u32 blocks_per_chunk) {
    const u32 j = blockIdx.x / blocks_per_chunk, part = blockIdx.x - j * blocks_per_chunk;
    const Chunk &ck = chunks[order[j]];
    const u8 *w = wins + (size_t)j * WIN;
    const u16 *s = syms + ck.out_off;
    u8 *o = text + text_off[j];
    const u32 n = ck.n_syms;
    for (u32 i = part * 256 + threadIdx.x; i < n; i += blocks_per_chunk * 256) {
        o[i] = resolve_sym(s[i], w);
    }
}

}  // namespace

struct snk_gunzip {
    int device = 0;
    uint64_t max_window = 0;
    uint32_t chunk_bytes = 0, spc = 0, epc = 0, max_chunks = 0;
    hipStream_t stream = nullptr;
    u8 *d_comp = nullptr;
    u64 *d_start = nullptr;
    Chunk *d_chunks = nullptr;
    MemberEnd *d_ends = nullptr;
    u16 *d_syms = nullptr;
    u8 *d_wins = nullptr, *d_text = nullptr;
    u32 *d_order = nullptr;
    u64 *d_toff = nullptr;
    uint64_t text_cap = 0, nbytes = 0;
    uint32_t nchunks = 0;
    std::vector<Chunk> h_chunks;
};

#define GZ_OK(call)                                                                         \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            snk_set_error((std::string("snk_gunzip: ") + #call + ": " + hipGetErrorString(e_)).c_str()); \
            return SNK_E_HIP;                                                               \
        }                                                                                   \
    } while (0)

extern "C" {

void snk_gunzip_destroy(snk_gunzip *g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    void *p[] = {g->d_comp, g->d_start, g->d_chunks, g->d_ends, g->d_syms, g->d_wins, g->d_text, g->d_order, g->d_toff};
    for (void *x : p) if (x) (void)hipFree(x);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

snk_gunzip *snk_gunzip_create(int device, uint64_t max_window_bytes, uint32_t chunk_bytes, uint32_t syms_per_chunk, uint32_t ends_per_chunk) {
    if (chunk_bytes < 4096 || max_window_bytes < chunk_bytes || syms_per_chunk < 1024 || ends_per_chunk < 1) {
        snk_set_error("snk_gunzip_create: bad sizes");
        return nullptr;
    }
    snk_gunzip *g = new snk_gunzip();
    g->device = device;
    g->max_window = max_window_bytes;
    g->chunk_bytes = chunk_bytes;
    g->spc = syms_per_chunk;
    g->epc = ends_per_chunk;
    g->max_chunks = (uint32_t)((max_window_bytes + chunk_bytes - 1) / chunk_bytes);
    const size_t nc = g->max_chunks;
    g->text_cap = (uint64_t)nc * syms_per_chunk;
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void **)&g->d_comp, max_window_bytes + PAD) == hipSuccess && hipMalloc((void **)&g->d_start, nc * sizeof(u64)) == hipSuccess &&
              hipMalloc((void **)&g->d_chunks, nc * sizeof(Chunk)) == hipSuccess && hipMalloc((void **)&g->d_ends, nc * ends_per_chunk * sizeof(MemberEnd)) == hipSuccess &&
              hipMalloc((void **)&g->d_syms, nc * (size_t)syms_per_chunk * sizeof(u16)) == hipSuccess &&
              hipMalloc((void **)&g->d_wins, (nc + 1) * (size_t)WIN) == hipSuccess && hipMalloc((void **)&g->d_text, g->text_cap) == hipSuccess &&
              hipMalloc((void **)&g->d_order, nc * sizeof(u32)) == hipSuccess && hipMalloc((void **)&g->d_toff, nc * sizeof(u64)) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        snk_set_error("snk_gunzip_create: device allocation failed");
        snk_gunzip_destroy(g);
        return nullptr;
    }
    return g;
}

int snk_gunzip_decode(snk_gunzip *g, const uint8_t *h_comp, uint64_t nbytes, uint64_t first_bit, int first_of_member,
                      snk_gunzip_chunk *h_chunks, snk_gunzip_member *h_ends) {
    if (!g || !h_comp || !h_chunks || !h_ends || nbytes == 0 || nbytes > g->max_window || first_bit >= nbytes * 8) {
        snk_set_error("snk_gunzip_decode: bad argument");
        return SNK_E_PARAM;
    }
    GZ_OK(hipSetDevice(g->device));
    const uint32_t nc = (uint32_t)((nbytes + g->chunk_bytes - 1) / g->chunk_bytes);
    g->nchunks = nc;
    g->nbytes = nbytes;
    GZ_OK(hipMemcpyAsync(g->d_comp, h_comp, nbytes, hipMemcpyHostToDevice, g->stream));
    GZ_OK(hipMemsetAsync(g->d_comp + nbytes, 0, PAD, g->stream));
    std::vector<u64> start(nc, ~0ull);
    if (nc > 1) {
        hipLaunchKernelGGL(inf_search_kernel, dim3(nc - 1), dim3(64), 0, g->stream, (const u8 *)g->d_comp, (u64)nbytes, g->chunk_bytes, (u64)first_bit, g->d_start);
        GZ_OK(hipGetLastError());
        GZ_OK(hipMemcpyAsync(start.data(), g->d_start, nc * sizeof(u64), hipMemcpyDeviceToHost, g->stream));
        GZ_OK(hipStreamSynchronize(g->stream));
    }
    start[0] = first_bit;
    // a chunk stops in front of the next chunk that has a start
    g->h_chunks.assign(nc, Chunk());
    u64 next = nbytes * 8 + 64;
    for (uint32_t c = nc; c-- > 0;) {
        Chunk &ck = g->h_chunks[c];
        memset(&ck, 0, sizeof ck);
        ck.start_bit = start[c];
        ck.stop_bit = next;
        ck.out_off = (u64)c * g->spc;
        ck.out_cap = g->spc;
        ck.first_of_member = (c == 0 && first_of_member) ? 1u : 0u;
        ck.ends_off = c * g->epc;
        ck.ends_cap = g->epc;
        if (start[c] != ~0ull) next = start[c];
    }
    GZ_OK(hipMemcpyAsync(g->d_chunks, g->h_chunks.data(), nc * sizeof(Chunk), hipMemcpyHostToDevice, g->stream));
    static const bool coop = !(getenv("SNK_DGZ_COOP") && atoi(getenv("SNK_DGZ_COOP")) == 0);      // (SNK_DGZ_COOP=0: one lane per chunk does everything)
    if (coop) hipLaunchKernelGGL(inf_decode_coop_kernel, dim3(nc), dim3(64), 0, g->stream, (const u8 *)g->d_comp, (u64)nbytes, g->d_chunks, g->d_syms, g->d_ends);
    else hipLaunchKernelGGL(inf_decode_kernel, dim3(nc), dim3(64), 0, g->stream, (const u8 *)g->d_comp, (u64)nbytes, g->d_chunks, g->d_syms, g->d_ends);
    GZ_OK(hipGetLastError());
    GZ_OK(hipMemcpyAsync(g->h_chunks.data(), g->d_chunks, nc * sizeof(Chunk), hipMemcpyDeviceToHost, g->stream));
    GZ_OK(hipMemcpyAsync(h_ends, g->d_ends, (size_t)nc * g->epc * sizeof(MemberEnd), hipMemcpyDeviceToHost, g->stream));
    GZ_OK(hipStreamSynchronize(g->stream));
    memcpy(h_chunks, g->h_chunks.data(), nc * sizeof(Chunk));
    return SNK_OK;
}

int snk_gunzip_resolve(snk_gunzip *g, const uint32_t *order, uint32_t k, const uint8_t *h_window_in, uint8_t *h_text, uint64_t text_bytes,
                       uint8_t *h_window_out) {
    if (!g || !order || k == 0 || k > g->nchunks || !h_text || !h_window_out) { snk_set_error("snk_gunzip_resolve: bad argument"); return SNK_E_PARAM; }
    GZ_OK(hipSetDevice(g->device));
    std::vector<u64> toff(k);
    u64 at = 0;
    for (uint32_t j = 0; j < k; ++j) {
        if (order[j] >= g->nchunks) { snk_set_error("snk_gunzip_resolve: chunk index out of range"); return SNK_E_PARAM; }
        toff[j] = at;
        at += g->h_chunks[order[j]].n_syms;
    }
    if (at != text_bytes || at > g->text_cap) { snk_set_error("snk_gunzip_resolve: text_bytes does not match the chunks"); return SNK_E_PARAM; }
    if (h_window_in) GZ_OK(hipMemcpyAsync(g->d_wins, h_window_in, WIN, hipMemcpyHostToDevice, g->stream));
    else GZ_OK(hipMemsetAsync(g->d_wins, 0, WIN, g->stream));
    GZ_OK(hipMemcpyAsync(g->d_order, order, k * sizeof(u32), hipMemcpyHostToDevice, g->stream));
    GZ_OK(hipMemcpyAsync(g->d_toff, toff.data(), k * sizeof(u64), hipMemcpyHostToDevice, g->stream));
    hipLaunchKernelGGL(inf_chain_kernel, dim3(1), dim3(1024), 0, g->stream, (const Chunk *)g->d_chunks, (const u32 *)g->d_order, k, (const u16 *)g->d_syms, g->d_wins);
    const u32 bpc = 16;
    hipLaunchKernelGGL(inf_resolve_kernel, dim3(k * bpc), dim3(256), 0, g->stream, (const Chunk *)g->d_chunks, (const u32 *)g->d_order, (const u64 *)g->d_toff,
                       (const u16 *)g->d_syms, (const u8 *)g->d_wins, g->d_text, bpc);
    GZ_OK(hipGetLastError());
    if (text_bytes) GZ_OK(hipMemcpyAsync(h_text, g->d_text, text_bytes, hipMemcpyDeviceToHost, g->stream));
    GZ_OK(hipMemcpyAsync(h_window_out, g->d_wins + (size_t)k * WIN, WIN, hipMemcpyDeviceToHost, g->stream));
    GZ_OK(hipStreamSynchronize(g->stream));
    return SNK_OK;
}

}  // extern "C"
