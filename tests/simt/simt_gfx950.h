// SIMT emulator: C++ bodies of the hand-placed gfx950 instructions of soapnuke_amd/csrc/snk_gfx950.hip.h (same names, same
// meaning).  LDS addresses are byte offsets into the dynamic shared memory of the kernel (HIP_DYNAMIC_SHARED, address 0 = its first byte).
// Asynchrony is not modelled: a read or DMA completes where it is issued, waits that order lanes against each other
// (a DMA chunk written by 64 lanes and read by others) are wave-level synchronisation points.  Test infrastructure only.
#pragma once
#include <hip/hip_runtime.h>

namespace snk {
namespace {
typedef uint32_t g9_u32;
static inline uint8_t *simt_lds_at(g9_u32 addr) {
    if (addr >= 160u * 1024u) { fprintf(stderr, "simt: LDS address %u out of range (thread %u)\n", addr, threadIdx.x); abort(); }
    return simt_dyn_shared() + addr;
}
#define SNK_LDS_ADDR(p) ((uint32_t)((const uint8_t *)(p) - (const uint8_t *)simt_dyn_shared()))
#define SNK_OPAQUE_S(x) ((void)(x))
#define SNK_KERNARG_PTR(T, A) ((const T *)(uintptr_t)&(A))      // (the emulator hands the argument struct to every thread by value)
#define SNK_FRESH_ARGS(p) ((void)(p))

static inline uint32_t mask_nonzero(unsigned long long m) { return m != 0ull ? 1u : 0u; }
static inline int wl(int dst, int val, int lane) { return simt_writelane(val, lane, dst); }
static inline int rl(int v, int lane) { return simt_readlane(v, lane); }

template <int OFF> static inline void lds_add_u32(g9_u32 addr, g9_u32 val) { __atomic_fetch_add((uint32_t *)simt_lds_at(addr + (g9_u32)OFF), val, __ATOMIC_RELAXED); }
template <int S, int E, int NS> static inline void lds_read_qstrips(g9_u32 (&q)[NS], g9_u32 addrq) {
    if constexpr (S < E) {
        q[S] = *simt_lds_at(addrq + 64 * S);
        lds_read_qstrips<S + 1, E>(q, addrq);
    }
}
template <int S, int E, int BASE, int NS> static inline void lds_read_qstrips_at(g9_u32 (&q)[NS], g9_u32 addrq) {
    if constexpr (S < E) {
        q[S] = *simt_lds_at(addrq + BASE + 64 * S);
        lds_read_qstrips_at<S + 1, E, BASE>(q, addrq);
    }
}
static inline g9_u32 simt_lds_b32(g9_u32 addr) { g9_u32 v; memcpy(&v, simt_lds_at(addr), 4); return v; }
template <int OFF> static inline void lds_read_b32_at(g9_u32 &d, g9_u32 addr) { d = simt_lds_b32(addr + (g9_u32)OFF); }
static inline void lds_read_b32(g9_u32 &d, g9_u32 addr) { d = simt_lds_b32(addr); }
static inline void lds_read_u8(g9_u32 &d, g9_u32 addr) { d = *simt_lds_at(addr); }
template <int N, int NS> static inline void lds_wait(g9_u32 &, g9_u32 &, g9_u32 (&)[NS]) {}
static inline void lds_wait_all() { SNK_WAVE_SYNC(); }
template <int N> static inline void vmem_wait() { SNK_WAVE_SYNC(); }

static inline g9_u32 clamp_row_addr(g9_u32 q, g9_u32 qlo, g9_u32 qhi, int lgb, g9_u32 base) {
    const g9_u32 lo = min(qlo, qhi), hi = max(qlo, qhi);             // v_med3_u32: the median of the three
    const g9_u32 c = q < lo ? lo : q > hi ? hi : q;
    return (c << lgb) + base;
}
template <int OFF> static inline g9_u32 clamp_row_addr_add(g9_u32 q, g9_u32 qlo, g9_u32 qhi, int lgb, g9_u32 base, g9_u32 val) {
    const g9_u32 a = clamp_row_addr(q, qlo, qhi, lgb, base);
    lds_add_u32<OFF>(a, val);
    return a;
}
static inline void dma_to_lds16(const uint8_t *g, uint8_t *dst) { memcpy(dst + 16 * (simt::cur->lane), g, 16); }
}  // namespace
}  // namespace snk
