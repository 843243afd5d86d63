// snk_gfx950.hip.h -- the instructions the tiled kernel places by hand (snk_tiled.hip, phase 1 / phase 3): LDS reads and adds with
// immediate offsets and counted waits, the fused clamp + row-address + histogram-add statement, the LDS DMA, v_writelane.
// They are inline asm on purpose (see the notes at each wrapper); everything here is gfx950 ISA.
//
// The CPU test tier (tests/simt, an emulator that runs the HIP sources on fibers) brings its own bodies with the same names and
// meaning: that is the one `#ifdef` of the device code, and nothing of the product is built with it.
#pragma once
#ifdef SNK_SIMT_EMUL
#include <simt_gfx950.h>
#else
#include <stdint.h>

extern "C" __device__ int __snk_writelane(int, int, int) __asm("llvm.amdgcn.writelane.i32");

namespace snk {
namespace {
typedef uint32_t g9_u32;
typedef __attribute__((address_space(3))) g9_u32 *lds_u32_ptr;
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;

// absolute LDS byte address of a pointer into __shared__ memory
#define SNK_LDS_ADDR(p) ((uint32_t)(uintptr_t)(snk::lds_u32_ptr)(p))
// the value lives in a scalar register from here on (the compiler may not fold what it knows about it into the code behind)
#define SNK_OPAQUE_S(x) asm volatile("" : "+s"(x))
// The kernel's own argument block (its single by-value struct argument A sits at offset 0 of the kernarg segment) as a pointer
// into the constant address space; SNK_FRESH_ARGS: the same pointer as a value the compiler cannot see through -- what is
// loaded and derived through it is loaded and derived THERE, not hoisted to the kernel's entry and carried (spilled) across loops
#define SNK_KERNARG_PTR(T, A) ((const T *)__builtin_amdgcn_kernarg_segment_ptr())
#define SNK_FRESH_ARGS(p) asm volatile("" : "+s"(p))

// clang exposes readlane but not writelane as a builtin; the LLVM intrinsic is bound above
// (v_writelane_b32: uniform value -> one lane of a VGPR; per-read scalars and the rare fix-up pass use it).
__device__ __forceinline__ int wl(int dst, int val, int lane) { return __snk_writelane(val, lane, dst); }
__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// 1 when any lane of a wave mask is set, else 0 -- on the scalar unit (s_cmp_lg_u64 + s_cselect_b32).  Written out because the
// compiler turns every C spelling of it (`m != 0`, min(popcount(m), 1), (m | -m) >> 63) into a boolean that it then moves through a
// vector select and a v_readfirstlane: two VALU instructions per read in phase 1's loop for a value that never leaves the scalar side.
__device__ __forceinline__ g9_u32 mask_nonzero(unsigned long long m) {
    g9_u32 f;
    asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(f) : "s"(m) : "scc");
    return f;
}

// Fire-and-forget LDS add with a compile-time offset.  Issued as inline asm on purpose: with a
// global_load_lds (LDS DMA) in flight hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS
// store/atomic it knows about, which would drain the prefetched chunk at the first histogram add
// of every chunk.  The histogram words never overlap the staging buffers; the flush waits
// lgkmcnt(0) explicitly before its barrier.  `addr` is an absolute LDS byte address.
template <int OFF>
__device__ __forceinline__ void lds_add_u32(g9_u32 addr, g9_u32 val) {
    asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(OFF));
}
// asynchronous LDS reads of one read's row: its bases and qualities as dwords (lane l: positions 4l..4l+3, for the
// bit collectors) and its qualities once more as one byte per lane and 64-position strip (lane = position, for the
// per-position histogram); pair with lds_wait
template <int S, int E, int NS>                 // strips S .. E-1
__device__ __forceinline__ void lds_read_qstrips(g9_u32 (&q)[NS], g9_u32 addrq) {
    if constexpr (S < E) {
        asm volatile("ds_read_u8 %0, %1 offset:%2" : "=&v"(q[S]) : "v"(addrq), "n"(64 * S));
        lds_read_qstrips<S + 1, E>(q, addrq);
    }
}
template <int S, int E, int BASE, int NS>              // strips S .. E-1 at immediate offsets BASE + 64 s
__device__ __forceinline__ void lds_read_qstrips_at(g9_u32 (&q)[NS], g9_u32 addrq) {
    if constexpr (S < E) {
        asm volatile("ds_read_u8 %0, %1 offset:%2" : "=&v"(q[S]) : "v"(addrq), "n"(BASE + 64 * S));
        lds_read_qstrips_at<S + 1, E, BASE>(q, addrq);
    }
}
template <int OFF>
__device__ __forceinline__ void lds_read_b32_at(g9_u32 &d, g9_u32 addr) { asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(OFF)); }
__device__ __forceinline__ void lds_read_b32(g9_u32 &d, g9_u32 addr) { asm volatile("ds_read_b32 %0, %1" : "=&v"(d) : "v"(addr)); }
__device__ __forceinline__ void lds_read_u8(g9_u32 &d, g9_u32 addr) { asm volatile("ds_read_u8 %0, %1" : "=&v"(d) : "v"(addr)); }
// wait until at most N LDS ops are outstanding; the registers of the (asm) reads being waited for
// are tied to the wait so that no use can be scheduled above it
template <int N, int NS>
__device__ __forceinline__ void lds_wait(g9_u32 &c4, g9_u32 &q4, g9_u32 (&q)[NS]) {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N));
    asm volatile("" : "+v"(c4), "+v"(q4));
#pragma unroll
    for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(q[s]));
}
// every LDS operation of the wave has completed (the asm adds and reads above are invisible to the compiler's own counting)
__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// at most N vector-memory operations (here: LDS DMA chunks) of the wave are outstanding
template <int N>
__device__ __forceinline__ void vmem_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// clamp(q, qlo, qhi) << lgb + base in ONE asm statement: the compiler pads every asm result that the next VALU instruction
// reads with an s_nop (it has to assume a dst_sel forwarding hazard), three per read (v_med3 takes one scalar operand only:
// qlo comes in a VGPR)
__device__ __forceinline__ g9_u32 clamp_row_addr(g9_u32 q, g9_u32 qlo_v, g9_u32 qhi, int lgb, g9_u32 base) {
    g9_u32 a;
    asm("v_med3_u32 %0, %1, %2, %3\n\tv_lshl_add_u32 %0, %0, %4, %5" : "=&v"(a) : "v"(q), "v"(qlo_v), "s"(qhi), "s"(lgb), "v"(base));
    return a;
}
// ... and the fire-and-forget add behind them
template <int OFF>
__device__ __forceinline__ g9_u32 clamp_row_addr_add(g9_u32 q, g9_u32 qlo_v, g9_u32 qhi, int lgb, g9_u32 base, g9_u32 val) {
    g9_u32 a;
    asm volatile("v_med3_u32 %0, %1, %2, %3\n\tv_lshl_add_u32 %0, %0, %4, %5\n\tds_add_u32 %0, %6 offset:%7"
                 : "=&v"(a) : "v"(q), "v"(qlo_v), "s"(qhi), "s"(lgb), "v"(base), "v"(val), "n"(OFF));
    return a;
}
// LDS DMA (global_load_lds_dwordx4): lane l's 16 bytes at g land at dst + 16 l (dst wave-uniform); counted by vmcnt
__device__ __forceinline__ void dma_to_lds16(const uint8_t *g, uint8_t *dst) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)dst, 16, 0, 0);
}
}  // namespace
}  // namespace snk
#endif
