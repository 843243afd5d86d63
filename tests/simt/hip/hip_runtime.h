// SIMT emulator: a stand-in for <hip/hip_runtime.h> that lets the HIP sources of soapnuke_amd/csrc and the CLI of
// soapnuke_amd/host compile with g++ and RUN on the CPU, wavefront semantics included.  TEST INFRASTRUCTURE ONLY (like oracle/):
// nothing in the product includes, links or loads it; it exists so that the kernels' logic, the C ABI around them and the CLI's
// host code can be checked against the oracle and the reference binary in the CPU-only test tier (tests/test_simt_*.py).
//
// Execution model
//   * a kernel launch runs its workgroups on a pool of OS threads (in workgroup order); a workgroup's threads are fibers
//     (one stack each) on one OS thread, so `__shared__` storage is `thread_local` and needs no locking;
//   * a fiber runs until it reaches a wave-level operation (ballot, shuffle, DPP, permlane, readlane, ds_bpermute ...) or
//     `__syncthreads`; there it publishes its operand and waits for the lanes it executes the operation with -- in wave-uniform
//     control flow every live lane of the wave -- so the operation sees the lanes' values exactly as the hardware's lock-step
//     execution does.  A lane that has returned from the kernel no longer takes part (the EXEC mask of an early exit);
//   * lanes that reach DIFFERENT wave operations (a vote inside `if (lane-dependent)`) are told apart by where they wait; when
//     nothing of the workgroup can run, the operation completes for the lanes that are there (wave_exchange below, and
//     release_divergent in simt_runtime.cpp) -- the lanes the EXEC mask would have enabled.  Only a workgroup in which no
//     operation can complete at all (a barrier some threads never reach) aborts, with the call chains of the waiting threads;
//   * memory: hipMalloc is malloc (filled with 0xEE so that reads of uninitialised device memory show), copies and memsets
//     are synchronous, streams and events keep their order trivially.  Device atomics are __atomic builtins.
//   * timing means nothing here.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <functional>
#include <mutex>
#include <tuple>
#include <vector>
#include <type_traits>

#define SNK_SIMT_EMUL 1
#ifndef __HIPCC__
#define __HIPCC__ 1          // the sources are compiled as HIP device code (their host-emulation branches belong to other harnesses)
#endif
#define __device__
#define __host__
#define __global__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ thread_local
#define HIP_SYMBOL(x) (x)
#define SNK_WAVE_UNIFORM_SHARED          /* state that all lanes of a wave write redundantly in lock-step: a copy per lane here */
uint8_t *simt_dyn_shared();
#define HIP_DYNAMIC_SHARED(type, var) type *var = (type *)simt_dyn_shared();
static constexpr int warpSize = 64;

// HIP's mixed-type min / max (the usual arithmetic conversions)
template <class A, class B> static inline constexpr typename std::common_type<A, B>::type max(A a, B b) {
    using C = typename std::common_type<A, B>::type;
    return (C)a < (C)b ? (C)b : (C)a;
}
template <class A, class B> static inline constexpr typename std::common_type<A, B>::type min(A a, B b) {
    using C = typename std::common_type<A, B>::type;
    return (C)b < (C)a ? (C)b : (C)a;
}
// kernel attributes of the amdgpu target that mean nothing to the host compiler
#define amdgpu_waves_per_eu(...) unused
#define amdgpu_flat_work_group_size(...) unused

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct ulonglong2 { unsigned long long x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

// ---------------------------------------------------------------------------------------------------------------- host API
typedef enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorUnknown = 999 } hipError_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
typedef enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 } hipFuncAttribute;
struct simt_stream;
struct simt_event;
typedef simt_stream *hipStream_t;
typedef simt_event *hipEvent_t;
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1, hipEventDefault = 0, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    size_t totalGlobalMem;
    int multiProcessorCount, warpSize, maxThreadsPerBlock, clockRate, memoryClockRate, memoryBusWidth;
    size_t sharedMemPerBlock, maxSharedMemoryPerMultiProcessor;
    int major, minor;
};
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipMallocAsync(void **p, size_t n, hipStream_t);
hipError_t hipFreeAsync(void *p, hipStream_t);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags = 0);
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr);
hipError_t hipMemcpyPeerAsync(void *d, int ddev, const void *s, int sdev, size_t n, hipStream_t = nullptr);
hipError_t hipMemset(void *d, int v, size_t n);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr);
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b);
hipError_t hipSetDevice(int);
hipError_t hipGetDevice(int *);
hipError_t hipGetDeviceCount(int *);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *, int);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
const char *hipGetErrorString(hipError_t);
hipError_t hipStreamCreate(hipStream_t *);
hipError_t hipStreamCreateWithFlags(hipStream_t *, unsigned);
hipError_t hipStreamDestroy(hipStream_t);
hipError_t hipStreamSynchronize(hipStream_t);
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0);
hipError_t hipEventCreate(hipEvent_t *);
hipError_t hipEventCreateWithFlags(hipEvent_t *, unsigned);
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventQuery(hipEvent_t);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
template <class F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
template <class T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f = 0) { return hipHostMalloc((void **)p, n, f); }
extern "C" void simt_dump_register(void *p, size_t n);
template <class T> static inline hipError_t hipMemcpyToSymbol(T &sym, const void *src, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyHostToDevice) {
    memcpy((char *)&sym + off, src, n);
    simt_dump_register((void *)&sym, sizeof(T));          // (a device variable the host fills: part of SIMT_DUMP_DIR's snapshots from here on)
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------------------------------- fibers
namespace simt {
struct Fiber;
struct Wave {
    int live = 0, arrived = 0;          // lanes that have not returned / that wait in a wave operation
    uint64_t live_mask = 0, pend_mask = 0;
    bool uniform = true;                // every waiting lane is at the same place (site0)
    uint64_t site0[3] = {0, 0, 0};
    uint64_t val[64];                   // operands of the waiting lanes
    uint64_t site[64][3];               // where each waits: call site of the operation, call site of the function around it, stack depth
    const char *kind[64];
    uint64_t snap[2][64];               // operands of the last two completed operations and who took part
    uint64_t snap_mask[2] = {0, 0};
    uint64_t seq = 0;                   // operations completed
    Fiber *first = nullptr;             // fiber of lane 0
};
struct Block;
struct Fiber {
    void *sp = nullptr;
    dim3 tid;
    int lane = 0, index = 0;
    bool done = false;
    Wave *wave = nullptr;
    Block *blk = nullptr;
    const uint64_t *wait_ptr = nullptr; // runnable again once *wait_ptr != wait_val
    uint64_t wait_val = 0;
    uint64_t wake = 0;                  // wave operations this lane has been released from
    int snap_idx = 0;
    const char *where = nullptr;
    uint64_t stack_top = 0;
};
struct Block {
    dim3 bid, bdim, gdim;
    int live = 0, bar_arrived = 0;
    uint64_t bar_gen = 0;
    size_t dyn_shared = 0;
    int vote_acc = 0, vote_res = 0;     // __syncthreads_or / _count: gathered before the barrier, published by its release
    void release() { bar_arrived = 0; vote_res = vote_acc; vote_acc = 0; ++bar_gen; }
};
extern thread_local Fiber *cur;
void yield_to_scheduler();
void wave_release(Wave &w, uint64_t group);
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body);

// A wave-level operation: the lane publishes its operand and waits for the lanes it executes the operation with; returns their
// operands (indexed by lane) and who they are.  In wave-uniform control flow that is every live lane, and the operation
// completes when the last one arrives.  Lanes that reach DIFFERENT operations (a vote inside `if (lane-dependent)`, the
// hardware's EXEC mask) are told apart by where they wait -- the call site of this function, its caller's call site and the stack
// depth; the scheduler completes such an operation for the lanes that are there once nothing else can run (simt_runtime.cpp:
// innermost call first, then code order -- the order in which the hardware serialises the sides of a branch does not matter to
// them, but a side must be done before the lanes meet again behind it).
// `convergent` is what the amdgpu target puts on these operations: the optimizer may not duplicate a call into the arms of a
// lane-dependent branch (jump threading would give the lanes of ONE source-level operation different call sites).
__attribute__((noinline, convergent)) const uint64_t *wave_exchange(uint64_t mine, uint64_t &mask, const char *where);
static inline void block_barrier(const char *where) {
    Fiber *f = cur;
    Block &b = *f->blk;
    if (++b.bar_arrived == b.live) b.release();
    else {
        f->wait_ptr = &b.bar_gen;
        f->wait_val = b.bar_gen;
        f->where = where;
        yield_to_scheduler();
    }
}
template <class T> static inline uint64_t bits_of(T v) {
    static_assert(sizeof(T) <= 8, "wave operand wider than 64 bits");
    uint64_t u = 0;
    memcpy(&u, &v, sizeof(T));
    return u;
}
template <class T> static inline T from_bits(uint64_t u) {
    T v;
    memcpy(&v, &u, sizeof(T));
    return v;
}
}  // namespace simt

#define threadIdx (simt::cur->tid)
#define blockIdx (simt::cur->blk->bid)
#define blockDim (simt::cur->blk->bdim)
#define gridDim (simt::cur->blk->gdim)

// SIMT_DUMP_DIR (tools/gfx950_interp.py, tests/test_isa_interp.py): the launch as the device would get it -- the kernel's address
// (resolved to its symbol by the reader), grid, the kernarg segment (every parameter at its natural alignment) and all device
// memory before and after the launch -- so that the kernel's gfx950 assembly can be run on the same state and compared.
namespace simt {
bool dump_wanted();
int dump_pre(const void *kernel, dim3 grid, dim3 block, size_t shmem, const void *kernarg, size_t kernarg_bytes);
void dump_post(int id);
std::mutex &dump_serial();          // while snapshots are taken, launches and copies of all host threads take turns (a snapshot pair must show one kernel's work only)
}  // namespace simt

template <class... P, class... A>
static inline void hipLaunchKernelGGL(void (*kernel)(P...), dim3 grid, dim3 block, size_t shmem, hipStream_t, A... args) {
    std::tuple<typename std::decay<P>::type...> params(static_cast<typename std::decay<P>::type>(args)...);
    int dump_id = -1;
    std::unique_lock<std::mutex> serial;
    if (simt::dump_wanted()) {
        serial = std::unique_lock<std::mutex>(simt::dump_serial());
        std::vector<char> ka;
        size_t off = 0;
        auto pack = [&](const auto &p) {
            const size_t a = alignof(typename std::decay<decltype(p)>::type);
            off = (off + a - 1) / a * a;
            ka.resize(off + sizeof(p));
            memcpy(ka.data() + off, &p, sizeof(p));
            off += sizeof(p);
        };
        std::apply([&](const auto &...p) { (pack(p), ...); }, params);
        dump_id = simt::dump_pre((const void *)kernel, grid, block, shmem, ka.data(), ka.size());
    }
    simt::launch(grid, block, shmem, [=] { std::apply(kernel, params); });
    if (dump_id >= 0) simt::dump_post(dump_id);
}

// ---------------------------------------------------------------------------------------------------------------- device side
static inline void __syncthreads() { simt::block_barrier("__syncthreads"); }
static inline int __syncthreads_or(int p) {
    simt::Block &b = *simt::cur->blk;
    if (p) b.vote_acc |= 1;
    simt::block_barrier("__syncthreads_or");
    return b.vote_res != 0;
}
static inline int __syncthreads_count(int p) {
    simt::Block &b = *simt::cur->blk;
    if (p) b.vote_acc += 1;
    simt::block_barrier("__syncthreads_count");
    return b.vote_res;
}
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
// lanes of a wave that hand data to each other through memory wait for each other here: a wave executes in lock-step on the
// device, one lane after the other in this emulator.  (__threadfence_block orders a lane's own accesses for the rest of the
// workgroup -- where the sources use it between lanes of one wave, it is such a point.)
#define SNK_WAVE_SYNC() do { uint64_t simt_m_; (void)simt::wave_exchange(0, simt_m_, "SNK_WAVE_SYNC"); } while (0)
__attribute__((always_inline)) static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); SNK_WAVE_SYNC(); }
static inline int __lane_id() { return simt::cur->lane; }

__attribute__((always_inline)) static inline unsigned long long __ballot(int p) {
    uint64_t m;
    const uint64_t *s = simt::wave_exchange(p ? 1 : 0, m, "__ballot");
    unsigned long long r = 0;
    for (uint64_t t = m; t; t &= t - 1) { const int l = __builtin_ctzll(t); r |= (unsigned long long)(s[l] & 1) << l; }
    return r;
}
__attribute__((always_inline)) static inline int __any(int p) { return __ballot(p) != 0; }
__attribute__((always_inline)) static inline int __all(int p) {
    uint64_t m;
    const uint64_t *s = simt::wave_exchange(p ? 1 : 0, m, "__all");
    for (uint64_t t = m; t; t &= t - 1) if (!s[__builtin_ctzll(t)]) return 0;
    return 1;
}
__attribute__((always_inline)) static inline unsigned long long __activemask() { return __ballot(1); }
template <class T> __attribute__((always_inline)) static inline T __shfl(T v, int src, int width = 64) {
    uint64_t m;
    const uint64_t *s = simt::wave_exchange(simt::bits_of(v), m, "__shfl");
    const int lane = simt::cur->lane;
    const int l = (lane & ~(width - 1)) | (src & (width - 1));
    return simt::from_bits<T>(s[l & 63]);
}
template <class T> __attribute__((always_inline)) static inline T __shfl_xor(T v, int x, int width = 64) {
    uint64_t m;
    const uint64_t *s = simt::wave_exchange(simt::bits_of(v), m, "__shfl_xor");
    const int lane = simt::cur->lane;
    int l = lane ^ x;
    if ((l & ~(width - 1)) != (lane & ~(width - 1))) l = lane;
    return simt::from_bits<T>(s[l & 63]);
}
template <class T> __attribute__((always_inline)) static inline T __shfl_up(T v, unsigned d, int width = 64) {
    uint64_t m;
    const uint64_t *s = simt::wave_exchange(simt::bits_of(v), m, "__shfl_up");
    const int lane = simt::cur->lane;
    const int l = ((lane & (width - 1)) >= (int)d) ? lane - (int)d : lane;
    return simt::from_bits<T>(s[l]);
}
template <class T> __attribute__((always_inline)) static inline T __shfl_down(T v, unsigned d, int width = 64) {
    uint64_t m;
    const uint64_t *s = simt::wave_exchange(simt::bits_of(v), m, "__shfl_down");
    const int lane = simt::cur->lane;
    const int l = ((lane & (width - 1)) + (int)d < width) ? lane + (int)d : lane;
    return simt::from_bits<T>(s[l]);
}

__attribute__((always_inline)) static inline int simt_readfirstlane(int v) {
    uint64_t m;
    const uint64_t *s = simt::wave_exchange((uint32_t)v, m, "readfirstlane");
    return (int)(uint32_t)s[__builtin_ctzll(m)];
}
__attribute__((always_inline)) static inline int simt_readlane(int v, int lane) {
    uint64_t m;
    const uint64_t *s = simt::wave_exchange((uint32_t)v, m, "readlane");
    return (int)(uint32_t)s[lane & 63];
}
static inline int simt_writelane(int val, int lane, int old) { return simt::cur->lane == (lane & 63) ? val : old; }
__attribute__((always_inline)) static inline int simt_ds_bpermute(int addr, int v) {
    uint64_t m;
    const uint64_t *s = simt::wave_exchange((uint32_t)v, m, "ds_bpermute");
    const int l = (addr >> 2) & 63;
    return (m >> l) & 1 ? (int)(uint32_t)s[l] : 0;
}
// v_mov_b32 with a DPP control (gfx9 encoding): the lane `old` keeps its value where the row / bank masks switch it off, or where
// the source lane does not exist (or is not live) and bound_ctrl is off; with bound_ctrl such a lane reads 0
__attribute__((always_inline)) static inline int simt_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    uint64_t m;
    const uint64_t *s = simt::wave_exchange((uint32_t)src, m, "update_dpp");
    const int lane = simt::cur->lane, row = lane >> 4, in = lane & 15;
    if (!((row_mask >> row) & 1) || !((bank_mask >> (in >> 2)) & 1)) return old;
    int from = -1;
    if (ctrl >= 0x00 && ctrl <= 0xFF) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);                 // quad_perm
    else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl & 15; if (in + n < 16) from = lane + n; }  // row_shl
    else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl & 15; if (in - n >= 0) from = lane - n; }  // row_shr
    else if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl & 15; from = (lane & ~15) | ((in - n) & 15); }   // row_ror
    else if (ctrl == 0x130) { if (lane + 1 < 64) from = lane + 1; }                                          // wave_shl:1
    else if (ctrl == 0x134) from = (lane + 1) & 63;                                                          // wave_rol:1
    else if (ctrl == 0x138) { if (lane >= 1) from = lane - 1; }                                              // wave_shr:1
    else if (ctrl == 0x13C) from = (lane - 1) & 63;                                                          // wave_ror:1
    else if (ctrl == 0x140) from = (lane & ~15) | (15 - in);                                                 // row_mirror
    else if (ctrl == 0x141) from = (lane & ~7) | (7 - (lane & 7));                                           // row_half_mirror
    else if (ctrl == 0x142) { if (row >= 1) from = row * 16 - 1; }                                           // row_bcast:15
    else if (ctrl == 0x143) { if (row >= 2) from = 31; }                                                     // row_bcast:31
    else { fprintf(stderr, "simt: DPP control 0x%x is not modelled\n", ctrl); abort(); }
    if (from < 0 || !((m >> from) & 1)) return bound_ctrl ? 0 : old;
    return (int)(uint32_t)s[from];
}
struct simt_u32x2 {
    unsigned v[2];
    unsigned operator[](int i) const { return v[i]; }
    unsigned &operator[](int i) { return v[i]; }
};
// v_permlane32_swap vdst, vsrc: lanes 32..63 of vdst <-> lanes 0..31 of vsrc; [0] = new vdst, [1] = new vsrc
static inline simt_u32x2 simt_permlane32_swap(unsigned vdst, unsigned vsrc, bool, bool) {
    uint64_t m;
    const uint64_t *s = simt::wave_exchange(((uint64_t)vsrc << 32) | vdst, m, "permlane32_swap");
    const int lane = simt::cur->lane;
    simt_u32x2 r;
    r.v[0] = lane >= 32 ? (unsigned)(s[lane - 32] >> 32) : vdst;
    r.v[1] = lane < 32 ? (unsigned)s[lane + 32] : vsrc;
    return r;
}
// v_permlane16_swap vdst, vsrc: the odd rows of vdst <-> the even rows of vsrc
static inline simt_u32x2 simt_permlane16_swap(unsigned vdst, unsigned vsrc, bool, bool) {
    uint64_t m;
    const uint64_t *s = simt::wave_exchange(((uint64_t)vsrc << 32) | vdst, m, "permlane16_swap");
    const int lane = simt::cur->lane, row = lane >> 4;
    simt_u32x2 r;
    r.v[0] = (row & 1) ? (unsigned)(s[lane - 16] >> 32) : vdst;
    r.v[1] = (row & 1) ? vsrc : (unsigned)s[lane + 16];
    return r;
}
#define __builtin_amdgcn_readfirstlane(x) simt_readfirstlane((int)(x))
#define __builtin_amdgcn_readlane(x, l) simt_readlane((int)(x), (l))
#define __builtin_amdgcn_ds_bpermute(a, v) simt_ds_bpermute((a), (v))
#define __builtin_amdgcn_update_dpp(o, s, c, r, b, bc) simt_update_dpp((int)(o), (int)(s), (c), (r), (b), (bc))
#define __builtin_amdgcn_mov_dpp(s, c, r, b, bc) simt_update_dpp(0, (int)(s), (c), (r), (b), (bc))
#define __builtin_amdgcn_permlane32_swap(a, b, f, c) simt_permlane32_swap((a), (b), (f), (c))
#define __builtin_amdgcn_permlane16_swap(a, b, f, c) simt_permlane16_swap((a), (b), (f), (c))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))

// ---- per-lane arithmetic builtins
static inline unsigned simt_alignbit(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (sh & 31u)); }
static inline unsigned simt_alignbyte(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (8 * (sh & 3u))); }
// v_perm_b32 D = perm(S0, S1, sel): byte k of D is picked by selector byte k out of {S0 (bytes 7..4), S1 (bytes 3..0)}
static inline unsigned simt_perm(unsigned s0, unsigned s1, unsigned sel) {
    const unsigned long long src = ((unsigned long long)s0 << 32) | s1;
    unsigned d = 0;
    for (int k = 0; k < 4; ++k) {
        const unsigned c = (sel >> (8 * k)) & 0xFF;
        unsigned b;
        if (c <= 7) b = (unsigned)(src >> (8 * c)) & 0xFF;
        else if (c == 8) b = (s1 >> 15) & 1 ? 0xFF : 0;        // sign of S1's low word
        else if (c == 9) b = (s1 >> 31) & 1 ? 0xFF : 0;
        else if (c == 10) b = (s0 >> 15) & 1 ? 0xFF : 0;
        else if (c == 11) b = (s0 >> 31) & 1 ? 0xFF : 0;
        else if (c == 12) b = 0;
        else b = 0xFF;
        d |= b << (8 * k);
    }
    return d;
}
static inline unsigned simt_sad_u8(unsigned a, unsigned b, unsigned c) {
    unsigned s = c;
    for (int k = 0; k < 4; ++k) { const int x = (a >> (8 * k)) & 0xFF, y = (b >> (8 * k)) & 0xFF; s += (unsigned)(x > y ? x - y : y - x); }
    return s;
}
static inline unsigned simt_sad_hi_u8(unsigned a, unsigned b, unsigned c) { return (simt_sad_u8(a, b, 0) << 16) + c; }
static inline unsigned simt_ubfe(unsigned v, unsigned off, unsigned w) {
    off &= 31; w &= 31;
    return w == 0 ? 0 : (v >> off) & ((1u << w) - 1u);
}
#define __builtin_amdgcn_alignbit(hi, lo, sh) simt_alignbit((hi), (lo), (sh))
#define __builtin_amdgcn_alignbyte(hi, lo, sh) simt_alignbyte((hi), (lo), (sh))
#define __builtin_amdgcn_perm(a, b, s) simt_perm((a), (b), (s))
#define __builtin_amdgcn_sad_u8(a, b, c) simt_sad_u8((a), (b), (c))
#define __builtin_amdgcn_sad_hi_u8(a, b, c) simt_sad_hi_u8((a), (b), (c))
#define __builtin_amdgcn_ubfe(v, o, w) simt_ubfe((v), (o), (w))

static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline unsigned __brev(unsigned x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
}
static inline unsigned long long __brevll(unsigned long long x) { return ((unsigned long long)__brev((unsigned)x) << 32) | __brev((unsigned)(x >> 32)); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }

// ---- atomics (global and LDS alike)
template <class T, class U> static inline T atomicAdd(T *p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicSub(T *p, U v) { return __atomic_fetch_sub(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicOr(T *p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicAnd(T *p, U v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicXor(T *p, U v) { return __atomic_fetch_xor(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicExch(T *p, U v) { return __atomic_exchange_n(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicMax(T *p, U v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o < (T)v && !__atomic_compare_exchange_n(p, &o, (T)v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
template <class T, class U> static inline T atomicMin(T *p, U v) {
    T o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o > (T)v && !__atomic_compare_exchange_n(p, &o, (T)v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
template <class T, class U, class V> static inline T atomicCAS(T *p, U cmp, V val) {
    T e = (T)cmp;
    __atomic_compare_exchange_n(p, &e, (T)val, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
    return e;
}
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or((p), (v), (order))
#define __hip_atomic_fetch_max(p, v, order, scope) atomicMax((p), (v))
