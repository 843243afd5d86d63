// Host stand-in for <hip/hip_runtime.h>: lets the device headers that hold pure per-lane logic (snk_common.hip.h,
// snk_adapter_bits.hip.h) compile with g++ as ONE lane of a wavefront, so that the bit-sliced adapter search can be fuzzed against
// the oracle without a GPU (tests/test_host_emul.py).  Wave votes see that one lane; nothing here is product code.
#pragma once
#include <stdint.h>
#include <string.h>
#include <algorithm>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(x)

using std::max;
using std::min;

struct uint4 { unsigned x, y, z, w; };

static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline unsigned snk_emul_alignbit(unsigned hi, unsigned lo, unsigned sh) {
    return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (sh & 31u));
}
#define __builtin_amdgcn_alignbit(hi, lo, sh) snk_emul_alignbit((hi), (lo), (sh))
#define __builtin_amdgcn_readfirstlane(x) (x)
static inline bool __any(bool p) { return p; }
static inline bool __all(bool p) { return p; }
static inline unsigned long long __ballot(bool p) { return p ? 1ull : 0ull; }
static inline int __lane_id() { return 0; }
template <class T> static inline T __shfl(T v, int, int) { return v; }
template <class T, class U> static inline T atomicAdd(T *p, U v) { T o = *p; *p = (T)(o + v); return o; }
template <class T, class U> static inline T atomicMax(T *p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicMin(T *p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
