// Checks snk::bit_transpose64 (csrc/snk_bittr.hip.h) against a host transpose on random matrices.
// hipcc --offload-arch=gfx950 -O3 -I soapnuke_amd/csrc tools/micro/bittr_test.hip -o tools/micro/bittr_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "snk_bittr.hip.h"

__global__ void k(const uint32_t *in, uint32_t *out, uint32_t *out_lo) {
    const int lane = threadIdx.x & 63, m = blockIdx.x;
    unsigned lo = in[(m * 64 + lane) * 2], hi = in[(m * 64 + lane) * 2 + 1];
    const unsigned l2 = snk::bit_transpose64_lo(lo, hi, lane);
    snk::bit_transpose64(lo, hi, lane);
    out[(m * 64 + lane) * 2] = lo;
    out[(m * 64 + lane) * 2 + 1] = hi;
    out_lo[m * 64 + lane] = l2;
}

int main() {
    const int M = 256;
    std::vector<uint32_t> h(M * 128), o(M * 128), ol(M * 64);
    uint64_t s = 88172645463325252ull;
    for (auto &x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)(s >> 16); }
    uint32_t *di, *dout, *dl;
    hipMalloc(&di, h.size() * 4); hipMalloc(&dout, h.size() * 4); hipMalloc(&dl, ol.size() * 4);
    hipMemcpy(di, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    k<<<M, 64>>>(di, dout, dl);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ol.data(), dl, ol.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0, badl = 0;
    for (int m = 0; m < M; ++m)
        for (int p = 0; p < 64; ++p)
            for (int r = 0; r < 64; ++r) {
                const int src = (h[(m * 64 + p) * 2 + (r >> 5)] >> (r & 31)) & 1;
                const int dst = (o[(m * 64 + r) * 2 + (p >> 5)] >> (p & 31)) & 1;
                bad += src != dst;
                if (p < 32) badl += src != (int)((ol[m * 64 + r] >> p) & 1);
            }
    printf("bit_transpose64: %ld wrong bits, lo variant: %ld wrong bits\n", bad, badl);
    return bad || badl ? 1 : 0;
}
