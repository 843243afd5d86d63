// Microbenchmark: issue rate of LDS ops per CU (conflict-free wave64 ds_add_u32 / ds_read_u8 / ds_read_b32).
// hipcc --offload-arch=gfx950 -O3 -o lds_rate lds_rate.hip && ./lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned *out, int iters) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned a = (wave * 256 + lane) * 4, one = 1, acc = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(a), "v"(one), "n"(0));
        } else if (MODE == 1) {
            unsigned t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(t[j]) : "v"(a >> 2), "n"(64 * 0));
            asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += t[j];
        } else if (MODE == 2) {
            unsigned t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t[j]) : "v"(a), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += t[j];
        } else if (MODE == 3) {   // same op mix as phase 1: 6 byte reads + 6 adds
            unsigned t[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(t[j]) : "v"(a >> 2), "n"(0));
#pragma unroll
            for (int j = 0; j < 6; ++j) asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(a), "v"(one), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(6)");
#pragma unroll
            for (int j = 0; j < 6; ++j) acc += t[j];
        }
        else if (MODE == 4) {   // op mix of phase 1 since v9: 2 dword reads + 3 byte reads + 3 adds per read
            unsigned t[5];
            asm volatile("ds_read_b32 %0, %1" : "=v"(t[0]) : "v"(a));
            asm volatile("ds_read_b32 %0, %1 offset:1024" : "=v"(t[1]) : "v"(a));
#pragma unroll
            for (int j = 2; j < 5; ++j) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(t[j]) : "v"(a >> 2), "n"(0));
#pragma unroll
            for (int j = 0; j < 3; ++j) asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(a), "v"(one), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(3)");
#pragma unroll
            for (int j = 0; j < 5; ++j) acc += t[j];
        } else if (MODE == 5) {   // adds of 22 active lanes (the third strip of a 150-base read)
            if (lane < 22) {
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("ds_add_u32 %0, %1 offset:%2" ::"v"(a), "v"(one), "n"(0));
            }
        } else if (MODE == 6) {   // 64-bit adds
            unsigned long long one64 = 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("ds_add_u64 %0, %1 offset:%2" ::"v"(a * 2), "v"(one64), "n"(0));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    __syncthreads();
    if (acc == 0xdeadbeef || lds[threadIdx.x] == 0xdeadbeef) out[0] = acc;
}
template <int MODE>
void run(const char *name, int opsPerIter, int waves) {
    unsigned *out;
    hipMalloc(&out, 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<MODE><<<256, waves * 64, 65536>>>(out, 100);
    hipEventRecord(e0);
    k<MODE><<<256, waves * 64, 65536>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9;     // nominal 2.4 GHz
    printf("%-28s waves/CU %2d: %.2f cycles per wave-op per CU\n", name, waves, cyc / ((double)iters * opsPerIter * waves));
}
int main() {
    for (int w : {4, 16}) {
        if (w == 4) { run<0>("ds_add_u32 (no rtn)", 8, 4); run<1>("ds_read_u8", 8, 4); run<2>("ds_read_b32", 8, 4); run<3>("6 rd_u8 + 6 add", 12, 4); }
        else { run<0>("ds_add_u32 (no rtn)", 8, 16); run<1>("ds_read_u8", 8, 16); run<2>("ds_read_b32", 8, 16); run<3>("6 rd_u8 + 6 add", 12, 16);
               run<4>("2 rd_b32 + 3 rd_u8 + 3 add", 8, 16); run<5>("ds_add_u32, 22 lanes", 8, 16); run<6>("ds_add_u64", 8, 16); }
    }
    return 0;
}
