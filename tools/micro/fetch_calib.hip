// Known-bytes calibration of the FETCH_SIZE counter on gfx950: each kernel reads every byte of a
// 2 GiB buffer exactly once with one access pattern; rocprofv3 --pmc FETCH_SIZE then tells how the counter's unit
// relates to the bytes really fetched for THAT pattern (the guide's x2 correction was derived for 16 B/lane streaming).
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ; tools/fetch_calib.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *glb_ptr_t;
__global__ void __launch_bounds__(256) rd16(const uint4 *p, size_t n16, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) rd4(const unsigned *p, size_t n4, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) rd1(const uint8_t *p, size_t n, unsigned *out) {       // 64 consecutive bytes per wave instruction
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) rd1_rows(const uint8_t *p, size_t rows, int pitch, int len, unsigned *out) {   // phase 3's pattern: rows of `len` bytes at `pitch`, 64-byte strips
    unsigned acc = 0;
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = wave; r < rows; r += nw)
        for (int s = 0; s * 64 < len; ++s) { const int o = s * 64 + lane; acc += p[r * pitch + (o < pitch ? o : pitch - 1)]; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) rd_dma(const uint8_t *p, size_t n16, unsigned *out) {     // global_load_lds, 16 B/lane
    __shared__ __attribute__((aligned(16))) uint8_t buf[256 * 16];
    for (size_t i = (size_t)blockIdx.x * blockDim.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const size_t j = i + threadIdx.x;
        if (j < n16) __builtin_amdgcn_global_load_lds((glb_ptr_t)(p + j * 16), (lds_ptr_t)(buf + (threadIdx.x & ~63) * 16), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (buf[threadIdx.x] == 0xA7 && buf[threadIdx.x + 1] == 0x11) out[0] = 1;
}
int main() {
    const size_t bytes = (size_t)2 << 30;
    uint8_t *d; unsigned *o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4);
    hipMemset(d, 1, bytes);
    hipDeviceSynchronize();
    rd16<<<4096, 256>>>((const uint4 *)d, bytes / 16, o);
    rd4<<<4096, 256>>>((const unsigned *)d, bytes / 4, o);
    rd1<<<4096, 256>>>(d, bytes, o);
    rd1_rows<<<4096, 256>>>(d, bytes / 160, 160, 150, o);
    rd_dma<<<4096, 256>>>(d, bytes / 16, o);
    hipDeviceSynchronize();
    printf("known bytes: rd16 %zu rd4 %zu rd1 %zu rd1_rows(150 of 160) %zu rd_dma %zu\n", bytes, bytes, bytes, (bytes / 160) * 150, bytes);
    return 0;
}
