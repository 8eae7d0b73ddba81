// Microbenchmark (dev only): LDS-DMA fill rate per CU for the row-gather patterns of the implicit-GEMM kernels:
//   mode 0: 1 KiB contiguous per instruction; mode 1: 8 rows x 128 B (row stride 768 B); mode 2: 16 rows x 64 B.
// All workgroups stream the same L2-resident 2.25 MiB buffer (rotated start), 64 KiB in flight per step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int MODE>
__global__ __launch_bounds__(512) void k(const char* src, unsigned nbytes, int steps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    unsigned pos = (blockIdx.x * 196608u) % nbytes;
    for (int s = 0; s < steps; ++s) {
        for (int i = wv; i < 64; i += 8) {
            unsigned off;
            if (MODE == 0) off = pos + i * 1024u + lane * 16u;
            else if (MODE == 1) off = pos + (i * 8u + (lane >> 3)) * 768u + (lane & 7) * 16u;
            else off = pos + (i * 16u + (lane >> 2)) * 768u + (lane & 3) * 16u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + i * 1024), 16, off % nbytes, 0, 0, 0);
        }
        pos = (pos + 65536u) % nbytes;
        __syncthreads();
    }
    if (tid == 0) sink[blockIdx.x] = *(float*)smem;
}
template <int MODE> void run(const char* d, unsigned nbytes, float* sink) {
    const int steps = 32;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 65536, 0, d, nbytes, steps, sink);
    hipEventRecord(a);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 65536, 0, d, nbytes, steps, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("mode %d: %7.1f us for 2 MiB/CU -> %6.1f GB/s per CU, %5.1f TB/s chip\n", MODE, ms * 1e3, 2.0 * 1.048576 / ms,
           2.0 * 1.048576 * 256 / ms / 1e3);
}
int main() {
    const unsigned nbytes = 2359296;
    char* d; float* sink; hipMalloc(&d, nbytes); hipMalloc(&sink, 4096); hipMemset(d, 1, nbytes);
    run<0>(d, nbytes, sink); run<1>(d, nbytes, sink); run<2>(d, nbytes, sink);
    return 0;
}
