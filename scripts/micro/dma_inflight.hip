// Microbenchmark (dev only): LDS-DMA fill rate per CU as a function of bytes in flight, weights-like
// access (all workgroups stream the same 2.4 MB L2-resident buffer, rotated start).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int KB>  // bytes in flight per step
__global__ __launch_bounds__(768) void k(const char* src, unsigned nbytes, int steps, int rot, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    constexpr int INSTR = KB;  // 1 KiB per DMA instruction
    unsigned pos = rot ? (blockIdx.x * 196608u) % nbytes : 0u;
    for (int s = 0; s < steps; ++s) {
        for (int i = wv; i < INSTR; i += 12) {
            unsigned off = (pos + i * 1024u + lane * 16u) % nbytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + (i % 128) * 1024), 16, off, 0, 0, 0);
        }
        pos = (pos + INSTR * 1024u) % nbytes;
        __syncthreads();
    }
    if (tid == 0) sink[blockIdx.x] = *(float*)smem;
}
template <int KB> void run(const char* d, unsigned nbytes, float* sink, int rot) {
    const int total_kb = 2048;  // per block
    const int steps = total_kb / KB;
    hipFuncSetAttribute((const void*)k<KB>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<KB>, dim3(256), dim3(768), 131072, 0, d, nbytes, steps, rot, sink);
    hipEventRecord(a);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<KB>, dim3(256), dim3(768), 131072, 0, d, nbytes, steps, rot, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("in-flight %3d KB rot=%d: %7.1f us for 2 MB/CU -> %6.1f GB/s per CU, %5.1f TB/s chip\n", KB, rot, ms * 1e3,
           2.0 * 1.048576 / ms, 2.0 * 1.048576 * 256 / ms / 1e3);
}
int main() {
    const unsigned nbytes = 2359296;  // 2.25 MiB (W1 + W2 of one ViT-S layer)
    char* d; float* sink; hipMalloc(&d, nbytes); hipMalloc(&sink, 4096); hipMemset(d, 1, nbytes);
    for (int rot = 0; rot < 2; ++rot) { run<16>(d, nbytes, sink, rot); run<32>(d, nbytes, sink, rot); run<64>(d, nbytes, sink, rot); run<96>(d, nbytes, sink, rot); run<128>(d, nbytes, sink, rot); }
    return 0;
}
