// dev only: what overlaps with what on one SIMD of gfx950. One workgroup of 8 waves (two per SIMD): waves 0-3 run role A,
// waves 4-7 role B; each role's elapsed cycles alone and beside the other. hipcc --offload-arch=gfx950 corun.hip -o corun
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
enum { IDLE = 0, M16 = 1, M32 = 2, VALU = 3, LDSR = 4, M16V2 = 5, M16L = 6, DMA = 7, M16V1 = 8 };
template <int R>
__device__ __forceinline__ void role(float* out, int n, char* smem, const char* gsrc) {
    const int lane = threadIdx.x & 63;
    if (R == IDLE) return;
    if (R == M16 || R == M16V2 || R == M16L || R == M16V1) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(1.0f - lane * 0.003f); }
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = lane + i;
        u32x4 ld[2];
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                if (R == M16V2) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(1.0001f)); asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[(i + 4) & 7]) : "v"(1.0001f)); }
                if (R == M16V1) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(1.0001f)); }
                if (R == M16L && (i & 1)) { asm volatile("ds_read_b128 %0, %1" : "=v"(ld[(i >> 1) & 1]) : "v"(lane * 16 + (i & 6) * 512)); }
            }
            if (R == M16L) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ld[0]), "+v"(ld[1]));
        }
        float s = 0;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + v[i];
        if (R == M16L) s += __builtin_bit_cast(float, ld[0][0] ^ ld[1][1]);
        out[threadIdx.x] = s;
    }
    if (R == M32) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(1.0f - lane * 0.003f); }
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0;
        for (int i = 0; i < 4; ++i) s += acc[i][0];
        out[threadIdx.x] = s;
    }
    if (R == VALU) {
        float r[16];
        for (int i = 0; i < 16; ++i) r[i] = lane + i;
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(1.0001f));
        }
        float s = 0;
        for (int i = 0; i < 16; ++i) s += r[i];
        out[threadIdx.x] = s;
    }
    if (R == LDSR) {
        u32x4 r[8];
        unsigned x = 0;
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(r[i]) : "v"(lane * 16 + i * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
            x ^= r[0][0] ^ r[7][3];
        }
        out[threadIdx.x] = __builtin_bit_cast(float, x);
    }
    if (R == DMA) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(gsrc), 0, 1u << 24, 0x00020000);
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 32768 + wv * 4096 + i * 1024), 16,
                                                         lane * 16, ((it * 4 + i) & 1023) * 1024 + wv * (1 << 20), 0, 0);
            __builtin_amdgcn_s_waitcnt((4 & 15) | (7 << 4) | (15 << 8));
        }
        __builtin_amdgcn_s_waitcnt((0) | (7 << 4) | (15 << 8));
        out[threadIdx.x] = smem[32768 + threadIdx.x];
    }
}
template <int RA, int RB>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* clk, int na, int nb, const char* gsrc) {
    extern __shared__ char smem[];
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<float*>(smem)[i] = i;
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (wv < 4) role<RA>(out + blockIdx.x * 512, na, smem, gsrc); else role<RB>(out + blockIdx.x * 512, nb, smem, gsrc);
    asm volatile("s_nop 0" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax((unsigned long long*)&clk[wv < 4 ? 0 : 1], (unsigned long long)(t1 - t0));
}
template <int RA, int RB>
void run(const char* name, int na, int nb, int ia, int ib, int grid = 1) {
    float* out; long long* clk; char* g;
    hipMalloc(&out, 512 * 4 * 256); hipMalloc(&clk, 16); hipMalloc(&g, 1 << 24);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<RA, RB>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) { hipMemset(clk, 0, 16); k<RA, RB><<<grid, 512, 65536>>>(out, clk, na, nb, g); }
    hipDeviceSynchronize();
    long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    printf("%-28s grid %3d:  A %8lld cycles", name, grid, c[0]);
    if (ia) printf(" = %6.2f per instr", (double)c[0] / ((double)na * ia));
    printf("   B %8lld cycles", c[1]);
    if (ib) printf(" = %6.2f per instr", (double)c[1] / ((double)nb * ib));
    printf("\n");
    hipFree(out); hipFree(clk); hipFree(g);
}
int main() {
    const int n = 20000;
    run<M16, IDLE>("mfma16 | idle", n, 0, 8, 0);
    run<M16, M16>("mfma16 | mfma16", n, n, 8, 8);
    run<M32, IDLE>("mfma32 | idle", n, 0, 4, 0);
    run<IDLE, VALU>("idle | valu", 0, n, 0, 16);
    run<VALU, VALU>("valu | valu", n, n, 16, 16);
    run<M16, VALU>("mfma16 | valu", n, n, 8, 16);
    run<M32, VALU>("mfma32 | valu", n, 2 * n, 4, 16);
    run<IDLE, LDSR>("idle | ldsread", 0, n, 0, 8);
    run<M16, LDSR>("mfma16 | ldsread", n, n, 8, 8);
    run<M32, LDSR>("mfma32 | ldsread", n, 2 * n, 4, 8);
    run<M16V1, IDLE>("mfma16+1valu | idle", n, 0, 8, 0);
    run<M16V2, IDLE>("mfma16+2valu | idle", n, 0, 8, 0);
    run<M16V2, M16V2>("mfma16+2valu | same", n, n, 8, 8);
    run<M16V1, M16V1>("mfma16+1valu | same", n, n, 8, 8);
    run<M16L, IDLE>("mfma16+.5read | idle", n, 0, 8, 0);
    run<M16L, M16L>("mfma16+.5read | same", n, n, 8, 8);
    run<IDLE, DMA>("idle | dma", 0, n / 4, 0, 4);
    run<M16, DMA>("mfma16 | dma", n, n / 4, 8, 4);
    run<IDLE, DMA>("idle | dma", 0, n / 4, 0, 4, 256);
    run<M16, DMA>("mfma16 | dma", n, n / 4, 8, 4, 256);
    return 0;
}
