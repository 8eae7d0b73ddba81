// dev probe (round 4): does buffer_store_dwordx4 with an SGPR soffset read its write-data registers late on gfx950?
// Every lane stores 16 bytes of a known pattern through a buffer descriptor (voffset in a VGPR, column offset in an SGPR), and NOPS
// wait states later the four data registers are overwritten with a poison value - what the register allocator does when it reuses the
// registers of a dead value. 768-thread workgroups (three waves per SIMD), many stores in flight. The host counts poison words in memory.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mubuf_store_hazard.hip -o scripts/micro/build/mubuf_store_hazard
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 768, ITER = 64;
constexpr unsigned POISON = 0xdeadbeefu;

template <int NOPS, bool SGPR_OFF>
__global__ __launch_bounds__(THREADS) void probe(unsigned* out, unsigned bytes) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, bytes, 0x00020000);
    const unsigned tid = blockIdx.x * THREADS + threadIdx.x;
    for (int it = 0; it < ITER; ++it) {
        const unsigned idx = (unsigned)it * gridDim.x * THREADS + tid;  // 16-byte slot
        const unsigned a0 = idx * 4u, a1 = idx * 4u + 1u, a2 = idx * 4u + 2u, a3 = idx * 4u + 3u;
        const unsigned vo = idx * 16u;
        const int so = __builtin_amdgcn_readfirstlane((int)(it >> 30));  // 0, in an SGPR
        // fixed registers v[4:7]: data in, store, NOPS wait states, first and last register overwritten
        if (SGPR_OFF)
            asm volatile("v_mov_b32 v4, %0\n\tv_mov_b32 v5, %1\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\ts_nop 7\n\t"
                         "buffer_store_dwordx4 v[4:7], %4, %5, %6 offen\n\t"
                         ".rept %7\n\ts_nop 0\n\t.endr\n\t"
                         "v_mov_b32 v4, 0xdeadbeef\n\tv_mov_b32 v7, 0xdeadbeef"
                         :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(vo), "s"(r), "s"(so), "n"(NOPS) : "v4", "v5", "v6", "v7", "memory");
        else
            asm volatile("v_mov_b32 v4, %0\n\tv_mov_b32 v5, %1\n\tv_mov_b32 v6, %2\n\tv_mov_b32 v7, %3\n\ts_nop 7\n\t"
                         "buffer_store_dwordx4 v[4:7], %4, %5, 0 offen\n\t"
                         ".rept %6\n\ts_nop 0\n\t.endr\n\t"
                         "v_mov_b32 v4, 0xdeadbeef\n\tv_mov_b32 v7, 0xdeadbeef"
                         :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(vo), "s"(r), "n"(NOPS) : "v4", "v5", "v6", "v7", "memory");
    }
}

template <int NOPS, bool SGPR_OFF>
static void run(unsigned* d, size_t words, int grid) {
    (void)hipMemset(d, 0, words * 4);
    hipLaunchKernelGGL((probe<NOPS, SGPR_OFF>), dim3(grid), dim3(THREADS), 0, 0, d, (unsigned)(words * 4));
    std::vector<unsigned> h(words);
    (void)hipMemcpy(h.data(), d, words * 4, hipMemcpyDeviceToHost);
    size_t poison = 0, wrong = 0;
    int lanes[64] = {0};
    for (size_t i = 0; i < words; ++i) {
        if (h[i] == POISON) { ++poison; ++lanes[(i / 4) % 64]; }
        else if (h[i] != (unsigned)i) ++wrong;
    }
    printf("soffset %s, %2d wait states between the store and the overwrite: %zu poison words, %zu other wrong words of %zu; by lane:", SGPR_OFF ? "SGPR" : "imm ", NOPS,
           poison, wrong, words);
    for (int l = 0; l < 64; ++l) if (lanes[l]) printf(" %d:%d", l, lanes[l]);
    printf("\n");
}

int main() {
    const int grid = 1024;
    const size_t words = (size_t)ITER * grid * THREADS * 4;
    unsigned* d;
    (void)hipMalloc(&d, words * 4);
    run<0, true>(d, words, grid);
    run<1, true>(d, words, grid);
    run<2, true>(d, words, grid);
    run<4, true>(d, words, grid);
    run<8, true>(d, words, grid);
    run<16, true>(d, words, grid);
    run<0, false>(d, words, grid);
    run<1, false>(d, words, grid);
    run<2, false>(d, words, grid);
    return 0;
}
