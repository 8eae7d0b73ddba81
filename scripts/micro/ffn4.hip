// dev micro-benchmark (round 3): the step loop of the f16x3 FFN kernel with ONE wave per SIMD - 4 waves x up to 512 registers,
// wave tiles twice as wide (A-step 48 rows x 64 units: 36 MFMAs for 14 fragment reads instead of 18 for 10; B-step 48 rows x 96
// outputs: 54 MFMAs for 12 (+6) reads instead of 27 for 6 (+6)), fragments of step t + 1 read into a second register set and
// the DMA pieces of step t + 3 issued BETWEEN the MFMAs of step t (one LDS / DMA instruction behind each MFMA, pinned), one
// barrier per step. Same LDS map, same bytes streamed, same MFMA count per SIMD as scripts/micro/ffn12.hip (8 waves: 577 ns
// per step) and the product kernel. Question: does a single wave per SIMD keep the matrix pipe fed when nothing competes with
// it for the SIMD's issue slots?
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/ffn4.hip -o scripts/micro/build/ffn4 [-DABL=n]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#ifndef ABL  // timing-only ablations: 1 no DMA traffic, 2 no fragment reads, 4 no MFMA, 8 no DMA instructions, 16 no barrier, 32 x pieces L2-hot, 64 x pieces with empty descriptor
#define ABL 0
#endif
#ifndef SPREAD
#define SPREAD 1  // 1: reads / DMA pieces one behind each MFMA; 0: all reads and pieces first, then the MFMAs
#endif
constexpr int BM = 96, E = 384, CHUNK = 128, NCH = 12;
constexpr int WAVES = 4, THREADS = WAVES * 64;
constexpr int G_KB = BM * 128, OFF_RING = 4 * G_KB, SLOTB = 28 * 1024, LDS = OFF_RING + 4 * SLOTB;
constexpr int NA = 12, NB = 8, STEPS = NA + NB;
constexpr int A_BLOCK = CHUNK * 128, B_BLOCK = 192 * 128, B_PART = NA * A_BLOCK, CHUNK_BYTES = B_PART + NB * B_BLOCK;
constexpr int X_OFF = 16 * 1024;
constexpr int PB = 6;  // DMA pieces per wave in a B-step (24); A-step: 16 W1 pieces by waves 0, 1 (8 each), 12 x pieces by waves 2, 3 (6 each)

#define WAITVM(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (7 << 4) | (0 << 8) | (((N) >> 4) << 14))  // vmcnt(N) lgkmcnt(0)

__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    if (ABL & 4) return c;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(THREADS, 1) void ffn4_kernel(const char* __restrict__ wpack, unsigned w_bytes, const char* __restrict__ h,
                                                          unsigned h_bytes, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wv >> 1, cg = wv & 1;
    const int f_row = lane & 15, f_kg = lane >> 4;
    const int m0 = (ABL & 32) ? 0 : blockIdx.x * BM;  // ABL 32: every workgroup streams the SAME 96 rows (L2-hot x pieces)
    char* const ring = smem + OFF_RING;
    for (int i = tid; i < 4 * G_KB / 4; i += THREADS) reinterpret_cast<unsigned*>(smem)[i] = 0x2c003c00u + ((i * 2654435761u) >> 20 & 0x007f007fu);
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wpack), 0, (ABL & 1) ? 0u : w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(h), 0, (ABL & (1 | 64)) ? 0u : h_bytes, 0x00020000);
    const unsigned v_w = (unsigned)lane * 16u;
    const int x_l = lane >> 3;
    auto v_x = [&](int q) { return (unsigned)(m0 + 8 * q + x_l) * (unsigned)(E * 4) + (unsigned)(((lane & 7) ^ x_l) << 4); };
    // piece u of this wave for step t (0 .. 19) of the chunk at byte offset cb -> ring slot t & 3. A-step: every wave fetches four
    // W1 pieces (u = 0 .. 3) and three x pieces (u = 4 .. 6) - the same instruction stream for all waves, no piece behind a branch.
    // Two SALU instructions (M0, scalar offset) and the load.
    const unsigned vx0 = v_x(3 * wv);
    auto issue_piece = [&](int cb, int t, int u) {
        if (ABL & 8) return;
        char* dst = ring + (t & 3) * SLOTB;
        if (t < NA) {
            if (u < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + (4 * wv + u) * 1024), 16, v_w, cb + t * A_BLOCK + (4 * wv + u) * 1024, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr_t)(dst + X_OFF + (3 * wv + u - 4) * 1024), 16, vx0, t * 128 + (u - 4) * 8 * E * 4, 0, 0);
        } else {
            const int q = PB * wv + u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + q * 1024), 16, v_w, cb + B_PART + (t - NA) * B_BLOCK + q * 1024, 0, 0);
        }
    };
    auto n_pieces = [](int t) { return t < NA ? 7 : PB; };
    const int sw = f_row & 7;
    const int ch_hi = (f_kg ^ sw) << 4, ch_lo = ((4 + f_kg) ^ sw) << 4;
    const int rows0 = rg * 48 + f_row;
    // as asm: the compiler puts `s_waitcnt vmcnt(0)` in front of an ordinary LDS read that follows an LDS-DMA instruction; the
    // step's own lgkmcnt(0) at its top covers these reads one step later. Address = per-lane constant + region base (through an
    // opaque SGPR: otherwise all ~300 distinct addresses of the unrolled chunk are hoisted out of the loop into registers) +
    // 2048 m as the instruction's immediate offset.
    auto rd = [&](int base, int m) -> u32x4 {
        if (ABL & 2) return u32x4{(unsigned)base, (unsigned)m, 2u, 3u};
        u32x4 v;
        switch (m) {
            case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(base)); break;
            case 1: asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(v) : "v"(base)); break;
            case 2: asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(v) : "v"(base)); break;
            case 3: asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(v) : "v"(base)); break;
            case 4: asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(v) : "v"(base)); break;
            default: asm volatile("ds_read_b128 %0, %1 offset:10240" : "=v"(v) : "v"(base)); break;
        }
        return v;
    };
    const int lw_a[2] = {(cg * 64 + f_row) * 128 + ch_hi, (cg * 64 + f_row) * 128 + ch_lo};
    const int lw_b[2] = {(cg * 96 + f_row) * 128 + ch_hi, (cg * 96 + f_row) * 128 + ch_lo};
    const int lx[2] = {rows0 * 128 + ch_hi, rows0 * 128 + ch_lo};
    auto region = [&](int off) { asm volatile("" : "+s"(off)); return off; };

    f32x4 acc[3][12], pacc[3][4];
#pragma unroll
    for (int rf = 0; rf < 3; ++rf) {
#pragma unroll
        for (int c = 0; c < 12; ++c) acc[rf][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) pacc[rf][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    u32x4 fw[2][12], fx[2][6];  // [register set]: weight fragments (hi 0.., lo n..); x fragments of an A-step / G fragments of a B-step pair (the two never overlap in a set)
#define gb fx

    // read k (0 ..) of step t into its register set; returns false when the step has no such read
    auto load_one = [&](int t, int k) -> bool {
        const int set = t & 1;
        const int so = OFF_RING + (t & 3) * SLOTB;
        if (t < NA) {
            if (k < 8) { const int nf = k & 3, lo = k >> 2; fw[set][lo * 4 + nf] = rd(lw_a[lo] + region(so), nf); return true; }
            if (k < 14) { const int rf = (k - 8) % 3, lo = (k - 8) / 3; fx[set][lo * 3 + rf] = rd(lx[lo] + region(so + X_OFF), rf); return true; }
            return false;
        }
        const int sb = t - NA;
        if (k < 12) { const int nf = k % 6, lo = k / 6; fw[set][lo * 6 + nf] = rd(lw_b[lo] + region(so), nf); return true; }
        if (k < 18 && (sb & 1) == 0) { const int rf = (k - 12) % 3, lo = (k - 12) / 3; gb[(sb >> 1) & 1][lo * 3 + rf] = rd(lx[lo] + region((sb >> 1) * G_KB), rf); return true; }
        return false;
    };
    // MFMA k of step t (A: 36, B: 54): three products per (rf, nf) pair, hi x hi, lo x hi, hi x lo as three sweeps
    auto mfma_one = [&](int t, int k) {
        const int set = t & 1;
        if (t < NA) {
            const int sweep = k / 12, r = k % 12, rf = r / 4, nf = r % 4;
            const u32x4& w = fw[set][(sweep == 1 ? 4 : 0) + nf];
            const u32x4& x = fx[set][(sweep == 2 ? 3 : 0) + rf];
            pacc[rf][nf] = mma(w, x, pacc[rf][nf]);
        } else {
            const int sb = t - NA, half = sb & 1, g = (sb >> 1) & 1;
            const int sweep = k / 18, r = k % 18, rf = r / 6, nf = r % 6;
            const u32x4& w = fw[set][(sweep == 1 ? 6 : 0) + nf];
            const u32x4& x = gb[g][(sweep == 2 ? 3 : 0) + rf];
            acc[rf][half * 6 + nf] = mma(w, x, acc[rf][half * 6 + nf]);
        }
    };
    auto waitn = [](int n) {
        switch (n) {
            case 6: WAITVM(6); break;
            case 7: WAITVM(7); break;
            default: WAITVM(0); break;
        }
    };

    int cb = (int)(blockIdx.x & 7) * CHUNK_BYTES;  // byte offset of the current chunk in the weight stream (rotated per XCD)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int u = 0; u < n_pieces(t); ++u) issue_piece(cb, t, u);
    WAITVM(0);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int k = 0; k < 18; ++k) load_one(0, k);
    for (int ci = 0; ci < NCH; ++ci) {
        int cb_next = cb + CHUNK_BYTES;
        if (cb_next == NCH * CHUNK_BYTES) cb_next = 0;
#pragma unroll
        for (int t = 0; t < STEPS; ++t) {
            const int t1 = (t + 1) % STEPS, t2 = (t + 2) % STEPS, t3 = (t + 3) % STEPS;
            const int c3 = t + 3 < STEPS ? cb : cb_next;
            __builtin_amdgcn_sched_barrier(0);
            waitn(n_pieces(t2));  // step t + 1 has landed (own pieces of step t + 2 may be outstanding), the fragments of step t are in
            if (!(ABL & 16)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const int nm = t < NA ? 36 : 54;
            if (SPREAD) {
#pragma unroll
                for (int k = 0; k < nm; ++k) {
                    mfma_one(t, k);
                    __builtin_amdgcn_sched_barrier(0);
                    // behind every MFMA one LDS read of step t + 1 (even k) or one DMA piece of step t + 3 (odd k)
                    if ((k & 1) == 0) { if (k / 2 < 18) load_one(t1, k / 2); }
                    else if (k / 2 < n_pieces(t3)) issue_piece(c3, t3, k / 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int u = 0; u < n_pieces(t3); ++u) issue_piece(c3, t3, u);
#pragma unroll
                for (int k = 0; k < 18; ++k) load_one(t1, k);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < nm; ++k) mfma_one(t, k);
            }
        }
        cb = cb_next;
    }
    __builtin_amdgcn_s_waitcnt((7 << 4) | (0 << 8) | 0);
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rf = 0; rf < 3; ++rf) {
#pragma unroll
        for (int c = 0; c < 12; ++c) sum += acc[rf][c];
#pragma unroll
        for (int c = 0; c < 4; ++c) sum += pacc[rf][c];
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {  // (keep every fragment register live)
#pragma unroll
        for (int i = 0; i < 12; ++i) sum[0] += __builtin_bit_cast(float, fw[s][i][0]);
    }
    reinterpret_cast<f32x4*>(out)[(size_t)blockIdx.x * THREADS + tid] = sum;
}

int main() {
    const size_t wbytes = (size_t)NCH * CHUNK_BYTES, M = 256 * BM, hbytes = M * E * 4;
    char *w, *h;
    float* out;
    hipMalloc(&w, wbytes);
    hipMalloc(&h, hbytes);
    hipMalloc(&out, 256 * THREADS * 16);
    std::vector<unsigned short> hw(wbytes / 2), hh(hbytes / 2);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x2c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
    for (size_t i = 0; i < hh.size(); ++i) hh[i] = 0x3800 + (rand() & 0x7ff) + ((rand() & 1) << 15);
    hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice);
    hipMemcpy(h, hh.data(), hbytes, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(ffn4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(ffn4_kernel, dim3(256), dim3(THREADS), LDS, 0, w, (unsigned)wbytes, h, (unsigned)hbytes, out);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(ffn4_kernel, dim3(256), dim3(THREADS), LDS, 0, w, (unsigned)wbytes, h, (unsigned)hbytes, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms / 20 < best) best = ms / 20;
    }
    hipError_t err = hipGetLastError();
    printf("ffn4 SPREAD=%d ABL=%d: %.1f us per launch (240 steps: %.0f ns per step), err=%s\n", SPREAD, ABL, best * 1e3, best * 1e6 / 240, hipGetErrorString(err));
    return 0;
}
