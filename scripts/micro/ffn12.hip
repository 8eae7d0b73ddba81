// dev micro-benchmark: the step loop of the f16x3 FFN kernel (pp_ffn_split.hip) in its simplest form - all waves in the same
// phase, one barrier per step, DMA three steps ahead on the ring of four 28 KiB slots - with the workgroup organised as
//   WAVES = 8   2 row groups x 4 column groups, 256 registers (what the product kernel has; there with role-alternating halves)
//   WAVES = 12  3 row groups x 4 column groups, 168 registers: three waves per SIMD
// Same LDS map (G 48 KiB + ring 112 KiB), same bytes streamed, same MFMA count per SIMD. Question: what does a third wave per
// SIMD buy over hand-scheduling two?
//   hipcc -O3 --offload-arch=gfx950 -DWAVES=12 scripts/micro/ffn12.hip -o scripts/micro/build/ffn12_12
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#ifndef WAVES
#define WAVES 12
#endif
#ifndef ROT
#define ROT 0
#endif
#ifndef OPAQUE
#define OPAQUE 0
#endif
#ifndef PIPE
#define PIPE 0  // 1: fragments of step t + 1 read under the MFMAs of step t (two register sets)
#endif
#ifndef ABL  // timing-only ablations: 1 no DMA traffic, 2 no fragment reads, 4 no MFMA, 8 no DMA instructions, 16 no barrier (PIPE form)
#define ABL 0
#endif
constexpr int BM = 96, E = 384, CHUNK = 128, NCH = 12;
constexpr int THREADS = WAVES * 64;
constexpr int RG = WAVES / 4;              // row groups (2 or 3), 4 column groups
constexpr int RF = BM / 16 / RG;           // row fragments per wave (3 or 2)
constexpr int G_KB = BM * 128, OFF_RING = 4 * G_KB, SLOTB = 28 * 1024, LDS = OFF_RING + 4 * SLOTB;
constexpr int NA = 12, NB = 8, STEPS = NA + NB;
constexpr int A_BLOCK = CHUNK * 128, B_BLOCK = 192 * 128, B_PART = NA * A_BLOCK, CHUNK_BYTES = B_PART + NB * B_BLOCK;
constexpr int X_OFF = 16 * 1024;

// DMA pieces per wave: A-step 16 W1 + 12 x, B-step 24 W2
__host__ __device__ constexpr int n_a(int wv) { return WAVES == 12 ? (wv < 4 ? 3 : 2) : (wv < 4 ? 4 : 3); }
__host__ __device__ constexpr int n_b(int) { return 24 / WAVES; }

#define WAITVM(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (7 << 4) | (0 << 8) | (((N) >> 4) << 14))

__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    if (ABL & 4) return c;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(THREADS, WAVES / 4) void ffn12_kernel(const char* __restrict__ wpack, unsigned w_bytes, const char* __restrict__ h,
                                                                   unsigned h_bytes, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wv >> 2, cg = wv & 3;
    const int f_row = lane & 15, f_kg = lane >> 4;
    const int m0 = blockIdx.x * BM;
    char* const ring = smem + OFF_RING;
    for (int i = tid; i < 4 * G_KB / 4; i += THREADS) reinterpret_cast<unsigned*>(smem)[i] = 0x2c003c00u + ((i * 2654435761u) >> 20 & 0x007f007fu);
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wpack), 0, (ABL & 1) ? 0u : w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(h), 0, (ABL & 1) ? 0u : h_bytes, 0x00020000);
    const unsigned v_w = (unsigned)lane * 16u;
    const int x_l = lane >> 3;
    // x piece q (0..11): rows 8 q + l
    auto v_x = [&](int q) { return (unsigned)(m0 + 8 * q + x_l) * (unsigned)(E * 4) + (unsigned)(((lane & 7) ^ x_l) << 4); };
    const int c_rot = (int)(blockIdx.x & 7);
    auto chunk_of = [&](int i) { const int c = (i + c_rot) % NCH; return c; };
    auto issue = [&](int ci, int t) {  // step t (0..19) of chunk ci into slot (t & 3)
        if (ABL & 8) return;  // no DMA instructions at all
        char* dst = ring + (t & 3) * SLOTB;
        const int base = chunk_of(ci % NCH) * CHUNK_BYTES;
        if (t < NA) {
#if ROT
            // the workgroups of an XCD (rank blockIdx.x >> 3) walk the k-blocks of a chunk in different rotations: at any moment
            // they ask the L2 for different lines instead of the same 1 KiB piece
            int tk = t + (int)((blockIdx.x >> 3) % NA);
            tk = tk >= NA ? tk - NA : tk;
#else
            const int tk = t;
#endif
            const int blk = base + tk * A_BLOCK;
            if (WAVES == 12) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + wv * 1024), 16, v_w, blk + wv * 1024, 0, 0);
                if (wv < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + (12 + wv) * 1024), 16, v_w, blk + (12 + wv) * 1024, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr_t)(dst + X_OFF + wv * 1024), 16, v_x(wv), tk * 128, 0, 0);
            } else {
                if (wv < 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + (4 * wv + u) * 1024), 16, v_w, blk + (4 * wv + u) * 1024, 0, 0);
                } else {
#pragma unroll
                    for (int u = 0; u < 3; ++u) __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr_t)(dst + X_OFF + (3 * (wv - 4) + u) * 1024), 16, v_x(3 * (wv - 4) + u), tk * 128, 0, 0);
                }
            }
        } else {
            const int blk = base + B_PART + (t - NA) * B_BLOCK;
#pragma unroll
            for (int u = 0; u < 24 / WAVES; ++u) {
                const int q = (24 / WAVES) * wv + u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + q * 1024), 16, v_w, blk + q * 1024, 0, 0);
            }
        }
    };
    const int sw = f_row & 7;
    const int ch_hi = (f_kg ^ sw) << 4, ch_lo = ((4 + f_kg) ^ sw) << 4;
    const int rows0 = rg * (16 * RF) + f_row;
    auto rd = [&](int off) -> u32x4 {
        if (ABL & 2) return u32x4{(unsigned)off, 1u, 2u, 3u};
#if PIPE == 2
        // as asm: the compiler puts `s_waitcnt vmcnt(0)` in front of an ordinary LDS read that follows an LDS-DMA instruction
        // (it cannot tell the ring slots apart); the step's own `lgkmcnt(0)` at its top covers these reads one step later
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(off));
        return v;
#else
        return *reinterpret_cast<const u32x4*>(smem + off);
#endif
    };

    f32x4 acc[RF][6], pacc[RF][2];
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) {
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[rf][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        pacc[rf][0] = pacc[rf][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    u32x4 bgh[RF], bgl[RF];

    issue(0, 0);
    issue(0, 1);
    issue(0, 2);
#if PIPE
    // software-pipelined form: the fragments of step t + 1 are read (into the other register set) under the MFMAs of step t
    u32x4 fw[2][6], fx[2][2 * RF], gb[2][2 * RF];
    auto load = [&](int t, int set) {  // t in 0..19
        int so = OFF_RING + (t & 3) * SLOTB;
#if OPAQUE
        asm volatile("" : "+s"(so));  // (as pp_ffn_split.hip's slot_off(): the slot offset is opaque to the compiler)
#endif
        if (t < NA) {
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                fw[set][nf] = rd(so + (cg * 32 + nf * 16 + f_row) * 128 + ch_hi);
                fw[set][2 + nf] = rd(so + (cg * 32 + nf * 16 + f_row) * 128 + ch_lo);
            }
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) {
                fx[set][rf] = rd(so + X_OFF + (rows0 + rf * 16) * 128 + ch_hi);
                fx[set][RF + rf] = rd(so + X_OFF + (rows0 + rf * 16) * 128 + ch_lo);
            }
        } else {
            const int sb = t - NA;
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) {
                fw[set][nf] = rd(so + (cg * 48 + nf * 16 + f_row) * 128 + ch_hi);
                fw[set][3 + nf] = rd(so + (cg * 48 + nf * 16 + f_row) * 128 + ch_lo);
            }
            if ((sb & 1) == 0) {
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) {
                    gb[(sb >> 1) & 1][rf] = rd((sb >> 1) * G_KB + (rows0 + rf * 16) * 128 + ch_hi);
                    gb[(sb >> 1) & 1][RF + rf] = rd((sb >> 1) * G_KB + (rows0 + rf * 16) * 128 + ch_lo);
                }
            }
        }
    };
    auto compute = [&](int t, int set) {
        if (t < NA) {
#pragma unroll
            for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) pacc[rf][nf] = mma(fw[set][nf], fx[set][rf], pacc[rf][nf]);
#pragma unroll
            for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) pacc[rf][nf] = mma(fw[set][2 + nf], fx[set][rf], pacc[rf][nf]);
#pragma unroll
            for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) pacc[rf][nf] = mma(fw[set][nf], fx[set][RF + rf], pacc[rf][nf]);
        } else {
            const int sb = t - NA, half = sb & 1, g = (sb >> 1) & 1;
#pragma unroll
            for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                for (int nf = 0; nf < 3; ++nf) acc[rf][half * 3 + nf] = mma(fw[set][nf], gb[g][rf], acc[rf][half * 3 + nf]);
#pragma unroll
            for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                for (int nf = 0; nf < 3; ++nf) acc[rf][half * 3 + nf] = mma(fw[set][3 + nf], gb[g][rf], acc[rf][half * 3 + nf]);
#pragma unroll
            for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                for (int nf = 0; nf < 3; ++nf) acc[rf][half * 3 + nf] = mma(fw[set][nf], gb[g][RF + rf], acc[rf][half * 3 + nf]);
        }
    };
#define WAITVM_ONLY(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (7 << 4) | (0 << 8) | (((N) >> 4) << 14))  // vmcnt(N) lgkmcnt(0): the asm reads of the previous step (this step's fragments) are in
    auto waitn2 = [](int n) {
        switch (n) {
            case 2: WAITVM_ONLY(2); break;
            case 3: WAITVM_ONLY(3); break;
            case 4: WAITVM_ONLY(4); break;
            default: WAITVM_ONLY(0); break;
        }
    };
    WAITVM(0);
    __builtin_amdgcn_s_barrier();
    load(0, 0);
    for (int ci = 0; ci < NCH; ++ci) {
#pragma unroll
        for (int t = 0; t < STEPS; ++t) {
            const int t2 = (t + 2) % STEPS;
            __builtin_amdgcn_sched_barrier(0);
            if (wv < 4) waitn2(t2 < NA ? n_a(0) : n_b(0)); else waitn2(t2 < NA ? n_a(4) : n_b(4));  // step t + 1 has landed
            if (!(ABL & 16)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            { const int t3 = t + 3; if (t3 < STEPS) issue(ci, t3); else issue(ci + 1, t3 - STEPS); }
            load((t + 1) % STEPS, (t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);  // the reads go out first, the MFMAs of this step cover their latency
            compute(t, t & 1);
        }
    }
#else
    for (int ci = 0; ci < NCH; ++ci) {
#pragma unroll
        for (int t = 0; t < STEPS; ++t) {
            // own pieces of steps t + 1, t + 2 may be outstanding
            const int t1 = (t + 1) % STEPS, t2 = (t + 2) % STEPS;
            __builtin_amdgcn_sched_barrier(0);
            auto waitn = [](int n) {  // (the builtin wants a literal; after unrolling n is a constant and the switch folds)
                switch (n) {
                    case 4: WAITVM(4); break;
                    case 5: WAITVM(5); break;
                    case 6: WAITVM(6); break;
                    case 7: WAITVM(7); break;
                    case 8: WAITVM(8); break;
                    default: WAITVM(0); break;
                }
            };
            if (wv < 4) waitn((t1 < NA ? n_a(0) : n_b(0)) + (t2 < NA ? n_a(0) : n_b(0)));
            else waitn((t1 < NA ? n_a(4) : n_b(4)) + (t2 < NA ? n_a(4) : n_b(4)));
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            { const int t3 = t + 3; if (t3 < STEPS) issue(ci, t3); else issue(ci + 1, t3 - STEPS); }
            const int so = OFF_RING + (t & 3) * SLOTB;
            if (t < NA) {
                u32x4 wh[2], wl[2], xh[RF], xl[RF];
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) {
                    wh[nf] = rd(so + (cg * 32 + nf * 16 + f_row) * 128 + ch_hi);
                    wl[nf] = rd(so + (cg * 32 + nf * 16 + f_row) * 128 + ch_lo);
                }
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) {
                    xh[rf] = rd(so + X_OFF + (rows0 + rf * 16) * 128 + ch_hi);
                    xl[rf] = rd(so + X_OFF + (rows0 + rf * 16) * 128 + ch_lo);
                }
#pragma unroll
                for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                    for (int nf = 0; nf < 2; ++nf) pacc[rf][nf] = mma(wh[nf], xh[rf], pacc[rf][nf]);
#pragma unroll
                for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                    for (int nf = 0; nf < 2; ++nf) pacc[rf][nf] = mma(wl[nf], xh[rf], pacc[rf][nf]);
#pragma unroll
                for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                    for (int nf = 0; nf < 2; ++nf) pacc[rf][nf] = mma(wh[nf], xl[rf], pacc[rf][nf]);
            } else {
                const int sb = t - NA, half = sb & 1;
                u32x4 wh[3], wl[3];
#pragma unroll
                for (int nf = 0; nf < 3; ++nf) {
                    wh[nf] = rd(so + (cg * 48 + nf * 16 + f_row) * 128 + ch_hi);
                    wl[nf] = rd(so + (cg * 48 + nf * 16 + f_row) * 128 + ch_lo);
                }
                if (half == 0) {
#pragma unroll
                    for (int rf = 0; rf < RF; ++rf) {
                        bgh[rf] = rd((sb >> 1) * G_KB + (rows0 + rf * 16) * 128 + ch_hi);
                        bgl[rf] = rd((sb >> 1) * G_KB + (rows0 + rf * 16) * 128 + ch_lo);
                    }
                }
#pragma unroll
                for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                    for (int nf = 0; nf < 3; ++nf) acc[rf][half * 3 + nf] = mma(wh[nf], bgh[rf], acc[rf][half * 3 + nf]);
#pragma unroll
                for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                    for (int nf = 0; nf < 3; ++nf) acc[rf][half * 3 + nf] = mma(wl[nf], bgh[rf], acc[rf][half * 3 + nf]);
#pragma unroll
                for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                    for (int nf = 0; nf < 3; ++nf) acc[rf][half * 3 + nf] = mma(wh[nf], bgl[rf], acc[rf][half * 3 + nf]);
            }
        }
    }
#endif
    __builtin_amdgcn_s_waitcnt((7 << 4) | (0 << 8) | 0);
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) {
#pragma unroll
        for (int c = 0; c < 6; ++c) sum += acc[rf][c];
        sum += pacc[rf][0] + pacc[rf][1];
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
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x2c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);  // ~N-like fp16 in [-0.1, 0.1]
    for (size_t i = 0; i < hh.size(); ++i) hh[i] = 0x3800 + (rand() & 0x7ff) + ((rand() & 1) << 15);
    hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice);
    hipMemcpy(h, hh.data(), hbytes, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(ffn12_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(ffn12_kernel, dim3(256), dim3(THREADS), LDS, 0, w, (unsigned)wbytes, h, (unsigned)hbytes, out);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(ffn12_kernel, dim3(256), dim3(THREADS), LDS, 0, w, (unsigned)wbytes, h, (unsigned)hbytes, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms / 20 < best) best = ms / 20;
    }
    hipError_t err = hipGetLastError();
    printf("WAVES=%d PIPE=%d ABL=%d: %.1f us per launch (240 steps: %.0f ns per step), err=%s\n", WAVES, PIPE, ABL, best * 1e3, best * 1e6 / 240, hipGetErrorString(err));
    return 0;
}
