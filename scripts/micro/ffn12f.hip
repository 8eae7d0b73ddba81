// dev micro-benchmark (round 4): the twelve-wave FFN step loop of ffn12d.hip (MODE 1, GELU 1) WITHOUT workgroup barriers in the loop:
// a ring of THREE 28 KiB slots (the fourth slot's LDS holds the flags), per-slot counters in LDS:
//   ready[slot]  += 1 by each of the four DMA waves when its pieces of the step have landed      (computing waves poll for 4 per use)
//   done[slot]   += 1 by each of the eight computing waves once its fragment reads are issued     (DMA waves poll for 8 per use; LDS
//                   executes a wave's operations in order, so the add lands behind the reads)
//   gw[rg] / gr[rg]  the G tile of a row group: written (4 waves) / read to the end (4 waves)
// Nothing keeps the waves in phase: a computing wave starts a step as soon as its slot is ready. Timing only (dummy data).
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize scripts/micro/ffn12f.hip -o scripts/micro/build/ffn12f
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#ifndef NSLOT
#define NSLOT 3
#endif
#ifndef ALT
#define ALT 0  // 1 (with SYNC 0): role-alternating computing waves - the two waves of a SIMD (row groups 0 / 1) are never in the same kind of
               // segment: two barriers per step, rg 0 reads its fragments while rg 1 multiplies the previous step's, then the other way round
#endif
#ifndef SYNC
#define SYNC 1  // 1 flags, 0 one barrier per step (the form of ffn12d.hip, here with NSLOT slots)
#endif
constexpr int BM = 96, E = 384, CHUNK = 128, NCH = 12;
constexpr int CW = 8, WAVES = 12, THREADS = WAVES * 64, RF = 3;
constexpr int G_KB = BM * 128, OFF_RING = 4 * G_KB, SLOTB = 28 * 1024;
constexpr int OFF_FLAGS = OFF_RING + NSLOT * SLOTB;
constexpr int LDS = OFF_FLAGS + (NSLOT < 4 ? 256 : 0);
constexpr int NA = 12, NB = 8, STEPS = NA + NB, TOTAL = NCH * STEPS;
constexpr int A_BLOCK = CHUNK * 128, B_BLOCK = 192 * 128, B_PART = NA * A_BLOCK, CHUNK_BYTES = B_PART + NB * B_BLOCK;
constexpr int X_OFF = 16 * 1024;
constexpr int DEPTH = NSLOT - 1;  // steps in flight ahead of the one being read
static_assert(LDS <= 160 * 1024, "LDS");

#define WAITVM_ONLY(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | (7 << 4) | (15 << 8) | (((N) >> 4) << 14))

__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void wait_vm_n(int n) {
    switch (n) {
        case 0: WAITVM_ONLY(0); break; case 6: WAITVM_ONLY(6); break; case 7: WAITVM_ONLY(7); break;
        case 12: WAITVM_ONLY(12); break; case 13: WAITVM_ONLY(13); break; case 14: WAITVM_ONLY(14); break;
        default: WAITVM_ONLY(0); break;
    }
}
__device__ __forceinline__ void mem_fence_compiler() { asm volatile("" ::: "memory"); }
// spin until *flag (LDS) has reached target
__device__ __forceinline__ void poll(const volatile unsigned* flag, unsigned target) {
    while (true) {
        const unsigned v = __builtin_amdgcn_readfirstlane(*flag);
        if ((int)(v - target) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
    mem_fence_compiler();
}
__device__ __forceinline__ void signal(unsigned* flag, int lane) {
    mem_fence_compiler();
    if (lane == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    mem_fence_compiler();
}

__global__ __launch_bounds__(THREADS) void ffn12f_kernel(const char* __restrict__ wpack, unsigned w_bytes, const char* __restrict__ h, unsigned h_bytes,
                                                         float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool dma_wave = wv >= CW;
    const int rg = (wv >> 2) & 1, cg = wv & 3;
    const int f_row = lane & 15, f_kg = lane >> 4;
    const int m0 = blockIdx.x * BM;
    char* const ring = smem + OFF_RING;
    unsigned* const flags = reinterpret_cast<unsigned*>(smem + OFF_FLAGS);
    unsigned* const f_ready = flags;        // [NSLOT]
    unsigned* const f_done = flags + 4;     // [NSLOT]
    unsigned* const f_gw = flags + 8;       // [2]
    unsigned* const f_gr = flags + 10;      // [2]
    for (int i = tid; i < 4 * G_KB / 4; i += THREADS) reinterpret_cast<unsigned*>(smem)[i] = 0x2c003c00u + ((i * 2654435761u) >> 20 & 0x007f007fu);
    if (SYNC && tid < 16) flags[tid] = 0;
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wpack), 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(h), 0, h_bytes, 0x00020000);
    const int c_rot = (int)(blockIdx.x & 7);

    if (dma_wave) {
        const int d = wv - CW;
        const unsigned v_w = (unsigned)lane * 16u;
        const int x_l = lane >> 3;
        const unsigned v_x = (unsigned)(m0 + x_l) * (unsigned)(E * 4) + (unsigned)(((lane & 7) ^ x_l) << 4);
        // step g (global index) -> chunk g / 20, t = g % 20, slot g % NSLOT
        auto issue = [&](int ci, int t, int slot) {
            char* dst = ring + slot * SLOTB;
            const int base = ((ci + c_rot) % NCH) * CHUNK_BYTES;
            if (t < NA) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = d + 4 * u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + q * 1024), 16, v_w, base + t * A_BLOCK + q * 1024, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int q = d + 4 * u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr_t)(dst + X_OFF + q * 1024), 16, v_x, t * 128 + q * 8 * E * 4, 0, 0);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int q = d + 4 * u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + q * 1024), 16, v_w, base + B_PART + (t - NA) * B_BLOCK + q * 1024, 0, 0);
                }
            }
        };
        int slot_i = 0;  // slot of the step to issue next
#pragma unroll
        for (int t = 0; t < DEPTH; ++t) { issue(0, t, slot_i); slot_i = slot_i + 1 == NSLOT ? 0 : slot_i + 1; }
        int slot_g = 0;
        for (int ci = 0; ci < NCH; ++ci) {
#pragma unroll
            for (int t = 0; t < STEPS; ++t) {
                const int g = ci * STEPS + t;
                __builtin_amdgcn_sched_barrier(0);
                // step g has landed when only this wave's pieces of the DEPTH - 1 steps behind it are out
                int allowed = 0;
#pragma unroll
                for (int a = 1; a < DEPTH; ++a) allowed += ((t + a) % STEPS) < NA ? 7 : 6;
                wait_vm_n(allowed);
                if (SYNC) signal(f_ready + slot_g, lane);
                else __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
#if ALT
                __builtin_amdgcn_s_barrier();  // the second barrier of the step (rg 0: reads | MFMAs, rg 1: MFMAs of the previous step | reads)
#endif
                // step g + DEPTH goes into the slot of step g - 1: every computing wave must be past its reads of that step
                if (SYNC && g >= 1) {
                    const int sp = slot_g == 0 ? NSLOT - 1 : slot_g - 1;
                    poll(f_done + sp, 8u * (unsigned)((g - 1) / NSLOT + 1));
                }
                const int tn = t + DEPTH;
                if (g + DEPTH < TOTAL) { if (tn < STEPS) issue(ci, tn, slot_i); else issue(ci + 1, tn - STEPS, slot_i); }
                else {  // keep the counts: empty descriptor
                    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wpack), 0, 0, 0x00020000);
                    const int n = (tn % STEPS) < NA ? 7 : 6;
                    for (int u = 0; u < n; ++u) __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (lds_ptr_t)(ring + slot_i * SLOTB + (d + 4 * u) * 1024), 16, v_w, 0, 0, 0);
                }
                slot_i = slot_i + 1 == NSLOT ? 0 : slot_i + 1;
                slot_g = slot_g + 1 == NSLOT ? 0 : slot_g + 1;
            }
        }
        WAITVM_ONLY(0);
#if ALT
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
#endif
        __syncthreads();
        return;
    }

    // ---------------- computing waves
    const int sw = f_row & 7;
    const int lane_hi = f_row * 128 + ((f_kg ^ sw) << 4), lane_lo = f_row * 128 + (((4 + f_kg) ^ sw) << 4);
    const int rows0 = rg * 48 + f_row;
    auto opq = [](int v) { asm volatile("" : "+s"(v)); return v; };
    auto rd = [&](int lane_off, int uni, int imm) -> u32x4 { return *reinterpret_cast<const u32x4*>(smem + (lane_off + uni) + imm); };
    f32x4 acc[RF][6], pacc[RF][2];
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) {
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[rf][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        pacc[rf][0] = pacc[rf][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    u32x4 bgh[RF], bgl[RF];
#if ALT
    {
        u32x4 wh[3], wl[3], xh[RF], xl[RF];
        auto Lseg = [&](int t, int slot) {  // all fragment reads of step t
            const int so = opq(slot * SLOTB);
            if (t < NA) {
                const int ua = opq(OFF_RING + cg * 32 * 128 + so), ux = opq(OFF_RING + X_OFF + rg * 48 * 128 + so);
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) wh[nf] = rd(lane_hi, ua, nf * 2048);
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) xh[rf] = rd(lane_hi, ux, rf * 2048);
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) wl[nf] = rd(lane_lo, ua, nf * 2048);
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) xl[rf] = rd(lane_lo, ux, rf * 2048);
            } else {
                const int sb = t - NA;
                const int ub = opq(OFF_RING + cg * 48 * 128 + so);
#pragma unroll
                for (int nf = 0; nf < 3; ++nf) wh[nf] = rd(lane_hi, ub, nf * 2048);
                if ((sb & 1) == 0) {
                    const int ug = opq(rg * 48 * 128 + (sb >> 1) * G_KB);
#pragma unroll
                    for (int rf = 0; rf < RF; ++rf) bgh[rf] = rd(lane_hi, ug, rf * 2048);
                }
#pragma unroll
                for (int nf = 0; nf < 3; ++nf) wl[nf] = rd(lane_lo, ub, nf * 2048);
                if ((sb & 1) == 0) {
                    const int ug = opq(rg * 48 * 128 + (sb >> 1) * G_KB);
#pragma unroll
                    for (int rf = 0; rf < RF; ++rf) bgl[rf] = rd(lane_lo, ug, rf * 2048);
                }
            }
        };
        auto Cseg = [&](int t) {  // the MFMAs of step t (+ the chunk's GELU behind its last A-step)
            if (t < NA) {
                if (t == 0) {
#pragma unroll
                    for (int rf = 0; rf < RF; ++rf) pacc[rf][0] = pacc[rf][1] = f32x4{0.f, 0.f, 0.f, 0.f};
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
                if (t == NA - 1) {
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                        for (int nf = 0; nf < 2; ++nf) {
                            h4 hv, lv;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float x = pacc[rf][nf][q];
                                const float z = fabsf(x) * 0.70710678118654752440f;
                                const float tt = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
                                float qq = __builtin_fmaf(tt, 1.061405429f, -1.453152027f);
                                qq = __builtin_fmaf(tt, qq, 1.421413741f);
                                qq = __builtin_fmaf(tt, qq, -0.284496736f);
                                qq = __builtin_fmaf(tt, qq, 0.254829592f);
                                const float e = __builtin_amdgcn_exp2f(-(z * z) * 1.44269504088896340736f);
                                const float ez = tt * qq * e;
                                float gv = 0.5f * x * (x < 0.f ? ez : 2.0f - ez);
                                asm("" : "+v"(gv));
                                hv[q] = (_Float16)gv;
                                lv[q] = (_Float16)(gv - (float)hv[q]);
                            }
                            char* gs = smem + cg * G_KB + (rows0 + rf * 16) * 128 + (f_kg & 1) * 8;
                            const int c = 2 * nf + (f_kg >> 1);
                            *reinterpret_cast<h4*>(gs + ((c ^ sw) << 4)) = hv;
                            *reinterpret_cast<h4*>(gs + (((4 + c) ^ sw) << 4)) = lv;
                        }
                }
            } else {
                const int half = (t - NA) & 1;
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
        };
        auto bar = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt((63 & 15) | (7 << 4) | (0 << 8) | ((63 >> 4) << 14));  // lgkmcnt(0): fragments in / G writes out
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        int slot = 0;
        if (rg == 0) {
            for (int ci = 0; ci < NCH; ++ci) {
#pragma clang loop unroll(full)
                for (int t = 0; t < STEPS; ++t) {
                    bar();
                    Lseg(t, slot);
                    bar();
                    Cseg(t);
                    slot = slot + 1 == NSLOT ? 0 : slot + 1;
                }
            }
            bar();
            bar();
        } else {
            bar();
            bar();
            Lseg(0, 0);
            slot = 1;
            for (int ci = 0; ci < NCH; ++ci) {
#pragma clang loop unroll(full)
                for (int t = 1; t <= STEPS; ++t) {  // step t of chunk ci (t = 20: step 0 of the next chunk)
                    bar();
                    Cseg(t - 1);
                    bar();
                    if (t < STEPS || ci + 1 < NCH) Lseg(t % STEPS, slot);
                    slot = slot + 1 == NSLOT ? 0 : slot + 1;
                }
            }
            // (barrier count: 2 + 2 * 20 * NCH = rg 0's 2 * 20 * NCH + 1 ... + 1: one more below)
        }
    }
#else
    int slot = 0;
    unsigned pre_flag = 0;
    for (int ci = 0; ci < NCH; ++ci) {
#pragma unroll
        for (int rf = 0; rf < RF; ++rf) pacc[rf][0] = pacc[rf][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma clang loop unroll(full)
        for (int t = 0; t < STEPS; ++t) {
            const int g = ci * STEPS + t;
            __builtin_amdgcn_sched_barrier(0);
            if (SYNC) {  // the flag of this step was read during the previous one: spin only if it was not up yet
                if (g == 0 || (int)(__builtin_amdgcn_readfirstlane(pre_flag) - 4u * (unsigned)(g / NSLOT + 1)) < 0) poll(f_ready + slot, 4u * (unsigned)(g / NSLOT + 1));
            } else __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const int so = opq(slot * SLOTB);
            if (t < NA) {
                u32x4 wh[2], wl[2], xh[RF], xl[RF];
                const int ua = opq(OFF_RING + cg * 32 * 128 + so), ux = opq(OFF_RING + X_OFF + rg * 48 * 128 + so);
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) wh[nf] = rd(lane_hi, ua, nf * 2048);
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) xh[rf] = rd(lane_hi, ux, rf * 2048);
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) wl[nf] = rd(lane_lo, ua, nf * 2048);
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) xl[rf] = rd(lane_lo, ux, rf * 2048);
                if (SYNC) {
                    signal(f_done + slot, lane);
                    pre_flag = *reinterpret_cast<const volatile unsigned*>(f_ready + (slot + 1 == NSLOT ? 0 : slot + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
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
                if (t == NA - 1) {
                    // the G tile of this row group is free once its four waves have read the previous chunk's to the end
                    if (SYNC) poll(f_gr + rg, 4u * (unsigned)ci);
                    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                        for (int nf = 0; nf < 2; ++nf) {
                            h4 hv, lv;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float x = pacc[rf][nf][q];
                                const float z = fabsf(x) * 0.70710678118654752440f;
                                const float tt = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
                                float qq = __builtin_fmaf(tt, 1.061405429f, -1.453152027f);
                                qq = __builtin_fmaf(tt, qq, 1.421413741f);
                                qq = __builtin_fmaf(tt, qq, -0.284496736f);
                                qq = __builtin_fmaf(tt, qq, 0.254829592f);
                                const float e = __builtin_amdgcn_exp2f(-(z * z) * 1.44269504088896340736f);
                                const float ez = tt * qq * e;
                                float gv = 0.5f * x * (x < 0.f ? ez : 2.0f - ez);
                                asm("" : "+v"(gv));
                                hv[q] = (_Float16)gv;
                                lv[q] = (_Float16)(gv - (float)hv[q]);
                            }
                            char* gs = smem + cg * G_KB + (rows0 + rf * 16) * 128 + (f_kg & 1) * 8;
                            const int c = 2 * nf + (f_kg >> 1);
                            *reinterpret_cast<h4*>(gs + ((c ^ sw) << 4)) = hv;
                            *reinterpret_cast<h4*>(gs + (((4 + c) ^ sw) << 4)) = lv;
                        }
                    if (SYNC) signal(f_gw + rg, lane);
                }
            } else {
                const int sb = t - NA, half = sb & 1;
                u32x4 wh[3], wl[3];
                const int ub = opq(OFF_RING + cg * 48 * 128 + so);
                if (SYNC && sb == 0) poll(f_gw + rg, 4u * (unsigned)(ci + 1));
#pragma unroll
                for (int nf = 0; nf < 3; ++nf) wh[nf] = rd(lane_hi, ub, nf * 2048);
                if (half == 0) {
                    const int ug = opq(rg * 48 * 128 + (sb >> 1) * G_KB);
#pragma unroll
                    for (int rf = 0; rf < RF; ++rf) bgh[rf] = rd(lane_hi, ug, rf * 2048);
#pragma unroll
                    for (int nf = 0; nf < 3; ++nf) wl[nf] = rd(lane_lo, ub, nf * 2048);
#pragma unroll
                    for (int rf = 0; rf < RF; ++rf) bgl[rf] = rd(lane_lo, ug, rf * 2048);
                    if (SYNC && sb == NB - 2) signal(f_gr + rg, lane);
                } else {
#pragma unroll
                    for (int nf = 0; nf < 3; ++nf) wl[nf] = rd(lane_lo, ub, nf * 2048);
                }
                if (SYNC) {
                    signal(f_done + slot, lane);
                    pre_flag = *reinterpret_cast<const volatile unsigned*>(f_ready + (slot + 1 == NSLOT ? 0 : slot + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
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
            slot = slot + 1 == NSLOT ? 0 : slot + 1;
        }
    }
#endif
    __builtin_amdgcn_s_waitcnt((7 << 4) | (0 << 8) | 0);
    __syncthreads();
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
    (void)hipMalloc(&w, wbytes);
    (void)hipMalloc(&h, hbytes);
    (void)hipMalloc(&out, 256 * THREADS * 16);
    std::vector<unsigned short> hw(wbytes / 2), hh(hbytes / 2);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x2c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
    for (size_t i = 0; i < hh.size(); ++i) hh[i] = 0x3800 + (rand() & 0x7ff) + ((rand() & 1) << 15);
    (void)hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice);
    (void)hipMemcpy(h, hh.data(), hbytes, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ffn12f_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(ffn12f_kernel, dim3(256), dim3(THREADS), LDS, 0, w, (unsigned)wbytes, h, (unsigned)hbytes, out);
        (void)hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(ffn12f_kernel, dim3(256), dim3(THREADS), LDS, 0, w, (unsigned)wbytes, h, (unsigned)hbytes, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms / 20 < best) best = ms / 20;
    }
    hipError_t err = hipGetLastError();
    printf("ALT=%d SYNC=%d NSLOT=%d: %.1f us per launch (240 steps: %.0f ns per step), err=%s\n", ALT, SYNC, NSLOT, best * 1e3, best * 1e6 / 240, hipGetErrorString(err));
    return 0;
}
