// dev micro-benchmark (round 5): step loop of the f16x3 FFN kernel with FOUR computing waves (one per SIMD) + FOUR helper waves (one per
// SIMD) at 256 registers, and the chunk's GELU taken OUT of the computing waves:
//   * computing wave (rg, cg): A-step 48 rows x 64 units (36 MFMAs, 14 fragment reads), B-step 48 rows x 96 outputs of the half block
//     (54 MFMAs, 12 weight reads + 6 G reads every other step): 56 KiB of fragment reads per A-step (eight-wave tiling: 80), 120 KiB
//     per k-block of B-steps (144). "Rolling" reads: a fragment register is re-read for its next use right behind the last MFMA that
//     needs it - no second register set; the barrier of step t + 1 sits inside step t.
//   * phases are SKEWED: A(0); A(1), B(0); A(2), B(1); ... A(11), B(10); B(11). P(c) (the fp32 accumulators of chunk c's A-steps) stays
//     in registers through B(c - 1), is written RAW into the G tile when B(c - 1) is done, and the helper waves turn it into GELU'd
//     (hi, lo) pairs IN PLACE while the computing waves run A(c + 1): a wave that owns no MFMA does the VALU work.
//   * helper wave: HMODE 0: every helper issues its quarter of the DMA pieces AND one GELU portion (8 lines x 128 B) per step;
//                  HMODE 1: helpers 0, 1 issue all DMA pieces, helpers 2, 3 do all the GELU.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/ffn44g.hip -o scripts/micro/build/ffn44g [-DHMODE=1] [-DABL=n] [-DSTAMP=1]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#ifndef HMODE
#define HMODE 0
#endif
#ifndef GELU
#define GELU 1  // 0: the helpers only split (timing)
#endif
#ifndef ABL  // timing-only ablations: 1 no DMA traffic (empty descriptors), 4 no MFMA, 8 helpers skip the GELU portions
#define ABL 0
#endif
#ifndef STAMP
#define STAMP 0
#endif
#ifndef CPRIO
#define CPRIO 0  // s_setprio of the computing waves
#endif
constexpr int BM = 96, E = 384, CHUNK = 128, NCH = 12;
constexpr int CW = 4, WAVES = 8, THREADS = WAVES * 64;
constexpr int G_KB = BM * 128, OFF_G = 0, OFF_RING = 4 * G_KB, SLOTB = 28 * 1024, LDS = OFF_RING + 4 * SLOTB;
constexpr int NA = 12, NB = 8, STEPS = NA + NB;
constexpr int A_BLOCK = CHUNK * 128, B_BLOCK = 192 * 128, B_PART = NA * A_BLOCK, CHUNK_BYTES = B_PART + NB * B_BLOCK;
constexpr int X_OFF = 16 * 1024;
constexpr int NDW = HMODE == 1 ? 2 : 4;  // helper waves that issue DMA pieces

template <int N>
__device__ __forceinline__ void waitvm() {
    static_assert(N >= 0 && N < 64, "vmcnt");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void waitvm_n(int n) {
    switch (n) {
        case 12: waitvm<12>(); break;
        case 13: waitvm<13>(); break;
        case 14: waitvm<14>(); break;
        case 24: waitvm<24>(); break;
        case 26: waitvm<26>(); break;
        case 28: waitvm<28>(); break;
        default: waitvm<0>(); break;
    }
}
__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    if (ABL & 4) return c;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
#define PINNED() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ float gelu_split_in(float x) {
#if GELU
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float tt = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float qq = __builtin_fmaf(tt, 1.061405429f, -1.453152027f);
    qq = __builtin_fmaf(tt, qq, 1.421413741f);
    qq = __builtin_fmaf(tt, qq, -0.284496736f);
    qq = __builtin_fmaf(tt, qq, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-(z * z) * 1.44269504088896340736f);
    const float ez = tt * qq * e;
    float gv = __builtin_fmaf(-0.5f * fabsf(x), ez, fmaxf(x, 0.f));
#else
    float gv = x;
#endif
    asm("" : "+v"(gv));
    return gv;
}

__global__ __launch_bounds__(THREADS) void ffn44g_kernel(const char* __restrict__ wpack, unsigned w_bytes, const char* __restrict__ h, unsigned h_bytes,
                                                         float* __restrict__ out, unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned long long t_start = __builtin_amdgcn_s_memtime(), r_start = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM;
    char* const ring = smem + OFF_RING;
    for (int i = tid; i < 4 * G_KB / 4; i += THREADS) reinterpret_cast<unsigned*>(smem)[i] = 0x2c003c00u + ((i * 2654435761u) >> 20 & 0x007f007fu);
    __syncthreads();
    const int c_rot = (int)(blockIdx.x & 7);
    auto stamp = [&](int i) {
        if (STAMP && blockIdx.x == 5 && threadIdx.x == 0) stamps[i] = __builtin_amdgcn_s_memtime();
    };

    if (wv >= CW) {
        // ---------------- helper waves. Barrier protocol of a step (pp_ffn_dma.hip): at the barrier of step s its pieces have landed, the
        // pieces of steps s + 1, s + 2 may be out; behind it step s + 3 goes into the slot of step s - 1.
        const int d = wv - CW;
        const bool dma = d < NDW;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wpack), 0, (ABL & 1) ? 0u : w_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(h), 0, (ABL & 1) ? 0u : h_bytes, 0x00020000);
        const unsigned v_w = (unsigned)lane * 16u;
        const int x_l = lane >> 3;
        const unsigned v_x = (unsigned)(m0 + x_l) * (unsigned)(E * 4) + (unsigned)(((lane & 7) ^ x_l) << 4);
        // step t of phase kind A (chunk ca) / B (chunk cb)
        auto issue_a = [&](int ca, int t) {
            int c = (ca % NCH) + c_rot;
            c = c >= NCH ? c - NCH : c;
            const int base = c * CHUNK_BYTES;
            char* dst = ring + (t & 3) * SLOTB;
            const int kb = (ca & 1) ? NA - 1 - t : t;
#pragma unroll
            for (int u = 0; u < 16 / NDW; ++u) {
                const int q = d + NDW * u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + q * 1024), 16, v_w, base + kb * A_BLOCK + q * 1024, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 12 / NDW; ++u) {
                const int q = d + NDW * u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr_t)(dst + X_OFF + q * 1024), 16, v_x, kb * 128 + q * 8 * E * 4, 0, 0);
            }
        };
        auto issue_b = [&](int cb, int sb) {
            int c = (cb % NCH) + c_rot;
            c = c >= NCH ? c - NCH : c;
            const int base = c * CHUNK_BYTES;
            char* dst = ring + (sb & 3) * SLOTB;
#pragma unroll
            for (int u = 0; u < 24 / NDW; ++u) {
                const int q = d + NDW * u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + q * 1024), 16, v_w, base + B_PART + sb * B_BLOCK + q * 1024, 0, 0);
            }
        };
        constexpr int PA = 28 / NDW, PB = 24 / NDW;  // pieces per issuing wave and step
        // GELU portion i (0 .. 11) of this helper: eight lines (rows) of k-block i / 3: raw fp32 P in, (hi, lo) halves out, in place.
        // LDS instructions as inline assembly: behind a buffer_load ... lds the compiler puts s_waitcnt vmcnt(0) in front of ordinary
        // LDS accesses. Line l of the eight: lane >> 3; the lane's physical 16-byte chunk lane & 7 holds logical chunk (lane & 7) ^ (row & 7).
        const int NGW = HMODE == 1 ? 2 : 4;        // helper waves that do GELU
        const int gw = HMODE == 1 ? d - 2 : d;     // index among them
        const bool gelu_wave = HMODE == 1 ? d >= 2 : true;
        auto gelu_portion = [&](int i) {
            if ((ABL & 8) || !gelu_wave) return;
            constexpr int ROWS_PER = 8;
            const int per_kb = (BM / NGW) / ROWS_PER;  // portions per k-block and helper: 3 (four GELU waves) or 6 (two)
            const int jb = i / per_kb, r0 = gw * (BM / NGW) + (i % per_kb) * ROWS_PER;
            const int row = r0 + (lane >> 3);
            const int lc = (lane & 7) ^ (row & 7);  // logical chunk = units 4 lc .. 4 lc + 3 of the k-block
            const int a_in = OFF_G + jb * G_KB + row * 128 + ((lane & 7) << 4);
            f32x4 p;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(p) : "v"(a_in) : "memory");
            h4 hv, lv;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float gv = gelu_split_in(p[q]);
                hv[q] = (_Float16)gv;
                lv[q] = (_Float16)(gv - (float)hv[q]);
            }
            const int a_hi = OFF_G + jb * G_KB + row * 128 + (((lc >> 1) ^ (row & 7)) << 4) + (lc & 1) * 8;
            const int a_lo = OFF_G + jb * G_KB + row * 128 + (((4 + (lc >> 1)) ^ (row & 7)) << 4) + (lc & 1) * 8;
            asm volatile("ds_write_b64 %0, %1" ::"v"(a_hi), "v"(__builtin_bit_cast(u32x2, hv)) : "memory");
            asm volatile("ds_write_b64 %0, %1" ::"v"(a_lo), "v"(__builtin_bit_cast(u32x2, lv)) : "memory");
        };
        const int NPORT = 12 * 4 / NGW;  // portions per GELU wave and chunk
        auto helper_barrier = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's G writes are in
            __builtin_amdgcn_s_barrier();
        };
        if (dma) {
            issue_a(0, 0);
            issue_a(0, 1);
            issue_a(0, 2);
        }
        // the step sequence: it = 0 .. 12: [A(it) if it < 12] [B(it - 1) if it >= 1]; the three steps after a step of the sequence
        auto issue_ahead = [&](int it, bool in_a, int t) {
            // step index t inside the phase; the step 3 ahead
            if (!dma) return;
            if (in_a) {
                if (t + 3 < NA) issue_a(it, t + 3);
                else if (it >= 1) issue_b(it - 1, t + 3 - NA);
                else issue_a(it + 1, t + 3 - NA);  // A(0) is followed by A(1)
            } else {
                if (t + 3 < NB) issue_b(it - 1, t + 3);
                else if (it < NCH - 1) issue_a(it + 1, t + 3 - NB);
                else if (it == NCH - 1) issue_b(it, t + 3 - NB);  // B(10) is followed by B(11)
                else issue_b(0, t + 3 - NB);                      // fillers behind the last phase (same counts)
            }
        };
        for (int it = 0; it <= NCH; ++it) {
            if (it < NCH) {
#pragma unroll
                for (int t = 0; t < NA; ++t) {
                    __builtin_amdgcn_sched_barrier(0);
                    // pieces allowed out: the two steps behind step t: A-steps except at the phase end (B-steps, or A(1) behind A(0))
                    const int n1 = t + 1 < NA ? PA : -1, n2 = t + 2 < NA ? PA : -1;
                    if (dma) {
                        if (n1 > 0 && n2 > 0) waitvm_n(2 * PA);
                        else if (n1 > 0) { if (it >= 1) waitvm_n(PA + PB); else waitvm_n(2 * PA); }
                        else { if (it >= 1) waitvm_n(2 * PB); else waitvm_n(2 * PA); }
                    }
                    helper_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    issue_ahead(it, true, t);
                    // (the computing waves pass the barrier of a phase's first step INSIDE the previous phase's last step: the barrier
                    // behind their dump of P(it - 1) comes next)
                    if (t == 0 && it >= 1) helper_barrier();
                    // GELU of chunk it - 1 during A(it)
                    if (it >= 1) {
#pragma unroll
                        for (int u = 0; u < NPORT / NA; ++u) gelu_portion(t * (NPORT / NA) + u);
                    }
                }
            }
            if (it >= 1) {
#pragma unroll
                for (int sb = 0; sb < NB; ++sb) {
                    __builtin_amdgcn_sched_barrier(0);
                    const int n1 = sb + 1 < NB ? PB : -1, n2 = sb + 2 < NB ? PB : -1;
                    if (dma) {
                        if (n1 > 0 && n2 > 0) waitvm_n(2 * PB);
                        else if (n1 > 0) { if (it < NCH - 1) waitvm_n(PB + PA); else waitvm_n(2 * PB); }
                        else { if (it < NCH - 1) waitvm_n(2 * PA); else waitvm_n(2 * PB); }
                    }
                    helper_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    issue_ahead(it, false, sb);
                    if (it == NCH && sb == 0) {
                        // the last chunk's GELU has no A-phase to hide behind
                        helper_barrier();
#pragma unroll
                        for (int i = 0; i < NPORT; ++i) gelu_portion(i);
                        helper_barrier();
                    }
                }
            }
        }
        waitvm<0>();
        helper_barrier();  // the barrier "of the step behind the last"
        return;
    }

    // ---------------- computing waves
    if (CPRIO) __builtin_amdgcn_s_setprio(CPRIO);
    const int rg = wv >> 1, cg = wv & 1;
    const int f_row = lane & 15, f_kg = lane >> 4;
    const int sw = f_row & 7;
    const int lane_hi = f_row * 128 + ((f_kg ^ sw) << 4), lane_lo = f_row * 128 + (((4 + f_kg) ^ sw) << 4);
    const int rows0 = rg * 48 + f_row;
    auto opaque_s = [](int v) { asm volatile("" : "+s"(v)); return v; };
    auto rd = [&](int lane_off, int uni, int imm) -> u32x4 { return *reinterpret_cast<const u32x4*>(smem + (lane_off + uni) + imm); };
    const int u_a = OFF_RING + cg * 64 * 128;          // A-step W1 lines (units 64 cg ..) inside a slot
    const int u_x = OFF_RING + X_OFF + rg * 48 * 128;  // x lines of an A slot (rows 48 rg ..)
    const int u_b = OFF_RING + cg * 96 * 128;          // B-step W2 lines (outputs 96 cg .. of the half block)
    const int u_g = OFF_G + rg * 48 * 128;             // row lines of a G buffer

    f32x4 acc[3][12];  // [row fragment][half * 6 + nf]: columns 192 half + 96 cg + 16 nf ..
    f32x4 pacc[3][4];  // P of a chunk: [row fragment][unit fragment]
    u32x4 rh_[3], rl_[3];  // row fragments of the running step: x (A) or G (B)
    u32x4 fh[2], fl[2];    // streamed weight fragment: ring of two pairs
#pragma unroll
    for (int rf = 0; rf < 3; ++rf)
#pragma unroll
        for (int c = 0; c < 12; ++c) acc[rf][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto step_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt((63 & 15) | (7 << 4) | (0 << 8) | ((63 >> 4) << 14));  // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // P -> the G tile, raw: lane holds units 64 cg + 16 nf + 4 f_kg + (0..3) = k-block 2 cg + (nf >> 1), logical 16-byte chunk
    // 4 (nf & 1) + f_kg of the row's line
    auto dump_p = [&]() {
#pragma unroll
        for (int rf = 0; rf < 3; ++rf)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                char* gs = smem + OFF_G + (2 * cg + (nf >> 1)) * G_KB + (rows0 + rf * 16) * 128;
                *reinterpret_cast<f32x4*>(gs + (((4 * (nf & 1) + f_kg) ^ sw) << 4)) = pacc[rf][nf];
            }
    };

    // the barrier of the very first step, then its fragments
    step_barrier();
    bool frags_loaded = false;
    int gstep = 0;  // for the stamps
    for (int it = 0; it <= NCH; ++it) {
        if (it < NCH) {
            if (it >= 1) {
                // P(it - 1) (held through B(it - 2)) -> G tile, raw; the barrier hands it to the helpers
                dump_p();
                step_barrier();
            }
#pragma unroll
            for (int rf = 0; rf < 3; ++rf)
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) pacc[rf][nf] = f32x4{0.01f, 0.02f, 0.03f, 0.04f};
            if (!frags_loaded) {
                const int ua = opaque_s(u_a), ux = opaque_s(u_x);
                fh[0] = rd(lane_hi, ua, 0);
                fl[0] = rd(lane_lo, ua, 0);
#pragma unroll
                for (int rf = 0; rf < 3; ++rf) { rh_[rf] = rd(lane_hi, ux, rf * 2048); rl_[rf] = rd(lane_lo, ux, rf * 2048); }
            }
            // ---- A(it): group nf = unit fragment nf (pair nf & 1): 9 MFMAs
#pragma unroll
            for (int t = 0; t < NA; ++t) {
                if (STAMP && it == 6) stamp(t);
                const int ua = opaque_s(u_a + (t & 3) * SLOTB);
                const int ua_n = opaque_s(u_a + ((t + 1) & 3) * SLOTB), ux_n = opaque_s(u_x + ((t + 1) & 3) * SLOTB);
                const int ub_n = opaque_s(u_b + ((t + 1) & 3) * SLOTB);  // (t + 1 == NA: slot 0)
                const int ug_n = opaque_s(u_g);
                const bool last = t + 1 == NA;
                const bool to_b = last && it >= 1;  // the next phase is B(it - 1); else A(it + 1) (only behind A(0))
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) {
                    const int p = nf & 1, pn = p ^ 1;
#pragma unroll
                    for (int j = 0; j < 9; ++j) {
                        const int rf = j % 3, sweep = j / 3;
                        pacc[rf][nf] = mma(sweep == 1 ? fl[p] : fh[p], sweep == 2 ? rl_[rf] : rh_[rf], pacc[rf][nf]);
                        PINNED();
                        if (nf < 3) {
                            if (j == 0) fh[pn] = rd(lane_hi, ua, (nf + 1) * 2048);
                            if (j == 1) fl[pn] = rd(lane_lo, ua, (nf + 1) * 2048);
                        } else if (!last) {
                            if (j == 0) fh[pn] = rd(lane_hi, ua_n, 0);
                            if (j == 1) fl[pn] = rd(lane_lo, ua_n, 0);
                            if (sweep == 1) rh_[rf] = rd(lane_hi, ux_n, rf * 2048);
                            if (sweep == 2) rl_[rf] = rd(lane_lo, ux_n, rf * 2048);
                        } else {
                            // the first step of the next phase (wave-uniform branch on `to_b`)
                            if (j == 0) fh[pn] = to_b ? rd(lane_hi, ub_n, 0) : rd(lane_hi, ua_n, 0);
                            if (j == 1) fl[pn] = to_b ? rd(lane_lo, ub_n, 0) : rd(lane_lo, ua_n, 0);
                            if (sweep == 1) rh_[rf] = to_b ? rd(lane_hi, ug_n, rf * 2048) : rd(lane_hi, ux_n, rf * 2048);
                            if (sweep == 2) rl_[rf] = to_b ? rd(lane_lo, ug_n, rf * 2048) : rd(lane_lo, ux_n, rf * 2048);
                        }
                        if (nf == 2 && j == 8) step_barrier();  // the barrier of step t + 1: its slot has landed, slot t is read out
                        PINNED();
                    }
                }
            }
            frags_loaded = true;
        }
        if (it >= 1) {
            if (it == NCH) {
                // the last chunk: dump, wait for the helpers' exposed GELU, then the fragments
                dump_p();
                step_barrier();
                step_barrier();
                const int ub = opaque_s(u_b), ug = opaque_s(u_g);
                fh[0] = rd(lane_hi, ub, 0);
                fl[0] = rd(lane_lo, ub, 0);
#pragma unroll
                for (int rf = 0; rf < 3; ++rf) { rh_[rf] = rd(lane_hi, ug, rf * 2048); rl_[rf] = rd(lane_lo, ug, rf * 2048); }
            }
            // ---- B(it - 1): group nf = output fragment nf of this wave's 96 columns of the half (pair nf & 1): 9 MFMAs
#pragma unroll
            for (int sb = 0; sb < NB; ++sb) {
                if (STAMP && it == 6) stamp(NA + 1 + sb);
                const int half = sb & 1, jb = sb >> 1;
                const int ub = opaque_s(u_b + (sb & 3) * SLOTB);
                const int ub_n = opaque_s(u_b + ((sb + 1) & 3) * SLOTB), ug_n = opaque_s(u_g + ((jb + 1) & 3) * G_KB);
                const int ua_n = opaque_s(u_a + ((sb + 1) & 3) * SLOTB), ux_n = opaque_s(u_x + ((sb + 1) & 3) * SLOTB);
                const bool last = sb + 1 == NB;
                const bool to_a = last && it < NCH - 1;  // the next phase is A(it + 1); else B(it) (only behind B(10)): no prefetch then
#pragma unroll
                for (int nf = 0; nf < 6; ++nf) {
                    const int p = nf & 1, pn = p ^ 1;
#pragma unroll
                    for (int j = 0; j < 9; ++j) {
                        const int rf = j % 3, sweep = j / 3;
                        acc[rf][half * 6 + nf] = mma(sweep == 1 ? fl[p] : fh[p], sweep == 2 ? rl_[rf] : rh_[rf], acc[rf][half * 6 + nf]);
                        PINNED();
                        if (nf < 5) {
                            if (j == 0) fh[pn] = rd(lane_hi, ub, (nf + 1) * 2048);
                            if (j == 1) fl[pn] = rd(lane_lo, ub, (nf + 1) * 2048);
                        } else if (!last) {
                            if (j == 0) fh[pn] = rd(lane_hi, ub_n, 0);
                            if (j == 1) fl[pn] = rd(lane_lo, ub_n, 0);
                            if (half == 1 && sweep == 1) rh_[rf] = rd(lane_hi, ug_n, rf * 2048);
                            if (half == 1 && sweep == 2) rl_[rf] = rd(lane_lo, ug_n, rf * 2048);
                        } else if (to_a) {
                            if (j == 0) fh[pn] = rd(lane_hi, ua_n, 0);
                            if (j == 1) fl[pn] = rd(lane_lo, ua_n, 0);
                            if (sweep == 1) rh_[rf] = rd(lane_hi, ux_n, rf * 2048);
                            if (sweep == 2) rl_[rf] = rd(lane_lo, ux_n, rf * 2048);
                        }
                        if (nf == 4 && j == 8) step_barrier();  // the barrier of step t + 1
                        PINNED();
                    }
                }
            }
            if (STAMP && it == 6) stamp(NA + 1 + NB);
        }
    }
    if (STAMP && blockIdx.x == 5 && threadIdx.x == 0) {
        stamps[30] = t_start; stamps[31] = __builtin_amdgcn_s_memtime(); stamps[32] = r_start; stamps[33] = __builtin_amdgcn_s_memrealtime();
    }
    __builtin_amdgcn_s_waitcnt((7 << 4) | (0 << 8) | 0);
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rf = 0; rf < 3; ++rf) {
#pragma unroll
        for (int c = 0; c < 12; ++c) sum += acc[rf][c];
#pragma unroll
        for (int c = 0; c < 4; ++c) sum += pacc[rf][c];
        sum[0] += __builtin_bit_cast(float, rh_[rf][0]) + __builtin_bit_cast(float, rl_[rf][1]);
    }
    sum[1] += __builtin_bit_cast(float, fh[0][0]) + __builtin_bit_cast(float, fl[1][1]);
    reinterpret_cast<f32x4*>(out)[(size_t)blockIdx.x * THREADS + tid] = sum;
}

int main() {
    const size_t wbytes = (size_t)NCH * CHUNK_BYTES, M = 256 * BM, hbytes = M * E * 4;
    char *w, *h;
    float* out;
    unsigned long long* stamps;
    hipMalloc(&w, wbytes);
    hipMalloc(&h, hbytes);
    hipMalloc(&out, 256 * THREADS * 16);
    hipMalloc(&stamps, 64 * 8);
    hipMemset(stamps, 0, 64 * 8);
    std::vector<unsigned short> hw(wbytes / 2), hh(hbytes / 2);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x2c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
    for (size_t i = 0; i < hh.size(); ++i) hh[i] = 0x3800 + (rand() & 0x7ff) + ((rand() & 1) << 15);
    hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice);
    hipMemcpy(h, hh.data(), hbytes, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(ffn44g_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(ffn44g_kernel, dim3(256), dim3(THREADS), LDS, 0, w, (unsigned)wbytes, h, (unsigned)hbytes, out, stamps);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(ffn44g_kernel, dim3(256), dim3(THREADS), LDS, 0, w, (unsigned)wbytes, h, (unsigned)hbytes, out, stamps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms / 20 < best) best = ms / 20;
    }
    hipError_t err = hipGetLastError();
    printf("ffn44g HMODE=%d GELU=%d CPRIO=%d ABL=%d: %.1f us per launch (240 steps: %.0f ns per step), err=%s\n", HMODE, GELU, CPRIO, ABL, best * 1e3,
           best * 1e6 / 240, hipGetErrorString(err));
    if (STAMP) {
        unsigned long long st[64];
        hipMemcpy(st, stamps, sizeof(st), hipMemcpyDeviceToHost);
        const double cyc = (double)(st[31] - st[30]), us = (double)(st[33] - st[32]) / 100.0;
        printf("  workgroup 5: %.0f cycles in %.1f us = %.0f MHz; iteration 6 steps (cycles): A", cyc, us, cyc / us);
        for (int i = 0; i < NA - 1; ++i) printf(" %llu", st[i + 1] - st[i]);
        printf(" | last A -> B0 %llu | B", st[NA + 1] - st[NA - 1]);
        for (int i = 0; i < NB; ++i) printf(" %llu", st[NA + 2 + i] - st[NA + 1 + i]);
        printf("\n");
    }
    return 0;
}
