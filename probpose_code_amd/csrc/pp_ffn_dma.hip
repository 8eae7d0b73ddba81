// The fused f16x3 feed-forward launch (entry points, packed weight streams: pp_ffn_split.hip) with the LDS-DMA issue in waves of its own - the
// eight-wave kernel of round 3 issued it from the computing waves (GELU written max(x, 0) - 0.5 |x| erfc(|x| / sqrt 2)):
//     [x <- x + att Wp^T + bp ; h <- LN2(x)]   x <- x + GELU(h W1^T + b1) W2^T + b2 ;   h_next <- LayerNorm(x)
// (mmpretrain TransformerEncoderLayer [3P]; call site mmpose/models/pose_estimators/base.py:206).
//
//   * 768 threads = 12 waves, three per SIMD, at <= 168 registers each:
//       waves 0-7   compute: wave (rg, cg) = rows 48 rg .. +47, column quarter cg - the tiles of pp_ffn_split.hip - and issue NO
//                   memory instruction inside the step loop: per step one barrier, the fragment reads, 18 / 27 MFMAs;
//       waves 8-11  one per SIMD: issue ALL buffer_load ... lds pieces (7 per A-step, 6 per B-step each) three steps ahead and
//                   do the counted s_waitcnt vmcnt(N) in front of every barrier.
//     A buffer_load ... lds holds its wave for 60 - 180 cycles when the texture path is busy; in the eight-wave kernel that
//     wave also owns MFMAs, and the role alternation there hides the stall behind the OTHER wave of the SIMD at the price of
//     two barriers per step. Here the stalls belong to a wave that has nothing else to do.
//   * the price is the register budget (512 / 3): no second accumulator set, so a chunk's GELU is not spread under the next
//     chunk's steps - it runs between the chunk's A-steps and its B-steps, all eight waves at once (~12 x 0.35 us per launch);
//   * steps, ring (four 28 KiB slots), G tile, LayerNorm epilogue, k-block sawtooth and chunk rotation as in pp_ffn_split.hip.
//   * the plain-load / LDS-DMA retire-order hazard of the eight-wave kernel (a younger plain load may retire before an older
//     piece's LDS write) cannot occur: the waves that count pieces issue nothing else.
// Measured as a skeleton first (scripts/micro/ffn12d.hip MODE=1 GELU=1): 148 - 150 us against 161 us for the eight-wave loop.
// TOOLCHAIN NOTE: the paired kernels issue their MFMAs as inline assembly (mma_ip), which the compiler's hazard recogniser does not
// see; the wait states in front of every VALU read of an accumulator are placed by hand (mfma_settle / valu_settle) and the schedule distance
// between two MFMAs on one accumulator is >= 6 MFMAs by construction of the sweeps. Built and validated with ROCm 7.2.0's hipcc (AMD clang 20);
// after a compiler upgrade or a change of the loop structure re-run tests/test_split_fp16.py - its `ffn_form` 1 runs the SAME
// arithmetic on the builtin-MFMA kernels (hazards handled by the compiler) and both forms are held to the fp64 reference.
#include "pp_common.h"
#include "pp_split.h"
#include "pp_ffn_params.h"

namespace pp {
namespace ffd {

using ffs::Params;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int BM = 96, E = 384, CHUNK = 128;
constexpr int CW = 8, WAVES = 12, THREADS = WAVES * 64;
constexpr int KB = E / 32;
constexpr int G_KB = BM * 128;
constexpr int OFF_G = 0;
constexpr int OFF_RING = 4 * G_KB;
constexpr int SLOTB = 28 * 1024, NSLOT = 4;
constexpr int LDS = OFF_RING + NSLOT * SLOTB;
constexpr int X_OFF = 16 * 1024;
constexpr int NA = KB, NB = 8, STEPS = NA + NB;
constexpr int A_BLOCK = CHUNK * 128;
constexpr int B_BLOCK = (E / 2) * 128;
constexpr int B_PART = NA * A_BLOCK;
constexpr int CHUNK_BYTES = B_PART + NB * B_BLOCK;
constexpr int NPROJ = 2 * KB;  // steps of the projection phase
static_assert(LDS == 160 * 1024, "LDS map");
static_assert(STEPS % NSLOT == 0 && NPROJ % NSLOT == 0, "ring positions must repeat");

#ifndef FFD_STAMP
#define FFD_STAMP 0  // dev: wave 0 of workgroups 0 and 131 leaves s_memtime stamps at the phase boundaries of the paired proj + FFN kernel (scripts/micro/ffd_stamps.sh)
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
// (n is a compile-time constant after unrolling; the builtin wants a literal)
__device__ __forceinline__ void wait_vm_n(int n) {
    switch (n) {
        case 0: wait_vm<0>(); break;
        case 6: wait_vm<6>(); break;
        case 7: wait_vm<7>(); break;
        case 9: wait_vm<9>(); break;
        case 10: wait_vm<10>(); break;
        case 11: wait_vm<11>(); break;
        case 12: wait_vm<12>(); break;
        case 13: wait_vm<13>(); break;
        case 14: wait_vm<14>(); break;
        case 15: wait_vm<15>(); break;
        default: wait_vm<0>(); break;
    }
}
// pieces one DMA wave issues for a step of the main loop / of the projection phase
__host__ __device__ constexpr int n_main(int t) { return (((t % STEPS) + STEPS) % STEPS) < NA ? 7 : 6; }
__host__ __device__ constexpr int n_proj(int s) { return s < NPROJ ? 6 + ((s & 1) == 0 ? 3 : 0) : 0; }

__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// In-place MFMA as inline assembly (PAIR form only): acc += a b with the accumulator tied to one register tuple. The builtin form lets the
// register allocator RENAME the destination of a step's first sweep inside the run-time k-block loop (+24 live registers: 146 spilled);
// tied, the loop holds at 160. The compiler's hazard recogniser does not see an MFMA in an asm statement: the VALU reads of the
// accumulators (GELU, LayerNorm) are preceded by explicit wait states (mfma_settle), the loads that overwrite fragment registers are
// ordered by s_waitcnt like any other use, and an accumulator is touched again six MFMAs (96 cycles) later at the earliest.
__device__ __forceinline__ void mma_ip(const u32x4& a, const u32x4& b, f32x4& c) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// (the accumulator fragments are in / out operands of the wait-state block: a read of them cannot be scheduled in front of it)
__device__ __forceinline__ void mfma_settle(f32x4 (&a)[3][2]) {
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]));
}
__device__ __forceinline__ void valu_settle(f32x4 (&a)[3][2]) {
    asm volatile("s_nop 4" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]));
}
__device__ __forceinline__ void mfma_settle(f32x4 (&a)[3][6]) {
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[0][4]), "+v"(a[0][5]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]),
                   "+v"(a[1][3]), "+v"(a[1][4]), "+v"(a[1][5]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[2][2]), "+v"(a[2][3]), "+v"(a[2][4]), "+v"(a[2][5]));
}

// ---------------------------------------------------------------- DMA waves
// Barrier protocol (every wave of the workgroup executes the same sequence of s_barrier):
//   PROJ: 24 step barriers | P1 | LayerNorm 2 | P2 ;   main: 20 per chunk | E1 | LayerNorm 2
template <bool PROJ, bool PAIR>
__device__ __forceinline__ void dma_role(const Params& p, char* smem, int d, int lane, int m0, int nchunks, int c_rot) {
    char* const ring = smem + OFF_RING;
    const int x_l = lane >> 3;
    const unsigned v_w = (unsigned)lane * 16u;
    // an x piece is 8 rows x 128 B: lane (row l = lane >> 3, physical chunk lane & 7) fetches logical chunk (lane & 7) ^ l
    const unsigned v_x = (unsigned)(m0 + x_l) * (unsigned)(E * 4) + (unsigned)(((lane & 7) ^ x_l) << 4);

    if constexpr (PROJ) {
        auto issue_p = [&](int s) {
            if (s >= NPROJ) return;
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wproj), 0, p.wproj_bytes, 0x00020000);
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int q = d + 4 * u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(ring + (s & 3) * SLOTB + q * 1024), 16, v_w, s * B_BLOCK + q * 1024, 0, 0);
            }
            if ((s & 1) == 0) {
                const int kb = s >> 1;
                const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.att), 0, p.att_bytes, 0x00020000);
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int q = d + 4 * u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(smem + OFF_G + (kb & 3) * G_KB + q * 1024), 16, v_x,
                                                             kb * 128 + q * 8 * E * 4, 0, 0);
                }
            }
        };
        issue_p(0);
        issue_p(1);
        issue_p(2);
#pragma unroll
        for (int s = 0; s < NPROJ; ++s) {
            // step s must have landed: the pieces of the steps behind it may be out
            __builtin_amdgcn_sched_barrier(0);
            wait_vm_n(n_proj(s + 1) + n_proj(s + 2));
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            issue_p(s + 3);
        }
        __builtin_amdgcn_s_barrier();  // P1
        __builtin_amdgcn_s_barrier();  // LayerNorm (ln2)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();  // P2: the ln2 rows are in L2
    }

    // piece set of step t (0 .. 19) of the chunk visited ci-th; past the last chunk the descriptors have no extent (the DMA writes
    // zeros, the counts stay the same)
    auto issue = [&](int ci, int t) {
        const bool live = ci < nchunks;
        int c = (live ? ci : 0) + c_rot;
        c = c >= nchunks ? c - nchunks : c;
        const int base = c * CHUNK_BYTES;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wpack), 0, live ? p.w_bytes : 0u, 0x00020000);
        char* dst = ring + (t & 3) * SLOTB;
        if (t < NA) {
            const int kb = (ci & 1) ? NA - 1 - t : t;  // odd visits walk the k-blocks backwards (pp_ffn_split.hip: the L2 finds the rows)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = d + 4 * u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + q * 1024), 16, v_w, base + kb * A_BLOCK + q * 1024, 0, 0);
            }
            const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.h), 0, live ? p.h_bytes : 0u, 0x00020000);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int q = d + 4 * u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr_t)(dst + X_OFF + q * 1024), 16, v_x, kb * 128 + q * 8 * E * 4, 0, 0);
            }
        } else {
            const int sb = t - NA;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int q = d + 4 * u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + q * 1024), 16, v_w, base + B_PART + sb * B_BLOCK + q * 1024, 0, 0);
            }
        }
    };
    if constexpr (PAIR) {
        // PAIRED chunks (see compute_role): a pair's 40 steps are  [W1(c0, kb) + x(kb)] [W1(c1, kb)]  x 12  |  B(c0) x 8  |  B(c1) x 8 ;
        // 7 / 4 / 6 pieces per DMA wave and step. Same protocol per step.
        const int npairs = nchunks >> 1;
        auto issue2 = [&](int cp_, int t) {
            // (the pair index through an opaque copy per step: otherwise the compiler computes the offsets of all forty steps at the loop head and
            // spills 180 SGPRs - every piece then waits for a v_readlane and its hazard states)
            int cp = cp_;
            asm volatile("" : "+s"(cp));
            const bool live = cp < npairs;
            const int cpl = live ? cp : 0;
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wpack), 0, live ? p.w_bytes : 0u, 0x00020000);
            char* dst = ring + (t & 3) * SLOTB;
            if (t < 2 * NA) {
                int c = 2 * cpl + (t & 1) + c_rot;
                c = c >= nchunks ? c - nchunks : c;
                const int kb = (cp & 1) ? NA - 1 - (t >> 1) : (t >> 1);  // odd pair visits walk the k-blocks backwards
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = d + 4 * u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + q * 1024), 16, v_w, c * CHUNK_BYTES + kb * A_BLOCK + q * 1024, 0, 0);
                }
                if ((t & 1) == 0) {
                    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.h), 0, live ? p.h_bytes : 0u, 0x00020000);
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int q = d + 4 * u;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, (lds_ptr_t)(dst + X_OFF + q * 1024), 16, v_x, kb * 128 + q * 8 * E * 4, 0, 0);
                    }
                }
            } else {
                const int which = (t - 2 * NA) >> 3, sb = (t - 2 * NA) & 7;
                int c = 2 * cpl + which + c_rot;
                c = c >= nchunks ? c - nchunks : c;
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int q = d + 4 * u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dst + q * 1024), 16, v_w, c * CHUNK_BYTES + B_PART + sb * B_BLOCK + q * 1024, 0, 0);
                }
            }
        };
        constexpr int PS = 2 * STEPS;
        auto n2 = [](int t) { t %= PS; return t < 2 * NA ? ((t & 1) == 0 ? 7 : 4) : 6; };
        issue2(0, 0);
        issue2(0, 1);
        issue2(0, 2);
        for (int cp = 0; cp < npairs; ++cp) {
#pragma unroll
            for (int t = 0; t < PS; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                wait_vm_n(n2(t + 1) + n2(t + 2));
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (t + 3 < PS) issue2(cp, t + 3); else issue2(cp + 1, t + 3 - PS);
                if (t == 2 * NA || t == 2 * NA + NB) __builtin_amdgcn_s_barrier();  // the computing waves' two G-tile barriers
            }
        }
    } else {
    issue(0, 0);
    issue(0, 1);
    issue(0, 2);
    for (int ci = 0; ci < nchunks; ++ci) {
#pragma unroll
        for (int t = 0; t < STEPS; ++t) {
            __builtin_amdgcn_sched_barrier(0);
            wait_vm_n(n_main(t + 1) + n_main(t + 2));
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // (all computing waves are past their reads of step t - 1: its slot takes step t + 3)
            if (t + 3 < STEPS) issue(ci, t + 3); else issue(ci + 1, t + 3 - STEPS);
            if (t == NA) __builtin_amdgcn_s_barrier();  // the computing waves' G-tile barrier (between their barriers of steps NA and NA + 1)
        }
    }
    }
    wait_vm<0>();                  // the fillers
    __builtin_amdgcn_s_barrier();  // E1
    __builtin_amdgcn_s_barrier();  // LayerNorm
    __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------- computing waves
#if FFD_STAMP
__device__ unsigned long long g_ffd_stamps[2][64];
#endif
struct TagF { static constexpr bool value = false; };
struct TagT { static constexpr bool value = true; };
// FOLD (bit 0: the residual rows come in the operand format; bit 1: the final LayerNorm is left to the next layer - rows out once, in the operand
// format, with their statistics): pp_proj_ffn_split_folded, the paired projection kernel only
template <bool PROJ, bool PAIR, int FOLD = 0>
__device__ __forceinline__ void compute_role(const Params& p, char* smem, int wv, int lane, int m0, int nchunks, int c_rot) {
    const int rg = wv >> 2, cg = wv & 3;
    const int f_row = lane & 15, f_kg = lane >> 4;
    auto chunk_of = [&](int i) { const int c = i + c_rot; return c >= nchunks ? c - nchunks : c; };
#if FFD_STAMP
    int n_stamp = 0;
    auto stamp = [&]() {
        if (wv == 0 && (blockIdx.x == 0 || blockIdx.x == 131)) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0 && n_stamp < 64) g_ffd_stamps[blockIdx.x == 0 ? 0 : 1][n_stamp] = t;
            ++n_stamp;
        }
    };
#else
    auto stamp = []() {};
#endif
    stamp();  // 0: start

    // fragment reads: hi halves in 16-byte chunk f_kg, lo halves in chunk 4 + f_kg of a line, swizzled by line & 7. The register
    // budget has no room for an address register per line set: every read address is ONE of two per-lane offsets (hi / lo chunk of
    // line f_row) plus a wave-uniform offset the compiler cannot fold or hoist (it lives in an SGPR, made opaque per step), plus
    // the fragment stride as an instruction offset.
    const int sw = f_row & 7;
    const int lane_hi = f_row * 128 + ((f_kg ^ sw) << 4), lane_lo = f_row * 128 + (((4 + f_kg) ^ sw) << 4);
    const int rows0 = rg * 48 + f_row;
    auto opaque_s = [](int v) { asm volatile("" : "+s"(v)); return v; };
    auto rd = [&](int lane_off, int uni, int imm) -> u32x4 { return *reinterpret_cast<const u32x4*>(smem + (lane_off + uni) + imm); };
    const int u_a = OFF_RING + cg * 32 * 128;           // A-step W1 lines of this wave (units 32 cg ..) inside a slot
    const int u_b = OFF_RING + cg * 48 * 128;           // B-step W2 lines (outputs 48 cg ..)
    const int u_x = OFF_RING + X_OFF + rg * 48 * 128;   // x lines of an A slot (rows 48 rg ..)
    const int u_g = OFF_G + rg * 48 * 128;              // row lines of a G buffer

    f32x4 acc[3][6];   // [row fragment][half * 3 + nf]: columns 192 half + 48 cg + 16 nf + 4 f_kg + (0..3)
    f32x4 pacc[3][2];  // P of the chunk in its A-steps
    f32x4 b1v[2];
    u32x4 bgh[3], bgl[3];

    auto step_barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        // lgkmcnt(0): this wave's LDS writes (the G tile) are in before anyone is let through; its reads were consumed by the MFMAs
        __builtin_amdgcn_s_waitcnt((63 & 15) | (7 << 4) | (0 << 8) | ((63 >> 4) << 14));
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // 48 x 48 tile of one column half: acc[:, half] += G-like rows (bgh, bgl) x weight half block in ring slot `so`
    auto b_step = [&](int slot, int half, int gbuf, bool load_g) {
        u32x4 wh[3], wl[3];
        const int ub = opaque_s(u_b + slot * SLOTB);
        const int ug = opaque_s(u_g + gbuf * G_KB);
        // reads in the order the MFMAs want them
        wh[0] = rd(lane_hi, ub, 0);
        if (load_g) bgh[0] = rd(lane_hi, ug, 0);
        wh[1] = rd(lane_hi, ub, 2048);
        wh[2] = rd(lane_hi, ub, 4096);
        if (load_g) {
            bgh[1] = rd(lane_hi, ug, 2048);
            bgh[2] = rd(lane_hi, ug, 4096);
        }
#pragma unroll
        for (int nf = 0; nf < 3; ++nf) wl[nf] = rd(lane_lo, ub, nf * 2048);
        if (load_g) {
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) bgl[rf] = rd(lane_lo, ug, rf * 2048);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rf = 0; rf < 3; ++rf)
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) acc[rf][half * 3 + nf] = mma(wh[nf], bgh[rf], acc[rf][half * 3 + nf]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rf = 0; rf < 3; ++rf)
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) acc[rf][half * 3 + nf] = mma(wl[nf], bgh[rf], acc[rf][half * 3 + nf]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nf = 0; nf < 3; ++nf)
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) acc[rf][half * 3 + nf] = mma(wh[nf], bgl[rf], acc[rf][half * 3 + nf]);
    };
    (void)b_step;  // (not every instantiation takes this path)
    // LayerNorm of the 96 x 384 block in the accumulators (pp_ffn_split.hip layernorm_rows: same operations in the same order)
    // Stores as buffer stores: the row part of the address (and the lane's column part) in the VGPR offset - rows past M fall out
    // of the descriptor's extent and are dropped by the hardware (the range check covers the VGPR offset, not the scalar one) -
    // the wave-uniform column part in the scalar offset: two address registers for the whole epilogue.
    // fold_c (std::true_type-like tag): the rows are NOT normalised - they leave once, in the operand format, to h_dst, with (mean, rstd) per row in
    // p.stats_out: the LayerNorm is applied by the layer that consumes them (pp_qkv_attention_split_folded)
    auto layernorm_rows = [&](const float* gamma, const float* beta, float* x_dst, void* h_dst, bool store_x, auto fold_c) {
        constexpr bool FO = decltype(fold_c)::value;
        // (row offsets recomputed here from an opaque copy of the lane id: kept as kernel-long constants they cost three registers the paired
        // A-steps do not have)
        unsigned ones_ = ~0u;
        asm volatile("" : "+s"(ones_));  // (opaque: the lane id is recomputed HERE by v_mbcnt, not kept - or spilled - through the step loops)
        const int ln_ = (int)__builtin_amdgcn_mbcnt_hi(ones_, __builtin_amdgcn_mbcnt_lo(ones_, 0u));
        const int fk_ = ln_ >> 4;
        const int rows0_ = rg * 48 + (ln_ & 15);
        // (the row sums' exchanges address their partner from ln_ too: __shfl_xor's own lane id is one value for both LayerNorms of the fused
        // kernel, which the paired form then carries - in scratch - through the whole chunk loop)
        auto shx = [&](float v, int m) -> float { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((ln_ ^ m) << 2, __builtin_bit_cast(int, v))); };
        const unsigned row_ = (unsigned)(m0 + rows0_) * (unsigned)(E * 4);
        const unsigned v_rowx = row_ + (unsigned)fk_ * 16u;                                            // fp32 rows
        const unsigned v_rowh = row_ + (unsigned)fk_ * 8u;                                             // split rows: the lane's four hi halves (lo: + 64)
        const unsigned v_rowh2 = row_ + (unsigned)(fk_ >> 1) * 16u + (unsigned)(fk_ & 1) * 64u;       // row-pair form: 16-byte hi chunk (even f_kg) / lo chunk (odd)
        (void)v_rowh;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(x_dst, 0, p.h_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(h_dst, 0, p.h_bytes, 0x00020000);
        float* stat = reinterpret_cast<float*>(smem + OFF_G);
        float mean[3], rstd[3];
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) {
            float sm = 0.f;
#pragma unroll
            for (int cf = 0; cf < 6; ++cf) {
                const f32x4 v = acc[rf][cf];
                sm += (v[0] + v[1]) + (v[2] + v[3]);
            }
            sm += shx(sm, 16);
            sm += shx(sm, 32);
            if (fk_ == 0) stat[cg * BM + rows0_ + rf * 16] = sm;
        }
        __syncthreads();
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) {
            const int r = rows0_ + rf * 16;
            mean[rf] = ((stat[r] + stat[BM + r]) + (stat[2 * BM + r] + stat[3 * BM + r])) * (1.0f / E);
            float q = 0.f;
#pragma unroll
            for (int cf = 0; cf < 6; ++cf)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dd = acc[rf][cf][k] - mean[rf];
                    q = __builtin_fmaf(dd, dd, q);
                }
            q += shx(q, 16);
            q += shx(q, 32);
            if (fk_ == 0) stat[(4 + cg) * BM + r] = q;
        }
        __syncthreads();
#if FFD_STAMP == 2
        stamp();  // (fine) row statistics done
#endif
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) {
            const int r = rows0_ + rf * 16;
            const float var = ((stat[4 * BM + r] + stat[5 * BM + r]) + (stat[6 * BM + r] + stat[7 * BM + r])) * (1.0f / E);
            rstd[rf] = 1.0f / sqrtf(var + p.eps);
            if (FO && fk_ == 0 && cg == 0 && m0 + r < p.M) {
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                *reinterpret_cast<f32x2_t*>(p.stats_out + (size_t)(m0 + r) * 2) = f32x2_t{mean[rf], rstd[rf]};
            }
        }
        // gamma / beta of all six column fragments in one round trip (48 registers: the operand fragments are dead): inside the store loop every
        // pair of loads sits behind the previous fragment's buffer stores - six dependent trips to the L2 per LayerNorm
        f32x4 gs_[6], bs_[6];
#pragma unroll
        for (int cf = 0; cf < 6; ++cf) {
            const int cb = (cf / 3) * 192 + cg * 48 + (cf % 3) * 16;
            gs_[cf] = FO ? f32x4{1.f, 1.f, 1.f, 1.f} : *reinterpret_cast<const f32x4*>(gamma + cb + fk_ * 4);
            bs_[cf] = FO ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(beta + cb + fk_ * 4);
        }
#pragma unroll
        for (int cf = 0; cf < 6; ++cf) {
            const int cb = (cf / 3) * 192 + cg * 48 + (cf % 3) * 16;  // (wave-uniform)
            const f32x4 g = gs_[cf], b = bs_[cf];
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) {
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                const f32x4 v = acc[rf][cf];
                float mu = mean[rf];
                const float rs = rstd[rf];
                asm("" : "+v"(mu));  // (a second, opaque copy: with the same value as in the variance pass the compiler keeps all 72 differences v - mean alive from there to here)
                f32x4 hv = {(v[0] - mu) * rs * g[0] + b[0], (v[1] - mu) * rs * g[1] + b[1], (v[2] - mu) * rs * g[2] + b[2],
                            (v[3] - mu) * rs * g[3] + b[3]};
                if (FO) hv = f32x4{v[0] - mu, v[1] - mu, v[2] - mu, v[3] - mu};  // (the CENTERED rows: the consumer's projection needs no mean * colsum term)
#pragma unroll
                for (int j = 0; j < 4; ++j) { float t = hv[j]; split_pin(t); hv[j] = t; }
                if (store_x) {
                    const u32x4 vq = __builtin_bit_cast(u32x4, v);
                    __builtin_amdgcn_raw_buffer_store_b128(vq, rx, v_rowx + rf * (16 * E * 4), cb * 4, 0);
                    asm volatile("s_nop 3" ::"v"(vq));  // (wait states behind a 16-byte buffer store: scripts/micro/mubuf_store_hazard.hip)
                }
                // WAIT STATES BEHIND 16-BYTE BUFFER STORES. A buffer_store_dwordx4 reads its data registers one cycle late for lanes
                // 12 - 15 of every row; an instruction that writes one of them directly behind the store changes what those lanes store.
                // The hardware wants one wait state (SGPR soffset) or two (immediate) there - scripts/micro/mubuf_store_hazard.hip
                // shows it in isolation - and the compiler inserts none for the SGPR form, which is the form used here. Seen as a
                // handful of stale 4-byte words per launch on a full chip (first behind the v_permlane16_swap below and wrongly blamed
                // on the swap; then in pp_linear_dma.hip's fp32 rows, where no swap is involved). Every 16-byte buffer store of this
                // file is followed by an explicit s_nop that depends on its data.
                // (hi, lo) of the four values by split_pair: eight VALU instructions instead of fourteen (the stamps of round 5 found the LayerNorm
                // epilogues still on the one-value-at-a-time split)
                u32x2_t hu, lu;
                { unsigned h__, l__; split_pair(hv[0], hv[1], h__, l__); hu[0] = h__; lu[0] = l__; }
                { unsigned h__, l__; split_pair(hv[2], hv[3], h__, l__); hu[1] = h__; lu[1] = l__; }
                const int so = (cb >> 5) * 128 + (cb & 16) * 2;
                {   // the row-pair form of split_store4_rowpair (lanes fk_, fk_ ^ 1 exchange halves: the even one stores the 16-byte hi chunk,
                    // the odd one the lo chunk), as ONE 16-byte buffer store followed by the wait states the compiler does not insert
                    const auto s0 = __builtin_amdgcn_permlane16_swap(hu[0], lu[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(hu[1], lu[1], false, false);
                    u32x4 q = {s0[0], s1[0], s0[1], s1[1]};
                    __builtin_amdgcn_raw_buffer_store_b128(q, rh, v_rowh2 + rf * (16 * E * 4), so, 0);
                    asm volatile("s_nop 3" ::"v"(q));
                }
            }
        }
    };

    // the residual rows, straight into the accumulators
    float res_mean[3] = {0.f, 0.f, 0.f};
    if constexpr ((FOLD & 1) != 0) {
#pragma unroll
        for (int rf = 0; rf < 3; ++rf) {
            const int r = m0 + rows0 + rf * 16;
            res_mean[rf] = r < p.M ? p.res_stats[(size_t)r * 2] : 0.f;
        }
    }
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual), 0, p.h_bytes, 0x00020000);
    const unsigned v_rowx = (unsigned)(m0 + rows0) * (unsigned)(E * 4) + (unsigned)f_kg * 16u;  // (dead after these loads)
#pragma unroll
    for (int rf = 0; rf < 3; ++rf)
#pragma unroll
        for (int cf = 0; cf < 6; ++cf) {
            const int cb = (cf / 3) * 192 + cg * 48 + (cf % 3) * 16;
            if constexpr ((FOLD & 1) != 0) {
                // operand-format rows, row-pair form: the even lane of a pair (f_kg, f_kg ^ 1) fetches the 16-byte hi chunk of both lanes' values, the odd
                // one the lo chunk; two row swaps give every lane its own (hi, lo) halves; x = hi + lo (22 significant bits)
                const unsigned vsp = (unsigned)(m0 + rows0) * (unsigned)(E * 4) + (unsigned)(f_kg >> 1) * 16u + (unsigned)(f_kg & 1) * 64u;
                const u32x4 q = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, vsp + rf * (16 * E * 4), (cb >> 5) * 128 + (cb & 16) * 2, 0));
                const auto s0 = __builtin_amdgcn_permlane16_swap(q[0], q[2], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(q[1], q[3], false, false);
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                const f16x4 hh = __builtin_bit_cast(f16x4, u32x2_t{s0[0], s1[0]}), ll = __builtin_bit_cast(f16x4, u32x2_t{s0[1], s1[1]});
                // (centered rows: the row's mean comes back here - after hi + lo, the order the producer's x - mean inverts)
                const float mu = res_mean[rf];
                acc[rf][cf] = f32x4{((float)hh[0] + (float)ll[0]) + mu, ((float)hh[1] + (float)ll[1]) + mu, ((float)hh[2] + (float)ll[2]) + mu,
                                    ((float)hh[3] + (float)ll[3]) + mu};
            } else {
                acc[rf][cf] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, v_rowx + rf * (16 * E * 4), cb * 4, 0));  // (rows past M: zeros, never stored)
            }
        }

    stamp();  // 1: residual rows requested
    // the accumulators collect products with weights stored as w * s: the fp32 values they start from are scaled to match (exact: powers of two)
    {
        const float s_in = PROJ ? p.s_p : p.s_2;
#pragma unroll
        for (int cf = 0; cf < 6; ++cf) {
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if constexpr (!PROJ) bv = *reinterpret_cast<const f32x4*>(p.b2 + (cf / 3) * 192 + cg * 48 + (cf % 3) * 16 + f_kg * 4);
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) acc[rf][cf] = (acc[rf][cf] + bv) * s_in;  // (!PROJ: + b2 before the first B-step accumulates)
        }
    }
    if constexpr (PROJ) {
        // ---- attention output projection + residual, then ln2:  acc <- x + att Wp^T + bp ;  h <- LN2(acc). 24 steps shaped like
        // the B-steps: step s = 2 kb + half takes the Wp half block from ring slot s & 3, the attention rows' k-block kb from G buffer kb & 3
        // (rolling fragment reads as in the FFN loop below: the same three sweeps, the barrier of step s + 1 behind the second MFMA of step s)
        {
            u32x4 pwh[3], pwl[3];
            step_barrier();  // the barrier of step 0
            {
                const int ub = opaque_s(u_b), ug = opaque_s(u_g);
#pragma unroll
                for (int nf = 0; nf < 3; ++nf) { pwh[nf] = rd(lane_hi, ub, nf * 2048); pwl[nf] = rd(lane_lo, ub, nf * 2048); }
#pragma unroll
                for (int rf = 0; rf < 3; ++rf) { bgl[rf] = rd(lane_lo, ug, rf * 2048); bgh[rf] = rd(lane_hi, ug, rf * 2048); }
            }
#pragma unroll 1
            for (int kp = 0; kp < NPROJ / 4; ++kp) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int half = q & 1;
                    const int ub_n = opaque_s(u_b + ((q + 1) & 3) * SLOTB);
                    const int ug_n = opaque_s(u_g + ((((4 * kp + q) >> 1) + 1) & 3) * G_KB);
                    const bool nb = q < 3 || kp + 1 < NPROJ / 4;  // a step follows
                    const bool newg = nb && half == 1;             // ... of the next k-block: new attention-row fragments
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 9; ++i) {  // hi x lo
                        const int rf = i / 3, nf = i % 3;
                        acc[rf][half * 3 + nf] = mma(pwh[nf], bgl[rf], acc[rf][half * 3 + nf]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (i == 1 && nb) step_barrier();
                        if (nf == 2 && newg) bgl[rf] = rd(lane_lo, ug_n, rf * 2048);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int i = 0; i < 9; ++i) {  // hi x hi
                        const int rf = i / 3, nf = i % 3;
                        acc[rf][half * 3 + nf] = mma(pwh[nf], bgh[rf], acc[rf][half * 3 + nf]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (rf == 2 && nb) pwh[nf] = rd(lane_hi, ub_n, nf * 2048);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int i = 0; i < 9; ++i) {  // lo x hi
                        const int rf = i / 3, nf = i % 3;
                        acc[rf][half * 3 + nf] = mma(pwl[nf], bgh[rf], acc[rf][half * 3 + nf]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (nf == 2 && newg) bgh[rf] = rd(lane_hi, ug_n, rf * 2048);
                        if (rf == 2 && nb) pwl[nf] = rd(lane_lo, ub_n, nf * 2048);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        // + bp (after the sums, as residual + (sum + bias) rounds closest to the reference's x + proj(...))
#pragma unroll
        for (int cf = 0; cf < 6; ++cf) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bp + (cf / 3) * 192 + cg * 48 + (cf % 3) * 16 + f_kg * 4);
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) acc[rf][cf] = acc[rf][cf] * p.inv_p + bv;
        }
        stamp();  // 2: projection steps done
        __syncthreads();  // P1: the G buffers are out of use
#if FFD_STAMP == 2
        stamp();  // (fine) P1 passed
#endif
        layernorm_rows(p.gamma2, p.beta2, nullptr, const_cast<void*>(p.h), false, TagF{});
#if FFD_STAMP == 2
        stamp();  // (fine) ln2 rows stored (issued)
#endif
        // the rows must be in L2 before the DMA waves ask for them (a store counts in vmcnt until the L2 has acknowledged it)
        __builtin_amdgcn_s_waitcnt((7 << 4) | (0 << 8) | (0));
#if FFD_STAMP == 2
        stamp();  // (fine) stores acknowledged
#endif
        __syncthreads();  // P2
        stamp();  // 3: ln2 done, rows in L2
    }

    // + b2 before the first B-step accumulates, in the scale of the W2 products
    if constexpr (PROJ) {
#pragma unroll
        for (int cf = 0; cf < 6; ++cf) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.b2 + (cf / 3) * 192 + cg * 48 + (cf % 3) * 16 + f_kg * 4);
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) acc[rf][cf] = (acc[rf][cf] + bv) * p.s_2;
        }
    }
    auto load_b1 = [&](int ci) {
        const int c = chunk_of(ci < nchunks ? ci : 0);
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) b1v[nf] = *reinterpret_cast<const f32x4*>(p.b1 + c * CHUNK + cg * 32 + nf * 16 + f_kg * 4);
    };
    load_b1(0);

    // ---- GELU(P) of a chunk -> (hi, lo) -> G tile
    auto gelu_chunk = [&](f32x4 (&pa)[3][2]) __attribute__((always_inline)) {
        // (hi, lo) pairs of GELU(P) -> G tile (gelu_erfc_as of pp_split.h, Abramowitz & Stegun 7.1.26: same operations in the same
        // order per value as pp_ffn_split.hip). The G tile is free: the previous chunk's B-steps ended before this chunk's A-steps.
        // Lane holds units 32 cg + 16 nf + 4 f_kg + (0..3) of its rows: k-block cg of the chunk, 16-byte chunk 2 nf + (f_kg >> 1)
        // (+ 4 for lo), upper or lower 8 bytes.
#pragma unroll
        for (int rf = 0; rf < 3; ++rf)
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                int ln_ = lane;
                asm volatile("" : "+v"(ln_));  // (addresses recomputed per fragment: no address register lives through the step loops)
                const int fk_ = ln_ >> 4, sw_ = ln_ & 7;
                char* gs = smem + OFF_G + cg * G_KB + (rg * 48 + (ln_ & 15) + rf * 16) * 128 + (fk_ & 1) * 8;
                const int c = 2 * nf + (fk_ >> 1);
                const int sw = sw_;
                // Round 5: the chunk's GELU is ~3.4 k of its ~22 k cycles (stamps of scripts/micro/ffn12d.hip -DSTAMP=1), VALU-bound with both
                // computing waves of a SIMD issuing at once, and it cannot be hidden behind MFMAs of the same SIMD.
                // What is left is fewer issue cycles per value (scripts/micro/valu_rate.hip, cycles per wave64 instruction and SIMD: plain fp32
                // 3.0, v_pk_*_f32 4.9 for TWO values, v_rcp / v_exp 8.5, v_cvt 4.5, v_cvt_pk_f16_f32 and v_fma_mix_f32 4.7): the same A & S 7.1.26
                // form on value PAIRS with packed fp32 instructions (constants folded: 1 + p z = 1 + (p / sqrt 2) |x|, exp(-z^2) =
                // exp2(-(x sqrt(log2(e) / 2))^2), max(x, 0) = 0.5 x + 0.5 |x| exactly) and the lo half as g - hi by ONE v_fma_mix_f32 that
                // reads hi straight from the packed fp16 pair (exact: the same difference as convert-back-and-subtract): ~56 instead of ~76
                // issue cycles per value. |difference to the unpacked form| ~1e-7 relative (rounding of the folded constants).
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                unsigned hq[2], lq[2];
#pragma unroll
                for (int u0 = 0; u0 < 2; ++u0) {
                    const f32x2 x = f32x2{pa[rf][nf][2 * u0], pa[rf][nf][2 * u0 + 1]} * p.inv_1;
                    const f32x2 ax = __builtin_elementwise_abs(x);
                    const f32x2 d = ax * 0.23164189265f + 1.0f;  // 0.3275911 / sqrt 2
                    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
                    const f32x2 uu = x * 0.84932180028801904272f;  // sqrt(log2(e) / 2)
                    const f32x2 u2 = uu * uu;
                    const f32x2 e = {__builtin_amdgcn_exp2f(-u2[0]), __builtin_amdgcn_exp2f(-u2[1])};
                    f32x2 q = t * 1.061405429f + -1.453152027f;
                    q = t * q + 1.421413741f;
                    q = t * q + -0.284496736f;
                    q = t * q + 0.254829592f;
                    const f32x2 erfc_z = (t * q) * e;
                    const f32x2 ma = ax * -0.5f;
                    const f32x2 mx = x * 0.5f - ma;  // = max(x, 0), exactly
                    f32x2 g = ma * erfc_z + mx;
                    asm("" : "+v"(g));  // (split_pin: no fusion of the arithmetic into the conversions)
                    const f16x2 hp = __builtin_convertvector(g, f16x2);  // v_cvt_pk_f16_f32, round to nearest even
                    const unsigned hu = __builtin_bit_cast(unsigned, hp);
                    f32x2 l;
                    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l[0]) : "v"(g[0]), "v"(hu));
                    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l[1]) : "v"(g[1]), "v"(hu));
                    hq[u0] = hu;
                    lq[u0] = __builtin_bit_cast(unsigned, __builtin_convertvector(l, f16x2));
                }
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2_t*>(gs + ((c ^ sw) << 4)) = u32x2_t{hq[0], hq[1]};
                *reinterpret_cast<u32x2_t*>(gs + (((4 + c) ^ sw) << 4)) = u32x2_t{lq[0], lq[1]};
            }
    };

    if constexpr (PAIR) {
        // PAIRED CHUNKS (round 5; needs an even number of chunks). The 96 x 384 block of x rows is re-streamed once per hidden chunk: 12 x 144 KiB per
        // workgroup, 4.6 MB of rows per XCD and pass through a 4 MB L2 - about half of those re-reads come from beyond the L2 (counter traffic
        // 395 MB per launch against 151 MB algorithmic), and the fill ablations of scripts/micro/ffn12d.hip price the x pieces at 9 % of the loop's
        // TIME (the chip is power-limited: same cycles, 1.68 -> 1.82 GHz without them; 7.5 % of it is their coming from beyond the L2). Two chunks
        // now share every x k-block: the A-steps of a pair run  [W1(c0, kb) + x(kb)] [W1(c1, kb)]  for kb = 0 .. 11, and with the rolling reads the
        // x fragments simply STAY in their registers for the second step - no second read of the slot, no slot lifetime problem - so x is
        // streamed six times per launch instead of twelve and the A-steps read a quarter less from LDS. Price: the second chunk's accumulators
        // (24 registers) live through the first chunk's GELU and B-steps; the B-steps' fragments leave room for them (72 + 24 + 48 = 144).
        f32x4 pacc1[3][2];  // P of the pair's second chunk (pacc: the first)
        u32x4 wh[2], wl[2], xh[3], xl[3], bwh[3], bwl[3];
        const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b1), 0, (unsigned)p.F * 4u, 0x00020000);
        auto load_b1x = [&](int ci, f32x4 (&dst)[2]) __attribute__((always_inline)) {  // (descriptor + lane & 0x30: no 64-bit address kept alive through the B-steps)
            const int c = chunk_of(ci < nchunks ? ci : 0);
            int vo = lane;
            asm volatile("" : "+v"(vo));
            vo &= 0x30;
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) dst[nf] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb1, vo, (c * CHUNK + cg * 32 + nf * 16) * 4, 0));
        };
        f32x4 b1w[2];  // the first chunk's bias is b1v (loaded above), the second's b1w
        load_b1x(1, b1w);
        step_barrier();  // the barrier of step 0
        {
            const int ua = opaque_s(u_a), ux = opaque_s(u_x);
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) { wh[nf] = rd(lane_hi, ua, nf * 2048); wl[nf] = rd(lane_lo, ua, nf * 2048); }
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) { xl[rf] = rd(lane_lo, ux, rf * 2048); xh[rf] = rd(lane_hi, ux, rf * 2048); }
        }
        // one A-step: the three sweeps into `pa`; `reload_x`: the row fragments are re-read for the next step (second step of a k-block);
        // `to_b`: the next step is a B-step (its weight fragments instead of W1's)
        auto a_step = [&](f32x4 (&pa)[3][2], int slot_next, bool reload_x, bool to_b) __attribute__((always_inline)) {
            const int sn = (slot_next & 3) * SLOTB;
            const int ua_n = opaque_s(u_a + sn), ux_n = opaque_s(u_x + sn), ub_n = opaque_s(u_b + sn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 6; ++i) {  // hi x lo
                const int rf = i >> 1, nf = i & 1;
                mma_ip(wh[nf], xl[rf], pa[rf][nf]);
                __builtin_amdgcn_sched_barrier(0);
                if (i == 1) step_barrier();  // the barrier of the next step
                if (reload_x && !to_b && nf == 1) xl[rf] = rd(lane_lo, ux_n, rf * 2048);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {  // hi x hi
                const int rf = i >> 1, nf = i & 1;
                mma_ip(wh[nf], xh[rf], pa[rf][nf]);
                __builtin_amdgcn_sched_barrier(0);
                if (rf == 2) { if (!to_b) wh[nf] = rd(lane_hi, ua_n, nf * 2048); else bwh[nf] = rd(lane_hi, ub_n, nf * 2048); }
                if (to_b && i == 5) bwh[2] = rd(lane_hi, ub_n, 2 * 2048);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {  // lo x hi
                const int rf = i >> 1, nf = i & 1;
                mma_ip(wl[nf], xh[rf], pa[rf][nf]);
                __builtin_amdgcn_sched_barrier(0);
                if (reload_x && !to_b && nf == 1) xh[rf] = rd(lane_hi, ux_n, rf * 2048);
                if (rf == 2) { if (!to_b) wl[nf] = rd(lane_lo, ua_n, nf * 2048); else bwl[nf] = rd(lane_lo, ub_n, nf * 2048); }
                if (to_b && i == 5) bwl[2] = rd(lane_lo, ub_n, 2 * 2048);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // the eight B-steps of one chunk (its G tile complete, bwh / bwl of its first step in registers). `next`: what follows the last step -
        // 0 the other chunk's B-steps (weights only: its G fragments come behind its GELU), 1 the next pair's first A-step. `more` (run time):
        // there IS a next pair - the launch's very last step has no barrier inside (the epilogue's E1 is the DMA waves' next one); its
        // prefetches still run (they read a slot of fillers): every path through the loop body redefines the fragment registers, so that
        // none of them is carried, spilled, through the GELU and the B-steps
        auto b_phase = [&](int t0, int next, bool more) __attribute__((always_inline)) {
            {
                const int ug = opaque_s(u_g);
#pragma unroll
                for (int rf = 0; rf < 3; ++rf) { bgl[rf] = rd(lane_lo, ug, rf * 2048); bgh[rf] = rd(lane_hi, ug, rf * 2048); }
            }
            // (the k-blocks as a run-time loop over one two-step body - the last one apart, for what follows it: forty fully unrolled steps
            // per pair sent the register allocator into spilling half the accumulators)
            auto two_steps = [&](int jb, bool last_kb) __attribute__((always_inline)) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int sb = 2 * jb + half;
                    const int sn = ((t0 + sb + 1) & 3) * SLOTB;
                    const int ub_n = opaque_s(u_b + sn), ua_n = opaque_s(u_a + sn), ux_n = opaque_s(u_x + sn);
                    const int ug_n = opaque_s(u_g + ((jb + 1) & 3) * G_KB);
                    const bool nb = !(last_kb && half == 1);
                    const bool newg = nb && half == 1;
                    const bool wnext = nb || next == 0;   // B-step weights follow
                    const bool anext = !nb && next == 1;  // an A-step follows
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 9; ++i) {  // hi x lo
                        const int rf = i / 3, nf = i % 3;
                        mma_ip(bwh[nf], bgl[rf], acc[rf][half * 3 + nf]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (i == 1 && (nb || next == 0 || more)) step_barrier();
                        if (nf == 2) { if (newg) bgl[rf] = rd(lane_lo, ug_n, rf * 2048); else if (anext) xl[rf] = rd(lane_lo, ux_n, rf * 2048); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int i = 0; i < 9; ++i) {  // hi x hi
                        const int rf = i / 3, nf = i % 3;
                        mma_ip(bwh[nf], bgh[rf], acc[rf][half * 3 + nf]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (rf == 2) { if (wnext) bwh[nf] = rd(lane_hi, ub_n, nf * 2048); else if (anext && nf < 2) wh[nf] = rd(lane_hi, ua_n, nf * 2048); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int i = 0; i < 9; ++i) {  // lo x hi
                        const int rf = i / 3, nf = i % 3;
                        mma_ip(bwl[nf], bgh[rf], acc[rf][half * 3 + nf]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (nf == 2) { if (newg) bgh[rf] = rd(lane_hi, ug_n, rf * 2048); else if (anext) xh[rf] = rd(lane_hi, ux_n, rf * 2048); }
                        if (rf == 2) { if (wnext) bwl[nf] = rd(lane_lo, ub_n, nf * 2048); else if (anext && nf < 2) wl[nf] = rd(lane_lo, ua_n, nf * 2048); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
#pragma unroll 1
            for (int jb = 0; jb < NB / 2 - 1; ++jb) two_steps(jb, false);
            two_steps(NB / 2 - 1, true);
        };
        const int npairs = nchunks >> 1;
        for (int cp = 0; cp < npairs; ++cp) {
#pragma unroll
            for (int rf = 0; rf < 3; ++rf)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) { pacc[rf][nf] = b1v[nf] * p.s_1; pacc1[rf][nf] = b1w[nf] * p.s_1; }
            valu_settle(pacc);  // (VALU writes -> MFMA SrcC reads: the wait states the compiler cannot know an asm statement needs)
            valu_settle(pacc1);
            // ---- A-steps of both chunks, k-block by k-block
#pragma unroll 1
            for (int kbi = 0; kbi < NA - 1; ++kbi) {  // (one k-block's two steps; the last k-block apart: B-steps follow it)
                a_step(pacc, 2 * kbi + 1, false, false);
                a_step(pacc1, 2 * kbi + 2, true, false);
            }
            a_step(pacc, 2 * NA - 1, false, false);
            a_step(pacc1, 2 * NA, true, true);
            stamp();  // 4 + 5 cp: A-steps of the pair done
            mfma_settle(pacc);
            gelu_chunk(pacc);
            step_barrier();  // the G tile is complete (the DMA waves pass it behind their barrier of step 2 NA)
            stamp();  // + 1: GELU of chunk 0
            load_b1x(2 * cp + 2, b1v);  // the next pair's biases, asked for while their sixteen registers are free
            b_phase(2 * NA, 0, true);
            stamp();  // + 2: B-steps of chunk 0
            mfma_settle(pacc1);
            gelu_chunk(pacc1);
            step_barrier();  // (behind the DMA waves' barrier of step 2 NA + NB)
            stamp();  // + 3: GELU of chunk 1
            load_b1x(2 * cp + 3, b1w);
            b_phase(2 * NA + NB, 1, cp + 1 < npairs);
            stamp();  // + 4: B-steps of chunk 1
        }
        mfma_settle(acc);
    } else {
    {
        // ROLLING FRAGMENT READS (round 5). In the loop below a step is [barrier | ten reads | wait | 18 / 27 MFMAs]: the reads' latency and the
        // barrier skew sit in front of every step's MFMAs - stamps of scripts/micro/ffn12d.hip: 850 - 880 cycles per A-step against 576 of MFMA
        // issue per SIMD, 1030 - 1100 per B-step against 864. There is no register set to read a step ahead into (168 registers), but none is
        // needed: a fragment register is re-read for the NEXT step right behind the last MFMA of THIS step that uses it. The three sweeps run
        // hi x lo, hi x hi, lo x hi, so the operands of the next step's first sweep (weights hi, rows lo) are free - and re-read - earliest,
        // twelve or more MFMAs ahead of their use. The barrier of step t + 1 moves INTO step t (behind its second MFMA: everything this wave read
        // from slot t is in registers since step t - 1; what it reads behind the barrier comes from slot t + 1, landed). The DMA waves' protocol
        // is unchanged: their barrier s still means "step s has landed, its predecessor's slot is read out". Skeleton: 640 - 660 / 900 - 960
        // cycles per A- / B-step, 234 k instead of 270 k cycles per launch (scripts/micro/ffn12d.hip -DROLL=1). The hi x lo product is now
        // added first (it was last): same terms, another rounding order.
        u32x4 wh[2], wl[2], xh[3], xl[3];  // A-step fragments
        u32x4 bwh[3], bwl[3];              // B-step weight fragments (bgh / bgl: the G fragments)
        step_barrier();  // the barrier of step 0
        {
            const int ua = opaque_s(u_a), ux = opaque_s(u_x);
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) { wh[nf] = rd(lane_hi, ua, nf * 2048); wl[nf] = rd(lane_lo, ua, nf * 2048); }
#pragma unroll
            for (int rf = 0; rf < 3; ++rf) { xl[rf] = rd(lane_lo, ux, rf * 2048); xh[rf] = rd(lane_hi, ux, rf * 2048); }
        }
        for (int ci = 0; ci < nchunks; ++ci) {
#pragma unroll
            for (int rf = 0; rf < 3; ++rf)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) pacc[rf][nf] = b1v[nf] * p.s_1;
            load_b1(ci + 1);
            // ---- A-steps: P += x[:, kb] W1[chunk, kb]^T, wave tile 48 rows x 32 units
#pragma unroll
            for (int t = 0; t < NA; ++t) {
                const int sn = ((t + 1) & 3) * SLOTB;  // slot of the next step (t + 1 == NA: the first B-step's, weights only - its G fragments follow the GELU)
                const int ua_n = opaque_s(u_a + sn), ux_n = opaque_s(u_x + sn), ub_n = opaque_s(u_b + sn);
                const bool na = t + 1 < NA;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 6; ++i) {  // hi x lo
                    const int rf = i >> 1, nf = i & 1;
                    pacc[rf][nf] = mma(wh[nf], xl[rf], pacc[rf][nf]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (i == 1) step_barrier();  // the barrier of step t + 1
                    if (na && nf == 1) xl[rf] = rd(lane_lo, ux_n, rf * 2048);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) {  // hi x hi
                    const int rf = i >> 1, nf = i & 1;
                    pacc[rf][nf] = mma(wh[nf], xh[rf], pacc[rf][nf]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (rf == 2) { if (na) wh[nf] = rd(lane_hi, ua_n, nf * 2048); else bwh[nf] = rd(lane_hi, ub_n, nf * 2048); }
                    if (!na && i == 5) bwh[2] = rd(lane_hi, ub_n, 2 * 2048);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) {  // lo x hi
                    const int rf = i >> 1, nf = i & 1;
                    pacc[rf][nf] = mma(wl[nf], xh[rf], pacc[rf][nf]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (na && nf == 1) xh[rf] = rd(lane_hi, ux_n, rf * 2048);
                    if (rf == 2) { if (na) wl[nf] = rd(lane_lo, ua_n, nf * 2048); else bwl[nf] = rd(lane_lo, ub_n, nf * 2048); }
                    if (!na && i == 5) bwl[2] = rd(lane_lo, ub_n, 2 * 2048);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            gelu_chunk(pacc);
            step_barrier();  // the G tile is complete (an extra barrier: the DMA waves pass it behind their barrier of step NA)
            {
                const int ug = opaque_s(u_g);
#pragma unroll
                for (int rf = 0; rf < 3; ++rf) { bgl[rf] = rd(lane_lo, ug, rf * 2048); bgh[rf] = rd(lane_hi, ug, rf * 2048); }
            }
            // ---- B-steps (j, half): acc[:, half] += G[:, j] W2[half, chunk j]^T, wave tile 48 rows x 48 outputs
#pragma unroll
            for (int sb = 0; sb < NB; ++sb) {
                const int half = sb & 1;
                const int sn = ((NA + sb + 1) & 3) * SLOTB;
                const int ub_n = opaque_s(u_b + sn), ua_n = opaque_s(u_a + sn), ux_n = opaque_s(u_x + sn);
                const int ug_n = opaque_s(u_g + (((sb >> 1) + 1) & 3) * G_KB);
                const bool nb = sb + 1 < NB;         // the next step is a B-step ...
                const bool newg = nb && half == 1;   // ... of the next k-block: new G fragments
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 9; ++i) {  // hi x lo
                    const int rf = i / 3, nf = i % 3;
                    acc[rf][half * 3 + nf] = mma(bwh[nf], bgl[rf], acc[rf][half * 3 + nf]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (i == 1 && (nb || ci + 1 < nchunks)) step_barrier();  // (the launch's last step has no successor: the epilogue's E1 is the DMA waves' next barrier)
                    if (nf == 2) { if (newg) bgl[rf] = rd(lane_lo, ug_n, rf * 2048); else if (!nb) xl[rf] = rd(lane_lo, ux_n, rf * 2048); }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 9; ++i) {  // hi x hi
                    const int rf = i / 3, nf = i % 3;
                    acc[rf][half * 3 + nf] = mma(bwh[nf], bgh[rf], acc[rf][half * 3 + nf]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (rf == 2) { if (nb) bwh[nf] = rd(lane_hi, ub_n, nf * 2048); else if (nf < 2) wh[nf] = rd(lane_hi, ua_n, nf * 2048); }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 9; ++i) {  // lo x hi
                    const int rf = i / 3, nf = i % 3;
                    acc[rf][half * 3 + nf] = mma(bwl[nf], bgh[rf], acc[rf][half * 3 + nf]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (nf == 2) { if (newg) bgh[rf] = rd(lane_hi, ug_n, rf * 2048); else if (!nb) xh[rf] = rd(lane_hi, ux_n, rf * 2048); }
                    if (rf == 2) { if (nb) bwl[nf] = rd(lane_lo, ub_n, nf * 2048); else if (nf < 2) wl[nf] = rd(lane_lo, ua_n, nf * 2048); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    }
    // ---- LayerNorm epilogue: G is out of use once every wave is past its last B-step
    __syncthreads();  // E1
    stamp();  // last but one: every wave past its last step
#pragma unroll
    for (int rf = 0; rf < 3; ++rf)
#pragma unroll
        for (int cf = 0; cf < 6; ++cf) acc[rf][cf] *= p.inv_2;  // (the W2 products carried the scale of W2)
    if constexpr ((FOLD & 2) != 0) layernorm_rows(nullptr, nullptr, nullptr, p.h_out, false, TagT{});
    else layernorm_rows(p.gamma, p.beta, p.x_out, p.h_out, true, TagF{});
    stamp();  // last: rows stored
}

template <bool PROJ, bool PAIR, int FOLD = 0>
__device__ __forceinline__ void body(const Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM;
    const int nchunks = p.F / CHUNK;
    const int c_rot = (int)(blockIdx.x & 7) % nchunks;  // the workgroups of an XCD walk the chunks in the same order
    if (wv >= CW) dma_role<PROJ, PAIR>(p, smem, wv - CW, lane, m0, nchunks, c_rot);
    else compute_role<PROJ, PAIR, FOLD>(p, smem, wv, lane, m0, nchunks, c_rot);
}

__global__ __launch_bounds__(THREADS) void ffn_dma_kernel(const Params p) { body<false, false>(p); }
__global__ __launch_bounds__(THREADS) void proj_ffn_dma_kernel(const Params p) { body<true, false>(p); }
__global__ __launch_bounds__(THREADS) void ffn_dma_pair_kernel(const Params p) { body<false, true>(p); }
__global__ __launch_bounds__(THREADS) void proj_ffn_dma_pair_kernel(const Params p) { body<true, true>(p); }
__global__ __launch_bounds__(THREADS) void proj_ffn_dma_pair_fold1_kernel(const Params p) { body<true, true, 1>(p); }  // split residual in, LayerNorm out (last layer)
__global__ __launch_bounds__(THREADS) void proj_ffn_dma_pair_fold2_kernel(const Params p) { body<true, true, 2>(p); }  // fp32 residual in (first layer), folded out
__global__ __launch_bounds__(THREADS) void proj_ffn_dma_pair_fold3_kernel(const Params p) { body<true, true, 3>(p); }  // split in, folded out

}  // namespace ffd

#if FFD_STAMP
extern "C" int pp_dev_ffd_stamps(unsigned long long* out) {  // dev: 2 x 64 stamps of the last launch (host pointer)
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pp::ffd::g_ffd_stamps), sizeof(unsigned long long) * 128, 0, hipMemcpyDeviceToHost);
}
#endif

namespace ffs {
// called from the entry points in pp_ffn_split.hip
int launch_dma_form(const Params& p, bool proj, hipStream_t s) {
    // an even number of hidden chunks: the paired form (two chunks share every streamed x k-block); option "ffn_pair" = 0 or an odd count: one at a time
    const bool pair = option("ffn_pair") != 0 && (p.F / ffd::CHUNK) % 2 == 0;
    auto kern = pair ? (proj ? ffd::proj_ffn_dma_pair_kernel : ffd::ffn_dma_pair_kernel) : (proj ? ffd::proj_ffn_dma_kernel : ffd::ffn_dma_kernel);
    const int fold = (p.res_split ? 1 : 0) | (p.fold_out ? 2 : 0);
    if (fold) {  // pp_proj_ffn_split_folded: the paired projection kernel only (its entry point checks)
        kern = fold == 1 ? ffd::proj_ffn_dma_pair_fold1_kernel : fold == 2 ? ffd::proj_ffn_dma_pair_fold2_kernel : ffd::proj_ffn_dma_pair_fold3_kernel;
    }
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ffd::LDS));
    hipLaunchKernelGGL(kern, dim3((p.M + ffd::BM - 1) / ffd::BM), dim3(ffd::THREADS), ffd::LDS, s, p);
    PP_LAUNCH_CHECK_AS(fold ? "ffn_dma_fold" : pair ? "ffn_dma_pair" : "ffn_dma_single");
    return PP_OK;
}
}  // namespace ffs
}  // namespace pp
