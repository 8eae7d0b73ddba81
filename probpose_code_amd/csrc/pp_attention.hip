// Multi-head self-attention of the ViT backbone for gfx950 (mmpretrain VisionTransformer [3P]:
// softmax(q k^T / sqrt(hd)) v per head, no mask, no dropout in eval).
//
// Sequence lengths on this path are tiny (192 tokens @256x192, 432 @384x288) so K and V of one
// (crop, head) stay resident in LDS and each query tile sees the whole score row at once: no
// online-softmax rescaling, no second pass.
//
//   * one 256-thread workgroup per (crop-pass, head); the 4 waves split the 16-query tiles;
//   * Q fragments go straight from HBM to registers (each is used by exactly one wave);
//     K is staged row-major [S][hd], V transposed [hd][S] so that both MFMA operands are
//     K-contiguous 16-byte reads;
//   * S^T = K Q^T with K as the MFMA "A" operand: a lane ends up with 4 consecutive keys of one
//     query in each 16x16 tile, the softmax row reduction is 47 in-register ops + 2 DPP hops, and
//     P in that layout IS the "B" operand of the O^T = V^T P^T MFMA (the key order inside a
//     K-block is a permutation shared by both operands) -- P never leaves registers;
//   * operand precision is a template parameter (bf16 -> v_mfma_f32_16x16x32_bf16, fp32 ->
//     v_mfma_f32_16x16x4_f32); softmax statistics are fp32 in both;
//   * blockIdx is remapped so that the heads of one crop (adjacent columns of the same qkv
//     rows) land on the same XCD / L2.
#include "pp_common.h"
#include "pp_split.h"

namespace pp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int ATT_THREADS = 256;
#ifndef ATT_ABL  // dev (scripts/micro/att_ablate.sh): 1 no V transpose, 2 no math, 3 no K / V loads - timing only, wrong results
#define ATT_ABL 0
#endif

__device__ __forceinline__ f32x4 att_mma(const u32x4& a, const u32x4& b, f32x4 c, __bf16) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                   0, 0);
}
__device__ __forceinline__ f32x4 att_mma(const u32x4& a, const u32x4& b, f32x4 c, float) {
    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], c, 0, 0, 0);
    return c;
}

template <typename T, int HD, int NT>
struct AttCfg {
    static constexpr int S = NT * 16;
    static constexpr int NTP = (NT + 1) & ~1;           // key tiles padded to an even count
    static constexpr int SP = NTP * 16;
    static constexpr int SPV = SP + 8;                  // V^T row pitch (elements): 8 * odd -> conflict-free reads
    static constexpr int CH = 16 / (int)sizeof(T);      // elements per 16-byte chunk
    static constexpr int RC = HD / CH;                  // chunks per K row
    static constexpr int NG = HD / (4 * CH);            // 4-chunk groups (one MFMA K-block each) per row
    static constexpr int DT = HD / 16;                  // output d tiles
    static constexpr size_t K_BYTES = (size_t)SP * HD * sizeof(T);
    static constexpr size_t V_BYTES = (size_t)HD * SPV * sizeof(T);
    static constexpr size_t LDS = K_BYTES + V_BYTES;
};

template <int RC>
__device__ __forceinline__ int kswz(int row, int chunk) {
    return RC >= 8 ? (chunk ^ (row & (RC - 1))) : chunk;
}

template <typename T, int HD, int NT>
__global__ __launch_bounds__(ATT_THREADS, (NT <= 12 ? 3 : 1)) void attention_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                                int n_seq, int heads, float scale_log2e) {
    using C = AttCfg<T, HD, NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                                        // [SP][HD], chunk-swizzled
    T* Vt = reinterpret_cast<T*>(smem + C::K_BYTES);        // [HD][SPV]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;

    // XCD-aware remap: hardware round-robins consecutive block ids over the 8 XCDs
    int id = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) id = (id & 7) * (nblk >> 3) + (id >> 3);
    const int seq = id / heads, head = id - seq * heads;
    const int E = heads * HD;
    const size_t row_stride = (size_t)3 * E;
    const T* base = qkv + (size_t)seq * C::S * row_stride + (size_t)head * HD;

    // ---- Q fragments of all of this wave's query tiles first: their HBM latency hides under the K/V staging
    constexpr int QPW = (NT + ATT_THREADS / 64 - 1) / (ATT_THREADS / 64);
    u32x4 qf_all[QPW][C::NG];
#pragma unroll
    for (int t = 0; t < QPW; ++t) {
        const int qt = wave + t * (ATT_THREADS / 64);
        const T* qrow = base + (size_t)((qt < NT ? qt : 0) * 16 + fr) * row_stride;
#pragma unroll
        for (int g = 0; g < C::NG; ++g) qf_all[t][g] = *reinterpret_cast<const u32x4*>(qrow + (g * 4 + fg) * C::CH);
    }

    // ---- stage K (row-major, swizzled) and V (transposed); zero the padded key rows
    for (int i = tid; i < C::SP * C::RC; i += ATT_THREADS) {
        const int r = i / C::RC, c = i - r * C::RC;
        u32x4 kv = u32x4{0, 0, 0, 0}, vv = u32x4{0, 0, 0, 0};
        if (r < C::S && ATT_ABL != 3) {
            kv = *reinterpret_cast<const u32x4*>(base + (size_t)r * row_stride + E + c * C::CH);
            vv = *reinterpret_cast<const u32x4*>(base + (size_t)r * row_stride + 2 * E + c * C::CH);
        }
        *reinterpret_cast<u32x4*>(Ks + ((size_t)r * C::RC + kswz<C::RC>(r, c)) * 16) = kv;
        if (ATT_ABL == 1) {
            *reinterpret_cast<u32x4*>(Vt + (size_t)i * C::CH) = vv;
            continue;
        }
        T ve[C::CH];
        *reinterpret_cast<u32x4*>(ve) = vv;
#pragma unroll
        for (int j = 0; j < C::CH; ++j) Vt[(c * C::CH + j) * C::SPV + r] = ve[j];
    }
    __syncthreads();

#pragma unroll
    for (int t = 0; t < QPW; ++t) {
        const int qt = wave + t * (ATT_THREADS / 64);
        if (qt >= NT) break;
        if (ATT_ABL == 2) {
            T* orow0 = out + ((size_t)seq * C::S + qt * 16 + fr) * E + head * HD;
            for (int dt = 0; dt < C::DT; ++dt) *reinterpret_cast<u32x2*>(orow0 + dt * 16 + 4 * fg) = u32x2{qf_all[t][0][0], (unsigned)Ks[lane * 4]};
            continue;
        }
        u32x4 qf[C::NG];
#pragma unroll
        for (int g = 0; g < C::NG; ++g) qf[g] = qf_all[t][g];

        // ---- scores: s[kt][i] = q . k for key 16 kt + 4 fg + i
        f32x4 s[C::NTP];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < C::NG; ++g) {
                const int r = kt * 16 + fr;
                const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + ((size_t)r * C::RC + kswz<C::RC>(r, g * 4 + fg)) * 16);
                acc = att_mma(kf, qf[g], acc, T{});
            }
            s[kt] = acc;
        }
        // ---- softmax over the 16 NT keys of this lane's query (fp32)
        float mx = -__builtin_inff();
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) mx = fmaxf(mx, s[kt][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mb = mx * scale_log2e;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][i], scale_log2e, -mb));  // arg <= 0: raw v_exp_f32
                s[kt][i] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        if (C::NTP > NT) s[C::NTP - 1] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- O^T = V^T P^T
        f32x4 o[C::DT];
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int blk = 0; blk < C::NTP / 2; ++blk) {
                const f32x4 p0 = s[2 * blk], p1 = s[2 * blk + 1];
                const bf16x8 pf = {(__bf16)p0[0], (__bf16)p0[1], (__bf16)p0[2], (__bf16)p0[3],
                                   (__bf16)p1[0], (__bf16)p1[1], (__bf16)p1[2], (__bf16)p1[3]};
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt) {
                    const T* vrow = Vt + (dt * 16 + fr) * C::SPV + blk * 32 + 4 * fg;
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(vrow);
                    const u32x2 hi = *reinterpret_cast<const u32x2*>(vrow + 16);
                    const u32x4 vf = {lo[0], lo[1], hi[0], hi[1]};
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf), pf, o[dt], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const u32x4 pf = __builtin_bit_cast(u32x4, s[kt]);
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt) {
                    const u32x4 vf = *reinterpret_cast<const u32x4*>(Vt + (dt * 16 + fr) * C::SPV + kt * 16 + 4 * fg);
                    o[dt] = att_mma(vf, pf, o[dt], T{});
                }
            }
        }
        // ---- normalise and store: lane holds d = 16 dt + 4 fg + (0..3) of query 16 qt + fr
        const float inv = 1.0f / sum;
        T* orow = out + ((size_t)seq * C::S + qt * 16 + fr) * E + head * HD;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            const f32x4 v = o[dt] * inv;
            if constexpr (sizeof(T) == 2) {
                const bf16x4 ov = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                *reinterpret_cast<bf16x4*>(orow + dt * 16 + 4 * fg) = ov;
            } else {
                *reinterpret_cast<f32x4*>(orow + dt * 16 + 4 * fg) = v;
            }
        }
    }
}

// Long sequences (432 tokens @384x288): the single pass above keeps all NTP score fragments of a query tile live - 112
// registers at 432 tokens, it spilled (233 us per launch at ViT-B). This form walks the keys in blocks of KB tiles with the
// online softmax (running maximum m, running sum l, O rescaled by 2^((m_old - m_new) scale) from the second block on), so
// only KB fragments are live; Q is loaded per tile inside a rolled loop; THREADS = 512 because K / V fill most of a CU's
// LDS and only one workgroup is resident: eight waves keep two per SIMD. 80 us per launch at ViT-B. (Also correct for
// the short shapes, where it measured 4 % slower than the single pass: 24.0 vs 23.0 us.)
template <typename T, int HD, int NT, int THREADS, int KB>
__global__ __launch_bounds__(THREADS, 1) void attention_stream_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                                int n_seq, int heads, float scale_log2e) {
    using C = AttCfg<T, HD, NT>;
    static_assert(KB % 2 == 0 && C::NTP % KB == 0, "softmax blocks are whole pairs of key tiles");
    constexpr int NBLK = C::NTP / KB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                                        // [SP][HD], chunk-swizzled
    T* Vt = reinterpret_cast<T*>(smem + C::K_BYTES);        // [HD][SPV]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;

    // XCD-aware remap: hardware round-robins consecutive block ids over the 8 XCDs
    int id = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) id = (id & 7) * (nblk >> 3) + (id >> 3);
    const int seq = id / heads, head = id - seq * heads;
    const int E = heads * HD;
    const size_t row_stride = (size_t)3 * E;
    const T* base = qkv + (size_t)seq * C::S * row_stride + (size_t)head * HD;

    // ---- short sequences: Q fragments of all of this wave's query tiles first, their HBM latency hides under the K/V
    // staging. Long ones (more tiles per wave than registers to park them in): loaded per tile inside a rolled loop.
    constexpr int QPW = (NT + THREADS / 64 - 1) / (THREADS / 64);
    constexpr bool QPRE = NT <= 12;
    u32x4 qf_all[QPRE ? QPW : 1][C::NG];
    if (QPRE) {
#pragma unroll
        for (int t = 0; t < QPW; ++t) {
            const int qt = wave + t * (THREADS / 64);
            const T* qrow = base + (size_t)((qt < NT ? qt : 0) * 16 + fr) * row_stride;
#pragma unroll
            for (int g = 0; g < C::NG; ++g) qf_all[t][g] = *reinterpret_cast<const u32x4*>(qrow + (g * 4 + fg) * C::CH);
        }
    }

    // ---- stage K (row-major, swizzled) and V (transposed); zero the padded key rows
    for (int i = tid; i < C::SP * C::RC; i += THREADS) {
        const int r = i / C::RC, c = i - r * C::RC;
        u32x4 kv = u32x4{0, 0, 0, 0}, vv = u32x4{0, 0, 0, 0};
        if (r < C::S && ATT_ABL != 3) {
            kv = *reinterpret_cast<const u32x4*>(base + (size_t)r * row_stride + E + c * C::CH);
            vv = *reinterpret_cast<const u32x4*>(base + (size_t)r * row_stride + 2 * E + c * C::CH);
        }
        *reinterpret_cast<u32x4*>(Ks + ((size_t)r * C::RC + kswz<C::RC>(r, c)) * 16) = kv;
        if (ATT_ABL == 1) {
            *reinterpret_cast<u32x4*>(Vt + (size_t)i * C::CH) = vv;
            continue;
        }
        T ve[C::CH];
        *reinterpret_cast<u32x4*>(ve) = vv;
#pragma unroll
        for (int j = 0; j < C::CH; ++j) Vt[(c * C::CH + j) * C::SPV + r] = ve[j];
    }
    __syncthreads();

    auto one_tile = [&](int qt, const u32x4 (&qf)[C::NG]) {
        if (ATT_ABL == 2) {
            T* orow0 = out + ((size_t)seq * C::S + qt * 16 + fr) * E + head * HD;
            for (int dt = 0; dt < C::DT; ++dt) *reinterpret_cast<u32x2*>(orow0 + dt * 16 + 4 * fg) = u32x2{qf[0][0], (unsigned)Ks[lane * 4]};
            return;
        }

        f32x4 o[C::DT];
        float m_run = 0.f, l_run = 0.f;  // running maximum (raw scores) and sum of this lane's query
#pragma unroll
        for (int blk0 = 0; blk0 < C::NTP; blk0 += KB) {
            // ---- scores of this block: s[kt][i] = q . k for key 16 (blk0 + kt) + 4 fg + i
            f32x4 s[KB];
#pragma unroll
            for (int kt = 0; kt < KB; ++kt) {
                if (blk0 + kt >= NT) continue;  // padded tile (compile-time)
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < C::NG; ++g) {
                    const int r = (blk0 + kt) * 16 + fr;
                    const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + ((size_t)r * C::RC + kswz<C::RC>(r, g * 4 + fg)) * 16);
                    acc = att_mma(kf, qf[g], acc, T{});
                }
                s[kt] = acc;
            }
            // ---- softmax (fp32): block maximum over this lane's query, the running maximum, the rescale factor
            float mx = -__builtin_inff();
#pragma unroll
            for (int kt = 0; kt < KB; ++kt) {
                if (blk0 + kt >= NT) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) mx = fmaxf(mx, s[kt][i]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (blk0 > 0) {  // (compile-time: the block loop is unrolled)
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt) o[dt] *= alpha;
                mx = m_new;
            }
            m_run = mx;
            const float mb = mx * scale_log2e;
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < KB; ++kt) {
                if (blk0 + kt >= NT) {
                    s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};  // padded keys weigh nothing
                    continue;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][i], scale_log2e, -mb));  // arg <= 0: raw v_exp_f32
                    s[kt][i] = p;
                    sum += p;
                }
            }
            l_run += sum;  // (this lane's keys only; the lanes of a query are added up once, at the end)

            // ---- O^T += V^T P^T over the block's keys
            if (blk0 == 0) {
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int b2 = 0; b2 < KB / 2; ++b2) {
                    const int blk = blk0 / 2 + b2;
                    const f32x4 p0 = s[2 * b2], p1 = s[2 * b2 + 1];
                    const bf16x8 pf = {(__bf16)p0[0], (__bf16)p0[1], (__bf16)p0[2], (__bf16)p0[3],
                                       (__bf16)p1[0], (__bf16)p1[1], (__bf16)p1[2], (__bf16)p1[3]};
#pragma unroll
                    for (int dt = 0; dt < C::DT; ++dt) {
                        const T* vrow = Vt + (dt * 16 + fr) * C::SPV + blk * 32 + 4 * fg;
                        const u32x2 lo = *reinterpret_cast<const u32x2*>(vrow);
                        const u32x2 hi = *reinterpret_cast<const u32x2*>(vrow + 16);
                        const u32x4 vf = {lo[0], lo[1], hi[0], hi[1]};
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf), pf, o[dt], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int kt = 0; kt < KB; ++kt) {
                    if (blk0 + kt >= NT) continue;
                    const u32x4 pf = __builtin_bit_cast(u32x4, s[kt]);
#pragma unroll
                    for (int dt = 0; dt < C::DT; ++dt) {
                        const u32x4 vf = *reinterpret_cast<const u32x4*>(Vt + (dt * 16 + fr) * C::SPV + (blk0 + kt) * 16 + 4 * fg);
                        o[dt] = att_mma(vf, pf, o[dt], T{});
                    }
                }
            }
            if (NBLK > 1) __builtin_amdgcn_sched_barrier(0);  // keep the next block's scores from being hoisted up here (registers)
        }
        float sum = l_run;
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        // ---- normalise and store: lane holds d = 16 dt + 4 fg + (0..3) of query 16 qt + fr
        const float inv = 1.0f / sum;
        T* orow = out + ((size_t)seq * C::S + qt * 16 + fr) * E + head * HD;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            const f32x4 v = o[dt] * inv;
            if constexpr (sizeof(T) == 2) {
                const bf16x4 ov = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                *reinterpret_cast<bf16x4*>(orow + dt * 16 + 4 * fg) = ov;
            } else {
                *reinterpret_cast<f32x4*>(orow + dt * 16 + 4 * fg) = v;
            }
        }
    };
    if (QPRE) {
#pragma unroll
        for (int t = 0; t < QPW; ++t) {
            const int qt = wave + t * (THREADS / 64);
            if (qt >= NT) break;
            one_tile(qt, qf_all[t]);
        }
    } else {
#pragma unroll 1
        for (int qt = wave; qt < NT; qt += THREADS / 64) {
            const T* qrow = base + (size_t)(qt * 16 + fr) * row_stride;
#pragma unroll
            for (int g = 0; g < C::NG; ++g) qf_all[0][g] = *reinterpret_cast<const u32x4*>(qrow + (g * 4 + fg) * C::CH);
            one_tile(qt, qf_all[0]);
        }
    }
}

template <typename T, int HD, int NT>
static int launch_attention(const void* qkv, void* out, int n_seq, int heads, float scale, hipStream_t s) {
    using C = AttCfg<T, HD, NT>;
    static_assert(C::LDS <= 160 * 1024, "K/V of one head must fit in one CU's LDS");
    if constexpr (NT > 12) {
        constexpr int THREADS = 2 * ATT_THREADS;
        constexpr int KB = sizeof(T) == 2 ? C::NTP / 2 : 4;  // 432 tokens: two softmax blocks of 14 key tiles (fp32: seven of 4)
        auto kern = attention_stream_kernel<T, HD, NT, THREADS, KB>;
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)C::LDS));
        hipLaunchKernelGGL(kern, dim3(n_seq * heads), dim3(THREADS), C::LDS, s, reinterpret_cast<const T*>(qkv),
                           reinterpret_cast<T*>(out), n_seq, heads, scale * 1.44269504088896340736f);
    } else {
        auto kern = attention_kernel<T, HD, NT>;
        PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)C::LDS));
        hipLaunchKernelGGL(kern, dim3(n_seq * heads), dim3(ATT_THREADS), C::LDS, s, reinterpret_cast<const T*>(qkv),
                           reinterpret_cast<T*>(out), n_seq, heads, scale * 1.44269504088896340736f);
    }
    PP_LAUNCH_CHECK();
    return PP_OK;
}

// ---------------------------------------------------------------------------------------------
// PP_PREC_F16X3: the same single-pass attention on split-fp16 operands (pp_split.h). qkv rows are 3 E elements of
// 4 bytes in 128-byte blocks (32 hi halves | 32 lo halves), so a head of 32 dims is exactly one block. K is staged as
// the raw blocks (chunk-swizzled), V transposed into a hi and a lo plane; every contraction is three fp16 MFMAs
// (lo*hi + hi*lo + hi*hi); the softmax probabilities are split in registers (p in [0, 1]: a subnormal low half is
// an absolute error below 2^-25). Output rows are written in the split format.
template <int HD, int NT>
struct AttSplitCfg {
    static constexpr int S = NT * 16;
    static constexpr int NTP = (NT + 1) & ~1;
    static constexpr int SP = NTP * 16;
    static constexpr int SPV = SP + 8;          // V^T row pitch in halves
    static constexpr int NB = HD / 32;          // 128-byte blocks per head row
    static constexpr int RC = NB * 8;           // 16-byte chunks per head row
    static constexpr int DT = HD / 16;
    static constexpr size_t K_BYTES = (size_t)SP * HD * 4;
    static constexpr size_t V_BYTES = (size_t)2 * HD * SPV * 2;
    static constexpr size_t LDS = K_BYTES + V_BYTES;
};

template <int HD, int NT>
__global__ __launch_bounds__(ATT_THREADS, (HD == 32 ? 3 : 1)) void attention_split_kernel(const char* __restrict__ qkv, char* __restrict__ out,
                                                                      int n_seq, int heads, float scale_log2e) {
    using C = AttSplitCfg<HD, NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                                                         // [SP][RC chunks], chunk-swizzled
    _Float16* Vh = reinterpret_cast<_Float16*>(smem + C::K_BYTES);           // [HD][SPV] hi halves of V^T
    _Float16* Vl = Vh + HD * C::SPV;                                         // [HD][SPV] lo halves
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;

    int id = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) id = (id & 7) * (nblk >> 3) + (id >> 3);  // XCD-aware remap (as above)
    const int seq = id / heads, head = id - seq * heads;
    const int E = heads * HD;
    const size_t row_bytes = (size_t)3 * E * 4;
    const char* base = qkv + (size_t)seq * C::S * row_bytes + (size_t)head * HD * 4;

    constexpr int QPW = (NT + ATT_THREADS / 64 - 1) / (ATT_THREADS / 64);
    f16x8 qh_all[QPW][C::NB], ql_all[QPW][C::NB];
#pragma unroll
    for (int t = 0; t < QPW; ++t) {
        const int qt = wave + t * (ATT_THREADS / 64);
        const char* qrow = base + (size_t)((qt < NT ? qt : 0) * 16 + fr) * row_bytes;
#pragma unroll
        for (int g = 0; g < C::NB; ++g) {
            qh_all[t][g] = *reinterpret_cast<const f16x8*>(qrow + g * 128 + fg * 16);
            ql_all[t][g] = *reinterpret_cast<const f16x8*>(qrow + g * 128 + 64 + fg * 16);
        }
    }

    // ---- stage K (raw blocks, swizzled) and V^T (hi / lo planes); zero the padded key rows
    for (int i = tid; i < C::SP * C::RC; i += ATT_THREADS) {
        const int r = i / C::RC, c = i - r * C::RC;
        u32x4 kv = u32x4{0, 0, 0, 0}, vv = u32x4{0, 0, 0, 0};
        if (r < C::S) {
            kv = *reinterpret_cast<const u32x4*>(base + (size_t)r * row_bytes + (size_t)E * 4 + c * 16);
            vv = *reinterpret_cast<const u32x4*>(base + (size_t)r * row_bytes + (size_t)2 * E * 4 + c * 16);
        }
        *reinterpret_cast<u32x4*>(Ks + ((size_t)r * C::RC + (c ^ (r & 7))) * 16) = kv;
        const f16x8 ve = __builtin_bit_cast(f16x8, vv);
        const int d0 = (c >> 3) * 32 + (c & 3) * 8;
        _Float16* plane = (c & 4) ? Vl : Vh;
#pragma unroll
        for (int j = 0; j < 8; ++j) plane[(d0 + j) * C::SPV + r] = ve[j];
    }
    __syncthreads();

#pragma unroll
    for (int t = 0; t < QPW; ++t) {
        const int qt = wave + t * (ATT_THREADS / 64);
        if (qt >= NT) break;
        // ---- scores: s[kt][i] = q . k for key 16 kt + 4 fg + i
        f32x4 s[C::NTP];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int r = kt * 16 + fr;
#pragma unroll
            for (int g = 0; g < C::NB; ++g) {
                const f16x8 kh = *reinterpret_cast<const f16x8*>(Ks + ((size_t)r * C::RC + ((g * 8 + fg) ^ (r & 7))) * 16);
                const f16x8 kl = *reinterpret_cast<const f16x8*>(Ks + ((size_t)r * C::RC + ((g * 8 + 4 + fg) ^ (r & 7))) * 16);
                acc = split_mma(kh, kl, qh_all[t][g], ql_all[t][g], acc);
            }
            s[kt] = acc;
        }
        // ---- softmax over the 16 NT keys of this lane's query (fp32)
        float mx = -__builtin_inff();
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) mx = fmaxf(mx, s[kt][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mb = mx * scale_log2e;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][i], scale_log2e, -mb));  // arg <= 0: raw v_exp_f32 (1 ulp)
                s[kt][i] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        if (C::NTP > NT) s[C::NTP - 1] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- O^T = V^T P^T
        f32x4 o[C::DT];
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < C::NTP / 2; ++blk) {
            const f32x4 p0 = s[2 * blk], p1 = s[2 * blk + 1];
            f16x8 ph, pl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ph[j] = split_hi(p0[j]);
                pl[j] = split_lo(p0[j], ph[j]);
                ph[4 + j] = split_hi(p1[j]);
                pl[4 + j] = split_lo(p1[j], ph[4 + j]);
            }
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) {
                const int off = (dt * 16 + fr) * C::SPV + blk * 32 + 4 * fg;
                const u32x2 h0 = *reinterpret_cast<const u32x2*>(Vh + off), h1 = *reinterpret_cast<const u32x2*>(Vh + off + 16);
                const u32x2 l0 = *reinterpret_cast<const u32x2*>(Vl + off), l1 = *reinterpret_cast<const u32x2*>(Vl + off + 16);
                const u32x4 vh = {h0[0], h0[1], h1[0], h1[1]}, vl = {l0[0], l0[1], l1[0], l1[1]};
                o[dt] = split_mma(__builtin_bit_cast(f16x8, vh), __builtin_bit_cast(f16x8, vl), ph, pl, o[dt]);
            }
        }
        // ---- normalise and store: lane holds d = 16 dt + 4 fg + (0..3) of query 16 qt + fr
        const float inv = 1.0f / sum;
        const size_t oidx = ((size_t)seq * C::S + qt * 16 + fr) * E + head * HD;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) split_store4(out, oidx + dt * 16 + 4 * fg, o[dt] * inv);
    }
}

template <int HD, int NT>
static int launch_attention_split(const void* qkv, void* out, int n_seq, int heads, float scale, hipStream_t s) {
    using C = AttSplitCfg<HD, NT>;
    static_assert(C::LDS <= 160 * 1024, "K/V of one head must fit in one CU's LDS");
    auto kern = attention_split_kernel<HD, NT>;
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
    hipLaunchKernelGGL(kern, dim3(n_seq * heads), dim3(ATT_THREADS), C::LDS, s, reinterpret_cast<const char*>(qkv),
                       reinterpret_cast<char*>(out), n_seq, heads, scale * 1.44269504088896340736f);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

// Long sequences in the split format (432 tokens @384x288; head dim 64 does not fit K and V^T of all keys in one CU's
// LDS - 229 KiB): the keys are walked in NSTAGE stages of KS key tiles. Per stage the workgroup stages that slice of K
// and V^T, then every wave runs its query tiles over it with the online softmax (running maximum m, running sum l, O
// rescaled when the maximum moves); the per-tile state lives in registers across the stages, so a wave owns ONE query
// tile (several per wave spilled: 283 scratch registers at head dim 64): the query tiles of a (sequence, head) are cut
// into QSPLIT workgroups of 7 waves, each of which stages the K / V slices for itself (L2-resident re-reads).
template <int HD, int NT, int NSTAGE>
struct AttSplitStreamCfg {
    static constexpr int NTP = ((NT + 2 * NSTAGE - 1) / (2 * NSTAGE)) * (2 * NSTAGE);  // key tiles padded to whole stages of pairs
    static constexpr int KS = NTP / NSTAGE;     // key tiles per stage (even)
    static constexpr int SS = KS * 16;          // keys per stage
    static constexpr int SPV = SS + 8;
    static constexpr int NB = HD / 32, RC = NB * 8, DT = HD / 16;
    static constexpr size_t K_BYTES = (size_t)SS * HD * 4;
    static constexpr size_t V_BYTES = (size_t)2 * HD * SPV * 2;
    static constexpr size_t LDS = K_BYTES + V_BYTES;
};

template <int HD, int NT, int NSTAGE, int THREADS, int QSPLIT>
__global__ __launch_bounds__(THREADS, 1) void attention_split_stream_kernel(const char* __restrict__ qkv, char* __restrict__ out,
                                                                          int n_seq, int heads, float scale_log2e) {
    using C = AttSplitStreamCfg<HD, NT, NSTAGE>;
    constexpr int S = NT * 16, NW = THREADS / 64, TPW = (NT + QSPLIT - 1) / QSPLIT, QPW = (TPW + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    _Float16* Vh = reinterpret_cast<_Float16*>(smem + C::K_BYTES);
    _Float16* Vl = Vh + HD * C::SPV;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    int id = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) id = (id & 7) * (nblk >> 3) + (id >> 3);
    const int qs = id % QSPLIT;  // the QSPLIT workgroups of a (sequence, head) are neighbours: same XCD, same L2 lines
    id /= QSPLIT;
    const int seq = id / heads, head = id - seq * heads;
    const int E = heads * HD;
    const size_t row_bytes = (size_t)3 * E * 4;
    const char* base = qkv + (size_t)seq * S * row_bytes + (size_t)head * HD * 4;

    f32x4 o[QPW][C::DT];
    float m_run[QPW], l_run[QPW];
#pragma unroll
    for (int t = 0; t < QPW; ++t) {
        m_run[t] = -__builtin_inff();
        l_run[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

#pragma unroll
    for (int st = 0; st < NSTAGE; ++st) {
        if (st > 0) __syncthreads();  // every wave is done with the previous slice
        for (int i = tid; i < C::SS * C::RC; i += THREADS) {
            const int r = i / C::RC, c = i - r * C::RC;
            const int key = st * C::SS + r;
            u32x4 kv = u32x4{0, 0, 0, 0}, vv = u32x4{0, 0, 0, 0};
            if (key < S) {
                kv = *reinterpret_cast<const u32x4*>(base + (size_t)key * row_bytes + (size_t)E * 4 + c * 16);
                vv = *reinterpret_cast<const u32x4*>(base + (size_t)key * row_bytes + (size_t)2 * E * 4 + c * 16);
            }
            *reinterpret_cast<u32x4*>(Ks + ((size_t)r * C::RC + (c ^ (r & 7))) * 16) = kv;
            const f16x8 ve = __builtin_bit_cast(f16x8, vv);
            const int d0 = (c >> 3) * 32 + (c & 3) * 8;
            _Float16* plane = (c & 4) ? Vl : Vh;
#pragma unroll
            for (int j = 0; j < 8; ++j) plane[(d0 + j) * C::SPV + r] = ve[j];
        }
        __syncthreads();

#pragma unroll
        for (int t = 0; t < QPW; ++t) {
            const int qt = qs * TPW + wave + t * NW;
            if (wave + t * NW < TPW && qt < NT) {
                const char* qrow = base + (size_t)(qt * 16 + fr) * row_bytes;
                f16x8 qh[C::NB], ql[C::NB];
#pragma unroll
                for (int g = 0; g < C::NB; ++g) {
                    qh[g] = *reinterpret_cast<const f16x8*>(qrow + g * 128 + fg * 16);
                    ql[g] = *reinterpret_cast<const f16x8*>(qrow + g * 128 + 64 + fg * 16);
                }
                f32x4 sc[C::KS];
                float mx = m_run[t];
#pragma unroll
                for (int kt = 0; kt < C::KS; ++kt) {
                    if (st * C::KS + kt >= NT) continue;  // padded key tile (compile-time)
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    const int r = kt * 16 + fr;
#pragma unroll
                    for (int g = 0; g < C::NB; ++g) {
                        const f16x8 kh = *reinterpret_cast<const f16x8*>(Ks + ((size_t)r * C::RC + ((g * 8 + fg) ^ (r & 7))) * 16);
                        const f16x8 kl = *reinterpret_cast<const f16x8*>(Ks + ((size_t)r * C::RC + ((g * 8 + 4 + fg) ^ (r & 7))) * 16);
                        acc = split_mma(kh, kl, qh[g], ql[g], acc);
                    }
                    sc[kt] = acc;
#pragma unroll
                    for (int i = 0; i < 4; ++i) mx = fmaxf(mx, acc[i]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                if (st > 0) {  // rescale what the earlier stages accumulated (exp2(-inf) = 0 never occurs: st 0 set m_run)
                    const float alpha = __builtin_amdgcn_exp2f((m_run[t] - mx) * scale_log2e);
                    l_run[t] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < C::DT; ++dt) o[t][dt] *= alpha;
                }
                m_run[t] = mx;
                const float mb = mx * scale_log2e;
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < C::KS; ++kt) {
                    if (st * C::KS + kt >= NT) {
                        sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};  // padded keys weigh nothing
                        continue;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][i], scale_log2e, -mb));
                        sc[kt][i] = p;
                        sum += p;
                    }
                }
                l_run[t] += sum;  // this lane's keys only; the four lanes of a query are added up at the end
#pragma unroll
                for (int blk = 0; blk < C::KS / 2; ++blk) {
                    if (st * C::KS + 2 * blk >= NT) continue;
                    const f32x4 p0 = sc[2 * blk], p1 = sc[2 * blk + 1];
                    f16x8 ph, pl;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        ph[j] = split_hi(p0[j]);
                        pl[j] = split_lo(p0[j], ph[j]);
                        ph[4 + j] = split_hi(p1[j]);
                        pl[4 + j] = split_lo(p1[j], ph[4 + j]);
                    }
#pragma unroll
                    for (int dt = 0; dt < C::DT; ++dt) {
                        const int off = (dt * 16 + fr) * C::SPV + blk * 32 + 4 * fg;
                        const u32x2 h0 = *reinterpret_cast<const u32x2*>(Vh + off), h1 = *reinterpret_cast<const u32x2*>(Vh + off + 16);
                        const u32x2 l0 = *reinterpret_cast<const u32x2*>(Vl + off), l1 = *reinterpret_cast<const u32x2*>(Vl + off + 16);
                        const u32x4 vh = {h0[0], h0[1], h1[0], h1[1]}, vl = {l0[0], l0[1], l1[0], l1[1]};
                        o[t][dt] = split_mma(__builtin_bit_cast(f16x8, vh), __builtin_bit_cast(f16x8, vl), ph, pl, o[t][dt]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // one tile's score fragments at a time (registers)
            }
        }
    }
#pragma unroll
    for (int t = 0; t < QPW; ++t) {
        const int qt = qs * TPW + wave + t * NW;
        if (wave + t * NW < TPW && qt < NT) {
            float sum = l_run[t];
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            const size_t oidx = ((size_t)seq * S + qt * 16 + fr) * E + head * HD;
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) split_store4(out, oidx + dt * 16 + 4 * fg, o[t][dt] * inv);
        }
    }
}

template <int HD, int NT, int NSTAGE>
static int launch_attention_split_stream(const void* qkv, void* out, int n_seq, int heads, float scale, hipStream_t s) {
    using C = AttSplitStreamCfg<HD, NT, NSTAGE>;
    static_assert(C::LDS <= 160 * 1024, "one stage of K/V must fit in one CU's LDS");
    constexpr int QSPLIT = 4, THREADS = 64 * ((NT + QSPLIT - 1) / QSPLIT);  // 27 query tiles: 4 workgroups of 7 waves
    auto kern = attention_split_stream_kernel<HD, NT, NSTAGE, THREADS, QSPLIT>;
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
    hipLaunchKernelGGL(kern, dim3(n_seq * heads * QSPLIT), dim3(THREADS), C::LDS, s, reinterpret_cast<const char*>(qkv),
                       reinterpret_cast<char*>(out), n_seq, heads, scale * 1.44269504088896340736f);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

int attention_split_dma(const void* qkv, void* out, int n_seq, int heads, int head_dim, float scale, hipStream_t s);  // pp_attention_dma.hip

}  // namespace pp

extern "C" int pp_attention(int prec, const void* qkv, void* out, int n_seq, int seq_len, int heads, int head_dim,
                            float scale, void* stream) {
    using namespace pp;
    PP_REQUIRE(qkv && out, PP_ERR_INVALID_ARG, "pp_attention: qkv and out must be non-NULL");
    PP_REQUIRE(n_seq > 0 && heads > 0, PP_ERR_INVALID_ARG, "pp_attention: n_seq and heads must be positive");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define PP_ATT_CASE(T, HD, NT) \
    if (head_dim == HD && seq_len == NT * 16) return launch_attention<T, HD, NT>(qkv, out, n_seq, heads, scale, s);
    if (prec == PP_PREC_BF16) {
        PP_ATT_CASE(__bf16, 32, 12)  // ProbPose-S 256x192
        PP_ATT_CASE(__bf16, 64, 12)  // ViT-B 256x192
        PP_ATT_CASE(__bf16, 32, 27)  // ProbPose-S 384x288
        PP_ATT_CASE(__bf16, 64, 27)  // ViT-B 384x288
    } else if (prec == PP_PREC_F32) {
        PP_ATT_CASE(float, 32, 12)
        PP_ATT_CASE(float, 64, 12)
        PP_ATT_CASE(float, 32, 27)
    } else if (prec == PP_PREC_F16X3) {
        if (head_dim == 32 && seq_len == 192) return launch_attention_split<32, 12>(qkv, out, n_seq, heads, scale, s);
        if (head_dim == 64 && seq_len == 192) return launch_attention_split<64, 12>(qkv, out, n_seq, heads, scale, s);
        if (seq_len == 432 && (head_dim == 32 || head_dim == 64) && option("attn_dma") != 0)  // K / V by LDS-DMA, V^T by transposing reads
            return attention_split_dma(qkv, out, n_seq, heads, head_dim, scale, s);
        if (head_dim == 32 && seq_len == 432) return launch_attention_split_stream<32, 27, 2>(qkv, out, n_seq, heads, scale, s);
        if (head_dim == 64 && seq_len == 432) return launch_attention_split_stream<64, 27, 2>(qkv, out, n_seq, heads, scale, s);
    } else {
        return fail(PP_ERR_INVALID_ARG, "pp_attention: unknown precision");
    }
#undef PP_ATT_CASE
    return fail(PP_ERR_UNSUPPORTED,
                "pp_attention: (seq_len, head_dim) not instantiated: supported 192/432 tokens x 32/64 "
                "(fp32 at 432 x 64 exceeds one CU's LDS; PP_PREC_F16X3 covers it in two key stages)");
}
