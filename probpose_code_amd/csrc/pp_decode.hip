// ProbMap decode for gfx950: flip-test average -> separable OKS-kernel convolution (f64
// accumulate, one rounding to f32) -> first-occurrence argmax -> one Newton sub-pixel step
// -> rescale to input-pixel space. One 256-thread workgroup per (crop, keypoint); the whole
// map lives in LDS, nothing but the inputs is read from HBM and nothing but the results
// (plus the optional maps) is written.
//
// Arithmetic follows mmpose/codecs/utils/post_processing.py:308-430 and
// mmpose/codecs/probmap.py:218 of the reference; see include/probpose_mi355x.h.
#include "pp_common.h"

// numpy evaluates the f32 sub-pixel expressions one rounding per operator; keep it so.
#pragma clang fp contract(off)

namespace pp {

constexpr int RM = PP_MAX_RADIUS;  // pad every map by the largest radius
constexpr int DEC_THREADS = 256;
constexpr int GX = 6;  // outputs per work item in the row pass (sliding register window)
constexpr int GY = 4;  // outputs per work item in the column pass
constexpr int RED_BYTES = 512;   // cross-wave reduction scratch at the head of the dynamic LDS region
constexpr int MAX_EPT = 28;      // Sparsemax keeps a whole row in registers: H*W <= 28 * 256

struct ArgBest {
    float v;
    int idx;
};

// np.argmax semantics: NaN counts as the maximum, first occurrence wins ties.
__device__ __forceinline__ bool better(float v, int idx, float bv, int bidx) {
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) return vn && (!bn || idx < bidx);
    return v > bv || (v == bv && idx < bidx);
}

// Block-wide reductions for the in-register Sparsemax (4 waves). `slot` alternates between two
// scratch areas so that one barrier per reduction suffices.
struct SmxStat {
    double s0, s1;
    int n0, n1;
};

__device__ __forceinline__ void block_max2(float& a, float& b, float* scratch) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        a = fmaxf(a, __shfl_xor(a, off));
        b = fmaxf(b, __shfl_xor(b, off));
    }
    if (lane_id() == 0) {
        scratch[2 * wave_id()] = a;
        scratch[2 * wave_id() + 1] = b;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < DEC_THREADS / WAVE; ++w) {
        a = fmaxf(a, scratch[2 * w]);
        b = fmaxf(b, scratch[2 * w + 1]);
    }
}

__device__ __forceinline__ SmxStat block_sum_stat(SmxStat v, SmxStat* scratch) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        v.s0 += __shfl_xor(v.s0, off);
        v.s1 += __shfl_xor(v.s1, off);
        v.n0 += __shfl_xor(v.n0, off);
        v.n1 += __shfl_xor(v.n1, off);
    }
    if (lane_id() == 0) scratch[wave_id()] = v;
    __syncthreads();
    SmxStat r = scratch[0];
#pragma unroll
    for (int w = 1; w < DEC_THREADS / WAVE; ++w) {  // fixed order: every thread gets the same bits
        r.s0 += scratch[w].s0;
        r.s1 += scratch[w].s1;
        r.n0 += scratch[w].n0;
        r.n1 += scratch[w].n1;
    }
    return r;
}

// FROM_LOGITS: `hm` / `hm_flip` hold the raw outputs of the head's final 1x1 conv; the kernel then also
// does  x / temperature -> Sparsemax over the H*W pixels -> * normalize -> clamp(0, 1)
// (probmap_head.py:637-646) before the flip-test average, so logits are read from HBM exactly once.
template <bool HAS_FLIP, bool FROM_LOGITS>
__global__ __launch_bounds__(DEC_THREADS) void probmap_decode_kernel(
    const float* __restrict__ hm, const float* __restrict__ hm_flip, const int32_t* __restrict__ flip_indices,
    const double* __restrict__ taps, const int32_t* __restrict__ radius, int K, int H, int W, double in_w,
    double in_h, float temperature, float normalize, float* __restrict__ avg_out, float* __restrict__ conv_out,
    float* __restrict__ locs, double* __restrict__ keypoints, float* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int bk = blockIdx.x;
    const int b = bk / K, k = bk - b * K;
    const int Wp = W + 2 * RM;
    const int HW = H * W;

    // all LDS in the one dynamic region (16-B aligned carve offsets)
    ArgBest* red = reinterpret_cast<ArgBest*>(smem);                                      // [4] cross-wave argmax
    float* mapf = reinterpret_cast<float*>(smem + RED_BYTES);                             // [H][Wp]    averaged map, x-padded
    double* rowd = reinterpret_cast<double*>(smem + RED_BYTES + ((H * Wp * 4 + 15) & ~15));  // [H+2RM][W] row pass, y-padded
    float* convf = reinterpret_cast<float*>(rowd + (H + 2 * RM) * W);                     // [H][W]     convolved map (f32)

    const int r = radius[k];
    // taps centred at RM: tapc[RM + t] multiplies the sample at offset t, |t| <= r
    double tapc[PP_MAX_TAPS];
#pragma unroll
    for (int j = 0; j < PP_MAX_TAPS; ++j) {
        const int t = j - (RM - r);
        tapc[j] = (t >= 0 && t <= 2 * r) ? taps[k * PP_MAX_TAPS + t] : 0.0;
    }

    // ---- load (+ flip-back + average), probmap_head.py:757-763 / tta.py:35-39
    const float* src = hm + (size_t)bk * HW;
    const float* srcf = nullptr;
    if (HAS_FLIP) srcf = hm_flip + ((size_t)b * K + flip_indices[k]) * HW;
    if constexpr (!FROM_LOGITS) {
        for (int i = tid; i < HW; i += DEC_THREADS) {
            const int y = i / W, x = i - y * W;
            float v = src[i];
            if (HAS_FLIP) v = (v + srcf[y * W + (W - 1 - x)]) * 0.5f;
            mapf[y * Wp + RM + x] = v;
            if (avg_out) avg_out[(size_t)bk * HW + i] = v;
        }
    } else {
        // ---- Sparsemax of this keypoint's row and (flip test) of its mirror partner's row, in registers.
        // Sort-free threshold search (Michelot): start from the candidates z > max - 1 (tau >= max - 1
        // always), tau <- (sum_cand z - 1) / |cand|, drop z <= tau, repeat until nothing is dropped.
        float* fscr = reinterpret_cast<float*>(smem);
        SmxStat* sscr = reinterpret_cast<SmxStat*>(smem + 64);
        float z0[MAX_EPT], z1[MAX_EPT];
        float m0 = -__builtin_inff(), m1 = -__builtin_inff();
#pragma unroll
        for (int e = 0; e < MAX_EPT; ++e) {
            const int i = tid + e * DEC_THREADS;
            z0[e] = -__builtin_inff();
            z1[e] = -__builtin_inff();
            if (i < HW) {
                z0[e] = src[i] / temperature;
                if (HAS_FLIP) z1[e] = srcf[i] / temperature;
            }
            m0 = fmaxf(m0, z0[e]);
            m1 = fmaxf(m1, z1[e]);
        }
        block_max2(m0, m1, fscr);
        float tau0 = -1.0f, tau1 = -1.0f;
        int prev0 = -1, prev1 = -1;
#pragma unroll
        for (int e = 0; e < MAX_EPT; ++e) {
            z0[e] -= m0;
            z1[e] -= m1;
        }
        for (int iter = 0; iter < 64; ++iter) {
            SmxStat st{0.0, 0.0, 0, 0};
#pragma unroll
            for (int e = 0; e < MAX_EPT; ++e) {
                if (z0[e] > tau0) {
                    st.s0 += (double)z0[e];
                    st.n0 += 1;
                }
                if (HAS_FLIP && z1[e] > tau1) {
                    st.s1 += (double)z1[e];
                    st.n1 += 1;
                }
            }
            st = block_sum_stat(st, sscr + (iter & 1) * (DEC_THREADS / WAVE));
            const bool done = st.n0 == prev0 && (!HAS_FLIP || st.n1 == prev1);
            if (done) break;
            prev0 = st.n0;
            prev1 = st.n1;
            tau0 = (float)((st.s0 - 1.0) / (double)st.n0);
            if (HAS_FLIP) tau1 = (float)((st.s1 - 1.0) / (double)st.n1);
        }
#pragma unroll
        for (int e = 0; e < MAX_EPT; ++e) {
            const int i = tid + e * DEC_THREADS;
            if (i < HW) {
                const int y = i / W, x = i - y * W;
                const float p = fminf(fmaxf(fmaxf(z0[e] - tau0, 0.0f) * normalize, 0.0f), 1.0f);
                mapf[y * Wp + RM + x] = p;
            }
        }
        if (HAS_FLIP) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < MAX_EPT; ++e) {
                const int i = tid + e * DEC_THREADS;
                if (i < HW) {
                    const int y = i / W, xf = i - y * W;
                    const float p = fminf(fmaxf(fmaxf(z1[e] - tau1, 0.0f) * normalize, 0.0f), 1.0f);
                    float* cell = mapf + y * Wp + RM + (W - 1 - xf);  // exactly one thread owns each cell
                    *cell = (*cell + p) * 0.5f;
                }
            }
        }
        if (avg_out) {
            __syncthreads();
            for (int i = tid; i < HW; i += DEC_THREADS) {
                const int y = i / W, x = i - y * W;
                avg_out[(size_t)bk * HW + i] = mapf[y * Wp + RM + x];
            }
        }
    }
    __syncthreads();
    // half-sample symmetric x-padding (scipy.ndimage 'reflect'): -1 -> 0, -2 -> 1, W -> W-1, ...
    for (int i = tid; i < H * 2 * RM; i += DEC_THREADS) {
        const int y = i / (2 * RM), p = i - y * (2 * RM);
        if (p < RM)
            mapf[y * Wp + (RM - 1 - p)] = mapf[y * Wp + RM + p];
        else
            mapf[y * Wp + RM + W + (p - RM)] = mapf[y * Wp + RM + W - 1 - (p - RM)];
    }
    __syncthreads();

    // ---- row pass: rowd[y][x] = sum_t map[y][x+t] * tap[t], t ascending, f64
    {
        const int nxg = (W + GX - 1) / GX;
        for (int it = tid; it < H * nxg; it += DEC_THREADS) {
            const int y = it / nxg, x0 = (it - y * nxg) * GX;
            const float* p = mapf + y * Wp + x0;
            double win[GX + 2 * RM];
#pragma unroll
            for (int c = 0; c < GX + 2 * RM; ++c) win[c] = (x0 + c < Wp) ? (double)p[c] : 0.0;
            double acc[GX];
#pragma unroll
            for (int g = 0; g < GX; ++g) acc[g] = 0.0;
#pragma unroll
            for (int j = 0; j < PP_MAX_TAPS; ++j) {
                if (j >= RM - r && j <= RM + r) {
                    const double t = tapc[j];
#pragma unroll
                    for (int g = 0; g < GX; ++g) acc[g] = fma(win[g + j], t, acc[g]);
                }
            }
#pragma unroll
            for (int g = 0; g < GX; ++g)
                if (x0 + g < W) rowd[(y + RM) * W + x0 + g] = acc[g];
        }
    }
    __syncthreads();
    // symmetric y-padding of the row-pass result
    for (int i = tid; i < 2 * RM * W; i += DEC_THREADS) {
        const int p = i / W, x = i - p * W;
        if (p < RM)
            rowd[(RM - 1 - p) * W + x] = rowd[(RM + p) * W + x];
        else
            rowd[(RM + H + (p - RM)) * W + x] = rowd[(RM + H - 1 - (p - RM)) * W + x];
    }
    __syncthreads();

    // ---- column pass + running argmax
    ArgBest best{-__builtin_inff(), 0x7fffffff};
    {
        const int nyg = (H + GY - 1) / GY;
        for (int it = tid; it < W * nyg; it += DEC_THREADS) {
            const int yg = it / W, x = it - yg * W, y0 = yg * GY;
            double win[GY + 2 * RM];
#pragma unroll
            for (int c = 0; c < GY + 2 * RM; ++c) win[c] = (y0 + c < H + 2 * RM) ? rowd[(y0 + c) * W + x] : 0.0;
            double acc[GY];
#pragma unroll
            for (int g = 0; g < GY; ++g) acc[g] = 0.0;
#pragma unroll
            for (int j = 0; j < PP_MAX_TAPS; ++j) {
                if (j >= RM - r && j <= RM + r) {
                    const double t = tapc[j];
#pragma unroll
                    for (int g = 0; g < GY; ++g) acc[g] = fma(win[g + j], t, acc[g]);
                }
            }
#pragma unroll
            for (int g = 0; g < GY; ++g) {
                if (y0 + g < H) {
                    const float v = (float)acc[g];  // the single rounding scipy does on output
                    const int idx = (y0 + g) * W + x;
                    convf[idx] = v;
                    if (better(v, idx, best.v, best.idx)) best = ArgBest{v, idx};
                }
            }
        }
    }
    // wave-level then block-level argmax
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best.v, off);
        const int oi = __shfl_xor(best.idx, off);
        if (better(ov, oi, best.v, best.idx)) best = ArgBest{ov, oi};
    }
    if (lane_id() == 0) red[wave_id()] = best;
    __syncthreads();
    if (conv_out)
        for (int i = tid; i < HW; i += DEC_THREADS) conv_out[(size_t)bk * HW + i] = convf[i];

    if (tid == 0) {
        for (int w = 1; w < DEC_THREADS / WAVE; ++w)
            if (better(red[w].v, red[w].idx, best.v, best.idx)) best = red[w];
        const int yi = best.idx / W, xi = best.idx - yi * W;
        float lx = (float)xi, ly = (float)yi;
        // post_processing.py:384-430 -- interior peaks only, f32, zero curvature -> 1e-6, no clamp
        if (xi > 0 && xi < W - 1 && yi > 0 && yi < H - 1) {
            const float c = convf[yi * W + xi];
            const float xp = convf[yi * W + xi + 1], xm = convf[yi * W + xi - 1];
            const float yp = convf[(yi + 1) * W + xi], ym = convf[(yi - 1) * W + xi];
            const float dx = (xp - xm) / 2.0f;
            const float dy = (yp - ym) / 2.0f;
            float dxx = xp + xm - 2.0f * c;
            float dyy = yp + ym - 2.0f * c;
            if (!(dxx != 0.0f)) dxx = 1e-6f;
            if (!(dyy != 0.0f)) dyy = 1e-6f;
            lx = lx + (-dx / dxx);
            ly = ly + (-dy / dyy);
        }
        locs[2 * bk + 0] = lx;
        locs[2 * bk + 1] = ly;
        // probmap.py:218 -- f32 locs promoted to f64 by the division
        keypoints[2 * bk + 0] = (double)lx / (double)(W - 1) * in_w;
        keypoints[2 * bk + 1] = (double)ly / (double)(H - 1) * in_h;
        scores[bk] = mapf[yi * Wp + RM + xi];  // raw (un-convolved) averaged map at the integer argmax
    }
}

static size_t decode_lds_bytes(int H, int W) {
    const size_t mapf = ((size_t)H * (W + 2 * RM) * 4 + 15) & ~(size_t)15;
    return RED_BYTES + mapf + (size_t)(H + 2 * RM) * W * 8 + (size_t)H * W * 4;
}

}  // namespace pp

static int decode_launch(bool from_logits, const float* hm, const float* hm_flip, const int32_t* flip_indices,
                         const double* taps, const int32_t* radius, int B, int K, int H, int W, double in_w,
                         double in_h, float temperature, float normalize, float* avg_out, float* conv_out, float* locs,
                         double* keypoints, float* scores, void* stream) {
    using namespace pp;
    PP_REQUIRE(B >= 0 && K > 0 && H > 0 && W > 0, PP_ERR_INVALID_ARG, "pp_probmap_(head_)decode: bad B/K/H/W");
    if (B == 0) return PP_OK;  // empty batch: nothing to read or write (buffers may be NULL)
    PP_REQUIRE(hm && taps && radius && locs && keypoints && scores, PP_ERR_INVALID_ARG,
               "pp_probmap_decode: hm, taps, radius, locs, keypoints and scores must be non-NULL");
    PP_REQUIRE(!hm_flip || flip_indices, PP_ERR_INVALID_ARG,
               "pp_probmap_decode: flip_indices is required when hm_flip is given");
    PP_REQUIRE(H >= RM && W >= RM, PP_ERR_UNSUPPORTED,
               "pp_probmap_decode: heatmap smaller than the largest OKS-kernel radius (9)");
    const size_t lds = decode_lds_bytes(H, W);
    PP_REQUIRE(lds <= 160 * 1024, PP_ERR_UNSUPPORTED, "pp_probmap_decode: heatmap too large for one CU's LDS");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (from_logits) {
        PP_REQUIRE(H * W <= MAX_EPT * DEC_THREADS, PP_ERR_UNSUPPORTED,
                   "pp_probmap_head_decode: H*W exceeds the in-register Sparsemax row capacity (7168)");
        PP_REQUIRE(temperature > 0.f, PP_ERR_INVALID_ARG, "pp_probmap_head_decode: temperature must be positive");
    }
    auto kern = from_logits ? (hm_flip ? probmap_decode_kernel<true, true> : probmap_decode_kernel<false, true>)
                            : (hm_flip ? probmap_decode_kernel<true, false> : probmap_decode_kernel<false, false>);
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    hipLaunchKernelGGL(kern, dim3(B * K), dim3(DEC_THREADS), lds, s, hm, hm_flip, flip_indices, taps, radius, K, H,
                       W, in_w, in_h, temperature, normalize, avg_out, conv_out, locs, keypoints, scores);
    PP_LAUNCH_CHECK();
    return PP_OK;
}

extern "C" int pp_probmap_decode(const float* hm, const float* hm_flip, const int32_t* flip_indices,
                                 const double* taps, const int32_t* radius, int B, int K, int H, int W,
                                 double in_w, double in_h, float* avg_out, float* conv_out, float* locs,
                                 double* keypoints, float* scores, void* stream) {
    return decode_launch(false, hm, hm_flip, flip_indices, taps, radius, B, K, H, W, in_w, in_h, 1.f, 1.f, avg_out,
                         conv_out, locs, keypoints, scores, stream);
}

extern "C" int pp_probmap_head_decode(const float* logits, const float* logits_flip, const int32_t* flip_indices,
                                      const double* taps, const int32_t* radius, int B, int K, int H, int W,
                                      double in_w, double in_h, float temperature, float normalize, float* avg_out,
                                      float* conv_out, float* locs, double* keypoints, float* scores, void* stream) {
    return decode_launch(true, logits, logits_flip, flip_indices, taps, radius, B, K, H, W, in_w, in_h, temperature,
                         normalize, avg_out, conv_out, locs, keypoints, scores, stream);
}
