// ProbMap decode for gfx950, fused with the head's Sparsemax and the flip-test average:
//   logits -> x / T -> Sparsemax over the H*W pixels -> * normalize -> clamp(0,1)      (probmap_head.py:637-646)
//   -> flip-back + channel permutation + average of the two test-time passes           (tta.py:35-39, probmap_head.py:757-763)
//   -> separable OKS-kernel convolution, f64 accumulate, one rounding to f32           (post_processing.py:13-39,347-352)
//   -> first-occurrence argmax -> one Newton sub-pixel step in f32 -> rescale (f64)     (post_processing.py:354-430, probmap.py:218)
// One 256-thread workgroup per (crop, keypoint); the whole map lives in LDS, the inputs are read from HBM exactly
// once with 16-byte lane loads and only the results (plus the optional maps) are written.
//
// Everything data-dependent but wave-uniform is a template parameter so the hot loops are branch-free register
// code: the OKS-kernel radius (0..9, dispatched by a scalar switch) and the row length per thread.
#include "pp_common.h"

// numpy evaluates the f32 sub-pixel expressions one rounding per operator; keep it so.
#pragma clang fp contract(off)

namespace pp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int RM = PP_MAX_RADIUS;  // every map is padded by the largest radius
constexpr int DEC_THREADS = 256;
constexpr int GX = 6;  // outputs per work item in the row pass (sliding register window)
constexpr int GY = 4;  // outputs per work item in the column pass
constexpr int RED_BYTES = 512;  // cross-wave reduction scratch at the head of the dynamic LDS region

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

// ---- separable convolution passes, radius known at compile time
// Both passes run over a BOX of outputs only (see the kernel: the support of the map dilated by the radius); everything
// outside is exactly zero in the reference's result too.
struct Box {
    int y0, y1, x0, x1;  // inclusive; empty when y1 < y0
};

template <int R>
__device__ __forceinline__ void row_pass(const float* __restrict__ mapf, double* __restrict__ rowd,
                                         const double* __restrict__ tap_g, int H, int W, int Wp, int tid, Box bx) {
    double tap[R + 1];  // the kernel is symmetric bit for bit (exp(-t^2/2s) / sum): R + 1 distinct factors, in SGPRs
#pragma unroll
    for (int j = 0; j <= R; ++j) tap[j] = tap_g[j];
    const int xa = max(bx.x0 - R, 0), xb = min(bx.x1 + R, W - 1);  // columns the support reaches through the row kernel
    const int nxg = (xb - xa + GX) / GX, ny = bx.y1 - bx.y0 + 1;
    for (int it = tid; it < ny * nxg; it += DEC_THREADS) {
        const int yr = it / nxg, y = bx.y0 + yr, x0 = xa + (it - yr * nxg) * GX;
        const float* p = mapf + y * Wp + x0 + (RM - R);  // window starts R samples left of output x0
        double win[GX + 2 * R];
#pragma unroll
        for (int c = 0; c < GX + 2 * R; ++c) win[c] = (double)p[c];
        double acc[GX];
#pragma unroll
        for (int g = 0; g < GX; ++g) acc[g] = 0.0;
#pragma unroll
        for (int j = 0; j <= 2 * R; ++j)  // t ascending, one fused multiply-add per tap
#pragma unroll
            for (int g = 0; g < GX; ++g) acc[g] = fma(win[g + j], tap[j <= R ? j : 2 * R - j], acc[g]);
#pragma unroll
        for (int g = 0; g < GX; ++g)
            if (x0 + g <= xb) rowd[(y + RM) * W + x0 + g] = acc[g];
    }
}

template <int R>
__device__ __forceinline__ ArgBest col_pass(const double* __restrict__ rowd, float* __restrict__ convf,
                                            const double* __restrict__ tap_g, int H, int W, int tid, Box bx) {
    double tap[R + 1];
#pragma unroll
    for (int j = 0; j <= R; ++j) tap[j] = tap_g[j];
    ArgBest best{-__builtin_inff(), 0x7fffffff};
    const int xa = max(bx.x0 - R, 0), xb = min(bx.x1 + R, W - 1);
    const int ya = max(bx.y0 - R, 0), yb = min(bx.y1 + R, H - 1);
    const int nyg = (yb - ya + GY) / GY, nx = xb - xa + 1;
    for (int it = tid; it < nx * nyg; it += DEC_THREADS) {
        const int yg = it / nx, x = xa + it - yg * nx, y0 = ya + yg * GY;
        const double* p = rowd + (y0 + RM - R) * W + x;
        double win[GY + 2 * R];
#pragma unroll
        for (int c = 0; c < GY + 2 * R; ++c) win[c] = (y0 + RM - R + c < H + 2 * RM) ? p[c * W] : 0.0;
        double acc[GY];
#pragma unroll
        for (int g = 0; g < GY; ++g) acc[g] = 0.0;
#pragma unroll
        for (int j = 0; j <= 2 * R; ++j)
#pragma unroll
            for (int g = 0; g < GY; ++g) acc[g] = fma(win[g + j], tap[j <= R ? j : 2 * R - j], acc[g]);
#pragma unroll
        for (int g = 0; g < GY; ++g) {
            if (y0 + g <= yb) {
                const float v = (float)acc[g];  // the single rounding scipy does on output
                const int idx = (y0 + g) * W + x;
                convf[idx] = v;
                if (better(v, idx, best.v, best.idx)) best = ArgBest{v, idx};
            }
        }
    }
    return best;
}

// ---- block-wide reductions for the in-register Sparsemax (4 waves)
struct SmxStat {
    float s0, s1;
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

__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

// NV = 16-byte vectors of the H*W row per thread (3 for 64x48, 7 for 96x72). Needs W % 4 == 0.
template <bool HAS_FLIP, bool FROM_LOGITS, int NV>
__global__ __launch_bounds__(DEC_THREADS) void probmap_decode_kernel(
    const float* __restrict__ hm, const float* __restrict__ hm_flip, const int32_t* __restrict__ flip_indices,
    const double* __restrict__ taps, const int32_t* __restrict__ radius, int K, int H, int W, double in_w,
    double in_h, float temperature, float normalize, float* __restrict__ avg_out, float* __restrict__ conv_out,
    float* __restrict__ locs, double* __restrict__ keypoints, float* __restrict__ scores, int phased) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int bk = blockIdx.x;
    const int b = bk / K, k = bk - b * K;
    const int Wp = W + 2 * RM;
    const int HW = H * W, HW4 = HW >> 2, W4 = W >> 2;

    // all LDS in the one dynamic region (16-B aligned carve offsets)
    ArgBest* red = reinterpret_cast<ArgBest*>(smem);                                         // [4] cross-wave argmax
    float* mapf = reinterpret_cast<float*>(smem + RED_BYTES);                                // [H][Wp] (+ slack) averaged map, x-padded
    double* rowd = reinterpret_cast<double*>(smem + RED_BYTES + (((H * Wp + 32) * 4 + 15) & ~15));  // [H+2RM][W] row pass, y-padded
    // [H][W] convolved map (f32): it takes the place of the averaged map, which is dead once the row pass is through - except
    // for the one value at the final argmax (the score), so every thread parks its share of the map (4 NV values) in
    // registers first. 49 KiB instead of 61 at 64 x 48: three workgroups per CU.
    float* convf = mapf;

    const f32x4* src = reinterpret_cast<const f32x4*>(hm + (size_t)bk * HW);
    const f32x4* srcf = nullptr;
    if (HAS_FLIP) srcf = reinterpret_cast<const f32x4*>(hm_flip + ((size_t)b * K + flip_indices[k]) * HW);

    if constexpr (!FROM_LOGITS) {
        // ---- load + flip-back + average: partner of pixels (y, x..x+3) is the reversed vector at (y, W-4-x)
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const int i4 = tid + e * DEC_THREADS;
            if (i4 < HW4) {
                const int y = i4 / W4, x = (i4 - y * W4) * 4;
                f32x4 v = src[i4];
                if (HAS_FLIP) {
                    const f32x4 f = srcf[y * W4 + (W4 - 1 - (x >> 2))];
                    v = (v + f32x4{f[3], f[2], f[1], f[0]}) * 0.5f;
                }
                float* d = mapf + y * Wp + RM + x;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
                if (avg_out) reinterpret_cast<f32x4*>(avg_out + (size_t)bk * HW)[i4] = v;
            }
        }
    } else {
        // ---- Sparsemax of this keypoint's row and (flip test) of its mirror partner's row, both in registers.
        // Sort-free threshold search (Michelot): candidates z > tau, tau <- tau + (sum_cand (z - tau) - 1) / |cand|,
        // starting from tau = max - 1 (a lower bound of the solution), until no candidate is dropped. The
        // correction form keeps the sums O(1), so fp32 accumulation loses nothing against the fp32 reference.
        float* fscr = reinterpret_cast<float*>(smem);
        SmxStat* sscr = reinterpret_cast<SmxStat*>(smem + 64);
        f32x4 z0[NV], z1[NV];
        float m0 = -__builtin_inff(), m1 = -__builtin_inff();
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const int i4 = tid + e * DEC_THREADS;
            const float ninf = -__builtin_inff();
            z0[e] = f32x4{ninf, ninf, ninf, ninf};
            z1[e] = z0[e];
            if (i4 < HW4) {
                z0[e] = src[i4] / temperature;
                if (HAS_FLIP) z1[e] = srcf[i4] / temperature;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m0 = fmaxf(m0, z0[e][j]);
                m1 = fmaxf(m1, z1[e][j]);
            }
        }
        block_max2(m0, m1, fscr);
        // normalize < 0 stands for the head's `normalize=None` (probmap_head.py:249,642-646): no Sparsemax, the map is
        // clamp(x / T, 0, 1) - the same code with threshold 0, no shift and scale 1
        const bool smx = normalize >= 0.f;
        if (!smx) {
            m0 = m1 = 0.f;
            normalize = 1.f;
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            z0[e] -= m0;
            z1[e] -= m1;
        }
        float tau0 = smx ? -1.0f : 0.f, tau1 = tau0;
        int prev0 = -1, prev1 = -1;
        for (int iter = 0; iter < (smx ? 64 : 0); ++iter) {
            SmxStat st{0.f, 0.f, 0, 0};
#pragma unroll
            for (int e = 0; e < NV; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d0 = z0[e][j] - tau0;
                    if (d0 > 0.f) {
                        st.s0 += d0;
                        st.n0 += 1;
                    }
                    if (HAS_FLIP) {
                        const float d1 = z1[e][j] - tau1;
                        if (d1 > 0.f) {
                            st.s1 += d1;
                            st.n1 += 1;
                        }
                    }
                }
            st = block_sum_stat(st, sscr + (iter & 1) * (DEC_THREADS / WAVE));
            if (st.n0 == prev0 && (!HAS_FLIP || st.n1 == prev1)) break;
            prev0 = st.n0;
            prev1 = st.n1;
            tau0 = tau0 + (st.s0 - 1.0f) / (float)st.n0;
            if (HAS_FLIP) tau1 = tau1 + (st.s1 - 1.0f) / (float)st.n1;
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const int i4 = tid + e * DEC_THREADS;
            if (i4 < HW4) {
                // memory order of the four values -> pixel (y, x0 + j * dx): row-major, or (logits written by the fused
                // deconvolution head) the four 2x2 output phases one after the other, each a (H/2, W/2) row-major block
                int y = i4 / W4, x0 = (i4 - y * W4) * 4, dx = 1;
                if (phased) {
                    const int e0 = i4 * 4, q = HW >> 2, z = e0 / q, rr = e0 - z * q, yy = rr / (W >> 1);
                    y = 2 * yy + (z >> 1);
                    x0 = 2 * (rr - yy * (W >> 1)) + (z & 1);
                    dx = 2;
                }
                float* d = mapf + y * Wp + RM + x0;
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j * dx] = clamp01(fmaxf(z0[e][j] - tau0, 0.0f) * normalize);
            }
        }
        if (HAS_FLIP) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                const int i4 = tid + e * DEC_THREADS;
                if (i4 < HW4) {
                    int y = i4 / W4, xf = (i4 - y * W4) * 4, dx = 1;
                    if (phased) {
                        const int e0 = i4 * 4, q = HW >> 2, z = e0 / q, rr = e0 - z * q, yy = rr / (W >> 1);
                        y = 2 * yy + (z >> 1);
                        xf = 2 * (rr - yy * (W >> 1)) + (z & 1);
                        dx = 2;
                    }
                    float* d = mapf + y * Wp + RM + (W - 1 - xf);  // pixel xf + j dx of the flipped pass lands at W-1-xf-j dx
#pragma unroll
                    for (int j = 0; j < 4; ++j) {                // exactly one thread owns each cell
                        const float p = clamp01(fmaxf(z1[e][j] - tau1, 0.0f) * normalize);
                        d[-j * dx] = (d[-j * dx] + p) * 0.5f;
                    }
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
    // ---- the averaged map is complete. Every thread parks its share (pixels tid + 256 i) in registers - the score is the
    // RAW map value at the final argmax, and the convolved map will take this region's place - and the workgroup finds
    // the bounding box of the non-zero pixels: a Sparsemax row has 2 - 10 px of support, and an output further than the
    // kernel radius from that box is a sum of exact zeros (fp64 accumulation of +0 terms, reflect padding included: a
    // mirrored sample lies within the radius of the border it mirrors). Only the dilated box is convolved; the rest
    // is written as 0.0f - bit for bit what scipy.ndimage.convolve returns there.
    float mine[4 * NV];
    Box bx{1 << 20, -1, 1 << 20, -1};
#pragma unroll
    for (int i = 0; i < 4 * NV; ++i) {
        const int px = tid + i * DEC_THREADS;
        const int y = px / W, x = px - y * W;
        mine[i] = px < HW ? mapf[y * Wp + RM + x] : 0.f;
        if (px < HW && mine[i] != 0.f) {  // (NaN counts as non-zero)
            bx.y0 = min(bx.y0, y); bx.y1 = max(bx.y1, y);
            bx.x0 = min(bx.x0, x); bx.x1 = max(bx.x1, x);
        }
    }
    {
#pragma unroll
        for (int off = WAVE / 2; off > 0; off >>= 1) {
            bx.y0 = min(bx.y0, __shfl_xor(bx.y0, off)); bx.y1 = max(bx.y1, __shfl_xor(bx.y1, off));
            bx.x0 = min(bx.x0, __shfl_xor(bx.x0, off)); bx.x1 = max(bx.x1, __shfl_xor(bx.x1, off));
        }
        int* bscr = reinterpret_cast<int*>(smem + 256);  // (the Sparsemax scratch below it is out of use past the barrier above)
        if (lane_id() == 0) {
            bscr[4 * wave_id() + 0] = bx.y0; bscr[4 * wave_id() + 1] = bx.y1;
            bscr[4 * wave_id() + 2] = bx.x0; bscr[4 * wave_id() + 3] = bx.x1;
        }
        // the row-pass result starts from zeros: rows outside the box feed the column pass as exact zeros
        double* rz = rowd;
        for (int i = tid; i < (H + 2 * RM) * W; i += DEC_THREADS) rz[i] = 0.0;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < DEC_THREADS / WAVE; ++w) {
            bx.y0 = min(bx.y0, bscr[4 * w + 0]); bx.y1 = max(bx.y1, bscr[4 * w + 1]);
            bx.x0 = min(bx.x0, bscr[4 * w + 2]); bx.x1 = max(bx.x1, bscr[4 * w + 3]);
        }
        bx.y0 = __builtin_amdgcn_readfirstlane(bx.y0); bx.y1 = __builtin_amdgcn_readfirstlane(bx.y1);
        bx.x0 = __builtin_amdgcn_readfirstlane(bx.x0); bx.x1 = __builtin_amdgcn_readfirstlane(bx.x1);
    }
    const bool empty = bx.y1 < bx.y0;
    // half-sample symmetric x-padding (scipy.ndimage 'reflect'): -1 -> 0, -2 -> 1, W -> W-1, ...
    for (int i = tid; i < H * 2 * RM; i += DEC_THREADS) {
        const int y = i / (2 * RM), p = i - y * (2 * RM);
        if (p < RM)
            mapf[y * Wp + (RM - 1 - p)] = mapf[y * Wp + RM + p];
        else
            mapf[y * Wp + RM + W + (p - RM)] = mapf[y * Wp + RM + W - 1 - (p - RM)];
    }
    __syncthreads();

    const int r = __builtin_amdgcn_readfirstlane(radius[k]);
    const double* tap_g = taps + k * PP_MAX_TAPS;
    // ---- row pass: rowd[y][x] = sum_t map[y][x+t] * tap[t], f64
    if (!empty) switch (r) {
        case 0: row_pass<0>(mapf, rowd, tap_g, H, W, Wp, tid, bx); break;
        case 1: row_pass<1>(mapf, rowd, tap_g, H, W, Wp, tid, bx); break;
        case 2: row_pass<2>(mapf, rowd, tap_g, H, W, Wp, tid, bx); break;
        case 3: row_pass<3>(mapf, rowd, tap_g, H, W, Wp, tid, bx); break;
        case 4: row_pass<4>(mapf, rowd, tap_g, H, W, Wp, tid, bx); break;
        case 5: row_pass<5>(mapf, rowd, tap_g, H, W, Wp, tid, bx); break;
        case 6: row_pass<6>(mapf, rowd, tap_g, H, W, Wp, tid, bx); break;
        case 7: row_pass<7>(mapf, rowd, tap_g, H, W, Wp, tid, bx); break;
        case 8: row_pass<8>(mapf, rowd, tap_g, H, W, Wp, tid, bx); break;
        default: row_pass<9>(mapf, rowd, tap_g, H, W, Wp, tid, bx); break;
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
    __syncthreads();  // the row pass has read the map for the last time: its region becomes the convolved map,
    for (int i = tid; i < HW; i += DEC_THREADS) convf[i] = 0.0f;  // zero outside the box
    __syncthreads();

    // ---- column pass + running argmax
    ArgBest best{-__builtin_inff(), 0x7fffffff};
    if (!empty) switch (r) {
        case 0: best = col_pass<0>(rowd, convf, tap_g, H, W, tid, bx); break;
        case 1: best = col_pass<1>(rowd, convf, tap_g, H, W, tid, bx); break;
        case 2: best = col_pass<2>(rowd, convf, tap_g, H, W, tid, bx); break;
        case 3: best = col_pass<3>(rowd, convf, tap_g, H, W, tid, bx); break;
        case 4: best = col_pass<4>(rowd, convf, tap_g, H, W, tid, bx); break;
        case 5: best = col_pass<5>(rowd, convf, tap_g, H, W, tid, bx); break;
        case 6: best = col_pass<6>(rowd, convf, tap_g, H, W, tid, bx); break;
        case 7: best = col_pass<7>(rowd, convf, tap_g, H, W, tid, bx); break;
        case 8: best = col_pass<8>(rowd, convf, tap_g, H, W, tid, bx); break;
        default: best = col_pass<9>(rowd, convf, tap_g, H, W, tid, bx); break;
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
    // The pixels outside the convolved box hold 0.0f: the first of them (row-major) competes as well - it wins when the
    // map is all zero, or when nothing inside the box is positive.
    int zidx = -1;
    {
        const int xa = max(bx.x0 - r, 0), xb = min(bx.x1 + r, W - 1), ya = max(bx.y0 - r, 0), yb = min(bx.y1 + r, H - 1);
        if (empty || ya > 0 || xa > 0) zidx = 0;
        else if (xb < W - 1) zidx = xb + 1;
        else if (yb < H - 1) zidx = (yb + 1) * W;
    }
    ArgBest bb = red[0];
    {  // every thread finds the winner; the owner of that pixel hands its parked map value over
        for (int w = 1; w < DEC_THREADS / WAVE; ++w)
            if (better(red[w].v, red[w].idx, bb.v, bb.idx)) bb = red[w];
        if (zidx >= 0 && better(0.0f, zidx, bb.v, bb.idx)) bb = ArgBest{0.0f, zidx};
        if ((bb.idx & (DEC_THREADS - 1)) == tid) {
            float sv = mine[0];
#pragma unroll
            for (int i = 1; i < 4 * NV; ++i) sv = (bb.idx / DEC_THREADS) == i ? mine[i] : sv;
            reinterpret_cast<float*>(red + DEC_THREADS / WAVE)[0] = sv;
        }
    }
    __syncthreads();
    if (conv_out)
        for (int i = tid; i < HW; i += DEC_THREADS) conv_out[(size_t)bk * HW + i] = convf[i];

    if (tid == 0) {
        best = bb;
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
        scores[bk] = reinterpret_cast<const float*>(red + DEC_THREADS / WAVE)[0];  // raw (un-convolved) averaged map at the integer argmax
    }
}

static size_t decode_lds_bytes(int H, int W) {
    const size_t mapf = (((size_t)H * (W + 2 * RM) + 32) * 4 + 15) & ~(size_t)15;
    return RED_BYTES + mapf + (size_t)(H + 2 * RM) * W * 8;  // (the convolved map reuses the averaged map's region)
}

typedef void (*DecodeKernel)(const float*, const float*, const int32_t*, const double*, const int32_t*, int, int, int,
                             double, double, float, float, float*, float*, float*, double*, float*, int);

template <int NV>
static DecodeKernel pick_kernel(bool from_logits, bool flip) {
    if (from_logits) return flip ? probmap_decode_kernel<true, true, NV> : probmap_decode_kernel<false, true, NV>;
    return flip ? probmap_decode_kernel<true, false, NV> : probmap_decode_kernel<false, false, NV>;
}

}  // namespace pp

static int decode_launch(bool from_logits, const float* hm, const float* hm_flip, const int32_t* flip_indices,
                         const double* taps, const int32_t* radius, int B, int K, int H, int W, double in_w,
                         double in_h, float temperature, float normalize, float* avg_out, float* conv_out, float* locs,
                         double* keypoints, float* scores, void* stream, int phased = 0) {
    using namespace pp;
    PP_REQUIRE(B >= 0 && K > 0 && H > 0 && W > 0, PP_ERR_INVALID_ARG, "pp_probmap_(head_)decode: bad B/K/H/W");
    PP_REQUIRE(!phased || (from_logits && H % 2 == 0 && W % 8 == 0), PP_ERR_UNSUPPORTED,
               "pp_probmap_head_decode_phased: needs an even height and a width that is a multiple of 8");
    if (B == 0) return PP_OK;  // empty batch: nothing to read or write (buffers may be NULL)
    PP_REQUIRE(hm && taps && radius && locs && keypoints && scores, PP_ERR_INVALID_ARG,
               "pp_probmap_decode: hm, taps, radius, locs, keypoints and scores must be non-NULL");
    PP_REQUIRE(!hm_flip || flip_indices, PP_ERR_INVALID_ARG,
               "pp_probmap_decode: flip_indices is required when hm_flip is given");
    PP_REQUIRE(H >= RM && W >= RM, PP_ERR_UNSUPPORTED,
               "pp_probmap_decode: heatmap smaller than the largest OKS-kernel radius (9)");
    PP_REQUIRE(W % 4 == 0, PP_ERR_UNSUPPORTED, "pp_probmap_decode: heatmap width must be a multiple of 4");
    const size_t lds = decode_lds_bytes(H, W);
    PP_REQUIRE(lds <= 160 * 1024, PP_ERR_UNSUPPORTED, "pp_probmap_decode: heatmap too large for one CU's LDS");
    if (from_logits)
        PP_REQUIRE(temperature > 0.f, PP_ERR_INVALID_ARG, "pp_probmap_head_decode: temperature must be positive");
    const int nv = (H * W / 4 + DEC_THREADS - 1) / DEC_THREADS;
    DecodeKernel kern = nullptr;
    if (nv <= 3) kern = pick_kernel<3>(from_logits, hm_flip != nullptr);        // 64 x 48
    else if (nv <= 7) kern = pick_kernel<7>(from_logits, hm_flip != nullptr);   // 96 x 72
    else if (nv <= 12) kern = pick_kernel<12>(from_logits, hm_flip != nullptr);
    PP_REQUIRE(kern != nullptr, PP_ERR_UNSUPPORTED, "pp_probmap_decode: H*W exceeds 12288 pixels");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    hipLaunchKernelGGL(kern, dim3(B * K), dim3(DEC_THREADS), lds, s, hm, hm_flip, flip_indices, taps, radius, K, H,
                       W, in_w, in_h, temperature, normalize, avg_out, conv_out, locs, keypoints, scores, phased);
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

extern "C" int pp_probmap_head_decode_phased(const float* logits, const float* logits_flip, const int32_t* flip_indices,
                                             const double* taps, const int32_t* radius, int B, int K, int H, int W,
                                             double in_w, double in_h, float temperature, float normalize,
                                             float* avg_out, float* conv_out, float* locs, double* keypoints,
                                             float* scores, void* stream) {
    return decode_launch(true, logits, logits_flip, flip_indices, taps, radius, B, K, H, W, in_w, in_h, temperature,
                         normalize, avg_out, conv_out, locs, keypoints, scores, stream, 1);
}
