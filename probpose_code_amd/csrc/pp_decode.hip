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

constexpr int RM = PP_MAX_RADIUS;  // largest radius: the x-padding the row pass may read
constexpr int PAD = 12;            // x-padding of the LDS map either side: >= RM and a multiple of 4, so that a pixel quad (y, 4 q .. 4 q + 3)
                                   // is a 16-byte aligned slot (ds_write_b128 / ds_read_b128: one conflict-free access per quad)
constexpr int ROWD_SLACK = 7;      // the fp64 row slots' pitch is W + 0..7, chosen per workgroup (conv_banded)
#ifndef PP_DEC_THREADS
#define PP_DEC_THREADS 256  // dev A/B: 128 (two waves per workgroup, 64 x 48 maps only) measured 53 us against 36: the chain gets longer, nothing is saved
#endif
constexpr int DEC_THREADS = PP_DEC_THREADS;
#ifndef PP_DEC_GX
#define PP_DEC_GX 7
#endif
constexpr int GX = PP_DEC_GX;  // outputs per work item in the row pass (sliding register window). Odd: consecutive lanes then read the map at a
                       // stride of 7 dwords (all 32 banks of a ds_read_b32 lane group distinct) and write the fp64 slots at 14 dwords
                       // (conflict-free per 16-lane group); 6 was 2-way on both
constexpr int GY = 4;  // outputs per work item in the column pass
constexpr int RED_BYTES = 512;  // cross-wave reduction scratch at the head of the dynamic LDS region

// i / d for 0 <= i < 2^20, 1 <= d <= 2^12 in three VALU instructions (the integer division is ~25, and the kernel did some forty
// of them per thread): (i + 0.5) / d is at least 0.5 / d away from an integer, the float product is off by < q 2^-22.
struct FastDiv {
    float r;
    __device__ __forceinline__ explicit FastDiv(int d) : r(1.0f / (float)d) {}
    __device__ __forceinline__ int operator()(int i) const { return (int)(((float)i + 0.5f) * r); }
};

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

// One BAND of output rows at a time. The row-pass results a band needs - its own rows and R rows either side, mirrored at the
// map's top / bottom edge (scipy 'reflect'), exact zeros where the source row lies outside the support box - go to `rowd`
// (NR row slots of W doubles); the column pass of the band reads nothing else. A Sparsemax map's dilated box fits one band;
// a dense map takes several (the R rows either side are then computed again for the next band). With NR slots instead of
// H + 2 RM rows a workgroup's LDS can be 31 KiB instead of 49 at 64 x 48 (five workgroups per CU) and 78 KiB instead of 100 at
// 96 x 72 (two instead of one). At bs 64 five and four per CU measured SLOWER than three (46 / 40 / 36 us: the launch is bound
// by instructions and the dispatcher balances 4.25 workgroups per CU better than residency does; a 35-slot buffer also
// splits the box of a radius-9 keypoint): decode_rowd_doubles sizes the buffer for option "decode_wgs_per_cu" (3) per CU.
//
// The convolved map takes the place of the averaged map (convf == mapf, pitch W < Wp: output rows <= y only ever cover
// map rows <= y), and later bands still read the map. So only the LAST band writes its outputs in place; an earlier band
// parks them behind its row slots (a third of the slots then: band = 2/3 (NR - 2R) rows) until the NEXT band's row pass
// is through - that pass needs map rows >= (first row of the next band) - R, all of them below the outputs flushed by
// then (band > R - 1: decode_rowd_doubles keeps NR >= 2 RM + 16). Every thread flushes exactly the cells it parks (same
// item -> thread map in every band), so the park needs no barrier of its own.
template <int R>
__device__ __forceinline__ ArgBest conv_banded(float* __restrict__ mapf, double* __restrict__ rowd,
                                               const double* __restrict__ tap_g, int H, int W, int Wp, int cap, int tid,
                                               Box bx) {
    double tap[R + 1];  // the kernel is symmetric bit for bit (exp(-t^2/2s) / sum): R + 1 distinct factors, in SGPRs
#pragma unroll
    for (int j = 0; j <= R; ++j) tap[j] = tap_g[j];
    const int xa = max(bx.x0 - R, 0), xb = min(bx.x1 + R, W - 1);  // columns the support reaches through the row kernel
    const int ya = max(bx.y0 - R, 0), yb = min(bx.y1 + R, H - 1);
    const int nx = xb - xa + 1, nxg = (nx + GX - 1) / GX;
    // Column pass: lane -> (output rows 4 yg .. 4 yg + 3, column x), x fastest, nx4 = nx rounded up to a multiple of 4 columns per
    // row group. The fp64 slots' pitch Wr is picked so that a step of one row group (4 rows = 8 Wr dwords) equals 2 nx4 dwords
    // mod 64: the lanes' ds_read_b64 addresses are then LINEAR in the lane index mod 64 dwords - no two lanes of a 32-lane
    // group share a bank whatever the box width (with pitch W = 48 a row group was 384 = 0 mod 64 dwords from the previous one:
    // every box narrower than 32 columns was 2-way).
    const int nx4 = (nx + 3) & ~3;
    int Wr = W + ((((nx4 >> 2) - W) % 8 + 8) & 7);  // 4 Wr == nx4 (mod 32)
    int NR = cap / Wr;                              // row slots the buffer (`cap` doubles) holds at this pitch
    if (NR < yb - ya + 1 + 2 * R) {                 // the box would no longer fit ONE band: the plain pitch (bands cost more than conflicts)
        Wr = W;
        NR = cap / W;
    }
    const FastDiv div_nx(nx4), div_nxg(nxg);
    const bool one_band = yb - ya + 1 <= NR - 2 * R;
    const int band = one_band ? yb - ya + 1 : ((NR - 2 * R) * 2) / 3;
    float* convf = mapf;
    float* park = reinterpret_cast<float*>(rowd + (band + 2 * R) * Wr);  // [band][nx], several bands only
    ArgBest best{-__builtin_inff(), 0x7fffffff};
    for (int by0 = ya; by0 <= yb; by0 += band) {
        const int by1 = min(by0 + band - 1, yb);
        const bool last = by1 == yb;
        // ---- row pass: slot s <- sum_t map[reflect(by0 - R + s)][x + t] * tap[t], f64
        const int nslot = by1 - by0 + 1 + 2 * R;
        for (int it = tid; it < nslot * nxg; it += DEC_THREADS) {
            const int s = div_nxg(it), x0 = xa + (it - s * nxg) * GX;
            int y = by0 - R + s;
            y = y < 0 ? -1 - y : (y >= H ? 2 * H - 1 - y : y);  // half-sample symmetric: -1 -> 0, H -> H - 1
            double acc[GX];
#pragma unroll
            for (int g = 0; g < GX; ++g) acc[g] = 0.0;
            if (y >= bx.y0 && y <= bx.y1) {
                const float* p = mapf + y * Wp + x0 + (PAD - R);  // window starts R samples left of output x0
                double win[GX + 2 * R];
#pragma unroll
                for (int c = 0; c < GX + 2 * R; ++c) win[c] = (double)p[c];
#pragma unroll
                for (int j = 0; j <= 2 * R; ++j)  // t ascending, one fused multiply-add per tap
#pragma unroll
                    for (int g = 0; g < GX; ++g) acc[g] = fma(win[g + j], tap[j <= R ? j : 2 * R - j], acc[g]);
            }
#pragma unroll
            for (int g = 0; g < GX; ++g)
                if (x0 + g <= xb) rowd[s * Wr + x0 + g] = acc[g];
        }
        __syncthreads();
        // ---- column pass + running argmax. Before a thread parks a cell it flushes what the previous band left there.
        const int nyg = (by1 - by0 + GY) / GY;
        for (int it = tid; it < nx4 * (one_band ? nyg : (band + GY - 1) / GY); it += DEC_THREADS) {
            const int yg = div_nx(it), xo = it - yg * nx4, x = xa + xo, s0 = yg * GY;
            if (xo >= nx) continue;  // (the up to three lanes that pad a row group to nx4)
            if (by0 > ya) {
#pragma unroll
                for (int g = 0; g < GY; ++g)
                    if (s0 + g < band) convf[(by0 - band + s0 + g) * W + x] = park[(s0 + g) * nx + xo];
            }
            if (yg >= nyg) continue;  // (a shorter last band: flush only)
            const double* p = rowd + s0 * Wr + x;
            double win[GY + 2 * R];
#pragma unroll
            for (int c = 0; c < GY + 2 * R; ++c) win[c] = (s0 + c < NR) ? p[c * Wr] : 0.0;  // (slots past the band's last feed only outputs that are dropped)
            double acc[GY];
#pragma unroll
            for (int g = 0; g < GY; ++g) acc[g] = 0.0;
#pragma unroll
            for (int j = 0; j <= 2 * R; ++j)
#pragma unroll
                for (int g = 0; g < GY; ++g) acc[g] = fma(win[g + j], tap[j <= R ? j : 2 * R - j], acc[g]);
#pragma unroll
            for (int g = 0; g < GY; ++g) {
                if (by0 + s0 + g <= by1) {
                    const float v = (float)acc[g];  // the single rounding scipy does on output
                    const int idx = (by0 + s0 + g) * W + x;
                    if (last) convf[idx] = v;
                    else park[(s0 + g) * nx + xo] = v;
                    if (better(v, idx, best.v, best.idx)) best = ArgBest{v, idx};
                }
            }
        }
        if (!last) __syncthreads();  // the slots are written again
    }
    return best;
}

// ---- lane exchanges of the xor butterfly (32, 16, 8, 4, 2, 1 - the pairing order __shfl_xor loops have, so sums keep their
// bits) without the LDS queue: the gfx950 row swaps for 32 / 16 (each lane ends up with its own and its partner's value: any
// commutative op takes them in either order), DPP row rotate / shifts / quad permutes below that.
__device__ __forceinline__ int dpp_xor8(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); }  // row_ror:8
__device__ __forceinline__ int dpp_xor4(int v) {
    const int t = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xf, 0x5, false);  // row_shl:4 into lanes 0-3, 8-11 of a row
    return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xf, 0xa, false);         // row_shr:4 into lanes 4-7, 12-15
}
__device__ __forceinline__ int dpp_xor2(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false); }  // quad_perm [2,3,0,1]
__device__ __forceinline__ int dpp_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false); }  // quad_perm [1,0,3,2]

template <class T, class Op>
__device__ __forceinline__ T wave_allreduce(T v, Op op) {
    static_assert(sizeof(T) == 4, "one dword");
    auto I = [](T x) { return __builtin_bit_cast(int, x); };
    auto V = [](int x) { return __builtin_bit_cast(T, x); };
    {
        const auto s = __builtin_amdgcn_permlane32_swap((unsigned)I(v), (unsigned)I(v), false, false);
        v = op(V((int)s[0]), V((int)s[1]));
    }
    {
        const auto s = __builtin_amdgcn_permlane16_swap((unsigned)I(v), (unsigned)I(v), false, false);
        v = op(V((int)s[0]), V((int)s[1]));
    }
    v = op(v, V(dpp_xor8(I(v))));
    v = op(v, V(dpp_xor4(I(v))));
    v = op(v, V(dpp_xor2(I(v))));
    v = op(v, V(dpp_xor1(I(v))));
    return v;
}

// ---- block-wide reductions for the in-register Sparsemax (4 waves)
struct SmxStat {
    float s0, s1;
    int n;  // candidates of the two rows, n0 | n1 << 16 (a row has at most 12 288 pixels)
};

__device__ __forceinline__ void block_max2(float& a, float& b, float* scratch) {
    a = wave_allreduce(a, [](float x, float y) { return fmaxf(x, y); });
    b = wave_allreduce(b, [](float x, float y) { return fmaxf(x, y); });
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

// `quiet`: no lane of this wave has a candidate left (wave-uniform) - its partial sums are zeros without the exchanges
__device__ __forceinline__ SmxStat block_sum_stat(SmxStat v, bool quiet, SmxStat* scratch) {
    if (!quiet) {
        v.s0 = wave_allreduce(v.s0, [](float x, float y) { return x + y; });
        v.s1 = wave_allreduce(v.s1, [](float x, float y) { return x + y; });
        v.n = wave_allreduce(v.n, [](int x, int y) { return x + y; });
    }
    if (lane_id() == 0) scratch[wave_id()] = v;
    __syncthreads();
    SmxStat r = scratch[0];
#pragma unroll
    for (int w = 1; w < DEC_THREADS / WAVE; ++w) {  // fixed order: every thread gets the same bits
        r.s0 += scratch[w].s0;
        r.s1 += scratch[w].s1;
        r.n += scratch[w].n;
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
    float* __restrict__ locs, double* __restrict__ keypoints, float* __restrict__ scores, int phased, int cap, int shift) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int bk = blockIdx.x;
    const int b = bk / K, k = bk - b * K;
    const int Wp = W + 2 * PAD;
    const int HW = H * W, HW4 = HW >> 2, W4 = W >> 2;
    const FastDiv div_w(W), div_w4(W4);
    // A thread owns the pixel QUADS q = tid + 256 e (e < NV): pixels (y, x0 .. x0 + 3), y = q / (W / 4), x0 = 4 (q % (W / 4)) - the same
    // quads whatever the memory layout of the input, so both layouts give the same bits; a quad is one 16-byte aligned LDS slot.
    // (Measured and dropped, round 4: lane pairs (l, l + 32) sharing an octet, numbered row-parity-major, so that a half-wave reads 512
    // contiguous bytes of one phase block and two v_permlane32_swap interleave the columns - 39.3 / 37.3 us against 37.8 / 35.8 for the
    // two 8-byte loads per quad below: the octet order puts the lanes' 16-byte LDS stores 32 bytes apart, 2-way conflicts.)
    auto quad = [&](int e, int& y, int& x0) -> bool {
        const int i4 = tid + e * DEC_THREADS;
        if (i4 >= HW4) return false;
        y = div_w4(i4);
        x0 = (i4 - y * W4) * 4;
        return true;
    };

    // all LDS in the one dynamic region (16-B aligned carve offsets)
    ArgBest* red = reinterpret_cast<ArgBest*>(smem);                                         // [4] cross-wave argmax
    float* mapf = reinterpret_cast<float*>(smem + RED_BYTES);                                // [H][Wp] (+ slack) averaged map, x-padded
    double* rowd = reinterpret_cast<double*>(smem + RED_BYTES + (((H * Wp + 32) * 4 + 15) & ~15));  // [NR][W + 0..7] row pass of one band (conv_banded)
    // [H][W] convolved map (f32): it takes the place of the averaged map, which is dead once the last row pass is through -
    // except for the one value at the final argmax (the score), so every thread parks its share of the map (4 NV values) in
    // registers first.
    float* convf = mapf;

    const f32x4* src = reinterpret_cast<const f32x4*>(hm + (size_t)bk * HW);
    const f32x4* srcf = nullptr;
    if (HAS_FLIP) srcf = reinterpret_cast<const f32x4*>(hm_flip + ((size_t)b * K + flip_indices[k]) * HW);

    if constexpr (!FROM_LOGITS) {
        // ---- load + flip-back + average: partner of pixels (y, x..x+3) is the reversed vector at (y, W-4-x)
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            int y, x;
            if (quad(e, y, x)) {
                f32x4 v = src[y * W4 + (x >> 2)];
                if (HAS_FLIP && !shift) {
                    const f32x4 f = srcf[y * W4 + (W4 - 1 - (x >> 2))];
                    v = (v + f32x4{f[3], f[2], f[1], f[0]}) * 0.5f;
                } else if (HAS_FLIP) {
                    // shift_heatmap (tta.py:64-66): the flipped-back map moves one pixel to the right, column 0 keeps its value:
                    // pixel x takes the flipped pass's pixel W - x (x >= 1), pixel 0 its pixel W - 1
                    const float* fr_ = reinterpret_cast<const float*>(srcf) + y * W;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int xx = x + j;
                        v[j] = (v[j] + fr_[xx >= 1 ? W - xx : W - 1]) * 0.5f;
                    }
                }
                *reinterpret_cast<f32x4*>(mapf + y * Wp + PAD + x) = v;  // 16-byte aligned slot
                if (avg_out) reinterpret_cast<f32x4*>(avg_out + (size_t)bk * HW)[y * W4 + (x >> 2)] = v;
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
        float m0 = -__builtin_inff(), m1 = -__builtin_inff(), chk = 0.f;
        // a division costs nine VALU instructions, the row has 24 per thread: multiply when the temperature is a normal power of two
        const unsigned t_bits = __builtin_bit_cast(unsigned, temperature);
        const bool t_pow2 = (t_bits & 0x007fffffu) == 0 && (t_bits >> 23) >= 2 && (t_bits >> 23) <= 252;
        const float t_inv = __builtin_bit_cast(float, (254u << 23) - t_bits);
        // The quad (y, x0 ..) of a row in memory. Row-major: one 16-byte vector. Phase-separated (the fused deconvolution head
        // writes the four 2x2 output phases one after the other, each a (H/2, W/2) row-major block): two 8-byte pairs - columns
        // x0 / 2, x0 / 2 + 1 of row y / 2 of the phases (y & 1, 0) and (y & 1, 1) - interleaved.
        auto load_quad = [&](const f32x4* base, bool ok, int y, int x0) -> f32x4 {
            const float ninf = -__builtin_inff();
            f32x4 r{ninf, ninf, ninf, ninf};
            if (!ok) return r;
            if (!phased) return base[y * W4 + (x0 >> 2)];
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2* pb = reinterpret_cast<const f32x2*>(base) + ((y & 1) * HW4 + (y >> 1) * W4 + (x0 >> 2));  // (in pairs: a phase block is HW4 / 2 pairs)
            const f32x2 a = pb[0], c = pb[HW4 >> 1];
            return f32x4{a[0], c[0], a[1], c[1]};
        };
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            int y = 0, x0 = 0;
            const bool ok = quad(e, y, x0);
            z0[e] = load_quad(src, ok, y, x0);
            z1[e] = z0[e];
            if (HAS_FLIP) z1[e] = load_quad(srcf, ok, y, x0);
            if (t_pow2) {  // x / 2^k == x * 2^-k bit for bit (-inf of a lane without a quad stays -inf)
                z0[e] = z0[e] * t_inv;
                if (HAS_FLIP) z1[e] = z1[e] * t_inv;
            } else {
                z0[e] = z0[e] / temperature;
                if (HAS_FLIP) z1[e] = z1[e] / temperature;
            }
            if (!HAS_FLIP) { const float ninf = -__builtin_inff(); z1[e] = f32x4{ninf, ninf, ninf, ninf}; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m0 = fmaxf(m0, z0[e][j]);
                m1 = fmaxf(m1, z1[e][j]);
                if (ok) {  // x * 0 is NaN for x = +-inf / NaN: one fma per logit finds a non-finite input (fmaxf would skip a NaN)
                    chk = __builtin_fmaf(z0[e][j], 0.f, chk);
                    if (HAS_FLIP) chk = __builtin_fmaf(z1[e][j], 0.f, chk);
                }
            }
        }
        if (chk != chk) m0 = __builtin_inff();
        block_max2(m0, m1, fscr);
        // A non-finite logit (an operand beyond the split-fp16 range upstream - numeric domain, include/probpose_mi355x.h - or a NaN input) must not
        // decode to pixel 0 with a plausible score: the keypoint comes out as NaN, which the host mirror turns into a FloatingPointError.
        if (!(fabsf(m0) < __builtin_inff()) || (HAS_FLIP && !(fabsf(m1) < __builtin_inff()))) {  // (workgroup-uniform)
            const float qnan = __builtin_nanf("");
            if (avg_out)
                for (int i = tid; i < HW; i += DEC_THREADS) avg_out[(size_t)bk * HW + i] = qnan;
            if (conv_out)
                for (int i = tid; i < HW; i += DEC_THREADS) conv_out[(size_t)bk * HW + i] = qnan;
            if (tid == 0) {
                locs[2 * bk + 0] = locs[2 * bk + 1] = qnan;
                keypoints[2 * bk + 0] = keypoints[2 * bk + 1] = (double)qnan;
                scores[bk] = qnan;
            }
            return;
        }
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
        int prev = -1;
        bool alive = true;  // the thresholds only rise: a thread without a candidate now has none in any later round
        // COMPACT form of the threshold search (round 6). The search only ever looks at the candidates of its FIRST threshold (z > max - 1: the
        // thresholds rise, nothing comes back) - a few dozen of a row's 3 072 logits on a peaked map - yet every round walked all 24 (48 with
        // the flipped pass) values of every thread and paid a block-wide reduction: ~260 instructions x 4 - 6 rounds of the kernel's ~2 550 per
        // wave. Now the first threshold's candidates are packed into an LDS list once (a block-wide prefix sum of the per-thread counts) and
        // every wave runs the rounds on the list by itself: a lane per candidate, wave reductions, no barrier. Same rounds, same thresholds up
        // to the order of the fp32 sums. Rows with more than SMX_CAP candidates (a flat map) keep the walk below.
        constexpr int SMX_CAP = 1024;
        bool compact_done = false;
        if (smx) {
            int c = 0;
#pragma unroll
            for (int e = 0; e < NV; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c += z0[e][j] > -1.0f ? 1 : 0;
                    if (HAS_FLIP) c += z1[e][j] > -1.0f ? (1 << 16) : 0;
                }
            int inc = c;  // inclusive prefix over the wave's lanes (both counts in one word: a row has 3 072 .. 12 288 values)
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) {
                const int t = __shfl_up(inc, o);
                if (lane_id() >= o) inc += t;
            }
            int* wtot = reinterpret_cast<int*>(smem + 192);
            if (lane_id() == WAVE - 1) wtot[wave_id()] = inc;
            __syncthreads();
            int base = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < DEC_THREADS / WAVE; ++w) {
                const int t = wtot[w];
                base += w < wave_id() ? t : 0;
                tot += t;
            }
            const int n0 = tot & 0xffff, n1 = tot >> 16;
            if (n0 <= SMX_CAP && n1 <= SMX_CAP) {  // (workgroup-uniform)
                float* L0 = mapf;            // the map region is not written before the thresholds are known
                float* L1 = mapf + SMX_CAP;
                const int off = base + inc - c;
                int o0 = off & 0xffff, o1 = off >> 16;
#pragma unroll
                for (int e = 0; e < NV; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (z0[e][j] > -1.0f) L0[o0++] = z0[e][j];
                        if (HAS_FLIP && z1[e][j] > -1.0f) L1[o1++] = z1[e][j];
                    }
                __syncthreads();
                for (int iter = 0; iter < 64; ++iter) {
                    float s0 = 0.f, s1 = 0.f;
                    int n = 0;
                    for (int i = lane_id(); i < n0; i += WAVE) {
                        const float d0 = L0[i] - tau0;
                        if (d0 > 0.f) {
                            s0 += d0;
                            n += 1;
                        }
                    }
                    if (HAS_FLIP)
                        for (int i = lane_id(); i < n1; i += WAVE) {
                            const float d1 = L1[i] - tau1;
                            if (d1 > 0.f) {
                                s1 += d1;
                                n += 1 << 16;
                            }
                        }
                    s0 = wave_allreduce(s0, [](float x, float y) { return x + y; });
                    if (HAS_FLIP) s1 = wave_allreduce(s1, [](float x, float y) { return x + y; });
                    n = wave_allreduce(n, [](int x, int y) { return x + y; });
                    if (n == prev) break;
                    prev = n;
                    tau0 = tau0 + (s0 - 1.0f) / (float)(n & 0xffff);
                    if (HAS_FLIP) tau1 = tau1 + (s1 - 1.0f) / (float)(n >> 16);
                }
                compact_done = true;
                __syncthreads();  // every wave is through with the lists before the map takes their place
            }
        }
        for (int iter = 0; iter < ((smx && !compact_done) ? 64 : 0); ++iter) {
            SmxStat st{0.f, 0.f, 0};
            if (alive) {
#pragma unroll
                for (int e = 0; e < NV; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d0 = z0[e][j] - tau0;
                        if (d0 > 0.f) {
                            st.s0 += d0;
                            st.n += 1;
                        }
                        if (HAS_FLIP) {
                            const float d1 = z1[e][j] - tau1;
                            if (d1 > 0.f) {
                                st.s1 += d1;
                                st.n += 1 << 16;
                            }
                        }
                    }
                alive = st.n != 0;
            }
            st = block_sum_stat(st, __builtin_amdgcn_ballot_w64(alive) == 0, sscr + (iter & 1) * (DEC_THREADS / WAVE));
            if (st.n == prev) break;
            prev = st.n;
            tau0 = tau0 + (st.s0 - 1.0f) / (float)(st.n & 0xffff);
            if (HAS_FLIP) tau1 = tau1 + (st.s1 - 1.0f) / (float)(st.n >> 16);
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            int y, x0;
            if (quad(e, y, x0)) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = clamp01(fmaxf(z0[e][j] - tau0, 0.0f) * normalize);
                *reinterpret_cast<f32x4*>(mapf + y * Wp + PAD + x0) = v;
            }
        }
        if (HAS_FLIP) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                int y, xf;
                if (quad(e, y, xf)) {
                    // pixels xf .. xf + 3 of the flipped pass land at W-1-xf .. W-4-xf: the reversed quad at column W-4-xf; exactly
                    // one thread owns each cell
                    if (!shift) {
                        f32x4* d = reinterpret_cast<f32x4*>(mapf + y * Wp + PAD + (W - 4 - xf));
                        f32x4 v = *d;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[3 - j] = (v[3 - j] + clamp01(fmaxf(z1[e][j] - tau1, 0.0f) * normalize)) * 0.5f;
                        *d = v;
                    } else {
                        // shift_heatmap (tta.py:64-66): flipped pixel q lands at column W - q (q >= 1; q = 0 falls off), and the last
                        // one, q = W - 1, also at column 0 - still exactly one thread per cell
                        float* row = mapf + y * Wp + PAD;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int q = xf + j;
                            const float pv = clamp01(fmaxf(z1[e][j] - tau1, 0.0f) * normalize);
                            if (q >= 1) row[W - q] = (row[W - q] + pv) * 0.5f;
                            if (q == W - 1) row[0] = (row[0] + pv) * 0.5f;
                        }
                    }
                }
            }
        }
        if (avg_out) {
            __syncthreads();
            for (int i4 = tid; i4 < HW4; i4 += DEC_THREADS) {
                const int y = div_w4(i4), x = (i4 - y * W4) * 4;
                reinterpret_cast<f32x4*>(avg_out + (size_t)bk * HW)[i4] = *reinterpret_cast<const f32x4*>(mapf + y * Wp + PAD + x);
            }
        }
    }
    __syncthreads();
    // ---- the averaged map is complete. Every thread parks its share (its quads) in registers - the score is the
    // RAW map value at the final argmax, and the convolved map will take this region's place - and the workgroup finds
    // the bounding box of the non-zero pixels: a Sparsemax row has 2 - 10 px of support, and an output further than the
    // kernel radius from that box is a sum of exact zeros (fp64 accumulation of +0 terms, reflect padding included: a
    // mirrored sample lies within the radius of the border it mirrors). Only the dilated box is convolved; the rest
    // is written as 0.0f - bit for bit what scipy.ndimage.convolve returns there.
    f32x4 mine[NV];
    Box bx{1 << 20, -1, 1 << 20, -1};
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        int y = 0, x = 0;
        const bool ok = quad(e, y, x);
        mine[e] = ok ? *reinterpret_cast<const f32x4*>(mapf + y * Wp + PAD + x) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (ok && mine[e][j] != 0.f) {  // (NaN counts as non-zero)
                bx.y0 = min(bx.y0, y); bx.y1 = max(bx.y1, y);
                bx.x0 = min(bx.x0, x + j); bx.x1 = max(bx.x1, x + j);
            }
    }
    {
        bx.y0 = wave_allreduce(bx.y0, [](int a, int b) { return min(a, b); });
        bx.y1 = wave_allreduce(bx.y1, [](int a, int b) { return max(a, b); });
        bx.x0 = wave_allreduce(bx.x0, [](int a, int b) { return min(a, b); });
        bx.x1 = wave_allreduce(bx.x1, [](int a, int b) { return max(a, b); });
        int* bscr = reinterpret_cast<int*>(smem + 256);  // (the Sparsemax scratch below it is out of use past the barrier above)
        if (lane_id() == 0) {
            bscr[4 * wave_id() + 0] = bx.y0; bscr[4 * wave_id() + 1] = bx.y1;
            bscr[4 * wave_id() + 2] = bx.x0; bscr[4 * wave_id() + 3] = bx.x1;
        }
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
    for (int i = tid; i < (empty ? 0 : (bx.y1 - bx.y0 + 1) * 2 * RM); i += DEC_THREADS) {  // (the row pass reads no other row)
        const int yr = i / (2 * RM), y = bx.y0 + yr, p = i - yr * (2 * RM);
        if (p < RM)
            mapf[y * Wp + (PAD - 1 - p)] = mapf[y * Wp + PAD + p];
        else
            mapf[y * Wp + PAD + W + (p - RM)] = mapf[y * Wp + PAD + W - 1 - (p - RM)];
    }
    __syncthreads();

    const int r = __builtin_amdgcn_readfirstlane(radius[k]);
    const double* tap_g = taps + k * PP_MAX_TAPS;
    // ---- separable convolution over the dilated box, band by band; the outputs land in convf, the rest of it is 0.0f
    ArgBest best{-__builtin_inff(), 0x7fffffff};
    if (!empty) switch (r) {
        case 0: best = conv_banded<0>(mapf, rowd, tap_g, H, W, Wp, cap, tid, bx); break;
        case 1: best = conv_banded<1>(mapf, rowd, tap_g, H, W, Wp, cap, tid, bx); break;
        case 2: best = conv_banded<2>(mapf, rowd, tap_g, H, W, Wp, cap, tid, bx); break;
        case 3: best = conv_banded<3>(mapf, rowd, tap_g, H, W, Wp, cap, tid, bx); break;
        case 4: best = conv_banded<4>(mapf, rowd, tap_g, H, W, Wp, cap, tid, bx); break;
        case 5: best = conv_banded<5>(mapf, rowd, tap_g, H, W, Wp, cap, tid, bx); break;
        case 6: best = conv_banded<6>(mapf, rowd, tap_g, H, W, Wp, cap, tid, bx); break;
        case 7: best = conv_banded<7>(mapf, rowd, tap_g, H, W, Wp, cap, tid, bx); break;
        case 8: best = conv_banded<8>(mapf, rowd, tap_g, H, W, Wp, cap, tid, bx); break;
        default: best = conv_banded<9>(mapf, rowd, tap_g, H, W, Wp, cap, tid, bx); break;
    }
    const int cxa = max(bx.x0 - r, 0), cxb = min(bx.x1 + r, W - 1), cya = max(bx.y0 - r, 0), cyb = min(bx.y1 + r, H - 1);
    if (conv_out) {  // zero outside the box (every read of the map is behind a barrier by now; disjoint from the band outputs)
        const int xa = cxa, xb = cxb, ya = cya, yb = cyb;
        for (int i = tid; i < HW; i += DEC_THREADS) {
            const int y = div_w(i), x = i - y * W;
            if (empty || y < ya || y > yb || x < xa || x > xb) convf[i] = 0.0f;
        }
    }
    // wave-level then block-level argmax
    {
        auto F = [](int x) { return __builtin_bit_cast(float, x); };
        auto I = [](float x) { return __builtin_bit_cast(int, x); };
        auto pick = [](ArgBest a, ArgBest b) { return better(a.v, a.idx, b.v, b.idx) ? a : b; };  // (a strict order: symmetric)
        {
            const auto sv = __builtin_amdgcn_permlane32_swap((unsigned)I(best.v), (unsigned)I(best.v), false, false);
            const auto si = __builtin_amdgcn_permlane32_swap((unsigned)best.idx, (unsigned)best.idx, false, false);
            best = pick(ArgBest{F((int)sv[0]), (int)si[0]}, ArgBest{F((int)sv[1]), (int)si[1]});
        }
        {
            const auto sv = __builtin_amdgcn_permlane16_swap((unsigned)I(best.v), (unsigned)I(best.v), false, false);
            const auto si = __builtin_amdgcn_permlane16_swap((unsigned)best.idx, (unsigned)best.idx, false, false);
            best = pick(ArgBest{F((int)sv[0]), (int)si[0]}, ArgBest{F((int)sv[1]), (int)si[1]});
        }
        best = pick(ArgBest{F(dpp_xor8(I(best.v))), dpp_xor8(best.idx)}, best);
        best = pick(ArgBest{F(dpp_xor4(I(best.v))), dpp_xor4(best.idx)}, best);
        best = pick(ArgBest{F(dpp_xor2(I(best.v))), dpp_xor2(best.idx)}, best);
        best = pick(ArgBest{F(dpp_xor1(I(best.v))), dpp_xor1(best.idx)}, best);
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
        const int wy = div_w(bb.idx), wx = bb.idx - wy * W, wq = wy * W4 + (wx >> 2);  // the winner's quad: its owner and slot
        const int own_e = wq / DEC_THREADS, own_tid = wq & (DEC_THREADS - 1);
        if (own_tid == tid) {
            float sv = mine[0][0];
#pragma unroll
            for (int e = 0; e < NV; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) sv = (own_e == e && (wx & 3) == j) ? mine[e][j] : sv;
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
            // (outside the convolved box the map is 0.0f; only written there when the caller asked for the map)
            auto cv = [&](int y, int x) { return (empty || y < cya || y > cyb || x < cxa || x > cxb) ? 0.0f : convf[y * W + x]; };
            const float c = cv(yi, xi);
            const float xp = cv(yi, xi + 1), xm = cv(yi, xi - 1);
            const float yp = cv(yi + 1, xi), ym = cv(yi - 1, xi);
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

static size_t decode_fixed_bytes(int H, int W) {
    return RED_BYTES + ((((size_t)H * (W + 2 * PAD) + 32) * 4 + 15) & ~(size_t)15);  // (the convolved map reuses the averaged map's region)
}

// Capacity of the band buffer in doubles: as many workgroups per CU as the map allows (five at 64 x 48, two at 96 x 72), never
// more than every row of the map at the widest pitch, a band never lower than 16 output rows at the largest radius and the
// widest pitch. 1.5 KiB of every share stay free for the allocation granule.
static long decode_rowd_doubles(int H, int W) {
    const size_t fixed = decode_fixed_bytes(H, W);
    const size_t full = (size_t)(H + 2 * RM) * (W + ROWD_SLACK);
    for (int n = std::min(5, std::max(1, pp::option("decode_wgs_per_cu"))); n >= 1; --n) {
        const size_t budget = (size_t)160 * 1024 / n - 1536;
        if (budget <= fixed) continue;
        const size_t cap = std::min(full, (budget - fixed) / 8);
        if (cap == full || cap >= (size_t)(2 * RM + 16) * (W + ROWD_SLACK)) return (long)cap;
    }
    return -1;
}

typedef void (*DecodeKernel)(const float*, const float*, const int32_t*, const double*, const int32_t*, int, int, int,
                             double, double, float, float, float*, float*, float*, double*, float*, int, int, int);

template <int NV>
static DecodeKernel pick_kernel(bool from_logits, bool flip) {
    if (from_logits) return flip ? probmap_decode_kernel<true, true, NV> : probmap_decode_kernel<false, true, NV>;
    return flip ? probmap_decode_kernel<true, false, NV> : probmap_decode_kernel<false, false, NV>;
}

}  // namespace pp

static int decode_launch(bool from_logits, const float* hm, const float* hm_flip, const int32_t* flip_indices,
                         const double* taps, const int32_t* radius, int B, int K, int H, int W, double in_w,
                         double in_h, float temperature, float normalize, float* avg_out, float* conv_out, float* locs,
                         double* keypoints, float* scores, void* stream, int phased = 0, int shift = 0) {
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
    const long cap = decode_rowd_doubles(H, W);
    PP_REQUIRE(cap > 0, PP_ERR_UNSUPPORTED, "pp_probmap_decode: heatmap too large for one CU's LDS");
    const size_t lds = decode_fixed_bytes(H, W) + (size_t)cap * 8;
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
                       W, in_w, in_h, temperature, normalize, avg_out, conv_out, locs, keypoints, scores, phased, (int)cap, (hm_flip && shift) ? 1 : 0);
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

extern "C" int pp_probmap_decode_flags(const float* maps, const float* maps_flip, const int32_t* flip_indices, const double* taps,
                                       const int32_t* radius, int B, int K, int H, int W, double in_w, double in_h, float temperature,
                                       float normalize, float* avg_out, float* conv_out, float* locs, double* keypoints, float* scores,
                                       int flags, void* stream) {
    using namespace pp;
    PP_REQUIRE((flags & ~(PP_DECODE_LOGITS | PP_DECODE_PHASED | PP_DECODE_SHIFT_HEATMAP)) == 0, PP_ERR_INVALID_ARG, "pp_probmap_decode_flags: unknown flag");
    const bool logits = (flags & PP_DECODE_LOGITS) != 0;
    PP_REQUIRE(logits || !(flags & PP_DECODE_PHASED), PP_ERR_INVALID_ARG, "pp_probmap_decode_flags: PP_DECODE_PHASED needs PP_DECODE_LOGITS");
    return decode_launch(logits, maps, maps_flip, flip_indices, taps, radius, B, K, H, W, in_w, in_h, logits ? temperature : 1.f,
                         logits ? normalize : 1.f, avg_out, conv_out, locs, keypoints, scores, stream, (flags & PP_DECODE_PHASED) ? 1 : 0,
                         (flags & PP_DECODE_SHIFT_HEATMAP) ? 1 : 0);
}
