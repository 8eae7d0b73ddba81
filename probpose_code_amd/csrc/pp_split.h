// Split-fp16 operand format of PP_PREC_F16X3 (the precision mode that meets the path's 1e-3 tolerance at MFMA
// fp16 rate instead of the 16x slower fp32 MFMA).
//
// A value x (fp32) is carried as  hi = fp16(x),  lo = fp16(x - hi):  hi + lo represents x to ~2^-23 relative - fp32's
// own rounding step - and a product is  a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  (three v_mfma_f32_16x16x32_f16 with
// fp32 accumulation; the dropped lo*lo term is 2^-24 relative). The MFMA keeps fp16 subnormals (probed on gfx950,
// scripts/micro/f16_denorm_probe.hip), so small weights' low halves need no scaling.
//
// Memory layout: 4 bytes per element like fp32, in BLOCKS OF 32 ELEMENTS along the contiguous (K) axis:
//     bytes [0, 64)   32 hi halves   | bytes [64, 128)   32 lo halves
// A 128-byte block is one LDS row of a K-tile; the 16-byte chunk c (0..3) holds the eight hi halves that ONE lane
// feeds to the K = 32 MFMA (k = 8 c .. 8 c + 7) and chunk 4 + c the matching lo halves. Every row length is a
// multiple of 32 elements, so element index -> block is global: block = idx >> 5, position = idx & 31.
#pragma once
#include <hip/hip_runtime.h>

namespace pp {

struct SplitH {
    unsigned raw;  // container only (sizeof == 4); never interpreted as a number
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float split_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ _Float16 split_hi(float x) { return (_Float16)x; }
__device__ __forceinline__ _Float16 split_lo(float x, _Float16 hi) { return (_Float16)(x - (float)hi); }
// Files built with -fno-slp-vectorize (pp_ffn_split.hip, pp_ffn_dma.hip) must pin a value in a register before they split it when it
// is the result of a multiplication: the compiler then fuses product, conversion and difference into
//     v_fma_mixlo_f16 hi, a, b, 0   ;   v_fma_mix_f32 d, a, b, -hi     (d = a * b - hi without the product's fp32 rounding)
// and the (hi, lo) pairs that come out of that sequence were measurably worse on gfx950: 2e-4 instead of 1e-5 on the FFN output
// (tests/test_split_fp16.py test_ffn_split_fused_vs_fp64; the instructions themselves keep fp16 subnormals,
// scripts/micro/mix_denorm_probe.hip). With the vectoriser on the conversions pair up into v_cvt_pk_f16_f32 and the pattern never forms.
__device__ __forceinline__ void split_pin(float& x) { asm("" : "+v"(x)); }

// (hi, lo) of a PAIR of values in 4 VALU instructions instead of 8 (round 5; issue cycles per wave64 instruction and SIMD, scripts/micro/
// valu_rate.hip: v_cvt_f16_f32 / v_cvt_f32_f16 4.5, v_cvt_pk_f16_f32 and v_fma_mix_f32 4.7, v_sub_f32 3.0): the two hi halves by one
// v_cvt_pk_f16_f32 (round to nearest even, like the single conversion), each lo half as fp16(g - hi) where g - hi is ONE v_fma_mix_f32
// that reads hi straight out of the packed pair (fma(g, 1.0, -hi): exact, the same fp32 difference as convert-back-and-subtract - NOT the
// fused product form of the trap described above: g is a finished fp32 value here), packed by a second v_cvt_pk_f16_f32.
// hi = {fp16(a) | fp16(b) << 16}, lo likewise.
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    asm("" : "+v"(a));
    asm("" : "+v"(b));
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, f16x2_t));
    float la, lb;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(la) : "v"(a), "v"(hi));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(lb) : "v"(b), "v"(hi));
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{la, lb}, f16x2_t));
}

// address of the hi half of element `idx` of a split tensor (the lo half lives 64 bytes further)
__device__ __forceinline__ char* split_addr(void* base, size_t idx) {
    return reinterpret_cast<char*>(base) + (idx >> 5) * 128 + (idx & 31) * 2;
}
__device__ __forceinline__ const char* split_addr(const void* base, size_t idx) {
    return reinterpret_cast<const char*>(base) + (idx >> 5) * 128 + (idx & 31) * 2;
}

// four consecutive elements (idx % 4 == 0): two 8-byte accesses
__device__ __forceinline__ void split_store4(void* base, size_t idx, split_f32x4 v) {
    f16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = split_hi(v[j]);
        l[j] = split_lo(v[j], h[j]);
    }
    char* p = split_addr(base, idx);
    *reinterpret_cast<f16x4*>(p) = h;
    *reinterpret_cast<f16x4*>(p + 64) = l;
}
// The same for an MFMA accumulator fragment, where lane and lane ^ 16 hold the two halves of one 8-element chunk (lanes with
// bit 4 clear: elements idx .. idx + 3 with idx % 8 == 0, their partners idx + 4 ..): one row swap per dword
// (v_permlane16_swap) gives the even-row lane the whole hi chunk and the odd-row lane the whole lo chunk - ONE 16-byte store
// per lane instead of two 8-byte ones. Every lane of the wave must call it; `store` masks the store itself.
__device__ __forceinline__ void split_store4_rowpair(void* base, size_t idx, split_f32x4 v, bool store) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x2_t hu, lu;
    { unsigned h__, l__; split_pair(v[0], v[1], h__, l__); hu[0] = h__; lu[0] = l__; }
    { unsigned h__, l__; split_pair(v[2], v[3], h__, l__); hu[1] = h__; lu[1] = l__; }
    // swap(a, b): odd rows of a <-> even rows of b. Even-row lane: (own hi, partner's hi); odd-row lane: (partner's lo, own lo)
    const auto s0 = __builtin_amdgcn_permlane16_swap(hu[0], lu[0], false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(hu[1], lu[1], false, false);
    const u32x4_t q = {s0[0], s1[0], s0[1], s1[1]};
    const bool odd = (threadIdx.x & 16) != 0;
    char* p = split_addr(base, idx & ~(size_t)7) + (odd ? 64 : 0);
    if (store) *reinterpret_cast<u32x4_t*>(p) = q;
}
__device__ __forceinline__ void split_store4_rowpair_nt(void* base, size_t idx, split_f32x4 v, bool store) {  // dev: the same, non-temporal
    f16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = split_hi(v[j]);
        l[j] = split_lo(v[j], h[j]);
    }
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const u32x2_t hu = __builtin_bit_cast(u32x2_t, h), lu = __builtin_bit_cast(u32x2_t, l);
    const auto s0 = __builtin_amdgcn_permlane16_swap(hu[0], lu[0], false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(hu[1], lu[1], false, false);
    const u32x4_t q = {s0[0], s1[0], s0[1], s1[1]};
    const bool odd = (threadIdx.x & 16) != 0;
    char* p = split_addr(base, idx & ~(size_t)7) + (odd ? 64 : 0);
    if (store) __builtin_nontemporal_store(q, reinterpret_cast<u32x4_t*>(p));
}
__device__ __forceinline__ split_f32x4 split_load4(const void* base, size_t idx) {
    const char* p = split_addr(base, idx);
    const f16x4 h = *reinterpret_cast<const f16x4*>(p), l = *reinterpret_cast<const f16x4*>(p + 64);
    return split_f32x4{(float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]};
}

// acc += a * b for one K = 32 block given the hi / lo fragments of both operands (small terms first)
__device__ __forceinline__ split_f32x4 split_mma(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl, split_f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
    return c;
}

// ReLU that keeps a NaN: v_max_f32 returns the OTHER operand for a NaN input (IEEE maxNum), so fmaxf(x, 0) turns the NaN an out-of-range
// split-fp16 operand produces upstream (include/probpose_mi355x.h, numeric domain) into a clean 0 - and the heatmap branch would decode a map of
// biases to "pixel 0" with a plausible score. The heatmap branch's ReLUs use this form: the NaN reaches the logits, pp_probmap_decode_flags writes
// NaN keypoints, the host mirror raises. Two instructions more per value, in epilogues only.
__device__ __forceinline__ float relu_keep_nan(float x) {
    const float r = fmaxf(x, 0.f);
    return x != x ? x : r;
}

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) for the parity mode's epilogues. libdevice erff costs ~57 VALU instructions per
// element - 55 us of a 166 us fc1 launch at bs 64 (scripts/bench_split_gemm.py). This form is Abramowitz & Stegun 7.1.26,
//     erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2),  t = 1 / (1 + p z),  z >= 0,  |error| <= 1.5e-7,
// used WITHOUT the cancellation 1 - erfc for negative arguments: 1 + erf(-z) = erfc(z) directly; 1 + erf(z) = 2 - erfc(z)
// for z >= 0. Evaluated in fp32 against the fp64 definition over [-12, 12] (2 M points): |error| <= 4.3e-7 absolute -
// 3.5e-8 relative to the activation where it is largest (x ~ 12); in the far negative tail (GELU ~ 1e-6) the relative
// error reaches 1.7e-3 of a value that is itself 1e-6. One v_rcp_f32, one v_exp_f32, a dozen plain fp32 instructions.
__device__ __forceinline__ float gelu_erfc_as(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float q = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    q = __builtin_fmaf(t, q, 1.421413741f);
    q = __builtin_fmaf(t, q, -0.284496736f);
    q = __builtin_fmaf(t, q, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-(z * z) * 1.44269504088896340736f);  // argument <= 0: raw v_exp_f32
    const float erfc_z = t * q * e;
    return 0.5f * x * (x < 0.f ? erfc_z : 2.0f - erfc_z);
}

}  // namespace pp
