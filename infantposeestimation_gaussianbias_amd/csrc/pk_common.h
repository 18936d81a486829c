// Shared device/host helpers for libposekernels (gfx950 only: 64-wide wavefronts, no other target).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/posekernels.h"

// Tuning knobs.  The RELEASE library reads nothing from the environment except the four documented routing switches of the two specialised
// convolution kernels (PK_CONV8P, PK_CONV8P_MIN_TILES, PK_CONV3H, PK_CONV3H_MIN_TILES: the parity tests lower the thresholds to push small
// and ragged shapes through them): every other knob of the measurement rounds is the compile-time constant it was left at.  A tuning build
// (`make TUNING=1`, -DPK_TUNING) turns the knobs back into environment reads for A/B runs on one box.
#ifdef PK_TUNING
#include <stdlib.h>
#define PK_KNOB(name, dflt) (getenv(name) ? atol(getenv(name)) : (long)(dflt))
#else
#define PK_KNOB(name, dflt) (dflt)
#endif


#define PK_WAVE 64

void pk_set_error(const char* fmt, ...);

#define PK_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            pk_set_error(__VA_ARGS__); \
            return PK_ERR_INVALID;     \
        }                              \
    } while (0)

#define PK_SUPPORTED(cond, ...)        \
    do {                               \
        if (!(cond)) {                 \
            pk_set_error(__VA_ARGS__); \
            return PK_ERR_UNSUPPORTED; \
        }                              \
    } while (0)

static inline int pk_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pk_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return PK_OK;
}

// ---- wave / block reductions (64-lane shuffles) ----------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Sum of `v` over the whole block, returned to every thread. `red` = LDS scratch of >= 16 floats.
// Safe to call repeatedly (barriers on both sides).
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// ---- bf16 <-> f32 ----------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16, round-to-nearest-even, NaN stays NaN: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two
// values per instruction); the integer bit trick costs ~6 VALU instructions per value and dominated the epilogues.
typedef __attribute__((ext_vector_type(2))) float pk_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 pk_bf16x2;
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((pk_f32x2){lo, hi}, pk_bf16x2));
}

// ---- GELU (exact-erf form) shared by the GEMM epilogues and the fused block kernels ---------------------------
// Phi(x) = 0.5 (1 + erf(x / sqrt 2)) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7: fp32 rounding level, four
// orders of magnitude below the bf16 rounding of every tensor these activations are stored to).  Written for instruction
// count -- the fused MLP kernels are VALU-bound on exactly this code (25 M hidden elements per block at B = 64):
//   hw(x) = 0.5 (1 - erf(|x| / sqrt 2)) = t (a1' + t (a2' + ...)) exp(-x^2 / 2),  t = 1 / (1 + p |x| / sqrt 2),  a' = a / 2
//   gelu(x)  = max(x, 0) - |x| hw          (x >= 0: x (1 - hw);  x < 0: x hw)
//   gelu'(x) = Phi(x) + x exp(-x^2/2) / sqrt(2 pi),   Phi = x >= 0 ? 1 - hw : hw
// 1 v_rcp + 1 v_exp + 11 (gelu) / 14 (gelu') plain VALU ops.  (libm erff is ~60 instructions with a divergent branch; the first
// version of this code used __frcp_rn, which hipcc expands to a correctly rounded division: 10 instructions instead of one.)
// (-DPK_GELU_POLY=1 selects the rcp/exp-free polynomial form further down: measured in the captured training step, same box,
// alternating runs: 17.30 / 17.25 ms with it, 17.39 / 17.02 ms without -- the fused MLP kernels wait on memory and on MFMA results
// (profiles/r03_pmc_sq_counters.json: VALU active 25-42 % of the wave cycles), they do not queue on the VALU; and its 4e-5 absolute
// error is visible to the whole-model golden test.  Kept for reference, off.)
#ifndef PK_GELU_POLY
#define PK_GELU_POLY 0
#endif
typedef __attribute__((ext_vector_type(4))) float pk_f32x4;
#if !PK_GELU_POLY
struct GeluTerms { float hw, e; };
__device__ __forceinline__ GeluTerms gelu_terms(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__fmaf_rn(0.3275911f * 0.70710678118654752440f, ax, 1.f));
    float q = __fmaf_rn(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    q = __fmaf_rn(q, t, 0.5f * 1.421413741f);
    q = __fmaf_rn(q, t, 0.5f * -0.284496736f);
    q = __fmaf_rn(q, t, 0.5f * 0.254829592f);
    const float e = __builtin_amdgcn_exp2f((x * x) * (-0.5f * 1.44269504088896340736f));
    return GeluTerms{(q * t) * e, e};
}
__device__ __forceinline__ float gelu_erf(float x) {
    const GeluTerms g = gelu_terms(x);
    return __fmaf_rn(-fabsf(x), g.hw, fmaxf(x, 0.f));
}
__device__ __forceinline__ float gelu_cdf(float x, float hw) { return 0.5f + copysignf(0.5f - hw, x); }
__device__ __forceinline__ float gelu_grad(float x) {
    const GeluTerms g = gelu_terms(x);
    return __fmaf_rn(x * 0.39894228040143267794f, g.e, gelu_cdf(x, g.hw));
}
// value and derivative from one evaluation of the shared terms
__device__ __forceinline__ void gelu_both(float x, float& val, float& grad) {
    const GeluTerms g = gelu_terms(x);
    val = __fmaf_rn(-fabsf(x), g.hw, fmaxf(x, 0.f));
    grad = __fmaf_rn(x * 0.39894228040143267794f, g.e, gelu_cdf(x, g.hw));
}
// componentwise 4-wide forms
__device__ __forceinline__ pk_f32x4 gelu_erf(pk_f32x4 x) { return (pk_f32x4){gelu_erf(x[0]), gelu_erf(x[1]), gelu_erf(x[2]), gelu_erf(x[3])}; }
__device__ __forceinline__ pk_f32x4 gelu_grad(pk_f32x4 x) { return (pk_f32x4){gelu_grad(x[0]), gelu_grad(x[1]), gelu_grad(x[2]), gelu_grad(x[3])}; }
__device__ __forceinline__ void gelu_both(pk_f32x4 x, pk_f32x4& val, pk_f32x4& grad) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a, b;
        gelu_both(x[i], a, b);
        val[i] = a;
        grad[i] = b;
    }
}
#endif
// The rcp and the exp of the form above are quarter-rate instructions (together as expensive as its eleven plain ones), and none of it maps
// onto the packed fp32 pipe.  erf(x / sqrt 2) = xc P(xc^2), xc = clamp(x, -4.2, 4.2), P of degree 8 (minimax fit, scripts/fit_gelu_poly.py;
// |error| <= 1.5e-5 inside the interval, 2.7e-5 = 1 - erf(4.2 / sqrt 2) beyond it; gelu: <= 4.4e-5 absolute, gelu': <= 1.1e-5 -- two
// orders of magnitude below the bf16 rounding of the tensors the results are stored to): one v_med3 + multiplies and FMAs only, and in the
// 4-wide form (an MFMA accumulator tile) every multiply / FMA is a v_pk_mul_f32 / v_pk_fma_f32 on two values: 7.5 issue slots per value
// instead of 19.  Used where GELU is the measured bound: the forward-only wide MLP kernel (k_mlp_fwd_w: 70 M hidden elements per launch
// at C = 80, 38 of its 83 us).  The training kernels keep the A&S form (measured there: no gain in the step, and their hidden is recomputed
// in backward with the same function either way); -DPK_GELU_POLY=1 switches every call site over (A/B builds).
#define PK_GELU_CLAMP 4.2f
#define PK_GELU_C0 0.79781485f
#define PK_GELU_C1 -0.13272066f
#define PK_GELU_C2 0.019660205f
#define PK_GELU_C3 -0.0022282742f
#define PK_GELU_C4 0.0001891444f
#define PK_GELU_C5 -1.1521039e-05f
#define PK_GELU_C6 4.6870278e-07f
#define PK_GELU_C7 -1.1265554e-08f
#define PK_GELU_C8 1.1994644e-10f
__device__ __forceinline__ float erf_rsqrt2(float x) {           // erf(x / sqrt 2)
    const float xc = __builtin_amdgcn_fmed3f(x, -PK_GELU_CLAMP, PK_GELU_CLAMP);
    const float s = xc * xc;
    float q = __fmaf_rn(PK_GELU_C8, s, PK_GELU_C7);
    q = __fmaf_rn(q, s, PK_GELU_C6);
    q = __fmaf_rn(q, s, PK_GELU_C5);
    q = __fmaf_rn(q, s, PK_GELU_C4);
    q = __fmaf_rn(q, s, PK_GELU_C3);
    q = __fmaf_rn(q, s, PK_GELU_C2);
    q = __fmaf_rn(q, s, PK_GELU_C1);
    q = __fmaf_rn(q, s, PK_GELU_C0);
    return xc * q;
}
__device__ __forceinline__ pk_f32x4 erf_rsqrt2(pk_f32x4 x) {
    pk_f32x4 xc;
#pragma unroll
    for (int i = 0; i < 4; ++i) xc[i] = __builtin_amdgcn_fmed3f(x[i], -PK_GELU_CLAMP, PK_GELU_CLAMP);
    const pk_f32x4 s = xc * xc;
    pk_f32x4 q = __builtin_elementwise_fma((pk_f32x4)(PK_GELU_C8), s, (pk_f32x4)(PK_GELU_C7));
    q = __builtin_elementwise_fma(q, s, (pk_f32x4)(PK_GELU_C6));
    q = __builtin_elementwise_fma(q, s, (pk_f32x4)(PK_GELU_C5));
    q = __builtin_elementwise_fma(q, s, (pk_f32x4)(PK_GELU_C4));
    q = __builtin_elementwise_fma(q, s, (pk_f32x4)(PK_GELU_C3));
    q = __builtin_elementwise_fma(q, s, (pk_f32x4)(PK_GELU_C2));
    q = __builtin_elementwise_fma(q, s, (pk_f32x4)(PK_GELU_C1));
    q = __builtin_elementwise_fma(q, s, (pk_f32x4)(PK_GELU_C0));
    return xc * q;
}
__device__ __forceinline__ float gelu_erf_poly(float x) {
    const float h = 0.5f * x;
    return __fmaf_rn(h, erf_rsqrt2(x), h);
}
__device__ __forceinline__ pk_f32x4 gelu_erf_poly(pk_f32x4 x) {
    const pk_f32x4 h = x * 0.5f;
    return __builtin_elementwise_fma(h, erf_rsqrt2(x), h);
}
#if PK_GELU_POLY
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf_poly(x); }
__device__ __forceinline__ pk_f32x4 gelu_erf(pk_f32x4 x) { return gelu_erf_poly(x); }
// gelu'(x) = Phi(x) + x exp(-x^2 / 2) / sqrt(2 pi)
__device__ __forceinline__ float gelu_grad(float x) {
    const float e = __builtin_amdgcn_exp2f((x * x) * (-0.5f * 1.44269504088896340736f));
    return __fmaf_rn(x * 0.39894228040143267794f, e, __fmaf_rn(0.5f, erf_rsqrt2(x), 0.5f));
}
__device__ __forceinline__ pk_f32x4 gelu_grad(pk_f32x4 x) {
    const pk_f32x4 a = (x * x) * (-0.5f * 1.44269504088896340736f);
    pk_f32x4 e;
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(a[i]);
    const pk_f32x4 cdf = __builtin_elementwise_fma((pk_f32x4)(0.5f), erf_rsqrt2(x), (pk_f32x4)(0.5f));
    return __builtin_elementwise_fma(x * 0.39894228040143267794f, e, cdf);
}
__device__ __forceinline__ void gelu_both(float x, float& val, float& grad) {
    const float er = erf_rsqrt2(x), h = 0.5f * x;
    const float e = __builtin_amdgcn_exp2f((x * x) * (-0.5f * 1.44269504088896340736f));
    val = __fmaf_rn(h, er, h);
    grad = __fmaf_rn(x * 0.39894228040143267794f, e, __fmaf_rn(0.5f, er, 0.5f));
}
__device__ __forceinline__ void gelu_both(pk_f32x4 x, pk_f32x4& val, pk_f32x4& grad) {
    const pk_f32x4 er = erf_rsqrt2(x), h = x * 0.5f;
    const pk_f32x4 a = (x * x) * (-0.5f * 1.44269504088896340736f);
    pk_f32x4 e;
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(a[i]);
    val = __builtin_elementwise_fma(h, er, h);
    grad = __builtin_elementwise_fma(x * 0.39894228040143267794f, e, __builtin_elementwise_fma((pk_f32x4)(0.5f), er, (pk_f32x4)(0.5f)));
}
#endif
__device__ __forceinline__ float softplus_(float v) { return v > 20.f ? v : log1pf(__expf(v)); }

// Sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), result in every lane of the row: two quad permutes and two row
// rotations, i.e. four VALU instructions.  (`__shfl_xor` compiles to ds_bpermute_b32 -- an LDS-crossbar instruction with LDS
// latency; the BatchNorm-statistics epilogue of the conv kernel issued 128 of them per wave and tile, ~1 us per wave, which is most
// of a shallow conv's run time.)  Fixed order: deterministic.
__device__ __forceinline__ float row16_sum(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, true));    // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true));    // row_ror:8
#endif
    return v;
}


// Cross-row combines for values laid out as 4 rows x 16 columns per wave (lane = 16 * row + column): the xor-16 / xor-32 butterfly
// steps.  xor 32: the gfx950 half swap v_permlane32_swap (VALU rate; with both operands the same register it returns low half twice |
// high half twice, and combining the two results is the butterfly step).  xor 16: ds_swizzle_b32 in bit-mask mode (lane ^ 16 inside each
// 32-lane half: the LDS crossbar, no LDS memory).  It was v_permlane16_swap until round 4: in k_attn_fwd_w the LayerNorm sums taken with
// it came out different from run to run (a few lanes of a few windows, one bf16 ulp in the output) whenever other kernels shared the chip --
// only with other streams active, only the 16-row swap, only in the reductions that run while buffer loads are in flight; padding it with
// wait states or a full s_waitcnt changed nothing, replacing it made 0 of 80 forwards differ instead of 50-60 (scripts/probes/
// cfg5_locate2.py).  Same operand pairs as before: bitwise the same sums.
__device__ __forceinline__ float xor16_partner(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));       // and 0x1f, or 0, xor 0x10
#else
    return v;
#endif
}
__device__ __forceinline__ float xor16_sum(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return v + xor16_partner(v);
#else
    return v;
#endif
}
__device__ __forceinline__ float xor32_sum(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#else
    return v;
#endif
}
__device__ __forceinline__ float xor16_max(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fmaxf(v, xor16_partner(v));
#else
    return v;
#endif
}
__device__ __forceinline__ float xor32_max(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
#else
    return v;
#endif
}

// All-reduce (sum) over groups of L consecutive lanes (L = 1, 2, 4, ..., 64), result in every lane: the xor butterfly with DPP
// (quad permutes, half-row / row mirrors) inside a 16-lane row and the row swaps above across rows.  Same operand pairs at every
// level as `v += __shfl_xor(v, o)`, i.e. bitwise the same sums, without ds_bpermute.
template <int L>
__device__ __forceinline__ float lanes_sum(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (L >= 2) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    if (L >= 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    if (L >= 8) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));     // row_half_mirror
    if (L >= 16) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));    // row_mirror
    if (L >= 32) v = xor16_sum(v);
    if (L >= 64) v = xor32_sum(v);
#endif
    return v;
}
