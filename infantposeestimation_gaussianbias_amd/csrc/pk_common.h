// Shared device/host helpers for libposekernels (gfx950 only: 64-wide wavefronts, no other target).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/posekernels.h"

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
