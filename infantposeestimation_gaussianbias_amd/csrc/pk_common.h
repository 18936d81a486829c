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
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {  // round-to-nearest-even, NaN stays NaN
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
