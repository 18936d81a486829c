// S1: AdamW over ONE flat fp32 parameter buffer (all 11.57 M HRFormer-small parameters = one launch).
// HBM-bound: reads p,g,m,v (16 B/elem) + flag, writes p,m,v (+ bf16 copy): 4 elements per lane, 16-byte accesses.
#include "pk_common.h"

__global__ void __launch_bounds__(256) k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, const uint8_t* __restrict__ flags, uint16_t* __restrict__ pb,
                                               int64_t n, const float* __restrict__ lr_dev, const int32_t* __restrict__ step_dev,
                                               float b1, float b2, float eps, float wd, float gscale) {
    const float lr = lr_dev[0];
    const float t = (float)step_dev[0];
    const float bc1 = 1.f - powf(b1, t), bc2s = sqrtf(1.f - powf(b2, t));
    const float step_size = lr / bc1;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 P = reinterpret_cast<float4*>(p)[i], G = reinterpret_cast<const float4*>(g)[i];
        float4 M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
        const uint32_t f4 = reinterpret_cast<const uint32_t*>(flags)[i];
        float* pp = &P.x; float* gg = &G.x; float* mm = &M.x; float* vv = &V.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t f = (f4 >> (8 * j)) & 0xff;
            if (f & 2u) {  // bit1: parameter takes part in optimisation (receives gradients)
                const float gr = gg[j] * gscale;
                if (f & 1u) pp[j] *= 1.f - lr * wd;  // bit0: decoupled weight decay
                mm[j] = b1 * mm[j] + (1.f - b1) * gr;
                vv[j] = b2 * vv[j] + (1.f - b2) * gr * gr;
                pp[j] -= step_size * mm[j] / (sqrtf(vv[j]) / bc2s + eps);
            }
        }
        reinterpret_cast<float4*>(p)[i] = P;
        reinterpret_cast<float4*>(m)[i] = M;
        reinterpret_cast<float4*>(v)[i] = V;
        if (pb) {
            uint2 o;
            o.x = pack_bf16x2(P.x, P.y);
            o.y = pack_bf16x2(P.z, P.w);
            reinterpret_cast<uint2*>(pb)[i] = o;
        }
    }
}

extern "C" int pk_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const uint8_t* decay_mask,
                             uint16_t* param_bf16, int64_t n, const float* lr_dev, const int32_t* step_dev, float beta1,
                             float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
    PK_REQUIRE(param && grad && exp_avg && exp_avg_sq && decay_mask && lr_dev && step_dev, "pk_adamw_step: null pointer");
    PK_REQUIRE(n > 0 && (n & 3) == 0, "pk_adamw_step: n=%lld must be a positive multiple of 4 (pad the flat buffer)", (long long)n);
    PK_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0 &&
                   ((uintptr_t)decay_mask & 3) == 0 && ((uintptr_t)param_bf16 & 7) == 0,
               "pk_adamw_step: buffers must be 16-byte aligned");
    int64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_adamw, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, decay_mask,
                       param_bf16, n, lr_dev, step_dev, beta1, beta2, eps, weight_decay, grad_scale);
    return pk_launch_status("pk_adamw_step");
}
