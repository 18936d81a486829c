// Fused halves of the HRFormer block (A3, hrformer.py:262-293 + Mlp :38-64) for the high-resolution branches (C = 32 / 64).
//
//   MLP half:   y = x + s * ( fc2( gelu_erf( fc1( LayerNorm2(x) ) ) ) )            one launch, forward
//               backward = two launches that RECOMPUTE LayerNorm / fc1 / GELU from x:
//                 k_mlp_bwd_dx : dx (through fc2^T, GELU', fc1^T, LayerNorm backward, + the residual gradient), dgamma/dbeta slabs
//                 k_mlp_bwd_dw : dW1, db1, dW2, db2 slabs (hidden dimension sliced over blockIdx.y so the fp32 accumulators stay
//                                in registers)
//   The 4C-wide hidden activation never exists in HBM: per token the unfused chain moved 2C (v) + 2*8C (z, h) + ... bytes per
//   direction; here a forward launch reads x once and writes y once (4C bytes per token), the backward reads x, dy twice and
//   writes dx once.
//
// Everything chains through MFMA accumulator registers (v_mfma_f32_16x16x32_bf16, weight tile = A operand):
//   * a lane of the accumulator tile D[i][j] holds column j = lane & 15 (a token) and rows i = 4g + r, g = lane >> 4 (channels);
//   * two accumulator tiles packed to bf16 ARE the B operand of the next MFMA if its contraction index is enumerated in
//     accumulator order (slot jj of lane group g  <->  row 16 (jj >> 2) + 4 g + (jj & 3) of the stacked tiles).  Hidden units
//     are an internal index, so instead of permuting the second weight matrix the ROWS of W1 (and of b1, W2^T) are gathered in
//     the order that makes accumulator order == natural order: tile t = 2s + u, row i  <->  hidden unit
//     32 s + 8 (i >> 2) + 4 u + (i & 3)   (hid() below).  Every weight fragment is then one contiguous 16-byte read.
//   * weights live in LDS in FRAGMENT ORDER: fragment f is the 1 KiB block [f][lane] of 16-byte pieces, filled once per
//     workgroup, read with lane-linear ds_read_b128 (conflict free by construction, immediate offsets).
//   * LayerNorm statistics: a token's C channels sit in the 4 lanes {j, j+16, j+32, j+48} -> two xor-shuffles.
//   * contractions over TOKENS (weight gradients) need "lane = channel, slots = tokens": activations are written row-major to a
//     wave-private LDS tile and read back with the gfx950 transpose read ds_read_b64_tr_b16 (as pk_attn.hip does for V).
// Waves are independent (one 32-token group per wave and iteration, no workgroup barrier inside the loops); all reductions
// have a fixed order (deterministic, no float atomics).
#include "pk_common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define OOB_OFF 0x80000000u
#define MAKE_RSRC(ptr) __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(ptr), 0, 0x7ffffff0, 0x00020000)
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")     // this wave's LDS writes are visible to its own reads

struct MlpArgs {
    const uint16_t* x;       // [M][C] bf16: input of the half (= residual)
    const uint16_t* dy;      // [M][C] bf16: gradient of the output (backward)
    uint16_t* out;           // forward: y; backward dx: dx
    const float *gamma, *beta, *b1, *b2;
    const float* scale;      // per-sample multiplier of the MLP branch (DropPath), or null
    const uint16_t* w1;      // fc1 weight  [4C][C]   (forward copy)
    const uint16_t* w2;      // fc2 weight  [C][4C]   (forward copy)
    const uint16_t* w1t;     // fc1 weight^T [C][4C]  (data-gradient copy)
    const uint16_t* w2t;     // fc2 weight^T [4C][C]  (data-gradient copy)
    float* part;             // backward: slabs (see the kernels)
    int M, rows_per_sample;
    float eps;
};

__device__ __forceinline__ int hid(int t, int i) { return 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3); }
__device__ __forceinline__ float blo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ bf16x8 pack2(const f32x4 a, const f32x4 b) {
    const u32x4 v = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ f32x4 unpack4(const u32x2 v) { return (f32x4){blo(v[0]), bhi(v[0]), blo(v[1]), bhi(v[1])}; }
__device__ __forceinline__ u32x2 pack4(const f32x4 v) { return (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}; }
__device__ __forceinline__ float sum8(const bf16x8 f) {
    const u32x4 v = __builtin_bit_cast(u32x4, f);
    return ((blo(v[0]) + bhi(v[0])) + (blo(v[1]) + bhi(v[1]))) + ((blo(v[2]) + bhi(v[2])) + (blo(v[3]) + bhi(v[3])));
}
// Fragment (column col0 + (lane & 15), k = the tile's 32 rows in ACCUMULATOR order: slots 0..3 = rows 4g .. 4g+3, slots 4..7 =
// rows 16 + 4g .. 16 + 4g + 3) of a row-major [32][pitch] bf16 LDS tile.  EXEC must be all ones (cross-lane gather).
__device__ __forceinline__ bf16x8 tr_frag32(const uint16_t* tile, int pitch, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const uint16_t* a0 = tile + (4 * g + (i >> 2)) * pitch + col0 + 4 * (i & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 16 * pitch));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// Stage `nfrag` weight fragments into LDS in fragment order: lane (i = l & 15, g = l >> 4) of fragment f receives the 16 bytes
// src[row(f, i)][k0(f) + 8 g .. + 7]  (`ld` = row pitch of src in elements).
template <typename RowFn, typename K0Fn>
__device__ __forceinline__ void stage_frags(u32x4* dst, const uint16_t* __restrict__ src, int nfrag, int ld, RowFn row, K0Fn k0) {
    for (int idx = threadIdx.x; idx < nfrag * 64; idx += blockDim.x) {
        const int f = idx >> 6, l = idx & 63;
        dst[idx] = *reinterpret_cast<const u32x4*>(src + (size_t)row(f, l & 15) * ld + k0(f) + 8 * (l >> 4));
    }
}
#define LDS_FRAG(base, f, lane) __builtin_bit_cast(bf16x8, (base)[(f) * 64 + (lane)])
// The weight fragments in LDS do not change from token group to token group, so LICM would hoist every ds_read out of the group
// loop into registers (C = 64: 64+ fragments = 256+ VGPRs -> occupancy 1 / spills).  An opaque copy of the lane id, made
// inside the loop, keeps the reads where they are; where the hoisted fragments are affordable (forward, C = 32: 64 VGPRs) the
// plain lane id is used and the weights end up register-resident.
template <bool OPAQUE>
__device__ __forceinline__ int opaque_lane(int lane) {
    if (OPAQUE) asm volatile("" : "+v"(lane));
    return lane;
}

// LayerNorm of one token row spread over the 4 lanes {j, j+16, j+32, j+48}: lane g holds channels 32k + 8g .. + 7 of K-step k
// (natural order = the B-operand fragment).  Two-pass statistics as k_ln_fwd (pk_norm.hip).
template <int NK>
__device__ __forceinline__ void ln_row(const u32x4 (&xr)[NK], const float (&gam)[NK][8], const float (&bet)[NK][8], float inv_c, float eps,
                                       bf16x8 (&vf)[NK], float& mean, float& rstd) {
    float v[NK][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[k][2 * j] = blo(xr[k][j]);
            v[k][2 * j + 1] = bhi(xr[k][j]);
            s += v[k][2 * j] + v[k][2 * j + 1];
        }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    mean = s * inv_c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[k][j] -= mean;
            q += v[k][j] * v[k][j];
        }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    rstd = rsqrtf(q * inv_c + eps);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = pack_bf16x2(v[k][2 * j] * rstd * gam[k][2 * j] + bet[k][2 * j], v[k][2 * j + 1] * rstd * gam[k][2 * j + 1] + bet[k][2 * j + 1]);
        vf[k] = __builtin_bit_cast(bf16x8, o);
    }
}
template <int NK>
__device__ __forceinline__ void scale_rows(const u32x4 (&r)[NK], float sc, bf16x8 (&f)[NK]) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(blo(r[k][j]) * sc, bhi(r[k][j]) * sc);
        f[k] = __builtin_bit_cast(bf16x8, o);
    }
}

// ================================================================================================ forward
template <int C>
__global__ void __launch_bounds__(256) k_mlp_fwd(MlpArgs p) {
    constexpr int HD = 4 * C, NK = C / 32, NT = HD / 16, NS = HD / 32, NCT = C / 16, RT = 2;
    __shared__ __attribute__((aligned(16))) u32x4 sW1[NT * NK * 64];       // fragment (t, k): rows hid(t, i), channels 32k + 8g ..
    __shared__ __attribute__((aligned(16))) u32x4 sW2[NCT * NS * 64];      // fragment (ct, s): rows 16ct + i, hidden 32s + 8g ..
    __shared__ __attribute__((aligned(16))) float sB1[HD];
    const int tid = threadIdx.x, lane_ = tid & 63, wave = tid >> 6, i16 = lane_ & 15, g = lane_ >> 4;
    stage_frags(sW1, p.w1, NT * NK, C, [](int f, int i) { return hid(f / NK, i); }, [](int f) { return 32 * (f % NK); });
    stage_frags(sW2, p.w2, NCT * NS, HD, [](int f, int i) { return 16 * (f / NS) + i; }, [](int f) { return 32 * (f % NS); });
    for (int i = tid; i < HD; i += 256) sB1[i] = p.b1[i];
    float gam[NK][8], bet[NK][8];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            gam[k][j] = p.gamma[32 * k + 8 * g + j];
            bet[k][j] = p.beta[32 * k + 8 * g + j];
        }
    f32x4 b2v[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) b2v[ct] = *reinterpret_cast<const f32x4*>(p.b2 + 16 * ct + 4 * g);
    __syncthreads();
    const auto rx = MAKE_RSRC(p.x);
    const auto ro = MAKE_RSRC(p.out);
    const int ngroups = (p.M + 16 * RT - 1) / (16 * RT);
    const float inv_c = 1.f / (float)C;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gridDim.x * 4) {
        const int lane = opaque_lane<C != 32>(lane_);
        u32x4 xr[RT][NK];
        u32x2 xo[RT][NCT];
        unsigned rbase[RT];
        float sc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row = grp * 16 * RT + 16 * rt + i16;
            const bool ok = row < p.M;
            rbase[rt] = ok ? (unsigned)row * (unsigned)(C * 2) : OOB_OFF;
#pragma unroll
            for (int k = 0; k < NK; ++k) xr[rt][k] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? rbase[rt] + (32 * k + 8 * g) * 2 : OOB_OFF, 0, 0);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) xo[rt][ct] = __builtin_amdgcn_raw_buffer_load_b64(rx, ok ? rbase[rt] + (16 * ct + 4 * g) * 2 : OOB_OFF, 0, 0);
            sc[rt] = (p.scale && ok) ? p.scale[row / p.rows_per_sample] : 1.f;
        }
        bf16x8 vf[RT][NK];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float mean, rstd;
            ln_row<NK>(xr[rt], gam, bet, inv_c, p.eps, vf[rt], mean, rstd);
        }
        f32x4 y[RT][NCT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) y[rt][ct] = zero;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            f32x4 h[RT][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * s + u;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(&sB1[32 * s + 8 * g + 4 * u]);     // b1[hid(t, 4g + r)]
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) h[rt][u] = bias;
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const bf16x8 a = LDS_FRAG(sW1, t * NK + k, lane);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) h[rt][u] = MFMA(a, vf[rt][k], h[rt][u]);
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[rt][u][r] = gelu_erf(h[rt][u][r]);
            }
            bf16x8 hf[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) hf[rt] = pack2(h[rt][0], h[rt][1]);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const bf16x8 a = LDS_FRAG(sW2, ct * NS + s, lane);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) y[rt][ct] = MFMA(a, hf[rt], y[rt][ct]);
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f32x4 o = (y[rt][ct] + b2v[ct]) * sc[rt] + unpack4(xo[rt][ct]);
                __builtin_amdgcn_raw_buffer_store_b64(pack4(o), ro, rbase[rt] == OOB_OFF ? OOB_OFF : rbase[rt] + (16 * ct + 4 * g) * 2, 0, 0);
            }
    }
}

// ================================================================================================ backward: dx
// dx = dy + LayerNorm_bwd( W1^T ( gelu'(z) * (W2^T (s dy)) ) ),  z = W1 LN(x) + b1 recomputed.  part[block][2][C] receives this
// workgroup's sums of dv * xhat (dgamma) and dv (dbeta), dv = gradient w.r.t. the LayerNorm output.
template <int C>
__global__ void __launch_bounds__(256) k_mlp_bwd_dx(MlpArgs p) {
    constexpr int HD = 4 * C, NK = C / 32, NT = HD / 16, NS = HD / 32, NCT = C / 16, RT = 2;
    __shared__ __attribute__((aligned(16))) u32x4 sW1[NT * NK * 64];       // (t, k): W1 rows hid(t, i), channels 32k + 8g ..
    __shared__ __attribute__((aligned(16))) u32x4 sW2T[NT * NK * 64];      // (t, k): W2^T rows hid(t, i), channels 32k + 8g ..
    __shared__ __attribute__((aligned(16))) u32x4 sW1T[NCT * NS * 64];     // (ct, s): W1^T rows 16ct + i, hidden 32s + 8g ..
    __shared__ __attribute__((aligned(16))) float sB1[HD];
    __shared__ float sRed[4][2][C];
    const int tid = threadIdx.x, lane_ = tid & 63, wave = tid >> 6, i16 = lane_ & 15, g = lane_ >> 4;
    stage_frags(sW1, p.w1, NT * NK, C, [](int f, int i) { return hid(f / NK, i); }, [](int f) { return 32 * (f % NK); });
    stage_frags(sW2T, p.w2t, NT * NK, C, [](int f, int i) { return hid(f / NK, i); }, [](int f) { return 32 * (f % NK); });
    stage_frags(sW1T, p.w1t, NCT * NS, HD, [](int f, int i) { return 16 * (f / NS) + i; }, [](int f) { return 32 * (f % NS); });
    for (int i = tid; i < HD; i += 256) sB1[i] = p.b1[i];
    float gam[NK][8], bet[NK][8];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            gam[k][j] = p.gamma[32 * k + 8 * g + j];
            bet[k][j] = p.beta[32 * k + 8 * g + j];
        }
    f32x4 gamA[NCT], dgam[NCT], dbet[NCT];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        gamA[ct] = *reinterpret_cast<const f32x4*>(p.gamma + 16 * ct + 4 * g);
        dgam[ct] = dbet[ct] = zero;
    }
    __syncthreads();
    const auto rx = MAKE_RSRC(p.x);
    const auto rg = MAKE_RSRC(p.dy);
    const auto ro = MAKE_RSRC(p.out);
    const int ngroups = (p.M + 16 * RT - 1) / (16 * RT);
    const float inv_c = 1.f / (float)C;
    for (int grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gridDim.x * 4) {
        const int lane = opaque_lane<true>(lane_);
        u32x4 xr[RT][NK], dyr[RT][NK];
        u32x2 xo[RT][NCT], dyo[RT][NCT];
        unsigned rbase[RT];
        float sc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row = grp * 16 * RT + 16 * rt + i16;
            const bool ok = row < p.M;
            rbase[rt] = ok ? (unsigned)row * (unsigned)(C * 2) : OOB_OFF;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const unsigned off = ok ? rbase[rt] + (32 * k + 8 * g) * 2 : OOB_OFF;
                xr[rt][k] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
                dyr[rt][k] = __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0);
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const unsigned off = ok ? rbase[rt] + (16 * ct + 4 * g) * 2 : OOB_OFF;
                xo[rt][ct] = __builtin_amdgcn_raw_buffer_load_b64(rx, off, 0, 0);
                dyo[rt][ct] = __builtin_amdgcn_raw_buffer_load_b64(rg, off, 0, 0);
            }
            sc[rt] = (p.scale && ok) ? p.scale[row / p.rows_per_sample] : 1.f;
        }
        bf16x8 vf[RT][NK], gf[RT][NK];
        float mean[RT], rstd[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            ln_row<NK>(xr[rt], gam, bet, inv_c, p.eps, vf[rt], mean[rt], rstd[rt]);
            scale_rows<NK>(dyr[rt], sc[rt], gf[rt]);
        }
        f32x4 dv[RT][NCT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) dv[rt][ct] = zero;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            f32x4 dz[RT][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * s + u;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(&sB1[32 * s + 8 * g + 4 * u]);
                f32x4 z[RT], dh[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    z[rt] = bias;
                    dh[rt] = zero;
                }
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const bf16x8 a1 = LDS_FRAG(sW1, t * NK + k, lane), a2 = LDS_FRAG(sW2T, t * NK + k, lane);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        z[rt] = MFMA(a1, vf[rt][k], z[rt]);
                        dh[rt] = MFMA(a2, gf[rt][k], dh[rt]);
                    }
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dz[rt][u][r] = dh[rt][r] * gelu_grad(z[rt][r]);
            }
            bf16x8 dzf[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) dzf[rt] = pack2(dz[rt][0], dz[rt][1]);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const bf16x8 a = LDS_FRAG(sW1T, ct * NS + s, lane);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) dv[rt][ct] = MFMA(a, dzf[rt], dv[rt][ct]);
            }
        }
        // LayerNorm backward in accumulator layout: this lane holds channels 16ct + 4g + r of token (rt, i16)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f32x4 xh[NCT];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                xh[ct] = (unpack4(xo[rt][ct]) - mean[rt]) * rstd[rt];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gh = dv[rt][ct][r] * gamA[ct][r];
                    s1 += gh;
                    s2 += gh * xh[ct][r];
                }
                dgam[ct] += dv[rt][ct] * xh[ct];
                dbet[ct] += dv[rt][ct];
            }
            s1 += __shfl_xor(s1, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64);
            s2 += __shfl_xor(s2, 32, 64);
            const float m1 = s1 * inv_c, m2 = s2 * inv_c;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f32x4 o = (dv[rt][ct] * gamA[ct] - m1 - xh[ct] * m2) * rstd[rt] + unpack4(dyo[rt][ct]);
                __builtin_amdgcn_raw_buffer_store_b64(pack4(o), ro, rbase[rt] == OOB_OFF ? OOB_OFF : rbase[rt] + (16 * ct + 4 * g) * 2, 0, 0);
            }
        }
    }
    // dgamma / dbeta: tokens of this wave -> sum over the 16 lanes of a lane group, then over the 4 waves (fixed order)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = dgam[ct][r], b = dbet[ct][r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                a += __shfl_xor(a, o, 64);
                b += __shfl_xor(b, o, 64);
            }
            if (i16 == 0) {
                sRed[wave][0][16 * ct + 4 * g + r] = a;
                sRed[wave][1][16 * ct + 4 * g + r] = b;
            }
        }
    __syncthreads();
    if (tid < 2 * C) {
        const int which = tid / C, c = tid - which * C;
        p.part[(size_t)blockIdx.x * 2 * C + tid] = ((sRed[0][which][c] + sRed[1][which][c]) + sRed[2][which][c]) + sRed[3][which][c];
    }
}

// ================================================================================================ backward: dW1, db1, dW2, db2
// blockIdx.y selects the hidden slice [h0, h0 + HS); slab of workgroup (x, y) at part + (y * gridDim.x + x) * SLAB:
//   [ dW1 slice [HS][C] | dW2 slice [C][HS] | db1 slice [HS] | db2 [C] ]   (fp32; db2 is the same in every slice: use slice 0)
template <int C, int HS>
__global__ void __launch_bounds__(256) k_mlp_bwd_dw(MlpArgs p) {
    constexpr int NK = C / 32, NTS = HS / 16, NSP = HS / 32, NCT = C / 16, RT = 2;
    constexpr int PV = C + 8, PH = 40;                               // row pitches (bf16) of the wave-private tiles
    constexpr int TILE_HALFS = 2 * 32 * PV + 2 * 32 * PH;            // v, g2, h, dz tiles of one wave
    constexpr int SLAB = 2 * HS * C + HS + C;
    constexpr int SCRATCH_BYTES = (4 * TILE_HALFS * 2 > SLAB * 4) ? 4 * TILE_HALFS * 2 : SLAB * 4;
    __shared__ __attribute__((aligned(16))) u32x4 sW1[NTS * NK * 64];      // (t, k): W1 rows h0 + 16t + i
    __shared__ __attribute__((aligned(16))) u32x4 sW2T[NTS * NK * 64];     // (t, k): W2^T rows h0 + 16t + i
    __shared__ __attribute__((aligned(16))) float sB1[HS];
    __shared__ __attribute__((aligned(16))) unsigned char sScratch[SCRATCH_BYTES];
    const int tid = threadIdx.x, lane_ = tid & 63, wave = tid >> 6, i16 = lane_ & 15, g = lane_ >> 4;
    const int h0 = blockIdx.y * HS;
    stage_frags(sW1, p.w1 + (size_t)h0 * C, NTS * NK, C, [](int f, int i) { return 16 * (f / NK) + i; }, [](int f) { return 32 * (f % NK); });
    stage_frags(sW2T, p.w2t + (size_t)h0 * C, NTS * NK, C, [](int f, int i) { return 16 * (f / NK) + i; }, [](int f) { return 32 * (f % NK); });
    for (int i = tid; i < HS; i += 256) sB1[i] = p.b1[h0 + i];
    uint16_t* tV = reinterpret_cast<uint16_t*>(sScratch) + wave * TILE_HALFS;
    uint16_t* tG = tV + 32 * PV;
    uint16_t* tH = tG + 32 * PV;
    uint16_t* tD = tH + 32 * PH;
    float gam[NK][8], bet[NK][8];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            gam[k][j] = p.gamma[32 * k + 8 * g + j];
            bet[k][j] = p.beta[32 * k + 8 * g + j];
        }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 dW1[NTS][NCT], dW2[NCT][NTS];
    float db1[NTS], db2[NCT];
#pragma unroll
    for (int t = 0; t < NTS; ++t) {
        db1[t] = 0.f;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) dW1[t][ct] = dW2[ct][t] = zero;
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) db2[ct] = 0.f;
    __syncthreads();
    const auto rx = MAKE_RSRC(p.x);
    const auto rg = MAKE_RSRC(p.dy);
    const int ngroups = (p.M + 16 * RT - 1) / (16 * RT);
    const float inv_c = 1.f / (float)C;
    for (int grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gridDim.x * 4) {
        const int lane = opaque_lane<true>(lane_);
        u32x4 xr[RT][NK], dyr[RT][NK];
        float sc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row = grp * 16 * RT + 16 * rt + i16;
            const bool ok = row < p.M;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const unsigned off = ok ? (unsigned)row * (unsigned)(C * 2) + (32 * k + 8 * g) * 2 : OOB_OFF;
                xr[rt][k] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
                dyr[rt][k] = __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0);
            }
            sc[rt] = (p.scale && ok) ? p.scale[row / p.rows_per_sample] : 1.f;
        }
        bf16x8 vf[RT][NK], gf[RT][NK];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float mean, rstd;
            ln_row<NK>(xr[rt], gam, bet, inv_c, p.eps, vf[rt], mean, rstd);
            scale_rows<NK>(dyr[rt], sc[rt], gf[rt]);
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                *reinterpret_cast<bf16x8*>(tV + (16 * rt + i16) * PV + 32 * k + 8 * g) = vf[rt][k];
                *reinterpret_cast<bf16x8*>(tG + (16 * rt + i16) * PV + 32 * k + 8 * g) = gf[rt][k];
            }
        }
        LDS_FENCE();
        bf16x8 vT[NCT], gT[NCT];          // lane = channel 16ct + i16, slots = the 32 tokens in accumulator order
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            vT[ct] = tr_frag32(tV, PV, 16 * ct, lane);
            gT[ct] = tr_frag32(tG, PV, 16 * ct, lane);
            db2[ct] += sum8(gT[ct]);
        }
#pragma unroll
        for (int sp = 0; sp < NSP; ++sp) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * sp + u;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(&sB1[16 * t + 4 * g]);
                f32x4 z[RT], dh[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    z[rt] = bias;
                    dh[rt] = zero;
                }
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const bf16x8 a1 = LDS_FRAG(sW1, t * NK + k, lane), a2 = LDS_FRAG(sW2T, t * NK + k, lane);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        z[rt] = MFMA(a1, vf[rt][k], z[rt]);
                        dh[rt] = MFMA(a2, gf[rt][k], dh[rt]);
                    }
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    f32x4 hh, dz;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float gv, gd;                       // gelu and gelu' share the exponential and the erf polynomial
                        gelu_both(z[rt][r], gv, gd);
                        hh[r] = gv;
                        dz[r] = dh[rt][r] * gd;
                    }
                    *reinterpret_cast<u32x2*>(tH + (16 * rt + i16) * PH + 16 * u + 4 * g) = pack4(hh);
                    *reinterpret_cast<u32x2*>(tD + (16 * rt + i16) * PH + 16 * u + 4 * g) = pack4(dz);
                }
            }
            LDS_FENCE();
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * sp + u;
                const bf16x8 hT = tr_frag32(tH, PH, 16 * u, lane), dT = tr_frag32(tD, PH, 16 * u, lane);
                db1[t] += sum8(dT);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    dW2[ct][t] = MFMA(gT[ct], hT, dW2[ct][t]);      // D[c][hd] = sum_m g2[m][c] h[m][hd]
                    dW1[t][ct] = MFMA(dT, vT[ct], dW1[t][ct]);      // D[hd][c] = sum_m dz[m][hd] v[m][c]
                }
            }
            LDS_FENCE();                                             // the reads above complete before the tiles are rewritten
        }
    }
    // ---- the four waves' accumulators are summed in LDS in wave order (fixed order), then written as ONE slab per workgroup
    float* slab = reinterpret_cast<float*>(sScratch);
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < NTS; ++t) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* a = slab + (16 * t + 4 * g + r) * C + 16 * ct + i16;                 // dW1[hd][c]: lane = c, rows = hd
                        float* b = slab + HS * C + (16 * ct + 4 * g + r) * HS + 16 * t + i16;       // dW2[c][hd]: lane = hd, rows = c
                        *a = (w ? *a : 0.f) + dW1[t][ct][r];
                        *b = (w ? *b : 0.f) + dW2[ct][t][r];
                    }
                float v = db1[t];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (g == 0) {
                    float* a = slab + 2 * HS * C + 16 * t + i16;
                    *a = (w ? *a : 0.f) + v;
                }
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                float v = db2[ct];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (g == 0) {
                    float* a = slab + 2 * HS * C + HS + 16 * ct + i16;
                    *a = (w ? *a : 0.f) + v;
                }
            }
        }
    }
    __syncthreads();
    float* dst = p.part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SLAB;
    for (int i = tid * 4; i < SLAB; i += 1024) *reinterpret_cast<f32x4*>(dst + i) = *reinterpret_cast<const f32x4*>(slab + i);
}

// ================================================================================================ C-ABI
static inline int mlp_hidden_slice(int C) {          // hidden units per blockIdx.y slice of the weight-gradient kernel
    static const int env = getenv("PK_MLP_HS") ? atoi(getenv("PK_MLP_HS")) : 0;
    if (env == 32 || env == 64 || (env == 128 && C == 32)) return env;
    // Measured (B = 64): the 128 accumulator registers of HS * C = 4096 hold the kernel at one wave per SIMD (C = 32: 66 us);
    // half of that (two waves per SIMD) runs the same work in 40 us although every slice recomputes LayerNorm.
    return C == 32 ? 64 : 32;
}
static inline int mlp_row_groups(int M) { return (M + 31) / 32; }
extern "C" int pk_ln_mlp_supported(int C) { return C == 32 || C == 64; }
extern "C" int pk_ln_mlp_hidden_slice(int C) { return mlp_hidden_slice(C); }
extern "C" int pk_ln_mlp_slab_floats(int C) {
    const int hs = mlp_hidden_slice(C);
    return 2 * hs * C + hs + C;
}
// workgroups (4 waves, one 32-token group per wave and iteration): enough to fill 256 CUs twice, never more than the work
static inline int mlp_blocks(int M, int target) {
    static const int env_target = getenv("PK_MLP_WGS") ? atoi(getenv("PK_MLP_WGS")) : 0;
    if (env_target > 0) target = env_target;
    const int need = (mlp_row_groups(M) + 3) / 4;
    return need < target ? (need < 1 ? 1 : need) : target;
}
extern "C" int pk_ln_mlp_dx_blocks(int M, int C) { return mlp_blocks(M, 512); }
// (fewer, longer-lived workgroups: each one stages its weight slice and ends with a 4-phase slab reduction; 256 x slices
// workgroups beat 512 and 1024 at every slice width)
// workgroups beat 512 and 1024 at every slice width; C = 64 with its 8 slices: 64 x 8 = 38 us, 128 x 8 = 46 us, 256 x 8 = 67 us)
extern "C" int pk_ln_mlp_dw_blocks(int M, int C) { return mlp_blocks(M, C == 32 ? 256 : 64); }

static int mlp_check(const char* who, const MlpArgs& a, int C) {
    PK_SUPPORTED(C == 32 || C == 64, "%s: C=%d (the fused MLP half is built for C = 32 / 64)", who, C);
    PK_REQUIRE(a.x && a.gamma && a.beta && a.b1 && a.M > 0, "%s: null pointer / bad size", who);
    PK_REQUIRE(!a.scale || a.rows_per_sample > 0, "%s: scale needs rows_per_sample", who);
    PK_REQUIRE((int64_t)a.M * C < 0x3fffffffLL, "%s: tensor too large for 32-bit byte offsets", who);
    PK_REQUIRE(((((uintptr_t)a.x) | ((uintptr_t)a.dy) | ((uintptr_t)a.out) | ((uintptr_t)a.w1) | ((uintptr_t)a.w2) | ((uintptr_t)a.w1t) |
                 ((uintptr_t)a.w2t) | ((uintptr_t)a.gamma) | ((uintptr_t)a.b2) | ((uintptr_t)a.part)) & 15) == 0, "%s: 16-byte alignment", who);
    return PK_OK;
}

extern "C" int pk_ln_mlp_fwd(const void* x, const float* gamma, const float* beta, const void* w1, const float* b1, const void* w2,
                             const float* b2, const float* row_scale, void* y, int M, int C, int rows_per_sample, float eps, void* stream) {
    MlpArgs a{};
    a.x = (const uint16_t*)x; a.out = (uint16_t*)y; a.gamma = gamma; a.beta = beta; a.b1 = b1; a.b2 = b2; a.scale = row_scale;
    a.w1 = (const uint16_t*)w1; a.w2 = (const uint16_t*)w2; a.M = M; a.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1; a.eps = eps;
    int rc = mlp_check("pk_ln_mlp_fwd", a, C);
    if (rc) return rc;
    PK_REQUIRE(w1 && w2 && b2 && y, "pk_ln_mlp_fwd: null pointer");
    const dim3 grid(mlp_blocks(M, 512)), block(256);
    if (C == 32) hipLaunchKernelGGL(k_mlp_fwd<32>, grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_mlp_fwd<64>, grid, block, 0, (hipStream_t)stream, a);
    return pk_launch_status("pk_ln_mlp_fwd");
}

extern "C" int pk_ln_mlp_bwd_dx(const void* dy, const void* x, const float* gamma, const float* beta, const void* w1, const float* b1,
                                const void* w1_t, const void* w2_t, const float* row_scale, void* dx, float* ln_partial, int M, int C,
                                int rows_per_sample, float eps, void* stream) {
    MlpArgs a{};
    a.x = (const uint16_t*)x; a.dy = (const uint16_t*)dy; a.out = (uint16_t*)dx; a.gamma = gamma; a.beta = beta; a.b1 = b1; a.scale = row_scale;
    a.w1 = (const uint16_t*)w1; a.w1t = (const uint16_t*)w1_t; a.w2t = (const uint16_t*)w2_t; a.part = ln_partial;
    a.M = M; a.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1; a.eps = eps;
    int rc = mlp_check("pk_ln_mlp_bwd_dx", a, C);
    if (rc) return rc;
    PK_REQUIRE(dy && w1 && w1_t && w2_t && dx && ln_partial, "pk_ln_mlp_bwd_dx: null pointer");
    const dim3 grid(pk_ln_mlp_dx_blocks(M, C)), block(256);
    if (C == 32) hipLaunchKernelGGL(k_mlp_bwd_dx<32>, grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_mlp_bwd_dx<64>, grid, block, 0, (hipStream_t)stream, a);
    return pk_launch_status("pk_ln_mlp_bwd_dx");
}

extern "C" int pk_ln_mlp_bwd_dw(const void* dy, const void* x, const float* gamma, const float* beta, const void* w1, const float* b1,
                                const void* w2_t, const float* row_scale, float* slabs, int M, int C, int rows_per_sample, float eps,
                                void* stream) {
    MlpArgs a{};
    a.x = (const uint16_t*)x; a.dy = (const uint16_t*)dy; a.gamma = gamma; a.beta = beta; a.b1 = b1; a.scale = row_scale;
    a.w1 = (const uint16_t*)w1; a.w2t = (const uint16_t*)w2_t; a.part = slabs;
    a.M = M; a.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1; a.eps = eps;
    int rc = mlp_check("pk_ln_mlp_bwd_dw", a, C);
    if (rc) return rc;
    PK_REQUIRE(dy && w1 && w2_t && slabs, "pk_ln_mlp_bwd_dw: null pointer");
    const int hs = mlp_hidden_slice(C);
    const dim3 grid(pk_ln_mlp_dw_blocks(M, C), (4 * C) / hs), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (C == 32 && hs == 128) hipLaunchKernelGGL((k_mlp_bwd_dw<32, 128>), grid, block, 0, st, a);
    else if (C == 32 && hs == 64) hipLaunchKernelGGL((k_mlp_bwd_dw<32, 64>), grid, block, 0, st, a);
    else if (C == 32) hipLaunchKernelGGL((k_mlp_bwd_dw<32, 32>), grid, block, 0, st, a);
    else if (hs == 64) hipLaunchKernelGGL((k_mlp_bwd_dw<64, 64>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_mlp_bwd_dw<64, 32>), grid, block, 0, st, a);
    return pk_launch_status("pk_ln_mlp_bwd_dw");
}
