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
#define LOG2E_F 1.44269504088896340736f
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0)
#define LO4(v) __builtin_shufflevector(v, v, 0, 1, 2, 3)        // the two 16-deep operands inside a 32-deep one
#define HI4(v) __builtin_shufflevector(v, v, 4, 5, 6, 7)
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
    s = xor16_sum(s);
    s = xor32_sum(s);
    mean = s * inv_c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[k][j] -= mean;
            q += v[k][j] * v[k][j];
        }
    q = xor16_sum(q);
    q = xor32_sum(q);
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
                for (int rt = 0; rt < RT; ++rt) h[rt][u] = gelu_erf(h[rt][u]);
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

// ================================================================================================ forward, wide channels
// C = 80 ... 320 (branches 2-3 of HRFormer-small, the 8-aligned twin of HRFormer-base): W1 and W2 (8 C^2 bytes each) do not fit the
// LDS any more, so the hidden dimension is walked in slices of 32 units and the 2 NK + NCT weight fragments of a slice (W1 rows of
// the slice, W2 columns of the slice) are STREAMED: global (L2-resident) -> registers -> a two-slot LDS ring in fragment order, the
// loads of slice s + 2 in flight while slice s is multiplied, one workgroup barrier per slice.  A workgroup is WAVES x 32 tokens
// (every wave reads every fragment: 8 waves = 256 tokens per 16 C^2 bytes of L2 traffic); x is read once for the LayerNorm / fc1
// operand and once more (L2 hit) for the residual, so the accumulators of fc2 (2 x NCT tiles) and the LayerNorm output (2 x NK
// fragments) are all a wave has to hold.  Channels beyond C inside the last 32-wide K step (C = 80: NK = 3) are zero operands: the
// loads of x, gamma, beta and of the weight columns are out of range there.  `c_real` < C (padded twins, models/padded.py): the
// LayerNorm statistics run over the real channels -- the padded ones hold exact zeros, so the sums only need the divisor, and the
// squared-deviation sum is corrected by (32 NK - c_real) mean^2.
#ifndef PK_MLP_WIDE_GELU_POLY
#define PK_MLP_WIDE_GELU_POLY 1
#endif
// RESIDENT (C = 80 with hidden 320: 110 KB of fragments): every slice is staged ONCE per workgroup, the workgroups are persistent over
// 32-token groups and the waves run through the slices on their own -- no barrier after the prologue.
template <int NK, int NCT, int WAVES, bool RESIDENT>
__global__ void __launch_bounds__(64 * WAVES) k_mlp_fwd_w(MlpArgs p, int C, int c_real, int HD) {
    constexpr int RT = 2, NF = 2 * NK + NCT, NLD = (NF + WAVES - 1) / WAVES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    const int NS = HD >> 5, n_slots = RESIDENT ? NS : 2;
    u32x4* ring = reinterpret_cast<u32x4*>(smem_w);                      // [n_slots][NF][64] fragments
    float* sB1 = reinterpret_cast<float*>(smem_w + n_slots * NF * 1024); // [HD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const auto rw1 = MAKE_RSRC(p.w1), rw2 = MAKE_RSRC(p.w2);
    // staging: fragment f = wave + WAVES j of a slice is fetched by this wave (lane l = the fragment's lane)
    unsigned soff[NLD], sstep[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int f = wave + WAVES * j;
        if (f < 2 * NK) {                    // W1 rows hid(2s + u, i), columns 32k + 8g ..
            const int u = f / NK, k = f - u * NK, col = 32 * k + 8 * g;
            soff[j] = col < C ? (unsigned)((hid(u, i16) * C + col) * 2) : OOB_OFF;
            sstep[j] = 64u * (unsigned)C;
        } else if (f < NF) {                 // W2 rows 16ct + i, columns 32s + 8g ..
            const int ct = f - 2 * NK;
            soff[j] = (unsigned)(((16 * ct + i16) * HD + 8 * g) * 2);
            sstep[j] = 64u;
        } else {
            soff[j] = OOB_OFF;
            sstep[j] = 0u;
        }
    }
    u32x4 st[NLD];
    auto load_slice = [&](int s) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int f = wave + WAVES * j;
            st[j] = f < 2 * NK ? __builtin_amdgcn_raw_buffer_load_b128(rw1, soff[j] + sstep[j] * (unsigned)s, 0, 0)
                               : __builtin_amdgcn_raw_buffer_load_b128(rw2, soff[j] + sstep[j] * (unsigned)s, 0, 0);
        }
    };
    auto write_slice = [&](int slot) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int f = wave + WAVES * j;
            if (f < NF) ring[(slot * NF + f) * 64 + lane] = st[j];
        }
    };
    load_slice(0);
    for (int i = tid; i < HD; i += 64 * WAVES) sB1[i] = p.b1[i];
    if (RESIDENT) {
        for (int s = 0; s < NS; ++s) {
            write_slice(s);
            if (s + 1 < NS) load_slice(s + 1);
        }
        __syncthreads();
    }
    const auto rx = MAKE_RSRC(p.x);
    const auto ro = MAKE_RSRC(p.out);
    const auto rgam = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gamma), 0, C * 4, 0x00020000);
    const auto rbet = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.beta), 0, C * 4, 0x00020000);
    const auto rb2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b2), 0, C * 4, 0x00020000);
    const float inv_c = 1.f / (float)c_real, n_pad = (float)(32 * NK - c_real);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const int ngroups = (p.M + 16 * RT - 1) / (16 * RT);
    // (streamed: exactly one group per wave, grid = ceil(groups / WAVES), so every wave reaches every barrier; a group beyond the end is
    //  all out-of-range rows: zero loads, dropped stores)
    auto process = [&](const int grp) {
        // ---- this wave's 32 tokens: LayerNorm -> B-operand fragments
        unsigned rbase[RT];
        float sc[RT];
        bf16x8 vf[RT][NK];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row = grp * 16 * RT + 16 * rt + i16;
            const bool ok = row < p.M;
            rbase[rt] = ok ? (unsigned)row * (unsigned)(C * 2) : OOB_OFF;
            sc[rt] = (p.scale && ok) ? p.scale[row / p.rows_per_sample] : 1.f;
            u32x4 xr[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) xr[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, (ok && 32 * k + 8 * g < C) ? rbase[rt] + (32 * k + 8 * g) * 2 : OOB_OFF, 0, 0);
            float v[NK][8];
            float sm = 0.f;
#pragma unroll
            for (int k = 0; k < NK; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[k][2 * j] = blo(xr[k][j]);
                    v[k][2 * j + 1] = bhi(xr[k][j]);
                    sm += v[k][2 * j] + v[k][2 * j + 1];
                }
            sm = xor16_sum(sm);
            sm = xor32_sum(sm);
            const float mean = sm * inv_c;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < NK; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[k][j] -= mean;
                    q += v[k][j] * v[k][j];
                }
            q = xor16_sum(q);
            q = xor32_sum(q);
            const float rstd = rsqrtf(fmaxf(q - n_pad * mean * mean, 0.f) * inv_c + p.eps);
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const f32x4 g0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rgam, (32 * k + 8 * g) * 4, 0, 0));
                const f32x4 g1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rgam, (32 * k + 8 * g + 4) * 4, 0, 0));
                const f32x4 b0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbet, (32 * k + 8 * g) * 4, 0, 0));
                const f32x4 b1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbet, (32 * k + 8 * g + 4) * 4, 0, 0));
                u32x4 o;
                o[0] = pack_bf16x2(v[k][0] * rstd * g0[0] + b0[0], v[k][1] * rstd * g0[1] + b0[1]);
                o[1] = pack_bf16x2(v[k][2] * rstd * g0[2] + b0[2], v[k][3] * rstd * g0[3] + b0[3]);
                o[2] = pack_bf16x2(v[k][4] * rstd * g1[0] + b1[0], v[k][5] * rstd * g1[1] + b1[1]);
                o[3] = pack_bf16x2(v[k][6] * rstd * g1[2] + b1[2], v[k][7] * rstd * g1[3] + b1[3]);
                vf[rt][k] = __builtin_bit_cast(bf16x8, o);
            }
        }
        if (!RESIDENT) {
            write_slice(0);
            if (NS > 1) load_slice(1);
        }
        f32x4 y[RT][NCT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) y[rt][ct] = zero;
#pragma unroll 1
        for (int s = 0; s < NS; ++s) {
            if (!RESIDENT) {
                __syncthreads();          // slice s is visible to everyone, everyone is done with the other slot (slice s - 1)
                if (s + 1 < NS) write_slice((s + 1) & 1);
                if (s + 2 < NS) load_slice(s + 2);
            }
            const u32x4* fr = ring + (RESIDENT ? s : (s & 1)) * NF * 64;
            f32x4 h[RT][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x4 bias = *reinterpret_cast<const f32x4*>(&sB1[32 * s + 8 * g + 4 * u]);     // b1[hid(2s + u, 4g + r)]
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) h[rt][u] = bias;
            }
#pragma unroll
            for (int k = 0; k < NK; ++k)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bf16x8 a = LDS_FRAG(fr, u * NK + k, lane);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) h[rt][u] = MFMA(a, vf[rt][k], h[rt][u]);
                }
            bf16x8 hf[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#if PK_MLP_WIDE_GELU_POLY
                hf[rt] = pack2(gelu_erf_poly(h[rt][0]), gelu_erf_poly(h[rt][1]));          // packed-fp32 polynomial form (pk_common.h)
#else
                hf[rt] = pack2(gelu_erf(h[rt][0]), gelu_erf(h[rt][1]));
#endif
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const bf16x8 a = LDS_FRAG(fr, 2 * NK + ct, lane);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) y[rt][ct] = MFMA(a, hf[rt], y[rt][ct]);
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const unsigned off = rbase[rt] == OOB_OFF ? OOB_OFF : rbase[rt] + (16 * ct + 4 * g) * 2;
                const u32x2 xo = __builtin_amdgcn_raw_buffer_load_b64(rx, off, 0, 0);
                const f32x4 b2v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb2, (16 * ct + 4 * g) * 4, 0, 0));
                const f32x4 o = (y[rt][ct] + b2v) * sc[rt] + unpack4(xo);
                __builtin_amdgcn_raw_buffer_store_b64(pack4(o), ro, off, 0, 0);
            }
    };
    if (RESIDENT) {
        for (int grp = blockIdx.x * WAVES + wave; grp < ngroups; grp += gridDim.x * WAVES) process(grp);
    } else {
        process(blockIdx.x * WAVES + wave);
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
                for (int rt = 0; rt < RT; ++rt) dz[rt][u] = dh[rt] * gelu_grad(z[rt]);
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
            s1 = xor16_sum(s1);
            s1 = xor32_sum(s1);
            s2 = xor16_sum(s2);
            s2 = xor32_sum(s2);
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
            a = lanes_sum<16>(a);
            b = lanes_sum<16>(b);
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
                    f32x4 hh, dz;                           // gelu and gelu' share the erf polynomial
                    gelu_both(z[rt], hh, dz);
                    dz *= dh[rt];
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
                v = xor16_sum(v);
                v = xor32_sum(v);
                if (g == 0) {
                    float* a = slab + 2 * HS * C + 16 * t + i16;
                    *a = (w ? *a : 0.f) + v;
                }
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                float v = db2[ct];
                v = xor16_sum(v);
                v = xor32_sum(v);
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

// ================================================================================================ attention half: forward
//   y = x + s * proj( window_attention( qkv( LayerNorm1(x) ) ) )         (hrformer.py:262-286, WindowAttention :174-200)
// One wave per 7x7 window (all heads, head_dim 32), tokens padded 49 -> 64 for the MFMA tiles.  The reference zero-pads the map
// to a multiple of 7 AFTER LayerNorm and attends the pad tokens without a mask: their LN output is forced to 0 here (q = b_q,
// k = b_k, v = b_v), only the tile padding 49..63 is masked.  Everything is chained through accumulators:
//   * Q^T, K^T tiles D[e][token] = W rows (A) x u (B): packed, they are the B operand (queries) / A operand (keys) of the score
//     MFMA S^T = K Q^T with the head dimension enumerated in accumulator order on both sides;
//   * V tiles D[token][e] = u (A) x W_v rows (B): packed over token tiles they are the A operand V^T of O^T = V^T P^T with the
//     keys in accumulator order -- the order the register-resident probabilities P^T come in (see pk_attn.hip);
//   * the W_v rows are gathered with the same hid() permutation as in the MLP, so the packed O^T tiles are the B operand of
//     the projection with the channels in natural order.
// Training mode also writes the attention output o (window order, [windows*49][C]) and the log-sum-exp for the backward pass.
struct AttnArgs {
    const uint16_t* x;        // [M][C] pixel rows
    const uint16_t* dy;       // backward
    uint16_t* out;            // forward: y; backward: dx
    const int32_t* rowmap;    // [windows*49] pixel row of each window token, -1 = zero-pad token
    const float *gamma, *beta, *table, *bqkv, *bproj, *scale;
    const uint16_t *wqkv, *wproj;          // forward copies [3C][C], [C][C]
    const uint16_t *wqkv_t, *wproj_t;      // data-gradient copies [C][3C], [C][C]
    uint16_t* o_save;         // [windows*49][C] attention output before the projection, or null
    float* lse;               // [windows][heads][49], or null
    uint16_t* dqkv;           // backward: [windows*49][3C]
    uint16_t* u_save;         // backward: LayerNorm output in window order [windows*49][C] (zero rows for the pad tokens)
    float* part;              // backward slabs
    int n_windows, windows_per_sample;
    float eps, softmax_scale;
};
#define AT_N 49
__device__ __forceinline__ int rel_a7(int t) { return 13 * (t / 7) + t % 7; }   // rel_index(i, j) = rel_a7(i) - rel_a7(j) + 84

template <int C>
__global__ void __launch_bounds__(256) k_attn_fwd(AttnArgs p) {
    constexpr int HEADS = C / 32, NK = C / 32, NCT = C / 16;
    __shared__ __attribute__((aligned(16))) u32x4 sWqkv[HEADS * 6 * NK * 64];   // fragment ((h*3 + part)*2 + et)*NK + k
    __shared__ __attribute__((aligned(16))) u32x4 sWp[NCT * HEADS * 64];        // fragment nt*HEADS + h: rows 16nt + i, channels 32h + 8g ..
    __shared__ __attribute__((aligned(16))) float sBqkv[3 * C];
    __shared__ float sBias[HEADS][176];
    const int tid = threadIdx.x, lane_ = tid & 63, wave = tid >> 6, g_ = lane_ >> 4;
    stage_frags(sWqkv, p.wqkv, HEADS * 6 * NK, C,
                [](int f, int i) {
                    const int k_ = f % NK, r = f / NK, et = r & 1, part = (r >> 1) % 3, h = r / 6;
                    (void)k_;
                    return part * C + 32 * h + (part == 2 ? hid(et, i) : 16 * et + i);
                },
                [](int f) { return 32 * (f % NK); });
    stage_frags(sWp, p.wproj, NCT * HEADS, C, [](int f, int i) { return 16 * (f / HEADS) + i; }, [](int f) { return 32 * (f % HEADS); });
    for (int i = tid; i < 3 * C; i += 256) sBqkv[i] = p.bqkv[i];
    for (int i = tid; i < HEADS * 169; i += 256) sBias[i / 169][i % 169] = p.table[(i % 169) * HEADS + i / 169];
    float gam[NK][8], bet[NK][8];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            gam[k][j] = p.gamma[32 * k + 8 * g_ + j];
            bet[k][j] = p.beta[32 * k + 8 * g_ + j];
        }
    f32x4 bpv[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) bpv[ct] = *reinterpret_cast<const f32x4*>(p.bproj + 16 * ct + 4 * g_);
    int aj[4][4];                       // 84 - A(j) for this lane's 16 keys j = 16cj + 4g + r (-1: tile padding)
#pragma unroll
    for (int cj = 0; cj < 4; ++cj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * cj + 4 * g_ + r;
            aj[cj][r] = j < AT_N ? 84 - rel_a7(j) : -1;
        }
    __syncthreads();
    const auto rx = MAKE_RSRC(p.x);
    const auto ro = MAKE_RSRC(p.out);
    const float inv_c = 1.f / (float)C;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int w = blockIdx.x * 4 + wave; w < p.n_windows; w += gridDim.x * 4) {
        // (opaque lane id: keeps the weight-fragment reads AND the 64 bias-table lookups per head inside the window loop)
        const int lane = opaque_lane<true>(lane_), i16 = lane & 15, g = lane >> 4;
        int row[4];
        u32x4 xr[4][NK];
        u32x2 xo[4][NCT];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = 16 * t + i16;
            row[t] = n < AT_N ? p.rowmap[w * AT_N + n] : -1;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned base = row[t] >= 0 ? (unsigned)row[t] * (unsigned)(C * 2) : OOB_OFF;
#pragma unroll
            for (int k = 0; k < NK; ++k) xr[t][k] = __builtin_amdgcn_raw_buffer_load_b128(rx, row[t] >= 0 ? base + (32 * k + 8 * g) * 2 : OOB_OFF, 0, 0);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) xo[t][ct] = __builtin_amdgcn_raw_buffer_load_b64(rx, row[t] >= 0 ? base + (16 * ct + 4 * g) * 2 : OOB_OFF, 0, 0);
        }
        const float sc = p.scale ? p.scale[w / p.windows_per_sample] : 1.f;
        bf16x8 uf[4][NK];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float mean, rstd;
            ln_row<NK>(xr[t], gam, bet, inv_c, p.eps, uf[t], mean, rstd);
            if (row[t] < 0) {          // zero-pad token (or tile padding): the reference pads AFTER LayerNorm
#pragma unroll
                for (int k = 0; k < NK; ++k) uf[t][k] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        f32x4 accY[4][NCT];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) accY[t][ct] = zero;
#pragma unroll 1
        for (int h = 0; h < HEADS; ++h) {
            bf16x8 qf[4], kf[4], vt[2][2];
            {
                f32x4 va[4][2];
#pragma unroll
                for (int ce = 0; ce < 2; ++ce) {
                    const float bv = sBqkv[2 * C + 32 * h + hid(ce, i16)];
#pragma unroll
                    for (int t = 0; t < 4; ++t) va[t][ce] = (f32x4){bv, bv, bv, bv};
#pragma unroll
                    for (int k = 0; k < NK; ++k) {
                        const bf16x8 wv = LDS_FRAG(sWqkv, ((h * 3 + 2) * 2 + ce) * NK + k, lane);
#pragma unroll
                        for (int t = 0; t < 4; ++t) va[t][ce] = MFMA(uf[t][k], wv, va[t][ce]);       // D[token][e]
                    }
                }
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int ce = 0; ce < 2; ++ce) vt[s][ce] = pack2(va[2 * s][ce], va[2 * s + 1][ce]);
            }
            {
                f32x4 qa[4][2], ka[4][2];
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(&sBqkv[32 * h + 16 * et + 4 * g]);
                    const f32x4 bk = *reinterpret_cast<const f32x4*>(&sBqkv[C + 32 * h + 16 * et + 4 * g]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        qa[t][et] = bq;
                        ka[t][et] = bk;
                    }
#pragma unroll
                    for (int k = 0; k < NK; ++k) {
                        const bf16x8 wq = LDS_FRAG(sWqkv, ((h * 3 + 0) * 2 + et) * NK + k, lane);
                        const bf16x8 wk = LDS_FRAG(sWqkv, ((h * 3 + 1) * 2 + et) * NK + k, lane);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            qa[t][et] = MFMA(wq, uf[t][k], qa[t][et]);                            // D[e][token]
                            ka[t][et] = MFMA(wk, uf[t][k], ka[t][et]);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    qf[t] = pack2(qa[t][0], qa[t][1]);
                    kf[t] = pack2(ka[t][0], ka[t][1]);
                }
            }
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                const int i = 16 * ci + i16;
                const int ai = rel_a7(i < AT_N ? i : 0);
                f32x4 st[4];
#pragma unroll
                for (int cj = 0; cj < 4; ++cj) st[cj] = MFMA(kf[cj], qf[ci], zero);
                float mx = -INFINITY;
#pragma unroll
                for (int cj = 0; cj < 4; ++cj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float bv = sBias[h][ai + (aj[cj][r] >= 0 ? aj[cj][r] : 0)];
                        st[cj][r] = aj[cj][r] >= 0 ? st[cj][r] * p.softmax_scale + bv : -INFINITY;
                        mx = fmaxf(mx, st[cj][r]);
                    }
                mx = xor16_max(mx);
                mx = xor32_max(mx);
                float sum = 0.f;
#pragma unroll
                for (int cj = 0; cj < 4; ++cj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        st[cj][r] = __expf(st[cj][r] - mx);
                        sum += st[cj][r];
                    }
                sum = xor16_sum(sum);
                sum = xor32_sum(sum);
                const float inv = 1.f / sum;
                if (g == 0 && i < AT_N && p.lse) p.lse[((size_t)w * HEADS + h) * AT_N + i] = mx + __logf(sum);
                const bf16x8 p0 = pack2(st[0] * inv, st[1] * inv), p1 = pack2(st[2] * inv, st[3] * inv);
                f32x4 o[2];
#pragma unroll
                for (int ce = 0; ce < 2; ++ce) {
                    o[ce] = MFMA(vt[0][ce], p0, zero);
                    o[ce] = MFMA(vt[1][ce], p1, o[ce]);         // rows 4g + r of tile ce  <->  channel 32h + 8g + 4ce + r
                    if (p.o_save && i < AT_N)
                        *reinterpret_cast<u32x2*>(p.o_save + ((size_t)w * AT_N + i) * C + 32 * h + 8 * g + 4 * ce) = pack4(o[ce]);
                }
                const bf16x8 of = pack2(o[0], o[1]);
#pragma unroll
                for (int nt = 0; nt < NCT; ++nt) accY[ci][nt] = MFMA(LDS_FRAG(sWp, nt * HEADS + h, lane), of, accY[ci][nt]);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f32x4 v = (accY[t][ct] + bpv[ct]) * sc + unpack4(xo[t][ct]);
                __builtin_amdgcn_raw_buffer_store_b64(pack4(v), ro, row[t] >= 0 ? (unsigned)row[t] * (unsigned)(C * 2) + (16 * ct + 4 * g) * 2 : OOB_OFF, 0, 0);
            }
    }
}

// ================================================================================================ attention half, head_dim 40: forward
// The 8-aligned twin of HRFormer-base (models/padded.py: C = heads x 40, head_dim 39 zero-padded to 40; hrformer.py:779-825) at its
// high-resolution widths.  Same chain as k_attn_fwd (one wave per 7x7 window, everything through MFMA accumulators), with the head
// dimension on THREE 16-row tiles (40 real + 8 zero rows) and two 32-deep contraction steps over it:
//   * q / k tiles et = 0..2: rows 16et + i of the head (rows >= 40: zero weights, zero bias); score operands = pack2(tile 0, tile 1) and
//     pack2(tile 2, 0);
//   * v tiles ce = 0, 1 gathered with hid() as before; tile 2 = head channels 32 + i, i < 8.  The packed output pack2(o[2], 0) then has
//     slot jj < 4 of lane group g = head channel 32 + 4g + jj (g < 2): the second projection fragment holds exactly those four W_proj
//     columns per lane (an 8-byte chunk, zero elsewhere);
//   * K-steps over the input channels: NK = ceil(C / 32), columns >= C are out-of-range (zero) loads, as in k_mlp_fwd_w.
// LayerNorm statistics over `c_real` channels (the padded channels of the twin hold zeros), softmax scale = p.softmax_scale (39^-0.5).
// Forward only (inference): neither the attention output nor the log-sum-exp is saved.  All heads' weight fragments are resident in LDS
// (C = 80: 74 KB), filled once per workgroup.
template <int NK, int NCT, int HEADS, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_attn_fwd_w(AttnArgs p, int C, int c_real) {
    constexpr int NT_ = 64 * WAVES;
    constexpr int ET = 3, HDP = 40;
    constexpr int F_QK = 2 * ET * NK, F_V = ET * NK, F_P = NCT, F_HEAD = F_QK + F_V + F_P;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_a[];
    u32x4* sW = reinterpret_cast<u32x4*>(smem_a);                                   // [HEADS][F_HEAD][64]: 16-byte fragments
    u32x2* sP2 = reinterpret_cast<u32x2*>(smem_a + HEADS * F_HEAD * 1024);           // [HEADS][NCT][64]: W_proj columns 32 + 4g .. of the head
    float* sBqkv = reinterpret_cast<float*>(smem_a + HEADS * (F_HEAD * 1024 + NCT * 512));      // [3C]
    // rel-pos bias (x log2 e) expanded to the score tiles' accumulator layout, key padding folded in as -inf:
    // [HEADS][ci][cj][lane] x 4 keys 16cj + 4g + r of query 16ci + (lane & 15)  -- one 16-byte read per score tile
    f32x4* sBias = reinterpret_cast<f32x4*>(sBqkv + 3 * HEADS * HDP);
    const int tid = threadIdx.x, lane_ = tid & 63, wave = tid >> 6, g_ = lane_ >> 4;
    for (int idx = tid; idx < HEADS * F_HEAD * 64; idx += NT_) {
        const int f = idx >> 6, l = idx & 63, i = l & 15, gg = l >> 4, h = f / F_HEAD, fl = f - h * F_HEAD;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (fl < F_QK + F_V) {
            int row, k;
            if (fl < F_QK) {                                   // q / k tile et: head rows 16et + i
                const int part = fl / (ET * NK), r = fl - part * ET * NK, et = r / NK;
                k = r - et * NK;
                const int e = 16 * et + i;
                row = e < HDP ? part * C + HDP * h + e : -1;
            } else {                                           // v tile ce
                const int r = fl - F_QK, ce = r / NK;
                k = r - ce * NK;
                const int e = ce < 2 ? hid(ce, i) : (i < 8 ? 32 + i : HDP);
                row = e < HDP ? 2 * C + HDP * h + e : -1;
            }
            const int col = 32 * k + 8 * gg;
            if (row >= 0 && col < C) v = *reinterpret_cast<const u32x4*>(p.wqkv + (size_t)row * C + col);
        } else {
            const int nt = fl - F_QK - F_V;
            v = *reinterpret_cast<const u32x4*>(p.wproj + (size_t)(16 * nt + i) * C + HDP * h + 8 * gg);
        }
        sW[idx] = v;
    }
    for (int idx = tid; idx < HEADS * NCT * 64; idx += NT_) {
        const int f = idx >> 6, l = idx & 63, i = l & 15, gg = l >> 4, h = f / NCT, nt = f - h * NCT;
        u32x2 v = {0u, 0u};
        if (gg < 2) v = *reinterpret_cast<const u32x2*>(p.wproj + (size_t)(16 * nt + i) * C + HDP * h + 32 + 4 * gg);
        sP2[idx] = v;
    }
    for (int i = tid; i < 3 * C; i += NT_) sBqkv[i] = p.bqkv[i];
    for (int idx = tid; idx < HEADS * 16 * 64; idx += NT_) {
        const int l = idx & 63, tile = (idx >> 6) & 15, h = idx >> 10, ci = tile >> 2, cj = tile & 3;
        const int i = 16 * ci + (l & 15), ai = rel_a7(i < AT_N ? i : 0);
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * cj + 4 * (l >> 4) + r;
            v[r] = j < AT_N ? p.table[(ai - rel_a7(j) + 84) * HEADS + h] * 1.44269504088896340736f : -INFINITY;
        }
        sBias[idx] = v;
    }
    __syncthreads();
    const auto rx = MAKE_RSRC(p.x);
    const auto ro = MAKE_RSRC(p.out);
    const auto rgam = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gamma), 0, C * 4, 0x00020000);
    const auto rbet = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.beta), 0, C * 4, 0x00020000);
    const auto rbp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bproj), 0, C * 4, 0x00020000);
    const float inv_c = 1.f / (float)c_real, n_pad = (float)(32 * NK - c_real);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const float qscale = p.softmax_scale * 1.44269504088896340736f;
    for (int w = blockIdx.x * WAVES + wave; w < p.n_windows; w += gridDim.x * WAVES) {
        const int lane = opaque_lane<true>(lane_), i16 = lane & 15, g = lane >> 4;
        int row[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = 16 * t + i16;
            row[t] = n < AT_N ? p.rowmap[w * AT_N + n] : -1;
        }
        const float sc = p.scale ? p.scale[w / p.windows_per_sample] : 1.f;
        bf16x8 uf[4][NK];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned base = row[t] >= 0 ? (unsigned)row[t] * (unsigned)(C * 2) : OOB_OFF;
            u32x4 xr[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k)
                xr[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, (row[t] >= 0 && 32 * k + 8 * g < C) ? base + (32 * k + 8 * g) * 2 : OOB_OFF, 0, 0);
            float v[NK][8];
            float sm = 0.f;
#pragma unroll
            for (int k = 0; k < NK; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[k][2 * j] = blo(xr[k][j]);
                    v[k][2 * j + 1] = bhi(xr[k][j]);
                    sm += v[k][2 * j] + v[k][2 * j + 1];
                }
            sm = xor16_sum(sm);
            sm = xor32_sum(sm);
            const float mean = sm * inv_c;
            float qv = 0.f;
#pragma unroll
            for (int k = 0; k < NK; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[k][j] -= mean;
                    qv += v[k][j] * v[k][j];
                }
            qv = xor16_sum(qv);
            qv = xor32_sum(qv);
            const float rstd = rsqrtf(fmaxf(qv - n_pad * mean * mean, 0.f) * inv_c + p.eps);
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const f32x4 g0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rgam, (32 * k + 8 * g) * 4, 0, 0));
                const f32x4 g1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rgam, (32 * k + 8 * g + 4) * 4, 0, 0));
                const f32x4 b0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbet, (32 * k + 8 * g) * 4, 0, 0));
                const f32x4 b1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbet, (32 * k + 8 * g + 4) * 4, 0, 0));
                u32x4 o;
                o[0] = pack_bf16x2(v[k][0] * rstd * g0[0] + b0[0], v[k][1] * rstd * g0[1] + b0[1]);
                o[1] = pack_bf16x2(v[k][2] * rstd * g0[2] + b0[2], v[k][3] * rstd * g0[3] + b0[3]);
                o[2] = pack_bf16x2(v[k][4] * rstd * g1[0] + b1[0], v[k][5] * rstd * g1[1] + b1[1]);
                o[3] = pack_bf16x2(v[k][6] * rstd * g1[2] + b1[2], v[k][7] * rstd * g1[3] + b1[3]);
                // zero-pad token (or tile padding): the reference pads AFTER LayerNorm
                uf[t][k] = row[t] >= 0 ? __builtin_bit_cast(bf16x8, o) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        f32x4 accY[4][NCT];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) accY[t][ct] = zero;
#pragma unroll 1
        for (int h = 0; h < HEADS; ++h) {
            const u32x4* hw = sW + h * F_HEAD * 64;
            const float* hb = sBqkv + HDP * h;
            bf16x8 kf[4], vt[2][ET];
            s16x4 kf2[4];
            {                                                   // V: D[token][e], packed over token tiles -> A operand V^T
                f32x4 va[4][ET];
#pragma unroll
                for (int ce = 0; ce < ET; ++ce) {
                    const int e = ce < 2 ? hid(ce, i16) : (i16 < 8 ? 32 + i16 : -1);
                    const float bv = e >= 0 ? hb[2 * C + e] : 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t) va[t][ce] = (f32x4){bv, bv, bv, bv};
#pragma unroll
                    for (int k = 0; k < NK; ++k) {
                        const bf16x8 wv = LDS_FRAG(hw, F_QK + ce * NK + k, lane);
#pragma unroll
                        for (int t = 0; t < 4; ++t) va[t][ce] = MFMA(uf[t][k], wv, va[t][ce]);
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int ce = 0; ce < ET; ++ce) vt[s2][ce] = pack2(va[2 * s2][ce], va[2 * s2 + 1][ce]);
            }
            {                                                   // K: D[e][token]
                f32x4 ka[4][ET];
#pragma unroll
                for (int et = 0; et < ET; ++et) {
                    f32x4 bk = zero;
                    if (et < 2 || g < 2) bk = *reinterpret_cast<const f32x4*>(&hb[C + 16 * et + 4 * g]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) ka[t][et] = bk;
#pragma unroll
                    for (int k = 0; k < NK; ++k) {
                        const bf16x8 wk = LDS_FRAG(hw, (ET + et) * NK + k, lane);
#pragma unroll
                        for (int t = 0; t < 4; ++t) ka[t][et] = MFMA(wk, uf[t][k], ka[t][et]);
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    kf[t] = pack2(ka[t][0], ka[t][1]);
                    kf2[t] = __builtin_bit_cast(s16x4, pack4(ka[t][2]));       // head rows 32 + 4g + r: the K = 16 operand layout
                }
            }
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                bf16x8 qf;
                s16x4 qf2;
                {                                               // Q of this query tile
                    f32x4 qa[ET];
#pragma unroll
                    for (int et = 0; et < ET; ++et) {
                        qa[et] = zero;
                        if (et < 2 || g < 2) qa[et] = *reinterpret_cast<const f32x4*>(&hb[16 * et + 4 * g]);
#pragma unroll
                        for (int k = 0; k < NK; ++k) qa[et] = MFMA(LDS_FRAG(hw, et * NK + k, lane), uf[ci][k], qa[et]);
                    }
                    qf = pack2(qa[0], qa[1]);
                    qf2 = __builtin_bit_cast(s16x4, pack4(qa[2]));
                }
                f32x4 st[4];
                // Scores and projection as chains of 16-deep MFMAs only (the 32-deep operands are two 16-deep ones side by side in the
                // same registers, and 3 x 4 passes cost what 8 + 4 do).  NOT a 32-deep MFMA followed by a 16-deep one on its result: hipcc
                // 7.2 pads nothing between  v_mfma_f32_16x16x32_bf16 D, ..  and a following  v_mfma_f32_16x16x16_bf16 D', .., C = D  with
                // D' != D, and the second then reads a stale C (seen as 0.5-0.9 relative error whenever the register allocator chose D' != D).
#pragma unroll
                for (int cj = 0; cj < 4; ++cj) {
                    st[cj] = MFMA16(LO4(kf[cj]), LO4(qf), zero);
                    st[cj] = MFMA16(HI4(kf[cj]), HI4(qf), st[cj]);
                    st[cj] = MFMA16(kf2[cj], qf2, st[cj]);
                }
                float mx = -INFINITY;
#pragma unroll
                for (int cj = 0; cj < 4; ++cj) {
                    st[cj] = __builtin_elementwise_fma(st[cj], (f32x4)(qscale), sBias[((h * 4 + ci) * 4 + cj) * 64 + lane]);      // log2 domain
                    mx = fmaxf(fmaxf(mx, fmaxf(st[cj][0], st[cj][1])), fmaxf(st[cj][2], st[cj][3]));
                }
                mx = xor16_max(mx);
                mx = xor32_max(mx);
                float sum = 0.f;
#pragma unroll
                for (int cj = 0; cj < 4; ++cj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        st[cj][r] = __builtin_amdgcn_exp2f(st[cj][r] - mx);
                        sum += st[cj][r];
                    }
                sum = xor16_sum(sum);
                sum = xor32_sum(sum);
                const float inv = 1.f / sum;
                const bf16x8 p0 = pack2(st[0] * inv, st[1] * inv), p1 = pack2(st[2] * inv, st[3] * inv);
                f32x4 o[ET];
#pragma unroll
                for (int ce = 0; ce < ET; ++ce) {
                    o[ce] = MFMA(vt[0][ce], p0, zero);
                    o[ce] = MFMA(vt[1][ce], p1, o[ce]);
                }
                const bf16x8 of0 = pack2(o[0], o[1]);
                const s16x4 of1 = __builtin_bit_cast(s16x4, pack4(o[2]));
#pragma unroll
                for (int nt = 0; nt < NCT; ++nt) {
                    const bf16x8 wp = LDS_FRAG(hw, F_QK + F_V + nt, lane);
                    accY[ci][nt] = MFMA16(LO4(wp), LO4(of0), accY[ci][nt]);
                    accY[ci][nt] = MFMA16(HI4(wp), HI4(of0), accY[ci][nt]);
                    accY[ci][nt] = MFMA16(__builtin_bit_cast(s16x4, sP2[(h * NCT + nt) * 64 + lane]), of1, accY[ci][nt]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const unsigned off = row[t] >= 0 ? (unsigned)row[t] * (unsigned)(C * 2) + (16 * ct + 4 * g) * 2 : OOB_OFF;
                const u32x2 xo = __builtin_amdgcn_raw_buffer_load_b64(rx, off, 0, 0);
                const f32x4 bp = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbp, (16 * ct + 4 * g) * 4, 0, 0));
                const f32x4 v = (accY[t][ct] + bp) * sc + unpack4(xo);
                __builtin_amdgcn_raw_buffer_store_b64(pack4(v), ro, off, 0, 0);
            }
    }
}

// ================================================================================================ attention half: backward
// One wave per window again; LayerNorm, q, k, v are recomputed from x, the projection's data gradient dO = (s dy) W_p is
// computed in-kernel, the attention core follows pk_attn.hip's two-pass formulation (P recomputed from the saved log-sum-exp in
// both layouts, delta_i = sum_e dO O from the saved output), and the qkv data gradient + LayerNorm backward + residual gradient
// are applied before anything leaves the registers:
//     dx = dy + dLN1( [dq | dk | dv] W_qkv ),   written to the pixel rows of the window's real tokens.
// Every MFMA operand is produced by an MFMA in the layout its consumer needs ("token form": lane = token, slots = head channels;
// "e form": lane = head channel, slots = tokens), so no tile is transposed through LDS.  The e-form weight rows are gathered with
// hid() so that the packed dq / dk / dv tiles are B operands in natural channel order for the qkv data-gradient MFMA.
// Emitted for the weight-gradient GEMMs (pk_wgrad_bf16): dqkv [windows*49][3C] and the LayerNorm output u in window order
// (zero rows for the pad tokens).  part_ln[block][2][C]: dgamma | dbeta sums; part_rpb[(block*4 + wave)][heads][169]: rel-pos-bias
// gradient folded per wave (fixed order, deterministic).
template <int C>
__global__ void __launch_bounds__(256, 2) k_attn_bwd(AttnArgs p, float* __restrict__ part_ln, float* __restrict__ part_rpb) {
    constexpr int HEADS = C / 32, NK = C / 32, NCT = C / 16, NKQ = 3 * C / 32;
    constexpr int F_A = 6 * NK, F_B = 4 * NK;                      // per head: token-form q/k/v rows, e-form q/k rows
    __shared__ __attribute__((aligned(16))) u32x4 sWa[HEADS * F_A * 64];       // ((h*3 + part)*2 + et)*NK + k : rows part*C + 32h + 16et + i
    __shared__ __attribute__((aligned(16))) u32x4 sWb[HEADS * F_B * 64];       // ((h*2 + part)*2 + ce)*NK + k : rows part*C + 32h + hid(ce, i)
    __shared__ __attribute__((aligned(16))) u32x4 sPa[HEADS * 2 * NK * 64];    // (h*2 + et)*NK + k : W_p^T rows 32h + 16et + i
    __shared__ __attribute__((aligned(16))) u32x4 sPb[HEADS * 2 * NK * 64];    // (h*2 + ce)*NK + k : W_p^T rows 32h + hid(ce, i)
    __shared__ __attribute__((aligned(16))) u32x4 sWt[NCT * NKQ * 64];         // ct*NKQ + kq : W_qkv^T rows 16ct + i, columns 32kq + 8g ..
    __shared__ __attribute__((aligned(16))) float sBqkv[3 * C];
    __shared__ __attribute__((aligned(16))) float sGam[C], sBet[C];
    __shared__ __attribute__((aligned(16))) int sA[64];                        // A(t) = 13 (t / 7) + t % 7 of token t, -1 for the tile padding
    __shared__ float sBias[HEADS][176];
    __shared__ __attribute__((aligned(16))) float sLse[4][64], sDelta[4][64];
    // LayerNorm statistics and pixel rows of the window's tokens, parked until phase 2 (12 registers that were live through both passes:
    // the kernel sits at the 256-register limit, 10 of them spilled to scratch before)
    __shared__ float sMean[4][64], sRstd[4][64];
    __shared__ int sRow[4][64];
    __shared__ float sTab[4][HEADS][192];                                      // rel-pos-bias gradient of this wave: entry e owned by lane e % 64
    // dS of the current (window, head), folded onto sTab once per window.  FOLD_PAD floats of slack on both sides: the fold reads (and
    // masks) words up to 294 before wave 0's tile and 293 after wave 3's -- they stay inside this array whatever the LDS layout is.
    constexpr int FOLD_PAD = 296;
    __shared__ float sFoldBuf[FOLD_PAD + 4 * AT_N * AT_N + FOLD_PAD];
    float* const sFold0 = sFoldBuf + FOLD_PAD;
    __shared__ float sRed[4][2][C];
    const int tid = threadIdx.x, lane_ = tid & 63, wave = tid >> 6;
    stage_frags(sWa, p.wqkv, HEADS * F_A, C,
                [](int f, int i) { const int r = f / NK, et = r & 1, part = (r >> 1) % 3, h = r / 6; return part * C + 32 * h + 16 * et + i; },
                [](int f) { return 32 * (f % NK); });
    stage_frags(sWb, p.wqkv, HEADS * F_B, C,
                [](int f, int i) { const int r = f / NK, ce = r & 1, part = (r >> 1) & 1, h = r / 4; return part * C + 32 * h + hid(ce, i); },
                [](int f) { return 32 * (f % NK); });
    stage_frags(sPa, p.wproj_t, HEADS * 2 * NK, C, [](int f, int i) { const int r = f / NK; return 32 * (r >> 1) + 16 * (r & 1) + i; },
                [](int f) { return 32 * (f % NK); });
    stage_frags(sPb, p.wproj_t, HEADS * 2 * NK, C, [](int f, int i) { const int r = f / NK; return 32 * (r >> 1) + hid(r & 1, i); },
                [](int f) { return 32 * (f % NK); });
    stage_frags(sWt, p.wqkv_t, NCT * NKQ, 3 * C, [](int f, int i) { return 16 * (f / NKQ) + i; }, [](int f) { return 32 * (f % NKQ); });
    for (int i = tid; i < 3 * C; i += 256) sBqkv[i] = p.bqkv[i];
    for (int i = tid; i < C; i += 256) {
        sGam[i] = p.gamma[i];
        sBet[i] = p.beta[i];
    }
    if (tid < 64) sA[tid] = tid < AT_N ? rel_a7(tid) : -1;
    // (the bias table is kept in units of log2 and the score / dP accumulators START at -lse / scale and -delta, so
    // P = exp2(fma(acc, scale * log2 e, bias')) and dS = P * acc': three instructions per score besides the bias gather)
    for (int i = tid; i < HEADS * 169; i += 256) sBias[i / 169][i % 169] = p.table[(i % 169) * HEADS + i / 169] * LOG2E_F;
    for (int i = tid; i < 4 * HEADS * 192; i += 256) (&sTab[0][0][0])[i] = 0.f;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 dgam[NCT], dbet[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) dgam[ct] = dbet[ct] = zero;
    __syncthreads();
    const auto rx = MAKE_RSRC(p.x);
    const auto rg = MAKE_RSRC(p.dy);
    const auto ro = MAKE_RSRC(p.out);
    const auto rs = MAKE_RSRC(p.o_save);
    const float inv_c = 1.f / (float)C, scale = p.softmax_scale, scale2 = p.softmax_scale * LOG2E_F, inv_scale = 1.f / p.softmax_scale;
    for (int w = blockIdx.x * 4 + wave; w < p.n_windows; w += gridDim.x * 4) {
        const int lane = opaque_lane<true>(lane_), i16 = lane & 15, g = lane >> 4;
        int row[4];
        // du = dqkv W_qkv, accumulated tile by tile from the packed dq / dk / dv fragments as the passes produce them: the lane that packs
        // the 8 channels 32 kq + 8g .. of token 16t + i16 for the store is the lane whose B operand they are (no read-back, no transpose)
        f32x4 du[4][NCT];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) du[t][ct] = zero;
        {   // ================= phase 1: the attention core; writes dqkv (and u) rows of this window
            u32x4 xr[4][NK], dyr[4][NK];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int n = 16 * t + i16;
                row[t] = n < AT_N ? p.rowmap[w * AT_N + n] : -1;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const unsigned off = row[t] >= 0 ? (unsigned)row[t] * (unsigned)(C * 2) + (32 * k + 8 * g) * 2 : OOB_OFF;
                    xr[t][k] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
                    dyr[t][k] = __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0);
                }
            const float sc = p.scale ? p.scale[w / p.windows_per_sample] : 1.f;
            bf16x8 uf[4][NK], gf[4][NK];
            {
                float gam[NK][8], bet[NK][8];
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    *reinterpret_cast<f32x4*>(&gam[k][0]) = *reinterpret_cast<const f32x4*>(&sGam[32 * k + 8 * g]);
                    *reinterpret_cast<f32x4*>(&gam[k][4]) = *reinterpret_cast<const f32x4*>(&sGam[32 * k + 8 * g + 4]);
                    *reinterpret_cast<f32x4*>(&bet[k][0]) = *reinterpret_cast<const f32x4*>(&sBet[32 * k + 8 * g]);
                    *reinterpret_cast<f32x4*>(&bet[k][4]) = *reinterpret_cast<const f32x4*>(&sBet[32 * k + 8 * g + 4]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float mean_t, rstd_t;
                    ln_row<NK>(xr[t], gam, bet, inv_c, p.eps, uf[t], mean_t, rstd_t);
                    if (g == 0) {
                        sMean[wave][16 * t + i16] = mean_t;
                        sRstd[wave][16 * t + i16] = rstd_t;
                        sRow[wave][16 * t + i16] = row[t];
                    }
                    scale_rows<NK>(dyr[t], sc, gf[t]);
                    if (row[t] < 0) {
#pragma unroll
                        for (int k = 0; k < NK; ++k) uf[t][k] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    }
                    if (16 * t + i16 < AT_N) {
#pragma unroll
                        for (int k = 0; k < NK; ++k)
                            *reinterpret_cast<bf16x8*>(p.u_save + ((size_t)w * AT_N + 16 * t + i16) * C + 32 * k + 8 * g) = uf[t][k];
                    }
                }
            }
            uint16_t* dq_base = p.dqkv + (size_t)w * AT_N * 3 * C;
#pragma unroll 1
            for (int h = 0; h < HEADS; ++h) {
                // ---- token-form operands: lane = token, slots = head channel e = 16 (jj >> 2) + 4g + (jj & 3)
                bf16x8 qf[4], kf[4], vf[4], dof[4];
                // the saved attention output and log-sum-exp of all four token tiles are requested up front (PMC: the waves of this kernel
                // sat parked on s_waitcnt 58 % of their cycles -- one dependent round trip per tile right where the value was needed)
                u32x2 osv[4][2];
                float lsv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int i = 16 * t + i16;
                    const unsigned ob = i < AT_N ? (unsigned)((w * AT_N + i) * C + 32 * h + 4 * g) * 2u : OOB_OFF;
                    osv[t][0] = __builtin_amdgcn_raw_buffer_load_b64(rs, ob, 0, 0);
                    osv[t][1] = __builtin_amdgcn_raw_buffer_load_b64(rs, i < AT_N ? ob + 32 : OOB_OFF, 0, 0);
                    lsv[t] = i < AT_N ? p.lse[((size_t)w * HEADS + h) * AT_N + i] : 0.f;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    f32x4 a[4][2];      // [q, k, v, dO][et]
#pragma unroll
                    for (int et = 0; et < 2; ++et) {
#pragma unroll
                        for (int part = 0; part < 3; ++part) a[part][et] = *reinterpret_cast<const f32x4*>(&sBqkv[part * C + 32 * h + 16 * et + 4 * g]);
                        a[3][et] = zero;
#pragma unroll
                        for (int k = 0; k < NK; ++k) {
#pragma unroll
                            for (int part = 0; part < 3; ++part)
                                a[part][et] = MFMA(LDS_FRAG(sWa, ((h * 3 + part) * 2 + et) * NK + k, lane), uf[t][k], a[part][et]);
                            a[3][et] = MFMA(LDS_FRAG(sPa, (h * 2 + et) * NK + k, lane), gf[t][k], a[3][et]);
                        }
                    }
                    qf[t] = pack2(a[0][0], a[0][1]);
                    kf[t] = pack2(a[1][0], a[1][1]);
                    vf[t] = pack2(a[2][0], a[2][1]);
                    dof[t] = pack2(a[3][0], a[3][1]);
                    // delta_i = sum_e dO[i][e] O[i][e]: this lane's 8 channels of token i = 16t + i16, then the 4 lanes of the token
                    const int i = 16 * t + i16;
                    const f32x4 o0 = unpack4(osv[t][0]), o1 = unpack4(osv[t][1]);
                    float de = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) de += a[3][0][r] * o0[r] + a[3][1][r] * o1[r];
                    de = xor16_sum(de);
                    de = xor32_sum(de);
                    if (g == 0) {
                        sLse[wave][i] = -lsv[t] * inv_scale;
                        sDelta[wave][i] = -de;
                    }
                }
                // ---- e-form operands (lane = head channel hid(ce, i16), slots = the 32 tokens of K-step s in accumulator order): K now,
                //      Q and dO after pass 1 (fewer live registers)
                bf16x8 Kt[2][2];
#pragma unroll
                for (int ce = 0; ce < 2; ++ce) {
                    f32x4 a[4];
                    const float bk = sBqkv[C + 32 * h + hid(ce, i16)];
#pragma unroll
                    for (int t = 0; t < 4; ++t) a[t] = (f32x4){bk, bk, bk, bk};
#pragma unroll
                    for (int k = 0; k < NK; ++k) {
                        const bf16x8 wk = LDS_FRAG(sWb, ((h * 2 + 1) * 2 + ce) * NK + k, lane);
#pragma unroll
                        for (int t = 0; t < 4; ++t) a[t] = MFMA(uf[t][k], wk, a[t]);
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) Kt[s][ce] = pack2(a[2 * s], a[2 * s + 1]);
                }
                LDS_FENCE();                // sLse / sDelta of this wave are visible to its own reads below
                // ---- pass 1: transposed scores, lane = query i:  dS^T, bias gradient, dQ^T = scale * K^T dS^T
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    // (validity from the tile index, not from loaded values: tiles 0..2 hold only real tokens, so their selects,
                    // clamps and predicated stores disappear at compile time; tile 3 has the single real token 48)
                    const int i = 16 * ci + i16;
                    const bool iok = ci < 3 || i16 == 0;
                    const int ai = rel_a7(iok ? i : 0) + 84;
                    const float li = sLse[wave][i], di = sDelta[wave][i];
                    const f32x4 li4 = {li, li, li, li}, di4 = {di, di, di, di};
                    f32x4 dq[2] = {zero, zero};
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        f32x4 ds[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int cj = 2 * s + u;
                            const f32x4 scr = MFMA(kf[cj], qf[ci], li4), dp = MFMA(vf[cj], dof[ci], di4);
                            const int4 aj = *reinterpret_cast<const int4*>(&sA[16 * cj + 4 * g]);
                            const int ajr[4] = {aj.x, aj.y, aj.z, aj.w};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const bool jv = cj < 3 || (g == 0 && r == 0);
                                const int e = ai - (cj < 3 ? ajr[r] : max(ajr[r], 0));
                                const float ev = __builtin_amdgcn_exp2f(__builtin_fmaf(scr[r], scale2, sBias[h][e]));
                                const float pv = (iok && jv) ? ev : 0.f;
                                ds[u][r] = pv * dp[r];
                                // d(bias)[i][j] of this window -> wave-private tile, folded onto the 169 table entries after the pass.
                                // (LDS float atomics straight into the table were measured: ~200 cycles per ds_add_f32 wave-instruction,
                                // 95 us of a 159 us kernel; 64 accumulator registers per head instead cost a wave of occupancy.)
                                if (iok && jv) sFold0[wave * (AT_N * AT_N) + i * AT_N + 16 * cj + 4 * g + r] = ds[u][r];
                            }
                        }
                        const bf16x8 df = pack2(ds[0], ds[1]);
#pragma unroll
                        for (int ce = 0; ce < 2; ++ce) dq[ce] = MFMA(Kt[s][ce], df, dq[ce]);
                    }
                    {       // rows 4g + r of tile ce  <->  head channel 8g + 4ce + r
                        const u32x2 q0 = pack4(dq[0] * scale), q1 = pack4(dq[1] * scale);
                        const u32x4 qv = iok ? (u32x4){q0[0], q0[1], q1[0], q1[1]} : (u32x4){0u, 0u, 0u, 0u};
                        if (iok) *reinterpret_cast<u32x4*>(dq_base + (size_t)i * 3 * C + 32 * h + 8 * g) = qv;
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct)
                            du[ci][ct] = MFMA(LDS_FRAG(sWt, ct * NKQ + h, lane), __builtin_bit_cast(bf16x8, qv), du[ci][ct]);
                    }
                }
                {        // fold: table entry e = (dy + 6) * 13 + dx + 6 sums dS[i][j] over the pairs with i - j = (dy, dx)
                    LDS_FENCE();
                    // (rows yj in a loop, the seven xj of a row unrolled with a select: the wave pays for its longest lane, 49 trips of a
                    // two-level loop before, 5 + 7 + 4 row trips over the three rounds now; terms in the same order)
                    for (int e = lane; e < 169; e += 64) {
                        const int dy_ = e / 13 - 6, dx_ = e % 13 - 6;
                        const int x0 = max(0, -dx_), x1 = min(7, 7 - dx_);
                        const int base = wave * (AT_N * AT_N) + (dy_ * 7 + dx_) * AT_N;        // + yj * 350 + xj * 50
                        unsigned keep[7];               // (bit masks, not selects: a select lets the compiler sink each load into a branch)
#pragma unroll
                        for (int xj = 0; xj < 7; ++xj) keep[xj] = (xj >= x0 && xj < x1) ? 0xffffffffu : 0u;
                        float acc = 0.f;
                        for (int yj = max(0, -dy_); yj < min(7, 7 - dy_); ++yj) {
#pragma unroll
                            for (int xj = 0; xj < 7; ++xj) {
                                // (an xj outside [x0, x1) addresses some other float of the padded tile array: read, masked to +0)
                                const float v = sFold0[base + yj * (7 * AT_N + 7) + xj * (AT_N + 1)];
                                acc += __uint_as_float(__float_as_uint(v) & keep[xj]);
                            }
                        }
                        sTab[wave][h][e] += acc;
                    }
                    LDS_FENCE();
                }
                bf16x8 Qt[2][2], dOt[2][2];
#pragma unroll
                for (int ce = 0; ce < 2; ++ce) {
                    f32x4 a[2][4];
                    const float bq = sBqkv[32 * h + hid(ce, i16)];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        a[0][t] = (f32x4){bq, bq, bq, bq};
                        a[1][t] = zero;
                    }
#pragma unroll
                    for (int k = 0; k < NK; ++k) {
                        const bf16x8 wq = LDS_FRAG(sWb, ((h * 2 + 0) * 2 + ce) * NK + k, lane), wp = LDS_FRAG(sPb, (h * 2 + ce) * NK + k, lane);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            a[0][t] = MFMA(uf[t][k], wq, a[0][t]);
                            a[1][t] = MFMA(gf[t][k], wp, a[1][t]);
                        }
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        Qt[s][ce] = pack2(a[0][2 * s], a[0][2 * s + 1]);
                        dOt[s][ce] = pack2(a[1][2 * s], a[1][2 * s + 1]);
                    }
                }
                // ---- pass 2: plain scores, lane = key j:  dV^T = dO^T P,  dK^T = scale * Q^T dS
#pragma unroll
                for (int cj = 0; cj < 4; ++cj) {
                    const int j = 16 * cj + i16;
                    const bool jok = cj < 3 || i16 == 0;
                    const int ajn = 84 - rel_a7(jok ? j : 0);
                    f32x4 dv[2] = {zero, zero}, dk[2] = {zero, zero};
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        f32x4 pp[2], ds[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int ci = 2 * s + u;
                            const f32x4 l4 = *reinterpret_cast<const f32x4*>(&sLse[wave][16 * ci + 4 * g]);
                            const f32x4 de4 = *reinterpret_cast<const f32x4*>(&sDelta[wave][16 * ci + 4 * g]);
                            const f32x4 scr = MFMA(qf[ci], kf[cj], l4), dp = MFMA(dof[ci], vf[cj], de4);
                            const int4 aq = *reinterpret_cast<const int4*>(&sA[16 * ci + 4 * g]);
                            const int aqr[4] = {aq.x, aq.y, aq.z, aq.w};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const bool iv = ci < 3 || (g == 0 && r == 0);
                                const float ev = __builtin_amdgcn_exp2f(__builtin_fmaf(scr[r], scale2, sBias[h][(ci < 3 ? aqr[r] : max(aqr[r], 0)) + ajn]));
                                const float pv = (jok && iv) ? ev : 0.f;
                                pp[u][r] = pv;
                                ds[u][r] = pv * dp[r];
                            }
                        }
                        const bf16x8 pf = pack2(pp[0], pp[1]), df = pack2(ds[0], ds[1]);
#pragma unroll
                        for (int ce = 0; ce < 2; ++ce) {
                            dv[ce] = MFMA(dOt[s][ce], pf, dv[ce]);
                            dk[ce] = MFMA(Qt[s][ce], df, dk[ce]);
                        }
                    }
                    {
                        const u32x2 k0 = pack4(dk[0] * scale), k1 = pack4(dk[1] * scale), v0 = pack4(dv[0]), v1 = pack4(dv[1]);
                        const u32x4 kv = jok ? (u32x4){k0[0], k0[1], k1[0], k1[1]} : (u32x4){0u, 0u, 0u, 0u};
                        const u32x4 vv = jok ? (u32x4){v0[0], v0[1], v1[0], v1[1]} : (u32x4){0u, 0u, 0u, 0u};
                        if (jok) {
                            *reinterpret_cast<u32x4*>(dq_base + (size_t)j * 3 * C + C + 32 * h + 8 * g) = kv;
                            *reinterpret_cast<u32x4*>(dq_base + (size_t)j * 3 * C + 2 * C + 32 * h + 8 * g) = vv;
                        }
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) {
                            du[cj][ct] = MFMA(LDS_FRAG(sWt, ct * NKQ + HEADS + h, lane), __builtin_bit_cast(bf16x8, kv), du[cj][ct]);
                            du[cj][ct] = MFMA(LDS_FRAG(sWt, ct * NKQ + 2 * HEADS + h, lane), __builtin_bit_cast(bf16x8, vv), du[cj][ct]);
                        }
                    }
                }
                LDS_FENCE();                // the reads of sLse / sDelta complete before the next head rewrites them
            }
        }
        // ================= phase 2: LayerNorm backward + residual gradient in accumulator layout, one token tile at a time; x and dy
        // of all four tiles are requested up front (one round trip instead of four)
        u32x2 xo4[4][NCT], dyo4[4][NCT];
        int row2[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) row2[t] = sRow[wave][16 * t + i16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const unsigned off = row2[t] >= 0 ? (unsigned)row2[t] * (unsigned)(C * 2) + (16 * ct + 4 * g) * 2 : OOB_OFF;
                xo4[t][ct] = __builtin_amdgcn_raw_buffer_load_b64(rx, off, 0, 0);
                dyo4[t][ct] = __builtin_amdgcn_raw_buffer_load_b64(rg, off, 0, 0);
            }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int rw = row2[t];
            const float mu = sMean[wave][16 * t + i16], rsd = sRstd[wave][16 * t + i16];
            const bool valid = rw >= 0;
            f32x4 xh[NCT], gamA[NCT];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                gamA[ct] = *reinterpret_cast<const f32x4*>(&sGam[16 * ct + 4 * g]);
                xh[ct] = (unpack4(xo4[t][ct]) - mu) * rsd;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gh = du[t][ct][r] * gamA[ct][r];
                    s1 += gh;
                    s2 += gh * xh[ct][r];
                }
                if (valid) {            // pad tokens are not LayerNorm outputs: their gradient is dropped (hrformer.py:103-114)
                    dgam[ct] += du[t][ct] * xh[ct];
                    dbet[ct] += du[t][ct];
                }
            }
            s1 = xor16_sum(s1);
            s1 = xor32_sum(s1);
            s2 = xor16_sum(s2);
            s2 = xor32_sum(s2);
            const float m1 = s1 * inv_c, m2 = s2 * inv_c;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f32x4 o = (du[t][ct] * gamA[ct] - m1 - xh[ct] * m2) * rsd + unpack4(dyo4[t][ct]);
                __builtin_amdgcn_raw_buffer_store_b64(pack4(o), ro, valid ? (unsigned)rw * (unsigned)(C * 2) + (16 * ct + 4 * g) * 2 : OOB_OFF, 0, 0);
            }
        }
    }
    // ---- dgamma / dbeta: over the 16 tokens of a lane group, then over the 4 waves
    const int i16 = lane_ & 15, g = lane_ >> 4;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = dgam[ct][r], b = dbet[ct][r];
            a = lanes_sum<16>(a);
            b = lanes_sum<16>(b);
            if (i16 == 0) {
                sRed[wave][0][16 * ct + 4 * g + r] = a;
                sRed[wave][1][16 * ct + 4 * g + r] = b;
            }
        }
    __syncthreads();
    // ---- rel-pos-bias gradient of each wave: [block*4 + wave][head][169]
    for (int i = tid; i < 4 * HEADS * 169; i += 256) {
        const int wv = i / (HEADS * 169), r = i % (HEADS * 169);
        part_rpb[((size_t)blockIdx.x * 4 + wv) * HEADS * 169 + r] = sTab[wv][r / 169][r % 169];
    }
    if (tid < 2 * C) {
        const int which = tid / C, c = tid - which * C;
        part_ln[(size_t)blockIdx.x * 2 * C + tid] = ((sRed[0][which][c] + sRed[1][which][c]) + sRed[2][which][c]) + sRed[3][which][c];
    }
}

static inline int attn_blocks(int n_windows) {
    // one 4-wave workgroup per CU, each wave walking ~4 windows (B = 64): 1 024 workgroups (two rounds of two per CU, one window per wave)
    // cost the step 0.13 ms more (15.49-15.56 vs 15.38 ms, three alternating runs; 128 / 192 / 512 / 2 048: 15.37 / 15.41 / 15.37 / 15.49) --
    // a quarter of the weight prologues, and room on every CU for the other branches' workgroups
    static const int cap = PK_KNOB("PK_ATTN_WGS", 256);
    const int need = (n_windows + 3) / 4;
    return need < cap ? (need < 1 ? 1 : need) : cap;
}
extern "C" int pk_attn_block_blocks(int n_windows) { return attn_blocks(n_windows); }
extern "C" int pk_attn_block_bwd(const void* dy, const void* x, const int32_t* rowmap, const float* gamma, const float* beta,
                                 const float* rel_table, const void* wqkv, const float* bqkv, const void* wqkv_t, const void* wproj_t,
                                 const float* row_scale, const void* o_saved, const float* lse, void* dx, void* dqkv, void* u_out,
                                 float* ln_partial, float* rpb_partial, int n_windows, int windows_per_sample, int heads, int C, float eps,
                                 void* stream) {
    PK_SUPPORTED((C == 32 || C == 64) && heads * 32 == C, "pk_attn_block_bwd: C=%d heads=%d (built for C = 32 / 64 with head_dim 32)", C, heads);
    PK_REQUIRE(dy && x && rowmap && gamma && beta && rel_table && wqkv && bqkv && wqkv_t && wproj_t && o_saved && lse && dx && dqkv && u_out &&
               ln_partial && rpb_partial && n_windows > 0, "pk_attn_block_bwd: null pointer");
    PK_REQUIRE(!row_scale || windows_per_sample > 0, "pk_attn_block_bwd: row_scale needs windows_per_sample");
    PK_REQUIRE(((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)wqkv) | ((uintptr_t)wqkv_t) | ((uintptr_t)wproj_t) |
                 ((uintptr_t)o_saved) | ((uintptr_t)dqkv) | ((uintptr_t)u_out) | ((uintptr_t)gamma)) & 15) == 0, "pk_attn_block_bwd: 16-byte alignment");
    PK_REQUIRE((int64_t)n_windows * AT_N * 3 * C < 0x3fffffffLL, "pk_attn_block_bwd: too large for 32-bit offsets");
    AttnArgs a{};
    a.x = (const uint16_t*)x; a.dy = (const uint16_t*)dy; a.out = (uint16_t*)dx; a.rowmap = rowmap; a.gamma = gamma; a.beta = beta;
    a.table = rel_table; a.bqkv = bqkv; a.scale = row_scale; a.wqkv = (const uint16_t*)wqkv; a.wqkv_t = (const uint16_t*)wqkv_t;
    a.wproj_t = (const uint16_t*)wproj_t; a.o_save = (uint16_t*)const_cast<void*>(o_saved); a.lse = const_cast<float*>(lse);
    a.dqkv = (uint16_t*)dqkv; a.u_save = (uint16_t*)u_out; a.n_windows = n_windows;
    a.windows_per_sample = windows_per_sample > 0 ? windows_per_sample : 1; a.eps = eps; a.softmax_scale = 1.f / sqrtf(32.f);
    const dim3 grid(attn_blocks(n_windows)), block(256);
    if (C == 32) hipLaunchKernelGGL(k_attn_bwd<32>, grid, block, 0, (hipStream_t)stream, a, ln_partial, rpb_partial);
    else hipLaunchKernelGGL(k_attn_bwd<64>, grid, block, 0, (hipStream_t)stream, a, ln_partial, rpb_partial);
    return pk_launch_status("pk_attn_block_bwd");
}
extern "C" int pk_attn_block_supported(int C, int heads) { return (C == 32 || C == 64) && heads * 32 == C; }
extern "C" int pk_attn_block_fwd(const void* x, const int32_t* rowmap, const float* gamma, const float* beta, const float* rel_table,
                                 const void* wqkv, const float* bqkv, const void* wproj, const float* bproj, const float* row_scale,
                                 void* y, void* o_save, float* lse, int n_windows, int windows_per_sample, int heads, int C, float eps,
                                 void* stream) {
    PK_SUPPORTED(pk_attn_block_supported(C, heads), "pk_attn_block_fwd: C=%d heads=%d (built for C = 32 / 64 with head_dim 32)", C, heads);
    PK_REQUIRE(x && rowmap && gamma && beta && rel_table && wqkv && bqkv && wproj && bproj && y && n_windows > 0, "pk_attn_block_fwd: null pointer");
    PK_REQUIRE(!row_scale || windows_per_sample > 0, "pk_attn_block_fwd: row_scale needs windows_per_sample");
    PK_REQUIRE(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)wqkv) | ((uintptr_t)wproj) | ((uintptr_t)o_save) | ((uintptr_t)bproj)) & 15) == 0,
               "pk_attn_block_fwd: 16-byte alignment");
    PK_REQUIRE((int64_t)n_windows * AT_N * 3 * C < 0x3fffffffLL, "pk_attn_block_fwd: too large for 32-bit offsets");
    AttnArgs a{};
    a.x = (const uint16_t*)x; a.out = (uint16_t*)y; a.rowmap = rowmap; a.gamma = gamma; a.beta = beta; a.table = rel_table; a.bqkv = bqkv;
    a.bproj = bproj; a.scale = row_scale; a.wqkv = (const uint16_t*)wqkv; a.wproj = (const uint16_t*)wproj; a.o_save = (uint16_t*)o_save;
    a.lse = lse; a.n_windows = n_windows; a.windows_per_sample = windows_per_sample > 0 ? windows_per_sample : 1; a.eps = eps;
    a.softmax_scale = 1.f / sqrtf(32.f);
    const dim3 grid(attn_blocks(n_windows)), block(256);
    if (C == 32) hipLaunchKernelGGL(k_attn_fwd<32>, grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_attn_fwd<64>, grid, block, 0, (hipStream_t)stream, a);
    return pk_launch_status("pk_attn_block_fwd");
}

// Head_dim-40 twins, forward only.  `n_windows` > 0 also asks whether the launch has enough windows to pay (one wave per window).
extern "C" int pk_attn_block_wide_supported(int C, int heads, int n_windows) {
    static const int on = PK_KNOB("PK_ATTN_WIDE", 1);
    static const int min_win = PK_KNOB("PK_ATTN_WIDE_MIN_WINDOWS", 1024);
    if (!(on && C == 80 && heads == 2)) return 0;
    return n_windows <= 0 || n_windows >= min_win;
}
template <int NK, int NCT, int HEADS, int WAVES>
static int attn_wide_launch(const AttnArgs& a, int C, int c_real, hipStream_t st) {
    const int lds = HEADS * ((3 * 3 * NK + NCT) * 1024 + NCT * 512) + 3 * C * 4 + HEADS * 16 * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_attn_fwd_w<NK, NCT, HEADS, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            pk_set_error("pk_attn_block_wide_fwd: cannot raise the LDS limit: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    static const int cap = PK_KNOB("PK_ATTN_WIDE_WGS", 256);      // one 8-wave workgroup per CU (104 KB of LDS)
    const int need = (a.n_windows + WAVES - 1) / WAVES;
    hipLaunchKernelGGL((k_attn_fwd_w<NK, NCT, HEADS, WAVES>), dim3(need < cap ? need : cap), dim3(64 * WAVES), lds, st, a, C, c_real);
    return pk_launch_status("pk_attn_block_wide_fwd");
}
extern "C" int pk_attn_block_wide_fwd(const void* x, const int32_t* rowmap, const float* gamma, const float* beta, const float* rel_table,
                                      const void* wqkv, const float* bqkv, const void* wproj, const float* bproj, const float* row_scale,
                                      void* y, int n_windows, int windows_per_sample, int heads, int C, int c_real, float softmax_scale,
                                      float eps, void* stream) {
    PK_SUPPORTED(pk_attn_block_wide_supported(C, heads, 0), "pk_attn_block_wide_fwd: C=%d heads=%d (built for C = 80 with 2 heads of 40)", C, heads);
    PK_REQUIRE(x && rowmap && gamma && beta && rel_table && wqkv && bqkv && wproj && bproj && y && n_windows > 0, "pk_attn_block_wide_fwd: null pointer");
    PK_REQUIRE(c_real > 0 && c_real <= C && softmax_scale > 0.f, "pk_attn_block_wide_fwd: c_real=%d / softmax_scale=%g", c_real, (double)softmax_scale);
    PK_REQUIRE(!row_scale || windows_per_sample > 0, "pk_attn_block_wide_fwd: row_scale needs windows_per_sample");
    PK_REQUIRE(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)wqkv) | ((uintptr_t)wproj) | ((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)bproj)) & 15) == 0,
               "pk_attn_block_wide_fwd: 16-byte alignment");
    PK_REQUIRE((int64_t)n_windows * AT_N * 3 * C < 0x3fffffffLL, "pk_attn_block_wide_fwd: too large for 32-bit offsets");
    AttnArgs a{};
    a.x = (const uint16_t*)x; a.out = (uint16_t*)y; a.rowmap = rowmap; a.gamma = gamma; a.beta = beta; a.table = rel_table; a.bqkv = bqkv;
    a.bproj = bproj; a.scale = row_scale; a.wqkv = (const uint16_t*)wqkv; a.wproj = (const uint16_t*)wproj;
    a.n_windows = n_windows; a.windows_per_sample = windows_per_sample > 0 ? windows_per_sample : 1; a.eps = eps; a.softmax_scale = softmax_scale;
    return attn_wide_launch<3, 5, 2, 8>(a, C, c_real, (hipStream_t)stream);
}

// ================================================================================================ C-ABI
static inline int mlp_hidden_slice(int C) {          // hidden units per blockIdx.y slice of the weight-gradient kernel
    static const int env = PK_KNOB("PK_MLP_HS", 0);
    if (env == 32 || env == 64 || (env == 128 && C == 32)) return env;
    // Measured (B = 64): the 128 accumulator registers of HS * C = 4096 hold the kernel at one wave per SIMD (C = 32: 66 us);
    // half of that (two waves per SIMD) runs the same work in 40 us although every slice recomputes LayerNorm.  (Round 4, with VGPR-form
    // MFMAs the 128-unit slice needs 255 registers, i.e. two waves per SIMD as well: 15.48-15.51 vs 15.43-15.51 ms per step -- no gain.)
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
    static const int env_target = PK_KNOB("PK_MLP_WGS", 0);
    if (env_target > 0) target = env_target;
    const int need = (mlp_row_groups(M) + 3) / 4;
    return need < target ? (need < 1 ? 1 : need) : target;
}
static inline int mlp_dx_target(int C) {
    static const int t32 = PK_KNOB("PK_MLP_DX32_WGS", 512), t64 = PK_KNOB("PK_MLP_DX64_WGS", 512);
    return C == 32 ? t32 : t64;
}
static inline int mlp_fwd_target(int C) {
    static const int t32 = PK_KNOB("PK_MLP_FWD32_WGS", 512), t64 = PK_KNOB("PK_MLP_FWD64_WGS", 512);
    return C == 32 ? t32 : t64;
}
static inline int mlp_dw_target(int C) {
    // (C = 64: 32 x 8 slices = one workgroup per CU; 64 x 8 ran 38 us in isolation and the step 0.07 ms slower, see PK_ATTN_WGS)
    static const int t32 = PK_KNOB("PK_MLP_DW32_WGS", 256), t64 = PK_KNOB("PK_MLP_DW64_WGS", 32);
    return C == 32 ? t32 : t64;
}
extern "C" int pk_ln_mlp_dx_blocks(int M, int C) { return mlp_blocks(M, mlp_dx_target(C)); }
// (fewer, longer-lived workgroups: each one stages its weight slice and ends with a 4-phase slab reduction; 256 x slices
// workgroups beat 512 and 1024 at every slice width)
// workgroups beat 512 and 1024 at every slice width; C = 64 with its 8 slices: 64 x 8 = 38 us, 128 x 8 = 46 us, 256 x 8 = 67 us)
extern "C" int pk_ln_mlp_dw_blocks(int M, int C) { return mlp_blocks(M, mlp_dw_target(C)); }

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
    const dim3 grid(mlp_blocks(M, mlp_fwd_target(C))), block(256);
    if (C == 32) hipLaunchKernelGGL(k_mlp_fwd<32>, grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_mlp_fwd<64>, grid, block, 0, (hipStream_t)stream, a);
    return pk_launch_status("pk_ln_mlp_fwd");
}

static int pk_cu_count_block() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return n;
}
// Wide channels, forward only (inference; training of these widths takes the unfused sequence, whose backward needs the hidden saved).
// `M` > 0 also asks whether the launch pays: a workgroup walks ALL hidden slices for its 128 / 256 tokens, so a launch with few
// workgroups is one long serial chain per CU -- measured in cfg 5 (HRFormer-base twin, B = 64 with the flip): C = 320 with 13 824 tokens
// (108 workgroups) 151 us against ~70 us for the unfused fc1 + fc2 GEMMs (step 29.35 ms with it, 27.45 ms without), while C = 80
// (864 workgroups) and C = 160 (216) win 2.4 ms per step.  Default: at least PK_MLP_WIDE_MIN_WGS = 160 workgroups.
extern "C" int pk_ln_mlp_wide_supported(int C, int hidden, int M) {
    static const int on = PK_KNOB("PK_MLP_WIDE", 1);
    static const int min_wgs = PK_KNOB("PK_MLP_WIDE_MIN_WGS", 160);
    if (!(on && (C == 80 || C == 128 || C == 160 || C == 256 || C == 320) && hidden > 0 && hidden % 32 == 0 && hidden <= 2048)) return 0;
    const int per_wg = C >= 256 ? 128 : 256;
    return M <= 0 || (M + per_wg - 1) / per_wg >= min_wgs;
}
template <int NK, int NCT, int WAVES>
static int mlp_wide_launch(const MlpArgs& a, int C, int c_real, int HD, hipStream_t st) {
    constexpr int NF = 2 * NK + NCT;
    static const int res_on = PK_KNOB("PK_MLP_WIDE_RESIDENT", 1);
    const int NS = HD / 32, lds_res = NS * NF * 1024 + HD * 4;
    const bool resident = res_on && lds_res <= 150 * 1024;         // every slice fits: stage once, persistent workgroups, no per-slice barrier
    const int lds = resident ? lds_res : 2 * NF * 1024 + HD * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_mlp_fwd_w<NK, NCT, WAVES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)k_mlp_fwd_w<NK, NCT, WAVES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NF * 1024 + 2048 * 4);
        if (e != hipSuccess) {
            pk_set_error("pk_ln_mlp_wide_fwd: cannot raise the LDS limit: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    const int need = (a.M + 32 * WAVES - 1) / (32 * WAVES);
    if (resident) {
        const int cus = pk_cu_count_block();
        hipLaunchKernelGGL((k_mlp_fwd_w<NK, NCT, WAVES, true>), dim3(need < cus ? need : cus), dim3(64 * WAVES), lds, st, a, C, c_real, HD);
    } else {
        hipLaunchKernelGGL((k_mlp_fwd_w<NK, NCT, WAVES, false>), dim3(need), dim3(64 * WAVES), lds, st, a, C, c_real, HD);
    }
    return pk_launch_status("pk_ln_mlp_wide_fwd");
}
extern "C" int pk_ln_mlp_wide_fwd(const void* x, const float* gamma, const float* beta, const void* w1, const float* b1, const void* w2,
                                  const float* b2, const float* row_scale, void* y, int M, int C, int c_real, int hidden,
                                  int rows_per_sample, float eps, void* stream) {
    PK_SUPPORTED(pk_ln_mlp_wide_supported(C, hidden, 0), "pk_ln_mlp_wide_fwd: C=%d hidden=%d (built for C = 80 / 128 / 160 / 256 / 320, hidden %% 32 == 0)", C, hidden);
    PK_REQUIRE(x && gamma && beta && w1 && b1 && w2 && b2 && y && M > 0, "pk_ln_mlp_wide_fwd: null pointer / bad size");
    PK_REQUIRE(c_real > 0 && c_real <= C, "pk_ln_mlp_wide_fwd: c_real=%d outside (0, C=%d]", c_real, C);
    PK_REQUIRE(!row_scale || rows_per_sample > 0, "pk_ln_mlp_wide_fwd: scale needs rows_per_sample");
    PK_REQUIRE((int64_t)M * C < 0x3fffffffLL && (int64_t)hidden * C < 0x1fffffffLL, "pk_ln_mlp_wide_fwd: tensor too large for 32-bit byte offsets");
    PK_REQUIRE(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)w1) | ((uintptr_t)w2) | ((uintptr_t)gamma) | ((uintptr_t)beta) | ((uintptr_t)b2)) & 15) == 0,
               "pk_ln_mlp_wide_fwd: 16-byte alignment");
    MlpArgs a{};
    a.x = (const uint16_t*)x; a.out = (uint16_t*)y; a.gamma = gamma; a.beta = beta; a.b1 = b1; a.b2 = b2; a.scale = row_scale;
    a.w1 = (const uint16_t*)w1; a.w2 = (const uint16_t*)w2; a.M = M; a.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1; a.eps = eps;
    hipStream_t st = (hipStream_t)stream;
    switch (C) {
        case 80: return mlp_wide_launch<3, 5, 8>(a, C, c_real, hidden, st);
        case 128: return mlp_wide_launch<4, 8, 8>(a, C, c_real, hidden, st);
        case 160: return mlp_wide_launch<5, 10, 8>(a, C, c_real, hidden, st);
        case 256: return mlp_wide_launch<8, 16, 4>(a, C, c_real, hidden, st);
        default: return mlp_wide_launch<10, 20, 4>(a, C, c_real, hidden, st);
    }
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
