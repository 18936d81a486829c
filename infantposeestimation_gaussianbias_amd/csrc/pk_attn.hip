// Window attention core of the HRFormer block (A1): per (window, head)
//     P = softmax(scale * Q K^T + relbias[h]),  O = P V            N = 49 tokens (7x7), head_dim d <= 32
// One 64-lane wave per (window, head); the 49x49 problem is padded to 64x64 MFMA tiles (v_mfma_f32_16x16x32_bf16) and
// ONLY tile padding is masked: the reference's zero-pad tokens are real tokens (q=b_q, k=b_k, v=b_v) and take softmax
// mass (hrformer.py:80-84, no mask).
//
// Register-resident formulation.  The scores are computed TRANSPOSED, S^T = K Q^T, so that in the MFMA accumulator
// layout a lane owns ONE query i (column l16) and 16 of the 64 keys j (rows 16cj + 4g + r):
//   * softmax over j is 16 in-lane values + two xor-shuffles (lanes 16 and 32 apart),
//   * the bf16 probabilities ARE the B-operand fragment of the next MFMA (column i, k = j) if the contraction index is
//     enumerated in accumulator order: k-slot (g, jj) of K-step s  <->  key j = 16(2s + jj/4) + 4g + jj%4.  The other
//     operand (V^T, rows = channel e) is read from a row-major LDS tile with the gfx950 transpose read
//     `ds_read_b64_tr_b16` using the same key enumeration.  P never goes through LDS.
//   * results come out transposed (O^T: rows = channel, column = token): a lane holds 4 consecutive channels of one token
//     -> 8-byte global stores.
// Q/K/V/dO row fragments (row = token, k = channel) are 16-byte global loads in exactly the MFMA A/B layout; only the
// tiles that are needed with the token index as contraction index are staged in LDS (forward: V, 5 KB per wave instead
// of 25 KB -> 4x the resident waves).  rel_index(i,j) = A(i) - A(j) + 84 with A(t) = 13*(t/7) + t%7, so the bias lookup
// needs no division in the inner loop.
// Backward recomputes P from the saved log-sum-exp in BOTH layouts: pass 1 (S^T, lane = query i) gives delta_i, dS^T,
// dQ and the relative-position-bias gradient; pass 2 (S, lane = key j) gives dV = P^T dO and dK = dS^T Q.  The bias
// gradient is accumulated in registers across the windows a workgroup walks (deterministic two-stage reduction after
// the kernel, no float atomics).
#include "pk_common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define AT_N 49
#define AT_WS 7
// row pitch (bf16) of the [64][32*DK] row-major LDS tiles: 32*DK + 8 (40 for head_dim <= 32, 72 for head_dim <= 64)
#define RPITCH(DK) (32 * (DK) + 8)

__device__ __forceinline__ bf16x8 lds_frag(const uint16_t* base, int row, int pitch, int k0) {
    return *reinterpret_cast<const bf16x8*>(base + row * pitch + k0);
}
// All global reads are branch-free `buffer_load_dwordx4` with the out-of-range offset trick (offset >= num_records returns
// zeros): rows >= 49 and channels >= d cost no branch, so the compiler issues every load of a window back to back and
// waits once, instead of serialising one HBM latency per conditional load (that was ~25 us per window).
#define OOB_OFF 0x80000000u
#define MAKE_RSRC(ptr) __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(ptr), 0, 0x7ffffff0, 0x00020000)
// Row fragment straight from global memory: token row `row`, channels 8g .. 8g+7 (zeros outside the 49 x d slice).
#define GLB_FRAG(rsrc, ld, d, row, c0) \
    __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((row) < AT_N && (c0) < (d)) ? (unsigned)(((row) * (ld) + (c0)) * 2) : OOB_OFF, 0, 0))
// Fragment (row/column = channel col0 + l16, k = tokens in ACCUMULATOR ORDER for K-step s): slots jj = 0..3 are tokens
// 32s + 4g + jj, slots 4..7 are tokens 32s + 16 + 4g + (jj-4), read from a row-major [64][RP] token tile.
// `pitch` = RPITCH(DK).  ds_read_b64_tr_b16: within a 16-lane group lane i addresses 4 elements of row (i>>2) at columns 4(i&3).. and receives
// column i of the 4 x 16 block.
__device__ __forceinline__ bf16x8 tok_frag(const uint16_t* tile, int pitch, int s, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const uint16_t* a0 = tile + (32 * s + 4 * g + (i >> 2)) * pitch + col0 + 4 * (i & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 16 * pitch));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ int rel_a(int t) { return 13 * (t / AT_WS) + t % AT_WS; }   // rel_index(i,j) = rel_a(i) - rel_a(j) + 84
__device__ __forceinline__ bf16x8 pack_frag(const f32x4 a, const f32x4 b, float m) {
    const u32x4 v = {pack_bf16x2(a[0] * m, a[1] * m), pack_bf16x2(a[2] * m, a[3] * m), pack_bf16x2(b[0] * m, b[1] * m),
                     pack_bf16x2(b[2] * m, b[3] * m)};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ void store4_bf16(uint16_t* dst, const f32x4 v, float m) {
    uint2 pk;
    pk.x = pack_bf16x2(v[0] * m, v[1] * m);
    pk.y = pack_bf16x2(v[2] * m, v[3] * m);
    *reinterpret_cast<uint2*>(dst) = pk;
}

// Load a [49][d] slice (row stride ld) into a zero-padded [64][RPITCH(DK)] row-major LDS tile (rows >= 49 and channels >= d
// are zero): TILE_LOAD issues this lane's 4*DK 16-byte loads, TILE_STORE writes them to LDS (call after all loads are issued).
#define TILE_LOAD(v, rsrc, ld, d, lane, DK)                                                                          \
    _Pragma("unroll") for (int k_ = 0; k_ < 4 * (DK); ++k_) {                                                        \
        const int idx_ = (lane) + 64 * k_, row_ = idx_ / (4 * (DK)), ch_ = idx_ % (4 * (DK));                         \
        v[k_] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (row_ < AT_N && ch_ * 8 < (d)) ? (unsigned)((row_ * (ld) + ch_ * 8) * 2) : OOB_OFF, 0, 0); \
    }
#define TILE_STORE(v, tile, lane, DK)                                                                                \
    _Pragma("unroll") for (int k_ = 0; k_ < 4 * (DK); ++k_) {                                                        \
        const int idx_ = (lane) + 64 * k_, row_ = idx_ / (4 * (DK)), ch_ = idx_ % (4 * (DK));                         \
        *reinterpret_cast<u32x4*>((tile) + row_ * RPITCH(DK) + ch_ * 8) = v[k_];                                      \
    }

// ================================================================================================ forward
template <int DK, int NCE>   // DK: 32-channel K-steps of the head dimension (1: d <= 32, 2: d <= 64); NCE = ceil(d / 16)
__global__ void __launch_bounds__(64) k_win_attn_fwd(const uint16_t* __restrict__ qkv, const float* __restrict__ table,
                                                     uint16_t* __restrict__ out, float* __restrict__ lse, int heads, int C, int d,
                                                     float scale) {
    constexpr int RP = RPITCH(DK);
    __shared__ __attribute__((aligned(16))) uint16_t sV[64 * RP];
    __shared__ float sBias[176];
    const int lane = threadIdx.x, g4 = lane >> 4, l16 = lane & 15;
    const int w = blockIdx.x / heads, h = blockIdx.x - w * heads;
    const uint16_t* base = qkv + (size_t)w * AT_N * 3 * C + h * d;
    const auto rq = MAKE_RSRC(base), rk = MAKE_RSRC(base + C), rv = MAKE_RSRC(base + 2 * C);
    u32x4 tv[4 * DK];
    TILE_LOAD(tv, rv, 3 * C, d, lane, DK);
    bf16x8 qf[4][DK], kf[4][DK];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) {
            qf[t][ks] = GLB_FRAG(rq, 3 * C, d, 16 * t + l16, 32 * ks + 8 * g4);
            kf[t][ks] = GLB_FRAG(rk, 3 * C, d, 16 * t + l16, 32 * ks + 8 * g4);
        }
    for (int i = lane; i < 169; i += 64) sBias[i] = table[i * heads + h];
    TILE_STORE(tv, sV, lane, DK);
    int aj[4][4];                       // 84 - A(j) for this lane's 16 keys j = 16cj + 4g4 + r (-1: tile padding)
#pragma unroll
    for (int cj = 0; cj < 4; ++cj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * cj + 4 * g4 + r;
            aj[cj][r] = j < AT_N ? 84 - rel_a(j) : -1;
        }
    __syncthreads();
    bf16x8 vt[2][NCE];                  // V^T fragments [token K-step][channel tile]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ce = 0; ce < NCE; ++ce) vt[s][ce] = tok_frag(sV, RP, s, 16 * ce, lane);

    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
        const int i = 16 * ci + l16;
        const int ai = rel_a(i < AT_N ? i : 0);
        f32x4 st[4];
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) {
            st[cj] = zero;
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) st[cj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[cj][ks], qf[ci][ks], st[cj], 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int cj = 0; cj < 4; ++cj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float bv = sBias[ai + (aj[cj][r] >= 0 ? aj[cj][r] : 0)];      // unconditional load, then select: no branches
                st[cj][r] = aj[cj][r] >= 0 ? st[cj][r] * scale + bv : -INFINITY;
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
        if (g4 == 0 && i < AT_N && lse) lse[((size_t)w * heads + h) * AT_N + i] = mx + __logf(sum);
        const bf16x8 p0 = pack_frag(st[0], st[1], inv), p1 = pack_frag(st[2], st[3], inv);
#pragma unroll
        for (int ce = 0; ce < NCE; ++ce) {
            f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt[0][ce], p0, zero, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt[1][ce], p1, o, 0, 0, 0);
            const int e = 16 * ce + 4 * g4;          // this lane: channels e..e+3 of token i
            if (i < AT_N && e < d) store4_bf16(out + ((size_t)w * AT_N + i) * C + h * d + e, o, 1.f);
        }
    }
}

// head_dim d (multiple of 8, <= 64) -> kernel instantiation <DK, NCE>
#define ATTN_DISPATCH(d, LAUNCH)            \
    do {                                    \
        if ((d) <= 16) { LAUNCH(1, 1); }    \
        else if ((d) <= 32) { LAUNCH(1, 2); } \
        else if ((d) <= 48) { LAUNCH(2, 3); } \
        else { LAUNCH(2, 4); }              \
    } while (0)

extern "C" int pk_window_attn_fwd(const void* qkv, const float* rel_table, void* out, float* lse, int n_windows, int heads, int C,
                                  float softmax_scale, void* stream) {
    PK_REQUIRE(qkv && rel_table && out && n_windows > 0 && heads > 0 && C > 0, "pk_window_attn_fwd: bad argument");
    PK_REQUIRE(C % heads == 0, "pk_window_attn_fwd: C=%d not divisible by heads=%d", C, heads);
    const int d = C / heads;
    PK_SUPPORTED(d <= 64 && (d & 7) == 0, "pk_window_attn_fwd: head_dim %d (supported: multiples of 8 up to 64)", d);
    PK_REQUIRE(((((uintptr_t)qkv) | ((uintptr_t)out)) & 15) == 0 && (C & 7) == 0, "pk_window_attn_fwd: alignment");
    const float scale = softmax_scale > 0.f ? softmax_scale : 1.f / sqrtf((float)d);
#define LAUNCH_FWD(DK_, NCE_)                                                                                                          \
    hipLaunchKernelGGL((k_win_attn_fwd<DK_, NCE_>), dim3(n_windows * heads), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)qkv, \
                       rel_table, (uint16_t*)out, lse, heads, C, d, scale)
    ATTN_DISPATCH(d, LAUNCH_FWD);
#undef LAUNCH_FWD
    return pk_launch_status("pk_window_attn_fwd");
}

// ================================================================================================ backward
// Workgroup g walks windows w = g/heads, g/heads + stride, ... of head h = g % heads; dS is summed in registers and
// written once to dbias_part[g][49*49].
// delta_i = sum_j P_ij dP_ij is taken from the identity delta_i = sum_e dO[i][e] * O[i][e] (O = forward output), so no
// quantity depends on a whole score row any more: both passes decompose into independent 32-key (pass 1) / 32-query
// (pass 2) chunks of two accumulator tiles, which keeps ~130 registers live instead of > 256 and lets three waves share a
// SIMD.  P is recomputed from the saved log-sum-exp in each layout.
template <int DK, int NCE>   // as in the forward kernel
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DK == 1 ? 2 : 1, DK == 1 ? 2 : 1))) k_win_attn_bwd(const uint16_t* __restrict__ qkv, const float* __restrict__ table,
                                                     const uint16_t* __restrict__ fwd_out, const uint16_t* __restrict__ dout,
                                                     const float* __restrict__ lse, uint16_t* __restrict__ dqkv,
                                                     float* __restrict__ dbias_part, int n_windows, int heads, int C, int d, float scale,
                                                     int wstride) {
    constexpr int RP = RPITCH(DK);
    __shared__ __attribute__((aligned(16))) uint16_t sTiles[3 * 64 * RP];      // Q, K, dO tiles; reused for the bias-gradient fold
    __shared__ __attribute__((aligned(16))) float sLse[64], sDelta[64];
    __shared__ float sBias[176];
    uint16_t *sQ = sTiles, *sK = sTiles + 64 * RP, *sdO = sTiles + 2 * 64 * RP;
    const int lane0 = threadIdx.x;
    const int h = blockIdx.x % heads;
    // Score arithmetic (as in pk_block.hip's k_attn_bwd): the bias table is kept in units of log2 and the score / dP accumulators START
    // at -lse / scale and -delta in pass 2 (pass 1 subtracts: no registers to spare), so P = exp2(fma(acc, scale * log2 e, bias')) and
    // dS = P * acc'; the clamps of the bias index exist only on the edge tiles (tiles 0..2 hold only real tokens).
    constexpr float LOG2E = 1.44269504088896340736f;
    const float scale2 = scale * LOG2E, inv_scale = 1.f / scale;
    for (int i = lane0; i < 169; i += 64) sBias[i] = table[i * heads + h] * LOG2E;
    int a4[4][4];                       // A(t) of the 16 tokens t = 16c + 4g4 + r this lane owns along accumulator rows (-1: padding)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = 16 * c + 4 * (lane0 >> 4) + r;
            a4[c][r] = t < AT_N ? rel_a(t) : -1;
        }
    f32x4 dsum[4][4];                   // [ci][cj][r]: d(bias) at (i = 16ci + l16, j = 16cj + 4g4 + r)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) dsum[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    for (int w = blockIdx.x / heads; w < n_windows; w += wstride) {
        __syncthreads();   // previous iteration's LDS reads are done
        const uint16_t* base = qkv + (size_t)w * AT_N * 3 * C + h * d;
        const uint16_t* dob = dout + (size_t)w * AT_N * C + h * d;
        const auto rq = MAKE_RSRC(base), rk = MAKE_RSRC(base + C), rv = MAKE_RSRC(base + 2 * C), rg = MAKE_RSRC(dob);
        const auto ro = MAKE_RSRC(fwd_out + (size_t)w * AT_N * C + h * d);
        u32x4 tq[4 * DK], tk[4 * DK], tg[4 * DK], orow[4 * DK], grow[4 * DK];
        TILE_LOAD(tq, rq, 3 * C, d, lane0, DK);
        TILE_LOAD(tk, rk, 3 * C, d, lane0, DK);
        TILE_LOAD(tg, rg, C, d, lane0, DK);
        bf16x8 vf[4][DK];               // V row fragments (token rows, k = channel) straight from global
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) vf[t][ks] = GLB_FRAG(rv, 3 * C, d, 16 * t + (lane0 & 15), 32 * ks + 8 * (lane0 >> 4));
#pragma unroll
        for (int c = 0; c < 4 * DK; ++c) {   // token `lane0`: its rows of O and dO for delta = sum_e dO*O
            const unsigned off = (lane0 < AT_N && c * 8 < d) ? (unsigned)((lane0 * C + c * 8) * 2) : OOB_OFF;
            orow[c] = __builtin_amdgcn_raw_buffer_load_b128(ro, off, 0, 0);
            grow[c] = __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0);
        }
        const float l = lane0 < AT_N ? lse[((size_t)w * heads + h) * AT_N + lane0] : 0.f;
        TILE_STORE(tq, sQ, lane0, DK);
        TILE_STORE(tk, sK, lane0, DK);
        TILE_STORE(tg, sdO, lane0, DK);
        {
            float de = 0.f;
#pragma unroll
            for (int c = 0; c < 4 * DK; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    de += __uint_as_float(orow[c][q] << 16) * __uint_as_float(grow[c][q] << 16);
                    de += __uint_as_float(orow[c][q] & 0xffff0000u) * __uint_as_float(grow[c][q] & 0xffff0000u);
                }
            sLse[lane0] = -l * inv_scale;
            sDelta[lane0] = -de;
        }
        __syncthreads();
        uint16_t* dq = dqkv + (size_t)w * AT_N * 3 * C + h * d;
        // The bias table in LDS and the lane's (i, j) pairs do not change from window to window, so LICM would hoist all
        // 512 bias lookups out of this loop into registers (it did: 256 VGPRs + 108 AGPRs).  An opaque zero added to the
        // index keeps the lookups inside the loop.
        int lz = 0;
        asm volatile("" : "+v"(lz));
        const int lane = lane0 + lz, g4 = lane >> 4, l16 = lane & 15;      // (same trick for the per-lane LDS addresses)

        // ---- pass 1: transposed scores, lane = query i:  dS^T, bias gradient, dQ^T = scale * K^T dS^T
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
            const int i = 16 * ci + l16;
            const bool iok = ci < 3 || l16 == 0;
            const int ai = rel_a(iok ? i : 0) + 84 + lz;
            // (pass 1 keeps plain subtractions: eight more live registers for the accumulator-start form put this kernel over 256 -- 39 spilled)
            const float li = -sLse[i] * scale2, di = -sDelta[i];
            bf16x8 qfi[DK], ofi[DK];
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) {
                qfi[ks] = lds_frag(sQ, i, RP, 32 * ks + g4 * 8);
                ofi[ks] = lds_frag(sdO, i, RP, 32 * ks + g4 * 8);
            }
            f32x4 acc[NCE];
#pragma unroll
            for (int ce = 0; ce < NCE; ++ce) acc[ce] = zero;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4 ds[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int cj = 2 * s + u;
                    f32x4 sc = zero, dp = zero;
#pragma unroll
                    for (int ks = 0; ks < DK; ++ks) {
                        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(sK, 16 * cj + l16, RP, 32 * ks + g4 * 8), qfi[ks], sc, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[cj][ks], ofi[ks], dp, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool jv = a4[cj][r] >= 0;        // (the tile-index form of this test costs 37 spilled registers here)
                        const float ev = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], scale2, sBias[ai - (cj < 3 ? a4[cj][r] : max(a4[cj][r], 0))]) - li);
                        const float pv = (iok && jv) ? ev : 0.f;      // select after the fact: straight-line code
                        ds[u][r] = pv * (dp[r] - di);
                        dsum[ci][cj][r] += ds[u][r];
                    }
                }
                const bf16x8 df = pack_frag(ds[0], ds[1], 1.f);
#pragma unroll
                for (int ce = 0; ce < NCE; ++ce)
                    acc[ce] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tok_frag(sK, RP, s, 16 * ce, lane), df, acc[ce], 0, 0, 0);
            }
#pragma unroll
            for (int ce = 0; ce < NCE; ++ce) {
                const int e = 16 * ce + 4 * g4;
                if (iok && e < d) store4_bf16(dq + (size_t)i * 3 * C + e, acc[ce], scale);
            }
        }

        // ---- pass 2: plain scores, lane = key j:  dV^T = dO^T P,  dK^T = scale * Q^T dS
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) {
            const int j = 16 * cj + l16;
            const bool jok = cj < 3 || l16 == 0;
            const int ajn = 84 - rel_a(jok ? j : 0) + lz;
            bf16x8 kfj[DK];
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) kfj[ks] = lds_frag(sK, j, RP, 32 * ks + g4 * 8);
            f32x4 dv[NCE], dk[NCE];
#pragma unroll
            for (int ce = 0; ce < NCE; ++ce) dv[ce] = dk[ce] = zero;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4 pp[2], ds[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int ci = 2 * s + u;
                    f32x4 sc = *reinterpret_cast<const f32x4*>(&sLse[16 * ci + 4 * g4]);
                    f32x4 dp = *reinterpret_cast<const f32x4*>(&sDelta[16 * ci + 4 * g4]);
#pragma unroll
                    for (int ks = 0; ks < DK; ++ks) {
                        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(sQ, 16 * ci + l16, RP, 32 * ks + g4 * 8), kfj[ks], sc, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(sdO, 16 * ci + l16, RP, 32 * ks + g4 * 8), vf[cj][ks], dp, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool iv = a4[ci][r] >= 0;
                        const float ev = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[r], scale2, sBias[(ci < 3 ? a4[ci][r] : max(a4[ci][r], 0)) + ajn]));
                        const float pv = (jok && iv) ? ev : 0.f;
                        pp[u][r] = pv;
                        ds[u][r] = pv * dp[r];
                    }
                }
                const bf16x8 pf = pack_frag(pp[0], pp[1], 1.f), df = pack_frag(ds[0], ds[1], 1.f);
#pragma unroll
                for (int ce = 0; ce < NCE; ++ce) {
                    dv[ce] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tok_frag(sdO, RP, s, 16 * ce, lane), pf, dv[ce], 0, 0, 0);
                    dk[ce] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tok_frag(sQ, RP, s, 16 * ce, lane), df, dk[ce], 0, 0, 0);
                }
            }
#pragma unroll
            for (int ce = 0; ce < NCE; ++ce) {
                const int e = 16 * ce + 4 * g4;
                if (jok && e < d) {
                    store4_bf16(dq + (size_t)j * 3 * C + C + e, dk[ce], scale);
                    store4_bf16(dq + (size_t)j * 3 * C + 2 * C + e, dv[ce], 1.f);
                }
            }
        }
    }
    // Fold the register-resident [49][49] bias gradient onto the 169 table entries inside the workgroup (fixed summation
    // order -> deterministic): the partial result per workgroup is 169 floats instead of 2401.
    __syncthreads();
    // (the tile sits 320 floats into the buffer: the fold below also reads -- and masks -- words up to 294 before and 293 after it)
    static_assert((320 + 6 * 7 * AT_N + 12 * AT_N + AT_N) * 4 <= 3 * 64 * RP * 2, "bias-gradient tile + slack must fit the Q/K/dO tiles");
    float* sD = reinterpret_cast<float*>(sTiles) + 320;      // 49*49*4 = 9604 B at +1280 B <= 3*64*RP*2 = 15360 B
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int cj = 0; cj < 4; ++cj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ci + (lane0 & 15), j = 16 * cj + 4 * (lane0 >> 4) + r;
                if (i < AT_N && j < AT_N) sD[i * AT_N + j] = dsum[ci][cj][r];
            }
    __syncthreads();
    // (rows yj in a loop, the seven xj of a row unrolled under bit masks: the wave pays for its longest lane -- 49 trips of a two-level
    // loop per round before, 5 + 7 + 4 row trips over the three rounds now; the terms are added in the same order)
    for (int e = lane0; e < 169; e += 64) {
        const int dy = e / 13 - 6, dx = e % 13 - 6;
        const int x0 = max(0, -dx), x1 = min(AT_WS, AT_WS - dx);
        unsigned keep[AT_WS];
#pragma unroll
        for (int xj = 0; xj < AT_WS; ++xj) keep[xj] = (xj >= x0 && xj < x1) ? 0xffffffffu : 0u;
        const int base = (dy * AT_WS + dx) * AT_N;                       // + yj * (7 * 49 + 7) + xj * (49 + 1)
        float acc = 0.f;
        for (int yj = max(0, -dy); yj < min(AT_WS, AT_WS - dy); ++yj) {
#pragma unroll
            for (int xj = 0; xj < AT_WS; ++xj) {
                // (an xj outside [x0, x1) addresses some other word of the tile buffer or just outside it: read, masked to +0)
                const float v = sD[base + yj * (AT_WS * AT_N + AT_WS) + xj * (AT_N + 1)];
                acc += __uint_as_float(__float_as_uint(v) & keep[xj]);
            }
        }
        dbias_part[(size_t)blockIdx.x * 169 + e] = acc;
    }
}

// Relative-position-bias gradient, second stage: dtable[e][h] = sum over the workgroups g of head h (g % heads == h) of
// part[g][e].  One 256-thread block per table entry: thread t adds groups t, t+256, ... in order, then a fixed-shape LDS
// tree combines the 256 partial sums (deterministic; the 11-block version with 90 dependent loads per thread took 31 us).
__global__ void __launch_bounds__(256) k_relbias_reduce(const float* __restrict__ part, int n_groups, int heads, float* __restrict__ dtable) {
    __shared__ float sh[256];
    const int e = blockIdx.x, h = blockIdx.y, t = threadIdx.x;
    const int per_head = n_groups / heads;
    float s = 0.f;
    for (int k = t; k < per_head; k += 256) s += part[(size_t)(k * heads + h) * 169 + e];
    sh[t] = s;
    __syncthreads();
#pragma unroll
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) sh[t] += sh[t + w];
        __syncthreads();
    }
    if (t == 0) dtable[e * heads + h] = sh[0];
}

extern "C" int pk_window_attn_bwd_groups(int n_windows, int heads) {
    if (n_windows <= 0 || heads <= 0) return 0;
    // One 64-lane workgroup per (window-group, head).  The kernel holds 2 waves per SIMD (2048 on the chip): give every
    // workgroup ceil(total / 2048) windows so the whole problem is resident at once and no workgroup walks more than that.
    const long total = (long)n_windows * heads;
    static const int slots = PK_KNOB("PK_WATTN_BWD_SLOTS", 2048);
    const int wpg = (int)((total + slots - 1) / slots);
    const int per_head = (n_windows + wpg - 1) / wpg;
    return per_head * heads;
}
extern "C" int pk_window_attn_bwd_ws_floats(int n_windows, int heads) { return pk_window_attn_bwd_groups(n_windows, heads) * 169; }
extern "C" int pk_window_attn_bwd(const void* qkv, const float* rel_table, const void* fwd_out, const void* dout, const float* lse,
                                  void* dqkv, float* dbias_partial, float* dtable, int n_windows, int heads, int C, float softmax_scale,
                                  void* stream) {
    PK_REQUIRE(qkv && rel_table && fwd_out && dout && lse && dqkv && dbias_partial, "pk_window_attn_bwd: null pointer");
    PK_REQUIRE(((((uintptr_t)qkv) | ((uintptr_t)fwd_out) | ((uintptr_t)dout) | ((uintptr_t)dqkv)) & 15) == 0 && (C & 7) == 0,
               "pk_window_attn_bwd: alignment");
    PK_REQUIRE(n_windows > 0 && heads > 0 && C > 0 && C % heads == 0, "pk_window_attn_bwd: bad sizes");
    const int d = C / heads;
    PK_SUPPORTED(d <= 64 && (d & 7) == 0, "pk_window_attn_bwd: head_dim %d (supported: multiples of 8 up to 64)", d);
    const float scale = softmax_scale > 0.f ? softmax_scale : 1.f / sqrtf((float)d);
    const int groups = pk_window_attn_bwd_groups(n_windows, heads);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_BWD(DK_, NCE_)                                                                                                     \
    hipLaunchKernelGGL((k_win_attn_bwd<DK_, NCE_>), dim3(groups), dim3(64), 0, st, (const uint16_t*)qkv, rel_table,              \
                       (const uint16_t*)fwd_out, (const uint16_t*)dout, lse, (uint16_t*)dqkv, dbias_partial, n_windows, heads, C, d, \
                       scale, groups / heads)
    ATTN_DISPATCH(d, LAUNCH_BWD);
#undef LAUNCH_BWD
    if (dtable)      // NULL: partials only ([groups][169], group g belongs to head g % heads), reduced later by pk_reduce_many
        hipLaunchKernelGGL(k_relbias_reduce, dim3(169, heads), dim3(256), 0, st, dbias_partial, groups, heads, dtable);
    return pk_launch_status("pk_window_attn_bwd");
}
