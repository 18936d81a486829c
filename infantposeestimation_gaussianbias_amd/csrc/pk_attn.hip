// Window attention core of the HRFormer block (A1): per (window, head)
//     P = softmax(scale * Q K^T + relbias[h]),  O = P V            N = 49 tokens (7x7), head_dim d <= 32
// One 64-lane wave per (window, head): Q/K/V tiles live in LDS, the 49x49 problem is padded to 64x64 MFMA tiles
// (v_mfma_f32_16x16x32_bf16) and ONLY tile padding is masked (-inf): the reference's zero-pad tokens are real
// tokens (q=b_q, k=b_k, v=b_v) and take softmax mass (hrformer.py:80-84, no mask).  Softmax reductions run on
// the accumulator layout with 16-lane xor shuffles.  Backward recomputes P from the saved log-sum-exp and
// accumulates the relative-position-bias gradient in registers across the windows a workgroup walks
// (deterministic two-stage reduction, no float atomics).
#include "pk_common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define AT_N 49
#define AT_WS 7
#define RP 40   // row pitch (bf16) of [64][32] row-major tiles
#define TP 72   // row pitch (bf16) of [*][64] tiles indexed by token

__device__ __forceinline__ bf16x8 lds_frag(const uint16_t* base, int row, int pitch, int k0) {
    return *reinterpret_cast<const bf16x8*>(base + row * pitch + k0);
}
typedef __attribute__((ext_vector_type(4))) short s16x4;
// Fragment whose 8 k-values run down the ROWS of a row-major LDS tile (k0 .. k0+31 = rows, col0 + lane&15 = column):
// two gfx950 transpose reads (ds_read_b64_tr_b16: 4 rows x 16 columns per 16-lane group, delivered column-major).
__device__ __forceinline__ bf16x8 tr_frag(const uint16_t* tile, int pitch, int k0, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pq = i & 3;
    const uint16_t* a0 = tile + (k0 + 8 * g + q) * pitch + col0 + 4 * pq;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * pitch));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ int rel_index(int i, int j) {
    const int yi = i / AT_WS, xi = i - yi * AT_WS, yj = j / AT_WS, xj = j - yj * AT_WS;
    return (yi - yj + AT_WS - 1) * (2 * AT_WS - 1) + (xi - xj + AT_WS - 1);
}

// Load a [49][d] slice (row stride ld) into a zero-padded [64][RP] row-major LDS tile (rows >= 49 and channels >= d are zero).
__device__ __forceinline__ void stage_tile(const uint16_t* __restrict__ g, int ld, int d, uint16_t* rowmajor, int lane) {
    for (int idx = lane; idx < 64 * 4; idx += 64) {
        const int row = idx >> 2, ch = idx & 3;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < AT_N && ch * 8 < d) v = *reinterpret_cast<const uint4*>(g + (size_t)row * ld + ch * 8);
        *reinterpret_cast<uint4*>(rowmajor + row * RP + ch * 8) = v;
    }
}

// ================================================================================================ forward
__global__ void __launch_bounds__(64) k_win_attn_fwd(const uint16_t* __restrict__ qkv, const float* __restrict__ table,
                                                     uint16_t* __restrict__ out, float* __restrict__ lse, int heads, int C, int d,
                                                     float scale) {
    __shared__ __attribute__((aligned(16))) uint16_t sQ[64 * RP], sK[64 * RP], sV[64 * RP], sP[64 * TP];
    __shared__ float sBias[176];
    const int lane = threadIdx.x, g4 = lane >> 4, l16 = lane & 15;
    const int w = blockIdx.x / heads, h = blockIdx.x - w * heads;
    const uint16_t* base = qkv + (size_t)w * AT_N * 3 * C + h * d;
    stage_tile(base, 3 * C, d, sQ, lane);
    stage_tile(base + C, 3 * C, d, sK, lane);
    stage_tile(base + 2 * C, 3 * C, d, sV, lane);
    for (int i = lane; i < 169; i += 64) sBias[i] = table[i * heads + h];
    __syncthreads();

    f32x4 s[4][4];
    bf16x8 qf[4], kf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        qf[t] = lds_frag(sQ, 16 * t + l16, RP, g4 * 8);
        kf[t] = lds_frag(sK, 16 * t + l16, RP, g4 * 8);
    }
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int cj = 0; cj < 4; ++cj)
            s[ci][cj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ci], kf[cj], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);

    // softmax over j for the rows i = 16ci + 4*g4 + r this lane shares with its 16-lane group
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * ci + 4 * g4 + r;
            const int ic = i < AT_N ? i : 0;
            float v[4], mx = -INFINITY;
#pragma unroll
            for (int cj = 0; cj < 4; ++cj) {
                const int j = 16 * cj + l16;
                v[cj] = (j < AT_N) ? s[ci][cj][r] * scale + sBias[rel_index(ic, j)] : -INFINITY;
                mx = fmaxf(mx, v[cj]);
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            float sum = 0.f;
#pragma unroll
            for (int cj = 0; cj < 4; ++cj) {
                v[cj] = __expf(v[cj] - mx);
                sum += v[cj];
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
            const float inv = 1.f / sum;
#pragma unroll
            for (int cj = 0; cj < 4; ++cj) sP[i * TP + 16 * cj + l16] = f32_to_bf16(v[cj] * inv);
            if (l16 == 0 && i < AT_N && lse) lse[((size_t)w * heads + h) * AT_N + i] = mx + __logf(sum);
        }
    }
    __syncthreads();

    // O = P V : A = P rows i (k = j); B[k=j][col=e] = V[j][e] read with the transpose read from the row-major V tile
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
        const bf16x8 p0 = lds_frag(sP, 16 * ci + l16, TP, g4 * 8), p1 = lds_frag(sP, 16 * ci + l16, TP, 32 + g4 * 8);
#pragma unroll
        for (int ce = 0; ce < 2; ++ce) {
            if (ce * 16 >= d) break;
            f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p0, tr_frag(sV, RP, 0, 16 * ce, lane), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p1, tr_frag(sV, RP, 32, 16 * ce, lane), o, 0, 0, 0);
            const int e = 16 * ce + l16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ci + 4 * g4 + r;
                if (i < AT_N && e < d) out[((size_t)w * AT_N + i) * C + h * d + e] = f32_to_bf16(o[r]);
            }
        }
    }
}

extern "C" int pk_window_attn_fwd(const void* qkv, const float* rel_table, void* out, float* lse, int n_windows, int heads, int C,
                                  void* stream) {
    PK_REQUIRE(qkv && rel_table && out && n_windows > 0 && heads > 0 && C > 0, "pk_window_attn_fwd: bad argument");
    PK_REQUIRE(C % heads == 0, "pk_window_attn_fwd: C=%d not divisible by heads=%d", C, heads);
    const int d = C / heads;
    PK_SUPPORTED(d <= 32 && (d & 7) == 0, "pk_window_attn_fwd: head_dim %d (supported: multiples of 8 up to 32)", d);
    PK_REQUIRE((((uintptr_t)qkv) & 15) == 0 && (C & 7) == 0, "pk_window_attn_fwd: alignment");
    hipLaunchKernelGGL(k_win_attn_fwd, dim3(n_windows * heads), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)qkv, rel_table,
                       (uint16_t*)out, lse, heads, C, d, 1.f / sqrtf((float)d));
    return pk_launch_status("pk_window_attn_fwd");
}

// ================================================================================================ backward
// Workgroup g walks windows w = g/heads, g/heads + stride, ... of head h = g % heads; dS is summed in registers and
// written once to dbias_part[g][49*49].
__global__ void __launch_bounds__(64) k_win_attn_bwd(const uint16_t* __restrict__ qkv, const float* __restrict__ table,
                                                     const uint16_t* __restrict__ dout, const float* __restrict__ lse,
                                                     uint16_t* __restrict__ dqkv, float* __restrict__ dbias_part, int n_windows, int heads,
                                                     int C, int d, float scale, int wstride) {
    __shared__ __attribute__((aligned(16))) uint16_t sQ[64 * RP], sK[64 * RP], sV[64 * RP], sdO[64 * RP];
    __shared__ __attribute__((aligned(16))) uint16_t sP[64 * TP], sdS[64 * TP];     // row-major [i][j]; transposes come from tr reads
    __shared__ float sBias[176];
    const int lane = threadIdx.x, g4 = lane >> 4, l16 = lane & 15;
    const int h = blockIdx.x % heads;
    for (int i = lane; i < 169; i += 64) sBias[i] = table[i * heads + h];
    f32x4 dsum[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) dsum[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int w = blockIdx.x / heads; w < n_windows; w += wstride) {
        __syncthreads();   // previous iteration's LDS reads are done
        const uint16_t* base = qkv + (size_t)w * AT_N * 3 * C + h * d;
        stage_tile(base, 3 * C, d, sQ, lane);
        stage_tile(base + C, 3 * C, d, sK, lane);
        stage_tile(base + 2 * C, 3 * C, d, sV, lane);
        stage_tile(dout + (size_t)w * AT_N * C + h * d, C, d, sdO, lane);
        __syncthreads();

        bf16x8 qf[4], kf[4], vf[4], of[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            qf[t] = lds_frag(sQ, 16 * t + l16, RP, g4 * 8);
            kf[t] = lds_frag(sK, 16 * t + l16, RP, g4 * 8);
            vf[t] = lds_frag(sV, 16 * t + l16, RP, g4 * 8);
            of[t] = lds_frag(sdO, 16 * t + l16, RP, g4 * 8);
        }
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
            f32x4 s[4], dp[4];
#pragma unroll
            for (int cj = 0; cj < 4; ++cj) {
                s[cj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ci], kf[cj], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                dp[cj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(of[ci], vf[cj], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ci + 4 * g4 + r;
                const bool iok = i < AT_N;
                const float l = iok ? lse[((size_t)w * heads + h) * AT_N + i] : 0.f;
                float p[4], delta = 0.f;
#pragma unroll
                for (int cj = 0; cj < 4; ++cj) {
                    const int j = 16 * cj + l16;
                    p[cj] = (iok && j < AT_N) ? __expf(s[cj][r] * scale + sBias[rel_index(i, j)] - l) : 0.f;
                    delta += p[cj] * dp[cj][r];
                }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) delta += __shfl_xor(delta, o, 64);
#pragma unroll
                for (int cj = 0; cj < 4; ++cj) {
                    const int j = 16 * cj + l16;
                    const float ds = p[cj] * (dp[cj][r] - delta);
                    dsum[ci][cj][r] += ds;
                    sP[i * TP + j] = f32_to_bf16(p[cj]);
                    sdS[i * TP + j] = f32_to_bf16(ds);
                }
            }
        }
        __syncthreads();
        uint16_t* dq = dqkv + (size_t)w * AT_N * 3 * C + h * d;
        // dV = P^T dO, dK = scale * dS^T Q (output rows j: A operands are transpose reads of the row-major P / dS tiles);
        // dQ = scale * dS K (output rows i: plain row fragments of dS).  B operands Q / K / dO: transpose reads (k runs down rows).
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const bf16x8 pt0 = tr_frag(sP, TP, 0, 16 * ct, lane), pt1 = tr_frag(sP, TP, 32, 16 * ct, lane);
            const bf16x8 st0 = tr_frag(sdS, TP, 0, 16 * ct, lane), st1 = tr_frag(sdS, TP, 32, 16 * ct, lane);
            const bf16x8 ds0 = lds_frag(sdS, 16 * ct + l16, TP, g4 * 8), ds1 = lds_frag(sdS, 16 * ct + l16, TP, 32 + g4 * 8);
#pragma unroll
            for (int ce = 0; ce < 2; ++ce) {
                if (ce * 16 >= d) break;
                const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                f32x4 dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pt0, tr_frag(sdO, RP, 0, 16 * ce, lane), z, 0, 0, 0);
                dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pt1, tr_frag(sdO, RP, 32, 16 * ce, lane), dv, 0, 0, 0);
                f32x4 dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st0, tr_frag(sQ, RP, 0, 16 * ce, lane), z, 0, 0, 0);
                dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st1, tr_frag(sQ, RP, 32, 16 * ce, lane), dk, 0, 0, 0);
                f32x4 dqa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ds0, tr_frag(sK, RP, 0, 16 * ce, lane), z, 0, 0, 0);
                dqa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ds1, tr_frag(sK, RP, 32, 16 * ce, lane), dqa, 0, 0, 0);
                const int e = 16 * ce + l16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = 16 * ct + 4 * g4 + r;
                    if (t < AT_N && e < d) {
                        uint16_t* row = dq + (size_t)t * 3 * C + e;
                        row[0] = f32_to_bf16(dqa[r] * scale);
                        row[C] = f32_to_bf16(dk[r] * scale);
                        row[2 * C] = f32_to_bf16(dv[r]);
                    }
                }
            }
        }
    }
    float* dst = dbias_part + (size_t)blockIdx.x * AT_N * AT_N;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int cj = 0; cj < 4; ++cj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ci + 4 * g4 + r, j = 16 * cj + l16;
                if (i < AT_N && j < AT_N) dst[i * AT_N + j] = dsum[ci][cj][r];
            }
}

// Relative-position-bias gradient, two deterministic stages:
//   A) tmp[h][i*49+j] = sum over the workgroups of head h of their register-accumulated dS (coalesced over ij)
//   B) dtable[e][h]   = sum over the <= 49 (i,j) pairs with rel_index(i,j) == e
__global__ void __launch_bounds__(256) k_relbias_reduce(const float* __restrict__ part, int n_groups, int heads, float* __restrict__ tmp) {
    // block = 16 (i,j) entries x 16 group-lanes; fixed-order combine (deterministic), chain length n_groups/(16*heads)
    __shared__ float sh[16][17];
    const int col = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int ij = blockIdx.x * 16 + col, h = blockIdx.y;
    float s = 0.f;
    if (ij < AT_N * AT_N)
        for (int g = h + rl * heads; g < n_groups; g += 16 * heads) s += part[(size_t)g * AT_N * AT_N + ij];
    sh[rl][col] = s;
    __syncthreads();
    if (rl != 0 || ij >= AT_N * AT_N) return;
    s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += sh[r][col];
    tmp[(size_t)h * AT_N * AT_N + ij] = s;
}
__global__ void __launch_bounds__(256) k_relbias_scatter(const float* __restrict__ tmp, int heads, float* __restrict__ dtable) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 169 * heads) return;
    const int e = t / heads, h = t - e * heads;
    const int dy = e / 13 - 6, dx = e % 13 - 6;
    const float* p = tmp + (size_t)h * AT_N * AT_N;
    float s = 0.f;
    for (int yj = 0; yj < AT_WS; ++yj) {
        const int yi = yj + dy;
        if (yi < 0 || yi >= AT_WS) continue;
        for (int xj = 0; xj < AT_WS; ++xj) {
            const int xi = xj + dx;
            if (xi < 0 || xi >= AT_WS) continue;
            s += p[(yi * AT_WS + xi) * AT_N + yj * AT_WS + xj];
        }
    }
    dtable[t] = s;
}

extern "C" int pk_window_attn_bwd_groups(int n_windows, int heads) {
    // one 64-lane workgroup per (window-group, head); aim for >= 1024 workgroups so every CU holds several waves
    int per_head = (1024 + heads - 1) / heads;
    if (per_head < 64) per_head = 64;
    if (per_head > n_windows) per_head = n_windows;
    return per_head * heads;
}
extern "C" int pk_window_attn_bwd_ws_floats(int n_windows, int heads) {
    return (pk_window_attn_bwd_groups(n_windows, heads) + heads) * AT_N * AT_N;
}
extern "C" int pk_window_attn_bwd(const void* qkv, const float* rel_table, const void* dout, const float* lse, void* dqkv,
                                  float* dbias_partial, float* dtable, int n_windows, int heads, int C, void* stream) {
    PK_REQUIRE(qkv && rel_table && dout && lse && dqkv && dbias_partial && dtable, "pk_window_attn_bwd: null pointer");
    PK_REQUIRE(n_windows > 0 && heads > 0 && C > 0 && C % heads == 0, "pk_window_attn_bwd: bad sizes");
    const int d = C / heads;
    PK_SUPPORTED(d <= 32 && (d & 7) == 0, "pk_window_attn_bwd: head_dim %d", d);
    const int groups = pk_window_attn_bwd_groups(n_windows, heads);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_win_attn_bwd, dim3(groups), dim3(64), 0, st, (const uint16_t*)qkv, rel_table, (const uint16_t*)dout, lse,
                       (uint16_t*)dqkv, dbias_partial, n_windows, heads, C, d, 1.f / sqrtf((float)d), groups / heads);
    // stage-A output reuses the tail of the partial buffer (caller sizes it for groups + heads tiles)
    float* tmp = dbias_partial + (size_t)groups * AT_N * AT_N;
    hipLaunchKernelGGL(k_relbias_reduce, dim3((AT_N * AT_N + 15) / 16, heads), dim3(256), 0, st, dbias_partial, groups, heads, tmp);
    hipLaunchKernelGGL(k_relbias_scatter, dim3((169 * heads + 255) / 256), dim3(256), 0, st, tmp, heads, dtable);
    return pk_launch_status("pk_window_attn_bwd");
}
