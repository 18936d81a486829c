// HBM-bound kernels around the MFMA contractions: BatchNorm (train statistics finalize / apply / backward),
// LayerNorm (forward / backward), exchange-unit sum with bilinear up-sampling, input layout conversion, weight
// packing.  All operate on NHWC bf16 rows of C channels, 8 channels (16 bytes) per lane.
#include "pk_common.h"

// floor(q / d) for 0 <= q < 2^24: float estimate + one correction (~8 VALU; a 32-bit division by a run-time divisor is ~35)
__device__ __forceinline__ uint32_t fdiv24(uint32_t q, uint32_t d, float inv) {
    int r = (int)((float)q * inv);
    const int rem = (int)q - r * (int)d;
    r += rem >= (int)d ? 1 : (rem < 0 ? -1 : 0);
    return (uint32_t)r;
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(w[i] << 16);
        f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack_bf16x2(f[0], f[1]);
    v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]);
    v.w = pack_bf16x2(f[6], f[7]);
    return v;
}

// ReLU masks as BITS (round 4): the forward BatchNorm + ReLU kernels can emit one byte per 16-byte chunk (bit j: channel j of the chunk is
// > 0 after the ReLU, taken from the bf16 value that is stored), and the backward kernels read that byte instead of the 16 bytes of the
// activated output: 2 of the 7 tensor passes of a ReLU layer's BatchNorm backward (dy, y, raw read twice; dx written) become 1/16 as large.
__device__ __forceinline__ uint32_t pos_mask8(const uint4& v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t lo = w[i] & 0xffffu, hi = w[i] >> 16;
        m |= ((lo != 0u && !(lo & 0x8000u)) ? 1u : 0u) << (2 * i);
        m |= ((hi != 0u && !(hi & 0x8000u)) ? 1u : 0u) << (2 * i + 1);
    }
    return m;
}

// ================================================================================================ generic partial sums
// out[k] (+)= scale * sum_b partial[b*stride + k], k < K.  Block = 16 columns x 64 row-lanes: lane r sums rows b = r (mod 64),
// the 64 lane sums are combined in fixed order -> deterministic, and the serial chain is nb/64 instead of nb (these kernels
// sit between two passes of BatchNorm backward, i.e. on the critical path, with only K/16 workgroups).
#define RED_LANES 64
__global__ void __launch_bounds__(16 * RED_LANES) k_sum_partials(const float* __restrict__ part, int nb, int K, int stride, float* __restrict__ out,
                                                      float scale, int accumulate, float* __restrict__ out_lo = nullptr,
                                                      float* __restrict__ out_hi = nullptr, int split = 0) {
    __shared__ double sh[RED_LANES][17];
    const int col = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int k = blockIdx.x * 16 + col;
    double s = 0.0;
    if (k < K) {
        // four independent loads per round: the serial chain is memory latency x nb / 256, not x nb / 64
        int b = rl;
        for (; b + 3 * RED_LANES < nb; b += 4 * RED_LANES) {
            const float v0 = part[(size_t)b * stride + k], v1 = part[(size_t)(b + RED_LANES) * stride + k];
            const float v2 = part[(size_t)(b + 2 * RED_LANES) * stride + k], v3 = part[(size_t)(b + 3 * RED_LANES) * stride + k];
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; b < nb; b += RED_LANES) s += (double)part[(size_t)b * stride + k];
    }
    sh[rl][col] = s;
    // fixed binary tree over the 64 lane sums (a single thread adding them one after the other was 64 dependent LDS round trips:
    // ~2.5 us of a 7 us kernel that sits between two passes of BatchNorm backward)
#pragma unroll
    for (int half = RED_LANES / 2; half >= 1; half >>= 1) {
        __syncthreads();
        if (rl < half) sh[rl][col] += sh[rl + half][col];
    }
    if (rl == 0 && k < K) {
        const double t = sh[0][col];
        const float v = (float)t * scale;
        if (out) out[k] = accumulate ? out[k] + v : v;
        if (out_lo && k < split) out_lo[k] = v;            // optional second copy, split into two destinations
        if (out_hi && k >= split) out_hi[k - split] = v;
    }
}
#define SUM_PARTIALS_GRID(K) dim3(((K) + 15) / 16)
extern "C" int pk_sum_partials(const float* partial, int nb, int K, int stride, float* out, float scale, int accumulate, void* stream) {
    PK_REQUIRE(partial && out && nb > 0 && K > 0 && stride >= K, "pk_sum_partials: bad argument");
    hipLaunchKernelGGL(k_sum_partials, SUM_PARTIALS_GRID(K), dim3(16 * RED_LANES), 0, (hipStream_t)stream, partial, nb, K, stride, out, scale, accumulate);
    return pk_launch_status("pk_sum_partials");
}

// ================================================================================================ BatchNorm (train)
// Finalize the per-tile column sums written by the conv epilogue: batch mean / biased var -> scale, shift for the
// apply kernel, saved mean / rstd for backward, running stats (momentum 0.1, UNBIASED var), num_batches_tracked += 1.
__global__ void __launch_bounds__(16 * RED_LANES) k_bn_finalize(const float* __restrict__ part, int tiles, int C, float count,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ run_mean, float* __restrict__ run_var,
                                                     long long* __restrict__ nbt, float momentum, float eps,
                                                     float* __restrict__ scale, float* __restrict__ shift,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    // block = 16 channels x 64 tile-lanes; fixed-order combination of the lane sums (deterministic)
    __shared__ double sh[2][RED_LANES][17];
    const int col = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + col;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += 1;
    double s = 0.0, q = 0.0;
    if (c < C) {
        int t = rl;
        for (; t + 3 * RED_LANES < tiles; t += 4 * RED_LANES) {       // eight independent loads per round
            const float* p0 = part + (size_t)t * 2 * C + c;
            const size_t st = (size_t)RED_LANES * 2 * C;
            const float s0 = p0[0], s1 = p0[st], s2 = p0[2 * st], s3 = p0[3 * st];
            const float q0 = p0[C], q1 = p0[st + C], q2 = p0[2 * st + C], q3 = p0[3 * st + C];
            s += ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
            q += ((double)q0 + (double)q1) + ((double)q2 + (double)q3);
        }
        for (; t < tiles; t += RED_LANES) {
            s += (double)part[(size_t)t * 2 * C + c];
            q += (double)part[(size_t)t * 2 * C + C + c];
        }
    }
    sh[0][rl][col] = s;
    sh[1][rl][col] = q;
#pragma unroll
    for (int half = RED_LANES / 2; half >= 1; half >>= 1) {      // fixed binary tree (see k_sum_partials)
        __syncthreads();
        if (rl < half) {
            sh[0][rl][col] += sh[0][rl + half][col];
            sh[1][rl][col] += sh[1][rl + half][col];
        }
    }
    if (rl != 0 || c >= C) return;
    s = sh[0][0][col];
    q = sh[1][0][col];
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma[c];
    scale[c] = g * rstd;
    shift[c] = beta[c] - (float)mean * g * rstd;
    mean_out[c] = (float)mean;
    rstd_out[c] = rstd;
    if (run_mean) {
        const double unbiased = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unbiased;
    }
}
extern "C" int pk_bn_finalize(const float* stats_partial, int tiles, int C, int count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                              float* scale, float* shift, float* save_mean, float* save_rstd, void* stream) {
    PK_REQUIRE(stats_partial && gamma && beta && scale && shift && save_mean && save_rstd, "pk_bn_finalize: null pointer");
    PK_REQUIRE(tiles > 0 && C > 0 && count > 0, "pk_bn_finalize: bad sizes");
    hipLaunchKernelGGL(k_bn_finalize, dim3((C + 15) / 16), dim3(16 * RED_LANES), 0, (hipStream_t)stream, stats_partial, tiles, C, (float)count,
                       gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, momentum, eps, scale, shift, save_mean,
                       save_rstd);
    return pk_launch_status("pk_bn_finalize");
}

// y = act(x*scale[c] + shift[c] (+ residual)); one 16-byte chunk per thread-iteration
__global__ void __launch_bounds__(256) k_bn_act(const uint4* __restrict__ x, const float* __restrict__ scale,
                                                const float* __restrict__ shift, const uint4* __restrict__ res, uint4* __restrict__ y,
                                                size_t chunks, int cchunks, int relu, uint8_t* __restrict__ mask) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (size_t)gridDim.x * blockDim.x) {
        // (32-bit: a size_t modulo is a ~60-instruction software division per chunk; a mask when the chunk count is a power of two)
        const int c0 = (int)(((cchunks & (cchunks - 1)) == 0) ? ((uint32_t)i & (uint32_t)(cchunks - 1)) : ((uint32_t)i % (uint32_t)cchunks)) * 8;
        float v[8], r[8];
        unpack8(x[i], v);
        if (res) unpack8(res[i], r);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = v[j] * scale[c0 + j] + shift[c0 + j];
            if (res) t += r[j];
            v[j] = relu ? fmaxf(t, 0.f) : t;
        }
        const uint4 o = pack8(v);
        y[i] = o;
        if (mask) mask[i] = (uint8_t)pos_mask8(o);
    }
}
extern "C" int pk_bn_act(const void* x, const float* scale, const float* shift, const void* residual, void* y, int64_t rows, int C,
                         int relu, uint8_t* relu_mask, void* stream) {
    PK_REQUIRE(x && scale && shift && y && rows > 0 && C > 0 && (C & 7) == 0, "pk_bn_act: bad argument (C=%d)", C);
    const size_t chunks = (size_t)rows * (C / 8);
    PK_SUPPORTED(chunks < 0xffffffffull, "pk_bn_act: tensor too large for 32-bit chunk indices");
    size_t nb = (chunks + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_bn_act, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, scale, shift,
                       (const uint4*)residual, (uint4*)y, chunks, C / 8, relu, relu_mask);
    return pk_launch_status("pk_bn_act");
}

// Small tensors (<= BNS_MAX_TILES statistics rows, i.e. the low-resolution branches and their exchange units): finalize + apply in ONE
// launch.  Each workgroup owns 32 channels x a block of rows and first reduces the partial statistics of ITS 32 channels itself (8 row
// lanes per channel, fixed order, double: every workgroup computes bit-identical scale / shift), row block 0 also writes save_mean /
// save_rstd and the running statistics.  One launch (~7 us of fixed cost on a latency-bound chain) less per layer.
#define BNS_MAX_TILES 128
#define BNS_CG 32
__device__ __forceinline__ void bn_act_fin_body(const uint4* __restrict__ x, const float* __restrict__ part, int tiles, int C, float count,
                                                const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ run_mean,
                                                float* __restrict__ run_var, long long* __restrict__ nbt, float momentum, float eps,
                                                float* __restrict__ mean_out, float* __restrict__ rstd_out, const uint4* __restrict__ res,
                                                uint4* __restrict__ y, int rows, int rows_per_block, int relu, const int cg, const int rb,
                                                uint8_t* __restrict__ mask) {
    __shared__ double shs[8][BNS_CG], shq[8][BNS_CG];
    __shared__ float s_scale[BNS_CG], s_shift[BNS_CG];
    const int c0 = cg * BNS_CG;
    {
        const int cl = threadIdx.x & (BNS_CG - 1), pl = threadIdx.x >> 5, c = c0 + cl;
        double sm = 0.0, sq = 0.0;
        if (c < C)
            for (int t = pl; t < tiles; t += 8) {
                sm += (double)part[(size_t)t * 2 * C + c];
                sq += (double)part[(size_t)t * 2 * C + C + c];
            }
        shs[pl][cl] = sm;
        shq[pl][cl] = sq;
    }
    __syncthreads();
    if (threadIdx.x < BNS_CG && c0 + threadIdx.x < C) {
        const int cl = threadIdx.x, c = c0 + cl;
        double sm = 0.0, sq = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            sm += shs[k][cl];
            sq += shq[k][cl];
        }
        const double mean = sm / count;
        double var = sq / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma[c];
        s_scale[cl] = g * rstd;
        s_shift[cl] = beta[c] - (float)mean * g * rstd;
        if (rb == 0) {
            mean_out[c] = (float)mean;
            rstd_out[c] = rstd;
            if (run_mean) {
                const double unbiased = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
                run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
                run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unbiased;
            }
            if (cg == 0 && cl == 0 && nbt) nbt[0] += 1;
        }
    }
    __syncthreads();
    const int cchunks = C / 8, ch = cg * (BNS_CG / 8) + (threadIdx.x & 3);
    if (ch >= cchunks) return;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = s_scale[(threadIdx.x & 3) * 8 + j];
        sh[j] = s_shift[(threadIdx.x & 3) * 8 + j];
    }
    const int r1 = min(rows, (rb + 1) * rows_per_block);
    for (int r = rb * rows_per_block + (threadIdx.x >> 2); r < r1; r += 64) {
        const size_t i = (size_t)r * cchunks + ch;
        float v[8], rr[8];
        unpack8(x[i], v);
        if (res) unpack8(res[i], rr);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = v[j] * sc[j] + sh[j];
            if (res) t += rr[j];
            v[j] = relu ? fmaxf(t, 0.f) : t;
        }
        const uint4 o = pack8(v);
        y[i] = o;
        if (mask) mask[i] = (uint8_t)pos_mask8(o);
    }
}
__global__ void __launch_bounds__(256) k_bn_act_fin(const uint4* __restrict__ x, const float* __restrict__ part, int tiles, int C, float count,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ run_mean,
                                                    float* __restrict__ run_var, long long* __restrict__ nbt, float momentum, float eps,
                                                    float* __restrict__ mean_out, float* __restrict__ rstd_out, const uint4* __restrict__ res,
                                                    uint4* __restrict__ y, int rows, int rows_per_block, int relu, uint8_t* __restrict__ mask) {
    bn_act_fin_body(x, part, tiles, C, count, gamma, beta, run_mean, run_var, nbt, momentum, eps, mean_out, rstd_out, res, y, rows, rows_per_block,
                    relu, (int)blockIdx.x, (int)blockIdx.y, mask);
}
// grouped form: up to PK_GROUP_MAX BatchNorm layers (any row count: every workgroup re-derives the statistics of its 32 channels)
struct BnFwdGroup {
    PkBnFwdDesc d[PK_GROUP_MAX];
    int first[PK_GROUP_MAX + 1], ncg[PK_GROUP_MAX], rpb[PK_GROUP_MAX];
    int n;
};
__global__ void __launch_bounds__(256) k_bn_act_fin_g(BnFwdGroup g) {
    const int L = (int)blockIdx.x;
    int i = 0;
    while (i + 1 < g.n && L >= g.first[i + 1]) ++i;
    const PkBnFwdDesc& d = g.d[i];
    const int local = L - g.first[i], ncg = g.ncg[i];
    bn_act_fin_body((const uint4*)d.raw, d.stats_partial, d.tiles, d.C, (float)d.rows, d.gamma, d.beta, d.running_mean, d.running_var,
                    (long long*)d.num_batches_tracked, d.momentum, d.eps, d.save_mean, d.save_rstd, (const uint4*)d.residual, (uint4*)d.y,
                    (int)d.rows, g.rpb[i], d.relu, local % ncg, local / ncg, d.relu_mask);
}
static inline void bn_fin_grid(int64_t rows, int C, int& ncg, int& nrb, int& rpb) {
    ncg = (C + BNS_CG - 1) / BNS_CG;
    nrb = (int)((rows + 127) / 128);                       // >= 128 rows per workgroup, ~256 workgroups in all
    const int want = (256 + ncg - 1) / ncg;
    if (nrb > want) nrb = want;
    if (nrb < 1) nrb = 1;
    rpb = (int)((rows + nrb - 1) / nrb);
}
extern "C" int pk_bn_train_fwd_group(const PkBnFwdDesc* d, int n, void* stream) {
    PK_REQUIRE(d && n > 0 && n <= PK_GROUP_MAX, "pk_bn_train_fwd_group: 1..%d members, got %d", PK_GROUP_MAX, n);
    BnFwdGroup g{};
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const PkBnFwdDesc& c = d[i];
        PK_REQUIRE(c.raw && c.stats_partial && c.gamma && c.beta && c.y && c.save_mean && c.save_rstd, "pk_bn_train_fwd_group: null pointer");
        PK_REQUIRE(c.tiles > 0 && c.C > 0 && (c.C & 7) == 0 && c.rows > 0 && c.rows < (1 << 30), "pk_bn_train_fwd_group: bad sizes");
        g.d[i] = c;
        int nrb;
        bn_fin_grid(c.rows, c.C, g.ncg[i], nrb, g.rpb[i]);
        g.first[i] = total;
        total += g.ncg[i] * nrb;
    }
    g.first[n] = total;
    g.n = n;
    hipLaunchKernelGGL(k_bn_act_fin_g, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, g);
    return pk_launch_status("pk_bn_train_fwd_group");
}

// Train-mode BatchNorm forward from the conv epilogue's partial statistics: finalize + apply.  One fused launch for small tensors, the
// two kernels above otherwise.  `scale` / `shift`: [C] workspaces of the two-launch path.
extern "C" int pk_bn_train_fwd(const void* raw, const float* stats_partial, int tiles, int C, int64_t rows, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                               float eps, const void* residual, void* y, float* save_mean, float* save_rstd, float* scale, float* shift,
                               int relu, uint8_t* relu_mask, void* stream) {
    PK_REQUIRE(raw && stats_partial && gamma && beta && y && save_mean && save_rstd && scale && shift, "pk_bn_train_fwd: null pointer");
    PK_REQUIRE(tiles > 0 && C > 0 && (C & 7) == 0 && rows > 0, "pk_bn_train_fwd: bad sizes");
    static const int fused_on = PK_KNOB("PK_BN_FUSED", 1);
    if (fused_on && tiles <= BNS_MAX_TILES && rows <= 32768) {
        const int ncg = (C + BNS_CG - 1) / BNS_CG;
        int nrb = (int)((rows + 127) / 128);                       // >= 128 rows per workgroup, ~256 workgroups in all
        const int want = (256 + ncg - 1) / ncg;
        if (nrb > want) nrb = want;
        if (nrb < 1) nrb = 1;
        const int rpb = (int)((rows + nrb - 1) / nrb);
        hipLaunchKernelGGL(k_bn_act_fin, dim3(ncg, nrb), dim3(256), 0, (hipStream_t)stream, (const uint4*)raw, stats_partial, tiles, C, (float)rows,
                           gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, momentum, eps, save_mean, save_rstd,
                           (const uint4*)residual, (uint4*)y, (int)rows, rpb, relu, relu_mask);
        return pk_launch_status("pk_bn_train_fwd");
    }
    int rc = pk_bn_finalize(stats_partial, tiles, C, (int)rows, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, scale,
                            shift, save_mean, save_rstd, stream);
    if (rc) return rc;
    return pk_bn_act(raw, scale, shift, residual, y, rows, C, relu, relu_mask, stream);
}

// Backward, pass 1: per-block partial sums over rows of g and g*xhat, g = dy * (y > 0 if relu).
// (Measured and dropped: recomputing the ReLU mask from `raw` (scale/shift of the forward) to skip the read of `y` in both passes
// made the step SLOWER, 21.97 -> 25.4 ms even with the path disabled at run time: the extra per-channel state costs these
// bandwidth-bound kernels more than the saved 2 bytes per element.)
// Block = 256 threads = (256/cchunks) row lanes x cchunks channel chunks; partial[block][2][C].
#define BNR_MAXC 1024
__device__ __forceinline__ void bn_bwd_reduce_body(const uint4* __restrict__ dy, const uint4* __restrict__ yact,
                                                   const uint4* __restrict__ raw, const float* __restrict__ mean,
                                                   const float* __restrict__ rstd, float* __restrict__ part, int64_t rows, int C,
                                                   int relu, int rows_per_block, const int bid, const uint8_t* __restrict__ mask) {
    __shared__ float sh[256 * 16];
    const int cchunks = C / 8, rlanes = 256 / cchunks;
    const int cc = threadIdx.x % cchunks, rl = threadIdx.x / cchunks;
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float mu[8], rs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mu[j] = mean[cc * 8 + j];
        rs[j] = rstd[cc * 8 + j];
    }
    const int64_t r0 = (int64_t)bid * rows_per_block, r1 = min(rows, r0 + (int64_t)rows_per_block);
    if (rl < rlanes) {
        for (int64_t r = r0 + rl; r < r1; r += rlanes) {
            const size_t i = (size_t)r * cchunks + cc;
            float g[8], xr[8], ya[8];
            unpack8(dy[i], g);
            unpack8(raw[i], xr);
            uint32_t mb = 0xffu;
            if (relu) {
                if (mask) mb = mask[i];
                else {
                    unpack8(yact[i], ya);
                    mb = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) mb |= (ya[j] > 0.f ? 1u : 0u) << j;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float gg = ((mb >> j) & 1u) ? g[j] : 0.f;
                s1[j] += gg;
                s2[j] += gg * (xr[j] - mu[j]) * rs[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sh[threadIdx.x * 16 + j] = s1[j];
        sh[threadIdx.x * 16 + 8 + j] = s2[j];
    }
    __syncthreads();
    if (threadIdx.x < cchunks) {
        float a1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < rlanes; ++k) {
            const int t = k * cchunks + threadIdx.x;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a1[j] += sh[t * 16 + j];
                a2[j] += sh[t * 16 + 8 + j];
            }
        }
        float* dst = part + (size_t)bid * 2 * C;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            dst[threadIdx.x * 8 + j] = a1[j];
            dst[C + threadIdx.x * 8 + j] = a2[j];
        }
    }
}
__global__ void __launch_bounds__(256) k_bn_bwd_reduce(const uint4* __restrict__ dy, const uint4* __restrict__ yact,
                                                       const uint4* __restrict__ raw, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, float* __restrict__ part, int64_t rows, int C,
                                                       int relu, int rows_per_block, const uint8_t* __restrict__ mask) {
    bn_bwd_reduce_body(dy, yact, raw, mean, rstd, part, rows, C, relu, rows_per_block, (int)blockIdx.x, mask);
}
// Backward, pass 2: dx = gamma*rstd*(g - sum_g/M - xhat*sum_gx/M); optional d_residual = g
__global__ void __launch_bounds__(256) k_bn_bwd_apply(const uint4* __restrict__ dy, const uint4* __restrict__ yact,
                                                      const uint4* __restrict__ raw, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ sums, float inv_count, uint4* __restrict__ dx,
                                                      uint4* __restrict__ dres, size_t chunks, int cchunks, int relu,
                                                      const uint8_t* __restrict__ mask) {
    const int C = cchunks * 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (size_t)gridDim.x * blockDim.x) {
        // (32-bit: a size_t modulo is a ~60-instruction software division per chunk; a mask when the chunk count is a power of two)
        const int c0 = (int)(((cchunks & (cchunks - 1)) == 0) ? ((uint32_t)i & (uint32_t)(cchunks - 1)) : ((uint32_t)i % (uint32_t)cchunks)) * 8;
        float g[8], xr[8], ya[8], o[8];
        unpack8(dy[i], g);
        unpack8(raw[i], xr);
        uint32_t mb = 0xffu;
        if (relu) {
            if (mask) mb = mask[i];
            else {
                unpack8(yact[i], ya);
                mb = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) mb |= (ya[j] > 0.f ? 1u : 0u) << j;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + j;
            const float gg = ((mb >> j) & 1u) ? g[j] : 0.f;
            g[j] = gg;
            const float xh = (xr[j] - mean[c]) * rstd[c];
            o[j] = gamma[c] * rstd[c] * (gg - sums[c] * inv_count - xh * sums[C + c] * inv_count);
        }
        dx[i] = pack8(o);
        if (dres) dres[i] = pack8(g);
    }
}
// (PK_BN_FUSED_ROWS / PK_BN_FUSED_NB, measured on HRNet-W32 384x288 training, 22.23 ms per step at the defaults: 262 144 rows with 128 / 256
// partial rows 23.57 / 23.06 ms, 65 536 rows with 128 partial rows 22.29 ms -- the large tensors keep the three-launch form)
static inline bool bn_bwd_small(int64_t rows) {
    static const int fused_on = PK_KNOB("PK_BN_FUSED", 1);
    static const long max_rows = PK_KNOB("PK_BN_FUSED_ROWS", 32768);
    return fused_on && rows <= max_rows;
}
extern "C" int pk_bn_bwd_blocks(int64_t rows) {
    // >= 32 rows per block, up to 1024 blocks: the low-resolution branches (3 072 .. 12 288 rows) still get hundreds of
    // workgroups (rows/256 left them with 12 .. 48 and a 55 us kernel for 3 MB of data).  Small tensors: at most 64 blocks, because
    // every workgroup of the fused apply kernel re-reduces the partial sums of its 32 channels itself (64 x 2 x 32 floats = 16 KB).
    int64_t nb = (rows + 31) / 32;
    static const int small_nb = PK_KNOB("PK_BN_FUSED_NB", 64);
    if (bn_bwd_small(rows) && nb > small_nb) nb = small_nb;
    return (int)(nb > 1024 ? 1024 : (nb < 1 ? 1 : nb));
}
// Small tensors: pass 2 with the partial-sum reduction folded in (see k_bn_act_fin): each workgroup owns 32 channels x a block of rows,
// reduces partial[nb][2][C] for its channels (8 lanes, fixed order, double), row block 0 writes dbeta / dgamma; then dx (and dres).
__device__ __forceinline__ void bn_bwd_apply_fin_body(const uint4* __restrict__ dy, const uint4* __restrict__ yact, const uint4* __restrict__ raw,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ part, int nb,
                                                      float inv_count, float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                      uint4* __restrict__ dx, uint4* __restrict__ dres, int rows, int rows_per_block, int C,
                                                      int relu, const int cg, const int rb, const uint8_t* __restrict__ mask) {
    __shared__ double sh1[8][BNS_CG], sh2[8][BNS_CG];
    __shared__ float s_1[BNS_CG], s_2[BNS_CG];
    const int c0 = cg * BNS_CG;
    {
        const int cl = threadIdx.x & (BNS_CG - 1), pl = threadIdx.x >> 5, c = c0 + cl;
        double a = 0.0, b = 0.0;
        if (c < C)
            for (int t = pl; t < nb; t += 8) {
                a += (double)part[(size_t)t * 2 * C + c];
                b += (double)part[(size_t)t * 2 * C + C + c];
            }
        sh1[pl][cl] = a;
        sh2[pl][cl] = b;
    }
    __syncthreads();
    if (threadIdx.x < BNS_CG && c0 + threadIdx.x < C) {
        const int cl = threadIdx.x;
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a += sh1[k][cl];
            b += sh2[k][cl];
        }
        s_1[cl] = (float)a;
        s_2[cl] = (float)b;
        if (rb == 0) {
            dbeta[c0 + cl] = (float)a;
            dgamma[c0 + cl] = (float)b;
        }
    }
    __syncthreads();
    const int cchunks = C / 8, ch = cg * (BNS_CG / 8) + (threadIdx.x & 3);
    if (ch >= cchunks) return;
    float mu[8], rs[8], gm[8], t1[8], t2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = ch * 8 + j;
        mu[j] = mean[c];
        rs[j] = rstd[c];
        gm[j] = gamma[c] * rs[j];
        t1[j] = s_1[(threadIdx.x & 3) * 8 + j] * inv_count;
        t2[j] = s_2[(threadIdx.x & 3) * 8 + j] * inv_count;
    }
    const int r1 = min(rows, (rb + 1) * rows_per_block);
    for (int r = rb * rows_per_block + (threadIdx.x >> 2); r < r1; r += 64) {
        const size_t i = (size_t)r * cchunks + ch;
        float g[8], xr[8], ya[8], o[8];
        unpack8(dy[i], g);
        unpack8(raw[i], xr);
        uint32_t mb = 0xffu;
        if (relu) {
            if (mask) mb = mask[i];
            else {
                unpack8(yact[i], ya);
                mb = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) mb |= (ya[j] > 0.f ? 1u : 0u) << j;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gg = ((mb >> j) & 1u) ? g[j] : 0.f;
            g[j] = gg;
            const float xh = (xr[j] - mu[j]) * rs[j];
            o[j] = gm[j] * (gg - t1[j] - xh * t2[j]);
        }
        dx[i] = pack8(o);
        if (dres) dres[i] = pack8(g);
    }
}
__global__ void __launch_bounds__(256) k_bn_bwd_apply_fin(const uint4* __restrict__ dy, const uint4* __restrict__ yact, const uint4* __restrict__ raw,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ part, int nb,
                                                          float inv_count, float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                          uint4* __restrict__ dx, uint4* __restrict__ dres, int rows, int rows_per_block, int C,
                                                          int relu, const uint8_t* __restrict__ mask) {
    bn_bwd_apply_fin_body(dy, yact, raw, mean, rstd, gamma, part, nb, inv_count, dbeta, dgamma, dx, dres, rows, rows_per_block, C, relu,
                          (int)blockIdx.x, (int)blockIdx.y, mask);
}
// grouped backward: ONE reduce launch + ONE apply launch for up to PK_GROUP_MAX BatchNorm layers
struct BnBwdGroup {
    PkBnBwdDesc d[PK_GROUP_MAX];
    int first_r[PK_GROUP_MAX + 1], first_a[PK_GROUP_MAX + 1];
    int nb[PK_GROUP_MAX], rpb_r[PK_GROUP_MAX], ncg[PK_GROUP_MAX], rpb_a[PK_GROUP_MAX];
    int n;
};
__global__ void __launch_bounds__(256) k_bn_bwd_reduce_g(BnBwdGroup g) {
    const int L = (int)blockIdx.x;
    int i = 0;
    while (i + 1 < g.n && L >= g.first_r[i + 1]) ++i;
    const PkBnBwdDesc& d = g.d[i];
    bn_bwd_reduce_body((const uint4*)d.dy, (const uint4*)d.y_act, (const uint4*)d.raw, d.save_mean, d.save_rstd, d.partial, d.rows, d.C, d.relu & 1,
                       g.rpb_r[i], L - g.first_r[i], d.relu_mask);
}
__global__ void __launch_bounds__(256) k_bn_bwd_apply_fin_g(BnBwdGroup g) {
    const int L = (int)blockIdx.x;
    int i = 0;
    while (i + 1 < g.n && L >= g.first_a[i + 1]) ++i;
    const PkBnBwdDesc& d = g.d[i];
    const int local = L - g.first_a[i], ncg = g.ncg[i];
    bn_bwd_apply_fin_body((const uint4*)d.dy, (const uint4*)d.y_act, (const uint4*)d.raw, d.save_mean, d.save_rstd, d.gamma, d.partial, g.nb[i],
                          (d.relu & 2) ? 0.f : 1.f / (float)d.rows, d.dbeta, d.dgamma, (uint4*)d.dx, (uint4*)d.dresidual, (int)d.rows, g.rpb_a[i],
                          d.C, d.relu & 1, local % ncg, local / ncg, d.relu_mask);
}
// partial rows [blocks][2][C] of a member of pk_bn_bwd_group: the small-tensor rule of pk_bn_bwd where it applies (same sums bit for bit)
extern "C" int pk_bn_bwd_group_blocks(int64_t rows) {
    if (bn_bwd_small(rows)) return pk_bn_bwd_blocks(rows);
    const int64_t nb = (rows + 127) / 128;
    return (int)(nb > 256 ? 256 : nb);
}
extern "C" int pk_bn_bwd_group(const PkBnBwdDesc* d, int n, void* stream) {
    PK_REQUIRE(d && n > 0 && n <= PK_GROUP_MAX, "pk_bn_bwd_group: 1..%d members, got %d", PK_GROUP_MAX, n);
    BnBwdGroup g{};
    int tr = 0, ta = 0;
    for (int i = 0; i < n; ++i) {
        const PkBnBwdDesc& c = d[i];
        PK_REQUIRE(c.dy && c.raw && c.save_mean && c.save_rstd && c.gamma && c.partial && c.dgamma && c.dbeta && c.dx, "pk_bn_bwd_group: null pointer");
        PK_REQUIRE(!(c.relu & 1) || c.y_act || c.relu_mask, "pk_bn_bwd_group: relu needs the activated output or its bit mask");
        PK_REQUIRE(c.rows > 0 && c.rows < (1 << 30) && c.C > 0 && (c.C & 7) == 0, "pk_bn_bwd_group: bad sizes");
        PK_SUPPORTED(c.C <= BNR_MAXC && c.C / 8 <= 256, "pk_bn_bwd_group: C=%d too large", c.C);
        g.d[i] = c;
        g.nb[i] = pk_bn_bwd_group_blocks(c.rows);
        g.rpb_r[i] = (int)((c.rows + g.nb[i] - 1) / g.nb[i]);
        int nrb;
        bn_fin_grid(c.rows, c.C, g.ncg[i], nrb, g.rpb_a[i]);
        g.first_r[i] = tr;
        g.first_a[i] = ta;
        tr += g.nb[i];
        ta += g.ncg[i] * nrb;
    }
    g.first_r[n] = tr;
    g.first_a[n] = ta;
    g.n = n;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bn_bwd_reduce_g, dim3((unsigned)tr), dim3(256), 0, st, g);
    hipLaunchKernelGGL(k_bn_bwd_apply_fin_g, dim3((unsigned)ta), dim3(256), 0, st, g);
    return pk_launch_status("pk_bn_bwd_group");
}

extern "C" int pk_bn_bwd(const void* dy, const void* y_act, const void* raw, const float* save_mean, const float* save_rstd,
                         const float* gamma, float* partial, float* sums, float* dgamma, float* dbeta, void* dx, void* dresidual,
                         int64_t rows, int C, int relu, const uint8_t* relu_mask, void* stream) {
    PK_REQUIRE(dy && raw && save_mean && save_rstd && gamma && partial && sums && dgamma && dbeta && dx, "pk_bn_bwd: null pointer");
    // relu: bit 0 = the forward applied ReLU; bit 1 = eval-mode BatchNorm (running statistics are constants: no batch-mean terms)
    const int eval_mode = relu & 2;
    relu &= 1;
    PK_REQUIRE(!relu || y_act || relu_mask, "pk_bn_bwd: relu needs the activated output or its bit mask");
    PK_REQUIRE(rows > 0 && C > 0 && (C & 7) == 0, "pk_bn_bwd: bad sizes");
    PK_SUPPORTED(C <= BNR_MAXC && C / 8 <= 256, "pk_bn_bwd: C=%d too large", C);
    const size_t chunks = (size_t)rows * (C / 8);
    PK_SUPPORTED(chunks < 0xffffffffull, "pk_bn_bwd: tensor too large for 32-bit chunk indices");
    hipStream_t st = (hipStream_t)stream;
    const int nb = pk_bn_bwd_blocks(rows);
    const int rpb = (int)((rows + nb - 1) / nb);
    hipLaunchKernelGGL(k_bn_bwd_reduce, dim3(nb), dim3(256), 0, st, (const uint4*)dy, (const uint4*)y_act, (const uint4*)raw, save_mean,
                       save_rstd, partial, rows, C, relu, rpb, relu_mask);
    if (bn_bwd_small(rows)) {          // the reduction of the partial sums rides in the apply launch
        const int ncg = (C + BNS_CG - 1) / BNS_CG;
        int nrb = (int)((rows + 127) / 128);
        const int want = (256 + ncg - 1) / ncg;
        if (nrb > want) nrb = want;
        if (nrb < 1) nrb = 1;
        const int rpb2 = (int)((rows + nrb - 1) / nrb);
        hipLaunchKernelGGL(k_bn_bwd_apply_fin, dim3(ncg, nrb), dim3(256), 0, st, (const uint4*)dy, (const uint4*)y_act, (const uint4*)raw, save_mean,
                           save_rstd, gamma, partial, nb, eval_mode ? 0.f : 1.f / (float)rows, dbeta, dgamma, (uint4*)dx, (uint4*)dresidual, (int)rows,
                           rpb2, C, relu, relu_mask);
        return pk_launch_status("pk_bn_bwd");
    }
    // sums = [sum g | sum g*xhat] for the apply kernel; the same values go to dbeta / dgamma (possibly flat-gradient views)
    hipLaunchKernelGGL(k_sum_partials, SUM_PARTIALS_GRID(2 * C), dim3(16 * RED_LANES), 0, st, partial, nb, 2 * C, 2 * C, sums, 1.f, 0, dbeta, dgamma, C);
    size_t gb = (chunks + 255) / 256;
    if (gb > 4096) gb = 4096;
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)gb), dim3(256), 0, st, (const uint4*)dy, (const uint4*)y_act, (const uint4*)raw,
                       save_mean, save_rstd, gamma, sums, eval_mode ? 0.f : 1.f / (float)rows, (uint4*)dx, (uint4*)dresidual, chunks, C / 8, relu, relu_mask);
    return pk_launch_status("pk_bn_bwd");
}

// relu backward alone (exchange-unit output / plain masks): dx = dy * (y > 0)
__global__ void __launch_bounds__(256) k_relu_bwd(const uint4* __restrict__ dy, const uint4* __restrict__ y, uint4* __restrict__ dx,
                                                  size_t chunks) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (size_t)gridDim.x * blockDim.x) {
        float g[8], v[8];
        unpack8(dy[i], g);
        unpack8(y[i], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = v[j] > 0.f ? g[j] : 0.f;
        dx[i] = pack8(g);
    }
}
extern "C" int pk_relu_bwd(const void* dy, const void* y, void* dx, int64_t numel, void* stream) {
    PK_REQUIRE(dy && y && dx && numel > 0 && (numel & 7) == 0, "pk_relu_bwd: bad argument");
    const size_t chunks = (size_t)numel / 8;
    size_t gb = (chunks + 255) / 256;
    if (gb > 4096) gb = 4096;
    hipLaunchKernelGGL(k_relu_bwd, dim3((unsigned)gb), dim3(256), 0, (hipStream_t)stream, (const uint4*)dy, (const uint4*)y, (uint4*)dx, chunks);
    return pk_launch_status("pk_relu_bwd");
}

// ================================================================================================ LayerNorm
// Row of C channels handled by L = pow2 >= C/(8*CPL) lanes, CPL chunks of 8 channels per lane (CPL = 2 only for C > 512);
// 64/L rows per wave.  `Cr` (<= C) is the number of REAL channels: statistics are taken over the first Cr channels only
// (the padded twin of HRFormer-base keeps 78 real channels in 80-wide rows; padded inputs are zero, padded outputs are
// forced to zero) -- Cr == C for every natively supported model.
template <int L, int CPL>
__global__ void __launch_bounds__(256) k_ln_fwd(const uint4* __restrict__ x, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, uint4* __restrict__ y, float* __restrict__ mean_out,
                                                float* __restrict__ rstd_out, int64_t rows, int C, int Cr, float eps) {
    const int cchunks = C / 8, sub = threadIdx.x % L;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / L;
    float v[CPL][8];
    bool on[CPL];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int ch = sub + c * L;
        on[c] = row < rows && ch < cchunks;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
        if (on[c]) unpack8(x[(size_t)row * cchunks + ch], v[c]);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (ch * 8 + j < Cr) ? v[c][j] : 0.f;
    }
    s = lanes_sum<L>(s);
    const float mean = s / (float)Cr;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int ch = sub + c * L;
        if (on[c]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) q += (ch * 8 + j < Cr) ? (v[c][j] - mean) * (v[c][j] - mean) : 0.f;
        }
    }
    q = lanes_sum<L>(q);
    const float rstd = rsqrtf(q / (float)Cr + eps);
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int ch = sub + c * L;
        if (!on[c]) continue;
        // gamma / beta have C entries (the padded ones are ignored): unconditional 16-byte loads, select afterwards
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + ch * 8), g1 = *reinterpret_cast<const float4*>(gamma + ch * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(beta + ch * 8), b1 = *reinterpret_cast<const float4*>(beta + ch * 8 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = (v[c][j] - mean) * rstd * gg[j] + bb[j];
            v[c][j] = (ch * 8 + j < Cr) ? t : 0.f;
        }
        y[(size_t)row * cchunks + ch] = pack8(v[c]);
    }
    if (sub == 0 && row < rows) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
}
#define LN_DISPATCH(KERNEL, L, CPL, ...)                                                                                \
    do {                                                                                                                \
        const int64_t threads = rows * (L);                                                                             \
        hipLaunchKernelGGL((KERNEL<L, CPL>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, __VA_ARGS__);   \
    } while (0)
static int ln_lanes(int C) {
    int l = 1;
    while (l * 8 < C && l < 64) l <<= 1;
    return l;
}
extern "C" int pk_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* save_mean, float* save_rstd,
                                int64_t rows, int C, int C_real, float eps, void* stream) {
    PK_REQUIRE(x && gamma && beta && y && save_mean && save_rstd && rows > 0, "pk_layernorm_fwd: bad argument");
    PK_SUPPORTED(C >= 8 && (C & 7) == 0 && C <= 1024, "pk_layernorm_fwd: C=%d (need a multiple of 8, <= 1024)", C);
    const int Cr = C_real > 0 ? C_real : C;
    PK_REQUIRE(Cr <= C, "pk_layernorm_fwd: C_real=%d > C=%d", Cr, C);
    hipStream_t st = (hipStream_t)stream;
    const uint4* X = (const uint4*)x;
    uint4* Y = (uint4*)y;
    if (C > 512) {
        LN_DISPATCH(k_ln_fwd, 64, 2, X, gamma, beta, Y, save_mean, save_rstd, rows, C, Cr, eps);
        return pk_launch_status("pk_layernorm_fwd");
    }
    switch (ln_lanes(C)) {
        case 1: LN_DISPATCH(k_ln_fwd, 1, 1, X, gamma, beta, Y, save_mean, save_rstd, rows, C, Cr, eps); break;
        case 2: LN_DISPATCH(k_ln_fwd, 2, 1, X, gamma, beta, Y, save_mean, save_rstd, rows, C, Cr, eps); break;
        case 4: LN_DISPATCH(k_ln_fwd, 4, 1, X, gamma, beta, Y, save_mean, save_rstd, rows, C, Cr, eps); break;
        case 8: LN_DISPATCH(k_ln_fwd, 8, 1, X, gamma, beta, Y, save_mean, save_rstd, rows, C, Cr, eps); break;
        case 16: LN_DISPATCH(k_ln_fwd, 16, 1, X, gamma, beta, Y, save_mean, save_rstd, rows, C, Cr, eps); break;
        case 32: LN_DISPATCH(k_ln_fwd, 32, 1, X, gamma, beta, Y, save_mean, save_rstd, rows, C, Cr, eps); break;
        default: LN_DISPATCH(k_ln_fwd, 64, 1, X, gamma, beta, Y, save_mean, save_rstd, rows, C, Cr, eps); break;
    }
    return pk_launch_status("pk_layernorm_fwd");
}

// dx = rstd*(gh - mean_c(gh) - xhat*mean_c(gh*xhat)) (+ dres), gh = dy*gamma; per-block partials of dgamma/dbeta.
// Each block owns `rows_per_block` consecutive rows so its column partials can be reduced deterministically.
template <int L, int CPL>
__global__ void __launch_bounds__(256) k_ln_bwd(const uint4* __restrict__ dy, const uint4* __restrict__ x, const float* __restrict__ mean,
                                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                const uint4* __restrict__ dres, uint4* __restrict__ dx, float* __restrict__ part,
                                                int64_t rows, int C, int Cr, int rows_per_block) {
    __shared__ float sh[256 * 16];
    const int cchunks = C / 8, sub = threadIdx.x % L, rl = threadIdx.x / L, rlanes = 256 / L;
    float* dst = part + (size_t)blockIdx.x * 2 * C;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + (int64_t)rows_per_block);
    float ag[CPL][8], ab[CPL][8], gm[CPL][8];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = (sub + c * L) * 8 + j;
            ag[c][j] = ab[c][j] = 0.f;
            gm[c][j] = (col < C) ? gamma[col] : 0.f;      // C entries; padded columns are masked where they are used
            if (col >= Cr) gm[c][j] = 0.f;
        }
    for (int64_t rb = r0; rb < r1; rb += rlanes) {        // uniform trip count: every lane executes the shuffles
        const int64_t row = rb + rl;
        float g[CPL][8], xh[CPL][8];
        float mu = 0.f, rs = 0.f;
        if (row < r1) {
            mu = mean[row];
            rs = rstd[row];
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = sub + c * L;
            const bool on = row < r1 && ch < cchunks;
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 8; ++j) g[c][j] = 0.f;
            if (on) {
                unpack8(dy[(size_t)row * cchunks + ch], g[c]);
                unpack8(x[(size_t)row * cchunks + ch], v);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool real = ch * 8 + j < Cr;
                xh[c][j] = real ? (v[j] - mu) * rs : 0.f;
                if (!real) g[c][j] = 0.f;
                const float gh = g[c][j] * gm[c][j];
                s1 += gh;
                s2 += gh * xh[c][j];
                ag[c][j] += g[c][j] * xh[c][j];
                ab[c][j] += g[c][j];
            }
        }
        s1 = lanes_sum<L>(s1);
        s2 = lanes_sum<L>(s2);
        const float m1 = s1 / (float)Cr, m2 = s2 / (float)Cr;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ch = sub + c * L;
            if (!(row < r1 && ch < cchunks)) continue;
            float o8[8], r8[8];
            if (dres) unpack8(dres[(size_t)row * cchunks + ch], r8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o8[j] = (ch * 8 + j < Cr) ? rs * (g[c][j] * gm[c][j] - m1 - xh[c][j] * m2) : 0.f;
                if (dres) o8[j] += r8[j];
            }
            dx[(size_t)row * cchunks + ch] = pack8(o8);
        }
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sh[threadIdx.x * 16 + j] = ag[c][j];
            sh[threadIdx.x * 16 + 8 + j] = ab[c][j];
        }
        __syncthreads();
        const int ch = threadIdx.x + c * L;
        if (threadIdx.x < L && ch < cchunks) {
            float a1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = 0; k < rlanes; ++k) {
                const int t = k * L + threadIdx.x;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    a1[j] += sh[t * 16 + j];
                    a2[j] += sh[t * 16 + 8 + j];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dst[ch * 8 + j] = a1[j];
                dst[C + ch * 8 + j] = a2[j];
            }
        }
    }
}
extern "C" int pk_ln_bwd_blocks(int64_t rows) {
    int64_t nb = (rows + 31) / 32;             // see pk_bn_bwd_blocks
    return (int)(nb > 1024 ? 1024 : (nb < 1 ? 1 : nb));
}
extern "C" int pk_layernorm_bwd(const void* dy, const void* x, const float* save_mean, const float* save_rstd, const float* gamma,
                                const void* dresidual, void* dx, float* partial, float* dgamma, float* dbeta, int64_t rows, int C,
                                int C_real, void* stream) {
    PK_REQUIRE(dy && x && save_mean && save_rstd && gamma && dx && partial && rows > 0 && (!dgamma == !dbeta), "pk_layernorm_bwd: bad argument");
    PK_SUPPORTED(C >= 8 && (C & 7) == 0 && C <= 1024, "pk_layernorm_bwd: C=%d", C);
    const int Cr = C_real > 0 ? C_real : C;
    PK_REQUIRE(Cr <= C, "pk_layernorm_bwd: C_real=%d > C=%d", Cr, C);
    hipStream_t st = (hipStream_t)stream;
    const int nb = pk_ln_bwd_blocks(rows);
    const int rpb = (int)((rows + nb - 1) / nb);
    const uint4 *DY = (const uint4*)dy, *X = (const uint4*)x, *DR = (const uint4*)dresidual;
    uint4* DX = (uint4*)dx;
#define LNB(L, CPL) hipLaunchKernelGGL((k_ln_bwd<L, CPL>), dim3(nb), dim3(256), 0, st, DY, X, save_mean, save_rstd, gamma, DR, DX, partial, rows, C, Cr, rpb)
    if (C > 512) LNB(64, 2);
    else switch (ln_lanes(C)) {
        case 1: LNB(1, 1); break;
        case 2: LNB(2, 1); break;
        case 4: LNB(4, 1); break;
        case 8: LNB(8, 1); break;
        case 16: LNB(16, 1); break;
        case 32: LNB(32, 1); break;
        default: LNB(64, 1); break;
    }
#undef LNB
    if (!dgamma) return pk_launch_status("pk_layernorm_bwd");   // partials only: reduced later by pk_reduce_many
    hipLaunchKernelGGL(k_sum_partials, SUM_PARTIALS_GRID(2 * C), dim3(16 * RED_LANES), 0, st, partial, nb, 2 * C, 2 * C, (float*)nullptr, 1.f, 0, dgamma,
                       dbeta, C);
    return pk_launch_status("pk_layernorm_bwd");
}

// column sums of a bf16 [rows][N] matrix (bias gradients): partial[block][N] then k_sum_partials
__global__ void __launch_bounds__(256) k_colsum(const uint4* __restrict__ g, const int32_t* __restrict__ rowmap, float* __restrict__ part,
                                                int64_t rows, int N, int rows_per_block, const float* __restrict__ row_scale,
                                                int rows_per_sample) {
    __shared__ float sh[256 * 8];
    const int cchunks = N / 8, rlanes = 256 / cchunks;
    const int cc = threadIdx.x % cchunks, rl = threadIdx.x / cchunks;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + (int64_t)rows_per_block);
    if (rl < rlanes)
        for (int64_t r = r0 + rl; r < r1; r += rlanes) {
            const int64_t src = rowmap ? rowmap[r] : r;
            if (src < 0) continue;
            float v[8];
            unpack8(g[(size_t)src * cchunks + cc], v);
            const float sc = row_scale ? row_scale[src / rows_per_sample] : 1.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += v[j] * sc;
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[threadIdx.x * 8 + j] = a[j];
    __syncthreads();
    if (threadIdx.x < cchunks) {
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < rlanes; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += sh[(k * cchunks + threadIdx.x) * 8 + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) part[(size_t)blockIdx.x * N + threadIdx.x * 8 + j] = s[j];
    }
}
extern "C" int pk_colsum_bf16(const void* g, const int32_t* rowmap, const float* row_scale, int rows_per_sample, float* partial,
                              float* out, int64_t rows, int N, void* stream) {
    PK_REQUIRE(g && partial && out && rows > 0 && N > 0 && (N & 7) == 0, "pk_colsum_bf16: bad argument");
    PK_REQUIRE(!row_scale || rows_per_sample > 0, "pk_colsum_bf16: row_scale needs rows_per_sample");
    PK_SUPPORTED(N / 8 <= 256, "pk_colsum_bf16: N=%d too wide", N);
    const int nb = pk_ln_bwd_blocks(rows);
    const int rpb = (int)((rows + nb - 1) / nb);
    hipLaunchKernelGGL(k_colsum, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const uint4*)g, rowmap, partial, rows, N, rpb, row_scale,
                       rows_per_sample > 0 ? rows_per_sample : 1);
    hipLaunchKernelGGL(k_sum_partials, SUM_PARTIALS_GRID(N), dim3(16 * RED_LANES), 0, (hipStream_t)stream, partial, nb, N, N, out, 1.f, 0);
    return pk_launch_status("pk_colsum_bf16");
}

// ================================================================================================ exchange-unit sum
// out = relu?( sum_i up_i(x_i) ): every input is NHWC bf16 with the same C; input i has spatial size (H>>s_i, W>>s_i)
// ... in general (Hi, Wi) and is bilinearly up-sampled (align_corners=False) to (H, W) when smaller.
struct FuseIn { const uint16_t* p; int H, W; };
struct FuseArgs { FuseIn in[4]; int n; uint16_t* out; int B, H, W, C, relu; const uint16_t* mask_y; };   // mask_y: term 0 counts only where mask_y > 0
__device__ __forceinline__ void bil_taps(int o, int n_in, int n_out, int& i0, int& i1, float& f) {
    float s = ((float)o + 0.5f) * ((float)n_in / (float)n_out) - 0.5f;
    s = fmaxf(s, 0.f);
    i0 = min((int)s, n_in - 1);
    i1 = min(i0 + 1, n_in - 1);
    f = s - (float)i0;
}
__device__ __forceinline__ void fuse_sum_body(const FuseArgs& a, const int bid, const int nblocks) {
    const int cchunks = a.C / 8;
    const size_t chunks = (size_t)a.B * a.H * a.W * cchunks;
    const float inv_c = 1.f / (float)cchunks, inv_w = 1.f / (float)a.W, inv_h = 1.f / (float)a.H;
    for (size_t i = (size_t)bid * blockDim.x + threadIdx.x; i < chunks; i += (size_t)nblocks * blockDim.x) {
        // (32-bit index arithmetic: five size_t divisions were ~300 VALU instructions per 16-byte output chunk)
        const uint32_t i32 = (uint32_t)i;
        const bool f24 = chunks < (1u << 24);
        const uint32_t pix0 = f24 ? fdiv24(i32, cchunks, inv_c) : i32 / (uint32_t)cchunks;
        const int cc = (int)(i32 - pix0 * (uint32_t)cchunks);
        const uint32_t pix1 = f24 ? fdiv24(pix0, a.W, inv_w) : pix0 / (uint32_t)a.W;
        const int x = (int)(pix0 - pix1 * (uint32_t)a.W);
        const int b = (int)(f24 ? fdiv24(pix1, a.H, inv_h) : pix1 / (uint32_t)a.H), y = (int)(pix1 - (uint32_t)b * (uint32_t)a.H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < a.n; ++k) {
            const FuseIn& in = a.in[k];
            float v[8];
            if (in.H == a.H && in.W == a.W) {
                unpack8(*reinterpret_cast<const uint4*>(in.p + (((size_t)b * a.H + y) * a.W + x) * a.C + cc * 8), v);
                if (k == 0 && a.mask_y) {         // relu backward folded into the sum of input gradients: g = dy * (y > 0)
                    float m[8];
                    unpack8(*reinterpret_cast<const uint4*>(a.mask_y + i * 8), m);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = m[j] > 0.f ? v[j] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            } else {
                int y0, y1, x0, x1;
                float fy, fx;
                bil_taps(y, in.H, a.H, y0, y1, fy);
                bil_taps(x, in.W, a.W, x0, x1, fx);
                const uint16_t* base = in.p + (size_t)b * in.H * in.W * a.C + cc * 8;
                float v01[8], v10[8], v11[8];
                unpack8(*reinterpret_cast<const uint4*>(base + ((size_t)y0 * in.W + x0) * a.C), v);
                unpack8(*reinterpret_cast<const uint4*>(base + ((size_t)y0 * in.W + x1) * a.C), v01);
                unpack8(*reinterpret_cast<const uint4*>(base + ((size_t)y1 * in.W + x0) * a.C), v10);
                unpack8(*reinterpret_cast<const uint4*>(base + ((size_t)y1 * in.W + x1) * a.C), v11);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc[j] += (v[j] * (1.f - fx) + v01[j] * fx) * (1.f - fy) + (v10[j] * (1.f - fx) + v11[j] * fx) * fy;
            }
        }
        if (a.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);
        }
        *reinterpret_cast<uint4*>(a.out + i * 8) = pack8(acc);
    }
}
__global__ void __launch_bounds__(256) k_fuse_sum(FuseArgs a) { fuse_sum_body(a, (int)blockIdx.x, (int)gridDim.x); }
struct FuseGroup {
    FuseArgs a[PK_GROUP_MAX];
    int first[PK_GROUP_MAX + 1];
    int n;
};
__global__ void __launch_bounds__(256) k_fuse_sum_g(FuseGroup g) {
    const int L = (int)blockIdx.x;
    int i = 0;
    while (i + 1 < g.n && L >= g.first[i + 1]) ++i;
    fuse_sum_body(g.a[i], L - g.first[i], g.first[i + 1] - g.first[i]);
}
static int fuse_args_from(const PkFuseDesc& c, FuseArgs& a, const char* who) {
    PK_REQUIRE(c.out && c.n_inputs >= 1 && c.n_inputs <= 4, "%s: bad argument", who);
    PK_REQUIRE(c.B > 0 && c.H > 0 && c.W > 0 && c.C > 0 && (c.C & 7) == 0, "%s: bad shape", who);
    for (int i = 0; i < c.n_inputs; ++i) {
        PK_REQUIRE(c.inputs[i] && c.in_h[i] > 0 && c.in_w[i] > 0 && c.in_h[i] <= c.H && c.in_w[i] <= c.W, "%s: input %d shape", who, i);
        a.in[i] = FuseIn{(const uint16_t*)c.inputs[i], c.in_h[i], c.in_w[i]};
    }
    PK_REQUIRE(!c.mask_y || (c.in_h[0] == c.H && c.in_w[0] == c.W), "%s: the masked term must have the output's size", who);
    a.n = c.n_inputs; a.out = (uint16_t*)c.out; a.B = c.B; a.H = c.H; a.W = c.W; a.C = c.C; a.relu = c.relu; a.mask_y = (const uint16_t*)c.mask_y;
    PK_SUPPORTED((size_t)c.B * c.H * c.W * (c.C / 8) < 0xffffffffull, "%s: tensor too large for 32-bit chunk indices", who);
    return PK_OK;
}
// up to PK_GROUP_MAX sums (the outputs of one exchange unit, or the input-gradient sums of its backward) in ONE launch
extern "C" int pk_fuse_sum_group(const PkFuseDesc* d, int n, void* stream) {
    PK_REQUIRE(d && n > 0 && n <= PK_GROUP_MAX, "pk_fuse_sum_group: 1..%d members, got %d", PK_GROUP_MAX, n);
    FuseGroup g{};
    int total = 0;
    for (int i = 0; i < n; ++i) {
        int rc = fuse_args_from(d[i], g.a[i], "pk_fuse_sum_group");
        if (rc) return rc;
        const size_t chunks = (size_t)d[i].B * d[i].H * d[i].W * (d[i].C / 8);
        size_t gb = (chunks + 255) / 256;
        if (gb > 2048) gb = 2048;
        g.first[i] = total;
        total += (int)gb;
    }
    g.first[n] = total;
    g.n = n;
    hipLaunchKernelGGL(k_fuse_sum_g, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, g);
    return pk_launch_status("pk_fuse_sum_group");
}
extern "C" int pk_fuse_sum(const void* const* inputs, const int* in_h, const int* in_w, int n_inputs, void* out, int B, int H, int W,
                           int C, int relu, void* stream) {
    PK_REQUIRE(inputs && in_h && in_w && out && n_inputs >= 1 && n_inputs <= 4, "pk_fuse_sum: bad argument");
    PK_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0, "pk_fuse_sum: bad shape");
    FuseArgs a{};
    for (int i = 0; i < n_inputs; ++i) {
        PK_REQUIRE(inputs[i] && in_h[i] > 0 && in_w[i] > 0 && in_h[i] <= H && in_w[i] <= W, "pk_fuse_sum: input %d shape", i);
        a.in[i] = FuseIn{(const uint16_t*)inputs[i], in_h[i], in_w[i]};
    }
    a.n = n_inputs; a.out = (uint16_t*)out; a.B = B; a.H = H; a.W = W; a.C = C; a.relu = relu; a.mask_y = nullptr;
    const size_t chunks = (size_t)B * H * W * (C / 8);
    PK_SUPPORTED(chunks < 0xffffffffull, "pk_fuse_sum: tensor too large for 32-bit chunk indices");
    size_t gb = (chunks + 255) / 256;
    if (gb > 4096) gb = 4096;
    hipLaunchKernelGGL(k_fuse_sum, dim3((unsigned)gb), dim3(256), 0, (hipStream_t)stream, a);
    return pk_launch_status("pk_fuse_sum");
}

// Backward of the bilinear up-sampling (gather form, deterministic): dsrc[b][ys][xs][c] = sum over the output pixels
// whose taps touch (ys,xs) of weight * dy.  `dy` is the (already relu-masked) gradient at the fused resolution.
template <int SPLIT>   // lanes that share one output chunk (each takes every SPLIT-th row of the gather window)
__device__ __forceinline__ void upsample_bwd_body(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ mask_y, uint16_t* __restrict__ dsrc,
                                                  int B, int H, int W, int Hs, int Ws, int C, const int bid, const int nblocks) {
    const int cchunks = C / 8;
    const size_t chunks = (size_t)B * Hs * Ws * cchunks;
    const int ry = (H + Hs - 1) / Hs, rx = (W + Ws - 1) / Ws;
    const int sub = threadIdx.x % SPLIT;
    const size_t groups_per_grid = (size_t)nblocks * blockDim.x / SPLIT;
    const size_t n_iter = (chunks + groups_per_grid - 1) / groups_per_grid;          // uniform trip count: all lanes reach the shuffles
    size_t i = ((size_t)bid * blockDim.x + threadIdx.x) / SPLIT;
    for (size_t it = 0; it < n_iter; ++it, i += groups_per_grid) {
        const bool on = i < chunks;
        const size_t ii = on ? i : 0;
        const uint32_t i32 = (uint32_t)ii;
        const bool f24 = chunks < (1u << 24);
        const uint32_t pix0 = f24 ? fdiv24(i32, cchunks, 1.f / (float)cchunks) : i32 / (uint32_t)cchunks;
        const int cc = (int)(i32 - pix0 * (uint32_t)cchunks);
        const uint32_t pix1 = f24 ? fdiv24(pix0, Ws, 1.f / (float)Ws) : pix0 / (uint32_t)Ws;
        const int xs = (int)(pix0 - pix1 * (uint32_t)Ws);
        const int b = (int)(f24 ? fdiv24(pix1, Hs, 1.f / (float)Hs) : pix1 / (uint32_t)Hs), ys = (int)(pix1 - (uint32_t)b * (uint32_t)Hs);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // support of source pixel ys: outputs whose source coordinate (oy + .5) * Hs/H - .5 lies in (ys-1, ys+1), i.e.
        // oy in ((ys-.5)*H/Hs - .5, (ys+1.5)*H/Hs - .5); H/Hs <= ry, one extra row each side covers non-integer ratios.
        // (The border clamps only move outputs that are already inside this range.)
        // For an exact even ratio r the support is EXACTLY the 2r outputs [ys r - r/2, ys r + 3r/2): the generic window above scans
        // 3r + 2 candidates per axis and computes the taps of each (r = 2: 64 candidates for 16 contributors -- the kernel was bound by
        // that arithmetic, ~19 us for a 6 MB gradient).
        const bool ey = (H == Hs * ry) && !(ry & 1), ex = (W == Ws * rx) && !(rx & 1);
        const int oy_lo = ey ? max(0, ys * ry - ry / 2) : max(0, (ys - 1) * ry - 1), oy_hi = ey ? min(H, ys * ry + 3 * ry / 2) : min(H, (ys + 2) * ry + 1);
        const int ox_lo = ex ? max(0, xs * rx - rx / 2) : max(0, (xs - 1) * rx - 1), ox_hi = ex ? min(W, xs * rx + 3 * rx / 2) : min(W, (xs + 2) * rx + 1);
        if (on)
            for (int oy = oy_lo + sub; oy < oy_hi; oy += SPLIT) {
                int y0, y1;
                float fy;
                bil_taps(oy, Hs, H, y0, y1, fy);
                const float wy = (y0 == ys ? 1.f - fy : 0.f) + (y1 == ys ? fy : 0.f);
                if (wy == 0.f) continue;
                for (int ox = ox_lo; ox < ox_hi; ++ox) {
                    int x0, x1;
                    float fx;
                    bil_taps(ox, Ws, W, x0, x1, fx);
                    const float wx = (x0 == xs ? 1.f - fx : 0.f) + (x1 == xs ? fx : 0.f);
                    if (wx == 0.f) continue;
                    float v[8];
                    unpack8(*reinterpret_cast<const uint4*>(dy + (((size_t)b * H + oy) * W + ox) * C + cc * 8), v);
                    if (mask_y) {                 // relu backward of the fused sum folded in: the gradient counts only where y > 0
                        float m[8];
                        unpack8(*reinterpret_cast<const uint4*>(mask_y + (((size_t)b * H + oy) * W + ox) * C + cc * 8), m);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = m[j] > 0.f ? v[j] : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += wy * wx * v[j];
                }
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = lanes_sum<SPLIT>(acc[j]);       // fixed butterfly over the SPLIT row-lanes: deterministic
        if (on && sub == 0) *reinterpret_cast<uint4*>(dsrc + i * 8) = pack8(acc);
    }
}
template <int SPLIT>
__global__ void __launch_bounds__(256) k_upsample_bwd(const uint16_t* __restrict__ dy, uint16_t* __restrict__ dsrc, int B, int H, int W,
                                                      int Hs, int Ws, int C) {
    upsample_bwd_body<SPLIT>(dy, nullptr, dsrc, B, H, W, Hs, Ws, C, (int)blockIdx.x, (int)gridDim.x);
}
static inline int upsample_split(int H, int Hs, size_t chunks) {
    const int ry = (H + Hs - 1) / Hs;
    // few output chunks with a tall gather window (scale 4 / 8): spread the window rows over 4 / 16 lanes per chunk
    return (ry >= 8 && chunks < (1u << 20)) ? 16 : ((ry >= 4 && chunks < (1u << 20)) ? 4 : 1);
}
struct UpBwdGroup {
    PkUpBwdDesc d[PK_GROUP_MAX];
    int first[PK_GROUP_MAX + 1], split[PK_GROUP_MAX];
    int n;
};
__global__ void __launch_bounds__(256) k_upsample_bwd_g(UpBwdGroup g) {
    const int L = (int)blockIdx.x;
    int i = 0;
    while (i + 1 < g.n && L >= g.first[i + 1]) ++i;
    const PkUpBwdDesc& d = g.d[i];
    const int bid = L - g.first[i], nb = g.first[i + 1] - g.first[i], sp = g.split[i];
    const uint16_t* dy = (const uint16_t*)d.dy;
    const uint16_t* my = (const uint16_t*)d.mask_y;
    uint16_t* ds = (uint16_t*)d.dsrc;
    if (sp == 16) upsample_bwd_body<16>(dy, my, ds, d.B, d.H, d.W, d.Hs, d.Ws, d.C, bid, nb);
    else if (sp == 4) upsample_bwd_body<4>(dy, my, ds, d.B, d.H, d.W, d.Hs, d.Ws, d.C, bid, nb);
    else upsample_bwd_body<1>(dy, my, ds, d.B, d.H, d.W, d.Hs, d.Ws, d.C, bid, nb);
}
// the up-sampling backward of every up-route of one exchange unit in ONE launch; mask_y (optional) folds the fused sum's ReLU backward in
extern "C" int pk_upsample_bwd_group(const PkUpBwdDesc* d, int n, void* stream) {
    PK_REQUIRE(d && n > 0 && n <= PK_GROUP_MAX, "pk_upsample_bwd_group: 1..%d members, got %d", PK_GROUP_MAX, n);
    UpBwdGroup g{};
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const PkUpBwdDesc& c = d[i];
        PK_REQUIRE(c.dy && c.dsrc && c.B > 0 && c.H >= c.Hs && c.W >= c.Ws && c.Hs > 0 && c.Ws > 0 && c.C > 0 && (c.C & 7) == 0,
                   "pk_upsample_bwd_group: bad argument");
        const size_t chunks = (size_t)c.B * c.Hs * c.Ws * (c.C / 8);
        PK_SUPPORTED(chunks < 0xffffffffull, "pk_upsample_bwd_group: tensor too large for 32-bit chunk indices");
        g.d[i] = c;
        g.split[i] = upsample_split(c.H, c.Hs, chunks);
        size_t gb = (chunks * g.split[i] + 255) / 256;
        if (gb > 2048) gb = 2048;
        g.first[i] = total;
        total += (int)gb;
    }
    g.first[n] = total;
    g.n = n;
    hipLaunchKernelGGL(k_upsample_bwd_g, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, g);
    return pk_launch_status("pk_upsample_bwd_group");
}
extern "C" int pk_sizeof_group_desc(int which) {
    switch (which) {
        case 0: return (int)sizeof(PkConvDesc);
        case 1: return (int)sizeof(PkBnFwdDesc);
        case 2: return (int)sizeof(PkBnBwdDesc);
        case 3: return (int)sizeof(PkFuseDesc);
        case 4: return (int)sizeof(PkUpBwdDesc);
        case 5: return (int)sizeof(PkWgradDesc);
        default: return -1;
    }
}
extern "C" int pk_upsample_bilinear_bwd(const void* dy, void* dsrc, int B, int H, int W, int Hs, int Ws, int C, void* stream) {
    PK_REQUIRE(dy && dsrc && B > 0 && H >= Hs && W >= Ws && Hs > 0 && Ws > 0 && C > 0 && (C & 7) == 0, "pk_upsample_bilinear_bwd: bad argument");
    const size_t chunks = (size_t)B * Hs * Ws * (C / 8);
    PK_SUPPORTED(chunks < 0xffffffffull, "pk_upsample_bilinear_bwd: tensor too large for 32-bit chunk indices");
    const int split = upsample_split(H, Hs, chunks);
    size_t gb = (chunks * split + 255) / 256;
    if (gb > 4096) gb = 4096;
    if (split == 16)
        hipLaunchKernelGGL(k_upsample_bwd<16>, dim3((unsigned)gb), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dy, (uint16_t*)dsrc, B, H, W, Hs, Ws, C);
    else if (split == 4)
        hipLaunchKernelGGL(k_upsample_bwd<4>, dim3((unsigned)gb), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dy, (uint16_t*)dsrc, B, H, W, Hs, Ws, C);
    else
        hipLaunchKernelGGL(k_upsample_bwd<1>, dim3((unsigned)gb), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dy, (uint16_t*)dsrc, B, H, W, Hs, Ws, C);
    return pk_launch_status("pk_upsample_bilinear_bwd");
}

// ================================================================================================ layout / packing
// (B,3,H,W) fp32 NCHW -> (B,H,W,Cp) bf16 NHWC with channels >= Cin zero-filled (Cp = 8: 16-byte pixels for the stem)
__global__ void __launch_bounds__(256) k_nchw_to_nhwc(const float* __restrict__ x, uint16_t* __restrict__ y, int B, int Cin, int H, int W,
                                                      int Cp, const float* __restrict__ sp_y) {
    const size_t pixels = (size_t)B * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pixels; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / ((size_t)H * W), r = i - b * H * W;
        for (int c0 = 0; c0 < Cp; c0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = 0.f;
                if (c0 + j < Cin) {
                    const size_t k = (b * Cin + c0 + j) * H * W + r;
                    t = x[k];
                    if (sp_y) t *= 1.f - __expf(-sp_y[k]);     // d softplus(z)/dz = sigmoid(z) = 1 - exp(-softplus(z))
                }
                v[j] = t;
            }
            *reinterpret_cast<uint4*>(y + i * Cp + c0) = pack8(v);
        }
    }
}
extern "C" int pk_nchw_f32_to_nhwc_bf16(const float* x, const float* softplus_out, void* y, int B, int Cin, int H, int W, int Cpad,
                                        void* stream) {
    PK_REQUIRE(x && y && B > 0 && Cin > 0 && H > 0 && W > 0 && Cpad >= Cin && (Cpad & 7) == 0, "pk_nchw_f32_to_nhwc_bf16: bad argument");
    const size_t pixels = (size_t)B * H * W;
    size_t gb = (pixels + 255) / 256;
    if (gb > 4096) gb = 4096;
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3((unsigned)gb), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)y, B, Cin, H, W, Cpad, softplus_out);
    return pk_launch_status("pk_nchw_f32_to_nhwc_bf16");
}

// Weight packing fp32 -> bf16 compute copies, table driven (ONE launch for the whole model).
//   mode 0: dst[n][t][cp]      = src[n][c][t]            conv forward   (OIHW -> [N][T][Cin_pad])
//   mode 1: dst[c][T-1-t][n]   = src[n][c][t]            conv data-grad (flipped taps, channels swapped), rows padded to Np
//   mode 2: dst[c][n]          = src[n][c]               linear data-grad (transpose)
//   (linear forward is mode 0 with T = 1)
struct PackDesc { const float* src; int64_t dst_off; int N, C, T, mode, Cp, Np; int64_t dst_numel; };
__global__ void __launch_bounds__(256) k_pack_weights(uint16_t* __restrict__ dst, const PackDesc* __restrict__ desc,
                                                      const int* __restrict__ blk_desc, const int* __restrict__ blk_first) {
    const PackDesc d = desc[blk_desc[blockIdx.x]];
    const float* __restrict__ src = d.src;
    const int64_t i0 = (int64_t)(blockIdx.x - blk_first[blockIdx.x]) * 1024;
    for (int k = 0; k < 4; ++k) {
        const int64_t i = i0 + k * 256 + threadIdx.x;
        if (i >= d.dst_numel) return;
        float v = 0.f;
        if (d.mode == 0) {
            const int c = (int)(i % d.Cp), t = (int)((i / d.Cp) % d.T), n = (int)(i / ((int64_t)d.Cp * d.T));
            if (c < d.C) v = src[((int64_t)n * d.C + c) * d.T + t];
        } else if (d.mode == 1) {
            const int n = (int)(i % d.Np), t = (int)((i / d.Np) % d.T), c = (int)(i / ((int64_t)d.Np * d.T));
            if (n < d.N) v = src[((int64_t)n * d.C + c) * d.T + (d.T - 1 - t)];
        } else {
            const int n = (int)(i % d.Np), c = (int)(i / d.Np);
            if (n < d.N) v = src[(int64_t)n * d.C + c];
        }
        dst[d.dst_off + i] = f32_to_bf16(v);
    }
}
extern "C" int pk_pack_weights(void* flat_dst_bf16, const void* desc_table, const int32_t* block_desc, const int32_t* block_first,
                               int n_blocks, void* stream) {
    PK_REQUIRE(flat_dst_bf16 && desc_table && block_desc && block_first && n_blocks > 0, "pk_pack_weights: bad argument");
    hipLaunchKernelGGL(k_pack_weights, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, (uint16_t*)flat_dst_bf16,
                       (const PackDesc*)desc_table, block_desc, block_first);
    return pk_launch_status("pk_pack_weights");
}

// ================================================================================================ padded twins
// A model whose channel counts are not multiples of 8 (HRFormer-base: C = 78/156/312/624, head_dim 39) runs through a
// "padded twin" with 8-aligned shapes.  Every real tensor is a box r[0..3] embedded at the origin of the twin's box p[0..3]
// (plain zero-extension, or head-structured: qkv rows (3, heads, 39 -> 40, C -> Cp), proj columns (C -> Cp, heads, 39 -> 40)).
// direction 0: twin[box] = real (parameters / buffers before a forward); 1: real = twin[box] (gradients into a gradient
// sink, BatchNorm running statistics after a step); 2: real += twin[box] (autograd-style accumulation into `.grad`).
// Table driven: ONE launch per direction for the whole model.
struct EmbedDesc { float* real; int64_t twin_off; int r[4]; int p[4]; int64_t numel; };
__global__ void __launch_bounds__(256) k_embed_boxes(float* __restrict__ twin, const EmbedDesc* __restrict__ desc,
                                                     const int* __restrict__ blk_desc, const int* __restrict__ blk_first, int direction) {
    const EmbedDesc d = desc[blk_desc[blockIdx.x]];
    const int64_t i0 = (int64_t)(blockIdx.x - blk_first[blockIdx.x]) * 1024;
    for (int k = 0; k < 4; ++k) {
        const int64_t i = i0 + k * 256 + threadIdx.x;
        if (i >= d.numel) return;
        int64_t t = i;
        const int i3 = (int)(t % d.r[3]);
        t /= d.r[3];
        const int i2 = (int)(t % d.r[2]);
        t /= d.r[2];
        const int i1 = (int)(t % d.r[1]), i0_ = (int)(t / d.r[1]);
        const int64_t j = d.twin_off + (((int64_t)i0_ * d.p[1] + i1) * d.p[2] + i2) * d.p[3] + i3;
        if (direction == 0) twin[j] = d.real[i];
        else if (direction == 1) d.real[i] = twin[j];
        else d.real[i] += twin[j];
    }
}
extern "C" int pk_embed_boxes(float* twin_base, const void* desc_table, const int* block_desc, const int* block_first, int n_blocks,
                              int direction, void* stream) {
    PK_REQUIRE(twin_base && desc_table && block_desc && block_first && n_blocks > 0 && direction >= 0 && direction <= 2,
               "pk_embed_boxes: bad argument");
    hipLaunchKernelGGL(k_embed_boxes, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, twin_base, (const EmbedDesc*)desc_table, block_desc,
                       block_first, direction);
    return pk_launch_status("pk_embed_boxes");
}
