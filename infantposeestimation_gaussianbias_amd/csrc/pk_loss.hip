// Fused losses (L1-L4): FusionPoseLoss forward + hand-derived backward, per-pixel losses, spatial statistics.
// HBM-bound fp32 reductions: one 256-thread workgroup per (b,k) map, wave-shuffle + LDS block reductions,
// a single-block deterministic finalize (no float atomics -> bitwise reproducible run to run).
#include "pk_common.h"

#define EPS8 1e-8f

__constant__ int c_skel[16][2] = {{0, 1}, {0, 2}, {1, 3}, {2, 4}, {5, 6}, {5, 7}, {7, 9}, {6, 8},
                                  {8, 10}, {5, 11}, {6, 12}, {11, 12}, {11, 13}, {13, 15}, {12, 14}, {14, 16}};

// per-map statistics slots (PK_LOSS_STAT = 24)
enum { ST_M, ST_Z, ST_CX, ST_CY, ST_MSE, ST_ENT, ST_UBAR, ST_R, ST_SPREAD, ST_SIG, ST_MEANVAR, ST_SSUM, ST_OX, ST_OY,
       ST_DOXX, ST_DOXY, ST_DOYX, ST_DOYY, ST_QX, ST_QY, ST_QSUM, ST_GX, ST_GY, ST_W };
// globals (16 floats after the pair block)
enum { GL_S, GL_DEN, GL_TENT, GL_S3, GL_UTW };   // S3 / UTW: normaliser of the heatmap / offset / peak terms and the use_target_weight flag

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// ------------------------------------------------------------------------------------------------ forward A
__global__ void __launch_bounds__(256) k_floss_map(const float* __restrict__ hm, const float* __restrict__ off,
                                                   const float* __restrict__ var, const float* __restrict__ tgt,
                                                   const float* __restrict__ wt, const float* __restrict__ gt,
                                                   float* __restrict__ stats, int H, int W, float in_w, float in_h,
                                                   const float* __restrict__ ext_c) {
    // ext_c (B*K,2) or NULL: coordinates handed in by the caller (the public methods of GaussianDistributionConstraint / FusionPoseLoss,
    // fusion_head.py:405-575,659-743, take ANY coordinates); NULL = the soft-argmax of the map itself (FusionPoseLoss.forward, :773)
    __shared__ float red[16];
    const int map = blockIdx.x, n = H * W;
    const float* h = hm + (size_t)map * n;
    const float* t = tgt + (size_t)map * n;
    const float* vr = var + (size_t)map * n;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) mx = fmaxf(mx, h[i]);
    mx = block_max(mx, red);
    float z = 0.f, sx = 0.f, sy = 0.f, mse = 0.f, R = 0.f, ss = 0.f, sv = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = h[i], e = __expf(v - mx), d = v - t[i];
        const int y = i / W;
        const float fx = (float)(i - y * W), fy = (float)y;
        z += e;
        sx += e * fx;
        sy += e * fy;
        mse += d * d;
        R += fmaxf(v, 0.f);
        ss += sigmoidf_(v);
        sv += vr[i];
    }
    z = block_sum(z, red);
    sx = block_sum(sx, red);
    sy = block_sum(sy, red);
    mse = block_sum(mse, red);
    R = block_sum(R, red);
    ss = block_sum(ss, red);
    sv = block_sum(sv, red);
    const float cx = ext_c ? ext_c[2 * map] : sx / z, cy = ext_c ? ext_c[2 * map + 1] : sy / z, rinv = 1.f / (R + EPS8), zinv = 1.f / z;
    float ent = 0.f, ub = 0.f, sp = 0.f, qx = 0.f, qy = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = h[i], p = __expf(v - mx) * zinv;
        const int y = i / W;
        const float fx = (float)(i - y * W), fy = (float)y;
        const float lg = __logf(p + EPS8);
        ent -= p * lg;
        ub += p * (-lg - p / (p + EPS8));
        const float q = fmaxf(v, 0.f) * rinv;
        sp += q * ((fx - cx) * (fx - cx) + (fy - cy) * (fy - cy));
        qx += q * fx;
        qy += q * fy;
    }
    ent = block_sum(ent, red);
    ub = block_sum(ub, red);
    sp = block_sum(sp, red);
    qx = block_sum(qx, red);
    qy = block_sum(qy, red);
    if (threadIdx.x != 0) return;
    float* s = stats + (size_t)map * PK_LOSS_STAT;
    s[ST_M] = mx; s[ST_Z] = z; s[ST_CX] = cx; s[ST_CY] = cy; s[ST_MSE] = mse / (float)n; s[ST_ENT] = ent; s[ST_UBAR] = ub;
    s[ST_R] = R; s[ST_SPREAD] = sp; s[ST_SIG] = sqrtf(sp + EPS8); s[ST_MEANVAR] = sv / (float)n; s[ST_SSUM] = ss;
    s[ST_QX] = qx; s[ST_QY] = qy; s[ST_QSUM] = R * rinv;
    s[ST_GX] = gt[2 * map] * ((float)W / in_w);
    s[ST_GY] = gt[2 * map + 1] * ((float)H / in_h);
    s[ST_W] = wt[map];
    // offsets sampled at c (border clamp; the soft-argmax c is a convex combination of pixel centres, i.e. always inside; external
    // coordinates on or beyond the border sample the clamped position and get no gradient along that axis, as grid_sample's
    // padding_mode='border' does)
    const float x = fminf(fmaxf(cx, 0.f), (float)(W - 1)), y = fminf(fmaxf(cy, 0.f), (float)(H - 1));
    const float inx = (cx > 0.f && cx < (float)(W - 1)) ? 1.f : 0.f, iny = (cy > 0.f && cy < (float)(H - 1)) ? 1.f : 0.f;
    const int x0 = (int)floorf(x), y0 = (int)floorf(y), x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const float fx = x - (float)x0, fy = y - (float)y0;
    for (int ch = 0; ch < 2; ++ch) {
        const float* o = off + ((size_t)map * 2 + ch) * n;
        const float v00 = o[y0 * W + x0], v01 = o[y0 * W + x1], v10 = o[y1 * W + x0], v11 = o[y1 * W + x1];
        s[ST_OX + ch] = v00 * (1.f - fx) * (1.f - fy) + v01 * fx * (1.f - fy) + v10 * (1.f - fx) * fy + v11 * fx * fy;
        s[ST_DOXX + 2 * ch] = inx * ((1.f - fy) * (v01 - v00) + fy * (v11 - v10));      // d o_ch / d cx
        s[ST_DOXX + 2 * ch + 1] = iny * ((1.f - fx) * (v10 - v00) + fx * (v11 - v01));  // d o_ch / d cy
    }
}

// ------------------------------------------------------------------------------------------------ forward B
// one workgroup per (b, skeleton pair): O = sum min(sigmoid(h_i), sigmoid(h_j))
__global__ void __launch_bounds__(256) k_floss_pair(const float* __restrict__ hm, float* __restrict__ pairs, int K, int n) {
    __shared__ float red[16];
    const int b = blockIdx.x / 16, p = blockIdx.x % 16;
    const int i = c_skel[p][0], j = c_skel[p][1];
    float o = 0.f;
    if (i < K && j < K) {
        const float* hi = hm + ((size_t)b * K + i) * n;
        const float* hj = hm + ((size_t)b * K + j) * n;
        for (int q = threadIdx.x; q < n; q += blockDim.x) o += fminf(sigmoidf_(hi[q]), sigmoidf_(hj[q]));
    }
    o = block_sum(o, red);
    if (threadIdx.x == 0) pairs[(size_t)blockIdx.x * 4] = o;
}

// ------------------------------------------------------------------------------------------------ forward C
__device__ __forceinline__ float smooth_l1(float e) { const float a = fabsf(e); return a < 1.f ? 0.5f * e * e : a - 0.5f; }

__global__ void __launch_bounds__(256) k_floss_final(float* __restrict__ stats, float* __restrict__ pairs, float* __restrict__ glob,
                                                     float* __restrict__ losses, int B, int K, float sigma_t,
                                                     const float* __restrict__ lam, int utw) {
    __shared__ float red[16];
    float sw = 0.f, a_hm = 0.f, a_off = 0.f, a_pk = 0.f, a_var = 0.f, a_sh = 0.f;
    const float tent = logf(2.f * 3.14159265358979323846f * 2.71828182845904523536f * sigma_t * sigma_t);
    for (int m = threadIdx.x; m < B * K; m += blockDim.x) {
        const float* s = stats + (size_t)m * PK_LOSS_STAT;
        const float w = s[ST_W];
        // use_target_weight=False (fusion_head.py:653-657,708-712,739-743): the heatmap / offset / peak terms are plain means over
        // (B, K); the Gaussian-constraint terms below are weighted either way (fusion_head.py:478-480,523-527,555-557)
        const float w3 = utw ? w : 1.f;
        sw += w;
        a_hm += w3 * s[ST_MSE];
        const float ex = s[ST_OX] - (s[ST_GX] - s[ST_CX]), ey = s[ST_OY] - (s[ST_GY] - s[ST_CY]);
        a_off += w3 * 0.5f * (smooth_l1(ex) + smooth_l1(ey));
        const float px = s[ST_CX] - s[ST_GX], py = s[ST_CY] - s[ST_GY];
        a_pk += w3 * (px * px + py * py);
        const float ds = s[ST_SIG] - sigma_t, dv = s[ST_MEANVAR] - sigma_t;
        a_var += w * (ds * ds + dv * dv);
        const float de = s[ST_ENT] - tent;
        a_sh += w * de * de;
    }
    float num = 0.f, den = 0.f;
    for (int q = threadIdx.x; q < B * 16; q += blockDim.x) {
        const int b = q / 16, p = q % 16, i = c_skel[p][0], j = c_skel[p][1];
        float r = 0.f, v = 0.f, mn = 0.f;
        if (i < K && j < K) {
            const float* si = stats + ((size_t)b * K + i) * PK_LOSS_STAT;
            const float* sj = stats + ((size_t)b * K + j) * PK_LOSS_STAT;
            mn = fminf(si[ST_SSUM], sj[ST_SSUM]) + EPS8;
            r = pairs[(size_t)q * 4] / mn;
            v = si[ST_W] * sj[ST_W];
            num += fmaxf(r - lam[6], 0.f) * v;          // lam[6] = overlap threshold (fusion_head.py:404,521)
            den += v;
        }
        pairs[(size_t)q * 4 + 1] = mn;
        pairs[(size_t)q * 4 + 2] = r;
        pairs[(size_t)q * 4 + 3] = v;
    }
    sw = block_sum(sw, red); a_hm = block_sum(a_hm, red); a_off = block_sum(a_off, red); a_pk = block_sum(a_pk, red);
    a_var = block_sum(a_var, red); a_sh = block_sum(a_sh, red); num = block_sum(num, red); den = block_sum(den, red);
    if (threadIdx.x != 0) return;
    const float S = sw + EPS8, S3 = utw ? S : (float)(B * K);
    glob[GL_S] = S;
    glob[GL_DEN] = den + EPS8;
    glob[GL_TENT] = tent;
    glob[GL_S3] = S3;
    glob[GL_UTW] = (float)utw;
    float l[6] = {lam[0] * a_hm / S3, lam[1] * a_off / S3, lam[2] * a_pk / S3, lam[3] * a_var / S, lam[4] * (num / (den + EPS8)),
                  lam[5] * a_sh / S};
    float tot = 0.f;
    for (int q = 0; q < 6; ++q) {
        losses[q] = l[q];
        tot += l[q];
    }
    losses[6] = tot;
}

// sigma (B,K) of compute_heatmap_variance (fusion_head.py:405-448) out of the per-map statistics
__global__ void k_floss_sigma(const float* __restrict__ stats, float* __restrict__ sigma, int BK) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < BK) sigma[m] = stats[(size_t)m * PK_LOSS_STAT + ST_SIG];
}

extern "C" int pk_fusion_terms_fwd(const float* heatmaps, const float* offsets, const float* variances, const float* target,
                                   const float* weight, const float* gt_keypoints, const float* coords, float* ws, float* losses,
                                   float* sigma, int B, int K, int H, int W, float in_w, float in_h, float sigma_t,
                                   const float* lambdas6, int use_target_weight, void* stream) {
    PK_REQUIRE(heatmaps && offsets && variances && target && weight && gt_keypoints && ws && losses && lambdas6,
               "pk_fusion_terms_fwd: null pointer");
    PK_REQUIRE(B > 0 && K > 0 && H > 1 && W > 1 && in_w > 0 && in_h > 0 && sigma_t > 0, "pk_fusion_terms_fwd: bad shape B=%d K=%d H=%d W=%d",
               B, K, H, W);
    float* stats = ws;
    float* pairs = ws + (size_t)B * K * PK_LOSS_STAT;
    float* glob = pairs + (size_t)B * 16 * 4;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_floss_map, dim3(B * K), dim3(256), 0, st, heatmaps, offsets, variances, target, weight, gt_keypoints, stats,
                       H, W, in_w, in_h, coords);
    hipLaunchKernelGGL(k_floss_pair, dim3(B * 16), dim3(256), 0, st, heatmaps, pairs, K, H * W);
    hipLaunchKernelGGL(k_floss_final, dim3(1), dim3(256), 0, st, stats, pairs, glob, losses, B, K, sigma_t, lambdas6, use_target_weight ? 1 : 0);
    if (sigma) hipLaunchKernelGGL(k_floss_sigma, dim3((B * K + 255) / 256), dim3(256), 0, st, stats, sigma, B * K);
    return pk_launch_status("pk_fusion_terms_fwd");
}

extern "C" int pk_fusion_loss_fwd(const float* heatmaps, const float* offsets, const float* variances, const float* target,
                                  const float* weight, const float* gt_keypoints, float* ws, float* losses, int B, int K, int H,
                                  int W, float in_w, float in_h, float sigma_t, const float* lambdas6, int use_target_weight, void* stream) {
    return pk_fusion_terms_fwd(heatmaps, offsets, variances, target, weight, gt_keypoints, nullptr, ws, losses, nullptr, B, K, H, W, in_w, in_h,
                               sigma_t, lambdas6, use_target_weight, stream);
}

// ------------------------------------------------------------------------------------------------ backward
__global__ void __launch_bounds__(256) k_floss_bwd(const float* __restrict__ hm, const float* __restrict__ tgt,
                                                   const float* __restrict__ ws, const float* __restrict__ gtot,
                                                   float* __restrict__ dhm, float* __restrict__ doff, float* __restrict__ dvar,
                                                   int B, int K, int H, int W, float sigma_t, const float* __restrict__ lam,
                                                   float* __restrict__ dcoords, const float* __restrict__ gsig,
                                                   const float* __restrict__ gvec) {
    // dcoords != NULL: the coordinates were handed in (k_floss_map's ext_c): their gradient goes to dcoords instead of through the
    // soft-argmax into the heatmap.  gsig (B*K) or NULL: upstream gradient on the per-map sigma (compute_heatmap_variance).
    // gvec (7 floats) or NULL: upstream gradients of the seven returned values; term q is scaled by gvec[q] + gvec[6] (entry 6 is the sum)
#define LG(q) (lam[q] * (gvec ? gvec[q] + gvec[6] : 1.f))
    const int map = blockIdx.x, b = map / K, k = map - b * K, n = H * W;
    const float* s = ws + (size_t)map * PK_LOSS_STAT;
    const float* pairs = ws + (size_t)B * K * PK_LOSS_STAT;
    const float* glob = pairs + (size_t)B * 16 * 4;
    const float G = gtot ? gtot[0] : 1.f;
    const float S = glob[GL_S], den = glob[GL_DEN], tent = glob[GL_TENT];
    const float w = s[ST_W], wS = G * w / S, w3S = G * (glob[GL_UTW] != 0.f ? w : 1.f) / glob[GL_S3];
    const float cx = s[ST_CX], cy = s[ST_CY], mx = s[ST_M], zinv = 1.f / s[ST_Z];
    // d total / d c
    const float ex = s[ST_OX] - (s[ST_GX] - cx), ey = s[ST_OY] - (s[ST_GY] - cy);
    const float sgx = fabsf(ex) < 1.f ? ex : (ex > 0.f ? 1.f : -1.f), sgy = fabsf(ey) < 1.f ? ey : (ey > 0.f ? 1.f : -1.f);
    const float koff = LG(1) * w3S * 0.5f;
    const float kvar = (LG(3) * wS * 2.f * (s[ST_SIG] - sigma_t) + (gsig ? gsig[map] : 0.f)) / (2.f * s[ST_SIG]);   // d total / d spread
    float gcx = LG(2) * w3S * 2.f * (cx - s[ST_GX]) + koff * (sgx * (s[ST_DOXX] + 1.f) + sgy * s[ST_DOYX]) +
                kvar * (-2.f) * (s[ST_QX] - cx * s[ST_QSUM]);
    float gcy = LG(2) * w3S * 2.f * (cy - s[ST_GY]) + koff * (sgx * s[ST_DOXY] + sgy * (s[ST_DOYY] + 1.f)) +
                kvar * (-2.f) * (s[ST_QY] - cy * s[ST_QSUM]);
    if (dcoords) {
        if (threadIdx.x == 0) {
            dcoords[2 * map] = gcx;
            dcoords[2 * map + 1] = gcy;
        }
        gcx = gcy = 0.f;
    }
    const float khm = LG(0) * w3S * 2.f / (float)n;
    const float ksh = LG(5) * wS * 2.f * (s[ST_ENT] - tent);
    const float ubar = s[ST_UBAR], spread = s[ST_SPREAD], rinv = 1.f / (s[ST_R] + EPS8);
    // overlap partners of this joint
    int other[4];
    float kin[4], kout[4];
    int np = 0;
    for (int p = 0; p < 16; ++p) {
        const int i = c_skel[p][0], j = c_skel[p][1];
        if (i >= K || j >= K || (i != k && j != k)) continue;
        const int o = (i == k) ? j : i;
        const float* pr = pairs + ((size_t)b * 16 + p) * 4;
        const float O = pr[0], mn = pr[1], r = pr[2], v = pr[3];
        const float kp = (r > lam[6]) ? G * LG(4) * v / den : 0.f;
        const float sk = s[ST_SSUM], so = ws[((size_t)b * K + o) * PK_LOSS_STAT + ST_SSUM];
        const float ind = sk < so ? 1.f : (sk == so ? 0.5f : 0.f);
        other[np] = o;
        kin[np] = kp / mn;                       // coefficient of 1[s_k < s_o] at the pixel
        kout[np] = -kp * ind * O / (mn * mn);    // coefficient through min(sum_k, sum_o)
        if (++np == 4) break;
    }
    const float* h = hm + (size_t)map * n;
    const float* t = tgt + (size_t)map * n;
    float* dh = dhm + (size_t)map * n;
    // offsets: only the 4 bilinear corners around c receive gradient
    const float x = fminf(fmaxf(cx, 0.f), (float)(W - 1)), y = fminf(fmaxf(cy, 0.f), (float)(H - 1));
    const int x0 = (int)floorf(x), y0 = (int)floorf(y), x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const float fx = x - (float)x0, fy = y - (float)y0;
    float* dox = doff + (size_t)map * 2 * n;
    float* doy = dox + n;
    const float dvu = G * LG(3) * w / S * 2.f * (s[ST_MEANVAR] - sigma_t) / (float)n;
    float* dv = dvar + (size_t)map * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int py = i / W, px = i - py * W;
        const float v = h[i], p = __expf(v - mx) * zinv;
        const float rx = (float)px - cx, ry = (float)py - cy;
        float g = khm * (v - t[i]) + p * (rx * gcx + ry * gcy);
        if (v > 0.f) g += kvar * ((rx * rx + ry * ry) - spread) * rinv;
        g += ksh * p * ((-__logf(p + EPS8) - p / (p + EPS8)) - ubar);
        if (np) {
            const float sk = sigmoidf_(v);
            float acc = 0.f;
            for (int q = 0; q < np; ++q) {
                const float so = sigmoidf_(hm[((size_t)b * K + other[q]) * n + i]);
                acc += kin[q] * (sk < so ? 1.f : (sk == so ? 0.5f : 0.f)) + kout[q];
            }
            g += acc * sk * (1.f - sk);
        }
        dh[i] = g;
        const float wxs = (px == x0 ? 1.f - fx : 0.f) + (px == x1 ? fx : 0.f);
        const float wys = (py == y0 ? 1.f - fy : 0.f) + (py == y1 ? fy : 0.f);
        const float cw = koff * wxs * wys;
        dox[i] = cw * sgx;
        doy[i] = cw * sgy;
        dv[i] = dvu;
    }
}
#undef LG

extern "C" int pk_fusion_terms_bwd(const float* heatmaps, const float* target, const float* ws, const float* grad_total,
                                   const float* grad_sigma, const float* grad_losses, float* d_heatmaps, float* d_offsets,
                                   float* d_variances, float* d_coords, int B, int K, int H, int W, float sigma_t, const float* lambdas6,
                                   void* stream) {
    PK_REQUIRE(heatmaps && target && ws && d_heatmaps && d_offsets && d_variances && lambdas6, "pk_fusion_terms_bwd: null pointer");
    PK_REQUIRE(B > 0 && K > 0 && H > 1 && W > 1, "pk_fusion_terms_bwd: bad shape");
    hipLaunchKernelGGL(k_floss_bwd, dim3(B * K), dim3(256), 0, (hipStream_t)stream, heatmaps, target, ws, grad_total, d_heatmaps,
                       d_offsets, d_variances, B, K, H, W, sigma_t, lambdas6, d_coords, grad_sigma, grad_losses);
    return pk_launch_status("pk_fusion_terms_bwd");
}

extern "C" int pk_fusion_loss_bwd(const float* heatmaps, const float* offsets, const float* variances, const float* target,
                                  const float* weight, const float* ws, const float* grad_total, float* d_heatmaps,
                                  float* d_offsets, float* d_variances, int B, int K, int H, int W, float sigma_t,
                                  const float* lambdas6, void* stream) {
    (void)offsets; (void)variances; (void)weight;
    return pk_fusion_terms_bwd(heatmaps, target, ws, grad_total, nullptr, nullptr, d_heatmaps, d_offsets, d_variances, nullptr, B, K, H, W,
                               sigma_t, lambdas6, stream);
}

// ================================================================================================ L3 / L4 pixel losses
__device__ __forceinline__ float pix_term(float p, float t, float w, int kind) {
    const float d = p - t;
    switch (kind) {
        case 0: return (d * w) * (d * w);
        case 1: return w * d * d;
        case 2: return w * smooth_l1(d);
        default: return 0.5f * (d * w) * (d * w);
    }
}
__global__ void __launch_bounds__(256) k_pixel_loss_partial(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                            const float* __restrict__ wt, float* __restrict__ partial, int BK,
                                                            int HW, int kind) {
    __shared__ float red[16];
    float acc = 0.f;
    const size_t total = (size_t)BK * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float w = wt ? wt[i / HW] : 1.f;
        acc += pix_term(pred[i], tgt[i], w, kind);
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_pixel_loss_final(const float* __restrict__ partial, int nb, float* __restrict__ loss,
                                                          float inv_count) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) acc += partial[i];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) loss[0] = acc * inv_count;
}
__global__ void __launch_bounds__(256) k_pixel_loss_bwd(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                        const float* __restrict__ wt, const float* __restrict__ gout,
                                                        float* __restrict__ dp, size_t total, int HW, int kind, float inv_count) {
    const float G = (gout ? gout[0] : 1.f) * inv_count;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float w = wt ? wt[i / HW] : 1.f, d = pred[i] - tgt[i];
        float g;
        if (kind == 0) g = 2.f * w * w * d;
        else if (kind == 1) g = 2.f * w * d;
        else if (kind == 2) g = w * (fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f));
        else g = w * w * d;
        dp[i] = G * g;
    }
}
static float pixel_inv_count(int B, int K, int HW, int kind) {
    // kinds 0-2: mean over every element; kind 3 (JointsMSELoss): sum_k 0.5*mean_{b,hw} / K  == 0.5*sum/(B*HW*K)
    return 1.f / ((float)B * (float)K * (float)HW);
}
extern "C" int pk_pixel_loss_fwd(const float* pred, const float* target, const float* weight, float* partial, float* loss, int B,
                                 int K, int HW, int kind, void* stream) {
    PK_REQUIRE(pred && target && partial && loss && B > 0 && K > 0 && HW > 0 && kind >= 0 && kind <= 3, "pk_pixel_loss_fwd: bad argument");
    const size_t total = (size_t)B * K * HW;
    int nb = (int)((total + 255) / 256);
    if (nb > PK_REDUCE_BLOCKS) nb = PK_REDUCE_BLOCKS;
    hipLaunchKernelGGL(k_pixel_loss_partial, dim3(nb), dim3(256), 0, (hipStream_t)stream, pred, target, weight, partial, B * K, HW, kind);
    hipLaunchKernelGGL(k_pixel_loss_final, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nb, loss, pixel_inv_count(B, K, HW, kind));
    return pk_launch_status("pk_pixel_loss_fwd");
}
extern "C" int pk_pixel_loss_bwd(const float* pred, const float* target, const float* weight, const float* grad_out, float* d_pred,
                                 int B, int K, int HW, int kind, void* stream) {
    PK_REQUIRE(pred && target && d_pred && B > 0 && K > 0 && HW > 0 && kind >= 0 && kind <= 3, "pk_pixel_loss_bwd: bad argument");
    const size_t total = (size_t)B * K * HW;
    int nb = (int)((total + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_pixel_loss_bwd, dim3(nb), dim3(256), 0, (hipStream_t)stream, pred, target, weight, grad_out, d_pred, total, HW,
                       kind, pixel_inv_count(B, K, HW, kind));
    return pk_launch_status("pk_pixel_loss_bwd");
}

// MorphologyShapeLoss.compute_spatial_statistics: centre of mass and per-axis variance of hm/(sum+1e-8)
__global__ void __launch_bounds__(256) k_spatial_stats(const float* __restrict__ hm, float* __restrict__ mean, float* __restrict__ var,
                                                       int H, int W) {
    __shared__ float red[16];
    const int map = blockIdx.x, n = H * W;
    const float* h = hm + (size_t)map * n;
    float s = 0.f, sx = 0.f, sy = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = i / W;
        const float v = h[i];
        s += v;
        sx += v * (float)(i - y * W);
        sy += v * (float)y;
    }
    s = block_sum(s, red); sx = block_sum(sx, red); sy = block_sum(sy, red);
    const float inv = 1.f / (s + EPS8), mx = sx * inv, my = sy * inv;
    float vx = 0.f, vy = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = i / W;
        const float p = h[i] * inv, dx = (float)(i - y * W) - mx, dy = (float)y - my;
        vx += p * dx * dx;
        vy += p * dy * dy;
    }
    vx = block_sum(vx, red); vy = block_sum(vy, red);
    if (threadIdx.x == 0) {
        mean[2 * map] = mx; mean[2 * map + 1] = my;
        var[2 * map] = vx; var[2 * map + 1] = vy;
    }
}
extern "C" int pk_spatial_stats(const float* heatmaps, float* mean, float* var, int BK, int H, int W, void* stream) {
    PK_REQUIRE(heatmaps && mean && var && BK > 0 && H > 0 && W > 0, "pk_spatial_stats: bad argument");
    hipLaunchKernelGGL(k_spatial_stats, dim3(BK), dim3(256), 0, (hipStream_t)stream, heatmaps, mean, var, H, W);
    return pk_launch_status("pk_spatial_stats");
}

// Backward of k_spatial_stats.  With S = sum h, inv = 1/(S + 1e-8), p = h*inv, q = S*inv:
//   d mean_x / d h_j = inv * (x_j - m_x)
//   d var_x  / d h_j = inv * ((x_j - m_x)^2 - v_x - 2 (x_j - m_x) m_x (1 - q))      (the last term is the 1e-8 in the normaliser)
// and the same along y.  One block per map: recompute S, then one pass over the pixels.
__global__ void __launch_bounds__(256) k_spatial_stats_bwd(const float* __restrict__ hm, const float* __restrict__ mean,
                                                           const float* __restrict__ var, const float* __restrict__ gmean,
                                                           const float* __restrict__ gvar, float* __restrict__ dhm, int H, int W) {
    __shared__ float red[16];
    const int map = blockIdx.x, n = H * W;
    const float* h = hm + (size_t)map * n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += h[i];
    s = block_sum(s, red);
    const float inv = 1.f / (s + EPS8), q = s * inv;
    const float mx = mean[2 * map], my = mean[2 * map + 1], vx = var[2 * map], vy = var[2 * map + 1];
    const float gmx = gmean[2 * map], gmy = gmean[2 * map + 1], gvx = gvar[2 * map], gvy = gvar[2 * map + 1];
    const float cx = 2.f * mx * (1.f - q), cy = 2.f * my * (1.f - q);
    float* d = dhm + (size_t)map * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = i / W;
        const float dx = (float)(i - y * W) - mx, dy = (float)y - my;
        d[i] = inv * (gmx * dx + gmy * dy + gvx * (dx * dx - vx - cx * dx) + gvy * (dy * dy - vy - cy * dy));
    }
}
extern "C" int pk_spatial_stats_bwd(const float* heatmaps, const float* mean, const float* var, const float* grad_mean,
                                    const float* grad_var, float* grad_heatmaps, int BK, int H, int W, void* stream) {
    PK_REQUIRE(heatmaps && mean && var && grad_mean && grad_var && grad_heatmaps && BK > 0 && H > 0 && W > 0, "pk_spatial_stats_bwd: bad argument");
    hipLaunchKernelGGL(k_spatial_stats_bwd, dim3(BK), dim3(256), 0, (hipStream_t)stream, heatmaps, mean, var, grad_mean, grad_var,
                       grad_heatmaps, H, W);
    return pk_launch_status("pk_spatial_stats_bwd");
}
