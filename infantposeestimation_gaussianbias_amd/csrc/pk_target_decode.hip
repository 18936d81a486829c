// Target generation (T1, T2) and heatmap decoders (D1-D4). All HBM-bound, fp32, one workgroup per map.
#include <stdarg.h>

#include "pk_common.h"

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void pk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* pk_last_error_string(void) { return g_err; }
extern "C" int pk_version(void) { return 100; }
// Profiling aid: an empty launch whose workgroup count is `id`, so a rocprofv3 kernel trace can be cut into program sections
// (scripts/trace_sections.py); never launched unless POSE_MARKERS=1.
__global__ void k_marker() {}
extern "C" int pk_marker(int id, void* stream) {
    PK_REQUIRE(id > 0 && id < 65536, "pk_marker: id out of range");
    hipLaunchKernelGGL(k_marker, dim3(id), dim3(64), 0, (hipStream_t)stream);
    return pk_launch_status("pk_marker");
}

// ================================================================================================ T1
// One 256-thread workgroup per (b,k) map. Each thread redoes the (cheap) float64 index arithmetic, then the
// block streams the whole map out with 16-byte stores: every element is written exactly once (zeros or LUT).
__global__ void __launch_bounds__(256) k_gaussian_target(const float* __restrict__ kp, const float* __restrict__ vis,
                                                         const float* __restrict__ lut, float* __restrict__ target,
                                                         float* __restrict__ weight, int Hh, int Wh, double sx, double sy,
                                                         double reach, int pn, int pc) {
    const int map = blockIdx.x;
    float w = vis[map];
    bool draw = !(w < 0.5f);
    int x_lo = 0, y_lo = 0, x_hi = 0, y_hi = 0;
    if (draw) {
        // float32 coordinate / float64 stride, then C truncation toward zero (Python int()).
        const double mx = (double)kp[2 * map] / sx, my = (double)kp[2 * map + 1] / sy;
        const double lim = 1.0e9;
        x_lo = (int)fmin(fmax(mx - reach, -lim), lim);
        y_lo = (int)fmin(fmax(my - reach, -lim), lim);
        x_hi = (int)fmin(fmax(mx + reach + 1.0, -lim), lim);
        y_hi = (int)fmin(fmax(my + reach + 1.0, -lim), lim);
        if (x_lo >= Wh || y_lo >= Hh || x_hi < 0 || y_hi < 0) {
            draw = false;
            w = 0.f;
        }
    }
    if (threadIdx.x == 0) weight[map] = w;
    const int c0 = max(0, x_lo), c1 = min(x_hi, Wh), r0 = max(0, y_lo), r1 = min(y_hi, Hh);
    float* out = target + (size_t)map * Hh * Wh;
    const int n = Hh * Wh;
    if ((Wh & 3) == 0) {
        for (int i = threadIdx.x * 4; i < n; i += blockDim.x * 4) {
            const int y = i / Wh, x = i - y * Wh;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = x + j;
                float val = 0.f;
                if (draw && y >= r0 && y < r1 && xx >= c0 && xx < c1) {
                    const int dx = xx - x_lo - pc, dy = y - y_lo - pc;
                    val = lut[dx * dx + dy * dy];
                }
                v[j] = val;
            }
            *reinterpret_cast<float4*>(out + i) = make_float4(v[0], v[1], v[2], v[3]);
        }
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int y = i / Wh, x = i - y * Wh;
            float val = 0.f;
            if (draw && y >= r0 && y < r1 && x >= c0 && x < c1) {
                const int dx = x - x_lo - pc, dy = y - y_lo - pc;
                val = lut[dx * dx + dy * dy];
            }
            out[i] = val;
        }
    }
}

extern "C" int pk_gaussian_target(const float* keypoints, const float* visible, const float* lut, int lut_len, float* target,
                                  float* weight, int B, int K, int Hh, int Wh, double stride_x, double stride_y,
                                  double reach, int patch_n, int patch_c, void* stream) {
    PK_REQUIRE(keypoints && visible && lut && target && weight, "pk_gaussian_target: null pointer");
    PK_REQUIRE(B > 0 && K > 0 && Hh > 0 && Wh > 0, "pk_gaussian_target: bad shape B=%d K=%d H=%d W=%d", B, K, Hh, Wh);
    PK_REQUIRE(stride_x > 0 && stride_y > 0 && reach > 0, "pk_gaussian_target: bad stride/reach");
    const int far = patch_c > patch_n - 1 - patch_c ? patch_c : patch_n - 1 - patch_c;
    PK_REQUIRE(patch_n > 0 && patch_c >= 0 && patch_c < patch_n && lut_len >= 2 * far * far + 1,
               "pk_gaussian_target: LUT of %d entries too short for a %d-wide patch centred at %d", lut_len, patch_n, patch_c);
    PK_REQUIRE((int)(2 * reach + 1) <= patch_n, "pk_gaussian_target: reach %.2f exceeds the %d-wide patch", reach, patch_n);
    PK_REQUIRE(((uintptr_t)target & 15) == 0, "pk_gaussian_target: target must be 16-byte aligned");
    hipLaunchKernelGGL(k_gaussian_target, dim3(B * K), dim3(256), 0, (hipStream_t)stream, keypoints, visible, lut, target, weight,
                       Hh, Wh, stride_x, stride_y, reach, patch_n, patch_c);
    return pk_launch_status("pk_gaussian_target");
}

// ================================================================================================ T2
__global__ void __launch_bounds__(256) k_dense_target(const float* __restrict__ kp, const float* __restrict__ vis,
                                                      float* __restrict__ hm, float* __restrict__ wts, int Hh, int Wh,
                                                      float scx, float scy, float sigma) {
    const int map = blockIdx.x;
    const float cx = kp[2 * map] * scx, cy = kp[2 * map + 1] * scy;   // float32 multiply, as `scaled_keypoints[:,0] *= scale_w`
    const bool on = vis[map] > 0.f && cx >= 0.f && cx < (float)Wh && cy >= 0.f && cy < (float)Hh;
    if (threadIdx.x == 0) wts[map] = on ? 1.f : 0.f;
    const float denom = 2.f * sigma * sigma;
    float* out = hm + (size_t)map * Hh * Wh;
    for (int i = threadIdx.x; i < Hh * Wh; i += blockDim.x) {
        const int y = i / Wh, x = i - y * Wh;
        float v = 0.f;
        if (on) {
            const float dx = (float)x - cx, dy = (float)y - cy;
            v = fmaxf(0.f, expf(-(dx * dx + dy * dy) / denom));
        }
        out[i] = v;
    }
}

extern "C" int pk_dense_target(const float* keypoints, const float* visible, float* heatmaps, float* weights, int B, int K,
                               int Hh, int Wh, float scale_x, float scale_y, float sigma, void* stream) {
    PK_REQUIRE(keypoints && visible && heatmaps && weights, "pk_dense_target: null pointer");
    PK_REQUIRE(B > 0 && K > 0 && Hh > 0 && Wh > 0 && sigma > 0.f, "pk_dense_target: bad shape");
    hipLaunchKernelGGL(k_dense_target, dim3(B * K), dim3(256), 0, (hipStream_t)stream, keypoints, visible, heatmaps, weights, Hh,
                       Wh, scale_x, scale_y, sigma);
    return pk_launch_status("pk_dense_target");
}

// ================================================================================================ argmax (D2, D3)
struct ArgMax {
    float v;
    int i;
};
// torch.max semantics: NaN counts as the maximum; ties resolve to the lowest flat index.
__device__ __forceinline__ bool am_better(float av, int ai, float bv, int bi) {
    const bool an = av != av, bn = bv != bv;
    if (an != bn) return an;
    if (!an && av != bv) return av > bv;
    return ai < bi;
}
__device__ __forceinline__ ArgMax am_wave(ArgMax a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(a.v, o, 64);
        const int oi = __shfl_xor(a.i, o, 64);
        if (am_better(ov, oi, a.v, a.i)) {
            a.v = ov;
            a.i = oi;
        }
    }
    return a;
}
__device__ __forceinline__ ArgMax am_block(const float* __restrict__ m, int n, float* sv, int* si) {
    ArgMax a{-INFINITY, 0x7fffffff};
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = m[i];
        if (am_better(v, i, a.v, a.i)) {
            a.v = v;
            a.i = i;
        }
    }
    a = am_wave(a);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) {
        sv[w] = a.v;
        si[w] = a.i;
    }
    __syncthreads();
    ArgMax r{sv[0], si[0]};
    for (int k = 1; k < nw; ++k)
        if (am_better(sv[k], si[k], r.v, r.i)) {
            r.v = sv[k];
            r.i = si[k];
        }
    return r;
}

__global__ void __launch_bounds__(256) k_argmax_decode(const float* __restrict__ hm, int32_t* __restrict__ index,
                                                       float* __restrict__ maxval, float* __restrict__ coords, int H, int W,
                                                       int mode) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int map = blockIdx.x;
    const float* m = hm + (size_t)map * H * W;
    const ArgMax r = am_block(m, H * W, sv, si);
    if (threadIdx.x != 0) return;
    if (index) index[map] = r.i;
    if (maxval) maxval[map] = r.v;
    if (!coords) return;
    const int x = r.i % W, y = r.i / W;
    float fx = (float)x, fy = (float)y;
    if (mode == 1) {  // quarter-pixel shift toward the larger neighbour; sign(0) = 0
        if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
            const float dx = m[y * W + x + 1] - m[y * W + x - 1], dy = m[(y + 1) * W + x] - m[(y - 1) * W + x];
            fx += 0.25f * (float)((dx > 0.f) - (dx < 0.f));
            fy += 0.25f * (float)((dy > 0.f) - (dy < 0.f));
        }
    } else if (mode == 2) {  // Taylor step, strict 1 < p < size-1, only where the 2nd difference is negative
        if (x > 1 && x < W - 1 && y > 1 && y < H - 1) {
            const float c = m[y * W + x];
            const float l = m[y * W + x - 1], rr = m[y * W + x + 1], u = m[(y - 1) * W + x], d = m[(y + 1) * W + x];
            const float dxx = (rr - 2.f * c) + l, dyy = (d - 2.f * c) + u;
            if (dxx < 0.f) fx += fminf(fmaxf((rr - l) / (2.f * fabsf(dxx)), -0.5f), 0.5f);
            if (dyy < 0.f) fy += fminf(fmaxf((d - u) / (2.f * fabsf(dyy)), -0.5f), 0.5f);
        }
    }
    coords[2 * map] = fx;
    coords[2 * map + 1] = fy;
}

extern "C" int pk_argmax_decode(const float* heatmaps, int32_t* index, float* maxval, float* coords, int BK, int H, int W,
                                int mode, void* stream) {
    PK_REQUIRE(heatmaps, "pk_argmax_decode: null heatmaps");
    PK_REQUIRE(BK > 0 && H > 0 && W > 0 && (int64_t)H * W < 0x7fffffff, "pk_argmax_decode: bad shape");
    PK_REQUIRE(mode >= 0 && mode <= 2, "pk_argmax_decode: mode %d", mode);
    hipLaunchKernelGGL(k_argmax_decode, dim3(BK), dim3(256), 0, (hipStream_t)stream, heatmaps, index, maxval, coords, H, W, mode);
    return pk_launch_status("pk_argmax_decode");
}

// ================================================================================================ D1
__device__ __forceinline__ float bilinear_border(const float* __restrict__ p, int H, int W, float x, float y) {
    x = fminf(fmaxf(x, 0.f), (float)(W - 1));
    y = fminf(fmaxf(y, 0.f), (float)(H - 1));
    const int x0 = (int)floorf(x), y0 = (int)floorf(y);
    const float fx = x - (float)x0, fy = y - (float)y0;
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);  // weight is 0 whenever the clamp bites
    return p[y0 * W + x0] * (1.f - fx) * (1.f - fy) + p[y0 * W + x1] * fx * (1.f - fy) + p[y1 * W + x0] * (1.f - fx) * fy +
           p[y1 * W + x1] * fx * fy;
}

__global__ void __launch_bounds__(256) k_softargmax_refine(const float* __restrict__ hm, const float* __restrict__ offsets,
                                                           const float* __restrict__ alpha_p, const float* __restrict__ fw_p,
                                                           float* __restrict__ coords, float* __restrict__ scores, int H, int W,
                                                           int radius) {
    __shared__ float red[16];
    const int map = blockIdx.x, n = H * W;
    const float* m = hm + (size_t)map * n;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += blockDim.x) mx = fmaxf(mx, m[i]);
    mx = block_max(mx, red);
    float z = 0.f, sx = 0.f, sy = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float e = __expf(m[i] - mx);
        const int y = i / W;
        z += e;
        sx += e * (float)(i - y * W);
        sy += e * (float)y;
    }
    z = block_sum(z, red);
    sx = block_sum(sx, red);
    sy = block_sum(sy, red);
    if (threadIdx.x != 0) return;
    const float gx = sx / z, gy = sy / z;
    // local softmax centroid of the clipped (2r+1)^2 patch around round-half-even(global)
    const int px = (int)fminf(fmaxf(rintf(gx), 0.f), (float)(W - 1)), py = (int)fminf(fmaxf(rintf(gy), 0.f), (float)(H - 1));
    const int x0 = max(0, px - radius), x1 = min(W, px + radius + 1), y0 = max(0, py - radius), y1 = min(H, py + radius + 1);
    float lm = -INFINITY;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) lm = fmaxf(lm, m[y * W + x]);
    float lz = 0.f, lx = 0.f, ly = 0.f;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            const float e = __expf(m[y * W + x] - lm);
            lz += e;
            lx += e * (float)x;
            ly += e * (float)y;
        }
    const float a = 1.f / (1.f + __expf(-alpha_p[0]));
    float cx = a * gx + (1.f - a) * (lx / lz), cy = a * gy + (1.f - a) * (ly / lz);
    if (offsets) {
        const float fw = 1.f / (1.f + __expf(-fw_p[0]));
        const float* o = offsets + (size_t)map * 2 * n;
        const float ox = bilinear_border(o, H, W, cx, cy), oy = bilinear_border(o + n, H, W, cx, cy);
        cx += fw * ox;
        cy += fw * oy;
    }
    coords[2 * map] = cx;
    coords[2 * map + 1] = cy;
    scores[map] = mx;
}

extern "C" int pk_softargmax_refine_decode(const float* heatmaps, const float* offsets, const float* alpha_param,
                                           const float* fusion_weight_param, float* coords, float* scores, int BK, int H, int W,
                                           int local_radius, void* stream) {
    PK_REQUIRE(heatmaps && alpha_param && coords && scores, "pk_softargmax_refine_decode: null pointer");
    PK_REQUIRE(!offsets || fusion_weight_param, "pk_softargmax_refine_decode: offsets given without fusion_weight");
    PK_REQUIRE(BK > 0 && H > 0 && W > 0 && local_radius >= 0, "pk_softargmax_refine_decode: bad shape");
    hipLaunchKernelGGL(k_softargmax_refine, dim3(BK), dim3(256), 0, (hipStream_t)stream, heatmaps, offsets, alpha_param,
                       fusion_weight_param, coords, scores, H, W, local_radius);
    return pk_launch_status("pk_softargmax_refine_decode");
}

// LocalGaussianRefinement.forward (fusion_head.py:74-128) about coordinates handed in by the caller: softmax-weighted centroid of the
// clipped (2r+1)^2 patch around clamp(round_half_even(c)).  One lane per map (25 pixels).
__global__ void k_local_refine(const float* __restrict__ hm, const float* __restrict__ cin, float* __restrict__ cout, int BK, int H,
                               int W, int radius) {
    const int map = blockIdx.x * blockDim.x + threadIdx.x;
    if (map >= BK) return;
    const float* m = hm + (size_t)map * H * W;
    const int px = (int)fminf(fmaxf(rintf(cin[2 * map]), 0.f), (float)(W - 1)), py = (int)fminf(fmaxf(rintf(cin[2 * map + 1]), 0.f), (float)(H - 1));
    const int x0 = max(0, px - radius), x1 = min(W, px + radius + 1), y0 = max(0, py - radius), y1 = min(H, py + radius + 1);
    float lm = -INFINITY;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) lm = fmaxf(lm, m[y * W + x]);
    float lz = 0.f, lx = 0.f, ly = 0.f;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            const float e = __expf(m[y * W + x] - lm);
            lz += e;
            lx += e * (float)x;
            ly += e * (float)y;
        }
    cout[2 * map] = lx / lz;
    cout[2 * map + 1] = ly / lz;
}
extern "C" int pk_local_gaussian_refine(const float* heatmaps, const float* coords_in, float* coords_out, int BK, int H, int W,
                                        int local_radius, void* stream) {
    PK_REQUIRE(heatmaps && coords_in && coords_out && BK > 0 && H > 0 && W > 0 && local_radius >= 0, "pk_local_gaussian_refine: bad argument");
    hipLaunchKernelGGL(k_local_refine, dim3((BK + 255) / 256), dim3(256), 0, (hipStream_t)stream, heatmaps, coords_in, coords_out, BK, H, W,
                       local_radius);
    return pk_launch_status("pk_local_gaussian_refine");
}

// Backward of SoftArgmax2D.forward (fusion_head.py:24-71): coords = sum softmax(h) (x, y), scores = max h.
//   d h_i = p_i ((x_i - c_x) g_cx + (y_i - c_y) g_cy) + g_score [i == first arg max]
__global__ void __launch_bounds__(256) k_softargmax_bwd(const float* __restrict__ hm, const float* __restrict__ coords,
                                                        const float* __restrict__ scores, const float* __restrict__ gco,
                                                        const float* __restrict__ gsc, float* __restrict__ dhm, int H, int W) {
    __shared__ float red[16];
    __shared__ int first;
    const int map = blockIdx.x, n = H * W;
    const float* m = hm + (size_t)map * n;
    const float mx = scores[map], cx = coords[2 * map], cy = coords[2 * map + 1];
    float z = 0.f;
    int arg = n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = m[i];
        z += __expf(v - mx);
        if (v == mx && i < arg) arg = i;
    }
    z = block_sum(z, red);
    if (threadIdx.x == 0) first = n;
    __syncthreads();
    atomicMin(&first, arg);
    __syncthreads();
    const float zinv = 1.f / z, gx = gco ? gco[2 * map] : 0.f, gy = gco ? gco[2 * map + 1] : 0.f, gs = gsc ? gsc[map] : 0.f;
    float* d = dhm + (size_t)map * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = i / W;
        const float p = __expf(m[i] - mx) * zinv;
        d[i] = p * (((float)(i - y * W) - cx) * gx + ((float)y - cy) * gy) + (i == first ? gs : 0.f);
    }
}
extern "C" int pk_softargmax_bwd(const float* heatmaps, const float* coords, const float* scores, const float* grad_coords,
                                 const float* grad_scores, float* d_heatmaps, int BK, int H, int W, void* stream) {
    PK_REQUIRE(heatmaps && coords && scores && d_heatmaps && BK > 0 && H > 0 && W > 0, "pk_softargmax_bwd: bad argument");
    hipLaunchKernelGGL(k_softargmax_bwd, dim3(BK), dim3(256), 0, (hipStream_t)stream, heatmaps, coords, scores, grad_coords, grad_scores,
                       d_heatmaps, H, W);
    return pk_launch_status("pk_softargmax_bwd");
}

// window_partition / window_reverse (hrformer.py:67-114) as row moves through the window row map (window-order token -> pixel row, -1 = one
// of the zero tokens appended at the bottom / right): gather = partition (pad tokens read as zeros), scatter = reverse (pad tokens dropped).
// Rows are `row_dwords` 4-byte words of any element type.
__global__ void k_rows_by_map(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, const int* __restrict__ map, long n_rows,
                              int row_dwords, int scatter) {
    const long total = n_rows * row_dwords;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / row_dwords;
        const int c = (int)(i - r * row_dwords), pr = map[r];
        if (scatter) {
            if (pr >= 0) dst[(long)pr * row_dwords + c] = src[i];
        } else {
            dst[i] = pr >= 0 ? src[(long)pr * row_dwords + c] : 0u;
        }
    }
}
extern "C" int pk_rows_by_map(const void* src, void* dst, const int* rowmap, int64_t n_rows, int row_bytes, int scatter, void* stream) {
    PK_REQUIRE(src && dst && rowmap && n_rows > 0 && row_bytes > 0 && row_bytes % 4 == 0, "pk_rows_by_map: bad argument (rows of whole dwords)");
    const long total = n_rows * (row_bytes / 4);
    const int nb = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(k_rows_by_map, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)src, (uint32_t*)dst, rowmap, (long)n_rows,
                       row_bytes / 4, scatter);
    return pk_launch_status("pk_rows_by_map");
}

// drop_path (hrformer.py:15-24): out = (x / keep) * mask[sample], mask in {0, 1}; fp32 elements, `per_sample` of them per sample.
__global__ void k_drop_path(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ out, long total, long per_sample,
                            float keep) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        out[i] = (x[i] / keep) * mask[i / per_sample];
}
extern "C" int pk_drop_path_f32(const float* x, const float* mask, float* out, int64_t batch, int64_t per_sample, float keep_prob, void* stream) {
    PK_REQUIRE(x && mask && out && batch > 0 && per_sample > 0 && keep_prob > 0.f, "pk_drop_path_f32: bad argument");
    const long total = batch * per_sample;
    const int nb = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(k_drop_path, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, mask, out, total, (long)per_sample, keep_prob);
    return pk_launch_status("pk_drop_path_f32");
}

// ================================================================================================ D3 tail
__global__ void k_window_refine(const float* __restrict__ hm, const float* __restrict__ cin, float* __restrict__ cout, int BK,
                                int H, int W, int window) {
    const int map = blockIdx.x * blockDim.x + threadIdx.x;
    if (map >= BK) return;
    const float* m = hm + (size_t)map * H * W;
    const int hw = window / 2;
    const int x = (int)cin[2 * map], y = (int)cin[2 * map + 1];  // int() truncation
    const int x0 = max(0, x - hw), x1 = min(W, x + hw + 1), y0 = max(0, y - hw), y1 = min(H, y + hw + 1);
    float ox = cin[2 * map], oy = cin[2 * map + 1];
    if (x1 > x0 && y1 > y0) {
        float s = 0.f, ax = 0.f, ay = 0.f;
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) {
                const float v = m[yy * W + xx];
                s += v;
                ax += v * (float)xx;
                ay += v * (float)yy;
            }
        ox = ax / (s + 1e-8f);
        oy = ay / (s + 1e-8f);
    }
    cout[2 * map] = ox;
    cout[2 * map + 1] = oy;
}
extern "C" int pk_window_refine(const float* heatmaps, const float* coords_in, float* coords_out, int BK, int H, int W,
                                int window, void* stream) {
    PK_REQUIRE(heatmaps && coords_in && coords_out && BK > 0 && H > 0 && W > 0 && window > 0, "pk_window_refine: bad argument");
    hipLaunchKernelGGL(k_window_refine, dim3((BK + 255) / 256), dim3(256), 0, (hipStream_t)stream, heatmaps, coords_in, coords_out,
                       BK, H, W, window);
    return pk_launch_status("pk_window_refine");
}

// fused_decode tail: hp scaled by (sx,sy); if regression given: a = mv/(mv+0.1); out = a*hp + (1-a)*reg*reg_scale
__global__ void k_fused_blend(const float* __restrict__ hp, const float* __restrict__ mv, const float* __restrict__ reg,
                              float* __restrict__ out, int BK, float sx, float sy, float reg_scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BK) return;
    const float hx = hp[2 * i] * sx, hy = hp[2 * i + 1] * sy;
    if (reg) {
        const float a = mv[i] / (mv[i] + 0.1f);
        out[2 * i] = a * hx + (1.f - a) * (reg[2 * i] * reg_scale);
        out[2 * i + 1] = a * hy + (1.f - a) * (reg[2 * i + 1] * reg_scale);
    } else {
        out[2 * i] = hx;
        out[2 * i + 1] = hy;
    }
}
extern "C" int pk_fused_blend(const float* hp, const float* maxvals, const float* regression, float* out, int BK, float sx,
                              float sy, float reg_scale, void* stream) {
    PK_REQUIRE(hp && out && BK > 0 && (!regression || maxvals), "pk_fused_blend: bad argument");
    hipLaunchKernelGGL(k_fused_blend, dim3((BK + 255) / 256), dim3(256), 0, (hipStream_t)stream, hp, maxvals, regression, out, BK,
                       sx, sy, reg_scale);
    return pk_launch_status("pk_fused_blend");
}

__global__ void k_affine_coords(const float* __restrict__ c, const float* __restrict__ center, const float* __restrict__ scale,
                                float* __restrict__ out, int B, int K, float mx, float my, const float* __restrict__ mvals,
                                float thr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const int b = i / K;
    const float keep = mvals ? (mvals[i] > thr ? 1.f : 0.f) : 1.f;
    out[2 * i] = (c[2 * i] * keep) * (scale[2 * b] * mx) + center[2 * b] - scale[2 * b] * 0.5f;
    out[2 * i + 1] = (c[2 * i + 1] * keep) * (scale[2 * b + 1] * my) + center[2 * b + 1] - scale[2 * b + 1] * 0.5f;
}
extern "C" int pk_affine_coords(const float* coords, const float* center, const float* scale, float* out, int B, int K,
                                float mul_x, float mul_y, const float* mask_maxvals, float threshold, void* stream) {
    PK_REQUIRE(coords && center && scale && out && B > 0 && K > 0, "pk_affine_coords: bad argument");
    hipLaunchKernelGGL(k_affine_coords, dim3((B * K + 255) / 256), dim3(256), 0, (hipStream_t)stream, coords, center, scale, out, B,
                       K, mul_x, mul_y, mask_maxvals, threshold);
    return pk_launch_status("pk_affine_coords");
}

// ================================================================================================ D4
__global__ void __launch_bounds__(256) k_flip_merge(const float* __restrict__ a, const float* __restrict__ bf,
                                                    const int32_t* __restrict__ partner, float* __restrict__ out, int K, int H,
                                                    int W) {
    const int map = blockIdx.x, b = map / K, k = map - b * K;
    const float* pa = a + (size_t)map * H * W;
    const float* pb = bf + ((size_t)b * K + partner[k]) * H * W;
    float* po = out + (size_t)map * H * W;
    for (int i = threadIdx.x; i < H * W; i += blockDim.x) {
        const int y = i / W, x = i - y * W;
        po[i] = (pa[i] + pb[y * W + (W - 1 - x)]) / 2.f;
    }
}
extern "C" int pk_flip_merge(const float* a, const float* b_flipped, const int32_t* partner, float* out, int B, int K, int H,
                             int W, void* stream) {
    PK_REQUIRE(a && b_flipped && partner && out && B > 0 && K > 0 && H > 0 && W > 0, "pk_flip_merge: bad argument");
    hipLaunchKernelGGL(k_flip_merge, dim3(B * K), dim3(256), 0, (hipStream_t)stream, a, b_flipped, partner, out, K, H, W);
    return pk_launch_status("pk_flip_merge");
}

// ================================================================================================ video post-processing
// utils/postprocess.py::temporal_smoothing (:187-223): per joint coordinate, edge-padded trajectory convolved (np.convolve,
// i.e. with the kernel FLIPPED) with `w` weights, evaluated in float64 and stored as float32 like the reference's
// numpy-float64 -> float32 tensor assignment.  coords/out: (T, C) with C = 2K; weights: w doubles; w must be odd (the
// reference's output has T+1 entries for even windows and its assignment fails).
__global__ void __launch_bounds__(256) k_temporal_smooth(const float* __restrict__ coords, float* __restrict__ out,
                                                         const double* __restrict__ weights, int T, int C, int w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * C) return;
    const int t = i / C, c = i - t * C, half = w / 2;
    double acc = 0.0;
    for (int j = 0; j < w; ++j) {
        int src = t + (w - 1 - j) - half;                 // padded index t + w-1-j, minus the left pad
        src = src < 0 ? 0 : (src >= T ? T - 1 : src);     // mode='edge'
        acc += (double)coords[(size_t)src * C + c] * weights[j];
    }
    out[i] = (float)acc;
}
extern "C" int pk_temporal_smooth(const float* coords, float* out, const double* weights, int T, int C, int window, void* stream) {
    PK_REQUIRE(coords && out && weights && T > 0 && C > 0 && window > 0, "pk_temporal_smooth: bad argument");
    PK_SUPPORTED((window & 1) == 1, "pk_temporal_smooth: window %d must be odd (the reference fails on even windows)", window);
    hipLaunchKernelGGL(k_temporal_smooth, dim3((T * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, coords, out, weights, T, C, window);
    return pk_launch_status("pk_temporal_smooth");
}

// utils/postprocess.py::nms_pose (:241-267): greedy, order-dependent suppression inside one sample -> one thread per sample
// walks the joints exactly like the reference's Python loop (distances on the ORIGINAL coordinates, `nearby` includes the
// joint itself and already-suppressed joints, first maximum wins ties).
__global__ void __launch_bounds__(64) k_nms_pose(const float* __restrict__ preds, const float* __restrict__ maxvals,
                                                 float* __restrict__ out, uint8_t* __restrict__ keep, int B, int K, float thr) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* p = preds + (size_t)b * K * 2;
    const float* v = maxvals + (size_t)b * K;
    uint8_t* kp = keep + (size_t)b * K;
    for (int k = 0; k < K; ++k) kp[k] = 1;
    for (int k = 0; k < K; ++k) {
        if (!kp[k]) continue;
        int count = 0, best = -1;
        float bv = -INFINITY;
        for (int j = 0; j < K; ++j) {
            const float dx = p[2 * j] - p[2 * k], dy = p[2 * j + 1] - p[2 * k + 1];
            if (sqrtf(dx * dx + dy * dy) < thr) {
                ++count;
                if (v[j] > bv) { bv = v[j]; best = j; }
            }
        }
        if (count > 1)
            for (int j = 0; j < K; ++j) {
                const float dx = p[2 * j] - p[2 * k], dy = p[2 * j + 1] - p[2 * k + 1];
                if (sqrtf(dx * dx + dy * dy) < thr && j != best) kp[j] = 0;
            }
    }
    for (int k = 0; k < K; ++k) {
        const float m = kp[k] ? 1.f : 0.f;
        out[((size_t)b * K + k) * 2] = p[2 * k] * m;
        out[((size_t)b * K + k) * 2 + 1] = p[2 * k + 1] * m;
    }
}
extern "C" int pk_nms_pose(const float* preds, const float* maxvals, float* out, uint8_t* keep, int B, int K, float distance_threshold,
                           void* stream) {
    PK_REQUIRE(preds && maxvals && out && keep && B > 0 && K > 0, "pk_nms_pose: bad argument");
    hipLaunchKernelGGL(k_nms_pose, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, preds, maxvals, out, keep, B, K, distance_threshold);
    return pk_launch_status("pk_nms_pose");
}

// ================================================================================================ f1: evaluator records
// COCOEvaluator.update (utils/metrics.py:84-106): per instance a (K,3) array [x, y, score] and the instance score = mean of the
// strictly positive keypoint scores (0 when there is none).  One 64-lane wave per instance; the mean is an fp32 sum in keypoint
// order divided by the count (numpy's float32 mean sums pairwise: equal to 1e-6 relative, stated in the test).
__global__ void __launch_bounds__(64) k_pose_records(const float* __restrict__ kp, const float* __restrict__ sc, float* __restrict__ rec,
                                                    float* __restrict__ inst, int K) {
    const int b = blockIdx.x, t = threadIdx.x;
    for (int k = t; k < K; k += 64) {
        rec[((size_t)b * K + k) * 3 + 0] = kp[((size_t)b * K + k) * 2 + 0];
        rec[((size_t)b * K + k) * 3 + 1] = kp[((size_t)b * K + k) * 2 + 1];
        rec[((size_t)b * K + k) * 3 + 2] = sc[(size_t)b * K + k];
    }
    if (t == 0) {
        float s = 0.f;
        int n = 0;
        for (int k = 0; k < K; ++k) {
            const float v = sc[(size_t)b * K + k];
            if (v > 0.f) {
                s += v;
                ++n;
            }
        }
        inst[b] = n > 0 ? s / (float)n : 0.f;
    }
}
extern "C" int pk_pose_records(const float* keypoints, const float* scores, float* records, float* instance_score, int B, int K,
                               void* stream) {
    PK_REQUIRE(keypoints && scores && records && instance_score && B > 0 && K > 0, "pk_pose_records: bad argument");
    hipLaunchKernelGGL(k_pose_records, dim3(B), dim3(64), 0, (hipStream_t)stream, keypoints, scores, records, instance_score, K);
    return pk_launch_status("pk_pose_records");
}

// ================================================================================================ f2: input pipeline
// TopdownAffine(+rotation) image crop + RandomFlip + ToTensor/normalise of the reference's data pipeline (datasets/transforms.py:42-47,
// 128-131, 212-217; datasets/coco_dataset.py:156-163; inference.py:93-110) for a whole batch in ONE launch, straight from the decoded
// uint8 images to the network input.  The warp follows OpenCV's 8-bit `warpAffine(INTER_LINEAR, BORDER_CONSTANT 0)` integer algorithm as
// restated in oracle/warp.py (coordinates in 10-bit fixed point rounded half-to-even, 5 bits of sub-pixel position, weights that sum to
// 2^15, (sum + 2^14) >> 15) -- integer work, bit-exact against that restatement; cv2 itself is not in the image (parity unpinned vs cv2).
// Normalisation is the reference's fp32 expression ((v / 255) - mean) / std with correctly rounded divisions.
struct CropDesc {
    int64_t src_offset;      // byte offset of this sample's (H, W, 3) uint8 image inside the source buffer
    int32_t H, W;
    int32_t flip;            // mirror the source columns (RandomFlip flips the image before the warp)
    int32_t bgr;             // source is BGR (cv2.imread / inference.py:83): swap to RGB
    double minv[6];          // inverse (destination -> source) affine matrix, float64 as OpenCV holds it
};
__global__ void __launch_bounds__(256) k_affine_crop(const unsigned char* __restrict__ src, const CropDesc* __restrict__ desc, int out_w, int out_h,
                                                     float* __restrict__ out_nchw, uint16_t* __restrict__ out_nhwc8, float m0, float m1, float m2,
                                                     float s0, float s1, float s2) {
    const CropDesc d = desc[blockIdx.y];
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= out_w * out_h) return;
    const int y = pix / out_w, x = pix - y * out_w;
    auto sat = [](double v) { return (long long)max(-2147483648.0, min(2147483647.0, rint(v))); };
    const long long adelta = sat(__dmul_rn(__dmul_rn(d.minv[0], (double)x), 1024.0)), bdelta = sat(__dmul_rn(__dmul_rn(d.minv[3], (double)x), 1024.0));
    const long long X0 = sat(__dmul_rn(__dadd_rn(__dmul_rn(d.minv[1], (double)y), d.minv[2]), 1024.0)) + 16;
    const long long Y0 = sat(__dmul_rn(__dadd_rn(__dmul_rn(d.minv[4], (double)y), d.minv[5]), 1024.0)) + 16;
    const long long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    const long long sx = max(-32768LL, min(32767LL, X >> 5)), sy = max(-32768LL, min(32767LL, Y >> 5));
    const int a = (int)(X & 31), b = (int)(Y & 31);
    const int w4[4] = {(32 - a) * (32 - b) * 32, a * (32 - b) * 32, (32 - a) * b * 32, a * b * 32};
    int acc[3] = {0, 0, 0};
    const unsigned char* img = src + d.src_offset;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const long long xx = sx + (t & 1), yy = sy + (t >> 1);
        if (xx < 0 || xx >= d.W || yy < 0 || yy >= d.H) continue;
        const long long xs = d.flip ? d.W - 1 - xx : xx;
        const unsigned char* p = img + (yy * d.W + xs) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += w4[t] * (int)p[d.bgr ? 2 - c : c];
    }
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int u = min(255, max(0, (acc[c] + (1 << 14)) >> 15));
        v[c] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)u, 255.0f), mean[c]), sd[c]);
    }
    const size_t plane = (size_t)out_w * out_h;
    if (out_nchw) {
#pragma unroll
        for (int c = 0; c < 3; ++c) out_nchw[((size_t)blockIdx.y * 3 + c) * plane + pix] = v[c];
    }
    if (out_nhwc8) {
        uint4 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], 0.f);
        o.z = o.w = 0u;
        *reinterpret_cast<uint4*>(out_nhwc8 + ((size_t)blockIdx.y * plane + pix) * 8) = o;
    }
}
extern "C" int pk_affine_crop_normalize(const void* src_u8, const void* desc_table, int n_samples, int out_w, int out_h, float* out_nchw_f32,
                                        void* out_nhwc8_bf16, const float* mean3, const float* std3, void* stream) {
    PK_REQUIRE(src_u8 && desc_table && n_samples > 0 && out_w > 0 && out_h > 0 && (out_nchw_f32 || out_nhwc8_bf16) && mean3 && std3,
               "pk_affine_crop_normalize: bad argument");
    PK_REQUIRE(!out_nhwc8_bf16 || (((uintptr_t)out_nhwc8_bf16) & 15) == 0, "pk_affine_crop_normalize: 16-byte alignment of the NHWC output");
    hipLaunchKernelGGL(k_affine_crop, dim3((out_w * out_h + 255) / 256, n_samples), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src_u8,
                       (const CropDesc*)desc_table, out_w, out_h, out_nchw_f32, (uint16_t*)out_nhwc8_bf16, mean3[0], mean3[1], mean3[2], std3[0],
                       std3[1], std3[2]);
    return pk_launch_status("pk_affine_crop_normalize");
}
