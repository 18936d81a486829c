// Implicit-GEMM on MFMA (bf16 in, fp32 accumulate) for every dense contraction of the network:
//   3x3 / 1x1 convolutions (forward, and data-gradient as a convolution with flipped weights), and linear layers
//   with an optional row gather/scatter (window partition / reverse of the HRFormer blocks).
//
//   out[m][n] = sum_{t < T} sum_{c < Cin}  A(m, t, c) * Wt[n][t][c]          m < M, n < N
//     conv   : m = (b,oy,ox), t = (kh,kw), A = X[b][oy*s+kh-p][ox*s+kw-p][c] (0 outside; NHWC)
//     dilated: A = X[b][(oy+kh-p)/2][(ox+kw-p)/2][c] only where both numerators are even (stride-2 dgrad)
//     linear : T = 1, A = X[rowmap ? rowmap[m] : m][c]   (rowmap -1 -> zero row: the reference's zero pad tokens)
//
// Tiling (one 256-thread workgroup = 4 waves): BM=128 output rows x BN (32/64/128) output columns, K-step 32
// (one MFMA k), per filter tap.  Operands are staged global -> registers -> LDS (double-buffered, one barrier per
// K-step, rows padded to 80 B against bank conflicts); fragments are read with 16-byte ds_read and fed to
// v_mfma_f32_16x16x32_bf16 with the WEIGHT tile as the A operand, so a lane's 4 accumulator registers are 4
// consecutive output channels of one pixel -> one 8-byte NHWC store per tile per lane.
//
// Epilogue options: bias, exact-erf GELU, residual add with a per-sample scale (DropPath), row scatter, bf16 or
// fp32 output, NCHW-planar fp32 output (head), softplus, and per-tile column sums / sums of squares of the fp32
// accumulators (train-mode BatchNorm statistics, reduced later in fixed order -> deterministic).
#include "pk_common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define STAT_ROWS 128      // pixel rows per BatchNorm partial-statistics tile (pk_conv_stats_tiles)
#define BKK 32
#define LDS_PITCH 40  // bf16 elements per LDS row (32 + 8 pad) = 80 bytes, keeps 16-byte alignment

struct IgemmArgs {
    const uint16_t* x;        // activations, bf16 NHWC [B][Hs][Ws][Cin] or [rows][Cin]
    const uint16_t* w;        // weights bf16 [N][T][Cin]
    void* out;                // see out_mode
    const float* bias;        // [N] or null
    const float* col_scale;   // [N] or null: the accumulator is multiplied by it before the bias (eval-mode BatchNorm folded into the conv)
    const uint16_t* res;      // residual, same layout as a bf16 row-major output, or null
    const float* res_scale;   // per-sample multiplier of the GEMM result before the residual add, [B] or null
    const int32_t* a_rowmap;  // [M] source row or -1 (linear mode only), or null
    const int32_t* o_rowmap;  // [M] destination row or -1 (skip), or null
    float* stats;             // [gridDim.x][2][N] per-M-tile column sums / sums of squares, or null
    uint16_t* preact;         // optional bf16 row-major copy of (acc + bias) BEFORE the activation (saved for GELU backward)
    const uint16_t* gelu_of;  // optional bf16 row-major z: result is multiplied by gelu'(z) (backward through GELU)
    int M, N, Cin, T;         // T = 1 or 9
    int Hs, Ws;               // source spatial size (conv)
    int Ho, Wo;               // output spatial size (conv)
    int stride, pad, dilated; // conv geometry
    int ldo;                  // output row pitch in elements (row-major modes)
    int rows_per_sample;      // rows of one sample (for res_scale): Ho*Wo or tokens per image
    int act;                  // 0 none, 1 GELU(erf), 2 softplus, 3 ReLU applied AFTER the residual add
    int out_mode;             // 0 bf16 row-major, 1 fp32 row-major, 2 fp32 NCHW planes [B][N][Ho*Wo]
    int xcd_remap;            // set by igemm_launch (PK_IGEMM_XCD=0 disables the XCD-contiguous tile order)
    int vec8;                 // set by igemm_launch: row-major pointers 16-byte aligned and ldo % 8 == 0 -> 16-byte epilogue I/O
    int chunk_major;          // set by igemm_launch: K order = all taps of one channel chunk back to back (N <= 32 tiles of deep 3x3 convs)
    int dil_group;            // set by igemm_launch (stride-2 data gradients): output pixels enumerated parity class by parity class
};

// ================================================================================================ main kernel (v2)
// Same contraction and epilogue as above, restructured so the K loop is MFMA-bound instead of issue-bound:
//   * filter taps are the OUTER loop: per tap each thread computes ONE byte offset per staged row (bounds, dilation);
//     the inner loop over channel chunks only bumps a scalar offset;
//   * every global read is a branch-free `buffer_load_dwordx4`: an out-of-image tap, a row beyond M, a pad token or a
//     channel beyond Cin gets the offset 0x80000000 >= num_records and the hardware returns zeros;
//   * LDS tiles are unpadded and XOR-swizzled per 16-byte chunk (conflict-free ds_read_b128 for both K-steps:
//     BK=64: chunk ^= (row>>1)&7, BK=32: chunk ^= (row>>3)<<1, found by exhaustive search over the b128 lane groups).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define OOB_OFF 0x80000000u

template <int BK>
__device__ __forceinline__ int swz(int row) { return BK == 64 ? ((row >> 1) & 7) : (((row >> 3) & 1) << 1); }
// floor(q / d) for 0 <= q < 2^24 (d >= 1): float estimate + one correction, ~8 VALU (a 32-bit integer division by a run-time divisor
// compiles to ~35; the pixel decompositions of the output-bound conv launches spent more issue slots on them than on MFMAs)
__device__ __forceinline__ int fast_div24(int q, int d, float inv) {
    int r = (int)((float)q * inv);
    const int rem = q - r * d;
    r += rem >= d ? 1 : (rem < 0 ? -1 : 0);
    return r;
}

// Epilogue of ONE output row segment: this lane owns output row `orow` (pixel / token) and the 8 consecutive columns
// n..n+7, handed over as two float4 read back from the LDS staging tile.  The accumulators are staged through LDS so
// that (a) every global access of the epilogue is a 16-byte access and a wave covers whole 128/256-byte row segments
// (the MFMA C layout would give 8-byte pieces of 16 different rows per instruction), and (b) the epilogue is ONE rolled
// loop: unrolled per accumulator tile it was ~2000 instructions x 16 tiles (erff inlined 64 times, 250 KB of code for
// one kernel), which made the K=32 GEMMs instruction-fetch bound.  Only static indexing of v[] (no scratch).
__device__ __forceinline__ float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ void igemm_epilogue_row8(const IgemmArgs& p, f32x4 lo, f32x4 hi, f32x4 slo, f32x4 shi, f32x4 blo, f32x4 bhi, int orow, int n,
                                                    float rs, int hw_out) {
    if (p.col_scale) {                                    // per-column scale of this lane's 8 columns, then the bias (zeros when absent)
        lo = lo * slo + blo;
        hi = hi * shi + bhi;
    } else {
        lo += blo;
        hi += bhi;
    }
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    const int nv = p.N - n;                               // valid columns of this segment (>= 1)
    const bool vec = p.vec8 && nv >= 8;
    const size_t base = (size_t)orow * p.ldo + n;
    if (p.preact) {
        if (vec) {
            *reinterpret_cast<u32x4*>(p.preact + base) =
                (u32x4){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nv) p.preact[base + j] = f32_to_bf16(v[j]);
        }
    }
    if (p.gelu_of) {
        if (vec) {
            const u32x4 zz = *reinterpret_cast<const u32x4*>(p.gelu_of + base);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[2 * j] *= gelu_grad(bf16_lo(zz[j]));
                v[2 * j + 1] *= gelu_grad(bf16_hi(zz[j]));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nv) v[j] *= gelu_grad(bf16_to_f32(p.gelu_of[base + j]));
        }
    }
    if (p.act == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_erf(v[j]);
    } else if (p.act == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = softplus_(v[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= rs;
    if (p.out_mode == 2) {
        const int bb = orow / hw_out, pix = orow - bb * hw_out;
        float* o = reinterpret_cast<float*>(p.out) + ((size_t)bb * p.N + n) * hw_out + pix;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < nv) o[(size_t)j * hw_out] = v[j];
        return;
    }
    if (p.res) {
        if (vec) {
            const u32x4 rv = *reinterpret_cast<const u32x4*>(p.res + base);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[2 * j] += bf16_lo(rv[j]);
                v[2 * j + 1] += bf16_hi(rv[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nv) v[j] += bf16_to_f32(p.res[base + j]);
        }
    }
    if (p.act == 3) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    if (p.out_mode == 0) {
        uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + base;
        if (vec) {
            *reinterpret_cast<u32x4*>(o) =
                (u32x4){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nv) o[j] = f32_to_bf16(v[j]);
        }
    } else {
        float* o = reinterpret_cast<float*>(p.out) + base;
        if (vec) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nv) o[j] = v[j];
        }
    }
}

// LEAN = 1: the token GEMMs of the HRFormer blocks only (linear rows with optional gather / scatter maps, bias, residual with a per-sample
// scale, bf16 output) -- the convolution addressing, the activations, the fp32 / NCHW outputs and the statistics are compiled out.  The
// full kernel is 27-30 KB of code; a small launch (100-900 workgroups, 1-4 K-steps) spends much of its few microseconds fetching it cold.
// The kernel body takes its tile coordinates and grid shape as arguments: k_igemm2 passes blockIdx / gridDim, the GROUPED launch
// k_igemm2g (below) cuts one 1-D grid into the tile grids of several independent problems.
template <int BM, int BN, int WM, int WN, int BK, int LEAN = 0>
__device__ __forceinline__ void igemm2_body(const IgemmArgs& p_in, const int bid_x, const int bid_y, const int grid_x, const int grid_y) {
    IgemmArgs p = p_in;
    if (LEAN == 3) {
        // every feature of the epilogue, but the PLAIN addressing in the K loop: no dilated gather (stride-2 data gradients), no parity
        // grouping, no chunk-major order.  The generic loop body is 681 instructions around 8 MFMAs on the 128 x 32 tile (295 VALU, 360 SALU,
        // 52 branches: the run-time flags of those three modes are tested in every advance / set_tap) -- ~0.8 us per K-step whatever the
        // tile does, which is what bounds the deep contractions of the low-resolution branches.  With the flags constant the compiler drops it.
        p.dilated = 0; p.dil_group = 0;       // (chunk-major stays a run-time flag of the 128 x 32 tiles: with the mask form a tap change per step is cheap)
    } else if (LEAN == 4) {          // the stride-2 data gradients (dilated gather, grouped or not): no chunk-major order
        p.dilated = 1; p.chunk_major = 0;
    } else if (LEAN) {               // LEAN = 2 keeps the GELU epilogues (fc1: GELU + saved pre-activation; fc2 data gradient: x gelu'(z))
        p.stats = nullptr;
        p.out_mode = 0; p.T = 1; p.Ho = 0; p.Wo = 0; p.dilated = 0; p.vec8 = 1; p.chunk_major = 0; p.dil_group = 0;
        p.col_scale = nullptr;
        if (LEAN == 1) { p.preact = nullptr; p.gelu_of = nullptr; p.act = 0; }
        else if (p.act != 1) p.act = 0;
    }
    constexpr int MI = BM / WM / 16, NI = BN / WN / 16;
    constexpr int CH = BK / 8;                       // 16-byte chunks per tile row
    constexpr int A_PT = BM * CH / 256;              // A chunks per thread (2 or 4)
    constexpr int B_PT = (BN * CH + 255) / 256;      // B chunks per thread (1..4)
    constexpr int KS = BK / 32;                      // MFMA k-steps per staged tile
    constexpr int TM = BM / WM, TN = BN / WN;        // tile of one wave
    constexpr int EP = TN + 4;                       // fp32 pitch of the epilogue staging tile (conflict-free b128 rows)
    constexpr int BPP = (MI > 4) ? 2 : MI;           // 16-row accumulator blocks staged per epilogue pass (all of them for BM = 128)
    constexpr int MAIN_HALFS = 2 * BM * BK + 2 * BN * BK, EPI_HALFS = 4 * (BPP * 16) * EP * 2;
    __shared__ __attribute__((aligned(16))) uint16_t smem[MAIN_HALFS > EPI_HALFS ? MAIN_HALFS : EPI_HALFS];
    uint16_t* sA = smem;                             // [2][BM*BK]
    uint16_t* sB = smem + 2 * BM * BK;               // [2][BN*BK]
    float* sStat = reinterpret_cast<float*>(smem);   // [WM][BN][2], reused after the main loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order.  Workgroups are handed to the 8 XCDs round-robin by linear id, and each XCD has its own 4 MB
    // L2: with the plain mapping the two N-tiles of one pixel tile (same input!) and the halo-sharing neighbours of a
    // 3x3 conv land on different L2s (measured: 11x the input bytes cross the fabric for the 256->256 head conv).  Give
    // every XCD one contiguous chunk of the tile sequence, N-tile index fastest.
    int mt = bid_x, nt = bid_y;
    {
        const int G = grid_x * grid_y, L = bid_x + grid_x * bid_y;
        if ((G & 7) == 0 && p.xcd_remap) {
            const int tile = (L & 7) * (G >> 3) + (L >> 3);
            mt = tile / grid_y;
            nt = tile - mt * grid_y;
        }
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const bool linear = (p.T == 1 && p.Ho == 0);
    const int kw_n = (p.T == 9) ? 3 : 1;
    const int kc = tid % CH;                         // this thread's chunk column (same for all its rows: 256 % CH == 0)
    const int row0 = tid / CH;                       // first staged row; further rows are +256/CH apart
    constexpr int RSTEP = 256 / CH;
    // LDS-DMA staging (tiles with BN >= 128: the MFMA-bound layers): `buffer_load_dwordx4 ... lds` writes 64 lanes x 16 bytes
    // to LDS at a wave-uniform base + lane * 16, with no VGPR destination and no ds_write.  The lanes of a wave stage 64/CH
    // consecutive rows x CH chunks, which is exactly lane-linear in the unpadded [row][BK] tile; the XOR swizzle therefore moves
    // to the SOURCE side: the lane at physical chunk position kc fetches logical chunk kc ^ swz(row) (the fragment reads are
    // unchanged; swz depends on row bits below the row step, so one value per thread).  Frees 24-44 VGPRs per tile variant.
    constexpr bool DMA = (BN >= 128);
    const int kcg = DMA ? (kc ^ swz<BK>(row0)) : kc;   // chunk index on the global side (swz depends on row bit 3 only: same for row0 + 64 i)

    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, 0x7ffffff0, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, 0x7ffffff0, 0x00020000);

    // Stride-2 data gradient (dilated gather): only the taps whose parity matches the output pixel's contribute -- 1, 2, 2 or 4 of the 9,
    // by parity class (oy & 1, ox & 1).  With p.dil_group the GEMM rows enumerate the output pixels class by class (class-major, then
    // sample, then row, then column of the class), so that a tile of BM rows lies in ONE class (except where two classes meet) and the
    // K loop runs over that class's taps only: 2.25 K-steps per channel chunk on average instead of 9 (the other 6.75 multiplied zeros).
    const bool grouped = !linear && p.dilated && p.dil_group;
    const int gny0 = (p.Ho + 1) >> 1, gny1 = p.Ho >> 1, gnx0 = (p.Wo + 1) >> 1, gnx1 = p.Wo >> 1, gB = grouped ? p.M / (p.Ho * p.Wo) : 0;
    const int gs1 = gB * gny0 * gnx0, gs2 = gs1 + gB * gny0 * gnx1, gs3 = gs2 + gB * gny1 * gnx0;      // first row of classes (0,1), (1,0), (1,1)
    auto group_class = [&](int m) { return (m >= gs1 ? 1 : 0) + (m >= gs2 ? 1 : 0) + (m >= gs3 ? 1 : 0); };
    const bool small_m24 = p.M < (1 << 24);          // exact range of fast_div24
    auto idiv = [&](int q, int d) { return small_m24 ? fast_div24(q, d, 1.f / (float)d) : q / d; };
    auto group_pixel = [&](int m, int& b, int& oy, int& ox) {
        const int c = group_class(m), cy = c >> 1, cx = c & 1;
        const int ny = cy ? gny1 : gny0, nx = cx ? gnx1 : gnx0;
        const int r = m - (c == 0 ? 0 : (c == 1 ? gs1 : (c == 2 ? gs2 : gs3)));
        b = idiv(r, ny * nx);
        const int r2 = r - b * ny * nx, y = idiv(r2, nx);
        oy = 2 * y + cy;
        ox = 2 * (r2 - y * nx) + cx;
    };
    unsigned tapmask = 0x1ffu;          // taps this tile walks (bit kh * 3 + kw)
    if (grouped) {
        const int c0 = group_class(m0), c1 = group_class(min(m0 + BM, p.M) - 1);
        if (c0 == c1) {                 // input index = (oy + kh - 1) / 2 must be whole: kh = 1 for even oy, kh in {0, 2} for odd oy
            const unsigned rows = (c0 >> 1) ? 0x5u : 0x2u, cols = (c0 & 1) ? 0x5u : 0x2u;
            tapmask = 0;
            for (int kh = 0; kh < 3; ++kh)
                if (rows >> kh & 1) tapmask |= cols << (3 * kh);
        }
    }
    // ---- fixed per-thread row descriptors
    int a_base[A_PT], a_iy[A_PT], a_ix[A_PT];
    bool a_ok[A_PT];
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
        const int m = m0 + row0 + RSTEP * i;
        a_ok[i] = m < p.M;
        a_base[i] = 0; a_iy[i] = 0; a_ix[i] = 0;
        if (a_ok[i]) {
            if (linear) {
                const int src = p.a_rowmap ? p.a_rowmap[m] : m;
                a_ok[i] = src >= 0;
                a_base[i] = src * p.Cin;
            } else {
                int b, oy, ox;
                if (grouped) {
                    group_pixel(m, b, oy, ox);
                } else {
                    const int hw = p.Ho * p.Wo;
                    b = idiv(m, hw);
                    const int r = m - b * hw;
                    oy = idiv(r, p.Wo);
                    ox = r - oy * p.Wo;
                }
                a_iy[i] = oy * p.stride - p.pad;
                a_ix[i] = ox * p.stride - p.pad;
                a_base[i] = b * p.Hs * p.Ws * p.Cin;
            }
        }
    }
    // Plain convolutions (LEAN = 3): per staged row ONE signed byte offset of its (virtual) top-left tap and a 9-bit mask of the taps that
    // fall inside the image; a tap change is then a scalar delta + bit test + select per row instead of re-deriving and bounds-checking
    // (iy, ix) with divergent branches (~35 instructions per row, on every K-step for 64-channel inputs).
    int a_off0[A_PT];
    unsigned a_tmask[A_PT];
    if (LEAN == 3 && !linear) {
#pragma unroll
        for (int i = 0; i < A_PT; ++i) {
            a_off0[i] = (a_base[i] + (a_iy[i] * p.Ws + a_ix[i]) * p.Cin + kcg * 8) * 2;
            unsigned mk = 0u;
            if (a_ok[i]) {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        if (kh < kw_n && kw < kw_n && a_iy[i] + kh >= 0 && a_iy[i] + kh < p.Hs && a_ix[i] + kw >= 0 && a_ix[i] + kw < p.Ws)
                            mk |= 1u << (kh * kw_n + kw);
            }
            a_tmask[i] = mk;
        }
    }
    unsigned b_off[B_PT];
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
        const int q = tid + 256 * i, n = n0 + q / CH;
        b_off[i] = (q < BN * CH && n < p.N) ? (unsigned)((n * p.T * p.Cin + kcg * 8) * 2) : OOB_OFF;
    }
    const int kchunks = (p.Cin + BK - 1) / BK;
    const int nk = (grouped ? __builtin_popcount(tapmask) : p.T) * kchunks;
    auto next_tap = [&](int t) {          // the next tap after t that this tile walks
        ++t;
        if (grouped) {                    // first set bit of the tap mask at or above t (a `while` here was unrolled x8 into ~120 scalar
            const unsigned m = tapmask >> t;                                  // instructions in the K loop of every stride-2 data gradient)
            t = m ? t + __builtin_ctz(m) : 9;
        }
        return t;
    };
    const int t_first = grouped ? next_tap(-1) : 0;

    unsigned a_voff[A_PT];           // byte offset of (row, current tap, channel kc*8), or OOB
    auto set_tap = [&](int t) {
        if (LEAN == 3 && !linear) {
            const int kh3 = kw_n == 3 ? (t * 11) >> 5 : 0, kw3 = t - kh3 * kw_n;          // t / 3 for t < 9
            const int delta = (kh3 * p.Ws + kw3) * p.Cin * 2;
#pragma unroll
            for (int i = 0; i < A_PT; ++i) a_voff[i] = ((a_tmask[i] >> t) & 1u) ? (unsigned)(a_off0[i] + delta) : OOB_OFF;
            return;
        }
        const int kh = kw_n == 3 ? (t * 11) >> 5 : 0, kw = t - kh * kw_n;          // t / 3 for t < 9 (no integer division in the loop)
        if (linear) {
#pragma unroll
            for (int i = 0; i < A_PT; ++i) a_voff[i] = a_ok[i] ? (unsigned)((a_base[i] + kcg * 8) * 2) : OOB_OFF;
            return;
        }
        // branch-free per row (the nested ifs compiled to a divergent branch pair per row and condition: ~35 instructions per row)
        const int dsh = p.dilated ? 1 : 0;
#pragma unroll
        for (int i = 0; i < A_PT; ++i) {
            const int sy = a_iy[i] + kh, sx = a_ix[i] + kw;
            const int iy = sy >> dsh, ix = sx >> dsh;
            const bool ok = a_ok[i] & ((((sy | sx) & dsh) == 0)) & ((unsigned)iy < (unsigned)p.Hs) & ((unsigned)ix < (unsigned)p.Ws);
            a_voff[i] = ok ? (unsigned)((a_base[i] + (iy * p.Ws + ix) * p.Cin + kcg * 8) * 2) : OOB_OFF;
        }
    };
    constexpr bool DEEP = (BN >= 128 && BK == 64 && !DMA);    // global loads issued TWO tiles ahead (second register set), see the main loop
    // Fragment double-buffering (below) and the second global-load register set together need > 256 registers (occupancy 1:
    // head conv 526 us).  Measured one at a time on the 3x3 256->256 head conv (fwd / dgrad): neither 370 / 308 us, fragment
    // prefetch 352 / 302 us, deep global prefetch 359 / 302 us -- both hide latency, neither removes the ceiling of this
    // tiling (64x64 wave tiles: 768 LDS cycles per 512 MFMA cycles and K-step).  The deep variant is kept for BK = 64.
    constexpr bool FRAG_PREFETCH = !DEEP;
    u32x4 ra[A_PT], rb[B_PT], ra2[DEEP ? A_PT : 1], rb2[DEEP ? B_PT : 1];
    auto dma_tiles = [&](int buf, int t, int c0) {       // global -> LDS without registers (DMA variant only)
#if defined(__HIP_DEVICE_COMPILE__)    // the builtin needs a gfx950 target feature: the host pass of hipcc must not see it
        c0 = __builtin_amdgcn_readfirstlane(c0);
        t = __builtin_amdgcn_readfirstlane(t);
        const bool c_ok = (c0 + kcg * 8) < p.Cin;
#pragma unroll
        for (int i = 0; i < A_PT; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(&sA[buf * BM * BK + (wave * (64 / CH) + RSTEP * i) * BK]),
                                                     16, c_ok ? a_voff[i] : OOB_OFF, c0 * 2, 0, 0);
#pragma unroll
        for (int i = 0; i < B_PT; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(&sB[buf * BN * BK + (wave * (64 / CH) + RSTEP * i) * BK]),
                                                     16, c_ok ? b_off[i] : OOB_OFF, (t * p.Cin + c0) * 2, 0, 0);
#endif
    };
    auto load_tiles = [&](u32x4* ra, u32x4* rb, int t, int c0) {       // c0: first channel of this K-chunk (wave-uniform)
        c0 = __builtin_amdgcn_readfirstlane(c0);        // scalar offset operands of the loads below: SGPRs, not a waterfall loop per load
        t = __builtin_amdgcn_readfirstlane(t);
        const bool c_ok = (c0 + kcg * 8) < p.Cin;
#pragma unroll
        for (int i = 0; i < A_PT; ++i)
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, c_ok ? a_voff[i] : OOB_OFF, c0 * 2, 0);
#pragma unroll
        for (int i = 0; i < B_PT; ++i)
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, c_ok ? b_off[i] : OOB_OFF, (t * p.Cin + c0) * 2, 0);
    };
    auto store_tiles = [&](const u32x4* ra, const u32x4* rb, int buf) {
#pragma unroll
        for (int i = 0; i < A_PT; ++i) {
            const int row = row0 + RSTEP * i;
            *reinterpret_cast<u32x4*>(&sA[buf * BM * BK + row * BK + ((kc ^ swz<BK>(row)) * 8)]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_PT; ++i) {
            const int q = tid + 256 * i, row = q / CH;
            if (q < BN * CH) *reinterpret_cast<u32x4*>(&sB[buf * BN * BK + row * BK + ((kc ^ swz<BK>(row)) * 8)]) = rb[i];
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int t_next = __builtin_amdgcn_readfirstlane(t_first), c_next = 0;     // (tap, chunk) of the tile being loaded
    // (Measured and dropped: chunk-major order -- all nine taps of one channel chunk back to back, so that the shifted re-reads of the
    // same pixels stay in L2 instead of cycling ~11 MB per XCD between two taps (PMC: the head conv fetches 4.1x its input).  The row
    // offsets must then be recomputed every step: head conv forward 335 -> 400 us, dgrad 290 -> 360 us; only N = 32 tiles gained.)
    // ... kept for BN = 32 with nine taps and >= 128 input channels (p.chunk_major, set by igemm_launch): 3x3 256 -> 32 @64x48 fetched
    // 681 MB for its 100 MB input at 5.8 TB/s of fabric traffic, 118 us.)
    // (tap, chunk) are the same for every lane, but the compiler's divergence analysis loses that across the lambdas: it kept c_next in a
    // VGPR, compared it with VALU instructions and -- since it is the scalar offset operand of every staging load -- wrapped EACH buffer load
    // of the K loop in a waterfall loop (v_readfirstlane / v_cmp_eq / s_and_saveexec / load / branch).  An explicit readfirstlane after every
    // update makes them SGPR values again: scalar compares, scalar branches, plain loads.
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto advance = [&]() {                // next (tap, chunk) in contraction order; recomputes the row offsets on a tap change
        if (BN == 32 && p.chunk_major) {
            if (++t_next == p.T) {
                t_next = 0;
                c_next += BK;
            }
            t_next = uni(t_next);
            c_next = uni(c_next);
            set_tap(t_next);
            return;
        }
        c_next = uni(c_next + BK);
        if (c_next >= p.Cin) {
            c_next = 0;
            t_next = uni(next_tap(t_next));
            set_tap(t_next);
        }
    };
    const int frow = lane & 15, fkc = lane >> 4;
    auto compute = [&](int buf) {
        // (Measured and dropped for the DMA tiles: fragment reads through inline assembly, as in the weight-gradient rings below, so
        // that the compiler's `s_waitcnt vmcnt(0)` in front of the first ds_read -- it cannot prove that the tile requested a moment
        // ago is a different LDS buffer -- disappears and a workgroup overlaps its own loads with its own MFMAs: head conv forward
        // 293 -> 327 us, dgrad 264 -> 273 us.  With two workgroups per CU the other workgroup already covers the loads, and the
        // hand-placed lgkmcnt(0) fences schedule worse than the compiler's counted waits.  The bound of this tiling is LDS read
        // bandwidth: 12 fragment reads per 32 MFMAs and wave (16x16x32 tiles); a 32x32x16 tiling would halve it.)
        if constexpr (BN >= 128 && KS == 2 && FRAG_PREFETCH) {
            // All fragments of the staged tile are read up front into separate registers (KS x (NI + MI) x 4 VGPRs; the kernel
            // sits at 2 waves/SIMD either way) and the scheduler is told to interleave the second K-step's LDS reads with the
            // first K-step's MFMAs: left alone it recycled six fragment registers and exposed an LDS round trip every 4-8 MFMAs.
            bf16x8 wf[KS][NI], af[KS][MI];
    #pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
    #pragma unroll
                for (int a = 0; a < NI; ++a) {
                    const int row = wn * (BN / WN) + a * 16 + frow;
                    wf[ks][a] = *reinterpret_cast<const bf16x8*>(&sB[buf * BN * BK + row * BK + (((fkc + 4 * ks) ^ swz<BK>(row)) * 8)]);
                }
    #pragma unroll
                for (int b = 0; b < MI; ++b) {
                    const int row = wm * (BM / WM) + b * 16 + frow;
                    af[ks][b] = *reinterpret_cast<const bf16x8*>(&sA[buf * BM * BK + row * BK + (((fkc + 4 * ks) ^ swz<BK>(row)) * 8)]);
                }
            }
    #pragma unroll
            for (int ks = 0; ks < KS; ++ks)
    #pragma unroll
                for (int a = 0; a < NI; ++a)
    #pragma unroll
                    for (int b = 0; b < MI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][a], af[ks][b], acc[a][b], 0, 0, 0);
            if (KS == 2 && NI * MI >= 8) {
                __builtin_amdgcn_sched_group_barrier(0x100, NI + MI, 0);                 // DS reads: fragments of K-step 0
    #pragma unroll
                for (int g = 0; g < (NI + MI) / 2; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, (NI * MI) / ((NI + MI) / 2), 0);   // a few MFMAs of K-step 0 ...
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                              // ... then two reads of K-step 1
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NI * MI, 0);                 // MFMAs of K-step 1
            }
        } else {       // smaller tiles: the extra fragment registers cost a wave of occupancy (measured: slower)
    #pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 wf[NI], af[MI];
    #pragma unroll
                for (int a = 0; a < NI; ++a) {
                    const int row = wn * (BN / WN) + a * 16 + frow;
                    wf[a] = *reinterpret_cast<const bf16x8*>(&sB[buf * BN * BK + row * BK + (((fkc + 4 * ks) ^ swz<BK>(row)) * 8)]);
                }
    #pragma unroll
                for (int b = 0; b < MI; ++b) {
                    const int row = wm * (BM / WM) + b * 16 + frow;
                    af[b] = *reinterpret_cast<const bf16x8*>(&sA[buf * BM * BK + row * BK + (((fkc + 4 * ks) ^ swz<BK>(row)) * 8)]);
                }
    #pragma unroll
                for (int a = 0; a < NI; ++a)
    #pragma unroll
                    for (int b = 0; b < MI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], af[b], acc[a][b], 0, 0, 0);
            }
        }
    };
    set_tap(t_first);
    if constexpr (DMA) {
        static_assert(BM % RSTEP == 0 && BN % RSTEP == 0 && (RSTEP % 16) == 0, "lane-linear LDS-DMA layout: whole 1-KiB pieces per wave instruction");
        // (Tried on top: three LDS stages, loads two tiles ahead, one raw s_barrier per K-step with a counted `s_waitcnt vmcnt(6)`
        // so the DMA stays in flight across the barrier -- correct, but 313 / 270 us instead of 304 / 264 us: at two workgroups
        // per CU the other workgroup already covers the load latency; the remaining bound is LDS bandwidth.  The same pipeline on
        // a 256 x 256 tile with 128 x 128 per wave (a third less LDS traffic again, one wave per SIMD): correct, but 256 + 256
        // registers are not enough -- 796 bytes of scratch per lane, 417 / 316 us.)
        dma_tiles(0, t_first, 0);
        __syncthreads();                              // drains vmcnt(0): tile 0 is in LDS
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) {
                advance();
                dma_tiles(buf ^ 1, t_next, c_next);   // the other buffer was last read before the barrier that ended step kt-1
            }
            compute(buf);
            __syncthreads();
        }
    } else {
    load_tiles(ra, rb, t_first, 0);
    store_tiles(ra, rb, 0);
    if constexpr (!DEEP) {
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            const bool more = kt + 1 < nk;
            if (more) {
                advance();
                load_tiles(ra, rb, t_next, c_next);
            }
            compute(buf);
            if (more) store_tiles(ra, rb, buf ^ 1);
            __syncthreads();
        }
    } else {
        // Prefetch distance 2: while tile kt is multiplied out of LDS, tile kt+1 is already in flight into one register set
        // and tile kt+2 is requested into the other; a set is written to LDS a full iteration after its loads were issued
        // (one iteration = 32 MFMAs ~ 0.25 us covers an L2 hit but not an HBM / Infinity-Cache access, and the SQ counters
        // showed 30 % of the wave cycles parked on s_waitcnt).  Unrolled by two so both sets are indexed statically.
        if (nk > 1) {
            advance();
            load_tiles(ra2, rb2, t_next, c_next);           // tile 1 -> set 2
        }
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            if (kt + 2 < nk) {
                advance();
                load_tiles(ra, rb, t_next, c_next);          // tile kt+2 -> set 1
            }
            compute(0);
            if (kt + 1 < nk) store_tiles(ra2, rb2, 1);       // tile kt+1 (requested one iteration ago) -> LDS buffer 1
            __syncthreads();
            if (kt + 1 >= nk) break;
            if (kt + 3 < nk) {
                advance();
                load_tiles(ra2, rb2, t_next, c_next);        // tile kt+3 -> set 2
            }
            compute(1);
            if (kt + 2 < nk) store_tiles(ra, rb, 0);         // tile kt+2 -> LDS buffer 0
            __syncthreads();
        }
    }

    }   // !DMA

    // ---- BatchNorm statistics of the raw fp32 accumulators (rows >= M contribute exact zeros)
    if (p.stats) {
#pragma unroll
        for (int a = 0; a < NI; ++a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int b = 0; b < MI; ++b) {
                    const float v = acc[a][b][r];
                    s += v;
                    q += v * v;
                }
                s = row16_sum(s);
                q = row16_sum(q);
                if ((lane & 15) == 0) {
                    const int nl = wn * (BN / WN) + a * 16 + (lane >> 4) * 4 + r;
                    sStat[(wm * BN + nl) * 2] = s;
                    sStat[(wm * BN + nl) * 2 + 1] = q;
                }
            }
        }
        __syncthreads();
        constexpr int GROUPS = BM / STAT_ROWS, WPG = WM / GROUPS;        // statistics tiles per workgroup, wave rows per tile
        static_assert(BM % STAT_ROWS == 0 && WM % GROUPS == 0, "statistics tiles must be whole wave rows");
        for (int e = tid; e < BN * GROUPS; e += 256) {       // (BN * GROUPS may exceed the 256 threads of the workgroup)
            const int g = e / BN, nl = e % BN;
            if (n0 + nl >= p.N) continue;
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WPG; ++w) {
                s += sStat[((g * WPG + w) * BN + nl) * 2];
                q += sStat[((g * WPG + w) * BN + nl) * 2 + 1];
            }
            float* dst = p.stats + (size_t)(mt * GROUPS + g) * 2 * p.N;
            dst[n0 + nl] = s;
            dst[p.N + n0 + nl] = q;
        }
    }

    // ---- epilogue: accumulators -> wave-private fp32 LDS tile -> rolled loop over 8-column row segments
    // (BPP 16-row blocks per pass: all of them for BM = 128; two at a time for the 256-row tile, whose 128 x 64 wave tiles
    // would need 139 KB of staging)
    if (p.stats) __syncthreads();                    // sStat (aliases the staging tile) has been consumed
    float* stage = reinterpret_cast<float*>(smem) + wave * (BPP * 16) * EP;
    constexpr int LPR = TN / 8, RPP = 64 / LPR;      // lanes per row, rows per pass
    const int hw_out = p.Ho * p.Wo;
    const int lr = lane / LPR, lc = (lane % LPR) * 8;
    const int n = n0 + wn * TN + lc;
    const bool n_ok = n < p.N;
    f32x4 blo = {0.f, 0.f, 0.f, 0.f}, bhi = {0.f, 0.f, 0.f, 0.f};     // the lane's columns are the same in every pass
    if (p.bias && n_ok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (n + j < p.N) blo[j] = p.bias[n + j];
            if (n + 4 + j < p.N) bhi[j] = p.bias[n + 4 + j];
        }
    }
    f32x4 slo = {1.f, 1.f, 1.f, 1.f}, shi = {1.f, 1.f, 1.f, 1.f};
    if (p.col_scale && n_ok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (n + j < p.N) slo[j] = p.col_scale[n + j];
            if (n + 4 + j < p.N) shi[j] = p.col_scale[n + 4 + j];
        }
    }
    // The staging tile is wave-private: only the lanes of ONE wave exchange data through it, so a wave-level fence orders the writes
    // against the reads (a workgroup barrier made every wave wait for the slowest of four, twice per pass -- these epilogues are
    // most of the run time of the shallow GEMMs).
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
#pragma unroll
    for (int pass = 0; pass < MI / BPP; ++pass) {
        if (pass) wave_sync();                       // the previous pass has been read out
#pragma unroll
        for (int bb = 0; bb < BPP; ++bb)
#pragma unroll
            for (int a = 0; a < NI; ++a)
                *reinterpret_cast<f32x4*>(&stage[(bb * 16 + (lane & 15)) * EP + a * 16 + (lane >> 4) * 4]) = acc[a][pass * BPP + bb];
        wave_sync();
        if (n_ok) {
#pragma unroll 2
            for (int ps = 0; ps < BPP * 16 / RPP; ++ps) {
                const int ml = ps * RPP + lr;
                const int m = m0 + wm * TM + pass * BPP * 16 + ml;
                int orow = (m < p.M) ? m : -1;
                if (orow >= 0 && p.o_rowmap) orow = p.o_rowmap[m];
                if (orow >= 0 && grouped) {                  // GEMM row of the parity-grouped enumeration -> output pixel row
                    int gb, goy, gox;
                    group_pixel(m, gb, goy, gox);
                    orow = (gb * p.Ho + goy) * p.Wo + gox;
                }
                if (orow < 0) continue;
                const float rs = p.res_scale ? p.res_scale[idiv(orow, p.rows_per_sample)] : 1.f;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(&stage[ml * EP + lc]);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(&stage[ml * EP + lc + 4]);
                igemm_epilogue_row8(p, lo, hi, slo, shi, blo, bhi, orow, n, rs, hw_out);
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int BK, int LEAN = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) k_igemm2(IgemmArgs p_in) {
    igemm2_body<BM, BN, WM, WN, BK, LEAN>(p_in, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y);
}

// GROUPED launch (round 4): up to PK_GROUP_MAX independent convolutions / linear layers of ONE tile shape in one launch.  The exchange
// units of HRNet / HRFormer (hrformer.py:420-491, hrnet.py:157-227) are 2-12 tiny conv + BatchNorm layers per level on 1 500 .. 50 000
// pixels; as separate launches each paid ~10 us of fixed cost (dispatch ramp, cold code, first-tile latency, epilogue tail) for 1-3 us of
// work, and the step time followed the NUMBER of launches (1 288 x 13 us = 17 ms).  The argument blocks travel BY VALUE in the kernel
// arguments (no descriptor upload: the pointers change every step in eager mode); `first` is the prefix sum of the members' tile counts.
struct IgemmGroup {
    IgemmArgs a[PK_GROUP_MAX];
    int first[PK_GROUP_MAX + 1];
    int gy[PK_GROUP_MAX];
    int n;
};
template <int BM, int BN, int WM, int WN, int BK, int LEAN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) k_igemm2g(IgemmGroup g) {
    const int L = (int)blockIdx.x;
    int i = 0;
    while (i + 1 < g.n && L >= g.first[i + 1]) ++i;          // workgroup-uniform (scalar) search over <= 12 entries
    const int local = L - g.first[i], gy = g.gy[i], gx = (g.first[i + 1] - g.first[i]) / gy;
    igemm2_body<BM, BN, WM, WN, BK, LEAN>(g.a[i], local % gx, local / gx, gx, gy);
}

// ================================================================================================ ring kernel (head convs)
// The 3x3 256 -> 256 convolutions of the fusion head (forward and data gradient: 8 launches of 232 GFLOP per step, 65 % of the
// model's flops) on the two-barrier-per-K-step tile above sit at that structure's ceiling (~850 TFLOP/s isolated: every K-step drains
// the LDS-DMA with vmcnt(0) in front of the barrier).  k_conv8p is the deep-pipeline structure instead:
//   * 256 pixels x 256 output channels per 512-thread workgroup (8 waves as 2 (pixels) x 4 (channels), 128 x 64 per wave = 128
//     accumulator registers), one workgroup per CU;
//   * K-tile = 32 channels of one filter tap = one MFMA k-step: [256 pixel rows | 256 weight rows] x 64 bytes = 32 KB, in a FIVE-stage
//     LDS ring (160 KB).  Operands travel global -> LDS by LDS-DMA (4 x 1 KiB pieces per wave and tile, XOR swizzle on the source side)
//     and stay in flight across barriers: tile t+3 is requested while tile t is multiplied, the only vmcnt wait is a counted one
//     (tile t+1 landed, tiles t+2 and t+3 still in flight: two whole steps of latency cover);
//   * per K-tile a wave reads 4 weight + 8 pixel fragments (12 ds_read_b128) and issues 32 MFMAs (16x16x32);
//   * ONE barrier per K-tile.  The two wave groups (waves 0-3 / 4-7: the two waves of every SIMD) run the step in opposite order: between
//     two barriers group 0 does [reads + DMA issue of tile t; 32 MFMAs of tile t], group 1 [32 MFMAs of tile t-1 from the fragments it
//     read in the previous interval; reads + DMA issue of tile t] -- while one wave of a SIMD loads, its partner holds the matrix pipe.
//   (Measured on the way: 64-channel K-tiles in two 64 KB buffers, four quadrant phases of 16 MFMAs and two barriers each, the groups half
//   a phase apart -- 233 us for the head conv's data gradient against 264 us on k_igemm2; ablations on that version: MFMAs + barriers
//   alone 157 us (~100 cycles of barrier overhead per 256-cycle MFMA slot), DMA + reads + barriers alone 174 us.  The ring with two
//   barriers per 32-MFMA step: 211 us.)
//   (Measured and dropped: PERSISTENT workgroups (one per CU walking its share of the output tiles, the K-tile ring continuing across
//   tiles, each wave storing its finished tile straight from the accumulators as 16-byte pieces after a v_permlane16_swap while its
//   partner multiplies) -- 224 / 244 us (data gradient / forward + statistics) against 212 / 229 us for one tile per workgroup with
//   the LDS-staged 128-byte row stores below; launched with one workgroup per tile the same code ran 224 / 255 us, i.e. the three
//   lock-step rounds of 256 workgroups cost nothing, the half-line stores do.  Ablations of this version (data gradient, 210 us):
//   MFMAs + barriers alone 158 us (= the guide's 256 x 256 GEMM rate, i.e. the sustained-clock MFMA ceiling), reads + DMA + barriers
//   alone 145 us, nothing but prologue / barriers / epilogue 56 us.)
// Hazards (J_t = interval between barriers t-1 and t; both groups read tile t and request tile t+3 in J_t):
//   RAW  every wave waits for ITS pieces of tile t+1 (counted vmcnt) before barrier t; the first read of tile t+1 is in J_t+1;
//   WAR  group 1 fences its reads of tile t-2 (lgkmcnt(0)) at the start of J_t-1, group 0 inside J_t-2; the DMA of tile t+3 into the
//        same ring stage is issued in J_t.
// LDS reads of DMA-filled tiles go through inline assembly (see ring_tr above: the compiler would drain the DMA in front of every
// ds_read it can see).  Supported: stride-1 3x3 / 1x1 convolutions, Cin % 32 == 0 with >= 18 K-tiles, N % 256 == 0, bf16 row-major
// output, optional BatchNorm statistics; everything else stays on k_igemm2.
#define C8_STAGE 32768          // bytes per ring stage: [pixels 256 x 64 B | weights 256 x 64 B]
#define C8_STAGES 5
template <int OFF>
__device__ __forceinline__ bf16x8 c8_read(uint32_t lds_byte_addr) {          // OFF: immediate (16-bit) byte offset -- no address VALU per read
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte_addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void c8_fence4(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}
__device__ __forceinline__ void c8_barrier() { asm volatile("s_barrier" ::: "memory"); }

__global__ void __launch_bounds__(512, 2) k_conv8p(IgemmArgs p) {
    __shared__ __attribute__((aligned(1024))) uint16_t c8_smem[C8_STAGES * C8_STAGE / 2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the LDS-DMA destinations (M0) and the group tests are SGPR arithmetic
    const int grp = wave >> 2;                 // wave group = pixel half of the tile (and the half-step stagger)
    const int wn = wave & 3;                   // 64-channel quarter
    // XCD-contiguous tile order (bijective for any grid size): blocks b and b + 8 share an XCD's L2, neighbouring pixel tiles share halo rows
    int mt, nt;
    {
        const int G = gridDim.x, L = blockIdx.x, q = G >> 3, r = G & 7, xcd = L & 7;
        const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
        const int ntiles = p.N >> 8;
        mt = tile / ntiles;
        nt = tile - mt * ntiles;
    }
    const int m0 = mt * 256, n0 = nt * 256;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, 0x7ffffff0, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w), 0, 0x7ffffff0, 0x00020000);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)c8_smem;

    // ---- DMA geometry: a 1 KiB piece = 16 rows x 64 bytes; in the pixel tile and in the weight tile this wave moves pieces wave and
    // 8 + wave, i.e. this thread rows (i * 8 + wave) * 16 + lane / 4, i = 0, 1, physical chunk lane % 4
    const int drow = wave * 16 + (lane >> 2);
    const int lchunk = (lane & 3) ^ (((lane >> 5) & 1) << 1);    // logical chunk fetched into physical position lane % 4: ^ swz<32>(row)
    unsigned pc[2];                                              // byte offset of (pixel, channel lchunk * 8) at the centre tap, or OOB
    int pedge[2];                                                // bit 0: row above exists, 1: below, 2: left, 3: right
    unsigned wo[2];                                              // byte offset of weight row n, tap 0, channel lchunk * 8
    {
        const int hw = p.Ho * p.Wo;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + i * 128 + drow;
            pc[i] = OOB_OFF;
            pedge[i] = 0;
            if (m < p.M) {
                const int b = m / hw, rr = m - b * hw, oy = rr / p.Wo, ox = rr - oy * p.Wo;
                pc[i] = (unsigned)(((b * p.Hs + oy) * p.Ws + ox) * p.Cin + lchunk * 8) * 2u;
                pedge[i] = (oy > 0 ? 1 : 0) | (oy < p.Hs - 1 ? 2 : 0) | (ox > 0 ? 4 : 0) | (ox < p.Ws - 1 ? 8 : 0);
            }
            wo[i] = (unsigned)(((n0 + i * 128 + drow) * p.T) * p.Cin + lchunk * 8) * 2u;
        }
    }
    const int nk = p.T * (p.Cin >> 5);
    // (tap, first channel) of the tile being requested, with the tap's edge requirements and pixel offset (all wave-uniform)
    int q_tap = 0, q_c0 = 0, q_need = 0, q_delta = 0;
    auto set_tap = [&]() {
        if (p.T == 9) {
            const int kh = q_tap / 3, kw = q_tap - kh * 3;
            q_need = (kh == 0 ? 1 : 0) | (kh == 2 ? 2 : 0) | (kw == 0 ? 4 : 0) | (kw == 2 ? 8 : 0);
            q_delta = ((kh - 1) * p.Ws + (kw - 1)) * p.Cin * 2;
        }
    };
    set_tap();
    auto issue_tile = [&](int stage) {          // K-tile (q_tap, q_c0) -> ring stage; then advance to the next K-tile
#if defined(__HIP_DEVICE_COMPILE__)
        uint16_t* sp = c8_smem + (stage * C8_STAGE + wave * 1024) / 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = pc[i] != OOB_OFF && (pedge[i] & q_need) == q_need;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(sp + i * 4096), 16,
                                                     ok ? pc[i] + (unsigned)q_delta : OOB_OFF, q_c0 * 2, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(sp + 8192 + i * 4096), 16, wo[i],
                                                     (q_tap * p.Cin + q_c0) * 2, 0, 0);
#endif
        // chunk-major K order: the nine taps of one 32-channel chunk back to back -- the shifted re-reads of the same pixel rows (64
        // bytes each) hit the XCD's L2 instead of cycling the whole 256-channel rows through it between two taps (PMC, tap-major:
        // 394 MB fetched for 100 MB of input).  A tap change costs this kernel two scalars, no per-row address work.
        if (++q_tap == p.T) {
            q_tap = 0;
            q_c0 += 32;
        }
        set_tap();
    };

    // ---- fragment addresses: row (lane & 15) of a 16-row block, logical chunk lane >> 4, swizzled; blocks are 1024 bytes apart
    const int frow = lane & 15;
    const uint32_t fchunk = (uint32_t)(((lane >> 4) ^ ((frow >> 3) << 1)) * 16);
    const uint32_t faddr_p = lds0 + (grp * 128 + frow) * 64 + fchunk;                        // + stage * 32768 + pixel block * 1024
    const uint32_t faddr_w = lds0 + 16384 + (wn * 64 + frow) * 64 + fchunk;                  // + stage * 32768 + channel block * 1024

    f32x4 acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: tiles 0, 1, 2 requested; tile 0 landed
    issue_tile(0);
    issue_tile(1);
    issue_tile(2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    c8_barrier();
    int st_r = 0, st_w = 3;                    // ring stage read by this step / filled with tile t + 3
    bf16x8 wf[4], pf[8];
    auto multiply = [&]() {
        c8_fence4(wf[0], wf[1], wf[2], wf[3]);
        c8_fence4(pf[0], pf[1], pf[2], pf[3]);
        c8_fence4(pf[4], pf[5], pf[6], pf[7]);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], pf[b], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int t = 0; t <= nk; ++t) {
        if (grp == 1 && t > 0) multiply();                         // group 1: tile t-1, from the fragments it read in the previous interval
        if (t < nk) {
            const uint32_t bp = faddr_p + st_r * C8_STAGE, bw = faddr_w + st_r * C8_STAGE;
            wf[0] = c8_read<0>(bw);
            wf[1] = c8_read<1024>(bw);
            wf[2] = c8_read<2048>(bw);
            wf[3] = c8_read<3072>(bw);
            pf[0] = c8_read<0>(bp);
            pf[1] = c8_read<1024>(bp);
            pf[2] = c8_read<2048>(bp);
            pf[3] = c8_read<3072>(bp);
            pf[4] = c8_read<4096>(bp);
            pf[5] = c8_read<5120>(bp);
            pf[6] = c8_read<6144>(bp);
            pf[7] = c8_read<7168>(bp);
            if (t + 3 < nk) issue_tile(st_w);
            __builtin_amdgcn_sched_barrier(0);
            if (grp == 0) multiply();                              // group 0: tile t
        }
        // counted wait: tile t+1 has landed (tiles t+2, t+3 may still be in flight)
        if (t + 3 < nk) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (t + 2 < nk) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        st_r = st_r == C8_STAGES - 1 ? 0 : st_r + 1;
        st_w = st_w == C8_STAGES - 1 ? 0 : st_w + 1;
        c8_barrier();
    }

    // ---- BatchNorm statistics: a wave's 128 pixels are exactly one statistics tile (STAT_ROWS = 128) of its 64 channels
    if (p.stats) {
        float* dst = p.stats + (size_t)(mt * 2 + grp) * 2 * p.N + n0 + wn * 64;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const float v = acc[a][b][r];
                    sm += v;
                    sq += v * v;
                }
                sm = row16_sum(sm);
                sq = row16_sum(sq);
                if ((lane & 15) == 0) {
                    const int nl = a * 16 + (lane >> 4) * 4 + r;
                    dst[nl] = sm;
                    dst[p.N + nl] = sq;
                }
            }
    }
    // ---- epilogue: bf16 tile of the wave (128 pixels x 64 channels = 16 KB) through its own LDS slice, 16-byte chunks XOR-swizzled by
    // (pixel & 7); then 128-byte row segments to global, 16 bytes per lane
    uint16_t* stage = c8_smem + wave * 8192;
    if (p.col_scale) {          // eval-mode BatchNorm folded in: y = relu?(scale * conv + shift); channel = n0 + 64 wn + 16 a + 4 (lane >> 4) + r
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int c = n0 + wn * 64 + a * 16 + (lane >> 4) * 4;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.col_scale + c), sh = *reinterpret_cast<const f32x4*>(p.bias + c);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                acc[a][b] = acc[a][b] * sc + sh;
                if (p.act == 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[a][b][r] = fmaxf(acc[a][b][r], 0.f);
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int px = b * 16 + (lane & 15);
            const int chunk = (a * 2 + (lane >> 5)) ^ (px & 7);
            uint2 v;
            v.x = pack_bf16x2(acc[a][b][0], acc[a][b][1]);
            v.y = pack_bf16x2(acc[a][b][2], acc[a][b][3]);
            *reinterpret_cast<uint2*>(stage + px * 64 + chunk * 8 + ((lane >> 4) & 1) * 4) = v;
        }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint16_t* out = reinterpret_cast<uint16_t*>(p.out);
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int px = it * 8 + (lane >> 3), ch = lane & 7;
        const int m = m0 + grp * 128 + px;
        const u32x4 v = *reinterpret_cast<const u32x4*>(stage + px * 64 + ((ch ^ (px & 7)) * 8));
        if (m < p.M) *reinterpret_cast<u32x4*>(out + (size_t)m * p.ldo + n0 + wn * 64 + ch * 8) = v;
    }
}
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
// ================================================================================================ halo kernel (small-channel 3x3 convs)
// 3x3 stride-1 convolutions with 32 / 64 input and output channels on large maps (layer1, transitions and exchange units of HRFormer,
// every BasicBlock of HRNet's high-resolution branches) are HBM-bound by arithmetic (64 -> 64 @64x48: 50 MB, 14.5 GFLOP) but ran at
// 0.14 of the MFMA peak on k_igemm2: every tap re-stages the 128-row input tile through registers and LDS (9x the input through the
// L2 -> LDS path, ~8 TB/s) and every workgroup re-reads the weights.  k_conv3h:
//   * the K loop runs over PADDED pixel positions q = (b, py, px) of the zero-bordered (H+2) x (W+2) image (the trick of k_wgrad4_3x3):
//     out[q] = sum_taps X[q + (kh-1)(W+2) + (kw-1)] . W[tap], so every tap of a tile of 128 consecutive positions is the SAME LDS tile read
//     at a row offset -- the input tile plus a (W+3)-row halo either side goes global -> LDS ONCE per tile, by LDS-DMA, border positions
//     fetched with the out-of-range offset (zeros); border outputs are computed and dropped;
//   * the weights of a wave's output columns (all nine taps) live in REGISTERS for the whole kernel (144 VGPRs at 64 -> 64): waves as
//     2 (positions) x 2 (channels), 64 positions x COUT/2 channels each;
//   * persistent workgroups (two per CU) walk the tiles of one XCD's contiguous range; the DMA of tile i+1 flies under the MFMAs of tile i
//     (two LDS buffers, one counted wait + two barriers per tile);
//   * fragment addresses do not depend on the tile: nine per-tap base addresses per lane, computed once (the XOR swizzle term of a row is
//     the same for all four 16-row blocks of a wave and both k-steps differ by one address bit).
// Epilogue straight from the accumulators: interior positions only, optional addend; the BatchNorm statistics of a wave are kept in
// registers over all its tiles and written once (one partial row per workgroup and position half: pk_conv_stats_rows).
// Measured (B = 64, rocprof): 64 -> 64 @64x48 43.5 -> 30.8 us, @48x36 33 -> 22, 32 -> 32 @96x72 (B = 64) 38.5 -> 24.5, @16x12 15.3 -> 11.4.
// Batch sweep of 64 -> 64 @64x48 (17.1 / 22.4 / 30.8 / 45.2 / 77.3 us at B = 16 / 32 / 64 / 128 / 256): ~11 us fixed (launch, 75 MB of
// weight fragments into 2 048 waves' registers, first tiles' DMA, last stores) + ~5 us per tile slot; at B = 64 a workgroup walks 3 or
// 4 tiles (1 650 tiles on 512 workgroups), so a quarter of the loop time is the fourth-tile tail.  Tried without effect: double-buffered
// fragment reads (the LDS latency is hidden by the other workgroup of the CU), carried instead of divided DMA row coordinates, letting
// a tile's stores drain under the next two tiles (kept: it is free), retiring the weight loads in front of the loop (kept: without it
// the compiler's lazy waits sit inside the MFMA stream).
template <int OFF>
__device__ __forceinline__ bf16x8 h3_read(uint32_t lds_byte_addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte_addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ int h3_div(int q, int d, float inv) {          // floor(q / d) for 0 <= q < 2^24: float estimate + one correction
    int r = (int)((float)q * inv);
    const int rem = q - r * d;
    r += rem >= d ? 1 : (rem < 0 ? -1 : 0);
    return r;
}
// s_waitcnt vmcnt(n) with a wave-uniform run-time n (the immediate is 6 bits: n > 63 waits for less than asked, which is always safe)
__device__ __forceinline__ void h3_wait_vm(int n) {
#define H3_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n < 0 ? 0 : (n > 63 ? 63 : n)) {
        H3_W(0) H3_W(1) H3_W(2) H3_W(3) H3_W(4) H3_W(5) H3_W(6) H3_W(7) H3_W(8) H3_W(9) H3_W(10) H3_W(11) H3_W(12) H3_W(13) H3_W(14) H3_W(15)
        H3_W(16) H3_W(17) H3_W(18) H3_W(19) H3_W(20) H3_W(21) H3_W(22) H3_W(23) H3_W(24) H3_W(25) H3_W(26) H3_W(27) H3_W(28) H3_W(29) H3_W(30) H3_W(31)
        H3_W(32) H3_W(33) H3_W(34) H3_W(35) H3_W(36) H3_W(37) H3_W(38) H3_W(39) H3_W(40) H3_W(41) H3_W(42) H3_W(43) H3_W(44) H3_W(45) H3_W(46) H3_W(47)
        H3_W(48) H3_W(49) H3_W(50) H3_W(51) H3_W(52) H3_W(53) H3_W(54) H3_W(55) H3_W(56) H3_W(57) H3_W(58) H3_W(59) H3_W(60) H3_W(61) H3_W(62) H3_W(63)
    }
#undef H3_W
}
template <int CIN, int COUT>
__global__ void __launch_bounds__(256, 2) k_conv3h(IgemmArgs p, int pieces_per_wave) {
    constexpr int RB = CIN * 2, CH = CIN / 8, RP = 1024 / RB, KS = CIN / 32, NI = COUT / 32;
    extern __shared__ __attribute__((aligned(1024))) uint16_t h3_smem[];          // [2][pieces_per_wave * 4 KiB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int H = p.Hs, W = p.Ws, W2 = W + 2, P = (H + 2) * W2, halo = W + 3;
    const int nsamples = p.M / (H * W), Mp = nsamples * P;                        // padded positions of the whole batch
    const int ntiles = (Mp + 127) >> 7;
    const float invP = 1.f / (float)P, invW2 = 1.f / (float)W2;
    const int bufbytes = pieces_per_wave * 4096;
    // persistent workgroups, XCD-contiguous tile ranges (neighbouring tiles share their halo rows in L2)
    int tile_first, tile_step, n_mine;
    {
        const int G = gridDim.x, L = blockIdx.x, q = ntiles >> 3, r = ntiles & 7, x = L & 7, j = L >> 3;
        const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q, cnt = q + (x < r ? 1 : 0);
        tile_step = (G - x + 7) >> 3;
        tile_first = start + j;
        n_mine = j < cnt ? (cnt - j + tile_step - 1) / tile_step : 0;
    }
    if (n_mine == 0) {          // (uneven XCD ranges can leave a workgroup without a tile: its statistics rows are zeros)
        if (p.stats)
            for (int i = tid; i < 4 * COUT; i += 256) p.stats[(size_t)blockIdx.x * 4 * COUT + i] = 0.f;
        return;
    }
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, 0x7ffffff0, 0x00020000);
    const auto ro = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint16_t*>(p.out), 0, 0x7ffffff0, 0x00020000);
    const auto rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.res ? p.res : p.x), 0, 0x7ffffff0, 0x00020000);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)h3_smem;

    // ---- per-tap fragment base addresses (bytes from the start of a buffer): LDS row halo + 64 wm + (lane & 15) + tap offset, chunk lane >> 4
    uint32_t fa[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int kh = t / 3, kw = t - kh * 3;
        const int row = halo + wm * 64 + (lane & 15) + (kh - 1) * W2 + (kw - 1);
        const int sw = CIN == 64 ? ((row >> 1) & 7) : (((row >> 3) & 1) << 1);
        fa[t] = lds0 + (uint32_t)(row * RB + (((lane >> 4) ^ sw) & (CH - 1)) * 16);
    }
    // ---- DMA of the input tile + halo: buffer row j <-> padded position q0 - halo + j; 1 KiB pieces of RP rows, piece = wave + 4 i
    const int prow = lane / CH, pchunk = lane % CH;
    auto issue_tile = [&](int tile, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
        // (b, py, px) of this lane's row in its first piece by two divisions, then carried from piece to piece (+ 4 RP positions)
        const int q = tile * 128 - halo + wave * RP + prow;
        int b = h3_div(q, P, invP);
        const int rem = q - b * P;
        int py = h3_div(rem, W2, invW2), px = rem - py * W2;
        for (int i = 0; i < pieces_per_wave; ++i) {
            const int piece = wave + 4 * i, j = piece * RP + prow;
            const int sw = CIN == 64 ? ((j >> 1) & 7) : (((j >> 3) & 1) << 1);
            const bool ok = b >= 0 && b < nsamples && py >= 1 && py <= H && px >= 1 && px <= W;
            const unsigned off = ok ? (unsigned)((((b * H + py - 1) * W + px - 1) * CIN + ((pchunk ^ sw) & (CH - 1)) * 8) * 2) : OOB_OFF;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(h3_smem + (buf * bufbytes + piece * 1024) / 2), 16, off, 0, 0, 0);
            px += 4 * RP;
            while (px >= W2) {
                px -= W2;
                if (++py == H + 2) {
                    py = 0;
                    ++b;
                }
            }
        }
#endif
    };
    f32x4 acc[NI][4];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    float ssum[NI][4], ssq[NI][4];              // BatchNorm statistics of this wave over ALL its tiles (lane: position lane % 16, channels 4 g ..)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) ssum[ni][r] = ssq[ni][r] = 0.f;
    // Vector-memory operations of one wave, in issue order: DMA(0) DMA(1) | DMA(2) ST(0) | DMA(3) ST(1) | ...  (iteration i computes tile
    // i, then requests tile i+2 into the buffer it has just read, then stores tile i).  vmcnt counts loads AND stores in order, so the
    // wait for DMA(i) at the top of iteration i names what may stay in flight behind it -- ST(i-2), DMA(i+1), ST(i-1): a tile's stores
    // drain under the NEXT TWO tiles' MFMAs instead of stalling the next iteration (first version: wait for everything but DMA(i+1) --
    // each tile paid an HBM write round trip).
    // (only the NI * 4 output stores per tile are counted -- buffer-store intrinsics the backend cannot merge; the statistics stores and
    // addend loads on top of them make the wave wait for MORE than it must, never less)
    constexpr int n_st = NI * 4;
    issue_tile(tile_first, 0);
    if (n_mine > 1) issue_tile(tile_first + tile_step, 1);
    // (the weight loads go out BEHIND the first two tiles' DMA: one memory round trip for both instead of two in a row)
    // ---- weights of this wave's COUT/2 output channels, all nine taps, as MFMA A fragments (row = channel, 8 k-values per lane)
    bf16x8 wreg[9][KS][NI];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int n = wn * (COUT / 2) + ni * 16 + (lane & 15);
                wreg[t][ks][ni] = *reinterpret_cast<const bf16x8*>(p.w + ((size_t)n * 9 + t) * CIN + ks * 32 + (lane >> 4) * 8);
            }
    // The weight loads are retired HERE, through an asm that redefines every fragment: otherwise the compiler waits for them lazily at
    // their first use INSIDE the tile loop -- counted vmcnt waits that, from the second tile on, wait for the DMA of the next tile and the
    // stores of the previous one in the middle of the MFMAs (first version: 32 us, the DMA never overlapped the compute).
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) asm volatile("s_waitcnt vmcnt(0)" : "+v"(wreg[t][ks][ni])::"memory");
    for (int it = 0; it < n_mine; ++it) {
        const int tile = tile_first + it * tile_step, buf = it & 1;
        {
            int allow = 0;                                  // operations issued after DMA(it)
            if (it + 1 < n_mine) allow += pieces_per_wave;  // DMA(it+1)
            if (it >= 1) allow += n_st;                     // ST(it-1)
            if (it >= 2) allow += n_st;                     // ST(it-2)
            h3_wait_vm(allow);
        }
        asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[ni][b] = zero;
        const uint32_t boff = (uint32_t)(buf * bufbytes);
        // 9 * KS steps of [4 fragment reads | 4 NI MFMAs]; the reads of step s+1 are issued before the MFMAs of step s (two fragment sets,
        // a counted lgkmcnt(4) keeps the younger four in flight): the LDS latency of every step was exposed otherwise
        bf16x8 pf[2][4];
        {
            const uint32_t a = fa[0] + boff;
            pf[0][0] = h3_read<0>(a);
            pf[0][1] = h3_read<16 * RB>(a);
            pf[0][2] = h3_read<32 * RB>(a);
            pf[0][3] = h3_read<48 * RB>(a);
        }
#pragma unroll
        for (int st = 0; st < 9 * KS; ++st) {
            const int t = st / KS, ks = st % KS, cur = st & 1;
            if (st + 1 < 9 * KS) {
                const int t2 = (st + 1) / KS, ks2 = (st + 1) % KS;
                const uint32_t a = (fa[t2] + boff) ^ (ks2 ? 64u : 0u);
                pf[cur ^ 1][0] = h3_read<0>(a);
                pf[cur ^ 1][1] = h3_read<16 * RB>(a);
                pf[cur ^ 1][2] = h3_read<32 * RB>(a);
                pf[cur ^ 1][3] = h3_read<48 * RB>(a);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(pf[cur][0]), "+v"(pf[cur][1]), "+v"(pf[cur][2]), "+v"(pf[cur][3])::"memory");
            } else {
                c8_fence4(pf[cur][0], pf[cur][1], pf[cur][2], pf[cur][3]);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[ni][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[t][ks][ni], pf[cur][b], acc[ni][b], 0, 0, 0);
        }
        asm volatile("s_barrier" ::: "memory");          // every wave has finished reading this buffer ...
        if (it + 2 < n_mine) issue_tile(tile + 2 * tile_step, buf);      // ... tile it+2 is requested into it
        // ---- epilogue: lane (row g = lane / 16, position lane % 16 of block b) holds channels 16 ni + 4 g .. + 3; interior positions only
        const int g = lane >> 4;
        // (b, py, px) of this lane's position in block 0 by two divisions, carried to blocks 1-3 (+ 16 positions each)
        int eb, epy, epx;
        {
            const int q = tile * 128 + wm * 64 + (lane & 15);
            eb = h3_div(q, P, invP);
            const int rem = q - eb * P;
            epy = h3_div(rem, W2, invW2);
            epx = rem - epy * W2;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const bool ok = eb < nsamples && epy >= 1 && epy <= H && epx >= 1 && epx <= W;
            const int m = (eb * H + epy - 1) * W + epx - 1;
            epx += 16;
            while (epx >= W2) {
                epx -= W2;
                if (++epy == H + 2) {
                    epy = 0;
                    ++eb;
                }
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                f32x4 v = acc[ni][b];
                if (p.stats && ok) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ssum[ni][r] += v[r];
                        ssq[ni][r] += v[r] * v[r];
                    }
                }
                const unsigned off = ok ? (unsigned)(m * COUT + wn * (COUT / 2) + ni * 16 + g * 4) * 2u : OOB_OFF;
                if (p.col_scale) {          // eval-mode BatchNorm folded in (scale, shift of this lane's four channels: L1-resident)
                    const int c = wn * (COUT / 2) + ni * 16 + g * 4;
                    v = v * *reinterpret_cast<const f32x4*>(p.col_scale + c) + *reinterpret_cast<const f32x4*>(p.bias + c);
                }
                if (p.res) {
                    const u32x2 rv = __builtin_amdgcn_raw_buffer_load_b64(rr, off, 0, 0);
                    v[0] += bf16_lo(rv[0]); v[1] += bf16_hi(rv[0]); v[2] += bf16_lo(rv[1]); v[3] += bf16_hi(rv[1]);
                }
                if (p.act == 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                const u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                __builtin_amdgcn_raw_buffer_store_b64(o, ro, off, 0, 0);
            }
        }
    }
    if (p.stats) {          // ONE partial row per (workgroup, position half): the sums of all its tiles, kept in registers until here
        const int g = lane >> 4;
        float* dst = p.stats + (size_t)(blockIdx.x * 2 + wm) * 2 * COUT + wn * (COUT / 2);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sm = row16_sum(ssum[ni][r]), sq = row16_sum(ssq[ni][r]);
                if ((lane & 15) == 0) {
                    dst[ni * 16 + g * 4 + r] = sm;
                    dst[COUT + ni * 16 + g * 4 + r] = sq;
                }
            }
    }
}
static inline int pk_cu_count() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
    }
    return cus;
}
static inline int conv3h_pieces_per_wave(int Ws, int Cin) {
    const int R = 128 + 2 * (Ws + 3), RP = 1024 / (Cin * 2);
    return ((R + RP - 1) / RP + 3) / 4;
}
static inline bool conv3h_takes(const IgemmArgs& a) {
    const char* on_env = getenv("PK_CONV3H");
    const char* mt_env = getenv("PK_CONV3H_MIN_TILES");
    const bool on = !on_env || atoi(on_env) != 0;
    const long min_tiles = mt_env ? atol(mt_env) : 96;         // (16x12 x 64 samples = 126 tiles: 11.4 us against 15.3 us on k_igemm2)
    if (!(on && a.Ho > 0 && a.T == 9 && a.stride == 1 && !a.dilated && a.Hs == a.Ho && a.Ws == a.Wo && (a.Cin == 32 || a.Cin == 64) &&
          (a.N == 32 || a.N == 64) && a.out_mode == 0 && (!a.bias || a.col_scale) && !a.res_scale && !a.a_rowmap && !a.o_rowmap && !a.preact &&
          !a.gelu_of && (a.act == 0 || (a.act == 3 && !a.stats)) && !(a.col_scale && (a.stats || !a.bias || (((uintptr_t)a.col_scale | (uintptr_t)a.bias) & 15))) &&
          a.ldo == a.N && !(a.res && a.stats)))
        return false;
    const int pp = conv3h_pieces_per_wave(a.Ws, a.Cin);
    if (pp < 3 || pp > 11) return false;                       // the counted vmcnt waits are compiled for 3 .. 11 pieces per wave (W <= ~96 at 64 channels)
    const long Mp = (long)(a.M / (a.Hs * a.Ws)) * (a.Hs + 2) * (a.Ws + 2);
    return Mp < (1 << 24) && (Mp + 127) / 128 >= min_tiles;
}
static inline int conv3h_grid(const IgemmArgs& a) {          // persistent workgroups: two per CU (the 64 -> 64 variant holds 232 VGPRs)
    const long Mp = (long)(a.M / (a.Hs * a.Ws)) * (a.Hs + 2) * (a.Ws + 2);
    const long ntiles = (Mp + 127) / 128;
    return (int)(ntiles < 2L * pk_cu_count() ? ntiles : 2L * pk_cu_count());
}
static int conv3h_launch(const IgemmArgs& a, hipStream_t st, const char* who) {
    const int pp = conv3h_pieces_per_wave(a.Ws, a.Cin);
    const int lds = 2 * pp * 4096;
    const int grid = conv3h_grid(a);
#define C3H_GO(CI, CO)                                                                                                                        \
    {                                                                                                                                         \
        static bool attr_done = false;                                                                                                        \
        if (!attr_done) {                                                                                                                     \
            hipError_t e = hipFuncSetAttribute((const void*)k_conv3h<CI, CO>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 11 * 4096);     \
            if (e != hipSuccess) {                                                                                                            \
                pk_set_error("%s: cannot raise the LDS limit of k_conv3h: %s", who, hipGetErrorString(e));                                    \
                return (int)e;                                                                                                                \
            }                                                                                                                                 \
            attr_done = true;                                                                                                                 \
        }                                                                                                                                     \
        hipLaunchKernelGGL((k_conv3h<CI, CO>), dim3((unsigned)grid), dim3(256), lds, st, a, pp);                                              \
    }
    if (a.Cin == 64 && a.N == 64) C3H_GO(64, 64)
    else if (a.Cin == 64 && a.N == 32) C3H_GO(64, 32)
    else if (a.Cin == 32 && a.N == 64) C3H_GO(32, 64)
    else C3H_GO(32, 32)
#undef C3H_GO
    return pk_launch_status(who);
}

static inline bool conv8p_takes(const IgemmArgs& a) {
    // one workgroup per CU: below one full round of 256 tiles the 128 x 128 tiles of k_igemm2 spread the work over more CUs.
    // (both switches are read per call: the parity tests lower the tile count to run small and ragged shapes through this kernel)
    const char* on_env = getenv("PK_CONV8P");
    const char* mt_env = getenv("PK_CONV8P_MIN_TILES");
    const bool on = !on_env || atoi(on_env) != 0;
    const long min_tiles = mt_env ? atol(mt_env) : 256;
    return on && a.Ho > 0 && (a.T == 9 || a.T == 1) && a.stride == 1 && !a.dilated && (a.N % 256) == 0 && (a.Cin % 32) == 0 && a.T * a.Cin >= 576 &&
           a.out_mode == 0 && (!a.bias || a.col_scale) && !a.res && !a.res_scale && !a.a_rowmap && !a.o_rowmap && !a.preact && !a.gelu_of &&
           (a.act == 0 || (a.act == 3 && !a.stats)) && !(a.col_scale && (a.stats || !a.bias || (((uintptr_t)a.col_scale | (uintptr_t)a.bias) & 15))) &&
           a.ldo == a.N && (((uintptr_t)a.out) & 15) == 0 && (long)((a.M + 255) / 256) * (a.N / 256) >= min_tiles && a.Hs == a.Ho && a.Ws == a.Wo;
}

// PK_IGEMM_LOG=1: the launch shapes of a run, counted on the host and printed at exit (profiling aid)
#include <map>
#include <string>
static std::map<std::string, int>& igemm_log() {
    static std::map<std::string, int>* m = new std::map<std::string, int>();      // leaked on purpose: read by an atexit handler
    return *m;
}
static void igemm_log_dump() {
    for (auto& kv : igemm_log()) fprintf(stderr, "# igemm %6d x %s\n", kv.second, kv.first.c_str());
}
static void igemm_log_add(const IgemmArgs& a) {
    static const bool on = PK_KNOB("PK_IGEMM_LOG", 0) != 0;
    if (!on) return;
    static bool reg = false;
    if (!reg) {
        reg = true;
        atexit(igemm_log_dump);
    }
    char buf[200];
    snprintf(buf, sizeof buf, "M=%-7d N=%-4d Cin=%-4d T=%d s=%d dil=%d conv=%d amap=%d omap=%d res=%d stats=%d act=%d gelu_of=%d out_mode=%d", a.M, a.N, a.Cin, a.T,
             a.stride, a.dilated, a.Ho > 0, a.a_rowmap != nullptr, a.o_rowmap != nullptr, a.res != nullptr, a.stats != nullptr, a.act,
             a.gelu_of != nullptr, a.out_mode);
    igemm_log()[buf]++;
}
// plain launches (no dilated gather, no chunk-major order) take the LEAN = 3 instantiation of the same tile: see the kernel's first lines
#define IGEMM_GO(BM_, BN_, WM_, WN_, BK_, GRID)                                                                      \
    do {                                                                                                             \
        if (plain) hipLaunchKernelGGL((k_igemm2<BM_, BN_, WM_, WN_, BK_, 3>), GRID, block, 0, st, a);                \
        else if (plain_on && a.dilated && !a.chunk_major) hipLaunchKernelGGL((k_igemm2<BM_, BN_, WM_, WN_, BK_, 4>), GRID, block, 0, st, a);       \
        else hipLaunchKernelGGL((k_igemm2<BM_, BN_, WM_, WN_, BK_>), GRID, block, 0, st, a);                         \
    } while (0)
static int igemm_launch(const IgemmArgs& a_in, hipStream_t st, const char* who) {
    IgemmArgs a = a_in;
    igemm_log_add(a);
    a.vec8 = (a.ldo % 8) == 0 &&
             ((((uintptr_t)a.out | (uintptr_t)a.res | (uintptr_t)a.preact | (uintptr_t)a.gelu_of) & 15) == 0);
    static const int xcd_on = PK_KNOB("PK_IGEMM_XCD", 1);
    a.xcd_remap = xcd_on;
    static const int cm_on = PK_KNOB("PK_IGEMM_CHUNK_MAJOR", 1);
    a.chunk_major = cm_on && a.T == 9 && a.N <= 32 && a.Cin >= 128 && (a.Cin % 64) == 0 && !a.dilated;      // (the dilated walk has its own tap list)
    static const int dg_on = PK_KNOB("PK_IGEMM_DILGROUP", 1);
    a.dil_group = dg_on && a.dilated && a.T == 9 && a.out_mode == 0 && !a.stats && !a.o_rowmap && !a.res_scale;
    if (conv3h_takes(a)) return conv3h_launch(a, st, who);
    if (conv8p_takes(a)) {
        hipLaunchKernelGGL(k_conv8p, dim3((unsigned)(((a.M + 255) / 256) * (a.N / 256))), dim3(512), 0, st, a);
        return pk_launch_status(who);
    }
    const dim3 block(256);
    static const int plain_on = PK_KNOB("PK_IGEMM_PLAIN", 1);
    const bool plain = plain_on && !a.dilated;
    const unsigned gm = (unsigned)((a.M + 127) / 128);
    // Deep contractions with wide outputs (the 3x3 convs of the head: K = 2304, N = 128/256): 256 x 128 workgroup tile, 128 x 64
    // per wave -- a third less LDS traffic per MFMA than the 64 x 64 wave tile, which is what bounds those kernels.
    static const int big_on = PK_KNOB("PK_IGEMM_BIG", 1);
    // (needs >= 1024 workgroups: with 768 -- N = 128 at M = 196 608 -- the second round of workgroups is half empty and the
    // kernel is slower than the 128 x 128 tile.  Measured at N = 256: fwd 361 -> 334 us, dgrad 306 -> 285 us.)
    if (big_on && (a.N % 128) == 0 && a.T * a.Cin >= 576 && (a.Cin % 32) == 0 && (long)((a.M + 255) / 256) * (a.N / 128) >= 1024) {
        IGEMM_GO(256, 128, 2, 2, 32, dim3((a.M + 255) / 256, a.N / 128));
        return pk_launch_status(who);
    }
    // deeper K-chunks when the channel count allows full 64-wide tiles -- except for contractions of <= 128 (one or two steps): the BK = 32
    // variants hold half the LDS and run 4-7 waves per SIMD instead of 3, which is what an output-bound launch needs (1x1 64 -> 256 @64x48
    // data gradient 48.5 -> 37 us)
    static const int shallow32 = PK_KNOB("PK_IGEMM_SHALLOW32", 128);
    const bool k64 = (a.Cin % 64) == 0 && !(a.T * a.Cin <= shallow32);
    static const int lean_on = PK_KNOB("PK_IGEMM_LEAN", 1);
    const bool lean_any = lean_on && a.T == 1 && a.Ho == 0 && !a.stats && a.act <= 1 && a.out_mode == 0 && a.vec8 && k64 && (a.N % 8) == 0;
    const bool lean = lean_any && !a.preact && !a.gelu_of && a.act == 0, lean_g = lean_any && !lean;
    // Shallow contractions (K = T*Cin <= 256: the token-MLP / qkv GEMMs) are bound by their output traffic, not MFMA:
    // 128x64 tiles need half the accumulators (4 waves/SIMD instead of 2) and hide the epilogue's memory latency better
    // (measured 24 vs 32 us for qkv K=32 N=96, 43 vs 51 us for fc1+GELU K=32 N=128, 17 vs 21 us for K=128 N=512).
    // Few pixel rows (low-resolution branches: 24 .. 96 row tiles): a 128-wide N tile leaves most of the 256 CUs idle, so take
    // the widest N tile that still gives >= 512 workgroups (they are latency-bound, not MFMA-bound, at that size).
    // (128 x 64 tiles for the large convs were measured too: head conv 421 us instead of 361 us.)
    static const int smallm_on = PK_KNOB("PK_IGEMM_SMALLM", 1);
    const bool small_m = smallm_on && a.N > 64 && (long)gm * ((a.N + 127) / 128) < 512;
    if (small_m && (long)gm * ((a.N + 63) / 64) < 512 && !a.stats) {
        if (lean) hipLaunchKernelGGL((k_igemm2<128, 32, 4, 1, 64, 1>), dim3(gm, (a.N + 31) / 32), block, 0, st, a);
        else if (lean_g) hipLaunchKernelGGL((k_igemm2<128, 32, 4, 1, 64, 2>), dim3(gm, (a.N + 31) / 32), block, 0, st, a);
        else if (k64) IGEMM_GO(128, 32, 4, 1, 64, dim3(gm, (a.N + 31) / 32));
        else IGEMM_GO(128, 32, 4, 1, 32, dim3(gm, (a.N + 31) / 32));
    } else if (a.N > 64 && (a.T * a.Cin <= 256 || small_m) && (!a.stats || a.T * a.Cin <= 256)) {       // (shallow convs with statistics too: 1x1 64 -> 256 forward 57 -> 47.6 us)
        if (lean) hipLaunchKernelGGL((k_igemm2<128, 64, 4, 1, 64, 1>), dim3(gm, (a.N + 63) / 64), block, 0, st, a);
        else if (lean_g) hipLaunchKernelGGL((k_igemm2<128, 64, 4, 1, 64, 2>), dim3(gm, (a.N + 63) / 64), block, 0, st, a);
        else if (k64) IGEMM_GO(128, 64, 4, 1, 64, dim3(gm, (a.N + 63) / 64));
        else IGEMM_GO(128, 64, 4, 1, 32, dim3(gm, (a.N + 63) / 64));
    } else if (a.N > 64) {
        if (k64) IGEMM_GO(128, 128, 2, 2, 64, dim3(gm, (a.N + 127) / 128));
        else IGEMM_GO(128, 128, 2, 2, 32, dim3(gm, (a.N + 127) / 128));
    } else if (a.N > 32) {
        if (lean) hipLaunchKernelGGL((k_igemm2<128, 64, 4, 1, 64, 1>), dim3(gm, 1), block, 0, st, a);
        else if (lean_g) hipLaunchKernelGGL((k_igemm2<128, 64, 4, 1, 64, 2>), dim3(gm, 1), block, 0, st, a);
        else if (k64) IGEMM_GO(128, 64, 4, 1, 64, dim3(gm, 1));
        else IGEMM_GO(128, 64, 4, 1, 32, dim3(gm, 1));
    } else {
        if (k64) IGEMM_GO(128, 32, 4, 1, 64, dim3(gm, 1));
        else IGEMM_GO(128, 32, 4, 1, 32, dim3(gm, 1));
    }
    return pk_launch_status(who);
}

static int check_common(const char* who, const void* x, const void* w, const void* out, int M, int N, int Cin, int ldo,
                        int out_mode) {
    PK_REQUIRE(x && w && out, "%s: null pointer", who);
    PK_REQUIRE(M > 0 && N > 0 && Cin > 0, "%s: bad sizes M=%d N=%d Cin=%d", who, M, N, Cin);
    PK_SUPPORTED((Cin & 7) == 0, "%s: Cin=%d must be a multiple of 8 (16-byte bf16 chunks)", who, Cin);
    PK_REQUIRE((((uintptr_t)x | (uintptr_t)w) & 15) == 0, "%s: x/w must be 16-byte aligned", who);
    if (out_mode != 2) {
        PK_REQUIRE(ldo >= N, "%s: ldo=%d < N=%d", who, ldo, N);
        PK_REQUIRE((ldo & 3) == 0 && ((uintptr_t)out & 15) == 0, "%s: output pitch must be a multiple of 4 and 16-byte aligned", who);
    }
    PK_REQUIRE((int64_t)M * (ldo > Cin ? ldo : Cin) < 0x3fffffffLL, "%s: tensor too large for 32-bit byte offsets", who);
    return PK_OK;
}

extern "C" int pk_conv2d_nhwc(const void* x, const void* w_packed, void* out, float* stats_partial, const float* bias,
                              int B, int Hs, int Ws, int Cin, int Cout, int ksize, int stride, int dilated_input, int Ho,
                              int Wo, int act, int out_mode, const void* addend, void* stream) {
    const int M = B * Ho * Wo;
    int rc = check_common("pk_conv2d_nhwc", x, w_packed, out, M, Cout, Cin, Cout, out_mode);
    if (rc) return rc;
    PK_REQUIRE(ksize == 1 || ksize == 3, "pk_conv2d_nhwc: ksize %d", ksize);
    PK_REQUIRE(stride == 1 || (stride == 2 && !dilated_input), "pk_conv2d_nhwc: stride %d", stride);
    PK_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0, "pk_conv2d_nhwc: bad geometry");
    if (dilated_input) {
        PK_REQUIRE(Ho <= 2 * Hs && Wo <= 2 * Ws && Ho >= 2 * Hs - 1 && Wo >= 2 * Ws - 1, "pk_conv2d_nhwc: dilated geometry %dx%d <- %dx%d", Ho, Wo, Hs, Ws);
    } else {
        const int pad = ksize / 2;
        PK_REQUIRE(Ho == (Hs + 2 * pad - ksize) / stride + 1 && Wo == (Ws + 2 * pad - ksize) / stride + 1,
                   "pk_conv2d_nhwc: output %dx%d does not match input %dx%d k=%d s=%d", Ho, Wo, Hs, Ws, ksize, stride);
    }
    PK_REQUIRE(out_mode >= 0 && out_mode <= 2 && act >= 0 && act <= 2, "pk_conv2d_nhwc: bad mode");
    PK_REQUIRE((int64_t)B * Hs * Ws * Cin < 0x3fffffffLL && (int64_t)Cout * ksize * ksize * Cin < 0x3fffffffLL,
               "pk_conv2d_nhwc: input too large for 32-bit byte offsets");
    PK_REQUIRE(out_mode == 2 || (Cout & 3) == 0, "pk_conv2d_nhwc: Cout=%d must be a multiple of 4 for row-major output", Cout);
    IgemmArgs a{};
    a.x = (const uint16_t*)x; a.w = (const uint16_t*)w_packed; a.out = out; a.bias = bias; a.stats = stats_partial;
    a.M = M; a.N = Cout; a.Cin = Cin; a.T = ksize * ksize; a.Hs = Hs; a.Ws = Ws; a.Ho = Ho; a.Wo = Wo;
    a.stride = stride; a.pad = ksize / 2; a.dilated = dilated_input; a.ldo = Cout; a.rows_per_sample = Ho * Wo;
    a.act = act; a.out_mode = out_mode;
    PK_REQUIRE(!addend || (out_mode == 0 && !stats_partial), "pk_conv2d_nhwc: an addend needs the bf16 row-major output and no statistics");
    a.res = (const uint16_t*)addend;          // out = conv(x) + addend (same shape, bf16): the skip connection's gradient in a data-gradient launch
    return igemm_launch(a, (hipStream_t)stream, "pk_conv2d_nhwc");
}

// conv -> eval-mode BatchNorm (-> + residual) (-> ReLU) in ONE launch: y = relu?(col_scale[n] * conv(x)[., n] + col_shift[n] + residual).
// Inference only (the scale / shift are the constants gamma * rsqrt(running_var + eps), beta - running_mean * that); stride 1 or 2.
extern "C" int pk_conv2d_affine_nhwc(const void* x, const void* w_packed, void* out, const float* col_scale, const float* col_shift,
                                     const void* residual, int relu, int B, int Hs, int Ws, int Cin, int Cout, int ksize, int stride, int Ho,
                                     int Wo, void* stream) {
    const int M = B * Ho * Wo;
    int rc = check_common("pk_conv2d_affine_nhwc", x, w_packed, out, M, Cout, Cin, Cout, 0);
    if (rc) return rc;
    PK_REQUIRE(col_scale && col_shift, "pk_conv2d_affine_nhwc: null scale / shift");
    PK_REQUIRE((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), "pk_conv2d_affine_nhwc: ksize %d stride %d", ksize, stride);
    PK_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0, "pk_conv2d_affine_nhwc: bad geometry");
    const int pad = ksize / 2;
    PK_REQUIRE(Ho == (Hs + 2 * pad - ksize) / stride + 1 && Wo == (Ws + 2 * pad - ksize) / stride + 1,
               "pk_conv2d_affine_nhwc: output %dx%d does not match input %dx%d k=%d s=%d", Ho, Wo, Hs, Ws, ksize, stride);
    PK_REQUIRE((int64_t)B * Hs * Ws * Cin < 0x3fffffffLL && (int64_t)Cout * ksize * ksize * Cin < 0x3fffffffLL,
               "pk_conv2d_affine_nhwc: input too large for 32-bit byte offsets");
    PK_REQUIRE((Cout & 3) == 0, "pk_conv2d_affine_nhwc: Cout=%d must be a multiple of 4", Cout);
    IgemmArgs a{};
    a.x = (const uint16_t*)x; a.w = (const uint16_t*)w_packed; a.out = out; a.bias = col_shift; a.col_scale = col_scale;
    a.M = M; a.N = Cout; a.Cin = Cin; a.T = ksize * ksize; a.Hs = Hs; a.Ws = Ws; a.Ho = Ho; a.Wo = Wo;
    a.stride = stride; a.pad = pad; a.dilated = 0; a.ldo = Cout; a.rows_per_sample = Ho * Wo;
    a.act = relu ? 3 : 0; a.out_mode = 0; a.res = (const uint16_t*)residual;
    return igemm_launch(a, (hipStream_t)stream, "pk_conv2d_affine_nhwc");
}

// Grouped convolutions: n <= PK_GROUP_MAX members, each what one pk_conv2d_nhwc (train: bf16 output + statistics partials; data gradient:
// dilated_input / addend) or pk_conv2d_affine_nhwc (col_scale / bias / residual / relu) call would do, all on the 128 x 32 tile of k_igemm2
// (any Cout % 4 == 0; deep-K members share BK = 64 only when every member allows it).  Members must agree on `dilated_input`.
extern "C" int pk_conv2d_group(const PkConvDesc* d, int n, void* stream) {
    PK_REQUIRE(d && n > 0 && n <= PK_GROUP_MAX, "pk_conv2d_group: 1..%d members, got %d", PK_GROUP_MAX, n);
    IgemmGroup g{};
    bool k64 = true;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const PkConvDesc& c = d[i];
        const int M = c.B * c.Ho * c.Wo;
        int rc = check_common("pk_conv2d_group", c.x, c.w, c.out, M, c.Cout, c.Cin, c.Cout, 0);
        if (rc) return rc;
        PK_REQUIRE(c.ksize == 1 || c.ksize == 3, "pk_conv2d_group: ksize %d", c.ksize);
        PK_REQUIRE(c.stride == 1 || (c.stride == 2 && !c.dilated_input), "pk_conv2d_group: stride %d", c.stride);
        PK_REQUIRE(c.B > 0 && c.Hs > 0 && c.Ws > 0 && c.Ho > 0 && c.Wo > 0, "pk_conv2d_group: bad geometry");
        PK_REQUIRE((c.dilated_input != 0) == (d[0].dilated_input != 0), "pk_conv2d_group: members must agree on dilated_input");
        if (c.dilated_input) {
            PK_REQUIRE(c.Ho <= 2 * c.Hs && c.Wo <= 2 * c.Ws && c.Ho >= 2 * c.Hs - 1 && c.Wo >= 2 * c.Ws - 1,
                       "pk_conv2d_group: dilated geometry %dx%d <- %dx%d", c.Ho, c.Wo, c.Hs, c.Ws);
        } else {
            const int pad = c.ksize / 2;
            PK_REQUIRE(c.Ho == (c.Hs + 2 * pad - c.ksize) / c.stride + 1 && c.Wo == (c.Ws + 2 * pad - c.ksize) / c.stride + 1,
                       "pk_conv2d_group: output %dx%d does not match input %dx%d k=%d s=%d", c.Ho, c.Wo, c.Hs, c.Ws, c.ksize, c.stride);
        }
        PK_REQUIRE((int64_t)c.B * c.Hs * c.Ws * c.Cin < 0x3fffffffLL && (int64_t)c.Cout * c.ksize * c.ksize * c.Cin < 0x3fffffffLL,
                   "pk_conv2d_group: input too large for 32-bit byte offsets");
        PK_REQUIRE((c.Cout & 3) == 0, "pk_conv2d_group: Cout=%d must be a multiple of 4", c.Cout);
        PK_REQUIRE(!(c.stats && (c.res || c.col_scale)), "pk_conv2d_group: statistics go with the plain bf16 output only");
        PK_REQUIRE(c.act == 0 || c.act == 3, "pk_conv2d_group: act %d (0 none, 3 ReLU after the residual)", c.act);
        IgemmArgs& a = g.a[i];
        a.x = (const uint16_t*)c.x; a.w = (const uint16_t*)c.w; a.out = c.out; a.bias = c.bias; a.col_scale = c.col_scale;
        a.stats = c.stats; a.res = (const uint16_t*)c.res;
        a.M = M; a.N = c.Cout; a.Cin = c.Cin; a.T = c.ksize * c.ksize; a.Hs = c.Hs; a.Ws = c.Ws; a.Ho = c.Ho; a.Wo = c.Wo;
        a.stride = c.stride; a.pad = c.ksize / 2; a.dilated = c.dilated_input ? 1 : 0; a.ldo = c.Cout; a.rows_per_sample = c.Ho * c.Wo;
        a.act = c.act; a.out_mode = 0;
        a.vec8 = (a.ldo % 8) == 0 && ((((uintptr_t)a.out | (uintptr_t)a.res) & 15) == 0);
        a.xcd_remap = 0;            // (the remap assumes a grid of its own; these tensors fit in any one L2)
        a.chunk_major = 0;
        a.dil_group = a.dilated && a.T == 9 && !a.stats;
        k64 = k64 && (a.Cin % 64) == 0 && a.T * a.Cin > 128;
        g.gy[i] = (a.N + 31) / 32;
        g.first[i] = total;
        total += ((M + 127) / 128) * g.gy[i];
    }
    g.first[n] = total;
    g.n = n;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)total), block(256);
    if (d[0].dilated_input) {
        if (k64) hipLaunchKernelGGL((k_igemm2g<128, 32, 4, 1, 64, 4>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((k_igemm2g<128, 32, 4, 1, 32, 4>), grid, block, 0, st, g);
    } else {
        if (k64) hipLaunchKernelGGL((k_igemm2g<128, 32, 4, 1, 64, 3>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((k_igemm2g<128, 32, 4, 1, 32, 3>), grid, block, 0, st, g);
    }
    return pk_launch_status("pk_conv2d_group");
}

extern "C" int pk_conv_stats_tiles(int M) { return (M + STAT_ROWS - 1) / STAT_ROWS; }
// rows of the [rows][2][Cout] partial-statistics buffer that pk_conv2d_nhwc(bf16 output, statistics) writes for this geometry: the halo
// kernel emits one row per 64 PADDED positions, every other kernel one per 128 output pixels (pk_bn_finalize sums whatever it is given)
extern "C" int pk_conv_stats_rows(int B, int Hs, int Ws, int Cin, int Cout, int ksize, int stride, int Ho, int Wo) {
    IgemmArgs a{};
    a.M = B * Ho * Wo; a.N = Cout; a.Cin = Cin; a.T = ksize * ksize; a.Hs = Hs; a.Ws = Ws; a.Ho = Ho; a.Wo = Wo; a.stride = stride;
    a.pad = ksize / 2; a.ldo = Cout; a.out_mode = 0; a.stats = reinterpret_cast<float*>(1);
    if (conv3h_takes(a)) return 2 * conv3h_grid(a);
    return pk_conv_stats_tiles(a.M);
}

extern "C" int pk_linear_bf16(const void* x, const void* w, void* out, const float* bias, const void* residual,
                              const float* res_scale, const int32_t* a_rowmap, const int32_t* o_rowmap, void* preact_out,
                              const void* gelu_grad_of, int M, int N, int K, int rows_per_sample, int act, int out_fp32,
                              void* stream) {
    int rc = check_common("pk_linear_bf16", x, w, out, M, N, K, N, out_fp32 ? 1 : 0);
    if (rc) return rc;
    PK_REQUIRE((N & 3) == 0, "pk_linear_bf16: N=%d must be a multiple of 4", N);
    PK_REQUIRE(!res_scale || rows_per_sample > 0, "pk_linear_bf16: res_scale needs rows_per_sample");
    PK_REQUIRE(act >= 0 && act <= 1, "pk_linear_bf16: act %d", act);
    IgemmArgs a{};
    a.x = (const uint16_t*)x; a.w = (const uint16_t*)w; a.out = out; a.bias = bias; a.res = (const uint16_t*)residual;
    a.res_scale = res_scale; a.a_rowmap = a_rowmap; a.o_rowmap = o_rowmap;
    a.preact = (uint16_t*)preact_out; a.gelu_of = (const uint16_t*)gelu_grad_of;
    a.M = M; a.N = N; a.Cin = K; a.T = 1; a.Ho = 0; a.Wo = 0; a.ldo = N; a.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
    a.act = act; a.out_mode = out_fp32 ? 1 : 0;
    return igemm_launch(a, (hipStream_t)stream, "pk_linear_bf16");
}

// ================================================================================================ weight gradient
// dW[n][t][c] = sum_m G(m, n) * A(m, t, c).  The contraction runs over pixels/tokens m, i.e. along the ROWS of both
// operands as they sit in HBM, so each MFMA fragment needs 8 consecutive m of one column.  Tiles are staged row-major
// ([m][n] and [m][c], 16-byte global loads -> 16-byte LDS stores) and the fragments are read with the gfx950 hardware
// transpose read `ds_read_b64_tr_b16` (4 rows x 16 columns per 16-lane group, delivered column-major): two reads per
// fragment, no scalar LDS traffic.  Row pitch = width + 16 elements keeps those reads bank-conflict free.
// One workgroup owns a TN(n) x TC(c) tile of one filter tap and one slice of M (split-M over gridDim.z); slices are
// written as fp32 slabs and summed in fixed order by k_wgrad_reduce (deterministic, no float atomics).
struct WgradArgs {
    const uint16_t* x;   // activations (layer input) bf16
    const uint16_t* g;   // output gradient bf16 [rows][N]
    float* part;         // [S][N][T][Cin] fp32
    const int32_t* a_rowmap;  // source row of x for GEMM row m (linear), -1 = zero row
    const int32_t* g_rowmap;  // source row of g for GEMM row m, -1 = zero row
    const float* g_scale;     // optional per-sample multiplier of g rows (DropPath), indexed by g_row / g_rows_per_sample
    int g_rows_per_sample;
    float* bias_part;         // optional [S][N] fp32: column sums of the (scaled, gathered) G rows = bias gradient slabs
    int M, N, Cin, T, Hs, Ws, Ho, Wo, stride, pad, m_per_slice, ctiles;
    int ntiles3, nslices3;    // k_wgrad3: output tiles per tap, number of M-slices
};

#define WG_MK 32
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ bf16x8 tr_frag(const uint16_t* tile, int pitch, int col0, int lane) {
    // fragment for MFMA lane (col = col0 + lane&15, k-group g = lane>>4): elements k = 8g .. 8g+7 of that column
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pq = i & 3;
    const uint16_t* a0 = tile + (8 * g + q) * pitch + col0 + 4 * pq;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * pitch));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int TN, int TC, int MK>   // output tile: TN rows (n) x TC columns (c), 4 waves as 2 x 2; MK pixel rows staged per step
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TN == 128 && MK == 32 ? 4 : 1))) k_wgrad2(WgradArgs p) {
    constexpr int PN = TN + 16, PC = TC + 16;            // LDS row pitches (elements)
    constexpr int GCH = TN / 8, XCH = TC / 8;            // 16-byte chunks per staged row
    constexpr int G_PT = MK * GCH / 256, X_PT = MK * XCH / 256;   // chunks per thread per step (1 or 2)
    constexpr int NI = TN / 2 / 16, CI = TC / 2 / 16;    // accumulator tiles per wave
    __shared__ __attribute__((aligned(16))) uint16_t sG[2][MK * PN];
    __shared__ __attribute__((aligned(16))) uint16_t sX[2][MK * PC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 3x3: 1-D grid, the nine taps (and the output tiles) of one M-slice on ONE XCD (slice = xcd + 8 * group): they read the same
    // rows of G and overlapping rows of X at about the same time, so eight of the nine reads are L2 hits.
    int bx = blockIdx.x, t = blockIdx.y, bz = blockIdx.z;
    if (p.nslices3 > 0) {
        const int xcd = blockIdx.x & 7, k_in = blockIdx.x >> 3, per_slice = 9 * p.ntiles3;
        bz = xcd + 8 * (k_in / per_slice);
        if (bz >= p.nslices3) return;            // padding workgroups of the last group (whole workgroup, before any barrier)
        const int rem = k_in % per_slice;
        t = rem % 9;
        bx = rem / 9;
    }
    const int ntile = bx / p.ctiles, ctile = bx - ntile * p.ctiles;
    const int n0 = ntile * TN, c0 = ctile * TC;
    const int kw_n = (p.T == 9) ? 3 : 1, kh = t / kw_n, kw = t - kh * kw_n;
    const int m_begin = bz * p.m_per_slice;
    const int m_end = min(p.M, m_begin + p.m_per_slice);
    const bool linear = (p.Ho == 0);
    const int hw = p.Ho * p.Wo;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, 0x7ffffff0, 0x00020000);
    const auto rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.g), 0, 0x7ffffff0, 0x00020000);

    // fixed (row-in-step, chunk) assignment
    int g_row[G_PT], g_col[G_PT], x_row[X_PT], x_col[X_PT];
#pragma unroll
    for (int i = 0; i < G_PT; ++i) { const int q = tid + 256 * i; g_row[i] = q / GCH; g_col[i] = (q % GCH) * 8; }
#pragma unroll
    for (int i = 0; i < X_PT; ++i) { const int q = tid + 256 * i; x_row[i] = q / XCH; x_col[i] = (q % XCH) * 8; }

    // The DropPath row scale is applied when the staged registers are written to LDS, not when they are loaded: scaling at
    // load time consumed the load result immediately and serialised every step of the proj / fc2 weight gradients on a
    // global-memory round trip.  (Tried and dropped: a second register set with loads two steps ahead -- the duplicated loop
    // body pushed the 128 x 128 tile into scratch, 523 -> 1 816 us, and bought nothing on the 64 x 64 tile.)
    u32x4 rgv[G_PT], rxv[X_PT];
    float rsc[G_PT];
    auto load = [&](int ms) {
#pragma unroll
        for (int i = 0; i < G_PT; ++i) {
            const int m = ms + g_row[i], n = n0 + g_col[i];
            unsigned off = OOB_OFF;
            float sc = 1.f;
            if (m < m_end && n < p.N) {
                const int gr = p.g_rowmap ? p.g_rowmap[m] : m;
                if (gr >= 0) {
                    off = (unsigned)((gr * p.N + n) * 2);
                    if (p.g_scale) sc = p.g_scale[gr / p.g_rows_per_sample];
                }
            }
            rgv[i] = __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0);
            rsc[i] = sc;
        }
#pragma unroll
        for (int i = 0; i < X_PT; ++i) {
            const int m = ms + x_row[i], c = c0 + x_col[i];
            unsigned off = OOB_OFF;
            if (m < m_end && c < p.Cin) {
                if (linear) {
                    const int xr = p.a_rowmap ? p.a_rowmap[m] : m;
                    if (xr >= 0) off = (unsigned)((xr * p.Cin + c) * 2);
                } else {
                    const int b = m / hw, r = m - b * hw, oy = r / p.Wo, ox = r - oy * p.Wo;
                    const int iy = oy * p.stride - p.pad + kh, ix = ox * p.stride - p.pad + kw;
                    if (iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws) off = (unsigned)((((b * p.Hs + iy) * p.Ws + ix) * p.Cin + c) * 2);
                }
            }
            rxv[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < G_PT; ++i) {
            u32x4 v = rgv[i];
            if (p.g_scale) {
                const float sc = rsc[i];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[j] = pack_bf16x2(__uint_as_float(v[j] << 16) * sc, __uint_as_float(v[j] & 0xffff0000u) * sc);
            }
            *reinterpret_cast<u32x4*>(&sG[buf][g_row[i] * PN + g_col[i]]) = v;
        }
#pragma unroll
        for (int i = 0; i < X_PT; ++i) *reinterpret_cast<u32x4*>(&sX[buf][x_row[i] * PC + x_col[i]]) = rxv[i];
    };
    f32x4 acc[NI][CI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < CI; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wn = wave >> 1, wc = wave & 1;
    const int nsteps = (m_end - m_begin + MK - 1) / MK;
    const bool do_bias = p.bias_part && ctile == 0 && t == 0;      // one workgroup column per n-tile owns the bias slab
    float bsum = 0.f;
    if (nsteps > 0) {
        load(m_begin);
        store(0);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) load(m_begin + (s + 1) * MK);
        if (do_bias && tid < TN) {          // column tid of the staged G tile: 32 rows, fixed order
#pragma unroll 8
            for (int r = 0; r < MK; ++r) bsum += bf16_to_f32(sG[buf][r * PN + tid]);
        }
#pragma unroll
        for (int ks = 0; ks < MK / 32; ++ks) {
            bf16x8 gf[NI], xf[CI];
#pragma unroll
            for (int a = 0; a < NI; ++a) gf[a] = tr_frag(sG[buf] + 32 * ks * PN, PN, wn * (TN / 2) + a * 16, lane);
#pragma unroll
            for (int b = 0; b < CI; ++b) xf[b] = tr_frag(sX[buf] + 32 * ks * PC, PC, wc * (TC / 2) + b * 16, lane);
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < CI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[a], xf[b], acc[a][b], 0, 0, 0);
        }
        if (s + 1 < nsteps) store(buf ^ 1);
        __syncthreads();
    }
    if (do_bias && tid < TN && n0 + tid < p.N) p.bias_part[(size_t)bz * p.N + n0 + tid] = bsum;
    float* dst = p.part + (size_t)bz * p.N * p.T * p.Cin;
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < CI; ++b) {
            const int c = c0 + wc * (TC / 2) + b * 16 + (lane & 15);
            const int n = n0 + wn * (TN / 2) + a * 16 + (lane >> 4) * 4;
            if (c < p.Cin) {
                if (n < p.N) dst[((size_t)n * p.T + t) * p.Cin + c] = acc[a][b][0];
                if (n + 1 < p.N) dst[((size_t)(n + 1) * p.T + t) * p.Cin + c] = acc[a][b][1];
                if (n + 2 < p.N) dst[((size_t)(n + 2) * p.T + t) * p.Cin + c] = acc[a][b][2];
                if (n + 3 < p.N) dst[((size_t)(n + 3) * p.T + t) * p.Cin + c] = acc[a][b][3];
            }
        }
}

// ------------------------------------------------------------------------------------------------ wide weight-gradient kernel
// The 3x3 convolutions of the fusion head (256 -> 256 channels at 64x48, M = 196 608 pixels: 232 GFLOP each, five per step) ran
// at 19 % of the MFMA peak on the 128 x 128 tile above: every (n-tile, c-tile, tap) workgroup streams its slice of G and X again
// (36 x 100 MB per launch), 16 KB per 64 MFMAs = the CU's 64 B/clk L1 port at full MFMA rate, and a one-step register
// prefetch does not cover an L2 round trip.  This kernel:
//   * 256 (n) x 256 (c) output tile per 512-thread workgroup (8 waves as 4 x 2, 64 x 128 per wave: 128 accumulator registers):
//     twice the MFMAs per staged byte, operand traffic 18 x 100 MB;
//   * operands go global -> LDS by LDS-DMA (`buffer_load ... lds`, no VGPR round trip) into a FOUR-stage ring, three 32-row
//     K-steps in flight ahead of the one being multiplied, one raw s_barrier per step with a counted `s_waitcnt vmcnt`, so the
//     DMA stays in flight across the barrier;
//   * tiles are row-major [32 rows][256 columns] (512-byte rows, what one DMA wave-instruction writes linearly: two rows per
//     1 KiB piece) with the 16-byte chunks XOR-swizzled on the SOURCE side by swz(row) = 2 (row & 3) + 8 ((row >> 3) & 1): the
//     transpose reads `ds_read_b64_tr_b16` of a 32-lane half (rows r..r+3 and r+8..r+11, 32 columns) then touch 32 distinct
//     8-byte bank pairs (without it all rows alias: 8-way conflicts).
// One workgroup per CU (128 KB of LDS), 1-D grid of 9 taps x S slices ~ one round of workgroups with equal work.
// (Measured and dropped: software-pipelining the fragment reads of step s + 1 under the MFMAs of step s inside every wave -- behind the
// per-step barrier the eight waves read together and multiply together -- needs a second register set for G (and X): 128 accumulators
// + 64..80 fragment registers + addresses do not fit 256 VGPRs at two waves per SIMD; 110 spills, and scratch traffic shares vmcnt
// with the DMA ring.)
// LDS reads of tiles that are filled by LDS-DMA go through inline assembly.  The compiler cannot tell which LDS bytes an outstanding
// `buffer_load ... lds` will write, so before any ds_read it can see it inserts `s_waitcnt vmcnt(0)`: the tile requested a moment ago
// is awaited BEFORE the current one is multiplied: a four-deep ring is drained on every step (first version of k_wgrad3: 25 % of
// the MFMA peak, 372 us; 289 us with the reads below).  The kernels do their own accounting
// (counted vmcnt / barrier before a stage is read), read through ring_tr(), and close each group of reads with a fence that waits
// for the LDS data and, by naming the fragments as in/out operands, keeps the MFMAs behind it.
__device__ __forceinline__ s16x4 ring_tr(const uint16_t* a) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)a) : "memory");
    return v;
}
__device__ __forceinline__ void ring_fence(s16x4& a, s16x4& b, s16x4& c, s16x4& d, s16x4& e, s16x4& f, s16x4& g, s16x4& h) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)::"memory");
}
__device__ __forceinline__ bf16x8 ring_join(const s16x4& lo, const s16x4& hi) {
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
#define W3_ROWS 32
#define W3_STAGES 5
__device__ __forceinline__ int w3_swz(int row) { return 2 * (row & 3) + 8 * ((row >> 3) & 1); }
__device__ __forceinline__ void w3_frag(const uint16_t* tile, int col0, int lane, s16x4& lo, s16x4& hi) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pq = i & 3;
    const int row = 8 * g + q;
    const int pchunk = ((col0 >> 3) + (pq >> 1)) ^ (2 * q + 8 * (g & 1));          // w3_swz(row) == w3_swz(row + 4)
    const uint16_t* a0 = tile + row * 256 + pchunk * 8 + (pq & 1) * 4;
    lo = ring_tr(a0);
    hi = ring_tr(a0 + 4 * 256);
}
__global__ void __launch_bounds__(512, 2) k_wgrad3(WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t w3_smem[];            // [stage][G tile 32 x 256 | X tile 32 x 256]
    constexpr int TILE = W3_ROWS * 256;                                            // elements per operand tile (16 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wc = wave & 1;
    // XCD-aware placement (1-D grid): workgroups are dealt to the 8 XCDs round-robin by linear id, each XCD has its own 4 MB L2.  The 9
    // taps (and the output tiles) of one M-slice read the same rows of G and X at about the same time, so they all go to ONE XCD
    // (slice z = xcd + 8 * group).
    const int xcd = blockIdx.x & 7, k_in = blockIdx.x >> 3, per_slice = 9 * p.ntiles3;
    const int zslice = xcd + 8 * (k_in / per_slice), rem = k_in % per_slice;
    if (zslice >= p.nslices3) return;            // padding workgroups of the last group (whole workgroup: no barrier was reached)
    const int t = rem % 9, tile3 = rem / 9;
    const int ntile = tile3 / p.ctiles, ctile = tile3 - ntile * p.ctiles;
    const int n0 = ntile * 256, c0 = ctile * 256;
    const int kh = t / 3, kw = t - kh * 3;
    const int m_begin = zslice * p.m_per_slice;
    const int m_end = min(p.M, m_begin + p.m_per_slice);
    const int nsteps = (max(m_end - m_begin, 0) + W3_ROWS - 1) / W3_ROWS;
    const int hw = p.Ho * p.Wo;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, 0x7ffffff0, 0x00020000);
    const auto rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.g), 0, 0x7ffffff0, 0x00020000);
    // this wave's DMA pieces: tile rows (4 wave + 2 j, + 1), j = 0, 1; lane -> (row within the pair, physical chunk).  The pixel
    // coordinates of the lane's row are carried from step to step (+32 rows with carries): the first version recomputed them with two
    // integer divisions per piece and step, ~200 VALU instructions per wave and step -- more issue time than the step's 32 MFMAs.
    const int prow = lane >> 5, pchunk = lane & 31;
    int pm[2], pb[2], poy[2], pox[2], colg[2], colx[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 4 * wave + 2 * j + prow;
        const int lch = pchunk ^ w3_swz(row);
        pm[j] = m_begin + row;
        pb[j] = pm[j] / hw;
        const int r = pm[j] - pb[j] * hw;
        poy[j] = r / p.Wo;
        pox[j] = r - poy[j] * p.Wo;
        colg[j] = n0 + lch * 8;
        colx[j] = c0 + lch * 8;
    }
    int issued = 0;
    auto issue_next = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
        const int st = issued;
        issued = issued == W3_STAGES - 1 ? 0 : issued + 1;
        uint16_t* sg = w3_smem + st * 2 * TILE;
        uint16_t* sx = sg + TILE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool live = pm[j] < m_end;
            const int iy = poy[j] * p.stride - p.pad + kh, ix = pox[j] * p.stride - p.pad + kw;
            const bool in = live && iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws;
            const unsigned og = live ? (unsigned)((pm[j] * p.N + colg[j]) * 2) : OOB_OFF;
            const unsigned ox_ = in ? (unsigned)((((pb[j] * p.Hs + iy) * p.Ws + ix) * p.Cin + colx[j]) * 2) : OOB_OFF;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (__attribute__((address_space(3))) void*)(sg + (4 * wave + 2 * j) * 256), 16, og, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(sx + (4 * wave + 2 * j) * 256), 16, ox_, 0, 0, 0);
            pm[j] += W3_ROWS;                       // next step: 32 rows further, with carries into (oy, b)
            pox[j] += W3_ROWS;
            while (pox[j] >= p.Wo) {
                pox[j] -= p.Wo;
                if (++poy[j] == p.Ho) {
                    poy[j] = 0;
                    ++pb[j];
                }
            }
        }
#endif
    };
    f32x4 acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // prologue: three steps in flight (steps beyond the slice issue out-of-range loads = zero tiles, so the counts stay uniform)
    issue_next();
    issue_next();
    issue_next();
    // ONE barrier per step, the two waves of every SIMD (wave groups 0-3 / 4-7) in opposite order between two barriers (k_conv8p has the
    // hazard analysis): group 0 [reads + DMA issue of step s; 32 MFMAs of step s], group 1 [32 MFMAs of step s-1 from the fragments it
    // read in the previous interval; reads + DMA issue of step s].  Step s+3 goes into the stage of step s-2, whose reads group 1 fenced
    // at the start of the previous interval: FIVE stages.  The counted wait (this wave's pieces of step s+1 landed, s+2 and s+3 may be
    // in flight) sits in front of the barrier that lets anyone read step s+1.  (Before: every wave read, then every wave multiplied,
    // behind one barrier per step -- the matrix pipe idled during each read phase: 300 us for the head conv.)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
    s16x4 gl[4], gh[4], xl[8], xh[8];
    auto multiply = [&]() {
        ring_fence(gl[0], gh[0], gl[1], gh[1], gl[2], gh[2], gl[3], gh[3]);
        ring_fence(xl[0], xh[0], xl[1], xh[1], xl[2], xh[2], xl[3], xh[3]);
        ring_fence(xl[4], xh[4], xl[5], xh[5], xl[6], xh[6], xl[7], xh[7]);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        bf16x8 gf[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) gf[a] = ring_join(gl[a], gh[a]);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bf16x8 xf = ring_join(xl[b], xh[b]);
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[a], xf, acc[a][b], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    int st_r = 0;
    for (int s = 0; s <= nsteps; ++s) {
        if (grp == 1 && s > 0) multiply();                          // group 1: step s-1
        if (s < nsteps) {
            const uint16_t* sg = w3_smem + st_r * 2 * TILE;
            const uint16_t* sx = sg + TILE;
#pragma unroll
            for (int a = 0; a < 4; ++a) w3_frag(sg, wn * 64 + a * 16, lane, gl[a], gh[a]);
#pragma unroll
            for (int b = 0; b < 8; ++b) w3_frag(sx, wc * 128 + b * 16, lane, xl[b], xh[b]);
            issue_next();                                           // step s+3 (zero tiles beyond the slice)
            __builtin_amdgcn_sched_barrier(0);
            if (grp == 0) multiply();                               // group 0: step s
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        st_r = st_r == W3_STAGES - 1 ? 0 : st_r + 1;
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the run-ahead zero tiles must have landed before the LDS is released
    float* dst = p.part + (size_t)zslice * p.N * p.T * p.Cin;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int c = c0 + wc * 128 + b * 16 + (lane & 15);
            const int n = n0 + wn * 64 + a * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[((size_t)(n + r) * p.T + t) * p.Cin + c] = acc[a][b][r];
        }
}
static inline bool wgrad_wide(int N, int Cin, int T) {
    static const int on = PK_KNOB("PK_WGRAD_WIDE", 1);
    return on && T == 9 && (N % 256) == 0 && (Cin % 256) == 0;
}

// ------------------------------------------------------------------------------------------------ streaming weight-gradient kernels
// Most weight gradients of the backbone are skinny GEMMs: M = 3 072 .. 219 520 rows against an N x Cin output of 32 .. 1 024 per
// side, i.e. a few flops per operand byte -- streaming work.  k_wgrad2 stages through registers one step ahead, so a workgroup
// covers a fraction of one memory round trip per step and the launcher needs >= 2 048 short slices to hide it; every slice writes
// a whole fp32 copy of its output tile (2.2 GB of slabs per step, read again by k_reduce_many).  k_wgrad4 keeps the k_wgrad3
// machinery (LDS-DMA ring with counted vmcnt across one raw barrier per step, source-side XOR swizzle, transpose reads) on
// 64/128-wide tiles with 256 threads and two workgroups per CU: 3 (ring of 4) or 7 (ring of 8) steps in flight per workgroup, so
// ~512 long slices fill the chip and the slab volume drops with the slice count.
//   k_wgrad4<TN, TC>: single tap (linear layers, 1x1 convolutions), bias gradient = one extra MFMA against a ones fragment.
//   k_wgrad4_3x3:     3x3 stride-1 convolutions, ALL NINE TAPS from one pass over G and X.  The K loop runs over PADDED pixel
//     coordinates p = (b, py, px) of the (Hs+2) x (Ws+2) zero-bordered image: dW[n][kh][kw][c] = sum_p Gpad[p][n] * Xpad[p + (kh-1)
//     (Ws+2) + (kw-1)][c], Gpad = 0 on the border, so every tap is the SAME 32 rows of G against a row-shifted window of one
//     circular X buffer (256 rows: 51 rows of halo either side + the rows in flight); border rows are fetched with the
//     out-of-range buffer offset (the DMA writes zeros).  64 x 64 output tile x 9 taps = 144 accumulator registers, waves split
//     the c range so each wave reads 4 G fragments + 9 X fragments for 36 MFMAs.
template <int W> __device__ __forceinline__ int w4_swz(int row) {            // in 16-byte chunks; uses row bits 0, 1, 3 only
    return W == 128 ? 2 * (row & 3) + 8 * ((row >> 3) & 1) : 2 * ((row >> 1) & 1) + 4 * ((row >> 3) & 1);
}
template <int W> __device__ __forceinline__ int w4_elem(int row, int col0, int pq) {
    return row * W + ((((col0 >> 3) + (pq >> 1)) ^ w4_swz<W>(row)) << 3) + (pq & 1) * 4;
}
template <int W> __device__ __forceinline__ void w4_frag(const uint16_t* tile, int col0, int lane, s16x4& lo, s16x4& hi) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pq = i & 3;
    const uint16_t* a0 = tile + w4_elem<W>(8 * g + q, col0, pq);             // rows r and r + 4 share the swizzle (bit 2 is not used)
    lo = ring_tr(a0);
    hi = ring_tr(a0 + 4 * W);
}
__device__ __forceinline__ void ring_fence4(s16x4& a, s16x4& b, s16x4& c, s16x4& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}
template <int N> __device__ __forceinline__ void w4_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define W4_LDS(ptr) ((__attribute__((address_space(3))) void*)(ptr))

// COLS = true ("column form", 3x3 convolutions the nine-tap kernel does not take: stride 2, or stride 1 on wide images): the X operand
// is the im2col matrix [M output pixels][9 * Cin], never materialised -- the lane that stages 16-byte chunk j of a row fetches
// tap j / (Cin / 8), channels 8 (j % (Cin / 8)) of that pixel's window (its own source address per DMA lane, zero outside the image),
// and the output columns of the tile are the contiguous (tap, c) columns of dW[n][tap][c].  For the stem (Cin = 8: one chunk per tap)
// all nine taps share one 128-column tile and one pass over G instead of nine.
template <int TN, int TC, bool COLS = false>
__device__ __forceinline__ void wgrad4_body(const WgradArgs& p, const int bid) {
    constexpr int ST = (TN + TC <= 128) ? 8 : 4;                  // ring depth: 64 KB (64+64: 8 x 8 KB, 128+128: 4 x 16 KB), 48 KB otherwise
    constexpr int PG = TN / 64, PX = TC / 64, PER = PG + PX;      // 1 KiB DMA pieces per wave and step
    constexpr int RG = 512 / TN, RX = 512 / TC;                   // tile rows per piece
    constexpr int NI = TN / 32, CI = TC / 32;                     // accumulator tiles per wave (waves 2 x 2)
    constexpr int STAGE = 32 * (TN + TC);                         // elements
    __shared__ __attribute__((aligned(1024))) uint16_t ring[ST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wc = wave & 1;
    const int xcd = bid & 7, k_in = bid >> 3;
    const int zslice = xcd + 8 * (k_in / p.ntiles3), tile3 = k_in % p.ntiles3;
    if (zslice >= p.nslices3) return;            // padding workgroups of the last group of 8 slices (whole workgroup, before any barrier)
    const int ntile = tile3 / p.ctiles, ctile = tile3 - ntile * p.ctiles;
    const int n0 = ntile * TN, c0 = ctile * TC;
    const int m_begin = zslice * p.m_per_slice;
    const int m_end = min(p.M, m_begin + p.m_per_slice);
    const int nsteps = (max(m_end - m_begin, 0) + 31) / 32;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, 0x7ffffff0, 0x00020000);
    const auto rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.g), 0, 0x7ffffff0, 0x00020000);
    // this wave's pieces: G pieces wave + 4 j (j < PG), X pieces wave + 4 j (j < PX); lane -> (row in piece, physical chunk)
    const int ncols = COLS ? 9 * p.Cin : p.Cin;                      // columns of the X operand = row length of one output row
    int gm[PG], xm[PX];
    unsigned goff[PG], xoff[PX];
    bool gcol[PG], xcol[PX];
    int xb[PX], xoy[PX], xox[PX], xkh[PX], xkw[PX], xch[PX];         // column form: pixel of the lane's row (carried), its tap and channel
#pragma unroll
    for (int j = 0; j < PG; ++j) {
        const int row = (wave + 4 * j) * RG + lane / (TN / 8);
        const int col = n0 + (((lane % (TN / 8)) ^ w4_swz<TN>(row)) << 3);
        gm[j] = m_begin + row;
        gcol[j] = col < p.N;
        goff[j] = (unsigned)((gm[j] * p.N + col) * 2);
    }
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int row = (wave + 4 * j) * RX + lane / (TC / 8);
        const int col = c0 + (((lane % (TC / 8)) ^ w4_swz<TC>(row)) << 3);
        xm[j] = m_begin + row;
        xcol[j] = col < ncols;
        xoff[j] = (unsigned)((xm[j] * p.Cin + col) * 2);
        if (COLS) {
            const int tap = col / p.Cin, hw = p.Ho * p.Wo;
            xch[j] = col - tap * p.Cin;
            xkh[j] = tap / 3 - p.pad;
            xkw[j] = tap % 3 - p.pad;
            xb[j] = xm[j] / hw;
            const int r = xm[j] - xb[j] * hw;
            xoy[j] = r / p.Wo;
            xox[j] = r - xoy[j] * p.Wo;
        }
    }
    const unsigned gstep = (unsigned)(64 * p.N), xstep = (unsigned)(64 * p.Cin);       // bytes per 32 rows
    int issued = 0;
    auto issue_next = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
        uint16_t* sg = ring + (issued & (ST - 1)) * STAGE;
        uint16_t* sx = sg + 32 * TN;
        ++issued;
#pragma unroll
        for (int j = 0; j < PG; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, W4_LDS(sg + (wave + 4 * j) * 512), 16, (gcol[j] && gm[j] < m_end) ? goff[j] : OOB_OFF, 0, 0, 0);
            gm[j] += 32;
            goff[j] += gstep;
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            unsigned off = (xcol[j] && xm[j] < m_end) ? xoff[j] : OOB_OFF;
            if (COLS) {
                const int iy = xoy[j] * p.stride + xkh[j], ix = xox[j] * p.stride + xkw[j];
                const bool in = xcol[j] && xm[j] < m_end && iy >= 0 && iy < p.Hs && ix >= 0 && ix < p.Ws;
                off = in ? (unsigned)((((xb[j] * p.Hs + iy) * p.Ws + ix) * p.Cin + xch[j]) * 2) : OOB_OFF;
                xox[j] += 32;                                 // next step: 32 output pixels further, with carries
                while (xox[j] >= p.Wo) {
                    xox[j] -= p.Wo;
                    if (++xoy[j] == p.Ho) {
                        xoy[j] = 0;
                        ++xb[j];
                    }
                }
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, W4_LDS(sx + (wave + 4 * j) * 512), 16, off, 0, 0, 0);
            xm[j] += 32;
            xoff[j] += xstep;
        }
#endif
    };
    f32x4 acc[NI][CI], accb[NI];
#pragma unroll
    for (int a = 0; a < NI; ++a) {
        accb[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < CI; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = p.bias_part && ctile == 0 && wc == 0;          // wave-uniform
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
#pragma unroll
    for (int j = 0; j < ST - 1; ++j) issue_next();       // steps beyond the slice fetch zero tiles: the counts stay uniform
    for (int s = 0; s < nsteps; ++s) {
        w4_wait_vm<(ST - 2) * PER>();                    // this wave's pieces of step s have landed ...
        __builtin_amdgcn_s_barrier();                    // ... everyone's have, and everyone is done with stage (s - 1) % ST
        issue_next();
        const uint16_t* sg = ring + (s & (ST - 1)) * STAGE;
        const uint16_t* sx = sg + 32 * TN;
        s16x4 gl[NI], gh[NI], xl[CI], xh[CI];
#pragma unroll
        for (int a = 0; a < NI; ++a) w4_frag<TN>(sg, wn * (TN / 2) + a * 16, lane, gl[a], gh[a]);
#pragma unroll
        for (int b = 0; b < CI; ++b) w4_frag<TC>(sx, wc * (TC / 2) + b * 16, lane, xl[b], xh[b]);
#pragma unroll
        for (int a = 0; a < NI; a += 2) ring_fence4(gl[a], gh[a], gl[a + 1], gh[a + 1]);
#pragma unroll
        for (int b = 0; b < CI; b += 2) ring_fence4(xl[b], xh[b], xl[b + 1], xh[b + 1]);
        bf16x8 gf[NI], xf[CI];
#pragma unroll
        for (int a = 0; a < NI; ++a) gf[a] = ring_join(gl[a], gh[a]);
#pragma unroll
        for (int b = 0; b < CI; ++b) xf[b] = ring_join(xl[b], xh[b]);
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int b = 0; b < CI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[a], xf[b], acc[a][b], 0, 0, 0);
        if (do_bias) {
#pragma unroll
            for (int a = 0; a < NI; ++a) accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[a], ones, accb[a], 0, 0, 0);
        }
    }
    w4_wait_vm<0>();                                     // the run-ahead zero tiles must have landed before the LDS is released
    float* dst = p.part + (size_t)zslice * p.N * ncols;
#pragma unroll
    for (int a = 0; a < NI; ++a) {
        const int n = n0 + wn * (TN / 2) + a * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int b = 0; b < CI; ++b) {
            const int c = c0 + wc * (TC / 2) + b * 16 + (lane & 15);
            if (c < ncols) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) dst[(size_t)(n + r) * ncols + c] = acc[a][b][r];
            }
        }
        if (do_bias && (lane & 15) == 0) {               // every column of the ones product holds the column sums of G
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.N) p.bias_part[(size_t)zslice * p.N + n + r] = accb[a][r];
        }
    }
}
template <int TN, int TC, bool COLS = false>
__global__ void __launch_bounds__(256, 2) k_wgrad4(WgradArgs p) { wgrad4_body<TN, TC, COLS>(p, (int)blockIdx.x); }
// grouped form (round 4): the weight gradients of all the conv layers of one exchange-unit level in ONE launch; every member's block
// range starts at a multiple of 8, so (local id & 7) is still the XCD a slice is dealt to
struct WgradGroup {
    WgradArgs a[PK_GROUP_MAX];
    int first[PK_GROUP_MAX + 1];
    int n;
};
template <int TN, int TC, bool COLS>
__global__ void __launch_bounds__(256, 2) k_wgrad4g(WgradGroup g) {
    const int L = (int)blockIdx.x;
    int i = 0;
    while (i + 1 < g.n && L >= g.first[i + 1]) ++i;
    wgrad4_body<TN, TC, COLS>(g.a[i], L - g.first[i]);
}

// k_wgrad4w: the single-tap kernel for operands whose rows are GATHERED through the 7x7 window partition and / or SCALED per sample
// (unfused attention: qkv weight gradient reads LN(x) in window order, proj reads dy in window order times the DropPath scale;
// unfused MLP: fc2 reads dy times the scale).  Loading the int32 row map would put a dependent global load in front of every DMA
// issue, so the map is RECOMPUTED: the caller passes the token grid (B, Hs, Ws) with the map (which must be nnops.window_rowmap of
// that grid; M = B * ceil(Hs/7) * ceil(Ws/7) * 49 is checked), and every DMA lane carries (token in window, window x, window y,
// sample) from step to step.  Row scales travel with the data: one 4-byte LDS-DMA per G piece fetches g_scale[sample] for the
// piece's rows into a per-stage slot (lane-linear, i.e. one copy per 16-byte chunk of the row), the fragments are scaled in
// registers after the transpose read (the bias gradient is the column sum of the SCALED rows, as in k_wgrad2).
struct WinPos { int t, wx, wy, b; };
__device__ __forceinline__ WinPos win_split(int m, int nw, int nh) {
    WinPos r;
    const int w = m / 49;
    r.t = m - 49 * w;
    const int q = w / nw;
    r.wx = w - q * nw;
    r.b = q / nh;
    r.wy = q - r.b * nh;
    return r;
}
__device__ __forceinline__ void win_advance(WinPos& r, int nw, int nh) {      // + 32 rows (< 49: at most one window further)
    r.t += 32;
    if (r.t >= 49) {
        r.t -= 49;
        if (++r.wx == nw) {
            r.wx = 0;
            if (++r.wy == nh) {
                r.wy = 0;
                ++r.b;
            }
        }
    }
}
__device__ __forceinline__ int win_row(const WinPos& r, int H, int W) {       // pixel row of the token, -1 = zero-pad token
    const int ty = (r.t * 37) >> 8, tx = r.t - 7 * ty;                         // t / 7, t % 7 for t < 49
    const int y = r.wy * 7 + ty, x = r.wx * 7 + tx;
    return (y < H && x < W) ? (r.b * H + y) * W + x : -1;
}
__device__ __forceinline__ float ring_f32(const float* a) {
    float v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)a) : "memory");
    return v;
}
__device__ __forceinline__ s16x4 scale_bf16x4(const s16x4& v, const float* sc) {
    const uint32_t lo = pack_bf16x2(__uint_as_float((uint32_t)(uint16_t)v[0] << 16) * sc[0], __uint_as_float((uint32_t)(uint16_t)v[1] << 16) * sc[1]);
    const uint32_t hi = pack_bf16x2(__uint_as_float((uint32_t)(uint16_t)v[2] << 16) * sc[2], __uint_as_float((uint32_t)(uint16_t)v[3] << 16) * sc[3]);
    return (s16x4){(short)(lo & 0xffff), (short)(lo >> 16), (short)(hi & 0xffff), (short)(hi >> 16)};
}
// MODE: which of the three run-time variations this instantiation serves (they were uniform run-time flags tested for every DMA piece of
// every 32-row step; a step of the 128 x 128 tile was 445 instructions around 20 MFMAs): 0 = any (generic), 1 = A rows gathered through the
// window partition only (qkv weight gradient), 2 = G rows gathered + scaled per sample (proj), 3 = G rows scaled only (fc2 of the unfused MLP).
template <int TN, int TC, int MODE = 0>
__global__ void __launch_bounds__(256, 2) k_wgrad4w(WgradArgs p) {
    constexpr int ST = (TN + TC >= 256) ? 3 : 4;                  // ring <= 48 KB + <= 8 KB of row scales: two workgroups per CU
    constexpr int PG = TN / 64, PX = TC / 64, PER = 2 * PG + PX;  // DMA instructions per wave and step: data pieces + one scale piece per G piece
    constexpr int RG = 512 / TN, RX = 512 / TC;
    constexpr int NI = TN / 32, CI = TC / 32;
    constexpr int STAGE = 32 * (TN + TC), SCW = 4 * PG * 64;      // elements per stage; floats of row scales per stage
    __shared__ __attribute__((aligned(1024))) uint16_t ring[ST * STAGE];
    __shared__ __attribute__((aligned(1024))) float sS[ST * SCW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wc = wave & 1;
    const int xcd = blockIdx.x & 7, k_in = blockIdx.x >> 3;
    const int zslice = xcd + 8 * (k_in / p.ntiles3), tile3 = k_in % p.ntiles3;
    if (zslice >= p.nslices3) return;
    const int ntile = tile3 / p.ctiles, ctile = tile3 - ntile * p.ctiles;
    const int n0 = ntile * TN, c0 = ctile * TC;
    const int m_begin = zslice * p.m_per_slice;
    const int m_end = min(p.M, m_begin + p.m_per_slice);
    const int nsteps = (max(m_end - m_begin, 0) + 31) / 32;
    const bool gwin = MODE == 0 ? p.g_rowmap != nullptr : MODE == 2, xwin = MODE == 0 ? p.a_rowmap != nullptr : MODE == 1,
               scaled = MODE == 0 ? p.g_scale != nullptr : MODE >= 2;     // workgroup-uniform; compile-time in the specialised instantiations
    const int nw = (p.Ws + 6) / 7, nh = (p.Hs + 6) / 7;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, 0x7ffffff0, 0x00020000);
    const auto rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.g), 0, 0x7ffffff0, 0x00020000);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(scaled ? const_cast<float*>(p.g_scale) : (float*)const_cast<uint16_t*>(p.g), 0, 0x7ffffff0, 0x00020000);
    int gm[PG], xm[PX], gsb[PG], gsr[PG];
    unsigned gcb[PG], xcb[PX];                      // byte offset of the lane's chunk inside a source row
    bool gcol[PG], xcol[PX];
    WinPos gw[PG], xw[PX];
#pragma unroll
    for (int j = 0; j < PG; ++j) {
        const int row = (wave + 4 * j) * RG + lane / (TN / 8);
        const int col = n0 + (((lane % (TN / 8)) ^ w4_swz<TN>(row)) << 3);
        gm[j] = m_begin + row;
        gcol[j] = col < p.N;
        gcb[j] = (unsigned)(col * 2);
        gw[j] = gwin ? win_split(gm[j], nw, nh) : WinPos{0, 0, 0, 0};
        gsb[j] = scaled ? gm[j] / p.g_rows_per_sample : 0;
        gsr[j] = scaled ? gm[j] - gsb[j] * p.g_rows_per_sample : 0;
    }
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int row = (wave + 4 * j) * RX + lane / (TC / 8);
        const int col = c0 + (((lane % (TC / 8)) ^ w4_swz<TC>(row)) << 3);
        xm[j] = m_begin + row;
        xcol[j] = col < p.Cin;
        xcb[j] = (unsigned)(col * 2);
        xw[j] = xwin ? win_split(xm[j], nw, nh) : WinPos{0, 0, 0, 0};
    }
    int wr = 0;
    auto issue_next = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
        uint16_t* sg = ring + wr * STAGE;
        uint16_t* sx = sg + 32 * TN;
        float* ss = sS + wr * SCW;
        wr = (wr + 1 == ST) ? 0 : wr + 1;
#pragma unroll
        for (int j = 0; j < PG; ++j) {
            const int src = gwin ? win_row(gw[j], p.Hs, p.Ws) : gm[j];
            const bool live = gm[j] < m_end && src >= 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, W4_LDS(sg + (wave + 4 * j) * 512), 16,
                                                     (live && gcol[j]) ? (unsigned)(src * p.N * 2) + gcb[j] : OOB_OFF, 0, 0, 0);
            const int sample = gwin ? gw[j].b : gsb[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, W4_LDS(ss + (wave + 4 * j) * 64), 4, (live && scaled) ? (unsigned)(sample * 4) : OOB_OFF, 0, 0, 0);
            gm[j] += 32;
            if (gwin) win_advance(gw[j], nw, nh);
            else if (scaled) {
                gsr[j] += 32;
                while (gsr[j] >= p.g_rows_per_sample) {
                    gsr[j] -= p.g_rows_per_sample;
                    ++gsb[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int src = xwin ? win_row(xw[j], p.Hs, p.Ws) : xm[j];
            const bool live = xm[j] < m_end && src >= 0 && xcol[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, W4_LDS(sx + (wave + 4 * j) * 512), 16, live ? (unsigned)(src * p.Cin * 2) + xcb[j] : OOB_OFF, 0, 0, 0);
            xm[j] += 32;
            if (xwin) win_advance(xw[j], nw, nh);
        }
#endif
    };
    f32x4 acc[NI][CI], accb[NI];
#pragma unroll
    for (int a = 0; a < NI; ++a) {
        accb[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < CI; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool do_bias = p.bias_part && ctile == 0 && wc == 0;
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    // row scales of this lane's fragment rows 8 g .. 8 g + 7: piece = row / RG, one copy per chunk lane -> take the first
    int sidx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = 8 * (lane >> 4) + i;
        sidx[i] = (r / RG) * 64 + (r % RG) * (TN / 8);
    }
#pragma unroll
    for (int j = 0; j < ST - 1; ++j) issue_next();
    int rd = 0;
    for (int s = 0; s < nsteps; ++s) {
        w4_wait_vm<(ST - 2) * PER>();
        __builtin_amdgcn_s_barrier();
        issue_next();
        const uint16_t* sg = ring + rd * STAGE;
        const uint16_t* sx = sg + 32 * TN;
        const float* ss = sS + rd * SCW;
        rd = (rd + 1 == ST) ? 0 : rd + 1;
        s16x4 gl[NI], gh[NI], xl[CI], xh[CI];
        float sc[8];
#pragma unroll
        for (int a = 0; a < NI; ++a) w4_frag<TN>(sg, wn * (TN / 2) + a * 16, lane, gl[a], gh[a]);
#pragma unroll
        for (int b = 0; b < CI; ++b) w4_frag<TC>(sx, wc * (TC / 2) + b * 16, lane, xl[b], xh[b]);
        if (scaled) {
#pragma unroll
            for (int i = 0; i < 8; ++i) sc[i] = ring_f32(ss + sidx[i]);
        }
#pragma unroll
        for (int a = 0; a < NI; a += 2) ring_fence4(gl[a], gh[a], gl[a + 1], gh[a + 1]);
#pragma unroll
        for (int b = 0; b < CI; b += 2) ring_fence4(xl[b], xh[b], xl[b + 1], xh[b + 1]);
        if (scaled) {
            asm volatile("" : "+v"(sc[0]), "+v"(sc[1]), "+v"(sc[2]), "+v"(sc[3]), "+v"(sc[4]), "+v"(sc[5]), "+v"(sc[6]), "+v"(sc[7]));
#pragma unroll
            for (int a = 0; a < NI; ++a) {
                gl[a] = scale_bf16x4(gl[a], sc);
                gh[a] = scale_bf16x4(gh[a], sc + 4);
            }
        }
        bf16x8 gf[NI], xf[CI];
#pragma unroll
        for (int a = 0; a < NI; ++a) gf[a] = ring_join(gl[a], gh[a]);
#pragma unroll
        for (int b = 0; b < CI; ++b) xf[b] = ring_join(xl[b], xh[b]);
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int b = 0; b < CI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[a], xf[b], acc[a][b], 0, 0, 0);
        if (do_bias) {
#pragma unroll
            for (int a = 0; a < NI; ++a) accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[a], ones, accb[a], 0, 0, 0);
        }
    }
    w4_wait_vm<0>();
    float* dst = p.part + (size_t)zslice * p.N * p.Cin;
#pragma unroll
    for (int a = 0; a < NI; ++a) {
        const int n = n0 + wn * (TN / 2) + a * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int b = 0; b < CI; ++b) {
            const int c = c0 + wc * (TC / 2) + b * 16 + (lane & 15);
            if (c < p.Cin) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) dst[(size_t)(n + r) * p.Cin + c] = acc[a][b][r];
            }
        }
        if (do_bias && (lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.N) p.bias_part[(size_t)zslice * p.N + n + r] = accb[a][r];
        }
    }
}

__global__ void __launch_bounds__(256, 2) k_wgrad4_3x3(WgradArgs p) {
    constexpr int W = 64, ST = 4, XR = 256;                                   // tile width, G stages, X ring rows
    __shared__ __attribute__((aligned(1024))) uint16_t sG[ST * 32 * W];       // 16 KB
    __shared__ __attribute__((aligned(1024))) uint16_t sX[XR * W];            // 32 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, k_in = blockIdx.x >> 3;
    const int zslice = xcd + 8 * (k_in / p.ntiles3), tile3 = k_in % p.ntiles3;
    if (zslice >= p.nslices3) return;
    const int ntile = tile3 / p.ctiles, ctile = tile3 - ntile * p.ctiles;
    const int n0 = ntile * W, c0 = ctile * W;
    const int PW = p.Ws + 2, PH = p.Hs + 2, PP = PH * PW;
    const int B = p.M / (p.Hs * p.Ws), MP = B * PP;                          // stride 1, pad 1: Ho = Hs, Wo = Ws
    const int p_begin = zslice * p.m_per_slice;                              // multiple of 32
    const int p_end = min(MP, p_begin + p.m_per_slice);
    const int nsteps = (max(p_end - p_begin, 0) + 31) / 32;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, 0x7ffffff0, 0x00020000);
    const auto rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.g), 0, 0x7ffffff0, 0x00020000);
    // DMA roles: 8 rows per 1 KiB piece, piece `wave` of every 32-row block; lane -> (row in piece, physical chunk).  The padded pixel
    // of the lane's row is carried as (b, py, px) and advanced by 32 per block.
    const int prow = 8 * wave + (lane >> 3);
    struct Pix { int b, py, px; };
    auto split = [&](int q) {                     // q >= -PP
        Pix r;
        const int qq = q + PP;
        r.b = qq / PP - 1;
        const int rem = qq - (r.b + 1) * PP;
        r.py = rem / PW;
        r.px = rem - r.py * PW;
        return r;
    };
    auto advance = [&](Pix& r) {
        r.px += 32;
        while (r.px >= PW) {
            r.px -= PW;
            if (++r.py == PH) {
                r.py = 0;
                ++r.b;
            }
        }
    };
    auto interior = [&](const Pix& r) { return r.b >= 0 && r.b < B && r.py >= 1 && r.py <= p.Hs && r.px >= 1 && r.px <= p.Ws; };
    Pix gp = split(p_begin + prow);
    int gq = p_begin + prow;                                                  // padded pixel of the lane's G row
    const int gcolumn = n0 + (((lane & 7) ^ w4_swz<W>(prow)) << 3);
    const bool gcol = gcolumn < p.N;
    int xq = p_begin - 64 + prow;                                             // X runs ahead: the buffer is filled from p_begin - 64
    Pix xp = split(xq);
    const int xcolumn = c0 + (((lane & 7) ^ w4_swz<W>(xq & (XR - 1))) << 3);  // (ring row bits 0, 1, 3 never change: +32 per block)
    const bool xcol = xcolumn < p.Cin;
    int g_issued = 0;
    auto issue_x = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
        const bool ok = xcol && interior(xp);
        const unsigned off = ok ? (unsigned)((((xp.b * p.Hs + xp.py - 1) * p.Ws + xp.px - 1) * p.Cin + xcolumn) * 2) : OOB_OFF;
        const int base_row = (xq - (lane >> 3)) & (XR - 1);                   // wave-uniform: first ring row of this piece
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, W4_LDS(sX + base_row * W), 16, off, 0, 0, 0);
        xq += 32;
        advance(xp);
#endif
    };
    auto issue_g = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
        const bool ok = gcol && gq < p_end && interior(gp);
        const unsigned off = ok ? (unsigned)((((gp.b * p.Hs + gp.py - 1) * p.Ws + gp.px - 1) * p.N + gcolumn) * 2) : OOB_OFF;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, W4_LDS(sG + (g_issued & (ST - 1)) * 32 * W + 8 * wave * W), 16, off, 0, 0, 0);
        ++g_issued;
        gq += 32;
        advance(gp);
#endif
    };
    // fragment addresses of the nine taps (bytes inside sX), advanced by 32 rows = 4 096 bytes per step (swizzle bits unchanged)
    const int g4 = lane >> 4, i4 = lane & 15, q4 = i4 >> 2, pq4 = i4 & 3;
    int alo[9], ahi[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int r0 = p_begin + (t / 3 - 1) * PW + (t % 3 - 1) + 8 * g4 + q4 + XR;          // >= 0: p_begin >= 0, halo < 256
        alo[t] = 2 * w4_elem<W>(r0 & (XR - 1), 16 * wave, pq4);
        ahi[t] = 2 * w4_elem<W>((r0 + 4) & (XR - 1), 16 * wave, pq4);
    }
    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[t][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // prologue: X blocks p_begin - 64 .. p_begin + 63, then three (G, X) steps in flight
    issue_x();
    issue_x();
    issue_x();
    issue_x();
#pragma unroll
    for (int j = 0; j < ST - 1; ++j) {
        issue_g();
        issue_x();
    }
    const char* sXb = reinterpret_cast<const char*>(sX);
    for (int s = 0; s < nsteps; ++s) {
        w4_wait_vm<(ST - 2) * 2>();          // landed: G of step s and X up to p0 + 95 (taps reach p0 + 31 + PW + 1 <= p0 + 82)
        __builtin_amdgcn_s_barrier();        // everyone is done with step s - 1: its G stage and the X rows below p0 - 64 may be overwritten
        issue_g();
        issue_x();
        const uint16_t* sg = sG + (s & (ST - 1)) * 32 * W;
        // three groups of reads (G + taps 0..1 | taps 2..5 | taps 6..8); each group is in flight under the previous group's MFMAs
        s16x4 gl[4], gh[4], xl[9], xh[9];
#pragma unroll
        for (int a = 0; a < 4; ++a) w4_frag<W>(sg, a * 16, lane, gl[a], gh[a]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t == 2) {
                ring_fence(gl[0], gh[0], gl[1], gh[1], gl[2], gh[2], gl[3], gh[3]);
                ring_fence4(xl[0], xh[0], xl[1], xh[1]);
            }
            if (t == 6) ring_fence(xl[2], xh[2], xl[3], xh[3], xl[4], xh[4], xl[5], xh[5]);
            xl[t] = ring_tr(reinterpret_cast<const uint16_t*>(sXb + alo[t]));
            xh[t] = ring_tr(reinterpret_cast<const uint16_t*>(sXb + ahi[t]));
            alo[t] = (alo[t] + 32 * W * 2) & (XR * W * 2 - 1);
            ahi[t] = (ahi[t] + 32 * W * 2) & (XR * W * 2 - 1);
            if (t == 5 || t == 8) {        // MFMAs of the group fenced before this one was issued
                const int t0 = (t == 5) ? 0 : 2, t1 = (t == 5) ? 2 : 6;
#pragma unroll
                for (int u = t0; u < t1; ++u) {
                    const bf16x8 xf = ring_join(xl[u], xh[u]);
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        acc[u][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring_join(gl[a], gh[a]), xf, acc[u][a], 0, 0, 0);
                }
            }
        }
        ring_fence4(xl[6], xh[6], xl[7], xh[7]);
        asm volatile("" : "+v"(xl[8]), "+v"(xh[8]));       // (the fence above waited for all of them; this pins tap 8 behind it)
#pragma unroll
        for (int u = 6; u < 9; ++u) {
            const bf16x8 xf = ring_join(xl[u], xh[u]);
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[u][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring_join(gl[a], gh[a]), xf, acc[u][a], 0, 0, 0);
        }
    }
    w4_wait_vm<0>();
    float* dst = p.part + (size_t)zslice * p.N * 9 * p.Cin;
    const int c = c0 + 16 * wave + (lane & 15);
    if (c < p.Cin) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int n = n0 + a * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) dst[((size_t)(n + r) * 9 + t) * p.Cin + c] = acc[t][a][r];
            }
    }
}
// which weight gradients the streaming kernels take: flags bit 0 = a_rowmap, 1 = g_rowmap, 2 = g_scale
static inline int wgrad4_kind(int N, int Cin, int ksize, int stride, int Hs, int Ws, int flags) {
    static const int on = PK_KNOB("PK_WGRAD4", 15);     // bit 0: single tap, 1: nine-tap 3x3, 2: column-form 3x3, 3: window / scaled rows
    if (flags) {       // gathered / scaled rows: linear form; a row map needs the token grid it is the window partition of
        if (ksize != 1 || stride != 1 || ((flags & 3) && (Hs <= 0 || Ws <= 0))) return 0;
        return (on & 8) ? 4 : 0;
    }
    if (ksize == 1 && stride == 1) return (on & 1) ? 1 : 0;
    if (ksize == 3 && wgrad_wide(N, Cin, 9)) return 0;
    if (ksize == 3 && stride == 1 && Ws >= 1 && Ws <= 48) return (on & 2) ? 2 : 0;
    if (ksize == 3) return (on & 4) ? 3 : 0;
    return 0;
}

// out[...] = sum_s part[s][n][t][c]; layout 0: [N][T][Cin]; layout 1: OIHW = [N][Cin][T] (reference conv weight layout).
// Block = 16 elements x 16 slice-lanes (lane r sums slices s = r mod 16, fixed-order combine): short dependency chains
// even with hundreds of slices, still deterministic.
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float* __restrict__ part, float* __restrict__ out, int S, int N, int T,
                                                      int Cin, int layout, const float* __restrict__ bias_part, float* __restrict__ dbias,
                                                      int n_bias, int w_blocks) {
    __shared__ float sh[16][17];
    const int col = threadIdx.x & 15, rl = threadIdx.x >> 4;
    if ((int)blockIdx.x >= w_blocks) {          // trailing blocks: bias-gradient slabs [S][N] -> dbias[0..n_bias)
        const int n = (blockIdx.x - w_blocks) * 16 + col;
        float s = 0.f;
        if (n < n_bias)
            for (int k = rl; k < S; k += 16) s += bias_part[(size_t)k * N + n];
        sh[rl][col] = s;
        __syncthreads();
        if (rl != 0 || n >= n_bias) return;
        s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += sh[r][col];
        dbias[n] = s;
        return;
    }
    const int total = N * T * Cin;
    const int i = blockIdx.x * 16 + col;
    float s = 0.f;
    if (i < total)
        for (int k = rl; k < S; k += 16) s += part[(size_t)k * total + i];
    sh[rl][col] = s;
    __syncthreads();
    if (rl != 0 || i >= total) return;
    s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += sh[r][col];
    if (layout == 0) out[i] = s;
    else {
        const int c = i % Cin, t = (i / Cin) % T, n = i / (Cin * T);
        out[((size_t)n * Cin + c) * T + t] = s;
    }
}

// Deferred parameter-gradient reductions.  Nothing in backward reads a parameter gradient, yet every split-M weight
// gradient, LayerNorm dgamma/dbeta and rel-pos-bias gradient ended with its own small slab-reduce launch in the middle of
// the data-gradient chain (~360 of the ~2 000 launches per step, each 6-12 us plus the dependency bubble around it).  With
// dw/dgamma/dtable == NULL the producers only write their slabs; ONE table-driven launch at the end of backward reduces
// them all:   out[index(i)] = sum_{s < S} part[s * slab_stride + i],  i < K
//   layout 0: index(i) = i * out_stride;   layout 1 (conv OIHW): i = (n*T + t)*Cin + c  ->  (n*Cin + c)*T + t;
//   layout 2 (2-D slice): i = a*Cin + b  ->  a*T + b*out_stride  (a column block of a wider row-major matrix).
// Fixed-order combine (deterministic).
struct ReduceDesc { const float* part; float* out; int64_t slab_stride; int S, K, layout, N, T, Cin, out_stride, pad_; };
// Block = 64 outputs (16 groups of 4 consecutive) x 16 slab-lanes: 16-byte loads, 256 contiguous bytes per slab row and block
// (the first version read 64-byte pieces and ran at a quarter of the HBM rate: 1.17 ms per step for ~1.5 GB of slabs.  Measured and
// dropped in round 2: 1 KiB contiguous per slab row x 4 slab-lanes with four rows in flight per lane -- 1 310 us instead of 686 us;
// and, for the 3x3 weights whose OIHW destination is written one float every 36 bytes, a block per (n, 64 channels, nine taps) with
// the destination run transposed through LDS and written contiguously -- step 18.3 -> 18.56 ms.  Round 3: four independent loads per
// thread and round (rows k, k+16, k+32, k+48) -- unchanged, 637 us for the step's 2.29 GB = 3.6 TB/s; four 64-output groups per workgroup
// (45 000 workgroups instead of 180 000, four loads in flight per thread) -- 1 072 us.)
__global__ void __launch_bounds__(256) k_reduce_many(const ReduceDesc* __restrict__ desc, const int* __restrict__ blk_desc,
                                                     const int* __restrict__ blk_first) {
    __shared__ float4 sh[16][17];
    const ReduceDesc d = desc[blk_desc[blockIdx.x]];
    const int col = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int i = ((blockIdx.x - blk_first[blockIdx.x]) * 16 + col) * 4;
    const bool vec = ((d.K | (int)d.slab_stride) & 3) == 0 && (((uintptr_t)d.part) & 15) == 0;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < d.K) {
        if (vec) {
            for (int k = rl; k < d.S; k += 16) {
                const float4 v = *reinterpret_cast<const float4*>(d.part + (size_t)k * d.slab_stride + i);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        } else {
            for (int k = rl; k < d.S; k += 16) {
                const float* p = d.part + (size_t)k * d.slab_stride + i;
                s.x += p[0];
                if (i + 1 < d.K) s.y += p[1];
                if (i + 2 < d.K) s.z += p[2];
                if (i + 3 < d.K) s.w += p[3];
            }
        }
    }
    sh[rl][col] = s;
    __syncthreads();
    if (rl != 0 || i >= d.K) return;
    s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float4 v = sh[r][col];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float o4[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ij = i + j;
        if (ij >= d.K) break;
        if (d.layout == 0) d.out[(size_t)ij * d.out_stride] = o4[j];
        else if (d.layout == 2) d.out[(size_t)(ij / d.Cin) * d.T + (size_t)(ij % d.Cin) * d.out_stride] = o4[j];   // 2-D: [a][b < Cin] -> a*T + b*out_stride
        else {
            const int c = ij % d.Cin, t = (ij / d.Cin) % d.T, n = ij / (d.Cin * d.T);
            d.out[((size_t)n * d.Cin + c) * d.T + t] = o4[j];
        }
    }
}
extern "C" int pk_reduce_many_cols(void) { return 64; }     /* outputs per block: n_blocks = sum over rows of ceil(K / this) */
extern "C" int pk_reduce_many(const void* desc_table, const int* block_desc, const int* block_first, int n_blocks, void* stream) {
    PK_REQUIRE(desc_table && block_desc && block_first && n_blocks > 0, "pk_reduce_many: bad argument");
    hipLaunchKernelGGL(k_reduce_many, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, (const ReduceDesc*)desc_table, block_desc, block_first);
    return pk_launch_status("pk_reduce_many");
}

// Output tile (TN x TC): 64 x 64 or 128 x 128.  Measured and dropped: a 256 x 128 tile (128 x 64 per wave, as in k_igemm2)
// is slower here (head conv 716 us vs 523 us) -- the weight-gradient kernel is bound by its global loads (46 % of the wave
// cycles parked on s_waitcnt) and the larger tile drops it from 3 to 2 waves per SIMD; 64 pixel rows per step instead of 32
// (more bytes in flight, half the barriers) costs occupancy as well: 734 us on the 128 tile, +0.4 ms per step on the 64 tile.
// What did help the 128 tile: all-VGPR accumulators at 4 waves per SIMD (527 -> 492 us).
static inline void wgrad_tile2(int N, int Cin, int T, int& tn, int& tc) {
    if (N >= 128 && Cin >= 128) tn = tc = 128;
    else tn = tc = 64;
}
static int wgrad_slices_old(int M, int N, int Cin, int T) {
    // enough workgroups to fill 256 CUs several times over (>= 2048), but no slice shorter than 256 rows
    static const int target = PK_KNOB("PK_WGRAD_WGS", 2048);
    // (shorter slices for the low-resolution branches were measured: 64-row slices cost +1.3 ms per step in slab traffic; longer ones
    // are slower as well -- 512 / 1 024 rows: +0.35 / +1.8 ms per step -- each k_wgrad2 workgroup is bound by its own load latency)
    static const int min_rows = PK_KNOB("PK_WGRAD_ROWS", 256);
    if (wgrad_wide(N, Cin, T)) {
        // one 512-thread workgroup per CU (128 KB LDS ring): ~one round of equal-sized workgroups over the 256 CUs, in whole groups of
        // 8 slices (one slice per XCD and group)
        static const int wide_target = PK_KNOB("PK_WGRAD_WIDE_WGS", 256);
        const int tiles3 = (N / 256) * (Cin / 256) * T;
        int s3 = wide_target / tiles3 / 8 * 8;
        if (s3 < 8) s3 = 8;
        const int max3 = (M + 1023) / 1024;
        if (s3 > max3) s3 = max3;
        return s3 < 1 ? 1 : s3;
    }
    int tn, tc;
    wgrad_tile2(N, Cin, T, tn, tc);
    const int tiles = ((N + tn - 1) / tn) * ((Cin + tc - 1) / tc) * T;
    int s = (target + tiles - 1) / tiles;
    const int max_s = (M + min_rows - 1) / min_rows;
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (s > 512) s = 512;
    return s;
}
static inline void wgrad4_tile(int N, int Cin, int kind, int& tn, int& tc) {
    tn = (kind == 2 || N <= 64) ? 64 : 128;
    tc = (kind == 2 || ((kind == 1 || kind == 4) && Cin <= 64)) ? 64 : 128;         // column form: 9 * Cin >= 72 columns
}
// padded-pixel count of the 3x3 form (the K range of k_wgrad4_3x3); M = B * Hs * Ws
static inline int wgrad4_rows(int M, int Hs, int Ws, int kind) { return kind == 2 ? (M / (Hs * Ws)) * (Hs + 2) * (Ws + 2) : M; }
static int wgrad4_slices(int rows, int N, int Cin, int kind) {
    // two workgroups per CU (single tap) / one (3x3: nine accumulator sets), slices of >= 512 rows: against 256 the step is unchanged
    // (17.3 / 17.6 ms on two boxes either way) and the slabs of the low-resolution branches halve (k_reduce_many 653 -> 570 us isolated);
    // 1 024 rows starve the small launches of workgroups (step + 0.6 ms)
    static const int t1 = PK_KNOB("PK_WGRAD4_WGS", 512);
    static const int t9 = PK_KNOB("PK_WGRAD4_WGS9", 256);
    static const int min_rows = PK_KNOB("PK_WGRAD4_ROWS", 512);
    int tn, tc;
    wgrad4_tile(N, Cin, kind, tn, tc);
    const int tiles = ((N + tn - 1) / tn) * (((kind == 3 ? 9 * Cin : Cin) + tc - 1) / tc);
    int s = ((kind == 2 ? t9 : t1) + tiles - 1) / tiles;
    static const int min_rows_w = PK_KNOB("PK_WGRAD4W_ROWS", min_rows);      // kind 4: window-gathered / row-scaled
    const int mr = kind == 4 ? min_rows_w : min_rows;
    const int max_s = (rows + mr - 1) / mr;
    if (s > max_s) s = max_s;
    return s < 1 ? 1 : s;
}
extern "C" int pk_wgrad_slices(int M, int N, int Cin, int ksize, int stride, int Hs, int Ws, int flags) {
    if (M <= 0 || N <= 0 || Cin <= 0 || ksize <= 0) return 0;          // (a size query must not divide by a zero tile count)
    const int kind = wgrad4_kind(N, Cin, ksize, stride, Hs, Ws, flags);
    if (kind) return wgrad4_slices(wgrad4_rows(M, Hs, Ws, kind), N, Cin, kind);
    return wgrad_slices_old(M, N, Cin, ksize * ksize);
}

// ---- grouped weight gradients (slabs only; the caller reduces them with pk_reduce_many).  Two member kinds: 1x1 stride-1 convs (single
// tap, 64 x 64 tiles) and 3x3 stride-2 convs (column form, 64 x 128 tiles); one fixed tile shape per kind so that one launch serves all.
static inline int wgrad_group_kind(int ksize, int stride) { return (ksize == 1 && stride == 1) ? 1 : ((ksize == 3 && stride == 2) ? 3 : 0); }
extern "C" int pk_wgrad_group_slices(int M, int N, int Cin, int ksize, int stride) {
    const int kind = wgrad_group_kind(ksize, stride);
    if (!kind) return 0;
    const int tn = 64, tc = kind == 3 ? 128 : 64;
    const int tiles = ((N + tn - 1) / tn) * (((kind == 3 ? 9 * Cin : Cin) + tc - 1) / tc);
    int s = (256 + tiles - 1) / tiles;                    // ~one round of workgroups per member; slices of >= 512 rows
    const int max_s = (M + 511) / 512;
    if (s > max_s) s = max_s;
    return s < 1 ? 1 : s;
}
extern "C" int pk_wgrad_group(const PkWgradDesc* d, int n, void* stream) {
    PK_REQUIRE(d && n > 0 && n <= PK_GROUP_MAX, "pk_wgrad_group: 1..%d members, got %d", PK_GROUP_MAX, n);
    WgradGroup g{};
    const int kind = wgrad_group_kind(d[0].ksize, d[0].stride);
    PK_SUPPORTED(kind != 0, "pk_wgrad_group: 1x1 stride-1 or 3x3 stride-2 convolutions only");
    const int tn = 64, tc = kind == 3 ? 128 : 64;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const PkWgradDesc& c = d[i];
        PK_REQUIRE(c.x && c.grad_out && c.workspace, "pk_wgrad_group: null pointer");
        PK_REQUIRE(wgrad_group_kind(c.ksize, c.stride) == kind, "pk_wgrad_group: members must be of one kind");
        const int M = c.B * c.Ho * c.Wo;
        PK_REQUIRE(M > 0 && c.N > 0 && c.Cin > 0 && c.Hs > 0 && c.Ws > 0, "pk_wgrad_group: bad sizes");
        PK_SUPPORTED((c.Cin & 7) == 0 && (c.N & 7) == 0, "pk_wgrad_group: Cin=%d and N=%d must be multiples of 8", c.Cin, c.N);
        PK_REQUIRE((int64_t)M * c.N < 0x3fffffffLL && (int64_t)c.B * c.Hs * c.Ws * c.Cin < 0x3fffffffLL, "pk_wgrad_group: tensor too large");
        PK_REQUIRE(kind == 3 || (c.Ho == c.Hs && c.Wo == c.Ws), "pk_wgrad_group: stride-1 geometry");
        WgradArgs& a = g.a[i];
        a.x = (const uint16_t*)c.x; a.g = (const uint16_t*)c.grad_out; a.part = c.workspace; a.g_rows_per_sample = 1;
        a.M = M; a.N = c.N; a.Cin = c.Cin; a.T = c.ksize * c.ksize; a.Hs = c.Hs; a.Ws = c.Ws; a.Ho = c.Ho; a.Wo = c.Wo; a.stride = c.stride;
        a.pad = c.ksize / 2;
        const int S = pk_wgrad_group_slices(M, c.N, c.Cin, c.ksize, c.stride);
        a.ctiles = ((kind == 3 ? 9 * c.Cin : c.Cin) + tc - 1) / tc;
        a.ntiles3 = ((c.N + tn - 1) / tn) * a.ctiles;
        a.nslices3 = S;
        a.m_per_slice = ((M + S - 1) / S + 31) / 32 * 32;
        g.first[i] = total;
        total += 8 * ((S + 7) / 8) * a.ntiles3;
    }
    g.first[n] = total;
    g.n = n;
    hipStream_t st = (hipStream_t)stream;
    if (kind == 3) hipLaunchKernelGGL((k_wgrad4g<64, 128, true>), dim3((unsigned)total), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((k_wgrad4g<64, 64, false>), dim3((unsigned)total), dim3(256), 0, st, g);
    return pk_launch_status("pk_wgrad_group");
}

extern "C" int pk_wgrad_bf16(const void* x, const void* grad_out, float* workspace, float* dw, float* dbias, int n_bias,
                             const int32_t* a_rowmap, const int32_t* g_rowmap, const float* g_scale, int g_rows_per_sample, int M,
                             int N, int Cin, int ksize, int stride, int B, int Hs, int Ws, int Ho, int Wo, int out_layout,
                             void* stream) {
    PK_REQUIRE(x && grad_out && workspace, "pk_wgrad_bf16: null pointer");
    PK_REQUIRE(M > 0 && N > 0 && Cin > 0 && (ksize == 1 || ksize == 3), "pk_wgrad_bf16: bad sizes");
    PK_SUPPORTED((Cin & 7) == 0 && (N & 7) == 0, "pk_wgrad_bf16: Cin=%d and N=%d must be multiples of 8", Cin, N);
    const bool linear = (Ho == 0);
    if (!linear) PK_REQUIRE(M == B * Ho * Wo && (int64_t)B * Hs * Ws * Cin < 0x7fffffffLL, "pk_wgrad_bf16: geometry");
    PK_REQUIRE(linear || (!a_rowmap && !g_rowmap), "pk_wgrad_bf16: row maps are for the linear form only");
    WgradArgs a{};
    PK_REQUIRE(!g_scale || g_rows_per_sample > 0, "pk_wgrad_bf16: g_scale needs g_rows_per_sample");
    a.x = (const uint16_t*)x; a.g = (const uint16_t*)grad_out; a.part = workspace; a.a_rowmap = a_rowmap; a.g_rowmap = g_rowmap;
    a.g_scale = g_scale; a.g_rows_per_sample = g_rows_per_sample > 0 ? g_rows_per_sample : 1;
    a.M = M; a.N = N; a.Cin = Cin; a.T = ksize * ksize; a.Hs = Hs; a.Ws = Ws; a.Ho = Ho; a.Wo = Wo; a.stride = stride; a.pad = ksize / 2;
    const int kind4 = wgrad4_kind(N, Cin, ksize, stride, Hs, Ws, (a_rowmap ? 1 : 0) | (g_rowmap ? 2 : 0) | (g_scale ? 4 : 0));
    const int S = pk_wgrad_slices(M, N, Cin, ksize, stride, Hs, Ws, (a_rowmap ? 1 : 0) | (g_rowmap ? 2 : 0) | (g_scale ? 4 : 0));
    PK_REQUIRE(n_bias >= 0 && n_bias <= N && (!dbias || n_bias > 0), "pk_wgrad_bf16: n_bias");
    a.bias_part = n_bias > 0 ? workspace + (size_t)S * N * a.T * Cin : nullptr;     // bias slabs follow the weight slabs
    a.m_per_slice = ((M + S - 1) / S + 63) / 64 * 64;
    int tn, tc;
    wgrad_tile2(N, Cin, a.T, tn, tc);
    a.ctiles = (Cin + tc - 1) / tc;
    hipStream_t st = (hipStream_t)stream;
    PK_REQUIRE((int64_t)M * N < 0x3fffffffLL && (linear || (int64_t)B * Hs * Ws * Cin < 0x3fffffffLL), "pk_wgrad_bf16: tensor too large for 32-bit byte offsets");
    if (kind4) {
        if (kind4 == 4 && (a_rowmap || g_rowmap)) {
            const int nwin = ((Hs + 6) / 7) * ((Ws + 6) / 7);
            PK_REQUIRE(B > 0 && M == B * nwin * 49, "pk_wgrad_bf16: M = %d is not the window-order row count of a (%d, %d, %d) token grid", M, B, Hs, Ws);
            PK_REQUIRE(!g_scale || !g_rowmap || g_rows_per_sample == Hs * Ws, "pk_wgrad_bf16: g_rows_per_sample must be Hs * Ws with a window map");
            PK_REQUIRE((int64_t)B * Hs * Ws * (N > Cin ? N : Cin) < 0x3fffffffLL, "pk_wgrad_bf16: tensor too large for 32-bit byte offsets");
        }
        PK_SUPPORTED(kind4 == 1 || kind4 == 4 || n_bias == 0, "pk_wgrad_bf16: the 3x3 streaming kernel has no bias-gradient path (convolutions here carry no bias)");
        PK_REQUIRE(linear || kind4 == 3 || (Ho == Hs && Wo == Ws), "pk_wgrad_bf16: stride-1 geometry");
        int t4n, t4c;
        wgrad4_tile(N, Cin, kind4, t4n, t4c);
        const int rows = wgrad4_rows(M, Hs, Ws, kind4);
        PK_REQUIRE((int64_t)rows * 2 < 0x3fffffffLL, "pk_wgrad_bf16: too many rows");
        a.ctiles = ((kind4 == 3 ? 9 * Cin : Cin) + t4c - 1) / t4c;
        a.ntiles3 = ((N + t4n - 1) / t4n) * a.ctiles;
        a.nslices3 = S;
        a.m_per_slice = ((rows + S - 1) / S + 31) / 32 * 32;
        const dim3 grid(8 * ((S + 7) / 8) * a.ntiles3);
        if (kind4 == 4) {
            static const int w4_modes = PK_KNOB("PK_WGRAD4W_MODES", 1);
            const bool gw_ = a.g_rowmap != nullptr, xw_ = a.a_rowmap != nullptr, sc_ = a.g_scale != nullptr;
            const int mode = !w4_modes ? 0 : (xw_ && !gw_ && !sc_) ? 1 : (gw_ && sc_ && !xw_) ? 2 : (sc_ && !gw_ && !xw_) ? 3 : 0;
#define W4W_GO(TN_, TC_)                                                                                     \
    do {                                                                                                     \
        if (mode == 1) hipLaunchKernelGGL((k_wgrad4w<TN_, TC_, 1>), grid, dim3(256), 0, st, a);              \
        else if (mode == 2) hipLaunchKernelGGL((k_wgrad4w<TN_, TC_, 2>), grid, dim3(256), 0, st, a);         \
        else if (mode == 3) hipLaunchKernelGGL((k_wgrad4w<TN_, TC_, 3>), grid, dim3(256), 0, st, a);         \
        else hipLaunchKernelGGL((k_wgrad4w<TN_, TC_, 0>), grid, dim3(256), 0, st, a);                        \
    } while (0)
            if (t4n == 64 && t4c == 64) W4W_GO(64, 64);
            else if (t4n == 64) W4W_GO(64, 128);
            else if (t4c == 64) W4W_GO(128, 64);
            else W4W_GO(128, 128);
#undef W4W_GO
        } else if (kind4 == 2) hipLaunchKernelGGL(k_wgrad4_3x3, grid, dim3(256), 0, st, a);
        else if (kind4 == 3 && t4n == 64) hipLaunchKernelGGL((k_wgrad4<64, 128, true>), grid, dim3(256), 0, st, a);
        else if (kind4 == 3) hipLaunchKernelGGL((k_wgrad4<128, 128, true>), grid, dim3(256), 0, st, a);
        else if (t4n == 64 && t4c == 64) hipLaunchKernelGGL((k_wgrad4<64, 64>), grid, dim3(256), 0, st, a);
        else if (t4n == 64) hipLaunchKernelGGL((k_wgrad4<64, 128>), grid, dim3(256), 0, st, a);
        else if (t4c == 64) hipLaunchKernelGGL((k_wgrad4<128, 64>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_wgrad4<128, 128>), grid, dim3(256), 0, st, a);
    } else if (wgrad_wide(N, Cin, a.T)) {
        PK_SUPPORTED(!linear && n_bias == 0, "pk_wgrad_bf16: the wide 3x3 kernel has no bias-gradient path (convolutions here carry no bias)");
        a.ctiles = Cin / 256;
        a.ntiles3 = (N / 256) * a.ctiles;
        a.nslices3 = S;
        static bool attr_set = false;
        constexpr int W3_LDS = W3_STAGES * 2 * W3_ROWS * 256 * 2;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)k_wgrad3, hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS);
            PK_REQUIRE(e == hipSuccess, "pk_wgrad_bf16: cannot reserve %d bytes of LDS: %s", W3_LDS, hipGetErrorString(e));
            attr_set = true;
        }
        hipLaunchKernelGGL(k_wgrad3, dim3(8 * ((S + 7) / 8) * 9 * a.ntiles3), dim3(512), W3_LDS, st, a);
    } else {
        dim3 grid(((N + tn - 1) / tn) * a.ctiles, a.T, S);
        static const int xcd9 = PK_KNOB("PK_WGRAD2_XCD", 1);
        if (a.T == 9 && xcd9) {
            a.ntiles3 = grid.x;
            a.nslices3 = S;
            grid = dim3(8 * ((S + 7) / 8) * 9 * a.ntiles3);
        }
        if (tn == 128) hipLaunchKernelGGL((k_wgrad2<128, 128, 32>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_wgrad2<64, 64, 32>), grid, dim3(256), 0, st, a);
    }
    if (!dw) return pk_launch_status("pk_wgrad_bf16");        // slabs only: the caller reduces them later (pk_reduce_many)
    const int total = N * a.T * Cin;
    const int w_blocks = (total + 15) / 16, b_blocks = dbias ? (n_bias + 15) / 16 : 0;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(w_blocks + b_blocks), dim3(256), 0, st, workspace, dw, S, N, a.T, Cin, out_layout,
                       a.bias_part, dbias, n_bias, w_blocks);
    return pk_launch_status("pk_wgrad_bf16");
}
