// Implicit-GEMM convolution (1 x 1 / 3 x 3, stride 1 / 2) on channels-last bf16 or fp16 (template parameter T) for gfx950, with the bias / ReLU / skip adds
// of DPT's decoder fused into the epilogue.  No vendor library.
//
// Replaces the 3 x 3 convolutions of the reference's (absent) third_party/dpt decoder inside `DPTDepthModel.forward`
// (call site /root/reference/hive/dataset_adaptors.py:1419): `scratch.layer{1..4}_rn`, the two convolutions of every
// `ResidualConvUnit_custom` of the four `FeatureFusionBlock_custom`s (out = conv2(relu(conv1(relu(x)))) + x (+ skip)) and
// `scratch.output_conv[0]` -- 272 of the network's 530 GFLOP per 480 x 640 frame.
//
// Formulation: out[m][co] = sum_k A[m][k] W[co][k] with m = (image, y, x), k = (ky, kx, ci).  With channels-last activations
// the 64-channel slice of one tap of one output pixel is 128 contiguous bytes of the input, so the im2col matrix is never
// built: the LDS-DMA that stages the A tile (global_load_lds_dwordx4, per-lane source address) reads it straight from the
// shifted pixel -- and from a 128-byte page of zeros where the tap falls into the padding.  The weight tensor
// [Cout][Cin][3][3] in channels-last memory format IS W[co][(ky, kx, ci)].
//
// Tile: TM = 256 (or 128, for small maps) output pixels x TN = 256 / 128 / 64 output channels, 8 waves (256 x 256: 2 x 4 waves of 128 x 64),
// K-step 64 = one tap x 64 input channels, v_mfma_f32_16x16x32_bf16, two LDS stages filled by LDS-DMA with the bank swizzle
// on the source side (conflict-free ds_read_b128), one raw s_barrier per K-step, XCD-aware tile order: consecutive tiles are
// consecutive image rows, so an XCD's run of tiles re-reads its three-row halo from its own L2.  K = 9 Cin >= 2304 gives
// 36+ K-steps per tile (the ViT GEMMs have 12), so the tile's ends are a small share.
//
// Measured (tools/probe_conv3x3.py, 24 frames): 840-910 TFLOP/s on the 120 x 160 / 60 x 80 / 240 x 320 shapes (MIOpen on the same
// tensors: 770-920), 500-580 at 30 x 40 (113 tiles for 256 CUs) and 140-160 at 15 x 20 (29 tiles).  Built, measured and taken
// out again: a deep LDS ring (K-step 32, 3 / 4 / 5 stages of 32 KiB in flight, counted vmcnt) -- 725-780 TFLOP/s whatever the
// depth, i.e. the fill latency is NOT what bounds the K-step, and the second barrier per 64 k costs 13 %.  What does bound it: a CU's
// vector-memory path moves one 64-byte line per ~2.3 cycles (tools/ubench/gather.hip, ldsdma.hip), so the 64 KiB of a 256 x 256 x 64
// stage take 2550+ cycles to arrive against 2048 cycles of MFMA -- per-workgroup clocks (tools/probe_gemm_stamps.py) show 2100 busy +
// 800 waiting cycles per K-step.  Also tried and taken out: start delays that de-phase the persistent workgroups (so that their
// epilogues' stores do not hit HBM together): no change where the delay is free (workgroups with one tile fewer), slower elsewhere.
// And a timing experiment that settles what a smarter A path could buy (an LDS-resident pixel window re-used by the 9 taps): with
// three quarters of the A pieces simply not issued the 3 x 3 convolutions run at 1144 instead of 1118 TFLOP/s -- the operand
// stream is not what the K-step waits for.
#include "hive_internal.hpp"
#include "mfma_pipe.hpp"

#include <algorithm>

// tuning builds only (make ablate_conv; tools/probe_resnet.py): phases of conv_kernel left out -- 1: the epilogue's turn through LDS and its
// stores, 2: the MFMAs (the stage pieces are still issued), 4: the LDS-DMA, 8: the GroupNorm sums' reduction and their store
#ifndef HIVE_CONV_AHEAD_GN2
#define HIVE_CONV_AHEAD_GN2 4  // the same for the second pass of the two-pass GroupNorm convolutions (fewer live registers there)
#endif
#ifndef HIVE_CONV_ABLATE
#define HIVE_CONV_ABLATE 0
#endif
#ifndef HIVE_CONV_AHEAD
#define HIVE_CONV_AHEAD 4  // fragment rows the shortcut loads of the epilogue run ahead (ONE shortcut; with two: half as many)
#endif

using hive_mfma::f32x4;
using hive_mfma::vec;  // vec<T, 8>: 8 elements of the 16-bit type T (__bf16 or _Float16), one 16-byte register quad

namespace {

template <typename T>
struct ConvParams {
    const T *x;      // [NB][H][W][Cin]
    const T *w;      // [Cout][R S Cin], k = (ky, kx, ci)
    const T *bias;   // [Cout] or nullptr
    const T *res1;   // [M][Cout] or nullptr
    const T *res2;   // [M][Cout] or nullptr
    T *out;          // [M][Cout]
    T *out_relu;     // [M][Cout] or nullptr: relu(out)
    const T *zeros;  // >= 128 bytes of zeros (padding taps)
    int H, W, Cin, Cout, relu;  // H, W: INPUT size
    int Ho, Wo;         // output size
    int S, taps;        // kernel width (1 or 3), R * S
    int stride, pad_t, pad_l;  // input row of tap ky for output row oy: oy * stride + ky - pad_t
    int M;              // NB * Ho * Wo
    float *gn_partial;  // GN kernels only: [M tile][2 images of the tile][sum, sum of squares][Cout] of the stored (rounded) outputs
    int stats_only;     // GN == 1: nothing is stored but gn_partial (first pass of hive_nhwc_conv_gn_apply)
    // GN == 2 (second pass): out = relu?(bf16(gn(bf16(conv))) + residual) with the (mean, rstd) of gn_stats[sample * gn_G + group]
    const float *gn_stats;
    const T *gn_gamma, *gn_beta;
    int gn_G, gn_cpg;   // groups, channels per group (>= 8: a lane's 8 channels share a group)
    // split-K (conv_deep_kernel only: mfma_pipe.hpp splitk_combine): the K-steps of a tile are dealt to split_k workgroups
    int split_k;
    hive_mfma::f32x4 *sk_ws;
    unsigned *sk_count;
};

// epilogue: through the wave's 4 KiB of LDS (mfma_pipe.hpp staged_rows) so that the residual loads and the stores are 16 bytes
// per lane on whole 128-byte lines; a lane gets 8 consecutive output channels (always the same ones) of one pixel; everything in
// f32, one rounding
//
// GN == 1 (the ResNetV2 convolutions, each followed by a GroupNorm): the statistics pass of that GroupNorm is folded in -- the sums of the
// stored (rounded) outputs and of their squares per channel, separately for the two images a tile of TM <= Ho Wo rows can touch (rows
// below / from `boundary`), taken from the accumulators (gn_sums_from_acc below), added over the waves of the tile through LDS in wave
// order -- a fixed order, run-to-run identical -- and written as the tile's row of `gn_partial`.  hive_nhwc_group_norm_stats (dpt_ops.hip)
// finishes from there: no pass over the tensor for statistics.  GN == 1 is launched for plain epilogues only (bias at most).
// y = x * (rstd * gamma) + (beta - mean * (rstd * gamma)) with every operation rounded on its own, as gn_apply_kernel (dpt_ops.hip, built
// with -ffp-contract=off) computes it (HIP's __fmul_rn / __fadd_rn are plain operators and contract to an FMA in this file)
__device__ __forceinline__ float gn_affine_exact(float x, float rstd, float gamma, float beta, float mean) {
#pragma clang fp contract(off)
    const float a = rstd * gamma;
    const float b = beta - mean * a;
    return x * a + b;
}

template <typename T, int MT, int GN, int NRES = 1>
__device__ __forceinline__ void conv_epilogue(const ConvParams<T> &p, f32x4 (&acc)[4][MT], int m_base, int n_base, unsigned char *stage, int lane,
                                              int boundary) {
    // `lane` made opaque per tile: otherwise the compiler computes every lane-constant of the epilogue (row numbers, LDS addresses) once in
    // front of the persistent tile loop, runs out of registers over the K loop and SPILLS them -- reloads between the hand-placed LDS-DMA
    asm volatile("" : "+v"(lane));
    const int n = n_base + (lane & 7) * 8;
    float b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        const vec<T, 8> bv = *reinterpret_cast<const vec<T, 8> *>(p.bias + n);
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = (float)bv[j];
    }
    float ggam[8], gbet[8], gmean[2] = {0.f, 0.f}, grstd[2] = {0.f, 0.f};
    if (GN == 2) {
        const vec<T, 8> gv = *reinterpret_cast<const vec<T, 8> *>(p.gn_gamma + n), bv = *reinterpret_cast<const vec<T, 8> *>(p.gn_beta + n);
#pragma unroll
        for (int j = 0; j < 8; ++j) ggam[j] = (float)gv[j], gbet[j] = (float)bv[j];
        const int hw = p.Ho * p.Wo, img0 = boundary / hw - 1, n_img = p.M / hw, g = n / p.gn_cpg;
        const float *st = p.gn_stats + ((size_t)img0 * p.gn_G + g) * 2;
        gmean[0] = st[0], grstd[0] = st[1];
        if (img0 + 1 < n_img) gmean[1] = st[2 * p.gn_G], grstd[1] = st[2 * p.gn_G + 1];
    }
    // the shortcut / skip rows: loaded AHEAD fragment rows before their use (AHEAD + 1 register sets), so that a load's trip to memory runs
    // under the previous rows' turns through LDS instead of in front of every store (as gemm_store_rows in vit.hip).  A fragment row's turn
    // is ~0.5 us, a trip to HBM under load 2 us: with one row ahead the second pass of conv3 at 120 x 160 took 660 us, with two 600, with
    // FOUR (8 KiB per wave, 16 MB over the chip in flight) 500.  NRES = 2 (the residual unit that adds its input AND the path from above:
    // twice the bytes per row) runs two ahead -- its six register sets are what four ahead costs with one shortcut.  A residual may BE the
    // output: every element is read by the lane that later writes it and rows of mt + AHEAD are read before rows of mt are written.
    constexpr int AHEAD = GN == 2 ? HIVE_CONV_AHEAD_GN2 : (NRES == 2 ? HIVE_CONV_AHEAD / 2 : HIVE_CONV_AHEAD);
    vec<T, 8> rs1[AHEAD + 1][2], rs2[NRES == 2 ? AHEAD + 1 : 1][2];
    const int row_in_frag = lane >> 3;
    auto pre = [&](int mt, int j) {
        const int m = min(m_base + mt * 16 + 8 * j + row_in_frag, p.M - 1);
        const size_t off = (size_t)m * p.Cout + n;
        if (GN != 1 && p.res1) rs1[mt % (AHEAD + 1)][j] = *reinterpret_cast<const vec<T, 8> *>(p.res1 + off);
        if (GN == 0 && NRES == 2) rs2[mt % (AHEAD + 1)][j] = *reinterpret_cast<const vec<T, 8> *>(p.res2 + off);
    };
    hive_mfma::staged_rows<MT, AHEAD>(stage, acc, lane, pre, [&](int r, int, const f32x4 &lo, const f32x4 &hi, int mt, int jrow) {
        const int m = m_base + r;
        if (m >= p.M) return;
        const size_t o_off = (size_t)m * p.Cout + n;
        float o[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = lo[j] + b[j], o[4 + j] = hi[j] + b[4 + j];
        if (GN == 0 && p.res1) {
            const vec<T, 8> rs = rs1[mt % (AHEAD + 1)][jrow];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += (float)rs[j];
        }
        if (GN == 0 && NRES == 2) {
            const vec<T, 8> rs = rs2[mt % (AHEAD + 1)][jrow];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += (float)rs[j];
        }
        if (GN == 0 && p.relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = fmaxf(o[j], 0.0f);
        }
        vec<T, 8> ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) ov[j] = (T)o[j];
        if (GN == 2) {
            // the GroupNorm behind this convolution, applied to the ROUNDED output with gn_apply_kernel's operations in its order
            // (no contraction: bit-identical to the separate pass), then the block's shortcut and ReLU
            const bool second = m >= boundary;
            const float mean = second ? gmean[1] : gmean[0], rstd = second ? grstd[1] : grstd[0];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = gn_affine_exact((float)ov[j], rstd, ggam[j], gbet[j], mean);
            if (p.res1) {
                const vec<T, 8> rs = rs1[mt % (AHEAD + 1)][jrow];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (float)(T)o[j] + (float)rs[j];
            }
            if (p.relu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = fmaxf(o[j], 0.0f);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[j] = (T)o[j];
            *reinterpret_cast<vec<T, 8> *>(p.out + o_off) = ov;
            return;
        }
        if (GN != 1 || !p.stats_only) *reinterpret_cast<vec<T, 8> *>(p.out + o_off) = ov;
        if (GN == 0 && p.out_relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[j] = (T)fmaxf(o[j], 0.0f);
            *reinterpret_cast<vec<T, 8> *>(p.out_relu + o_off) = ov;
        }
    });
}

// GroupNorm sums straight from the accumulators (GN == 1 where the epilogue has no shortcut / ReLU: every ResNetV2 convolution).  In the
// MFMA layout a lane holds 4 consecutive channels (x 4 fragments along N) of pixel `fr` of each of its MT fragment rows: it rounds them to
// T (the values the GroupNorm will read), and adds them and their squares with dot instructions -- 1.5 VALU instructions per
// output instead of the ~11 of the path through the LDS turn-around (measured: the statistics-only pass of conv3 at 120 x 160 spent
// 245 of its 322 us there).  The 16 pixels of a fragment row sit in the 16 lanes of a DPP row: four row_ror additions leave the row's
// total in each of its lanes (no LDS, unlike the ds_bpermute behind __shfl_xor); a fixed order, run-to-run identical.
// wsum[(h 2 + k) 64 + c]: sums (k = 0) / sums of squares (k = 1) of channel n_base + c over the wave's rows of the tile's first (h = 0: rows
// below `boundary`) and second image -- written AFTER the epilogue's stores (the accumulators are still there; the wave's 4 KiB of LDS is free).
__device__ __forceinline__ float dot2acc(vec<__bf16, 2> a, vec<__bf16, 2> b, float c) { return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false); }
__device__ __forceinline__ float dot2acc(vec<_Float16, 2> a, vec<_Float16, 2> b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
__device__ __forceinline__ float dpp_row_total(float v) {
#define HIVE_ROR_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    HIVE_ROR_ADD(0x128);  // row_ror:8
    HIVE_ROR_ADD(0x124);
    HIVE_ROR_ADD(0x122);
    HIVE_ROR_ADD(0x121);
#undef HIVE_ROR_ADD
    return v;
}
template <typename T, int MT>
__device__ __forceinline__ void gn_sums_from_acc(const ConvParams<T> &p, const f32x4 (&acc)[4][MT], int m_base, int n_base, int lane, int boundary,
                                                 float *wsum) {
    asm volatile("" : "+v"(lane));  // (as in conv_epilogue)
    const int fr = lane & 15, fq = lane >> 4;
    const int m_end = m_base + MT * 16;                                                     // (wave-uniform, as everything below that decides a branch)
    const bool one_image = m_end <= p.M && (m_end <= boundary || m_base >= boundary);  // every row of the wave is stored and lies in ONE image: no masks
    const bool second = m_base >= boundary;
    // one fragment column (16 channels) at a time, so that only its sums are live beside the accumulators.  Two pixels of one channel are
    // rounded into one packed register (v_cvt_pk) and v_dot2c_f32_{bf16,f16} adds both (against packed ones) and both squares (against
    // itself) into the f32 sums: 1.5 instructions per output.
    vec<T, 2> ones;
    ones[0] = ones[1] = (T)1.0f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            const vec<T, 4> bv = *reinterpret_cast<const vec<T, 4> *>(p.bias + n_base + nt * 16 + fq * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[e] = (float)bv[e];
        }
        f32x4 sv[2], qv[2];  // [image][channel]
        sv[0] = sv[1] = qv[0] = qv[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (one_image) {
            f32x4 ts = f32x4{0.f, 0.f, 0.f, 0.f}, tq = ts;
#pragma unroll
            for (int mt = 0; mt < MT; mt += 2) {
                const f32x4 a0 = acc[nt][mt] + bias, a1 = acc[nt][mt + 1] + bias;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vec<T, 2> y;
                    y[0] = (T)a0[e], y[1] = (T)a1[e];
                    ts[e] = dot2acc(y, ones, ts[e]);
                    tq[e] = dot2acc(y, y, tq[e]);
                }
            }
            const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
            sv[0] = second ? zero4 : ts;
            sv[1] = second ? ts : zero4;
            qv[0] = second ? zero4 : tq;
            qv[1] = second ? tq : zero4;
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; mt += 2) {
                const int m0 = m_base + mt * 16 + fr, m1 = m0 + 16;
                const bool v0 = m0 < p.M, v1 = m1 < p.M, s0 = m0 >= boundary, s1 = m1 >= boundary;
                const f32x4 a0 = acc[nt][mt] + bias, a1 = acc[nt][mt + 1] + bias;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vec<T, 2> y, z;  // the pair's members of the first / second image, zero otherwise
                    y[0] = (T)(v0 && !s0 ? a0[e] : 0.f), y[1] = (T)(v1 && !s1 ? a1[e] : 0.f);
                    z[0] = (T)(v0 && s0 ? a0[e] : 0.f), z[1] = (T)(v1 && s1 ? a1[e] : 0.f);
                    sv[0][e] = dot2acc(y, ones, sv[0][e]);
                    qv[0][e] = dot2acc(y, y, qv[0][e]);
                    sv[1][e] = dot2acc(z, ones, sv[1][e]);
                    qv[1][e] = dot2acc(z, z, qv[1][e]);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, qt = f32x4{0.f, 0.f, 0.f, 0.f};
            if (h == 0 || m_end > boundary) {  // (the wave reaches into the second image at all)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st[e] = dpp_row_total(sv[h][e]);
                    qt[e] = dpp_row_total(qv[h][e]);
                }
            }
            if (fr == 0) {  // a row's total is in each of its lanes: lane 16 fq writes channels nt 16 + fq 4 + (0..3)
                *reinterpret_cast<f32x4 *>(wsum + (h * 2 + 0) * 64 + nt * 16 + fq * 4) = st;
                *reinterpret_cast<f32x4 *>(wsum + (h * 2 + 1) * 64 + nt * 16 + fq * 4) = qt;
            }
        }
    }
}

constexpr int BK = 64;


template <typename T, int TM, int TN, int GN>
__global__ __launch_bounds__(512, 1) void conv_kernel(ConvParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 stages x (A tile TM x 64, W tile TN x 64) + 8 x 4 KiB for the epilogue
    constexpr int A_GROUPS = TM / 8, A_PW = A_GROUPS / 8;  // 1 KiB groups of the A tile, and how many each wave stages
    constexpr int W_GROUPS = TN / 8, GROUPS = A_GROUPS + W_GROUPS, PER_WAVE = GROUPS / 8;
    constexpr int STAGE_BYTES = GROUPS * 1024;
    constexpr int WN = TN / 64, WM = 8 / WN, RW = TM / WM, MT = RW / 16;  // waves along N / M, rows per wave, M fragments (TN = 64: 8 x 1 waves of 32 x 64)
    const int tid = threadIdx.x;
    int lane = tid & 63;  // made opaque once per tile (top of the tile loop): the lane-constants of the stage pieces are then recomputed per tile
                          // instead of living -- spilled, with both skip connections' register sets in the epilogue -- across the whole loop
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    // PERSISTENT workgroups, XCD-aware (one per CU: the two stages take 96-128 KiB of LDS): the grid is a multiple of 8;
    // workgroups are dealt round-robin over the 8 XCDs, each XCD owns a contiguous run of tiles (consecutive image rows: its
    // tiles re-read their three-row halo from its own L2) and its workgroups walk the run with a stride of gridDim / 8.  The
    // K-steps of a workgroup's tiles form ONE stream: the first stage of the next tile is issued between the MFMAs of the last
    // K-step of the current one and lands while its epilogue runs -- what the 1 x 1 convolutions of the ResNet stages need
    // (K = 64 .. 1024: one to sixteen K-steps per tile, then 64-128 KiB of output to store).
    const int tiles_n = p.Cout / TN, n_tiles = ((p.M + TM - 1) / TM) * tiles_n;
    const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3, tq = n_tiles >> 3, tr = n_tiles & 7;
    const int run0 = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, run_n = tq + (xcd < tr ? 1 : 0);
    int tl = blockIdx.x >> 3;  // position in the XCD's run
    if (tl >= run_n) return;   // (whole workgroup)
    const int K = p.taps * p.Cin, CPT = p.Cin / BK;  // CPT: K-steps (channel blocks) per tap

    // per tile: the A rows this lane stages (the same ones in every K-step): groups wave, wave + 8, ...
    struct Tile {
        int m0, n0;
        int py[A_PW], px[A_PW];
        const T *pbase[A_PW];
    };
    int a_chunk[A_PW];
    auto lane_constants = [&]() {
#pragma unroll
        for (int j = 0; j < A_PW; ++j) {
            const int row = (wave + 8 * j) * 8 + (lane >> 3);
            a_chunk[j] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;  // source-side swizzle: LDS slot (lane & 7) receives this chunk
        }
    };
    lane_constants();
    auto setup = [&](int t, Tile &tile) {
        tile.m0 = (t / tiles_n) * TM;
        tile.n0 = (t % tiles_n) * TN;
#pragma unroll
        for (int j = 0; j < A_PW; ++j) {
            const int row = (wave + 8 * j) * 8 + (lane >> 3);
            const int m = min(tile.m0 + row, p.M - 1);  // rows past the end are never stored
            const int img = m / (p.Ho * p.Wo), rem = m - img * (p.Ho * p.Wo);
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            tile.py[j] = oy * p.stride - p.pad_t;  // input pixel of tap (0, 0)
            tile.px[j] = ox * p.stride - p.pad_l;
            tile.pbase[j] = p.x + (((long long)img * p.H + tile.py[j]) * p.W + tile.px[j]) * p.Cin + a_chunk[j];  // may point before the image: used only when inside
        }
    };

    // one LDS-DMA wave-instruction of a stage: j < A_PW an A group (8 output pixels x 128 B of one tap), else a W group
    auto issue_piece = [&](const Tile &tile, int stage, int tap, int cc, int j) {
        unsigned char *st = lds + stage * STAGE_BYTES;
        if (HIVE_CONV_ABLATE & 4) return;
        if (j < A_PW) {
            const int dy = tap / p.S, dx = tap - dy * p.S;
            const long long shift = ((long long)dy * p.W + dx) * p.Cin + cc * BK;
            const bool inside = (unsigned)(tile.py[j] + dy) < (unsigned)p.H && (unsigned)(tile.px[j] + dx) < (unsigned)p.W;
            const T *g = inside ? tile.pbase[j] + shift : p.zeros + a_chunk[j];
            __builtin_amdgcn_global_load_lds((const void *)g, (__attribute__((address_space(3))) void *)(st + (wave + 8 * j) * 1024), 16, 0, 0);
        } else {
            const int grp = wave + 8 * (j - A_PW);  // W group: rows grp * 8 .. + 7 of the weight tile
            const int row = grp * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            const T *g = p.w + (size_t)(tile.n0 + row) * K + tap * p.Cin + cc * BK + chunk * 8;
            __builtin_amdgcn_global_load_lds((const void *)g, (__attribute__((address_space(3))) void *)(st + (A_GROUPS + grp) * 1024), 16, 0, 0);
        }
    };

    const int KT = p.taps * CPT, last_tap = p.taps - 1;
    // K order within a tile: channel block OUTER, tap INNER.  The taps of one 64-channel block read the same input rows shifted
    // by a pixel or a row, so consecutive K-steps re-read bytes the previous ones just brought into the XCD's L2 (a tile's window
    // for one channel block is ~74 KB; 32 concurrent tiles per XCD: 2.4 MB of its 4 MB); with taps outer the re-use distance is
    // a whole sweep over the channels (tools/ubench/ldsdma.hip: a 64 KiB stage takes 2550 cycles from L2, 5400 from beyond it).
    Tile tile;  // the tile whose stages are being ISSUED (one K-step ahead of the MFMAs: the next tile's during a tile's last step)
    setup(run0 + tl, tile);
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) issue_piece(tile, 0, 0, 0, j);
    int buf = 0;  // LDS stage of the current K-step (alternates along the whole stream)
    for (;;) {
        asm volatile("" : "+v"(lane));
        lane_constants();
        const int fr = lane & 15, fq = lane >> 4;
        const bool has_next = tl + per_xcd < run_n;
        const int em0 = tile.m0, en0 = tile.n0;  // the tile being multiplied (for its epilogue)
        f32x4 acc[4][MT];  // acc[nt][mt] = W_frag . A_frag^T : rows = output channel, cols = pixel
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int nx_tap = 1, nx_cc = 0;  // (tap, channel block) of K-step kt + 1
        if (nx_tap == p.taps) nx_tap = 0, nx_cc = 1;
        for (int kt = 0; kt < KT; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // everyone's current stage landed; everyone finished reading the previous one
            // The next stage of the stream (next K-step; first K-step of the next tile; at the very end the last stage again, into
            // the buffer nobody reads any more) is issued UNCONDITIONALLY, piece by piece BETWEEN this step's MFMA slots
            // (mfma_pipe.hpp): a CU's texture-address unit accepts a vector-memory wave-instruction every ~40 cycles, and an
            // in-order wave that issues its 8 back to back stands in that queue before its first MFMA.
            int is_tap = nx_tap, is_cc = nx_cc;
            if (kt + 1 == KT) {
                if (has_next) {
                    setup(run0 + tl + per_xcd, tile);  // this tile's rows are not needed any more: its last stage is in LDS
                    is_tap = 0, is_cc = 0;
                } else {
                    is_tap = last_tap, is_cc = CPT - 1;
                }
            }
            if (++nx_tap == p.taps) nx_tap = 0, ++nx_cc;
            const unsigned char *a_t = lds + buf * STAGE_BYTES, *w_t = a_t + A_GROUPS * 1024;
            if (HIVE_CONV_ABLATE & 2) {
                for (int j = 0; j < PER_WAVE; ++j) issue_piece(tile, buf ^ 1, is_tap, is_cc, j);
            } else {
                hive_mfma::kstep64<T, MT, false>(a_t, w_t, wr * RW, wc * 64, fr, fq, acc, PER_WAVE, [&](int j) { issue_piece(tile, buf ^ 1, is_tap, is_cc, j); });
            }
            buf ^= 1;
        }
        unsigned char *stage = lds + 2 * STAGE_BYTES + wave * 4096;
        const int hw = p.Ho * p.Wo, boundary = (em0 / hw + 1) * hw;  // first row of the tile's second image
        if (!(HIVE_CONV_ABLATE & 1) && !(GN == 1 && p.stats_only)) {
            if (GN == 0 && p.res2)  // (kernel-uniform) both skip connections
                conv_epilogue<T, MT, GN, 2>(p, acc, em0 + wr * RW, en0 + wc * 64, stage, lane, boundary);
            else
                conv_epilogue<T, MT, GN, 1>(p, acc, em0 + wr * RW, en0 + wc * 64, stage, lane, boundary);
        }
        if (GN == 1 && !(HIVE_CONV_ABLATE & 8)) {
            __builtin_amdgcn_wave_barrier();  // behind the epilogue's last reads of this LDS
            gn_sums_from_acc<T, MT>(p, acc, em0 + wr * RW, en0 + wc * 64, lane, boundary, reinterpret_cast<float *>(stage));
            __syncthreads();
            const int tile_m = em0 / TM;
            for (int t = tid; t < 4 * TN; t += 512) {
                const int hq = t / TN, ch = t - hq * TN, cw = ch >> 6, c = ch & 63;
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) a += reinterpret_cast<const float *>(lds + 2 * STAGE_BYTES + (w * WN + cw) * 4096)[hq * 64 + c];
                p.gn_partial[((size_t)tile_m * 4 + hq) * p.Cout + en0 + ch] = a;
            }
        }
        if (!has_next) break;
        tl += per_xcd;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant last stage
}

// The same convolution for launches that do NOT fill the chip (small batches -- the reference's literal loop is one frame per forward -- and the small
// maps of the decoder): 128 x 128 tiles on a ring of NST = 4 stages (three in flight) and, for long K loops, split-K.  With fewer tiles than CUs a
// launch lasts as long as ONE workgroup's K loop, and with two stages each of its steps waited a whole trip to L2 / HBM for the next stage (2 us
// per step for a 48 KiB stage: 72 us for a 3 x 3 convolution of 256 channels on 6 tiles).  Differences from conv_kernel: the stage being issued runs
// AHEAD = NST - 1 steps in front of the one being multiplied, along ONE stream of (item, K-step)s; the pieces are issued through
// hive_mfma::lds_dma16_untracked and counted by hand (no wait of the compiler's own may sit in the K loop: mfma_pipe.hpp);
// an item = (tile, split s) multiplies K-steps [s KT / S, (s + 1) KT / S) and the last of a tile's items to finish adds the partials in a fixed
// order and runs the epilogue (splitk_combine).  Epilogues, K order (channel block outer, tap inner) and results are conv_kernel's.
template <typename T, int TN, int GN, int NST>
__global__ __launch_bounds__(512, 1) void conv_deep_kernel(ConvParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // NST stages x (A tile 128 x 64, W tile TN x 64) + 8 x 4 KiB for the epilogue
    constexpr int TM = 128, AHEAD = NST - 1;
    constexpr int A_GROUPS = TM / 8, A_PW = A_GROUPS / 8;
    constexpr int W_GROUPS = TN / 8, GROUPS = A_GROUPS + W_GROUPS, PER_WAVE = GROUPS / 8;
    constexpr int STAGE_BYTES = GROUPS * 1024;
    constexpr int WN = TN / 64, WM = 8 / WN, RW = TM / WM, MT = RW / 16;
    static_assert(AHEAD * PER_WAVE < 64, "vmcnt is a 6-bit counter");
    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int S = p.split_k, KT = p.taps * (p.Cin / BK);
    const int tiles_n = p.Cout / TN, n_items = ((p.M + TM - 1) / TM) * tiles_n * S;
    const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3, tq = n_items >> 3, tr = n_items & 7;
    const int run0 = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, run_n = tq + (xcd < tr ? 1 : 0);
    int tl = blockIdx.x >> 3;  // position in the XCD's run of items
    if (tl >= run_n) return;   // (whole workgroup)

    struct Tile {
        int m0, n0, kt, k1, tl;  // kt: the K-step to issue next
        int py[A_PW], px[A_PW];
        const T *pbase[A_PW];
    };
    int a_chunk[A_PW];
    auto lane_constants = [&]() {
#pragma unroll
        for (int j = 0; j < A_PW; ++j) {
            const int row = (wave + 8 * j) * 8 + (lane >> 3);
            a_chunk[j] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        }
    };
    lane_constants();
    auto coords = [&](int item, int &m0, int &n0, int &k0, int &k1) {
        const int t = item / S, s = item - t * S;
        m0 = (t / tiles_n) * TM;
        n0 = (t % tiles_n) * TN;
        k0 = s * KT / S;
        k1 = (s + 1) * KT / S;
    };
    auto setup = [&](Tile &tile) {  // the rows this lane stages (the same ones in every K-step of the item)
        coords(run0 + tile.tl, tile.m0, tile.n0, tile.kt, tile.k1);
#pragma unroll
        for (int j = 0; j < A_PW; ++j) {
            const int row = (wave + 8 * j) * 8 + (lane >> 3);
            const int m = min(tile.m0 + row, p.M - 1);
            const int img = m / (p.Ho * p.Wo), rem = m - img * (p.Ho * p.Wo);
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            tile.py[j] = oy * p.stride - p.pad_t;
            tile.px[j] = ox * p.stride - p.pad_l;
            tile.pbase[j] = p.x + (((long long)img * p.H + tile.py[j]) * p.W + tile.px[j]) * p.Cin + a_chunk[j];
        }
    };
    auto issue_piece = [&](const Tile &tile, int stage, int j) {
        unsigned char *st = lds + stage * STAGE_BYTES;
        const int cc = tile.kt / p.taps, tap = tile.kt - cc * p.taps;  // channel block outer, tap inner
        if (j < A_PW) {
            const int dy = tap / p.S, dx = tap - dy * p.S;
            const long long shift = ((long long)dy * p.W + dx) * p.Cin + cc * BK;
            const bool inside = (unsigned)(tile.py[j] + dy) < (unsigned)p.H && (unsigned)(tile.px[j] + dx) < (unsigned)p.W;
            const T *g = inside ? tile.pbase[j] + shift : p.zeros + a_chunk[j];
            hive_mfma::lds_dma16_untracked((const void *)g, __builtin_amdgcn_readfirstlane(hive_mfma::lds_address(st + (wave + 8 * j) * 1024)));
        } else {
            const int grp = wave + 8 * (j - A_PW);
            const int row = grp * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            const T *g = p.w + (size_t)(tile.n0 + row) * (p.taps * p.Cin) + tap * p.Cin + cc * BK + chunk * 8;
            hive_mfma::lds_dma16_untracked((const void *)g, __builtin_amdgcn_readfirstlane(hive_mfma::lds_address(st + (A_GROUPS + grp) * 1024)));
        }
    };
    // past the end of the stream the cursor parks on the last stage, issued again into a buffer nobody reads any more: every step issues a stage
    auto advance = [&](Tile &tile) {
        if (tile.kt + 1 < tile.k1) {
            ++tile.kt;
        } else if (tile.tl + per_xcd < run_n) {
            tile.tl += per_xcd;
            setup(tile);
        }
    };
    Tile is;
    is.tl = tl;
    setup(is);
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) {
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) issue_piece(is, a, j);
        advance(is);
    }
    int buf = 0;
    for (;;) {
        asm volatile("" : "+v"(lane));
        lane_constants();
        const int fr = lane & 15, fq = lane >> 4;
        const bool has_next = tl + per_xcd < run_n;
        int em0, en0, k0, k1;
        coords(run0 + tl, em0, en0, k0, k1);
        f32x4 acc[4][MT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = k0; kt < k1; ++kt) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PER_WAVE) : "memory");  // this step's stage: all but the AHEAD - 1 younger ones have landed
            __builtin_amdgcn_s_barrier();
            const int ib = buf + AHEAD >= NST ? buf + AHEAD - NST : buf + AHEAD;  // the buffer the previous step read
            const unsigned char *a_t = lds + buf * STAGE_BYTES, *w_t = a_t + A_GROUPS * 1024;
            hive_mfma::kstep64<T, MT, false>(a_t, w_t, wr * RW, wc * 64, fr, fq, acc, PER_WAVE, [&](int j) { issue_piece(is, ib, j); });
            advance(is);
            buf = buf + 1 == NST ? 0 : buf + 1;
        }
        unsigned char *epi = lds + NST * STAGE_BYTES;
        bool store = true;
        if (S > 1) store = hive_mfma::splitk_combine<512>(S, p.sk_ws, p.sk_count, run0 + tl, acc, tid, reinterpret_cast<int *>(epi));
        if (store) {
            unsigned char *stage = epi + wave * 4096;
            const int hw = p.Ho * p.Wo, boundary = (em0 / hw + 1) * hw;
            if (!(GN == 1 && p.stats_only)) {
                if (GN == 0 && p.res2)
                    conv_epilogue<T, MT, GN, 2>(p, acc, em0 + wr * RW, en0 + wc * 64, stage, lane, boundary);
                else
                    conv_epilogue<T, MT, GN, 1>(p, acc, em0 + wr * RW, en0 + wc * 64, stage, lane, boundary);
            }
            if (GN == 1) {
                __builtin_amdgcn_wave_barrier();
                gn_sums_from_acc<T, MT>(p, acc, em0 + wr * RW, en0 + wc * 64, lane, boundary, reinterpret_cast<float *>(stage));
                __syncthreads();
                const int tile_m = em0 / TM;
                for (int t = tid; t < 4 * TN; t += 512) {
                    const int hq = t / TN, ch = t - hq * TN, cw = ch >> 6, c = ch & 63;
                    float a = 0.f;
#pragma unroll
                    for (int w = 0; w < WM; ++w) a += reinterpret_cast<const float *>(epi + (w * WN + cw) * 4096)[hq * 64 + c];
                    p.gn_partial[((size_t)tile_m * 4 + hq) * p.Cout + en0 + ch] = a;
                }
            }
        }
        if (!has_next) break;
        // nothing but the ring's own pieces may be outstanding in the K loop, in fact (stores and loads retire in no common order) and in the
        // compiler's books (a shortcut row loaded ahead for a row past M is never consumed: hipcc would wait for it at the top of every K-step)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): a wait the compiler sees
        __syncthreads();                     // (the GroupNorm sums' reads of the epilogue LDS stay in front of the next item's)
        tl += per_xcd;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the parked cursor's stages
}

constexpr int conv_lds(int tm, int tn, int nst = 2) { return nst * (tm / 8 + tn / 8) * 1024 + hive_mfma::STAGED_ROWS_LDS; }  // two stages of (A tile + W tile), 128-byte rows; the epilogue's 8 x 4 KiB

constexpr int CONV_DEEP_NST = 4;  // 4 x 32 KiB of stages + 32 KiB for the epilogue = the CU's 160 KiB

template <typename T>
int ensure_conv_attrs(hive_ctx *ctx) {
    static bool set[64] = {};
    if (ctx->device < 64 && set[ctx->device]) return HIVE_OK;
#define HIVE_CONV_ATTR(TM_, TN_, GN_) \
    HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)conv_kernel<T, TM_, TN_, GN_>, hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds(TM_, TN_)))
    HIVE_CONV_ATTR(256, 256, 0);
    HIVE_CONV_ATTR(256, 256, 1);
    HIVE_CONV_ATTR(256, 128, 0);
    HIVE_CONV_ATTR(256, 128, 1);
    HIVE_CONV_ATTR(256, 64, 0);
    HIVE_CONV_ATTR(256, 64, 1);
    HIVE_CONV_ATTR(128, 256, 0);
    HIVE_CONV_ATTR(128, 256, 1);
    HIVE_CONV_ATTR(128, 128, 0);
    HIVE_CONV_ATTR(128, 128, 1);
    HIVE_CONV_ATTR(256, 256, 2);
    HIVE_CONV_ATTR(128, 256, 2);
#undef HIVE_CONV_ATTR
    HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)conv_deep_kernel<T, 128, 0, CONV_DEEP_NST>, hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds(128, 128, CONV_DEEP_NST)));
    HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)conv_deep_kernel<T, 128, 1, CONV_DEEP_NST>, hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds(128, 128, CONV_DEEP_NST)));
    if (ctx->device < 64) set[ctx->device] = true;
    return HIVE_OK;
}

// the two-pass GroupNorm convolution's mode (hive_nhwc_conv_gn_apply): statistics only / normalise in the epilogue
struct GnMode {
    int stats_only = 0;
    const float *gn_stats = nullptr;
    const void *gn_gamma = nullptr, *gn_beta = nullptr;
    int gn_G = 0, gn_cpg = 0;
};

template <typename T>
int launch_conv_t(hive_ctx *ctx, const char *what, const void *d_x, int N, int H, int W, int C_in, int C_out, int R, int stride, int pad_t,
                  int pad_l, int Ho, int Wo, const void *d_w, const void *d_bias, int relu, const void *d_residual, const void *d_residual2,
                  void *d_out, void *d_out_relu, void *d_gn_partial, long long gn_partial_floats, int *gn_tile_rows, const GnMode *gn_mode) {
    HIVE_REQUIRE(ctx, d_x && d_w && d_out, "%s: NULL argument", what);
    if (gn_tile_rows) *gn_tile_rows = 0;
    HIVE_REQUIRE(ctx, N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && (long long)N * H * W < (1ll << 31) && (long long)N * Ho * Wo < (1ll << 31),
                 "%s: bad sizes %d x %d x %d -> %d x %d", what, N, H, W, Ho, Wo);
    HIVE_REQUIRE(ctx, (R == 1 || R == 3) && (stride == 1 || stride == 2) && pad_t >= 0 && pad_t < R && pad_l >= 0 && pad_l < R,
                 "%s: kernel %d, stride %d, padding (%d, %d)", what, R, stride, pad_t, pad_l);
    HIVE_REQUIRE(ctx, (long long)(Ho - 1) * stride - pad_t < H && (long long)(Wo - 1) * stride - pad_l < W, "%s: output %d x %d reaches outside the %d x %d input", what, Ho,
                 Wo, H, W);
    HIVE_REQUIRE(ctx, C_in > 0 && C_in % 64 == 0 && C_out > 0 && C_out % 64 == 0, "%s: need C_in %% 64 == 0 and C_out %% 64 == 0, got %d -> %d", what, C_in, C_out);
    HIVE_REQUIRE(ctx, (R == 1 && stride == 1) || (d_out != d_x && d_out_relu != d_x), "%s: the output must not alias the input", what);
    ConvParams<T> p{};
    p.x = (const T *)d_x;
    p.w = (const T *)d_w;
    p.bias = (const T *)d_bias;
    p.res1 = (const T *)d_residual;
    p.res2 = (const T *)d_residual2;
    p.out = (T *)d_out;
    p.out_relu = (T *)d_out_relu;
    p.zeros = (const T *)ctx->d_zeros;
    p.H = H;
    p.W = W;
    p.Cin = C_in;
    p.Cout = C_out;
    p.relu = relu;
    p.Ho = Ho;
    p.Wo = Wo;
    p.S = R;
    p.taps = R * R;
    p.stride = stride;
    p.pad_t = pad_t;
    p.pad_l = pad_l;
    p.M = N * Ho * Wo;
    int rc = ensure_conv_attrs<T>(ctx);
    if (rc) return rc;
    const int tn = C_out % 256 == 0 ? 256 : (C_out % 128 == 0 ? 128 : 64);
    // 256 output pixels per tile, or 128 where that leaves fewer CU-rounds of work: one persistent workgroup per CU, so a launch costs
    // rounds x tile size (a 128-row tile does ~0.85 of a 256-row tile's rate).  Large batches: thousands of tiles, 256 wins.  Small maps and
    // small batches (30 x 40 / 15 x 20 maps: 113 / 29 tiles of 256; one 240 x 320 frame: 300 tiles = 2 rounds of 256 rows against 3 of 128).
    int tm = 256;
    if (tn >= 128) {
        const long long per = C_out / tn, t256 = (long long)((p.M + 255) / 256) * per, t128 = (long long)((p.M + 127) / 128) * per;
        const long long r256 = (t256 + ctx->num_cus - 1) / ctx->num_cus, r128 = (t128 + ctx->num_cus - 1) / ctx->num_cus;
        if (t256 < ctx->num_cus || r128 * 128 * 100 < r256 * 256 * 85) tm = 128;
    }
    if (gn_mode) {
        p.stats_only = gn_mode->stats_only;
        p.gn_stats = gn_mode->gn_stats;
        p.gn_gamma = (const T *)gn_mode->gn_gamma;
        p.gn_beta = (const T *)gn_mode->gn_beta;
        p.gn_G = gn_mode->gn_G;
        p.gn_cpg = gn_mode->gn_cpg;
    }
    // GroupNorm statistics from the epilogue: a tile may touch two images at most (tm <= Ho Wo); smaller maps keep the stand-alone pass
    // (and plain epilogues: with a shortcut / ReLU in the epilogue no statistics are left and *gn_tile_rows stays 0)
    if (d_gn_partial && gn_tile_rows && (long long)Ho * Wo >= tm && !relu && !d_residual && !d_residual2 && !d_out_relu) {
        HIVE_REQUIRE(ctx, (long long)((p.M + tm - 1) / tm) * 4 * C_out <= gn_partial_floats, "%s: gn_partial holds %lld floats, %lld needed", what,
                     gn_partial_floats, (long long)((p.M + tm - 1) / tm) * 4 * C_out);
        p.gn_partial = (float *)d_gn_partial;
        *gn_tile_rows = tm;
    }
    // Launches that do not fill the chip: conv_deep_kernel (128 x 128 tiles, four-stage ring, split-K for long K loops).  HIVE_CONV_DEEP=0 switches it off.
    {
        const long long tiles128 = (long long)((p.M + 127) / 128) * (C_out / 128);
        const char *deep_env = getenv("HIVE_CONV_DEEP");
        const bool allowed = !(deep_env && deep_env[0] == '0');
        if (allowed && !p.gn_stats && C_out % 128 == 0 && tiles128 <= ctx->num_cus && tiles128 <= HIVE_SPLITK_TILES) {
            const int KT = p.taps * (C_in / BK);
            const char *sk_env = getenv("HIVE_SPLITK");
            p.split_k = sk_env ? std::max(1, std::min(atoi(sk_env), KT)) : hive_mfma::splitk_ways(tiles128, KT, ctx->num_cus);
            p.split_k = (int)std::max<long long>(1, std::min<long long>(p.split_k, ctx->num_cus / tiles128));
            if (ctx->deterministic) p.split_k = 1;
            ++ctx->n_deep_ring_launches;
            if (p.split_k > 1) {
                ++ctx->n_splitk_launches;
                void *ws = nullptr;
                rc = hive_splitk_workspace(ctx, (size_t)tiles128 * p.split_k * 128 * 128 * sizeof(float), &ws, &p.sk_count);
                if (rc) return rc;
                p.sk_ws = reinterpret_cast<hive_mfma::f32x4 *>(ws);
            }
            const bool stats = d_gn_partial && gn_tile_rows && (long long)Ho * Wo >= 128 && !relu && !d_residual && !d_residual2 && !d_out_relu;
            if (stats) {
                HIVE_REQUIRE(ctx, (long long)((p.M + 127) / 128) * 4 * C_out <= gn_partial_floats, "%s: gn_partial holds %lld floats, %lld needed", what,
                             gn_partial_floats, (long long)((p.M + 127) / 128) * 4 * C_out);
                p.gn_partial = (float *)d_gn_partial;
                *gn_tile_rows = 128;
            }
            const long long items = tiles128 * p.split_k;
            const dim3 dgrid((unsigned)std::min<long long>((items + 7) / 8 * 8, (long long)ctx->num_cus / 8 * 8));
            if (stats)
                hipLaunchKernelGGL((conv_deep_kernel<T, 128, 1, CONV_DEEP_NST>), dgrid, dim3(512), (size_t)conv_lds(128, 128, CONV_DEEP_NST), ctx->stream, p);
            else
                hipLaunchKernelGGL((conv_deep_kernel<T, 128, 0, CONV_DEEP_NST>), dgrid, dim3(512), (size_t)conv_lds(128, 128, CONV_DEEP_NST), ctx->stream, p);
            HIVE_CHECK_HIP(ctx, hipGetLastError());
            return HIVE_OK;
        }
    }
    // persistent workgroups, one per CU, a multiple of 8 so that every XCD gets the same number
    const long long tiles = (long long)((p.M + tm - 1) / tm) * (C_out / tn);
    const dim3 grid((unsigned)std::min<long long>((tiles + 7) / 8 * 8, (long long)ctx->num_cus / 8 * 8));
    const size_t lds = (size_t)conv_lds(tm, tn);
#define HIVE_CONV_LAUNCH(TM_, TN_)                                                                                   \
    do {                                                                                                             \
        if (p.gn_partial)                                                                                            \
            hipLaunchKernelGGL((conv_kernel<T, TM_, TN_, 1>), grid, dim3(512), lds, ctx->stream, p);                    \
        else                                                                                                         \
            hipLaunchKernelGGL((conv_kernel<T, TM_, TN_, 0>), grid, dim3(512), lds, ctx->stream, p);                    \
    } while (0)
    if (p.gn_stats) {  // second pass of hive_nhwc_conv_gn_apply: C_out % 256 == 0 (checked there)
        if (tm == 256)
            hipLaunchKernelGGL((conv_kernel<T, 256, 256, 2>), grid, dim3(512), lds, ctx->stream, p);
        else
            hipLaunchKernelGGL((conv_kernel<T, 128, 256, 2>), grid, dim3(512), lds, ctx->stream, p);
    } else if (tm == 256 && tn == 256)
        HIVE_CONV_LAUNCH(256, 256);
    else if (tm == 256 && tn == 128)
        HIVE_CONV_LAUNCH(256, 128);
    else if (tm == 256)
        HIVE_CONV_LAUNCH(256, 64);
    else if (tn == 256)
        HIVE_CONV_LAUNCH(128, 256);
    else
        HIVE_CONV_LAUNCH(128, 128);
#undef HIVE_CONV_LAUNCH
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

// dtype dispatch: HIVE_BF16 (north_star's contract) or HIVE_F16 (the reference's `model.half()`)
int launch_conv(hive_ctx *ctx, const char *what, const void *d_x, int dtype, int N, int H, int W, int C_in, int C_out, int R, int stride, int pad_t,
                int pad_l, int Ho, int Wo, const void *d_w, const void *d_bias, int relu, const void *d_residual, const void *d_residual2,
                void *d_out, void *d_out_relu, void *d_gn_partial = nullptr, long long gn_partial_floats = 0, int *gn_tile_rows = nullptr,
                const GnMode *gn_mode = nullptr) {
    if (dtype == HIVE_BF16)
        return launch_conv_t<__bf16>(ctx, what, d_x, N, H, W, C_in, C_out, R, stride, pad_t, pad_l, Ho, Wo, d_w, d_bias, relu, d_residual, d_residual2, d_out, d_out_relu,
                                     d_gn_partial, gn_partial_floats, gn_tile_rows, gn_mode);
    if (dtype == HIVE_F16)
        return launch_conv_t<_Float16>(ctx, what, d_x, N, H, W, C_in, C_out, R, stride, pad_t, pad_l, Ho, Wo, d_w, d_bias, relu, d_residual, d_residual2, d_out,
                                       d_out_relu, d_gn_partial, gn_partial_floats, gn_tile_rows, gn_mode);
    if (gn_tile_rows) *gn_tile_rows = 0;
    return hive_fail(ctx, HIVE_ERR_INVALID, "%s: dtype must be HIVE_F16 or HIVE_BF16", what);
}

}  // namespace

extern "C" int hive_nhwc_conv3x3(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C_in, int C_out, const void *d_w,
                                 const void *d_bias, int relu, const void *d_residual, const void *d_residual2, void *d_out,
                                 void *d_out_relu) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, C_out % 128 == 0, "nhwc_conv3x3: need C_out %% 128 == 0, got %d", C_out);
    HIVE_REQUIRE(ctx, d_out != d_x && d_out_relu != d_x, "nhwc_conv3x3: the output must not alias the input (3 x 3 halo)");
    return launch_conv(ctx, "nhwc_conv3x3", d_x, dtype, N, H, W, C_in, C_out, 3, 1, 1, 1, H, W, d_w, d_bias, relu, d_residual, d_residual2, d_out, d_out_relu);
}

extern "C" int64_t hive_nhwc_conv_gn_partial_floats(int64_t n_px, int C_out) { return (n_px / 128 + 1) * 4 * (int64_t)C_out; }

extern "C" int hive_nhwc_conv_gn(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C_in, int C_out, int kernel, int stride, int pad_top,
                                 int pad_left, int H_out, int W_out, const void *d_w, const void *d_bias, int relu, const void *d_residual,
                                 const void *d_residual2, void *d_out, void *d_out_relu, void *d_gn_partial, int64_t gn_partial_floats, int *gn_tile_rows) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_gn_partial && gn_tile_rows, "nhwc_conv_gn: NULL argument");
    return launch_conv(ctx, "nhwc_conv_gn", d_x, dtype, N, H, W, C_in, C_out, kernel, stride, pad_top, pad_left, H_out, W_out, d_w, d_bias, relu, d_residual,
                       d_residual2, d_out, d_out_relu, d_gn_partial, gn_partial_floats, gn_tile_rows);
}

// out = relu?(GroupNorm(conv(x)) + residual) WITHOUT the convolution's output ever reaching memory: the convolution runs twice.  Pass 1
// keeps only the per-tile channel sums of its (rounded) outputs; (mean, rstd) per (sample, group) follow; pass 2 recomputes the tile
// and normalises it in the epilogue.  Worth it where the output is wider than the input -- the ResNetV2 bottlenecks' expanding 1 x 1
// convolutions (conv3: C -> 4 C, and the stage's downsample convolution): per element of the 4 C-wide tensor the separate sequence
// moves write + read (statistics, now folded) + read + write, this one a single write, at the price of reading the C-wide input twice.
extern "C" int hive_nhwc_conv_gn_apply(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C_in, int C_out, int kernel, int stride,
                                       int pad_top, int pad_left, int H_out, int W_out, const void *d_w, int G, const void *d_gamma, const void *d_beta,
                                       float eps, const void *d_residual, int relu, void *d_out, void *d_scratch, int64_t scratch_floats, int *fused) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_gamma && d_beta && d_scratch && fused, "nhwc_conv_gn_apply: NULL argument");
    *fused = 0;
    const long long hw = (long long)H_out * W_out;
    // eligible: whole 256-channel tiles, a lane's 8 channels inside one group, a tile inside two samples
    if (G <= 0 || C_out % 256 != 0 || C_out % G != 0 || (C_out / G) % 8 != 0 || hw < 256) return HIVE_OK;
    const int64_t partial_floats = hive_nhwc_conv_gn_partial_floats((int64_t)N * hw, C_out);
    HIVE_REQUIRE(ctx, scratch_floats >= partial_floats + 2ll * N * G, "nhwc_conv_gn_apply: scratch holds %lld floats, %lld needed", (long long)scratch_floats,
                 (long long)(partial_floats + 2ll * N * G));
    float *partial = (float *)d_scratch, *stats = partial + partial_floats;
    GnMode mode;
    mode.stats_only = 1;
    int tile_rows = 0;
    int rc = launch_conv(ctx, "nhwc_conv_gn_apply", d_x, dtype, N, H, W, C_in, C_out, kernel, stride, pad_top, pad_left, H_out, W_out, d_w, nullptr, 0, nullptr,
                         nullptr, d_out, nullptr, partial, partial_floats, &tile_rows, &mode);
    if (rc) return rc;
    HIVE_REQUIRE(ctx, tile_rows > 0, "nhwc_conv_gn_apply: no statistics from a %lld-pixel map", hw);
    rc = hive_gn_finalize_tiles(ctx, partial, N, (int)hw, C_out, G, tile_rows, eps, stats);
    if (rc) return rc;
    mode.stats_only = 0;
    mode.gn_stats = stats;
    mode.gn_gamma = d_gamma;
    mode.gn_beta = d_beta;
    mode.gn_G = G;
    mode.gn_cpg = C_out / G;
    rc = launch_conv(ctx, "nhwc_conv_gn_apply", d_x, dtype, N, H, W, C_in, C_out, kernel, stride, pad_top, pad_left, H_out, W_out, d_w, nullptr, relu, d_residual,
                     nullptr, d_out, nullptr, nullptr, 0, nullptr, &mode);
    if (rc) return rc;
    *fused = 1;
    return HIVE_OK;
}

// The same operation for 1 x 1 convolutions with the statistics from the input's Gram matrices (gram.hip) instead of a first pass of the convolution:
// d_tables from hive_gn_gram_prepare(d_w).  d_scratch: float, >= 2 N G elements.  *fused = 0: not eligible (as above, or C_in not 64 / 128 / 256).
extern "C" int hive_nhwc_conv_gn_apply_gram(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C_in, int C_out, int stride, int H_out, int W_out,
                                            const void *d_w, const float *d_tables, int G, const void *d_gamma, const void *d_beta, float eps, const void *d_residual,
                                            int relu, void *d_out, void *d_scratch, int64_t scratch_floats, int *fused) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_gamma && d_beta && d_scratch && d_tables && fused, "nhwc_conv_gn_apply_gram: NULL argument");
    *fused = 0;
    const long long hw = (long long)H_out * W_out;
    if (G < 4 || 256 % G != 0 || G > 64 || C_out % 256 != 0 || C_out % G != 0 || (C_out / G) % 8 != 0 || hw < 256 || !(C_in == 64 || C_in == 128 || C_in == 256)) return HIVE_OK;
    HIVE_REQUIRE(ctx, scratch_floats >= 2ll * N * G, "nhwc_conv_gn_apply_gram: scratch holds %lld floats, %lld needed", (long long)scratch_floats, 2ll * N * G);
    HIVE_REQUIRE(ctx, (long long)(H_out - 1) * stride < H && (long long)(W_out - 1) * stride < W, "nhwc_conv_gn_apply_gram: output %d x %d reaches outside the %d x %d input", H_out,
                 W_out, H, W);
    float *stats = (float *)d_scratch;
    int rc = hive_gram_gn_stats(ctx, d_x, dtype, N, H, W, C_in, C_out, stride, H_out, W_out, G, d_tables, eps, stats, nullptr, nullptr);
    if (rc) return rc;
    GnMode mode;
    mode.gn_stats = stats;
    mode.gn_gamma = d_gamma;
    mode.gn_beta = d_beta;
    mode.gn_G = G;
    mode.gn_cpg = C_out / G;
    rc = launch_conv(ctx, "nhwc_conv_gn_apply_gram", d_x, dtype, N, H, W, C_in, C_out, 1, stride, 0, 0, H_out, W_out, d_w, nullptr, relu, d_residual, nullptr, d_out, nullptr,
                     nullptr, 0, nullptr, &mode);
    if (rc) return rc;
    *fused = 1;
    return HIVE_OK;
}

extern "C" int hive_nhwc_conv(hive_ctx *ctx, const void *d_x, int dtype, int N, int H, int W, int C_in, int C_out, int kernel, int stride, int pad_top,
                              int pad_left, int H_out, int W_out, const void *d_w, const void *d_bias, int relu, const void *d_residual,
                              const void *d_residual2, void *d_out, void *d_out_relu) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    return launch_conv(ctx, "nhwc_conv", d_x, dtype, N, H, W, C_in, C_out, kernel, stride, pad_top, pad_left, H_out, W_out, d_w, d_bias, relu, d_residual,
                       d_residual2, d_out, d_out_relu);
}
