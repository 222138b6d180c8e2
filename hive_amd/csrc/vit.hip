// ViT-B encoder blocks of DPT-Hybrid on gfx950: LayerNorm, bf16 MFMA GEMM with fused epilogues, and a
// flash-style attention with LDS-tiled QK^T.  No vendor BLAS.
//
// Replaces the transformer blocks inside `dpt.models.DPTDepthModel.forward` of the reference's (absent)
// third_party/dpt + timm==0.5.4 (call site /root/reference/hive/dataset_adaptors.py:1419):
//     x = x + proj(softmax(q k^T / sqrt(64)) v),  q,k,v = qkv(LN1(x));   x = x + fc2(gelu(fc1(LN2(x))))
//
// Data layout (device, bf16 unless noted), sized for HBM residency of a whole batch:
//   tokens      x      [B][Np][D]      Np = tokens per image padded to a multiple of 64 (pad rows are zero)
//   q | k       qk     [B*Np][2D]      head h, channel c at column h*64 + c  (k at D + h*64 + c)
//   v^T         vT     [B][H][64][Np]  (token quads 4..7 and 8..11 of every 16 swapped: vt_slot) written transposed by the QKV GEMM epilogue, so that P.V contracts
//                                      over a contiguous axis and the attention needs no transposed reads
//   weights            [out][in] row-major (nn.Linear), bias / LayerNorm affine in f32
//
// GEMM: C[M][N] = A[M][K] W[N][K]^T on v_mfma_f32_16x16x32_bf16, K-step 64.  Two tilings: 128 x 128 (4 waves of 64 x 64,
// 2-stage ring of 64 KiB so that two workgroups share a CU) and, for wide N with plenty of tiles, 256 x 256 (8 waves of
// 128 x 64).  Operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4); the bank swizzle is applied on the
// SOURCE address and again on the ds_read_b128 (conflict-free); one raw s_barrier per K-step; XCD-aware tile order.
// The MFMA is issued as W.A^T so that a lane owns 4 consecutive output columns (8-byte stores); the v^T
// kernel variant issues A.W^T so that a lane owns 4 consecutive tokens.  Epilogues: bias, erf-GELU, residual.
//
// Attention: one workgroup = 128 queries of one (image, head); each wave 32 queries; K / V^T tiles of 64 keys staged by
// LDS-DMA into a 2-stage ring.  S^T = K Q^T puts the
// query on the lane and the keys in the accumulator registers, so the online softmax (base 2, deferred maximum)
// is in-register with one cross-half shuffle, and the probabilities are already the B operand of O^T += V^T P^T.
#include "hive_internal.hpp"
#include "mfma_pipe.hpp"

#include <algorithm>
#include <cstdlib>
#include <cmath>

using hive_mfma::f32x4;
using hive_mfma::f32x16;
using hive_mfma::vec;  // vec<T, 8>: 8 elements of the 16-bit element type T = __bf16 (north_star's contract) or _Float16 (what the
                       // reference runs, /root/reference/hive/dataset_adaptors.py:1394-1401, 1415-1417): same MFMA shapes and rates

#ifndef HIVE_GEMM_AHEAD
#define HIVE_GEMM_AHEAD 3
#endif
enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESIDUAL = 2, EPI_QKV = 3, EPI_QKV_ALL = 4 };  // (_ALL: gemm_kernel only -- q | k and v^T tiles in ONE launch, chosen per tile)

// v^T key order.  The attention's P.V step holds the probabilities of a lane in accumulator order: its 8 k-slots of one
// 32x32x16 MFMA are keys {0..3, 8..11} (+ 4 for the upper half-wave) of a 16-key group -- the S^T accumulator layout, not a
// choice.  V^T is therefore STORED with the two middle quads of every 16-token group swapped (slot = token with bits 2 and 3
// exchanged; an involution on multiples of 4), so that those 8 keys are 16 contiguous bytes: one conflict-free ds_read_b128
// per operand instead of two 8-byte reads that collide under the 16-byte chunk swizzle (39 % LDS bank-conflict rate in
// round 1's counters).
__device__ __forceinline__ int vt_slot(int tok) { return (tok & ~12) | ((tok & 4) << 1) | ((tok & 8) >> 1); }

using hive_mfma::swz;  // byte offset of 16-byte chunk c of row r in a 128-byte-row tile, chunk XORed with (r >> 1) & 7

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dimension D (multiple of 256), one wave per row, f32 statistics
template <typename T, int CH>  // D = 256 * CH
__global__ __launch_bounds__(256) void layernorm_kernel(const T *__restrict__ x, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, T *__restrict__ out, int M, float eps) {
    constexpr int D = 256 * CH;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const T *xr = x + (size_t)row * D;
    float v[4 * CH];
    constexpr int n_chunks = CH;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < n_chunks; ++i) {
        const vec<T, 4> t = *reinterpret_cast<const vec<T, 4> *>(xr + i * 256 + lane * 4);
        for (int j = 0; j < 4; ++j) {
            v[4 * i + j] = (float)t[j];
            sum += v[4 * i + j];
        }
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)D;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < 4 * n_chunks; ++i) {
        const float d = v[i] - mean;
        var += d * d;
    }
    for (int off = 32; off > 0; off >>= 1) var += __shfl_xor(var, off);
    const float rstd = rsqrtf(var / (float)D + eps);
    T *orow = out + (size_t)row * D;
#pragma unroll
    for (int i = 0; i < n_chunks; ++i) {
        const int c = i * 256 + lane * 4;
        const float4 g = *reinterpret_cast<const float4 *>(gamma + c);
        const float4 b = *reinterpret_cast<const float4 *>(beta + c);
        vec<T, 4> o;
        o[0] = (T)((v[4 * i + 0] - mean) * rstd * g.x + b.x);
        o[1] = (T)((v[4 * i + 1] - mean) * rstd * g.y + b.y);
        o[2] = (T)((v[4 * i + 2] - mean) * rstd * g.z + b.z);
        o[3] = (T)((v[4 * i + 3] - mean) * rstd * g.w + b.w);
        *reinterpret_cast<vec<T, 4> *>(orow + c) = o;
    }
}

// ------------------------------------------------------------------------------------------------
template <typename T>
struct GemmParams {
    const T *A;       // [M][K]
    const T *W;       // [N][K]
    const float *bias;   // [N]
    const T *residual;  // [M][ldc] (EPI_BIAS_RESIDUAL)
    T *C;             // [M][ldc]
    T *vT;            // EPI_QKV: [B][H][64][Np]
    int M, N, K, ldc;
    int Np, H;           // EPI_QKV: tokens per image (padded), heads
    int n_split;         // EPI_QKV: columns >= n_split are V columns
    int q_cols;          // EPI_BIAS: columns < q_cols (a multiple of 8) are multiplied by q_scale after the bias (the q part of q|k)
    float q_scale;
    // LayerNorm folded into the GEMM that consumes it (hive_vit_forward): with W' = gamma o W stored as the weight, c1[n] = sum_k W'[n][k] and
    // bias[n] = sum_k beta[k] W[n][k] + b[n], LayerNorm(x) W^T + b = rstd (x W'^T - mean c1) + bias -- the normalised tensor is never written.
    const float *ln_stats;  // consumer: [M][2] = (mean, rstd) of the rows of A, or null: the plain epilogue
    const float *ln_c1;     // consumer: [N]
    float *ln_partial;      // producer (EPI_BIAS_RESIDUAL), or null: [M][N / 64][2] = per 64 stored columns of a row (their sum, the sum of squares about their own mean)
    // split-K (gemm_kernel only, small M: mfma_pipe.hpp splitk_combine): the K-steps of a tile are dealt to split_k workgroups
    int split_k = 1;
    hive_mfma::f32x4 *sk_ws = nullptr;
    unsigned *sk_count = nullptr;
};

constexpr int BN = 128, BK = 64;

// (attention_kernel, tuning builds of rounds 4-5 that lost and were taken out: the softmax denominators from an extra all-ones channel block of P.V instead of 32 v_add_f32 per
// tile -- 75.41 -> 75.85 ms per 107-frame forward, the four extra MFMAs cost more than the additions; the same sums as 16 v_pk_add_f32 -- 75.78-75.99 -> 75.96-76.19 ms.)
#ifndef HIVE_GEMM_ABLATE
#define HIVE_GEMM_ABLATE 0  // tuning builds only (make ablate_gemm; tools/probe_gemm_tiles.py with HIVE_AMD_LIB=...): 1 = gemm256p_kernel without its epilogue
#endif

// erf-GELU (nn.GELU default): 0.5 x (1 + erf(x / sqrt 2)) = max(x, 0) - |x| E(|x|) / 2 with E(a) = erfc(a / sqrt 2).
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef HIVE_GELU_AS
#define HIVE_GELU_AS 0  // tuning build (make gelu_as): 1 = rounds 1-4's form, erf by Abramowitz-Stegun 7.1.26 on rcp + exp2 (two transcendentals, ~16 issue slots per value)
#endif
#if HIVE_GELU_AS
__device__ __forceinline__ f32x2 gelu_exact2(f32x2 x) {
    const f32x2 ax = f32x2{fabsf(x.x), fabsf(x.y)};
    const f32x2 z = ax * 0.70710678118654752440f;
    const f32x2 d = 1.0f + 0.3275911f * z;
    const f32x2 t = f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2 poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const f32x2 a = -z * z * 1.44269504088896340736f;
    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    const f32x2 erf_abs = 1.0f - poly * e;
    const f32x2 s = f32x2{copysignf(erf_abs.x, x.x), copysignf(erf_abs.y, x.y)};
    return 0.5f * x * (1.0f + s);
}
#else
// Round 5: ONE transcendental per value.  E(a) / 2 = 2^-(1 + a q(a)) with q a degree-4 polynomial fitted (weighted minimax of the GELU's absolute error,
// tools/fit_gelu.py) so that |gelu - exact| <= 6e-7 for every float32 x -- the size of Abramowitz-Stegun's 1.5e-7 |x| / 2 at |x| = 4-8, three orders of
// magnitude below a bfloat16 / two below a float16 output step -- and 1 + a q(a) increases monotonically to +inf, so large |x| need no clamp (2^-inf = 0).
// Evaluated for two values at once on the packed f32 pipe (v_pk_fma_f32: IEEE per element): 5 packed + 2 v_and + 2 v_exp per pair, ~9.5 issue slots per value
// against ~16.5 -- the fc1 epilogue was as long as its tile's 12-step K loop, 6,600 of its 13,800 cycles the GELU arithmetic.
__device__ __forceinline__ f32x2 gelu_exact2(f32x2 x) {
    const f32x2 ax = f32x2{fabsf(x.x), fabsf(x.y)};
    f32x2 q = 4.88117491e-04f * ax - 7.19880622e-03f;
    q = q * ax + 5.21468023e-02f;
    q = q * ax + 4.59595714e-01f;
    q = q * ax + 1.15100057e+00f;
    const f32x2 P = ax * q + 1.0f;
    const f32x2 e = f32x2{__builtin_amdgcn_exp2f(-P.x), __builtin_amdgcn_exp2f(-P.y)};
    return (x + ax) * 0.5f - ax * e;  // max(x, 0) = (x + |x|) / 2 exactly
}
#endif


// Global -> LDS staging with LDS-DMA (global_load_lds_dwordx4): one wave instruction deposits 64 x 16 B =
// 8 rows of 128 B, lane-linear.  The bank swizzle therefore lives on the SOURCE side: LDS slot (row, s)
// receives global chunk s ^ ((row >> 1) & 7), and the fragment reads apply the same XOR (swz()).
template <typename T, bool UNTRACKED = false>
__device__ __forceinline__ void stage_group(const T *__restrict__ src, int ld, int row0, int row_max, int k0,
                                            unsigned char *tile, int grp, int lane) {
    const int row = grp * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    const int grow = min(row0 + row, row_max);  // clamp: rows past the end are never stored
    const T *g = src + (size_t)grow * ld + k0 + chunk * 8;
    if constexpr (UNTRACKED)
        hive_mfma::lds_dma16_untracked((const void *)g, __builtin_amdgcn_readfirstlane(hive_mfma::lds_address(tile + grp * 1024)));
    else
        __builtin_amdgcn_global_load_lds((const void *)g, (__attribute__((address_space(3))) void *)(tile + grp * 1024), 16, 0, 0);
}

// Sum over the 8 lanes 8 g .. 8 g + 7 (half of a 16-lane DPP row), left in all of them: two quad permutes and a half-row mirror, three DPP
// additions on the VALU (the ds_bpermute behind __shfl_xor goes through the LDS pipe, which the epilogue's turn-around already keeps busy)
__device__ __forceinline__ float dpp_oct_total(float v) {
#define HIVE_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    HIVE_DPP_ADD(0xB1);   // quad_perm [1, 0, 3, 2]
    HIVE_DPP_ADD(0x4E);   // quad_perm [2, 3, 0, 1]
    HIVE_DPP_ADD(0x141);  // row_half_mirror: lane i of an 8-lane half reads lane 7 - i, which holds the other quad's sum
#undef HIVE_DPP_ADD
    return v;
}

// Epilogue of the A.W^T GEMM tiles (EPI_BIAS / _GELU / _RESIDUAL): bias, GELU or residual in f32, one rounding to bf16; through the
// wave's 4 KiB of LDS (mfma_pipe.hpp staged_rows) so that the residual loads and the stores are 16 bytes per lane on whole lines.
template <typename T, int EPI, int MT>
__device__ __forceinline__ void gemm_store_rows(const GemmParams<T> &p, const f32x4 (&acc)[4][MT], int m_base, int n_base, unsigned char *stage, int lane) {
    const int n = n_base + (lane & 7) * 8, rr = lane >> 3;
    const float4 b0 = *reinterpret_cast<const float4 *>(p.bias + n), b1 = *reinterpret_cast<const float4 *>(p.bias + n + 4);
    const bool ln_in = EPI != EPI_BIAS_RESIDUAL && p.ln_stats != nullptr;    // (kernel-uniform) the rows of A are normalised here, not by a LayerNorm pass
    const bool ln_out = EPI == EPI_BIAS_RESIDUAL && p.ln_partial != nullptr;  // the stored rows' statistics are left for the GEMM that reads them next
    float4 c0 = float4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
    if (ln_in) {
        c0 = *reinterpret_cast<const float4 *>(p.ln_c1 + n);
        c1 = *reinterpret_cast<const float4 *>(p.ln_c1 + n + 4);
    }
    // residual rows: loaded one fragment row ahead (two register sets), not in front of each store.  The residual may BE the output
    // (x += proj(...)): every element is read by the lane that later writes it, rows of fragment row mt + 1 are read before rows of
    // mt are written -- no hazard, but the compiler cannot know, so the order is set by hand.
    constexpr int AHEAD = HIVE_GEMM_AHEAD;  // fragment rows the residual loads run ahead (a row's turn is ~0.5 us, a trip to HBM under load 2 us: conv.hip)
    vec<T, 8> rs[AHEAD + 1][2];
    // Round 5: FEWER vector-memory instructions.  A CU's vector-memory path takes one wave-instruction per ~45 cycles whatever its width, and the epilogue is paced
    // by it: the q|k / fc1 epilogues issued as many (mean, rstd) loads -- one 8-byte broadcast load per fragment-row half -- as stores (16 + 16 per wave and tile), the
    // proj / fc2 epilogues 16 eight-lane stores of the LayerNorm partials beside their 16 + 16.  Now the wave's rows' statistics come in with TWO loads per tile:
    // "pair" q = 2 mt + j of the epilogue's 2 MT steps covers rows 16 mt + 8 j + rr; lane (rr, c) loads the rows of pairs c and c + 8, and step q takes its values
    // from lane c = q & 7 of its own 8-lane group with a ds_swizzle broadcast (LDS crossbar, no memory, no address register).  The partials go the other way: at
    // step q lane c = q & 7 keeps the group's (sum, M2), and two 64-lane stores write them all at the end.
    float2 st_in[2] = {float2{0.f, 0.f}, float2{0.f, 0.f}}, st_out[2] = {float2{0.f, 0.f}, float2{0.f, 0.f}};
    const int pc = lane & 7;
    auto pair_row = [&](int q) { return m_base + (q >> 1) * 16 + 8 * (q & 1) + rr; };
    if (ln_in) {
        st_in[0] = *reinterpret_cast<const float2 *>(p.ln_stats + 2 * (size_t)min(pair_row(pc), p.M - 1));
        if (MT > 4) st_in[1] = *reinterpret_cast<const float2 *>(p.ln_stats + 2 * (size_t)min(pair_row(pc + 8), p.M - 1));
    }
    auto pre = [&](int mt, int j) {
        const int m = min(m_base + mt * 16 + 8 * j + rr, p.M - 1);
        if (EPI == EPI_BIAS_RESIDUAL) rs[mt % (AHEAD + 1)][j] = *reinterpret_cast<const vec<T, 8> *>(p.residual + (size_t)m * p.ldc + n);
    };
    // value of lane (lane & ~7) | k, for all lanes: ds_swizzle in bit-mask mode (and 0x18, or k, xor 0 -- within each 32-lane half)
#define HIVE_BCAST8(v, k) __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, (v)), 0x18 | ((k) << 5)))
    hive_mfma::staged_rows<MT, AHEAD>(stage, acc, lane, pre, [&](int r, int, const f32x4 &lo, const f32x4 &hi, int mt, int j) {
        const int m = m_base + r;
        if (m >= p.M) return;
        float o[8];
        if (ln_in) {
            float2 st;  // (mean, rstd) of row m: held by lane q & 7 of this 8-lane group, slot q >> 3 (q = 2 mt + j: compile-time after unrolling)
            {
                const int q = 2 * mt + j;
                const float2 src = st_in[q >> 3];
                switch (q & 7) {
                    case 0: st = float2{HIVE_BCAST8(src.x, 0), HIVE_BCAST8(src.y, 0)}; break;
                    case 1: st = float2{HIVE_BCAST8(src.x, 1), HIVE_BCAST8(src.y, 1)}; break;
                    case 2: st = float2{HIVE_BCAST8(src.x, 2), HIVE_BCAST8(src.y, 2)}; break;
                    case 3: st = float2{HIVE_BCAST8(src.x, 3), HIVE_BCAST8(src.y, 3)}; break;
                    case 4: st = float2{HIVE_BCAST8(src.x, 4), HIVE_BCAST8(src.y, 4)}; break;
                    case 5: st = float2{HIVE_BCAST8(src.x, 5), HIVE_BCAST8(src.y, 5)}; break;
                    case 6: st = float2{HIVE_BCAST8(src.x, 6), HIVE_BCAST8(src.y, 6)}; break;
                    default: st = float2{HIVE_BCAST8(src.x, 7), HIVE_BCAST8(src.y, 7)}; break;
                }
            }
            o[0] = st.y * (lo[0] - st.x * c0.x) + b0.x, o[1] = st.y * (lo[1] - st.x * c0.y) + b0.y;
            o[2] = st.y * (lo[2] - st.x * c0.z) + b0.z, o[3] = st.y * (lo[3] - st.x * c0.w) + b0.w;
            o[4] = st.y * (hi[0] - st.x * c1.x) + b1.x, o[5] = st.y * (hi[1] - st.x * c1.y) + b1.y;
            o[6] = st.y * (hi[2] - st.x * c1.z) + b1.z, o[7] = st.y * (hi[3] - st.x * c1.w) + b1.w;
        } else {
            o[0] = lo[0] + b0.x, o[1] = lo[1] + b0.y, o[2] = lo[2] + b0.z, o[3] = lo[3] + b0.w;
            o[4] = hi[0] + b1.x, o[5] = hi[1] + b1.y, o[6] = hi[2] + b1.z, o[7] = hi[3] + b1.w;
        }
        if (EPI == EPI_BIAS && n < p.q_cols) {
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] *= p.q_scale;
        }
        if (EPI == EPI_BIAS_GELU) {
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const f32x2 g = gelu_exact2(f32x2{o[k], o[k + 1]});
                o[k] = g.x;
                o[k + 1] = g.y;
            }
        }
        if (EPI == EPI_BIAS_RESIDUAL) {
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] += (float)rs[mt % (AHEAD + 1)][j][k];
        }
        vec<T, 8> ov;
#pragma unroll
        for (int k = 0; k < 8; ++k) ov[k] = (T)o[k];
        if constexpr (HIVE_GEMM_ABLATE & 2)  // tuning build: everything but the store itself
            asm volatile("" ::"v"(ov));
        else
            *reinterpret_cast<vec<T, 8> *>(p.C + (size_t)m * p.ldc + n) = ov;
        if (ln_out) {
            // statistics of the STORED (rounded) values of this row's 64 columns: the 8 lanes of the row sit side by side (lane & 7); the sum
            // of squares is taken about the 64 values' own mean (ln_finalize_kernel merges the groups exactly: no E[x^2] - mean^2 cancellation)
            float v[8], sum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[k] = (float)ov[k];
                sum += v[k];
            }
            sum = dpp_oct_total(sum);
            const float mean = sum * (1.0f / 64.0f);
            float m2 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) m2 += (v[k] - mean) * (v[k] - mean);
            m2 = dpp_oct_total(m2);
            {  // lane q & 7 of the row's group keeps the pair's partial; stored behind the loop, two instructions per wave
                const int q = 2 * mt + j;
                if (pc == (q & 7)) st_out[q >> 3] = float2{sum, m2};
            }
        }
    });
    if (ln_out) {
#pragma unroll
        for (int h = 0; h < (MT > 4 ? 2 : 1); ++h) {
            const int m = pair_row(pc + 8 * h);
            if (m < p.M) *reinterpret_cast<float2 *>(p.ln_partial + 2 * ((size_t)m * (p.N >> 6) + (n_base >> 6))) = st_out[h];
        }
    }
#undef HIVE_BCAST8
}

// (mean, rstd) of every row of x [M][256 CH], exactly as layernorm_kernel computes them: the statistics of a ViT's first LayerNorm, whose input no
// GEMM epilogue produced
template <typename T, int CH>
__global__ __launch_bounds__(256) void ln_row_stats_kernel(const T *__restrict__ x, float *__restrict__ stats, int M, float eps) {
    constexpr int D = 256 * CH;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const T *xr = x + (size_t)row * D;
    float v[4 * CH];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const vec<T, 4> t = *reinterpret_cast<const vec<T, 4> *>(xr + i * 256 + lane * 4);
        for (int j = 0; j < 4; ++j) {
            v[4 * i + j] = (float)t[j];
            sum += v[4 * i + j];
        }
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum / (float)D;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < 4 * CH; ++i) {
        const float d = v[i] - mean;
        var += d * d;
    }
    for (int off = 32; off > 0; off >>= 1) var += __shfl_xor(var, off);
    if (lane == 0) *reinterpret_cast<float2 *>(stats + 2 * (size_t)row) = float2{mean, rsqrtf(var / (float)D + eps)};
}

// per-row groups (sum, M2 about the group's own mean) of 64 columns each -> (mean, rstd) of the row: Chan's merge of the groups.  16 lanes per row
// (a DPP row; groups <= 16, i.e. D <= 1024): lane g reads group g -- a row's partials are 8 `groups` contiguous bytes, four rows per wave-instruction.
__device__ __forceinline__ float dpp_row16_total(float v) {
#define HIVE_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    HIVE_DPP_ADD(0x128);  // row_ror:8
    HIVE_DPP_ADD(0x124);  // row_ror:4
    HIVE_DPP_ADD(0x122);  // row_ror:2
    HIVE_DPP_ADD(0x121);  // row_ror:1
#undef HIVE_DPP_ADD
    return v;
}
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float *__restrict__ partial, int M, int groups, float eps, float *__restrict__ stats) {
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4), g = threadIdx.x & 15;
    float2 pg = float2{0.f, 0.f};
    if (row < M && g < groups) pg = reinterpret_cast<const float2 *>(partial)[(size_t)row * groups + g];
    const float mean = dpp_row16_total(pg.x) / (float)(64 * groups);
    const float d = pg.x * (1.0f / 64.0f) - mean;
    const float m2 = dpp_row16_total(g < groups ? pg.y + 64.0f * d * d : 0.f);
    if (row < M && g == 0) *reinterpret_cast<float2 *>(stats + 2 * (size_t)row) = float2{mean, rsqrtf(m2 / (float)(64 * groups) + eps)};
}

// Weights of a GEMM with its LayerNorm folded in (one thread block per output column n): W'[n][k] = T(gamma[k] W[n][k]),
// c1[n] = sum_k W'[n][k] (of the STORED values), c2[n] = sum_k beta[k] W[n][k] + b[n]; sums in float64.
template <typename T>
__global__ __launch_bounds__(256) void ln_fold_weights_kernel(const T *__restrict__ W, const float *__restrict__ bias, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, int K, T *__restrict__ Wf, float *__restrict__ c1, float *__restrict__ c2) {
    __shared__ double red[2][256];
    const int n = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float w = (float)W[(size_t)n * K + k];
        const T wf = (T)(gamma[k] * w);
        Wf[(size_t)n * K + k] = wf;
        s1 += (double)(float)wf;
        s2 += (double)beta[k] * (double)w;
    }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[0][threadIdx.x] += red[0][threadIdx.x + off];
            red[1][threadIdx.x] += red[1][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        c1[n] = (float)red[0][0];
        c2[n] = (float)(red[1][0] + (double)bias[n]);
    }
}

// C tile = TM x 128, K-step 64, TM/32 waves (each a 64 x 64 sub-tile = 4 x 4 MFMA 16x16x32 accumulators), an
// NST-stage LDS ring filled by LDS-DMA, ONE raw s_barrier per K-step.
//   NST = 3: stages kt+1 and kt+2 in flight while kt is multiplied, counted s_waitcnt vmcnt (never 0 in the loop);
//            144 KiB at TM = 256: one workgroup per CU.
//   NST = 4 (round 4, where the items do not fill the CUs -- small batches): three stages in flight.  With one workgroup on a CU nothing hides a
//            stage's trip from HBM / L2 (~1.2 us behind ~0.5 us of issue for 32 KiB), and with two stages a K-step lasted as long as that trip.
//   NST = 2 (used otherwise): 64 KiB at TM = 128, so TWO workgroups share a CU and one's prologue (first stage in flight) and
//            epilogue (GELU, stores) overlap the other's K loop -- at K = 768 a tile is only 12 K-steps long and those
//            ends were a third of its time.  Measured at M = 19456: 640 -> 664 (qkv), 491 -> 529 (proj), 689 -> 730
//            (fc2) TFLOP/s against TM = 256 / NST = 3.  Also measured and not kept: 256 x 256 tiles with 128 x 128 per
//            wave (half the LDS bytes per flop, one wave per SIMD: 570-680 TFLOP/s, register-staged variant spills),
//            LDS-DMA pieces interleaved between the MFMAs (no change).
// EPI_QKV here means "v^T tile": orientation A.W^T and the transposed store; q|k columns use EPI_BIAS.
template <typename T, int EPI, int TM, int NST>
__global__ __launch_bounds__(TM * 2, 1) void gemm_kernel(GemmParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // NST stages x (A tile TM x 64, W tile 128 x 64) + 4 KiB per wave for the epilogue
    constexpr int NWAVES = TM / 32;
    constexpr int A_GROUPS = TM / 8, GROUPS = A_GROUPS + 16, PER_WAVE = GROUPS / NWAVES;
    constexpr int STAGE_BYTES = GROUPS * 1024;
    constexpr bool VT = (EPI == EPI_QKV);
    constexpr bool ALL = (EPI == EPI_QKV_ALL);  // round 5, small batches: tiles with n0 < n_split are q | k tiles (EPI_BIAS), the others v^T tiles -- one launch instead of two
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // PERSISTENT workgroups, XCD-aware: the grid is a multiple of 8 (<= 2 workgroups per CU); workgroups are dealt round-robin
    // over the 8 XCDs, each XCD owns a contiguous run of tiles (n fastest: neighbours share their A panel in that XCD's L2)
    // and its workgroups walk the run with a stride of gridDim / 8.  The K-steps of a workgroup's tiles form ONE stream: the
    // first stage of the next tile is prefetched between the MFMAs of the last K-step of the current one and lands while its
    // epilogue (GELU, residual, stores) runs -- at K = 768 a tile is 12 steps, and its exposed first fill was a step's worth.
    // Items = tiles x split_k (split_k = 1 except at small M): item i multiplies K-steps [k0, k1) of tile i / split_k.
    const int S = p.split_k, KT = p.K / BK;
    const int tiles_n = p.N / BN, n_tiles = ((p.M + TM - 1) / TM) * tiles_n * S;
    const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3, tq = n_tiles >> 3, tr = n_tiles & 7;
    const int run0 = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, run_n = tq + (xcd < tr ? 1 : 0);
    int tl = blockIdx.x >> 3;  // position in the XCD's run
    if (tl >= run_n) return;   // (whole workgroup)
    auto coords = [&](int item, int &m, int &n, int &ka, int &kb) {
        const int t = item / S, s = item - t * S;
        m = (t / tiles_n) * TM;
        n = (t % tiles_n) * BN;
        ka = s * KT / S;
        kb = (s + 1) * KT / S;
    };
    int m0, n0, k0, k1;
    coords(run0 + tl, m0, n0, k0, k1);

    // one LDS-DMA wave-instruction: group g = wave + j * NWAVES (8 rows x 128 B of A or of W) of K-step kt of tile (tm0, tn0)
    auto issue_piece = [&](int tm0, int tn0, int kt, int buf, int j) {
        unsigned char *st = lds + buf * STAGE_BYTES;
        const int g = wave + j * NWAVES;
        if (g < A_GROUPS)
            stage_group<T, (NST > 2)>(p.A, p.K, tm0, p.M - 1, kt * BK, st, g, lane);
        else
            stage_group<T, (NST > 2)>(p.W, p.K, tn0, p.N - 1, kt * BK, st + A_GROUPS * 1024, g - A_GROUPS, lane);
    };
    // The K-steps of a workgroup's items form ONE stream; the stage being ISSUED runs AHEAD = NST - 1 steps in front of the one being multiplied
    // (across item boundaries: the next item's first stages land while this one's epilogue runs).  Past the end of the stream the cursor parks on
    // the last stage, which is issued again into a buffer nobody reads any more: the counted waits below rely on every step issuing a stage.
    constexpr int AHEAD = NST - 1;
    struct Cursor {
        int m0, n0, kt, k1, tl;
    } is = {m0, n0, k0, k1, tl};
    auto advance = [&](Cursor &c) {
        if (c.kt + 1 < c.k1) {
            ++c.kt;
        } else if (c.tl + per_xcd < run_n) {
            c.tl += per_xcd;
            coords(run0 + c.tl, c.m0, c.n0, c.kt, c.k1);
        }
    };
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) {
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) issue_piece(is.m0, is.n0, is.kt, a, j);
        advance(is);
    }
    const int fr = lane & 15, fq = lane >> 4;
    int buf = 0;  // LDS stage of the current K-step (cycles along the whole stream)
    for (;;) {
    const bool has_next = tl + per_xcd < run_n;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = k0; kt < k1; ++kt) {
        // this step's stage has landed when at most the AHEAD - 1 younger stages' pieces are outstanding (LDS-DMA loads retire in order)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PER_WAVE) : "memory");
        __builtin_amdgcn_s_barrier();  // everyone's current stage landed; everyone finished reading the previous one
        // the stage AHEAD steps on goes out piece by piece between this step's MFMA slots (mfma_pipe.hpp), into the buffer the previous step read
        const int ib = buf + AHEAD >= NST ? buf + AHEAD - NST : buf + AHEAD;
        const unsigned char *a_t = lds + buf * STAGE_BYTES, *w_t = a_t + A_GROUPS * 1024;
        // VT: acc[mt][nt] = A_frag . W_frag^T (rows = m, cols = n); else acc[nt][mt] = W_frag . A_frag^T (rows = n, cols = m)
        if (ALL && n0 >= p.n_split)  // (workgroup-uniform)
            hive_mfma::kstep64<T, 4, true>(a_t, w_t, wr * 64, wc * 64, fr, fq, acc, PER_WAVE, [&](int j) { issue_piece(is.m0, is.n0, is.kt, ib, j); });
        else
            hive_mfma::kstep64<T, 4, VT>(a_t, w_t, wr * 64, wc * 64, fr, fq, acc, PER_WAVE, [&](int j) { issue_piece(is.m0, is.n0, is.kt, ib, j); });
        advance(is);
        buf = buf + 1 == NST ? 0 : buf + 1;
    }

    bool store = true;
    if (S > 1) store = hive_mfma::splitk_combine<TM * 2>(S, p.sk_ws, p.sk_count, run0 + tl, acc, tid, reinterpret_cast<int *>(lds + NST * STAGE_BYTES));
    if (store) {
    // epilogue
    if (!VT && !(ALL && n0 >= p.n_split)) {
        if constexpr (!VT) gemm_store_rows<T, ALL ? EPI_BIAS : EPI, 4>(p, acc, m0 + wr * 64, n0 + wc * 64, lds + NST * STAGE_BYTES + wave * 4096, lane);
    } else if constexpr (VT || ALL) {
        // v^T[b][h][c][token]: the wave's 64 tokens x 64 channels are one head of one image (Np % 64 == 0): 64 rows of 128 contiguous
        // bytes.  A lane owns 4 consecutive tokens of one channel, so a direct store is 16 rows x 32 bytes per instruction; instead
        // the block is turned around in the wave's 4 KiB of LDS, 32 channels at a time ([channel][64 token slots] bf16, 16-byte chunk c
        // of row r at c ^ (r & 7)) and leaves as 8 rows x 128 bytes per instruction.  vt_slot exchanges bits 2 and 3 of the token
        // index = the two bits of fq: the lane's quad stays contiguous.
        unsigned char *ot = lds + NST * STAGE_BYTES + wave * 4096;
        const int mw = m0 + wr * 64;  // M = B Np is a multiple of 64: the block is inside or outside as a whole
        const int fqs = ((fq & 1) << 1) | (fq >> 1);
        float4 ls[4][2];  // LayerNorm folded in: (mean, rstd) of the lane's 4 tokens per fragment row, loaded once up front (not per channel block, each wait behind a store)
        if (p.ln_stats) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float *st = p.ln_stats + 2 * (size_t)min(mw + mt * 16 + fq * 4, p.M - 4);
                ls[mt][0] = *reinterpret_cast<const float4 *>(st), ls[mt][1] = *reinterpret_cast<const float4 *>(st + 4);
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int ntl = 0; ntl < 2; ++ntl) {
                const int nt = half * 2 + ntl, row = ntl * 16 + fr;
                const float b = p.bias[n0 + wc * 64 + nt * 16 + fr];
                const float c1v = p.ln_stats ? p.ln_c1[n0 + wc * 64 + nt * 16 + fr] : 0.f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    vec<T, 4> ov;
                    if (p.ln_stats) {  // (kernel-uniform) LayerNorm folded in: the lane's 4 tokens' (mean, rstd)
                        const float4 s0 = ls[mt][0], s1 = ls[mt][1];
                        ov[0] = (T)(s0.y * (acc[mt][nt][0] - s0.x * c1v) + b);
                        ov[1] = (T)(s0.w * (acc[mt][nt][1] - s0.z * c1v) + b);
                        ov[2] = (T)(s1.y * (acc[mt][nt][2] - s1.x * c1v) + b);
                        ov[3] = (T)(s1.w * (acc[mt][nt][3] - s1.z * c1v) + b);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) ov[j] = (T)(acc[mt][nt][j] + b);
                    }
                    *reinterpret_cast<vec<T, 4> *>(ot + row * 128 + (((mt * 2 + (fqs >> 1)) ^ (row & 7)) << 4) + (fqs & 1) * 8) = ov;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (mw < p.M) {
                const int img = mw / p.Np, tok0 = mw - img * p.Np, head = (n0 - (ALL ? p.n_split : 0) + wc * 64) >> 6;  // (_ALL: n0 counts from the q columns)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 8 * i + (lane >> 3), c = lane & 7;
                    const vec<T, 8> o8 = *reinterpret_cast<const vec<T, 8> *>(ot + row * 128 + ((c ^ (row & 7)) << 4));
                    *reinterpret_cast<vec<T, 8> *>(p.vT + (((size_t)img * p.H + head) * 64 + half * 32 + row) * p.Np + tok0 + 8 * c) = o8;
                }
            }
            __builtin_amdgcn_wave_barrier();  // the second half's writes stay behind these reads
        }
    }
    }  // store
    if (!has_next) break;
    // The deep ring counts its own (untracked) pieces, so nothing else may be outstanding in the K loop -- neither in fact (an epilogue's stores and a
    // stage's loads do not retire in one order) nor in the compiler's books: a residual row loaded ahead for a row past M is never consumed, and hipcc
    // then waits for it where its register is next written -- an s_waitcnt vmcnt(0..1) at the top of EVERY K-step, which drained the ring.  A wait the
    // compiler can see (the builtin, not an asm statement) settles both.
    if constexpr (NST > 2) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), nothing else
    tl += per_xcd;
    coords(run0 + tl, m0, n0, k0, k1);
    }  // persistent tile loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant last stage
}

// ------------------------------------------------------------------------------------------------
// Large-M variant: C tile 256 x 256, K-step 64, EIGHT waves as 2 (M) x 4 (N), each a 128 x 64 sub-tile (8 x 4 MFMA
// 16x16x32 accumulators, 128 VGPRs: two waves per SIMD).  Half the L2 operand bytes per flop of the 128 x 128 tile
// (the operand stream, not the MFMAs or LDS, bounds that one: DESIGN.md section 5.3).  Two 64-KiB LDS-DMA stages.
//
// What the K-step costs, measured (tools/probe_gemm_stamps.py: per-workgroup clocks; timing builds with parts of the loop removed):
// 2700-2750 shader cycles against 2048 of MFMA issue.  Without the LDS-DMA stream the same loop (MFMAs, fragment reads, barrier) runs
// 4096^3 at 1478 TFLOP/s instead of 1230-1290 (hipBLASLt: 1470); without the barrier as well, 1506.  It is not the pieces' latency:
// a wave's own pieces have landed when it reaches the barrier (vmcnt wait: 10 % of its waiting), and a FOUR-stage ring of 32-deep
// steps (stages issued three steps ahead, counted vmcnt -- built, bit-identical results, 1241 vs 1280) is no faster; issuing the
// pieces later in the step is slower (1290 -> 1180 -> 1156 for a start at slot 0 / 3 / 6), two per slot marginally faster.  What
// did help: the rotated K-step of mfma_pipe.hpp (KPipe: +2-5 %) and the epilogue through LDS (+10-20 % at K = 768).  The decisive
// timing build: every piece still issued and still writing its 1 KiB into LDS, but all reading the SAME 128 bytes -- 1670 TFLOP/s.
// So neither the issue of the LDS-DMA instructions, nor their address arithmetic (hoisted out of the loop since: +1-2 %), nor the LDS
// writes cost the 20 %: fetching 64 KiB of distinct lines per 2048 MFMA cycles and CU from L2 does (32 B/clk/CU wanted, ~26 B/clk
// delivered with all 256 CUs streaming: tools/ubench/ldsdma.hip).  At 256 x 256 x 64 the tile is L2-bandwidth-bound; a larger tile does
// not fit the register file.
constexpr int T256 = 256, T256_STAGE = 2 * T256 / 8 * 1024;

#ifdef HIVE_GEMM_STAMPS  // tuning builds only (make stamps): per-workgroup phase clocks of gemm256_kernel, read back by tools/probe_gemm_stamps.py
__device__ unsigned long long g_stamps[8 * 8192];
extern "C" int hive_debug_read_stamps(void *host, size_t bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#define HIVE_STAMP(i) do { if (tid == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + (i)] = clock64(); } while (0)
#define HIVE_STAMP_ADD(i, v) do { if (tid == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + (i)] = (v); } while (0)
#else
#define HIVE_STAMP(i) do { } while (0)
#define HIVE_STAMP_ADD(i, v) do { } while (0)
#endif

template <typename T, int EPI>
__global__ __launch_bounds__(512, 1) void gemm256_kernel(GemmParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 stages x (A tile 256 x 64, W tile 256 x 64) + 8 x 4 KiB for the epilogue
    constexpr int A_GROUPS = T256 / 8, GROUPS = 2 * A_GROUPS, PER_WAVE = GROUPS / 8;
    static_assert(EPI != EPI_QKV, "the transposed v^T store stays with the 128-row kernel (N = 768)");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;  // 128 rows x 64 columns per wave
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    const int tiles_n = p.N / T256;
    const int m0 = (tile / tiles_n) * T256, n0 = (tile % tiles_n) * T256;

    // LDS-DMA addressing, hoisted: piece j of this wave (group g = wave + 8 j: j < 4 an A group, else a W group; 8 rows x 128 B) reads
    // [uniform panel base + 128 kt] + [per-lane byte offset of (row, swizzled chunk)], and the per-lane part does not depend on kt --
    // 8 registers computed once instead of a clamp, two 64-bit multiply-adds and three 64-bit adds per piece and K-step (~90 VALU
    // instructions per wave and step beside its 64 MFMAs).
    unsigned piece_off[PER_WAVE];
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        const int g = wave + j * 8, row = (g & (A_GROUPS - 1)) * 8 + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int rel = j < PER_WAVE / 2 ? min(row, p.M - 1 - m0) : min(row, p.N - 1 - n0);  // rows past the end: the last row again (never stored)
        piece_off[j] = (unsigned)rel * (unsigned)p.K * 2u + (unsigned)chunk * 16u;
    }
    const char *a_panel = reinterpret_cast<const char *>(p.A + (size_t)m0 * p.K), *w_panel = reinterpret_cast<const char *>(p.W + (size_t)n0 * p.K);
    auto issue_piece = [&](int kt, int stage, int j) {
        const char *g = (j < PER_WAVE / 2 ? a_panel : w_panel) + (size_t)kt * (BK * 2) + piece_off[j];
        __builtin_amdgcn_global_load_lds((const void *)g, (__attribute__((address_space(3))) void *)(lds + stage * T256_STAGE + (wave + j * 8) * 1024), 16, 0, 0);
    };

    f32x4 acc[4][8];  // acc[nt][mt] = W_frag . A_frag^T : rows = n, cols = m
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int KT = p.K / BK;
    HIVE_STAMP(0);
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) issue_piece(0, 0, j);
    const int fr = lane & 15, fq = lane >> 4;
#ifdef HIVE_GEMM_STAMPS
    unsigned long long waited = 0, waited_vm = 0;
#endif
    hive_mfma::KPipe<T, 8, false> pipe;
    pipe.a_row0 = wr * 128, pipe.w_row0 = wc * 64, pipe.fr = fr, pipe.fq = fq;
    for (int kt = 0; kt < KT; ++kt) {
#ifdef HIVE_GEMM_STAMPS
        const unsigned long long w0 = clock64();
#endif
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (lgkmcnt: the held-back slots' fragment reads of stage kt - 1)
#ifdef HIVE_GEMM_STAMPS
        const unsigned long long w1 = clock64();
#endif
        __builtin_amdgcn_s_barrier();  // everyone's stage kt landed; everyone finished reading stage kt - 1
#ifdef HIVE_GEMM_STAMPS
        if (kt == 0) HIVE_STAMP(1); else waited += clock64() - w0, waited_vm += w1 - w0;
#endif
        const int nk = min(kt + 1, KT - 1);  // next stage (past the end: the last one again), issued between the MFMA slots
        const unsigned char *a_t = lds + (kt & 1) * T256_STAGE, *w_t = a_t + A_GROUPS * 1024;
        pipe.begin(a_t, w_t);
        if (kt > 0) pipe.flush(acc);  // the previous step's last slots, under the latency of this step's first reads
        pipe.body(acc, PER_WAVE, [&](int j) { issue_piece(nk, (kt + 1) & 1, j); });
    }
    pipe.flush(acc);
    HIVE_STAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant last stage
    HIVE_STAMP(3);
    HIVE_STAMP_ADD(6, waited);
    HIVE_STAMP_ADD(7, waited_vm);

    gemm_store_rows<T, EPI, 8>(p, acc, m0 + wr * 128, n0 + wc * 64, lds + 2 * T256_STAGE + wave * 4096, lane);
    HIVE_STAMP(4);
#ifdef HIVE_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HIVE_STAMP(5);
#endif
}

// What the epilogues cost at the bench batch (M = 130112; `make ablate_gemm`, tools/probe_gemm_tiles.py): without its epilogue the K loops of q|k run in 231 us
// (1330 TFLOP/s) against 347 with it, proj 115 / 209, fc1 + GELU 417 / 756, fc2 507 / 572 -- 7.3 ms of a 76 ms forward; with everything but the store
// instructions 261 / 174 / 587 / 541.  Measured INTERLEAVED in one process (the clock drifts over a run: a first comparison across processes showed gains that
// were drift): non-temporal stores of C change nothing (+-0.5 %; fc1 1.4 % slower), nor does starting every other workgroup half a tile late so that the
// epilogues' stores do not hit HBM together, nor does letting a tile's first K-step start before the previous tile's stores have retired (vmcnt(16) instead of
// vmcnt(0): the stage it needs is older than those stores).  The epilogue cannot overlap the next tile's K loop inside one workgroup (the accumulators are the registers), and two
// co-resident workgroups need tiles of 128 x 256 at most (LDS), whose operand stream -- 1.5 x the bytes per flop through the CU's ~26 B/clk vector-memory path --
// costs what the overlap gains (the 128 x 128 two-workgroup form measures 450 vs 347 us on q|k).
// The same tile as PERSISTENT workgroups (one per CU, XCD-aware runs of tiles as in csrc/conv.hip): the K-steps of a workgroup's tiles
// form one stream, the first stage of the next tile is issued during the last K-step of the current one and lands under its epilogue.
template <typename T, int EPI>
__global__ __launch_bounds__(512, 1) void gemm256p_kernel(GemmParams<T> p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int A_GROUPS = T256 / 8, GROUPS = 2 * A_GROUPS, PER_WAVE = GROUPS / 8;
    constexpr bool VT = (EPI == EPI_QKV);  // the v columns of the QKV projection: orientation A.W^T (a lane owns 4 consecutive TOKENS of one channel)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_n = p.N / T256, n_tiles = ((p.M + T256 - 1) / T256) * tiles_n;
    const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3, tq = n_tiles >> 3, tr = n_tiles & 7;
    const int run0 = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, run_n = tq + (xcd < tr ? 1 : 0);
    int tl = blockIdx.x >> 3;
    if (tl >= run_n) return;  // (whole workgroup)
    int m0, n0;
    unsigned piece_off[PER_WAVE];
    const char *a_panel, *w_panel;
    auto setup = [&](int tile) {
        m0 = (tile / tiles_n) * T256;
        n0 = (tile % tiles_n) * T256;
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const int g = wave + j * 8, row = (g & (A_GROUPS - 1)) * 8 + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
            const int rel = j < PER_WAVE / 2 ? min(row, p.M - 1 - m0) : min(row, p.N - 1 - n0);
            piece_off[j] = (unsigned)rel * (unsigned)p.K * 2u + (unsigned)chunk * 16u;
        }
        a_panel = reinterpret_cast<const char *>(p.A + (size_t)m0 * p.K);
        w_panel = reinterpret_cast<const char *>(p.W + (size_t)n0 * p.K);
    };
    auto issue_piece = [&](int kt, int stage, int j) {
        const char *g = (j < PER_WAVE / 2 ? a_panel : w_panel) + (size_t)kt * (BK * 2) + piece_off[j];
        __builtin_amdgcn_global_load_lds((const void *)g, (__attribute__((address_space(3))) void *)(lds + stage * T256_STAGE + (wave + j * 8) * 1024), 16, 0, 0);
    };
    const int KT = p.K / BK;
    setup(run0 + tl);
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) issue_piece(0, 0, j);
    hive_mfma::KPipe<T, 8, VT> pipe;
    pipe.a_row0 = wr * 128, pipe.w_row0 = wc * 64, pipe.fr = lane & 15, pipe.fq = lane >> 4;
    int buf = 0;
    for (;;) {
        const bool has_next = tl + per_xcd < run_n;
        const int em0 = m0, en0 = n0;
        f32x4 acc[VT ? 8 : 4][VT ? 4 : 8];  // [nt][mt], or [mt][nt] for the v^T tiles
#pragma unroll
        for (int i = 0; i < (VT ? 8 : 4); ++i)
#pragma unroll
            for (int j = 0; j < (VT ? 4 : 8); ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < KT; ++kt) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int nk = kt + 1;
            if (nk == KT) {
                if (has_next) {
                    setup(run0 + tl + per_xcd);  // this tile's last stage is in LDS: its offsets are not needed any more
                    nk = 0;
                } else {
                    nk = KT - 1;  // the last stage again, into the buffer nobody reads any more
                }
            }
            const unsigned char *a_t = lds + buf * T256_STAGE, *w_t = a_t + A_GROUPS * 1024;
            pipe.begin(a_t, w_t);
            if (kt > 0) pipe.flush(acc);
            pipe.body(acc, PER_WAVE, [&](int j) { issue_piece(nk, buf ^ 1, j); });
            buf ^= 1;
        }
        pipe.flush(acc);
        if constexpr (HIVE_GEMM_ABLATE & 1) {  // tuning build (make ablate_gemm): the K loops alone -- the accumulators are kept alive, nothing is stored
#pragma unroll
            for (int i = 0; i < (VT ? 8 : 4); ++i)
#pragma unroll
                for (int j = 0; j < (VT ? 4 : 8); ++j) asm volatile("" ::"v"(acc[i][j]));
        } else if constexpr (!VT) {
            gemm_store_rows<T, EPI, 8>(p, acc, em0 + wr * 128, en0 + wc * 64, lds + 2 * T256_STAGE + wave * 4096, lane);
        } else {
            // v^T[b][h][c][token]: the wave's 128 tokens x 64 channels are two 64-token blocks of one head (Np % 64 == 0), each 64 rows of
            // 128 contiguous bytes; turned around in the wave's 4 KiB of LDS, 64 tokens x 32 channels at a time (see gemm_kernel's v^T path)
            unsigned char *ot = lds + 2 * T256_STAGE + wave * 4096;
            const int fr = lane & 15, fq = lane >> 4, fqs = ((fq & 1) << 1) | (fq >> 1);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int mw = em0 + wr * 128 + blk * 64;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int ntl = 0; ntl < 2; ++ntl) {
                        const int nt = half * 2 + ntl, row = ntl * 16 + fr;
                        const float bv = p.bias[en0 + wc * 64 + nt * 16 + fr];
                        const float c1v = p.ln_stats ? p.ln_c1[en0 + wc * 64 + nt * 16 + fr] : 0.f;
#pragma unroll
                        for (int mtl = 0; mtl < 4; ++mtl) {
                            vec<T, 4> ov;
                            if (p.ln_stats) {  // (kernel-uniform) LayerNorm folded in: the lane's 4 tokens' (mean, rstd).  (Hoisting the 8 float4 of a 64-token
                                // block out of the channel loops spills 27 registers here; gemm_kernel's v^T path has the room and does.)
                                const float *st = p.ln_stats + 2 * (size_t)min(mw + mtl * 16 + fq * 4, p.M - 4);
                                const float4 s0 = *reinterpret_cast<const float4 *>(st), s1 = *reinterpret_cast<const float4 *>(st + 4);
                                ov[0] = (T)(s0.y * (acc[blk * 4 + mtl][nt][0] - s0.x * c1v) + bv);
                                ov[1] = (T)(s0.w * (acc[blk * 4 + mtl][nt][1] - s0.z * c1v) + bv);
                                ov[2] = (T)(s1.y * (acc[blk * 4 + mtl][nt][2] - s1.x * c1v) + bv);
                                ov[3] = (T)(s1.w * (acc[blk * 4 + mtl][nt][3] - s1.z * c1v) + bv);
                            } else
#pragma unroll
                            for (int j = 0; j < 4; ++j) ov[j] = (T)(acc[blk * 4 + mtl][nt][j] + bv);
                            *reinterpret_cast<vec<T, 4> *>(ot + row * 128 + (((mtl * 2 + (fqs >> 1)) ^ (row & 7)) << 4) + (fqs & 1) * 8) = ov;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (mw < p.M) {
                        const int img = mw / p.Np, tok0 = mw - img * p.Np, head = (en0 + wc * 64) >> 6;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int row = 8 * i + (lane >> 3), c = lane & 7;
                            const vec<T, 8> o8 = *reinterpret_cast<const vec<T, 8> *>(ot + row * 128 + ((c ^ (row & 7)) << 4));
                            *reinterpret_cast<vec<T, 8> *>(p.vT + (((size_t)img * p.H + head) * 64 + half * 32 + row) * p.Np + tok0 + 8 * c) = o8;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (!has_next) break;
        tl += per_xcd;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
template <typename T>
struct AttnParams {
    const T *qk;  // [B*Np][2D]; q pre-multiplied by head_dim^-0.5 * log2(e) (hive_vit_qkv)
    const T *vT;  // [B][H][64][Np]
    T *out;       // [B*Np][D]
    int B, H, N, Np, D;
};

constexpr int ATT_KV = 64;  // keys per tile
constexpr float ATT_DEFER = 8.0f;  // margin of the deferred maximum, in the scores' (base-2 exponent) units

__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// KS = 2 (round 5, small grids -- the reference's one-frame forward is 120 workgroups on 256 CUs, each a chain of 19 key tiles of ~1.2 us): eight waves, the
// second four take the second half of the KEYS for the same 128 queries with a ring of their own, and the halves' (m, l, O) are merged through LDS at the end
// (the usual two-way merge of online softmax).  Another summation order than KS = 1: results agree to rounding, not bit for bit.
template <typename T, int KS>
__global__ __launch_bounds__(256 * KS) void attention_kernel(AttnParams<T> p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[KS * 2 * 2 * ATT_KV * 128];  // per key half: 2 stages x (K tile, V^T tile)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all & 3, kh = KS > 1 ? wave_all >> 2 : 0;  // query group of 32, key half
    unsigned char *lds = lds_all + kh * (2 * 2 * ATT_KV * 128);
    const int q_blocks = (p.Np + 127) / 128;
    const int qb = blockIdx.x % q_blocks;
    const int head = (blockIdx.x / q_blocks) % p.H;
    const int img = blockIdx.x / (q_blocks * p.H);
    const int lq = lane & 31, hh = lane >> 5;
    const int q_row = qb * 128 + wave * 32 + lq;                 // token index of this lane's query
    const int q_tok = min(q_row, p.Np - 1);
    const size_t row0 = (size_t)img * p.Np;
    const int ld = 2 * p.D;

    // Q as the B operand of S^T = K Q^T: lane holds Q[q][16 ks + 8 hh + j]
    vec<T, 8> qf[4];
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const vec<T, 8> *>(p.qk + (row0 + q_tok) * ld + head * 64 + ks * 16 + hh * 8);

    // staging of one K tile [64 keys][64 ch] and one V^T tile [64 ch][64 keys] by LDS-DMA: 8 + 8 groups of 8 rows x 128 B, a wave takes groups
    // wave and wave + 4 of each, with the GEMM's source-side chunk swizzle (slot = chunk ^ ((row >> 1) & 7)).  Round 5: the lane's four byte offsets are
    // loop constants and a tile's base is a scalar (Np % 64 == 0: no row of a tile is past the end) -- the address arithmetic was ~45 of a tile's ~200
    // vector instructions in a kernel the vector ALU bounds.
    const char *k_base = reinterpret_cast<const char *>(p.qk + row0 * ld + p.D + head * 64);
    const char *v_base = reinterpret_cast<const char *>(p.vT + ((size_t)img * p.H + head) * 64 * p.Np);
    unsigned k_off[2], v_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave + 4 * j) * 8 + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
        k_off[j] = (unsigned)row * (unsigned)ld * 2u + (unsigned)chunk * 16u;
        v_off[j] = (unsigned)row * (unsigned)p.Np * 2u + (unsigned)chunk * 16u;
    }
    auto issue_tile = [&](int t, int stage) {
        unsigned char *k_t = lds + stage * 2 * ATT_KV * 128 + wave * 1024, *v_t = k_t + ATT_KV * 128;
        const char *kp = k_base + (size_t)t * ((size_t)ATT_KV * ld * 2), *vp = v_base + (size_t)t * (ATT_KV * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const void *)(kp + k_off[j]), (__attribute__((address_space(3))) void *)(k_t + j * 4096), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const void *)(vp + v_off[j]), (__attribute__((address_space(3))) void *)(v_t + j * 4096), 16, 0, 0);
    };

    f32x16 oacc[2];
    for (int i = 0; i < 16; ++i) oacc[0][i] = oacc[1][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 neg_m;  // -m_run in every element (0 before the first tile)
    for (int i = 0; i < 16; ++i) neg_m[i] = 0.f;

    const int n_tiles = p.Np / ATT_KV;
    const int per_half = (n_tiles + KS - 1) / KS;  // every wave runs per_half iterations (the barriers are the workgroup's); the last half may have fewer tiles
    const int t_begin = kh * per_half, t_end = min(n_tiles, t_begin + per_half);
    if (KS == 1 || t_begin < t_end) issue_tile(t_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = 0; it < per_half; ++it) {
        const int t = t_begin + it;
        const int stage = it & 1;
        if (t + 1 < t_end) issue_tile(t + 1, stage ^ 1);  // that buffer was last read in the previous iteration, before its barrier
        if (KS == 1 || t < t_end) {
            const unsigned char *k_t = lds + stage * 2 * ATT_KV * 128, *v_t = k_t + ATT_KV * 128;
            // S^T[key][q] for 64 keys x 32 queries: rows (keys) in registers, query on the lane
            // q is pre-scaled, so the products are the softmax's base-2 exponents; the accumulators start at -m (the running maximum of
            // this lane's query), so they come out of the MFMAs already shifted: p = exp2(acc) -- no multiply-subtract per score
            // (-m sits in 16 registers of its own, rewritten only when the maximum moves, and is the C operand of each chain's first MFMA:
            // no per-tile initialisation of the 32 accumulators)
            const float m_used = it == 0 ? 0.f : m_run;
            f32x16 sacc[2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const vec<T, 8> kf = *reinterpret_cast<const vec<T, 8> *>(k_t + swz(kb * 32 + lq, ks * 2 + hh));
                    sacc[kb] = hive_mfma::mfma32(kf, qf[ks], ks == 0 ? neg_m : sacc[kb]);
                }
            // key row of register i: 32 kb + (i & 3) + 8 (i >> 2) + 4 hh.  Keys >= N exist only in the last tile.
            if (t == n_tiles - 1) {
                asm volatile("; pad keys: last tile only" ::: "memory");  // keeps this a branch (if-converted it is 31 v_cndmask in EVERY tile)
                const int key0 = t * ATT_KV + 4 * hh;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (key0 + kb * 32 + (i & 3) + 8 * (i >> 2) >= p.N) sacc[kb][i] = -INFINITY;
            }
            // v_max3_f32 by hand: fmaxf() makes the compiler canonicalise every MFMA result first (one v_max_f32 x, x, x per score:
            // 50 v_max + 8 v_max3 per tile where 16 v_max3 do; the softmax VALU work, not the MFMAs, bounds this kernel)
            float m_tile = max3_raw(sacc[0][0], sacc[1][0], sacc[0][1]);
            m_tile = max3_raw(m_tile, sacc[1][1], sacc[0][2]);
#pragma unroll
            for (int i = 2; i < 16; i += 2) {
                m_tile = max3_raw(m_tile, sacc[1][i], sacc[0][i + 1]);
                if (i + 2 < 16)
                    m_tile = max3_raw(m_tile, sacc[1][i + 1], sacc[0][i + 2]);
                else
                    m_tile = max3_raw(m_tile, sacc[1][i + 1], sacc[1][i + 1]);
            }
            m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));  // the tile's maximum RELATIVE to m_used
            // deferred maximum: the running maximum only moves when some query of the wave exceeds it by more than ATT_DEFER
            // (probabilities then stay below 2^8 -- harmless in f32 / bf16), so the rescale of the O accumulators and the re-shift of
            // the scores are skipped for almost every tile.  Any m gives the same softmax.
            if (it == 0 || __any(m_tile > ATT_DEFER)) {
                const float m_new = it == 0 ? m_tile : fmaxf(m_run, m_used + m_tile);
                const float delta = m_new - m_used;
                const float alpha = it == 0 ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    oacc[0][i] *= alpha;
                    oacc[1][i] *= alpha;
                    sacc[0][i] -= delta;
                    sacc[1][i] -= delta;
                    neg_m[i] = -m_new;
                }
            }
            vec<T, 8> pf[2][2];
            float l_tile = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float e = __builtin_amdgcn_exp2f(sacc[kb][i]);
                    l_tile += e;
                    pf[kb][i >> 3][i & 7] = (T)e;
                }
            l_run += l_tile;
            // O^T[ch][q] += V^T[ch][key] P^T[key][q]; k index j of half hh <-> key 32 kb + 16 s + 8 (j >> 2) + 4 hh + (j & 3) (see vt_slot)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        // the lane's 8 keys {16 s + 4 hh + 0..3, + 8..11} of key block kb sit in slots 16 s + 8 hh .. + 7 of the stored
                        // (quad-swapped) order: chunk 4 kb + 2 s + hh of the row, one 16-byte read through the tile's swizzle
                        const vec<T, 8> vf = *reinterpret_cast<const vec<T, 8> *>(v_t + swz(db * 32 + lq, 4 * kb + 2 * s + hh));
                        oacc[db] = hive_mfma::mfma32(vf, pf[kb][s], oacc[db]);
                    }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of tile t+1 have landed
        __syncthreads();
    }
    float l_tot = l_run + __shfl_xor(l_run, 32);
    if constexpr (KS > 1) {
        // the second key half hands over (O, m, l) -- 34 floats per lane, behind the four 4 KiB turn-around blocks below (the stages are free: the loop's last
        // barrier is behind every wave) -- and leaves; the first merges: m = max(m0, m1), O = O0 2^(m0 - m) + O1 2^(m1 - m), l likewise.  Every half has at least
        // one tile with a real key (n_tiles >= 4 and Np - N < 64: checked on the host), so both maxima are finite.
        float *xch = reinterpret_cast<float *>(lds_all + 4 * 4096) + wave * (34 * 64) + lane;
        if (kh == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                xch[i * 64] = oacc[0][i];
                xch[(16 + i) * 64] = oacc[1][i];
            }
            xch[32 * 64] = m_run;
            xch[33 * 64] = l_tot;
        }
        __syncthreads();
        if (kh == 1) return;  // (whole waves; nothing below synchronises the workgroup)
        const float m1 = xch[32 * 64], l1 = xch[33 * 64];
        const float m = fmaxf(m_run, m1);
        const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
        l_tot = l_tot * a0 + l1 * a1;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            oacc[0][i] = oacc[0][i] * a0 + xch[i * 64] * a1;
            oacc[1][i] = oacc[1][i] * a0 + xch[(16 + i) * 64] * a1;
        }
    }
    const float inv = 1.0f / l_tot;
    // O leaves through LDS (the K / V stages are free: the loop's last barrier is behind every wave): in the accumulator layout a lane
    // owns 4 consecutive channels of its query, so a direct store is 32 rows x 16 bytes per instruction (8 per wave, store-issue
    // bound); turned around in the wave's private 4 KiB -- [query][64 channels], 16-byte chunk c of row r at slot c ^ (r & 7) -- the
    // wave stores 4 x (8 rows x 128 contiguous bytes).
    unsigned char *ot = lds_all + wave * 4096;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            vec<T, 4> ov;
            for (int j = 0; j < 4; ++j) ov[j] = (T)(oacc[db][4 * g + j] * inv);
            *reinterpret_cast<vec<T, 4> *>(ot + lq * 128 + (((4 * db + g) ^ (lq & 7)) << 4) + 8 * hh) = ov;  // channels 32 db + 8 g + 4 hh ..+3
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * j + (lane >> 3), c = lane & 7;
        const vec<T, 8> row = *reinterpret_cast<const vec<T, 8> *>(ot + r * 128 + ((c ^ (r & 7)) << 4));
        const int q = qb * 128 + wave * 32 + r;
        if (q < p.Np) *reinterpret_cast<vec<T, 8> *>(p.out + (row0 + q) * p.D + head * 64 + 8 * c) = row;
    }
}

// ------------------------------------------------------------------------------------------------
// token (un)padding: [B][N][D] <-> [B][Np][D], pad rows zero
typedef unsigned short half_bits;  // either 16-bit type: copied, never interpreted
__global__ __launch_bounds__(256) void pad_tokens_kernel(const half_bits *__restrict__ src, half_bits *__restrict__ dst, int B, int N, int Np,
                                                         int D, int to_padded) {
    const size_t chunks_per_row = D / 8;
    const size_t total = (size_t)B * Np * chunks_per_row;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / chunks_per_row, c = i % chunks_per_row;
        const int img = (int)(row / Np), tok = (int)(row % Np);
        if (to_padded) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (tok < N) v = *reinterpret_cast<const uint4 *>(src + ((size_t)img * N + tok) * D + c * 8);
            *reinterpret_cast<uint4 *>(dst + row * D + c * 8) = v;
        } else if (tok < N) {
            *reinterpret_cast<uint4 *>(dst + ((size_t)img * N + tok) * D + c * 8) = *reinterpret_cast<const uint4 *>(src + row * D + c * 8);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// DPT head tail, fused and in f32: the last 1x1 convolution (C -> 1) of the depth head on the bf16 / f16
// channels-last feature map, ReLU (non_negative), depth = 1 / max(scale * x + shift, 1e-8) (invert), and
// optionally the uint16-millimetre hand-off of dataset_adaptors.py:1432-1433 + io.py:1032-1039.
struct HeadTailParams {
    float w[64];
    float pre_bias[64];  // bias of the preceding convolution (all zero when it was already applied)
    int pre_relu;        // ReLU between that convolution and this one (head: conv 128->32, ReLU, conv 32->1)
    float bias, scale, shift;
    int C, non_negative, invert;
    long long n_px;
    float depth_scale, max_depth;  // quantised outputs
};

template <typename T>
__global__ __launch_bounds__(256) void head_tail_kernel(const T *__restrict__ feat, HeadTailParams p, float *__restrict__ out_depth,
                                                        uint16_t *__restrict__ out_mm, float *__restrict__ out_m) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n_px) return;
    const T *f = feat + i * p.C;
    float acc = p.bias;
    for (int c = 0; c < p.C; c += 8) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(f + c);
        const T *v = reinterpret_cast<const T *>(&raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = (float)v[j] + p.pre_bias[c + j];
            if (p.pre_relu) x = fmaxf(x, 0.0f);
            acc += x * p.w[c + j];
        }
    }
    if (p.non_negative) acc = fmaxf(acc, 0.0f);
    float depth = acc;
    if (p.invert) depth = 1.0f / fmaxf(p.scale * acc + p.shift, 1e-8f);
    if (out_depth) out_depth[i] = depth;
    if (out_mm || out_m) {
        const uint16_t mm = (uint16_t)(int)fminf(fmaxf(depth * 1000.0f, 0.0f), 65535.0f);
        float m = p.depth_scale * (float)mm;
        if (m > p.max_depth) m = 0.0f;
        if (out_mm) out_mm[i] = mm;
        if (out_m) out_m[i] = m;
    }
}

// uint8 RGB [n] -> network input, channels-last.  The reference evaluates ((x / 255.0) - mean) / std in float64
// (numpy), casts to float32 (PrepareForNet) and then to the 16-bit network type (dataset_adaptors.py:1407-1417);
// with only 256 possible inputs that is a table, built on the host in float64.
struct PreprocessLut {
    float v[256];
};

template <typename T>
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t *__restrict__ rgb, long long n, PreprocessLut lut,
                                                         T *__restrict__ out) {
    __shared__ float tab[256];
    tab[threadIdx.x] = lut.v[threadIdx.x];
    __syncthreads();
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        const uchar4 v = *reinterpret_cast<const uchar4 *>(rgb + i);
        T o[4] = {(T)tab[v.x], (T)tab[v.y], (T)tab[v.z], (T)tab[v.w]};
        *reinterpret_cast<uint2 *>(out + i) = *reinterpret_cast<const uint2 *>(o);
    } else {
        for (long long j = i; j < n; ++j) out[j] = (T)tab[rgb[j]];
    }
}

// ------------------------------------------------------------------------------------------------
struct hive_vit {
    hive_ctx *ctx = nullptr;
    int dtype = HIVE_BF16;
    int depth = 0, dim = 0, heads = 0, mlp = 0;
    float eps = 1e-6f;
    std::vector<hive_vit_block_weights> blocks;
    // workspace
    void *ws = nullptr;
    size_t ws_bytes = 0;
    // LayerNorm folded into the q|k|v and fc1 GEMMs (hive_vit_forward): per block the weights gamma o W in the network's 16-bit type and the
    // f32 columns c1 = sum_k W', c2 = sum_k beta W + b; one allocation
    struct Folded {
        const void *qkv_w = nullptr, *fc1_w = nullptr;
        const float *qkv_c1 = nullptr, *qkv_c2 = nullptr, *fc1_c1 = nullptr, *fc1_c2 = nullptr;
    };
    std::vector<Folded> folded;
    void *fold_mem = nullptr;
};

template <typename T>
static int launch_layernorm(hive_ctx *ctx, const void *x, const float *g, const float *b, void *out, int M, int D, float eps) {
    const dim3 grid((M + 3) / 4), block(256);
    const T *xi = (const T *)x;
    T *xo = (T *)out;
    switch (D / 256) {
        case 1: hipLaunchKernelGGL((layernorm_kernel<T, 1>), grid, block, 0, ctx->stream, xi, g, b, xo, M, eps); break;
        case 2: hipLaunchKernelGGL((layernorm_kernel<T, 2>), grid, block, 0, ctx->stream, xi, g, b, xo, M, eps); break;
        case 3: hipLaunchKernelGGL((layernorm_kernel<T, 3>), grid, block, 0, ctx->stream, xi, g, b, xo, M, eps); break;
        case 4: hipLaunchKernelGGL((layernorm_kernel<T, 4>), grid, block, 0, ctx->stream, xi, g, b, xo, M, eps); break;
        default: return hive_fail(ctx, HIVE_ERR_INVALID, "layernorm: unsupported D %d", D);
    }
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

constexpr int GEMM_TM = 128, GEMM_NST = 2, GEMM_NST_DEEP = 4;
constexpr size_t gemm_lds(int nst) { return (size_t)nst * (GEMM_TM / 8 + 16) * 1024 + (GEMM_TM / 32) * 4096; }  // the stages + 4 KiB per wave for the epilogue
constexpr size_t GEMM_LDS = gemm_lds(GEMM_NST), GEMM_LDS_DEEP = gemm_lds(GEMM_NST_DEEP);  // 80 KiB: two workgroups per CU; 144 KiB: one
constexpr int GEMM256_LDS = 2 * T256_STAGE + hive_mfma::STAGED_ROWS_LDS;

template <typename T>
static int launch_gemm256(hive_ctx *ctx, int epi, const GemmParams<T> &p) {
    const dim3 grid((unsigned)(((p.M + T256 - 1) / T256) * (p.N / T256))), block(512);
    const size_t lds_bytes = GEMM256_LDS;
    // persistent workgroups by default (+2-4 % where a workgroup gets more than one tile: the next tile's first fill is hidden);
    // HIVE_GEMM_PERSIST=0 selects the one-tile-per-workgroup kernel, the one the phase clocks of `make stamps` instrument
    static const char *persist = getenv("HIVE_GEMM_PERSIST");
    if (!(persist && persist[0] == '0')) {
        const dim3 pgrid((unsigned)std::min<long long>(((long long)grid.x + 7) / 8 * 8, (long long)ctx->num_cus / 8 * 8));
        switch (epi) {
            case EPI_BIAS: hipLaunchKernelGGL((gemm256p_kernel<T, EPI_BIAS>), pgrid, block, lds_bytes, ctx->stream, p); break;
            case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm256p_kernel<T, EPI_BIAS_GELU>), pgrid, block, lds_bytes, ctx->stream, p); break;
            case EPI_BIAS_RESIDUAL: hipLaunchKernelGGL((gemm256p_kernel<T, EPI_BIAS_RESIDUAL>), pgrid, block, lds_bytes, ctx->stream, p); break;
            case EPI_QKV: hipLaunchKernelGGL((gemm256p_kernel<T, EPI_QKV>), pgrid, block, lds_bytes, ctx->stream, p); break;
            default: return hive_fail(ctx, HIVE_ERR_INVALID, "gemm: unknown epilogue %d", epi);
        }
        HIVE_CHECK_HIP(ctx, hipGetLastError());
        return HIVE_OK;
    }
    switch (epi) {
        case EPI_BIAS: hipLaunchKernelGGL((gemm256_kernel<T, EPI_BIAS>), grid, block, lds_bytes, ctx->stream, p); break;
        case EPI_BIAS_GELU: hipLaunchKernelGGL((gemm256_kernel<T, EPI_BIAS_GELU>), grid, block, lds_bytes, ctx->stream, p); break;
        case EPI_BIAS_RESIDUAL: hipLaunchKernelGGL((gemm256_kernel<T, EPI_BIAS_RESIDUAL>), grid, block, lds_bytes, ctx->stream, p); break;
        default: return hive_fail(ctx, HIVE_ERR_INVALID, "gemm: unknown epilogue %d", epi);
    }
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

template <typename T>
static int launch_gemm(hive_ctx *ctx, int epi, const GemmParams<T> &p) {
    // 256 x 256 tiles (half the operand bytes per flop: 690-970 TFLOP/s against 510-680 for the 128-row tiles on the ViT shapes)
    // wherever they fill the chip: one workgroup per CU, so what matters is how full the last round of tiles is -- at M = 29184,
    // N = 768 there are 342 tiles for 256 CUs (67 %), at M = 19456 228 (89 %: 970 TFLOP/s), at M = 9728 114 (44 %); below 60 % the persistent
    // 128-row tiles (two workgroups per CU, 1368 tiles for 512 slots) win.
    const char *force = getenv("HIVE_GEMM_TILE");  // "256" / "128": tuning override (read per call: tools/probe_gemm_tiles.py switches it inside one process)
    const long long tiles256 = (long long)((p.M + T256 - 1) / T256) * (p.N / T256);
    const long long rounds = (tiles256 + ctx->num_cus - 1) / ctx->num_cus;
    const bool fills = tiles256 * 5 >= rounds * ctx->num_cus * 3;  // >= 60 % of the CU slots of its rounds (67 %: 635 / 763 vs 612 / 705 TFLOP/s for proj / fc2 at M = 29184; 44 %: 655 vs 858)
    static const char *persist_env = getenv("HIVE_GEMM_PERSIST");
    const bool one_tile_kernel = persist_env && persist_env[0] == '0';  // that kernel has no v^T epilogue
    if ((epi != EPI_QKV || !one_tile_kernel) && p.N % T256 == 0 && ((force && force[0] == '2') || (!force && fills))) return launch_gemm256<T>(ctx, epi, p);
    // persistent workgroups: two per CU (64 KiB of LDS each), a multiple of 8 so that every XCD gets the same number
    const long long tiles = (long long)((p.M + GEMM_TM - 1) / GEMM_TM) * (p.N / BN);
    // (Round 4, measured and taken out: 64-row tiles (two waves, 56 KiB of LDS) where fewer 128-row tiles than CUs exist -- the reference's literal
    // loop is batch 1, M = 1216: 60 tiles for proj / fc2.  5.38 vs 5.35 ms per one-frame forward: at one tile per CU a GEMM is as long as its K loop
    // (12-48 steps of ~0.8-1.5 us), whatever the tile's height; small batches want the K loop split, not the tile.)
    // Split-K for long K loops on few tiles (mfma_pipe.hpp splitk_combine / splitk_ways): the reference's literal loop is batch 1 -- M = 1216: 60 tiles
    // for fc2, whose K loop of 48 steps WAS that GEMM's duration.  HIVE_SPLITK=0 switches it off, n forces n ways.
    GemmParams<T> q = p;
    const long long slots = (long long)(2 * ctx->num_cus) / 8 * 8;
    const char *sk_env = getenv("HIVE_SPLITK");
    q.split_k = sk_env ? std::max(1, std::min(atoi(sk_env), p.K / BK)) : hive_mfma::splitk_ways(tiles, p.K / BK, ctx->num_cus);
    if (tiles > HIVE_SPLITK_TILES || ctx->deterministic) q.split_k = 1;
    if (q.split_k > 1) {
        ++ctx->n_splitk_launches;
        void *ws = nullptr;
        int rc = hive_splitk_workspace(ctx, (size_t)tiles * q.split_k * GEMM_TM * BN * sizeof(float), &ws, &q.sk_count);
        if (rc) return rc;
        q.sk_ws = reinterpret_cast<hive_mfma::f32x4 *>(ws);
    }
    const long long items = tiles * q.split_k;
    // the deep ring (one workgroup per CU) where the items fit the CUs in one round anyway; HIVE_GEMM_RING=2 / 4 forces a depth
    const char *ring_env = getenv("HIVE_GEMM_RING");
    const bool deep = ring_env ? ring_env[0] == '4' : items <= ctx->num_cus;
    if (deep) ++ctx->n_deep_ring_launches;
    const dim3 grid((unsigned)std::min<long long>((items + 7) / 8 * 8, deep ? (long long)ctx->num_cus / 8 * 8 : slots)), block(GEMM_TM * 2);
#define HIVE_GEMM_CASE(EPI_)                                                                                                          \
    case EPI_:                                                                                                                        \
        if (deep)                                                                                                                     \
            hipLaunchKernelGGL((gemm_kernel<T, EPI_, GEMM_TM, GEMM_NST_DEEP>), grid, block, GEMM_LDS_DEEP, ctx->stream, q);            \
        else                                                                                                                          \
            hipLaunchKernelGGL((gemm_kernel<T, EPI_, GEMM_TM, GEMM_NST>), grid, block, GEMM_LDS, ctx->stream, q);                      \
        break
    switch (epi) {
        HIVE_GEMM_CASE(EPI_BIAS);
        HIVE_GEMM_CASE(EPI_BIAS_GELU);
        HIVE_GEMM_CASE(EPI_BIAS_RESIDUAL);
        HIVE_GEMM_CASE(EPI_QKV);
#undef HIVE_GEMM_CASE
        default: return hive_fail(ctx, HIVE_ERR_INVALID, "gemm: unknown epilogue %d", epi);
    }
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

template <typename T>
static int ensure_gemm_attrs(hive_ctx *ctx) {
    static bool set[64] = {false};
    if (ctx->device < 64 && set[ctx->device]) return HIVE_OK;
#define HIVE_GEMM_ATTR(EPI_)                                                                                                                                  \
    HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)gemm_kernel<T, EPI_, GEMM_TM, GEMM_NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS)); \
    HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)gemm_kernel<T, EPI_, GEMM_TM, GEMM_NST_DEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS_DEEP)); \
    HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)gemm256p_kernel<T, EPI_>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM256_LDS))
    HIVE_GEMM_ATTR(EPI_BIAS);
    HIVE_GEMM_ATTR(EPI_BIAS_GELU);
    HIVE_GEMM_ATTR(EPI_BIAS_RESIDUAL);
    HIVE_GEMM_ATTR(EPI_QKV);
#undef HIVE_GEMM_ATTR
    HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)gemm_kernel<T, EPI_QKV_ALL, GEMM_TM, GEMM_NST_DEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS_DEEP));
    HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)gemm_kernel<T, EPI_QKV_ALL, GEMM_TM, GEMM_NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS));
    HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)gemm256_kernel<T, EPI_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM256_LDS));
    HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)gemm256_kernel<T, EPI_BIAS_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM256_LDS));
    HIVE_CHECK_HIP(ctx, hipFuncSetAttribute((const void *)gemm256_kernel<T, EPI_BIAS_RESIDUAL>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM256_LDS));
    if (ctx->device < 64) set[ctx->device] = true;
    return HIVE_OK;
}

static inline int pad64(int n) { return (n + 63) / 64 * 64; }

struct LnFold {  // LayerNorm folded into a GEMM: what the consumer reads (stats, c1) or the producer leaves (partial)
    const float *stats = nullptr, *c1 = nullptr;
    float *partial = nullptr;
};

template <typename T>
static int linear_t(hive_ctx *ctx, const void *A, const void *W, const float *bias, const void *residual, void *C, int M, int N, int K, int epilogue,
                    const LnFold &ln = LnFold()) {
    int rc = ensure_gemm_attrs<T>(ctx);
    if (rc) return rc;
    GemmParams<T> p{};
    p.A = (const T *)A;
    p.W = (const T *)W;
    p.bias = bias;
    p.residual = (const T *)residual;
    p.C = (T *)C;
    p.M = M;
    p.N = N;
    p.K = K;
    p.ldc = N;
    p.ln_stats = ln.stats;
    p.ln_c1 = ln.c1;
    p.ln_partial = ln.partial;
    return launch_gemm<T>(ctx, epilogue, p);
}

template <typename T>
static int qkv_t(hive_ctx *ctx, const void *x, const void *W, const float *bias, void *qk, void *vT, int B, int Np, int D, int H, const LnFold &ln = LnFold()) {
    int rc = ensure_gemm_attrs<T>(ctx);
    if (rc) return rc;
    // q | k columns: plain bias epilogue into qk [M][2D]
    GemmParams<T> p{};
    p.A = (const T *)x;
    p.W = (const T *)W;
    p.bias = bias;
    p.C = (T *)qk;
    p.M = B * Np;
    p.N = 2 * D;
    p.K = D;
    p.ldc = 2 * D;
    p.q_cols = D;  // q leaves the GEMM in the softmax's base-2 exponent units: (x Wq + b) * head_dim^-0.5 * log2(e), rounded once
    p.q_scale = 0.125f * 1.44269504088896340736f;
    p.ln_stats = ln.stats;
    p.ln_c1 = ln.c1;
    // Round 5, small batches (the reference's literal loop is one frame per forward): where ALL 3 D columns' 128 x 128 tiles fit the CUs in one round, q | k and v^T are ONE launch
    // of the four-stage-ring kernel -- 180 tiles at one frame instead of 120 + 60 in two dependent launches of ~14 us each; a tile's arithmetic is what it was (same K order, same
    // epilogue), so the results are bit-identical.  HIVE_QKV_MERGE=0 keeps the two launches.
    {
        const long long tiles_all = (long long)((p.M + GEMM_TM - 1) / GEMM_TM) * (3 * D / BN);
        const char *merge_env = getenv("HIVE_QKV_MERGE");
        const long long merge_max = merge_env && merge_env[0] == '2' ? (long long)(2 * ctx->num_cus) / 8 * 8 : ctx->num_cus;  // ("2": also where the tiles fit two workgroups per CU -- measured below)
        if (!(merge_env && merge_env[0] == '0') && tiles_all <= merge_max && !getenv("HIVE_GEMM_TILE") && !getenv("HIVE_GEMM_RING")) {
            GemmParams<T> q = p;
            q.N = 3 * D;
            q.n_split = 2 * D;
            q.vT = (T *)vT;
            q.Np = Np;
            q.H = H;
            const bool deep = tiles_all <= ctx->num_cus;
            if (deep) ++ctx->n_deep_ring_launches;
            const dim3 grid((unsigned)((tiles_all + 7) / 8 * 8)), block(GEMM_TM * 2);
            if (deep)
                hipLaunchKernelGGL((gemm_kernel<T, EPI_QKV_ALL, GEMM_TM, GEMM_NST_DEEP>), grid, block, GEMM_LDS_DEEP, ctx->stream, q);
            else
                hipLaunchKernelGGL((gemm_kernel<T, EPI_QKV_ALL, GEMM_TM, GEMM_NST>), grid, block, GEMM_LDS, ctx->stream, q);
            HIVE_CHECK_HIP(ctx, hipGetLastError());
            return HIVE_OK;
        }
    }
    if ((rc = launch_gemm<T>(ctx, EPI_BIAS, p))) return rc;
    // v columns: transposed store into vT [B][H][64][Np]
    GemmParams<T> pv{};
    pv.A = (const T *)x;
    pv.W = (const T *)W + (size_t)2 * D * D;
    pv.bias = bias + 2 * D;
    pv.vT = (T *)vT;
    pv.M = B * Np;
    pv.N = D;
    pv.K = D;
    pv.ldc = D;
    pv.Np = Np;
    pv.H = H;
    pv.ln_stats = ln.stats;
    pv.ln_c1 = ln.c1 ? ln.c1 + 2 * D : nullptr;
    return launch_gemm<T>(ctx, EPI_QKV, pv);
}

template <typename T>
static int attention_t(hive_ctx *ctx, const void *qk, const void *vT, void *out, int B, int N, int Np, int D, int H) {
    AttnParams<T> p{};
    p.qk = (const T *)qk;
    p.vT = (const T *)vT;
    p.out = (T *)out;
    p.B = B;
    p.H = H;
    p.N = N;
    p.Np = Np;
    p.D = D;
    const int q_blocks = (p.Np + 127) / 128;
    // the keys split two ways inside the workgroup where the grid leaves a CU one workgroup at most (HIVE_ATT_KSPLIT=0 / 1 overrides: read per call)
    const long long grid = (long long)p.B * p.H * q_blocks;
    const char *ks_env = getenv("HIVE_ATT_KSPLIT");
    const bool split = p.Np / ATT_KV >= 4 && (ks_env ? ks_env[0] == '1' : (grid <= ctx->num_cus && !ctx->deterministic));  // (deterministic: no path whose use depends on the batch size)
    if (split)
        hipLaunchKernelGGL((attention_kernel<T, 2>), dim3((unsigned)grid), dim3(512), 0, ctx->stream, p);
    else
        hipLaunchKernelGGL((attention_kernel<T, 1>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, p);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

#define HIVE_REQUIRE_16BIT(ctx, dtype, what) HIVE_REQUIRE(ctx, (dtype) == HIVE_BF16 || (dtype) == HIVE_F16, what ": dtype must be HIVE_F16 or HIVE_BF16, got %d", (int)(dtype))

// (Re)build the private copies hive_vit_forward reads instead of the caller's tensors: W' = gamma o W for qkv / fc1 and the two constant rows c1, c2 of the folded
// LayerNorm.  Launched on the context's stream: ordered behind whatever wrote the caller's weights there, in front of the next forward.
static void fold_layernorm_weights(hive_vit *v) {
    hive_ctx *ctx = v->ctx;
    const int dim = v->dim, mlp_dim = v->mlp, depth = v->depth;
    const size_t esz = sizeof(half_bits), rows = (size_t)3 * dim + mlp_dim;
    const size_t w_bytes = (rows * dim * esz + 255) & ~(size_t)255, c_bytes = (2 * rows * sizeof(float) + 255) & ~(size_t)255;
    for (int i = 0; i < depth; ++i) {
        char *base = (char *)v->fold_mem + (size_t)i * (w_bytes + c_bytes);
        float *c = (float *)(base + w_bytes);
        hive_vit::Folded &f = v->folded[i];
        f.qkv_w = base;
        f.fc1_w = base + (size_t)3 * dim * dim * esz;
        f.qkv_c1 = c, f.qkv_c2 = c + 3 * dim, f.fc1_c1 = c + 6 * dim, f.fc1_c2 = c + 6 * dim + mlp_dim;
        const hive_vit_block_weights &b = v->blocks[i];
#define HIVE_FOLD(T_)                                                                                                                                   \
    hipLaunchKernelGGL(ln_fold_weights_kernel<T_>, dim3(3 * dim), dim3(256), 0, ctx->stream, (const T_ *)b.qkv_w, (const float *)b.qkv_b, (const float *)b.ln1_g, \
                       (const float *)b.ln1_b, dim, (T_ *)f.qkv_w, (float *)f.qkv_c1, (float *)f.qkv_c2);                                                \
    hipLaunchKernelGGL(ln_fold_weights_kernel<T_>, dim3(mlp_dim), dim3(256), 0, ctx->stream, (const T_ *)b.fc1_w, (const float *)b.fc1_b, (const float *)b.ln2_g, \
                       (const float *)b.ln2_b, dim, (T_ *)f.fc1_w, (float *)f.fc1_c1, (float *)f.fc1_c2)
        if (v->dtype == HIVE_BF16) {
            HIVE_FOLD(__bf16);
        } else {
            HIVE_FOLD(_Float16);
        }
#undef HIVE_FOLD
    }
}

extern "C" {

int hive_vit_layernorm(hive_ctx *ctx, const void *x, int dtype, const float *gamma, const float *beta, void *out, int M, int D, float eps) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, x && gamma && beta && out, "layernorm: NULL argument");
    HIVE_REQUIRE_16BIT(ctx, dtype, "layernorm");
    HIVE_REQUIRE(ctx, M > 0 && D > 0 && D % 256 == 0 && D <= 1024, "layernorm: D must be a multiple of 256 and <= 1024, got %d", D);
    return dtype == HIVE_BF16 ? launch_layernorm<__bf16>(ctx, x, gamma, beta, out, M, D, eps) : launch_layernorm<_Float16>(ctx, x, gamma, beta, out, M, D, eps);
}

int hive_vit_linear(hive_ctx *ctx, const void *A, int dtype, const void *W, const float *bias, const void *residual, void *C, int M, int N,
                    int K, int epilogue) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, A && W && bias && C, "linear: NULL argument");
    HIVE_REQUIRE_16BIT(ctx, dtype, "linear");
    HIVE_REQUIRE(ctx, M > 0 && N > 0 && K > 0 && N % BN == 0 && K % BK == 0, "linear: need N %% 128 == 0 and K %% 64 == 0 (M=%d N=%d K=%d)", M, N, K);
    HIVE_REQUIRE(ctx, epilogue == EPI_BIAS || epilogue == EPI_BIAS_GELU || (epilogue == EPI_BIAS_RESIDUAL && residual),
                 "linear: bad epilogue %d", epilogue);
    return dtype == HIVE_BF16 ? linear_t<__bf16>(ctx, A, W, bias, residual, C, M, N, K, epilogue) : linear_t<_Float16>(ctx, A, W, bias, residual, C, M, N, K, epilogue);
}

int hive_vit_qkv(hive_ctx *ctx, const void *x, int dtype, const void *W, const float *bias, void *qk, void *vT, int B, int Np, int D, int H) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, x && W && bias && qk && vT, "qkv: NULL argument");
    HIVE_REQUIRE_16BIT(ctx, dtype, "qkv");
    HIVE_REQUIRE(ctx, B > 0 && Np > 0 && Np % 64 == 0 && D == H * 64 && D % 128 == 0, "qkv: need Np %% 64 == 0, D == 64 H, D %% 128 == 0");
    return dtype == HIVE_BF16 ? qkv_t<__bf16>(ctx, x, W, bias, qk, vT, B, Np, D, H) : qkv_t<_Float16>(ctx, x, W, bias, qk, vT, B, Np, D, H);
}

int hive_vit_attention(hive_ctx *ctx, const void *qk, int dtype, const void *vT, void *out, int B, int N, int Np, int D, int H) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, qk && vT && out, "attention: NULL argument");
    HIVE_REQUIRE_16BIT(ctx, dtype, "attention");
    HIVE_REQUIRE(ctx, B > 0 && N > 0 && N <= Np && Np % 64 == 0 && Np - N < 64 && D == H * 64, "attention: need Np - 64 < N <= Np, Np %% 64 == 0, head dim 64");
    // the kernel masks pad keys in the LAST 64-key tile only: Np must be N rounded up to a multiple of 64
    HIVE_REQUIRE(ctx, Np - N < 64, "attention: Np (%d) must be N (%d) rounded up to a multiple of 64", Np, N);
    return dtype == HIVE_BF16 ? attention_t<__bf16>(ctx, qk, vT, out, B, N, Np, D, H) : attention_t<_Float16>(ctx, qk, vT, out, B, N, Np, D, H);
}

int hive_dpt_preprocess(hive_ctx *ctx, const uint8_t *d_rgb, int64_t n_values, float mean, float std, int dtype, void *d_out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_rgb && d_out && n_values > 0 && std != 0.f, "dpt_preprocess: bad arguments");
    HIVE_REQUIRE(ctx, ((uintptr_t)d_rgb % 4 == 0) && ((uintptr_t)d_out % 8 == 0), "dpt_preprocess: unaligned buffers");
    const dim3 grid((unsigned)((n_values / 4 + 256) / 256));
    PreprocessLut lut;
    for (int i = 0; i < 256; ++i) lut.v[i] = (float)(((double)i / 255.0 - (double)mean) / (double)std);
    if (dtype == HIVE_BF16)
        hipLaunchKernelGGL(preprocess_kernel<__bf16>, grid, dim3(256), 0, ctx->stream, d_rgb, (long long)n_values, lut, (__bf16 *)d_out);
    else if (dtype == HIVE_F16)
        hipLaunchKernelGGL(preprocess_kernel<_Float16>, grid, dim3(256), 0, ctx->stream, d_rgb, (long long)n_values, lut, (_Float16 *)d_out);
    else
        return hive_fail(ctx, HIVE_ERR_INVALID, "dpt_preprocess: dtype must be HIVE_F16 or HIVE_BF16");
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

int hive_dpt_head_tail(hive_ctx *ctx, const void *d_feat, int dtype, int64_t n_px, int C, const float *h_pre_bias, int pre_relu,
                       const float *h_weight, float bias, int non_negative, int invert, float scale, float shift, float *d_depth,
                       float depth_scale, float max_depth, uint16_t *d_out_mm, float *d_out_m) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, d_feat && h_weight && n_px > 0, "dpt_head_tail: bad arguments");
    HIVE_REQUIRE(ctx, C > 0 && C <= 64 && C % 8 == 0, "dpt_head_tail: C must be a multiple of 8 and <= 64, got %d", C);
    HIVE_REQUIRE(ctx, d_depth || d_out_mm || d_out_m, "dpt_head_tail: no output requested");
    HeadTailParams p{};
    memcpy(p.w, h_weight, sizeof(float) * C);
    if (h_pre_bias) memcpy(p.pre_bias, h_pre_bias, sizeof(float) * C);
    p.pre_relu = pre_relu;
    p.bias = bias;
    p.scale = scale;
    p.shift = shift;
    p.C = C;
    p.non_negative = non_negative;
    p.invert = invert;
    p.n_px = n_px;
    p.depth_scale = depth_scale;
    p.max_depth = max_depth;
    const dim3 grid((unsigned)((n_px + 255) / 256));
    if (dtype == HIVE_BF16)
        hipLaunchKernelGGL(head_tail_kernel<__bf16>, grid, dim3(256), 0, ctx->stream, (const __bf16 *)d_feat, p, d_depth, d_out_mm, d_out_m);
    else if (dtype == HIVE_F16)
        hipLaunchKernelGGL(head_tail_kernel<_Float16>, grid, dim3(256), 0, ctx->stream, (const _Float16 *)d_feat, p, d_depth, d_out_mm, d_out_m);
    else
        return hive_fail(ctx, HIVE_ERR_INVALID, "dpt_head_tail: dtype must be HIVE_F16 or HIVE_BF16");
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    return HIVE_OK;
}

int hive_vit_create(hive_ctx *ctx, int dtype, int depth, int dim, int heads, int mlp_dim, float ln_eps, const hive_vit_block_weights *blocks,
                    hive_vit **out) {
    HIVE_ENTER(ctx);
    if (!ctx) return hive_fail(nullptr, HIVE_ERR_INVALID, "ctx is NULL");
    HIVE_REQUIRE(ctx, out && blocks && depth > 0, "vit_create: NULL argument");
    HIVE_REQUIRE_16BIT(ctx, dtype, "vit_create");
    HIVE_REQUIRE(ctx, dim == heads * 64 && dim % 256 == 0 && dim <= 1024 && mlp_dim % 128 == 0,
                 "vit_create: need head dim 64, dim %% 256 == 0, dim <= 1024, mlp %% 128 == 0 (dim=%d heads=%d mlp=%d)", dim, heads, mlp_dim);
    for (int i = 0; i < depth; ++i) {
        const hive_vit_block_weights &b = blocks[i];
        HIVE_REQUIRE(ctx, b.ln1_g && b.ln1_b && b.qkv_w && b.qkv_b && b.proj_w && b.proj_b && b.ln2_g && b.ln2_b && b.fc1_w && b.fc1_b &&
                              b.fc2_w && b.fc2_b, "vit_create: block %d has a NULL weight pointer", i);
    }
    hive_vit *v = new hive_vit();
    v->ctx = ctx;
    v->dtype = dtype;
    v->depth = depth;
    v->dim = dim;
    v->heads = heads;
    v->mlp = mlp_dim;
    v->eps = ln_eps;
    v->blocks.assign(blocks, blocks + depth);
    // folded weights: per block (3 D + mlp) rows of D 16-bit values + 2 (3 D + mlp) floats
    {
        const size_t esz = sizeof(half_bits), rows = (size_t)3 * dim + mlp_dim;
        const size_t w_bytes = (rows * dim * esz + 255) & ~(size_t)255, c_bytes = (2 * rows * sizeof(float) + 255) & ~(size_t)255;
        hipError_t e = hipMalloc(&v->fold_mem, (size_t)depth * (w_bytes + c_bytes));
        if (e != hipSuccess) {
            int rc = hive_fail(ctx, HIVE_ERR_NOMEM, "vit_create: allocating the folded LayerNorm weights failed: %s", hipGetErrorString(e));
            delete v;
            return rc;
        }
        v->folded.resize(depth);
        fold_layernorm_weights(v);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the caller may free or overwrite its weight tensors' temporaries after create
        if (e != hipSuccess) {
            int rc = hive_fail(ctx, HIVE_ERR_DEVICE, "vit_create: folding the LayerNorm weights failed: %s", hipGetErrorString(e));
            (void)hipFree(v->fold_mem);
            delete v;
            return rc;
        }
    }
    *out = v;
    return HIVE_OK;
}

int hive_vit_weights_modified(hive_vit *v) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vit is NULL");
    fold_layernorm_weights(v);
    HIVE_CHECK_HIP(v->ctx, hipGetLastError());
    return HIVE_OK;
}

int hive_vit_destroy(hive_vit *v) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (!v) return HIVE_OK;
    (void)hipStreamSynchronize(v->ctx->stream);
    if (v->ws) (void)hipFree(v->ws);
    if (v->fold_mem) (void)hipFree(v->fold_mem);
    delete v;
    return HIVE_OK;
}

int hive_vit_forward(hive_vit *v, const void *x, int B, int N, const int *tap_blocks, int n_taps, void *const *tap_out) {
    HIVE_ENTER(v ? v->ctx : nullptr);
    if (!v) return hive_fail(nullptr, HIVE_ERR_INVALID, "vit is NULL");
    hive_ctx *ctx = v->ctx;
    HIVE_REQUIRE(ctx, x && B > 0 && N > 0, "vit_forward: bad arguments");
    HIVE_REQUIRE(ctx, n_taps >= 0 && (n_taps == 0 || (tap_blocks && tap_out)), "vit_forward: bad taps");
    int rc;
    const int D = v->dim, H = v->heads, Np = pad64(N), M = B * Np, dt = v->dtype;
    const size_t tok = (size_t)M * D * sizeof(half_bits);
    // LayerNorm folded into the GEMMs that consume it (default; HIVE_LN_FOLD=0: a LayerNorm pass in front of q|k|v and of fc1, the normalised tensor
    // written and read once more).  Folded: the proj / fc2 epilogues leave the statistics of the rows they store (per 64 columns), ln_finalize_kernel
    // turns them into (mean, rstd) per row, and the q|k, v^T and fc1 epilogues apply them to x (gamma o W)^T: -24 LayerNorm launches (1.8 ms per 107-frame
    // step) for 24 finalize launches of a few microseconds, and the token tensor makes one round trip less per half block.
    const char *fold_env = getenv("HIVE_LN_FOLD");  // (read per call: tests toggle it inside one process)
    const bool fold = !(fold_env && fold_env[0] == '0');
    const int groups = D / 64;
    // workspace: x (residual stream), ln, qk (2D), vT (D), attn, hidden (mlp), LayerNorm statistics (partial sums, (mean, rstd))
    const size_t part_bytes = ((size_t)M * groups * 2 * sizeof(float) + 255) & ~(size_t)255, stat_bytes = ((size_t)M * 2 * sizeof(float) + 255) & ~(size_t)255;
    const size_t off_x = 0, off_ln = off_x + tok, off_qk = off_ln + tok, off_vt = off_qk + 2 * tok, off_attn = off_vt + tok,
                 off_hid = off_attn + tok, off_part = off_hid + (size_t)M * v->mlp * sizeof(half_bits), off_stat = off_part + part_bytes,
                 total = off_stat + stat_bytes;
    if ((rc = hive_reserve_device(ctx, &v->ws, &v->ws_bytes, total))) return rc;
    char *ws = (char *)v->ws;
    half_bits *xs = (half_bits *)(ws + off_x), *ln = (half_bits *)(ws + off_ln), *qk = (half_bits *)(ws + off_qk), *vt = (half_bits *)(ws + off_vt),
              *attn = (half_bits *)(ws + off_attn), *hid = (half_bits *)(ws + off_hid);
    float *part = (float *)(ws + off_part), *stats = (float *)(ws + off_stat);
    const int cp_blocks = std::min<int>((int)(((size_t)M * D / 8 + 255) / 256), ctx->num_cus * 8);
    hipLaunchKernelGGL(pad_tokens_kernel, dim3(cp_blocks), dim3(256), 0, ctx->stream, (const half_bits *)x, xs, B, N, Np, D, 1);
    HIVE_CHECK_HIP(ctx, hipGetLastError());
    auto finalize = [&]() -> int {
        hipLaunchKernelGGL(ln_finalize_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, ctx->stream, (const float *)part, M, groups, v->eps, stats);
        HIVE_CHECK_HIP(ctx, hipGetLastError());
        return HIVE_OK;
    };
    auto linear = [&](const void *A, const void *W, const float *bias, const void *residual, void *C, int Mm, int Nn, int Kk, int epi, const LnFold &lf) {
        return dt == HIVE_BF16 ? linear_t<__bf16>(ctx, A, W, bias, residual, C, Mm, Nn, Kk, epi, lf) : linear_t<_Float16>(ctx, A, W, bias, residual, C, Mm, Nn, Kk, epi, lf);
    };
    for (int i = 0; i < v->depth; ++i) {
        const hive_vit_block_weights &w = v->blocks[i];
        if (!fold) {
            if ((rc = hive_vit_layernorm(ctx, xs, dt, (const float *)w.ln1_g, (const float *)w.ln1_b, ln, M, D, v->eps))) return rc;
            if ((rc = hive_vit_qkv(ctx, ln, dt, w.qkv_w, (const float *)w.qkv_b, qk, vt, B, Np, D, H))) return rc;
            if ((rc = hive_vit_attention(ctx, qk, dt, vt, attn, B, N, Np, D, H))) return rc;
            if ((rc = hive_vit_linear(ctx, attn, dt, w.proj_w, (const float *)w.proj_b, xs, xs, M, D, D, EPI_BIAS_RESIDUAL))) return rc;
            if ((rc = hive_vit_layernorm(ctx, xs, dt, (const float *)w.ln2_g, (const float *)w.ln2_b, ln, M, D, v->eps))) return rc;
            if ((rc = hive_vit_linear(ctx, ln, dt, w.fc1_w, (const float *)w.fc1_b, nullptr, hid, M, v->mlp, D, EPI_BIAS_GELU))) return rc;
            if ((rc = hive_vit_linear(ctx, hid, dt, w.fc2_w, (const float *)w.fc2_b, xs, xs, M, D, v->mlp, EPI_BIAS_RESIDUAL))) return rc;
        } else {
            const hive_vit::Folded &f = v->folded[i];
            if (i == 0) {  // the first LayerNorm's input comes from no GEMM: its rows' statistics by a pass of their own
                const dim3 grid((M + 3) / 4), block(256);
#define HIVE_STATS(T_, CH_) hipLaunchKernelGGL((ln_row_stats_kernel<T_, CH_>), grid, block, 0, ctx->stream, (const T_ *)xs, stats, M, v->eps)
                if (dt == HIVE_BF16) {
                    switch (D / 256) { case 1: HIVE_STATS(__bf16, 1); break; case 2: HIVE_STATS(__bf16, 2); break; case 3: HIVE_STATS(__bf16, 3); break; default: HIVE_STATS(__bf16, 4); }
                } else {
                    switch (D / 256) { case 1: HIVE_STATS(_Float16, 1); break; case 2: HIVE_STATS(_Float16, 2); break; case 3: HIVE_STATS(_Float16, 3); break; default: HIVE_STATS(_Float16, 4); }
                }
#undef HIVE_STATS
                HIVE_CHECK_HIP(ctx, hipGetLastError());
            } else if ((rc = finalize())) {  // statistics of the rows the previous block's fc2 stored
                return rc;
            }
            LnFold in1, in2, out;
            in1.stats = stats, in1.c1 = f.qkv_c1;
            in2.stats = stats, in2.c1 = f.fc1_c1;
            out.partial = part;
            rc = dt == HIVE_BF16 ? qkv_t<__bf16>(ctx, xs, f.qkv_w, f.qkv_c2, qk, vt, B, Np, D, H, in1) : qkv_t<_Float16>(ctx, xs, f.qkv_w, f.qkv_c2, qk, vt, B, Np, D, H, in1);
            if (rc) return rc;
            if ((rc = hive_vit_attention(ctx, qk, dt, vt, attn, B, N, Np, D, H))) return rc;
            if ((rc = linear(attn, w.proj_w, (const float *)w.proj_b, xs, xs, M, D, D, EPI_BIAS_RESIDUAL, out))) return rc;
            if ((rc = finalize())) return rc;
            if ((rc = linear(xs, f.fc1_w, f.fc1_c2, nullptr, hid, M, v->mlp, D, EPI_BIAS_GELU, in2))) return rc;
            if ((rc = linear(hid, w.fc2_w, (const float *)w.fc2_b, xs, xs, M, D, v->mlp, EPI_BIAS_RESIDUAL, i + 1 < v->depth ? out : LnFold()))) return rc;
        }
        for (int t = 0; t < n_taps; ++t)
            if (tap_blocks[t] == i) {
                hipLaunchKernelGGL(pad_tokens_kernel, dim3(cp_blocks), dim3(256), 0, ctx->stream, (const half_bits *)xs, (half_bits *)tap_out[t], B,
                                   N, Np, D, 0);
                HIVE_CHECK_HIP(ctx, hipGetLastError());
            }
    }
    return HIVE_OK;
}

}  // extern "C"
