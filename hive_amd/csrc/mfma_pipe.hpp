// One 64-deep K-step of the 16-bit MFMA tile kernels (csrc/vit.hip GEMMs, csrc/conv.hip), shared so that the schedule that was
// measured once is the schedule everywhere.  gfx950 only.  The element type T is __bf16 (north_star's contract) or _Float16 (what the
// reference runs: `model.half()`, /root/reference/hive/dataset_adaptors.py:1394-1401): v_mfma_f32_16x16x32_{bf16,f16} have the same
// shape, operand layout and rate, so one schedule serves both; only the instruction and the epilogue's conversion differ.
//
// A wave owns MT x 4 accumulators of 16 x 16 (MT fragments along the A rows, 4 along the W rows); operands sit in LDS as
// 128-byte rows with the source-side chunk swizzle (swz).  The step is 2 MT "slots" (sub-step of 32 k, A fragment) of 4
// v_mfma_f32_16x16x32_bf16 each.  Two things are woven between the slots, in SOURCE order (the compiler keeps it: an LDS-DMA
// and the ds_reads around it may alias, so neither moves across the other):
//   * fragment reads run ahead: the A fragment of slot s + 3 and the W fragments of the second sub-step are read while the
//     MFMAs of slot s issue (one exposed LDS round trip per step instead of one per 8 MFMAs);
//   * the NEXT stage's LDS-DMA pieces (`issue(j)`, j < n_pieces) go out one per slot instead of back to back behind the
//     barrier: a CU's texture-address unit accepts one vector-memory wave-instruction per ~40 cycles (64 per 64 KiB stage and CU
//     = the 2550 cycles tools/ubench/ldsdma.hip measures for a bare fill), and an in-order wave that issues its 8 at once
//     stands in that queue before its first MFMA.
// Measured on the decoder convolutions (tools/probe_conv3x3.py): 820-910 -> 900-1030 TFLOP/s.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

namespace hive_mfma {

template <typename T, int N>
using vec = T __attribute__((ext_vector_type(N)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// acc += A . B^T on one 16 x 16 (k = 32) or 32 x 32 (k = 16) tile; 8 elements of T per lane and operand
__device__ __forceinline__ f32x4 mfma16(vec<__bf16, 8> a, vec<__bf16, 8> b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(vec<_Float16, 8> a, vec<_Float16, 8> b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(vec<__bf16, 8> a, vec<__bf16, 8> b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(vec<_Float16, 8> a, vec<_Float16, 8> b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// One LDS-DMA wave instruction (64 lanes x 16 B, lane-linear, to the wave-uniform LDS address lds_dst) outside the compiler's wait-count bookkeeping.  The rings
// deeper than two stages count vmcnt BY HAND (s_waitcnt vmcnt(pieces of the younger stages)): any wait hipcc adds inside the K loop on its own account
// drains the ring every step, silently (it shows only in the ISA).  The one such wait found here came from ordinary loads (see the builtin s_waitcnt behind the
// deep kernels' epilogues); hipcc 7.2 did not add one for the builtin LDS-DMA itself in these kernels (-DHIVE_LDSDMA_TRACKED builds them that way: same ISA
// waits), but it can (the CDNA4 guide reports an s_waitcnt vmcnt(0) in front of the first ds_read of every K-step once a second LDS object exists), so the
// deep rings keep their pieces out of its books altogether.  M0 holds the LDS base and is compiler-reserved: saved, set, put back (the guide's recipe).
__device__ __forceinline__ void lds_dma16_untracked(const void *gsrc, unsigned lds_dst) {
#ifdef HIVE_LDSDMA_TRACKED  // (tuning build: the compiler's own LDS-DMA)
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void *)(size_t)lds_dst, 16, 0, 0);
    return;
#endif
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// LDS byte address of a (generic) pointer into the kernel's LDS array: the low half of the flat address (the high half is the shared aperture)
__device__ __forceinline__ unsigned lds_address(const void *p) { return (unsigned)(size_t)p; }

// byte offset of 16-byte chunk `c` (0..7) of row `r` in a tile with 128-byte rows: the chunk is XORed with (r >> 1) & 7 so that
// any 16 consecutive rows at one chunk index land on 16 distinct 16-byte slots
__device__ __forceinline__ int swz(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// SWAP = false: acc[nt][mt] += W_frag[nt] . A_frag[mt]^T (rows = W rows: a lane owns 4 consecutive W rows of one A row)
// SWAP = true : acc[mt][nt] += A_frag[mt] . W_frag[nt]^T (MT == 4 only: the transposed v^T store of the QKV GEMM)
template <typename T, int MT, bool SWAP, typename Acc, typename Issue>
__device__ __forceinline__ void kstep64(const unsigned char *a_t, const unsigned char *w_t, int a_row0, int w_row0, int fr, int fq, Acc &acc,
                                        int n_pieces, Issue &&issue) {
    constexpr int SLOTS = 2 * MT, D = 3;
    constexpr int WPS = MT >= 4 ? 1 : 4 / MT;  // W fragments of sub-step 1 prefetched per slot
    static_assert(MT == 2 || MT == 4 || MT == 8, "A fragments per wave");
    typedef vec<T, 8> frag;
    auto rd_a = [&](int sl) { return *reinterpret_cast<const frag *>(a_t + swz(a_row0 + (sl % MT) * 16 + fr, (sl / MT) * 4 + fq)); };
    auto rd_w = [&](int sub, int t) { return *reinterpret_cast<const frag *>(w_t + swz(w_row0 + t * 16 + fr, sub * 4 + fq)); };
    frag ring[4], wfr[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) wfr[0][t] = rd_w(0, t);
#pragma unroll
    for (int sl = 0; sl < D; ++sl) ring[sl] = rd_a(sl);
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        if (sl + D < SLOTS) ring[(sl + D) & 3] = rd_a(sl + D);
        if (sl < 4 / WPS) {  // the second sub-step's W fragments, all four before slot MT
#pragma unroll
            for (int k = 0; k < WPS; ++k) wfr[1][sl * WPS + k] = rd_w(1, sl * WPS + k);
        }
        const int sub = sl / MT, mt = sl % MT;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (SWAP)
                acc[mt][nt] = mfma16(ring[sl & 3], wfr[sub][nt], acc[mt][nt]);
            else
                acc[nt][mt] = mfma16(wfr[sub][nt], ring[sl & 3], acc[nt][mt]);
        }
        if (sl < n_pieces) issue(sl);
    }
    for (int j = SLOTS; j < n_pieces; ++j) issue(j);  // more pieces than slots (narrow tiles)
}

// The same K-step ROTATED ACROSS THE BARRIER.  With one s_barrier per K-step and all 8 waves (two per SIMD) behind it, every wave
// starts its step with an exposed LDS round trip (its first fragments cannot be read before the stage is known to have landed), both
// waves of a SIMD at the same moment: the MFMA pipe idles.  KPipe holds the MFMAs of the step's LAST `DEFER` slots back (their
// fragments are in registers), and the next step, after the barrier, first issues its fragment reads (`begin`), then runs the
// held-back MFMAs under that latency (`flush`), then its own slots (`body`).  Per tile: begin/body for the first step, barrier +
// begin/flush/body for the others, one flush before the epilogue.
#ifndef HIVE_PIPE_ABLATE_READS
#define HIVE_PIPE_ABLATE_READS 0  // tuning build (make ablate_reads; WRONG results): 1 = a third of KPipe's fragment reads left out -- what a 128 x 128-per-wave register
                                  // blocking would save of the LDS port's 192 KiB per K-step -- to see whether the K loop is LDS-port-bound before building that kernel
#endif
template <typename T, int MT, bool SWAP>
struct KPipe {
    typedef vec<T, 8> frag;
    static constexpr int SLOTS = 2 * MT, DEFER = 2, D = 3;
    static constexpr int WPS = MT >= 4 ? 1 : 4 / MT;
    static_assert(MT == 2 || MT == 4 || MT == 8, "A fragments per wave");
    frag ring[4], wfr[2][4];  // slot sl uses ring[sl & 3]; the deferred slots SLOTS - 2, SLOTS - 1 sit in ring[2], ring[3]
    const unsigned char *a_t, *w_t;
    int a_row0, w_row0, fr, fq;

    __device__ __forceinline__ frag rd_a(int sl) const {
        if (HIVE_PIPE_ABLATE_READS && sl >= MT + MT / 2) return wfr[0][sl & 3];  // (timing experiment: no LDS read for the second half of sub-step 1's A fragments)
        return *reinterpret_cast<const frag *>(a_t + swz(a_row0 + (sl % MT) * 16 + fr, (sl / MT) * 4 + fq));
    }
    __device__ __forceinline__ frag rd_w(int sub, int t) const {
        if (HIVE_PIPE_ABLATE_READS && sub == 1) return wfr[0][t];  // (timing experiment: sub-step 1 reuses sub-step 0's W fragments)
        return *reinterpret_cast<const frag *>(w_t + swz(w_row0 + t * 16 + fr, sub * 4 + fq));
    }
    template <typename Acc>
    __device__ __forceinline__ void mfma_slot(int sl, Acc &acc) {
        const int sub = sl / MT, mt = sl % MT;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (SWAP)
                acc[mt][nt] = mfma16(ring[sl & 3], wfr[sub][nt], acc[mt][nt]);
            else
                acc[nt][mt] = mfma16(wfr[sub][nt], ring[sl & 3], acc[nt][mt]);
        }
    }
    // right behind the barrier: point at the new stage and issue its first reads (W of sub-step 0, A of slots 0 and 1)
    __device__ __forceinline__ void begin(const unsigned char *a_tile, const unsigned char *w_tile) {
        a_t = a_tile;
        w_t = w_tile;
#pragma unroll
        for (int t = 0; t < 4; ++t) wfr[0][t] = rd_w(0, t);
        ring[0] = rd_a(0);
        ring[1] = rd_a(1);
    }
    // the MFMAs held back from the previous step (fragments in ring[2], ring[3], wfr[1])
    template <typename Acc>
    __device__ __forceinline__ void flush(Acc &acc) {
#pragma unroll
        for (int sl = SLOTS - DEFER; sl < SLOTS; ++sl) mfma_slot(sl, acc);
    }
    template <typename Acc, typename Issue>
    __device__ __forceinline__ void body(Acc &acc, int n_pieces, Issue &&issue) {
        ring[2] = rd_a(2);
#pragma unroll
        for (int sl = 0; sl < SLOTS - DEFER; ++sl) {
            if (sl + D < SLOTS) ring[(sl + D) & 3] = rd_a(sl + D);
            if (sl < 4 / WPS) {
#pragma unroll
                for (int k = 0; k < WPS; ++k) wfr[1][sl * WPS + k] = rd_w(1, sl * WPS + k);
            }
            mfma_slot(sl, acc);
            if (sl < n_pieces) issue(sl);
        }
        for (int j = SLOTS - DEFER; j < n_pieces; ++j) issue(j);
    }
};

// Epilogue of the SWAP = false accumulators through LDS.  In the MFMA layout a lane owns 4 consecutive W rows (output columns) of
// one A row, so a direct store writes 16 rows x 32 bytes per wave-instruction, 8 bytes per lane.  A CU's vector-memory path takes a
// wave-instruction every 40-47 cycles whatever its width (per-workgroup clocks, tools/probe_gemm_stamps.py: 23 000 cycles for the
// 8 x 32 such stores of a 256 x 256 tile -- as long as 8 of the tile's 12 K-steps at K = 768), so the epilogue wants FEW, WIDE
// instructions on whole lines.  Each wave turns one 16-row fragment row (16 x 64 f32 = 4 KiB) around in a PRIVATE 4 KiB of LDS:
// ds_write_b128 in the MFMA layout, two ds_read_b128 with 8 lanes along a row; `finish(row, col, lo, hi)` then gets 8 consecutive
// columns with the lanes of a row side by side: its 16-byte loads / stores cover 8 rows x 128 contiguous bytes per instruction
// (16 stores per wave and tile instead of 32, 12 000 -> 6 000 cycles).
// Image: 256-byte rows, 16-byte chunk c of row r at slot c ^ r (both the writes -- 8 consecutive rows at one chunk index -- and the
// reads -- the four 16-lane groups of ds_read_b128 -- are conflict-free).  Wave-private, LDS operations of a wave execute in order:
// no workgroup barrier.
// `pre(mt, j)` (optional) is called one fragment row AHEAD of `finish(row, col, lo, hi, mt, j)`: the place to issue the global loads
// (residual rows) that finish will consume, so that they fly during the previous fragment row's turn-around instead of being
// waited for in front of every store.
// AHEAD = 2: `pre` runs two fragment rows ahead (three register sets in the caller): a row's turn takes ~0.5 us, less than a trip to HBM
// under load.
template <int MT, int AHEAD = 1, typename Acc, typename Pre, typename Finish>
__device__ __forceinline__ void staged_rows(unsigned char *stage, const Acc &acc, int lane, Pre &&pre, Finish &&finish) {
    const int fr = lane & 15, fq = lane >> 4;
    const int rr = lane >> 3, c0 = 2 * (lane & 7);
#pragma unroll
    for (int a = 0; a < AHEAD && a < MT; ++a) {
        pre(a, 0);
        pre(a, 1);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4 *>(stage + fr * 256 + (((nt * 4 + fq) ^ fr) << 4)) = acc[nt][mt];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (mt + AHEAD < MT) {
            pre(mt + AHEAD, 0);
            pre(mt + AHEAD, 1);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 8 * j + rr;
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(stage + r * 256 + ((c0 ^ r) << 4));
            const f32x4 hi = *reinterpret_cast<const f32x4 *>(stage + r * 256 + (((c0 + 1) ^ r) << 4));
            finish(mt * 16 + r, c0 * 4, lo, hi, mt, j);
        }
        __builtin_amdgcn_wave_barrier();  // the next fragment row's writes stay behind these reads
    }
}

template <int MT, typename Acc, typename Finish>
__device__ __forceinline__ void staged_rows(unsigned char *stage, const Acc &acc, int lane, Finish &&finish) {
    staged_rows<MT, 1>(stage, acc, lane, [](int, int) {}, [&](int r, int c, const f32x4 &lo, const f32x4 &hi, int, int) { finish(r, c, lo, hi); });
}

constexpr int STAGED_ROWS_LDS = 8 * 4096;  // 8 waves

// Split-K for small batches.  With fewer tiles than CUs a tile kernel lasts as long as ONE workgroup's K loop (12-48 steps of 1-2 us, each
// waiting for a stage to arrive from L2 / HBM) while most of the chip idles, so the K-steps of a tile are dealt to S workgroups ("items":
// item = tile * S + s multiplies steps [s KT / S, (s + 1) KT / S)).  Every item leaves its f32 accumulators in the workspace as they sit in
// the registers (NT threads x R x C f32x4, lane-linear: 16 bytes per lane, whole lines per wave instruction) and counts itself in on the
// tile's counter; the item whose add comes LAST adds the S partials in the order s = 0 .. S - 1 (its own included, re-read: the sum does not
// depend on who is last) and runs the tile's ordinary epilogue.  No workgroup waits for another.
// Visibility (gfx950: per-CU L1s, per-XCD L2s, none refreshed by another CU's stores) follows the in-launch split-K recipe of the CDNA4 guide:
// the partials are stored WRITE-THROUGH (raw buffer stores with aux = sc1: they leave the XCD's L2, so no release fence -- a fence per workgroup
// writes back and invalidates whole caches: 5.35 -> 8.7 ms per one-frame forward when first built that way), every storing wave drains its stores
// (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE lane adds to the counter (relaxed, agent scope); the last arriver's lane makes ONE
// agent-scope acquire (drops this CU's stale L1 lines), waits for it, and the workgroup meets again before anyone loads a partial.
// Returns whether this workgroup holds the tile's sum; `flag`: 4 bytes of the kernel's one LDS array.  The last arriver puts the counter back to
// zero for the stream's next launch (the counters are zeroed when allocated: hive_splitk_workspace).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NT, int R, int C>
__device__ __forceinline__ bool splitk_combine(int S, f32x4 *ws, unsigned *count, int item, f32x4 (&acc)[R][C], int tid, int *flag) {
    constexpr unsigned SLAB = R * C * NT * 16;  // bytes per partial
    const int tile = item / S;
    {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char *>(ws) + (size_t)item * SLAB, 0, (int)SLAB, 0x00020000);
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int j = 0; j < C; ++j) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rsrc, ((i * C + j) * NT + tid) * 16, 0, 16);  // aux 16 = sc1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave, before the barrier
    __syncthreads();
    if (tid == 0) {
        const unsigned arrived = __hip_atomic_fetch_add(count + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (unsigned)(S - 1)) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (holds the barrier below until the invalidate has completed)
            __hip_atomic_store(count + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *flag = (int)arrived;
    }
    __syncthreads();
    const bool last = *flag == S - 1;
    __syncthreads();  // (the caller's epilogue reuses flag's LDS)
    if (!last) return false;
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < C; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 *src = ws + (size_t)tile * S * (R * C * NT) + tid;
    for (int s = 0; s < S; ++s, src += R * C * NT) {
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int j = 0; j < C; ++j) acc[i][j] += src[(i * C + j) * NT];
    }
    return true;
}

// How many ways to split (measured, tools/probe_splitk.py): an item costs ~4 us of its own (write-through stores, the counter, the acquire) and the last
// arriver ~1 us per 64 KiB partial it reads, a K-step of the deep ring 0.6-0.7 us: splitting pays for LONG K loops only (fc2's 48 steps: 32.6 -> 21 us four
// ways; the 12-step GEMMs lose), and never beyond one round of workgroups on the CUs.
inline int splitk_ways(long long tiles, int KT, long long cus) {
    if (tiles <= 0 || KT < 32) return 1;
    return (int)std::max<long long>(1, std::min<long long>(4, cus / tiles));
}

}  // namespace hive_mfma
