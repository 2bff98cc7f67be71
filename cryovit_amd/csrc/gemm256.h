// 256x256x64 bf16 MFMA GEMM tile for gfx950, 8 waves, phase-pipelined LDS-DMA.   T[r][l] = sum_k R[r][k] * L[l][k]
//
// The workhorse of the ViT (and of the head's 1x1x1 projection): every linear layer with N % 256 == 0 runs here.
// Same operand / epilogue conventions as gemm_core.h (R side -> accumulator registers, L side -> lanes, sigma row
// permutation on the R side so a lane owns 16 contiguous output features), different pipeline:
//
//   * 8 waves = 4 (R) x 2 (L); a wave owns 64 R rows x 128 L rows = 4 x 8 MFMA tiles (128 accumulator registers).
//   * A K tile (64 deep) is cut into FOUR 16-KiB half-tiles laid out in LDS in the order the waves consume them:
//       q%4 = 0: R-lo (first 32 R rows of every wave)   1: L-lo (first 64 L rows of every wave)
//             2: R-hi                                     3: L-hi
//     and one K tile is computed in four phases of 16 MFMAs:  (R-lo,L-lo) (R-hi,L-lo) (R-hi,L-hi) (R-lo,L-hi).
//     Fragments stay in registers after their phase, so half-tile q is READ only in phase rd(q) in {q-1, q}.
//   * LDS holds a ring of 8 half-tiles (2 K tiles, 128 KiB).  Every half-tile q is READ exactly once, into registers, in
//     the load segment of phase q-1 (R-lo of the NEXT K tile is read during phase 3, into the register set the dead R-hi
//     fragments just vacated -- the two R register sets swap roles every K tile).  Phase g issues the LDS-DMA of
//     half-tile g+7: FIVE half-tiles (80 KiB) are always in flight across barriers behind a counted
//     s_waitcnt vmcnt(10) -- the kernel is DMA-latency bound (64 KiB per 2048 MFMA cycles per CU against ~1 us of
//     loaded L2/HBM latency), so bytes in flight, not issue slots, set its speed.
//   * Two barriers per phase split it into a LOAD segment (ds_read fragments, issue DMA, counted wait) and an MFMA
//     segment.  Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA segment while
//     its partner is in its load segment: the matrix pipe never waits for LDS latency.
//   Hazards (barrier instances I_k; group 0 load segment g = (I_2g+1, I_2g+2), group 1 = (I_2g+2, I_2g+3)):
//     RAW  half-tile q (read in phase q-1) is waited for by EVERY wave (its own DMA pieces) at the end of its load
//          segment of phase q-2, i.e. before I_2q-1 at the latest; the earliest read of q is after I_2q-1.
//     WAR  the slot of q is rewritten by DMA(q+8), issued in load segment q+1 (after I_2q+3); the late group's reads of
//          q completed (lgkmcnt(0) opens its MFMA segment q-1) before I_2q+2.
#pragma once
#include "gemm_core.h"
#include <type_traits>

namespace cvx {

constexpr int G256_THREADS = 512;
constexpr int G256_HALF_BYTES = 128 * 128;            // 128 rows x 64 bf16
constexpr int G256_LDS_BYTES = 8 * G256_HALF_BYTES;   // ring of 8 half-tiles

template <class E, class = void> struct epi_has_produce : std::false_type {};
template <class E> struct epi_has_produce<E, std::enable_if_t<(E::OUT16 > 0)>> : std::true_type {};
template <class E, class = void> struct epi_has_preload : std::false_type {};
template <class E> struct epi_has_preload<E, std::enable_if_t<E::HAS_PRELOAD>> : std::true_type {};

// diagnostic build only (VARIANT 20): per-wave cycle sums of the four parts of a phase, block 0 -> g_gemm256_dbg
__device__ unsigned long long g_gemm256_dbg[8 * 4];
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// wait until at most `pending` half-tiles (2 LDS-DMA instructions each) of this wave are still in flight
__device__ __forceinline__ void wait_halftiles(int pending) {
    switch (pending) {  // wave-uniform
        case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// De-synchronise the first round of workgroups: all CUs start together and every tile takes the same time, so without
// this the whole chip alternates between a compute phase (HBM idle) and an epilogue phase in which 256 workgroups hit HBM
// at once (a 512-KB fp32 read-modify-write per tile took as long as the K loop of the proj GEMM).  Workgroups of the
// first round sleep group * quarter-tile; the hardware dispatcher keeps the offsets for the rest of the launch.
__device__ __forceinline__ void stagger_first_round(int stagger_cycles) {
    if (stagger_cycles <= 0 || blockIdx.x >= 256) return;
    const int group = (blockIdx.x >> 3) & 3;  // blocks b, b+8 share an XCD: spread the groups inside every XCD
    if (group == 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long want = (unsigned long long)group * (unsigned)stagger_cycles;
    while (__builtin_amdgcn_s_memtime() - t0 < want) __builtin_amdgcn_s_sleep(32);
}

template <int VARIANT, class Epi>
__device__ __forceinline__ void gemm256_body(const uint16_t* __restrict__ Rmat, long ldr, const uint16_t* __restrict__ Lmat,
                                             long ldl, int nk, long r0, long l0, const Epi& epi, char* smem, int koff = 0) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 3, wl = wave >> 2;  // waves 0-3 / 4-7 = L half 0 / 1 = the two stagger groups
    const int total = 4 * nk;

    // ---- LDS-DMA source offsets: 2 x 16-B pieces per thread per half-tile ----
    uint32_t offR[2][2], offL[2][2];  // [half][piece], BYTES relative to the tile origin, K tile 0 (saddr-form DMA, common.h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = j * G256_THREADS + tid;
        const int hr = c >> 3, ch = ((c & 7) ^ swz_chunk(hr)) << 3;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int tr = (hr >> 5) * 64 + half * 32 + (hr & 31);   // R tile row: wave hr>>5, rows [32*half, +32)
            offR[half][j] = (uint32_t)(sigma_row<4>(tr) * ldr + ch) * 2u;
            const int tl = (hr >> 6) * 128 + half * 64 + (hr & 63);  // L tile row: wave hr>>6, rows [64*half, +64)
            offL[half][j] = (uint32_t)(tl * ldl + ch) * 2u;
        }
    }
    const uint16_t* Rb = Rmat + r0 * ldr;
    const uint16_t* Lb = Lmat + l0 * ldl;

    // timing-only ablations (results are garbage): 10 = no DMA in the loop, 11 = DMA always from K tile 0 (L2-resident),
    // 12 = no fragment reads in the loop
    constexpr bool ABL_NO_DMA = VARIANT == 10, ABL_SAME_K = VARIANT == 11, ABL_NO_READ = VARIANT == 12;
    constexpr bool ABL_NO_R_DMA = VARIANT == 13;  // only the L half-tiles are DMA'd (what a weights-bypass-LDS kernel would move)
    auto issue = [&](int q) {  // half-tile q -> ring slot q & 7
        // koff rotates the K order per workgroup (variant 7): workgroups that share an operand panel then request its K
        // slabs at different times, so the later ones find the lines IN L2 instead of queueing behind the same miss
        int kt = ABL_SAME_K ? 0 : (q >> 2) + koff;
        if (kt >= nk) kt -= nk;
        const int kind = q & 3;
        const uint32_t dst = lds_addr(smem) + (q & 7) * G256_HALF_BYTES + wave * 1024;
        // slot order inside a K tile: variants 0-5: R-lo, L-lo, R-hi, L-hi;  variant 6: R-lo, R-hi, L-lo, L-hi
        const bool isL = VARIANT == 6 ? (kind >= 2) : (kind & 1);
        const int half = VARIANT == 6 ? (kind & 1) : (kind >> 1);
        const uint16_t* src = (isL ? Lb : Rb) + kt * BK;
        const uint32_t o0 = isL ? offL[half][0] : offR[half][0];
        const uint32_t o1 = isL ? offL[half][1] : offR[half][1];
        if (ABL_NO_R_DMA && !isL) {  // keep the vmcnt arithmetic: two cheap L2-resident pieces instead
            glds16_saddr2<G256_THREADS * 16>(Lb, o0 % 128, o1 % 128, dst);
            return;
        }
        glds16_saddr2<G256_THREADS * 16>(src, o0, o1, dst);
    };

    // ---- fragment read offsets inside a half-tile (k-step 0; k-step 1 = ^64) ----
    const int sw = (lane & 15) >> 1;
    const int fo = (lane & 15) * 128 + (((lane >> 4) ^ sw) << 4);
    const int foR = wr * 32 * 128 + fo;  // + f*2048, f = 0,1
    const int foL = wl * 64 * 128 + fo;  // + f*2048, f = 0..3

    f32x4 acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    [[maybe_unused]] unsigned long long dbg_load = 0, dbg_lbar = 0, dbg_mma = 0, dbg_mbar = 0;
    constexpr bool STAMP = VARIANT == 20;
    constexpr bool COARSE = VARIANT == 21;
    [[maybe_unused]] unsigned long long c0 = 0, c1 = 0, c2 = 0;
    if constexpr (COARSE) c0 = stamp();
    if constexpr (VARIANT == 6) {
    // ---- variant 6: TWO phases of 32 MFMAs per K tile (half the barriers): A = (R, L-lo), B = (R, L-hi); the DMA runs
    //      exactly one K tile ahead (R halves issued in phase A, L halves in phase B) ----
    bf16x8 rlo[2][2], rhi[2][2], lf[4][2];
    auto read_r = [&](bf16x8 (&dst)[2][2], const char* half) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) dst[f][ks] = *(const bf16x8*)(half + ((foR + f * 2048) ^ (ks << 6)));
    };
    auto read_l = [&](const char* half) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) lf[f][ks] = *(const bf16x8*)(half + ((foL + f * 2048) ^ (ks << 6)));
    };
    auto mma32 = [&](int b0) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    acc[a][b0 + b] = mfma16x16x32<epi_is_f16<Epi>::value>(a < 2 ? rlo[a][ks] : rhi[a - 2][ks], lf[b][ks], acc[a][b0 + b]);
        __builtin_amdgcn_s_setprio(0);
    };
    for (int q = 0; q < 4; ++q) issue(q);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // R-lo, R-hi, L-lo of K tile 0 landed
    __builtin_amdgcn_s_barrier();
    if (wl == 1) __builtin_amdgcn_s_barrier();
    auto ktile = [&](int t, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        const char* st = smem + (t & 1) * 4 * G256_HALF_BYTES;
        const int g = 4 * t;
        read_r(rlo, st);
        read_r(rhi, st + G256_HALF_BYTES);
        read_l(st + 2 * G256_HALF_BYTES);
        if (STEADY || g + 4 < total) { issue(g + 4); issue(g + 5); }
        if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        mma32(0);
        __builtin_amdgcn_s_barrier();
        read_l(st + 3 * G256_HALF_BYTES);
        if (STEADY || g + 6 < total) { issue(g + 6); issue(g + 7); }
        if constexpr (STEADY) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        mma32(4);
        __builtin_amdgcn_s_barrier();
    };
    {
        int t = 0;
        for (; t < nk - 1; ++t) ktile(t, std::true_type{});
        for (; t < nk; ++t) ktile(t, std::false_type{});
    }
    if (wl == 0) __builtin_amdgcn_s_barrier();
    } else if constexpr (VARIANT == 0 || VARIANT == 4 || VARIANT == 5 || VARIANT == 8 || VARIANT >= 10) {
    // ---- variants 0/4/5: reads at rd(q) in {q-1,q}, 3-4 half-tiles in flight (A = 5) ----
    bf16x8 rlo[2][2], rhi[2][2], lf[4][2];  // [frag][k-step]

    auto read_r = [&](bf16x8 (&dst)[2][2], const char* half) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) dst[f][ks] = *(const bf16x8*)(half + ((foR + f * 2048) ^ (ks << 6)));
    };
    auto read_l = [&](const char* half) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) lf[f][ks] = *(const bf16x8*)(half + ((foL + f * 2048) ^ (ks << 6)));
    };
    auto mma = [&](const bf16x8 (&r)[2][2], int a0, int b0) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int b = 0; b < 4; ++b)  // (B operand kept across consecutive MFMAs: gemm256p.h)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    acc[a0 + a][b0 + b] = mfma16x16x32<epi_is_f16<Epi>::value>(r[a][ks], lf[b][ks], acc[a0 + a][b0 + b]);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: half-tiles 0..A-1 in flight, 0 and 1 landed ----
#pragma unroll
    for (int q = 0; q < 5; ++q)
        if (q < total) issue(q);
    if (total > 5) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wl == 1) __builtin_amdgcn_s_barrier();  // stagger: waves 4-7 run one barrier behind
    if constexpr (COARSE) c1 = stamp();

    // one K tile.  STEADY: all four DMA issues exist -> unconditional issue + exact counted waits (no branches).
    // ISSUE_IN_MMA: the DMA is issued from the MFMA segment (in the issue gaps of the wave's own MFMAs) instead of the
    // load segment, which is the critical one (it competes for issue slots with the partner wave's MFMA stream).
    constexpr bool ISSUE_IN_MMA = (VARIANT == 5 || VARIANT == 8 || VARIANT == 21);  // 8 = 5 with the un-staged bf16 epilogue (A/B)
    auto ktile = [&](int t, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        const char* st = smem + (t & 1) * 4 * G256_HALF_BYTES;
        const int g = 4 * t;
        [[maybe_unused]] unsigned long long t0 = 0;
        if constexpr (STAMP) t0 = stamp();
        auto do_issue = [&](int q) {
            if constexpr (ABL_NO_DMA) return;
            if (STEADY || q < total) issue(q);
        };
        auto do_wait = [&](auto n_tag) {
            constexpr int N = decltype(n_tag)::value;
            if constexpr (!STEADY) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        };
        using W6 = std::integral_constant<int, ISSUE_IN_MMA ? 4 : 6>;
        using W8 = std::integral_constant<int, ISSUE_IN_MMA ? 6 : 8>;
        // ---- phase 0: (R-lo, L-lo) ----
        if (!ABL_NO_READ || t == 0) {
            read_r(rlo, st);
            read_l(st + G256_HALF_BYTES);
        }
        if constexpr (!ISSUE_IN_MMA) do_issue(g + 5);
        do_wait(W6{});
        if constexpr (STAMP) { const auto t1 = stamp(); dbg_load += t1 - t0; t0 = t1; }
        __builtin_amdgcn_s_barrier();
        if constexpr (STAMP) { const auto t1 = stamp(); dbg_lbar += t1 - t0; t0 = t1; }
        if constexpr (ISSUE_IN_MMA) do_issue(g + 5);
        mma(rlo, 0, 0);
        if constexpr (STAMP) { const auto t1 = stamp(); dbg_mma += t1 - t0; t0 = t1; }
        __builtin_amdgcn_s_barrier();
        if constexpr (STAMP) { const auto t1 = stamp(); dbg_mbar += t1 - t0; t0 = t1; }
        // ---- phase 1: (R-hi, L-lo) ----
        if (!ABL_NO_READ || t == 0) read_r(rhi, st + 2 * G256_HALF_BYTES);
        if constexpr (!ISSUE_IN_MMA) do_issue(g + 6);
        do_wait(W6{});
        if constexpr (STAMP) { const auto t1 = stamp(); dbg_load += t1 - t0; t0 = t1; }
        __builtin_amdgcn_s_barrier();
        if constexpr (STAMP) { const auto t1 = stamp(); dbg_lbar += t1 - t0; t0 = t1; }
        if constexpr (ISSUE_IN_MMA) do_issue(g + 6);
        mma(rhi, 2, 0);
        if constexpr (STAMP) { const auto t1 = stamp(); dbg_mma += t1 - t0; t0 = t1; }
        __builtin_amdgcn_s_barrier();
        if constexpr (STAMP) { const auto t1 = stamp(); dbg_mbar += t1 - t0; t0 = t1; }
        // ---- phase 2: (R-hi, L-hi) ----
        if (!ABL_NO_READ || t == 0) read_l(st + 3 * G256_HALF_BYTES);
        if constexpr (!ISSUE_IN_MMA) do_issue(g + 7);
        do_wait(W8{});
        if constexpr (STAMP) { const auto t1 = stamp(); dbg_load += t1 - t0; t0 = t1; }
        __builtin_amdgcn_s_barrier();
        if constexpr (STAMP) { const auto t1 = stamp(); dbg_lbar += t1 - t0; t0 = t1; }
        if constexpr (ISSUE_IN_MMA) do_issue(g + 7);
        mma(rhi, 2, 4);
        if constexpr (STAMP) { const auto t1 = stamp(); dbg_mma += t1 - t0; t0 = t1; }
        __builtin_amdgcn_s_barrier();
        if constexpr (STAMP) { const auto t1 = stamp(); dbg_mbar += t1 - t0; t0 = t1; }
        // ---- phase 3: (R-lo, L-hi) ----
        if constexpr (!ISSUE_IN_MMA) do_issue(g + 8);
        do_wait(W6{});
        if constexpr (STAMP) { const auto t1 = stamp(); dbg_load += t1 - t0; t0 = t1; }
        __builtin_amdgcn_s_barrier();
        if constexpr (STAMP) { const auto t1 = stamp(); dbg_lbar += t1 - t0; t0 = t1; }
        if constexpr (ISSUE_IN_MMA) do_issue(g + 8);
        mma(rlo, 0, 4);
        if constexpr (STAMP) { const auto t1 = stamp(); dbg_mma += t1 - t0; t0 = t1; }
        __builtin_amdgcn_s_barrier();
        if constexpr (STAMP) { const auto t1 = stamp(); dbg_mbar += t1 - t0; t0 = t1; }
    };
    {   // (a per-tile runtime choice between the two instantiations makes hipcc spill the accumulators: keep two loops)
        int t = 0;
        for (; t < nk - 2; ++t) ktile(t, std::true_type{});
        for (; t < nk; ++t) ktile(t, std::false_type{});
    }
    if (wl == 0) __builtin_amdgcn_s_barrier();  // pairs with the stagger barrier of waves 4-7

    } else {
    // ---- variant 1: every half-tile read in phase q-1, AHEAD-2 half-tiles in flight ----
    constexpr int AHEAD = VARIANT == 1 ? 7 : (VARIANT == 2 ? 6 : 5);
    bf16x8 rx[2][2], ry[2][2], lf[4][2];  // [frag][k-step]; rx / ry swap the R-lo / R-hi roles every K tile

    auto read_r = [&](bf16x8 (&dst)[2][2], const char* half) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) dst[f][ks] = *(const bf16x8*)(half + ((foR + f * 2048) ^ (ks << 6)));
    };
    auto read_l = [&](const char* half) {
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) lf[f][ks] = *(const bf16x8*)(half + ((foL + f * 2048) ^ (ks << 6)));
    };
    auto mma = [&](const bf16x8 (&r)[2][2], int a0, int b0) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int b = 0; b < 4; ++b)  // (B operand kept across consecutive MFMAs: gemm256p.h)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    acc[a0 + a][b0 + b] = mfma16x16x32<epi_is_f16<Epi>::value>(r[a][ks], lf[b][ks], acc[a0 + a][b0 + b]);
        __builtin_amdgcn_s_setprio(0);
    };
    // end of the load segment of phase g: issue half-tile g+AHEAD, then retire (own pieces of) half-tile g+2
    auto issue_and_wait = [&](int g) {
        if (g + AHEAD < total) issue(g + AHEAD);
        const int last_issued = min(g + AHEAD, total - 1);
        wait_halftiles(last_issued - (g + 2));
        __builtin_amdgcn_s_barrier();
    };
    // one K tile = 4 phases; `lo` holds R-lo(t) on entry, `hi` receives R-hi(t) and then R-lo(t+1)
    auto ktile = [&](int t, bf16x8 (&lo)[2][2], bf16x8 (&hi)[2][2]) {
        const char* st = smem + (t & 1) * 4 * G256_HALF_BYTES;
        const char* nx = smem + ((t + 1) & 1) * 4 * G256_HALF_BYTES;
        const int g = 4 * t;
        read_l(st + G256_HALF_BYTES);            // phase 0: L-lo(t)
        issue_and_wait(g);
        mma(lo, 0, 0);
        __builtin_amdgcn_s_barrier();
        read_r(hi, st + 2 * G256_HALF_BYTES);    // phase 1: R-hi(t)
        issue_and_wait(g + 1);
        mma(hi, 2, 0);
        __builtin_amdgcn_s_barrier();
        read_l(st + 3 * G256_HALF_BYTES);        // phase 2: L-hi(t)
        issue_and_wait(g + 2);
        mma(hi, 2, 4);
        __builtin_amdgcn_s_barrier();
        if (t + 1 < nk) read_r(hi, nx);          // phase 3: R-lo(t+1) into the registers R-hi(t) just vacated
        issue_and_wait(g + 3);
        mma(lo, 0, 4);
        __builtin_amdgcn_s_barrier();
    };

    // ---- prologue: half-tiles 0..A-1 in flight; 0 and 1 landed; R-lo(0) read ----
#pragma unroll
    for (int q = 0; q < AHEAD; ++q)
        if (q < total) issue(q);
    wait_halftiles(min(AHEAD - 1, total - 1) - 1);
    __builtin_amdgcn_s_barrier();
    read_r(rx, smem);
    if (wl == 1) __builtin_amdgcn_s_barrier();  // stagger: waves 4-7 run one barrier behind

    for (int t = 0; t < nk; t += 2) {
        ktile(t, rx, ry);
        if (t + 1 < nk) ktile(t + 1, ry, rx);
    }
    if (wl == 0) __builtin_amdgcn_s_barrier();  // pairs with the stagger barrier of waves 4-7

    }
    if constexpr (COARSE) c2 = stamp();
    if constexpr (STAMP) {
        if (blockIdx.x == 0 && lane == 0) {
            g_gemm256_dbg[wave * 4 + 0] = dbg_load; g_gemm256_dbg[wave * 4 + 1] = dbg_lbar;
            g_gemm256_dbg[wave * 4 + 2] = dbg_mma; g_gemm256_dbg[wave * 4 + 3] = dbg_mbar;
        }
    }
    // ---- epilogue: lane group g owns 16 contiguous R rows, one L row per fragment ----
    const int gq = lane >> 4;
    const long rbase = r0 + wr * 64 + gq * 16;
    typename Epi::template Ctx<16> ctx;
    epi.template prep<16>(ctx, rbase);
    if constexpr (epi_has_preload<Epi>::value) {
        // fp32 read-modify-write epilogue (residual stream), staged through LDS.  In accumulator layout a wave
        // instruction touches 16 rows x four 16-B pieces: every load/store moved 1 KiB but opened 32 cache lines, and
        // the 512-KB RMW of a tile took as long as the whole K loop of the proj GEMM (tools/stamp_gemm_coarse.py).
        // Each wave transposes its 64 x 128 tile through a private 32-row x 272-B LDS buffer (the DMA ring is idle
        // now; no barrier: a wave only reads what it wrote and LDS ops of one wave complete in order) so that one
        // instruction covers 4 rows x 256 contiguous bytes.
        constexpr int PITCH = 68;  // floats per staged row (64 + 4 pad)
        float* stg = (float*)smem + wave * 32 * PITCH;
        const int mrow = lane >> 4, ncol = (lane & 15) * 4;
        const long nglob = r0 + wr * 64 + ncol;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    float4 d;
                    d.x = ctx.gamma[f * 4 + 0] * (acc[f][2 * q + bb][0] + ctx.bias[f * 4 + 0]);
                    d.y = ctx.gamma[f * 4 + 1] * (acc[f][2 * q + bb][1] + ctx.bias[f * 4 + 1]);
                    d.z = ctx.gamma[f * 4 + 2] * (acc[f][2 * q + bb][2] + ctx.bias[f * 4 + 2]);
                    d.w = ctx.gamma[f * 4 + 3] * (acc[f][2 * q + bb][3] + ctx.bias[f * 4 + 3]);
                    *(float4*)(stg + (16 * bb + (lane & 15)) * PITCH + 16 * gq + 4 * f) = d;
                }
            float4 xv[8];
            const long mbase = l0 + wl * 128 + 32 * q + mrow;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long m = mbase + 4 * i;
                xv[i] = (Epi::ACCUM && m < epi.m_valid && nglob < epi.n_valid) ? ld_stream(epi.x + m * epi.ldx + nglob) : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long m = mbase + 4 * i;
                const float4 d = *(const float4*)(stg + (4 * i + mrow) * PITCH + ncol);
                if (m < epi.m_valid && nglob < epi.n_valid) {
                    float4 o;
                    o.x = xv[i].x + d.x; o.y = xv[i].y + d.y; o.z = xv[i].z + d.z; o.w = xv[i].w + d.w;
                    st_stream(epi.x + m * epi.ldx + nglob, o);
                }
            }
        }
    } else if constexpr (epi_has_produce<Epi>::value && VARIANT != 8) {
        // bf16 row-major epilogues, staged through LDS like the fp32 one: in accumulator layout every 16-B store of a wave
        // instruction lands in a different half-used cache line (16 rows x 4 pieces); staged, an instruction writes whole
        // 128-B (BF16) / 64-B (SwiGLU) row segments.  Per wave: 32 rows x (ROWB + 16) bytes of the idle DMA ring.
        constexpr int O16 = Epi::OUT16, ROWB = 4 * O16 * 2, PITCHB = ROWB + 16, LPR = ROWB / 16, RPI = 64 / LPR;
        char* stg = smem + wave * 32 * PITCHB;
        const int srow = lane / LPR, spiece = lane % LPR;
        const long ocol = ((r0 + wr * 64) >> Epi::OUT_SHIFT) + spiece * 8;       // first output column of this lane's 16 B
        const long ovalid = epi.n_valid >> Epi::OUT_SHIFT;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                float v[16];
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[f * 4 + e] = acc[f][2 * q + bb][e];
                uint32_t w[O16 / 2];
                epi.produce(ctx, v, w);
                char* dst = stg + (16 * bb + (lane & 15)) * PITCHB + gq * (O16 * 2);
                if constexpr (O16 == 16) {
                    *(uint4*)dst = uint4{w[0], w[1], w[2], w[3]};
                    *(uint4*)(dst + 16) = uint4{w[4], w[5], w[6], w[7]};
                } else {
                    *(uint4*)dst = uint4{w[0], w[1], w[2], w[3]};
                }
            }
#pragma unroll
            for (int i = 0; i < 32 / RPI; ++i) {
                const int row = RPI * i + srow;
                const long m = l0 + wl * 128 + 32 * q + row;
                const uint4 d = *(const uint4*)(stg + row * PITCHB + spiece * 16);
                if (m < epi.m_valid && ocol < ovalid) {
                    // 128-B row segments stream past the L2; the 64-B segments of the gated output are left to the L2's write
                    // combining (as streaming stores they reached memory as half lines: WRITE_SIZE 1.31 GB for 1.08 GB written)
                    if constexpr (O16 == 16) st_stream(epi.out + m * epi.ldc + ocol, d);
                    else *(uint4*)(epi.out + m * epi.ldc + ocol) = d;
                }
            }
        }
    } else {
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const long l = l0 + wl * 128 + b * 16 + (lane & 15);
        float v[16];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[f * 4 + e] = acc[f][b][e];
        epi.template store<16>(ctx, rbase, l, v);
    }
    }
    if constexpr (COARSE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long c3 = stamp();
        if (blockIdx.x == 8 * 20 && lane == 0) {  // a block in the middle of an XCD's first round
            g_gemm256_dbg[wave * 4 + 0] = c1 - c0; g_gemm256_dbg[wave * 4 + 1] = c2 - c1;
            g_gemm256_dbg[wave * 4 + 2] = c3 - c2; g_gemm256_dbg[wave * 4 + 3] = c3 - c0;
        }
    }
}

}  // namespace cvx
