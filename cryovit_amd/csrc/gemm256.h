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
//   * LDS holds a ring of 8 half-tiles (2 K tiles, 128 KiB).  Phase g issues the LDS-DMA of half-tile g+A (A = 5):
//     3-4 half-tiles are always in flight ACROSS barriers behind a counted s_waitcnt vmcnt(6/8) -- never vmcnt(0)
//     in the steady state.
//   * Two barriers per phase split it into a LOAD segment (ds_read fragments, issue DMA, counted wait) and an MFMA
//     segment.  Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA segment while
//     its partner is in its load segment: the matrix pipe never waits for LDS latency.
//   Hazards (one barrier of stagger included):
//     RAW  half-tile q is waited for (every wave, its own DMA pieces) at the END of the load segment of phase
//          rd(q)-1, i.e. ahead of a barrier that every reader passes before its first ds_read of q.
//     WAR  slot of q is rewritten by DMA(q+8) issued in load segment q+3 >= rd(q)+3: the late group's reads of q
//          completed (lgkmcnt(0) opens its MFMA segment) two barriers earlier.
#pragma once
#include "gemm_core.h"

namespace cvx {

constexpr int G256_THREADS = 512;
constexpr int G256_HALF_BYTES = 128 * 128;            // 128 rows x 64 bf16
constexpr int G256_LDS_BYTES = 8 * G256_HALF_BYTES;   // ring of 8 half-tiles
constexpr int G256_AHEAD = 5;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else static_assert(N == 0, "unsupported count");
}

template <class Epi>
__device__ __forceinline__ void gemm256_body(const uint16_t* __restrict__ Rmat, long ldr, const uint16_t* __restrict__ Lmat,
                                             long ldl, int nk, long r0, long l0, const Epi& epi, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 3, wl = wave >> 2;  // waves 0-3 / 4-7 = L half 0 / 1 = the two stagger groups
    const int total = 4 * nk;

    // ---- LDS-DMA source offsets: 2 x 16-B pieces per thread per half-tile ----
    uint32_t offR[2][2], offL[2][2];  // [half][piece], elements relative to the tile origin, K tile 0
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = j * G256_THREADS + tid;
        const int hr = c >> 3, ch = ((c & 7) ^ swz_chunk(hr)) << 3;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int tr = (hr >> 5) * 64 + half * 32 + (hr & 31);   // R tile row: wave hr>>5, rows [32*half, +32)
            offR[half][j] = (uint32_t)(sigma_row<4>(tr) * ldr + ch);
            const int tl = (hr >> 6) * 128 + half * 64 + (hr & 63);  // L tile row: wave hr>>6, rows [64*half, +64)
            offL[half][j] = (uint32_t)(tl * ldl + ch);
        }
    }
    const uint16_t* Rb = Rmat + r0 * ldr;
    const uint16_t* Lb = Lmat + l0 * ldl;

    auto issue = [&](int q) {  // half-tile q -> ring slot q & 7
        const int kt = q >> 2, kind = q & 3;
        char* dst = smem + (q & 7) * G256_HALF_BYTES + wave * 1024;
        const uint16_t* src = ((kind & 1) ? Lb : Rb) + kt * BK;
        const uint32_t o0 = (kind & 1) ? offL[kind >> 1][0] : offR[kind >> 1][0];
        const uint32_t o1 = (kind & 1) ? offL[kind >> 1][1] : offR[kind >> 1][1];
        glds16(src + o0, dst);
        glds16(src + o1, dst + G256_THREADS * 16);
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
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a0 + a][b0 + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r[a][ks], lf[b][ks], acc[a0 + a][b0 + b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: half-tiles 0..A-1 in flight, 0 and 1 landed ----
#pragma unroll
    for (int q = 0; q < G256_AHEAD; ++q)
        if (q < total) issue(q);
    if (total > G256_AHEAD) wait_vmcnt<6>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (wl == 1) __builtin_amdgcn_s_barrier();  // stagger: waves 4-7 run one barrier behind

    for (int t = 0; t < nk; ++t) {
        const char* st = smem + (t & 1) * 4 * G256_HALF_BYTES;
        const int g = 4 * t;
        const bool steady = g + 3 + G256_AHEAD < total;  // all four issues of this K tile exist: counted waits are exact
        // ---- phase 0: (R-lo, L-lo) ----
        read_r(rlo, st);
        read_l(st + G256_HALF_BYTES);
        if (g + G256_AHEAD < total) issue(g + G256_AHEAD);
        if (steady) wait_vmcnt<6>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        mma(rlo, 0, 0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 1: (R-hi, L-lo) ----
        read_r(rhi, st + 2 * G256_HALF_BYTES);
        if (g + 1 + G256_AHEAD < total) issue(g + 1 + G256_AHEAD);
        if (steady) wait_vmcnt<6>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        mma(rhi, 2, 0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: (R-hi, L-hi) ----
        read_l(st + 3 * G256_HALF_BYTES);
        if (g + 2 + G256_AHEAD < total) issue(g + 2 + G256_AHEAD);
        if (steady) wait_vmcnt<8>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        mma(rhi, 2, 4);
        __builtin_amdgcn_s_barrier();
        // ---- phase 3: (R-lo, L-hi) ----
        if (g + 3 + G256_AHEAD < total) issue(g + 3 + G256_AHEAD);
        if (steady) wait_vmcnt<6>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        mma(rlo, 0, 4);
        __builtin_amdgcn_s_barrier();
    }
    if (wl == 0) __builtin_amdgcn_s_barrier();  // pairs with the stagger barrier of waves 4-7

    // ---- epilogue: lane group g owns 16 contiguous R rows, one L row per fragment ----
    const int gq = lane >> 4;
    const long rbase = r0 + wr * 64 + gq * 16;
    typename Epi::template Ctx<16> ctx;
    epi.template prep<16>(ctx, rbase);
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

}  // namespace cvx
