// bf16 MFMA tile GEMM core for gfx950:  T[r][l] = sum_k R[r][k] * L[l][k]   (both operands K-contiguous)
//
// One kernel template serves every dense contraction on the hot path (ViT linears, patch embed,
// the head's 1x1x1 / dilated 3x3x3 / transposed convolutions as implicit GEMMs).
//
//   * "R side": the operand whose row index lands in the MFMA accumulator REGISTERS
//     (v_mfma_f32_16x16x32_bf16 A operand), "L side": the operand whose row index lands on LANES.
//     NREG orientation: R = weights [N][K], L = activations [M][K]  -> each lane owns 4*FR CONTIGUOUS
//     output features of one activation row -> 16/32-byte row-major stores, 128-B lines per row.
//     MREG orientation: R = activations, L = weights -> each lane owns contiguous ROWS of one feature
//     (used to write V transposed for the attention kernel).
//   * LDS tiles are [rows][64] bf16 (128-B rows), filled by global_load_lds_dwordx4 (LDS-DMA, no VGPR
//     round trip).  The DMA destination is lane-linear, so the bank swizzle (16-B chunk ^= (row>>1)&7,
//     conflict-free for ds_read_b128 over 16 consecutive rows) and the R-side row permutation sigma
//     (which makes a lane's accumulator registers contiguous in the output) are both applied to the
//     per-lane SOURCE address; reads apply the same XOR.
//   * K loop: double-buffered, one barrier per 64-deep K tile (next tile's DMA issued before the MFMAs).
#pragma once
#include "common.h"
#include <type_traits>

namespace cvx {

constexpr int BK = 64;
constexpr int GEMM_THREADS = 256;

__device__ __forceinline__ int swz_chunk(int row) { return (row >> 1) & 7; }

// LDS row -> source row inside a group of 16*FRG rows: lane group g=(i>>2) ends up holding source rows
// [g*4*FRG, (g+1)*4*FRG) of the group, ordered (frag, reg).
template <int FRG>
__device__ __forceinline__ int sigma_row(int lds_row) {
    if constexpr (FRG == 0) {
        return lds_row;
    } else {
        constexpr int G = 16 * FRG;
        const int grp = lds_row / G, in = lds_row % G;
        const int f = in >> 4, i = in & 15;
        return grp * G + (i >> 2) * (4 * FRG) + 4 * f + (i & 3);
    }
}

// ------------------------------------------------------------------------------------------------
// Tile loaders.  issue(lds_tile, kt) starts the LDS-DMA of K tile kt into lds_tile ([ROWS][64] bf16).
// ------------------------------------------------------------------------------------------------
template <int ROWS, int FRG>
struct PlainLoader {
    static constexpr int NJ = (ROWS * 8 + GEMM_THREADS - 1) / GEMM_THREADS;
    const uint16_t* base;  // tile origin: matrix + row0*ld
    uint32_t off[NJ];      // per-thread BYTE offsets (row*ld + chunk*8) * 2: saddr-form LDS-DMA (common.h)
    int wave;

    __device__ __forceinline__ void init(const uint16_t* mat, long ld, long row0, int tid) {
        base = mat + row0 * ld;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = j * GEMM_THREADS + tid;
            const int row = c >> 3, pos = c & 7;
            off[j] = (uint32_t)(sigma_row<FRG>(row) * ld + ((pos ^ swz_chunk(row)) << 3)) * 2u;
        }
    }
    __device__ __forceinline__ void issue(char* lds_tile, int kt) const {
        const uint16_t* b = base + kt * BK;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if ((ROWS * 8) % GEMM_THREADS != 0 && (j * GEMM_THREADS + wave * 64) >= ROWS * 8) break;
            glds16_saddr(b, off[j], lds_addr(lds_tile) + (j * GEMM_THREADS + wave * 64) * 16);
        }
    }
};

// Implicit-GEMM gather for a 3x3x3 convolution with dilation (dil,1,1), zero "same" padding, over a
// channels-last volume in[D][H][W][C] (bf16).  Row = output voxel, K index = tap*C + c (tap = (kz*3+ky)*3+kx).
// Out-of-volume taps and K padding read from a zero page.
template <int ROWS>
struct Conv3Loader {
    static constexpr int NJ = (ROWS * 8 + GEMM_THREADS - 1) / GEMM_THREADS;
    const uint16_t* in;
    const uint16_t* zero;
    int C, D, H, W, dil;
    int z[NJ], y[NJ], x[NJ];
    int chunk[NJ];  // source 16-B chunk within the 64-wide K tile (swizzle folded in)
    bool rowok[NJ];
    int wave;
    // C % 64 == 0 (every layer this loader still serves: the few-channel ones run in conv_halo.hip): a K tile lies inside ONE
    // tap, so the tap, its (dz, dy, dx) and the channel offset are wave-uniform scalar work and a piece costs three adds, the
    // bounds test and one 64-bit multiply-add.  The general form below divides by C and by 9 / 3 PER PIECE: ~60 vector
    // instructions x 8 pieces per K tile against 32 MFMAs -- address arithmetic, not the L2, was what bounded sb1.c1 (658 TFLOP/s).
    bool tap_uniform;
    uint32_t inv_kpt;  // ceil(2^20 / (C / 64)): tap = kt * inv_kpt >> 20, exact for kt < 27 * C / 64 <= 864
    int lin[NJ];       // linear voxel index of the piece's row

    __device__ __forceinline__ void init(const uint16_t* in_, const uint16_t* zero_, int C_, int D_, int H_, int W_,
                                         int dil_, long row0, long nvox, int tid) {
        in = in_; zero = zero_; C = C_; D = D_; H = H_; W = W_; dil = dil_;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        tap_uniform = (C & 63) == 0 && C <= 2048 && nvox < (1L << 31);
        inv_kpt = tap_uniform ? ((1u << 20) + (uint32_t)(C >> 6) - 1u) / (uint32_t)(C >> 6) : 0u;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = j * GEMM_THREADS + tid;
            const int row = c >> 3, pos = c & 7;
            chunk[j] = pos ^ swz_chunk(row);
            long v = row0 + row;
            rowok[j] = v < nvox;
            if (!rowok[j]) v = 0;
            lin[j] = (int)v;
            x[j] = (int)(v % W);
            const long t = v / W;
            y[j] = (int)(t % H);
            z[j] = (int)(t / H);
        }
    }
    __device__ __forceinline__ void issue(char* lds_tile, int kt) const {
        if (tap_uniform) {
            const int tap = (int)(((uint32_t)kt * inv_kpt) >> 20);
            const int c0 = kt * BK - tap * C;
            const int kz = tap / 9, r9 = tap - kz * 9, ky = r9 / 3, kx = r9 - ky * 3;
            const int dz = (kz - 1) * dil, dy = ky - 1, dx = kx - 1;
            const int dlin = (dz * H + dy) * W + dx;
            const bool tapok = tap < 27;  // (K padding: zero weights, zero data)
            const uint16_t* base = in + c0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if ((ROWS * 8) % GEMM_THREADS != 0 && (j * GEMM_THREADS + wave * 64) >= ROWS * 8) break;
                const bool ok = rowok[j] && tapok && (unsigned)(z[j] + dz) < (unsigned)D && (unsigned)(y[j] + dy) < (unsigned)H &&
                                (unsigned)(x[j] + dx) < (unsigned)W;
                const uint16_t* g = ok ? base + ((long)(lin[j] + dlin) * C + chunk[j] * 8) : zero;
                glds16_vaddr(g, lds_addr(lds_tile) + (j * GEMM_THREADS + wave * 64) * 16);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if ((ROWS * 8) % GEMM_THREADS != 0 && (j * GEMM_THREADS + wave * 64) >= ROWS * 8) break;
            const int kk = kt * BK + chunk[j] * 8;
            const int tap = kk / C, c = kk - tap * C;
            const int kz = tap / 9, r9 = tap - kz * 9, ky = r9 / 3, kx = r9 - ky * 3;
            const int zz = z[j] + (kz - 1) * dil, yy = y[j] + ky - 1, xx = x[j] + kx - 1;
            const bool ok = rowok[j] && tap < 27 && (unsigned)zz < (unsigned)D && (unsigned)yy < (unsigned)H &&
                            (unsigned)xx < (unsigned)W;
            const uint16_t* g = ok ? in + (((long)zz * H + yy) * W + xx) * C + c : zero;
            glds16_vaddr(g, lds_addr(lds_tile) + (j * GEMM_THREADS + wave * 64) * 16);
        }
    }
};

// ------------------------------------------------------------------------------------------------
// Kernel
// ------------------------------------------------------------------------------------------------
// STAGES > 2: a ring of STAGES K tiles with STAGES - 1 in flight and counted waits -- for launches of at most one workgroup
// per CU (the tails of the persistent kernel), where nothing else on the CU overlaps the DMA round trip of a K tile
template <int BR_, int BL_, int WAVES_R_, int STAGES_ = 2>
struct TileCfg {
    static constexpr int BR = BR_, BL = BL_, WAVES_R = WAVES_R_, WAVES_L = 4 / WAVES_R_, STAGES = STAGES_;
    static constexpr int WR = BR / WAVES_R, WL = BL / WAVES_L;
    static constexpr int FR = WR / 16, FL = WL / 16;
    static constexpr int FRG = FR > 4 ? 4 : FR;  // sigma group = min(wave tile, 64) rows
    static constexpr int TILE_R_BYTES = BR * 128, TILE_L_BYTES = BL * 128;
    static constexpr int STAGE_BYTES = TILE_R_BYTES + TILE_L_BYTES;
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
    static_assert(WR % 16 == 0 && WL % 16 == 0, "wave tile must be a multiple of 16");
    static_assert(FR <= 4 || FR % 4 == 0, "R wave tile > 64 must be a multiple of 64");
};

// XCD-aware, grouped tile order: blocks b and b+8 share an XCD (private L2), so each XCD gets a contiguous
// chunk of the logical tile sequence (bijective for any grid size); inside the sequence, tiles sweep the
// R dimension within bands of GROUP_L L-tiles so concurrently resident blocks share operand panels.
__device__ int g_tile_group_l = 8;  // tuning knob (cvx_set_option "tile_group_l")
__device__ __forceinline__ void tile_coords(int bid, int nblk, int tiles_r, int tiles_l, int& tr, int& tl) {
    const int q = nblk >> 3, rem = nblk & 7, xcd = bid & 7;
    const int id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    const int GROUP_L = g_tile_group_l;
    const int per_band = GROUP_L * tiles_r;
    const int band = id / per_band, in = id - band * per_band;
    const int l_first = band * GROUP_L;
    const int gl = min(GROUP_L, tiles_l - l_first);
    tl = l_first + in % gl;
    tr = in / gl;
}

template <class Cfg, class LoaderR, class LoaderL, class Epi>
__device__ __forceinline__ void gemm_tile_body(const LoaderR& ldr, const LoaderL& ldl, const Epi& epi, int nk,
                                               long r0, long l0, char* smem) {
    constexpr int FR = Cfg::FR, FL = Cfg::FL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr0 = (wave / Cfg::WAVES_L) * Cfg::WR, wl0 = (wave % Cfg::WAVES_L) * Cfg::WL;

    f32x4 acc[FR][FL];
#pragma unroll
    for (int a = 0; a < FR; ++a)
#pragma unroll
        for (int b = 0; b < FL; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane fragment read offsets (k-step 0); k-step 1 is the same offset ^ 64
    const int sw = (lane & 15) >> 1;
    const int fo = (lane & 15) * 128 + (((lane >> 4) ^ sw) << 4);
    const int offR = wr0 * 128 + fo;
    const int offL = Cfg::TILE_R_BYTES + wl0 * 128 + fo;

    auto mfma_tile = [&](const char* cur) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 rf[FR], lf[FL];
#pragma unroll
            for (int a = 0; a < FR; ++a) rf[a] = *(const bf16x8*)(cur + ((offR + a * 2048) ^ (ks << 6)));
#pragma unroll
            for (int b = 0; b < FL; ++b) lf[b] = *(const bf16x8*)(cur + ((offL + b * 2048) ^ (ks << 6)));
            // (issue order: the B operand -- the L-side fragment -- is the one kept across consecutive MFMAs; see gemm256p.h)
#ifndef CVX_MMA_ORDER_AB
#pragma unroll
            for (int b = 0; b < FL; ++b)
#pragma unroll
                for (int a = 0; a < FR; ++a)
                    acc[a][b] = mfma16x16x32<epi_is_f16<Epi>::value>(rf[a], lf[b], acc[a][b]);
#else
#pragma unroll
            for (int a = 0; a < FR; ++a)
#pragma unroll
                for (int b = 0; b < FL; ++b)
                    acc[a][b] = mfma16x16x32<epi_is_f16<Epi>::value>(rf[a], lf[b], acc[a][b]);
#endif
        }
    };
    if constexpr (Cfg::STAGES > 2) {
        // K tiles kt+1 .. kt+S-2 stay in flight while tile kt is multiplied; every wave issues the same PER pieces per K tile
        // (both loaders divide evenly), so "tile kt has landed" is vmcnt(PER x tiles issued after it)
        constexpr int S = Cfg::STAGES, PER = LoaderR::NJ + LoaderL::NJ;
        static_assert((Cfg::BR * 8) % GEMM_THREADS == 0 && (Cfg::BL * 8) % GEMM_THREADS == 0, "counted waits need whole passes");
        static_assert(S <= 6 && (S - 2) * PER < 64, "vmcnt is a 6-bit field");
        auto stage = [&](int kt) { return smem + (kt % S) * Cfg::STAGE_BYTES; };
        for (int kt = 0; kt < S - 1 && kt < nk; ++kt) {
            ldr.issue(stage(kt), kt);
            ldl.issue(stage(kt) + Cfg::TILE_R_BYTES, kt);
        }
        for (int kt = 0; kt < nk; ++kt) {
            const int ahead = min(S - 2, nk - 1 - kt);
            auto wait_ahead = [&](auto tag) {  // (tag::value tiles may stay in flight)
                constexpr int N = decltype(tag)::value * PER;
                if constexpr (N < 64) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
            };
            if (ahead >= 4) wait_ahead(std::integral_constant<int, 4>{});
            else if (ahead == 3) wait_ahead(std::integral_constant<int, 3>{});
            else if (ahead == 2) wait_ahead(std::integral_constant<int, 2>{});
            else if (ahead == 1) wait_ahead(std::integral_constant<int, 1>{});
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // tile kt landed for every wave, and every wave is done with tile kt-1 (whose slot is refilled next)
            if (kt + S - 1 < nk) {
                ldr.issue(stage(kt + S - 1), kt + S - 1);
                ldl.issue(stage(kt + S - 1) + Cfg::TILE_R_BYTES, kt + S - 1);
            }
            mfma_tile(stage(kt));
        }
    } else {
    ldr.issue(smem, 0);
    ldl.issue(smem + Cfg::TILE_R_BYTES, 0);
    // the LDS-DMA is issued from inline asm (saddr form, common.h), which hipcc's own vmcnt bookkeeping does not see:
    // the drain before the barrier is explicit
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * Cfg::STAGE_BYTES;
        if (kt + 1 < nk) {
            char* nxt = smem + ((kt + 1) & 1) * Cfg::STAGE_BYTES;
            ldr.issue(nxt, kt + 1);
            ldl.issue(nxt + Cfg::TILE_R_BYTES, kt + 1);
        }
        mfma_tile(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // next tile landed and everyone is done reading `cur`
    }
    }

    // epilogue: lane (g = lane>>4) owns R rows [g*4*FRG, +4*FRG) of each 16*FRG-row group, one L row per frag
    constexpr int FRG = Cfg::FRG, NG = FR / FRG;
    const int g = lane >> 4;
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
        const long rbase = r0 + wr0 + grp * 16 * FRG + g * 4 * FRG;
        typename Epi::template Ctx<4 * FRG> ctx;
        epi.template prep<4 * FRG>(ctx, rbase);  // per-R constants (bias, gamma) loaded once per lane
#pragma unroll
        for (int b = 0; b < FL; ++b) {
            const long l = l0 + wl0 + b * 16 + (lane & 15);
            float v[4 * FRG];
#pragma unroll
            for (int f = 0; f < FRG; ++f)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[f * 4 + e] = acc[grp * FRG + f][b][e];
            epi.template store<4 * FRG>(ctx, rbase, l, v);
        }
    }
}

}  // namespace cvx
