// 256x256 bf16 MFMA GEMM tile for gfx950, FOUR waves = one wave per SIMD, each owning a 128x128 sub-tile
// (256 accumulator registers in the AGPR half of the unified 512-entry file).   T[r][l] = sum_k R[r][k] * L[l][k]
//
// Why this shape: s_memtime stamps of the 8-wave kernel (gemm256.h, tools/stamp_gemm.py) showed ~110 cycles of
// s_barrier latency on each of its 8 barriers per K tile and a 35 % stretch of every 16-MFMA burst by the SIMD partner's
// load instructions -- the matrix pipe was busy 57 % of the time.  Here a wave never shares its SIMD: its own ds_reads
// and LDS-DMA issues fill the issue slots between its own MFMAs, fragments are double-buffered in registers one k-step
// ahead, and there is ONE barrier per 64 MFMAs.
//
//   * k-step = 32 (one v_mfma_f32_16x16x32_bf16 deep).  LDS is a ring of 4 sub-stages of 32 KiB
//     ([256 R rows][32] + [256 L rows][32] bf16, 64-B rows, 16-B chunk index ^= (-(row>>2))&3: conflict-free
//     ds_read_b128 over 16 consecutive rows).
//   * k-step j:  counted vmcnt wait (own LDS-DMA pieces of sub-stage j+1) -> s_barrier -> issue the DMA of sub-stage
//     j+4 into the slot of sub-stage j (whose fragments every wave already holds in registers) -> 64 MFMAs on the
//     fragments of k-step j interleaved with the 16 ds_read_b128 of k-step j+1.
//     Three sub-stages (96 KiB) are in flight across barriers; a DMA has three k-steps (~3000 cycles) to land.
//   Hazards.  RAW: sub-stage j+1 is read after the barrier of k-step j, ahead of which every wave waited for its own
//   DMA pieces of it.  WAR: slot j&3 is rewritten after that same barrier; its last ds_reads were issued in k-step j-1
//   and waited for (lgkmcnt) before the first MFMA of k-step j, i.e. before the wave arrived at the barrier.
#pragma once
#include "gemm_core.h"
#include <type_traits>

namespace cvx {

constexpr int G4W_THREADS = 256;
constexpr int G4W_KS = 32;                            // k-step depth (bf16 elements)
constexpr int G4W_SUB_BYTES = 2 * 256 * G4W_KS * 2;   // R + L sub-tiles of one k-step: 32 KiB
constexpr int G4W_RING = 4;
constexpr int G4W_LDS_BYTES = G4W_RING * G4W_SUB_BYTES;

__device__ __forceinline__ int swz64(int row) { return (-(row >> 2)) & 3; }  // chunk XOR for 64-B rows

template <int VARIANT, class Epi>
__device__ __forceinline__ void gemm4w_body(const uint16_t* __restrict__ Rmat, long ldr, const uint16_t* __restrict__ Lmat,
                                            long ldl, int nks, long r0, long l0, const Epi& epi, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wl = wave & 1;

    // ---- LDS-DMA: a wave-instruction writes 1 KiB = 16 rows x 64 B; sub-tile = 16 instructions, 4 per wave ----
    uint32_t offR[4], offL[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 4 + wave) * 16 + (lane >> 2);  // LDS row of this lane's 16-B piece
        const int ch = ((lane & 3) ^ swz64(row)) << 3;      // source chunk (elements)
        offR[i] = (uint32_t)(sigma_row<4>(row) * ldr + ch) * 2u;  // bytes (saddr-form DMA)
        offL[i] = (uint32_t)(row * ldl + ch) * 2u;
    }
    const uint16_t* Rb = Rmat + r0 * ldr;
    const uint16_t* Lb = Lmat + l0 * ldl;
    auto issue = [&](int j) {  // sub-stage j -> ring slot j & 3
        const uint32_t dst = lds_addr(smem) + (j & (G4W_RING - 1)) * G4W_SUB_BYTES;
        const uint16_t* rs = Rb + j * G4W_KS;
        const uint16_t* ls = Lb + j * G4W_KS;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_saddr(rs, offR[i], dst + (i * 4 + wave) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_saddr(ls, offL[i], dst + 16384 + (i * 4 + wave) * 1024);
    };

    // ---- fragment reads: frag f of the wave's 128 rows: row = w*128 + f*16 + (lane&15), chunk = (lane>>4) ^ swz ----
    const int fo = (lane & 15) * 64 + (((lane >> 4) ^ swz64(lane & 15)) << 4);
    const int foR = wr * 128 * 64 + fo;
    const int foL = 16384 + wl * 128 * 64 + fo;

    f32x4 acc[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 ra[8], la[8], rb[8], lb[8];  // fragment double buffer (k-step j / j+1)

    auto read_frags = [&](bf16x8 (&r)[8], bf16x8 (&l)[8], int j) {
        const char* st = smem + (j & (G4W_RING - 1)) * G4W_SUB_BYTES;
#pragma unroll
        for (int f = 0; f < 8; ++f) r[f] = *(const bf16x8*)(st + foR + f * 1024);
#pragma unroll
        for (int f = 0; f < 8; ++f) l[f] = *(const bf16x8*)(st + foL + f * 1024);
    };
    // The 64 accumulators are pinned to the AGPR half of the register file through the "a" constraint: with the
    // builtin, hipcc kept part of them in VGPRs and emitted ~200 v_accvgpr copies per k-step.  Operands come from
    // ds_reads the compiler waits for itself; consecutive MFMAs chain on C = D (no wait states needed).
    auto mfma = [&](f32x4& c, const bf16x8& a, const bf16x8& b) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    };
    // k-step j: `cur` holds its fragments; `nxt` receives those of k-step j+1
    auto kstep = [&](int j, const bf16x8 (&cr)[8], const bf16x8 (&cl)[8], bf16x8 (&nr)[8], bf16x8 (&nl)[8], auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        // sub-stage j+1 (8 pieces per wave) must have landed; j+2, j+3 may stay in flight
        // (lgkmcnt(0): this wave's ds_reads of sub-stage j, issued a whole k-step ago, are complete -> its slot may be
        //  rewritten by whoever passes the barrier first)
        if constexpr (STEADY && VARIANT != 1) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool do_issue = VARIANT != 1 && (STEADY || j + 4 < nks), do_read = STEADY || j + 1 < nks;
        const uint32_t dst = lds_addr(smem) + ((j + 4) & (G4W_RING - 1)) * G4W_SUB_BYTES;
        const char* nst = smem + ((j + 1) & (G4W_RING - 1)) * G4W_SUB_BYTES;
        const uint16_t* rs = Rb + (VARIANT == 2 ? 0 : (j + 4) * G4W_KS);  // VARIANT 1/2: timing-only ablations (no DMA / L2-resident DMA)
        const uint16_t* ls = Lb + (VARIANT == 2 ? 0 : (j + 4) * G4W_KS);
        // hand interleave: per row of 8 MFMAs one LDS-DMA piece of sub-stage j+4 and two fragment reads of k-step j+1
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            if (do_issue) {
                if (a < 4) glds16_saddr(rs, offR[a], dst + (a * 4 + wave) * 1024);
                else glds16_saddr(ls, offL[a - 4], dst + 16384 + ((a - 4) * 4 + wave) * 1024);
            }
            if (do_read) {
                nr[a] = *(const bf16x8*)(nst + foR + a * 1024);
                nl[a] = *(const bf16x8*)(nst + foL + a * 1024);
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) mfma(acc[a][b], cr[a], cl[b]);
        }
    };

    // ---- prologue: sub-stages 0..3 in flight, 0 landed, fragments of k-step 0 in registers ----
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < nks) issue(j);
    if (nks >= 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(ra, la, 0);

    int j = 0;
    // steady state needs j+4 < nks for the issue and exact counts: after issue(j+4) there are sub-stages j+1..j+4
    // outstanding at most; the wait at the top of k-step j (before that issue) sees j+1, j+2, j+3 -> vmcnt(16)
    for (; j + 5 < nks; j += 2) {
        kstep(j, ra, la, rb, lb, std::true_type{});
        kstep(j + 1, rb, lb, ra, la, std::true_type{});
    }
    for (; j + 1 < nks; j += 2) {
        kstep(j, ra, la, rb, lb, std::false_type{});
        kstep(j + 1, rb, lb, ra, la, std::false_type{});
    }
    if (j < nks) kstep(j, ra, la, rb, lb, std::false_type{});

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results -> v_accvgpr_read hazard (inline-asm MFMAs are not padded)
    // ---- epilogue: 2 sigma groups of 64 R rows; lane group g owns 16 contiguous R rows of each ----
    const int gq = lane >> 4;
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
        const long rbase = r0 + wr * 128 + grp * 64 + gq * 16;
        typename Epi::template Ctx<16> ctx;
        epi.template prep<16>(ctx, rbase);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const long l = l0 + wl * 128 + b * 16 + (lane & 15);
            float v[16];
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[f * 4 + e] = acc[grp * 4 + f][b][e];
            epi.template store<16>(ctx, rbase, l, v);
        }
    }
}

}  // namespace cvx
