// Persistent form of the 256x256x64 phase-pipelined bf16 GEMM tile (gemm256.h, schedule "variant 5").
//
// One workgroup per CU walks a static list of output tiles; the LDS-DMA operand stream NEVER stops at a tile boundary:
//
//   * while a tile's last five MFMA phases run, the DMA pieces they issue already belong to the NEXT tile's first K tiles
//     (a workgroup's last tile re-fetches its own first K tile: harmless reads, drained before exit), so no tile but the
//     first pays the ~3.2 k-cycle pipeline fill and no K tile runs on the drained `vmcnt(0)` schedule;
//   * the epilogue (convert / gate / residual update, staged through LDS into whole-row stores) runs with five half-tiles
//     of the next tile in flight.  Its staging buffers are the three ring slots of the finished tile that no DMA targets
//     until the next tile's first phase (L-lo, R-hi, L-hi of the last K tile: 48 KiB, 6 KiB per wave);
//   * `vmcnt` retires loads, stores and LDS-DMA together IN ISSUE ORDER, so the S stores a wave issues in the epilogue (and
//     the constants pieces, below) sit between half-tile 4 and half-tile 5 of the next tile in its queue: the first three
//     counted waits of that tile's first K tile are 4+S / 4+S+NC / 6+S+NC instead of 4 / 4 / 6.  S is exact only when every store executes, i.e. for
//     launches whose tiles are all interior (FULL: M % 256 == 0, no padded output columns): there the stores are
//     unpredicated.  Other launches keep predicated stores and the conservative 4 / 4 / 6 (correct, they only over-wait).
//
// Barrier structure per tile (I = all 8 waves, G0 / G1 = waves 0-3 / 4-7, G1 runs one barrier behind inside the K loop):
//   [G1: I] K loop ... [G0: I]  epilogue  I  -> next tile
// The barrier after the epilogue orders every wave's staging reads before the first DMA into those slots (issued in the
// first MFMA segment of the next tile).
#pragma once
#include "gemm256.h"

#ifndef CVX_BF16_STORE_NT
#define CVX_BF16_STORE_NT 1  // streaming stores for the 16-wide bf16 epilogue (qk, fc1): A/B with -DCVX_BF16_STORE_NT=0
#endif
#ifndef CVX_RESID_LOAD_NT
#define CVX_RESID_LOAD_NT 1  // streaming loads of the fp32 residual stream in the epilogue (A/B: -DCVX_RESID_LOAD_NT=0)
#endif
#if CVX_RESID_LOAD_NT
#define CVX_RESID_LOAD(p) ld_stream(p)
#else
#define CVX_RESID_LOAD(p) ([](const float* q) { const f32x4 v = *(const f32x4*)q; return float4{v[0], v[1], v[2], v[3]}; })(p)
#endif
#ifndef CVX_RESID_STORE_NT
#define CVX_RESID_STORE_NT 1  // streaming stores for the fp32 residual stream (A/B: -DCVX_RESID_STORE_NT=0)
#endif

namespace cvx {

constexpr bool DIRECT8 = true;  // SwiGLU epilogue: store straight from the accumulator layout (A/B switch for tools/)
constexpr int G256P_LDS_BYTES = G256_LDS_BYTES + 8 * 2048;  // ring + two 1-KiB constants pieces (bias, gamma) per wave

// vector-memory operations ONE wave issues in the staged epilogue of one interior tile and that may still be in flight
// when the next tile starts (the residual epilogue's loads are consumed, hence complete, before its last stores)
// MREG orientation (R = activation rows -> accumulator registers, L = weight rows -> lanes): a lane owns 16 consecutive ROWS of
// one output feature; such epilogues (V^T for the attention kernel) declare `static constexpr bool MREG16 = true` and
// `store16(m0, n, v)`, which issues exactly two 16-B stores
template <class E, class = void> struct epi_is_mreg : std::false_type {};
template <class E> struct epi_is_mreg<E, std::enable_if_t<E::MREG16>> : std::true_type {};

template <class E, class = void> struct epi_has_hl : std::false_type {};
template <class E> struct epi_has_hl<E, std::enable_if_t<E::HAS_HL>> : std::true_type {};
template <class E, class = void> struct epi_is_ln : std::false_type {};
template <class E> struct epi_is_ln<E, std::enable_if_t<E::LNFOLD>> : std::true_type {};

template <class Epi> constexpr int epi_stores_per_wave() {
    if constexpr (epi_is_mreg<Epi>::value) return 16;                            // 8 fragments x 2 stores of 8 rows
    else if constexpr (epi_has_hl<Epi>::value) return 33;                        // 8 units x (2 hi + 2 lo) + 1 row-statistics store
#ifdef CVX_LN_EMIT_PROTO
    else if constexpr (epi_has_preload<Epi>::value) return 49;                   // + 2 x 8 bf16 stores + 1 row-statistics store (prototype)
#else
    else if constexpr (epi_has_preload<Epi>::value) return 32;                   // 4 x 8 float4 stores
#endif
    else return 4 * (32 / (64 / (4 * Epi::OUT16 * 2 / 16)));                      // 4 x (32 rows / rows per instruction)
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// position `id` of the XCD-grouped tile sequence -> tile coordinates (the mapping of tile_coords() in gemm_core.h)
__device__ __forceinline__ void tile_from_seq(int id, int tiles_r, int tiles_l, int GROUP_L, int& tr, int& tl) {
    const int per_band = GROUP_L * tiles_r;
    const int band = id / per_band, in = id - band * per_band;
    const int l_first = band * GROUP_L;
    const int gl = min(GROUP_L, tiles_l - l_first);
    tl = l_first + in % gl;
    tr = in / gl;
}

// diagnostic build only (DBG, -DCVX_ABLATION): cycle stamps of one workgroup's first 8 tiles, waves 0 and 4:
// [wave>>2][tile][0..5] = K loop start, K loop end, epilogue start (after the un-stagger barrier), epilogue end, after the
// post-epilogue barriers, end of the tile's first K tile
__device__ unsigned long long g_gemm256p_dbg[2 * 8 * 6];
#ifdef CVX_LN_EMIT_PROTO
// timing prototype (tools/bench_ln_emit.py, never in the product library): the residual epilogue also emits bf16(x) -- what a
// LayerNorm folded into the consuming GEMMs would need -- as 16-B stores after a lane-pair exchange
#endif

template <class Epi, bool FULL, bool DBG = false>
__device__ __forceinline__ void gemm256p_body(const uint16_t* __restrict__ Rmat, long ldr, const uint16_t* __restrict__ Lmat, long ldl,
                                              int nk, int tiles_r, int tiles_l, int group_l, int xcd_stagger, const Epi& epi, char* smem) {
    constexpr bool MREG = epi_is_mreg<Epi>::value;
    static_assert(MREG || epi_has_preload<Epi>::value || epi_has_produce<Epi>::value || epi_has_hl<Epi>::value, "persistent tile: counted-store epilogues only");
    constexpr bool LN = epi_is_ln<Epi>::value;   // LayerNorm folded into this (consuming) GEMM's epilogue
    constexpr bool HL = epi_has_hl<Epi>::value;  // bf16 hi / lo residual stream
    constexpr bool F16 = epi_is_f16<Epi>::value;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave & 3, wl = wave >> 2;
    const int total = 4 * nk;

    // ---- this workgroup's tiles: XCD x = blockIdx & 7 owns a contiguous chunk of the sequence (its L2 sees neighbouring
    //      tiles); the 32 workgroups of an XCD take 32 consecutive positions per round ----
    const int ntiles = tiles_r * tiles_l;
    const int xcd = blockIdx.x & 7, lslot = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int q8 = ntiles >> 3, rem = ntiles & 7;
    const int c0 = xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8;
    const int cnt = q8 + (xcd < rem ? 1 : 0);
    int id = lslot;
    if (id >= cnt) return;  // (uniform for the workgroup; no barrier has been executed)
    // De-phase the XCDs once: tiles take the same time everywhere, so without this all 256 workgroups reach their epilogue
    // together for the whole launch and every tile boundary is a chip-wide store burst (tens of MB) that the in-order vmcnt
    // queue then has to see retired a few phases into the next tile.  The 32 workgroups of ONE XCD stay in step (they
    // stream the same operand panels through their L2); XCD x starts x * xcd_stagger cycles late and keeps that offset.
    if (xcd_stagger > 0 && xcd > 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const unsigned long long want = (unsigned long long)xcd * (unsigned)xcd_stagger;
        while (__builtin_amdgcn_s_memtime() - t0 < want) __builtin_amdgcn_s_sleep(32);
    }

    uint32_t offR[2][2], offL[2][2];  // [half][piece] LDS-DMA source offsets, BYTES relative to the tile origin
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = j * G256_THREADS + tid;
        const int hr = c >> 3, ch = ((c & 7) ^ swz_chunk(hr)) << 3;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int tr_ = (hr >> 5) * 64 + half * 32 + (hr & 31);
            offR[half][j] = (uint32_t)(sigma_row<4>(tr_) * ldr + ch) * 2u;
            const int tl_ = (hr >> 6) * 128 + half * 64 + (hr & 63);
            offL[half][j] = (uint32_t)(tl_ * ldl + ch) * 2u;
        }
    }
    long r0, l0;
    const uint16_t *Rc, *Lc, *Rn, *Ln;  // operand panels of the current / next tile
    // group_l < 0: this workgroup's chunk of the sequence is walked from its END (the launch then finishes where the
    // sequence starts; used for GEMMs whose activation operand was written front to back by the previous kernel, so its last
    // rows are the ones still in the Infinity Cache)
    const bool rev_walk = group_l < 0;
    group_l = rev_walk ? -group_l : group_l;
    auto panels = [&](int pos, const uint16_t*& Rp, const uint16_t*& Lp, long& rr, long& ll) {
        int tr, tl;
        tile_from_seq(c0 + (rev_walk ? cnt - 1 - pos : pos), tiles_r, tiles_l, group_l, tr, tl);
        rr = (long)tr * 256; ll = (long)tl * 256;
        Rp = Rmat + rr * ldr; Lp = Lmat + ll * ldl;
    };
    panels(id, Rc, Lc, r0, l0);
    long rn = r0, ln = l0;
    Rn = Rc; Ln = Lc;
    if (id + per < cnt) panels(id + per, Rn, Ln, rn, ln);

    // Per-column constants of the epilogue (bias, LayerScale gamma) reach the wave through LDS-DMA pieces issued in the tile's
    // FIRST K tile into 1-KiB slots behind the ring: a register load in the epilogue would be the YOUNGEST entry of the
    // in-order vmcnt queue, and waiting for it would drain the five half-tiles in flight for the next tile.
    // One piece per vector: lanes 0-15 carry the wave's 64 columns (16 B each), the other lanes re-read the same bytes.
    // LN consumers: piece 0 = b' and the column sums cs of the wave's features (the two arrays sit cs_off floats apart), piece 1 =
    // the row constants (rstd, -mu * rstd) of the wave's activation rows (8 B per row).
    constexpr int NC = (epi_has_preload<Epi>::value || HL || LN) ? 2 : 1;
    char* cbuf = smem + G256_LDS_BYTES + wave * 2048;
    auto issue_consts = [&](long rr, long ll) {  // (r0, l0) of the tile the constants are for
        int l2 = tid & 63;
        asm volatile("" : "+v"(l2));
        if constexpr (MREG) {  // features sit on the L side: the wave's 128 features = lanes 0-31 x 16 B
            if constexpr (LN) {
                glds16_saddr(epi.bias + ll + wl * 128, (uint32_t)(l2 & 31) * 16u + (uint32_t)(l2 >> 5) * (uint32_t)epi.cs_off * 4u, lds_addr(cbuf));
                glds16_saddr(epi.rowstat + (rr + wr * 64 + epi.m_off) * 2, (uint32_t)(l2 & 31) * 16u, lds_addr(cbuf) + 1024);
            } else {
                glds16_saddr(epi.bias + ll + wl * 128, (uint32_t)(l2 & 31) * 16u, lds_addr(cbuf));
            }
            return;
        }
        const uint32_t voff = (uint32_t)(l2 & 15) * 16u;
        if constexpr (LN) {
            glds16_saddr(epi.bias + rr + wr * 64, voff + (uint32_t)((l2 >> 4) & 1) * (uint32_t)epi.cs_off * 4u, lds_addr(cbuf));
            glds16_saddr(epi.rowstat + (ll + wl * 128) * 2, (uint32_t)l2 * 16u, lds_addr(cbuf) + 1024);
        } else {
            glds16_saddr(epi.bias + rr + wr * 64, voff, lds_addr(cbuf));
            if constexpr (epi_has_preload<Epi>::value || HL) glds16_saddr(epi.gamma + rr + wr * 64, voff, lds_addr(cbuf) + 1024);
        }
    };

    int slot0 = 0;  // ring slot of the current tile's half-tile 0 (0 or 4)
    // One LDS-DMA half-tile = two 1-KiB pieces per wave.  The source is a wave-uniform byte pointer (SGPR pair) + a 32-bit
    // per-lane byte offset (glds16_saddr2, common.h: `global_load_lds_dwordx4 voff, s[base]`, no per-piece VALU work);
    // `slot` is the ring slot counted from the K tile's own group (values 5..8 = the next K tile's L-lo, R-hi, L-hi and the
    // R-lo after that).  Everything that could be selected at run time per piece (which tile, which K tile) is decided by the
    // CALLER once per K tile: scalar work in the MFMA segments is on the loop's critical path (measured: 25 extra SALU
    // instructions per K tile cost 135 of 2810 cycles).
    const uint32_t lds0 = lds_addr(smem) + wave * 1024;  // LDS byte address of this wave's first piece in ring slot 0
    auto dma = [&](const char* src, const uint32_t (&off)[2], const char* st, int slot) {
        const uint32_t dst = lds0 + (((uint32_t)(st - smem) + slot * G256_HALF_BYTES) & (G256_LDS_BYTES - 1));
        glds16_saddr2<G256_THREADS * 16>(src, off[0], off[1], dst);
    };
    const int sw = (lane & 15) >> 1;
    const int fo = (lane & 15) * 128 + (((lane >> 4) ^ sw) << 4);
    const int foR = wr * 32 * 128 + fo, foL = wl * 64 * 128 + fo;

    f32x4 acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
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
    // ZERO: first K tile of a tile after the first -- the k-step-0 MFMAs start from C = 0 (no accumulator clearing pass)
    auto mma = [&](const bf16x8 (&r)[2][2], int a0, int b0, auto zero_tag) {
        constexpr bool ZERO = decltype(zero_tag)::value;
        __builtin_amdgcn_s_setprio(1);
#if !defined(CVX_MMA_ORDER_AB)
        // issue order: the L-side fragment is the operand kept across consecutive MFMAs (b outer, a inner).  Against the order that
        // keeps the R-side fragment for four MFMAs (a outer, b inner: -DCVX_MMA_ORDER_AB) this is +0.3 ... +0.9 % on every ViT-g GEMM
        // in alternating runs on one board (profiles/r03_mma_order.txt); a boustrophedon order measured the same as this one
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    acc[a0 + a][b0 + b] = mfma16x16x32<F16>(r[a][ks], lf[b][ks], (ZERO && ks == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[a0 + a][b0 + b]);
#else
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a0 + a][b0 + b] = mfma16x16x32<F16>(r[a][ks], lf[b][ks], (ZERO && ks == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[a0 + a][b0 + b]);
#endif
        __builtin_amdgcn_s_setprio(0);
    };

    constexpr int S = FULL ? epi_stores_per_wave<Epi>() : 0;
    // One K tile.  Rk1 / Lk1: operand streams at the NEXT K tile (the next tile's K tile 0 after this tile's last one),
    // Rk2: the R stream one K tile further -- wave-uniform byte pointers chosen by the caller.
    auto ktile = [&](const char* st, const char* Rk1, const char* Lk1, const char* Rk2, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int E = FIRST ? S : 0;
        read_r(rlo, st);                                  // ---- phase 0: (R-lo, L-lo) ----
        read_l(st + G256_HALF_BYTES);
        wait_vmcnt<4 + E>();
        __builtin_amdgcn_s_barrier();
        if constexpr (FIRST) issue_consts(r0, l0);            // (NC more entries between the epilogue's stores and half-tile 5)
        dma(Lk1, offL[0], st, 5);
        mma(rlo, 0, 0, first_tag);
        __builtin_amdgcn_s_barrier();
        read_r(rhi, st + 2 * G256_HALF_BYTES);            // ---- phase 1: (R-hi, L-lo) ----
        wait_vmcnt<4 + E + (FIRST ? NC : 0)>();
        __builtin_amdgcn_s_barrier();
        dma(Rk1, offR[1], st, 6);
        mma(rhi, 2, 0, first_tag);
        __builtin_amdgcn_s_barrier();
        read_l(st + 3 * G256_HALF_BYTES);                 // ---- phase 2: (R-hi, L-hi) ----
        wait_vmcnt<6 + E + (FIRST ? NC : 0)>();
        __builtin_amdgcn_s_barrier();
        dma(Lk1, offL[1], st, 7);
        mma(rhi, 2, 4, first_tag);
        __builtin_amdgcn_s_barrier();
        wait_vmcnt<4>();                                  // ---- phase 3: (R-lo, L-hi) ---- (retires the epilogue's stores too)
        __builtin_amdgcn_s_barrier();
        dma(Rk2, offR[0], st, 8);
        mma(rlo, 0, 4, first_tag);
        __builtin_amdgcn_s_barrier();
    };
    // all K tiles of the current tile after the first: the plain ones, then the last two, whose look-ahead crosses into the
    // next tile (peeled so that the loop body carries no per-tile selection at all)
    constexpr long KB = BK * 2;  // bytes per K tile along a row
    auto ktiles_1_to_end = [&]() {
        const char* rc = (const char*)Rc;
        const char* lc = (const char*)Lc;
        const char* rn = (const char*)Rn;
        const char* lnx = (const char*)Ln;
        int par = slot0 >> 2;
        for (int t = 1; t < nk - 2; ++t) {
            par ^= 1;
            ktile(smem + par * 4 * G256_HALF_BYTES, rc + (t + 1) * KB, lc + (t + 1) * KB, rc + (t + 2) * KB, std::false_type{});
        }
        par ^= 1;  // t = nk - 2: R-lo two K tiles ahead is the next tile's first
        ktile(smem + par * 4 * G256_HALF_BYTES, rc + (long)(nk - 1) * KB, lc + (long)(nk - 1) * KB, rn, std::false_type{});
        par ^= 1;  // t = nk - 1: the whole look-ahead belongs to the next tile
        ktile(smem + par * 4 * G256_HALF_BYTES, rn, lnx, rn + KB, std::false_type{});
    };

    // ---- epilogue of the finished tile at (r0, l0); staging = ring slots 1..3 of the last K tile's group ----
    auto epilogue = [&]() {
        // every per-lane quantity of the epilogue is re-derived from this opaque copy of the lane id: otherwise hipcc hoists
        // them (store addresses, staging offsets: ~40 VGPRs) out of the tile loop and keeps them live across the K loop,
        // which already runs at the 256-register limit of a 512-thread workgroup
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        char* stg_base = smem + (((slot0 >> 2) + nk - 1) & 1) * 4 * G256_HALF_BYTES + G256_HALF_BYTES + wave * 6144;
        const int gq = lane >> 4;
        if constexpr (MREG) {
            // lane group gq owns rows r0 + wr*64 + 16 gq .. +15 of feature l0 + wl*128 + 16 b + (lane & 15): 32 contiguous
            // bytes of the transposed output per fragment, stored straight from the accumulator layout
            const long m0 = r0 + wr * 64 + gq * 16;
            [[maybe_unused]] float2 rs[16];  // LN: (rstd, -mu * rstd) of the lane group's 16 rows (the same for its 16 lanes: broadcast reads)
            if constexpr (LN) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 t = *(const float4*)(cbuf + 1024 + (gq * 16 + 2 * i) * 8);
                    rs[2 * i] = float2{t.x, t.y}; rs[2 * i + 1] = float2{t.z, t.w};
                }
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int nl = b * 16 + (lane & 15);
                const float bias = *(const float*)(cbuf + nl * 4);
                float v[16];
                if constexpr (LN) {
                    const float cs = *(const float*)(cbuf + 512 + nl * 4);
#pragma unroll
                    for (int f = 0; f < 4; ++f)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[f * 4 + e] = fmaf(acc[f][b][e], rs[f * 4 + e].x, fmaf(rs[f * 4 + e].y, cs, bias));
                } else {
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[f * 4 + e] = acc[f][b][e] + bias;
                }
                epi.template store16<FULL>(m0, l0 + wl * 128 + nl, v);
            }
        } else {
        typename Epi::template Ctx<16> ctx;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float4 t = *(const float4*)(cbuf + (gq * 16 + 4 * f) * 4);
            ctx.bias[f * 4 + 0] = t.x; ctx.bias[f * 4 + 1] = t.y; ctx.bias[f * 4 + 2] = t.z; ctx.bias[f * 4 + 3] = t.w;
            if constexpr (epi_has_preload<Epi>::value || HL || LN) {  // LayerScale gamma, or (LN) the column sums behind b'
                const float4 u = *(const float4*)(cbuf + (LN ? 256 : 1024) + (gq * 16 + 4 * f) * 4);
                ctx.gamma[f * 4 + 0] = u.x; ctx.gamma[f * 4 + 1] = u.y; ctx.gamma[f * 4 + 2] = u.z; ctx.gamma[f * 4 + 3] = u.w;
            }
        }
        // LN: the row constants of this lane's row in accumulator block `blk` (rows 16 blk + (lane & 15) of the wave's 128)
        auto row_consts = [&](int blk) {
            if constexpr (LN) return *(const float2*)(cbuf + 1024 + (16 * blk + (lane & 15)) * 8);
            else return float2{0.f, 0.f};
        };
        if constexpr (epi_has_preload<Epi>::value) {
            // fp32 residual update x += gamma * (acc + bias): 16 rows x 64 columns of the wave's tile at a time are transposed
            // through LDS so one instruction covers 4 rows x 256 contiguous bytes; the x values of the NEXT 32 rows are
            // requested before the current ones are consumed (two register sets) -- the round trip to HBM, not the
            // arithmetic, is what this epilogue waits for.  Addresses are a wave-uniform base (SGPRs) + 8 loop-invariant
            // 32-bit lane offsets: per-access 64-bit address pairs would not fit beside 128 accumulators + 64 x registers.
            constexpr int PITCH = 68;
            float* stg = (float*)stg_base;
            const int mrow = lane >> 4, ncol = (lane & 15) * 4;
            const uint32_t ldx = (uint32_t)epi.ldx;
            const long mw = l0 + wl * 128, nw = r0 + wr * 64;      // first row / column of this wave's 128 x 64 part
            float* xw = epi.x + mw * epi.ldx + nw;                 // wave-uniform
            const long mleft = epi.m_valid - mw;                   // rows / columns of the part inside the problem
            const bool nok = FULL || nw + ncol < epi.n_valid;
            uint32_t loff[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) loff[i] = ((uint32_t)(mrow + 4 * i) * ldx + (uint32_t)ncol) * 4u;  // BYTES (saddr + voffset form)
            // units of 16 rows (one accumulator column block b): the x values of unit u + 2 are requested before unit u is consumed
            float4 xv[3][4];
            auto request = [&](float4 (&dst)[4], int u) {
                const char* xu = (const char*)(xw + (long)(16 * u) * ldx);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    dst[i] = (Epi::ACCUM && (FULL || (16 * u + 4 * i + mrow < mleft && nok))) ? CVX_RESID_LOAD((const float*)(xu + loff[i])) : float4{0.f, 0.f, 0.f, 0.f};
            };
            request(xv[0], 0);
            request(xv[1], 1);
            // gamma * (acc + bias) in place, so the 32 bias / gamma registers are dead before the third x set is requested
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[f][b][e] = ctx.gamma[f * 4 + e] * (acc[f][b][e] + ctx.bias[f * 4 + e]);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u + 2 < 8) request(xv[(u + 2) % 3], u + 2);
                __builtin_amdgcn_sched_barrier(0);  // (hipcc otherwise hoists all the requests to the top and spills)
                char* xu = (char*)(xw + (long)(16 * u) * ldx);
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const f32x4 a = acc[f][u];
                    *(float4*)(stg + (lane & 15) * PITCH + 16 * gq + 4 * f) = float4{a[0], a[1], a[2], a[3]};
                }
                float4 d[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) d[i] = *(const float4*)(stg + (4 * i + mrow) * PITCH + ncol);
#ifdef CVX_LN_EMIT_PROTO
                uint2 pk[4];
#endif
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 xq4 = xv[u % 3][i];
                    if (FULL || (16 * u + 4 * i + mrow < mleft && nok)) {
                        float4 o;
                        o.x = xq4.x + d[i].x; o.y = xq4.y + d[i].y; o.z = xq4.z + d[i].z; o.w = xq4.w + d[i].w;
#ifdef CVX_LN_EMIT_PROTO
                        pk[i] = uint2{pack2bf(o.x, o.y), pack2bf(o.z, o.w)};
                        if constexpr (FULL) {   // row partial sums of these 64 columns: in-lane, then across the 16 lanes that share the row (DPP)
                            float rs = (o.x + o.y) + (o.z + o.w), rq = (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
                            rs += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rs), 0xB1, 0xF, 0xF, true));
                            rq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rq), 0xB1, 0xF, 0xF, true));
                            rs += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rs), 0x4E, 0xF, 0xF, true));
                            rq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rq), 0x4E, 0xF, 0xF, true));
                            rs += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rs), 0x141, 0xF, 0xF, true));
                            rq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rq), 0x141, 0xF, 0xF, true));
                            rs += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rs), 0x140, 0xF, 0xF, true));
                            rq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rq), 0x140, 0xF, 0xF, true));
                            // (bias / gamma left the constants buffer at the top of the epilogue: it stages the 128 rows' statistics)
                            if ((lane & 15) == 0) *(float2*)(cbuf + (16 * u + 4 * i + mrow) * 8) = float2{rs, rq};
                        }
#endif
#if CVX_RESID_STORE_NT
                        gst16_saddr_nt(xu, loff[i], __builtin_bit_cast(u32x4, o));
#else
                        gst16_saddr(xu, loff[i], __builtin_bit_cast(u32x4, o));
#endif
                    }
                }
#ifdef CVX_LN_EMIT_PROTO
                if (FULL && epi.emit_xb) {
                    // lane pairs (even, odd column lane) exchange halves so each lane stores 8 consecutive columns of ONE row
                    const bool odd = lane & 1;
                    const long g_emit_ldb = epi.emit_ldb;
                    const uintptr_t pb = (uintptr_t)(epi.emit_xb + (mw + 16 * u) * g_emit_ldb + nw);  // (wave-uniform: into SGPRs)
                    char* bu = (char*)(((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pb >> 32)) << 32) |
                                       (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pb));
#pragma unroll
                    for (int p2 = 0; p2 < 2; ++p2) {
                        const uint2 keep = odd ? pk[2 * p2 + 1] : pk[2 * p2];
                        const uint2 send = odd ? pk[2 * p2] : pk[2 * p2 + 1];
                        const uint32_t r0 = __builtin_amdgcn_mov_dpp(send.x, 0xB1, 0xF, 0xF, true);
                        const uint32_t r1 = __builtin_amdgcn_mov_dpp(send.y, 0xB1, 0xF, 0xF, true);
                        const u32x4 v = odd ? u32x4{r0, r1, keep.x, keep.y} : u32x4{keep.x, keep.y, r0, r1};
                        const uint32_t bo = ((uint32_t)(4 * (2 * p2 + (odd ? 1 : 0)) + mrow) * (uint32_t)g_emit_ldb + (uint32_t)(8 * ((lane & 15) >> 1))) * 2u;
                        gst16_saddr(bu, bo, v);
                    }
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef CVX_LN_EMIT_PROTO
            if (FULL && epi.emit_part) {
                // P[slot = 4 * (N tile) + wr][row][2]: the wave's 128 rows x (sum, sum of squares) are 1 KiB = one store
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const f32x4 st4 = *(const f32x4*)(cbuf + lane * 16);
                const long slot = (r0 >> 8) * 4 + wr;
                const uintptr_t pp = (uintptr_t)(epi.emit_part + (slot * epi.emit_rows + mw) * 2);
                char* pbase = (char*)(((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pp >> 32)) << 32) |
                                      (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pp));
                gst16_saddr(pbase, (uint32_t)lane * 16u, __builtin_bit_cast(u32x4, st4));
            }
#endif
        } else if constexpr (HL) {
            // bf16 hi / lo residual stream: x = hi + lo (exact in fp32), x += gamma * (acc + bias), hi' = bf16(x), lo' = bf16(x - hi').
            // Units of 16 rows x 64 columns are transposed through LDS as in the fp32 form; afterwards a lane owns 8 consecutive
            // columns of rows r8 and r8 + 8 of the unit, so each of the unit's four loads and four stores covers 8 rows x 128
            // contiguous bytes of one array.  The row sums of the new x (64 columns: 8 lanes, three DPP steps) are staged in the
            // constants buffer (bias / gamma left it at the top of the epilogue) and leave as ONE 1-KiB store per wave:
            // part[slot = column / 64][row][2] -- what cvx_rowstat_finalize turns into the next GEMM's (rstd, -mu * rstd).
            constexpr int PITCH = 68;
            float* stg = (float*)stg_base;
            const int r8 = lane >> 3, c8 = (lane & 7) * 8;
            const uint32_t ldx = (uint32_t)epi.ldx;
            const long mw = l0 + wl * 128, nw = r0 + wr * 64;      // first row / column of this wave's 128 x 64 part
            const char* hw = (const char*)(epi.xh + mw * epi.ldx + nw);   // wave-uniform
            const char* lw = (const char*)(epi.xl + mw * epi.ldx + nw);
            const long mleft = epi.m_valid - mw;                   // rows of the part inside the problem
            // columns: n_valid is a multiple of 64, so a wave's 64 columns are inside the problem or outside it as a whole (wave-uniform)
            const bool nok = FULL || nw + 64 <= epi.n_valid;
            uint32_t loff[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) loff[j] = ((uint32_t)(r8 + 8 * j) * ldx + (uint32_t)c8) * 2u;  // BYTES
            u32x4 xv[3][4];  // [set][hi row A, lo row A, hi row B, lo row B]; unit u + 2 is requested before unit u is consumed
            auto request = [&](u32x4 (&dst)[4], int u) {
                const long ub = (long)(16 * u) * ldx * 2;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool ok = FULL || (nok && 16 * u + 8 * j + r8 < mleft);
                    dst[2 * j] = ok ? __builtin_nontemporal_load((const u32x4*)(hw + ub + loff[j])) : u32x4{0u, 0u, 0u, 0u};
                    dst[2 * j + 1] = ok ? __builtin_nontemporal_load((const u32x4*)(lw + ub + loff[j])) : u32x4{0u, 0u, 0u, 0u};
                }
            };
            request(xv[0], 0);
            request(xv[1], 1);
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[f][b][e] = ctx.gamma[f * 4 + e] * (acc[f][b][e] + ctx.bias[f * 4 + e]);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u + 2 < 8) request(xv[(u + 2) % 3], u + 2);
                __builtin_amdgcn_sched_barrier(0);  // (hipcc otherwise hoists all the requests to the top and spills)
                const long ub = (long)(16 * u) * ldx * 2;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const f32x4 a = acc[f][u];
                    *(float4*)(stg + (lane & 15) * PITCH + 16 * gq + 4 * f) = float4{a[0], a[1], a[2], a[3]};
                }
                float4 d[2][2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h) d[j][h] = *(const float4*)(stg + (r8 + 8 * j) * PITCH + c8 + 4 * h);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const u32x4 vh = xv[u % 3][2 * j], vl = xv[u % 3][2 * j + 1];
                    const float dd[8] = {d[j][0].x, d[j][0].y, d[j][0].z, d[j][0].w, d[j][1].x, d[j][1].y, d[j][1].z, d[j][1].w};
                    u32x4 nh, nl;
                    float rs = 0.f, rq = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float x0 = (bflo(vh[k]) + bflo(vl[k])) + dd[2 * k];
                        const float x1 = (bfhi(vh[k]) + bfhi(vl[k])) + dd[2 * k + 1];
                        nh[k] = pack2bf(x0, x1);
                        nl[k] = pack2bf(x0 - bflo(nh[k]), x1 - bfhi(nh[k]));
                        rs += x0 + x1;
                        rq = fmaf(x0, x0, fmaf(x1, x1, rq));
                    }
                    // across the 8 lanes that share the row: xor 1, xor 2 (quad permutes), then the other quad of the half row
                    rs += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rs), 0xB1, 0xF, 0xF, true));
                    rq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rq), 0xB1, 0xF, 0xF, true));
                    rs += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rs), 0x4E, 0xF, 0xF, true));
                    rq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rq), 0x4E, 0xF, 0xF, true));
                    rs += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rs), 0x141, 0xF, 0xF, true));
                    rq += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rq), 0x141, 0xF, 0xF, true));
                    if ((lane & 7) == 0) *(float2*)(cbuf + (16 * u + 8 * j + r8) * 8) = float2{rs, rq};
                    if (FULL || (nok && 16 * u + 8 * j + r8 < mleft)) {
                        gst16_saddr((void*)(hw + ub), loff[j], nh);        // the next GEMM reads hi: plain policy
                        gst16_saddr_nt((void*)(lw + ub), loff[j], nl);     // lo is read once, a whole layer later: streaming
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (FULL || nok) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const f32x4 st4 = *(const f32x4*)(cbuf + lane * 16);  // rows 2 lane, 2 lane + 1 of the wave's 128
                const uintptr_t pp = (uintptr_t)(epi.part + (((r0 >> 6) + wr) * epi.part_rows + mw) * 2);  // (wave-uniform: into SGPRs)
                char* pbase = (char*)(((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pp >> 32)) << 32) |
                                      (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pp));
                gst16_saddr(pbase, (uint32_t)lane * 16u, __builtin_bit_cast(u32x4, st4));  // (rows past m_valid: allocated, never read)
            }
        } else {
            constexpr int O16 = Epi::OUT16, ROWB = 4 * O16 * 2, PITCHB = ROWB + 16, LPR = ROWB / 16, RPI = 64 / LPR;
            static_assert(32 * PITCHB <= 6144, "staging buffer of a wave");
            char* stg = stg_base;
            const int srow = lane / LPR, spiece = lane % LPR;
            const uint32_t ldc = (uint32_t)epi.ldc;
            const long mw = l0 + wl * 128, ow = (r0 + wr * 64) >> Epi::OUT_SHIFT;   // first row / output column of this wave's part
            uint16_t* outw = epi.out + mw * epi.ldc + ow;                           // wave-uniform
            const long mleft = epi.m_valid - mw;
            const bool ook = FULL || ow + spiece * 8 < (epi.n_valid >> Epi::OUT_SHIFT);
            uint32_t loff[32 / RPI];
#pragma unroll
            for (int i = 0; i < 32 / RPI; ++i) loff[i] = ((uint32_t)(RPI * i + srow) * ldc + (uint32_t)(spiece * 8)) * 2u;  // BYTES
            if constexpr (O16 == 8 && DIRECT8) {
                // 8 packed outputs = 16 B per lane: in accumulator layout a wave instruction already writes 16 rows x 64
                // contiguous bytes (the four lane groups of a row sit side by side) -- the same shape the staged path
                // produces, without the LDS round trip
                const uint32_t doff = ((uint32_t)(lane & 15) * ldc + (uint32_t)(gq * 8)) * 2u;
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    float v[16];
#pragma unroll
                    for (int f = 0; f < 4; ++f)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[f * 4 + e] = acc[f][b][e];
                    uint32_t w[O16 / 2];
                    epi.produce(ctx, v, w, row_consts(b));
                    char* ob = (char*)(outw + (long)(16 * b) * ldc);
                    if (FULL || (16 * b + (lane & 15) < mleft && ow + gq * 8 < (epi.n_valid >> Epi::OUT_SHIFT)))
                        gst16_saddr(ob, doff, u32x4{w[0], w[1], w[2], w[3]});
                }
            } else
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                char* oq = (char*)(outw + (long)(32 * q) * ldc);
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    float v[16];
#pragma unroll
                    for (int f = 0; f < 4; ++f)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[f * 4 + e] = acc[f][2 * q + bb][e];
                    uint32_t w[O16 / 2];
                    epi.produce(ctx, v, w, row_consts(2 * q + bb));
                    char* dst = stg + (16 * bb + (lane & 15)) * PITCHB + gq * (O16 * 2);
                    if constexpr (O16 == 16) {
                        *(uint4*)dst = uint4{w[0], w[1], w[2], w[3]};
                        *(uint4*)(dst + 16) = uint4{w[4], w[5], w[6], w[7]};
                    } else {
                        *(uint4*)dst = uint4{w[0], w[1], w[2], w[3]};
                    }
                }
                // (all the staged rows are read BEFORE the first store: the asm stores are ordering points for hipcc, and read /
                //  wait / store one row at a time exposed the LDS latency 16 times per tile)
                uint4 d[32 / RPI];
#pragma unroll
                for (int i = 0; i < 32 / RPI; ++i) d[i] = *(const uint4*)(stg + (RPI * i + srow) * PITCHB + spiece * 16);
#pragma unroll
                for (int i = 0; i < 32 / RPI; ++i) {
                    const int row = RPI * i + srow;
                    if (FULL || (32 * q + row < mleft && ook)) {
                        if constexpr (O16 == 16 && CVX_BF16_STORE_NT) gst16_saddr_nt(oq, loff[i], u32x4{d[i].x, d[i].y, d[i].z, d[i].w});
                        else gst16_saddr(oq, loff[i], u32x4{d[i].x, d[i].y, d[i].z, d[i].w});  // 64-B segments: left to the L2's write combining
                    }
                }
            }
        }
        }  // !MREG
    };

    // ---- pipeline fill (first tile only): half-tiles 0..4 = K tile 0 and R-lo of K tile 1 ----
    issue_consts(r0, l0);
    dma((const char*)Rc, offR[0], smem, 0);
    dma((const char*)Lc, offL[0], smem, 1);
    dma((const char*)Rc, offR[1], smem, 2);
    dma((const char*)Lc, offL[1], smem, 3);
    dma((const char*)Rc + KB, offR[0], smem, 4);
    wait_vmcnt<6>();
    __builtin_amdgcn_s_barrier();
    if (wl == 1) __builtin_amdgcn_s_barrier();  // stagger: waves 4-7 run one barrier behind
    [[maybe_unused]] int dbg_tile = 0;
    const bool dbg_on = DBG && blockIdx.x == 8 * 10 && (wave & 3) == 0 && (tid & 63) == 0;
    auto dbg = [&](int k) {
        if constexpr (DBG) {
            const unsigned long long t = stamp();
            if (dbg_on && dbg_tile < 8) g_gemm256p_dbg[((wave >> 2) * 8 + dbg_tile) * 6 + k] = t;
        }
    };
    dbg(0);
    ktile(smem, (const char*)Rc + KB, (const char*)Lc + KB, (const char*)Rc + 2 * KB, std::false_type{});
    dbg(5);
    for (;;) {
        ktiles_1_to_end();
        dbg(1);
        if (wl == 0) __builtin_amdgcn_s_barrier();  // pairs with the stagger barrier: both groups are in step again
        dbg(2);
        epilogue();
        dbg(3);
        id += per;
        if (id >= cnt) break;
        slot0 = (slot0 + total) & 7;
        Rc = Rn; Lc = Ln; r0 = rn; l0 = ln;
        if (id + per < cnt) panels(id + per, Rn, Ln, rn, ln);  // (else: the last tile prefetches itself)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // every wave is done with its staging slots
        if (wl == 1) __builtin_amdgcn_s_barrier();  // stagger again
        dbg(4);
        if constexpr (DBG) ++dbg_tile;
        dbg(0);
        ktile(smem + (slot0 >> 2) * 4 * G256_HALF_BYTES, (const char*)Rc + KB, (const char*)Lc + KB, (const char*)Rc + 2 * KB, std::true_type{});
        dbg(5);
    }
    wait_vmcnt<0>();  // the last tile's look-ahead DMA must have landed before the workgroup releases its LDS
}

}  // namespace cvx
