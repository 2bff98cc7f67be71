// Flash-style multi-head attention for the ViT blocks, head_dim 64, gfx950.
//
// One workgroup = 4 waves = 128 query rows of one (slice, head); each wave owns 32 query rows.
// Per 64-key tile:   S^T = K Q^T  (v_mfma_f32_32x32x16_bf16, "swapped" so a query row sits on ONE lane and
// the softmax row reductions are lane-local + one cross-half exchange), online softmax in fp32, then
// O^T += V^T P^T where the S^T accumulator registers are converted in place into the B operand (no LDS
// round trip for P).  K rows are read through the permutation pi (swap bits 2,3 of the row index) so
// that the k order the accumulator-as-operand trick imposes becomes 8 CONTIGUOUS keys of V^T -- which the
// QKV GEMM epilogue writes transposed (CVX_EPI_VT), so V^T fragments are plain ds_read_b128.
// K and V^T tiles are double-buffered in LDS by LDS-DMA (global_load_lds_dwordx4), one barrier per tile,
// XOR-swizzled on the source address (conflict-free ds_read_b128).
#include "common.h"
#include "../../include/cryovit_hip.h"
#include "host_util.h"
#include <atomic>

namespace cvx {

constexpr int ATT_THREADS = 256;
constexpr int KV_TILE = 64;
constexpr int ATT_TILE_BYTES = 64 * 128;  // 64 rows x 64 bf16

__device__ __forceinline__ int pi_row(int r) { return (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1); }

// VARIANT: 0 = default (the O rescale is skipped when no row maximum of the wave moved), 1 = always rescale,
//          10-13 = timing-only ablations (10: exp2 replaced by a multiply, 11: K/V tile 0 reused, no DMA / wait / barrier,
//          12: DMA issued but no wait / barrier, 13: wait + barrier but no DMA)
//          3 = THREE K/V^T buffers: tile j+2 is issued in iteration j and waited for with a counted vmcnt(4)
//          8 / 9 = variant 7 with EIGHT waves per workgroup (256 query rows share every K / V^T tile: half the LDS-DMA instructions
//          and half the L2 -> LDS bytes per query row).  Two such workgroups per CU keep four waves per SIMD, and their 2 x 48 KB of
//          LDS have room for a THIRD tile buffer (8: two tiles of look-ahead behind a counted wait; 9: two buffers, for A/B runs) --
//          the three-buffer form of the 4-wave kernel (variant 3) paid for its buffer with the fourth wave per SIMD.
// VROW: V is NOT pre-transposed -- it sits row-major beside Q and K in one [rows][3C] buffer (columns 2C ..), written by ONE qkv GEMM
//       with the plain row-major epilogue (no separate V^T GEMM launch, no scattered 16-B stores of a transposed epilogue, no vt buffer).
//       The V tile is staged exactly like the K tile ([key][64 dims], same swizzle); the A operand of O^T += V^T P^T (32 dims x 16 keys,
//       lane (dim r, h): keys 8h .. 8h+7) is produced by two ds_read_b64_tr_b16 per MFMA: a 16-lane group reads a 4-key x 16-dim block
//       (lane li: key li / 4, dims 4 (li % 4) ..) and receives it transposed (lane li: dim li, the 4 keys).
typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef short v8s_t __attribute__((ext_vector_type(8)));
template <int VARIANT, bool VROW = false>
__global__ __launch_bounds__((VARIANT == 8 || VARIANT == 9) ? 512 : ATT_THREADS) void k_attention(const uint16_t* __restrict__ qk, long ldqk,
                                                           const uint16_t* __restrict__ vt, uint16_t* __restrict__ out,
                                                           long ldo, int heads, int ntok, int ntp, int kp, int C, int nqb,
                                                           int xcd_remap) {
    constexpr int NW = (VARIANT == 8 || VARIANT == 9) ? 8 : 4;       // waves per workgroup, 32 query rows each
    constexpr int NT = NW * 64, QB = NW * 32;                        // threads, query rows per workgroup
    constexpr int PPT = 512 / NT;                                    // 16-B pieces per thread and tile (K and V^T alike)
    constexpr int NBUF = (VARIANT == 3 || VARIANT == 8) ? 3 : 2;
    __shared__ __attribute__((aligned(16))) char smem[NBUF * 2 * ATT_TILE_BYTES];  // [buf][K | V^T]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware block order: blocks b and b+8 share an XCD (private L2).  All nqb query blocks of one (slice, head) --
    // which stream the SAME K / V^T -- are placed on one XCD, back to back in its dispatch sequence, so a K/V tile is an
    // L2 miss once and an L2 hit for the other query blocks.  (Placement only changes speed, never results.)
    int qb, pair;
    const bool prio_qk = (xcd_remap & 2) != 0, prio_pv = (xcd_remap & 4) != 0;  // s_setprio 1 around the wave's MFMA blocks (cvx_set_option "attn_mfma_prio": bit 0 S^T, bit 1 O^T)
    if (xcd_remap & 1) {
        const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
        pair = (w / nqb) * 8 + xcd;
        qb = w % nqb;
    } else {
        pair = blockIdx.x / nqb;
        qb = blockIdx.x % nqb;
    }
    const int slice = pair / heads, head = pair - slice * heads;
    const long row0 = (long)slice * ntp;
    const uint16_t* Qp = qk + row0 * ldqk + head * 64;
    const uint16_t* Kp = Qp + C;
    const uint16_t* Vp = VROW ? Kp + C : vt + ((long)(slice * heads + head) * 64) * kp;
    const long vpitch = VROW ? ldqk : (long)kp;  // elements between consecutive rows of the V source (VROW: token rows, else V^T dim rows)

    const int r = lane & 31, h = lane >> 5;
    const int q0 = qb * QB + wave * 32;
    const int qrow = min(q0 + r, ntp - 1);  // rows past the slice are clamped for loads, never stored
    const bool wave_idle = VARIANT != 1 && q0 >= ntok;  // (variant 1 keeps the old behaviour for A/B runs)

    // Q fragments: B operand of S^T = K Q^T.  lane (r,h) holds Q[q0+r][16*ks + 8*h + j]
    // The loads are inline asm with their own wait (below, before the key loop): left to hipcc, the waits for these four
    // loads land at their first use INSIDE the loop as vmcnt(3)..vmcnt(0) -- and since the LDS-DMA of the loop is issued
    // from asm (invisible to hipcc's counters), that vmcnt(0) drained the tile prefetch it had just issued, every tile.
    bf16x8 qf[4];
    {
        const uint16_t* qsrc = Qp + (long)qrow * ldqk + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qf[ks]) : "v"(qsrc + 16 * ks) : "memory");
    }
    // Scores arrive in LOG2 units: the caller folds head_dim^-0.5 * log2(e) into the Q projection (one rounding, at weight
    // packing), so p = exp2(S - m) with no multiply.
    // VARIANT 6: the running maximum is subtracted INSIDE the matrix product: a fifth
    // k-step whose K operand is the constant column e_0 and whose Q operand carries -m of the lane's query row, so
    //   S' = K Q^T - m   and   p = exp2(S')   with no per-element VALU work at all
    // as long as the row maxima of the tile stay within DEFER (log2 units) of the value subtracted.  The VALU, not the
    // matrix pipe, bounds this kernel at head_dim 64 (PMC: VALU active 73 % of SIMD cycles, MFMA 39 %); the 32 fused
    // multiply-adds per tile this removes were 15 % of its vector instructions, for 2 more MFMAs on the idle pipe.
    constexpr bool AUG = VARIANT == 6 || VARIANT == 7 || VARIANT == 8 || VARIANT == 9;
    constexpr bool PSUM_TRIGGER = VARIANT == 7 || VARIANT == 8 || VARIANT == 9;  // the maximum is only looked at in tile 0; later tiles watch their probability sums
    constexpr float DEFER = 3.0f;  // p <= 2^3 before a row's maximum is raised (bf16 P is floating point: same relative precision)
    bf16x8 kaug, qaug;
    [[maybe_unused]] float m_used = 0.f;  // what is currently subtracted (exactly representable in bf16)
    if constexpr (AUG) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { kaug[e] = (__bf16)0.f; qaug[e] = (__bf16)0.f; }
        if (h == 0) kaug[0] = (__bf16)1.0f;
    }

    // LDS-DMA source offsets for this thread's two 16-B pieces of each tile
    // (saddr-form DMA, common.h: wave-uniform tile base + tile-invariant 32-bit lane byte offsets -- no address VALU per piece)
    uint32_t koffs[2], voffs[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int c = (jj < PPT ? jj : 0) * NT + tid;  // (PPT = 1: the second entry is unused)
        const int srow = c >> 3, schunk = ((c & 7) ^ ((srow >> 1) & 7)) << 3;
        koffs[jj] = (uint32_t)(srow * ldqk + schunk) * 2u;
        voffs[jj] = (uint32_t)(srow * vpitch + schunk) * 2u;
    }
    // A query block whose rows all belong to wave 0 (1029 tokens = 8 x 128 + 5: every ninth block) runs as ONE wave: waves 1-3
    // leave at once (a hardware barrier only counts the waves still alive) and wave 0 issues their DMA pieces too -- their wave
    // slots go back to the CU instead of idling through 17 tiles of barriers.
    const bool lone = AUG && qb * QB + 32 >= ntok;  // (uniform per workgroup)
    if (lone && wave != 0) return;
    auto issue = [&](int j, int buf) {
        const uint32_t kt = lds_addr(smem) + buf * 2 * ATT_TILE_BYTES + wave * 1024;
        const long kv0 = (long)j * KV_TILE;
        if constexpr (PPT == 2) {
            glds16_saddr2<NT * 16>(Kp + kv0 * ldqk, koffs[0], koffs[1], kt);
            glds16_saddr2<NT * 16>(Vp + (VROW ? kv0 * ldqk : kv0), voffs[0], voffs[1], kt + ATT_TILE_BYTES);
        } else {
            glds16_saddr(Kp + kv0 * ldqk, koffs[0], kt);
            glds16_saddr(Vp + (VROW ? kv0 * ldqk : kv0), voffs[0], kt + ATT_TILE_BYTES);
        }
        if (lone) {
            // the pieces of the departed waves: wave w2's thread sits 8 w2 tile rows below this one, same chunk, and its swizzle
            // term (row >> 1) & 7 differs by 4 w2 & 7, i.e. the 16-B chunk index flips bit 2 for odd w2 (ldqk and kp are multiples
            // of 64 elements, so that bit is bit 6 of the byte offset).  Derived from opaque copies: hoisted out of the key loop, 12
            // more offsets cost the common path a wave per SIMD.
            uint32_t k0 = koffs[0], k1 = koffs[1], v0 = voffs[0], v1 = voffs[1];
            asm volatile("" : "+v"(k0), "+v"(k1), "+v"(v0), "+v"(v1));
#pragma unroll 1
            for (int w2 = 1; w2 < NW; ++w2) {
                const uint32_t fl = (w2 & 1) ? 64u : 0u;
                const uint32_t dk = (uint32_t)(w2 * 16) * (uint32_t)ldqk, dv = (uint32_t)(w2 * 16) * (uint32_t)vpitch;
                if constexpr (PPT == 2) {
                    glds16_saddr2<NT * 16>(Kp + kv0 * ldqk, (k0 + dk) ^ fl, (k1 + dk) ^ fl, kt + w2 * 1024);
                    glds16_saddr2<NT * 16>(Vp + (VROW ? kv0 * ldqk : kv0), (v0 + dv) ^ fl, (v1 + dv) ^ fl, kt + ATT_TILE_BYTES + w2 * 1024);
                } else {
                    glds16_saddr(Kp + kv0 * ldqk, (k0 + dk) ^ fl, kt + w2 * 1024);
                    glds16_saddr(Vp + (VROW ? kv0 * ldqk : kv0), (v0 + dv) ^ fl, kt + ATT_TILE_BYTES + w2 * 1024);
                }
            }
        }
    };

    // fragment read offsets
    const int krow = pi_row(r);
    const int koff = krow * 128, ksw = (krow >> 1) & 7;  // K tile: row 32t + pi(r), 16-B chunk 2*ks + h
    const int voff = r * 128, vsw = (r >> 1) & 7;        // V^T tile: row 32dt + r, chunk 4t + 2s + h
    // VROW: byte offset inside a V tile of this lane's transposed read for dim block dt, key quad j (keys 8h + 4j .. +3 of a 16-key
    // step; the step itself adds (32 t + 16 sk) rows = a compile-time offset): row 8h + 4j + li / 4, dims 32 dt + 16 dgrp + 4 (li % 4)
    [[maybe_unused]] uint32_t vtr[2][2];
    if constexpr (VROW) {
        const int li = lane & 15, dgrp = (lane >> 4) & 1;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 8 * h + 4 * j + (li >> 2);
                const int chunk = (4 * dt + 2 * dgrp + ((li & 3) >> 1)) ^ ((row >> 1) & 7);
                vtr[dt][j] = (uint32_t)(row * 128 + (chunk << 4) + (li & 1) * 8);
            }
    }

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const float LOG2E = 1.0f;  // (scores are already in log2 units: see the note at the Q fragments)

    const int nkv = (ntok + KV_TILE - 1) / KV_TILE;
    constexpr bool SKIP_RESCALE = VARIANT != 1, ABL_NOEXP = VARIANT == 10;
    constexpr bool ABL_NOSYNC = VARIANT == 11 || VARIANT == 12, ABL_NODMA = VARIANT == 11 || VARIANT == 13;
    issue(0, 0);
    if (NBUF == 3 && nkv > 1) issue(1, 1);
    // Q has landed (and tile 0 / 1 with it: the loop's own first wait is then a no-op); "+v" pins every use of qf below this
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3])::"memory");
    int buf = 0;  // buffer of tile j
    const char* kt = smem;
    [[maybe_unused]] bool redo = false;  // VARIANT 7: tile j is computed again, with the row maxima (no new wait / barrier / DMA)
    for (int j = 0; j < nkv; ++j) {
        if (!PSUM_TRIGGER || !redo) {
        if (!ABL_NOSYNC || j == 0) {
            // tile j landed (with three buffers tile j+1 may stay in flight); every wave is done with tile j-1
            // (a wave's own pieces of tile j+1 are its 2 PPT youngest entries -- 2 PPT NW when it issues for the whole workgroup)
            if (NBUF == 3 && j + 1 < nkv) {
                if (lone) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPT * NW) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPT) : "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (NBUF == 3) {
            if (j + 2 < nkv) issue(j + 2, buf == 0 ? 2 : buf - 1);  // (j+2) % 3 == (buf + 2) % 3
        } else if (!ABL_NODMA && j + 1 < nkv) {
            issue(ABL_NOSYNC ? 0 : j + 1, ABL_NOSYNC ? 1 : (buf ^ 1));
        }
        kt = smem + (ABL_NOSYNC || ABL_NODMA ? 0 : buf) * 2 * ATT_TILE_BYTES;
        buf = NBUF == 3 ? (buf == 2 ? 0 : buf + 1) : (buf ^ 1);
        }
        [[maybe_unused]] const bool anchor = j == 0 || redo;
        redo = false;
        const char* vtile = kt + ATT_TILE_BYTES;
        // 1029 tokens = 8 x 128 + 5: in the last query block only wave 0 owns real rows.  The other three keep feeding the
        // LDS-DMA and the barriers (the tile is a workgroup effort) but skip the products and the softmax -- their issue
        // slots go to the other workgroups on the SIMD (wave-uniform branch: EXEC stays full).
        if (wave_idle) continue;

        // ---- S^T[t] = K_t Q^T : rows = keys (registers), col = query (lane) ----
        f32x16 s[2];
        const int kv0 = j * KV_TILE;
        // 1029 keys = 16 x 64 + 5: the upper 32 keys of the last tile are all padding.  Their products, S^T and O^T alike, are
        // skipped (wave-uniform): half a tile of MFMAs in 17, which at the board's power limit is time
        const bool half_tile = !(xcd_remap & 8) && kv0 + 32 >= ntok;  // (bit 3: A/B switch, cvx_set_option "attn_half_tile" 0)
        auto compute_s = [&]() {
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[0][i] = 0.f; s[1][i] = 0.f; }
            if (prio_qk) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t == 1 && half_tile) break;  // (the mask below sets all of S^T[1] to -inf)
                if constexpr (AUG) s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kaug, qaug, s[t], 0, 0, 0);  // -m_used for every key
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 kf = *(const bf16x8*)(kt + t * 4096 + koff + (((2 * ks + h) ^ ksw) << 4));
                    s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[t], 0, 0, 0);
                }
            }
            if (prio_qk) __builtin_amdgcn_s_setprio(0);
            // ---- mask keys >= ntok (last tile only; wave-uniform test) ----
            if (kv0 + KV_TILE > ntok) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int rho = (i & 3) + 8 * (i >> 2) + 4 * h;
                        if (kv0 + 32 * t + pi_row(rho) >= ntok) s[t][i] = -INFINITY;
                    }
            }
        };
        if constexpr (!PSUM_TRIGGER) compute_s();

        if constexpr (PSUM_TRIGGER) {
            // ---- online softmax on S' = S - m_used (log2 units), VARIANT 7 ----
            // Tile 0 anchors every row at its own maximum (16 v_max3 + one cross-half exchange, once per query block).  After that the
            // row maximum is not computed on the common path: p = exp2(S') is floating point, so a row whose later scores exceed its
            // anchor simply produces p > 1 -- the same relative precision in bf16 P, in the fp32 sums and in O.  What has to be
            // prevented is overflow, and for that the probability SUM the tile computes anyway is enough: when a lane's sum passes
            // 2^24 the whole tile is computed AGAIN (same LDS tile: no wait, barrier or DMA) in the anchored form.  Per 64-key tile
            // this removes 16 v_max3, the exchange and the compare chain from a loop whose vector ISSUE slots, not its matrix pipe,
            // set the pace.
            float psum;
            compute_s();
            if (anchor) {
                // tile 0, or a tile whose probability sums ran away: the rows are (re-)anchored at their maximum exactly as variant 6
                // does in every tile.  Tile 0 takes the maximum whatever its sign (a row whose scores are all far below zero must not
                // underflow to l = 0); later the anchor only ever goes up.  The anchor is bf16-representable so that the products of
                // the next tiles subtract exactly what the rescale here assumes.
                float mloc = s[0][0];
#pragma unroll
                for (int i = 1; i < 16; ++i) mloc = fmaxf(mloc, s[0][i]);
#pragma unroll
                for (int i = 0; i < 16; ++i) mloc = fmaxf(mloc, s[1][i]);
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
                const float m_new = (float)(__bf16)(m_used + (j == 0 ? mloc : fmaxf(mloc, 0.f)));
                const float delta = m_new - m_used;
                m_used = m_new;
                if (h == 0) qaug[0] = (__bf16)(-m_new);
                const float alpha = j == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);  // (tile 0: O = l = 0, and -delta may be huge)
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; s[0][i] -= delta; s[1][i] -= delta; }
            }
            // (accumulating the sums two at a time with v_pk_add_f32 -- 16 instead of 32 adds -- measured 3 % SLOWER: 1.100 vs 1.064 ms)
            psum = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = __builtin_amdgcn_exp2f(s[t][i]);
                    s[t][i] = p;
                    psum += p;
                }
            // some row's probabilities outgrew its anchor by 2^19 or more (or overflowed): the tile is computed again, this time
            // with the maximum (never twice: the anchored pass is final -- variant 6's own bound, p <= 2^(half a bf16 spacing of m))
            if (!anchor && __builtin_amdgcn_ballot_w64(!(psum <= 16777216.0f)) != 0) {
                redo = true;
                --j;
                continue;
            }
            l_run += psum;
        } else if constexpr (AUG) {
            // ---- online softmax on S' = S - m_used (log2 units) ----
            float mloc = s[0][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mloc = fmaxf(mloc, s[0][i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) mloc = fmaxf(mloc, s[1][i]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            // tile 0 anchors every row at its own maximum, whatever its sign (m_used starts at 0: a row whose scores are all
            // far below zero must not underflow to l = 0); later tiles only ever raise it
            if (j == 0 || __builtin_amdgcn_ballot_w64(mloc > DEFER) != 0) {
                // the subtracted maximum becomes the row's new maximum rounded to bf16 (the value the next tiles' products
                // subtract must be the one used here); delta is exact in fp32
                const float m_new = (float)(__bf16)(m_used + (j == 0 ? mloc : fmaxf(mloc, 0.f)));
                const float delta = m_new - m_used;
                m_used = m_new;
                if (h == 0) qaug[0] = (__bf16)(-m_new);
                const float alpha = j == 0 ? 1.0f : __builtin_amdgcn_exp2f(-delta);  // (tile 0: O = l = 0, and -delta may be huge)
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; s[0][i] -= delta; s[1][i] -= delta; }
            }
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = __builtin_amdgcn_exp2f(s[t][i]);
                    s[t][i] = p;
                    psum += p;
                }
            l_run += psum;
        } else {
        // ---- online softmax (scores already carry head_dim^-0.5 through Q) ----
        float mloc = s[0][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mloc = fmaxf(mloc, s[0][i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) mloc = fmaxf(mloc, s[1][i]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
        const float mb = m_new * LOG2E;
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = ABL_NOEXP ? fmaf(s[t][i], LOG2E, -mb) * 0.001f : __builtin_amdgcn_exp2f(fmaf(s[t][i], LOG2E, -mb));
                s[t][i] = p;
                psum += p;
            }
        l_run = fmaf(l_run, alpha, psum);
        // alpha == 1 exactly when the row maximum did not move: skip the 32-register rescale unless some lane needs it
        if (!SKIP_RESCALE || __builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
        }
        }  // !AUG

        // ---- O^T[dt] += V^T[dt] P^T : accumulator registers 8s..8s+7 of S^T[t] are k-step s of the B operand ----
        if (prio_pv) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t == 1 && half_tile) break;  // P is exactly 0 there
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                bf16x8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (__bf16)s[t][8 * sk + e];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    bf16x8 vf;
                    if constexpr (VROW) {
                        typedef __attribute__((address_space(3))) v4s_t* lds_v4;
                        const char* vb = vtile + (32 * t + 16 * sk) * 128;
                        const v4s_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(vb + vtr[dt][0]));
                        const v4s_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(vb + vtr[dt][1]));
                        vf = __builtin_bit_cast(bf16x8, v8s_t{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
                    } else {
                        vf = *(const bf16x8*)(vtile + dt * 4096 + voff + (((4 * t + 2 * sk + h) ^ vsw) << 4));
                    }
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
                }
            }
        }
        if (prio_pv) __builtin_amdgcn_s_setprio(0);
    }

    // ---- normalise and store: lane (r,h) holds O[q0+r][32dt + 8g + 4h + (0..3)] in regs 4g..4g+3 ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q0 + r < ntok) {
        uint16_t* orow = out + (row0 + q0 + r) * ldo + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 w;
                w.x = pack2bf(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv);
                w.y = pack2bf(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
                *(uint2*)(orow + 32 * dt + 8 * g + 4 * h) = w;
            }
    }
}

#ifdef CVX_ABLATION
// ---------------------------------------------------------------------------------------------------------------
// 64 query rows per wave (two 32-row groups that share every K / V^T fragment read, the LDS-DMA and the barrier of a
// tile): half the LDS reads, DMA instructions and barriers per query row, and inside ONE wave the softmax VALU work of
// one group can overlap the MFMAs of the other.  NW waves per workgroup (NW*64 query rows).
// ---------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_attention64(const uint16_t* __restrict__ qk, long ldqk, const uint16_t* __restrict__ vt,
                                                         uint16_t* __restrict__ out, long ldo, int heads, int ntok, int ntp, int kp,
                                                         int C, int nqb, int xcd_remap) {
    constexpr int NT = NW * 64;
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * ATT_TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int qb, pair;
    if (xcd_remap & 1) {
        const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
        pair = (w / nqb) * 8 + xcd;
        qb = w % nqb;
    } else {
        pair = blockIdx.x / nqb;
        qb = blockIdx.x % nqb;
    }
    const int slice = pair / heads, head = pair - slice * heads;
    const long row0 = (long)slice * ntp;
    const uint16_t* Qp = qk + row0 * ldqk + head * 64;
    const uint16_t* Kp = Qp + C;
    const uint16_t* Vp = vt + ((long)(slice * heads + head) * 64) * kp;
    const int r = lane & 31, h = lane >> 5;
    const int q0 = qb * NT + wave * 64;

    bf16x8 qf[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int qrow = min(q0 + 32 * g + r, ntp - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[g][ks] = *(const bf16x8*)(Qp + (long)qrow * ldqk + 16 * ks + 8 * h);
    }
    // LDS-DMA: 512 16-B pieces per K tile and per V^T tile, NT threads
    constexpr int NP = (512 + NT - 1) / NT;
    uint32_t koffs[NP], voffs[NP];
#pragma unroll
    for (int jj = 0; jj < NP; ++jj) {
        const int c = jj * NT + tid;
        const int srow = (c >> 3) & 63, schunk = ((c & 7) ^ ((srow >> 1) & 7)) << 3;
        koffs[jj] = (uint32_t)(srow * ldqk + schunk) * 2u;
        voffs[jj] = (uint32_t)(srow * kp + schunk) * 2u;
    }
    auto issue = [&](int j, int buf) {
        const uint32_t kt = lds_addr(smem) + buf * 2 * ATT_TILE_BYTES;
        const long kv0 = (long)j * KV_TILE;
#pragma unroll
        for (int jj = 0; jj < NP; ++jj) {
            if (jj * NT + wave * 64 >= 512) break;  // wave-uniform: the last pass may cover only some waves
            glds16_saddr(Kp + kv0 * ldqk, koffs[jj], kt + (jj * NT + wave * 64) * 16);
            glds16_saddr(Vp + kv0, voffs[jj], kt + ATT_TILE_BYTES + (jj * NT + wave * 64) * 16);
        }
    };
    const int krow = pi_row(r);
    const int koff = krow * 128, ksw = (krow >> 1) & 7;
    const int voff = r * 128, vsw = (r >> 1) & 7;

    f32x16 o[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[g][0][i] = 0.f; o[g][1][i] = 0.f; }
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    const float LOG2E = 1.0f;  // (scores are already in log2 units: see the note at the Q fragments)
    const int nkv = (ntok + KV_TILE - 1) / KV_TILE;

    issue(0, 0);
    for (int j = 0; j < nkv; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (j + 1 < nkv) issue(j + 1, (j + 1) & 1);
        const char* kt = smem + (j & 1) * 2 * ATT_TILE_BYTES;
        const char* vtile = kt + ATT_TILE_BYTES;

        f32x16 s[2][2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[g][0][i] = 0.f; s[g][1][i] = 0.f; }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kt + t * 4096 + koff + (((2 * ks + h) ^ ksw) << 4));
                s[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0][ks], s[0][t], 0, 0, 0);
                s[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[1][ks], s[1][t], 0, 0, 0);
            }
        const int kv0 = j * KV_TILE;
        if (kv0 + KV_TILE > ntok) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int rho = (i & 3) + 8 * (i >> 2) + 4 * h;
                        if (kv0 + 32 * t + pi_row(rho) >= ntok) s[g][t][i] = -INFINITY;
                    }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float mloc = s[g][0][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mloc = fmaxf(mloc, s[g][0][i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) mloc = fmaxf(mloc, s[g][1][i]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float m_new = fmaxf(m_run[g], mloc);
            const float alpha = __builtin_amdgcn_exp2f((m_run[g] - m_new) * LOG2E);
            const float mb = m_new * LOG2E;
            m_run[g] = m_new;
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[g][t][i], LOG2E, -mb));
                    s[g][t][i] = p;
                    psum += p;
                }
            l_run[g] = fmaf(l_run[g], alpha, psum);
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { o[g][0][i] *= alpha; o[g][1][i] *= alpha; }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) {
                bf16x8 pf[2];
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[g][e] = (__bf16)s[g][t][8 * sk + e];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16x8 vf = *(const bf16x8*)(vtile + dt * 4096 + voff + (((4 * t + 2 * sk + h) ^ vsw) << 4));
                    o[0][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[0], o[0][dt], 0, 0, 0);
                    o[1][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[1], o[1][dt], 0, 0, 0);
                }
            }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float l_tot = l_run[g] + __shfl_xor(l_run[g], 32, 64);
        const float inv = 1.0f / l_tot;
        const int q = q0 + 32 * g + r;
        if (q < ntok) {
            uint16_t* orow = out + (row0 + q) * ldo + head * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    uint2 w;
                    w.x = pack2bf(o[g][dt][4 * gg + 0] * inv, o[g][dt][4 * gg + 1] * inv);
                    w.y = pack2bf(o[g][dt][4 * gg + 2] * inv, o[g][dt][4 * gg + 3] * inv);
                    *(uint2*)(orow + 32 * dt + 8 * gg + 4 * h) = w;
                }
        }
    }
}

#endif  // CVX_ABLATION

}  // namespace cvx

using namespace cvx;

std::atomic<int> g_attn_variant{7};     // cvx_set_option("attn_variant"); 7 = maximum subtracted inside the product, re-anchoring triggered by the probability sums (default); 6 = by the tile maxima
std::atomic<int> g_attn_xcd_remap{1};   // cvx_set_option("attn_xcd_remap")
std::atomic<int> g_attn_half_tile{1};   // cvx_set_option("attn_half_tile"): skip the all-padding upper half of the last key tile (0: A/B runs)
std::atomic<int> g_attn_mfma_prio{2};   // cvx_set_option("attn_mfma_prio"): s_setprio 1 around the wave's MFMA blocks (bit 0: S^T, bit 1: O^T).  Measured per layer:
                                        // 0: 1.028 ms, 1: 1.018, 2: 0.999 (default), 3: 1.003 -- the O^T MFMAs wait behind other waves' softmax VALU otherwise

extern "C" int cvx_attention_qkv_bf16(const void* qkv, long ld, void* out, long ldo, int slices, int heads, int ntok, int ntp,
                                      hipStream_t st) {
    if (slices <= 0) return 0;
    if (!qkv || !out) return cvx_fail("attention_qkv: null pointer");
    if (ntp % 8 || ntp < ntok || ld % 64 || ld < 3L * heads * 64 || ldo % 4)
        return cvx_fail("attention_qkv: need ntp%8==0, ntp>=ntok, ld%64==0 (the default kernel's lone-wave DMA offsets), ld >= 3*heads*64");
    const int nqb = (ntok + 127) / 128;
    const long nblk = (long)nqb * heads * slices;
    if (nblk > 0x7fffffff) return cvx_fail("attention_qkv: grid too large");
    const int xcd_remap = (((long)heads * slices) % 8 == 0 && g_attn_xcd_remap ? 1 : 0) | ((g_attn_mfma_prio.load() & 3) << 1) | (g_attn_half_tile.load() ? 0 : 8);
    hipLaunchKernelGGL((k_attention<7, true>), dim3((unsigned)nblk), dim3(ATT_THREADS), 0, st, (const uint16_t*)qkv, ld, (const uint16_t*)nullptr,
                       (uint16_t*)out, ldo, heads, ntok, ntp, /*kp (unused)*/ 0, heads * 64, nqb, xcd_remap);
    return cvx_check_launch();
}

extern "C" int cvx_attention_bf16(const void* qk, long ldqk, const void* vt, void* out, long ldo, int slices, int heads,
                                  int ntok, int ntp, int kp, hipStream_t st) {
    if (slices <= 0) return 0;
    if (ntp % 8 || kp % 64 || kp < ntok || ntp < ntok || ldqk % 8 || ldo % 4)
        return cvx_fail("attention: need ntp%8==0, kp%64==0, kp>=ntok, ntp>=ntok, ldqk%8==0");
    int variant = g_attn_variant;  // one read per call: a concurrent cvx_set_option cannot give a mixed launch
    // Variant 6's single-wave query blocks derive the departed waves' DMA offsets by XOR-ing 64 into wave 0's own, which is the other
    // waves' row term only while a K row's byte pitch is a multiple of 128 (ldqk % 64 == 0: every DINOv2 width).  Other leading
    // dimensions take the general kernel (variant 0: a running maximum per tile).
    if ((variant == 6 || variant == 7 || variant == 8 || variant == 9) && ldqk % 64 != 0) variant = 0;
    const int rows_per_block = variant == 4 ? 192 : (variant == 5 || variant == 8 || variant == 9) ? 256 : 128;
    const int nqb = (ntok + rows_per_block - 1) / rows_per_block;
    const long nblk = (long)nqb * heads * slices;
    if (nblk > 0x7fffffff) return cvx_fail("attention: grid too large");
    const int xcd_remap = (((long)heads * slices) % 8 == 0 && g_attn_xcd_remap ? 1 : 0) | ((g_attn_mfma_prio.load() & 3) << 1) | (g_attn_half_tile.load() ? 0 : 8);
    dim3 grid((unsigned)nblk);
    void (*k)(const uint16_t*, long, const uint16_t*, uint16_t*, long, int, int, int, int, int, int, int);
    switch (variant) {
        case 7: k = k_attention<7>; break;
#ifdef CVX_ABLATION  // earlier / rejected schedules (parity-tested on the ablation build) and timing-only ones with garbage output
        case 8: k = k_attention<8>; break;  // 8-wave workgroups, three buffers: measured 1.001 -> 1.123 ms per layer
        case 9: k = k_attention<9>; break;  // 8-wave workgroups, two buffers: 1.082 ms
        case 1: k = k_attention<1>; break;
        case 3: k = k_attention<3>; break;
        case 6: k = k_attention<6>; break;
        case 10: k = k_attention<10>; break;
        case 11: k = k_attention<11>; break;
        case 12: k = k_attention<12>; break;
        case 13: k = k_attention<13>; break;
#endif
        default: k = k_attention<0>; break;
    }
#ifdef CVX_ABLATION
    if (variant == 4 || variant == 5) {
        auto k64 = variant == 4 ? k_attention64<3> : k_attention64<4>;
        hipLaunchKernelGGL(k64, grid, dim3(rows_per_block), 0, st, (const uint16_t*)qk, ldqk, (const uint16_t*)vt, (uint16_t*)out, ldo,
                           heads, ntok, ntp, kp, heads * 64, nqb, xcd_remap);
        return cvx_check_launch();
    }
#endif
    hipLaunchKernelGGL(k, grid, dim3((variant == 8 || variant == 9) ? 512 : ATT_THREADS), 0, st, (const uint16_t*)qk, ldqk, (const uint16_t*)vt, (uint16_t*)out, ldo, heads,
                       ntok, ntp, kp, heads * 64, nqb, xcd_remap);
    return cvx_check_launch();
}
