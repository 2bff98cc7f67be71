// Kernels of the alternate encoder (SAM2.1 Hiera image encoder + FPN neck, BASELINE configs[4]) that the GEMM / LayerNorm
// kernels of the ViT path do not already cover:
//   k_sam_patches     bilinear resize to the encoder's image size + 7x7/stride-4/pad-3 patch gather -> bf16 GEMM operand
//   k_win_attention   windowed / global multi-head attention on a token grid, head dims 56..96 (72 for Hiera-L), with the
//                     queries optionally living on a 2x2-pooled grid (stage transitions)
//   k_pool2x2         2x2 max pool over the token grid (fp32 residual shortcut, bf16 queries)
//   k_cast_bf16       fp32 residual stream -> bf16 GEMM operand (FPN lateral convs)
//   k_fpn_out         lateral (+ nearest-upsampled coarser level) -> float16 [D][C][h][w]
// Activations are channels-last rows: row = (slice * G + y) * G + x.
#include "common.h"
#include "../../include/cryovit_hip.h"
#include "host_util.h"
#include <atomic>

namespace cvx {

typedef short v4s __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------
// Patch gather.  out[(d*G + py)*G + px][c*49 + ky*7 + kx] = img(d, c, 4*py + ky - 3, 4*px + kx - 3)  (0 outside),
// img = the [S][S] bilinear resample (align_corners = False, as F.interpolate(..., "trilinear") with an unchanged channel
// extent does: src = max(0, (dst + 0.5) * H/S - 0.5)) of the source slice; identity when H == W == S.
// mode 0: src uint8 [D][H][W] scaled by 1/255, 3 equal channels; 1: float [D][H][W], 3 equal channels; 2: float [D][3][H][W].
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float src_px(const void* __restrict__ src, int mode, long plane, int H, int W, int y, int x) {
    const long i = plane * H * W + (long)y * W + x;
    return mode == 0 ? (float)((const uint8_t*)src)[i] * (1.0f / 255.0f) : ((const float*)src)[i];
}

__global__ __launch_bounds__(256) void k_sam_patches(const void* __restrict__ src, int mode, int D, int H, int W, int S,
                                                     uint16_t* __restrict__ out, int ldo) {
    const int G = S / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)D * G * G * 160;  // 147 taps rounded up to a multiple of 32 lanes' worth
    if (idx >= total) return;
    const int k = (int)(idx % 160);
    const long row = idx / 160;
    if (k >= 147) return;
    const int px = (int)(row % G), py = (int)((row / G) % G), d = (int)(row / ((long)G * G));
    const int c = k / 49, ky = (k % 49) / 7, kx = k % 7;
    const int y = 4 * py + ky - 3, x = 4 * px + kx - 3;
    float v = 0.f;
    if ((unsigned)y < (unsigned)S && (unsigned)x < (unsigned)S) {
        const long plane = mode == 2 ? (long)d * 3 + c : d;
        if (H == S && W == S) {
            v = src_px(src, mode, plane, H, W, y, x);
        } else {
            const float fy = fmaxf(((float)y + 0.5f) * ((float)H / (float)S) - 0.5f, 0.f);
            const float fx = fmaxf(((float)x + 0.5f) * ((float)W / (float)S) - 0.5f, 0.f);
            const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
            const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
            const float ly = fy - (float)y0, lx = fx - (float)x0;
            const float a = src_px(src, mode, plane, H, W, y0, x0), b = src_px(src, mode, plane, H, W, y0, x1);
            const float e = src_px(src, mode, plane, H, W, y1, x0), f = src_px(src, mode, plane, H, W, y1, x1);
            v = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * e + lx * f);
        }
    }
    out[row * ldo + k] = f2bf(v);
}

// ---------------------------------------------------------------------------------------------------
// Windowed attention.  One workgroup = one (slice, window, head, tile of 64 queries); wave w owns queries 16w..16w+15.
// Swapped products on v_mfma_f32_16x16x16_bf16 (A rows x B columns):
//   S^T[key][q] = K[key][:] . Q[q][:]     A = K rows from LDS (ds_read_b64), B = Q (registers)
//   O^T[d][q]  += V^T[d][key] P^T[key][q]  A = V^T via ds_read_b64_tr_b16 (V stays row-major in LDS), B = P^T
// The accumulator layout of S^T (lane: q = lane%16, keys 4g..4g+3) IS the B-operand layout of P^T: probabilities go
// from the softmax to the second product without leaving the lane.  Online softmax over key tiles of 64.
// ---------------------------------------------------------------------------------------------------
// PF (256-thread blocks, head_dim % 8 == 0, power-of-two window, 16-B aligned rows): the K / V tile of the NEXT 64 keys is fetched into
// registers while the current one is consumed and written to LDS after the barrier (16-B pieces, row / column of every piece
// computed once) -- the plain form stages each tile synchronously with 8-B pieces and two divisions per piece.
// X32 (head_dim % 8 == 0): both products on v_mfma_f32_16x16x32_bf16, twice the rate of the 16x16x16 form on gfx950.  The k-slots
// 8g .. 8g+7 of a lane are 4 + 4 values: for S^T the head dims 32s + 8g .. +7 (one 16-B read of a K row, head dim padded to a
// multiple of 32 with zero columns); for O^T the keys 16j + 4g .. +3 and 16(j+1) + 4g .. +3 of TWO 16-key sub-tiles -- exactly the
// two accumulator fragments (-> P^T) and the two transposed V reads the 16x16x16 form fed to two separate MFMAs.
typedef short v8s __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 cat8(v4s a, v4s b) { return __builtin_bit_cast(bf16x8, v8s{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}); }

template <int DSTEPS, bool PF = false, bool X32 = false, bool PF2 = false>
__global__ __launch_bounds__(256) void k_win_attention(const uint16_t* __restrict__ q, long ldq, const uint16_t* __restrict__ k,
                                                       const uint16_t* __restrict__ v, long ldkv, uint16_t* __restrict__ out,
                                                       long ldo, int hd, int heads, int G, int ws, int Gq, int wsq,
                                                       float scale_log2e) {
    constexpr int NQ = (DSTEPS + 1) / 2;                      // X32: 32-deep k-steps of the S^T product
    constexpr int KW = X32 ? 32 * NQ : 16 * DSTEPS;           // staged row width (head dim padded with zero columns)
    constexpr int LDR = KW + 8;  // LDS row stride (elements): 16-lane row reads land on distinct bank groups (8-B and 16-B reads)
    __shared__ __attribute__((aligned(16))) uint16_t Ks[64 * LDR];
    __shared__ __attribute__((aligned(16))) uint16_t Vs[64 * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    // XCD-aware block order: linear ids b and b + 8 share an XCD (and its L2).  The query blocks of one (slice, window, head) read the
    // same K / V rows: re-numbering the grid so that an XCD takes a contiguous run of linear ids puts them behind one L2.
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned long nb = (unsigned long)gridDim.x * gridDim.y * gridDim.z;
        if ((nb & 7) == 0) {
            const unsigned long lin = bx + (unsigned long)gridDim.x * (by + (unsigned long)gridDim.y * bz);
            const unsigned long vb = (lin & 7) * (nb >> 3) + (lin >> 3);
            bx = (unsigned)(vb % gridDim.x);
            const unsigned long t = vb / gridDim.x;
            by = (unsigned)(t % gridDim.y);
            bz = (unsigned)(t / gridDim.y);
        }
    }
    const int head = by % heads, win = by / heads;
    const int nwx = G / ws, wy = win / nwx, wx = win % nwx;
    const long d = bz;
    const int nq = wsq * wsq, nk = ws * ws;
    const int nthreads = blockDim.x;

    // zero the pad columns [hd, KW) of both tiles once: staging never touches them
    for (int i = tid; i < 64 * (KW - hd); i += nthreads) {
        const int r = i / (KW - hd), c = hd + i % (KW - hd);
        Ks[r * LDR + c] = 0;
        Vs[r * LDR + c] = 0;
    }
    if constexpr (X32) {
        // a 32-deep k-step of the second product spans TWO 16-key sub-tiles: when the last one is missing (16 / 48-key tiles) its V
        // rows are multiplied by P = 0 -- they must be finite, so the rows no tile will write start out as zeros
        for (int i = tid; i < 64 * (hd >> 1); i += nthreads) ((uint32_t*)Vs)[(i / (hd >> 1)) * (LDR >> 1) + i % (hd >> 1)] = 0u;
    }

    // this lane's query (B operand: column q = li, k = 16s + 4g .. +3)
    const int ql = bx * 64 + wave * 16 + li;
    const bool q_ok = ql < nq;
    const int qc = q_ok ? ql : 0;
    const long qrow = d * Gq * Gq + (long)(wy * wsq + qc / wsq) * Gq + wx * wsq + qc % wsq;
    v4s qf[X32 ? 1 : DSTEPS];
    [[maybe_unused]] bf16x8 qf8[X32 ? NQ : 1];
    if constexpr (X32) {
#pragma unroll
        for (int s = 0; s < NQ; ++s) {
            const int c = 32 * s + 8 * g;
            const uint4 z{0u, 0u, 0u, 0u};
            qf8[s] = __builtin_bit_cast(bf16x8, c < hd ? *(const uint4*)(q + qrow * ldq + (long)head * hd + c) : z);
        }
    } else {
#pragma unroll
    for (int s = 0; s < DSTEPS; ++s) {
        const int c = 16 * s + 4 * g;
        if (c < hd) qf[s] = *(const v4s*)(q + qrow * ldq + (long)head * hd + c);
        else qf[s] = v4s{0, 0, 0, 0};
    }
    }

    f32x4 o[DSTEPS];
#pragma unroll
    for (int t = 0; t < DSTEPS; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;

    const int chunks = hd >> 2;  // 8-byte pieces per row
    // PF: this thread's (at most three) 16-B pieces of a tile: tile row, column, LDS offset -- the same for every tile
    constexpr int NPF = 3;
    [[maybe_unused]] int prow[NPF], pcol[NPF];
    // PF2: TWO tiles of look-ahead (two register sets, the tile loop unrolled by two): with one, the 64-key tile's compute time
    // (~0.35 us) is shorter than the L2 / HBM round trip of the next tile's loads and every tile waits for its data
    [[maybe_unused]] uint4 rk[NPF], rv[NPF], rk2[PF2 ? NPF : 1], rv2[PF2 ? NPF : 1];
    [[maybe_unused]] const int wsh = 31 - __builtin_clz(ws);
    if constexpr (PF) {
        const int ch16 = hd >> 3;
#pragma unroll
        for (int pp = 0; pp < NPF; ++pp) {
            const int i = tid + 256 * pp;
            prow[pp] = i < 64 * ch16 ? i / ch16 : -1;
            pcol[pp] = i < 64 * ch16 ? (i % ch16) * 8 : 0;
            rk[pp] = uint4{0u, 0u, 0u, 0u};
            rv[pp] = uint4{0u, 0u, 0u, 0u};
            if constexpr (PF2) { rk2[pp] = uint4{0u, 0u, 0u, 0u}; rv2[pp] = uint4{0u, 0u, 0u, 0u}; }
        }
    }
    [[maybe_unused]] auto prefetch = [&](int k0, uint4 (&rk)[NPF], uint4 (&rv)[NPF]) {
        const int nkt = min(64, nk - k0);
#pragma unroll
        for (int pp = 0; pp < NPF; ++pp) {
            // (unconditional loads from a clamped row: a conditional assignment sent the staging registers to scratch)
            const int kl = k0 + ((prow[pp] >= 0 && prow[pp] < nkt) ? prow[pp] : 0);
            const long krow = d * G * G + (long)(wy * ws + (kl >> wsh)) * G + wx * ws + (kl & (ws - 1));
            rk[pp] = *(const uint4*)(k + krow * ldkv + (long)head * hd + pcol[pp]);
            rv[pp] = *(const uint4*)(v + krow * ldkv + (long)head * hd + pcol[pp]);
        }
    };
    if constexpr (PF) prefetch(0, rk, rv);
    if constexpr (PF2) { if (nk > 64) prefetch(64, rk2, rv2); }
    auto tile = [&](int k0, uint4 (&rk)[NPF], uint4 (&rv)[NPF]) {
        const int nkt = min(64, nk - k0), nsub = nkt >> 4;
        __syncthreads();  // previous tile fully consumed (also orders the pad zeroing before the first reads)
        if constexpr (PF) {
#pragma unroll
            for (int pp = 0; pp < NPF; ++pp) {
                if (prow[pp] >= 0 && prow[pp] < nkt) {
                    *(uint4*)(Ks + prow[pp] * LDR + pcol[pp]) = rk[pp];
                    *(uint4*)(Vs + prow[pp] * LDR + pcol[pp]) = rv[pp];
                }
            }
        } else {
        // four pieces of K and of V are loaded before any of them is stored: one load -> store pair per iteration made every
        // iteration wait for its own HBM round trip (the 64-thread blocks of the last stage spent 18 of them per tile)
        for (int i0 = tid; i0 < nkt * chunks; i0 += 4 * nthreads) {
            uint2 tk[4], tv[4];
            int off[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * nthreads;
                const int ic = i < nkt * chunks ? i : i0;  // (clamped duplicate, not stored)
                const int r = ic / chunks, c = (ic % chunks) * 4;
                const int kl = k0 + r;
                const long krow = d * G * G + (long)(wy * ws + kl / ws) * G + wx * ws + kl % ws;
                tk[u] = *(const uint2*)(k + krow * ldkv + (long)head * hd + c);
                tv[u] = *(const uint2*)(v + krow * ldkv + (long)head * hd + c);
                off[u] = i < nkt * chunks ? r * LDR + c : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (off[u] >= 0) {
                    *(uint2*)(Ks + off[u]) = tk[u];
                    *(uint2*)(Vs + off[u]) = tv[u];
                }
            }
        }
        }
        __syncthreads();
        if constexpr (PF) {
            if (k0 + (PF2 ? 128 : 64) < nk) prefetch(k0 + (PF2 ? 128 : 64), rk, rv);  // in flight while this tile (and, PF2, the next) is consumed
        }

        f32x4 sacc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < nsub) {
                if constexpr (X32) {
#pragma unroll
                    for (int s = 0; s < NQ; ++s) {
                        const bf16x8 kf = *(const bf16x8*)(Ks + (16 * j + li) * LDR + 32 * s + 8 * g);
                        sacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf8[s], sacc[j], 0, 0, 0);
                    }
                } else {
#pragma unroll
                for (int s = 0; s < DSTEPS; ++s) {
                    const v4s kf = *(const v4s*)(Ks + (16 * j + li) * LDR + 16 * s + 4 * g);
                    sacc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kf, qf[s], sacc[j], 0, 0, 0);
                }
                }
            }
        }
        // online softmax for query li: this lane holds keys 16j + 4g + i; the other keys sit in lanes li + 16, 32, 48
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nsub) mx = fmaxf(mx, fmaxf(fmaxf(sacc[j][0], sacc[j][1]), fmaxf(sacc[j][2], sacc[j][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx * scale_log2e);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        float ps = 0.f;
        v4s pf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < nsub) {
                float p[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    p[i] = __builtin_amdgcn_exp2f(fmaf(sacc[j][i], scale_log2e, -m_new));
                    ps += p[i];
                }
                const uint32_t w0 = pack2bf(p[0], p[1]), w1 = pack2bf(p[2], p[3]);
                pf[j] = v4s{(short)(w0 & 0xffff), (short)(w0 >> 16), (short)(w1 & 0xffff), (short)(w1 >> 16)};
            } else {
                pf[j] = v4s{0, 0, 0, 0};
            }
        }
        ps += __shfl_xor(ps, 16, 64);
        ps += __shfl_xor(ps, 32, 64);
        l = l * alpha + ps;
        m = m_new;
        // V^T fragments: block of keys 16j+4g .. +3 x dims 16t .. 16t+15; lane 4r+p of a 16-lane group supplies row r, dims 4p..
        // All 4*DSTEPS transposed reads are issued back to back and waited for once (rows past the tile are read, not used).
        const uint32_t vbase = (uint32_t)(size_t)(Vs + (4 * g + (li >> 2)) * LDR + 4 * (li & 3));
        v4s vf[DSTEPS][4];
#pragma unroll
        for (int t = 0; t < DSTEPS; ++t)
            asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%5\n\t"
                         "ds_read_b64_tr_b16 %1, %4 offset:%6\n\t"
                         "ds_read_b64_tr_b16 %2, %4 offset:%7\n\t"
                         "ds_read_b64_tr_b16 %3, %4 offset:%8"
                         : "=&v"(vf[t][0]), "=&v"(vf[t][1]), "=&v"(vf[t][2]), "=&v"(vf[t][3])
                         : "v"(vbase), "n"(32 * t), "n"(32 * t + 32 * LDR), "n"(32 * t + 64 * LDR), "n"(32 * t + 96 * LDR)
                         : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < DSTEPS; ++t) {
            // ties the fragments to the wait above: the MFMAs below consume the outputs of this (ordered) statement
            asm volatile("" : "+v"(vf[t][0]), "+v"(vf[t][1]), "+v"(vf[t][2]), "+v"(vf[t][3]));
            o[t] *= alpha;
            if constexpr (X32) {
                o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(vf[t][0], vf[t][1]), cat8(pf[0], pf[1]), o[t], 0, 0, 0);  // (pf[1] = 0 on a 16-key tile)
                if (nsub > 2) o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(vf[t][2], vf[t][3]), cat8(pf[2], pf[3]), o[t], 0, 0, 0);
                continue;
            }
            o[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf[t][0], pf[0], o[t], 0, 0, 0);
            if (nsub > 1) o[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf[t][1], pf[1], o[t], 0, 0, 0);
            if (nsub > 2) o[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf[t][2], pf[2], o[t], 0, 0, 0);
            if (nsub > 3) o[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf[t][3], pf[3], o[t], 0, 0, 0);
        }
    };
    if constexpr (PF2) {
        static_assert(PF, "PF2 extends PF");
        for (int k0 = 0; k0 < nk; k0 += 128) {
            tile(k0, rk, rv);
            if (k0 + 64 < nk) tile(k0 + 64, rk2, rv2);
        }
    } else {
        for (int k0 = 0; k0 < nk; k0 += 64) tile(k0, rk, rv);
    }
    if (!q_ok) return;
    const float inv = 1.0f / l;
    uint16_t* dst = out + qrow * ldo + (long)head * hd;
#pragma unroll
    for (int t = 0; t < DSTEPS; ++t) {
        const int c = 16 * t + 4 * g;  // O^T rows (dims) 16t + 4g .. +3 of column q = li
        if (c < hd) {
            uint2 w;
            w.x = pack2bf(o[t][0] * inv, o[t][1] * inv);
            w.y = pack2bf(o[t][2] * inv, o[t][3] * inv);
            *(uint2*)(dst + c) = w;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 2x2 max pool over the token grid: in rows (d, y, x) on a G x G grid -> out rows on (G/2) x (G/2); C channels from column 0.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pool2x2_f32(const float* __restrict__ in, long ldi, float* __restrict__ out, long ldo,
                                                     int D, int G, int C) {
    const int Go = G / 2, c4 = C / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)D * Go * Go * c4) return;
    const int c = (int)(idx % c4) * 4;
    const long ro = idx / c4;
    const int x = (int)(ro % Go), y = (int)((ro / Go) % Go);
    const long d = ro / ((long)Go * Go);
    const float* p = in + ((d * G + 2 * y) * G + 2 * x) * ldi + c;
    const float4 a = *(const float4*)p, b = *(const float4*)(p + ldi), e = *(const float4*)(p + (long)G * ldi),
                 f = *(const float4*)(p + (long)G * ldi + ldi);
    float4 r;
    r.x = fmaxf(fmaxf(a.x, b.x), fmaxf(e.x, f.x)); r.y = fmaxf(fmaxf(a.y, b.y), fmaxf(e.y, f.y));
    r.z = fmaxf(fmaxf(a.z, b.z), fmaxf(e.z, f.z)); r.w = fmaxf(fmaxf(a.w, b.w), fmaxf(e.w, f.w));
    *(float4*)(out + ro * ldo + c) = r;
}

__device__ __forceinline__ uint32_t bfmax2(uint32_t a, uint32_t b) { return pack2bf(fmaxf(bflo(a), bflo(b)), fmaxf(bfhi(a), bfhi(b))); }

__global__ __launch_bounds__(256) void k_pool2x2_bf16(const uint16_t* __restrict__ in, long ldi, uint16_t* __restrict__ out,
                                                      long ldo, int D, int G, int C) {
    const int Go = G / 2, c4 = C / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)D * Go * Go * c4) return;
    const int c = (int)(idx % c4) * 4;
    const long ro = idx / c4;
    const int x = (int)(ro % Go), y = (int)((ro / Go) % Go);
    const long d = ro / ((long)Go * Go);
    const uint16_t* p = in + ((d * G + 2 * y) * G + 2 * x) * ldi + c;
    const uint2 a = *(const uint2*)p, b = *(const uint2*)(p + ldi), e = *(const uint2*)(p + (long)G * ldi),
                f = *(const uint2*)(p + (long)G * ldi + ldi);
    uint2 r;
    r.x = bfmax2(bfmax2(a.x, b.x), bfmax2(e.x, f.x));
    r.y = bfmax2(bfmax2(a.y, b.y), bfmax2(e.y, f.y));
    *(uint2*)(out + ro * ldo + c) = r;
}

__global__ __launch_bounds__(256) void k_cast_bf16(const float* __restrict__ in, long ldi, uint16_t* __restrict__ out, long ldo,
                                                   long rows, int C) {
    const int c4 = C / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * c4) return;
    const int c = (int)(idx % c4) * 4;
    const long r = idx / c4;
    const float4 a = *(const float4*)(in + r * ldi + c);
    uint2 w;
    w.x = pack2bf(a.x, a.y);
    w.y = pack2bf(a.z, a.w);
    *(uint2*)(out + r * ldo + c) = w;
}

// FPN level output: out[d][c][y][x] (float16) = lat[(d*g + y)*g + x][c] (+ coarse[(d*g/2 + y/2)*g/2 + x/2][c]), 64x64 tiles
__global__ __launch_bounds__(256) void k_fpn_out(const float* __restrict__ lat, const float* __restrict__ coarse, int C, int g,
                                                 _Float16* __restrict__ out) {
    __shared__ float tile[64][65];
    const int hw = g * g;
    const long d = blockIdx.z;
    const int t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int t = t0 + i, c = c0 + tx;
        float v = 0.f;
        if (t < hw && c < C) {
            v = lat[(d * hw + t) * C + c];
            if (coarse) {
                const int y = t / g, x = t % g, gc = g / 2;
                v += coarse[(d * gc * gc + (long)(y / 2) * gc + x / 2) * C + c];
            }
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, t = t0 + tx;
        if (c < C && t < hw) out[(d * C + c) * hw + t] = (_Float16)tile[tx][i];
    }
}

}  // namespace cvx

using namespace cvx;

std::atomic<int> g_win_attn_prefetch{1};  // cvx_set_option("win_attn_prefetch", 0 / 1 / 2): K / V register prefetch, one tile ahead (default) or two (measured
                                          // 104.5 -> 106.9 ms per tomogram: 164 VGPRs, a wave per SIMD less) -- gemm.hip
std::atomic<int> g_win_attn_x32{0};       // cvx_set_option("win_attn_x32", 0 / 1): 16x16x32 products.  Parity-green, measured 113.4 -> 115.4 ms per tomogram on
                                          // sam_features (the kernel is not matrix-pipe bound: wider padded rows cost more than the halved MFMA count saves): off

extern "C" int cvx_sam_patches(const void* src, int mode, int D, int H, int W, int S, void* out, long ldo, hipStream_t st) {
    if (D <= 0) return 0;
    if (mode < 0 || mode > 2 || S <= 0 || S % 4 || ldo < 147 || H <= 0 || W <= 0) return cvx_fail("sam_patches: bad arguments");
    const long total = (long)D * (S / 4) * (S / 4) * 160;
    hipLaunchKernelGGL(k_sam_patches, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, mode, D, H, W, S, (uint16_t*)out,
                       (int)ldo);
    return cvx_check_launch();
}

extern "C" int cvx_window_attention_bf16(const void* q, long ldq, const void* k, const void* v, long ldkv, void* out, long ldo,
                                         int slices, int heads, int head_dim, int grid, int window, int q_grid, int q_window,
                                         hipStream_t st) {
    if (slices <= 0) return 0;
    if (window <= 0 || grid % window || q_window <= 0 || q_grid % q_window || grid / window != q_grid / q_window)
        return cvx_fail("window_attention: the key and query grids must hold the same whole number of windows");
    const int nk = window * window, nq = q_window * q_window;
    if (nk % 16) return cvx_fail("window_attention: keys per window must be a multiple of 16");
    if (head_dim % 4 || head_dim < 8 || head_dim > 96 || ldq % 4 || ldkv % 4 || ldo % 4)
        return cvx_fail("window_attention: head_dim must be a multiple of 4 in [8, 96], leading dimensions multiples of 4");
    const int nwin = (grid / window) * (grid / window);
    if ((long)nwin * heads > 65535 || slices > 65535) return cvx_fail("window_attention: grid too large");
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)head_dim);
    const dim3 blocks((nq + 63) / 64, nwin * heads, slices);
    const dim3 threads(64 * (nq >= 64 ? 4 : (nq + 15) / 16));
    const uint16_t *qq = (const uint16_t*)q, *kk = (const uint16_t*)k, *vv = (const uint16_t*)v;
    // register-prefetch form: full 256-thread blocks, 16-B pieces (head_dim % 8, row starts 16-B aligned), power-of-two window
    const bool pf = g_win_attn_prefetch && threads.x == 256 && head_dim % 8 == 0 && ldkv % 8 == 0 && (window & (window - 1)) == 0 &&
                    ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0;
    const bool pf2 = pf && g_win_attn_prefetch >= 2 && nk > 64;  // two tiles of look-ahead (windows of more than one 64-key tile)
    // 16x16x32 products: 16-B aligned 8-element pieces of Q and of the staged rows
    const bool x32 = g_win_attn_x32 && head_dim % 8 == 0 && ldq % 8 == 0 && ((uintptr_t)q & 15) == 0;
#define CVX_WIN_LAUNCH(DS)                                                                                                          \
    do {                                                                                                                            \
        if (pf2 && !x32) hipLaunchKernelGGL((k_win_attention<DS, true, false, true>), blocks, threads, 0, st, qq, ldq, kk, vv, ldkv, (uint16_t*)out, ldo, \
                                    head_dim, heads, grid, window, q_grid, q_window, scale_log2e);                                  \
        else if (x32 && pf) hipLaunchKernelGGL((k_win_attention<DS, true, true>), blocks, threads, 0, st, qq, ldq, kk, vv, ldkv, (uint16_t*)out, ldo, \
                                    head_dim, heads, grid, window, q_grid, q_window, scale_log2e);                                  \
        else if (x32) hipLaunchKernelGGL((k_win_attention<DS, false, true>), blocks, threads, 0, st, qq, ldq, kk, vv, ldkv, (uint16_t*)out, ldo, \
                                    head_dim, heads, grid, window, q_grid, q_window, scale_log2e);                                  \
        else if (pf) hipLaunchKernelGGL((k_win_attention<DS, true>), blocks, threads, 0, st, qq, ldq, kk, vv, ldkv, (uint16_t*)out, ldo, \
                                   head_dim, heads, grid, window, q_grid, q_window, scale_log2e);                                   \
        else hipLaunchKernelGGL((k_win_attention<DS, false>), blocks, threads, 0, st, qq, ldq, kk, vv, ldkv, (uint16_t*)out, ldo,    \
                                head_dim, heads, grid, window, q_grid, q_window, scale_log2e);                                      \
    } while (0)
    if (head_dim <= 64) CVX_WIN_LAUNCH(4);
    else if (head_dim <= 80) CVX_WIN_LAUNCH(5);
    else CVX_WIN_LAUNCH(6);
#undef CVX_WIN_LAUNCH
    return cvx_check_launch();
}

extern "C" int cvx_pool2x2(const void* in, long ldi, void* out, long ldo, int slices, int grid, int C, int is_bf16,
                           hipStream_t st) {
    if (slices <= 0) return 0;
    if (grid % 2 || C % 4 || ldi % 4 || ldo % 4) return cvx_fail("pool2x2: even grid, C and leading dimensions multiples of 4 required");
    const long total = (long)slices * (grid / 2) * (grid / 2) * (C / 4);
    const dim3 blocks((unsigned)((total + 255) / 256));
    if (is_bf16)
        hipLaunchKernelGGL(k_pool2x2_bf16, blocks, dim3(256), 0, st, (const uint16_t*)in, ldi, (uint16_t*)out, ldo, slices, grid, C);
    else
        hipLaunchKernelGGL(k_pool2x2_f32, blocks, dim3(256), 0, st, (const float*)in, ldi, (float*)out, ldo, slices, grid, C);
    return cvx_check_launch();
}

extern "C" int cvx_cast_bf16(const float* in, long ldi, void* out, long ldo, long rows, int C, hipStream_t st) {
    if (rows <= 0) return 0;
    if (C % 4 || ldi % 4 || ldo % 4) return cvx_fail("cast_bf16: C and leading dimensions must be multiples of 4");
    const long total = rows * (C / 4);
    hipLaunchKernelGGL(k_cast_bf16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, ldi, (uint16_t*)out, ldo, rows, C);
    return cvx_check_launch();
}

extern "C" int cvx_fpn_level_out(const float* lateral, const float* coarse, int slices, int C, int grid, void* out_f16,
                                 hipStream_t st) {
    if (slices <= 0) return 0;
    if (coarse && grid % 2) return cvx_fail("fpn_level_out: top-down addition needs an even grid");
    const dim3 blocks((grid * grid + 63) / 64, (C + 63) / 64, slices);
    hipLaunchKernelGGL(k_fpn_out, blocks, dim3(256), 0, st, lateral, coarse, C, grid, (_Float16*)out_f16);
    return cvx_check_launch();
}
