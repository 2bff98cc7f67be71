// Normalisation and layout kernels (HBM-bound): LayerNorm fp32 -> bf16, final LayerNorm + feature layouts,
// GroupNorm over channels-last fp16 volumes (the head), fp16 [C][D][h][w] -> fp16 channels-last.
#include "common.h"
#include "../../include/cryovit_hip.h"
#include "host_util.h"
#include <atomic>

namespace cvx {

constexpr int LN_MAXJ = 8;  // float4 per lane: C <= 64*4*8 = 2048

// One wave per row; the row lives in registers (two-pass mean / centred variance, fp32).
// The token stream is either one fp32 array or the bf16 (hi, lo) pair of the ViT path (x = hi + lo, exact in fp32)
struct RowF32 {
    const float* row;
    __device__ __forceinline__ float4 ld4(int i) const { return *(const float4*)(row + 4 * i); }
    __device__ __forceinline__ float ld1(int c) const { return row[c]; }
};
struct RowHL {
    const uint16_t* hi; const uint16_t* lo;
    __device__ __forceinline__ float4 ld4(int i) const {
        const uint2 a = *(const uint2*)(hi + 4 * i), b = *(const uint2*)(lo + 4 * i);
        return float4{bflo(a.x) + bflo(b.x), bfhi(a.x) + bfhi(b.x), bflo(a.y) + bflo(b.y), bfhi(a.y) + bfhi(b.y)};
    }
    __device__ __forceinline__ float ld1(int c) const { return bf2f(hi[c]) + bf2f(lo[c]); }
};

template <class Row>
__device__ __forceinline__ void ln_row_stats(const Row& row, int C4, int lane, float4 (&v)[LN_MAXJ], float& mean,
                                             float& rstd, int C, float eps) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
        const int i = lane + 64 * j;
        if (i < C4) {
            v[j] = row.ld4(i);
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
    }
    mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
        const int i = lane + 64 * j;
        if (i < C4) {
            const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    rstd = rsqrtf(wave_sum(q) / (float)C + eps);
}

// Narrow rows (C < 1024: ViT-S/B, the Hiera stages): 4 channels per lane and step keeps more of the 64 lanes busy; plain
// loads, because these activations are small enough to still sit in the L2 / MALL when the LayerNorm reads them.
__global__ __launch_bounds__(256) void k_layernorm_bf16_v4(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                        const float* __restrict__ b, uint16_t* __restrict__ out, long ldo,
                                                        long rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int C4 = C >> 2;
    float4 v[LN_MAXJ];
    float mean, rstd;
    ln_row_stats(RowF32{x + row * ldx}, C4, lane, v, mean, rstd, C, eps);
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
        const int i = lane + 64 * j;
        if (i < C4) {
            const float4 ww = *(const float4*)(w + 4 * i), bb = *(const float4*)(b + 4 * i);
            uint2 o;
            o.x = pack2bf((v[j].x - mean) * rstd * ww.x + bb.x, (v[j].y - mean) * rstd * ww.y + bb.y);
            o.y = pack2bf((v[j].z - mean) * rstd * ww.z + bb.z, (v[j].w - mean) * rstd * ww.w + bb.w);
            *(uint2*)(out + row * ldo + 4 * i) = o;
        }
    }
}


// Narrower still (C <= 384: Hiera stages 1 and 2 with 144 / 288 channels): at one row per wave only 36 / 72 of the 64 / 128 lane slots
// carry data.  Here a row takes LPR = 16 / 32 lanes, a wave 4 / 2 rows, a lane NJ float4s (lane-strided): 75 % of the slots, and
// four / two rows' loads in flight per wave.  Reductions stay inside the LPR-lane group (xor butterflies below LPR).
template <int LPR, int NJ>
__global__ __launch_bounds__(256) void k_layernorm_bf16_rows(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                             const float* __restrict__ b, uint16_t* __restrict__ out, long ldo,
                                                             long rows, int C, float eps) {
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, sub = lane % LPR;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool live = row < rows;
    const long rr = live ? row : rows - 1;  // (the clamped duplicate is computed and not stored: keeps the shuffles convergent)
    const int C4 = C >> 2;
    float4 v[NJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = sub + LPR * j;
        v[j] = i < C4 ? *(const float4*)(x + rr * ldx + 4 * i) : float4{0.f, 0.f, 0.f, 0.f};
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (sub + LPR * j < C4) {
            const float a0 = v[j].x - mean, a1 = v[j].y - mean, a2 = v[j].z - mean, a3 = v[j].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / (float)C + eps);
    if (!live) return;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = sub + LPR * j;
        if (i < C4) {
            const float4 ww = *(const float4*)(w + 4 * i), bb = *(const float4*)(b + 4 * i);
            uint2 o2;
            o2.x = pack2bf((v[j].x - mean) * rstd * ww.x + bb.x, (v[j].y - mean) * rstd * ww.y + bb.y);
            o2.y = pack2bf((v[j].z - mean) * rstd * ww.z + bb.z, (v[j].w - mean) * rstd * ww.w + bb.w);
            *(uint2*)(out + row * ldo + 4 * i) = o2;
        }
    }
}

// LN_V8: 8 consecutive channels per lane and step (two adjacent 16-B loads, ONE 16-B store of 8 bf16) -- full-width stores
// instead of the 8-B ones of the float4 mapping; LN_ROWS rows per wave keep twice the loads in flight.
constexpr int LN_MAXJ8 = 4;   // 8-channel chunks per lane: C <= 64*8*4 = 2048
constexpr int LN_ROWS = 2;

// POLICY bit 0: plain (cacheable) loads instead of streaming ones; bit 1: blocks walk the rows from the END of the stream (the rows
// the producing GEMM wrote last may still sit in the Infinity Cache) -- A/B switches, cvx_set_option("ln_policy", v)
template <int POLICY>
__global__ __launch_bounds__(256) void k_layernorm_bf16(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                        const float* __restrict__ b, uint16_t* __restrict__ out, long ldo,
                                                        long rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long blk = (POLICY & 2) ? (long)gridDim.x - 1 - blockIdx.x : (long)blockIdx.x;
    const long row0 = (blk * 4 + (threadIdx.x >> 6)) * LN_ROWS;
    if (row0 >= rows) return;
    const int C8 = C >> 3;
    f32x4 v[LN_ROWS][LN_MAXJ8][2];
    float s[LN_ROWS];
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        s[r] = 0.f;
        const long row = row0 + r < rows ? row0 + r : rows - 1;  // the clamped duplicate is computed and not stored
        const float* xr = x + row * ldx;
#pragma unroll
        for (int j = 0; j < LN_MAXJ8; ++j) {
            const int i = lane + 64 * j;
            if (i < C8) {
                v[r][j][0] = (POLICY & 1) ? *(const f32x4*)(xr + 8 * i) : __builtin_nontemporal_load((const f32x4*)(xr + 8 * i));
                v[r][j][1] = (POLICY & 1) ? *(const f32x4*)(xr + 8 * i + 4) : __builtin_nontemporal_load((const f32x4*)(xr + 8 * i + 4));
                s[r] += ((v[r][j][0].x + v[r][j][0].y) + (v[r][j][0].z + v[r][j][0].w)) +
                        ((v[r][j][1].x + v[r][j][1].y) + (v[r][j][1].z + v[r][j][1].w));
            }
        }
    }
    float mean[LN_ROWS], rstd[LN_ROWS];
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        mean[r] = wave_sum(s[r]) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXJ8; ++j) {
            if (lane + 64 * j < C8) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float a0 = v[r][j][h].x - mean[r], a1 = v[r][j][h].y - mean[r], a2 = v[r][j][h].z - mean[r],
                                a3 = v[r][j][h].w - mean[r];
                    q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                }
            }
        }
        rstd[r] = rsqrtf(wave_sum(q) / (float)C + eps);
    }
#pragma unroll
    for (int j = 0; j < LN_MAXJ8; ++j) {
        const int i = lane + 64 * j;
        if (i < C8) {
            const float4 w0 = *(const float4*)(w + 8 * i), w1 = *(const float4*)(w + 8 * i + 4);
            const float4 b0 = *(const float4*)(b + 8 * i), b1 = *(const float4*)(b + 8 * i + 4);
#pragma unroll
            for (int r = 0; r < LN_ROWS; ++r) {
                if (row0 + r >= rows) continue;
                const float m = mean[r], rs = rstd[r];
                uint4 o;
                o.x = pack2bf((v[r][j][0].x - m) * rs * w0.x + b0.x, (v[r][j][0].y - m) * rs * w0.y + b0.y);
                o.y = pack2bf((v[r][j][0].z - m) * rs * w0.z + b0.z, (v[r][j][0].w - m) * rs * w0.w + b0.w);
                o.z = pack2bf((v[r][j][1].x - m) * rs * w1.x + b1.x, (v[r][j][1].y - m) * rs * w1.y + b1.y);
                o.w = pack2bf((v[r][j][1].z - m) * rs * w1.z + b1.z, (v[r][j][1].w - m) * rs * w1.w + b1.w);
                *(uint4*)(out + (row0 + r) * ldo + 8 * i) = o;
            }
        }
    }
}

// Final LayerNorm of the patch tokens of `slices` slices, written as
//   feats_cl  fp16 [slice][p][C]                 (row-major, for the head)
//   feats_f16 fp16 [C][d_total][npatch] at depth d0+slice   (the reference's `dino_features` layout)
// A workgroup owns 64 consecutive patch tokens of one slice; the fp16 transposition goes through an LDS tile
// [256 channels][64 tokens] so global stores are 128-B runs along the token axis.
constexpr int FN_TOK = 64, FN_CH = 256, FN_PITCH = FN_TOK + 8;  // pitch in halfwords (144 B, 16-B aligned)

// One expression for all three outputs (fp32 tokens, fp16 channels-last, fp16 [C][D][h][w]): the explicit fmaf pins the
// contraction, so the two fp16 copies hold bit-identical values (the head gives the same result from the file as in HBM).
__device__ __forceinline__ float ln_out(float x, float mean, float rstd, float w, float b) { return fmaf((x - mean) * rstd, w, b); }

template <bool HL>  // HL: x = (xh, xl) bf16 pair (xl passed behind x), ldx in elements of either array
__global__ __launch_bounds__(256) void k_final_norm(const void* __restrict__ x, const void* __restrict__ xlo, long ldx, const float* __restrict__ w,
                                                    const float* __restrict__ b, float eps, int ntp, int tok0, int npatch,
                                                    int C, _Float16* __restrict__ f16, long d_total, long d0,
                                                    uint16_t* __restrict__ cl, float* __restrict__ f32) {
    __shared__ __attribute__((aligned(16))) _Float16 tile[FN_CH * FN_PITCH];
    __shared__ float s_mean[FN_TOK], s_rstd[FN_TOK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = blockIdx.y, p0 = blockIdx.x * FN_TOK;
    const int ntile = min(FN_TOK, npatch - p0);
    const int C4 = C >> 2;
    const long row0 = (long)slice * ntp + tok0 + p0;
    auto row_of = [&](int t) {
        if constexpr (HL) return RowHL{(const uint16_t*)x + (row0 + t) * ldx, (const uint16_t*)xlo + (row0 + t) * ldx};
        else return RowF32{(const float*)x + (row0 + t) * ldx};
    };

    // phase 1: statistics (and the channels-last fp16 copy) -- one wave per token, 16 tokens per wave
    for (int t = wave; t < ntile; t += 4) {
        float4 v[LN_MAXJ];
        float mean, rstd;
        ln_row_stats(row_of(t), C4, lane, v, mean, rstd, C, eps);
        if (lane == 0) { s_mean[t] = mean; s_rstd[t] = rstd; }
        if (f32) {  // x_norm_patchtokens [slice][p][C] in fp32 (the encoder-protocol output)
            float* orow = f32 + ((long)slice * npatch + p0 + t) * C;
#pragma unroll
            for (int j = 0; j < LN_MAXJ; ++j) {
                const int i = lane + 64 * j;
                if (i < C4) {
                    const float4 ww = *(const float4*)(w + 4 * i), bb = *(const float4*)(b + 4 * i);
                    float4 o;
                    o.x = ln_out(v[j].x, mean, rstd, ww.x, bb.x); o.y = ln_out(v[j].y, mean, rstd, ww.y, bb.y);
                    o.z = ln_out(v[j].z, mean, rstd, ww.z, bb.z); o.w = ln_out(v[j].w, mean, rstd, ww.w, bb.w);
                    *(float4*)(orow + 4 * i) = o;
                }
            }
        }
        if (cl) {
            uint16_t* orow = cl + ((long)slice * npatch + p0 + t) * C;
#pragma unroll
            for (int j = 0; j < LN_MAXJ; ++j) {
                const int i = lane + 64 * j;
                if (i < C4) {
                    const float4 ww = *(const float4*)(w + 4 * i), bb = *(const float4*)(b + 4 * i);
                    uint2 o;
                    o.x = pack2h(ln_out(v[j].x, mean, rstd, ww.x, bb.x), ln_out(v[j].y, mean, rstd, ww.y, bb.y));
                    o.y = pack2h(ln_out(v[j].z, mean, rstd, ww.z, bb.z), ln_out(v[j].w, mean, rstd, ww.w, bb.w));
                    *(uint2*)(orow + 4 * i) = o;
                }
            }
        }
    }
    if (!f16) return;
    __syncthreads();

    // phase 2: fp16 [C][depth][token] through the LDS transpose tile, 256 channels at a time
    for (int c0 = 0; c0 < C; c0 += FN_CH) {
        const int nch = min(FN_CH, C - c0);
        // thread -> channel (tid), loop tokens: global reads are 256 consecutive floats per token (coalesced)
        if (tid < nch) {
            const float ww = w[c0 + tid], bb = b[c0 + tid];
            for (int t = 0; t < ntile; ++t) {
                const float val = ln_out(row_of(t).ld1(c0 + tid), s_mean[t], s_rstd[t], ww, bb);
                tile[tid * FN_PITCH + t] = __builtin_bit_cast(_Float16, f2h(val));
            }
        }
        __syncthreads();
        // thread -> (channel row, 16-B piece of 8 tokens)
        for (int idx = tid; idx < nch * (FN_TOK / 8); idx += 256) {
            const int c = idx >> 3, piece = idx & 7;
            _Float16* dst = f16 + ((long)(c0 + c) * d_total + d0 + slice) * npatch + p0 + piece * 8;
            const _Float16* src = tile + c * FN_PITCH + piece * 8;
            if (piece * 8 + 8 <= ntile && (((uintptr_t)dst) & 15) == 0) {
                *(uint4*)dst = *(const uint4*)src;
            } else {
                for (int e = 0; e < 8; ++e)
                    if (piece * 8 + e < ntile) dst[e] = src[e];
            }
        }
        __syncthreads();
    }
}

// fp16 [C][nvox] -> fp16 [nvox][C]   (HDF5 `dino_features` -> head input: a pure transpose, exact), 64x64 LDS tiles
__global__ __launch_bounds__(256) void k_f16_to_cl(const _Float16* __restrict__ in, uint16_t* __restrict__ out, int C,
                                                   long nvox) {
    __shared__ float tile[64][65];
    const long v0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i;
        const long v = v0 + tx;
        tile[i][tx] = (c < C && v < nvox) ? (float)in[(long)c * nvox + v] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const long v = v0 + i;
        const int c = c0 + tx;
        if (v < nvox && c < C) out[v * C + c] = f2h(tile[tx][i]);
    }
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm over x[nvox][C] fp16 (channels-last), G groups of C/G adjacent channels.
// ---------------------------------------------------------------------------------------------------
// Statistics are reduced WITHOUT atomics so that the result does not depend on scheduling: lane-private sums -> LDS, summed
// per channel in thread order -> per-group block partials in global memory -> k_gn_finalize adds the blocks in a fixed order.
__global__ __launch_bounds__(256) void k_gn_stats(const uint16_t* __restrict__ x, float* __restrict__ partials, long nvox, int C,
                                                  int G, long vox_per_block) {
    extern __shared__ float red[];  // [2*C] channel sums | channel sums of squares
    __shared__ float part[256][17]; // one row per thread: s[8] | q[8]
    const int tid = threadIdx.x;
    const int cpt = C >> 3;         // 16-B chunks per voxel (<= 256); cpt * vstride threads are active
    const int cc = tid % cpt, vstride = 256 / cpt;
    const int vsub = tid / cpt < vstride ? tid / cpt : -1;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    const long v0 = (long)blockIdx.x * vox_per_block, v1 = min(nvox, v0 + vox_per_block);
    auto add = [&](const uint4& u) {
        const uint32_t wds[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = hlo(wds[e]), b = hhi(wds[e]);
            s[2 * e] += a; q[2 * e] += a * a;
            s[2 * e + 1] += b; q[2 * e + 1] += b * b;
        }
    };
    // four 16-B loads in flight per thread (a block covers 64+ voxels: enough blocks to fill the chip even at C = 1024);
    // the order in which a thread adds its voxels is the same with and without the unrolling
    const uint16_t* xp = x + cc * 8;
    long v = vsub < 0 ? v1 : v0 + vsub;
    for (; v + 3 * vstride < v1; v += 4 * vstride) {
        const uint4 u0 = ld_stream16(xp + v * C), u1 = ld_stream16(xp + (v + vstride) * C);
        const uint4 u2 = ld_stream16(xp + (v + 2 * vstride) * C), u3 = ld_stream16(xp + (v + 3 * vstride) * C);
        add(u0); add(u1); add(u2); add(u3);
    }
    for (; v < v1; v += vstride) add(ld_stream16(xp + v * C));
#pragma unroll
    for (int e = 0; e < 8; ++e) { part[tid][e] = s[e]; part[tid][8 + e] = q[e]; }
    __syncthreads();
    for (int i = tid; i < 2 * C; i += 256) {  // channel c (i < C: sum, else: sum of squares) over the vstride voxel lanes, in order
        const int c = i < C ? i : i - C, col = (c & 7) + (i < C ? 0 : 8);
        float acc = 0.f;
        for (int vs = 0; vs < vstride; ++vs) acc += part[vs * cpt + (c >> 3)][col];
        red[i] = acc;
    }
    __syncthreads();
    const int cpg = C / G;
    for (int g = tid; g < 2 * G; g += 256) {
        const float* src = red + (g < G ? g * cpg : C + (g - G) * cpg);
        float acc = 0.f;
        for (int e = 0; e < cpg; ++e) acc += src[e];
        partials[(long)blockIdx.x * 2 * G + g] = acc;
    }
}

// One 64-lane group per GROUP: stats[g] / stats[G + g] = the block partials added in a fixed order (double accumulation), then
// the per-channel affine form of the normalisation, coef[c] = rstd * w[c], coef[C + c] = b[c] - mean * coef[c]  (fp64 statistics,
// one rounding to fp32 each: the arithmetic k_gn_apply used to repeat in every block)
__global__ __launch_bounds__(64) void k_gn_finalize(const float* __restrict__ partials, int nblk, int G, float* __restrict__ stats,
                                                    const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ coef,
                                                    long nvox, int C, float eps) {
    const int g = blockIdx.x, lane = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int k = lane; k < nblk; k += 64) {
        s += (double)partials[(long)k * 2 * G + g];
        q += (double)partials[(long)k * 2 * G + G + g];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    s = (double)(float)s; q = (double)(float)q;  // (the statistics are published, and used, as fp32 sums)
    if (lane == 0) { stats[g] = (float)s; stats[G + g] = (float)q; }
    const int cpg = C / G;
    const double cnt = (double)nvox * cpg;
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int e = lane; e < cpg; e += 64) {
        const int c = g * cpg + e;
        const float sc = rstd * w[c];
        coef[c] = sc;
        coef[C + c] = b[c] - (float)mean * sc;
    }
}

// out rows may be wider than C (`ldo`: the result lands in a column block of a wider channels-last buffer -- the concatenated input
// of a UNet3D synthesis block, written in place instead of through a copy kernel); out2 (nullable): a second, dense copy
template <int ACT>
__global__ __launch_bounds__(256) void k_gn_apply(const uint16_t* __restrict__ x, const float* __restrict__ coef,
                                                  uint16_t* __restrict__ out, long ldo, uint16_t* __restrict__ out2, long nvox, int C,
                                                  long vox_per_block) {
    const int tid = threadIdx.x;
    const int cpt = C >> 3;
    const int cc = tid % cpt, vsub = tid / cpt, vstride = 256 / cpt;
    if (vsub >= vstride) return;
    float sc[8], sh[8];
    {
        const float4 a0 = *(const float4*)(coef + cc * 8), a1 = *(const float4*)(coef + cc * 8 + 4);
        const float4 b0 = *(const float4*)(coef + C + cc * 8), b1 = *(const float4*)(coef + C + cc * 8 + 4);
        sc[0] = a0.x; sc[1] = a0.y; sc[2] = a0.z; sc[3] = a0.w; sc[4] = a1.x; sc[5] = a1.y; sc[6] = a1.z; sc[7] = a1.w;
        sh[0] = b0.x; sh[1] = b0.y; sh[2] = b0.z; sh[3] = b0.w; sh[4] = b1.x; sh[5] = b1.y; sh[6] = b1.z; sh[7] = b1.w;
    }
    auto fn = [&](float v, int e) {
        const float t = fmaf(v, sc[e], sh[e]);
        return ACT == 1 ? gelu_erf(t) : t;  // (UNet3D: InstanceNorm + GELU in one pass)
    };
    auto norm = [&](const uint4& u) {
        uint4 o;
        o.x = pack2h(fn(hlo(u.x), 0), fn(hhi(u.x), 1));
        o.y = pack2h(fn(hlo(u.y), 2), fn(hhi(u.y), 3));
        o.z = pack2h(fn(hlo(u.z), 4), fn(hhi(u.z), 5));
        o.w = pack2h(fn(hlo(u.w), 6), fn(hhi(u.w), 7));
        return o;
    };
    const long v0 = (long)blockIdx.x * vox_per_block, v1 = min(nvox, v0 + vox_per_block);
    const long co = cc * 8;
    long v = v0 + vsub;
    for (; v + 3 * vstride < v1; v += 4 * vstride) {
        const uint4 u0 = ld_stream16(x + v * C + co), u1 = ld_stream16(x + (v + vstride) * C + co);
        const uint4 u2 = ld_stream16(x + (v + 2 * vstride) * C + co), u3 = ld_stream16(x + (v + 3 * vstride) * C + co);
        const uint4 n0 = norm(u0), n1 = norm(u1), n2 = norm(u2), n3 = norm(u3);
        *(uint4*)(out + v * ldo + co) = n0;
        *(uint4*)(out + (v + vstride) * ldo + co) = n1;
        *(uint4*)(out + (v + 2 * vstride) * ldo + co) = n2;
        *(uint4*)(out + (v + 3 * vstride) * ldo + co) = n3;
        if (out2) {
            *(uint4*)(out2 + v * C + co) = n0;
            *(uint4*)(out2 + (v + vstride) * C + co) = n1;
            *(uint4*)(out2 + (v + 2 * vstride) * C + co) = n2;
            *(uint4*)(out2 + (v + 3 * vstride) * C + co) = n3;
        }
    }
    for (; v < v1; v += vstride) {
        const uint4 n0 = norm(ld_stream16(x + v * C + co));
        *(uint4*)(out + v * ldo + co) = n0;
        if (out2) *(uint4*)(out2 + v * C + co) = n0;
    }
}

// ---------------------------------------------------------------------------------------------------
// The ViT's residual stream as a bf16 (hi, lo) pair with the LayerNorm folded into the consuming GEMMs.
// ---------------------------------------------------------------------------------------------------
// fp32 rows -> hi = bf16(x), lo = bf16(x - hi) and the row constants (rstd, -mean * rstd) of the FIRST LayerNorm: once per slice
// batch, behind the patch-embedding GEMM (every later split and every later statistic comes out of a residual GEMM's epilogue).
// One wave per row, 8 consecutive channels per lane and step (16-B stores); two-pass variance on the registers.
__global__ __launch_bounds__(256) void k_split_stream(const float* __restrict__ x, long ldx, uint16_t* __restrict__ xh, uint16_t* __restrict__ xl,
                                                      long ld, float* __restrict__ rowstat, long rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int C8 = C >> 3;
    const float* xr = x + row * ldx;
    f32x4 v[LN_MAXJ8][2];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ8; ++j) {
        const int i = lane + 64 * j;
        if (i < C8) {
            v[j][0] = *(const f32x4*)(xr + 8 * i);
            v[j][1] = *(const f32x4*)(xr + 8 * i + 4);
            s += ((v[j][0].x + v[j][0].y) + (v[j][0].z + v[j][0].w)) + ((v[j][1].x + v[j][1].y) + (v[j][1].z + v[j][1].w));
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ8; ++j) {
        const int i = lane + 64 * j;
        if (i < C8) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float a0 = v[j][h].x - mean, a1 = v[j][h].y - mean, a2 = v[j][h].z - mean, a3 = v[j][h].w - mean;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
            uint4 oh, ol;
            uint32_t* ph = &oh.x;
            uint32_t* pl = &ol.x;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float x0 = v[j][h][2 * k], x1 = v[j][h][2 * k + 1];
                    const uint32_t hw = pack2bf(x0, x1);
                    ph[2 * h + k] = hw;
                    pl[2 * h + k] = pack2bf(x0 - bflo(hw), x1 - bfhi(hw));
                }
            *(uint4*)(xh + row * ld + 8 * i) = oh;
            *(uint4*)(xl + row * ld + 8 * i) = ol;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) *(float2*)(rowstat + 2 * row) = float2{rstd, -mean * rstd};
}

// (hi, lo) -> fp32 rows (x = hi + lo, exact): where a folded stretch of the stream hands over to kernels that take fp32 (Hiera's
// stage transitions)
__global__ __launch_bounds__(256) void k_merge_stream(const uint16_t* __restrict__ xh, const uint16_t* __restrict__ xl, long ld, float* __restrict__ x,
                                                      long ldx, long rows, int C8) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C8) return;
    const long row = idx / C8;
    const int c = (int)(idx - row * C8) * 8;
    const uint4 a = *(const uint4*)(xh + row * ld + c), b = *(const uint4*)(xl + row * ld + c);
    float* o = x + row * ldx + c;
    *(float4*)o = float4{bflo(a.x) + bflo(b.x), bfhi(a.x) + bfhi(b.x), bflo(a.y) + bflo(b.y), bfhi(a.y) + bfhi(b.y)};
    *(float4*)(o + 4) = float4{bflo(a.z) + bflo(b.z), bfhi(a.z) + bfhi(b.z), bflo(a.w) + bflo(b.w), bfhi(a.w) + bfhi(b.w)};
}

// part[slot][part_rows][2] (sum, sum of squares of a row over 64 columns, written by the hi/lo residual epilogue) ->
// rowstat[row] = (rstd, -mean * rstd).  Fixed order, double accumulation: 24 partials of fp32 sums lose nothing further.
__global__ __launch_bounds__(64) void k_rowstat_finalize(const float* __restrict__ part, int nslot, long part_rows, float* __restrict__ rowstat,
                                                          long rows, float inv_c, float eps) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (64-thread blocks: 2064 of them at 128 slices -- 8 per CU)
    if (row >= rows) return;
    double s = 0.0, q = 0.0;
    // eight independent loads in flight per thread (a dependent load -> add chain of 24 round trips made this 13 us per launch,
    // 80 times per slice batch); the summation order stays the slot order
    int k = 0;
    for (; k + 8 <= nslot; k += 8) {
        float2 p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = *(const float2*)(part + ((long)(k + u) * part_rows + row) * 2);
#pragma unroll
        for (int u = 0; u < 8; ++u) { s += (double)p[u].x; q += (double)p[u].y; }
    }
    for (; k < nslot; ++k) {
        const float2 p = *(const float2*)(part + ((long)k * part_rows + row) * 2);
        s += (double)p.x;
        q += (double)p.y;
    }
    const double mean = s * (double)inv_c;
    double var = q * (double)inv_c - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    *(float2*)(rowstat + 2 * row) = float2{rstd, -(float)mean * rstd};
}

}  // namespace cvx

using namespace cvx;

std::atomic<int> g_ln_policy{3};  // cvx_set_option("ln_policy", 0..3) -- gemm.hip; 3 (cacheable loads, rows from the end) measured 17.4 -> 16.2-16.4 ms per tomogram

extern "C" int cvx_layernorm_bf16(const float* x, long ldx, const float* w, const float* b, void* out, long ldo,
                                  long rows, int C, float eps, hipStream_t st) {
    if (rows <= 0) return 0;
    if (C % 4 || C > 2048 || ldx % 4 || ldo % 4) return cvx_fail("layernorm: C%4==0, C<=2048, ld%4==0 required");
    if (C <= 192) {  // 16 lanes per row, 4 rows per wave
        hipLaunchKernelGGL((k_layernorm_bf16_rows<16, 3>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, x, ldx, w, b, (uint16_t*)out, ldo, rows,
                           C, eps);
        return cvx_check_launch();
    }
    if (C <= 384) {  // 32 lanes per row, 2 rows per wave
        hipLaunchKernelGGL((k_layernorm_bf16_rows<32, 3>), dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, st, x, ldx, w, b, (uint16_t*)out, ldo, rows,
                           C, eps);
        return cvx_check_launch();
    }
    if (C < 1024 || C % 8 || ldo % 8) {
        hipLaunchKernelGGL(k_layernorm_bf16_v4, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, ldx, w, b, (uint16_t*)out, ldo, rows, C,
                           eps);
        return cvx_check_launch();
    }
    const dim3 grid((unsigned)((rows + 4 * LN_ROWS - 1) / (4 * LN_ROWS)));
    switch (g_ln_policy.load()) {
        case 1: hipLaunchKernelGGL(k_layernorm_bf16<1>, grid, dim3(256), 0, st, x, ldx, w, b, (uint16_t*)out, ldo, rows, C, eps); break;
        case 2: hipLaunchKernelGGL(k_layernorm_bf16<2>, grid, dim3(256), 0, st, x, ldx, w, b, (uint16_t*)out, ldo, rows, C, eps); break;
        case 3: hipLaunchKernelGGL(k_layernorm_bf16<3>, grid, dim3(256), 0, st, x, ldx, w, b, (uint16_t*)out, ldo, rows, C, eps); break;
        default: hipLaunchKernelGGL(k_layernorm_bf16<0>, grid, dim3(256), 0, st, x, ldx, w, b, (uint16_t*)out, ldo, rows, C, eps); break;
    }
    return cvx_check_launch();
}

extern "C" int cvx_final_norm_features(const float* x, long ldx, const float* w, const float* b, float eps, int slices,
                                       int ntp, int tok0, int hp, int wp, int C, void* feats_f16, long d_total, long d0,
                                       void* feats_cl, float* tokens_f32, hipStream_t st) {
    if (slices <= 0) return 0;
    if (C % 4 || C > 64 * 4 * LN_MAXJ || ldx % 4) return cvx_fail("final_norm: C%4==0, C<=2048, ldx%4==0 required");
    const int npatch = hp * wp;
    dim3 grid((npatch + FN_TOK - 1) / FN_TOK, slices);
    hipLaunchKernelGGL(k_final_norm<false>, grid, dim3(256), 0, st, (const void*)x, (const void*)nullptr, ldx, w, b, eps, ntp, tok0, npatch, C,
                       (_Float16*)feats_f16, d_total, d0, (uint16_t*)feats_cl, tokens_f32);
    return cvx_check_launch();
}

extern "C" int cvx_final_norm_features_hl(const void* xh, const void* xl, long ld, const float* w, const float* b, float eps, int slices,
                                          int ntp, int tok0, int hp, int wp, int C, void* feats_f16, long d_total, long d0,
                                          void* feats_cl, float* tokens_f32, hipStream_t st) {
    if (slices <= 0) return 0;
    if (!xh || !xl) return cvx_fail("final_norm_hl: null stream");
    if (C % 4 || C > 64 * 4 * LN_MAXJ || ld % 4) return cvx_fail("final_norm_hl: C%4==0, C<=2048, ld%4==0 required");
    const int npatch = hp * wp;
    dim3 grid((npatch + FN_TOK - 1) / FN_TOK, slices);
    hipLaunchKernelGGL(k_final_norm<true>, grid, dim3(256), 0, st, xh, xl, ld, w, b, eps, ntp, tok0, npatch, C, (_Float16*)feats_f16,
                       d_total, d0, (uint16_t*)feats_cl, tokens_f32);
    return cvx_check_launch();
}

extern "C" int cvx_split_stream(const float* x, long ldx, void* xh, void* xl, long ld, float* rowstat, long rows, int C, float eps,
                                hipStream_t st) {
    if (rows <= 0) return 0;
    if (!x || !xh || !xl || !rowstat) return cvx_fail("split_stream: null pointer");
    if (C % 8 || C > 64 * 8 * LN_MAXJ8 || ldx % 4 || ld % 8) return cvx_fail("split_stream: C%8==0, C<=2048, ldx%4==0, ld%8==0 required");
    hipLaunchKernelGGL(k_split_stream, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, ldx, (uint16_t*)xh, (uint16_t*)xl, ld, rowstat, rows,
                       C, eps);
    return cvx_check_launch();
}

extern "C" int cvx_merge_stream(const void* xh, const void* xl, long ld, float* x, long ldx, long rows, int C, hipStream_t st) {
    if (rows <= 0) return 0;
    if (!xh || !xl || !x) return cvx_fail("merge_stream: null pointer");
    if (C % 8 || ld % 8 || ldx % 4) return cvx_fail("merge_stream: C%8==0, ld%8==0, ldx%4==0 required");
    const long total = rows * (C / 8);
    hipLaunchKernelGGL(k_merge_stream, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const uint16_t*)xh, (const uint16_t*)xl, ld, x, ldx,
                       rows, C / 8);
    return cvx_check_launch();
}

extern "C" int cvx_rowstat_finalize(const float* part, int nslot, long part_rows, float* rowstat, long rows, int C, float eps, hipStream_t st) {
    if (rows <= 0) return 0;
    if (!part || !rowstat) return cvx_fail("rowstat_finalize: null pointer");
    if (nslot <= 0 || nslot * 64 != C || rows > part_rows) return cvx_fail("rowstat_finalize: nslot must be C / 64 and rows <= part_rows");
    hipLaunchKernelGGL(k_rowstat_finalize, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, st, part, nslot, part_rows, rowstat, rows,
                       1.0f / (float)C, eps);
    return cvx_check_launch();
}

extern "C" int cvx_features_to_channels_last(const void* feats_f16, void* out_cl, int C, long nvox, hipStream_t st) {
    if (nvox <= 0) return 0;
    dim3 grid((unsigned)((nvox + 63) / 64), (C + 63) / 64);
    hipLaunchKernelGGL(k_f16_to_cl, grid, dim3(256), 0, st, (const _Float16*)feats_f16, (uint16_t*)out_cl, C, nvox);
    return cvx_check_launch();
}

static int groupnorm_impl(const void* x, const float* w, const float* b, void* out, float* stats, long nvox, int C, int G, float eps,
                          int act, hipStream_t st, long ldo = 0, void* out2 = nullptr) {
    if (nvox <= 0) return 0;
    if (ldo == 0) ldo = C;
    if (ldo < C || ldo % 8) return cvx_fail("groupnorm: the output leading dimension must be >= C and a multiple of 8");
    const int cpt = C / 8;
    if (C % 8 || cpt > 256 || C % G || G > CVX_GN_MAX_GROUPS)
        return cvx_fail("groupnorm: C must be a multiple of 8, <= 2048, divisible by G, G <= 512");
    // stats[0 .. 2G) = the sums; then per-block partials of nstat blocks; the per-channel coefficients (2C floats = C/G more
    // "blocks") behind them: 2G * (1 + CVX_GN_BLOCKS) floats in all
    const int vstride = 256 / cpt;
    long vpb_stat = std::max<long>(64, 8L * vstride);           // >= 8 voxels per thread
    long nstat = (nvox + vpb_stat - 1) / vpb_stat;
    if (C / G >= CVX_GN_BLOCKS / 2) return cvx_fail("groupnorm: C / G must be < CVX_GN_BLOCKS / 2 (the 2C coefficients sit behind the block partials)");
    const long cap = CVX_GN_BLOCKS - C / G;
    if (nstat > cap) { nstat = cap; vpb_stat = (nvox + nstat - 1) / nstat; nstat = (nvox + vpb_stat - 1) / vpb_stat; }
    float* partials = stats + 2 * G;
    float* coef = partials + nstat * 2 * G;
    hipLaunchKernelGGL(k_gn_stats, dim3((unsigned)nstat), dim3(256), sizeof(float) * 2 * C, st, (const uint16_t*)x, partials, nvox, C, G,
                       vpb_stat);
    int rc = cvx_check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(k_gn_finalize, dim3(G), dim3(64), 0, st, partials, (int)nstat, G, stats, w, b, coef, nvox, C, eps);
    rc = cvx_check_launch();
    if (rc) return rc;
    const long vox_per_block = 8L * vstride;                     // 8 voxels per thread, four 16-B loads in flight
    const unsigned nblk = (unsigned)((nvox + vox_per_block - 1) / vox_per_block);
    if (act) hipLaunchKernelGGL(k_gn_apply<1>, dim3(nblk), dim3(256), 0, st, (const uint16_t*)x, coef, (uint16_t*)out, ldo, (uint16_t*)out2, nvox, C, vox_per_block);
    else hipLaunchKernelGGL(k_gn_apply<0>, dim3(nblk), dim3(256), 0, st, (const uint16_t*)x, coef, (uint16_t*)out, ldo, (uint16_t*)out2, nvox, C, vox_per_block);
    return cvx_check_launch();
}

extern "C" int cvx_groupnorm_f16(const void* x, const float* w, const float* b, void* out, float* stats, long nvox, int C,
                                  int G, float eps, hipStream_t st) {
    return groupnorm_impl(x, w, b, out, stats, nvox, C, G, eps, 0, st);
}

extern "C" int cvx_groupnorm_act_f16(const void* x, const float* w, const float* b, void* out, float* stats, long nvox, int C,
                                      int G, float eps, int act, hipStream_t st) {
    if (act != 0 && act != 1) return cvx_fail("groupnorm_act: act is 0 (none) or 1 (GELU)");
    return groupnorm_impl(x, w, b, out, stats, nvox, C, G, eps, act, st);
}

extern "C" int cvx_groupnorm_act_strided_f16(const void* x, const float* w, const float* b, void* out, long ldo, void* out2_dense, float* stats,
                                              long nvox, int C, int G, float eps, int act, hipStream_t st) {
    if (act != 0 && act != 1) return cvx_fail("groupnorm_act_strided: act is 0 (none) or 1 (GELU)");
    if (!out) return cvx_fail("groupnorm_act_strided: null output");
    return groupnorm_impl(x, w, b, out, stats, nvox, C, G, eps, act, st, ldo, out2_dense);
}
