// Dense / implicit-GEMM launchers and fused epilogues (gfx950).  See gemm_core.h for the tile kernel.
#include "gemm_core.h"
#include "gemm256.h"
#include "gemm256p.h"
#ifdef CVX_ABLATION
#include "gemm4w.h"  // the 4-wave tile: measured 3 % slower, kept for A/B runs only
#endif
#include "../../include/cryovit_hip.h"
#include "host_util.h"
#include <atomic>
#include <initializer_list>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>

namespace cvx {

// ------------------------------------------------------------------------------------------------
// Epilogues.  NREG orientation: store<NV>(n0, m, v) -- v[i] is output feature n0+i of row m.
//             MREG orientation: store<NV>(m0, n, v) -- v[i] is row m0+i of output feature n.
// ------------------------------------------------------------------------------------------------
template <int NV, bool F16 = false>
__device__ __forceinline__ void store_bf16_chunked(uint16_t* dst, const float* v, long n0, long n_valid) {
    if constexpr (NV >= 8) {
#pragma unroll
        for (int h = 0; h < NV / 8; ++h) {
            if (n0 + h * 8 < n_valid) {
                uint4 w;
                w.x = pack2x<F16>(v[h * 8 + 0], v[h * 8 + 1]);
                w.y = pack2x<F16>(v[h * 8 + 2], v[h * 8 + 3]);
                w.z = pack2x<F16>(v[h * 8 + 4], v[h * 8 + 5]);
                w.w = pack2x<F16>(v[h * 8 + 6], v[h * 8 + 7]);
                *(uint4*)(dst + h * 8) = w;
            }
        }
    } else {
        static_assert(NV == 4, "NV must be 4 or a multiple of 8");
        if (n0 < n_valid) {
            uint2 w;
            w.x = pack2x<F16>(v[0], v[1]);
            w.y = pack2x<F16>(v[2], v[3]);
            *(uint2*)dst = w;
        }
    }
}

template <int NV>
struct VecCtx {
    float bias[NV];
    float gamma[NV];
};
template <int NV>
__device__ __forceinline__ void load_vec(float* dst, const float* src) {
#pragma unroll
    for (int h = 0; h < NV / 4; ++h) {
        const float4 t = *(const float4*)(src + h * 4);
        dst[h * 4 + 0] = t.x; dst[h * 4 + 1] = t.y; dst[h * 4 + 2] = t.z; dst[h * 4 + 3] = t.w;
    }
}

// LayerNorm folded into a consuming GEMM (LN = true; DESIGN.md s.4 "LayerNorm fold"): the A operand is bf16(x) itself (the `hi`
// half of the residual stream), the LayerNorm gain is folded into the weights at packing (W' = bf16(W * gamma_ln)), and the
// normalisation is applied to the ACCUMULATOR:  y[m][n] = rstd[m] * acc - mu[m] * rstd[m] * cs[n] + b'[n]  with
// cs[n] = sum_c W'[n][c] and b'[n] = b[n] + sum_c W[n][c] * beta_ln[c].  `bias` then points at [2][cs_off] floats (b' | cs) and
// `rowstat` at fp32 [rows][2] = (rstd, -mu * rstd), written by cvx_rowstat_finalize / cvx_split_stream.
__device__ __forceinline__ float ln_fold(float acc, float2 rs, float cs, float b) { return fmaf(acc, rs.x, fmaf(rs.y, cs, b)); }

// out[m][n] = bf16 | fp16 (act(acc + bias[n]));  ACT: 0 none, 1 GELU(erf);  HALF: fp16 operands and output (the head)
template <int ACT, bool HALF = false, bool LN = false>
struct EpiBF16 {
    static constexpr bool F16 = HALF;
    static constexpr bool LNFOLD = LN;
    uint16_t* out; long ldc; const float* bias; long m_valid, n_valid;
    const float* rowstat = nullptr; long cs_off = 0;  // LN only
    template <int NV> using Ctx = VecCtx<NV>;
    template <int NV>
    __device__ __forceinline__ void prep(Ctx<NV>& c, long n0) const {
        load_vec<NV>(c.bias, bias + n0);
        if constexpr (LN) load_vec<NV>(c.gamma, bias + cs_off + n0);  // (the gamma slot carries the column sums)
    }
    template <int NV>
    __device__ __forceinline__ void store(const Ctx<NV>& c, long n0, long m, const float* acc) const {
        if (m >= m_valid) return;
        float2 rs{0.f, 0.f};
        if constexpr (LN) rs = *(const float2*)(rowstat + 2 * m);
        float v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float t = LN ? ln_fold(acc[i], rs, c.gamma[i], c.bias[i]) : acc[i] + c.bias[i];
            v[i] = ACT == 1 ? gelu_erf(t) : t;
        }
        store_bf16_chunked<NV, HALF>(out + m * ldc + n0, v, n0, n_valid);
    }
    // the same epilogue for rows [m_off, ...) of the problem when A is passed advanced by m_off rows (tail launches)
    EpiBF16 shifted(long m_off) const {
        EpiBF16 e = *this;
        e.out = out + m_off * ldc; e.m_valid = m_valid - m_off;
        if (rowstat) e.rowstat = rowstat + 2 * m_off;
        return e;
    }
    // LDS-staged row-major store (gemm256.h): 16 accumulators -> OUT16 packed bf16 outputs of column n0 >> OUT_SHIFT
    static constexpr int OUT16 = 16, OUT_SHIFT = 0;
    __device__ __forceinline__ void produce(const Ctx<16>& c, const float* acc, uint32_t (&w)[8], float2 rs = float2{0.f, 0.f}) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float a = LN ? ln_fold(acc[2 * i], rs, c.gamma[2 * i], c.bias[2 * i]) : acc[2 * i] + c.bias[2 * i];
            const float b = LN ? ln_fold(acc[2 * i + 1], rs, c.gamma[2 * i + 1], c.bias[2 * i + 1]) : acc[2 * i + 1] + c.bias[2 * i + 1];
            w[i] = pack2x<HALF>(ACT == 1 ? gelu_erf(a) : a, ACT == 1 ? gelu_erf(b) : b);
        }
    }
};

// SwiGLU gate: packed W12 rows are interleaved in blocks of 8 (a[8j..8j+7], b[8j..8j+7]) so a lane's 16
// contiguous accumulators are 8 a's and the 8 matching b's:  out[m][n0/2 + i] = silu(a_i) * b_i
template <bool LN = false>
struct EpiSwiGLUT {
    static constexpr bool LNFOLD = LN;
    uint16_t* out; long ldc; const float* bias; long m_valid, n_valid;
    const float* rowstat = nullptr; long cs_off = 0;  // LN only (see EpiBF16)
    template <int NV> using Ctx = VecCtx<NV>;
    template <int NV>
    __device__ __forceinline__ void prep(Ctx<NV>& c, long n0) const {
        load_vec<NV>(c.bias, bias + n0);
        if constexpr (LN) load_vec<NV>(c.gamma, bias + cs_off + n0);
    }
    template <int NV>
    __device__ __forceinline__ void store(const Ctx<NV>& c, long n0, long m, const float* acc) const {
        static_assert(NV == 16, "SwiGLU epilogue needs 16 contiguous features per lane");
        if (m >= m_valid || n0 >= n_valid) return;
        float2 rs{0.f, 0.f};
        if constexpr (LN) rs = *(const float2*)(rowstat + 2 * m);
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float a = LN ? ln_fold(acc[i], rs, c.gamma[i], c.bias[i]) : acc[i] + c.bias[i];
            const float b = LN ? ln_fold(acc[8 + i], rs, c.gamma[8 + i], c.bias[8 + i]) : acc[8 + i] + c.bias[8 + i];
            v[i] = silu(a) * b;
        }
        store_bf16_chunked<8>(out + m * ldc + (n0 >> 1), v, 0, 1);
    }
    EpiSwiGLUT shifted(long m_off) const {
        EpiSwiGLUT e = *this;
        e.out = out + m_off * ldc; e.m_valid = m_valid - m_off;
        if (rowstat) e.rowstat = rowstat + 2 * m_off;
        return e;
    }
    static constexpr int OUT16 = 8, OUT_SHIFT = 1;  // 8 gated outputs per 16 accumulators, output column = n0 / 2
    // Written on register-adjacent PAIRS (accumulator elements 2i, 2i+1 of one fragment): packed adds / multiplies, two exp2,
    // two rcp and one packed convert per two outputs.  Left to the SLP vectoriser the same arithmetic came out with the pairs
    // crossed (a_i with b_i): 10 instructions per output, a quarter of them v_mov / v_or shuffles.
    __device__ __forceinline__ void produce(const Ctx<16>& c, const float* acc, uint32_t (&w)[4], float2 rs = float2{0.f, 0.f}) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x2 a, b;
            if constexpr (LN) {
                a = f32x2{acc[2 * i], acc[2 * i + 1]} * rs.x + (f32x2{c.gamma[2 * i], c.gamma[2 * i + 1]} * rs.y + f32x2{c.bias[2 * i], c.bias[2 * i + 1]});
                b = f32x2{acc[8 + 2 * i], acc[9 + 2 * i]} * rs.x + (f32x2{c.gamma[8 + 2 * i], c.gamma[9 + 2 * i]} * rs.y + f32x2{c.bias[8 + 2 * i], c.bias[9 + 2 * i]});
            } else {
                a = f32x2{acc[2 * i], acc[2 * i + 1]} + f32x2{c.bias[2 * i], c.bias[2 * i + 1]};
                b = f32x2{acc[8 + 2 * i], acc[9 + 2 * i]} + f32x2{c.bias[8 + 2 * i], c.bias[9 + 2 * i]};
            }
            const f32x2 t = a * -1.4426950408889634f;
            const f32x2 u = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.0f;
            const f32x2 r = f32x2{__builtin_amdgcn_rcpf(u[0]), __builtin_amdgcn_rcpf(u[1])};
            const f32x2 o = (a * b) * r;
            w[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(o, bf16x2));
        }
    }
};
using EpiSwiGLU = EpiSwiGLUT<false>;

// residual stream update in fp32:  x[m][n] += gamma[n] * (acc + bias[n])
// x[m][n] (+)= gamma[n] * (acc + bias[n]) on the fp32 stream.  ACC = true: read-modify-write (LayerScale + residual);
// ACC = false: plain fp32 output (x is only written -- Hiera's projected shortcut, the FPN lateral convs)
template <bool ACC>
struct EpiResidT {
    static constexpr bool ACCUM = ACC;
    float* x; long ldx; const float* bias; const float* gamma; long m_valid, n_valid;
#ifdef CVX_LN_EMIT_PROTO
    uint16_t* emit_xb = nullptr; long emit_ldb = 0;  // timing prototype (tools/bench_ln_emit.py): bf16 copy of the updated x
    float* emit_part = nullptr; long emit_rows = 0;  // ... and row partial sums P[4 * N tiles][emit_rows][2]
#endif
    template <int NV> using Ctx = VecCtx<NV>;
    template <int NV>
    __device__ __forceinline__ void prep(Ctx<NV>& c, long n0) const {
        load_vec<NV>(c.bias, bias + n0);
        load_vec<NV>(c.gamma, gamma + n0);
    }
    // the residual values are PRE-LOADED for a batch of output columns before any of them is stored: issued one after
    // the other, each load -> add -> store round trip exposed a full HBM latency (32 per lane per tile: the epilogue
    // took as long as the whole K loop of the proj GEMM -- tools/stamp_gemm_coarse.py)
    EpiResidT shifted(long m_off) const {
        EpiResidT r = *this;
        r.x = x + m_off * ldx;
        r.m_valid = m_valid - m_off;
#ifdef CVX_LN_EMIT_PROTO
        if (r.emit_xb) r.emit_xb = emit_xb + m_off * emit_ldb;
        if (r.emit_part) r.emit_part = emit_part + m_off * 2;
#endif
        return r;
    }
    static constexpr bool HAS_PRELOAD = true;
    template <int NV> struct Pre { float4 x[NV / 4]; };
    template <int NV>
    __device__ __forceinline__ void preload(Pre<NV>& p, long n0, long m) const {
        const long mm = m < m_valid ? m : 0;
#pragma unroll
        for (int h = 0; h < NV / 4; ++h)
            p.x[h] = ACC ? *(const float4*)(x + mm * ldx + (n0 + h * 4 < n_valid ? n0 + h * 4 : 0)) : float4{0.f, 0.f, 0.f, 0.f};
    }
    template <int NV>
    __device__ __forceinline__ void store(const Ctx<NV>& c, const Pre<NV>& p, long n0, long m, const float* acc) const {
        if (m >= m_valid) return;
#pragma unroll
        for (int h = 0; h < NV / 4; ++h) {
            const long n = n0 + h * 4;
            if (n >= n_valid) continue;
            float4 xv = p.x[h];
            xv.x += c.gamma[h * 4 + 0] * (acc[h * 4 + 0] + c.bias[h * 4 + 0]);
            xv.y += c.gamma[h * 4 + 1] * (acc[h * 4 + 1] + c.bias[h * 4 + 1]);
            xv.z += c.gamma[h * 4 + 2] * (acc[h * 4 + 2] + c.bias[h * 4 + 2]);
            xv.w += c.gamma[h * 4 + 3] * (acc[h * 4 + 3] + c.bias[h * 4 + 3]);
            *(float4*)(x + m * ldx + n) = xv;
        }
    }
    template <int NV>
    __device__ __forceinline__ void store(const Ctx<NV>& c, long n0, long m, const float* acc) const {
        Pre<NV> p;
        preload<NV>(p, n0, m);
        store<NV>(c, p, n0, m, acc);
    }
};
using EpiResid = EpiResidT<true>;
using EpiF32 = EpiResidT<false>;

// The residual stream as a PAIR of bf16 arrays: hi = bf16(x) is at the same time the A operand of the next
// GEMM (whose epilogue applies the folded LayerNorm), lo = bf16(x - hi) keeps 16 significant bits of the running sum
// (x is read back as hi + lo, exact in fp32).  Same bytes as one fp32 array, but the separate LayerNorm pass (read 4 B, write
// 2 B per element, 80 times per slice batch) is gone.   x += gamma * (acc + bias);  hi, lo = split(x);  and per 64-column slot
// the row sums  part[slot][m] = (sum x, sum x^2)  of the NEW x (fp32, before the split) for cvx_rowstat_finalize.
struct EpiResidHL {
    static constexpr bool HAS_HL = true;
    uint16_t* xh; uint16_t* xl; long ldx; const float* bias; const float* gamma; float* part; long part_rows; long m_valid, n_valid;
    template <int NV> using Ctx = VecCtx<NV>;
    template <int NV>
    __device__ __forceinline__ void prep(Ctx<NV>& c, long n0) const {
        load_vec<NV>(c.bias, bias + n0);
        load_vec<NV>(c.gamma, gamma + n0);
    }
    EpiResidHL shifted(long m_off) const {
        EpiResidHL r = *this;
        r.xh = xh + m_off * ldx; r.xl = xl + m_off * ldx; r.part = part + m_off * 2; r.m_valid = m_valid - m_off;
        return r;
    }
    // accumulator-layout form (the 128 / 64-wide tiles: tails and small problems): 16 contiguous columns n0.. of row m per lane,
    // the four lane groups of a wave cover one 64-column slot.  Every lane runs the shuffles (no early exit).
    template <int NV>
    __device__ __forceinline__ void store(const Ctx<NV>& c, long n0, long m, const float* acc) const {
        static_assert(NV == 16, "hi/lo residual epilogue: 16 contiguous columns per lane (64-column slots)");
        const bool ok = m < m_valid && n0 < n_valid;  // (n_valid % 64 == 0: the four lane groups of a slot agree)
        const long mm = ok ? m : 0;
        const uint16_t* ph = xh + mm * ldx + (ok ? n0 : 0);
        const uint16_t* pl = xl + mm * ldx + (ok ? n0 : 0);
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint4 vh = *(const uint4*)(ph + 8 * h), vl = *(const uint4*)(pl + 8 * h);
            const uint32_t wh[4] = {vh.x, vh.y, vh.z, vh.w}, wl[4] = {vl.x, vl.y, vl.z, vl.w};
            uint32_t nh[4], nl[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = 8 * h + 2 * k;
                const float x0 = (bflo(wh[k]) + bflo(wl[k])) + c.gamma[i] * (acc[i] + c.bias[i]);
                const float x1 = (bfhi(wh[k]) + bfhi(wl[k])) + c.gamma[i + 1] * (acc[i + 1] + c.bias[i + 1]);
                nh[k] = pack2bf(x0, x1);
                nl[k] = pack2bf(x0 - bflo(nh[k]), x1 - bfhi(nh[k]));
                s += x0 + x1;
                q = fmaf(x0, x0, fmaf(x1, x1, q));
            }
            if (ok) {
                *(uint4*)(xh + m * ldx + n0 + 8 * h) = uint4{nh[0], nh[1], nh[2], nh[3]};
                *(uint4*)(xl + m * ldx + n0 + 8 * h) = uint4{nl[0], nl[1], nl[2], nl[3]};
            }
        }
        s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
        s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
        if (ok && (threadIdx.x & 48) == 0) *(float2*)(part + ((n0 >> 6) * part_rows + m) * 2) = float2{s, q};
    }
};

// patch embedding: GEMM row m = slice*npatch + p  ->  token row slice*ntp + tok0 + p of the fp32 stream,
// value = acc + bias[n] + pos[(1+p)][n]
struct EpiPatch {
    float* x; long ldx; const float* bias; const float* pos; long ldpos; int npatch, ntp, tok0; long m_valid, n_valid;
    template <int NV> using Ctx = VecCtx<NV>;
    template <int NV>
    __device__ __forceinline__ void prep(Ctx<NV>& c, long n0) const { load_vec<NV>(c.bias, bias + n0); }
    template <int NV>
    __device__ __forceinline__ void store(const Ctx<NV>& c, long n0, long m, const float* acc) const {
        if (m >= m_valid) return;
        const long s = m / npatch, p = m - s * npatch;
        float* dst = x + (s * ntp + tok0 + p) * ldx;
        const float* pp = pos + (1 + p) * ldpos;
#pragma unroll
        for (int h = 0; h < NV / 4; ++h) {
            const long n = n0 + h * 4;
            if (n >= n_valid) continue;
            const float4 q = *(const float4*)(pp + n);
            float4 o;
            o.x = acc[h * 4 + 0] + c.bias[h * 4 + 0] + q.x;
            o.y = acc[h * 4 + 1] + c.bias[h * 4 + 1] + q.y;
            o.z = acc[h * 4 + 2] + c.bias[h * 4 + 2] + q.z;
            o.w = acc[h * 4 + 3] + c.bias[h * 4 + 3] + q.w;
            *(float4*)(dst + n) = o;
        }
    }
};

// V written TRANSPOSED for the attention kernel (MREG orientation: a lane owns NV consecutive tokens of one
// feature n = head*64 + d):   vt[slice][head][d][t] = acc + bias[n],   token row m = slice*ntp + t
template <bool LN = false>
struct EpiVTT {
    static constexpr bool LNFOLD = LN;
    uint16_t* vt; const float* bias; int heads, ntp, kp; long m_valid, n_valid; long m_off = 0;  // m_off: row of the problem that A row 0 is
    const float* rowstat = nullptr; long cs_off = 0;  // LN only (see EpiBF16); rowstat is indexed by the PROBLEM row (m + m_off)
    EpiVTT shifted(long off) const { EpiVTT e = *this; e.m_off = m_off + off; e.m_valid = m_valid - off; return e; }
    template <int NV> struct Ctx {};
    template <int NV>
    __device__ __forceinline__ void prep(Ctx<NV>&, long) const {}
    // persistent tile (gemm256p.h): 16 consecutive token rows m0.. of feature n, bias already added; exactly two 16-B stores
    // (inline asm) when UNPREDICATED (interior tiles: every row and feature is inside the problem)
    static constexpr bool MREG16 = true;
    template <bool UNPREDICATED>
    __device__ __forceinline__ void store16(long m0, long n, const float* v) const {
        const long head = n >> 6, d = n & 63;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long m8 = m0 + h * 8 + m_off;
            const long s = m8 / ntp, t = m8 - s * ntp;  // ntp % 8 == 0: a group of 8 never straddles slices
            const u32x4 w{pack2bf(v[h * 8 + 0], v[h * 8 + 1]), pack2bf(v[h * 8 + 2], v[h * 8 + 3]), pack2bf(v[h * 8 + 4], v[h * 8 + 5]),
                          pack2bf(v[h * 8 + 6], v[h * 8 + 7])};
            if (UNPREDICATED || (m0 + h * 8 < m_valid && n < n_valid)) gst16_vaddr(vt + ((s * heads + head) * 64 + d) * (long)kp + t, w);
        }
    }
    template <int NV>
    __device__ __forceinline__ void store(const Ctx<NV>&, long m0, long n, const float* acc) const {
        static_assert(NV % 8 == 0, "V^T epilogue stores 8 tokens (16 B) at a time");
        if (n >= n_valid) return;
        const float b = bias[n];
        const float cs = LN ? bias[cs_off + n] : 0.f;
        const long head = n >> 6, d = n & 63;
#pragma unroll
        for (int h = 0; h < NV / 8; ++h) {
            if (m0 + h * 8 >= m_valid) continue;
            const long m8 = m0 + h * 8 + m_off;
            const long s = m8 / ntp, t = m8 - s * ntp;  // ntp % 8 == 0: a group of 8 never straddles slices
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (LN) v[i] = ln_fold(acc[h * 8 + i], *(const float2*)(rowstat + 2 * (m8 + i)), cs, b);
                else v[i] = acc[h * 8 + i] + b;
            }
            store_bf16_chunked<8>(vt + ((s * heads + head) * 64 + d) * (long)kp + t, v, 0, 1);
        }
    }
};
using EpiVT = EpiVTT<false>;

// ConvTranspose3d kernel=stride=(1,2,2) as a GEMM with N = 4*Cout (n = (i*2+j)*Cout + o), pixel-shuffle
// scatter into the channels-last output [D][2H][2W][Cout], GELU fused.  bias is pre-expanded to N entries.
template <int ACT, bool HALF = false>
struct EpiConvT {
    static constexpr bool F16 = HALF;
    uint16_t* out; const float* bias; int H, W, cout; long m_valid, n_valid;
    int up_z = 0;  // 1: kernel = stride = (2,2,2) (UNet3D's upconv, unet3d.py:166-170): n = ((iz*2+i)*2+j)*Cout + o, output [2D][2H][2W][Cout]
    template <int NV> using Ctx = VecCtx<NV>;
    template <int NV>
    __device__ __forceinline__ void prep(Ctx<NV>& c, long n0) const { load_vec<NV>(c.bias, bias + n0); }
    template <int NV>
    __device__ __forceinline__ void store(const Ctx<NV>& c, long n0, long m, const float* acc) const {
        if (m >= m_valid) return;
        const long xq = m % W, t = m / W, yq = t % H, zq0 = t / H;
        constexpr int CH = NV >= 8 ? 8 : 4;
#pragma unroll
        for (int h = 0; h < NV / CH; ++h) {
            const long n = n0 + h * CH;
            if (n >= n_valid) continue;
            const int ijk = (int)(n / cout), o = (int)(n - (long)ijk * cout);
            const int ij = ijk & 3;
            const long zq = up_z ? 2 * zq0 + (ijk >> 2) : zq0;
            float v[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const float tt = acc[h * CH + i] + c.bias[h * CH + i];
                v[i] = ACT == 1 ? gelu_erf(tt) : tt;
            }
            uint16_t* dst = out + ((zq * (2L * H) + 2 * yq + (ij >> 1)) * (2L * W) + 2 * xq + (ij & 1)) * cout + o;
            store_bf16_chunked<CH, HALF>(dst, v, 0, 1);
        }
    }
};

// ------------------------------------------------------------------------------------------------
// Kernels
// ------------------------------------------------------------------------------------------------
template <class Cfg, class Epi>
__global__ __launch_bounds__(GEMM_THREADS) void k_gemm_nreg(const uint16_t* A, long lda, const uint16_t* Wt, long ldw,
                                                             int nk, int tiles_n, int tiles_m, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int tr, tl;
    tile_coords(blockIdx.x, gridDim.x, tiles_n, tiles_m, tr, tl);
    PlainLoader<Cfg::BR, Cfg::FRG> ldr;
    PlainLoader<Cfg::BL, 0> ldl;
    ldr.init(Wt, ldw, (long)tr * Cfg::BR, threadIdx.x);
    ldl.init(A, lda, (long)tl * Cfg::BL, threadIdx.x);
    gemm_tile_body<Cfg>(ldr, ldl, epi, nk, (long)tr * Cfg::BR, (long)tl * Cfg::BL, smem);
}

template <class Cfg, class Epi>
__global__ __launch_bounds__(GEMM_THREADS) void k_gemm_mreg(const uint16_t* A, long lda, const uint16_t* Wt, long ldw,
                                                             int nk, int tiles_n, int tiles_m, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int tr, tl;
    tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, tr, tl);
    PlainLoader<Cfg::BR, Cfg::FRG> ldr;
    PlainLoader<Cfg::BL, 0> ldl;
    ldr.init(A, lda, (long)tr * Cfg::BR, threadIdx.x);
    ldl.init(Wt, ldw, (long)tl * Cfg::BL, threadIdx.x);
    gemm_tile_body<Cfg>(ldr, ldl, epi, nk, (long)tr * Cfg::BR, (long)tl * Cfg::BL, smem);
}

template <class Cfg, class Epi>
__global__ __launch_bounds__(GEMM_THREADS) void k_conv3_nreg(const uint16_t* in, const uint16_t* zero, int C, int D, int H,
                                                              int W, int dil, const uint16_t* Wt, long ldw, int nk,
                                                              int tiles_n, int tiles_m, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int tr, tl;
    tile_coords(blockIdx.x, gridDim.x, tiles_n, tiles_m, tr, tl);
    PlainLoader<Cfg::BR, Cfg::FRG> ldr;
    Conv3Loader<Cfg::BL> ldl;
    ldr.init(Wt, ldw, (long)tr * Cfg::BR, threadIdx.x);
    ldl.init(in, zero, C, D, H, W, dil, (long)tl * Cfg::BL, (long)D * H * W, threadIdx.x);
    gemm_tile_body<Cfg>(ldr, ldl, epi, nk, (long)tr * Cfg::BR, (long)tl * Cfg::BL, smem);
}

// Conv3d kernel = stride = 2 (UNet3D's pooling convolution, unet3d.py:127-131) as an implicit GEMM: row = output voxel,
// K index = ((iz*2+iy)*2+ix)*C + c gathered from the 2x2x2 input voxels it covers (never out of the volume; rows past the last
// output voxel read the zero page).  C is a power of two >= 8, so 8 C is a multiple of the 64-deep K tile.
template <int ROWS>
struct Pool2Loader {
    static constexpr int NJ = (ROWS * 8 + GEMM_THREADS - 1) / GEMM_THREADS;
    const uint16_t* in;
    const uint16_t* zero;
    int cshift, Hi, Wi;
    long vin[NJ];  // linear index of input voxel (2z, 2y, 2x)
    int chunk[NJ];
    bool rowok[NJ];
    int wave;
    __device__ __forceinline__ void init(const uint16_t* in_, const uint16_t* zero_, int cshift_, int Do, int Ho, int Wo, long row0, int tid) {
        in = in_; zero = zero_; cshift = cshift_; Hi = 2 * Ho; Wi = 2 * Wo;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const long nvox = (long)Do * Ho * Wo;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = j * GEMM_THREADS + tid;
            const int row = c >> 3, pos = c & 7;
            chunk[j] = pos ^ swz_chunk(row);
            long v = row0 + row;
            rowok[j] = v < nvox;
            if (!rowok[j]) v = 0;
            const long x = v % Wo, t = v / Wo, y = t % Ho, z = t / Ho;
            vin[j] = ((2 * z) * Hi + 2 * y) * Wi + 2 * x;
        }
    }
    __device__ __forceinline__ void issue(char* lds_tile, int kt) const {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if ((ROWS * 8) % GEMM_THREADS != 0 && (j * GEMM_THREADS + wave * 64) >= ROWS * 8) break;
            const int kk = kt * BK + chunk[j] * 8;
            const int tap = kk >> cshift, c = kk - (tap << cshift);
            const long vs = vin[j] + ((long)(tap >> 2) * Hi + ((tap >> 1) & 1)) * Wi + (tap & 1);
            const uint16_t* g = (rowok[j] && tap < 8) ? in + ((vs << cshift) + c) : zero;
            glds16_vaddr(g, lds_addr(lds_tile) + (j * GEMM_THREADS + wave * 64) * 16);
        }
    }
};

template <class Cfg, class Epi>
__global__ __launch_bounds__(GEMM_THREADS) void k_pool2_nreg(const uint16_t* in, const uint16_t* zero, int cshift, int Do, int Ho, int Wo,
                                                              const uint16_t* Wt, long ldw, int nk, int tiles_n, int tiles_m, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int tr, tl;
    tile_coords(blockIdx.x, gridDim.x, tiles_n, tiles_m, tr, tl);
    PlainLoader<Cfg::BR, Cfg::FRG> ldr;
    Pool2Loader<Cfg::BL> ldl;
    ldr.init(Wt, ldw, (long)tr * Cfg::BR, threadIdx.x);
    ldl.init(in, zero, cshift, Do, Ho, Wo, (long)tl * Cfg::BL, threadIdx.x);
    gemm_tile_body<Cfg>(ldr, ldl, epi, nk, (long)tr * Cfg::BR, (long)tl * Cfg::BL, smem);
}

// 256x256 phase-pipelined tile (gemm256.h): NREG (R = weights) and MREG (R = activations) orientations
template <class Epi, int VARIANT>
__global__ __launch_bounds__(G256_THREADS) void k_gemm256_nreg(const uint16_t* A, long lda, const uint16_t* Wt, long ldw, int nk,
                                                                int tiles_n, int tiles_m, Epi epi, int stagger) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stagger_first_round(stagger);
    int tr, tl;
    tile_coords(blockIdx.x, gridDim.x, tiles_n, tiles_m, tr, tl);
    const int koff = VARIANT == 7 ? ((tr & 3) + (tl & 7)) % nk : 0;
    gemm256_body<(VARIANT == 7 ? 5 : VARIANT)>(Wt, ldw, A, lda, nk, (long)tr * 256, (long)tl * 256, epi, smem, koff);
}
template <class Epi, int VARIANT>
__global__ __launch_bounds__(G256_THREADS) void k_gemm256_mreg(const uint16_t* A, long lda, const uint16_t* Wt, long ldw, int nk,
                                                                int tiles_n, int tiles_m, Epi epi, int stagger) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stagger_first_round(stagger);
    int tr, tl;
    tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, tr, tl);
    gemm256_body<VARIANT>(A, lda, Wt, ldw, nk, (long)tr * 256, (long)tl * 256, epi, smem);
}

// persistent form (gemm256p.h): one workgroup per CU walks its tiles, the operand stream runs across tile boundaries
template <class Epi, bool FULL, bool DBG = false>
__global__ __launch_bounds__(G256_THREADS) void k_gemm256p_nreg(const uint16_t* A, long lda, const uint16_t* Wt, long ldw, int nk,
                                                                 int tiles_n, int tiles_m, int group_l, int xcd_stagger, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm256p_body<Epi, FULL, DBG>(Wt, ldw, A, lda, nk, tiles_n, tiles_m, group_l, xcd_stagger, epi, smem);
}

template <class Epi, bool FULL>
__global__ __launch_bounds__(G256_THREADS) void k_gemm256p_mreg(const uint16_t* A, long lda, const uint16_t* Wt, long ldw, int nk,
                                                                 int tiles_n, int tiles_m, int group_l, int xcd_stagger, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm256p_body<Epi, FULL, false>(A, lda, Wt, ldw, nk, tiles_m, tiles_n, group_l, xcd_stagger, epi, smem);
}

// tuning switches (cvx_set_option): A/B the tile kernels and pipeline schedules inside ONE process
static std::atomic<int> g_resid_stagger{0};  // cycles between the XCDs' start offsets in the residual GEMMs ("gemm_resid_stagger")
static std::atomic<int> g_resid_reverse{0};  // residual GEMMs walk their tile sequence from the end (A/B: "gemm_resid_reverse")
static std::atomic<int> g_tile_group_l_host{8};  // host copy of g_tile_group_l (the persistent kernel takes it as an argument)
static std::atomic<int> g_use_gemm256{1}, g_gemm256_variant{9}, g_gemm_stagger{0};  // 9 = persistent (gemm256p.h)  // stagger: measured no gain (tools/bench_gemm.py 5 vs 1005)
template <class Epi> static constexpr int epilogue_cycles() { return 12000; }       // bf16 store epilogues (stamped)
template <> constexpr int epilogue_cycles<EpiResid>() { return 40000; }              // fp32 read-modify-write
template <> constexpr int epilogue_cycles<EpiF32>() { return 20000; }

#ifdef CVX_ABLATION
template <class Epi, bool MREG, int VARIANT>
__global__ __launch_bounds__(G4W_THREADS) void k_gemm4w(const uint16_t* A, long lda, const uint16_t* Wt, long ldw, int nks,
                                                         int tiles_n, int tiles_m, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int tr, tl;
    if constexpr (MREG) {
        tile_coords(blockIdx.x, gridDim.x, tiles_m, tiles_n, tr, tl);
        gemm4w_body<VARIANT>(A, lda, Wt, ldw, nks, (long)tr * 256, (long)tl * 256, epi, smem);
    } else {
        tile_coords(blockIdx.x, gridDim.x, tiles_n, tiles_m, tr, tl);
        gemm4w_body<VARIANT>(Wt, ldw, A, lda, nks, (long)tr * 256, (long)tl * 256, epi, smem);
    }
}
#endif

static bool use_gemm256(long M, long Npad, long Kpad) {
    return g_use_gemm256 && Npad % 256 == 0 && Kpad % BK == 0 && Kpad / BK >= 4 && M >= 1024;
}

template <class Epi, bool MREG>
static int launch_256(const uint16_t* A, long lda, const uint16_t* Wt, long ldw, long M, long Npad, long Kpad, const Epi& epi,
                      hipStream_t st) {
    const int tiles_n = (int)(Npad / 256), tiles_m = (int)((M + 255) / 256);
#ifdef CVX_ABLATION
    if (g_use_gemm256 == 2) {  // one-wave-per-SIMD tile (gemm4w.h)
        auto k4 = g_gemm256_variant == 1 ? k_gemm4w<Epi, MREG, 1> : g_gemm256_variant == 2 ? k_gemm4w<Epi, MREG, 2> : k_gemm4w<Epi, MREG, 0>;
        CVX_HIP(hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, G4W_LDS_BYTES));
        hipLaunchKernelGGL(k4, dim3(tiles_n * tiles_m), dim3(G4W_THREADS), G4W_LDS_BYTES, st, A, lda, Wt, ldw, (int)(Kpad / G4W_KS),
                           tiles_n, tiles_m, epi);
        return cvx_check_launch();
    }
#endif
    const int variant = g_gemm256_variant;
    if constexpr ((!MREG && (epi_has_preload<Epi>::value || epi_has_produce<Epi>::value || epi_has_hl<Epi>::value)) || (MREG && epi_is_mreg<Epi>::value)) {
        if (variant == 9 || variant == 29) {
            // one workgroup per CU (128 KiB of LDS each), a multiple of 8 so every XCD gets the same number
            // (per DEVICE: a process may drive several GPUs, one volume per stream each)
            static std::atomic<int> n_cu_of[64];
            int dev = 0;
            CVX_HIP(hipGetDevice(&dev));
            int n_cu = dev >= 0 && dev < 64 ? n_cu_of[dev].load() : 0;
            if (!n_cu) {
                int n = 0;
                CVX_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
                n_cu = n >= 8 ? n / 8 * 8 : 8;
                if (dev >= 0 && dev < 64) n_cu_of[dev] = n_cu;
            }
            const int ntiles = tiles_n * tiles_m;
            const int grid = ntiles >= n_cu ? n_cu : (ntiles + 7) / 8 * 8;
            const bool full = M % 256 == 0 && epi.n_valid == Npad;
            void (*kp)(const uint16_t*, long, const uint16_t*, long, int, int, int, int, int, Epi);
            if constexpr (MREG) {
                kp = full ? k_gemm256p_mreg<Epi, true> : k_gemm256p_mreg<Epi, false>;
            } else {
                kp = full ? k_gemm256p_nreg<Epi, true> : k_gemm256p_nreg<Epi, false>;
#ifdef CVX_ABLATION
                if (variant == 29) kp = k_gemm256p_nreg<Epi, true, true>;  // stamped (interior tiles only)
#endif
            }
            CVX_HIP(hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, G256P_LDS_BYTES));
            hipLaunchKernelGGL(kp, dim3(grid), dim3(G256_THREADS), G256P_LDS_BYTES, st, A, lda, Wt, ldw, (int)(Kpad / BK), tiles_n, tiles_m,
                               ((epi_has_preload<Epi>::value || epi_has_hl<Epi>::value) && g_resid_reverse ? -1 : 1) * (int)g_tile_group_l_host,
                               ntiles > grid ? ((epi_has_preload<Epi>::value || epi_has_hl<Epi>::value) ? (int)g_resid_stagger : (int)g_gemm_stagger) : 0, epi);
            return cvx_check_launch();
        }
    }
    void (*k)(const uint16_t*, long, const uint16_t*, long, int, int, int, Epi, int);
    if constexpr (MREG) {
        k = k_gemm256_mreg<Epi, 0>;
    } else {
        switch (variant) {
#ifdef CVX_ABLATION  // the measured-and-rejected schedules and the timing-only builds (stamps / garbage-output ablations): never in the product library
            case 1: k = k_gemm256_nreg<Epi, 1>; break;
            case 5: k = k_gemm256_nreg<Epi, 5>; break;
            case 6: k = k_gemm256_nreg<Epi, 6>; break;
            case 7: k = k_gemm256_nreg<Epi, 7>; break;
            case 8: k = k_gemm256_nreg<Epi, 8>; break;
            case 20: k = k_gemm256_nreg<Epi, 20>; break;
            case 21: k = k_gemm256_nreg<Epi, 21>; break;
            case 10: k = k_gemm256_nreg<Epi, 10>; break;
            case 11: k = k_gemm256_nreg<Epi, 11>; break;
            case 12: k = k_gemm256_nreg<Epi, 12>; break;
            case 13: k = k_gemm256_nreg<Epi, 13>; break;
#endif
            default: k = k_gemm256_nreg<Epi, 0>; break;
        }
    }
    CVX_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS_BYTES));
    // quarter of one tile's duration (~2800 cycles per K tile + epilogue), only when the launch has several rounds
    const long nk = Kpad / BK;
    const int stagger = (g_gemm_stagger && (long)tiles_n * tiles_m > 512) ? (int)((nk * 2800 + epilogue_cycles<Epi>()) / 4) : 0;
    hipLaunchKernelGGL(k, dim3(tiles_n * tiles_m), dim3(G256_THREADS), G256_LDS_BYTES, st, A, lda, Wt, ldw, (int)nk, tiles_n, tiles_m,
                       epi, stagger);
    return cvx_check_launch();
}

template <class E, class = void> struct epi_can_shift : std::false_type {};
template <class E> struct epi_can_shift<E, std::void_t<decltype(std::declval<const E&>().shifted(0L))>> : std::true_type {};

// Tail split: a 256-tile grid whose last round would keep only a few CUs busy (e.g. 3096 tiles = 12 rounds + 24 tiles for
// the N = 1536 GEMMs of one 128-slice batch) is cut into a main launch of whole rounds and a tail launch over the remaining
// M rows with 128x128 tiles (4x as many, quarter-size tiles: the tail costs ~0.3 of a round instead of a full one).
static std::atomic<int> g_tail_split{2};  // 2: also for the residual epilogue with a short K loop (proj), which pays off since the tail runs on 64 x 128 tiles
static std::atomic<int> g_tail_deep{1};  // cvx_set_option("gemm_tail_deep"): tails run on the multi-stage ring (0: the double-buffered tiles, for A/B)
static std::atomic<int> g_tail_tile{1};  // tail launches: 1 = 64 x 128 tiles (twice the workgroups on the idle chip), 0 = 128 x 128
static std::atomic<int> g_tail_max{128};  // (64 -> 128 in round 3: 281.26 -> 280.26 ms per tomogram; 0 = never: 282.89) largest last partial round (in 256 x 256 tiles) that is cut off into a tail launch ("gemm_tail_max")
static long tail_split_rows(long M, long Npad, bool allow = true) {
    const long tiles_n = Npad / 256, tiles_m = (M + 255) / 256, tiles = tiles_n * tiles_m, rem = tiles % 256;
    if (!g_tail_split || !allow || tiles < 512 || rem == 0 || rem > g_tail_max) return M;
    const long main_mtiles = (tiles - rem) / tiles_n;
    return main_mtiles > 0 ? main_mtiles * 256 : M;
}

template <class Cfg, class Epi>
static int launch_nreg(const uint16_t* A, long lda, const uint16_t* Wt, long ldw, long M, long Npad, long Kpad,
                       const Epi& epi, hipStream_t st) {
    const int tiles_n = (int)(Npad / Cfg::BR), tiles_m = (int)((M + Cfg::BL - 1) / Cfg::BL);
    auto k = k_gemm_nreg<Cfg, Epi>;
    CVX_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
    hipLaunchKernelGGL(k, dim3(tiles_n * tiles_m), dim3(GEMM_THREADS), Cfg::LDS_BYTES, st, A, lda, Wt, ldw,
                       (int)(Kpad / BK), tiles_n, tiles_m, epi);
    return cvx_check_launch();
}

// 256-tile launch with the tail split (NREG epilogues that can be re-based on a row offset)
template <class Epi>
static int launch_256_split(const uint16_t* A, long lda, const uint16_t* Wt, long ldw, long M, long Npad, long Kpad, const Epi& epi,
                            hipStream_t st) {
    if constexpr (epi_can_shift<Epi>::value) {
        // (the 128-tile kernel's fp32 read-modify-write epilogue is not LDS-staged: with a short K loop the tail would cost
        //  more than the idle round it removes -- measured on the proj GEMM)
        const long m_main = tail_split_rows(M, Npad, !(epi_has_preload<Epi>::value || epi_has_hl<Epi>::value) || Kpad >= 2048 || g_tail_split == 2);
        if (m_main < M) {
            int rc = launch_256<Epi, false>(A, lda, Wt, ldw, m_main, Npad, Kpad, epi, st);
            if (rc) return rc;
            // the tail is a handful of tiles on an otherwise idle chip, one workgroup per CU at most: what it waits for is the DMA
            // round trip of each K tile, so it gets the ring with 2-3 K tiles in flight (gemm_core.h, TileCfg STAGES)
            const long rows = M - m_main;
            if (g_tail_deep) {
                // (the residual GEMMs' tails only: 192 workgroups.  For the qkv / w12 tails -- 288 / 512 workgroups of 128 x 128 -- a
                //  128 x 256 tile on a 3-deep ring, one workgroup per CU, measured SLOWER than two double-buffered workgroups per
                //  CU: 34.5 / 36.1 us against 29.1 / 30.7; 64 x 128 tiles on a 3-deep ring, two per CU: +0.36 ms per tomogram)
                if constexpr (epi_has_preload<Epi>::value || epi_has_hl<Epi>::value) {
                    if ((Npad / 64) * ((rows + 127) / 128) <= 256) {
                        return launch_nreg<TileCfg<64, 128, 1, 4>>(A + m_main * lda, lda, Wt, ldw, rows, Npad, Kpad, epi.shifted(m_main), st);
                    }
                }
            }
            // (before the ring: more, smaller workgroups finish sooner)
            if (g_tail_tile == 2 && (M - m_main) / 256 * (Npad / 256) * 16 <= 1024)
                // 64 x 64 tiles, 32 KB of LDS each: three workgroups per CU.  The tail is LATENCY-bound (one K tile of look-ahead, a
                // K tile per DMA round trip): co-resident workgroups overlap each other's waits
                return launch_nreg<TileCfg<64, 64, 1>>(A + m_main * lda, lda, Wt, ldw, M - m_main, Npad, Kpad, epi.shifted(m_main), st);
            if (g_tail_tile && (M - m_main) / 256 * (Npad / 256) * 8 <= 256)  // (8 small tiles per 256 x 256 tile: one round at most)
                return launch_nreg<TileCfg<64, 128, 1>>(A + m_main * lda, lda, Wt, ldw, M - m_main, Npad, Kpad, epi.shifted(m_main), st);
            return launch_nreg<TileCfg<128, 128, 2>>(A + m_main * lda, lda, Wt, ldw, M - m_main, Npad, Kpad, epi.shifted(m_main), st);
        }
    }
    return launch_256<Epi, false>(A, lda, Wt, ldw, M, Npad, Kpad, epi, st);
}

template <class Epi>
static int dispatch_nreg(const uint16_t* A, long lda, const uint16_t* Wt, long ldw, long M, long Npad, long Kpad,
                         const Epi& epi, hipStream_t st) {
    if (Kpad % BK) return cvx_fail("gemm: K must be padded to a multiple of 64");
    if (use_gemm256(M, Npad, Kpad)) return launch_256_split(A, lda, Wt, ldw, M, Npad, Kpad, epi, st);
    if (Npad % 128 == 0) return launch_nreg<TileCfg<128, 128, 2>>(A, lda, Wt, ldw, M, Npad, Kpad, epi, st);
    if (Npad % 64 == 0) return launch_nreg<TileCfg<64, 256, 1>>(A, lda, Wt, ldw, M, Npad, Kpad, epi, st);
    if constexpr (!epi_has_hl<Epi>::value) {  // (the hi/lo residual epilogue works on 64-column slots)
        if (Npad % 32 == 0) return launch_nreg<TileCfg<32, 256, 1>>(A, lda, Wt, ldw, M, Npad, Kpad, epi, st);
        if (Npad % 16 == 0) return launch_nreg<TileCfg<16, 256, 1>>(A, lda, Wt, ldw, M, Npad, Kpad, epi, st);
    }
    return cvx_fail("gemm: N must be padded to a multiple of 16 (64 for the hi/lo residual epilogue)");
}

template <class Cfg, class Epi>
static int launch_conv3(const cvx_conv3d_desc& d, const Epi& epi, hipStream_t st) {
    const long M = (long)d.D * d.H * d.W;
    const int tiles_n = (int)(d.n_pad / Cfg::BR), tiles_m = (int)((M + Cfg::BL - 1) / Cfg::BL);
    auto k = k_conv3_nreg<Cfg, Epi>;
    CVX_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
    hipLaunchKernelGGL(k, dim3(tiles_n * tiles_m), dim3(GEMM_THREADS), Cfg::LDS_BYTES, st, (const uint16_t*)d.in,
                       (const uint16_t*)d.zero_page, d.C, d.D, d.H, d.W, d.dil, (const uint16_t*)d.w, (long)d.k_pad,
                       (int)(d.k_pad / BK), tiles_n, tiles_m, epi);
    return cvx_check_launch();
}

template <class Cfg, class Epi>
static int launch_pool2(const cvx_conv3d_desc& d, int cshift, const Epi& epi, hipStream_t st) {
    const int Do = d.D / 2, Ho = d.H / 2, Wo = d.W / 2;
    const long M = (long)Do * Ho * Wo;
    const int tiles_n = (int)(d.n_pad / Cfg::BR), tiles_m = (int)((M + Cfg::BL - 1) / Cfg::BL);
    auto k = k_pool2_nreg<Cfg, Epi>;
    CVX_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
    hipLaunchKernelGGL(k, dim3(tiles_n * tiles_m), dim3(GEMM_THREADS), Cfg::LDS_BYTES, st, (const uint16_t*)d.in, (const uint16_t*)d.zero_page,
                       cshift, Do, Ho, Wo, (const uint16_t*)d.w, (long)d.k_pad, (int)(d.k_pad / BK), tiles_n, tiles_m, epi);
    return cvx_check_launch();
}

}  // namespace cvx

namespace cvx {  // convt.hip: few-channel (1,2,2) transposed convolutions without LDS / K padding
bool convt_small_eligible(const cvx_gemm_desc& d);
int convt_small_dispatch(const cvx_gemm_desc& d, hipStream_t st);
}

using namespace cvx;

extern std::atomic<int> g_attn_variant, g_attn_xcd_remap, g_attn_mfma_prio, g_attn_half_tile;  // attention.hip
extern std::atomic<int> g_ln_policy;                         // norm.hip
extern std::atomic<int> g_win_attn_prefetch, g_win_attn_x32; // hiera.hip

extern "C" int cvx_debug_read_gemm256(unsigned long long* out32) {
    CVX_HIP(hipMemcpyFromSymbol(out32, HIP_SYMBOL(cvx::g_gemm256_dbg), sizeof(unsigned long long) * 32));
    return 0;
}
#ifdef CVX_LN_EMIT_PROTO
static uint16_t* g_emit_xb_host = nullptr;
static long g_emit_ldb_host = 0;
static float* g_emit_part_host = nullptr;
static long g_emit_rows_host = 0;
extern "C" int cvx_debug_set_emit(void* xb, long ldb, float* part, long rows) {  // timing prototype only (tools/bench_ln_emit.py)
    g_emit_xb_host = (uint16_t*)xb;
    g_emit_ldb_host = ldb;
    g_emit_part_host = part;
    g_emit_rows_host = rows;
    return 0;
}
#endif
extern "C" int cvx_debug_read_gemm256p(unsigned long long* out96) {
    CVX_HIP(hipMemcpyFromSymbol(out96, HIP_SYMBOL(cvx::g_gemm256p_dbg), sizeof(unsigned long long) * 96));
    return 0;
}

// few-channel 3x3x3 convolutions: 2 = z-marching LDS-ring kernel (default), 1 = per-tile halo kernel, 0 = implicit GEMM (A/B runs, tests)
static std::atomic<int> g_conv_halo{2};
static std::atomic<int> g_convt_small{1};  // dedicated kernel for the head's 16->8 and 32->32 transposed convolutions (0: GEMM tile)
static std::atomic<int> g_conv_wide{1};  // 192-wide implicit-GEMM tile for C_out % 192 == 0 (0: three 64-wide tiles)

extern "C" int cvx_set_option(const char* name, int value) {
    if (!name) return cvx_fail("set_option: null name");
    // Process-wide tuning defaults, read ONCE at the start of every launch call.  Every accepted value selects a kernel that
    // computes the same result (parity-tested); the timing-only ablation kernels exist only in -DCVX_ABLATION builds.
    auto one_of = [&](std::initializer_list<int> ok) { for (int v : ok) if (v == value) return true; return false; };
#ifdef CVX_ABLATION
    const bool abl = true;
#else
    const bool abl = false;
#endif
    if (!strcmp(name, "use_gemm256")) {
        if (!one_of({0, 1}) && !(abl && value == 2)) return cvx_fail("set_option: use_gemm256 must be 0 or 1 (2, the 4-wave tile, needs a -DCVX_ABLATION build)");
        g_use_gemm256 = value;
    } else if (!strcmp(name, "gemm256_variant")) {
        if (!one_of({0, 9}) && !(abl && one_of({1, 2, 5, 6, 7, 8, 10, 11, 12, 13, 20, 21, 29})))
            return cvx_fail("set_option: unknown gemm256_variant (ablation variants need a -DCVX_ABLATION build)");
        g_gemm256_variant = value;
    } else if (!strcmp(name, "gemm_stagger")) {
        if (value < 0 || value > 1000000) return cvx_fail("set_option: gemm_stagger is a cycle count in [0, 1e6]");
        g_gemm_stagger = value;  // persistent kernel: start offset between XCDs, cycles; the one-shot kernels: on / off
    }
    else if (!strcmp(name, "gemm_tail_split")) {
        if (!one_of({0, 1, 2})) return cvx_fail("set_option: gemm_tail_split is 0 (off), 1 (on; residual epilogues only for K >= 2048) or 2 (always)");
        g_tail_split = value;
    }
    else if (!strcmp(name, "gemm_tail_max")) {
        if (value < 0 || value > 255) return cvx_fail("set_option: gemm_tail_max is a tile count in [0, 255]");
        g_tail_max = value;
    }
    else if (!strcmp(name, "gemm_tail_tile")) {
        if (!one_of({0, 1, 2})) return cvx_fail("set_option: gemm_tail_tile is 0 (128 x 128), 1 (64 x 128) or 2 (64 x 64 tiles)");
        g_tail_tile = value;
    }
    else if (!strcmp(name, "gemm_resid_reverse")) g_resid_reverse = value != 0;
    else if (!strcmp(name, "gemm_resid_stagger")) {
        if (value < 0 || value > 1000000) return cvx_fail("set_option: gemm_resid_stagger is a cycle count in [0, 1e6]");
        g_resid_stagger = value;
    }
    else if (!strcmp(name, "conv_halo")) {
        if (!one_of({0, 2}) && !(abl && value == 1))
            return cvx_fail("set_option: conv_halo is 0 (implicit GEMM) or 2 (z-marching ring); 1 (the round-1 tile-halo kernel) needs a -DCVX_ABLATION build");
        g_conv_halo = value;
    }
    else if (!strcmp(name, "conv_wide")) g_conv_wide = value != 0;
    else if (!strcmp(name, "convt_small")) g_convt_small = value != 0;
    else if (!strcmp(name, "win_attn_prefetch")) {
        if (!one_of({0, 1, 2})) return cvx_fail("set_option: win_attn_prefetch is 0 (off), 1 (one tile ahead) or 2 (two tiles ahead)");
        g_win_attn_prefetch = value;
    }
    else if (!strcmp(name, "win_attn_x32")) g_win_attn_x32 = value != 0;
    else if (!strcmp(name, "ln_policy")) {
        if (!one_of({0, 1, 2, 3})) return cvx_fail("set_option: ln_policy is a 2-bit mask (1: cacheable loads, 2: rows walked from the end)");
        g_ln_policy = value;
    }
    else if (!strcmp(name, "attn_variant")) {
        if (!one_of({0, 7}) && !(abl && one_of({1, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13})))
            return cvx_fail("set_option: unknown attn_variant (ablation variants need a -DCVX_ABLATION build)");
        g_attn_variant = value;
    } else if (!strcmp(name, "attn_xcd_remap")) g_attn_xcd_remap = value != 0;
    else if (!strcmp(name, "gemm_tail_deep")) g_tail_deep = value != 0;  // 0: double-buffered tail tiles (A/B); a 6-deep ring measured equal to the 4-deep one
    else if (!strcmp(name, "attn_mfma_prio")) g_attn_mfma_prio = value & 3;
    else if (!strcmp(name, "attn_half_tile")) g_attn_half_tile = value != 0;
    else if (!strcmp(name, "tile_group_l")) {
        if (value < 1) return cvx_fail("set_option: tile_group_l must be >= 1");
        g_tile_group_l_host = value;
        CVX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(cvx::g_tile_group_l), &value, sizeof(int)));
    }
    else return cvx_fail("set_option: unknown option");
    return 0;
}

// measurement hook (bench.py): HIP events recorded on the launch stream around every GEMM of one epilogue kind
static struct { int epi; hipEvent_t* start; hipEvent_t* stop; int cap, count; } g_hook = {-1, nullptr, nullptr, 0, 0};

extern "C" int cvx_set_gemm_event_hook(int epilogue, void** start_events, void** stop_events, int capacity) {
    g_hook.epi = (start_events && stop_events && capacity > 0) ? epilogue : -1;
    g_hook.start = (hipEvent_t*)start_events;
    g_hook.stop = (hipEvent_t*)stop_events;
    g_hook.cap = capacity;
    g_hook.count = 0;
    return 0;
}
extern "C" int cvx_get_gemm_event_count(void) { return g_hook.count; }

static int gemm_dispatch(const cvx_gemm_desc* d, hipStream_t st);

extern "C" int cvx_gemm_bf16(const cvx_gemm_desc* d, hipStream_t st) {
    if (!d) return cvx_fail("gemm: null descriptor");
    if (d->epilogue == g_hook.epi && g_hook.count < g_hook.cap) {
        const int i = g_hook.count++;
        CVX_HIP(hipEventRecord(g_hook.start[i], st));
        const int rc = gemm_dispatch(d, st);
        CVX_HIP(hipEventRecord(g_hook.stop[i], st));
        return rc;
    }
    return gemm_dispatch(d, st);
}

// V^T GEMM (MREG orientation): whole rounds on the persistent 256 tile, the rest (or a small problem) on 64 x 128 / 128 x 128 tiles
template <class E>
static int dispatch_vt(const cvx_gemm_desc* d, const uint16_t* A, const uint16_t* W, const E& e, hipStream_t st) {
    using Cfg = TileCfg<128, 128, 2>;
    auto tail128 = [&](const uint16_t* a, long m, const E& ev) {
        const int tiles_n = (int)(d->n_pad / Cfg::BL), tiles_m = (int)((m + Cfg::BR - 1) / Cfg::BR);
        auto k = k_gemm_mreg<Cfg, E>;
        CVX_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        hipLaunchKernelGGL(k, dim3(tiles_n * tiles_m), dim3(GEMM_THREADS), Cfg::LDS_BYTES, st, a, d->lda, W, d->ldw,
                           (int)(d->k_pad / BK), tiles_n, tiles_m, ev);
        return cvx_check_launch();
    };
    if (use_gemm256(d->m, d->n_pad, d->k_pad)) {
        const long m_main = tail_split_rows(d->m, d->n_pad);
        int rc = launch_256<E, true>(A, d->lda, W, d->ldw, m_main, d->n_pad, d->k_pad, e, st);
        if (rc || m_main == d->m) return rc;
        if (g_tail_tile && (d->m - m_main) / 256 * (d->n_pad / 256) * 8 <= 256) {
            using CfgS = TileCfg<64, 128, 1>;
            const long m = d->m - m_main;
            const int tiles_n = (int)(d->n_pad / CfgS::BL), tiles_m = (int)((m + CfgS::BR - 1) / CfgS::BR);
            auto k = k_gemm_mreg<CfgS, E>;
            CVX_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, CfgS::LDS_BYTES));
            hipLaunchKernelGGL(k, dim3(tiles_n * tiles_m), dim3(GEMM_THREADS), CfgS::LDS_BYTES, st, A + m_main * d->lda, d->lda, W, d->ldw,
                               (int)(d->k_pad / BK), tiles_n, tiles_m, e.shifted(m_main));
            return cvx_check_launch();
        }
        return tail128(A + m_main * d->lda, d->m - m_main, e.shifted(m_main));
    }
    return tail128(A, d->m, e);
}

static int gemm_dispatch(const cvx_gemm_desc* d, hipStream_t st) {
    const uint16_t* A = (const uint16_t*)d->a;
    const uint16_t* W = (const uint16_t*)d->w;
    if (d->m <= 0) return 0;
    if (!A || !W || !d->out || !d->bias) return cvx_fail("gemm: a, w, out and bias must be device pointers");
    if (d->dtype != CVX_DTYPE_BF16 && d->dtype != CVX_DTYPE_F16) return cvx_fail("gemm: unknown dtype");
    if (d->dtype == CVX_DTYPE_F16 && d->epilogue != CVX_EPI_BF16 && d->epilogue != CVX_EPI_BF16_GELU && d->epilogue != CVX_EPI_CONVT)
        return cvx_fail("gemm: fp16 operands are built for the plain / GELU / ConvT epilogues (the segmentation head)");
    switch (d->epilogue) {
        case CVX_EPI_BF16: {
            if (d->ln_rowstat) {
                EpiBF16<0, false, true> e{(uint16_t*)d->out, d->ldc, d->bias, d->m, d->n, d->ln_rowstat, d->n_pad};
                return dispatch_nreg(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
            }
            if (d->dtype == CVX_DTYPE_F16) {
                EpiBF16<0, true> e{(uint16_t*)d->out, d->ldc, d->bias, d->m, d->n};
                return dispatch_nreg(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
            }
            EpiBF16<0> e{(uint16_t*)d->out, d->ldc, d->bias, d->m, d->n};
            return dispatch_nreg(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
        }
        case CVX_EPI_BF16_GELU: {
            if (d->ln_rowstat) {
                EpiBF16<1, false, true> e{(uint16_t*)d->out, d->ldc, d->bias, d->m, d->n, d->ln_rowstat, d->n_pad};
                return dispatch_nreg(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
            }
            if (d->dtype == CVX_DTYPE_F16) {
                EpiBF16<1, true> e{(uint16_t*)d->out, d->ldc, d->bias, d->m, d->n};
                return dispatch_nreg(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
            }
            EpiBF16<1> e{(uint16_t*)d->out, d->ldc, d->bias, d->m, d->n};
            return dispatch_nreg(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
        }
        case CVX_EPI_SWIGLU: {
            if (d->n_pad % 128) return cvx_fail("gemm: SwiGLU needs N padded to 128");
            if (d->ln_rowstat) {
                EpiSwiGLUT<true> e{(uint16_t*)d->out, d->ldc, d->bias, d->m, d->n, d->ln_rowstat, d->n_pad};
                if (use_gemm256(d->m, d->n_pad, d->k_pad)) return launch_256_split(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
                return launch_nreg<TileCfg<128, 128, 2>>(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
            }
            EpiSwiGLU e{(uint16_t*)d->out, d->ldc, d->bias, d->m, d->n};
            if (use_gemm256(d->m, d->n_pad, d->k_pad)) return launch_256_split(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
            return launch_nreg<TileCfg<128, 128, 2>>(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
        }
        case CVX_EPI_RESID: {
            if (!d->out || !d->bias || !d->gamma) return cvx_fail("gemm: the residual epilogue needs out, bias and gamma (LayerScale)");
            EpiResid e{(float*)d->out, d->ldc, d->bias, d->gamma, d->m, d->n};
#ifdef CVX_LN_EMIT_PROTO
            if (!g_emit_xb_host) return cvx_fail("emit prototype: cvx_debug_set_emit first (the store counts of this build assume the emission)");
            e.emit_xb = g_emit_xb_host; e.emit_ldb = g_emit_ldb_host;
            if (!g_emit_part_host) return cvx_fail("emit prototype: the partial-sum buffer is required too");
            e.emit_part = g_emit_part_host; e.emit_rows = g_emit_rows_host;
#endif
            return dispatch_nreg(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
        }
        case CVX_EPI_RESID_HL: {
            if (!d->out || !d->out2 || !d->bias || !d->gamma || !d->stat_part)
                return cvx_fail("gemm: the hi/lo residual epilogue needs out (hi), out2 (lo), bias, gamma and stat_part");
            if (d->n % 64 || d->n_pad % 64 || d->n > d->n_pad || d->ldc < d->n)
                return cvx_fail("gemm: the hi/lo residual epilogue needs N a multiple of 64 (64-column statistics slots), N <= n_pad, ldc >= N");
            if (d->stat_rows < (d->m + 255) / 256 * 256) return cvx_fail("gemm: stat_rows must cover M rounded up to 256 rows");
            EpiResidHL e{(uint16_t*)d->out, (uint16_t*)d->out2, d->ldc, d->bias, d->gamma, d->stat_part, d->stat_rows, d->m, d->n};
            return dispatch_nreg(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
        }
        case CVX_EPI_F32: {
            if (!d->out || !d->bias || !d->gamma) return cvx_fail("gemm: the fp32 epilogue needs out, bias and gamma");
            EpiF32 e{(float*)d->out, d->ldc, d->bias, d->gamma, d->m, d->n};
            return dispatch_nreg(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
        }
        case CVX_EPI_PATCH: {
            EpiPatch e{(float*)d->out, d->ldc, d->bias, d->pos, d->ldpos, d->npatch, d->ntp, d->tok0, d->m, d->n};
            return dispatch_nreg(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
        }
        case CVX_EPI_VT: {
            if (d->n_pad % 128 || d->k_pad % BK) return cvx_fail("gemm: V^T epilogue needs N padded to 128, K to 64");
            if (d->ln_rowstat) {
                EpiVTT<true> e{(uint16_t*)d->out, d->bias, d->heads, d->ntp, d->kp, d->m, d->n, 0, d->ln_rowstat, d->n_pad};
                return dispatch_vt(d, A, W, e, st);
            }
            EpiVT e{(uint16_t*)d->out, d->bias, d->heads, d->ntp, d->kp, d->m, d->n};
            return dispatch_vt(d, A, W, e, st);
        }
        case CVX_EPI_CONVT: {
            if (d->cout % 8) return cvx_fail("gemm: ConvT C_out must be a multiple of 8");
            if (g_convt_small && convt_small_eligible(*d)) return convt_small_dispatch(*d, st);
            if (d->dtype == CVX_DTYPE_F16) {  // the head's activations are fp16
                if (d->act) {
                    EpiConvT<1, true> e{(uint16_t*)d->out, d->bias, d->H, d->W, d->cout, d->m, d->n, d->convt_up_z ? 1 : 0};
                    return dispatch_nreg(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
                }
                EpiConvT<0, true> e{(uint16_t*)d->out, d->bias, d->H, d->W, d->cout, d->m, d->n, d->convt_up_z ? 1 : 0};
                return dispatch_nreg(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
            }
            if (d->act) {
                EpiConvT<1> e{(uint16_t*)d->out, d->bias, d->H, d->W, d->cout, d->m, d->n, d->convt_up_z ? 1 : 0};
                return dispatch_nreg(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
            }
            EpiConvT<0> e{(uint16_t*)d->out, d->bias, d->H, d->W, d->cout, d->m, d->n, d->convt_up_z ? 1 : 0};
            return dispatch_nreg(A, d->lda, W, d->ldw, d->m, d->n_pad, d->k_pad, e, st);
        }
        default:
            return cvx_fail("gemm: unknown epilogue");
    }
}

template <int ACT>
static int conv3_dispatch(const cvx_conv3d_desc& d, hipStream_t st) {
    const long M = (long)d.D * d.H * d.W;
    EpiBF16<ACT, true> e{(uint16_t*)d.out, (long)d.cout, d.bias, M, (long)d.cout};  // head activations: fp16
    if (d.n_pad % 128 == 0) return launch_conv3<TileCfg<128, 128, 2>>(d, e, st);
    // 192 output channels (sb1): all of them in one tile, so the gathered activation tile is fetched once instead of three times
    if (d.n_pad % 192 == 0 && g_conv_wide) return launch_conv3<TileCfg<192, 128, 1>>(d, e, st);
    if (d.n_pad % 64 == 0) return launch_conv3<TileCfg<64, 256, 1>>(d, e, st);
    if (d.n_pad % 32 == 0) return launch_conv3<TileCfg<32, 256, 1>>(d, e, st);
    if (d.n_pad % 16 == 0) return launch_conv3<TileCfg<16, 256, 1>>(d, e, st);
    return cvx_fail("conv3d: C_out must be padded to a multiple of 16");
}

namespace cvx {  // conv_halo.hip: LDS-halo kernel for the full-resolution few-channel layers
bool conv3_halo_eligible(const cvx_conv3d_desc& d);
int conv3_halo_dispatch(const cvx_conv3d_desc& d, hipStream_t st);
bool conv3_march_eligible(const cvx_conv3d_desc& d);
int conv3_march_dispatch(const cvx_conv3d_desc& d, hipStream_t st);
}

template <int ACT>
static int pool2_dispatch(const cvx_conv3d_desc& d, int cshift, hipStream_t st) {
    const long M = (long)(d.D / 2) * (d.H / 2) * (d.W / 2);
    EpiBF16<ACT, true> e{(uint16_t*)d.out, (long)d.cout, d.bias, M, (long)d.cout};
    if (d.n_pad % 128 == 0) return launch_pool2<TileCfg<128, 128, 2>>(d, cshift, e, st);
    if (d.n_pad % 64 == 0) return launch_pool2<TileCfg<64, 256, 1>>(d, cshift, e, st);
    if (d.n_pad % 32 == 0) return launch_pool2<TileCfg<32, 256, 1>>(d, cshift, e, st);
    if (d.n_pad % 16 == 0) return launch_pool2<TileCfg<16, 256, 1>>(d, cshift, e, st);
    return cvx_fail("conv2s2: C_out must be padded to a multiple of 16");
}

extern "C" int cvx_conv2s2_f16(const cvx_conv3d_desc* d, hipStream_t st) {
    if (!d) return cvx_fail("conv2s2: null descriptor");
    if (!d->in || !d->w || !d->bias || !d->out || !d->zero_page) return cvx_fail("conv2s2: null pointer");
    if (d->D % 2 || d->H % 2 || d->W % 2 || d->D <= 0 || d->H <= 0 || d->W <= 0) return cvx_fail("conv2s2: D, H, W must be even");
    int cshift = 3;
    while ((1 << cshift) < d->C) ++cshift;
    if ((1 << cshift) != d->C || d->C < 8) return cvx_fail("conv2s2: C_in must be a power of two >= 8");
    if (d->k_pad != 8 * d->C) return cvx_fail("conv2s2: K must be 8*C_in");
    if (d->cout % 4) return cvx_fail("conv2s2: C_out must be a multiple of 4");
    return d->act ? pool2_dispatch<1>(*d, cshift, st) : pool2_dispatch<0>(*d, cshift, st);
}

extern "C" int cvx_conv3d_f16(const cvx_conv3d_desc* d, hipStream_t st) {
    if (!d) return cvx_fail("conv3d: null descriptor");
    if (d->C % 8) return cvx_fail("conv3d: C_in must be a multiple of 8");
    if (d->k_pad % BK || d->k_pad < 27 * d->C) return cvx_fail("conv3d: K must be 27*C_in padded to a multiple of 64");
    if (d->cout % 4) return cvx_fail("conv3d: C_out must be a multiple of 4");
    const int halo = g_conv_halo;
    if (halo == 2 && conv3_march_eligible(*d)) return conv3_march_dispatch(*d, st);
#ifdef CVX_ABLATION
    if (halo && conv3_halo_eligible(*d)) return conv3_halo_dispatch(*d, st);
#endif
    return d->act ? conv3_dispatch<1>(*d, st) : conv3_dispatch<0>(*d, st);
}
