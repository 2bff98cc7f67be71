// ViT input stage: tomogram slices -> bf16 patch matrix (fused uint8 scale, edge pad, bicubic x14/16, im2col),
// and the non-patch rows of the token stream.
#include "common.h"
#include "../../include/cryovit_hip.h"
#include "host_util.h"

namespace cvx {

// Keys cubic-convolution taps, A = -0.75 (PyTorch bicubic), for fractional offset t in [0,1)
__device__ __forceinline__ void cubic_taps(float t, float (&w)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = 2.0f - t;
    w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    w[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
    w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

// out[m][k], m = (slice*hp + ph)*wp + pw, k = py*14 + px  (k >= 196 zero).  One thread = 8 consecutive k.
// Source coordinate as torch: s = scale*(dst+0.5)-0.5 with scale = (float)(1/0.875); taps floor(s)-1..+2 clamped
// to the (edge-padded) image, which equals clamping to the raw image because the padding replicates the border.
template <bool U8>
__global__ __launch_bounds__(256) void k_preprocess(const void* __restrict__ src, int H, int W, int hp, int wp, int k_pad,
                                                    long total_chunks, uint16_t* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total_chunks) return;
    const int kc = k_pad >> 3;
    const long m = idx / kc;
    const int k0 = (int)(idx - m * kc) * 8;
    const int pw = (int)(m % wp);
    const long t1 = m / wp;
    const int ph = (int)(t1 % hp);
    const long slice = t1 / hp;
    const float scale = (float)(1.0 / 0.875);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        if (k >= 196) { v[e] = 0.f; continue; }
        const int py = k / 14, px = k - py * 14;
        const int oy = ph * 14 + py, ox = pw * 14 + px;
        const float sy = scale * ((float)oy + 0.5f) - 0.5f, sx = scale * ((float)ox + 0.5f) - 0.5f;
        const float fy = floorf(sy), fx = floorf(sx);
        float wy[4], wx[4];
        cubic_taps(sy - fy, wy);
        cubic_taps(sx - fx, wx);
        const int iy = (int)fy, ix = (int)fx;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(iy - 1 + a, 0), H - 1);
            float rowv = 0.f;
#pragma unroll
            for (int bq = 0; bq < 4; ++bq) {
                const int xx = min(max(ix - 1 + bq, 0), W - 1);
                const long off = (slice * H + yy) * (long)W + xx;
                float px_v;
                if constexpr (U8) px_v = (float)((const uint8_t*)src)[off] / 255.0f;
                else px_v = ((const float*)src)[off];
                rowv += wx[bq] * px_v;
            }
            acc += wy[a] * rowv;
        }
        v[e] = acc;
    }
    uint4 o;
    o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
    *(uint4*)(out + m * k_pad + k0) = o;
}

// out[m][k], k = c*196 + py*14 + px, from x fp32 [b][3][Hi][Wi]; one thread = 8 consecutive k
__global__ __launch_bounds__(256) void k_im2col(const float* __restrict__ x, int Hi, int Wi, int hp, int wp, int k_pad,
                                                long total_chunks, uint16_t* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total_chunks) return;
    const int kc = k_pad >> 3;
    const long m = idx / kc;
    const int k0 = (int)(idx - m * kc) * 8;
    const int pw = (int)(m % wp);
    const long t1 = m / wp;
    const int ph = (int)(t1 % hp);
    const long slice = t1 / hp;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        if (k >= 588) { v[e] = 0.f; continue; }
        const int c = k / 196, r = k - c * 196, py = r / 14, px = r - py * 14;
        v[e] = x[((slice * 3 + c) * Hi + ph * 14 + py) * (long)Wi + pw * 14 + px];
    }
    uint4 o;
    o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
    *(uint4*)(out + m * k_pad + k0) = o;
}

__global__ __launch_bounds__(256) void k_init_tokens(float* __restrict__ x, long ldx, const float* __restrict__ cls_pos0,
                                                     const float* __restrict__ reg, int n_reg, int ntok, int ntp, int C) {
    const int slice = blockIdx.x;
    const int nspecial = 1 + n_reg, npad = ntp - ntok;
    for (long i = threadIdx.x; i < (long)(nspecial + npad) * C; i += 256) {
        const int rr = (int)(i / C), c = (int)(i - (long)rr * C);
        float val;
        int row;
        if (rr == 0) { val = cls_pos0[c]; row = 0; }
        else if (rr < nspecial) { val = reg[(rr - 1) * C + c]; row = rr; }
        else { val = 0.f; row = ntok + (rr - nspecial); }
        x[((long)slice * ntp + row) * ldx + c] = val;
    }
}

}  // namespace cvx

using namespace cvx;

extern "C" int cvx_preprocess_patches(const void* slices, int is_u8, int b, int H, int W, void* out, int k_pad,
                                      hipStream_t st) {
    if (b <= 0) return 0;
    if (k_pad < 196 || k_pad % 64) return cvx_fail("preprocess: k_pad must be a multiple of 64 >= 196");
    const int Hp = (H + 15) / 16 * 16, Wp = (W + 15) / 16 * 16;
    const int hp = Hp / 16, wp = Wp / 16;  // (Hp*14/16)/14
    const long total = (long)b * hp * wp * (k_pad / 8);
    const unsigned nblk = (unsigned)((total + 255) / 256);
    if (is_u8)
        hipLaunchKernelGGL(k_preprocess<true>, dim3(nblk), dim3(256), 0, st, slices, H, W, hp, wp, k_pad, total, (uint16_t*)out);
    else
        hipLaunchKernelGGL(k_preprocess<false>, dim3(nblk), dim3(256), 0, st, slices, H, W, hp, wp, k_pad, total, (uint16_t*)out);
    return cvx_check_launch();
}

extern "C" int cvx_im2col_patches(const float* x, int b, int Hi, int Wi, void* out, int k_pad, hipStream_t st) {
    if (b <= 0) return 0;
    if (Hi % 14 || Wi % 14) return cvx_fail("im2col: image size must be a multiple of 14");
    if (k_pad < 588 || k_pad % 64) return cvx_fail("im2col: k_pad must be a multiple of 64 >= 588");
    const int hp = Hi / 14, wp = Wi / 14;
    const long total = (long)b * hp * wp * (k_pad / 8);
    hipLaunchKernelGGL(k_im2col, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, Hi, Wi, hp, wp, k_pad, total,
                       (uint16_t*)out);
    return cvx_check_launch();
}

extern "C" int cvx_init_tokens(float* x, long ldx, const float* cls_pos0, const float* reg, int n_reg, int slices, int ntok,
                               int ntp, int C, hipStream_t st) {
    if (slices <= 0) return 0;
    hipLaunchKernelGGL(k_init_tokens, dim3(slices), dim3(256), 0, st, x, ldx, cls_pos0, reg, n_reg, ntok, ntp, C);
    return cvx_check_launch();
}
