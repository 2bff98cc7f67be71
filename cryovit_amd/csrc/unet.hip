// Small streaming kernels of the UNet3D baseline (/root/reference/src/cryovit/models/unet3d.py):
// the channel concatenation of a synthesis block (l.64) and the 1x1x1 output layer + clip + sigmoid (l.45, 69-71, 96).
// The convolutions, the pooling convolution, the transposed convolution and InstanceNorm + GELU are the head's kernels
// (cvx_conv3d_f16, cvx_conv2s2_f16, cvx_gemm_bf16 with the ConvT epilogue, cvx_groupnorm_act_f16 with G = C).
#include "common.h"
#include "../../include/cryovit_hip.h"
#include "host_util.h"

namespace cvx {

__global__ __launch_bounds__(256) void k_concat2(const uint4* __restrict__ a, int pa, const uint4* __restrict__ b, int pb, uint4* __restrict__ out,
                                                 long npieces) {
    const int po = pa + pb;  // 16-B pieces per output voxel
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npieces; i += (long)gridDim.x * 256) {
        const long v = i / po;
        const int p = (int)(i - v * po);
        out[i] = p < pa ? a[v * pa + p] : b[v * pb + (p - pa)];
    }
}

template <int C>
__global__ __launch_bounds__(256) void k_pointwise_out(const uint16_t* __restrict__ in, const float* __restrict__ w, float bias,
                                                       float* __restrict__ logits, float* __restrict__ probs, long nvox) {
    float wr[C];
#pragma unroll
    for (int c = 0; c < C; ++c) wr[c] = w[c];
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (long)gridDim.x * 256) {
        float acc = bias;
#pragma unroll
        for (int p = 0; p < C / 8; ++p) {
            const uint4 u = *(const uint4*)(in + v * C + p * 8);
            acc = fmaf(hlo(u.x), wr[8 * p + 0], acc); acc = fmaf(hhi(u.x), wr[8 * p + 1], acc);
            acc = fmaf(hlo(u.y), wr[8 * p + 2], acc); acc = fmaf(hhi(u.y), wr[8 * p + 3], acc);
            acc = fmaf(hlo(u.z), wr[8 * p + 4], acc); acc = fmaf(hhi(u.z), wr[8 * p + 5], acc);
            acc = fmaf(hlo(u.w), wr[8 * p + 6], acc); acc = fmaf(hhi(u.w), wr[8 * p + 7], acc);
        }
        const float lg = fminf(fmaxf(acc, -5.0f), 5.0f);
        if (logits) logits[v] = lg;
        if (probs) probs[v] = 1.0f / (1.0f + __expf(-lg));
    }
}

}  // namespace cvx

using namespace cvx;

extern "C" int cvx_concat_channels_f16(const void* a, int Ca, const void* b, int Cb, void* out, long nvox, hipStream_t st) {
    if (!a || !b || !out) return cvx_fail("concat_channels: null pointer");
    if (Ca % 8 || Cb % 8 || Ca <= 0 || Cb <= 0) return cvx_fail("concat_channels: channel counts must be positive multiples of 8");
    if (nvox <= 0) return 0;
    const long np = nvox * ((Ca + Cb) / 8);
    const long nb = (np + 2047) / 2048;
    hipLaunchKernelGGL(k_concat2, dim3((unsigned)(nb < 65536 ? nb : 65536)), dim3(256), 0, st, (const uint4*)a, Ca / 8, (const uint4*)b, Cb / 8,
                       (uint4*)out, np);
    return cvx_check_launch();
}

extern "C" int cvx_pointwise_out_f16(const void* in, const float* w, float bias, float* logits, float* probs, long nvox, int C, hipStream_t st) {
    if (!in || !w) return cvx_fail("pointwise_out: null pointer");
    if (nvox <= 0) return 0;
    const long nb = (nvox + 1023) / 1024;
    const dim3 grid((unsigned)(nb < 65536 ? nb : 65536));
    switch (C) {
        case 8: hipLaunchKernelGGL(k_pointwise_out<8>, grid, dim3(256), 0, st, (const uint16_t*)in, w, bias, logits, probs, nvox); break;
        case 16: hipLaunchKernelGGL(k_pointwise_out<16>, grid, dim3(256), 0, st, (const uint16_t*)in, w, bias, logits, probs, nvox); break;
        case 32: hipLaunchKernelGGL(k_pointwise_out<32>, grid, dim3(256), 0, st, (const uint16_t*)in, w, bias, logits, probs, nvox); break;
        case 64: hipLaunchKernelGGL(k_pointwise_out<64>, grid, dim3(256), 0, st, (const uint16_t*)in, w, bias, logits, probs, nvox); break;
        default: return cvx_fail("pointwise_out: C must be 8, 16, 32 or 64");
    }
    return cvx_check_launch();
}
