// Last layer of the segmentation head at full resolution (HBM-bound):
//   Conv3d(C=8 -> 1, k=3, "same") -> clip(+-5) -> logits / sigmoid -> probs, fused with the masked Dice sums.
// Input is the channels-last bf16 volume [D][H][W][8] (one voxel = one 16-B load).
#include "common.h"
#include "../../include/cryovit_hip.h"
#include "host_util.h"

namespace cvx {

__global__ __launch_bounds__(256) void k_conv3_out(const uint16_t* __restrict__ in, const float* __restrict__ w /*[27][8]*/,
                                                   float bias, float* __restrict__ logits, float* __restrict__ probs,
                                                   const int8_t* __restrict__ labels, float* __restrict__ dice, int D, int H,
                                                   int W) {
    __shared__ float sw[27 * 8];
    for (int i = threadIdx.x; i < 27 * 8; i += 256) sw[i] = w[i];
    __syncthreads();
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int z = blockIdx.z;
    float inter = 0.f, ysum = 0.f, psum = 0.f;
    if (x < W && y < H) {
        float acc = bias;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) {
            const int zz = z + kz - 1;
            if ((unsigned)zz >= (unsigned)D) continue;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = y + ky - 1;
                if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = x + kx - 1;
                    if ((unsigned)xx >= (unsigned)W) continue;
                    const uint4 u = *(const uint4*)(in + (((long)zz * H + yy) * W + xx) * 8);
                    const float* ww = sw + ((kz * 3 + ky) * 3 + kx) * 8;
                    acc = fmaf(bflo(u.x), ww[0], acc); acc = fmaf(bfhi(u.x), ww[1], acc);
                    acc = fmaf(bflo(u.y), ww[2], acc); acc = fmaf(bfhi(u.y), ww[3], acc);
                    acc = fmaf(bflo(u.z), ww[4], acc); acc = fmaf(bfhi(u.z), ww[5], acc);
                    acc = fmaf(bflo(u.w), ww[6], acc); acc = fmaf(bfhi(u.w), ww[7], acc);
                }
            }
        }
        const float lg = fminf(fmaxf(acc, -5.0f), 5.0f);            // cryovit.py:39
        const float p = 1.0f / (1.0f + __expf(-lg));                 // cryovit.py:49
        const long v = ((long)z * H + y) * W + x;
        if (logits) logits[v] = lg;
        if (probs) probs[v] = p;
        if (labels) {
            const int lab = labels[v];
            if (lab > -1) {                                          // base_model.py:99
                const float ph = p < 0.5f ? 0.f : 1.f;               // metrics.py:38
                inter = (float)lab * ph; ysum = (float)lab; psum = ph;
            }
        }
    }
    if (labels) {
        inter = wave_sum(inter); ysum = wave_sum(ysum); psum = wave_sum(psum);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&dice[0], inter);
            atomicAdd(&dice[1], ysum);
            atomicAdd(&dice[2], psum);
        }
    }
}

// Standalone masked Dice sums over probs/labels (used when predictions come from elsewhere)
__global__ __launch_bounds__(256) void k_dice(const float* __restrict__ probs, const int8_t* __restrict__ labels,
                                              float* __restrict__ dice, long n, float thr) {
    float inter = 0.f, ysum = 0.f, psum = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int lab = labels[i];
        if (lab > -1) {
            const float ph = probs[i] < thr ? 0.f : 1.f;
            inter += (float)lab * ph; ysum += (float)lab; psum += ph;
        }
    }
    inter = wave_sum(inter); ysum = wave_sum(ysum); psum = wave_sum(psum);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&dice[0], inter);
        atomicAdd(&dice[1], ysum);
        atomicAdd(&dice[2], psum);
    }
}

}  // namespace cvx

using namespace cvx;

extern "C" int cvx_conv3_out_fused(const void* in, const float* w, float bias, float* logits, float* probs,
                                   const int8_t* labels, float* dice, int D, int H, int W, hipStream_t st) {
    if (D <= 0 || H <= 0 || W <= 0) return 0;
    if (labels && !dice) return cvx_fail("conv3_out: labels given without a dice accumulator");
    if (D > 65535 || (H + 3) / 4 > 65535) return cvx_fail("conv3_out: volume exceeds grid limits");
    dim3 grid((W + 63) / 64, (H + 3) / 4, D);
    hipLaunchKernelGGL(k_conv3_out, grid, dim3(256), 0, st, (const uint16_t*)in, w, bias, logits, probs, labels, dice, D, H, W);
    return cvx_check_launch();
}

extern "C" int cvx_dice_sums(const float* probs, const int8_t* labels, float* dice, long n, float thr, hipStream_t st) {
    if (n <= 0) return 0;
    const unsigned nblk = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_dice, dim3(nblk), dim3(256), 0, st, probs, labels, dice, n, thr);
    return cvx_check_launch();
}
