// Last layer of the segmentation head at full resolution (HBM-bound):
//   Conv3d(C=8 -> 1, k=3, "same") -> clip(+-5) -> logits / sigmoid -> probs, fused with the masked Dice sums.
// Input is the channels-last fp16 volume [D][H][W][8] (one voxel = one 16-B load).
#include "common.h"
#include "../../include/cryovit_hip.h"
#include "host_util.h"

namespace cvx {

// Persistent-style grid: each block walks tiles of 64(x) x 4(y) voxels of one z plane.  The tile's input neighbourhood --
// 3 planes x 6 rows x 66 voxels x 16 B, the halo included -- is staged ONCE in LDS (zero-filled outside the volume) and the
// 27 taps of every output are read from there: 4.6 global reads per output instead of 27 (the un-staged version was bound by
// L2 bandwidth: 14.5 GB of cache traffic for a 537-MB input).  Dice partial sums stay in registers and leave the block as
// ONE row of `partials` (no same-address atomics: 1.5 M of them cost 19 ms on this volume).
constexpr int CO_TX = 64, CO_TY = 4, CO_HX = CO_TX + 2, CO_HY = CO_TY + 2;
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_conv3_out(const uint16_t* __restrict__ in, const float* __restrict__ w /*[27][8]*/,
                                                   float bias, float* __restrict__ logits, float* __restrict__ probs,
                                                   const int8_t* __restrict__ labels, float* __restrict__ partials,
                                                   uint8_t* __restrict__ mask, float mask_thr, int D, int H, int W, int tiles_x,
                                                   int tiles_y, long ntiles) {
    // weights as fp16 pairs (the reference runs this layer under fp16 autocast like the others): a tap is four v_dot2_f32_f16
    // (fp16 products, fp32 accumulation) instead of eight conversions + eight FMAs -- the kernel was bound by those ~430 vector
    // instructions per voxel, not by HBM (0.51 ms = 1.4 TB/s of compulsory traffic)
    __shared__ __attribute__((aligned(16))) uint32_t sw2[27 * 4];
    __shared__ float red[3][4];
    __shared__ __attribute__((aligned(16))) uint4 halo[3][CO_HY][CO_HX];  // 19 KB
    for (int i = threadIdx.x; i < 27 * 4; i += 256) sw2[i] = pack2h(w[2 * i], w[2 * i + 1]);
    float inter = 0.f, ysum = 0.f, psum = 0.f;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    // XCD-aware order (workgroups b and b + 8 share an L2): each XCD walks a contiguous run of tiles per pass, so the rows a tile's
    // neighbours staged are found in the same L2
    const long vb = (gridDim.x & 7) == 0 ? (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
    for (long tile = vb; tile < ntiles; tile += gridDim.x) {
        const int tx = (int)(tile % tiles_x);
        const long t2 = tile / tiles_x;
        const int ty = (int)(t2 % tiles_y), z = (int)(t2 / tiles_y);
        const int x0 = tx * CO_TX - 1, y0 = ty * CO_TY - 1;
        __syncthreads();  // the previous tile's reads are done (also orders the weight copy before the first use)
        // all of a thread's (at most five) halo voxels are loaded before the first is stored: load -> store pairs in a loop made
        // each iteration wait for its own round trip
        {
            constexpr int NH = 3 * CO_HY * CO_HX, NIT = (NH + 255) / 256;
            uint4 hu[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = threadIdx.x + 256 * it;
                const int ic = i < NH ? i : 0;
                const int hx = ic % CO_HX, hy = (ic / CO_HX) % CO_HY, hz = ic / (CO_HX * CO_HY);
                const int xx = x0 + hx, yy = y0 + hy, zz = z + hz - 1;
                hu[it] = uint4{0u, 0u, 0u, 0u};  // "same" padding: zeros outside the volume
                if (i < NH && (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H && (unsigned)zz < (unsigned)D)
                    hu[it] = *(const uint4*)(in + (((long)zz * H + yy) * W + xx) * 8);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = threadIdx.x + 256 * it;
                if (i < NH) ((uint4*)halo)[i] = hu[it];
            }
        }
        __syncthreads();
        const int x = tx * CO_TX + lx, y = ty * CO_TY + ly;
        if (x >= W || y >= H) continue;
        float acc = bias;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const uint4 u = halo[kz][ly + ky][lx + kx];
                    const uint4 ww = *(const uint4*)(sw2 + ((kz * 3 + ky) * 3 + kx) * 4);
                    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, u.x), __builtin_bit_cast(h16x2, ww.x), acc, false);
                    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, u.y), __builtin_bit_cast(h16x2, ww.y), acc, false);
                    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, u.z), __builtin_bit_cast(h16x2, ww.z), acc, false);
                    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2, u.w), __builtin_bit_cast(h16x2, ww.w), acc, false);
                }
        const float lg = fminf(fmaxf(acc, -5.0f), 5.0f);            // cryovit.py:39
        const float p = 1.0f / (1.0f + __expf(-lg));                 // cryovit.py:49
        const long v = ((long)z * H + y) * W + x;
        if (logits) logits[v] = lg;
        if (probs) probs[v] = p;
        if (mask) mask[v] = p >= mask_thr ? 1 : 0;                   // callbacks.py:100-102 (PredictionWriter)
        if (labels) {
            const int lab = labels[v];
            if (lab > -1) {                                          // base_model.py:99
                const float ph = p < mask_thr ? 0.f : 1.f;           // metrics.py:38 (DiceMetric.thresh: the SAME threshold as the mask)
                inter += (float)lab * ph; ysum += (float)lab; psum += ph;
            }
        }
    }
    if (labels) {
        inter = wave_sum(inter); ysum = wave_sum(ysum); psum = wave_sum(psum);
        const int wv = threadIdx.x >> 6;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { red[0][wv] = inter; red[1][wv] = ysum; red[2][wv] = psum; }
        __syncthreads();
        if (threadIdx.x < 3)
            partials[(long)blockIdx.x * 3 + threadIdx.x] =
                (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    }
}

// fixed-order (bitwise reproducible) reduction of the per-block Dice partials: dice[k] += sum_b partials[b][k]
__global__ __launch_bounds__(64) void k_dice_finalize(const float* __restrict__ partials, int nblk, float* __restrict__ dice) {
    const int lane = threadIdx.x;
    for (int k = 0; k < 3; ++k) {
        double s = 0.0;
        for (int b = lane; b < nblk; b += 64) s += (double)partials[(long)b * 3 + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) dice[k] += (float)s;
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
                                   const int8_t* labels, float* dice, float* scratch, uint8_t* mask, float mask_threshold, int D,
                                   int H, int W, hipStream_t st) {
    if (D <= 0 || H <= 0 || W <= 0) return 0;
    if (labels && (!dice || !scratch)) return cvx_fail("conv3_out: labels need a dice accumulator and a scratch buffer");
    const int tiles_x = (W + 63) / 64, tiles_y = (H + 3) / 4;
    const long ntiles = (long)tiles_x * tiles_y * D;
    const int nblk = (int)(ntiles < CVX_DICE_BLOCKS ? ntiles : CVX_DICE_BLOCKS);
    hipLaunchKernelGGL(k_conv3_out, dim3(nblk), dim3(256), 0, st, (const uint16_t*)in, w, bias, logits, probs, labels, scratch, mask,
                       mask_threshold, D, H, W, tiles_x, tiles_y, ntiles);
    int rc = cvx_check_launch();
    if (rc || !labels) return rc;
    hipLaunchKernelGGL(k_dice_finalize, dim3(1), dim3(64), 0, st, scratch, nblk, dice);
    return cvx_check_launch();
}

extern "C" int cvx_dice_sums(const float* probs, const int8_t* labels, float* dice, long n, float thr, hipStream_t st) {
    if (n <= 0) return 0;
    const unsigned nblk = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_dice, dim3(nblk), dim3(256), 0, st, probs, labels, dice, n, thr);
    return cvx_check_launch();
}
