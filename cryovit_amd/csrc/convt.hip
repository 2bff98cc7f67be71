// ConvTranspose3d kernel = stride = (1,2,2) for the head's few-channel blocks (C_in 16 / 32, C_out <= 32; cryovit.py:74-77):
// out[z][2y+i][2x+j][o] = act(b[o] + sum_c in[z][y][x][c] W[c][o][i][j]).  As a GEMM tile (K padded from 16 / 32 to 64, scattered 8-B
// stores) these layers ran at 2.1 TB/s of compulsory traffic; they are a K = 16 / 32 product with 4 C_out outputs per voxel, bound by
// the GELUs and the output stream.  Here: v_mfma_f32_16x16x16_f16 with the VOXELS as the B operand -- lane (voxel = lane % 16,
// g = lane / 16) loads channels 4g .. 4g+3 of its voxel straight from global memory (a wave reads 16 voxels = one contiguous run;
// no LDS) -- and the weight rows ordered so that lane group g ends up with ALL C_out channels of output pixel (i, j) = (g / 2, g % 2)
// of its voxel: row 4g + r of fragment a is output channel 4a + r of pixel g.  A lane then stores its pixel as whole 16-B pieces, and
// the 16 voxels x 2 pixels of an output row are one contiguous run.
#include "common.h"
#include "../../include/cryovit_hip.h"
#include "host_util.h"

namespace cvx {

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

template <int C2, int C3, int ACT>
__global__ __launch_bounds__(256) void k_convt_small(const uint16_t* __restrict__ in, const uint16_t* __restrict__ wt /*[4*C3][ldw], n = ij*C3+o*/,
                                                     long ldw, const float* __restrict__ bias /*[4*C3] (b[o] repeated)*/,
                                                     uint16_t* __restrict__ out, int H, int W, long nfrag) {
    constexpr int NA = C3 / 4, KS = C2 / 16, F = 4;
    const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    // A fragments: row (lane % 16) = 4 gg + r  ->  weight row n = gg * C3 + 4 a + r; k = 16 ks + 4 g .. +3
    h16x4 wa[NA][KS];
    {
        const int gg = li >> 2, r = li & 3;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                wa[a][ks] = *(const h16x4*)(wt + (long)(gg * C3 + 4 * a + r) * ldw + 16 * ks + 4 * g);
    }
    float bv[NA][4];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[a][r] = bias[4 * a + r];
    const int fpr = W >> 4;  // fragments per input row (W % 16 == 0)
    const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    for (long f0 = wave_id * F; f0 < nfrag; f0 += nwaves * F) {
        h16x4 xb[F][KS];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const long fr = f0 + f < nfrag ? f0 + f : nfrag - 1;  // (the clamped duplicate is computed and not stored)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xb[f][ks] = *(const h16x4*)(in + (fr * 16 + li) * C2 + 16 * ks + 4 * g);
        }
#pragma unroll
        for (int f = 0; f < F; ++f) {
            if (f0 + f >= nfrag) break;  // (uniform)
            f32x4 acc[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc[a] = __builtin_amdgcn_mfma_f32_16x16x16f16(wa[a][ks], xb[f][ks], acc[a], 0, 0, 0);
            }
            const long fr = f0 + f;
            const long rowi = fr / fpr;                 // z * H + y
            const int x = (int)(fr - rowi * fpr) * 16 + li;
            const long z = rowi / H;
            const int y = (int)(rowi - z * H);
            uint16_t* dst = out + (((z * 2 * H + 2 * y + (g >> 1)) * (2L * W)) + 2 * x + (g & 1)) * C3;
#pragma unroll
            for (int a2 = 0; a2 < NA; a2 += 2) {
                float v[8];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t = acc[a2 + h][r] + bv[a2 + h][r];
                        v[4 * h + r] = ACT == 1 ? gelu_erf(t) : t;
                    }
                *(uint4*)(dst + 4 * a2) = uint4{pack2h(v[0], v[1]), pack2h(v[2], v[3]), pack2h(v[4], v[5]), pack2h(v[6], v[7])};
            }
        }
    }
}

template <int C2, int C3>
static int launch_convt_small(const cvx_gemm_desc& d, hipStream_t st) {
    const long nfrag = d.m / 16;
    const long want = (nfrag + 15) / 16;  // 4 waves x 4 fragments per block and pass
    const unsigned nblk = (unsigned)(want < 8192 ? (want < 1 ? 1 : want) : 8192);
    auto k = d.act ? k_convt_small<C2, C3, 1> : k_convt_small<C2, C3, 0>;
    hipLaunchKernelGGL(k, dim3(nblk), dim3(256), 0, st, (const uint16_t*)d.a, (const uint16_t*)d.w, d.ldw, d.bias, (uint16_t*)d.out, d.H, d.W, nfrag);
    return cvx_check_launch();
}

// used by cvx_gemm_bf16 (gemm.hip) for the ConvT epilogue
bool convt_small_eligible(const cvx_gemm_desc& d) {
    return d.dtype == CVX_DTYPE_F16 && !d.convt_up_z && d.W % 16 == 0 && d.H > 0 && d.m % ((long)d.H * d.W) == 0 && d.n == 4L * d.cout &&
           d.ldc == d.cout && ((d.lda == 16 && d.cout == 8) || (d.lda == 32 && d.cout == 32)) && d.k_pad >= d.lda;
}
int convt_small_dispatch(const cvx_gemm_desc& d, hipStream_t st) {
    if (d.cout == 8) return launch_convt_small<16, 8>(d, st);
    return launch_convt_small<32, 32>(d, st);
}

}  // namespace cvx
