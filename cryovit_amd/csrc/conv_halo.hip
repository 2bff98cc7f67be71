// Dilated 3x3x3 "same" convolution for the head's full-resolution, few-channel layers (C_in 8 / 16 / 32, C_out <= 16):
// SynthesisBlock 4 at 256^2 and output_layer.0 at 512^2 (cryovit.py:26-33).  There the implicit GEMM of gemm_core.h is
// bound by the L2: its LDS-DMA gathers every one of the 27 taps of every voxel separately (14-17 GB of cache traffic per
// layer for a 0.3-0.5 GB input).  Here one workgroup stages the input neighbourhood of a 4 x 64-voxel tile -- 3 planes
// (z-d, z, z+d) x 6 rows x 66 voxels x C_in channels, zero-filled outside the volume -- ONCE in LDS, and the MFMA operand
// fragments are read straight from that halo tile: no im2col image exists, global traffic is the input (x1.55 for the halo)
// plus the output.
//   D[n][voxel] = sum_k W[n][k] * X[voxel][k],  k = tap*C_in + c     v_mfma_f32_16x16x32_f16, A = weights (16 rows = C_out
//   padded), B = 16 voxels; lane (voxel = lane%16, g = lane/16) ends up with channels 4g..4g+3 of its voxel: one 8-B store.
#include "common.h"
#include "../../include/cryovit_hip.h"
#include "host_util.h"

namespace cvx {

constexpr int CH_TX = 64, CH_TY = 4, CH_HX = CH_TX + 2, CH_HY = CH_TY + 2;

template <int CIN>
struct HaloCfg {
    static constexpr int K = 27 * CIN;
    static constexpr int KSTEPS = (K + 31) / 32;
    static constexpr int WPITCH = KSTEPS * 32 + 8;                       // fp16 elements per weight row (+8: 16 rows -> 16 distinct 16-B slots)
    // C_in = 32: the three z planes are staged and consumed one after the other (a k-step is exactly one tap, a plane 9
    // k-steps), 25 KB instead of 76 KB of halo: three workgroups per CU instead of one, so staging overlaps compute
    static constexpr bool PLANEWISE = CIN == 32;
    static constexpr int PLANES = PLANEWISE ? 1 : 3;
    static constexpr int HALO_BYTES = PLANES * CH_HY * CH_HX * CIN * 2;
    static constexpr int W_BYTES = 16 * WPITCH * 2;
    static constexpr int LDS_BYTES = HALO_BYTES + W_BYTES;
};

template <int CIN, int ACT>
__global__ __launch_bounds__(256) void k_conv3_halo(const uint16_t* __restrict__ in, const uint16_t* __restrict__ wt /*[16][ldw]*/,
                                                    long ldw, const float* __restrict__ bias, uint16_t* __restrict__ out, int cout,
                                                    int D, int H, int W, int dil, int tiles_x, int tiles_y, long ntiles) {
    using Cfg = HaloCfg<CIN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t* halo = (uint16_t*)smem;                       // [3][CH_HY][CH_HX][CIN]
    uint16_t* wl = (uint16_t*)(smem + Cfg::HALO_BYTES);     // [16][WPITCH]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;

    // weights -> LDS once per workgroup (rows >= C_out and k >= 27*C_in are zero in the packed matrix)
    for (int i = tid; i < 16 * (Cfg::KSTEPS * 4); i += 256) {
        const int n = i / (Cfg::KSTEPS * 4), c8 = i % (Cfg::KSTEPS * 4);
        uint4 u = uint4{0u, 0u, 0u, 0u};
        if (c8 * 8 < Cfg::K) u = *(const uint4*)(wt + (long)n * ldw + c8 * 8);  // K % 8 == 0
        *(uint4*)(wl + n * Cfg::WPITCH + c8 * 8) = u;
    }
    float bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = bias[4 * g + i];

    // per-lane halo offsets (elements) of the k-steps' fragments relative to the voxel's own halo position:
    // k-step s, lane group g covers k = 32 s + 8 g .. +7  ->  tap = k / CIN, first channel = k % CIN
    int foff[Cfg::KSTEPS];
#pragma unroll
    for (int s = 0; s < Cfg::KSTEPS; ++s) {
        const int k = 32 * s + 8 * g;
        int tap = k / CIN;
        const int c0 = k - tap * CIN;
        tap = tap < 27 ? tap : 26;  // padded k: the weights are zero, any finite data will do
        const int kz = tap / 9, r9 = tap - kz * 9, ky = r9 / 3, kx = r9 - ky * 3;
        foff[s] = ((kz * CH_HY + ky) * CH_HX + kx) * CIN + c0;
    }

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = (int)(tile % tiles_x);
        const long t2 = tile / tiles_x;
        const int ty = (int)(t2 % tiles_y), z = (int)(t2 / tiles_y);
        const int x0 = tx * CH_TX - 1, y0 = ty * CH_TY - 1;
        f32x4 acc[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint16_t* hrow = halo + ((long)wave * CH_HX + li) * CIN;  // halo position (ky=0, kx=0) of voxel li of fragment 0
        constexpr int PV = CIN / 8;  // 16-B pieces per voxel
#pragma unroll
        for (int pass = 0; pass < (Cfg::PLANEWISE ? 3 : 1); ++pass) {
            __syncthreads();  // the previous fragment reads are done
            for (int i = tid; i < Cfg::PLANES * CH_HY * CH_HX * PV; i += 256) {
                const int p = i % PV, vox = i / PV;
                const int hx = vox % CH_HX, hy = (vox / CH_HX) % CH_HY, hz = Cfg::PLANEWISE ? pass : vox / (CH_HX * CH_HY);
                const int xx = x0 + hx, yy = y0 + hy, zz = z + (hz - 1) * dil;
                uint4 u = uint4{0u, 0u, 0u, 0u};  // "same" padding
                if ((unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H && (unsigned)zz < (unsigned)D)
                    u = *(const uint4*)(in + (((long)zz * H + yy) * W + xx) * CIN + p * 8);
                *(uint4*)(halo + (long)vox * CIN + p * 8) = u;
            }
            __syncthreads();
            constexpr int S0 = 0, SN = Cfg::PLANEWISE ? 9 : Cfg::KSTEPS;
#pragma unroll
            for (int ss = S0; ss < SN; ++ss) {
                const int s = Cfg::PLANEWISE ? 9 * pass + ss : ss;
                const bf16x8 wf = *(const bf16x8*)(wl + li * Cfg::WPITCH + 32 * s + 8 * g);
                // plane-wise: the staged plane sits at plane index 0, the tap offsets of plane `pass` are those of plane 0
                const int fo = Cfg::PLANEWISE ? foff[ss] : foff[s];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const bf16x8 xf = *(const bf16x8*)(hrow + (long)(16 * f) * CIN + fo);
                    acc[f] = mfma16x16x32<true>(wf, xf, acc[f]);
                }
            }
        }
        const int y = ty * CH_TY + wave;
        if (y < H && 4 * g < cout) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int x = tx * CH_TX + 16 * f + li;
                if (x >= W) continue;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = acc[f][i] + bv[i];
                    v[i] = ACT == 1 ? gelu_erf(t) : t;
                }
                uint2 o;
                o.x = pack2h(v[0], v[1]);
                o.y = pack2h(v[2], v[3]);
                *(uint2*)(out + (((long)z * H + y) * W + x) * cout + 4 * g) = o;
            }
        }
    }
}

template <int CIN>
static int launch_halo(const cvx_conv3d_desc& d, hipStream_t st) {
    using Cfg = HaloCfg<CIN>;
    const int tiles_x = (d.W + CH_TX - 1) / CH_TX, tiles_y = (d.H + CH_TY - 1) / CH_TY;
    const long ntiles = (long)tiles_x * tiles_y * d.D;
    const int per_cu = Cfg::LDS_BYTES > 80 * 1024 ? 1 : (Cfg::LDS_BYTES > 40 * 1024 ? 3 : 6);
    const long want = 256L * per_cu;
    const unsigned nblk = (unsigned)(ntiles < want ? ntiles : want);
    auto k = d.act ? k_conv3_halo<CIN, 1> : k_conv3_halo<CIN, 0>;
    CVX_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
    hipLaunchKernelGGL(k, dim3(nblk), dim3(256), Cfg::LDS_BYTES, st, (const uint16_t*)d.in, (const uint16_t*)d.w, (long)d.k_pad, d.bias,
                       (uint16_t*)d.out, d.cout, d.D, d.H, d.W, d.dil, tiles_x, tiles_y, ntiles);
    return cvx_check_launch();
}

// used by cvx_conv3d_f16 (gemm.hip) for the shapes this kernel is built for
bool conv3_halo_eligible(const cvx_conv3d_desc& d) {
    return d.n_pad == 16 && d.cout % 4 == 0 && d.cout <= 16 && (d.C == 8 || d.C == 16 || d.C == 32) && d.k_pad >= 27 * d.C;
}
int conv3_halo_dispatch(const cvx_conv3d_desc& d, hipStream_t st) {
    switch (d.C) {
        case 8: return launch_halo<8>(d, st);
        case 16: return launch_halo<16>(d, st);
        default: return launch_halo<32>(d, st);
    }
}

}  // namespace cvx
