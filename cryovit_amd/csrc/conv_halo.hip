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

#ifdef CVX_ABLATION  // k_conv3_halo: the round-1 per-tile form (A/B runs only; the product runs k_conv3_march)
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
#endif  // CVX_ABLATION

// ---------------------------------------------------------------------------------------------------
// Z-MARCHING form (round 2).  The tile kernel above stages all three planes of every tile from scratch, synchronously: a
// 4.6x read amplification out of the L2 and no overlap inside a workgroup.  Here a workgroup owns one 4 x 64-voxel COLUMN and
// one residue class of z (z = r, r + dil, r + 2 dil, ...: along that sequence a (dil,1,1)-dilated stencil is a plain 3-plane
// stencil), keeps the planes z - dil, z, z + dil in an LDS ring of four slots and fetches ONE new plane per step by LDS-DMA
// while the current step computes: 1.55x instead of 4.6x L2 reads per voxel, one barrier per step, the fetch a whole step ahead.
// Out-of-volume voxels (the "same" padding) are fetched from the zero page, like every other voxel: no special cases in the
// ring.  C_out up to 32 (NF = 2 weight fragments): SynthesisBlock 3's 32 -> 32 convolutions run here too.  With C_out <= 8 only
// lane groups 0 and 1 of the accumulator layout carry channels: fragments 2 / 3 move to lanes 32-63 (v_permlane32_swap) so all
// 64 lanes evaluate GELUs (the epilogue, not the MFMAs, is the larger part of the 8-channel layer).
// ---------------------------------------------------------------------------------------------------
template <int CIN, int NF>
struct MarchCfg {
    static constexpr int K = 27 * CIN;
    static constexpr int KSTEPS = (K + 31) / 32;
    static constexpr int WPITCH = KSTEPS * 32 + 8;
    static constexpr int PV = CIN / 8;                                  // 16-B pieces per voxel
    static constexpr int NP = CH_HY * CH_HX * PV;                       // pieces per plane
    static constexpr int NWI = (NP + 63) / 64;                          // wave-instructions per plane (1 KiB each)
    static constexpr int NQ = (NWI + 3) / 4;                            // ... per wave
    static constexpr int SLOT_BYTES = NWI * 1024;
    static constexpr int W_BYTES = 16 * NF * WPITCH * 2;
    static constexpr int LDS_BYTES = 4 * SLOT_BYTES + W_BYTES;
};

// C_in = 32: a voxel is 64 B, so the four 16-lane groups of a ds_read_b128 (16 consecutive voxels, lane group g reads 16-B
// chunk g) would touch only a quarter of the banks, 2 .. 4 lanes each.  The ring therefore holds chunk c of halo voxel v at chunk
// position c ^ 2 * ((v >> 2) & 1) -- applied on the SOURCE side of the LDS-DMA, whose LDS image is lane-linear: for both lane
// patterns of a group ({0-3, 12-15} with one g and {4-11} with g ^ 1) and any alignment of the 16 voxels the 16 reads then hit
// 16 distinct 16-B bank slots.
__device__ __forceinline__ int ring_swizzle(int v) { return ((v >> 2) & 1) << 1; }

template <int CIN, int NF, int ACT>
__global__ __launch_bounds__(256) void k_conv3_march(const uint16_t* __restrict__ in, const uint16_t* __restrict__ wt /*[16 NF][ldw]*/,
                                                     long ldw, const float* __restrict__ bias, const void* __restrict__ zero_page,
                                                     uint16_t* __restrict__ out, int cout, int D, int H, int W, int dil, int tiles_x,
                                                     int tiles_y, int nitems) {
    using Cfg = MarchCfg<CIN, NF>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                                             // 4 x [CH_HY][CH_HX][CIN] fp16 (+ the last wave-instruction's tail)
    uint16_t* wl = (uint16_t*)(smem + 4 * Cfg::SLOT_BYTES);        // [16 NF][WPITCH]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;

    for (int i = tid; i < 16 * NF * (Cfg::KSTEPS * 4); i += 256) {
        const int n = i / (Cfg::KSTEPS * 4), c8 = i % (Cfg::KSTEPS * 4);
        uint4 u = uint4{0u, 0u, 0u, 0u};
        if (c8 * 8 < Cfg::K) u = *(const uint4*)(wt + (long)n * ldw + c8 * 8);
        *(uint4*)(wl + n * Cfg::WPITCH + c8 * 8) = u;
    }
    const bool swap8 = cout <= 8;           // (uniform) fragments 2 / 3 are finished by lanes 32-63
    const int ge = swap8 ? (g & 1) : g;     // lane group whose channels this lane finishes
    float bv[NF][4];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[nf][i] = bias[16 * nf + 4 * ge + i];

    // k-step s, lane group g covers k = 32 s + 8 g .. +7 -> tap = k / CIN: plane kz of the ring + an in-plane byte offset
    int foff[Cfg::KSTEPS], fkz[Cfg::KSTEPS];
#pragma unroll
    for (int s = 0; s < Cfg::KSTEPS; ++s) {
        const int k = 32 * s + 8 * g;
        int tap = k / CIN;
        const int c0 = k - tap * CIN;
        tap = tap < 27 ? tap : 26;  // padded k: zero weights
        const int kz = tap / 9, r9 = tap - kz * 9, ky = r9 / 3, kx = r9 - ky * 3;
        fkz[s] = kz;
        const int hv = (wave + ky) * CH_HX + li + kx;  // halo voxel of fragment 0 (fragment f: + 16 f, same swizzle)
        foff[s] = (hv * CIN + (CIN == 32 ? c0 ^ ring_swizzle(hv) * 8 : c0)) * 2;
    }
    const char* zp = (const char*)zero_page;
    __syncthreads();

    // XCD-aware order: workgroups b and b + 8 share an XCD (and its L2), so each XCD takes a contiguous run of columns -- neighbouring
    // columns read each other's halo rows / voxels from the same L2 instead of eight different ones
    const int vb = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    for (int item = vb; item < nitems; item += gridDim.x) {
        const int r = item % dil, col = item / dil;
        const int tx = col % tiles_x, ty = col / tiles_x;
        const int x0 = tx * CH_TX - 1, y0 = ty * CH_TY - 1;
        const int J = (D - r + dil - 1) / dil;  // z_j = r + j dil, j < J
        // this lane's pieces of a plane: wave-instruction q*4 + wave, piece index i -> (voxel, 16-B part)
        uint32_t poff[Cfg::NQ];
        uint32_t pvalid = 0;
#pragma unroll
        for (int q = 0; q < Cfg::NQ; ++q) {
            const int i = (q * 4 + wave) * 64 + lane;
            const int vox = i / Cfg::PV, p = i - vox * Cfg::PV;
            const int hy = vox / CH_HX, hx = vox - hy * CH_HX;
            const int xx = x0 + hx, yy = y0 + hy;
            const bool ok = i < Cfg::NP && (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H;
            const int ps = CIN == 32 ? p ^ ring_swizzle(vox) : p;  // the ring holds chunk ps of the voxel at chunk position p
            poff[q] = ok ? (uint32_t)(((yy * W + xx) * CIN + ps * 8) * 2) : 0u;
            pvalid |= ok ? (1u << q) : 0u;
        }
        auto fetch = [&](int jp) {  // plane index jp (z = r + jp dil; -1 and J are padding planes) -> ring slot (jp + 1) & 3
            const int zz = r + jp * dil;
            const bool zok = jp >= 0 && zz < D;
            const char* base = (const char*)(in + (long)(zok ? zz : 0) * H * W * CIN);
            const uint32_t dst = lds_addr(ring) + (uint32_t)((jp + 1) & 3) * Cfg::SLOT_BYTES;
#pragma unroll
            for (int q = 0; q < Cfg::NQ; ++q) {
                if ((q * 4 + wave) >= Cfg::NWI) break;  // (uniform)
                const char* src = (zok && (pvalid >> q & 1)) ? base + poff[q] : zp;
                glds16_vaddr(src, dst + (uint32_t)(q * 4 + wave) * 1024u);
            }
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // the previous column's last fragment reads are done
        fetch(-1);
        fetch(0);
        fetch(1);
        const int y = ty * CH_TY + wave;
        // A step's results are stored at the START of the next step, in front of that step's plane fetch: issued at the end of
        // their own step they would be the youngest entries of the in-order vmcnt queue, and the wait for the fetched plane at
        // the top of the next step would be a wait for their write acknowledgements (measured: 7 k instead of 3.5 k cycles / step).
        uint2 pend[NF][4];
        auto flush = [&](int z) {
            uint16_t* orow = out + ((long)z * H + y) * W * cout;
            if (y >= H) return;
            if (swap8) {
                if (4 * ge >= cout) return;
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int x = tx * CH_TX + 16 * (f + 2 * (lane >> 5)) + li;
                    if (x < W) *(uint2*)(orow + (long)x * cout + 4 * ge) = pend[0][f];
                }
                return;
            }
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                if (16 * nf + 4 * g >= cout) continue;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int x = tx * CH_TX + 16 * f + li;
                    if (x < W) *(uint2*)(orow + (long)x * cout + 16 * nf + 4 * g) = pend[nf][f];
                }
            }
        };
        for (int j = 0; j < J; ++j) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // planes j-1, j, j+1 have landed (step j-2's stores retired long ago)
            __builtin_amdgcn_s_barrier();                     // ... for every wave; step j-1's reads of slot (j+3)&3 are done
            if (j > 0) flush(r + (j - 1) * dil);
            if (j + 2 <= J) fetch(j + 2);
            f32x4 acc[NF][4];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[nf][f] = f32x4{0.f, 0.f, 0.f, 0.f};
            // Fragment reads run PF k-steps ahead of the MFMAs that consume them (hipcc's own schedule kept one read in flight
            // and waited for it in front of every MFMA: at one wave per SIMD the LDS latency was exposed 108 times per step)
            constexpr int PF = CIN == 8 ? 1 : 3, NB = PF + 1;  // (C_in 8: four workgroups per CU hide it; more registers would cost one)
            bf16x8 wf[NB][NF], xf[NB][4];
            auto read_frags = [&](int s) {
                const char* pl = ring + ((j + fkz[s]) & 3) * Cfg::SLOT_BYTES + foff[s];
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) wf[s % NB][nf] = *(const bf16x8*)(wl + (16 * nf + li) * Cfg::WPITCH + 32 * s + 8 * g);
#pragma unroll
                for (int f = 0; f < 4; ++f) xf[s % NB][f] = *(const bf16x8*)(pl + 16 * f * CIN * 2);
            };
#pragma unroll
            for (int s = 0; s < PF; ++s) read_frags(s);
#pragma unroll
            for (int s = 0; s < Cfg::KSTEPS; ++s) {
                if (s + PF < Cfg::KSTEPS) read_frags(s + PF);
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) acc[nf][f] = mfma16x16x32<true>(wf[s % NB][nf], xf[s % NB][f], acc[nf][f]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (swap8) {
                // lanes 32-63 (groups 2, 3: padding rows of the 16-row weight fragment) take fragments 2 / 3 of lanes 0-31
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0][0][i]), __float_as_uint(acc[0][2][i]), false, false);
                    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0][1][i]), __float_as_uint(acc[0][3][i]), false, false);
                    acc[0][0][i] = __uint_as_float(a[0]);
                    acc[0][1][i] = __uint_as_float(b[0]);
                }
            }
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    if (swap8 && f >= 2) continue;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float t = acc[nf][f][i] + bv[nf][i];
                        v[i] = ACT == 1 ? gelu_erf(t) : t;
                    }
                    pend[nf][f] = uint2{pack2h(v[0], v[1]), pack2h(v[2], v[3])};
                }
        }
        if (J > 0) flush(r + (J - 1) * dil);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (no DMA may be in flight when the workgroup releases its LDS)
}

template <int CIN, int NF>
static int launch_march(const cvx_conv3d_desc& d, hipStream_t st) {
    using Cfg = MarchCfg<CIN, NF>;
    static_assert(Cfg::LDS_BYTES <= 160 * 1024, "ring + weights must fit the CU's LDS");
    const int tiles_x = (d.W + CH_TX - 1) / CH_TX, tiles_y = (d.H + CH_TY - 1) / CH_TY;
    const int dil = d.dil < d.D ? d.dil : d.D;  // (dil >= D: every z is its own sequence of one plane; the neighbours are padding)
    const long nitems = (long)tiles_x * tiles_y * dil;
    if (nitems > 0x7fffffffL) return cvx_fail("conv3d: volume too large for the marching kernel");
    const int per_cu = (160 * 1024) / Cfg::LDS_BYTES;
    const long want = 256L * (per_cu > 8 ? 8 : per_cu);
    const unsigned nblk = (unsigned)(nitems < want ? nitems : want);
    auto k = d.act ? k_conv3_march<CIN, NF, 1> : k_conv3_march<CIN, NF, 0>;
    CVX_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
    hipLaunchKernelGGL(k, dim3(nblk), dim3(256), Cfg::LDS_BYTES, st, (const uint16_t*)d.in, (const uint16_t*)d.w, (long)d.k_pad, d.bias,
                       d.zero_page, (uint16_t*)d.out, d.cout, d.D, d.H, d.W, dil, tiles_x, tiles_y, (int)nitems);
    return cvx_check_launch();
}

bool conv3_march_eligible(const cvx_conv3d_desc& d) {
    return (d.n_pad == 16 || d.n_pad == 32) && d.cout % 4 == 0 && d.cout <= d.n_pad && (d.C == 8 || d.C == 16 || d.C == 32) &&
           d.k_pad >= 27 * d.C && d.zero_page != nullptr && (long)d.H * d.W * d.C * 2 < (1L << 31);
}
int conv3_march_dispatch(const cvx_conv3d_desc& d, hipStream_t st) {
    if (d.n_pad == 32) {
        switch (d.C) {
            case 8: return launch_march<8, 2>(d, st);
            case 16: return launch_march<16, 2>(d, st);
            default: return launch_march<32, 2>(d, st);
        }
    }
    switch (d.C) {
        case 8: return launch_march<8, 1>(d, st);
        case 16: return launch_march<16, 1>(d, st);
        default: return launch_march<32, 1>(d, st);
    }
}

// used by cvx_conv3d_f16 (gemm.hip) for the shapes this kernel is built for
#ifdef CVX_ABLATION  // the round-1 per-tile halo kernel: superseded by the z-marching ring, kept for A/B runs
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
#endif  // CVX_ABLATION

}  // namespace cvx
