// Shared device helpers for the gfx950 (CDNA4) kernels of the CryoVIT hot path.
// Wave = 64 lanes everywhere; bf16 storage, fp32 accumulation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cvx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CVX_GLOBAL_AS __attribute__((address_space(1)))
#define CVX_LDS_AS __attribute__((address_space(3)))

#ifndef CVX_STREAM_IO
#define CVX_STREAM_IO 1
#endif
constexpr bool g_stream_io = CVX_STREAM_IO;

__device__ __forceinline__ uint16_t f2bf(float f) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 on gfx950 (RNE, NaN stays NaN)
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
// two floats -> one dword of two bf16 (lo in bits 0-15): ONE v_cvt_pk_bf16_f32.  (Written as two scalar casts OR-ed together,
// hipcc still converts pairwise but then re-packs with v_and / v_lshl / v_or_sdwa: three more VALU instructions per dword --
// half of the bf16 GEMM epilogue's vector instructions.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
// fp16 storage (the segmentation head: the reference runs it under fp16 autocast, configs `precision: "16-mixed"`).
// Conversions saturate at the largest finite half instead of producing inf.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ uint16_t f2h(float f) {
    _Float16 h = (_Float16)__builtin_amdgcn_fmed3f(f, -65504.0f, 65504.0f);
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) { return (uint32_t)f2h(lo) | ((uint32_t)f2h(hi) << 16); }
__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ float hlo(uint32_t w) { return h2f((uint16_t)(w & 0xffff)); }
__device__ __forceinline__ float hhi(uint32_t w) { return h2f((uint16_t)(w >> 16)); }
template <bool F16> __device__ __forceinline__ uint32_t pack2x(float lo, float hi) { return F16 ? pack2h(lo, hi) : pack2bf(lo, hi); }

// v_mfma_f32_16x16x32 on bf16 or fp16 operands (same register layouts; the LDS images are type-agnostic 16-bit data)
template <bool F16> __device__ __forceinline__ f32x4 mfma16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// epilogues that store fp16 (and therefore consume fp16 operands) declare `static constexpr bool F16 = true`
template <class E, class = void> struct epi_is_f16 { static constexpr bool value = false; };
template <class E> struct epi_is_f16<E, decltype((void)E::F16)> { static constexpr bool value = E::F16; };

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// branch-free erf (Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7): keeps fused epilogues straight-line
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __expf(-ax * ax);
    const float r = fmaf(-p * t, e, 1.0f);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division (~12 VALU instructions: the 64 gates per
// lane were the larger half of the SwiGLU epilogue's 1360 instructions); the result is rounded to bf16 afterwards
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// async global -> LDS, 16 B per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const CVX_GLOBAL_AS void*)gsrc, (CVX_LDS_AS void*)lds_wave_base, 16, 0, 0);
}

// The same in SADDR form: wave-uniform byte pointer (SGPR pair) + 32-bit per-lane BYTE offset, LDS destination given as a
// wave-uniform LDS byte address.  The builtin above takes a per-lane 64-bit pointer: every piece then costs a 64-bit VALU add
// (v_lshl_add_u64) right in front of it, inside the MFMA segment of the GEMM / attention loops -- measured on the 256x256 GEMM
// tile: 2810 -> 2573 cycles per K tile from this change alone.  M0 (the LDS address) is written in the SAME asm statement
// that uses it and M0 is declared clobbered: hipcc may not assume it still holds what the compiler last put there.
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(CVX_LDS_AS const char*)p; }
__device__ __forceinline__ void glds16_saddr(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
// two pieces from one base: lane offsets voff0 / voff1, LDS destinations lds_dst and lds_dst + STEP bytes
template <int STEP>
__device__ __forceinline__ void glds16_saddr2(const void* sbase, uint32_t voff0, uint32_t voff1, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\t"
                 "s_add_u32 m0, m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 ::"v"(voff0), "v"(voff1), "s"(sbase), "s"(lds_dst), "i"(STEP) : "memory", "m0");
}

// per-lane 64-bit source address (gathers whose lanes cannot share a base), same M0 discipline
__device__ __forceinline__ void glds16_vaddr(const void* gsrc, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory", "m0");
}

// Streaming (non-temporal) 16-B accesses for tensors that are written once and read once by a later kernel: the
// residual stream and the GEMM outputs are far larger than the L2, keeping them out of it leaves the L2 to the operand panels.
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__device__ __forceinline__ float4 ld_stream(const float* p) {
    const f32x4 v = g_stream_io ? __builtin_nontemporal_load((const f32x4*)p) : *(const f32x4*)p;
    return float4{v[0], v[1], v[2], v[3]};
}
// 16 B of 16-bit data, plain cache policy (tensors a following kernel reads again: the Infinity Cache may still hold them)
__device__ __forceinline__ uint4 ld_stream16(const uint16_t* p) { return *(const uint4*)p; }
__device__ __forceinline__ void st_stream(float* p, const float4& v) {
    const f32x4 t{v.x, v.y, v.z, v.w};
    if (g_stream_io) __builtin_nontemporal_store(t, (f32x4*)p); else *(f32x4*)p = t;
}
__device__ __forceinline__ void st_stream(uint16_t* p, const uint4& v) {
    const u32x4 t{v.x, v.y, v.z, v.w};
    if (g_stream_io) __builtin_nontemporal_store(t, (u32x4*)p); else *(u32x4*)p = t;
}

// 16-B stores in saddr form issued from inline asm: ONE store instruction per call, guaranteed (hipcc may split or merge the
// stores it generates itself -- a `*(uint4*)p = v` came out as two dwordx2 -- and the persistent GEMM tile counts its stores
// in hand-written vmcnt waits).  The trailing s_nop keeps the next VALU write of the data registers behind the store's read.
__device__ __forceinline__ void gst16_saddr(void* sbase, uint32_t voff, const u32x4& v) {
    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void gst16_saddr_nt(void* sbase, uint32_t voff, const u32x4& v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}

__device__ __forceinline__ void gst16_vaddr(void* p, const u32x4& v) {  // per-lane 64-bit address
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace cvx
