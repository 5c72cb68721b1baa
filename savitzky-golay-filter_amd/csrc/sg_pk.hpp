// sg_pk.hpp -- packed-FP32 building blocks shared by the kernels (gfx950).
//
// A wave64 VALU instruction costs 4 cycles whether it is v_fma_f32 or v_pk_fma_f32, so every fp32 kernel here keeps
// its data as pairs (two adjacent samples / columns / streams per VGPR pair) and does two multiply-adds per
// instruction.  A tap is ONE float broadcast to both halves: it is read out of an aligned SGPR (or VGPR) pair with
// op_sel, so a filter's taps occupy half as many pairs as floats.  These are inline asm because the compiler's own
// packing of such loops builds (w[k], w[k-1]) SGPR pairs instead and spills hundreds of SGPRs.
//
// Scheduling note: the hazard recogniser counts no wait states for inline asm, so an asm result consumed by the very
// next instruction costs an s_nop unless a compiler-visible instruction sits in between; callers interleave
// independent chains (or issue the producer one step ahead) accordingly.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

namespace sg {

typedef float        f32x2 __attribute__((ext_vector_type(2)));
typedef float        f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));      // the nontemporal builtins want a native vector type
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// acc += w[SEL] * x, tap pair in SGPRs
template <int SEL>
__device__ __forceinline__ void pk_fma_sgpr(f32x2 &acc, const f32x2 wpair, const f32x2 x)
{
    if constexpr (SEL == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(wpair), "v"(x));
    else                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(wpair), "v"(x));
}
// acc += w[SEL] * x, tap pair in VGPRs (taps that change per term at run time)
template <int SEL>
__device__ __forceinline__ void pk_fma_vgpr(f32x2 &acc, const f32x2 wpair, const f32x2 x)
{
    if constexpr (SEL == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(wpair), "v"(x));
    else                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(wpair), "v"(x));
}
// w[SEL] * x, rounded on its own (kernels that must reproduce the reference's separate multiply and add)
template <int SEL>
__device__ __forceinline__ f32x2 pk_mul_sgpr(const f32x2 wpair, const f32x2 x)
{
    f32x2 p;
    if constexpr (SEL == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "s"(wpair), "v"(x));
    else                    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(p) : "s"(wpair), "v"(x));
    return p;
}
// (a.y, b.x): the pair that straddles two aligned pairs -- one v_pk_mov_b32, left to the compiler (a shuffle it
// knows has no forwarding hazard, and a counted wait state between the asm instructions around it)
__device__ __forceinline__ f32x2 pk_straddle(const f32x2 a, const f32x2 b)
{
    return __builtin_shufflevector(a, b, 1, 2);
}

// lanes of one wave exchanging data through LDS: order the compiler's memory operations, nothing else (LDS
// operations of one wave execute in order; no s_barrier, no s_waitcnt is emitted for a wavefront-scope fence)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// f(integral_constant<0>) && f(<1>) && ... : a loop whose index is a literal in every iteration (register arrays
// indexed by it stay in registers); returning false breaks out
template <int... I, typename F>
__device__ __forceinline__ bool static_for(std::integer_sequence<int, I...>, F &&f)
{
    return (f(std::integral_constant<int, I>{}) && ...);
}
template <int COUNT, typename F>
__device__ __forceinline__ bool static_for(F &&f)
{
    return static_for(std::make_integer_sequence<int, COUNT>{}, static_cast<F &&>(f));
}

}  // namespace sg
