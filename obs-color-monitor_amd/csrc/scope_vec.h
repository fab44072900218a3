/*
 * scope_vec.h -- wave-level vocabulary of the kernels: a native 2 x fp32 vector for the packed CDNA4 VALU
 * ops (v_pk_mul_f32, v_pk_add_f32) and a full-wave sum.
 * (tests/emul/ provides a scalar stand-in of the same name for the CPU emulation build.)
 */
#ifndef SCOPE_VEC_H
#define SCOPE_VEC_H

typedef float scope_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ scope_f2 scope_f2_make(float a, float b)
{
	scope_f2 v;
	v.x = a;
	v.y = b;
	return v;
}
__device__ __forceinline__ float scope_f2_lo(scope_f2 v)
{
	return v.x;
}
__device__ __forceinline__ float scope_f2_hi(scope_f2 v)
{
	return v.y;
}

/* Sum over the 64 lanes of a wave; every lane must be active (uniform control flow). */
__device__ __forceinline__ unsigned scope_wave_sum(unsigned x)
{
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1)
		x += __shfl_xor(x, m);
	return x;
}

#endif
