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

/* Load of a V at `base + lane_off`, where `base` is wave-uniform: the base is pinned into SGPRs and the access is
 * typed as global memory, so the load uses the scalar-base + 32-bit lane-offset addressing form and no 64-bit
 * per-lane address arithmetic is needed.  NT = streaming (non-temporal) load. */
template <bool NT, typename V> __device__ __forceinline__ V scope_load_base(const unsigned char *base, unsigned lane_off)
{
	typedef const __attribute__((address_space(1))) unsigned char *gptr;
	const unsigned long long a = (unsigned long long)base;
	const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
	const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
	gptr b = (gptr)(((unsigned long long)hi << 32) | lo);
	const __attribute__((address_space(1))) V *p = (const __attribute__((address_space(1))) V *)(b + lane_off);
	if (NT)
		return __builtin_nontemporal_load(p);
	return *p;
}

/* Streaming (non-temporal) load of one dword at `row + lane_off`, where `row` is wave-uniform.  The row
 * pointer is pinned into SGPRs and the access is typed as global memory so that the load uses the
 * scalar-base + 32-bit lane-offset addressing form: no 64-bit per-lane address arithmetic. */
__device__ __forceinline__ unsigned scope_load_row_nt(const unsigned char *row, unsigned lane_off)
{
	typedef const __attribute__((address_space(1))) unsigned char *gptr;
	const unsigned long long a = (unsigned long long)row;
	const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
	const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
	gptr base = (gptr)(((unsigned long long)hi << 32) | lo);
	return __builtin_nontemporal_load((const __attribute__((address_space(1))) unsigned *)(base + lane_off));
}

/* The same for two adjacent texels (8 bytes, 8-byte aligned). */
typedef unsigned scope_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ scope_u2 scope_load_row2_nt(const unsigned char *row, unsigned lane_off)
{
	typedef const __attribute__((address_space(1))) unsigned char *gptr;
	const unsigned long long a = (unsigned long long)row;
	const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
	const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
	gptr base = (gptr)(((unsigned long long)hi << 32) | lo);
	return __builtin_nontemporal_load((const __attribute__((address_space(1))) scope_u2 *)(base + lane_off));
}

/* ... and the same as a plain (cacheable) load: lines a partner workgroup on the same XCD will want stay in the L2. */
__device__ __forceinline__ scope_u2 scope_load_row2(const unsigned char *row, unsigned lane_off)
{
	typedef const __attribute__((address_space(1))) unsigned char *gptr;
	const unsigned long long a = (unsigned long long)row;
	const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
	const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
	gptr base = (gptr)(((unsigned long long)hi << 32) | lo);
	return *(const __attribute__((address_space(1))) scope_u2 *)(base + lane_off);
}

/* Buffer addressing: a 128-bit resource descriptor in SGPRs (base, size) and ONE 32-bit byte offset per lane -- no
 * 64-bit address arithmetic at all, and the hardware's own range check: a load whose offset lies at or beyond `bytes`
 * returns zeros instead of faulting, so lanes past the end of a plane need no offset select (only their results are
 * ignored).  gfx9-family descriptor: dword 3 = 0x00020000 (raw buffer, 32-bit data format); stride 0.
 * scope_buf_load2: 8 bytes (two texels) at base + off; NT = streaming. */
typedef int scope_i4 __attribute__((ext_vector_type(4)));
typedef int scope_i2 __attribute__((ext_vector_type(2)));
typedef scope_i4 scope_buf;
__device__ scope_i2 scope_llvm_raw_buffer_load_v2i32(scope_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2i32");
__device__ __forceinline__ scope_buf scope_buf_make(const unsigned char *base, unsigned bytes)
{
	const unsigned long long a = (unsigned long long)base;
	scope_i4 r;
	r.x = (int)__builtin_amdgcn_readfirstlane((unsigned)a);
	r.y = (int)(__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu);
	r.z = (int)__builtin_amdgcn_readfirstlane(bytes);
	r.w = 0x00020000;
	return r;
}
template <bool NT> __device__ __forceinline__ scope_u2 scope_buf_load2(scope_buf buf, unsigned off)
{
	const scope_i2 t = scope_llvm_raw_buffer_load_v2i32(buf, (int)off, 0, NT ? 2 : 0); /* aux bit 1 = nt */
	scope_u2 r;
	r.x = (unsigned)t.x;
	r.y = (unsigned)t.y;
	return r;
}

/* The in-launch hand-off of partial tables between workgroups (vss_fold, scope_device.h): WRITE-THROUGH (sc1) 16-byte and 1-byte
 * stores and sc1 loads through a buffer resource -- MI355X_MICROARCH.md, inter-workgroup visibility: "16 B sc1 stores AND sc1
 * loads" need no release / acquire fence (the per-XCD L2s are not coherent with each other; an sc1 store leaves the L2 for
 * memory and drops the line, an sc1 load bypasses the CU's L1).  aux bit 4 = sc1 on gfx94x / gfx950. */
__device__ void scope_llvm_raw_buffer_store_v4i32(scope_i4 data, scope_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4i32");
__device__ scope_i4 scope_llvm_raw_buffer_load_v4i32(scope_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
__device__ void scope_llvm_raw_buffer_store_i8(char data, scope_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.i8");
__device__ char scope_llvm_raw_buffer_load_i8(scope_i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i8");
typedef unsigned scope_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void scope_buf_store16_sc1(scope_buf buf, unsigned off, scope_u4 v)
{
	scope_i4 d;
	d.x = (int)v.x, d.y = (int)v.y, d.z = (int)v.z, d.w = (int)v.w;
	scope_llvm_raw_buffer_store_v4i32(d, buf, (int)off, 0, 16);
}
__device__ __forceinline__ scope_u4 scope_buf_load16_sc1(scope_buf buf, unsigned off)
{
	const scope_i4 d = scope_llvm_raw_buffer_load_v4i32(buf, (int)off, 0, 16);
	scope_u4 v;
	v.x = (unsigned)d.x, v.y = (unsigned)d.y, v.z = (unsigned)d.z, v.w = (unsigned)d.w;
	return v;
}
__device__ __forceinline__ void scope_buf_store1_sc1(scope_buf buf, unsigned off, unsigned byte)
{
	scope_llvm_raw_buffer_store_i8((char)byte, buf, (int)off, 0, 16);
}
__device__ __forceinline__ unsigned scope_buf_load1_sc1(scope_buf buf, unsigned off)
{
	return (unsigned)(unsigned char)scope_llvm_raw_buffer_load_i8(buf, (int)off, 0, 16);
}

/* v_perm_b32: byte i of the result is byte sel[i] of the 8 bytes {hi (4..7), lo (0..3)}; 0x0c gives 0x00, 0x0d gives 0xff. */
__device__ __forceinline__ unsigned scope_perm(unsigned hi, unsigned lo, unsigned sel)
{
	return __builtin_amdgcn_perm(hi, lo, sel);
}

/* v_pk_min_u16: the two 16-bit halves independently. */
__device__ __forceinline__ unsigned scope_pk_min_u16(unsigned a, unsigned b)
{
	typedef unsigned short scope_us2 __attribute__((ext_vector_type(2)));
	const scope_us2 r = __builtin_elementwise_min(__builtin_bit_cast(scope_us2, a), __builtin_bit_cast(scope_us2, b));
	return __builtin_bit_cast(unsigned, r);
}

/* v_bitop3_b32: any bitwise function of three operands; bit (a << 2 | b << 1 | c) of TT is the result for the bits a, b, c. */
template <unsigned TT> __device__ __forceinline__ unsigned scope_bitop3(unsigned a, unsigned b, unsigned c)
{
	return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
}

/* ds_add_u32 of n at byte offset `byte` of the kernel's dynamic LDS block.  The address is built from the offset alone:
 * through a (tab + offset) pointer hipcc adds the block's LDS address -- the literal 0, resolved too late to fold --
 * with a v_add_u32 in front of every atomic, one VALU instruction per pixel in the vectorscope.  Only for kernels whose
 * dynamic block starts at LDS address 0, i.e. that declare no static __shared__ (scope_lds_is_at_zero traps otherwise). */
typedef __attribute__((address_space(3))) unsigned scope_lds_word;
__device__ __forceinline__ void scope_lds_add_at(unsigned *tab, unsigned byte, unsigned n)
{
	(void)tab;
	__hip_atomic_fetch_add((scope_lds_word *)(__UINTPTR_TYPE__)byte, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void scope_lds_is_at_zero(unsigned *tab)
{
	if ((unsigned)(__UINTPTR_TYPE__)(scope_lds_word *)tab != 0u)
		__builtin_trap();
}

/* Byte B of x times a 24-bit factor, in ONE instruction: the SDWA form of v_mul_*_u24/i24 takes the byte straight out of
 * the register.  hipcc finds this form for some shapes of the surrounding code and not for others (then it is a shift,
 * a mask and a multiply: +4 VALU instructions per texel in the fused kernels), hence the inline asm.  The factor is
 * wave-uniform (a kernel argument or a constant): SDWA encodes no literal, so it travels in an SGPR. */
#define SCOPE_MUL24_BYTE(name, insn, type, sel)                                                                             \
	__device__ __forceinline__ type name(unsigned x, type factor)                                                      \
	{                                                                                                                    \
		type r;                                                                                                      \
		asm(insn " %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" sel " src1_sel:DWORD"                    \
		    : "=v"(r)                                                                                                \
		    : "v"(x), "s"(factor));                                                                                  \
		return r;                                                                                                    \
	}
SCOPE_MUL24_BYTE(scope_mul24_byte1, "v_mul_i32_i24_sdwa", int, "BYTE_1")
SCOPE_MUL24_BYTE(scope_mul24_byte2, "v_mul_i32_i24_sdwa", int, "BYTE_2")
SCOPE_MUL24_BYTE(scope_umul24_byte1, "v_mul_u32_u24_sdwa", unsigned, "BYTE_1")
SCOPE_MUL24_BYTE(scope_umul24_byte2, "v_mul_u32_u24_sdwa", unsigned, "BYTE_2")
#undef SCOPE_MUL24_BYTE

/* Two 16-bit lanes per register (VOP3P): what the vectorscope's pair addressing (vss_add_pair) is made of. */
typedef unsigned short scope_us2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned scope_pk_lshr_u16(unsigned a, unsigned n)
{
	const scope_us2v r = __builtin_bit_cast(scope_us2v, a) >> (unsigned short)n; /* v_pk_lshrrev_b16 */
	return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ unsigned scope_pk_mad_u16(unsigned a, unsigned b, unsigned c)
{
	unsigned r;
	asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c)); /* a * b + c in each half, mod 2^16 */
	return r;
}
/* v_dot4_u32_u8: a.b0 * b.b0 + a.b1 * b.b1 + a.b2 * b.b2 + a.b3 * b.b3 + c.  With b = a factor in ONE byte it is "that byte of a,
 * times the factor, plus c" in one instruction -- a table address from a texel's channel (as a shift and a mask it is two). */
__device__ __forceinline__ unsigned scope_udot4(unsigned a, unsigned b, unsigned c)
{
	return __builtin_amdgcn_udot4(a, b, c, false);
}
/* (half H of x) << n and value << (byte B of count, its low five bits): the SDWA forms pick the half / the byte out of
 * the register, one instruction each (as a shift and a mask they are two). */
#define SCOPE_SHL_SEL(name, sel0, sel1)                                                                                 \
	__device__ __forceinline__ unsigned name(unsigned count, unsigned value)                                         \
	{                                                                                                                \
		unsigned r;                                                                                              \
		asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" sel0 " src1_sel:" sel1  \
		    : "=v"(r)                                                                                            \
		    : "v"(count), "v"(value));                                                                           \
		return r;                                                                                                \
	}
SCOPE_SHL_SEL(scope_shl_word0, "DWORD", "WORD_0") /* (value & 0xffff) << count */
SCOPE_SHL_SEL(scope_shl_word1, "DWORD", "WORD_1") /* (value >> 16) << count */
SCOPE_SHL_SEL(scope_shl_by_byte0, "BYTE_0", "DWORD") /* value << (count & 31) */
SCOPE_SHL_SEL(scope_shl_by_byte2, "BYTE_2", "DWORD") /* value << (count >> 16 & 31) */
#undef SCOPE_SHL_SEL

/* min(byte 3 of x, cap) in the low half / in the high half (low half zero) of the result: the waveform's alpha test as an
 * increment, one SDWA instruction (byte select on the source, word select + zero padding on the destination). */
__device__ __forceinline__ unsigned scope_min_byte3_lo(unsigned x, unsigned cap)
{
	unsigned r;
	asm("v_min_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(x), "s"(cap));
	return r;
}
__device__ __forceinline__ unsigned scope_min_byte3_hi(unsigned x, unsigned cap)
{
	unsigned r;
	asm("v_min_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(x), "s"(cap));
	return r;
}

/* A wave-uniform value pinned into a VECTOR register for the rest of the kernel.  v_mad_i32_i24 takes one scalar operand
 * (gfx9 constant bus): with coefficient AND constant term in SGPRs hipcc moves the constant into a VGPR again before every
 * multiply-add -- one VALU instruction per matrix row and cell in the fused kernels.  The asm is opaque, so the copy is
 * made once and kept. */
__device__ __forceinline__ int scope_pin_vgpr(int uniform)
{
	int v;
	asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(uniform));
	return v;
}

/* Every VMEM operation of this wave (loads, stores, atomics without return) has been acknowledged.  Inline asm:
 * hipcc (ROCm 7.2) drops a builtin wait when its own scoreboard says nothing is outstanding (MI355X_MICROARCH.md). */
__device__ __forceinline__ void scope_drain_vmem()
{
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

/* *dst += the sum of x over the 16 lanes of a DPP row (lanes 16r .. 16r+15 of the wave; every lane active): four
 * row_shr adds leave the row's total in its last lane, which adds it to LDS.  (LDS atomic: see the caller.) */
__device__ __forceinline__ void scope_row16_add(unsigned *dst, unsigned x)
{
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false); /* row_shr:1 */
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false); /* row_shr:2 */
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false); /* row_shr:4 */
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false); /* row_shr:8 */
	if ((threadIdx.x & 15u) == 15u && x)
		atomicAdd(dst, x);
}

/* The value of the lane below (lane - 1), `fill` for the first lane of a DPP row of 16 (row_shr:1, bound_ctrl off).  Every lane
 * must be active. */
__device__ __forceinline__ unsigned scope_lane_below(unsigned x, unsigned fill)
{
	return (unsigned)__builtin_amdgcn_update_dpp((int)fill, (int)x, 0x111, 0xf, 0xf, false);
}
/* The same with zero for the first lane of a row, in the form the compiler can fold into the instruction that uses the value
 * (bound_ctrl on: no `old` operand to initialise). */
__device__ __forceinline__ unsigned scope_lane_below0(unsigned x)
{
	return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);
}
/* The value of the wave's first active lane, for a value that is not wave-uniform yet (v_readfirstlane_b32). */
__device__ __forceinline__ unsigned scope_first_lane(unsigned x)
{
	return (unsigned)__builtin_amdgcn_readfirstlane((int)x);
}
/* How many bits of `mask` belong to lanes below this one (v_mbcnt_lo / v_mbcnt_hi). */
__device__ __forceinline__ unsigned scope_lanes_below(unsigned long long mask)
{
	return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
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
