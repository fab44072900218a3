/*
 * scope_device.h -- device-side vocabulary shared by the kernel translation units (scope_kernels.hip, scope_trio.hip):
 * K0 (fp32 definition + the 24-bit fixed-point fast path), the vectorscope's LDS table (bin key, swizzled address,
 * absorb mode, clamp pass), the arrival ticket of the in-launch combines, the launch macro.  Internal to libscope_hip.so.
 */
#ifndef SCOPE_DEVICE_H
#define SCOPE_DEVICE_H

#include "scope_kernels.h"
#include <hip/hip_ext.h>
#include <type_traits>
#include <scope_vec.h>

typedef uint32_t u32;
typedef uint64_t u64;

/* All LDS scratch lives in the dynamic region so its base stays 16-byte aligned. */
extern __shared__ __attribute__((aligned(16))) u32 scope_lds[];
typedef u32 scope_v4 __attribute__((vector_size(16)));

/* ------------------------------------------------------------------------------------------
 * K0: per-texel RGB -> Y'CbCr, bit-exact with oracle/scope_oracle.c:k0_texel
 * ---------------------------------------------------------------------------------------- */

/* byte/255 correctly rounded to fp32 without a divide: fma(b, hi, b*lo) with hi+lo ~= 1/255
 * (checked exhaustively for the 256 inputs in tests/test_k0_constants.py). */
__device__ __forceinline__ float scope_unorm8(u32 byte)
{
	const float b = (float)byte;
	return __fmaf_rn(b, 0x1.010102p-8f, __fmul_rn(b, -0x1.fdfdfep-33f));
}

/* FLOAT -> UNORM8 store rule pinned by the build: floor(clamp01(x)*255 + 0.5). */
__device__ __forceinline__ u32 scope_k0_quantise(float x)
{
	x = fminf(fmaxf(x, 0.0f), 1.0f);
	return (u32)__fadd_rn(__fmul_rn(x, 255.0f), 0.5f);
}

/* px bytes B,G,R,A -> bytes U,Y,V,255 (data/common.effect:23-43; the shader returns
 * float4(V,Y,U,1) into a BGRA target).  need_uv / need_y are wave-uniform. */
__device__ __forceinline__ u32 scope_k0_texel(u32 px, const ScopeK0 &k, bool need_uv, bool need_y)
{
	const float b = scope_unorm8(px & 0xffu);
	const float g = scope_unorm8((px >> 8) & 0xffu);
	const float r = scope_unorm8((px >> 16) & 0xffu);
	u32 out = 0xff000000u;
	if (need_uv) {
		/* U and V side by side in one 2-vector: the same operations in the same order on both halves,
		 * so they issue as v_pk_mul_f32 / v_pk_add_f32 (no FMA: -ffp-contract=off).  V's "- 0" is exact. */
		scope_f2 uv = scope_f2_make(k.ur, k.vr) * scope_f2_make(r, r);
		uv = uv + scope_f2_make(k.ug, k.vg) * scope_f2_make(g, g);
		uv = uv + scope_f2_make(k.ub, k.vb) * scope_f2_make(b, b);
		uv = uv + scope_f2_make(0.5f, 0.5f);
		uv = uv - scope_f2_make(0.00390625f, 0.0f);
		out |= scope_k0_quantise(scope_f2_lo(uv)) | (scope_k0_quantise(scope_f2_hi(uv)) << 16);
	}
	if (need_y) {
		float y = __fmul_rn(k.yr, r);
		y = __fadd_rn(y, __fmul_rn(k.yg, g));
		y = __fadd_rn(y, __fmul_rn(k.yb, b));
		out |= scope_k0_quantise(y) << 8;
	}
	return out;
}

/* ---- the fast path ------------------------------------------------------------------------------
 * The fp32 chain above costs ~33 VALU instructions per texel and makes every fused kernel VALU-bound.
 * The value it approximates is T = c_r R + c_g G + c_b B + 255 off + 0.5 with output floor(T) (no clamp is
 * ever active for 8-bit inputs), which three 24-bit integer multiply-adds give directly in 2^-24 fixed
 * point: E = C_r R + C_g G + C_b B + K, and the output is byte 3 of E (chroma |c| < 0.5 fits v_mul_i32_i24,
 * luma 0 <= c < 1 fits v_mul_u32_u24; the sum is taken mod 2^32 and ends below 256 * 2^24).  The two disagree
 * only where T is within ~2.5e-5 of an integer, i.e. where fp32 rounding decides: an exhaustive run over all
 * 2^24 colours, both matrices and all three rows finds every such colour within 422 units (2^-24) of a step.
 * So K carries a guard of 512 units and a value whose fraction is below 1024 -- bytes 1..2 of E, read as a
 * number, below 4 -- is "near a step": its cell is recomputed with the fp32 chain (a wave votes per cell;
 * 6e-5 .. 1e-3 of all colours).  Two rows' fractions travel as the halves of one register (v_perm_b32) and the
 * running minimum over a cell is a v_pk_min_u16.  Black is an exact tie (V = 128.0) on which both paths agree,
 * and common (letterbox bars): black texels are exempt.
 * tests/test_parity_gpu.py runs all 2^24 colours through this code against the oracle. */
__device__ __forceinline__ int scope_mad24(int a, int b, int c)
{
	return __mul24(a, b) + c; /* v_mad_i32_i24 */
}

__device__ __forceinline__ u32 scope_umad24(u32 a, u32 b, u32 c)
{
	return __umul24(a, b) + c; /* v_mad_u32_u24: the luma coefficients are positive and may exceed 0.5 */
}

#define SCOPE_K0_NEAR 4u /* 2 * SCOPE_K0_FX_GUARD >> 8 */

/* The vectorscope's bin key of a Y'CbCr texel: U in byte 1, V in byte 2, bytes 0 and 3 zero (see vss_add). */
__device__ __forceinline__ u32 vss_bin(u32 px)
{
	return scope_perm(0u, px, 0x0c02000cu);
}

/* One row of the matrix in 2^-24 fixed point: f[0] R + f[1] G + f[2] B + f[3] (mod 2^32), texel bytes B, G, R, A.
 * Four instructions, the masked blue byte `b` is shared by the rows: two byte multiplies, a multiply-add, a 3-way add. */
__device__ __forceinline__ u32 scope_k0_row(u32 px, int b, const int (&f)[4])
{
	return (u32)(scope_mad24(f[2], b, f[3]) + scope_mul24_byte1(px, f[1]) + scope_mul24_byte2(px, f[0]));
}
__device__ __forceinline__ u32 scope_k0_row_y(u32 px, u32 b, const int (&f)[4])
{
	return scope_umad24((u32)f[2], b, (u32)f[3]) + scope_umul24_byte1(px, (u32)f[1]) + scope_umul24_byte2(px, (u32)f[0]);
}

/* worst: running per-half minimum of the fractions' upper 16 bits.  VSSKEY: the result is the vectorscope's bin key
 * (0, U, V, 0) instead of the texel (U, Y, V, 255) -- the same single v_perm_b32 with another selector. */
template <bool VSSKEY>
__device__ __forceinline__ u32 scope_k0_fixed(u32 px, const ScopeK0 &k, bool need_uv, bool need_y, u32 &worst)
{
	const u32 b = px & 0xffu;
	u32 out = 0xff000000u;
	if (need_uv) {
		const u32 eu = scope_k0_row(px, (int)b, k.fu);
		const u32 ev = scope_k0_row(px, (int)b, k.fv);
		worst = scope_pk_min_u16(worst, scope_perm(ev, eu, 0x06050201u));
		out = scope_perm(ev, eu, VSSKEY ? 0x0c07030cu : 0x0d070c03u); /* 0, U, V, 0 or U, 0, V, 255 */
	}
	if (need_y) {
		const u32 ey = scope_k0_row_y(px, b, k.fy);
		worst = scope_pk_min_u16(worst, scope_perm(ey, ey, 0x06050201u));
		out = scope_perm(ey, out, 0x03020700u);
	}
	return out;
}

/* N texels of one lane, in place.  The votes are wave-wide: a texel is redone in fp32 by the whole wave when any
 * lane holds one near a step in that position (identical results for the other lanes, that is what the exhaustive test
 * says).  Black is exempt: an exact tie both paths agree on. */
template <int N, bool VSSKEY = false>
__device__ __forceinline__ void scope_k0_cell(u32 (&p)[N], const ScopeK0 &k, bool need_uv, bool need_y)
{
	u32 t[N], worst = 0xffffffffu;
#pragma unroll
	for (int j = 0; j < N; j++)
		t[j] = scope_k0_fixed<VSSKEY>(p[j], k, need_uv, need_y, worst);
	const bool near_step = min(worst & 0xffffu, worst >> 16) < SCOPE_K0_NEAR;
	if (__ballot(near_step) != 0) { /* 3 % of the cells of a wave on noise (64 lanes x 4 texels x 1.2e-4), and black */
		u32 any = 0;
#pragma unroll
		for (int j = 0; j < N; j++)
			any |= p[j];
		if (__ballot(near_step && (any & 0x00ffffffu) != 0) != 0) {
			/* which of the N texels: the wave redoes only those (one of four, as a rule) */
#pragma unroll
			for (int j = 0; j < N; j++) {
				u32 wj = 0xffffffffu;
				(void)scope_k0_fixed<VSSKEY>(p[j], k, need_uv, need_y, wj);
				const bool nj = min(wj & 0xffffu, wj >> 16) < SCOPE_K0_NEAR && (p[j] & 0x00ffffffu) != 0;
				if (N == 1 || __ballot(nj) != 0) {
					t[j] = scope_k0_texel(p[j], k, need_uv, need_y);
					if (VSSKEY)
						t[j] = vss_bin(t[j]);
				}
			}
		}
	}
#pragma unroll
	for (int j = 0; j < N; j++)
		p[j] = t[j];
}

/* The vectorscope's pair form (HASH 2, vss_add_pair): 2 NP texels in, the chroma of texels (0, 1), (2, 3) out as
 * pu[i] = U_even | U_odd << 16 and pv[i] likewise -- the top bytes of the fixed-point sums picked up two texels at a time
 * by one v_perm_b32 per row, instead of a key per texel.  Same votes, same fp32 redo as scope_k0_cell. */
template <int NP>
__device__ __forceinline__ void scope_k0_cell_uv(const u32 (&p)[2 * NP], const ScopeK0 &k, u32 (&pu)[NP], u32 (&pv)[NP])
{
	u32 eu[2 * NP], ev[2 * NP], worst = 0xffffffffu, any = 0;
#pragma unroll
	for (int j = 0; j < 2 * NP; j++) {
		const u32 b = p[j] & 0xffu;
		eu[j] = scope_k0_row(p[j], (int)b, k.fu);
		ev[j] = scope_k0_row(p[j], (int)b, k.fv);
		worst = scope_pk_min_u16(worst, scope_perm(ev[j], eu[j], 0x06050201u));
	}
	const bool near_step = min(worst & 0xffffu, worst >> 16) < SCOPE_K0_NEAR;
	if (__ballot(near_step) != 0) {
#pragma unroll
		for (int j = 0; j < 2 * NP; j++)
			any |= p[j];
		if (__ballot(near_step && (any & 0x00ffffffu) != 0) != 0) {
#pragma unroll
			for (int j = 0; j < 2 * NP; j++) {
				const u32 f = scope_perm(ev[j], eu[j], 0x06050201u);
				const bool nj = min(f & 0xffffu, f >> 16) < SCOPE_K0_NEAR && (p[j] & 0x00ffffffu) != 0;
				if (__ballot(nj) != 0) { /* the wave redoes the texels that need it: one of them, as a rule */
					const u32 t = scope_k0_texel(p[j], k, true, false); /* U, 0, V, 255 */
					eu[j] = t << 24;
					ev[j] = t << 8;
				}
			}
		}
	}
#pragma unroll
	for (int i = 0; i < NP; i++) {
		pu[i] = scope_perm(eu[2 * i + 1], eu[2 * i], 0x0c070c03u);
		pv[i] = scope_perm(ev[2 * i + 1], ev[2 * i], 0x0c070c03u);
	}
}

/* The matrix with the rows' constant terms pinned into vector registers (scope_pin_vgpr): what a fused kernel hands to
 * scope_k0_cell.  -DSCOPE_PIN_K0=0 keeps them scalar (A/B). */
#ifndef SCOPE_PIN_K0
#define SCOPE_PIN_K0 1
#endif
__device__ __forceinline__ ScopeK0 scope_k0_pinned(const ScopeK0 &k, bool need_uv, bool need_y)
{
	ScopeK0 r = k;
	if (SCOPE_PIN_K0 && need_uv) {
		r.fu[3] = scope_pin_vgpr(k.fu[3]);
		r.fv[3] = scope_pin_vgpr(k.fv[3]);
	}
	if (SCOPE_PIN_K0 && need_y)
		r.fy[3] = scope_pin_vgpr(k.fy[3]);
	return r;
}

/* "Last workgroup finishes the frame": every workgroup of a group (the slices of one frame) calls this after it has
 * added its table to the frame's accumulator with agent-scope atomics; it returns true, in every thread, for the
 * workgroup that arrives last, which may then read (exchange) the accumulator.  The caller resets the ticket when
 * done, so a zero-initialised ticket array serves every launch on the stream.
 *
 * This is the "agent atomics on both sides" hand-off of MI355X_MICROARCH.md (Workgroup dispatch, XCD placement &
 * inter-workgroup visibility): the payload is written AND read with device-scope atomic read-modify-writes, which are
 * performed at the memory side, below the per-XCD L2s, so no cache can hold a stale or an unwritten copy and no
 * L2 write-back / invalidate (buffer_wbl2 / buffer_inv, ~1.7 us each) is needed.  Every wave waits until its atomics
 * have been acknowledged (the asm wait: hipcc may drop a builtin one) before the barrier, and one lane then takes an
 * agent-scope ticket.  Round 1 handed over sc1-stored partial tables here; that form is gone.
 * tests: test_histogram_in_kernel_combine_stress (thousands of launches, every count checked). */
#ifndef SCOPE_TICKET_ORDER
#define SCOPE_TICKET_ORDER __ATOMIC_RELAXED /* A/B: -DSCOPE_TICKET_ORDER=__ATOMIC_ACQ_REL (what it costs: DESIGN.md 3.3) */
#endif
__device__ __forceinline__ bool scope_last_arrival(u32 *ticket, u32 group_size, u32 *lds_flag)
{
	scope_drain_vmem(); /* this wave's atomics have been performed */
	__syncthreads();
	if (threadIdx.x == 0)
		*lds_flag = __hip_atomic_fetch_add(ticket, 1u, SCOPE_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT) == group_size - 1u;
	__syncthreads();
	return *lds_flag != 0;
}

/* Fair shares of the SIMD.  The instruction arbiter serves the OLDEST ready wave first; in a loop with workgroup barriers that
 * starves the youngest waves for good -- in scope_trio_kernel waves 0-3 of a workgroup (the oldest on their SIMDs) were through
 * their share of a 64-frame launch in 321 us, waves 12-15 in 503 us, and everybody waited for those at every barrier while the
 * SIMD ran on fewer and fewer waves (tools/trio_trace.py, profiles/round4/trio_trace.txt).  So the issue priority goes round:
 * in round r the age class (wave / 4 + r) mod 4 is served first.  wave and round are wave-uniform. */
#ifndef SCOPE_PRIO_ROUNDS
#define SCOPE_PRIO_ROUNDS 1 /* 0: left to the arbiter (A/B) */
#endif
__device__ __forceinline__ void scope_prio_round(u32 wave, u32 round)
{
	if (!SCOPE_PRIO_ROUNDS)
		return;
	switch (((wave >> 2) + round) & 3u) {
	case 0:
		__builtin_amdgcn_s_setprio(0);
		break;
	case 1:
		__builtin_amdgcn_s_setprio(1);
		break;
	case 2:
		__builtin_amdgcn_s_setprio(2);
		break;
	default:
		__builtin_amdgcn_s_setprio(3);
		break;
	}
}

/* The lanes of ONE wave hand data to each other through the LDS (bin_rle's scratch row): a wave's LDS operations are performed
 * in order by the hardware; this tells the COMPILER not to move or forward the plain stores and loads across the point.  No
 * instructions on gfx9. */
__device__ __forceinline__ void scope_wave_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* Slice s of S of `total` cells. */
__device__ __forceinline__ u32 scope_slice_begin(u32 total, u32 s, u32 S)
{
	return (u32)(((u64)total * s) / S);
}
/* ------------------------------------------------------------------------------------------
 * K1 vectorscope
 *
 * LDS table: 32768 words, two u16 counters per word.  Bin (u, v) lives in table row v (the flush turns the
 * table upside down into output rows 255 - v), half u & 1 of
 *   HASH 0: word v*128 + ((u >> 1) ^ 8 (v & 15))     (default)
 *   HASH 1: word v*128 + (u >> 1)                     (kept for A/B: two instructions fewer, but the rows of a
 *                                                      chroma cluster then share their banks: -17 % on
 *                                                      concentrated chroma, +1..5 % elsewhere)
 * With HASH 0 sixteen consecutive rows start on sixteen different multiples of eight words, so a chroma cluster
 * of 16 x 16 bins is spread over all banks.  (Round 2 also tried getting that spread from a row pitch of 136 words,
 * with the LDS address from one multiply-add and the pixel travelling as (address, increment): five VALU
 * instructions per pixel instead of six, yet 1-2 % slower on every kind of content -- the adds of a cell then issue
 * back to back instead of between the address arithmetic of their neighbours.  Not kept.)
 *
 * A pixel's bin travels as a KEY = u << 8 | v << 16 (vss_bin: one v_perm_b32 from the texel, or straight from the
 * fixed-point sums when K0 is fused).  Equal keys <=> equal bins; the bin index u + 256*(255 - v) is never
 * materialised, the LDS byte address comes straight from the key's bits: key >> 7 = v << 9 | u << 1 is already
 * "row base + 4 * (u >> 1)" up to bit 1 (each VALU instruction occupies a SIMD for a quad-cycle here, so
 * instructions are what counts).
 *
 * Counters are 16 bit: between two clamp passes a workgroup adds at most 65280 pixels, and
 * every counter is <= 255 after a pass, so no counter can exceed 65535.  The clamp is exact
 * because the reference saturates at 255 (src/vectorscope.c:233-234).
 * ---------------------------------------------------------------------------------------- */

#define VSS_WORDS 32768u
#define VSS_EPOCH_PIXELS 65280u
/* HASH 2 (the fused-K0 kernel, which is short of VALU issue slots): rows of 136 words, word v*136 + (u >> 1), no
 * exclusive-or -- 16 consecutive rows start on 16 different multiples of eight words as well (136 = 2 * 64 + 8) -- and
 * 34 816 words = 136 KiB of the CU's 160.  Every word index fits 16 bits, so TWO pixels' indices are worked out in the
 * halves of one register (vss_add_pair): 4.5 VALU instructions per pixel from the fixed-point sums to the two LDS
 * operands instead of 7. */
#define VSS_PITCH2 136u
/* HASH 3 (scope_trio_kernel, where K0 is fused and the table shares the LDS with the waveform's): only the chroma K0 can
 * produce.  Over all 2^24 colours and both matrices U lies in [15, 239] and V in [16, 240] (data/common.effect:23-43 are
 * limited-range chroma rows; tests/test_oracle.py::test_k0_chroma_range enumerates them, test_dock_kernel_all_colours
 * runs every colour through the kernel), so the table holds rows v = 16 .. 240 and the columns u = 8 .. 239:
 * word (v - 16) * PITCH + ((u - 8) >> 1), 225 rows of PITCH >= 116 words.  PITCH = 122 (109 800 bytes instead of 131 072): 122 = 26
 * mod 32, so consecutive rows start 26 banks apart and sixteen of them start on sixteen different banks -- a chroma cluster of
 * a few rows by a few words (what a wave sees of any coherent picture) is spread over the banks.  Round 4 ran 116 (= 20 mod 32:
 * rows eight apart share their banks); on the ramp content, whose chroma lies on ONE line through the CbCr plane, the clusters of a
 * wave's four image rows then met in the same banks: a model of the table's bank traffic over sampled wave instructions
 * (profiles/round5/vss3_pitch_model.txt) put 122 at 0.77 of 116's conflict cost there and level elsewhere, and the same-process
 * A/B agreed: ramp 3084 -> 3394 GB/s (+10 %), noise / natural / screen +-0.1 %, concentrated -0.5 % (visit b5,
 * profiles/round5/ab_trio_pitch_depth.txt).  An even pitch keeps the read-out's 8-byte loads aligned; the table lies at byte
 * VSS3_BASE of the LDS block, behind the waveform's; the origin (row 16, column 8) and the base are ONE constant, carried by the
 * ds instruction's immediate offset, so the address arithmetic is HASH 2's: the raw word v * PITCH + (u >> 1) fits 16 bits
 * (vss_add_pair). */
#ifndef VSS3_PITCH
#define VSS3_PITCH 122u /* even (the read-out's 8-byte loads); < 128 (the table must fit behind the waveform's 48 KiB) */
#endif
#define VSS3_ROWS 225u
#define VSS3_V0 16u
#define VSS3_U0 8u
#define VSS3_BASE 49152u
#define VSS3_BIAS (VSS3_BASE - 4u * (VSS3_V0 * VSS3_PITCH + (VSS3_U0 >> 1)))
static_assert(VSS3_PITCH % 2u == 0u && VSS3_PITCH >= 116u && VSS3_BASE + ((VSS3_ROWS * VSS3_PITCH + 3u) & ~3u) * 4u <= 160u * 1024u,
	      "HASH 3: an even pitch that holds u = 8 .. 239 and leaves the table inside the CU's 160 KiB");
static_assert(VSS3_BASE >= 4u * (VSS3_V0 * VSS3_PITCH + (VSS3_U0 >> 1)) && VSS3_BIAS < 65536u && 240u * VSS3_PITCH + 119u < 65536u,
	      "HASH 3: the bias is a ds instruction's offset field, the raw word index a 16-bit half");
template <int HASH> struct VssTable {
	static constexpr u32 words = HASH == 3 ? ((VSS3_ROWS * VSS3_PITCH + 3u) & ~3u) : HASH == 2 ? 256u * VSS_PITCH2 : VSS_WORDS; /* whole uint4s */
	static constexpr u32 pitch = HASH == 3 ? VSS3_PITCH : HASH == 2 ? VSS_PITCH2 : 128u;
	static constexpr u32 bias = HASH == 3 ? VSS3_BIAS : 0u; /* bytes added to 4 * (v * pitch + (u >> 1)) */
};

template <int HASH> __device__ __forceinline__ void vss_add(u32 *tab, u32 key, u32 n)
{
	const u32 t = key >> 7; /* v << 9 | u << 1 */
	u32 byte;
	if (HASH == 2 || HASH == 3) /* the slow door of the pair layouts: see vss_add_pair */
		byte = (key >> 16) * (VssTable<HASH>::pitch * 4u) + (t & 0x1fcu) + VssTable<HASH>::bias;
	else if (HASH == 1)
		byte = t & 0x1fffcu;
	else
		byte = scope_bitop3<0x78>(t, key >> 11, 0x1e0u) & 0x1fffcu; /* t ^ (v << 5 & 0x1e0); key >> 11 = v << 5 | u >> 3 */
	/* half u & 1 = bit 8 of the key; bits 4..7 are zero, so the shift count is 0 or 16 */
	scope_lds_add_at(tab, byte, n << ((key >> 4) & 31u));
}

/* HASH 2, two pixels at once: pu = U0 | U1 << 16, pv = V0 | V1 << 16 (scope_k0_cell_uv).  Word indices in 16-bit halves
 * (v_pk_lshrrev_b16, v_pk_mad_u16), then per pixel one SDWA shift for the byte address and one for the increment
 * 1 << 16 (u & 1), whose count is the byte (u & 15) << 4 of pu << 4 (the shifter reads five bits). */
template <int HASH = 2> __device__ __forceinline__ void vss_add_pair(u32 *tab, u32 pu, u32 pv)
{
	constexpr u32 pitch = VssTable<HASH>::pitch, bias = VssTable<HASH>::bias;
	const u32 w = scope_pk_mad_u16(pv, pitch | (pitch << 16), scope_pk_lshr_u16(pu, 1));
	const u32 q = pu << 4;
	scope_lds_add_at(tab, scope_shl_word0(2u, w) + bias, scope_shl_by_byte0(q, 1u)); /* bias: the ds offset field */
	scope_lds_add_at(tab, scope_shl_word1(2u, w) + bias, scope_shl_by_byte2(q, 1u));
}
/* the key (0, U, V, 0) of either pixel of a pair: what the absorb mode and the votes compare */
__device__ __forceinline__ u32 vss_pair_key(u32 pu, u32 pv, int odd)
{
	return scope_perm(pv, pu, odd ? 0x0c06020cu : 0x0c04000cu);
}

/* Flat content (screen capture, graphics, letterbox, a paused scene): almost every pixel of a lane's
 * stream has the colour the lane saw last.  Same-address LDS atomics serialise at ~4 clk per lane
 * (profiles/round1/lds_atomic_microbench.txt) -- a flat frame costs 127 clk per wave instruction --
 * so in *absorb mode* a lane keeps (colour, count) in two registers, counts matching pixels there and
 * only sends the others to LDS; the pair is added when the colour changes, before every clamp pass and
 * at the end of the slice.  Whether a batch runs in absorb mode is one wave vote (vss_region). */
struct VssLane {
	u32 colour; /* key of the pending pixels; 0xffffffff = none (no key has bits in bytes 0 and 3) */
	u32 count;
};

#ifndef VSS_ABSORB_WHOLE
#define VSS_ABSORB_WHOLE 1 /* 0: the first form -- matching pixels of a mixed cell counted, the others sent one by one (A/B) */
#endif
template <int HASH> __device__ __forceinline__ void vss_absorb4(u32 *tab, VssLane &st, u32 k0, u32 k1, u32 k2, u32 k3)
{
	bool e0 = k0 == st.colour, e1 = k1 == st.colour, e2 = k2 == st.colour, e3 = k3 == st.colour;
	if (VSS_ABSORB_WHOLE) {
		/* whole cells only (see vss_absorb_pairs): the lane's colour throughout -> the register; one other colour
		 * throughout -> that becomes the lane's colour; anything else -> the table, as it is */
		if (e0 & e1 & e2 & e3) {
			st.count += 4;
		} else if ((k0 == k1) & (k1 == k2) & (k2 == k3)) {
			if (st.count)
				vss_add<HASH>(tab, st.colour, st.count);
			st.colour = k0;
			st.count = 4;
		} else {
			vss_add<HASH>(tab, k0, 1);
			vss_add<HASH>(tab, k1, 1);
			vss_add<HASH>(tab, k2, 1);
			vss_add<HASH>(tab, k3, 1);
		}
		return;
	}
	if (!(e0 | e1 | e2 | e3)) { /* none of the four is the lane's colour: it changed (rare; divergent) */
		if (st.count)
			vss_add<HASH>(tab, st.colour, st.count); /* hand in the old one ... */
		st.colour = k0;                                /* ... and start counting the new one */
		st.count = 0;
		e0 = true;
		e1 = k1 == k0;
		e2 = k2 == k0;
		e3 = k3 == k0;
	}
	st.count += (u32)e0 + (u32)e1 + (u32)e2 + (u32)e3;
	if (!(e0 & e1 & e2 & e3)) { /* one exec test skips all four when the whole wave was absorbed */
		if (!e0)
			vss_add<HASH>(tab, k0, 1);
		if (!e1)
			vss_add<HASH>(tab, k1, 1);
		if (!e2)
			vss_add<HASH>(tab, k2, 1);
		if (!e3)
			vss_add<HASH>(tab, k3, 1);
	}
}

/* The same in pair form (HASH 2): lane_u / lane_v are the lane's colour as U | U << 16, V | V << 16 (0xffffffff while it
 * has none: no pair looks like that), x = 0 in the half whose pixel has that colour.  Whole cells only: a cell of the
 * lane's colour throughout -- nearly all of them, in this mode -- is counted in the register (six instructions); a cell of
 * ONE other colour makes that the lane's colour (the old count goes to the table); any other cell goes to the table as it
 * is, pair by pair.  (The first form counted the matching pixels of a mixed cell and sent only the others to the table, as
 * vss_absorb4 does: on screen content -- flat areas with a few per cent of single-pixel detail, where every cell of a wave
 * runs the fast and the slow path -- that cost 28.3 VALU instructions per pixel against 26.2 with keys.) */
__device__ __forceinline__ void vss_absorb_pairs(u32 *tab, VssLane &st, u32 &lane_u, u32 &lane_v, u32 pu0, u32 pv0, u32 pu1, u32 pv1)
{
	const u32 x0 = scope_bitop3<0xf6>(pu0 ^ lane_u, pv0, lane_v); /* (pu ^ lane_u) | (pv ^ lane_v) */
	const u32 x1 = scope_bitop3<0xf6>(pu1 ^ lane_u, pv1, lane_v);
	if ((x0 | x1) == 0u) {
		st.count += 4;
		return;
	}
	/* one colour throughout?  Then the four differences are equal (necessary; the exact test only where that holds) */
	if (x0 == x1 && x0 == scope_perm(0u, x0, 0x01000302u) && pu0 == pu1 && pv0 == pv1 && pu0 == scope_perm(0u, pu0, 0x01000302u) &&
	    pv0 == scope_perm(0u, pv0, 0x01000302u)) {
		if (st.count)
			vss_add<2>(tab, st.colour, st.count);
		st.colour = vss_pair_key(pu0, pv0, 0);
		st.count = 4;
		lane_u = pu0;
		lane_v = pv0;
		return;
	}
	vss_add_pair(tab, pu0, pv0);
	vss_add_pair(tab, pu1, pv1);
}

/* one pair (scope_trio_kernel's slot) */
template <int HASH = 2> __device__ __forceinline__ void vss_absorb_pair(u32 *tab, VssLane &st, u32 &lane_u, u32 &lane_v, u32 pu, u32 pv)
{
	constexpr u32 pitch = VssTable<HASH>::pitch, bias = VssTable<HASH>::bias;
	u32 x = scope_bitop3<0xf6>(pu ^ lane_u, pv, lane_v);
	if (x == 0u) {
		st.count += 2;
		return;
	}
	bool e0 = (x & 0xffffu) == 0u, e1 = (x >> 16) == 0u;
	if (!(e0 | e1)) {
		if (st.count)
			vss_add<HASH>(tab, st.colour, st.count);
		st.colour = vss_pair_key(pu, pv, 0);
		st.count = 0;
		lane_u = scope_perm(0u, pu, 0x0c000c00u);
		lane_v = scope_perm(0u, pv, 0x0c000c00u);
		x = scope_bitop3<0xf6>(pu ^ lane_u, pv, lane_v);
		e0 = true;
		e1 = (x >> 16) == 0u;
	}
	st.count += (u32)e0 + (u32)e1;
	if (!(e0 & e1)) {
		const u32 w = scope_pk_mad_u16(pv, pitch | (pitch << 16), scope_pk_lshr_u16(pu, 1)), q = pu << 4;
		if (!e0)
			scope_lds_add_at(tab, scope_shl_word0(2u, w) + bias, scope_shl_by_byte0(q, 1u));
		if (!e1)
			scope_lds_add_at(tab, scope_shl_word1(2u, w) + bias, scope_shl_by_byte2(q, 1u));
	}
}

/* Add what the lanes still hold.  Called by whole waves (uniform control flow).  When every pending
 * lane of the wave holds the same colour -- the flat-frame case -- one lane adds the wave's total. */
template <int HASH> __device__ __forceinline__ void vss_flush_lane(u32 *tab, VssLane &st)
{
	const u64 pending = __ballot(st.count != 0);
	if (pending == 0)
		return;
	const u32 leader = (u32)(__ffsll((long long)pending) - 1);
	const u32 colour = __builtin_amdgcn_readlane(st.colour, leader);
	if (__all(st.count == 0 || st.colour == colour)) {
		const u32 total = scope_wave_sum(st.count);
		if ((threadIdx.x & 63u) == leader)
			vss_add<HASH>(tab, colour, total);
	} else if (st.count) {
		vss_add<HASH>(tab, st.colour, st.count);
	}
	st.count = 0;
}

/* The absorb mode's state in ONE register, S = count | U << 16 | V << 24 (round 6; a lane's count stays below 65 281, an epoch's
 * texels): with (colour, count, U | U << 16, V | V << 16) in four registers the compiler copied all four at every join of the
 * binning code's paths -- four to twelve v_mov per slot of the three-scope kernel on the path noise takes.  What the compare needs
 * is two byte permutes away.  Flush: every lane of the wave (uniform control flow); one colour throughout the wave -- the flat
 * frame -- and one lane adds the wave's total. */
template <int HASH> __device__ __forceinline__ void vss_flush_packed(u32 *tab, u32 &S)
{
	/* every lane of the wave (uniform control flow); one colour throughout the wave: one lane adds the wave's total */
	const u32 n = S & 0xffffu;
	const u64 pending = __ballot(n != 0);
	if (pending == 0)
		return;
	const u32 leader = (u32)(__ffsll((long long)pending) - 1);
	const u32 colour = __builtin_amdgcn_readlane(S >> 16, leader);
	if (__all(n == 0 || (S >> 16) == colour)) {
		const u32 total = scope_wave_sum(n);
		if ((threadIdx.x & 63u) == leader)
			vss_add<HASH>(tab, colour << 8, total); /* key = U << 8 | V << 16 */
	} else if (n) {
		vss_add<HASH>(tab, (S >> 16) << 8, n);
	}
	S &= 0xffff0000u;
}

/* two pairs (one 16-byte cell) against the lane's colour: the packed form of vss_absorb_pairs */
__device__ __forceinline__ void vss_absorb_pairs1(u32 *tab, u32 &S, u32 pu0, u32 pv0, u32 pu1, u32 pv1)
{
	const u32 lane_u = scope_perm(0u, S, 0x0c020c02u), lane_v = scope_perm(0u, S, 0x0c030c03u); /* S = 0: no pair has U = V = 0 */
	const u32 x0 = scope_bitop3<0xf6>(pu0 ^ lane_u, pv0, lane_v); /* (pu ^ lane_u) | (pv ^ lane_v) */
	const u32 x1 = scope_bitop3<0xf6>(pu1 ^ lane_u, pv1, lane_v);
	if ((x0 | x1) == 0u) {
		S += 4u;
		return;
	}
	/* one colour throughout?  Then the four differences are equal (necessary; the exact test only where that holds) */
	if (x0 == x1 && x0 == scope_perm(0u, x0, 0x01000302u) && pu0 == pu1 && pv0 == pv1 && pu0 == scope_perm(0u, pu0, 0x01000302u) &&
	    pv0 == scope_perm(0u, pv0, 0x01000302u)) {
		if (S & 0xffffu)
			vss_add<2>(tab, (S >> 16) << 8, S & 0xffffu);
		S = scope_perm(pv0, pu0, 0x04000c0cu) | 4u;
		return;
	}
	vss_add_pair(tab, pu0, pv0);
	vss_add_pair(tab, pu1, pv1);
}

__device__ __forceinline__ u32 vss_clamp_word(u32 w)
{
	const u32 lo = min(w & 0xffffu, 255u);
	const u32 hi = min(w >> 16, 255u);
	return lo | (hi << 16);
}

/* two table words = four consecutive bins -> four clamped bytes */
__device__ __forceinline__ u32 vss_clamp_pair(u32 w0, u32 w1)
{
	return min(w0 & 0xffffu, 255u) | (min(w0 >> 16, 255u) << 8) | (min(w1 & 0xffffu, 255u) << 16) | (min(w1 >> 16, 255u) << 24);
}

__device__ __forceinline__ void vss_clamp_pass(u32 *tab, u32 words = VSS_WORDS)
{
	for (u32 i = threadIdx.x * 4; i < words; i += blockDim.x * 4) {
		uint4 x = *(uint4 *)&tab[i];
		if ((x.x | x.y | x.z | x.w) & 0xff00ff00u) {
			x.x = vss_clamp_word(x.x);
			x.y = vss_clamp_word(x.y);
			x.z = vss_clamp_word(x.z);
			x.w = vss_clamp_word(x.w);
			*(uint4 *)&tab[i] = x;
		}
	}
}

/* ------------------------------------------------------------------------------------------
 * The vectorscope's tables leave a workgroup here (scope_vss_kernel, scope_trio_kernel).
 *
 *   slices == 1 (pmask NULL): the final clamped table, dst[frame].
 *   otherwise a SPARSE partial -- only the 16-bin groups that hold a count, flagged in one mask byte per thread -- laid out for
 *   scope_sat_reduce_sparse_kernel: group g of slice s is cell ((frame * 256 + g / 16) * slices + s) * 16 + g % 16.
 *   ... and with fold.out set (round 6; launches of a few slices per frame: the 64-frame batches have four) the slices of a frame
 *   meet INSIDE the launch: partials and masks are stored write-through (sc1), every storing wave drains, the workgroup takes the
 *   frame's arrival ticket, and the workgroup that arrives last adds the other slices' flagged groups (sc1 loads, all of a
 *   thread's in flight at once) to its own -- which it still holds in registers -- saturates, and writes the final table; the
 *   ticket goes back to zero.  The hand-off is MI355X_MICROARCH.md's "16 B sc1 stores AND sc1 loads" form: no release or acquire
 *   fence, no dependence on which XCDs the slices run on.  It takes scope_sat_reduce_sparse_kernel<1> (6.8 us + a launch boundary
 *   per 318 us step of the headline) out of the call.  1024 threads per workgroup, at most VSS_FOLD_MAX slices.
 * group(g, lo, hi): the eight table words (16 bins) of output group g, which is layout business of the caller. */
#define VSS_FOLD_MAX 4u
struct VssFold {
	uint8_t *out;   /* final tables: uint8[frames][65536], or uint32[frames][32768] (packed16); NULL: no fold */
	u32 *tickets;   /* [SCOPE_MAX_FRAMES], zero between launches */
	u32 packed16;
};

template <typename Group>
__device__ __forceinline__ void vss_write_out(Group &&group, u32 frame, u32 slice, u32 slices, uint8_t *dst, uint8_t *pmask, const VssFold &fold,
					      u32 *lds_flag)
{
	const u32 unit = frame * slices + slice;
	uint4 *out = (uint4 *)(dst + (size_t)unit * SCOPE_VSS_BINS);
	const u32 cell0 = (frame * 256u * slices + slice) * 16u; /* this slice's first cell (16 bytes each) of the frame's block of partials */
	const u32 mask_at = ((frame * (blockDim.x >> 4) + (threadIdx.x >> 4)) * slices + slice) * 16u + (threadIdx.x & 15u);
	const bool folding = fold.out != NULL && pmask != NULL; /* uniform */
	const scope_buf pbuf = scope_buf_make(dst, folding ? SCOPE_MAX_FRAMES * VSS_FOLD_MAX * SCOPE_VSS_BINS : 0u);
	const scope_buf mbuf = scope_buf_make(pmask, folding ? SCOPE_MAX_FRAMES * VSS_FOLD_MAX * 1024u : 0u);
	scope_u4 mine[4]; /* folding: this thread's four clamped groups */
	u32 wrote = 0;
	/* thread t takes the groups t + k * blockDim.x: four with 1024 threads (whose clamped values the fold keeps, hence the unrolled
	 * first four), eight with 512 (the most a mask byte covers), more only for the dense tables of smaller workgroups */
	auto emit = [&](u32 g, u32 k) {
		uint4 lo, hi;
		group(g, lo, hi);
		const scope_u4 t = {vss_clamp_pair(lo.x, lo.y), vss_clamp_pair(lo.z, lo.w), vss_clamp_pair(hi.x, hi.y), vss_clamp_pair(hi.z, hi.w)};
		if (!pmask) {
			out[g] = make_uint4(t.x, t.y, t.z, t.w);
		} else if ((lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) != 0) {
			const u32 cell = cell0 + (g >> 4) * slices * 16u + (g & 15u);
			if (folding) {
				scope_buf_store16_sc1(pbuf, cell * 16u, t);
			} else {
				/* streaming store: a partial is read once, by other CUs; leaving it dirty in this XCD's L2 only moves the
				 * write-back to the end of the kernel */
				const scope_v4 tv = {t.x, t.y, t.z, t.w};
				__builtin_nontemporal_store(tv, (scope_v4 *)dst + cell);
			}
			wrote |= 1u << k;
		}
		return t;
	};
#pragma unroll
	for (u32 k = 0; k < 4u; k++) {
		const u32 g = threadIdx.x + k * blockDim.x;
		mine[k] = scope_u4{0u, 0u, 0u, 0u};
		if (g < SCOPE_VSS_BINS / 16)
			mine[k] = emit(g, k);
	}
	for (u32 k = 4u, g = threadIdx.x + 4u * blockDim.x; g < SCOPE_VSS_BINS / 16; g += blockDim.x, k++)
		(void)emit(g, k);
	/* the masks likewise: byte t of slice s is ((frame * blockDim / 16 + t / 16) * slices + s) * 16 + t % 16 */
	if (pmask && !folding)
		pmask[mask_at] = (uint8_t)wrote;
	if (!folding)
		return;
	scope_buf_store1_sc1(mbuf, mask_at, wrote);
	if (!scope_last_arrival(&fold.tickets[frame], slices, lds_flag)) /* every wave drains its stores, barrier, one agent-scope ticket */
		return;
	/* the frame's last arrival: the other slices' partials are in memory.  Every load of this thread is issued before the first
	 * is used (a group nobody flagged re-reads cell 0 of the partials and is dropped). */
	u32 mb[VSS_FOLD_MAX];
#pragma unroll
	for (u32 s = 0; s < VSS_FOLD_MAX; s++)
		mb[s] = (s < slices && s != slice) ? scope_buf_load1_sc1(mbuf, mask_at + (s - slice) * 16u) : 0u;
	scope_u4 q[4][VSS_FOLD_MAX];
#pragma unroll
	for (u32 kk = 0; kk < 4u; kk++) {
		const u32 g = threadIdx.x + kk * 1024u;
#pragma unroll
		for (u32 s = 0; s < VSS_FOLD_MAX; s++) {
			const bool on = ((mb[s] >> kk) & 1u) != 0u;
			q[kk][s] = scope_buf_load16_sc1(pbuf, on ? ((frame * 256u * slices + s) * 16u + (g >> 4) * slices * 16u + (g & 15u)) * 16u : 0u);
			if (!on)
				q[kk][s] = scope_u4{0u, 0u, 0u, 0u};
		}
	}
#pragma unroll
	for (u32 kk = 0; kk < 4u; kk++) {
		const u32 g = threadIdx.x + kk * 1024u;
		/* sixteen bytes <= 255 each from up to four tables: sums in 16-bit lanes (even bytes, odd bytes), then the clamp */
		u32 ev[4] = {mine[kk].x & 0x00ff00ffu, mine[kk].y & 0x00ff00ffu, mine[kk].z & 0x00ff00ffu, mine[kk].w & 0x00ff00ffu};
		u32 od[4] = {(mine[kk].x >> 8) & 0x00ff00ffu, (mine[kk].y >> 8) & 0x00ff00ffu, (mine[kk].z >> 8) & 0x00ff00ffu, (mine[kk].w >> 8) & 0x00ff00ffu};
#pragma unroll
		for (u32 s = 0; s < VSS_FOLD_MAX; s++) {
			ev[0] += q[kk][s].x & 0x00ff00ffu, od[0] += (q[kk][s].x >> 8) & 0x00ff00ffu;
			ev[1] += q[kk][s].y & 0x00ff00ffu, od[1] += (q[kk][s].y >> 8) & 0x00ff00ffu;
			ev[2] += q[kk][s].z & 0x00ff00ffu, od[2] += (q[kk][s].z >> 8) & 0x00ff00ffu;
			ev[3] += q[kk][s].w & 0x00ff00ffu, od[3] += (q[kk][s].w >> 8) & 0x00ff00ffu;
		}
		u32 b[16];
#pragma unroll
		for (u32 j = 0; j < 4; j++) {
			b[4 * j + 0] = min(ev[j] & 0xffffu, 255u);
			b[4 * j + 1] = min(od[j] & 0xffffu, 255u);
			b[4 * j + 2] = min(ev[j] >> 16, 255u);
			b[4 * j + 3] = min(od[j] >> 16, 255u);
		}
		if (!fold.packed16) {
			((uint4 *)fold.out)[(size_t)frame * (SCOPE_VSS_BINS / 16) + g] =
				make_uint4(b[0] | b[1] << 8 | b[2] << 16 | b[3] << 24, b[4] | b[5] << 8 | b[6] << 16 | b[7] << 24,
					   b[8] | b[9] << 8 | b[10] << 16 | b[11] << 24, b[12] | b[13] << 8 | b[14] << 16 | b[15] << 24);
		} else { /* the wire format of the multi-GPU sum: two bins per 32-bit word, lo | hi << 16 */
			uint4 *wire = (uint4 *)fold.out + ((size_t)frame * (SCOPE_VSS_BINS / 16) + g) * 2;
			wire[0] = make_uint4(b[0] | b[1] << 16, b[2] | b[3] << 16, b[4] | b[5] << 16, b[6] | b[7] << 16);
			wire[1] = make_uint4(b[8] | b[9] << 16, b[10] | b[11] << 16, b[12] | b[13] << 16, b[14] | b[15] << 16);
		}
	}
	if (threadIdx.x == 0)
		__hip_atomic_store(&fold.tickets[frame], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* HIS: the histogram of the same channels comes out of this pass for nothing.  A strip's table IS a histogram per
 * column -- exact 16-bit counts until the flush clamps them -- so the flush, which reads every counter anyway, also
 * adds up each value's 64 columns (the same texels pass the same alpha test in both loops, src/waveform.c:247-248 and
 * src/histogram.c:386-387), a 16-lane row sum hands the totals to LDS, and 12 wave instructions per unit add them to
 * the frame's accumulator his.acc[frame][slot][value] with device-scope atomics.  The workgroup that finishes a
 * frame's last unit (ticket) moves the sums to his.out[frame][value][slot] and leaves accumulator and ticket zero.
 * No LDS atomic per pixel is added: the dock's waveform + histogram cost one read of the plane (src/roi.c:329-341
 * hands one staged surface to every scope). */
struct ScopeWvsHis {
	u32 *acc;     /* [SCOPE_MAX_FRAMES][4][256], zero between launches */
	u32 *tickets; /* [SCOPE_MAX_FRAMES], zero between launches */
	u32 *out;     /* [n_frames][256][4]: slots R,G,B,unused as scope_his_kernel writes them */
	u32 chmask;   /* channels the caller wants counted (a subset of CH); the others stay zero */
};

/* A launch, optionally with the kernel's own begin / end time stamps delivered into two HIP events
 * (hipExtLaunchKernelGGL: the stamps come out of the dispatch packet, as rocprofv3's do -- events recorded around a
 * launch also time the gap the command processor leaves in front of the kernel, 5 % of a 0.3 ms launch). */
#define SCOPE_LAUNCH(kernel, grid, block, lds, st, tm, ...)                                                        \
	do {                                                                                                       \
		if (tm)                                                                                            \
			hipExtLaunchKernelGGL(kernel, grid, block, lds, st, (tm)->start, (tm)->stop, 0, __VA_ARGS__); \
		else                                                                                               \
			hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                             \
	} while (0)

#endif
