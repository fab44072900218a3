/*
 * scope_trio.hip -- scope_trio_kernel: the dock's three scopes (vectorscope through the fused K0, waveform, histogram) of
 * every frame of a launch from ONE read of its RGB plane, both LDS tables resident (src/roi.c:329-341 hands one staged
 * surface to every registered scope; the reference walks it three times).  A translation unit of its own (it compiles in
 * seconds where scope_kernels.hip takes minutes: `make trio_variant` builds A/B variants of it alone).
 *
 * Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (K0 must not be FMA-contracted).
 */
#include "scope_device.h"

/* ------------------------------------------------------------------------------------------
 * The dock's three scopes from ONE read, for ANY number of frames: scope_trio_kernel.
 *
 * BOTH tables are resident in the LDS: the vectorscope's (16-bit counters; the compact HASH 3 form of scope_device.h: only
 * the chroma K0 can produce, 109 800 bytes) and a waveform table of ONE strip of columns (48 KiB; the map is described at
 * TRIO_COMPACT below).  A workgroup owns strips of one frame and streams them like any other kernel here: every texel is
 * loaded once (8 bytes = a column pair per lane, TRIO_D loads per lane in flight, buffer addressing), counted in the waveform
 * table (alpha test, src/waveform.c:247-248), sent through the fixed-point K0 and counted in the vectorscope table.  Per
 * strip: barrier, the waveform's 256 x SW output texels leave (the thread that reads a counter zeroes it; the histogram's
 * totals accumulate in three registers per thread over all strips of the slice), barrier; the loads of the next strip are in
 * flight across it.  The vectorscope's clamp pass runs before the round of loads that could overflow 16 bits.  Per slice:
 * ONE vectorscope partial (sparse, as scope_vss_kernel writes them; the final table when the frame has one slice) and
 * <= 768 device atomics for the histogram.
 *
 * Launch shapes (scope_trio_strip_width / scope_trio_slices): batches run 32-column strips -- 64 frames of 4K: 4 slices per
 * frame, slice s walks strips s, s + 4, ... (30 each); the reference's call shape runs 16-column strips so that it still
 * spreads over the chip -- three frames: 80 slices of three strips, one frame: 240 slices of one.  The rows of a wave
 * instruction lie H / 4 (32 columns) or H / 8 (16) apart, so that the equal values of vertically neighbouring texels do not
 * meet on one ADDRESS.  With 16-column strips a row segment is half a cache line: slices come in pairs that split a range of
 * strips even / odd, and the two workgroups of a pair are b and b + 8 (same XCD, same L2, same moment).
 *
 * What bounds it (profiles/round4/trio_variants.txt, ab_prio_rounds.txt, ab_dot4_addresses.txt): 26 VALU instructions
 * per texel (K0 and its vote 11.5, the waveform's three adds 4 -- their addresses come out of v_dot4_u32_u8, trio_count2 -- the
 * vectorscope's pair 4.5, the rest loop, read-outs, clamp passes) keep each SIMD's VALU busy 68 % of
 * the kernel with the four waves a 1024-thread workgroup gives it (the LDS footprint allows one workgroup per CU), and the
 * waveform's three LDS atomics per texel are about as near their limit: with the vectorscope's adds AND K0 compiled out the
 * kernel reaches 0.59 of the HBM peak, without the waveform's adds 0.565.  Measured in round 4, each in a same-process A/B: two
 * waveform replicas (bank conflicts halved) +0 %, whole-line loads and half the barriers (32-column strips) +3 ... 5 %, pair-form
 * vectorscope adds +2 %, a slot without any exec-mask branch +0.5 %, skipping the slots a strip's rounding leaves empty +2 %,
 * contiguous read-out stores +2.4 %, the waves' issue priority going round (scope_prio_round: the arbiter's oldest-first order
 * starved waves 12-15 of every workgroup) +2.3 % on noise and +7.4 % on screen content, more or fewer loads in flight 0 %, the
 * waveform's addresses by v_dot4_u32_u8 +1.7 ... 4.7 %.
 * Tried and dropped in round 3: deeper or shallower load queues, skipping K0 work of empty slots in front of EVERY slot.
 * Vector shape only (W % 4 == 0, 8-byte aligned rows, 16-byte aligned tables), H <= 65535 (a column's rows fit the waveform's
 * 16-bit counters).
 * ---------------------------------------------------------------------------------------- */
#ifndef TRIO_D
#define TRIO_D 6 /* 8-byte loads in flight per lane; a strip's loads per lane are rounded up to a multiple of it */
#endif
#ifndef TRIO_NT
#define TRIO_NT 0 /* plain loads: the other half of every cache line is wanted by the partner workgroup, so lines should stay in
		   * the L2 with normal priority (64 x 4K: 637 us against 642 with streaming loads; three frames 45.9 / 46.4) */
#endif
#ifndef TRIO_BUF
#define TRIO_BUF 1 /* buffer addressing for the pixel loads (see the issue side of the kernel); 0: global loads + offset select */
#endif
#ifndef TRIO_WABSORB
#define TRIO_WABSORB 0 /* 1: the waveform's counters in absorb mode where a whole wave sits on flat content (see the kernel).
			* Measured, 64 x 4K (profiles/round3/trio_variants.txt, visits f3): a flat frame 628 -> 890 Gpix/s, but
			* its mere presence costs the kernel 3.5 % on noise and natural content (850 -> 817, 836 -> 808) and 5 % on
			* screen content (721 -> 685; with the vectorscope's looser vote 9 %): entirely flat frames are not worth
			* that, so it is compiled out */
#endif
#ifndef TRIO_INTERLEAVE
#define TRIO_INTERLEAVE 1 /* 32-column strips: a frame's slices interleave its strips (see the kernel); 0: contiguous ranges (A/B) */
#endif
#ifndef TRIO_PRIO
#define TRIO_PRIO 1 /* the waves' issue priority goes round (see the kernel); 0: left to the arbiter (oldest first) */
#endif
#ifndef TRIO_SKIP_EMPTY
#define TRIO_SKIP_EMPTY 1
#endif
#ifndef TRIO_FASTSLOT
#define TRIO_FASTSLOT 1 /* 0: every slot runs the checked form (round 3's; A/B) */
#endif
#ifndef TRIO_DOT4
#define TRIO_DOT4 1 /* the waveform's table addresses by v_dot4_u32_u8, one instruction per add (trio_count); 0: shift and mask (A/B) */
#endif
#ifndef TRIO_EXP
#define TRIO_EXP 0 /* measurement only (wrong tables): bit 0 no waveform adds, 1 no vectorscope adds, 2 no K0, 3 no clamp pass */
#endif
#ifndef TRIO_RLE
#define TRIO_RLE 1 /* round 6: the vectorscope's run-length form (trio_rle_pair below) where the vote finds flat content; 0: the per-lane
		    * absorb mode alone (rounds 3-5; A/B) */
#endif
#ifndef TRIO_RLE_HI
#define TRIO_RLE_HI 64u /* only with TRIO_ABSORB 1: flat lanes of 64 at the vote from which the per-lane absorb mode runs instead of the
			 * run-length form (64 = where the WHOLE wave is flat; 65 = never) */
#endif
#ifndef TRIO_ABSORB
#define TRIO_ABSORB 1 /* the per-lane absorb mode: 1 = on one state register (trio_absorb_pair; with TRIO_RLE only where the vote finds the
		       * whole wave flat, TRIO_RLE_HI), 4 = rounds 3-5's four-register form (A/B), 0 = none (the run-length form takes all
		       * flat content; A/B) */
#endif
#define TRIO_MODE_ABSORB 1u
#define TRIO_MODE_RLE 2u
#define TRIO_MODE_RLE_WHOLE 6u /* TRIO_MODE_RLE | 4: with the whole-wave question */
#ifndef TRIO_RLE_LO
#define TRIO_RLE_LO 48u /* flat lanes of 64 at the vote from which a wave leaves the plain adds (three quarters: rounds 3-5's absorb threshold) */
#endif
#ifndef TRIO_RLE_WHOLE
#define TRIO_RLE_WHOLE (!TRIO_ABSORB) /* the run-length form's whole-wave shortcut (trio_rle_pair): what stands in for the absorb mode where
					* that is compiled out (A/B: flat frames 13 % behind the absorb mode, visit d6) */
#endif
#ifndef TRIO_STORE_SC1
#define TRIO_STORE_SC1 0 /* the strip's read-out stores: 0 streaming (nt), 1 write-through (sc1, as the waveform kernel's batch form; A/B) */
#endif
#ifndef TRIO_RLE_FASTONLY
#define TRIO_RLE_FASTONLY 0 /* 1: only the branch-free slot form carries the run-length form (the checked slots -- two of 36 per 4K strip -- keep
			      * the absorb mode): less code (A/B) */
#endif
#ifndef TRIO_COMPACT
#define TRIO_COMPACT 1 /* round 4: the LDS map below.  0: round 3's map (A/B: `make trio_variant VARIANT_FLAGS=-DTRIO_COMPACT=0`) */
#endif
/* ---- the LDS map (TRIO_COMPACT) ---------------------------------------------------------------------------------------
 * Round 3's map -- the vectorscope's full 128 KiB table and ONE waveform replica of 8 KiB per channel -- left the kernel waiting
 * for the LDS: a half wave is 4 rows x 8 column pairs and a counter's bank was (value mod 4) * 8 + pair, so two of the four rows
 * met in a bank whenever their values agreed mod 4 (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.64,
 * profiles/round3/pmc_sq_three_scope_x64_f1.txt), and four rows of flat content met on one ADDRESS.
 * K0 is fused here, and its chroma rows are limited-range: U in [15, 239], V in [16, 240] for every colour and both matrices
 * (scope_device.h, HASH 3).  The vectorscope table shrinks to that rectangle -- 225 rows of 122 words (round 4: 116), 109 800 bytes -- and the
 * 21 KiB that frees buy a SECOND waveform replica (four for a one-channel mask):
 *     bytes      0 ..  49 151   waveform: [channel][256 values][REP replicas][8 column pairs] words of two 16-bit counters
 *     bytes 49 152 .. 158 951   vectorscope, HASH 3
 * A lane's replica is its row within the half wave mod REP, so the bank is (value mod (4 / REP)) * 8 REP + replica * 8 + pair:
 * with two replicas only the rows r and r + 2 can meet, two-way at most -- which costs a ds_add nothing
 * (profiles/round1/lds_atomic_microbench.txt: "same bank, distinct address x 2" = the conflict-free 4.5 clk) -- and flat content
 * puts two lanes on an address instead of four; with four replicas (one channel) no two lanes of a half wave share a bank.
 * The strip's read-out adds the replicas (packed 16-bit adds: a column's total is at most H <= 65 535).
 *
 * The same 48 KiB hold ONE replica of a strip twice as wide (SW = 32 columns: 16 column pairs, a wave is 4 rows x 32 columns,
 * word = value * 16 + pair): the bank is (value mod 2) * 16 + pair, the two rows of a half wave meet two-way at most as
 * above -- and a strip's row segment is now a whole 128-byte line (no partner workgroup fetching the other half from the L2),
 * a launch has half the strips (half the barriers and read-outs per texel) and the read-out stores 128 contiguous bytes per
 * value instead of 64.  Batches run this form; launches too small to give every CU four such strips (one frame, the ring of
 * three) keep 16-column strips with the replicas, so that a one-frame call still spreads over 240 workgroups.
 * scope_trio_slices decides; "trio.strip" (scope_ctx_set_option) forces 16 or 32. */
#if TRIO_COMPACT
#undef TRIO_HASH
#define TRIO_HASH 3
#ifndef TRIO_PAIR
#define TRIO_PAIR 1 /* the vectorscope's adds in pair form (vss_add_pair<3>): with the LDS relieved, VALU issue is what is short */
#endif
#define TRIO_REP(nch, sw) ((sw) == 32u ? 1u : (nch) == 1u ? 4u : 2u)
#define TRIO_WVS_BASE 0u
#define TRIO_VSS_BASE VSS3_BASE
#define TRIO_LDS_BYTES(nch) (VSS3_BASE + VssTable<3>::words * 4u)
#else
#ifndef TRIO_HASH
#define TRIO_HASH 0 /* the vectorscope table's layout.  0: as scope_vss_kernel's plain form.  2: rows of 136 words, a slot's two
		     * texels addressed as a pair (vss_add_pair: 4.5 VALU instructions per texel instead of 7, what gave the fused-K0
		     * vectorscope 4 %); with the waveform's 24 KiB that is all 160 KiB of the CU.  Measured in round 3 (visit p10, same
		     * process, 64 x 4K): noise 3779 -> 3758 GB/s, natural 3724 -> 3746, flat 2890 -> 2938, screen 3074 -> 2847 */
#endif
#define TRIO_PAIR (TRIO_HASH == 2)
#define TRIO_REP(nch, sw) 1u
#define TRIO_VSS_BASE 0u
#define TRIO_WVS_BASE (VssTable<TRIO_HASH>::words * 4u)
#define TRIO_LDS_BYTES(nch) (VssTable<TRIO_HASH>::words * 4u + (nch) * 8192u)
#endif
#define TRIO_VSS_WORDS (VssTable<TRIO_HASH>::words)

/* Development build only (`make trio_variant VARIANT_FLAGS=-DTRIO_TRACE VARIANT_TAG=trace`, tools/trio_trace.py): every wave adds up
 * where its time goes -- waiting at the strip's barriers, the read-out, the clamp passes (their barriers included), everything --
 * in 100 MHz ticks, and leaves the four sums in a device array.  Never the product. */
#ifdef TRIO_TRACE
__device__ unsigned long long scope_trio_dbg[256 * 16 * 4];
extern "C" int scope_trio_debug_read(unsigned long long *out)
{
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(scope_trio_dbg), sizeof(scope_trio_dbg));
}
#define TRIO_T(var) const unsigned long long var = wall_clock64()
#define TRIO_T_ADD(acc, a, b) acc += (b) - (a)
#else
#define TRIO_T(var) do { } while (0)
#define TRIO_T_ADD(acc, a, b) do { } while (0)
#endif

/* one texel into the waveform table: value v of channel c -> byte c * CHB + v * 4 VS + lane_base (replica and column pair) */
template <int CH, u32 VS> __device__ __forceinline__ void trio_count(u32 *tab, u32 px, u32 lane_base, u32 inc)
{
	constexpr u32 slot1 = (CH & 1), slot2 = slot1 + ((CH >> 1) & 1);
	constexpr u32 CHB = 256u * VS * 4u, SH = VS == 32 ? 7u : VS == 16 ? 6u : 5u; /* VS = words per value: 8, 16 or 32 */
	if (TRIO_DOT4) {
		/* value << SH + lane_base in ONE instruction: v_dot4_u32_u8 with the factor 1 << SH (32, 64 or 128) in the channel's byte
		 * (scope_udot4); the channel's table comes in as the ds_add's immediate offset */
		if (CH & 1)
			scope_lds_add_at(tab, scope_udot4(px, 1u << SH, lane_base), inc);
		if (CH & 2)
			scope_lds_add_at(tab, scope_udot4(px, 1u << (SH + 8u), lane_base) + slot1 * CHB, inc);
		if (CH & 4)
			scope_lds_add_at(tab, scope_udot4(px, 1u << (SH + 16u), lane_base) + slot2 * CHB, inc);
		return;
	}
	if (CH & 1)
		scope_lds_add_at(tab, lane_base + ((px & 0xffu) << SH), inc);
	if (CH & 2)
		scope_lds_add_at(tab, lane_base + slot1 * CHB + (((px >> 8) & 0xffu) << SH), inc);
	if (CH & 4)
		scope_lds_add_at(tab, lane_base + slot2 * CHB + (((px >> 16) & 0xffu) << SH), inc);
}

/* The two texels of a lane's column pair at once: the (up to) six addresses first, then the six adds -- an address straight out of
 * v_dot4_u32_u8 into the ds_add that uses it costs three wait states each time (the compiler pads with s_nop). */
template <int CH, u32 VS> __device__ __forceinline__ void trio_count2(u32 *tab, u32 p0, u32 p1, u32 lane_base, u32 inc0, u32 inc1)
{
	constexpr u32 slot1 = (CH & 1), slot2 = slot1 + ((CH >> 1) & 1);
	constexpr u32 CHB = 256u * VS * 4u, SH = VS == 32 ? 7u : VS == 16 ? 6u : 5u;
	if (!TRIO_DOT4) {
		trio_count<CH, VS>(tab, p0, lane_base, inc0);
		trio_count<CH, VS>(tab, p1, lane_base, inc1);
		return;
	}
	u32 a0[3], a1[3];
#pragma unroll
	for (u32 c = 0; c < 3; c++) {
		a0[c] = scope_udot4(p0, 1u << (SH + 8u * c), lane_base);
		a1[c] = scope_udot4(p1, 1u << (SH + 8u * c), lane_base);
	}
	if (CH & 1) {
		scope_lds_add_at(tab, a0[0], inc0);
		scope_lds_add_at(tab, a1[0], inc1);
	}
	if (CH & 2) {
		scope_lds_add_at(tab, a0[1] + slot1 * CHB, inc0);
		scope_lds_add_at(tab, a1[1] + slot1 * CHB, inc1);
	}
	if (CH & 4) {
		scope_lds_add_at(tab, a0[2] + slot2 * CHB, inc0);
		scope_lds_add_at(tab, a1[2] + slot2 * CHB, inc1);
	}
}

/* ---- the per-lane absorb mode on ONE state register (round 6) -----------------------------------------------------------------
 * Rounds 3-5 carried the mode's state in four registers (colour key, count, the colour as U | U << 16 and V | V << 16), and the
 * compiler copied all four at every join of a slot's paths: four v_mov per slot on the path NOISE takes, twelve once the
 * run-length form added a third way through the slot (visit a6: -6 % on noise).  Here the state is S = count | U << 16 | V << 24
 * (a lane's count stays below 65 281, an epoch's texels); what the compare needs is two byte permutes away.  The mode runs only
 * where the vote found the WHOLE wave flat (flat frames, the inside of large areas: a lane's colour then rarely changes and a
 * slot costs six instructions and no LDS operation); everything between that and noise is the run-length form's. */
template <int HASH> __device__ __forceinline__ void trio_absorb_pair(u32 *tab, u32 &S, u32 pu, u32 pv)
{
	constexpr u32 pitch = VssTable<HASH>::pitch, bias = VssTable<HASH>::bias;
	/* x = 0 in the half whose texel has the lane's colour (S = 0, no colour yet: no pair has U = V = 0) */
	u32 x = scope_bitop3<0xf6>(pu ^ scope_perm(0u, S, 0x0c020c02u), pv, scope_perm(0u, S, 0x0c030c03u));
	if (x == 0u) {
		S += 2u;
		return;
	}
	bool e0 = (x & 0xffffu) == 0u, e1 = (x >> 16) == 0u;
	if (!(e0 | e1)) { /* the colour changed: the old count goes to the table, texel 0's colour is the lane's */
		if (S & 0xffffu)
			vss_add<HASH>(tab, (S >> 16) << 8, S & 0xffffu);
		S = scope_perm(pv, pu, 0x04000c0cu);
		x = scope_bitop3<0xf6>(pu ^ scope_perm(0u, S, 0x0c020c02u), pv, scope_perm(0u, S, 0x0c030c03u));
		e0 = true;
		e1 = (x >> 16) == 0u;
	}
	S += (u32)e0 + (u32)e1;
	if (!(e0 & e1)) {
		const u32 w = scope_pk_mad_u16(pv, pitch | (pitch << 16), scope_pk_lshr_u16(pu, 1)), q = pu << 4;
		if (!e0)
			scope_lds_add_at(tab, scope_shl_word0(2u, w) + bias, scope_shl_by_byte0(q, 1u));
		if (!e1)
			scope_lds_add_at(tab, scope_shl_word1(2u, w) + bias, scope_shl_by_byte2(q, 1u));
	}
}

/* ---- the vectorscope's run-length form (round 6) --------------------------------------------------------------------
 * Screen capture -- flat rectangles with a little single-texel detail -- was this kernel's worst content: 38 VALU instructions per
 * texel against 26 on noise (profiles/round5/pmc_sq_three_scope_x64_screen_*.txt).  The per-lane absorb mode keeps (colour, count)
 * per LANE, and a lane's consecutive slots lie sixteen image rows apart: when a row of the picture crosses a rectangle's edge all
 * sixteen lanes of the row segment change colour in the same slot and hand the same old colour in (16- to 64-way same-address
 * atomics), and every wave holds a lane that meets detail in almost every slot, so the wave runs the register path, the colour
 * change AND the table path.  What is coherent is the row segment ACROSS lanes: the sixteen lanes of a DPP row hold 32 adjacent
 * texels (32-column strips; with 16-column strips two segments of eight lanes, rows H / 8 apart -- equal keys merge all the same:
 * the form needs equal bins, not neighbours).  So, with no per-lane state (scope_vss_kernel's keys_rle after K0):
 *   - a pair is FLAT when its two texels have one chroma; a flat pair whose key equals the key of the flat pair in the lane below
 *     (one DPP move and a compare; the first lane of a DPP row has no lane below) adds nothing;
 *   - the HEAD of a run adds 2 x the run's length, read off the ballot of the continuing lanes (a 64-bit shift and a find-first-bit:
 *     the next lane that continues nothing ends the run; a run ends with its DPP row);
 *   - a pair that is not flat adds its two texels as the plain form does.
 * Heads and rough pairs share ONE address computation (vss_add_pair's) and the first ds_add; only the second texel of a rough pair
 * is a second, exec-masked add.  19 VALU instructions per slot whatever the content (plain adds: 7), one LDS atomic per run.
 * Every texel is counted once in the bin K0 gave it: exact.  `v`: the lane holds a real texel pair (lanes that do not break every
 * run and add nothing); every lane of the wave must be active. */
template <int HASH, bool ALLV> __device__ __forceinline__ void trio_rle_pair(u32 *tab, u32 pu, u32 pv, bool v, u32 lane, u32 wave_w, bool whole)
{
	constexpr u32 pitch = VssTable<HASH>::pitch, bias = VssTable<HASH>::bias;
	/* the pair's two colours as 16-bit halves (U | V << 8 each): flat <=> the halves agree, and a flat pair's word is its tag --
	 * never zero (U >= 15) and never a lane's own tag, whose halves differ (0x4000 and the lane) */
	const u32 c = pu | (pv << 8);
	if (ALLV && TRIO_RLE_WHOLE && whole) {
		/* The whole wave ONE colour (a flat frame, the inside of a rectangle, letterbox bars)?  Then its first lane adds the wave's
		 * 2 x 64 texels and nobody else does anything: the question is one readfirstlane and one compare, the bin's address is
		 * worked out on the scalar side.  (What the per-lane absorb mode of rounds 3-5 was for -- without its four state registers,
		 * whose copies at every join of the slot cost the path NOISE takes four to twelve v_mov per slot.)  Asked only where the
		 * last vote found EVERY lane of the wave flat (`whole`, wave-uniform): in front of every slot the question's chain of
		 * scalar and vector steps cost screen content 8 % (visit c6), where it is almost never answered with yes. */
		const u32 c0 = scope_first_lane(c);
		if (((c0 ^ (c0 >> 16)) & 0xffffu) == 0u && __all(c == c0)) {
			if (scope_lanes_below(__ballot(true)) == 0u)
				vss_add<HASH>(tab, (c0 & 0xffffu) << 8, 2u * wave_w); /* key = U << 8 | V << 16 */
			return;
		}
	}
	const bool same = v & ((c & 0xffffu) == (c >> 16));
	const u32 tag = same ? c : (0x40000000u | lane);
	const bool cont = tag == scope_lane_below0(tag); /* the first lane of a DPP row reads zero: it continues nothing */
	const u64 breaks = ~__ballot(cont); /* the lanes that continue nothing, and the lanes there are none of */
	if (v && !cont) {
		/* bit i of `above`: lane + 1 + i continues nothing.  A run ends within its sixteen lanes (the next DPP row's first lane, or
		 * the sentinel behind lane 63), so the low half always holds a bit */
		const u64 above = ((breaks >> 1) | (1ull << 63)) >> lane;
		const u32 n0 = same ? 2u * (u32)__builtin_ctz((u32)above) + 2u : 1u;
		const u32 w = scope_pk_mad_u16(pv, pitch | (pitch << 16), scope_pk_lshr_u16(pu, 1)), q = pu << 4;
		scope_lds_add_at(tab, scope_shl_word0(2u, w) + bias, scope_shl_by_byte0(q, n0));
		if (!same)
			scope_lds_add_at(tab, scope_shl_word1(2u, w) + bias, scope_shl_by_byte2(q, 1u));
	}
}

template <int CH, bool HIS, int SW>
__global__ __launch_bounds__(SCOPE_BLOCK) void scope_trio_kernel(ScopeFrameSet fs, u32 strips, u32 slices, ScopeK0 k0_arg,
								uint8_t *wvs_dst, ScopeWvsHis his, uint8_t *vss_dst, uint8_t *vss_mask, u32 rle_force, VssFold fold)
{
	const ScopeK0 k0 = scope_k0_pinned(k0_arg, true, false);
	constexpr u32 slot1 = (CH & 1), slot2 = slot1 + ((CH >> 1) & 1), nch = slot2 + ((CH >> 2) & 1);
	constexpr u32 slot_of[3] = {0, slot1, slot2};
	constexpr int D = TRIO_D;
	u32 *tab = scope_lds;
	scope_lds_is_at_zero(tab);
	constexpr u32 P = SW / 2, PL = SW == 32 ? 4u : 3u, RW = 64u / P; /* column pairs per strip (and its log2); rows of a wave */
	constexpr u32 REP = TRIO_REP(nch, (u32)SW), VS = P * REP, CHW = 256u * VS; /* replicas; words per value / per channel */
	u32 *wt = tab + TRIO_WVS_BASE / 4u; /* [nch][256 values][REP replicas][P column pairs] */
	u32 *vt = tab + TRIO_VSS_BASE / 4u; /* the vectorscope table (the add functions address it from LDS byte 0 themselves) */
	const u32 W = fs.width, H = fs.height;
	const u32 lane = threadIdx.x & 63u;
	const u32 pair = lane & (P - 1u), rsub = lane >> PL;
	const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const u32 nwaves = blockDim.x >> 6;

	/* ---- which strips are mine ------------------------------------------------------------------------------------ */
	const u32 b = blockIdx.x, total = gridDim.x;
	u32 frame, slice, first, last, stride;
	if (SW == 16 && (slices & 1u) == 0) {
		/* pairs: workgroups b and b + 8 of a block of 16 share a range of strip PAIRS (a pair = one 128-byte line per row) and
		 * take its even / odd strips */
		const u32 half = slices >> 1, base = total & ~15u;
		u32 p, parity;
		if (b < base) {
			p = ((b >> 4) << 3) + (b & 7u);
			parity = (b >> 3) & 1u;
		} else {
			p = (base >> 1) + ((b - base) >> 1);
			parity = (b - base) & 1u;
		}
		frame = p / half;
		const u32 q = p - frame * half;
		const u32 spairs = (strips + 1u) >> 1;
		first = 2u * scope_slice_begin(spairs, q, half) + parity;
		last = min(2u * scope_slice_begin(spairs, q + 1u, half), strips);
		stride = 2;
		slice = 2u * q + parity;
	} else if (TRIO_INTERLEAVE && SW == 32) {
		/* the slices of a frame take its strips in turn -- slice s walks strips s, s + slices, ... -- so that the workgroups of a
		 * frame, which start together and run at one pace, are on NEIGHBOURING strips at any moment: their 128-byte row segments
		 * are pieces of the same DRAM pages, asked for at about the same time */
		frame = b / slices;
		slice = b - frame * slices;
		first = slice;
		last = strips;
		stride = slices;
	} else {
		frame = b / slices;
		slice = b - frame * slices;
		first = scope_slice_begin(strips, slice, slices);
		last = scope_slice_begin(strips, slice + 1u, slices);
		stride = 1;
	}
	const u32 nst = first < last ? (last - first + stride - 1u) / stride : 0u;
	const uint8_t *plane = fs.plane[frame];

	/* rows of a lane: rsub * rps + i, i = wave, wave + nwaves, ...; G loads per lane and strip, rounded up to whole rounds of D */
	const u32 rps = (H + RW - 1u) / RW;
	const u32 row0 = rsub * rps;
	const u32 G = (rps + nwaves - 1u) / nwaves;
	const u32 Gp = ((G + (u32)D - 1u) / (u32)D) * (u32)D;
	const u32 dstep = nwaves * fs.linesize;
	const u32 lane_base = TRIO_WVS_BASE + (rsub & (REP - 1u)) * (P * 4u) + (pair << 2); /* the lane's replica: its row in the half wave mod REP */

	/* ---- issue side ------------------------------------------------------------------------------------------------
	 * TRIO_BUF (default): buffer addressing -- the plane is a buffer resource of H * linesize bytes in SGPRs, a lane holds ONE
	 * 32-bit byte offset that advances by nwaves rows per slot, and a load past the plane's end returns zeros by the
	 * hardware's range check: issuing a slot is the load and one add.  Which slots of a lane are real is decided where they
	 * are binned: slot g < nv (a per-lane constant: the lane's rows form a prefix of its slots) and the strip's column
	 * exists.  TRIO_BUF 0: global loads with a validity test and an offset select per slot (A/B). */
	u32 px[D][2];
#if !TRIO_BUF
	bool ok[D];
#endif
	u32 s_i = 0, g_i = 0, off_i = 0;
	bool live_i = false;
	const u32 lim = row0 < H ? min(rps, H - row0) : 0u; /* rows of this lane's range that exist */
	const u32 nv = lim > wave ? (lim - wave + nwaves - 1u) / nwaves : 0u;
	/* the resource ends with the last row's last texel, not with its pitch padding: the API does not ask for the padding
	 * behind the last row to be allocated, and lanes of a ragged last strip (x0 >= W) must not touch it */
	const u32 plane_bytes = (H - 1u) * fs.linesize + W * 4u;
	const scope_buf buf = scope_buf_make(plane, plane_bytes);
	/* the shortest row range of the wave is the last lane's (the highest row range ends with the frame) */
	const u32 nv_min = TRIO_BUF ? (u32)__builtin_amdgcn_readlane((int)nv, 63) : 0u;
	(void)nv_min;
	(void)buf;
	(void)nv;
	(void)live_i;
	auto strip_start = [&](u32 s) {
		const u32 x0 = (first + s * stride) * (u32)SW + 2u * pair;
		live_i = s < nst && x0 < W; /* W % 4 == 0: both texels of the pair or none */
		off_i = (row0 + wave) * fs.linesize + x0 * 4u;
		if (TRIO_BUF && s >= nst)
			off_i = plane_bytes; /* past the last strip: every load out of range */
	};
	strip_start(0);
#if TRIO_BUF
#define TRIO_ISSUE(d)                                                                                                  \
	do {                                                                                                           \
		const scope_u2 t_ = scope_buf_load2<TRIO_NT != 0>(buf, off_i);                                         \
		px[d][0] = t_.x;                                                                                       \
		px[d][1] = t_.y;                                                                                       \
		off_i += dstep;                                                                                        \
		if (++g_i == Gp) {                                                                                     \
			g_i = 0;                                                                                       \
			strip_start(++s_i);                                                                            \
		}                                                                                                      \
	} while (0)
#define TRIO_OK(d, g) (live_b && (g) < nv)
#else
#define TRIO_ISSUE(d)                                                                                                  \
	do {                                                                                                           \
		const u32 i_ = wave + g_i * nwaves;                                                                    \
		ok[d] = live_i && i_ < rps && row0 + i_ < H;                                                           \
		scope_u2 t_;                                                                                           \
		if (TRIO_NT)                                                                                           \
			t_ = scope_load_row2_nt(plane, ok[d] ? off_i : 0u);                                            \
		else                                                                                                   \
			t_ = scope_load_row2(plane, ok[d] ? off_i : 0u);                                               \
		px[d][0] = t_.x;                                                                                       \
		px[d][1] = t_.y;                                                                                       \
		off_i += dstep;                                                                                        \
		if (++g_i == Gp) {                                                                                     \
			g_i = 0;                                                                                       \
			strip_start(++s_i);                                                                            \
		}                                                                                                      \
	} while (0)
#define TRIO_OK(d, g) (ok[d])
#endif
#pragma unroll
	for (int d = 0; d < D; d++)
		TRIO_ISSUE(d);

	/* both tables zeroed under the first loads */
	{
		const uint4 zero = make_uint4(0, 0, 0, 0);
		for (u32 i = threadIdx.x * 4; i < TRIO_LDS_BYTES(nch) / 4u; i += blockDim.x * 4)
			*(uint4 *)&tab[i] = zero;
	}
	__syncthreads();

	VssLane st = {0xffffffffu, 0};
	u32 lane_u = 0xffffffffu, lane_v = 0xffffffffu; /* TRIO_ABSORB 4: the lane's colour (vss_absorb_pair) */
	u32 S = 0; /* TRIO_ABSORB 1: the absorb mode's state in one register (trio_absorb_pair) */
	(void)lane_u;
	(void)lane_v;
	(void)S;
#ifndef TRIO_SCALAR_VERDICT
#define TRIO_SCALAR_VERDICT 1 /* 0: the vote's verdict as a bool (round 4; A/B) */
#endif
	/* wave-uniform, and kept in a scalar register (readfirstlane): a bool carried round the loop becomes a lane mask and every
	 * `if (absorb)` an exec-mask dance of six scalar instructions and two branches per slot (as in vss_region) */
	u32 absorb = 0;
	u32 mode = rle_force ? TRIO_MODE_RLE : 0u; /* the pair form's mode (wave-uniform): the vote's, or the run-length form forced (tests) */
	bool wflat = false;
	const u32 wave_w = (u32)__popcll(__ballot(true)); /* 64; one on the CPU emulation (tests/emul), whose votes are a lane's own */
	u32 wp0 = 0, wp1 = 0, wn = 0; /* the waveform's pending texel pair and how often it came (absorb mode) */
	u32 since = 0;       /* texels the vectorscope table may have taken since its last clamp pass (uniform) */
	const u32 round_px = (u32)D * blockDim.x * 2u; /* texels a round of D slots can add (upper bound: every lane valid) */
	/* The strip's read-out: 256 values x SW / 4 groups of four columns = NV items per thread (1024 threads).  Item j of thread t
	 * is value (t >> GL) + (1024 >> GL) j, group t & (2^GL - 1): the lanes of ONE store instruction cover whole value rows --
	 * 64 contiguous bytes per value with 16-column strips, 128 with 32 (a thread holding two NEIGHBOURING groups instead made
	 * every store instruction write every other 16 bytes: PMC WRITE_SIZE 1.36 x, visit d1). */
	constexpr u32 NV = (u32)SW / 16u, GL = SW == 32 ? 3u : 2u;
	u32 hs[3][NV] = {}; /* HIS: this thread's (value, column group) totals over the slice's strips */
	const u32 value0 = threadIdx.x >> GL, g4 = threadIdx.x & ((1u << GL) - 1u);

#ifdef TRIO_TRACE
	unsigned long long t_barrier = 0, t_readout = 0, t_clamp = 0;
	const unsigned long long t_start = wall_clock64();
#endif
	u32 prio_round = 0; /* uniform */
	for (u32 s_b = 0; s_b < nst; s_b++) {
		const bool live_b = (first + s_b * stride) * (u32)SW + 2u * pair < W; /* this lane's column pair exists in the strip */
		const bool strip_whole = (first + s_b * stride + 1u) * (u32)SW <= W;  /* ... and so does every lane's (uniform) */
		(void)live_b;
		(void)strip_whole;
		for (u32 g_b = 0; g_b < Gp; g_b += (u32)D) {
			if (TRIO_PRIO)
				scope_prio_round(wave, prio_round++); /* fair shares of the SIMD: scope_device.h */
			/* the vectorscope's 16-bit counters: a clamp pass (every counter min(c, 255): exact, the reference saturates
			 * at 255) before the round that could carry one past 65535 -- at round granularity, not per strip: a 4K strip is
			 * 34 560 texels, two strips would overflow, but five rounds of 12 288 fit, so 60 strips need 36 passes and
			 * not 59 (a pass reads the whole 128 KiB table: 5 % of the kernel when taken once per strip) */
			if (since + round_px > VSS_EPOCH_PIXELS && !(TRIO_EXP & 8)) { /* uniform across the workgroup */
				TRIO_T(tc0);
				if (TRIO_PAIR && TRIO_ABSORB == 1)
					vss_flush_packed<TRIO_HASH>(tab, S); /* pending texels belong to the epoch that ends here */
				else
					vss_flush_lane<TRIO_HASH>(tab, st);
				__syncthreads();
				vss_clamp_pass(vt, TRIO_VSS_WORDS);
				__syncthreads();
				since = 0;
				TRIO_T(tc1);
				TRIO_T_ADD(t_clamp, tc0, tc1);
			}
			since += round_px;
			/* One slot = the two texels a lane loaded.  FAST: every lane of the wave holds a real texel pair (the strip is whole
			 * and the slot lies inside the shortest row range of the wave: all but the last slots of a strip, all but a ragged
			 * last strip) -- no validity mask and no exec-mask branches at all; the other slots run the checked form. */
			auto slot = [&](auto fast_tag, const int d) __attribute__((always_inline)) {
				constexpr bool FAST = decltype(fast_tag)::value;
				const bool v = FAST || TRIO_OK(d, g_b + (u32)d);
				if (!FAST && TRIO_SKIP_EMPTY && __ballot(v) == 0)
					return; /* a slot no lane of the wave has a texel for: the rounding of a strip's slots up to whole rounds of D
						 * (two of 36 at 4K) and the rows past a short last row range.  (Round 3 found this test too dear --
						 * it sat in front of every slot; here only the checked form carries it.) */
				const u32 p0 = px[d][0], p1 = px[d][1];
				u32 key[2] = {p0, p1};
				u32 pu[1] = {0}, pv[1] = {0}; /* pair form: the slot's chroma */
				if (v && !(TRIO_EXP & 1)) {
					if (TRIO_WABSORB && wflat) {
						/* Flat content -- EVERY lane of the wave held a horizontally equal texel pair at the last vote
						 * (stricter than the vectorscope's three quarters: screen content with a few per cent of detail
						 * flushed its pending pairs so often that this mode lost 9 % there): a lane's consecutive
						 * slots then hold the same pair too, and with ONE
						 * replica four rows of a half wave would meet on one counter at 4 clk per lane (a flat
						 * frame ran at 628 Gpix/s, noise at 859).  The lane counts repeats of its pending pair in a
						 * register and adds them in one go when the pair changes, at the strip's end at the latest.
						 * Either mode is exact; `wflat` is wave-uniform. */
						if (p0 == wp0 && p1 == wp1) {
							wn++;
						} else {
							if (wn) {
								if ((wp0 >> 24) != 0)
									trio_count<CH, VS>(tab, wp0, lane_base, wn);
								if ((wp1 >> 24) != 0)
									trio_count<CH, VS>(tab, wp1, lane_base, wn << 16);
							}
							wp0 = p0;
							wp1 = p1;
							wn = 1;
						}
					} else {
						if (FAST) {
							/* the alpha test (src/waveform.c:247-248) as the increment: min(alpha, 1), in the low half for the even
							 * column and in the high half for the odd one -- one SDWA instruction each, no exec mask, and the whole
							 * slot is one basic block (adding 0 for a transparent texel changes nothing) */
							trio_count2<CH, VS>(tab, p0, p1, lane_base, scope_min_byte3_lo(p0, 1u), scope_min_byte3_hi(p1, 1u));
						} else {
							if ((p0 >> 24) != 0) /* src/waveform.c:247-248 */
								trio_count<CH, VS>(tab, p0, lane_base, 1u);
							if ((p1 >> 24) != 0)
								trio_count<CH, VS>(tab, p1, lane_base, 65536u);
						}
					}
				}
				if (TRIO_EXP & 4) {
					key[0] = vss_bin(p0);
					key[1] = vss_bin(p1);
				} else if (TRIO_PAIR) {
					scope_k0_cell_uv<1>(key, k0, pu, pv);
				} else {
					scope_k0_cell<2, true>(key, k0, true, false); /* wave votes inside: every lane goes through, valid or not */
				}
				if (((g_b + (u32)d) & 3u) == 0) { /* content is coherent: the verdict holds for four loads */
					const bool same = TRIO_PAIR ? ((pu[0] == scope_perm(0u, pu[0], 0x01000302u)) & (pv[0] == scope_perm(0u, pv[0], 0x01000302u)))
									 : key[0] == key[1];
					const u32 flat_lanes = (u32)__popcll(__ballot(!v || same));
					absorb = TRIO_SCALAR_VERDICT ? (u32)__builtin_amdgcn_readfirstlane((int)(flat_lanes * 64u >= wave_w * TRIO_RLE_LO))
								     : (u32)(flat_lanes * 64u >= wave_w * TRIO_RLE_LO); /* three quarters of the wave: 48 lanes */
					wflat = flat_lanes >= wave_w;
					/* ONE mode word for the pair form (noise asks ONE question per slot): 0 plain adds; TRIO_MODE_ABSORB the per-lane
					 * absorb mode -- where the WHOLE wave is flat (TRIO_RLE_HI; a one-lane wave of the CPU emulation always is); from
					 * three quarters of the wave up to that the run-length form, with its whole-wave question behind an all-flat vote */
					if (TRIO_PAIR) {
						const bool flat_enough = flat_lanes * 64u >= wave_w * TRIO_RLE_LO;
						const bool to_absorb = TRIO_ABSORB && (!TRIO_RLE || flat_lanes * 64u >= TRIO_RLE_HI * wave_w);
						mode = (u32)__builtin_amdgcn_readfirstlane((int)((TRIO_RLE && rle_force != 0) ? TRIO_MODE_RLE
												 : !flat_enough	      ? 0u
												 : to_absorb	      ? TRIO_MODE_ABSORB
												 : !TRIO_RLE	      ? 0u
												 : wflat	      ? TRIO_MODE_RLE_WHOLE
														      : TRIO_MODE_RLE));
					}
				}
				/* ONE structured exit from here on (no early return): with several ways out the compiler copied the absorb mode's
				 * state registers at every join -- sixteen v_mov per slot on the path noise takes, 6 % of the kernel (visit a6) */
				if (TRIO_PAIR) {
					const u32 m = (u32)__builtin_amdgcn_readfirstlane((int)mode);
					if (__builtin_expect(m != 0u, 0)) { /* unlikely: laid out behind the loop */
						if (TRIO_EXP & 2) {
						} else if (TRIO_RLE && (m & TRIO_MODE_RLE) && (FAST || !TRIO_RLE_FASTONLY)) {
							trio_rle_pair<TRIO_HASH, FAST>(tab, pu[0], pv[0], v, lane, wave_w, m == TRIO_MODE_RLE_WHOLE); /* every lane goes through */
						} else if (v) {
							if (TRIO_ABSORB == 1)
								trio_absorb_pair<TRIO_HASH>(tab, S, pu[0], pv[0]);
							else if (TRIO_ABSORB)
								vss_absorb_pair<TRIO_HASH>(tab, st, lane_u, lane_v, pu[0], pv[0]);
							else
								vss_add_pair<TRIO_HASH>(tab, pu[0], pv[0]); /* (a checked slot under TRIO_RLE_FASTONLY) */
						}
					} else if (v && !(TRIO_EXP & 2)) {
						vss_add_pair<TRIO_HASH>(tab, pu[0], pv[0]);
					}
				} else if (v && !(TRIO_EXP & 2)) {
					if (absorb) { /* the key forms (TRIO_COMPACT 0; A/B) */
						bool e0 = key[0] == st.colour, e1 = key[1] == st.colour;
						if (!(e0 | e1)) {
							if (st.count)
								vss_add<TRIO_HASH>(tab, st.colour, st.count);
							st.colour = key[0];
							st.count = 0;
							e0 = true;
							e1 = key[1] == key[0];
						}
						st.count += (u32)e0 + (u32)e1;
						if (!e0)
							vss_add<TRIO_HASH>(tab, key[0], 1);
						if (!e1)
							vss_add<TRIO_HASH>(tab, key[1], 1);
					} else {
						vss_add<TRIO_HASH>(tab, key[0], 1);
						vss_add<TRIO_HASH>(tab, key[1], 1);
					}
				}
			};
#pragma unroll
			for (int d = 0; d < D; d++) {
				if (TRIO_FASTSLOT && strip_whole && g_b + (u32)d < nv_min) /* wave-uniform */
					slot(std::true_type{}, d);
				else
					slot(std::false_type{}, d);
				TRIO_ISSUE(d); /* ONE issue point per slot, behind both forms: the load queue has the same shape (and the same
						* registers) on every path */
			}
		}
		if (TRIO_WABSORB && wn) { /* the waveform's pending pair belongs to this strip's table */
			if ((wp0 >> 24) != 0)
				trio_count<CH, VS>(tab, wp0, lane_base, wn);
			if ((wp1 >> 24) != 0)
				trio_count<CH, VS>(tab, wp1, lane_base, wn << 16);
			wn = 0;
		}
		const bool last_strip = s_b + 1u == nst;
		if (last_strip)
			(TRIO_PAIR && TRIO_ABSORB == 1) ? vss_flush_packed<TRIO_HASH>(tab, S) : vss_flush_lane<TRIO_HASH>(tab, st); /* the lanes' pending texels, before the table is read out */
		TRIO_T(tb0);
		__syncthreads();
		TRIO_T(tb1);
		TRIO_T_ADD(t_barrier, tb0, tb1);
		{
			/* the strip's 256 x 16 waveform texels: thread -> (value, group of four columns); the reader of a counter zeroes it */
			const u32 xs = (first + s_b * stride) * (u32)SW;
			const u32 groups = (min(W - xs, (u32)SW)) >> 2; /* groups of four columns that exist */
			u32 o[NV][4] = {};
#pragma unroll
			for (u32 c = 0; c < 3; c++) {
				if (!((CH >> c) & 1))
					continue;
#pragma unroll
				for (u32 j = 0; j < NV; j++) {
					uint2 *w = (uint2 *)&wt[slot_of[c] * CHW + (value0 + (1024u >> GL) * j) * VS + 2u * g4];
					uint2 q = w[0];
					w[0] = make_uint2(0, 0);
#pragma unroll
					for (u32 r = 1; r < REP; r++) { /* a column's total over the replicas is at most H <= 65 535: the halves never carry */
						const uint2 qr = w[(P / 2u) * r];
						w[(P / 2u) * r] = make_uint2(0, 0);
						q.x += qr.x;
						q.y += qr.y;
					}
					o[j][0] |= min(q.x & 0xffffu, 255u) << (8u * c);
					o[j][1] |= min(q.x >> 16, 255u) << (8u * c);
					o[j][2] |= min(q.y & 0xffffu, 255u) << (8u * c);
					o[j][3] |= min(q.y >> 16, 255u) << (8u * c);
					if (HIS) /* columns past the right edge were never counted: their counters are zero */
						hs[c][j] += (q.x & 0xffffu) + (q.x >> 16) + (q.y & 0xffffu) + (q.y >> 16);
				}
			}
			if (g4 < groups) {
				uint8_t *table = wvs_dst + (size_t)frame * ((size_t)SCOPE_WVS_ROWS * W * 4);
#pragma unroll
				for (u32 j = 0; j < NV; j++) {
					const scope_v4 t = {o[j][0], o[j][1], o[j][2], o[j][3]};
					if (TRIO_STORE_SC1)
						scope_llvm_raw_buffer_store_v4i32(scope_i4{(int)o[j][0], (int)o[j][1], (int)o[j][2], (int)o[j][3]},
										  scope_buf_make(table, SCOPE_WVS_ROWS * W * 4u),
										  (int)(((xs + (g4 << 2)) + (255u - (value0 + (1024u >> GL) * j)) * W) * 4u), 0, 16);
					else
						__builtin_nontemporal_store(t, (scope_v4 *)(table + (size_t)(xs + (g4 << 2)) * 4) +
										       (size_t)(255u - (value0 + (1024u >> GL) * j)) * (W >> 2));
				}
			}
		}
		TRIO_T(tb2);
		TRIO_T_ADD(t_readout, tb1, tb2);
		if (last_strip)
			break;
		__syncthreads();
		TRIO_T(tb3);
		TRIO_T_ADD(t_barrier, tb2, tb3);
	}
	if (TRIO_PRIO)
		__builtin_amdgcn_s_setprio(0); /* the slice's epilogue (partial table, histogram atomics, ticket) at the usual priority */
#ifdef TRIO_TRACE
	if (lane == 0) {
		unsigned long long *d = &scope_trio_dbg[((size_t)(blockIdx.x & 255u) * 16u + wave) * 4u];
		d[0] = wall_clock64() - t_start;
		d[1] = t_barrier;
		d[2] = t_readout;
		d[3] = t_clamp;
	}
#endif
#undef TRIO_ISSUE
#undef TRIO_OK

	/* ---- the slice's vectorscope table: final (one slice per frame), a sparse partial, or the frame's slices meeting inside the
	 * launch -- as scope_vss_kernel: vss_write_out (scope_device.h).  The arrival flag lives in the waveform table's space (read
	 * out by now; the histogram's hand-off below takes the same space behind a barrier of its own). ---------- */
	vss_write_out(
		[&](u32 g, uint4 &lo, uint4 &hi) {
			const u32 row = 255u - (g >> 4);
			lo = make_uint4(0, 0, 0, 0);
			hi = lo;
			if (TRIO_HASH == 3) {
				/* output group (row, k) = bins u in [16 k, 16 k + 15] = words 8 k - 4 .. 8 k + 3 of table row `row` - 16; what lies
				 * outside the table (rows 0..15 and 241..255, u < 8, u >= 240) no colour reaches: zero */
				const u32 k16 = g & 15u;
				if (row - VSS3_V0 < VSS3_ROWS && k16 != 15u) {
					const u32 w0 = (row - VSS3_V0) * VSS3_PITCH + (k16 << 3);
					if (VSS3_PITCH % 4u == 0u) {
						hi = *(const uint4 *)&vt[w0];
						if (k16 != 0u)
							lo = *(const uint4 *)&vt[w0 - 4u];
					} else { /* a pitch that is only even: 8-byte loads (once per slice) */
						const uint2 h0 = *(const uint2 *)&vt[w0], h1 = *(const uint2 *)&vt[w0 + 2u];
						hi = make_uint4(h0.x, h0.y, h1.x, h1.y);
						if (k16 != 0u) {
							const uint2 l0 = *(const uint2 *)&vt[w0 - 4u], l1 = *(const uint2 *)&vt[w0 - 2u];
							lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
						}
					}
				}
			} else {
				const u32 w0 = TRIO_HASH == 2 ? row * VSS_PITCH2 + ((g & 15u) << 3) : (row << 7) + (((g & 15u) << 3) ^ ((row & 15u) << 3));
				lo = *(const uint4 *)&vt[w0];
				hi = *(const uint4 *)&vt[w0 + 4];
			}
		},
		frame, slice, slices, vss_dst, vss_mask, fold, &tab[TRIO_WVS_BASE / 4u + 8u]);
	if (HIS) {
		/* the slice's totals -> the frame's accumulator, then the arrival ticket; the workgroup that completes the frame moves the
		 * sums out and leaves accumulator and ticket zero.  The column groups of a value meet in LDS first (the waveform
		 * table's space, read out by now): ONE device atomic per value and channel and workgroup.  (Four lanes of a wave
		 * instruction on one address cost dearly at the memory side: a one-frame launch of 240 workgroups took 440 us that way.) */
		u32 *acc_g = his.acc + (size_t)frame * SCOPE_HIS_WORDS;
		__syncthreads();
#pragma unroll
		for (u32 c = 0; c < 3; c++)
#pragma unroll
			for (u32 j = 0; j < NV; j++)
				tab[(c * NV + j) * 1024u + threadIdx.x] = hs[c][j]; /* = [c][value][group]: item j's value is (1024 >> GL) j further */
		__syncthreads();
		for (u32 t2 = threadIdx.x; t2 < 768u; t2 += blockDim.x) {
			const u32 c = t2 >> 8, val = t2 & 255u;
			u32 sum = 0;
#pragma unroll
			for (u32 i = 0; i < NV; i++) {
				const uint4 q = *(const uint4 *)&tab[c * NV * 1024u + (val << GL) + 4u * i];
				sum += q.x + q.y + q.z + q.w;
			}
			if (sum && (((CH & his.chmask) >> c) & 1u))
				__hip_atomic_fetch_add(&acc_g[(2u - c) * 256u + val], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		/* (the barrier inside scope_last_arrival comes after every thread's reads of the table: word 0 may carry the verdict) */
		if (scope_last_arrival(&his.tickets[frame], slices, &tab[0])) {
			for (u32 t2 = threadIdx.x; t2 < SCOPE_HIS_WORDS; t2 += blockDim.x) {
				const u32 val = t2 >> 2, slot = t2 & 3u;
				his.out[(size_t)frame * SCOPE_HIS_WORDS + t2] =
					__hip_atomic_exchange(&acc_g[slot * 256u + val], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			if (threadIdx.x == 0)
				__hip_atomic_store(&his.tickets[frame], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

namespace {

struct TrioArgs {
	const ScopeFrameSet *fs;
	u32 strips, slices;
	ScopeK0 k0;
	uint8_t *wvs_dst;
	ScopeWvsHis his;
	uint8_t *vss_dst, *vss_mask;
	u32 rle_force;
	VssFold fold;
	hipStream_t st;
	const ScopeTiming *tm;
};
template <int CH, bool HIS, int SW> hipError_t trio_variant(const TrioArgs *a)
{
	constexpr u32 nch = (CH & 1) + ((CH >> 1) & 1) + ((CH >> 2) & 1);
	constexpr u32 lds = TRIO_LDS_BYTES(nch);
	(void)nch; /* the compact map has one size for every mask */
	if (!a)
		return hipFuncSetAttribute((const void *)scope_trio_kernel<CH, HIS, SW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	SCOPE_LAUNCH((scope_trio_kernel<CH, HIS, SW>), dim3(a->fs->n_frames * a->slices), dim3(SCOPE_BLOCK), lds, a->st, a->tm, *a->fs,
		     a->strips, a->slices, a->k0, a->wvs_dst, a->his, a->vss_dst, a->vss_mask, a->rle_force, a->fold);
	return hipGetLastError();
}
template <bool HIS, int SW> hipError_t trio_pick_ch(u32 chmask, const TrioArgs *a)
{
	switch (chmask & 7u) {
	case 1:
		return trio_variant<1, HIS, SW>(a);
	case 2:
		return trio_variant<2, HIS, SW>(a);
	case 3:
		return trio_variant<3, HIS, SW>(a);
	case 4:
		return trio_variant<4, HIS, SW>(a);
	case 5:
		return trio_variant<5, HIS, SW>(a);
	case 6:
		return trio_variant<6, HIS, SW>(a);
	default:
		return trio_variant<7, HIS, SW>(a);
	}
}
template <bool HIS> hipError_t trio_pick_sw(u32 sw, u32 chmask, const TrioArgs *a)
{
	return sw == 32u && TRIO_COMPACT ? trio_pick_ch<HIS, TRIO_COMPACT ? 32 : 16>(chmask, a) : trio_pick_ch<HIS, 16>(chmask, a);
}

} // namespace

hipError_t scope_trio_init(void)
{
	hipError_t e = hipSuccess;
	for (u32 sw = 16; sw <= 32; sw += 16)
		for (u32 chmask = 1; chmask <= 7; chmask++) {
			if (e == hipSuccess)
				e = trio_pick_sw<false>(sw, chmask, NULL);
			if (e == hipSuccess)
				e = trio_pick_sw<true>(sw, chmask, NULL);
		}
	return e;
}

/* The compact vectorscope table (HASH 3) holds the bins U in [8, 239], V in [16, 240] and nothing else: a chroma outside that
 * rectangle would be counted in the waveform's table or in a neighbouring row without any fault.  Today's two matrices stay
 * inside (tests/test_oracle.py::test_k0_chroma_range, test_dock_kernel_all_colours); this is the launch-time guard for
 * a third one, a full-range option or other coefficients: the matrix rows are linear, so their extremes over all colours lie
 * at the eight corners of the RGB cube, evaluated here in the kernel's own fixed point with the guard band on either side
 * (a result the kernel recomputes in fp32 moves by one step at most). */
static bool trio_chroma_fits(const ScopeK0 &k)
{
	for (int corner = 0; corner < 8; corner++) {
		const long long r = (corner & 1) ? 255 : 0, g = (corner & 2) ? 255 : 0, b = (corner & 4) ? 255 : 0;
		const long long su = k.fu[0] * r + k.fu[1] * g + k.fu[2] * b + (long long)(uint32_t)k.fu[3];
		const long long sv = k.fv[0] * r + k.fv[1] * g + k.fv[2] * b + (long long)(uint32_t)k.fv[3];
		for (long long d = -2 * SCOPE_K0_FX_GUARD; d <= 0; d += 2 * SCOPE_K0_FX_GUARD) {
			const long long u = (su + d) >> SCOPE_K0_FX_SHIFT, v = (sv + d) >> SCOPE_K0_FX_SHIFT;
			if (u < (long long)VSS3_U0 || u > 239 || v < (long long)VSS3_V0 || v >= (long long)(VSS3_V0 + VSS3_ROWS))
				return false;
		}
	}
	return true;
}

/* the verdict per matrix the library knows, worked out once (scope_k0_matrix: 1 -> 601, anything else -> 709) */
static bool trio_chroma_fits_cs(int colorspace)
{
	static const bool ok[2] = {trio_chroma_fits(scope_k0_matrix(1)), trio_chroma_fits(scope_k0_matrix(2))};
	return !TRIO_COMPACT || ok[colorspace == 1 ? 0 : 1];
}

uint32_t scope_trio_chroma_selftest(void)
{
	return (trio_chroma_fits(scope_k0_matrix(1)) ? 1u : 0u) | (trio_chroma_fits(scope_k0_matrix(2)) ? 2u : 0u) |
	       (!trio_chroma_fits(scope_k0_matrix_full_range(1)) ? 4u : 0u) | (!trio_chroma_fits(scope_k0_matrix_full_range(2)) ? 8u : 0u);
}

bool scope_trio_fits(const ScopeFrameSet &fs, const uint8_t *wvs_dst, const ScopeTuning &tn)
{
	if (!trio_chroma_fits_cs(1) || !trio_chroma_fits_cs(2))
		return false; /* the two-pass path then (scope_api.hip): its tables hold every bin */
	/* the vector shape, a column's rows within the waveform's 16-bit counters, 32-bit offsets */
	return scope_wvs_vector_shape(fs, wvs_dst, tn) && fs.height >= 1 && fs.height <= 65535u &&
	       (u64)fs.height * fs.linesize < (1ull << 31);
}

/* 32-column strips (whole 128-byte lines, one waveform replica) where every CU still gets at least four of them; 16-column
 * strips with replicas below that: the reference's call shape (one frame, the ring of three) then spreads over 240 workgroups */
uint32_t scope_trio_strip_width(const ScopeFrameSet &fs, uint32_t num_cu, const ScopeTuning &tn)
{
	if (!TRIO_COMPACT || tn.trio_strip == 16)
		return 16u;
	if (tn.trio_strip == 32)
		return 32u;
	return (u64)fs.n_frames * ((fs.width + 31u) / 32u) >= 4ull * num_cu ? 32u : 16u;
}

uint32_t scope_trio_slices(const ScopeFrameSet &fs, uint32_t num_cu, uint32_t strip_width)
{
	/* one workgroup per CU, every slice the same number of strips where that divides: 64 frames -> 4 slices of 30 strips of 32
	 * columns, three frames -> 80 slices of three 16-column strips, one -> 240 of one; even counts so that 16-column slices pair
	 * up (see the kernel) */
	const u32 strips = (fs.width + strip_width - 1u) / strip_width;
	u32 s0 = max(1u, num_cu / max(1u, fs.n_frames));
	s0 = min(min(s0, strips), 256u);
	const u32 per = (strips + s0 - 1u) / s0;
	u32 s = (strips + per - 1u) / per;
	/* only 16-column slices pair up; an odd count of 32-column slices is fine, and rounding it up could exceed the strips there are
	 * (three strips -> four slices, one of them empty: legal for the kernel -- it writes an empty partial and takes its ticket --
	 * but 150 KiB of LDS zeroed for nothing) */
	if (strip_width == 16u && s > 1u && (s & 1u))
		s++;
	return min(s, 256u);
}

hipError_t scope_launch_trio(const ScopeFrameSet &fs, uint32_t slices, uint32_t strip_width, uint32_t chmask, int colorspace,
			     uint8_t *wvs_dst, const ScopeHisSink *his, uint8_t *vss_dst, uint8_t *vss_mask, hipStream_t st,
			     const ScopeTiming *tm, uint32_t rle_force, const ScopeVssFoldSink *fold)
{
	if ((chmask & 7u) == 0 || slices == 0 || slices > 256u || (slices > 1 && !vss_mask) || (strip_width != 16u && strip_width != 32u))
		return hipErrorInvalidValue;
	if (!trio_chroma_fits_cs(colorspace))
		return hipErrorInvalidValue; /* never a silent write outside the compact table (the verdict is cached per matrix) */
	if (fold && (slices < 2 || slices > VSS_FOLD_MAX || !vss_mask))
		return hipErrorInvalidValue;
	const VssFold vf = fold ? VssFold{fold->out, fold->tickets, fold->packed16 ? 1u : 0u} : VssFold{NULL, NULL, 0u};
	ScopeWvsHis h = {NULL, NULL, NULL, 0};
	if (his)
		h = {his->acc, his->tickets, his->out, his->chmask & chmask & 7u};
	const TrioArgs a = {&fs, (fs.width + strip_width - 1u) / strip_width, slices, scope_k0_matrix(colorspace), wvs_dst, h, vss_dst,
			    slices > 1 ? vss_mask : NULL, rle_force, vf, st, tm};
	return his ? trio_pick_sw<true>(strip_width, chmask, &a) : trio_pick_sw<false>(strip_width, chmask, &a);
}
