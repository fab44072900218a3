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
 * The dock's three scopes from ONE read, for ANY number of frames: scope_trio_kernel (round 3).
 *
 * scope_dock_kernel (scope_kernels.hip, round 2) reads a frame once but only where one round of strips covers the launch (one 4K frame), because it
 * time-shares the LDS between the waveform's and the vectorscope's table and has to hold a strip's texels in registers in
 * between; the ring of three and every batch fell back to two reads of the planes (1.91 x the algorithmic traffic).
 *
 * Here BOTH tables are resident: the vectorscope's 128 KiB (16-bit counters, swizzled rows, as scope_vss_kernel) plus a
 * waveform table of ONE replica for a 16-column strip -- 256 values x 8 column pairs x 4 bytes = 8 KiB per channel, 24 KiB
 * for RGB -- 152 KiB of the CU's 160.  A workgroup owns the strips of one slice of one frame and streams them like any
 * other kernel here: every texel is loaded once, counted in the waveform table (alpha test, src/waveform.c:247-248), sent
 * through the fixed-point K0 and counted in the vectorscope table, with TRIO_D loads per lane in flight and no register-held
 * strip.  Per strip: barrier, the waveform's 256 x 16 output texels leave (the thread that reads a counter zeroes it, the
 * histogram's totals accumulate in three registers per thread over all strips of the slice), barrier; the loads of the next
 * strip are in flight across it.  The vectorscope's clamp pass runs before the round of loads that could overflow 16 bits.  Per
 * slice: ONE vectorscope partial (sparse, as scope_vss_kernel writes them; the final table when the frame has one slice)
 * and <= 768 device atomics for the histogram.  64 frames: 4 slices of 60 strips per frame; three frames: 80 slices of
 * three strips; one frame: 240 slices of one -- the dock kernel's shape.
 *
 * What the single replica costs: a half wave is 4 rows x 8 column pairs, word = value * 8 + pair, so two rows whose values
 * agree mod 4 meet in one bank (2.1 x the conflict-free time on noise by the bank model of profiles/round1/
 * lds_atomic_microbench.txt) -- LDS issue, not HBM, bounds this kernel.  The eight rows of a wave instruction lie H/8 apart
 * (as in scope_dock_kernel), so that the equal values of vertically neighbouring texels do not meet on one ADDRESS.
 *
 * A strip's row segment is 64 bytes, half a cache line.  Slices come in pairs that split a range of strips even / odd, and
 * the two workgroups of a pair are b and b + 8 (same XCD, same L2, same moment): the other half of every line is wanted by
 * the partner while it is still in the L2.
 * Tried and dropped: skipping the K0 / vote work of slots that are empty for a whole wave (one of 18 per strip at 4K; the
 * uniform branch around a slot's work cost hipcc's schedule of the round 8 %: 632 -> 685 us for 64 x 4K), deeper or
 * shallower load queues (4 / 8 / 12 per lane: 674 / 739 / 738 us against 641 at 6 -- a strip's 17 slots are rounded up to
 * whole rounds of D, and empty slots still run K0); profiles/round3/trio_variants.txt.
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
#ifndef TRIO_EXP
#define TRIO_EXP 0 /* measurement only (wrong tables): bit 0 no waveform adds, 1 no vectorscope adds, 2 no K0, 3 no clamp pass */
#endif
#ifndef TRIO_HASH
#define TRIO_HASH 0 /* the vectorscope table's layout.  0: as scope_vss_kernel's plain form.  2: rows of 136 words, a slot's two
		     * texels addressed as a pair (vss_add_pair: 4.5 VALU instructions per texel instead of 7, what gave the fused-K0
		     * vectorscope 4 %); with the waveform's 24 KiB that is all 160 KiB of the CU.  Measured here (visit p10, same
		     * process, 64 x 4K): noise 3779 -> 3758 GB/s, natural 3724 -> 3746, flat 2890 -> 2938, screen 3074 -> 2847 --
		     * this kernel waits for the LDS (the waveform's single replica), not for VALU issue slots: left at 0 */
#endif
#define TRIO_CH_BYTES 8192u
#define TRIO_VSS_WORDS (VssTable<TRIO_HASH>::words)
#define TRIO_WVS_BASE (TRIO_VSS_WORDS * 4u)

template <int CH> __device__ __forceinline__ void trio_count(u32 *tab, u32 px, u32 lane_base, u32 inc)
{
	constexpr u32 slot1 = (CH & 1), slot2 = slot1 + ((CH >> 1) & 1);
	if (CH & 1)
		scope_lds_add_at(tab, lane_base + ((px & 0xffu) << 5), inc);
	if (CH & 2)
		scope_lds_add_at(tab, lane_base + slot1 * TRIO_CH_BYTES + (((px >> 8) & 0xffu) << 5), inc);
	if (CH & 4)
		scope_lds_add_at(tab, lane_base + slot2 * TRIO_CH_BYTES + (((px >> 16) & 0xffu) << 5), inc);
}

template <int CH, bool HIS>
__global__ __launch_bounds__(SCOPE_BLOCK) void scope_trio_kernel(ScopeFrameSet fs, u32 strips, u32 slices, ScopeK0 k0_arg,
								uint8_t *wvs_dst, ScopeWvsHis his, uint8_t *vss_dst, uint8_t *vss_mask)
{
	const ScopeK0 k0 = scope_k0_pinned(k0_arg, true, false);
	constexpr u32 slot1 = (CH & 1), slot2 = slot1 + ((CH >> 1) & 1), nch = slot2 + ((CH >> 2) & 1);
	constexpr u32 slot_of[3] = {0, slot1, slot2};
	constexpr int D = TRIO_D;
	u32 *tab = scope_lds;
	scope_lds_is_at_zero(tab);
	u32 *wt = tab + TRIO_VSS_WORDS; /* [nch][256 values][8 column pairs] */
	const u32 W = fs.width, H = fs.height;
	const u32 lane = threadIdx.x & 63u;
	const u32 pair = lane & 7u, rsub = lane >> 3;
	const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const u32 nwaves = blockDim.x >> 6;

	/* ---- which strips are mine ------------------------------------------------------------------------------------ */
	const u32 b = blockIdx.x, total = gridDim.x;
	u32 frame, slice, first, last, stride;
	if ((slices & 1u) == 0) {
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
	const u32 rps = (H + 7u) >> 3;
	const u32 row0 = rsub * rps;
	const u32 G = (rps + nwaves - 1u) / nwaves;
	const u32 Gp = ((G + (u32)D - 1u) / (u32)D) * (u32)D;
	const u32 dstep = nwaves * fs.linesize;
	const u32 lane_base = TRIO_WVS_BASE + (pair << 2);

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
	(void)buf;
	(void)nv;
	(void)live_i;
	auto strip_start = [&](u32 s) {
		const u32 x0 = (first + s * stride) * 16u + 2u * pair;
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
		for (u32 i = threadIdx.x * 4; i < TRIO_VSS_WORDS + nch * (TRIO_CH_BYTES / 4u); i += blockDim.x * 4)
			*(uint4 *)&tab[i] = zero;
	}
	__syncthreads();

	VssLane st = {0xffffffffu, 0};
	u32 lane_u = 0xffffffffu, lane_v = 0xffffffffu; /* TRIO_HASH 2: the lane's colour in pair form (vss_absorb_pair) */
	(void)lane_u;
	(void)lane_v;
	bool absorb = false, wflat = false; /* wave-uniform */
	u32 wp0 = 0, wp1 = 0, wn = 0; /* the waveform's pending texel pair and how often it came (absorb mode) */
	u32 since = 0;       /* texels the vectorscope table may have taken since its last clamp pass (uniform) */
	const u32 round_px = (u32)D * blockDim.x * 2u; /* texels a round of D slots can add (upper bound: every lane valid) */
	u32 hs[3] = {0, 0, 0}; /* HIS: this thread's (value, column group) totals over the slice's strips */
	const u32 value = threadIdx.x >> 2, g4 = threadIdx.x & 3u;

	for (u32 s_b = 0; s_b < nst; s_b++) {
		const bool live_b = (first + s_b * stride) * 16u + 2u * pair < W; /* this lane's column pair exists in the strip */
		(void)live_b;
		for (u32 g_b = 0; g_b < Gp; g_b += (u32)D) {
			/* the vectorscope's 16-bit counters: a clamp pass (every counter min(c, 255): exact, the reference saturates
			 * at 255) before the round that could carry one past 65535 -- at round granularity, not per strip: a 4K strip is
			 * 34 560 texels, two strips would overflow, but five rounds of 12 288 fit, so 60 strips need 36 passes and
			 * not 59 (a pass reads the whole 128 KiB table: 5 % of the kernel when taken once per strip) */
			if (since + round_px > VSS_EPOCH_PIXELS && !(TRIO_EXP & 8)) { /* uniform across the workgroup */
				vss_flush_lane<TRIO_HASH>(tab, st); /* pending texels belong to the epoch that ends here */
				__syncthreads();
				vss_clamp_pass(tab, TRIO_VSS_WORDS);
				__syncthreads();
				since = 0;
			}
			since += round_px;
#pragma unroll
			for (int d = 0; d < D; d++) {
				const bool v = TRIO_OK(d, g_b + (u32)d);
				const u32 p0 = px[d][0], p1 = px[d][1];
				u32 key[2] = {p0, p1};
				u32 pu[1] = {0}, pv[1] = {0}; /* TRIO_HASH 2: the slot's chroma in pair form */
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
									trio_count<CH>(tab, wp0, lane_base, wn);
								if ((wp1 >> 24) != 0)
									trio_count<CH>(tab, wp1, lane_base, wn << 16);
							}
							wp0 = p0;
							wp1 = p1;
							wn = 1;
						}
					} else {
						if ((p0 >> 24) != 0) /* src/waveform.c:247-248 */
							trio_count<CH>(tab, p0, lane_base, 1u);
						if ((p1 >> 24) != 0)
							trio_count<CH>(tab, p1, lane_base, 65536u);
					}
				}
				if (TRIO_EXP & 4) {
					key[0] = vss_bin(p0);
					key[1] = vss_bin(p1);
				} else if (TRIO_HASH == 2) {
					scope_k0_cell_uv<1>(key, k0, pu, pv);
				} else {
					scope_k0_cell<2, true>(key, k0, true, false); /* wave votes inside: every lane goes through, valid or not */
				}
				if (((g_b + (u32)d) & 3u) == 0) { /* content is coherent: the verdict holds for four loads */
					const bool same = TRIO_HASH == 2 ? ((pu[0] == scope_perm(0u, pu[0], 0x01000302u)) & (pv[0] == scope_perm(0u, pv[0], 0x01000302u)))
									 : key[0] == key[1];
					const u32 flat_lanes = (u32)__popcll(__ballot(!v || same));
					absorb = flat_lanes >= 48;
					wflat = flat_lanes >= 64;
				}
				TRIO_ISSUE(d); /* ONE issue point per slot, whatever the slot held: the load queue has the same shape on every path */
				if (!v || (TRIO_EXP & 2))
					continue;
				if (TRIO_HASH == 2) {
					if (absorb)
						vss_absorb_pair(tab, st, lane_u, lane_v, pu[0], pv[0]);
					else
						vss_add_pair(tab, pu[0], pv[0]);
				} else if (absorb) {
					bool e0 = key[0] == st.colour, e1 = key[1] == st.colour;
					if (!(e0 | e1)) {
						if (st.count)
							vss_add<TRIO_HASH == 2 ? 0 : TRIO_HASH>(tab, st.colour, st.count);
						st.colour = key[0];
						st.count = 0;
						e0 = true;
						e1 = key[1] == key[0];
					}
					st.count += (u32)e0 + (u32)e1;
					if (!e0)
						vss_add<TRIO_HASH == 2 ? 0 : TRIO_HASH>(tab, key[0], 1);
					if (!e1)
						vss_add<TRIO_HASH == 2 ? 0 : TRIO_HASH>(tab, key[1], 1);
				} else {
					vss_add<TRIO_HASH == 2 ? 0 : TRIO_HASH>(tab, key[0], 1);
					vss_add<TRIO_HASH == 2 ? 0 : TRIO_HASH>(tab, key[1], 1);
				}
			}
		}
		if (TRIO_WABSORB && wn) { /* the waveform's pending pair belongs to this strip's table */
			if ((wp0 >> 24) != 0)
				trio_count<CH>(tab, wp0, lane_base, wn);
			if ((wp1 >> 24) != 0)
				trio_count<CH>(tab, wp1, lane_base, wn << 16);
			wn = 0;
		}
		const bool last_strip = s_b + 1u == nst;
		if (last_strip)
			vss_flush_lane<TRIO_HASH>(tab, st); /* the lanes' pending texels, before the table is read out */
		__syncthreads();
		{
			/* the strip's 256 x 16 waveform texels: thread -> (value, group of four columns); the reader of a counter zeroes it */
			const u32 xs = (first + s_b * stride) * 16u;
			const u32 groups = (min(W - xs, 16u)) >> 2;
			u32 o[4] = {0, 0, 0, 0};
#pragma unroll
			for (u32 c = 0; c < 3; c++) {
				if (!((CH >> c) & 1))
					continue;
				uint2 *w = (uint2 *)&wt[slot_of[c] * (TRIO_CH_BYTES / 4u) + (value << 3) + 2u * g4];
				const uint2 q = *w;
				*w = make_uint2(0, 0);
				o[0] |= min(q.x & 0xffffu, 255u) << (8u * c);
				o[1] |= min(q.x >> 16, 255u) << (8u * c);
				o[2] |= min(q.y & 0xffffu, 255u) << (8u * c);
				o[3] |= min(q.y >> 16, 255u) << (8u * c);
				if (HIS) /* columns past the right edge were never counted: their counters are zero */
					hs[c] += (q.x & 0xffffu) + (q.x >> 16) + (q.y & 0xffffu) + (q.y >> 16);
			}
			if (g4 < groups) {
				uint8_t *table = wvs_dst + (size_t)frame * ((size_t)SCOPE_WVS_ROWS * W * 4);
				const scope_v4 t = {o[0], o[1], o[2], o[3]};
				__builtin_nontemporal_store(t, (scope_v4 *)(table + (size_t)(xs + (g4 << 2)) * 4) + (size_t)(255u - value) * (W >> 2));
			}
		}
		if (last_strip)
			break;
		__syncthreads();
	}
#undef TRIO_ISSUE
#undef TRIO_OK

	/* ---- the slice's vectorscope table: final (one slice per frame) or a sparse partial, as scope_vss_kernel ---------- */
	{
		const u32 unit = frame * slices + slice;
		uint4 *out = (uint4 *)(vss_dst + (size_t)unit * SCOPE_VSS_BINS);
		uint4 *outs = (uint4 *)vss_dst + ((size_t)frame * 256u * slices + slice) * 16u;
		u32 wrote = 0, k = 0;
		for (u32 g = threadIdx.x; g < SCOPE_VSS_BINS / 16; g += blockDim.x, k++) {
			const u32 row = 255u - (g >> 4);
			const u32 w0 = TRIO_HASH == 2 ? row * VSS_PITCH2 + ((g & 15u) << 3) : (row << 7) + (((g & 15u) << 3) ^ ((row & 15u) << 3));
			const uint4 lo = *(const uint4 *)&tab[w0], hi = *(const uint4 *)&tab[w0 + 4];
			if (!vss_mask) {
				out[g] = make_uint4(vss_clamp_pair(lo.x, lo.y), vss_clamp_pair(lo.z, lo.w), vss_clamp_pair(hi.x, hi.y),
						    vss_clamp_pair(hi.z, hi.w));
			} else if ((lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) != 0) {
				const scope_v4 t = {vss_clamp_pair(lo.x, lo.y), vss_clamp_pair(lo.z, lo.w), vss_clamp_pair(hi.x, hi.y),
						    vss_clamp_pair(hi.z, hi.w)};
				__builtin_nontemporal_store(t, (scope_v4 *)outs + ((size_t)(g >> 4) * slices * 16u + (g & 15u)));
				wrote |= 1u << k;
			}
		}
		if (vss_mask)
			vss_mask[(((size_t)frame * (blockDim.x >> 4) + (threadIdx.x >> 4)) * slices + slice) * 16u + (threadIdx.x & 15u)] =
				(uint8_t)wrote;
	}
	if (HIS) {
		/* the slice's totals -> the frame's accumulator, then the arrival ticket; the workgroup that completes the frame moves the
		 * sums out and leaves accumulator and ticket zero.  The four column groups of a value meet in LDS first (the vectorscope
		 * table's space, read out by now): ONE device atomic per value and channel and workgroup.  (Four lanes of a wave
		 * instruction on one address cost dearly at the memory side: a one-frame launch of 240 workgroups took 440 us that way.) */
		u32 *acc_g = his.acc + (size_t)frame * SCOPE_HIS_WORDS;
		__syncthreads();
#pragma unroll
		for (u32 c = 0; c < 3; c++)
			tab[c * 1024u + threadIdx.x] = hs[c];
		__syncthreads();
		for (u32 t2 = threadIdx.x; t2 < 768u; t2 += blockDim.x) {
			const u32 c = t2 >> 8, val = t2 & 255u;
			const uint4 q = *(const uint4 *)&tab[c * 1024u + 4u * val];
			const u32 sum = q.x + q.y + q.z + q.w;
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
	hipStream_t st;
	const ScopeTiming *tm;
};
template <int CH, bool HIS> hipError_t trio_variant(const TrioArgs *a)
{
	constexpr u32 nch = (CH & 1) + ((CH >> 1) & 1) + ((CH >> 2) & 1);
	constexpr u32 lds = TRIO_VSS_WORDS * 4u + nch * TRIO_CH_BYTES;
	if (!a)
		return hipFuncSetAttribute((const void *)scope_trio_kernel<CH, HIS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	SCOPE_LAUNCH((scope_trio_kernel<CH, HIS>), dim3(a->fs->n_frames * a->slices), dim3(SCOPE_BLOCK), lds, a->st, a->tm, *a->fs,
		     a->strips, a->slices, a->k0, a->wvs_dst, a->his, a->vss_dst, a->vss_mask);
	return hipGetLastError();
}
template <bool HIS> hipError_t trio_pick_ch(u32 chmask, const TrioArgs *a)
{
	switch (chmask & 7u) {
	case 1:
		return trio_variant<1, HIS>(a);
	case 2:
		return trio_variant<2, HIS>(a);
	case 3:
		return trio_variant<3, HIS>(a);
	case 4:
		return trio_variant<4, HIS>(a);
	case 5:
		return trio_variant<5, HIS>(a);
	case 6:
		return trio_variant<6, HIS>(a);
	default:
		return trio_variant<7, HIS>(a);
	}
}

} // namespace

hipError_t scope_trio_init(void)
{
	hipError_t e = hipSuccess;
	for (u32 chmask = 1; chmask <= 7; chmask++) {
		if (e == hipSuccess)
			e = trio_pick_ch<false>(chmask, NULL);
		if (e == hipSuccess)
			e = trio_pick_ch<true>(chmask, NULL);
	}
	return e;
}

bool scope_trio_fits(const ScopeFrameSet &fs, const uint8_t *wvs_dst, const ScopeTuning &tn)
{
	/* the vector shape, a column's rows within the waveform's 16-bit counters, 32-bit offsets */
	return scope_wvs_vector_shape(fs, wvs_dst, tn) && fs.height >= 1 && fs.height <= 65535u &&
	       (u64)fs.height * fs.linesize < (1ull << 31);
}

uint32_t scope_trio_slices(const ScopeFrameSet &fs, uint32_t num_cu)
{
	/* one workgroup per CU, every slice the same number of strips where that divides: 64 frames -> 4 slices of 60 strips,
	 * three frames -> 80 of three, one -> 240 of one; even counts so that slices pair up (see the kernel) */
	const u32 strips = scope_wvs_strips(fs.width, 1);
	u32 s0 = max(1u, num_cu / max(1u, fs.n_frames));
	s0 = min(min(s0, strips), 256u);
	const u32 per = (strips + s0 - 1u) / s0;
	u32 s = (strips + per - 1u) / per;
	if (s > 1u && (s & 1u))
		s++;
	return min(s, 256u);
}

hipError_t scope_launch_trio(const ScopeFrameSet &fs, uint32_t slices, uint32_t chmask, int colorspace, uint8_t *wvs_dst,
			     const ScopeHisSink *his, uint8_t *vss_dst, uint8_t *vss_mask, hipStream_t st, const ScopeTiming *tm)
{
	if ((chmask & 7u) == 0 || slices == 0 || slices > 256u || (slices > 1 && !vss_mask))
		return hipErrorInvalidValue;
	ScopeWvsHis h = {NULL, NULL, NULL, 0};
	if (his)
		h = {his->acc, his->tickets, his->out, his->chmask & chmask & 7u};
	const TrioArgs a = {&fs, scope_wvs_strips(fs.width, 1), slices, scope_k0_matrix(colorspace), wvs_dst, h, vss_dst,
			    slices > 1 ? vss_mask : NULL, st, tm};
	return his ? trio_pick_ch<true>(chmask, &a) : trio_pick_ch<false>(chmask, &a);
}

