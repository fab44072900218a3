/* tests/emul/scope_vec.h -- CPU stand-in of csrc/scope_vec.h (TEST INFRASTRUCTURE ONLY). */
#pragma once
struct scope_f2 {
	float x, y;
};
static inline scope_f2 scope_f2_make(float a, float b)
{
	return scope_f2{a, b};
}
static inline float scope_f2_lo(scope_f2 v)
{
	return v.x;
}
static inline float scope_f2_hi(scope_f2 v)
{
	return v.y;
}
static inline scope_f2 operator+(scope_f2 a, scope_f2 b)
{
	return scope_f2{a.x + b.x, a.y + b.y};
}
static inline scope_f2 operator-(scope_f2 a, scope_f2 b)
{
	return scope_f2{a.x - b.x, a.y - b.y};
}
static inline scope_f2 operator*(scope_f2 a, scope_f2 b)
{
	return scope_f2{a.x * b.x, a.y * b.y};
}
template <bool NT, typename V> static inline V scope_load_base(const unsigned char *base, unsigned lane_off)
{
	return *(const V *)(base + lane_off);
}
/* the emulated wave has one active lane */
static inline unsigned scope_load_row_nt(const unsigned char *row, unsigned lane_off)
{
	return *(const unsigned *)(row + lane_off);
}

struct scope_u2 {
	unsigned x, y;
};
static inline scope_u2 scope_load_row2_nt(const unsigned char *row, unsigned lane_off)
{
	const unsigned *p = (const unsigned *)(row + lane_off);
	return scope_u2{p[0], p[1]};
}

static inline scope_u2 scope_load_row2(const unsigned char *row, unsigned lane_off)
{
	return scope_load_row2_nt(row, lane_off);
}

struct scope_buf {
	const unsigned char *base;
	unsigned bytes;
};
static inline scope_buf scope_buf_make(const unsigned char *base, unsigned bytes)
{
	return scope_buf{base, bytes};
}
template <bool NT> static inline scope_u2 scope_buf_load2(scope_buf buf, unsigned off)
{
	if ((unsigned long long)off + 8 > buf.bytes) /* the hardware's range check: zeros */
		return scope_u2{0, 0};
	const unsigned *p = (const unsigned *)(buf.base + off);
	return scope_u2{p[0], p[1]};
}

/* a 16-byte store through a buffer resource, whatever its cache flavour (aux) */
struct scope_i4 {
	int x, y, z, w;
};
static inline void scope_llvm_raw_buffer_store_v4i32(scope_i4 d, scope_buf buf, int off, int, int)
{
	if ((unsigned long long)(unsigned)off + 16 <= buf.bytes)
		__builtin_memcpy((unsigned char *)buf.base + (unsigned)off, &d, 16);
}
/* the hand-off vocabulary: the emulation's memory is one coherent array */
struct scope_u4 {
	unsigned x, y, z, w;
};
static inline void scope_buf_store16_sc1(scope_buf buf, unsigned off, scope_u4 v)
{
	if ((unsigned long long)off + 16 <= buf.bytes)
		__builtin_memcpy((unsigned char *)buf.base + off, &v, 16);
}
static inline scope_u4 scope_buf_load16_sc1(scope_buf buf, unsigned off)
{
	scope_u4 v = {0, 0, 0, 0};
	if ((unsigned long long)off + 16 <= buf.bytes)
		__builtin_memcpy(&v, buf.base + off, 16);
	return v;
}
static inline void scope_buf_store1_sc1(scope_buf buf, unsigned off, unsigned byte)
{
	if (off < buf.bytes)
		((unsigned char *)buf.base)[off] = (unsigned char)byte;
}
static inline unsigned scope_buf_load1_sc1(scope_buf buf, unsigned off)
{
	return off < buf.bytes ? buf.base[off] : 0u;
}

static inline unsigned scope_perm(unsigned hi, unsigned lo, unsigned sel)
{
	const unsigned long long src = ((unsigned long long)hi << 32) | lo;
	unsigned out = 0;
	for (int i = 0; i < 4; i++) {
		const unsigned c = (sel >> (8 * i)) & 0xffu;
		unsigned byte;
		if (c <= 7)
			byte = (unsigned)(src >> (8 * c)) & 0xffu;
		else if (c <= 11)
			byte = ((src >> (16 * (c - 8) + 15)) & 1u) ? 0xffu : 0u;
		else
			byte = c == 12 ? 0u : 0xffu;
		out |= byte << (8 * i);
	}
	return out;
}

static inline unsigned scope_pk_min_u16(unsigned a, unsigned b)
{
	const unsigned lo = (a & 0xffffu) < (b & 0xffffu) ? (a & 0xffffu) : (b & 0xffffu);
	const unsigned hi = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16);
	return lo | (hi << 16);
}

template <unsigned TT> static inline unsigned scope_bitop3(unsigned a, unsigned b, unsigned c)
{
	unsigned out = 0;
	for (int i = 0; i < 32; i++) {
		const unsigned idx = (((a >> i) & 1u) << 2) | (((b >> i) & 1u) << 1) | ((c >> i) & 1u);
		out |= ((TT >> idx) & 1u) << i;
	}
	return out;
}

/* one-lane waves: a lane is a DPP row of its own -- there is no lane below, and no bit of a ballot belongs to one.
 * SCOPE_EMUL_WAVE64: row_shr:1 within rows of 16 lanes (bound_ctrl off: the first lane of a row keeps `fill`), v_mbcnt. */
static inline unsigned scope_lane_below(unsigned x, unsigned fill)
{
	if (emul::wave64()) {
		uint64_t m;
		const unsigned lane = threadIdx.x & 63u;
		const uint64_t *v = emul::wave_gather(x, &m, emul::SITE_BELOW);
		return (lane & 15u) == 0u || !((m >> (lane - 1u)) & 1u) ? fill : (unsigned)v[lane - 1u];
	}
	return fill;
}
/* the value of the wave's first live lane, for a value that is NOT wave-uniform yet (the emulation's __builtin_amdgcn_readfirstlane is
 * the identity: the kernels use that one on uniform values only) */
static inline unsigned scope_first_lane(unsigned x)
{
	if (emul::wave64()) {
		uint64_t m;
		const uint64_t *v = emul::wave_gather(x, &m, emul::SITE_READFIRST);
		return (unsigned)v[__builtin_ctzll(m)];
	}
	return x;
}
static inline unsigned scope_lane_below0(unsigned x)
{
	return scope_lane_below(x, 0u);
}
static inline unsigned scope_lanes_below(unsigned long long mask)
{
	if (emul::wave64())
		return (unsigned)__builtin_popcountll(mask & ((1ull << (threadIdx.x & 63u)) - 1ull));
	return 0;
}
/* the emulated wave has one lane = one DPP row of its own */
static inline void scope_row16_add(unsigned *dst, unsigned x)
{
	if (x)
		__atomic_fetch_add(dst, x, __ATOMIC_RELAXED);
}

static inline void scope_drain_vmem()
{
}

static inline int scope_mul24_byte1(unsigned x, int f)
{
	return (int)((x >> 8) & 0xffu) * f;
}
static inline int scope_mul24_byte2(unsigned x, int f)
{
	return (int)((x >> 16) & 0xffu) * f;
}
static inline unsigned scope_umul24_byte1(unsigned x, unsigned f)
{
	return ((x >> 8) & 0xffu) * f;
}
static inline unsigned scope_umul24_byte2(unsigned x, unsigned f)
{
	return ((x >> 16) & 0xffu) * f;
}
/* the emulated LDS block is an ordinary array: byte offset -> word */
static inline void scope_lds_add_at(unsigned *tab, unsigned byte, unsigned n)
{
	__atomic_fetch_add(tab + (byte >> 2), n, __ATOMIC_RELAXED);
}
static inline void scope_lds_is_at_zero(unsigned *)
{
}

static inline unsigned scope_pk_lshr_u16(unsigned a, unsigned n)
{
	return ((a & 0xffffu) >> n) | ((a >> 16) >> n << 16);
}
static inline unsigned scope_pk_mad_u16(unsigned a, unsigned b, unsigned c)
{
	const unsigned lo = ((a & 0xffffu) * (b & 0xffffu) + (c & 0xffffu)) & 0xffffu;
	const unsigned hi = ((a >> 16) * (b >> 16) + (c >> 16)) & 0xffffu;
	return lo | (hi << 16);
}
static inline unsigned scope_udot4(unsigned a, unsigned b, unsigned c)
{
	for (int i = 0; i < 4; i++)
		c += ((a >> (8 * i)) & 0xffu) * ((b >> (8 * i)) & 0xffu);
	return c;
}
static inline unsigned scope_min_byte3_lo(unsigned x, unsigned cap)
{
	return (x >> 24) < cap ? (x >> 24) : cap;
}
static inline unsigned scope_min_byte3_hi(unsigned x, unsigned cap)
{
	return scope_min_byte3_lo(x, cap) << 16;
}
static inline unsigned scope_shl_word0(unsigned count, unsigned value)
{
	return (value & 0xffffu) << (count & 31u);
}
static inline unsigned scope_shl_word1(unsigned count, unsigned value)
{
	return (value >> 16) << (count & 31u);
}
static inline unsigned scope_shl_by_byte0(unsigned count, unsigned value)
{
	return value << (count & 31u);
}
static inline unsigned scope_shl_by_byte2(unsigned count, unsigned value)
{
	return value << ((count >> 16) & 31u);
}

static inline int scope_pin_vgpr(int uniform)
{
	return uniform;
}

static inline unsigned scope_wave_sum(unsigned x)
{
	if (emul::wave64()) {
		uint64_t m;
		unsigned sum = 0;
		const uint64_t *v = emul::wave_gather(x, &m, emul::SITE_SUM);
		for (unsigned l = 0; l < 64; l++)
			if ((m >> l) & 1u)
				sum += (unsigned)v[l];
		return sum;
	}
	return x;
}
