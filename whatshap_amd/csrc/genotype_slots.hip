// genotype_slots.hip -- run-fused device path of GenotypeDPTable (genotype.h; src/genotypedptable.cpp:200-441): the slot runs of
// slots.h with SUMS instead of minima -- no records, no ties.
//
// One launch = one run of consecutive columns (the planner of the phasing path in genotype_mode: one (cell, transmission
// value) per lane, 6 - log2 T lane slots, up to 3 wave slots, the reads that stay through the run select the workgroup).
//   forward chain  (src/genotypedptable.cpp:292-441): a lane holds A_{c-1}[back(x)][i]; per column: transition over the
//       previous transmission value (butterfly over the low lane bits: P(j -> i) is a product over the bits of i ^ j), store the
//       transitioned value, multiply by S_i(x) = sum_a prior_c(i, a) * cost_i(x, a), sum out the reads that END in the column;
//   backward chain (:200-289), the same runs walked from their last column to their first: a lane holds B_c[fwd(x)][i];
//       per column: store it, multiply by S_i(x), sum out the reads that START in the column, transition;
//   combine: ONE full-chip launch over (column, cell) after both chains: the genotype-likelihood sums
//       L_c[individual][genotype] = sum over x, i, a of stored forward * prior * cost * stored backward (:376-383), per-block partial
//       sums, then the normalisation of every column.  Any constant factor on a whole stored column cancels there, so the
//       chains rescale freely: every run divides what it hands on by the total of what it received.
// cost_i(x, a) = prod_p W_i(x)[p][a_p]; W splits over the slot classes into TABLES computed once per solve at full-chip width
// (geno_slot_tables): G[workgroup], V[wave], S[lane] per (column, transmission value): a column costs two LDS reads of 2P doubles
// and 2P multiplications per lane for W, then 2^(P+1) - 2 multiplications for the products of all assignments.
// The two chains are independent until the combine: they run on two streams side by side.
// Arithmetic is f64 (the reference: long double): parity to a relative tolerance (tests/test_gpu_genotype.py, rtol 1e-9).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "debug_build.h"
#include "device_pool.h"
#include "genotype.h"
#include "slots.h"

namespace whamd {

namespace {

constexpr int GS_MAXLOCAL = 12;      // local slots of a run (<= 6 lane + 3 wave)
constexpr int GS_MAXA = 16;          // allele assignments (P <= 4)
constexpr int GS_MAXGL = 1 + 3 * 4;  // normaliser + 3 genotypes of up to 4 individuals

// Per column: which reads end / start in it (local slots, the order does not matter for sums) and which slots hold a read.
struct GsCol {
	uint32_t active;                 // slots (local and grid) that hold a read in this column
	uint8_t n_end, n_start, first_of_table, last_of_table;
	uint8_t end_slot[GS_MAXLOCAL], start_slot[GS_MAXLOCAL];
};
static_assert(sizeof(GsCol) == 32, "GsCol layout");
// Per column: the read in every slot (tables kernel and combine kernel only).
struct GsRow {
	double pe[SLOT_MAXSLOTS];        // error probability of the read's entry (src/genotypecolumncostcomputer.cpp:26-48)
	uint8_t ind[SLOT_MAXSLOTS + 2];
	uint8_t allele[SLOT_MAXSLOTS + 2];   // 0 REF, 1 ALT, 2 BLANK
};
// Per run.
struct GsRun {
	uint32_t c0, ncols, g, L, lw, threads, has_prev, has_next;
	uint32_t in_occ, in_identity, out_occ, pad0;
	uint32_t in_pos[8], out_pos[8];      // entry / exit index bit of every slot (SlotRun)
	unsigned long long tab_off;          // tables of the run: G [2^g][ncols][T][E], V [2^lw][ncols][T][E], S [ncols][64][E]  (doubles, E = 2 P)
	unsigned long long store_off;        // the run's columns in the two column stores: [ncols][2^g * threads] doubles
	uint32_t v_off, s_off;               // V and S relative to tab_off
	uint32_t part_in_f, part_out_f, part_in_b, part_out_b;   // first per-wave partial sum of the exchange columns read / written (forward, backward)
	uint32_t n_part_in_f, n_part_in_b;   // how many (0: this run does not rescale -- only every GS_RESCALE-th run does)
	uint32_t emit_f, emit_b;             // 1: the neighbour rescales, leave the per-wave sums of what is handed on
};
constexpr uint32_t GS_RESCALE = 4;       // runs between two rescalings of a chain (a run shrinks the values by ~1e-10 at most: far from 1e-308)
struct GsDev {
	const GsCol* cols;        // by column
	const GsRow* rows;        // by column
	const double* prior;      // [n_cols][T][A]
	const double* rho;        // [n_cols]: P(one transmission bit differs) / P(it does not)  (src/transitionprobabilitycomputer.cpp:22-45)
	const uint8_t* gidx;      // [T][A][n_ind]
	const int8_t* h2p;        // [T][n_ind][2]
	double* tab;
	double* fstore;           // forward column store
	double* bstore;           // backward column store
	double* partials;         // per-wave sums of the exchange columns
	uint32_t T, A, P, n_ind, n_cols, pad;
	unsigned long long* dbg;  // -DWHAMD_GENO_STAMPS: cycle sums of the run kernel's phases (wave 0 of workgroup 0)
};

__device__ __forceinline__ uint32_t gs_uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t gs_pos(const uint32_t (&w)[8], uint32_t s) { return (w[s >> 2] >> ((s & 3u) * 8u)) & 31u; }

// ---- tables: blockIdx.y = run; one thread per (kind, unit, column, transmission value), E = 2 P outputs each
__global__ __launch_bounds__(256) void geno_slot_tables(GsDev G, const GsRun* __restrict__ runs) {
	const GsRun run = runs[blockIdx.y];
	const uint32_t T = G.T, P = G.P, E = 2u * P, tb = 31u - (uint32_t)__clz((int)T), nls = 6u - tb;
	const uint32_t per_unit = run.ncols * T;   // entries per workgroup / wave
	const uint32_t n_g = per_unit << run.g, n_v = per_unit << run.lw, n_s = run.ncols * 64u;
	double* __restrict__ out = G.tab + run.tab_off;
	for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n_g + n_v + n_s; idx += gridDim.x * blockDim.x) {
		uint32_t kind, unit, c, i, s0, s1, bits;
		if (idx < n_g + n_v) {
			kind = idx < n_g ? 0u : 1u;
			const uint32_t r = kind ? idx - n_g : idx;
			unit = r / per_unit; c = (r % per_unit) / T; i = r % T;
			if (kind == 0u) { s0 = run.L; s1 = run.L + run.g; } else { s0 = nls; s1 = run.L; }
			bits = unit;
		} else {
			kind = 2u;
			const uint32_t r = idx - n_g - n_v;
			c = r >> 6; unit = r & 63u; i = unit & (T - 1u);
			s0 = 0; s1 = nls; bits = unit >> tb;
		}
		const GsRow& row = G.rows[run.c0 + c];
		const uint32_t active = G.cols[run.c0 + c].active;
		double w[8];
#pragma unroll
		for (int q = 0; q < 8; ++q) w[q] = 1.0;
		for (uint32_t s = s0; s < s1; ++s) {
			if (!((active >> s) & 1u)) continue;
			const uint32_t al = row.allele[s];
			if (al > 1u) continue;   // BLANK
			const uint32_t hap = ((bits >> (s - s0)) & 1u) ^ 1u;   // bit 0 <-> haplotype 1 (src/genotypecolumncostcomputer.cpp:61)
			const uint32_t part = (uint32_t)G.h2p[((size_t)i * G.n_ind + row.ind[s]) * 2u + hap];
			const double pe = row.pe[s], ok = 1.0 - pe;
#pragma unroll
			for (int q = 0; q < 8; ++q) {
				const uint32_t p = (uint32_t)q >> 1, a = (uint32_t)q & 1u;
				w[q] *= (p == part) ? (a == al ? ok : pe) : 1.0;
			}
		}
		// (the same index arithmetic in the kernels: entry = ((unit * ncols + c) * T + i) * E for G / V, (c * 64 + lane) * E for S)
		double* e = kind == 0u ? out + ((size_t)(unit * run.ncols + c) * T + i) * E
		          : (kind == 1u ? out + run.v_off + ((size_t)(unit * run.ncols + c) * T + i) * E : out + run.s_off + ((size_t)c * 64u + unit) * E);
#pragma unroll
		for (int q = 0; q < 8; ++q) if ((uint32_t)q < E) e[q] = w[q];
	}
}

// value of lane (l ^ mask), f64; mask is wave-uniform.  Masks inside a row of 16 lanes are DPP moves (no LDS crossbar round trip).
template <int CTRL>
__device__ __forceinline__ double gs_dpp(double v) {
	const unsigned long long u = __double_as_longlong(v);
	const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)u, CTRL, 0xF, 0xF, true);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(u >> 32), CTRL, 0xF, 0xF, true);
	return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double gs_lane_xor(double v, uint32_t mask) {
	if (mask == 1u) return gs_dpp<0xB1>(v);                       // quad_perm [1,0,3,2]
	if (mask == 2u) return gs_dpp<0x4E>(v);                       // quad_perm [2,3,0,1]
	if (mask == 4u) return gs_dpp<0x1B>(gs_dpp<0x141>(v));        // row_half_mirror (i ^ 7), quad_perm [3,2,1,0] (i ^ 3)
	if (mask == 8u) return gs_dpp<0x141>(gs_dpp<0x140>(v));       // row_mirror (i ^ 15), row_half_mirror (i ^ 7)
	return __shfl_xor(v, (int)mask);
}

template <int P>
__device__ __forceinline__ void gs_products(const double (&W)[2 * P], double (&prod)[1 << P]) {
	// prod[a] = prod_p W[p][a_p], built bit by bit: 2^(P+1) - 2 multiplications
	prod[0] = W[0]; prod[1] = W[1];
#pragma unroll
	for (int p = 1; p < P; ++p) {
#pragma unroll
		for (int a = (1 << p) - 1; a >= 0; --a) {
			prod[a | (1 << p)] = prod[a] * W[2 * p + 1];
			prod[a] = prod[a] * W[2 * p];
		}
	}
}

// One run, forward (DIR 0) or backward (DIR 1).  LDS: exchange 2 x [threads] doubles | A [waves][ncols][T][E] | prior [ncols][T][A] |
// rho [ncols] | reduction scratch | GsCol [ncols].  The lane part S of W is the same for every workgroup of every run that
// shares the column: each lane reads its 2 P doubles per column straight from the table (L2), one column ahead of their use --
// copying the run's whole S table into LDS up front was most of a launch (57 KB per workgroup for a trio).
template <int TB, int P, int DIR>
__global__ __launch_bounds__(512) void geno_slot_run(GsDev G, GsRun run, const double* __restrict__ prev, double* __restrict__ cur) {
	constexpr uint32_t T = 1u << TB, E = 2u * P, A = 1u << P;
	constexpr int NLS = 6 - TB;
	extern __shared__ __attribute__((aligned(16))) double gs_smem[];
	const uint32_t w = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = gs_uni(tid >> 6);
#ifdef WHAMD_GENO_STAMPS
	const unsigned long long st0 = __builtin_readcyclecounter();
#define GS_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && G.dbg) atomicAdd(&G.dbg[DIR * 8 + (k)], __builtin_readcyclecounter() - st0); } while (0)
#else
#define GS_STAMP(k)
#endif
	const uint32_t threads = run.threads, ncols = run.ncols, L = run.L, nwaves = threads >> 6;
	const uint32_t i = lane & (T - 1u);
	const uint32_t lcell = tid >> TB, Pcell = (w << L) | lcell;
	double* xbuf = gs_smem;
	double* a_lds = xbuf + 2u * threads;
	double* pr_lds = a_lds + (size_t)nwaves * ncols * T * E;
	double* rho_lds = pr_lds + (size_t)ncols * T * A;
	double* red = rho_lds + ((ncols + 1u) & ~1u);
	GsCol* col_lds = reinterpret_cast<GsCol*>(red + 16);
	const double* __restrict__ tabG = G.tab + run.tab_off;
	const bool from_other = DIR == 0 ? run.has_prev != 0u : run.has_next != 0u;
	const uint32_t np = DIR == 0 ? run.n_part_in_f : run.n_part_in_b;   // partial sums of the entering column (0: this run does not rescale)
	double val = 1.0;   // forward: column 0 starts from 1 (:313, `prev ? ... : 1`); backward: B of the last column is 1
	double psum = 0.0;
	// Two shapes of prologue and column loop, chosen by measurement: a single individual (TB = 0) gains 8 % from every prologue load in one
	// batch and the lane tables requested three columns ahead; a pedigree (16 table words per lane and column, 114 VGPRs that way) loses 4 %.
	if constexpr (TB == 0) {
		// ---- prologue: tables, priors, descriptors, the entering value, the partial sums of the entering column.  EVERY global load is issued before
		// the first wait -- unconditional loads from clamped addresses, the values stored (or dropped) afterwards: written as `q < n ? src[q] : 0`
		// loops each copy was load, wait, store: seven L2 round trips in a row.  (Measured: chains of 50 000 columns 42.3 -> 41.9 ms -- the two chains
		// hide most of each other's prologue; the rotation of the lane tables below is what moved them, to 39.0 ms.)
		const uint32_t per_wave = ncols * T * E;   // A = G[w] * V[wave]
		const double* __restrict__ gsrc = tabG + (size_t)w * per_wave;
		const double* __restrict__ vsrc = tabG + run.v_off + (size_t)wave * per_wave;
		double gv[8], vv[8];
	#pragma unroll
		for (uint32_t u = 0; u < 8; ++u) {
			const uint32_t q = min(u * 64u + lane, per_wave - 1u);
			gv[u] = gsrc[q];
			vv[u] = vsrc[q];
		}
		constexpr uint32_t PR_N = 4, PS_N = 4;
		const uint32_t n_pr = ncols * T * A;
		const double* __restrict__ pr = G.prior + (size_t)run.c0 * T * A;
		double prv[PR_N];
	#pragma unroll
		for (uint32_t u = 0; u < PR_N; ++u) prv[u] = pr[min(u * threads + tid, n_pr - 1u)];
		const double rho_v = G.rho[run.c0 + min(tid, ncols - 1u)];
		const uint4 col_v = reinterpret_cast<const uint4*>(G.cols + run.c0)[min(tid, ncols * 2u - 1u)];
		{
			// forward reads the exchange column in its ENTRY layout, backward in its EXIT layout (the same index space: the reads that
			// continue across the boundary)
			const uint32_t occ = from_other ? (DIR == 0 ? run.in_occ : run.out_occ) : 0u;   // (nothing to read: entry 0 is fetched and dropped)
			uint32_t idx = 0;
			if (DIR == 0 && run.in_identity) idx = Pcell & occ;
			else {
	#pragma unroll
				for (int s = 0; s < SLOT_MAXSLOTS; ++s) idx |= (((Pcell & occ) >> s) & 1u) << (DIR == 0 ? gs_pos(run.in_pos, s) : gs_pos(run.out_pos, s));
			}
			val = prev[(size_t)idx * T + i];
		}
		// total of the entering column when this run rescales (every thread ends up with the same number): wave sums, then across the waves
		const uint32_t p0 = DIR == 0 ? run.part_in_f : run.part_in_b;
		double psv[PS_N];
	#pragma unroll
		for (uint32_t u = 0; u < PS_N; ++u) psv[u] = G.partials[np ? p0 + min(u * threads + tid, np - 1u) : 0u];
		// ---- (everything requested) now the copies into LDS
	#pragma unroll
		for (uint32_t u = 0; u < 8; ++u) {
			const uint32_t q = u * 64u + lane;
			if (q < per_wave) a_lds[(size_t)wave * per_wave + q] = gv[u] * vv[u];
		}
		for (uint32_t q = 512u + lane; q < per_wave; q += 64u) a_lds[(size_t)wave * per_wave + q] = gsrc[q] * vsrc[q];   // (runs beyond 512 table entries per wave)
	#pragma unroll
		for (uint32_t u = 0; u < PR_N; ++u) {
			const uint32_t q = u * threads + tid;
			if (q < n_pr) pr_lds[q] = prv[u];
		}
		for (uint32_t q = PR_N * threads + tid; q < n_pr; q += threads) pr_lds[q] = pr[q];
		if (tid < ncols) rho_lds[tid] = rho_v;
		if (tid < ncols * 2u) reinterpret_cast<uint4*>(col_lds)[tid] = col_v;
		GS_STAMP(0);   // prologue loads issued and staged
		if (!from_other) val = 1.0;
		if (np) {
	#pragma unroll
			for (uint32_t u = 0; u < PS_N; ++u) if (u * threads + tid < np) psum += psv[u];
			for (uint32_t q = PS_N * threads + tid; q < np; q += threads) psum += G.partials[p0 + q];
			for (int off = 32; off > 0; off >>= 1) psum += __shfl_xor(psum, off);
			if (lane == 0) red[wave] = psum;
		}
	} else {
		// ---- prologue: tables, priors, descriptors, the entering value, the scale of the entering column
		{
			const uint32_t per_wave = ncols * T * E;   // A = G[w] * V[wave]
			const double* __restrict__ g = tabG + (size_t)w * per_wave;
			const double* __restrict__ v = tabG + run.v_off + (size_t)wave * per_wave;
			for (uint32_t q0 = 0; q0 < per_wave; q0 += 512u) {   // eight loads in flight per lane before the first is used
				double gv[8], vv[8];
	#pragma unroll
				for (uint32_t u = 0; u < 8; ++u) {
					const uint32_t q = q0 + u * 64u + lane;
					gv[u] = q < per_wave ? g[q] : 0.0;
					vv[u] = q < per_wave ? v[q] : 0.0;
				}
	#pragma unroll
				for (uint32_t u = 0; u < 8; ++u) {
					const uint32_t q = q0 + u * 64u + lane;
					if (q < per_wave) a_lds[(size_t)wave * per_wave + q] = gv[u] * vv[u];
				}
			}
			const double* __restrict__ pr = G.prior + (size_t)run.c0 * T * A;
			for (uint32_t q = tid; q < ncols * T * A; q += threads) pr_lds[q] = pr[q];
			for (uint32_t q = tid; q < ncols; q += threads) rho_lds[q] = G.rho[run.c0 + q];
			const uint4* __restrict__ cg = reinterpret_cast<const uint4*>(G.cols + run.c0);
			for (uint32_t q = tid; q < ncols * 2u; q += threads) reinterpret_cast<uint4*>(col_lds)[q] = cg[q];
		}
		GS_STAMP(0);   // prologue loads issued and staged
		if (from_other) {
			// forward reads the exchange column in its ENTRY layout, backward in its EXIT layout (the same index space: the reads that
			// continue across the boundary)
			const uint32_t occ = DIR == 0 ? run.in_occ : run.out_occ;
			uint32_t idx = 0;
			if (DIR == 0 && run.in_identity) idx = Pcell & occ;
			else {
	#pragma unroll
				for (int s = 0; s < SLOT_MAXSLOTS; ++s) idx |= (((Pcell & occ) >> s) & 1u) << (DIR == 0 ? gs_pos(run.in_pos, s) : gs_pos(run.out_pos, s));
			}
			val = prev[(size_t)idx * T + i];
		}
		// total of the entering column when this run rescales (every thread ends up with the same number): wave sums, then across the waves
		if (np) {
			const uint32_t p0 = DIR == 0 ? run.part_in_f : run.part_in_b;
			for (uint32_t q = tid; q < np; q += threads) psum += G.partials[p0 + q];
			for (int off = 32; off > 0; off >>= 1) psum += __shfl_xor(psum, off);
			if (lane == 0) red[wave] = psum;
		}
	}
	__syncthreads();
	GS_STAMP(1);   // entering value loaded, partial sums reduced, barrier
	double inv = 1.0;
	if (np) {
		double total = 0.0;
		for (uint32_t q = 0; q < nwaves; ++q) total += red[q];
		inv = total > 0.0 ? 1.0 / total : 1.0;
	}
	double* __restrict__ store = (DIR == 0 ? G.fstore : G.bstore) + run.store_off + (size_t)w * threads + tid;
	const size_t col_stride = (size_t)threads << run.g;
	uint32_t xsel = 0;
	auto sum_out = [&](uint32_t slot) {   // both lanes of a pair end up with the pair's sum
		double other;
		if (slot < (uint32_t)NLS) other = gs_lane_xor(val, 1u << (slot + TB));
		else {
			double* xb = xbuf + xsel * threads;
			xb[tid] = val;
			__syncthreads();
			other = xb[tid ^ (64u << (slot - NLS))];
			xsel ^= 1u;
		}
		val += other;
	};
	auto transition = [&](double rho) {   // sum over the other transmission value of val * P(. -> .): one fused multiply-add per bit
#pragma unroll
		for (int s = 0; s < TB; ++s) val = fma(rho, gs_lane_xor(val, 1u << s), val);
	};
	const double2* __restrict__ tabS = reinterpret_cast<const double2*>(tabG + run.s_off) + (size_t)lane * (E / 2u);
	auto load_s = [&](uint32_t ci, double (&sv)[E]) {   // the lane part of W for column ci (global: the same 64 entries for everybody, L2-resident)
		const double2* sp = tabS + (size_t)ci * 32u * E;
#pragma unroll
		for (uint32_t q = 0; q < E; q += 2) { const double2 t2 = sp[q / 2u]; sv[q] = t2.x; sv[q + 1] = t2.y; }
	};
	auto cell_sum = [&](uint32_t ci, const double (&sv)[E]) -> double {   // S_i(x) of this lane's cell in column ci
		const double* ap = a_lds + ((size_t)(wave * ncols + ci) * T + i) * E;
		double W[E];
#pragma unroll
		for (uint32_t q = 0; q < E; q += 2) {
			const double2 av = *reinterpret_cast<const double2*>(ap + q);
			W[q] = av.x * sv[q]; W[q + 1] = av.y * sv[q + 1];
		}
		double prod[A];
		gs_products<P>(W, prod);
		const double* pp = pr_lds + ((size_t)ci * T + i) * A;
		double s = 0.0;
#pragma unroll
		for (uint32_t a = 0; a < A; ++a) s = fma(pp[a], prod[a], s);
		return s;
	};
	if constexpr (TB == 0) {
		// The lane part of W is requested THREE columns ahead, into four register sets used in rotation (the loop is unrolled four times, no set is
		// ever copied): the counter that orders vector-memory operations is shared by loads and stores and retires in order, so the wait for a
		// column's S is also a wait for every store issued before that load -- one column ahead (and a copy of the set at the end of each
		// column, which waited for the load just issued) a column cost a full store + load round trip.
		double r0[E], r1[E], r2[E], r3[E];
		auto col_at = [&](uint32_t k) { return DIR == 0 ? k : ncols - 1u - k; };        // k-th column of the walk
		auto request = [&](uint32_t k, double (&sv)[E]) { load_s(col_at(k < ncols ? k : ncols - 1u), sv); };
		auto column = [&](uint32_t k, const double (&sv)[E], double (&fill)[E]) -> bool {   // false: the walk ends here (backward, first column of the table)
			const uint32_t ci = col_at(k);
			const GsCol& cd = col_lds[ci];
			const uint32_t first = gs_uni(cd.first_of_table);
			request(k + 3u, fill);
			if (DIR == 0) {
				const uint32_t n_end = gs_uni(cd.n_end);
				if (!first) transition(rho_lds[ci]);
				store[(size_t)ci * col_stride] = val;   // sum_j A_{c-1}[back(x)][j] P(j -> i): what the likelihood sums need
				val *= cell_sum(ci, sv);
				for (uint32_t e = 0; e < n_end; ++e) sum_out(gs_uni(cd.end_slot[e]));
				return true;
			}
			const uint32_t n_start = gs_uni(cd.n_start);
			store[(size_t)ci * col_stride] = val;   // B_c[fwd(x)][i]
			if (first) return false;                // (B_{-1} is never needed, :200-289 stops at column 1)
			val *= cell_sum(ci, sv);
			for (uint32_t e = 0; e < n_start; ++e) sum_out(gs_uni(cd.start_slot[e]));
			transition(rho_lds[ci]);
			return true;
		};
		request(0, r0); request(1, r1); request(2, r2);
		for (uint32_t k = 0; k < ncols; k += 4u) {
			if (!column(k, r0, r3)) break;
			if (k + 1u >= ncols || !column(k + 1u, r1, r0)) break;
			if (k + 2u >= ncols || !column(k + 2u, r2, r1)) break;
			if (k + 3u >= ncols || !column(k + 3u, r3, r2)) break;
		}
	} else {
		double s_cur[E], s_next[E];
		if (DIR == 0) {
			load_s(0, s_cur);
			for (uint32_t ci = 0; ci < ncols; ++ci) {
				const GsCol& cd = col_lds[ci];
				const uint32_t n_end = gs_uni(cd.n_end), first = gs_uni(cd.first_of_table);
				load_s(ci + 1 < ncols ? ci + 1 : ci, s_next);
				if (!first) transition(rho_lds[ci]);
				store[(size_t)ci * col_stride] = val;   // sum_j A_{c-1}[back(x)][j] P(j -> i): what the likelihood sums need
				val *= cell_sum(ci, s_cur);
				for (uint32_t e = 0; e < n_end; ++e) sum_out(gs_uni(cd.end_slot[e]));
	#pragma unroll
				for (uint32_t q = 0; q < E; ++q) s_cur[q] = s_next[q];
			}
		} else {
			load_s(ncols - 1u, s_cur);
			for (uint32_t ci = ncols; ci-- > 0;) {
				const GsCol& cd = col_lds[ci];
				const uint32_t n_start = gs_uni(cd.n_start), first = gs_uni(cd.first_of_table);
				load_s(ci ? ci - 1 : 0u, s_next);
				store[(size_t)ci * col_stride] = val;   // B_c[fwd(x)][i]
				if (first) break;                       // (B_{-1} is never needed, :200-289 stops at column 1)
				val *= cell_sum(ci, s_cur);
				for (uint32_t e = 0; e < n_start; ++e) sum_out(gs_uni(cd.start_slot[e]));
				transition(rho_lds[ci]);
	#pragma unroll
				for (uint32_t q = 0; q < E; ++q) s_cur[q] = s_next[q];
			}
		}
	}
	GS_STAMP(2);   // column loop
	// ---- exit: hand on what was received times 1 / (total received), and the per-wave sums of what is handed on
	const bool to_other = DIR == 0 ? run.has_next != 0u : run.has_prev != 0u;
	if (to_other) {
		const uint32_t occ = DIR == 0 ? run.out_occ : run.in_occ;
		const uint32_t localmask = (1u << L) - 1u;
		const bool writes = (lcell & ~occ & localmask) == 0u;   // representatives: free-slot bits zero
		uint32_t idx = 0;
		if (DIR == 1 && run.in_identity) idx = Pcell & occ;
		else {
#pragma unroll
			for (int s = 0; s < SLOT_MAXSLOTS; ++s) idx |= (((Pcell & occ) >> s) & 1u) << (DIR == 0 ? gs_pos(run.out_pos, s) : gs_pos(run.in_pos, s));
		}
		const double outv = val * inv;
		if (writes) cur[(size_t)idx * T + i] = outv;
		if (DIR == 0 ? run.emit_f : run.emit_b) {
			double ps = writes ? outv : 0.0;
			for (int off = 32; off > 0; off >>= 1) ps += __shfl_xor(ps, off);
			if (lane == 0) G.partials[(DIR == 0 ? run.part_out_f : run.part_out_b) + w * nwaves + wave] = ps;
		}
	}
	GS_STAMP(3);   // exit
#ifdef WHAMD_GENO_STAMPS
	if (blockIdx.x == 0 && threadIdx.x == 0 && G.dbg) { atomicAdd(&G.dbg[DIR * 8 + 6], 1ull); atomicAdd(&G.dbg[DIR * 8 + 7], (unsigned long long)ncols); }
#endif
#undef GS_STAMP
}

// ---- combine: blockIdx.y = column, blockIdx.x = 256-thread block of the column's lanes (workgroup-major, as the chains stored them)
struct GsCombineCol {
	unsigned long long tab_off, store_off;   // the column's run
	uint32_t v_off, s_off;
	uint32_t ci, ncols, g, L, threads, n_blocks;
};
constexpr uint32_t GS_COMBINE_LANES = 8;   // lanes of a column one thread of the combine kernel goes through (one reduction for all of them)
// Per lane only U(x, i, a) = forward * backward * prod_p W_i(x)[p][a_p] is formed and summed over the cells x -- the prior and the
// genotype of every individual depend on (i, a) alone, they are applied to the T x A sums of a column by geno_slot_finish.
// A thread's lanes share lane & 63, hence the transmission value i: it accumulates A numbers; the block reduces them over the
// lanes with the same i and leaves u_partials[column][block][i][a].
template <int TB, int P>
__global__ __launch_bounds__(256) void geno_slot_combine(GsDev G, const GsCombineCol* __restrict__ ccols, uint32_t c_first, uint32_t max_blocks,
                                                          double* __restrict__ u_partials) {
	constexpr uint32_t T = 1u << TB, E = 2u * P, A = 1u << P;
	const uint32_t c = c_first + blockIdx.y;
	const GsCombineCol cc = ccols[c];
	if (blockIdx.x >= cc.n_blocks) return;
	__shared__ double red[4][T * A];
	__shared__ __attribute__((aligned(16))) double gv[GS_COMBINE_LANES * 4u][T * E];   // G[workgroup] * V[wave] of the 32 (lane group, wave) pieces of this block
	const GsCol cd = G.cols[c];
	const uint32_t localmask = (1u << cc.L) - 1u;
	const uint32_t n_lanes = cc.threads << cc.g, tshift = 31u - (uint32_t)__clz((int)cc.threads);
	const double* __restrict__ tab = G.tab + cc.tab_off;
	const size_t col_at = cc.store_off + (size_t)cc.ci * n_lanes;
	const uint32_t lane = threadIdx.x & 63u, i = lane & (T - 1u);
	// all the loads of the thread's lanes first: one memory round trip.  UNCONDITIONAL loads (the address clamped into the column, the value
	// dropped afterwards): under `counts ? load : 0` every pair of loads sat in its own branch with a wait behind it -- eight round trips in a row.
	double fv[GS_COMBINE_LANES], bv[GS_COMBINE_LANES], fb[GS_COMBINE_LANES];
#pragma unroll
	for (uint32_t j = 0; j < GS_COMBINE_LANES; ++j) {
		const uint32_t gt = (blockIdx.x * GS_COMBINE_LANES + j) * 256u + threadIdx.x;   // lane of the column: workgroup * threads + tid
		const size_t at = col_at + (gt < n_lanes ? gt : 0u);
		fv[j] = G.fstore[at];
		bv[j] = G.bstore[at];
	}
	double sw[E];   // the lane part of W is the same for all of the thread's lanes
	{
		const double* sp = tab + cc.s_off + ((size_t)cc.ci * 64u + lane) * E;
#pragma unroll
		for (uint32_t q = 0; q < E; ++q) sw[q] = sp[q];
	}
	// the workgroup and wave parts of W: the same T x E numbers for the 64 lanes of a piece -- fetched once per block, coalesced, into LDS (as
	// per-lane loads they were 2 E loads with four distinct addresses per piece and lane: the kernel waited on them, 24 ms for a trio's 42 GB)
	{
		constexpr uint32_t N_GV = GS_COMBINE_LANES * 4u * T * E, PER = (N_GV + 255u) / 256u;
		double gq[PER], vq[PER];
#pragma unroll
		for (uint32_t u = 0; u < PER; ++u) {   // (loads first, unconditional: a piece beyond the column reads the column's first one)
			const uint32_t idx = u * 256u + threadIdx.x, piece = (idx / (T * E)) % (GS_COMBINE_LANES * 4u), q = idx % (T * E);
			const uint32_t gt0 = ((blockIdx.x * GS_COMBINE_LANES + (piece >> 2)) * 4u + (piece & 3u)) * 64u;   // first lane of the piece
			const uint32_t g0 = gt0 < n_lanes ? gt0 : 0u;
			const uint32_t w = g0 >> tshift, wave = (g0 & (cc.threads - 1u)) >> 6;
			gq[u] = tab[((size_t)(w * cc.ncols + cc.ci) * T) * E + q];
			vq[u] = tab[cc.v_off + ((size_t)(wave * cc.ncols + cc.ci) * T) * E + q];
		}
#pragma unroll
		for (uint32_t u = 0; u < PER; ++u) {
			const uint32_t idx = u * 256u + threadIdx.x;
			if (idx < N_GV) gv[idx / (T * E)][idx % (T * E)] = gq[u] * vq[u];
		}
	}
	__syncthreads();
	// (the stored values are first used here, behind the table loads: everything above is one memory round trip)
#pragma unroll
	for (uint32_t j = 0; j < GS_COMBINE_LANES; ++j) {
		const uint32_t gt = (blockIdx.x * GS_COMBINE_LANES + j) * 256u + threadIdx.x;
		const uint32_t lcell = (gt & (cc.threads - 1u)) >> TB;   // (threads = 64 << lw: a power of two)
		const bool counts = gt < n_lanes && (lcell & ~cd.active & localmask) == 0u;   // one lane per DISTINCT cell: free-slot bits zero
		fb[j] = counts ? fv[j] * bv[j] : 0.0;
	}
	double acc[A];
#pragma unroll
	for (uint32_t a = 0; a < A; ++a) acc[a] = 0.0;
#pragma unroll
	for (uint32_t j = 0; j < GS_COMBINE_LANES; ++j) {
		const uint32_t gt = (blockIdx.x * GS_COMBINE_LANES + j) * 256u + threadIdx.x;
		if (gt >= n_lanes) continue;   // (wave-uniform: whole waves lie inside or outside the column)
		const double2* gvp = reinterpret_cast<const double2*>(&gv[j * 4u + (threadIdx.x >> 6)][i * E]);
		double W[E];
#pragma unroll
		for (uint32_t q = 0; q < E; q += 2) { const double2 t2 = gvp[q / 2u]; W[q] = t2.x * sw[q]; W[q + 1] = t2.y * sw[q + 1]; }
		double prod[A];
		gs_products<P>(W, prod);
#pragma unroll
		for (uint32_t a = 0; a < A; ++a) acc[a] = fma(fb[j], prod[a], acc[a]);
	}
	// sum over the lanes with the same transmission value: the lane bits above the TB low ones, then the four waves
#pragma unroll
	for (uint32_t a = 0; a < A; ++a) {
		double v = acc[a];
#pragma unroll
		for (int bit = TB; bit < 6; ++bit) v += gs_lane_xor(v, 1u << bit);
		if (lane < T) red[threadIdx.x >> 6][lane * A + a] = v;
	}
	__syncthreads();
	if (threadIdx.x < T * A)
		u_partials[((size_t)blockIdx.y * max_blocks + blockIdx.x) * (T * A) + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// normalised genotype likelihoods of a batch of columns: block = column.  L_c[individual][genotype] = sum over (i, a) with that
// genotype of prior_c(i, a) * U_c[i][a]  (src/genotypedptable.cpp:376-383, :444-451)
__global__ __launch_bounds__(256) void geno_slot_finish(GsDev G, const double* __restrict__ u_partials, const GsCombineCol* __restrict__ ccols, uint32_t c_first,
                                                        uint32_t max_blocks, double* __restrict__ gl_out) {
	const uint32_t c = c_first + blockIdx.x, TA = G.T * G.A, n_ind = G.n_ind, n_gl = 1u + 3u * n_ind, nb = ccols[c].n_blocks;
	__shared__ double u[16 * GS_MAXA];
	__shared__ double tot[GS_MAXGL];
	__shared__ uint8_t gsh[16 * GS_MAXA * 4];   // gidx [T][A][n_ind <= 4]: read TA times per output below (from global memory: a round trip each)
	for (uint32_t q = threadIdx.x; q < TA * n_ind; q += 256u) gsh[q] = G.gidx[q];
	const double* p = u_partials + (size_t)blockIdx.x * max_blocks * TA;
	if (threadIdx.x < TA) {
		// (eight partial sums requested at a time, added in block order: written as one load per trip the loop was a memory round trip per block --
		// 64 of them in a row for a trio's column)
		double v = 0.0;
		uint32_t blk = 0;
		for (; blk + 8u <= nb; blk += 8u) {
			double t8[8];
#pragma unroll
			for (uint32_t q = 0; q < 8u; ++q) t8[q] = p[(size_t)(blk + q) * TA + threadIdx.x];
#pragma unroll
			for (uint32_t q = 0; q < 8u; ++q) v += t8[q];
		}
		for (; blk < nb; ++blk) v += p[(size_t)blk * TA + threadIdx.x];
		u[threadIdx.x] = v * G.prior[(size_t)c * TA + threadIdx.x];
	}
	__syncthreads();
	if (threadIdx.x < n_gl) {
		double v = 0.0;
		const uint32_t s = threadIdx.x ? (threadIdx.x - 1) / 3 : 0, g = threadIdx.x ? (threadIdx.x - 1) % 3 : 0;
		for (uint32_t q = 0; q < TA; ++q)
			if (threadIdx.x == 0 || gsh[q * n_ind + s] == g) v += u[q];
		tot[threadIdx.x] = v;
	}
	__syncthreads();
	if (threadIdx.x >= 1 && threadIdx.x < n_gl) {
		const uint32_t s = (threadIdx.x - 1) / 3, g = (threadIdx.x - 1) % 3;
		gl_out[((size_t)s * G.n_cols + c) * 3 + g] = tot[threadIdx.x] / tot[0];
	}
}

struct Cleanup {
	std::vector<void*> allocations;
	std::vector<hipStream_t> streams;
	std::vector<hipEvent_t> events;
	int slab_device = -1;
	~Cleanup() {
		if (slab_device >= 0) genotype_slab_release(slab_device);
		for (hipEvent_t e : events) (void)hipEventDestroy(e);
		for (hipStream_t s : streams) (void)hipStreamDestroy(s);
		for (void* a : allocations) (void)hipFree(a);
	}
};

}  // namespace

// Returns WHAMD_OK with `used` = false when the table is not eligible (the caller takes the per-column kernels): a pedigree the
// planner does not cover, a column that fits no run, stores that do not fit in HBM.
whamd_status_t genotype_solve_slots(const Problem& p, const GenotypeModel& m, int device, std::vector<double>& gl_out, GenotypeStats& st,
                                    bool& used, std::string& msg) {
	used = false;
	const uint32_t n = p.n_cols, T = p.T, ni = p.n_ind;
	if (n < 2 || ni == 0 || ni > 4 || !(p.P == 2 || p.P == 4) || !(T == 1 || T == 4 || T == 16) || (T == 1) != (p.P == 2)) return WHAMD_OK;
	int l_pref = 0;
	if (const char* e = debug_env("WHAMD_GENO_SLOT_L")) l_pref = atoi(e);
	SlotPlan plan;
	if (!plan_forward_slots(p, l_pref > 0 ? -l_pref : 0, 0, plan, 0, /*genotype_mode=*/true)) return WHAMD_OK;
	for (const Step& s : plan.steps) if (s.kind != 2) return WHAMD_OK;   // a column no run can take
	const uint32_t tb = T == 1 ? 0u : (T == 4 ? 2u : 4u), E = 2u * p.P, A = m.A;
#define GS_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { msg = std::string(#expr) + " failed: " + hipGetErrorString(e_); return WHAMD_ERR_DEVICE; } } while (0)
	GS_TRY(hipSetDevice(device));
	const size_t n_runs = plan.runs.size();
	// ---- host descriptors
	std::vector<GsCol> cols(n);
	std::vector<GsRow> rows(n);
	std::vector<GsRun> runs(n_runs);
	std::vector<GsCombineCol> ccols(n);
	unsigned long long tab_words = 0, store_words = 0;
	uint32_t n_partials = 0, max_f = 0, max_blocks = 1;
	size_t max_lds = 0;
	for (size_t ri = 0; ri < n_runs; ++ri) {
		const SlotRun& sr = plan.runs[ri];
		GsRun& r = runs[ri];
		r.c0 = sr.c0; r.ncols = sr.ncols; r.g = sr.g; r.L = sr.L; r.lw = sr.lw; r.threads = sr.threads;
		r.has_prev = sr.c0 > 0 ? 1u : 0u;
		r.has_next = sr.c0 + sr.ncols < n ? 1u : 0u;
		r.in_occ = sr.in_occ; r.in_identity = sr.in_identity; r.out_occ = sr.out_occ;
		std::memcpy(r.in_pos, sr.in_pos, sizeof r.in_pos);
		std::memcpy(r.out_pos, sr.out_pos, sizeof r.out_pos);
		const unsigned long long per_unit = (unsigned long long)sr.ncols * T * E;
		r.tab_off = tab_words;
		r.v_off = (uint32_t)(per_unit << sr.g);
		r.s_off = r.v_off + (uint32_t)(per_unit << sr.lw);
		tab_words += (unsigned long long)r.s_off + (unsigned long long)sr.ncols * 64u * E;
		r.store_off = store_words;
		store_words += (unsigned long long)sr.ncols * ((unsigned long long)sr.threads << sr.g);
		max_f = std::max(max_f, sr.L + sr.g);
		const uint32_t nw = (sr.threads >> 6) << sr.g;   // per-wave partial sums of what the run hands on, one set per direction
		r.part_out_f = n_partials; n_partials += nw;
		r.part_out_b = n_partials; n_partials += nw;
		const uint32_t blocks = (uint32_t)((((size_t)sr.threads << sr.g) + 256u * GS_COMBINE_LANES - 1u) / (256u * GS_COMBINE_LANES));
		max_blocks = std::max(max_blocks, blocks);
		const size_t waves = sr.threads >> 6;
		max_lds = std::max(max_lds, ((size_t)2 * sr.threads + waves * sr.ncols * T * E + (size_t)sr.ncols * T * A + ((sr.ncols + 1) & ~1u) + 16) * 8 + (size_t)sr.ncols * sizeof(GsCol));
		for (uint32_t ci = 0; ci < sr.ncols; ++ci) {
			const uint32_t c = sr.c0 + ci;
			const PedSlotRow& pr = plan.prows[c];
			const SlotBtCol& bc = plan.bt_cols[c];
			GsCol& cd = cols[c];
			GsRow& rw = rows[c];
			std::memset(&cd, 0, sizeof cd);
			std::memset(&rw, 0, sizeof rw);
			const ColumnEntry* col = p.col_begin(c);
			for (uint32_t j = 0; j < p.k[c]; ++j) {
				const uint32_t s = bc.slot[j];
				cd.active |= 1u << s;
				rw.pe[s] = m.error_prob[p.col_ptr[c] + j];
				rw.ind[s] = col[j].sample;
				rw.allele[s] = col[j].allele;
			}
			cd.first_of_table = c == 0;
			cd.last_of_table = c + 1 == n;
			if (pr.n_end > (uint32_t)GS_MAXLOCAL || pr.pad[0] > (uint32_t)GS_MAXLOCAL) return WHAMD_OK;
			cd.n_end = (uint8_t)pr.n_end;
			for (uint32_t e = 0; e < pr.n_end; ++e) cd.end_slot[e] = plan.end_slots[plan.end_off[ri] + bc.kf + e];
			cd.n_start = (uint8_t)pr.pad[0];
			for (uint32_t e = 0; e < pr.pad[0]; ++e) cd.start_slot[e] = plan.start_slots[plan.start_off[ri] + pr.pad[1] + e];
			GsCombineCol& cc = ccols[c];
			cc.tab_off = r.tab_off; cc.store_off = r.store_off; cc.v_off = r.v_off; cc.s_off = r.s_off;
			cc.ci = ci; cc.ncols = sr.ncols; cc.g = sr.g; cc.L = sr.L; cc.threads = sr.threads; cc.n_blocks = blocks;
		}
	}
	for (size_t ri = 0; ri < n_runs; ++ri) {   // the partial sums a run reads are the ones its neighbour writes; every GS_RESCALE-th run of a chain rescales
		GsRun& r = runs[ri];
		if (ri > 0 && ri % GS_RESCALE == 0) {
			r.part_in_f = runs[ri - 1].part_out_f; r.n_part_in_f = (plan.runs[ri - 1].threads >> 6) << plan.runs[ri - 1].g;
			runs[ri - 1].emit_f = 1;
		}
		if (ri + 1 < n_runs && (n_runs - 1 - ri) % GS_RESCALE == 0) {
			r.part_in_b = runs[ri + 1].part_out_b; r.n_part_in_b = (plan.runs[ri + 1].threads >> 6) << plan.runs[ri + 1].g;
			runs[ri + 1].emit_b = 1;
		}
	}
	size_t free_b = 0, total_b = 0;
	GS_TRY(hipMemGetInfo(&free_b, &total_b));
	if (free_b < total_b / 2) {   // a phasing table of this process may have left its arena in the cache (dp_device.hip)
		dptable_release_arena_cache();
		GS_TRY(hipMemGetInfo(&free_b, &total_b));
	}
	free_b += genotype_slab_idle_bytes(device);   // the column store kept from an earlier call is available to this one
	constexpr uint32_t BATCH = 512;
	const double fixed = (double)tab_words * 8 + (double)BATCH * max_blocks * T * A * 8 + 4.0 * ((double)(1ull << max_f) * T * 8) +
	                     (double)n * (sizeof(GsCol) + sizeof(GsRow) + sizeof(GsCombineCol) + 8.0 * T * A + 8);
	if (max_lds > 150 * 1024) return WHAMD_OK;
	// ---- windows.  The column stores are what grows with the table (a trio at coverage 15: 2 MiB per column and chain).  When both do not fit,
	// the runs are cut into WINDOWS (the reference keeps sqrt(n) columns and recomputes, src/genotypedptable.cpp:116-157,159-195,324): pass 1
	// runs the whole forward chain keeping only the exchange column at every window boundary (and the columns of the newest window); then,
	// newest window first, the forward columns of a window are recomputed from its kept exchange column, the backward chain runs through the
	// window, and the window's likelihoods are formed.  Two sets of window stores: the recomputation of window w - 1 runs beside the backward
	// chain and the combine of window w.  One more forward pass, any table length.
	struct GsWindow { size_t r0, r1; unsigned long long words; uint32_t c0, c1; };
	std::vector<GsWindow> windows;
	{
		const double room = 0.8 * (double)free_b - fixed - (double)(2ull << 30);
		unsigned long long budget_words = ~0ull;   // per store
		if (2.0 * (double)store_words * 8 > room) budget_words = room > 0 ? (unsigned long long)(room / 4.0 / 8.0) : 0ull;
		if (const char* e = getenv("WHAMD_GENO_WINDOW_BYTES")) budget_words = std::min<unsigned long long>(budget_words, std::strtoull(e, nullptr, 10) / 8);
		GsWindow cur{0, 0, 0, 0, 0};
		for (size_t ri = 0; ri < n_runs; ++ri) {
			const unsigned long long words = (unsigned long long)plan.runs[ri].ncols * ((unsigned long long)plan.runs[ri].threads << plan.runs[ri].g);
			if (words > budget_words) return WHAMD_OK;   // a single run beyond the budget: the per-column path
			if (cur.words + words > budget_words) {
				cur.r1 = ri;
				windows.push_back(cur);
				cur = GsWindow{ri, ri, 0, 0, 0};
			}
			runs[ri].store_off = cur.words;   // (relative to the window's stores)
			cur.words += words;
		}
		cur.r1 = n_runs;
		windows.push_back(cur);
		for (GsWindow& wdw : windows) {
			wdw.c0 = plan.runs[wdw.r0].c0;
			wdw.c1 = plan.runs[wdw.r1 - 1].c0 + plan.runs[wdw.r1 - 1].ncols;
			for (uint32_t c = wdw.c0; c < wdw.c1; ++c) ccols[c].store_off = runs[0].store_off;   // (set per column below)
		}
		for (size_t ri = 0; ri < n_runs; ++ri)
			for (uint32_t ci = 0; ci < plan.runs[ri].ncols; ++ci) ccols[plan.runs[ri].c0 + ci].store_off = runs[ri].store_off;
	}
	const size_t n_windows = windows.size();
	unsigned long long window_words = 0;
	for (const GsWindow& wdw : windows) window_words = std::max(window_words, wdw.words);
	const size_t n_sets = n_windows > 1 ? 2 : 1;   // (fstore, bstore) pairs
	used = true;
	gl_out.assign((size_t)ni * n * 3, 0.0);
	st = GenotypeStats();
	st.n_columns = n;
	st.transmissions = T;
	st.window = n_windows > 1 ? windows[0].c1 - windows[0].c0 : n;
	for (uint32_t c = 0; c < n; ++c) { st.n_cells += 1ull << p.k[c]; st.max_coverage = std::max<uint32_t>(st.max_coverage, p.k[c]); }
	Cleanup keep;
	hipStream_t sf = nullptr, sb = nullptr, sc = nullptr;
	GS_TRY(hipStreamCreateWithFlags(&sf, hipStreamNonBlocking)); keep.streams.push_back(sf);
	GS_TRY(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); keep.streams.push_back(sb);
	if (n_windows > 1 || debug_env("WHAMD_GENO_PIECES")) { GS_TRY(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); keep.streams.push_back(sc); }   // (likelihood sums beside the chains)
	auto alloc = [&](void** dptr, size_t bytes) -> hipError_t {
		hipError_t e = hipMalloc(dptr, std::max<size_t>(bytes, 16));
		if (e != hipSuccess) {   // idle blocks of the phasing / heuristic pool may hold the memory: give them back, try once more
			(void)hipGetLastError();
			devpool_release();
			e = hipMalloc(dptr, std::max<size_t>(bytes, 16));
		}
		if (e == hipSuccess) keep.allocations.push_back(*dptr);
		return e;
	};
	auto up = [&](void** dptr, const void* src, size_t bytes) -> hipError_t {
		hipError_t e = alloc(dptr, bytes);
		if (e == hipSuccess && bytes) e = hipMemcpyAsync(*dptr, src, bytes, hipMemcpyHostToDevice, sf);
		return e;
	};
	std::vector<double> rho(n, 0.0);
	const uint32_t nb = 2 * p.n_triples + 1;
	if (T > 1) for (uint32_t c = 0; c < n; ++c) rho[c] = m.transition_bern[(size_t)c * nb + 1] / m.transition_bern[(size_t)c * nb];
	GsDev G{};
	void *d_cols, *d_rows, *d_prior, *d_rho, *d_gidx, *d_h2p, *d_runs, *d_ccols, *d_tab, *d_fs, *d_bs, *d_part, *d_glpart, *d_gl;
	double* d_x[4];
	GS_TRY(up(&d_cols, cols.data(), cols.size() * sizeof(GsCol)));
	GS_TRY(up(&d_rows, rows.data(), rows.size() * sizeof(GsRow)));
	GS_TRY(up(&d_prior, m.allele_prior.data(), m.allele_prior.size() * 8));
	GS_TRY(up(&d_rho, rho.data(), rho.size() * 8));
	GS_TRY(up(&d_gidx, m.genotype_index.data(), m.genotype_index.size()));
	GS_TRY(up(&d_h2p, p.h2p.data(), p.h2p.size()));
	GS_TRY(up(&d_runs, runs.data(), runs.size() * sizeof(GsRun)));
	GS_TRY(up(&d_ccols, ccols.data(), ccols.size() * sizeof(GsCombineCol)));
	GS_TRY(alloc(&d_tab, (size_t)tab_words * 8));
	d_fs = genotype_slab_acquire(device, 2 * n_sets * (size_t)window_words * 8);   // the column stores in the block kept between calls
	if (d_fs) keep.slab_device = device;
	else GS_TRY(alloc(&d_fs, 2 * n_sets * (size_t)window_words * 8));
	d_bs = (double*)d_fs + n_sets * window_words;
	void* d_check = nullptr;   // the forward exchange column entering every window but the first
	const size_t check_bytes = ((size_t)1 << max_f) * T * 8;
	if (n_windows > 1) GS_TRY(alloc(&d_check, (n_windows - 1) * check_bytes));
	GS_TRY(alloc(&d_part, (size_t)n_partials * 8));
	GS_TRY(alloc(&d_glpart, (size_t)BATCH * max_blocks * T * A * 8));
	GS_TRY(alloc(&d_gl, gl_out.size() * 8));
	for (double*& x : d_x) GS_TRY(alloc((void**)&x, ((size_t)1 << max_f) * T * 8));
	G.cols = (const GsCol*)d_cols; G.rows = (const GsRow*)d_rows; G.prior = (const double*)d_prior; G.rho = (const double*)d_rho;
	G.gidx = (const uint8_t*)d_gidx; G.h2p = (const int8_t*)d_h2p; G.tab = (double*)d_tab; G.fstore = (double*)d_fs; G.bstore = (double*)d_bs;
	G.dbg = nullptr;
#ifdef WHAMD_GENO_STAMPS
	{ void* d_dbg = nullptr; GS_TRY(alloc(&d_dbg, 128)); GS_TRY(hipMemset(d_dbg, 0, 128)); G.dbg = (unsigned long long*)d_dbg; }
#endif
	G.partials = (double*)d_part; G.T = T; G.A = A; G.P = p.P; G.n_ind = ni; G.n_cols = n;
	hipEvent_t ev[4];
	for (hipEvent_t& e : ev) { GS_TRY(hipEventCreate(&e)); keep.events.push_back(e); }
	using RunFn = void (*)(GsDev, GsRun, const double*, double*);
	RunFn fwd = nullptr, bwd = nullptr;
	if (tb == 0) { fwd = geno_slot_run<0, 2, 0>; bwd = geno_slot_run<0, 2, 1>; }
	else if (tb == 2) { fwd = geno_slot_run<2, 4, 0>; bwd = geno_slot_run<2, 4, 1>; }
	else { fwd = geno_slot_run<4, 4, 0>; bwd = geno_slot_run<4, 4, 1>; }
	GS_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fwd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	GS_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(bwd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	uint64_t launches = 0;
	GS_TRY(hipEventRecord(ev[0], sf));
	{
		uint32_t most = 0;
		for (const GsRun& r : runs) most = std::max<uint32_t>(most, ((r.ncols * T) << r.g) + ((r.ncols * T) << r.lw) + r.ncols * 64u);
		const uint32_t bx = std::max(1u, std::min(256u, (most + 255u) / 256u));
		for (size_t r0 = 0; r0 < n_runs; r0 += 32768) {
			const uint32_t ny = (uint32_t)std::min<size_t>(32768, n_runs - r0);
			hipLaunchKernelGGL(geno_slot_tables, dim3(bx, ny), dim3(256), 0, sf, G, (const GsRun*)d_runs + r0);
			++launches;
		}
		GS_TRY(hipGetLastError());
	}
	GS_TRY(hipEventRecord(ev[1], sf));
	GS_TRY(hipStreamWaitEvent(sb, ev[1], 0));   // uploads and tables are complete
	auto lds_of = [&](const GsRun& r) {
		const size_t waves = r.threads >> 6;
		return ((size_t)2 * r.threads + waves * r.ncols * T * E + (size_t)r.ncols * T * A + ((r.ncols + 1) & ~1u) + 16) * 8 + (size_t)r.ncols * sizeof(GsCol);
	};
	auto with_stores = [&](size_t set) {
		GsDev g = G;
		g.fstore = (double*)d_fs + set * window_words;
		g.bstore = (double*)d_bs + set * window_words;
		return g;
	};
	auto launch_fwd = [&](size_t ri, const GsDev& g) {
		const GsRun& r = runs[ri];
		hipLaunchKernelGGL(fwd, dim3(1u << r.g), dim3(r.threads), lds_of(r), sf, g, r, (const double*)d_x[ri & 1], d_x[(ri & 1) ^ 1]);
		++launches;
	};
	auto launch_bwd = [&](size_t ri, const GsDev& g) {
		const GsRun& r = runs[ri];
		hipLaunchKernelGGL(bwd, dim3(1u << r.g), dim3(r.threads), lds_of(r), sb, g, r, (const double*)d_x[2 + (ri & 1)], d_x[2 + ((ri & 1) ^ 1)]);
		++launches;
	};
	auto launch_combine = [&](const GsWindow& wdw, const GsDev& g, hipStream_t stream) {
		for (uint32_t c0 = wdw.c0; c0 < wdw.c1; c0 += BATCH) {
			const uint32_t ncol = std::min(BATCH, wdw.c1 - c0);
			uint32_t gx = 1;
			for (uint32_t c = c0; c < c0 + ncol; ++c) gx = std::max(gx, ccols[c].n_blocks);
			if (tb == 0) hipLaunchKernelGGL((geno_slot_combine<0, 2>), dim3(gx, ncol), dim3(256), 0, stream, g, (const GsCombineCol*)d_ccols, c0, max_blocks, (double*)d_glpart);
			else if (tb == 2) hipLaunchKernelGGL((geno_slot_combine<2, 4>), dim3(gx, ncol), dim3(256), 0, stream, g, (const GsCombineCol*)d_ccols, c0, max_blocks, (double*)d_glpart);
			else hipLaunchKernelGGL((geno_slot_combine<4, 4>), dim3(gx, ncol), dim3(256), 0, stream, g, (const GsCombineCol*)d_ccols, c0, max_blocks, (double*)d_glpart);
			hipLaunchKernelGGL(geno_slot_finish, dim3(ncol), dim3(256), 0, stream, g, (const double*)d_glpart, (const GsCombineCol*)d_ccols, c0, max_blocks, (double*)d_gl);
			launches += 2;
		}
	};
	if (n_windows == 1) {
		// the two chains, submissions interleaved so that neither hardware queue runs dry.  (WHAMD_GENO_PIECES = k forms the likelihood sums of
		// a k-th of the table on a third stream as soon as both chains have passed it, beside the rest of the chains.  Measured: no gain --
		// the combine streams the stores at 3 TB/s and the chains slow down by what it saves: trio of 20 000 columns, chains 37 -> 54 ms,
		// combine 24 -> 6 ms.  One piece after the chains is the default.)
		const GsDev g = with_stores(0);
		size_t n_pieces = 1;
		const bool third_stream = debug_env("WHAMD_GENO_PIECES") != nullptr;
		if (const char* e = debug_env("WHAMD_GENO_PIECES")) n_pieces = std::max<size_t>(1, std::min<size_t>((size_t)atoi(e), std::max<size_t>(1, n_runs / 64)));
		auto piece_lo = [&](size_t k) { return n_runs * k / n_pieces; };
		std::vector<hipEvent_t> pf(n_pieces), pb(n_pieces);
		for (size_t k = 0; k < n_pieces; ++k)
			for (hipEvent_t* e : {&pf[k], &pb[k]}) { GS_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming)); keep.events.push_back(*e); }
		size_t rf = 0, rb = n_runs, kf = 0, kb = n_pieces;
		while (rf < n_runs || rb > 0) {
			if (rf < n_runs) {
				launch_fwd(rf++, g);
				if (rf == piece_lo(kf + 1)) { GS_TRY(hipEventRecord(pf[kf], sf)); ++kf; }
			}
			if (rb > 0) {
				launch_bwd(--rb, g);
				if (rb == piece_lo(kb - 1)) { --kb; GS_TRY(hipEventRecord(pb[kb], sb)); }
			}
		}
		GS_TRY(hipGetLastError());
		GS_TRY(hipEventRecord(ev[2], sb));
		GS_TRY(hipStreamWaitEvent(sf, ev[2], 0));
		GS_TRY(hipEventRecord(ev[2], sf));
		// pieces in the order both chains have passed them: from the middle outwards
		std::vector<size_t> order;
		for (size_t d = 0; order.size() < n_pieces; ++d) {
			const size_t mid = n_pieces / 2;
			if (d == 0) { order.push_back(mid); continue; }
			if (mid >= d) order.push_back(mid - d);
			if (mid + d < n_pieces) order.push_back(mid + d);
		}
		// (one piece: on the forward chain's stream, which has just waited for the backward chain)
		const hipStream_t cs = third_stream ? sc : sf;
		for (size_t k : order) {
			if (third_stream) {
				GS_TRY(hipStreamWaitEvent(sc, pf[k], 0));
				GS_TRY(hipStreamWaitEvent(sc, pb[k], 0));
			}
			const size_t r0 = piece_lo(k), r1 = piece_lo(k + 1);
			GsWindow piece{r0, r1, 0, plan.runs[r0].c0, plan.runs[r1 - 1].c0 + plan.runs[r1 - 1].ncols};
			launch_combine(piece, g, cs);
		}
		if (third_stream) {
			hipEvent_t done = nullptr;
			GS_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming)); keep.events.push_back(done);
			GS_TRY(hipEventRecord(done, sc));
			GS_TRY(hipStreamWaitEvent(sf, done, 0));
		}
	} else {
		// pass 1: the whole forward chain; the exchange column entering every window is kept.  Every window writes its columns into its set and
		// only the newest window's survive: the store of the run kernel is unconditional (under a `keep` branch the counter wait at the join
		// became a wait for the store itself, every column: 42 -> 48 ms for the chains of 50 000 columns)
		const size_t last = n_windows - 1;
		std::vector<hipEvent_t> ef(n_windows), eb(n_windows), ec(n_windows);
		for (size_t w = 0; w < n_windows; ++w)
			for (hipEvent_t* e : {&ef[w], &eb[w], &ec[w]}) { GS_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming)); keep.events.push_back(*e); }
		for (size_t w = 0; w < n_windows; ++w) {
			const GsWindow& wdw = windows[w];
			if (w > 0) GS_TRY(hipMemcpyAsync((char*)d_check + (w - 1) * check_bytes, d_x[wdw.r0 & 1], check_bytes, hipMemcpyDeviceToDevice, sf));
			const GsDev g = with_stores(w % 2);
			for (size_t ri = wdw.r0; ri < wdw.r1; ++ri) launch_fwd(ri, g);
		}
		GS_TRY(hipGetLastError());
		GS_TRY(hipEventRecord(ef[last], sf));
		// newest window first: (recompute the forward columns,) backward chain, likelihoods -- forward of window w - 1 beside backward / combine of w
		for (size_t w = n_windows; w-- > 0;) {
			const GsWindow& wdw = windows[w];
			const GsDev g = with_stores(w % 2);
			if (w != last) {
				if (w + 2 < n_windows) GS_TRY(hipStreamWaitEvent(sf, ec[w + 2], 0));   // the stores of this set are free again
				if (w > 0) GS_TRY(hipMemcpyAsync(d_x[wdw.r0 & 1], (char*)d_check + (w - 1) * check_bytes, check_bytes, hipMemcpyDeviceToDevice, sf));
				for (size_t ri = wdw.r0; ri < wdw.r1; ++ri) launch_fwd(ri, g);
				GS_TRY(hipEventRecord(ef[w], sf));
			}
			GS_TRY(hipStreamWaitEvent(sb, ef[w], 0));
			if (w + 2 < n_windows) GS_TRY(hipStreamWaitEvent(sb, ec[w + 2], 0));
			for (size_t ri = wdw.r1; ri-- > wdw.r0;) launch_bwd(ri, g);
			GS_TRY(hipEventRecord(eb[w], sb));
			GS_TRY(hipStreamWaitEvent(sc, eb[w], 0));
			launch_combine(wdw, g, sc);
			GS_TRY(hipEventRecord(ec[w], sc));
			GS_TRY(hipGetLastError());
		}
		GS_TRY(hipEventRecord(ev[2], sb));
		GS_TRY(hipStreamWaitEvent(sf, ec[0], 0));
		GS_TRY(hipStreamWaitEvent(sf, ec[n_windows > 1 ? 1 : 0], 0));
	}
	GS_TRY(hipGetLastError());
	GS_TRY(hipEventRecord(ev[3], sf));
	GS_TRY(hipMemcpyAsync(gl_out.data(), d_gl, gl_out.size() * 8, hipMemcpyDeviceToHost, sf));
	GS_TRY(hipStreamSynchronize(sf));
	GS_TRY(hipStreamSynchronize(sb));
	if (sc) GS_TRY(hipStreamSynchronize(sc));
#ifdef WHAMD_GENO_STAMPS
	{
		unsigned long long d[16];
		GS_TRY(hipMemcpy(d, G.dbg, sizeof d, hipMemcpyDeviceToHost));
		for (int dir = 0; dir < 2; ++dir) {
			const double runs = (double)std::max<unsigned long long>(d[dir * 8 + 6], 1);
			fprintf(stderr, "[whamd geno stamps] %s: %llu runs, %.1f columns each; cycles since kernel start (wave 0 / workgroup 0): staged %.0f, entered + reduced %.0f, loop done %.0f, exit %.0f\n",
			        dir ? "backward" : "forward", d[dir * 8 + 6], d[dir * 8 + 7] / runs, d[dir * 8 + 0] / runs, d[dir * 8 + 1] / runs, d[dir * 8 + 2] / runs, d[dir * 8 + 3] / runs);
		}
	}
#endif
	float t_tab = 0, t_chain = 0, t_all = 0;
	GS_TRY(hipEventElapsedTime(&t_tab, ev[0], ev[1]));
	GS_TRY(hipEventElapsedTime(&t_chain, ev[1], ev[2]));
	GS_TRY(hipEventElapsedTime(&t_all, ev[0], ev[3]));
	st.backward_ms = t_chain;              // the two chains side by side
	st.forward_ms = t_all - t_chain;       // tables + combine
	st.total_ms = t_all;
	st.launches = launches;
	st.slot_runs = (uint32_t)n_runs;
	if (getenv("WHAMD_DEBUG_TIMING"))
		fprintf(stderr, "[whamd timing] genotype slot runs: %zu runs (%.1f columns per run), %zu window(s), tables %.2f ms, chains %.2f ms, combine %.2f ms; tables %.1f MB, stores %zu x %.1f MB\n",
		        n_runs, (double)n / n_runs, n_windows, t_tab, t_chain, t_all - t_chain - t_tab, tab_words * 8e-6, 2 * n_sets, window_words * 8e-6);
#undef GS_TRY
	return WHAMD_OK;
}

}  // namespace whamd
