// kernels_backtrace.h -- backtrace_kernel: follows the stored argmins from a job's last column to its first (src/pedigreedptable.cpp:137-173).
// Included by dp_device.hip inside namespace whamd { namespace { ... } }: not a stand-alone header.
// Backtrace (src/pedigreedptable.cpp:137-173); out: index / transmission per column, out_score[0] = optimum.
// The steps of the forward plan are walked in reverse (`units`, newest first).  For a resident run the argmin bits the
// path can touch all belong to ONE workgroup's record (the grid-read bits of the path do not change inside a run), so
// the workgroup copies that record (a few KiB) into LDS with one coalesced load while it prefetches the NEXT run's
// column records and the header of the run after that; one wave then follows the path with LDS latency instead of one
// dependent HBM access per column.
//
// One workgroup per job (BtJob).  `with_last_column` != 0: units[0] is the table's last column (its optimum comes from
// P.last_keys), the walk starts at units[1].  == 0: the units are the steps of ONE connected component that ends before
// the table does (its last column projects onto a single entry): the walk starts at units[0] with entry 0 and no score
// is written.
constexpr int BT_CELLS = 128;
constexpr int BT_CHUNK_BLOB = RES_MAXCOLS * 32;   // words of the chunk walker's per-unit descriptor area (>= SLOT_MAXCOLS * 8 + 32)
constexpr int BT_CHUNK_RUNS = 16;   // slot runs per chunk of the speculative backtrace  // >= RES_MAXCOLS and >= SLOT_MAXENDS_RUN + 1

// One pedigree slot run (slots.h, kernels_pedslots.h) for the calling wave: the record holds one byte per (lane, column) =
// argj | ending-read decisions << 4, four columns per word.  Walks the columns newest first from local cell `l` and
// transmission value `tcur` (src/pedigreedptable.cpp:151-172: the ending reads of the column are undone in reverse order at
// the side-0 lane of each pair, then the lane of the complete cell holds the transmission value of the column before).
// cells[c] / cells[64 + c]: local cell and transmission value of the path at column c; returns the value handed down.
__device__ __forceinline__ uint32_t pedslot_walk(const SlotBtCol* bcols, const uint8_t* stage8, uint32_t ncols, uint32_t threads, uint32_t tb,
                                                uint32_t l, uint32_t tcur, uint32_t* cells, bool writer) {
	for (uint32_t ci = ncols; ci-- > 0;) {
		const SlotBtCol& bc = bcols[ci];
		const uint32_t base = (ci >> 2) * threads * 4u + (ci & 3u);
		for (uint32_t e = bc.pad[0]; e-- > 0;) {
			const uint32_t slot = e < 3u ? bc.slot[25u + e] : bc.pad[1];
			const uint32_t look = l & ~(1u << slot);
			const uint32_t byte = stage8[base + ((look << tb) | tcur) * 4u];
			l = look | (((byte >> (4u + e)) & 1u) << slot);
		}
		if (writer) { cells[ci] = l; cells[64u + ci] = tcur; }
		tcur = stage8[base + ((l << tb) | tcur) * 4u] & 15u;
	}
	return tcur;
}

__global__ __launch_bounds__(1024) void backtrace_kernel(DevProblem P, const BtUnit* __restrict__ all_units, const BtJob* __restrict__ jobs,
                                                         uint32_t* __restrict__ path_index, uint32_t* __restrict__ path_trans,
                                                         uint32_t* __restrict__ out_score) {
	const BtJob job = jobs[blockIdx.x];  // one workgroup per job: jobs are independent (DeviceTable jobs)
	const BtUnit* __restrict__ units = all_units + job.unit_off;
	const uint32_t n_units = job.unit_count, with_last_column = job.with_last_column;
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
	uint32_t* recs0 = smem;                                   // 2 x RES_MAXCOLS * 32 words: column records (double buffer)
	uint32_t* hdr = smem + 2 * RES_MAXCOLS * 32;              // 4 x 32 words: unit headers (ring)
	uint32_t* xshare = hdr + 128;                             // 4 words
	uint32_t* cells = xshare + 4;                             // BT_CELLS words: local cell index of the path per column (slot runs: per chain position)
	uint32_t* tsarr = cells + BT_CELLS;                       // RES_MAXCOLS words: transmission value of the path per column
	unsigned long long* stage = reinterpret_cast<unsigned long long*>(tsarr + RES_MAXCOLS);
	const uint32_t lane = threadIdx.x, NT = blockDim.x;
	const uint32_t n = P.n_cols, T = P.T;
	const uint32_t u_first = with_last_column == 1u ? 1u : 0u;
	uint32_t x = 0, tprev = 0;
	if (with_last_column == 2u) {   // windowed solve: continue where the walk of the next newer window stopped
		x = P.bt_state[0];
		tprev = P.bt_state[1];
	}
	if (with_last_column == 1u) {
		// optimum of the last column: first (rank(x), i) attaining the minimum (strict '<' scan, :306-315)
		unsigned long long bestk = ~0ull;
		uint32_t t = 0;
		for (uint32_t i = 0; i < T; ++i) {
			const unsigned long long key = P.last_keys[i];
			if ((key >> KEY_JBITS) < (bestk >> KEY_JBITS)) { bestk = key; t = i; }
		}
		if (bestk == ~0ull) {  // unreachable for valid inputs (the host rejects Mendelian conflicts); keep defined output
			if (lane == 0) out_score[0] = 0xFFFFFFFFu;
			bestk = 0;
		} else if (lane == 0) {
			out_score[0] = (uint32_t)(bestk >> 32);
		}
		const uint32_t rlast = (uint32_t)(bestk >> KEY_JBITS) & BT_STATE_XMASK;
		x = rlast ^ (rlast >> 1);
		tprev = (uint32_t)bestk & KEY_JMASK;
		if (lane == 0) {
			path_index[n - 1] = x;
			path_trans[n - 1] = t;
		}
	}
	// every unit from u_first on yields x_c from x_{c+1}.
	// prime the pipeline: headers of the first two units, records of the first
	if (lane < 64) {
		const uint32_t u = u_first + (lane >> 5);
		if (u < n_units) hdr[(u & 3u) * 32 + (lane & 31u)] = reinterpret_cast<const uint32_t*>(units + u)[lane & 31u];
	}
	__syncthreads();
	if (n_units > u_first && hdr[(u_first & 3u) * 32] == 1u) {
		const uint32_t* h1 = hdr + (u_first & 3u) * 32;
		const uint32_t* __restrict__ g1 = reinterpret_cast<const uint32_t*>(P.res_bt + h1[3]);
		for (uint32_t i = lane; i < h1[2] * 32; i += NT) recs0[(u_first & 1u) * RES_MAXCOLS * 32 + i] = g1[i];
	} else if (n_units > u_first && (hdr[(u_first & 3u) * 32] == 2u || hdr[(u_first & 3u) * 32] == 3u)) {
		const uint32_t* h1 = hdr + (u_first & 3u) * 32;
		const uint32_t* __restrict__ g1 = P.slot_blob + h1[3];
		for (uint32_t i = lane; i < h1[11]; i += NT) recs0[(u_first & 1u) * RES_MAXCOLS * 32 + i] = g1[i];
	}
	__syncthreads();
	unsigned long long bt_load = 0, bt_walk = 0, bt_runs = 0, bt_a = 0, bt_b = 0, bt_c = 0;
	for (uint32_t ui = u_first; ui < n_units; ++ui) {
		const unsigned long long tb0 = P.dbg ? __builtin_readcyclecounter() : 0ull;
		const uint32_t* h = hdr + (ui & 3u) * 32;
		const uint32_t kind = h[0], c0 = h[1], ncols = h[2];
		uint32_t* recs = recs0 + (ui & 1u) * RES_MAXCOLS * 32;
		// prefetch: header of unit ui + 2, records of unit ui + 1 (its header arrived one iteration ago)
		uint32_t hv = 0, wrun = 0;
		const bool hload = lane < 32 && ui + 2 < n_units;
		if (hload) hv = reinterpret_cast<const uint32_t*>(units + ui + 2)[lane];
		const uint32_t* hn = hdr + ((ui + 1) & 3u) * 32;
		const bool next_run = ui + 1 < n_units && hn[0] == 1u, next_slots = ui + 1 < n_units && (hn[0] == 2u || hn[0] == 3u);
		const uint32_t nrec = next_run ? hn[2] * 32 : (next_slots ? hn[11] : 0u);
		const uint32_t* __restrict__ gnext = next_slots ? P.slot_blob + hn[3] : reinterpret_cast<const uint32_t*>(P.res_bt + (next_run ? hn[3] : 0u));
		uint32_t slot_w = 0, slot_l = 0;
		bool slot_mirrored = false;
		uint32_t rv[2];
#pragma unroll
		for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + lane; rv[u] = i < nrec ? gnext[i] : 0u; }
		if (kind == 0) {
			// ---- one column through the column kernels' records (global loads; rare in steady state)
			const uint32_t c = c0;
			// header words of a column unit: 4 f, 5 mode, 6 nplanes, 7 ebits, 8/9 record offset, 10 nseg_fwd, 11 nseg_end,
			// 12..27 deposit runs (if word 28 is set; else they are read from the column descriptor)
			const uint32_t cf = h[4], cmode = h[5], cnplanes = h[6], cebits = h[7], nsf = h[10], nse = h[11];
			const unsigned long long cbt = ((unsigned long long)h[9] << 32) | h[8];
			const uint32_t* segs = h[28] ? (h + 12) : (P.segs + P.cols[c].seg_off);
			const uint32_t y = x & ((1u << cf) - 1u);
			uint32_t xp, aj;
			if (cmode == 0) {
				const unsigned long long* planes = reinterpret_cast<const unsigned long long*>(P.bt + cbt);
				const uint32_t words = 1u << (cf - 6);
				unsigned long long wv[8];
#pragma unroll
				for (int p = 0; p < 8; ++p) wv[p] = (uint32_t)p < cnplanes ? planes[(size_t)(p * T + tprev) * words + (y >> 6)] : 0ull;
				uint32_t v = 0;
#pragma unroll
				for (int p = 0; p < 8; ++p) v |= (uint32_t)((wv[p] >> (y & 63u)) & 1ull) << p;
				const uint32_t e = v & ((1u << cebits) - 1u);
				aj = v >> cebits;
				xp = deposit(y, segs, nsf) | deposit(e, segs + nsf, nse);
			} else {
				const uint32_t raw = reinterpret_cast<const uint32_t*>(P.bt + cbt)[(size_t)y * T + tprev];
				const uint32_t r = raw >> KEY_JBITS;
				xp = r ^ (r >> 1);
				aj = raw & KEY_JMASK;
			}
			if (lane == 0) {
				path_index[c] = xp;
				path_trans[c] = tprev;
			}
			tprev = aj;
			x = xp;
		} else if (kind == 3) {
			// ---- pedigree slot run: physical exit index from the logical one, the record of the path's workgroup -> LDS
			const uint32_t L = h[5], rec_words = h[6], f_exit = h[12];
			const uint8_t* exit_slot = reinterpret_cast<const uint8_t*>(h + 16);
			uint32_t pexit = 0;
			for (uint32_t j = 0; j < f_exit; ++j) pexit |= ((x >> j) & 1u) << exit_slot[j];
			slot_w = pexit >> L;
			slot_l = pexit & ((1u << L) - 1u);
			const uint32_t stage_words = rec_words / 2u;
			const unsigned long long* __restrict__ gst = reinterpret_cast<const unsigned long long*>(
				P.bt + (((unsigned long long)h[9] << 32) | h[8]) + (size_t)slot_w * rec_words * 4u);
			unsigned long long sv[2];
#pragma unroll
			for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + lane; sv[u] = i < stage_words ? gst[i] : 0ull; }
#pragma unroll
			for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + lane; if (i < stage_words) stage[i] = sv[u]; }
			for (uint32_t i = 2 * NT + lane; i < stage_words; i += NT) stage[i] = gst[i];
		} else if (kind == 2) {
			// ---- slot run (slots.h): physical exit index from the logical one, then the record of the workgroup the path
			// runs through (the complement workgroup's when that half was not computed) -> LDS
			const uint32_t g = h[4], L = h[5], n_ends = h[6], threads = h[7], f_exit = h[12];
			const uint8_t* exit_slot = reinterpret_cast<const uint8_t*>(h + 16);
			uint32_t pexit = 0;
			for (uint32_t j = 0; j < f_exit; ++j) pexit |= ((x >> j) & 1u) << exit_slot[j];
			slot_w = pexit >> L;
			slot_l = pexit & ((1u << L) - 1u);
			slot_mirrored = h[10] && ((slot_w >> (g - 1u)) & 1u);
			const uint32_t wrec = slot_mirrored ? (~slot_w & ((1u << g) - 1u)) : slot_w;
			const uint32_t stage_words = n_ends * threads / 8u;
			const unsigned long long* __restrict__ gst = reinterpret_cast<const unsigned long long*>(
				P.bt + (((unsigned long long)h[9] << 32) | h[8]) + (size_t)wrec * n_ends * threads);
			unsigned long long sv[2];
#pragma unroll
			for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + lane; sv[u] = i < stage_words ? gst[i] : 0ull; }
#pragma unroll
			for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + lane; if (i < stage_words) stage[i] = sv[u]; }
			for (uint32_t i = 2 * NT + lane; i < stage_words; i += NT) stage[i] = gst[i];
		} else {
			// ---- resident run [c0, c0 + ncols): this workgroup's record -> LDS
			const uint32_t g = h[4], Lf_last = h[5], stage_words = h[6], n_wext = h[7];
			const uint32_t yexit = x & ((1u << (Lf_last + g)) - 1u);
			uint32_t w = 0;
			for (uint32_t i = 0; i < n_wext; ++i) {
				const uint32_t r = h[12 + i];
				w |= ((yexit >> (r & 31u)) & ((1u << ((r >> 16) & 31u)) - 1u)) << ((r >> 8) & 31u);
			}
			wrun = w;
			// halved run (ResSegment::half) and the path lies in the half that was not computed: the record of the
			// complement workgroup holds the mirror-image decisions (bits 4..7 of every record byte)
			const uint32_t wrec = (((h[11] >> 20) & 1u) && ((w >> (g - 1u)) & 1u)) ? (~w & ((1u << g) - 1u)) : w;
			const unsigned long long* __restrict__ gst = reinterpret_cast<const unsigned long long*>(
				P.bt + (((unsigned long long)h[9] << 32) | h[8])) + (size_t)wrec * stage_words;
			unsigned long long sv[2];
#pragma unroll
			for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + lane; sv[u] = i < stage_words ? gst[i] : 0ull; }
#pragma unroll
			for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + lane; if (i < stage_words) stage[i] = sv[u]; }
			for (uint32_t i = 2 * NT + lane; i < stage_words; i += NT) stage[i] = gst[i];
		}
		// land the prefetches
		if (hload) hdr[((ui + 2) & 3u) * 32 + lane] = hv;
		{
			uint32_t* rnext = recs0 + ((ui + 1) & 1u) * RES_MAXCOLS * 32;
#pragma unroll
			for (int u = 0; u < 2; ++u) { const uint32_t i = u * NT + lane; if (i < nrec) rnext[i] = rv[u]; }
			for (uint32_t i = 2 * NT + lane; i < nrec; i += NT) rnext[i] = gnext[i];
		}
		__syncthreads();
		const unsigned long long tb1 = P.dbg ? __builtin_readcyclecounter() : 0ull;
		if (kind == 2) {
			if (lane < 64) {   // one wave follows the path
				const uint32_t g = h[4], L = h[5], n_ends = h[6], threads = h[7], lr = h[13];
				(void)g;
				const SlotBtCol* bcols = reinterpret_cast<const SlotBtCol*>(recs);
				const uint8_t* ends = reinterpret_cast<const uint8_t*>(recs + ncols * 8);
				// lane k holds the slot of ending read k (and k + 64): the chain fetches it with v_readlane
				const uint32_t e_lo = lane < n_ends ? ends[lane] : 0u, e_hi = lane + 64u < n_ends ? ends[lane + 64u] : 0u;
				const uint8_t* stage8 = reinterpret_cast<const uint8_t*>(stage);
				const uint32_t lmask = (1u << L) - 1u;
				uint32_t l = slot_l;
				if (lane == 0) cells[n_ends] = l;
				// state S[k] = local index of the path after undoing the ending reads k, k+1, ...: one dependent LDS byte per step
				for (uint32_t k = n_ends; k-- > 0;) {
					const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)(k < 64u ? e_lo : e_hi), (int)(k & 63u));
					const uint32_t look = slot_mirrored ? ((~l & lmask) | (1u << j)) : (l & ~(1u << j));
					const uint32_t byte = stage8[k * threads + (look >> lr)];
					const uint32_t bit = (byte >> (look & ((1u << lr) - 1u))) & 1u;
					l = (l & ~(1u << j)) | (bit << j);
					if (lane == 0) cells[k] = l;
				}
				__builtin_amdgcn_wave_barrier();
				uint32_t xl = 0;
				if (lane < ncols) {
					const SlotBtCol& bc = bcols[lane];
					const uint32_t pc = (slot_w << L) | cells[bc.kf];
					for (uint32_t j = 0; j < bc.k; ++j) xl |= ((pc >> bc.slot[j]) & 1u) << j;
					path_index[c0 + lane] = xl;
					path_trans[c0 + lane] = 0u;
				}
				if (lane == 0) { xshare[0] = xl; xshare[1] = 0u; }
			}
			__syncthreads();
			x = xshare[0];
			tprev = xshare[1];
		}
		if (kind == 3) {
			if (lane < 64) {   // one wave follows the path
				const uint32_t L = h[5], threads = h[7], tb = h[13];
				const SlotBtCol* bcols = reinterpret_cast<const SlotBtCol*>(recs);
				const uint32_t thand = pedslot_walk(bcols, reinterpret_cast<const uint8_t*>(stage), ncols, threads, tb, slot_l, tprev, cells, lane == 0);
				__builtin_amdgcn_wave_barrier();
				uint32_t xl = 0;
				if (lane < ncols) {
					const SlotBtCol& bc = bcols[lane];
					const uint32_t pc = (slot_w << L) | cells[lane];
					for (uint32_t j = 0; j < bc.k; ++j) xl |= ((pc >> bc.slot[j]) & 1u) << j;
					path_index[c0 + lane] = xl;
					path_trans[c0 + lane] = cells[64u + lane];
				}
				if (lane == 0) { xshare[0] = xl; xshare[1] = thand; }
			}
			__syncthreads();
			x = xshare[0];
			tprev = xshare[1];
		}
		if (kind == 1) {
			if (lane < 64) {  // one wave follows the path; the others only helped with the copies
				// local exit index of the path
				const uint32_t yexit = x & ((1u << (h[5] + h[4])) - 1u);
				uint32_t l = 0;
				for (uint32_t i = 0; i < h[10]; ++i) {
					const uint32_t r = h[18 + i];
					l |= ((yexit >> (r & 31u)) & ((1u << ((r >> 16) & 31u)) - 1u)) << ((r >> 8) & 31u);
				}
				// sequential part, in local index space: only columns where a read ends touch the record.  Lanes keep the
				// per-column parameters in registers; the loop fetches them with v_readlane (off the dependent chain), so the
				// chain per visited column is: mask, record byte from LDS, bit insert.
				const unsigned long long tw0 = P.dbg ? __builtin_readcyclecounter() + (l & 0u) : 0ull;
				const uint32_t n_active = h[11] & 0xFFFFu, simple = (h[11] >> 16) & 15u;
				const uint32_t* rmine = recs + (lane < ncols ? lane : 0u) * 32;
				uint32_t tcur = tprev, mycell = 0, myts = 0;
				if (simple == 2u) {
					// trio, at most one read ends per column: every column reads one record byte (the transmission argmin lives
					// there), the chain per column is mask, byte, (bit insert); parameters by v_readlane from lane ci
					const uint4 q0 = *reinterpret_cast<const uint4*>(rmine);      // Lf, ebits, layout, stage_off
					const uint32_t p_mask = (1u << q0.x) - 1u, p_soff = q0.w * 8u, p_eb = q0.y | (rmine[5] << 8);
					const uint8_t* stage8 = reinterpret_cast<const uint8_t*>(stage);
					for (uint32_t ci = ncols; ci-- > 0;) {
						const uint32_t s_mask = __builtin_amdgcn_readlane(p_mask, ci), s_soff = __builtin_amdgcn_readlane(p_soff, ci),
						               s_eb = __builtin_amdgcn_readlane(p_eb, ci);
						const uint32_t lout = l & s_mask;
						const uint32_t fld = stage8[s_soff + lout * 4u + tcur];
						const uint32_t e0 = s_eb >> 8;
						const uint32_t with_bit = insert_zero(lout, e0) | ((fld & 1u) << e0);
						const uint32_t cell = (s_eb & 255u) ? with_bit : lout;
						if (lane == ci) { mycell = cell; myts = tcur; }
						tcur = (fld >> 3) & 3u;
						l = cell;
					}
				} else if (simple) {
					// single individual, every record one byte per thread: the chain visits only the columns in which a read ends
					// (ResBacktrace kpos / src / cmask / kcol); lane k holds the parameters of chain position k
					const uint32_t kc = rmine[31] < ncols ? rmine[31] : 0u;
					const uint32_t* rk = recs + kc * 32;
					const uint32_t c_cmask = rk[30], c_soff = rk[3] * 8u, c_e0 = rk[5], c_lfmask = (1u << rk[0]) - 1u;
					const uint32_t gg = h[4];
					const bool mirrored = ((h[11] >> 20) & 1u) && ((wrun >> (gg - 1u)) & 1u);
					const uint32_t mxor = mirrored ? 0xFFFFFFFFu : 0u, mshift = mirrored ? 4u : 0u;
					const uint32_t my_kpos = rmine[28], my_src = rmine[29], my_cmask = rmine[30];
					const uint8_t* stage8 = reinterpret_cast<const uint8_t*>(stage);
					const uint32_t l_exit = l;
					for (uint32_t k = 0; k < n_active; ++k) {
						const uint32_t s_cmask = __builtin_amdgcn_readlane(c_cmask, k), s_soff = __builtin_amdgcn_readlane(c_soff, k),
						               s_e0 = __builtin_amdgcn_readlane(c_e0, k);
						const uint32_t lout = l & s_cmask;
						const uint32_t look = (lout ^ mxor) & __builtin_amdgcn_readlane(c_lfmask, k);  // mirrored: the complement entry
						const uint32_t byte = stage8[s_soff + (look >> 2)];
						const uint32_t cell = insert_zero(lout, s_e0) | (((byte >> ((look & 3u) + mshift)) & 1u) << s_e0);
						if (my_kpos == k) mycell = cell;
						l = cell;
					}
					// columns without an ending read: the cell of the next active column above (or the exit index), masked
					const uint32_t from = __shfl(mycell, my_src == RES_BT_NONE ? 0u : my_src);
					if (my_kpos == RES_BT_NONE) mycell = (my_src == RES_BT_NONE ? l_exit : from) & my_cmask;
				} else {
					const uint4 q0 = *reinterpret_cast<const uint4*>(rmine);      // Lf, ebits, layout, stage_off
					const uint4 q1 = *reinterpret_cast<const uint4*>(rmine + 4);  // nwords, epos0, epos1, epos2
					const uint32_t p_mask = (1u << q0.x) - 1u, p_eb = q0.y | (q0.z << 8), p_soff = q0.w * 8u, p_e0 = q1.y, p_e1 = q1.z, p_e2 = q1.w, p_nw = q1.x;
					const uint8_t* stage8 = reinterpret_cast<const uint8_t*>(stage);
					for (uint32_t ci = ncols; ci-- > 0;) {
						const uint32_t s_mask = __builtin_amdgcn_readlane(p_mask, ci), s_eb = __builtin_amdgcn_readlane(p_eb, ci);
						const uint32_t lout = l & s_mask;
						uint32_t cell = lout;
						const uint32_t eb = s_eb & 255u, layout = s_eb >> 8;
						if (layout == 2u) {  // trio: one byte per (entry, transmission value): ending-read bits | argj << 3
							const uint32_t s_soff = __builtin_amdgcn_readlane(p_soff, ci);
							const uint32_t fld = stage8[s_soff + lout * 4u + tcur] & 31u;
							if (eb) {
								const uint32_t epos[3] = {(uint32_t)__builtin_amdgcn_readlane(p_e0, ci), (uint32_t)__builtin_amdgcn_readlane(p_e1, ci),
								                          (uint32_t)__builtin_amdgcn_readlane(p_e2, ci)};
								uint32_t bits = 0;
	#pragma unroll
								for (int q = 0; q < 3; ++q) {
									if ((uint32_t)q < eb) {
										cell = insert_zero(cell, epos[q]);
										bits |= ((fld >> q) & 1u) << epos[q];
									}
								}
								cell |= bits;
							}
							if (lane == ci) myts = tcur;
							tcur = fld >> 3;
						} else if (eb) {
							const uint32_t s_soff = __builtin_amdgcn_readlane(p_soff, ci), s_e0 = __builtin_amdgcn_readlane(p_e0, ci);
							if (layout == 1u) {  // one byte per thread: bit (lout & 3) of byte lout >> 2
								const uint32_t byte = stage8[s_soff + (lout >> 2)];
								cell = insert_zero(lout, s_e0) | (((byte >> (lout & 3u)) & 1u) << s_e0);
							} else {             // ballot planes, up to 3 ending reads (ascending positions)
								const uint32_t epos[3] = {s_e0, (uint32_t)__builtin_amdgcn_readlane(p_e1, ci), (uint32_t)__builtin_amdgcn_readlane(p_e2, ci)};
								const uint32_t s_nw = __builtin_amdgcn_readlane(p_nw, ci);
								uint32_t bits = 0;
	#pragma unroll
								for (int q = 0; q < 3; ++q) {
									if ((uint32_t)q < eb) {
										cell = insert_zero(cell, epos[q]);
										const unsigned long long word = stage[(s_soff >> 3) + q * s_nw + (lout >> 6)];
										bits |= (uint32_t)((word >> (lout & 63u)) & 1ull) << epos[q];
									}
								}
								cell |= bits;
							}
						}
						if (lane == ci) mycell = cell;
						l = cell;
					}
				}
				const unsigned long long tw1 = P.dbg ? __builtin_readcyclecounter() + (l & 0u) : 0ull;
				// logical indices, one lane per column
				uint32_t xl = 0;
				if (lane < ncols) {
					const uint32_t* rb = recs + lane * 32;
					const uint32_t cell = mycell;
					const uint32_t ng = rb[8], nl = rb[9];
					for (uint32_t i = 0; i < ng; ++i) {
						const uint32_t r = rb[10 + i];
						xl |= ((wrun >> (r & 31u)) & ((1u << ((r >> 16) & 31u)) - 1u)) << ((r >> 8) & 31u);
					}
					for (uint32_t i = 0; i < nl; ++i) {
						const uint32_t r = rb[18 + i];
						xl |= ((cell >> (r & 31u)) & ((1u << ((r >> 16) & 31u)) - 1u)) << ((r >> 8) & 31u);
					}
					path_index[c0 + lane] = xl;
					path_trans[c0 + lane] = rb[2] == 2u ? myts : 0u;
				}
				if (lane == 0) { xshare[0] = xl; xshare[1] = tcur; }
				if (P.dbg) { const unsigned long long tw2 = __builtin_readcyclecounter() + (xl & 0u); bt_a += tw0 - tb1; bt_b += tw1 - tw0; bt_c += tw2 - tw1; }
			}
			__syncthreads();
			x = xshare[0];
			tprev = xshare[1];
			if (P.dbg) { bt_load += tb1 - tb0; bt_walk += __builtin_readcyclecounter() - tb1; bt_runs++; }
		}
	}
	if (P.bt_state && lane == 0) {
		P.bt_state[0] = x;
		P.bt_state[1] = tprev;
	}
	if (P.dbg && lane == 0) {
		unsigned long long* d = P.dbg + P.dbg_wg_off + 4 * 512 * 2;
		d[0] = bt_load; d[1] = bt_walk; d[2] = bt_runs; d[3] = bt_a; d[4] = bt_b; d[5] = bt_c;
	}
}

// ------------------------------------------------------------------------------------------------ chunked backtrace
// The walk above is one dependent chain over the whole table (a run's record cannot be fetched before the path's
// workgroup index in that run is known).  For a table made of slot runs the chain is cut into CHUNKS of consecutive
// units: every chunk but the newest starts from a *guess* of the path's state at its end -- the minimum of the exit
// column there, left by the forward pass (SlotRun::spec_id) -- and all chunks are walked at once, one workgroup each
// (mode 0).  A second launch (mode 1, one workgroup) goes over the chunk boundaries newest to oldest: where the state
// the true path arrives with equals the guess, the chunk's speculative walk IS the true path; where it does not, the
// chunk is walked again from the true state until the path reaches a state the speculative walk went through at the
// same unit boundary (from there on the two are identical) or the chunk ends.  By induction the result is exactly
// the path the sequential walk finds; the guess only decides how much is walked twice.
// Orientations.  The minimum of a projection column is attained by every image of a state under the table's symmetries:
// relabelling the two haplotypes of a FOUNDER (complement the bits of its reads, and flip the transmission bits of the trios
// it is a parent of) leaves every cost unchanged.  A single individual has one such generator (all bits), a trio or a quartet
// two (father, mother); which image the true path runs through is decided only at the table's last column -- all of them are
// walked (one path buffer each), the verification picks.  (An image's walk is NOT the image of the walk: ties break
// differently, the records hold every decision.)
// Where every genotype is heterozygous (the synthetic benchmarks; long stretches of real data) complementing a CHILD's reads
// together with both of its transmission bits is a symmetry too: the child then carries its parents' other haplotypes, which
// hold the complementary alleles.  It is not exact in general -- a guess only has to be right often, the verification keeps the
// result exact -- so children are generators as well: up to 4 generators (a quartet), 16 guesses per chunk.
constexpr uint32_t BT_GENERATORS = 4;
constexpr uint32_t BT_ORIENT = 1u << BT_GENERATORS;
struct BtChunk {
	uint32_t unit_off, unit_count;
	uint32_t spec_id;   // 0: the newest chunk (units[0] is the table's last column, the optimum comes from P.last_keys)
	uint32_t n_orient;  // 2^generators in use
	uint32_t flip[BT_GENERATORS];   // packed-state XOR of each generator at this chunk's entry (index bits | transmission bits << BT_STATE_TSHIFT)
};
__device__ __forceinline__ uint32_t bt_orient(const BtChunk& ch, uint32_t state, uint32_t o) {
#pragma unroll
	for (uint32_t q = 0; q < BT_GENERATORS; ++q) state ^= ((o >> q) & 1u) ? ch.flip[q] : 0u;
	return state;
}

// State of the walk between units, packed into one word: logical index of the path at the first column of the unit walked
// before (bits 0..25) | transmission value handed down (bits 26..31; 0 for a single individual): device_types.h.

// One unit for the whole workgroup (256 threads): takes the packed state at the first column of the unit walked before (later
// in the table), returns the packed state at this unit's first column.  Column steps (any T), slot runs (T = 1), trio runs.
__device__ __forceinline__ uint32_t chunk_walk_unit(const DevProblem& P, const BtUnit* __restrict__ unit, uint32_t state, uint32_t* hdr, uint32_t* blob,
                                                    uint32_t* cells, uint32_t* xshare, unsigned long long* stage,
                                                    uint32_t* __restrict__ path_index, uint32_t* __restrict__ path_trans) {
	const uint32_t tid = threadIdx.x, NT = blockDim.x;
	const uint32_t x = state & BT_STATE_XMASK, tprev = state >> BT_STATE_TSHIFT;
	__syncthreads();   // the previous unit's readers are done with the LDS areas
	if (tid < 32) hdr[tid] = reinterpret_cast<const uint32_t*>(unit)[tid];
	__syncthreads();
	const uint32_t kind = hdr[0], c0 = hdr[1], ncols = hdr[2];
	if (kind == 0) {
		// one column through the column kernels' records (word layout: backtrace_kernel above)
		const uint32_t T = P.T;
		const uint32_t cf = hdr[4], cmode = hdr[5], cnplanes = hdr[6], cebits = hdr[7], nsf = hdr[10], nse = hdr[11];
		const unsigned long long cbt = ((unsigned long long)hdr[9] << 32) | hdr[8];
		const uint32_t* segs = hdr[28] ? (hdr + 12) : (P.segs + P.cols[c0].seg_off);
		const uint32_t y = cf >= 32 ? x : (x & ((1u << cf) - 1u));
		uint32_t xp, aj;
		if (cmode == 0) {
			const unsigned long long* planes = reinterpret_cast<const unsigned long long*>(P.bt + cbt);
			const uint32_t words = 1u << (cf - 6);
			uint32_t v = 0;
			for (uint32_t p = 0; p < cnplanes; ++p) v |= (uint32_t)((planes[(size_t)(p * T + tprev) * words + (y >> 6)] >> (y & 63u)) & 1ull) << p;
			const uint32_t e = v & ((1u << cebits) - 1u);
			aj = v >> cebits;
			xp = deposit(y, segs, nsf) | deposit(e, segs + nsf, nse);
		} else {
			const uint32_t raw = reinterpret_cast<const uint32_t*>(P.bt + cbt)[(size_t)y * T + tprev];
			const uint32_t r = raw >> KEY_JBITS;
			xp = r ^ (r >> 1);
			aj = raw & KEY_JMASK;
		}
		if (tid == 0) { path_index[c0] = xp; path_trans[c0] = tprev; }
		return xp | (aj << BT_STATE_TSHIFT);
	}
	if (kind == 1) {
		// ---- LDS-resident run of a trio (kernels_trio.h): per-column parameters (ResBacktrace, 32 words each) and the record of the
		// workgroup the path runs through -> LDS; one wave follows the argmins column by column (every column carries the
		// transmission argmin).  Same record semantics as the walk in backtrace_kernel.
		const uint32_t g = hdr[4], Lf_last = hdr[5], stage_words = hdr[6], n_wext = hdr[7];
		const uint32_t yexit = x & ((1u << (Lf_last + g)) - 1u);
		uint32_t w = 0;
		for (uint32_t i = 0; i < n_wext; ++i) {
			const uint32_t r = hdr[12 + i];
			w |= ((yexit >> (r & 31u)) & ((1u << ((r >> 16) & 31u)) - 1u)) << ((r >> 8) & 31u);
		}
		const uint32_t* __restrict__ grecs = reinterpret_cast<const uint32_t*>(P.res_bt + hdr[3]);
		for (uint32_t i = tid; i < ncols * 32u; i += NT) blob[i] = grecs[i];
		const unsigned long long* __restrict__ gst = reinterpret_cast<const unsigned long long*>(
			P.bt + (((unsigned long long)hdr[9] << 32) | hdr[8])) + (size_t)w * stage_words;
		for (uint32_t i = tid; i < stage_words; i += NT) stage[i] = gst[i];
		__syncthreads();
		if (tid < 64) {
			uint32_t l = 0;
			for (uint32_t i = 0; i < hdr[10]; ++i) {
				const uint32_t r = hdr[18 + i];
				l |= ((yexit >> (r & 31u)) & ((1u << ((r >> 16) & 31u)) - 1u)) << ((r >> 8) & 31u);
			}
			const uint8_t* stage8 = reinterpret_cast<const uint8_t*>(stage);
			uint32_t tcur = tprev;
			for (uint32_t ci = ncols; ci-- > 0;) {
				const uint32_t* rb = blob + ci * 32u;   // Lf, ebits, layout (2 for a trio column), stage_off | nwords, epos0, epos1, epos2
				const uint32_t lout = l & ((1u << rb[0]) - 1u), eb = rb[1];
				const uint32_t fld = stage8[rb[3] * 8u + lout * 4u + tcur] & 31u;   // ending-read bits | argj << 3
				uint32_t cell = lout, bits = 0;
				for (uint32_t q = 0; q < eb && q < 3u; ++q) {
					cell = insert_zero(cell, rb[5 + q]);
					bits |= ((fld >> q) & 1u) << rb[5 + q];
				}
				cell |= bits;
				if (tid == 0) { cells[ci] = cell; cells[64 + ci] = tcur; }
				tcur = fld >> 3;
				l = cell;
			}
			if (tid == 0) xshare[1] = tcur;
		}
		__syncthreads();
		if (tid < ncols) {   // logical index of column tid: deposits of the workgroup index and of the local cell
			const uint32_t* rb = blob + tid * 32u;
			const uint32_t cell = cells[tid], ng = rb[8], nl = rb[9];
			uint32_t xl = 0;
			for (uint32_t i = 0; i < ng; ++i) {
				const uint32_t r = rb[10 + i];
				xl |= ((w >> (r & 31u)) & ((1u << ((r >> 16) & 31u)) - 1u)) << ((r >> 8) & 31u);
			}
			for (uint32_t i = 0; i < nl; ++i) {
				const uint32_t r = rb[18 + i];
				xl |= ((cell >> (r & 31u)) & ((1u << ((r >> 16) & 31u)) - 1u)) << ((r >> 8) & 31u);
			}
			path_index[c0 + tid] = xl;
			path_trans[c0 + tid] = cells[64 + tid];
			if (tid == 0) xshare[0] = xl;
		}
		__syncthreads();
		return xshare[0] | (xshare[1] << BT_STATE_TSHIFT);
	}
	if (kind == 3) {
		// ---- pedigree slot run: blob (column slot lists), physical exit index, record of the path's workgroup -> LDS
		const uint32_t L = hdr[5], rec_words = hdr[6], threads = hdr[7], f_exit = hdr[12], tb = hdr[13];
		const uint32_t* __restrict__ gblob = P.slot_blob + hdr[3];
		for (uint32_t i = tid; i < hdr[11]; i += NT) blob[i] = gblob[i];
		const uint8_t* exit_slot = reinterpret_cast<const uint8_t*>(hdr + 16);
		uint32_t pexit = 0;
		for (uint32_t j = 0; j < f_exit; ++j) pexit |= ((x >> j) & 1u) << exit_slot[j];
		const uint32_t w = pexit >> L;
		const unsigned long long* __restrict__ gst = reinterpret_cast<const unsigned long long*>(
			P.bt + (((unsigned long long)hdr[9] << 32) | hdr[8]) + (size_t)w * rec_words * 4u);
		for (uint32_t i = tid; i < rec_words / 2u; i += NT) stage[i] = gst[i];
		__syncthreads();
		const SlotBtCol* bcols = reinterpret_cast<const SlotBtCol*>(blob);
		if (tid < 64) {
			const uint32_t thand = pedslot_walk(bcols, reinterpret_cast<const uint8_t*>(stage), ncols, threads, tb, pexit & ((1u << L) - 1u), tprev, cells, tid == 0);
			if (tid == 0) xshare[1] = thand;
		}
		__syncthreads();
		for (uint32_t c = tid >> 6; c < ncols; c += NT >> 6) {
			const SlotBtCol& bc = bcols[c];
			const uint32_t pc = (w << L) | cells[c];
			const uint32_t j = tid & 63u;
			const bool bit = j < bc.k && j < 25u && ((pc >> bc.slot[j < 25u ? j : 0u]) & 1u);
			const uint32_t xl = (uint32_t)__ballot(bit);
			if (j == 0) {
				path_index[c0 + c] = xl;
				path_trans[c0 + c] = cells[64u + c];
				if (c == 0) xshare[0] = xl;
			}
		}
		__syncthreads();
		return xshare[0] | (xshare[1] << BT_STATE_TSHIFT);
	}
	// ---- slot run: blob (column slot lists + ending slots), physical exit index, record of the path's workgroup -> LDS
	const uint32_t g = hdr[4], L = hdr[5], n_ends = hdr[6], threads = hdr[7], f_exit = hdr[12], lr = hdr[13];
	const uint32_t* __restrict__ gblob = P.slot_blob + hdr[3];
	for (uint32_t i = tid; i < hdr[11]; i += NT) blob[i] = gblob[i];
	const uint8_t* exit_slot = reinterpret_cast<const uint8_t*>(hdr + 16);
	uint32_t pexit = 0;
	for (uint32_t j = 0; j < f_exit; ++j) pexit |= ((x >> j) & 1u) << exit_slot[j];
	const uint32_t w = pexit >> L, lmask = (1u << L) - 1u;
	const bool mirrored = hdr[10] && ((w >> (g - 1u)) & 1u);
	const uint32_t wrec = mirrored ? (~w & ((1u << g) - 1u)) : w;
	const uint32_t stage_words = n_ends * threads / 8u;
	const unsigned long long* __restrict__ gst = reinterpret_cast<const unsigned long long*>(
		P.bt + (((unsigned long long)hdr[9] << 32) | hdr[8]) + (size_t)wrec * n_ends * threads);
	for (uint32_t i = tid; i < stage_words; i += NT) stage[i] = gst[i];
	__syncthreads();
	const SlotBtCol* bcols = reinterpret_cast<const SlotBtCol*>(blob);
	if (tid < 64) {   // one wave follows the path: state S[k] = local index after undoing the ending reads k, k + 1, ...
		const uint8_t* ends = reinterpret_cast<const uint8_t*>(blob + ncols * 8);
		const uint8_t* stage8 = reinterpret_cast<const uint8_t*>(stage);
		uint32_t l = pexit & lmask;
		if (tid == 0) cells[n_ends] = l;
		for (uint32_t k = n_ends; k-- > 0;) {
			const uint32_t j = ends[k];
			const uint32_t look = mirrored ? ((~l & lmask) | (1u << j)) : (l & ~(1u << j));
			const uint32_t byte = stage8[k * threads + (look >> lr)];
			l = (l & ~(1u << j)) | (((byte >> (look & ((1u << lr) - 1u))) & 1u) << j);
			if (tid == 0) cells[k] = l;
		}
	}
	__syncthreads();
	// logical index of every column: lane j of a wave tests the slot of logical bit j, the ballot is the index
	for (uint32_t c = tid >> 6; c < ncols; c += NT >> 6) {
		const SlotBtCol& bc = bcols[c];
		const uint32_t pc = (w << L) | cells[bc.kf];
		const uint32_t j = tid & 63u;
		const bool bit = j < bc.k && j < 28u && ((pc >> bc.slot[j < 28u ? j : 0u]) & 1u);
		const uint32_t xl = (uint32_t)__ballot(bit);
		if (j == 0) {
			path_index[c0 + c] = xl;
			path_trans[c0 + c] = 0u;
			if (c == 0) xshare[0] = xl;
		}
	}
	__syncthreads();
	return xshare[0];
}

// mode 0: blockIdx.x = 2 * chunk + orientation.  The single-individual table is symmetric under complementing every read
// (D[~x] == D[x]), so the minimum of an exit column is always attained twice, by a state and by its complement, and which of
// the two the true path runs through is decided only at the table's last column: both are walked (into path buffer 0 / 1),
// the verification picks the one the true path arrives at.  (The complement is NOT simply the complement path: ties break
// differently for the two, the records hold both decisions -- slots.h.)
// (`bx`: the block's index in x -- blockIdx.x of a launch for ONE table, or of a launch whose blockIdx.y selects the table: backtrace_chunks_group)
__device__ __forceinline__ void backtrace_chunks_body(const DevProblem& P, const BtUnit* __restrict__ units, const BtChunk* __restrict__ chunks,
                                                      uint32_t n_chunks, uint32_t n_units, uint32_t mode, uint32_t n_orient_max, uint32_t* __restrict__ path2,
                                                      uint32_t* __restrict__ trans2, uint32_t* __restrict__ out_score,
                                                      uint32_t* __restrict__ unit_x2, uint32_t* __restrict__ guess, uint8_t* __restrict__ sel,
                                                      uint32_t* __restrict__ counters, const uint32_t bx) {
	extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
	uint32_t* hdr = smem;                         // 32 words
	uint32_t* xshare = hdr + 32;                  // 4 words
	uint32_t* cells = xshare + 4;                 // BT_CELLS words
	uint32_t* blob = cells + BT_CELLS;            // BT_CHUNK_BLOB words: slot blob, or the ResBacktrace records of a trio run
	unsigned long long* stage = reinterpret_cast<unsigned long long*>(blob + BT_CHUNK_BLOB);
	const uint32_t tid = threadIdx.x, n = P.n_cols;
	if (mode == 0) {
		const uint32_t ci = bx / n_orient_max, o = bx % n_orient_max;   // (the grid is n_chunks x the most orientations any chunk has)
		const BtChunk ch = chunks[ci];
		if (o >= ch.n_orient) return;
		const BtUnit* __restrict__ cu = units + ch.unit_off;
		uint32_t* __restrict__ path = path2 + (size_t)o * n;
		uint32_t* __restrict__ ptrans = trans2 + (size_t)o * n;
		uint32_t* __restrict__ unit_x = unit_x2 + (size_t)o * n_units;
		uint32_t x, first = 0;
		if (ch.spec_id == 0) {
			if (o) return;
			// the table's last column: first (rank(x), i) attaining the minimum (strict '<' scan, src/pedigreedptable.cpp:306-315)
			unsigned long long key = ~0ull;
			uint32_t t_last = 0;
			for (uint32_t i = 0; i < P.T; ++i) {
				const unsigned long long k2 = P.last_keys[i];
				if ((k2 >> KEY_JBITS) < (key >> KEY_JBITS)) { key = k2; t_last = i; }
			}
			const uint32_t rlast = (uint32_t)(key >> KEY_JBITS) & BT_STATE_XMASK;
			x = (rlast ^ (rlast >> 1)) | (((uint32_t)key & KEY_JMASK) << BT_STATE_TSHIFT);   // index of the last column | the argj it hands down
			if (tid == 0) {
				out_score[0] = key == ~0ull ? 0xFFFFFFFFu : (uint32_t)(key >> 32);
				path[n - 1] = x & BT_STATE_XMASK;
				ptrans[n - 1] = t_last;
				unit_x[ch.unit_off] = x;
			}
			first = 1;
		} else if (cu->kind == 3u) {
			// pedigree slot run: the smallest entry of the exit column (key = value << 32 | exit index * T + t)
			const unsigned long long* __restrict__ cand = P.spec_keys + (size_t)(ch.spec_id - 1u) * P.spec_stride;
			unsigned long long best = ~0ull;
			for (uint32_t i = tid; i < P.spec_stride; i += blockDim.x) best = min(best, cand[i]);
#pragma unroll
			for (int m = 1; m < 64; m <<= 1) {
				const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)best, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), m);
				best = min(best, ((unsigned long long)hi << 32) | lo);
			}
			unsigned long long* red = reinterpret_cast<unsigned long long*>(cells);
			if ((tid & 63u) == 0) red[tid >> 6] = best;
			__syncthreads();
			for (uint32_t i = 0; i < (blockDim.x >> 6); ++i) best = min(best, red[i]);
			const SlotBtUnit* su = reinterpret_cast<const SlotBtUnit*>(cu);
			const uint32_t idx = (uint32_t)best >> su->lr, tt = (uint32_t)best & ((1u << su->lr) - 1u);   // (lr holds log2 T for these units)
			x = 0;
			for (uint32_t j = 0; j < su->f_exit; ++j) x |= ((idx >> su->exit_pos[j]) & 1u) << j;
			x |= tt << BT_STATE_TSHIFT;
			if (tid == 0 && o == 0) guess[ci] = x;
			x = bt_orient(ch, x, o);
		} else if (cu->kind == 1u) {
			// trio: the smallest entry (y, t) of the projection column this chunk's newest run left (key = value << 32 | y * 4 + t)
			const unsigned long long* __restrict__ cand = P.spec_keys + (size_t)(ch.spec_id - 1u) * P.spec_stride;
			unsigned long long best = ~0ull;
			for (uint32_t i = tid; i < P.spec_stride; i += blockDim.x) best = min(best, cand[i]);
#pragma unroll
			for (int m = 1; m < 64; m <<= 1) {
				const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)best, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), m);
				best = min(best, ((unsigned long long)hi << 32) | lo);
			}
			unsigned long long* red = reinterpret_cast<unsigned long long*>(cells);
			if ((tid & 63u) == 0) red[tid >> 6] = best;
			__syncthreads();
			for (uint32_t i = 0; i < (blockDim.x >> 6); ++i) best = min(best, red[i]);
			const uint32_t idx = (uint32_t)best;
			x = (idx >> 2) | ((idx & 3u) << BT_STATE_TSHIFT);
			if (tid == 0 && o == 0) guess[ci] = x;
			x = bt_orient(ch, x, o);
		} else {
			// guess: the smallest entry of the column this chunk's newest run left (exit index -> logical exit index), or its complement
			const unsigned long long* __restrict__ cand = P.spec_keys + (size_t)(ch.spec_id - 1u) * P.spec_stride;
			unsigned long long best = ~0ull;
			for (uint32_t i = tid; i < P.spec_stride; i += blockDim.x) best = min(best, cand[i]);
#pragma unroll
			for (int m = 1; m < 64; m <<= 1) {
				const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)best, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), m);
				best = min(best, ((unsigned long long)hi << 32) | lo);
			}
			unsigned long long* red = reinterpret_cast<unsigned long long*>(cells);
			if ((tid & 63u) == 0) red[tid >> 6] = best;
			__syncthreads();
			for (uint32_t i = 0; i < (blockDim.x >> 6); ++i) best = min(best, red[i]);
			const uint32_t idx = (uint32_t)best;
			const SlotBtUnit* su = reinterpret_cast<const SlotBtUnit*>(cu);
			x = 0;
			for (uint32_t j = 0; j < su->f_exit; ++j) x |= ((idx >> su->exit_pos[j]) & 1u) << j;
			if (tid == 0 && o == 0) guess[ci] = x;
			x = bt_orient(ch, x, o);
		}
		for (uint32_t u = first; u < ch.unit_count; ++u) {
			x = chunk_walk_unit(P, cu + u, x, hdr, blob, cells, xshare, stage, path, ptrans);
			if (tid == 0) unit_x[ch.unit_off + u] = x;
		}
		return;
	}
	// ---- mode 1: boundaries newest to oldest; sel[u] = path buffer that holds the true path of unit u
	uint32_t missed = 0, rewalked = 0, index_only = 0;
	for (uint32_t u = tid; u < chunks[0].unit_count; u += blockDim.x) sel[chunks[0].unit_off + u] = 0;
	uint32_t entry = unit_x2[chunks[0].unit_off + chunks[0].unit_count - 1u];   // the newest chunk started from the true optimum
	for (uint32_t ci = 1; ci < n_chunks; ++ci) {
		const BtChunk ch = chunks[ci];
		const BtUnit* __restrict__ cu = units + ch.unit_off;
		uint32_t x = entry;                                           // the true path's state at the first column of the unit walked before
		const SlotBtUnit* su = reinterpret_cast<const SlotBtUnit*>(cu);
		// the part of the state the chunk's first unit looks at: the low f bits of the index (+ the transmission value)
		const uint32_t fbits = cu->kind == 1u ? cu->g + cu->Lf_last : su->f_exit;
		const uint32_t imask = fbits >= BT_STATE_TSHIFT ? BT_STATE_XMASK : ((1u << fbits) - 1u);
		const uint32_t fmask = imask | ~BT_STATE_XMASK;
		const uint32_t g0 = guess[ci];
		uint32_t from = 0;                                            // units [from, unit_count) come from buffer `o`
		uint32_t o = BT_ORIENT;
		for (uint32_t q = 0; q < ch.n_orient; ++q)
			if (o == BT_ORIENT && ((x ^ bt_orient(ch, g0, q)) & fmask) == 0u) o = q;
		if (o == BT_ORIENT) {
			for (uint32_t q = 0; q < ch.n_orient; ++q)
				if (((x ^ bt_orient(ch, g0, q)) & imask) == 0u) { ++index_only; break; }
			// none of the guesses: walk from the true state (into buffer 0) until the path reaches a state one of the walks went through
			++missed;
			o = 0;
			from = ch.unit_count;
			for (uint32_t u = 0; u < ch.unit_count; ++u) {
				x = chunk_walk_unit(P, cu + u, x, hdr, blob, cells, xshare, stage, path2, trans2);
				++rewalked;
				uint32_t hit = BT_ORIENT;
				for (uint32_t q = 0; q < ch.n_orient; ++q)
					if (hit == BT_ORIENT && x == unit_x2[(size_t)q * n_units + ch.unit_off + u]) hit = q;   // (not written by this launch)
				if (hit != BT_ORIENT) { from = u + 1u; o = hit; break; }
			}
		}
		for (uint32_t u = tid; u < ch.unit_count; u += blockDim.x) sel[ch.unit_off + u] = u < from ? (uint8_t)0 : (uint8_t)o;
		entry = from == ch.unit_count ? x : unit_x2[(size_t)o * n_units + ch.unit_off + ch.unit_count - 1u];
	}
	if (tid == 0) { counters[0] = missed; counters[1] = rewalked; counters[2] = index_only; }
}

__global__ __launch_bounds__(256) void backtrace_chunks(DevProblem P, const BtUnit* __restrict__ units, const BtChunk* __restrict__ chunks,
                                                         uint32_t n_chunks, uint32_t n_units, uint32_t mode, uint32_t n_orient_max, uint32_t* __restrict__ path2,
                                                         uint32_t* __restrict__ trans2, uint32_t* __restrict__ out_score,
                                                         uint32_t* __restrict__ unit_x2, uint32_t* __restrict__ guess, uint8_t* __restrict__ sel,
                                                         uint32_t* __restrict__ counters) {
	backtrace_chunks_body(P, units, chunks, n_chunks, n_units, mode, n_orient_max, path2, trans2, out_score, unit_x2, guess, sel, counters, blockIdx.x);
}

// get_super_reads of a table with ONE individual and trusted genotypes, on the device (src/pedigreedptable.cpp:344-388 + get_alleles,
// src/pedigreecolumncostcomputer.cpp:117-175; the host's statement of the same: problem.cpp finish_columns, first branch): one thread per column, from the signed
// per-bit deltas the forward pass reads (REF +q, ALT -q, BLANK 0) and the index path the backtrace has just written.  cp[side][1] += q for REF, cp[side][0] += q for
// ALT (set_partitioning, :53-76); genotype 0/1: a = 1 costs cp[0][1] + cp[1][0], a = 2 costs cp[0][0] + cp[1][1], `<=` lets the later one win (:131), the quality is
// the difference; homozygous: one assignment, the other allele's best stays INF = -1 as an int (:162).  A column the host has to look at (Mendelian conflict: a
// genotype outside 0..2, a cost of INF) gets the allele SUPERREAD_CONFLICT: the host finds it and runs its own loop, which makes the reference's error.
// 9 MB of column entries per coverage-15 table of 50 000 columns were the host's tail of every solve (2.3 ms per table, 96 tables behind one group launch);
// here they are 3 MB of deltas that are in HBM anyway.
struct SuperreadArgs {
	const int32_t* delta;                // [entries]  (n_ind == 1: a column's deltas lie at its entry offset)
	const unsigned long long* col_ptr;   // [n_cols + 1] entry offsets
	const uint8_t* genotype;             // [n_cols]
	const uint32_t* path_index;          // [n_cols]
	uint32_t* out;                       // [n_cols] qualities, then n_cols bytes haplotype 0, then n_cols bytes haplotype 1
	uint32_t n_cols, pad;
};
constexpr uint8_t SUPERREAD_CONFLICT = 0xFF;
__device__ __forceinline__ void superreads_column(const SuperreadArgs& a, uint32_t c) {
	const uint32_t x = a.path_index[c];
	const unsigned long long off = a.col_ptr[c];
	const uint32_t k = (uint32_t)(a.col_ptr[c + 1] - off);
	const int32_t* __restrict__ d = a.delta + off;
	uint32_t cp[2][2] = {{0u, 0u}, {0u, 0u}};
	for (uint32_t j = 0; j < k; ++j) {
		const int32_t v = d[j];
		const uint32_t ref = v > 0 ? (uint32_t)v : 0u, alt = v < 0 ? (uint32_t)(-v) : 0u;
		if ((x >> j) & 1u) { cp[1][1] += ref; cp[1][0] += alt; }
		else { cp[0][1] += ref; cp[0][0] += alt; }
	}
	const uint32_t g = a.genotype[c];
	uint8_t a0, a1;
	uint32_t quality = 0;
	if (g == 1u) {
		const uint32_t cost1 = cp[0][1] + cp[1][0], cost2 = cp[0][0] + cp[1][1];
		const bool second = cost2 <= cost1;
		a0 = second ? 0 : 1;
		a1 = second ? 1 : 0;
		const int diff = (int)cost1 - (int)cost2;
		quality = (uint32_t)(diff < 0 ? -diff : diff);
		if (quality == 0u) a0 = a1 = 3;   // WHAMD_ALLELE_EQUAL_SCORES
		if ((second ? cost2 : cost1) == 0xFFFFFFFFu) a0 = a1 = SUPERREAD_CONFLICT;
	} else if (g == 0u || g == 2u) {
		const uint32_t al = g == 2u ? 1u : 0u;
		const uint32_t cost = cp[0][al] + cp[1][al];
		a0 = a1 = (uint8_t)al;
		const int diff = (int)cost - (int)0xFFFFFFFFu;
		quality = (uint32_t)(diff < 0 ? -diff : diff);
		if (quality == 0u) a0 = a1 = 3;
		if (cost == 0xFFFFFFFFu) a0 = a1 = SUPERREAD_CONFLICT;
	} else {
		a0 = a1 = SUPERREAD_CONFLICT;
	}
	a.out[c] = quality;
	uint8_t* h = (uint8_t*)(a.out + a.n_cols);
	h[c] = a0;
	h[(size_t)a.n_cols + c] = a1;
}
__global__ __launch_bounds__(256) void superreads_single(SuperreadArgs a) {
	const uint32_t c = blockIdx.x * 256u + threadIdx.x;
	if (c < a.n_cols) superreads_column(a, c);
}

// The chunked backtrace of SEVERAL tables in one launch (whamd_dptable_enqueue_many: the tables of a group finish their forward passes together): blockIdx.y
// selects the table's record, which holds what the single-table launch takes as arguments.  One launch per mode and one gather for the whole group, on the
// group's stream -- 96 tables used to be 288 launches on 96 streams, each behind a barrier on the group's event.
struct BtGroupEntry {
	DevProblem P;
	const BtUnit* units;
	const BtChunk* chunks;
	uint32_t n_chunks, n_units, n_orient_max, n_cols;
	uint32_t* path2;
	uint32_t* trans2;
	uint32_t* out_score;
	uint32_t* unit_x2;
	uint32_t* guess;
	uint8_t* sel;
	uint32_t* counters;
	uint32_t* path_index;
	uint32_t* path_trans;
	SuperreadArgs super;   // (delta == nullptr: the host makes this table's superreads)
};
constexpr int BT_GROUP_MAX = 248;
struct BtGroupArgs {
	uint32_t n, pad;
	const BtGroupEntry* entry[BT_GROUP_MAX];
};
__global__ __launch_bounds__(256) void backtrace_chunks_group(BtGroupArgs args, uint32_t mode) {
	const BtGroupEntry& e = *args.entry[blockIdx.y];
	if (mode == 0 ? blockIdx.x >= e.n_orient_max * e.n_chunks : blockIdx.x != 0u) return;   // (uniform per block: the whole block leaves)
	backtrace_chunks_body(e.P, e.units, e.chunks, e.n_chunks, e.n_units, mode, e.n_orient_max, e.path2, e.trans2, e.out_score, e.unit_x2, e.guess, e.sel, e.counters, blockIdx.x);
}
__global__ __launch_bounds__(64) void backtrace_gather_group(BtGroupArgs args) {
	const BtGroupEntry& e = *args.entry[blockIdx.y];
	const uint32_t u = blockIdx.x;
	if (u >= e.n_units) return;
	const uint32_t c0 = e.units[u].c0, ncols = e.units[u].ncols;
	const uint32_t* __restrict__ src = e.path2 + (size_t)e.sel[u] * e.n_cols;
	const uint32_t* __restrict__ tsrc = e.trans2 + (size_t)e.sel[u] * e.n_cols;
	for (uint32_t c = threadIdx.x; c < ncols; c += blockDim.x) {
		e.path_index[c0 + c] = src[c0 + c];
		e.path_trans[c0 + c] = tsrc[c0 + c];
	}
}

// The superreads of the group's tables (those whose record carries the arrays), behind the gather: blockIdx.y = table.
__global__ __launch_bounds__(256) void superreads_group(BtGroupArgs args) {
	const BtGroupEntry& e = *args.entry[blockIdx.y];
	const uint32_t c = blockIdx.x * 256u + threadIdx.x;
	if (e.super.delta == nullptr || c >= e.super.n_cols) return;
	superreads_column(e.super, c);
}

// Gathers the final index path: unit u's columns from the path buffer the verification selected.
__global__ __launch_bounds__(64) void backtrace_gather(const BtUnit* __restrict__ units, uint32_t n_units, uint32_t n_cols, const uint32_t* __restrict__ path2,
                                                       const uint32_t* __restrict__ trans2, const uint8_t* __restrict__ sel,
                                                       uint32_t* __restrict__ path_index, uint32_t* __restrict__ path_trans) {
	const uint32_t u = blockIdx.x;
	if (u >= n_units) return;
	const uint32_t c0 = units[u].c0, ncols = units[u].ncols;
	const uint32_t* __restrict__ src = path2 + (size_t)sel[u] * n_cols;
	const uint32_t* __restrict__ tsrc = trans2 + (size_t)sel[u] * n_cols;
	for (uint32_t c = threadIdx.x; c < ncols; c += blockDim.x) {
		path_index[c0 + c] = src[c0 + c];
		path_trans[c0 + c] = tsrc[c0 + c];
	}
}
