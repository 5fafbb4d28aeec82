/*
 * pedmec_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the reference's wMEC / PedMEC dynamic program
 * (whatshap/whatshap @ 2025-07-11, src/pedigreedptable.cpp and the classes it drives).  It exists
 * only so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can check the HIP
 * path; nothing under whatshap_amd/ may import, link or call it.
 *
 * It deliberately follows the reference's *control flow* (Gray-code enumeration with incremental
 * cost updates, strict-< scatter into the projection column, sqrt(n) checkpointing with
 * recomputation during the backtrace) instead of the closed forms the device kernels use, so the
 * two implementations share no tricks.  Every function cites the reference lines it restates.
 *
 * Parity pin: checked against (i) the known-answer cases of the reference's own tests
 * (tests/test_phasing.py, tests/test_pedigreephasing.py, tests/test_verification.py + tests/test.matrix;
 * fixtures in tests/golden/) and (ii) the compiled reference itself (oracle/_ref, built by
 * oracle/Makefile from the sources under /root/reference) on random tie-heavy instances
 * (tests/test_oracle_vs_reference.py).
 *
 * Input/output types are the views of include/whatshap_amd.h so the oracle, the compiled reference
 * driver (oracle/ref_driver.cpp) and the product library consume byte-identical inputs.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/whatshap_amd.h"

#define INF 0xFFFFFFFFu /* numeric_limits<unsigned int>::max() */

typedef struct entry_t { /* src/entry.h:22-24 */
	uint32_t read_id;
	uint8_t allele;
	uint32_t phred;
} entry_t;

typedef struct column_t {
	uint32_t n;
	entry_t* e;
} column_t;

typedef struct assignment_t { /* allele_assignment_t, src/pedigreecolumncostcomputer.h:27-34 */
	uint32_t assignment;
	uint32_t cost;
} assignment_t;

typedef struct costcomputer_t { /* PedigreeColumnCostComputer */
	const column_t* column;
	const uint32_t* read_marks; /* read -> individual index */
	const int* h2p;             /* [n_ind][2] haplotype -> partition for this transmission value */
	uint32_t n_partitions;
	uint32_t partitioning;
	uint32_t (*cost_partition)[2];
	assignment_t* assignments;
	uint32_t n_assignments;
} costcomputer_t;

typedef struct indexer_t { /* ColumnIndexingScheme, src/columnindexingscheme.h */
	uint32_t k;
	uint32_t* read_ids;
	uint32_t backward_width;
	int* forward_mask; /* [k]: bit index in the forward projection or -1; NULL for the last column */
} indexer_t;

typedef struct pmo_table {
	/* inputs (copied) */
	uint32_t n_reads;
	uint64_t* read_ptr;
	int32_t* var_position;
	uint8_t* var_allele;
	uint32_t* var_quality;
	uint32_t* read_sources; /* individual index per read, src/pedigreedptable.cpp:32-34 */
	uint32_t n_ind, n_triples, n_variants;
	uint32_t* individual_id;
	uint32_t (*triples)[3]; /* by index */
	uint8_t* genotype;
	double* gl;
	uint8_t* gl_present;
	int distrust;
	uint32_t* recomb;
	size_t n_recomb;
	uint32_t n_cols;
	uint32_t* positions;
	size_t* first_reads;
	/* derived */
	uint32_t T;     /* 4^triples */
	uint32_t P;     /* partitions */
	int* h2p;       /* [T][n_ind][2] */
	indexer_t* indexers;
	uint32_t** proj; /* projection_column_table[c]      : [2^f][T] or NULL */
	uint32_t** ibt;  /* index_backtrace_table[c]        */
	uint32_t** tbt;  /* transmission_backtrace_table[c] */
	uint32_t optimal_score, optimal_score_index, optimal_transmission_value, previous_transmission_value;
	uint32_t* path_index;
	uint32_t* path_trans;
	uint64_t cells; /* sum over unique columns of 2^k */
	char err[256];
	int status;
} pmo_table;

static void fail(pmo_table* t, int status, const char* msg) {
	if (t->status == 0) {
		t->status = status;
		snprintf(t->err, sizeof t->err, "%s", msg);
	}
}

/* ---------------------------------------------------------------- Read helpers (src/read.cpp) */
static uint32_t read_len(const pmo_table* t, uint32_t r) { return (uint32_t)(t->read_ptr[r + 1] - t->read_ptr[r]); }
static int32_t read_pos(const pmo_table* t, uint32_t r, uint32_t i) { return t->var_position[t->read_ptr[r] + i]; }
static int32_t first_position(const pmo_table* t, uint32_t r) { return read_pos(t, r, 0); }                 /* read.cpp:75-78 */
static int32_t last_position(const pmo_table* t, uint32_t r) { return read_pos(t, r, read_len(t, r) - 1); } /* read.cpp:81-84 */

static int cmp_u32(const void* a, const void* b) {
	uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
	return x < y ? -1 : x > y;
}

static long find_position(const pmo_table* t, int32_t pos) {
	long lo = 0, hi = (long)t->n_cols - 1;
	if (pos < 0) return -1;
	while (lo <= hi) {
		long mid = (lo + hi) / 2;
		if (t->positions[mid] == (uint32_t)pos) return mid;
		if (t->positions[mid] < (uint32_t)pos) lo = mid + 1; else hi = mid - 1;
	}
	return -1;
}

/* ColumnIterator::ColumnIterator, src/columniterator.cpp:10-59 */
static void column_iterator_init(pmo_table* t, const uint32_t* positions, size_t n_positions) {
	if (positions == NULL) {
		/* ReadSet::get_positions, src/readset.cpp:54-62: sorted set of all variant positions */
		uint64_t nnz = t->read_ptr[t->n_reads];
		uint32_t* all = (uint32_t*)malloc((nnz ? nnz : 1) * sizeof(uint32_t));
		for (uint64_t i = 0; i < nnz; ++i) all[i] = (uint32_t)t->var_position[i];
		qsort(all, nnz, sizeof(uint32_t), cmp_u32);
		uint64_t m = 0;
		for (uint64_t i = 0; i < nnz; ++i) if (i == 0 || all[i] != all[i - 1]) all[m++] = all[i];
		t->positions = all;
		t->n_cols = (uint32_t)m;
	} else {
		t->positions = (uint32_t*)malloc((n_positions ? n_positions : 1) * sizeof(uint32_t));
		memcpy(t->positions, positions, n_positions * sizeof(uint32_t));
		t->n_cols = (uint32_t)n_positions;
	}
	t->first_reads = (size_t*)malloc((t->n_cols ? t->n_cols : 1) * sizeof(size_t));
	for (uint32_t i = 0; i < t->n_cols; ++i) t->first_reads[i] = SIZE_MAX;
	int pos = 0;
	for (uint32_t i = 0; i < t->n_reads; ++i) {
		if (read_len(t, i) == 0) { fail(t, WHAMD_ERR_INVALID, "No variants present"); return; } /* read.cpp:76 */
		if (first_position(t, i) < pos) { /* :28-30 */
			fail(t, WHAMD_ERR_UNSORTED, "ColumnIterator: reads in ReadSet are not sorted.");
			return;
		}
		for (uint32_t j = 1; j < read_len(t, i); ++j) { /* Read::isSorted, read.cpp:210-218 */
			if (!(read_pos(t, i, j - 1) < read_pos(t, i, j))) {
				fail(t, WHAMD_ERR_UNSORTED, "ColumnIterator: encountered read with unsorted variants.");
				return;
			}
		}
		long fc = find_position(t, first_position(t, i)), lc = find_position(t, last_position(t, i));
		if (fc < 0 || lc < 0) { /* asserts at :36-37 */
			fail(t, WHAMD_ERR_INVALID, "read starts or ends at a position that is not in the position list");
			return;
		}
		for (long j = fc; j <= lc; ++j) if (t->first_reads[j] == SIZE_MAX) t->first_reads[j] = i; /* :40-44 */
		pos = first_position(t, i);
	}
	if (t->n_cols >= 2) { /* :49-58 */
		size_t next_index = t->first_reads[t->n_cols - 1];
		for (long i = (long)t->n_cols - 2; i >= 0; --i) {
			if (t->first_reads[i] == SIZE_MAX) t->first_reads[i] = next_index; else next_index = t->first_reads[i];
		}
	}
}

/* ColumnIterator::jump_to_column(c) followed by get_next(), src/columniterator.cpp:91-169.
 * Active reads are those with first <= pos <= last, in read-index order; a read that does not
 * cover pos contributes a BLANK entry (:127-134). */
static column_t get_column(const pmo_table* t, uint32_t c) {
	column_t col = {0, NULL};
	int pos = (int)t->positions[c];
	size_t r = t->first_reads[c];
	uint32_t cap = 8;
	col.e = (entry_t*)malloc(cap * sizeof(entry_t));
	while (r < t->n_reads) {
		if (last_position(t, (uint32_t)r) < pos) { r += 1; continue; }
		if (first_position(t, (uint32_t)r) <= pos) {
			uint32_t active_entry = 0;
			while (read_pos(t, (uint32_t)r, active_entry) < pos) active_entry += 1;
			if (col.n == cap) { cap *= 2; col.e = (entry_t*)realloc(col.e, cap * sizeof(entry_t)); }
			entry_t* e = &col.e[col.n++];
			e->read_id = (uint32_t)r; /* ids are ReadSet indices after reassignReadIds, pedigreedptable.cpp:24 */
			if (read_pos(t, (uint32_t)r, active_entry) == pos) {
				e->allele = t->var_allele[t->read_ptr[r] + active_entry];
				e->phred = t->var_quality[t->read_ptr[r] + active_entry];
			} else {
				e->allele = WHAMD_ALLELE_BLANK;
				e->phred = 0;
			}
			r += 1;
		} else {
			break;
		}
	}
	return col;
}

/* ------------------------------------------- PedigreePartitions, src/pedigreepartitions.cpp:7-42 */
static void h2p_rec(const pmo_table* t, uint32_t tv, int* map, const int* triple_indices, uint32_t i) {
	if (map[2 * i] != -1) return;
	int ti = triple_indices[i];
	uint32_t parent0 = t->triples[ti][0], parent1 = t->triples[ti][1];
	h2p_rec(t, tv, map, triple_indices, parent0);
	h2p_rec(t, tv, map, triple_indices, parent1);
	int a = map[2 * parent0 + !((tv >> (2 * ti)) & 1)];
	int b = map[2 * parent1 + !((tv >> (2 * ti + 1)) & 1)];
	map[2 * i] = a;
	map[2 * i + 1] = b;
}

static void build_partitions(pmo_table* t) {
	t->P = 2 * (t->n_ind - t->n_triples);
	t->h2p = (int*)malloc((size_t)t->T * (t->n_ind ? t->n_ind : 1) * 2 * sizeof(int));
	int* triple_indices = (int*)malloc((t->n_ind ? t->n_ind : 1) * sizeof(int));
	for (uint32_t tv = 0; tv < t->T; ++tv) {
		int* map = t->h2p + (size_t)tv * t->n_ind * 2;
		for (uint32_t i = 0; i < t->n_ind; ++i) { triple_indices[i] = -1; map[2 * i] = map[2 * i + 1] = -1; }
		for (uint32_t i = 0; i < t->n_triples; ++i) triple_indices[t->triples[i][2]] = (int)i;
		int p = 0;
		for (uint32_t i = 0; i < t->n_ind; ++i) if (triple_indices[i] == -1) { map[2 * i] = p; map[2 * i + 1] = p + 1; p += 2; }
		for (uint32_t i = 0; i < t->n_ind; ++i) h2p_rec(t, tv, map, triple_indices, i);
	}
	free(triple_indices);
}

/* ------------------------- PedigreeColumnCostComputer, src/pedigreecolumncostcomputer.cpp:14-175 */
static void cc_init(const pmo_table* t, costcomputer_t* cc, const column_t* column, uint32_t column_index, uint32_t tv) {
	cc->column = column;
	cc->read_marks = t->read_sources;
	cc->h2p = t->h2p + (size_t)tv * t->n_ind * 2;
	cc->n_partitions = t->P;
	cc->partitioning = 0;
	cc->cost_partition = (uint32_t(*)[2])calloc(t->P ? t->P : 1, sizeof(uint32_t[2]));
	cc->assignments = (assignment_t*)malloc(((size_t)1 << t->P) * sizeof(assignment_t));
	cc->n_assignments = 0;
	for (uint32_t i = 0; i < (1u << t->P); ++i) { /* :25-49 */
		int compatible = 1;
		uint32_t cost = 0;
		for (uint32_t ind = 0; ind < t->n_ind; ++ind) {
			uint32_t allele0 = (i >> cc->h2p[2 * ind]) & 1, allele1 = (i >> cc->h2p[2 * ind + 1]) & 1;
			uint32_t gt_index = allele0 + allele1; /* Genotype::get_index of a diploid bi-allelic genotype */
			size_t gi = (size_t)ind * t->n_variants + column_index;
			if (t->distrust) {
				cost = (uint32_t)((double)cost + t->gl[gi * 3 + gt_index]); /* `cost += gls->get(genotype)`, :37 */
			} else if (t->genotype[gi] != gt_index) { /* :39-43 */
				compatible = 0;
				break;
			}
		}
		if (compatible) { cc->assignments[cc->n_assignments].assignment = i; cc->assignments[cc->n_assignments].cost = cost; cc->n_assignments++; }
	}
}

static void cc_free(costcomputer_t* cc) { free(cc->cost_partition); free(cc->assignments); }

static void cc_set_partitioning(costcomputer_t* cc, uint32_t partitioning) { /* :53-76 */
	for (uint32_t p = 0; p < cc->n_partitions; ++p) cc->cost_partition[p][0] = cc->cost_partition[p][1] = 0;
	/* `partitioning = partitioning;` at :56 assigns the parameter to itself: the member keeps its value */
	for (uint32_t i = 0; i < cc->column->n; ++i) {
		const entry_t* e = &cc->column->e[i];
		int in_partition1 = (partitioning & 1u) == 0;
		uint32_t ind = cc->read_marks[e->read_id];
		int p = in_partition1 ? cc->h2p[2 * ind] : cc->h2p[2 * ind + 1];
		if (e->allele == WHAMD_ALLELE_REF) cc->cost_partition[p][1] += e->phred;
		else if (e->allele == WHAMD_ALLELE_ALT) cc->cost_partition[p][0] += e->phred;
		partitioning >>= 1;
	}
}

static void cc_update_partitioning(costcomputer_t* cc, int bit_to_flip) { /* :79-98 */
	const entry_t* e = &cc->column->e[bit_to_flip];
	cc->partitioning ^= 1u << bit_to_flip;
	int in_partition1 = (cc->partitioning & (1u << bit_to_flip)) == 0;
	uint32_t ind = cc->read_marks[e->read_id];
	int p_from = in_partition1 ? cc->h2p[2 * ind + 1] : cc->h2p[2 * ind];
	int p_to = in_partition1 ? cc->h2p[2 * ind] : cc->h2p[2 * ind + 1];
	if (e->allele == WHAMD_ALLELE_REF) { cc->cost_partition[p_from][1] -= e->phred; cc->cost_partition[p_to][1] += e->phred; }
	else if (e->allele == WHAMD_ALLELE_ALT) { cc->cost_partition[p_from][0] -= e->phred; cc->cost_partition[p_to][0] += e->phred; }
}

static uint32_t cc_get_cost(const costcomputer_t* cc) { /* :101-114 */
	uint32_t best = INF;
	for (uint32_t k = 0; k < cc->n_assignments; ++k) {
		uint32_t cost = cc->assignments[k].cost;
		for (uint32_t p = 0; p < cc->n_partitions; ++p) cost += cc->cost_partition[p][(cc->assignments[k].assignment >> p) & 1];
		if (cost < best) best = cost;
	}
	return best;
}

/* get_alleles, :117-175.  out: per individual allele0, allele1, quality. Returns 0 on Mendelian conflict. */
static int cc_get_alleles(const pmo_table* t, const costcomputer_t* cc, uint8_t* a0, uint8_t* a1, uint32_t* q) {
	uint32_t best = INF;
	uint32_t* bca = (uint32_t*)malloc((size_t)(t->n_ind ? t->n_ind : 1) * 4 * sizeof(uint32_t)); /* [ind][hap][allele] */
	for (uint32_t i = 0; i < t->n_ind * 4; ++i) bca[i] = INF;
	for (uint32_t i = 0; i < t->n_ind; ++i) { a0[i] = a1[i] = WHAMD_ALLELE_BLANK; q[i] = 0; }
	for (uint32_t k = 0; k < cc->n_assignments; ++k) {
		uint32_t cost = cc->assignments[k].cost;
		uint32_t a = cc->assignments[k].assignment;
		for (uint32_t p = 0; p < cc->n_partitions; ++p) cost += cc->cost_partition[p][(a >> p) & 1];
		int new_best = 0;
		if (cost <= best) { best = cost; new_best = 1; } /* `<=`: the last minimum wins, :131 */
		for (uint32_t ind = 0; ind < t->n_ind; ++ind) {
			uint32_t allele0 = (a >> cc->h2p[2 * ind]) & 1, allele1 = (a >> cc->h2p[2 * ind + 1]) & 1;
			if (new_best) { a0[ind] = (uint8_t)allele0; a1[ind] = (uint8_t)allele1; }
			if (cost < bca[ind * 4 + 0 + allele0]) bca[ind * 4 + 0 + allele0] = cost;
			if (cost < bca[ind * 4 + 2 + allele1]) bca[ind * 4 + 2 + allele1] = cost;
		}
	}
	if (best == INF) { free(bca); return 0; } /* :155-157 */
	for (uint32_t ind = 0; ind < t->n_ind; ++ind) { /* :160-172 */
		for (int hap = 0; hap < 2; ++hap) {
			int quality = abs((int)bca[ind * 4 + 2 * hap + 0] - (int)bca[ind * 4 + 2 * hap + 1]);
			q[ind] = (uint32_t)quality; /* overwritten by haplotype 1 */
			if (quality == 0) { if (hap == 0) a0[ind] = WHAMD_ALLELE_EQUAL_SCORES; else a1[ind] = WHAMD_ALLELE_EQUAL_SCORES; }
		}
	}
	free(bca);
	return 1;
}

/* ------------------------------------ ColumnIndexingScheme, src/columnindexingscheme.cpp:7-34,62-85 */
static void indexer_init(indexer_t* ix, const indexer_t* prev, const column_t* col) {
	ix->k = col->n;
	ix->read_ids = (uint32_t*)malloc((col->n ? col->n : 1) * sizeof(uint32_t));
	for (uint32_t i = 0; i < col->n; ++i) ix->read_ids[i] = col->e[i].read_id;
	ix->backward_width = 0;
	ix->forward_mask = NULL;
	if (prev) {
		uint32_t i = 0, j = 0;
		while (i < prev->k && j < ix->k) {
			if (prev->read_ids[i] == ix->read_ids[j]) { ix->backward_width++; i++; j++; }
			else if (prev->read_ids[i] < ix->read_ids[j]) i++; else j++;
		}
	}
}

static void indexer_set_next(indexer_t* ix, const indexer_t* next) { /* set_next_column, :62-85 */
	ix->forward_mask = (int*)malloc((ix->k ? ix->k : 1) * sizeof(int));
	for (uint32_t j = 0; j < ix->k; ++j) ix->forward_mask[j] = -1;
	uint32_t i = 0, j = 0;
	int n = 0;
	while (i < next->k && j < ix->k) {
		if (next->read_ids[i] == ix->read_ids[j]) { ix->forward_mask[j] = n++; i++; j++; }
		else if (next->read_ids[i] < ix->read_ids[j]) i++; else j++;
	}
}

static uint32_t forward_bits(const indexer_t* ix) {
	uint32_t f = 0;
	if (ix->forward_mask) for (uint32_t j = 0; j < ix->k; ++j) if (ix->forward_mask[j] >= 0) f++;
	return f;
}

static uint32_t popcount32(uint32_t x) { uint32_t c = 0; for (; x; x >>= 1) c += x & 1; return c; } /* pedigreedptable.cpp:58-64 */

static uint32_t recomb_at(const pmo_table* t, uint32_t c) {
	/* The reference reads recombcost[column_index] unchecked (:289).  Past the end we define the
	 * value as the last given entry (0 if none) -- include/whatshap_amd.h states the same rule. */
	if (c < t->n_recomb) return t->recomb[c];
	return t->n_recomb ? t->recomb[t->n_recomb - 1] : 0;
}

/* PedigreeDPTable::compute_column, src/pedigreedptable.cpp:177-335 */
static void compute_column(pmo_table* t, uint32_t c, int count_cells) {
	if (t->proj[c] != NULL) return; /* :181-185 */
	indexer_t* ix = &t->indexers[c];
	uint32_t T = t->T;
	column_t col = get_column(t, c);
	size_t column_size = (size_t)1 << ix->k;
	uint32_t* dp_column = (uint32_t*)calloc(column_size * T, sizeof(uint32_t)); /* :200 */
	const uint32_t* prev = c > 0 ? t->proj[c - 1] : NULL;
	uint32_t *cur_proj = NULL, *tbt = NULL, *ibt = NULL;
	int last = !(c + 1 < t->n_cols);
	if (!last) { /* :213-229; the reference sizes these 2^k (forward_projection_size, columnindexingscheme.cpp:47-49), only rows < 2^f are touched */
		size_t n = ((size_t)1 << forward_bits(ix)) * T;
		cur_proj = (uint32_t*)malloc(n * sizeof(uint32_t));
		tbt = (uint32_t*)malloc(n * sizeof(uint32_t));
		ibt = (uint32_t*)malloc(n * sizeof(uint32_t));
		memset(cur_proj, 0xFF, n * sizeof(uint32_t));
		memset(tbt, 0xFF, n * sizeof(uint32_t));
		memset(ibt, 0xFF, n * sizeof(uint32_t));
	}
	costcomputer_t* ccs = (costcomputer_t*)malloc(T * sizeof(costcomputer_t));
	for (uint32_t i = 0; i < T; ++i) cc_init(t, &ccs[i], &col, c, i); /* :232-236 */
	uint32_t* min_recomb_index = (uint32_t*)malloc(T * sizeof(uint32_t));
	if (count_cells) t->cells += column_size;

	/* GrayCodes (src/graycodes.cpp:9-43) driven through ColumnIndexingIterator::advance
	 * (src/columnindexingiterator.cpp:26-49): visits 0,1,3,2,6,7,5,4,... */
	int length = (int)ix->k;
	uint32_t gs = ~0u, gc = 0;
	int gi = -1, gchanged = -1;
	uint32_t forward_projection = 0;
	while (gi < length) { /* has_next(), graycodes.cpp:20-22 */
		uint32_t index = gc;
		int bit_changed = gchanged;
		gi = 0;
		while (gi < length) { /* get_next, graycodes.cpp:26-43 */
			uint32_t mask = 1u << gi;
			if (((gc & mask) ^ (gs & mask)) != 0) { gc ^= mask; gchanged = gi; break; }
			gs ^= mask;
			gi += 1;
		}
		if (bit_changed >= 0) { /* pedigreedptable.cpp:243-251 */
			if (ix->forward_mask && ix->forward_mask[bit_changed] >= 0) forward_projection ^= 1u << ix->forward_mask[bit_changed];
			for (uint32_t i = 0; i < T; ++i) cc_update_partitioning(&ccs[i], bit_changed);
		} else {
			forward_projection = 0;
			for (uint32_t i = 0; i < T; ++i) cc_set_partitioning(&ccs[i], index);
		}
		size_t bidx = 0;
		if (c > 0) bidx = index & ((1u << ix->backward_width) - 1); /* :254-257 */
		int found_valid = 0;
		for (uint32_t i = 0; i < T; ++i) { /* :264-300 */
			uint32_t current_cost = cc_get_cost(&ccs[i]);
			uint32_t min = INF;
			uint32_t min_index = 0;
			if (current_cost < INF) found_valid = 1;
			for (uint32_t j = 0; j < T; ++j) {
				uint32_t val, previous_cost = 0;
				if (c > 0) previous_cost = prev[bidx * T + j];
				if (current_cost < INF && previous_cost < INF) val = current_cost + previous_cost; else val = INF;
				if (val < INF) val += popcount32(i ^ j) * recomb_at(t, c);
				if (val < min) { min = val; min_index = j; }
			}
			dp_column[(size_t)index * T + i] = min;
			min_recomb_index[i] = min_index;
		}
		if (!found_valid) { fail(t, WHAMD_ERR_MENDELIAN_CONFLICT, "Error: Mendelian conflict"); break; } /* :301-303 */
		if (last) { /* :306-315 */
			for (uint32_t i = 0; i < T; ++i) {
				if (dp_column[(size_t)index * T + i] < t->optimal_score) {
					t->optimal_score = dp_column[(size_t)index * T + i];
					t->optimal_score_index = index;
					t->optimal_transmission_value = i;
					t->previous_transmission_value = min_recomb_index[i];
				}
			}
		} else { /* :317-325 */
			for (uint32_t i = 0; i < T; ++i) {
				if (dp_column[(size_t)index * T + i] < cur_proj[(size_t)forward_projection * T + i]) {
					cur_proj[(size_t)forward_projection * T + i] = dp_column[(size_t)index * T + i];
					ibt[(size_t)forward_projection * T + i] = index;
					tbt[(size_t)forward_projection * T + i] = min_recomb_index[i];
				}
			}
		}
	}
	if (!last) { t->proj[c] = cur_proj; t->ibt[c] = ibt; t->tbt[c] = tbt; }
	for (uint32_t i = 0; i < T; ++i) cc_free(&ccs[i]);
	free(ccs); free(min_recomb_index); free(dp_column); free(col.e);
}

static void free_column_tables(pmo_table* t, uint32_t c) {
	free(t->proj[c]); free(t->ibt[c]); free(t->tbt[c]);
	t->proj[c] = t->ibt[c] = t->tbt[c] = NULL;
}

/* PedigreeDPTable::compute_table, src/pedigreedptable.cpp:84-174 */
static void compute_table(pmo_table* t) {
	uint32_t n = t->n_cols;
	t->optimal_score = INF; /* clear_table, :67-81 */
	t->optimal_score_index = t->optimal_transmission_value = t->previous_transmission_value = 0;
	if (n == 0) { t->optimal_score = 0; return; } /* :88-92 */
	t->indexers = (indexer_t*)calloc(n, sizeof(indexer_t));
	t->proj = (uint32_t**)calloc(n, sizeof(uint32_t*));
	t->ibt = (uint32_t**)calloc(n, sizeof(uint32_t*));
	t->tbt = (uint32_t**)calloc(n, sizeof(uint32_t*));
	for (uint32_t c = 0; c < n; ++c) { /* the indexers are built ahead of the forward pass here; same content as :99-117 */
		column_t col = get_column(t, c);
		if (col.n > 31) { free(col.e); fail(t, WHAMD_ERR_UNSUPPORTED, "coverage too high for the oracle"); return; }
		indexer_init(&t->indexers[c], c ? &t->indexers[c - 1] : NULL, &col);
		if (c) indexer_set_next(&t->indexers[c - 1], &t->indexers[c]);
		free(col.e);
	}
	size_t k = (size_t)sqrt((double)n); /* :104 */
	for (uint32_t c = 0; c < n; ++c) { /* :105-135 */
		compute_column(t, c, 1);
		if (t->status) return;
		if (k > 1 && c > 0 && ((c - 1) % k) != 0) free_column_tables(t, c - 1);
	}
	/* backtrace, :137-173 */
	t->path_index = (uint32_t*)calloc(n, sizeof(uint32_t));
	t->path_trans = (uint32_t*)calloc(n, sizeof(uint32_t));
	uint32_t prev_inheritance_value = t->previous_transmission_value;
	uint32_t v_index = t->optimal_score_index, v_inh = t->optimal_transmission_value;
	t->path_index[n - 1] = v_index;
	t->path_trans[n - 1] = v_inh;
	for (size_t i = n - 1; i > 0; --i) {
		if (t->proj[i - 1] == NULL) { /* :146-153 */
			size_t j = (i - 1) / k * k;
			for (j = j + 1; j < i; ++j) { compute_column(t, (uint32_t)j, 0); if (t->status) return; }
		}
		uint32_t bt_index = v_index & ((1u << t->indexers[i].backward_width) - 1); /* :155-156 */
		v_index = t->ibt[i - 1][(size_t)bt_index * t->T + prev_inheritance_value];
		v_inh = prev_inheritance_value;
		prev_inheritance_value = t->tbt[i - 1][(size_t)bt_index * t->T + v_inh];
		t->path_index[i - 1] = v_index;
		t->path_trans[i - 1] = v_inh;
		if (i % k == 0) { /* :162-172 */
			for (size_t j = i; j < i + k && j < (size_t)n - 1; ++j) free_column_tables(t, (uint32_t)j);
		}
	}
}

/* ============================================================================ public API */

int pmo_create(const whamd_readset_view* rs, const uint32_t* recombcost, size_t n_recombcost,
               const whamd_pedigree_view* ped, int distrust_genotypes, const uint32_t* positions,
               size_t n_positions, pmo_table** out) {
	pmo_table* t = (pmo_table*)calloc(1, sizeof(pmo_table));
	*out = t;
	uint64_t nnz = rs->n_reads ? rs->read_ptr[rs->n_reads] : 0;
	t->n_reads = rs->n_reads;
	t->read_ptr = (uint64_t*)calloc(rs->n_reads + 1, sizeof(uint64_t));
	if (rs->n_reads) memcpy(t->read_ptr, rs->read_ptr, (rs->n_reads + 1) * sizeof(uint64_t));
	t->var_position = (int32_t*)malloc((nnz ? nnz : 1) * sizeof(int32_t));
	t->var_allele = (uint8_t*)malloc(nnz ? nnz : 1);
	t->var_quality = (uint32_t*)malloc((nnz ? nnz : 1) * sizeof(uint32_t));
	memcpy(t->var_position, rs->var_position, nnz * sizeof(int32_t));
	memcpy(t->var_allele, rs->var_allele, nnz);
	memcpy(t->var_quality, rs->var_quality, nnz * sizeof(uint32_t));
	for (uint64_t i = 0; i < nnz; ++i) if (t->var_allele[i] > WHAMD_ALLELE_BLANK) fail(t, WHAMD_ERR_INVALID, "read allele must be 0 (REF), 1 (ALT) or 2 (BLANK)"); /* BLANK is skipped, :69-70, 93-94 */
	t->n_ind = ped->n_individuals;
	t->n_triples = ped->n_triples;
	t->n_variants = ped->n_variants;
	t->individual_id = (uint32_t*)malloc((t->n_ind ? t->n_ind : 1) * sizeof(uint32_t));
	memcpy(t->individual_id, ped->individual_id, t->n_ind * sizeof(uint32_t));
	size_t ng = (size_t)t->n_ind * t->n_variants;
	t->genotype = (uint8_t*)malloc(ng ? ng : 1);
	if (ng) memcpy(t->genotype, ped->genotype, ng);
	t->distrust = distrust_genotypes != 0;
	if (ped->genotype_likelihoods) {
		t->gl = (double*)malloc((ng ? ng : 1) * 3 * sizeof(double));
		memcpy(t->gl, ped->genotype_likelihoods, ng * 3 * sizeof(double));
	}
	t->triples = (uint32_t(*)[3])malloc((t->n_triples ? t->n_triples : 1) * sizeof(uint32_t[3]));
	for (uint32_t i = 0; i < t->n_triples; ++i) { /* Pedigree::addRelationship -> id_to_index, src/pedigree.cpp:41-55 */
		for (int m = 0; m < 3; ++m) {
			uint32_t id = ped->triple_ids[3 * i + m];
			long idx = -1;
			for (uint32_t q = 0; q < t->n_ind; ++q) if (t->individual_id[q] == id) idx = q; /* later insertion wins, as id_to_index_map[id] = ... */
			if (idx < 0) { char m2[96]; snprintf(m2, sizeof m2, "Individual with ID %u not present in pedigree.", id); fail(t, WHAMD_ERR_INVALID, m2); idx = 0; }
			t->triples[i][m] = (uint32_t)idx;
		}
	}
	t->n_recomb = n_recombcost;
	t->recomb = (uint32_t*)malloc((n_recombcost ? n_recombcost : 1) * sizeof(uint32_t));
	memcpy(t->recomb, recombcost, n_recombcost * sizeof(uint32_t));
	t->T = 1;
	for (uint32_t i = 0; i < t->n_triples; ++i) t->T *= 4; /* std::pow(4, triple_count), :27 */
	if (t->status) return t->status;

	column_iterator_init(t, positions, n_positions); /* member initialiser, :22 */
	if (t->status) return t->status;
	t->read_sources = (uint32_t*)malloc((t->n_reads ? t->n_reads : 1) * sizeof(uint32_t));
	for (uint32_t r = 0; r < t->n_reads; ++r) { /* :32-34 */
		long idx = -1;
		for (uint32_t q = 0; q < t->n_ind; ++q) if ((int64_t)t->individual_id[q] == (int64_t)rs->read_sample_id[r]) idx = q;
		if (idx < 0) { char m2[96]; snprintf(m2, sizeof m2, "Individual with ID %u not present in pedigree.", (unsigned)rs->read_sample_id[r]); fail(t, WHAMD_ERR_INVALID, m2); return t->status; }
		t->read_sources[r] = (uint32_t)idx;
	}
	if (t->n_ind && t->n_cols > t->n_variants) { fail(t, WHAMD_ERR_INVALID, "pedigree has fewer variants than there are columns"); return t->status; }
	if (t->distrust && t->n_cols && t->n_ind && !t->gl) { fail(t, WHAMD_ERR_INVALID, "distrust_genotypes requires genotype likelihoods"); return t->status; }
	build_partitions(t);
	compute_table(t); /* :36 */
	return t->status;
}

const char* pmo_error(const pmo_table* t) { return t->err; }
uint32_t pmo_column_count(const pmo_table* t) { return t->n_cols; }
uint64_t pmo_cell_count(const pmo_table* t) { return t->cells; }
uint32_t pmo_optimal_score(const pmo_table* t) { return t->optimal_score; } /* :338-341 */
void pmo_positions(const pmo_table* t, uint32_t* out) { memcpy(out, t->positions, t->n_cols * sizeof(uint32_t)); }

void pmo_index_path(const pmo_table* t, uint32_t* index_out, uint32_t* trans_out) {
	if (t->n_cols == 0) return;
	memcpy(index_out, t->path_index, t->n_cols * sizeof(uint32_t));
	memcpy(trans_out, t->path_trans, t->n_cols * sizeof(uint32_t));
}

/* get_super_reads, src/pedigreedptable.cpp:344-388 */
int pmo_super_reads(pmo_table* t, uint8_t* allele0, uint8_t* allele1, uint32_t* quality, uint32_t* transmission, uint32_t* sample_id) {
	uint32_t n = t->n_cols;
	for (uint32_t i = 0; i < t->n_ind; ++i) sample_id[i] = t->individual_id[i];
	uint8_t* a0 = (uint8_t*)malloc(t->n_ind ? t->n_ind : 1);
	uint8_t* a1 = (uint8_t*)malloc(t->n_ind ? t->n_ind : 1);
	uint32_t* q = (uint32_t*)malloc((t->n_ind ? t->n_ind : 1) * sizeof(uint32_t));
	for (uint32_t c = 0; c < n; ++c) {
		column_t col = get_column(t, c);
		costcomputer_t cc;
		cc_init(t, &cc, &col, c, t->path_trans[c]);
		cc_set_partitioning(&cc, t->path_index[c]);
		int ok = cc_get_alleles(t, &cc, a0, a1, q);
		cc_free(&cc);
		free(col.e);
		if (!ok) { free(a0); free(a1); free(q); fail(t, WHAMD_ERR_MENDELIAN_CONFLICT, "Error: Mendelian conflict"); return t->status; }
		for (uint32_t i = 0; i < t->n_ind; ++i) {
			allele0[(size_t)i * n + c] = a0[i];
			allele1[(size_t)i * n + c] = a1[i];
			quality[(size_t)i * n + c] = q[i];
		}
		transmission[c] = t->path_trans[c];
	}
	free(a0); free(a1); free(q);
	return 0;
}

/* get_optimal_partitioning, src/pedigreedptable.cpp:391-406, then core.pyx:413-416 (true -> 0, false -> 1) */
void pmo_partitioning(const pmo_table* t, uint8_t* out) {
	for (uint32_t r = 0; r < t->n_reads; ++r) out[r] = 1;
	for (uint32_t c = 0; c < t->n_cols; ++c) {
		for (uint32_t j = 0; j < t->indexers[c].k; ++j) if ((t->path_index[c] & (1u << j)) == 0) out[t->indexers[c].read_ids[j]] = 0;
	}
}

void pmo_destroy(pmo_table* t) {
	if (!t) return;
	if (t->indexers) for (uint32_t c = 0; c < t->n_cols; ++c) { free(t->indexers[c].read_ids); free(t->indexers[c].forward_mask); }
	if (t->proj) for (uint32_t c = 0; c < t->n_cols; ++c) { free(t->proj[c]); free(t->ibt[c]); free(t->tbt[c]); }
	free(t->indexers); free(t->proj); free(t->ibt); free(t->tbt);
	free(t->read_ptr); free(t->var_position); free(t->var_allele); free(t->var_quality); free(t->read_sources);
	free(t->individual_id); free(t->triples); free(t->genotype); free(t->gl); free(t->recomb);
	free(t->positions); free(t->first_reads); free(t->h2p); free(t->path_index); free(t->path_trans);
	free(t);
}
