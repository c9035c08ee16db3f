/*
 * bslv_hip.h -- C ABI of libbslv_hip.so, the MI355X-native engine behind the two inner loops of
 * BENSOLVE's Benson outer-approximation algorithm.
 *
 * Every entry point is plain C: pointers, sizes, ints.  No torch / HIP types cross the boundary.
 * All pointers are HOST pointers unless the name ends in _dev.  Return value 0 = success,
 * non-zero = BSLV_E_* error (bslv_last_error() has the text).  The library never falls back
 * to a CPU path: if no gfx950 device is usable every constructor fails with BSLV_E_NODEVICE.
 *
 * Reference interfaces replaced (file:line relative to the reference tree):
 *   - bslv_lp.h:27-105   the 17 lp_* functions bslv_algs.o / bslv_main.o import
 *                        (one global glp_prob, bslv_lp.c:31)            -> section 1 + section 3
 *   - bslv_poly.h:90-118 the poly__* functions and the polytope / poly_args structs
 *                        (bslv_poly.h:55-82)                            -> section 2 + section 3
 *   - bslv_algs.c:958-1161 phase2_primal's inner loop (one vertex -> one LP -> one cut)
 *                        re-shaped into batch -> kernels -> gather      -> section 4
 */
#ifndef BSLV_HIP_H
#define BSLV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    BSLV_OK = 0,
    BSLV_E_NODEVICE = 1,   /* no usable HIP device / HIP call failed */
    BSLV_E_ARG = 2,        /* bad argument */
    BSLV_E_NOMEM = 3,      /* device or host allocation failed / pool exhausted */
    BSLV_E_CAPACITY = 4,   /* a fixed-capacity device array overflowed */
    BSLV_E_STATE = 5       /* call out of order */
};

/* LP status codes: same numbering as lp_status_type (bslv_lp.h:47). */
enum { BSLV_LP_INFEASIBLE = 0, BSLV_LP_UNBOUNDED = 1, BSLV_LP_UNEXPECTED = 2, BSLV_LP_UNDEFINED = 3, BSLV_LP_OPTIMAL = 4 };

const char *bslv_last_error(void);
int bslv_device_count(void);
/* the GPU the calling thread's engines live on from now (one process per GPU: LOCAL_RANK); engines are created on the
 * current HIP device of the thread that creates them */
int bslv_set_device(int device);
/* name / CU count / total global memory of the device the calling thread uses */
int bslv_device_info(char *name, int name_len, int *cus, size_t *mem_bytes);

/* ------------------------------------------------------------------------------------------
 * 1. Batched scalar-LP engine  (replaces lp_solve + getters, bslv_lp.c:219-308, for batches)
 *
 * Model (GLPK's, as bslv_lp.c uses it): auxiliary variables r = A x; every variable -- aux
 * 0..M-1, structural M..M+N-1 -- has bounds [lb,ub] (+-INFINITY allowed); minimise cost.x + c0.
 * A contiguous range of variables [var_first, var_first+var_cnt) gets PER-LP bounds: these are
 * the r rows  R_j.y - z <= R_j.v  of P2(v) (bslv_algs.c:637-649,1041-1048).
 *
 * State lives in a pool of TABLEAU SLOTS resident in HBM.  A slot holds one LP's dense compact
 * simplex tableau x_B = T x_N (with reduced-cost row and basic values), so a solved slot is a
 * warm start for any other right-hand side: every slot that reached optimality is dual feasible
 * for every v.  solve_batch copies src -> dst, installs the new bounds, and runs the bounded dual
 * simplex in lock step over the batch.
 * ------------------------------------------------------------------------------------------ */
typedef struct bslv_lpq bslv_lpq;

int  bslv_lpq_create(bslv_lpq **out, int M, int N,
                     const double *A /* M*N row-major */,
                     const double *lb, const double *ub /* M+N each */,
                     const double *cost /* N+1: cost[0] = constant shift (bslv_lp.h:33) */,
                     int var_first, int var_cnt, int pool_slots);
void bslv_lpq_destroy(bslv_lpq *h);
/* PRESOLVE at this boundary: a row of A with a single non-zero outside the per-LP range is a bound on its column (the rows `d 0 1`
 * over free columns of ex/example10.m:21-24); it is folded into the column's bounds and leaves the tableau.  Every index of this
 * interface stays an index of the model AS GIVEN (GLPK's, bslv_lp.c:60-70): get_primal of a folded row returns a_ij x_j, get_dual
 * the column's reduced cost over a_ij when the column sits on the bound that row gave it.  BSLV_NO_PRESOLVE=1 switches it off. */
int  bslv_lpq_rows_folded(const bslv_lpq *h);
/* REVISED FORM (the reference hands A to its solver as COO, bslv_lp.c:60-70 -> glp_load_matrix): for wide sparse
 * problems the engine can keep the BASIS INVERSE of every LP (M x M) instead of its tableau ((M+1) x N) and A once, as CSC and CSR,
 * for the whole pool; tableau rows and columns are sparse products, the delayed update and its pass kernel run on B^-1 (ex09 of the
 * reference's suite: 171 MB per LP instead of 1.36 GB).  Same interface, same slots and warm starts; bslv_lpq_solve_batch_obj is not
 * available in this form, and nothing refactorises B^-1: an LP whose inverse has drifted (the pivot element from its row and from its
 * column disagree) comes back BSLV_LP_UNDEFINED for the caller's retry from the standard basis (bslv_lp.c:222-227).  Chosen by itself
 * for N >= 2 M, < 2 % non-zeros and tableaux of 4 GiB and more; BSLV_LP_REV=0 / 1 forces the form.  Returns 1 in the revised form. */
int  bslv_lpq_is_revised(const bslv_lpq *h);
int  bslv_lpq_pool_slots(const bslv_lpq *h);
size_t bslv_lpq_slot_bytes(const bslv_lpq *h);
/* LAZY TABLEAUX.  The first pass of a solve writes the LP's tableau into its own slot -- the largest memory item of a batch, and
 * wasted where nobody uses the slot as a parent again (the reference keeps ONE basis and warm-starts from it, bslv_lp.c:170-173; here
 * only the LP of a vertex that yields a new cut becomes a parent).  After bslv_lpq_set_lazy(h, 1) an LP that is finished when its
 * pass would be due keeps its pending pivots; values, duals and objective are served from its vectors as always.  The caller then
 * names the slots it will start later LPs from: bslv_lpq_materialise gives those their tableau (the same pass), and
 * bslv_lpq_discard_pending drops the rest.  A slot that was not materialised must not be passed as `src`.  A solve_batch that finds
 * slots still open materialises all of them first.  bslv_lpq_lazy_stats: [0] LP passes skipped, [1] passes made on request, [2] host
 * microseconds spent in bslv_lpq_materialise (totals). */
int  bslv_lpq_set_lazy(bslv_lpq *h, int on);
int  bslv_lpq_materialise(bslv_lpq *h, int n, const int *slots);
int  bslv_lpq_discard_pending(bslv_lpq *h);
int  bslv_lpq_lazy_stats(const bslv_lpq *h, long out[3]);
/* replace the shared bounds (lp_set_rows / lp_set_cols, bslv_lp.c:112-134) */
int  bslv_lpq_set_bounds(bslv_lpq *h, const double *lb, const double *ub);
/* put the standard basis (all aux basic; glp_std_basis, bslv_lp.c:101,225) into a slot */
int  bslv_lpq_reset_slot(bslv_lpq *h, int slot);
/* Solve B LPs.  LP b starts from slot src[b], works in slot dst[b] (src==dst allowed: in place),
 * with bounds vlo/vup[b*var_cnt + j] on variable var_first+j.  status[b] gets BSLV_LP_*,
 * iters[b] the number of dual-simplex pivots.  Any of status/iters may be NULL. */
int  bslv_lpq_solve_batch(bslv_lpq *h, int B, const int *src, const int *dst,
                          const double *vlo, const double *vup, int *status, int *iters);
/* The LPs of the batch differ in their OBJECTIVE (lp_set_obj_coeffs + lp_solve, bslv_lp.c:141-151, 219-259, as phase2_dual
 * drives them, bslv_algs.c:1469-1477): cost costs[b*cost_cnt + t] on variable cost_first + t, zero elsewhere (the engine must
 * have been created with a zero cost vector).  LP b starts from the basis of slot src[b], which has to be primal feasible
 * (any solved slot is: the bounds do not change) and runs primal simplex steps.  vlo/vup as in solve_batch. */
int  bslv_lpq_solve_batch_obj(bslv_lpq *h, int B, const int *src, const int *dst, const double *vlo, const double *vup,
                              int cost_first, int cost_cnt, const double *costs, int *status, int *iters);
/* getters for solved slots: out[b*cnt + j] = value for variable first+j of slot[b]
 * (lp_primal_solution_rows/cols, lp_dual_solution_rows/cols, lp_obj_val: bslv_lp.c:261-308) */
int  bslv_lpq_get_primal(bslv_lpq *h, int B, const int *slot, int first, int cnt, double *out);
int  bslv_lpq_get_dual(bslv_lpq *h, int B, const int *slot, int first, int cnt, double *out);
int  bslv_lpq_get_obj(bslv_lpq *h, int B, const int *slot, double *out);
/* statistics of the last solve_batch: lock-step iterations, tableau-update kernel launches,
 * and the HIP-event time (ms) spent in the tableau-update kernel */
/* when on, every tableau-update launch is bracketed by HIP events on the engine's stream */
int  bslv_lpq_set_profile(bslv_lpq *h, int on);
int  bslv_lpq_last_stats(const bslv_lpq *h, int *lockstep_iters, long *pivots, double *update_ms, double *total_ms);
/* lockstep_iters = rounds (each: up to KP selections on vectors + one pass of k_flush over the tableaux with pending pivots);
 * update_ms = time in k_flush (HIP events, with set_profile).  Tableau passes of the last batch, i.e. how many (LP, round)
 * pairs k_flush read and wrote: */
long bslv_lpq_last_passes(const bslv_lpq *h);
/* k_flush launches behind those passes: one per lock-step round, plus one per bslv_lpq_materialise that had something to do
 * (what a kernel trace counts; update_ms / this = the average launch) */
long bslv_lpq_last_launches(const bslv_lpq *h);
/* extended selection of the last solve_batch (LPs with boxed variables): out[0] iterations in which boxed columns switched
 * bound (long-step ratio test), [1] cost perturbations switched on, [2] primal simplex steps, [3] perturbation removals
 * that left reduced costs of the wrong sign (what the bound switches / primal steps then repaired) */
int  bslv_lpq_last_ext_stats(const bslv_lpq *h, long out[4]);
/* of the iterations with bound switches (out[0] above): those whose switches were carried into beta by a vector update, i.e.
 * without a pass over the tableau (at most 48 switches in the iteration) */
long bslv_lpq_last_flip_updates(const bslv_lpq *h);
/* The extended selection (bound flipping where variables are boxed, cost perturbation against dual-degenerate stalling, primal
 * clean-up) is compiled in for LPs with a boxed variable and, by itself, for tableaux of 1 GiB and more (ex09 of the reference's
 * suite: the plain dual simplex stalls there until the iteration limit).  on != 0 switches it on for every later solve of this
 * engine; the callers' retry does that when an LP is still undefined after the restart from the standard basis
 * (bslv_lp.c:222-227 is the reference's two-stage retry).  Same optimal values, other pivots. */
int  bslv_lpq_set_extended(bslv_lpq *h, int on);
int  bslv_lpq_get_extended(const bslv_lpq *h);

/* ------------------------------------------------------------------------------------------
 * 2. Polyhedron engine  (replaces bslv_poly.h:90-118)
 *
 * Same object as the reference's poly_args: a primal polyhedron (points + directions) and its
 * dual (one dual vertex per halfspace); `val`/`ideal` arguments are what the reference passes in
 * poly_args.val / poly_args.ideal.  v2h selects the dualV2primalH callback (a function pointer
 * without context in the reference, bslv_poly.h:79-80):
 *   0 cone_polar (bslv_poly.c:30-39)  1 lowerV2upperH (bslv_algs.c:287-305, parameter c)
 *   2 upperV2lowerH (bslv_algs.c:307-313, parameter c)
 * rc_out follows the reference's EXIT_SUCCESS(0) / EXIT_FAILURE(1) convention.
 * ------------------------------------------------------------------------------------------ */
typedef struct bslv_poly bslv_poly;

int  bslv_poly_create(bslv_poly **out, int dim, int v2h, const double *c /* dim, may be NULL */);
                                                              /* poly__set_default_args + poly__initialise :41-102 */
void bslv_poly_destroy(bslv_poly *h);                         /* poly__kill :258 */
int  bslv_poly_dual0_apex(bslv_poly *h);                      /* cone_vertenum's tweak, bslv_algs.c:338-339 */
int  bslv_poly_add(bslv_poly *h, const double *val, int ideal, int *rc_out);       /* poly__add_vrtx :104 */
int  bslv_poly_add_cuts(bslv_poly *h, int B, const double *val /* B*dim */, const int *ideal /* may be NULL */,
                        int *rc_out /* B */);                 /* batched poly__add_vrtx */
/* add_cuts strategy: 0 = one cut at a time (slot numbering identical to the sequential definition),
 * 1 = rounds of independent cuts applied in one pass each (default; same sets, different slot numbers) */
int  bslv_poly_set_batch_mode(bslv_poly *h, int mode);
long bslv_poly_rounds_run(const bslv_poly *h);
/* which paths of the single-cut pipeline ran (tests, tuning): out[0] chunks in hot mode, [1] cuts whose round B was
 * queued speculatively, [2] of those declined by the device and rerun, [3] prunes redone by the multi-kernel path,
 * [4] cuts applied one at a time, [5] multi-kernel prunes that confirmed their edges through facet-major member lists */
int  bslv_poly_path_stats(const bslv_poly *h, long out[6]);
/* multi-GPU (4a): adjacency prunes whose pair space was dealt to the ranks (facets of at least 32768 elements; bslv_poly_debug_set
 * key 10 changes the threshold): every rank tests a contiguous share of the rows, the adjacent pairs found are all-gathered and
 * appended in rank order, which is the order a single GPU writes them in */
long bslv_poly_sharded_prunes(const bslv_poly *h);
/* one number per cut of the NEXT bslv_poly_add_cuts call (e.g. the depth z of the cut); with BSLV_R2_ORDER=1 / 2 the rounds of independent
 * cuts rank the cuts of a chunk by it, ascending / descending, instead of by a pseudo-random shuffle (an experiment: see DESIGN.md 4e) */
int  bslv_poly_set_cut_priorities(bslv_poly *h, int n, const double *prio);
/* the projection sub-band of poly__cut (bslv_poly.c:666-674): with on = 1 an element that lies between 1e-2 POLY_EPS and POLY_EPS above a cut
 * that removes something is moved onto the hyperplane before it is treated as lying on it, exactly as the reference does, and cuts are applied
 * one at a time in the order handed in (the rounds of independent cuts classify ahead and stay off).  Default 0: such an element keeps its
 * coordinates (same index sets, coordinates within 1e-9: tests/test_oracle_poly.py::test_snap_band_*).  Also BSLV_POLY_SNAP=1. */
int  bslv_poly_set_snap(bslv_poly *h, int on);
int  bslv_poly_snapped(bslv_poly *h, long *moved);        /* elements moved so far */
/* capacity ahead of need (elements = vertices + directions, edges, 32-bit words of incidence lists; 0 = leave alone): the arrays otherwise
 * double when they fill up, a hipMalloc + copy + hipFree in the middle of a batch of cuts.  The reference grows its lists the same way
 * in blocks of VRTXBLCK / LSTBLCK (bslv_poly.c:415-440 `add_vrtx`, :452 list blocks).  BSLV_E_ARG while a chunk of cuts is open. */
int  bslv_poly_reserve(bslv_poly *h, long elements, long edges, long pool_words);
int  bslv_poly_largest_facet(const bslv_poly *h);     /* members of the largest new facet that went through the multi-kernel prune */
long bslv_poly_noflag_prunes(const bslv_poly *h);      /* large-facet prunes that kept no flag byte per pair (k_pair_retest_emit) */
/* cuts that were still untouched when a chunk's rounds ended on "no cut alive" and were handed to the one-cut pipeline instead
 * (0 in every run but one of round 2's test runs; kept as a counter so that it cannot hide) */
long bslv_poly_rounds2_late_left(const bslv_poly *h);
/* reads of a round's mailbox (mapped pinned memory, four cache lines) whose sequence number had arrived before the rest of the
 * state: detected by the checksum the state carries and repeated */
long bslv_poly_rounds2_torn_reads(const bslv_poly *h);
/* test hook, same switches as the BSLV_* environment variables but at run time: key 0 dynamic LDS bytes of the one-workgroup
 * prune (64 forces the multi-kernel prune), 1 speculative launch on/off, 2 hot mode on/off, 3 CROSS_UB, 4 size of a new facet
 * from which the multi-kernel prune builds facet-major member lists (default 4096), 5 member lists on/off, 6 device-selected
 * rounds of independent cuts inside a hot chunk on/off, 7 cuts per chunk (32..4096, default 512) */
/* bslv_poly_add_cuts may hand cuts BACK (rc 2: not applied, dual slot left unused) when the rounds of a chunk get thinner than
 * min_cuts cuts -- the tail of a chunk is its cliques, one cut each per pass over the polyhedron; the caller hands the same
 * halfspaces in again with its next batch (phase2_primal has no counterpart: bslv_algs.c:1041-1080 applies one cut per LP).
 * 0 (default): every cut handed in is applied or found redundant, as poly__add_vrtx does (bslv_poly.c:104-151). */
int  bslv_poly_set_defer(bslv_poly *h, int min_cuts);
int  bslv_poly_debug_set(bslv_poly *h, int key, long value);
/* rounds of independent cuts chosen and applied on the device inside a hot chunk (debug_set key 6 switches them off, key 7 sets
 * the number of cuts classified and applied together): out[0] rounds, [1] cuts applied in them, [2] chunks, [3] prunes that took
 * the multi-kernel path, [4] rounds taken back for want of capacity */
int  bslv_poly_rounds2_stats(const bslv_poly *h, long out[5]);
int  bslv_poly_init(bslv_poly *h, int *rc_out);               /* poly__intl_apprx :153 */
int  bslv_poly_next(bslv_poly *h, double *val, int *ideal, int *idx, int *rc_out); /* poly__get_vrtx :210 */
int  bslv_poly_unprocessed(bslv_poly *h, int max_out, int *idx, double *val, int *ideal, int *count);
int  bslv_poly_unprocessed2(bslv_poly *h, int max_out, int from_end, int *idx, double *val, int *ideal, int *parent, int *count);
/* families: counts[k] = unprocessed elements whose parent (the newest facet through them) is dual slot first_facet + k; the
 * unprocessed children of a list of facets (at most max_out, the newest slots, ascending; parent[] = dual slots) */
int  bslv_poly_children_hist(bslv_poly *h, int first_facet, int n, int *counts, int *total, int *older);
int  bslv_poly_children_of(bslv_poly *h, int nfacets, const int *facets, int max_out, int *idx, double *val, int *ideal, int *parent, int *n_out);
int  bslv_poly_mark(bslv_poly *h, int n, const int *idx);     /* ST_BT(primal.sltn, idx) */
int  bslv_poly_dual_adjacency(bslv_poly *h);                  /* poly__update_adjacence(&dual) :992 */
/* batched incidence kernel on the current elements: hps = B x (dim+1) halfspaces (normal, alpha);
 * words_out: ceil(B/32) x nprimal 64-bit words, 2 bits per class (0 dead 1 MINUS 2 ZERO 3 PLUS) */
int  bslv_poly_classify_batch(bslv_poly *h, int B, const double *hps, unsigned long long *words_out,
                              int *anyminus_out, int repeats, float *ms_out);
/* TEST: the incidence kernel as the chunked cut application launches it: tc_out[i] = halfspaces element i is not strictly
 * inside, t1_out[i] = the first of them or -1 (nv ints each); words_out as bslv_poly_classify_batch (may be NULL) */
int  bslv_poly_classify_batch_touch(bslv_poly *h, int B, const double *hps, unsigned long long *words_out, int *tc_out, int *t1_out);
/* TEST: the incidence kernel has a variant that runs its dot products on v_mfma_f64_16x16x4 (from 16 halfspaces on; selected by
 * bslv_poly_debug_set key 9 = 1 or BSLV_K1_MFMA=1; the scalar kernel is the default because it is faster on gfx950):
 * ntiles random tiles, chained MFMAs against the scalar fma chain, *mismatches = results that differ in any bit (0: same order
 * of accumulation, one rounding per step) */
int  bslv_k1_mfma_selftest(int dim, int ntiles, unsigned long long seed, long *mismatches);
/* MEASUREMENT ONLY: replace the polyhedron by nv synthetic live points (for timing the incidence kernel) */
int  bslv_poly_bench_fill(bslv_poly *h, int nv, unsigned long long seed);
/* counts and slot-indexed dumps (poly__vrtx2file / adj2file / inc2file write these, :341-414) */
int  bslv_poly_dim(const bslv_poly *h);
int  bslv_poly_nprimal(const bslv_poly *h);
int  bslv_poly_ndual(const bslv_poly *h);
long bslv_poly_nedges(const bslv_poly *h);
long bslv_poly_ninc(bslv_poly *h);
long bslv_poly_ndual_edges(const bslv_poly *h);
long bslv_poly_pair_tests(const bslv_poly *h);
long bslv_poly_new_vertices(const bslv_poly *h);
int  bslv_poly_get_primal(bslv_poly *h, unsigned char *used, unsigned char *ideal, unsigned char *sltn, double *coords);
int  bslv_poly_get_dual(bslv_poly *h, unsigned char *used, unsigned char *ideal, double *coords);
/* coordinates of the primal slots first .. first+count-1 only (count x dim, row-major): the coordinates of a slot never change, so
 * a caller that mirrors `polytope` (poly_compat.hip, bslv_poly.h:55-69) fetches only what a cut added */
int  bslv_poly_get_primal_range(bslv_poly *h, int first, int count, double *coords);
int  bslv_poly_get_edges(bslv_poly *h, int *ab);
int  bslv_poly_get_inc(bslv_poly *h, int *pairs);
int  bslv_poly_get_dual_edges(bslv_poly *h, int *ab);
/* host-only self-test of the two mailbox readers of the polyhedron engine (no GPU needed): a writer thread publishes `rounds`
 * messages with the sequence number FIRST and the content afterwards; 0 = every message was taken with its own content
 * (*torn_out: reads that had to be repeated), negative = a reader accepted foreign content.  which: 0 Mail, 1 round state */
int  bslv_selftest_mailbox(int which, int rounds, long *torn_out);

/* ------------------------------------------------------------------------------------------
 * 4. Batched Benson phase-2 driver  (replaces phase2_primal's loop, bslv_algs.c:958-1082,
 *    and init_P2, :574-664).  Problem in the reference's normal form "min, c_q > 0"
 *    (sol_init, bslv_vlp.c:845-861); A is m x n, P is q x n, dense row-major; bound types
 *    'f','l','u','d','s' as in the .vlp format; R is q x r with generators as columns
 *    (bslv_algs.c:599) -- R = Z = I for the default cone in the bounded case (-b, :943-956).
 *    One outer iteration = collect -> solve_local -> (all-gather records) -> apply.
 * ------------------------------------------------------------------------------------------ */
typedef struct bslv_benson bslv_benson;

int  bslv_benson_create(bslv_benson **out, int m, int n, int q, const double *A, const double *P,
                        const char *rtype, const double *rlb, const double *rub,
                        const char *ctype, const double *clb, const double *cub,
                        const double *R, int r, const double *c, double eps, int pool_slots);
/* the same with the homogeneous problem of phases 0 and 1 (init_P2(..., HOMOGENEOUS), bslv_algs.c:574-664): hom != 0 zeroes
 * every bound of the VLP ('d' becomes fixed; lp_set_rows_hom / lp_set_cols_hom, bslv_lp.c:118-134), R holds the generators Z
 * of the dual ordering cone and the last row reads eta.y <= 1 (eta NULL = 0).  The cut of a vertex is then
 * y* = (w + alpha eta, alpha), alpha = dual of that row (phase1_primal, bslv_algs.c:876-883). */
int  bslv_benson_create_ex(bslv_benson **out, int m, int n, int q, const double *A, const double *P,
                           const char *rtype, const double *rlb, const double *rub,
                           const char *ctype, const double *clb, const double *cub,
                           const double *R, int r, const double *c, const double *eta, int hom, int flags, double eps, int pool_slots);
/* flags of bslv_benson_create_ex.  PREIMAGES = option -s (opt->solution == PRE_IMG_ON): the engine keeps x of every confirmed
 * vertex and (u, w) of every cut (bslv_algs.c:1064-1079); every row of A stays in the LP (no presolve) */
enum { BSLV_BENSON_PREIMAGES = 2 };
/* 0 + data, or 1 when nothing is stored for that element */
int  bslv_benson_preimage_p(const bslv_benson *h, int element, double *x /* n */);
int  bslv_benson_preimage_d(const bslv_benson *h, int facet, double *uw /* m + q */);
int  bslv_benson_set_preimage_p(bslv_benson *h, int element, const double *x /* n */);
void bslv_benson_destroy(bslv_benson *h);
int  bslv_benson_start(bslv_benson *h, int *vlp_status /* 0 ok, 1 infeasible, 2 unbounded */);
int  bslv_benson_collect(bslv_benson *h, int max_batch, int rank, int world, int *n_local, int *n_total);
int  bslv_benson_record_len(const bslv_benson *h);             /* q + 5 doubles */
int  bslv_benson_solve_local(bslv_benson *h, double *records, int *pivots_out, int *lockstep_out);
int  bslv_benson_apply(bslv_benson *h, int nrec, const double *records, long *stats /* 5, may be NULL */);
int  bslv_benson_step(bslv_benson *h, int max_batch, long *stats /* 8 */, double *ms /* 3 */);
/* two batch contexts (ctx 0/1): the LPs of batch k (solve_local_ctx, may run on a second host thread) overlap with
 * the cut application of batch k-1 (apply_ctx).  set_pipelined(1): batch members are marked when collected. */
int  bslv_benson_set_pipelined(bslv_benson *h, int on);
int  bslv_benson_collect_ctx(bslv_benson *h, int ctx, int max_batch, int rank, int world, int *n_local, int *n_total);
int  bslv_benson_solve_local_ctx(bslv_benson *h, int ctx, double *records, int *pivots_out, int *lockstep_out);
int  bslv_benson_apply_ctx(bslv_benson *h, int ctx, int nrec, const double *records, long *stats);
int  bslv_benson_unprocessed_left(const bslv_benson *h);
/* batch selection: 1 = newest vertices first (default), 2 = spread evenly over the unprocessed queue, 3 = newest first but at
 * most `cap` children of one cut per batch, chosen from a window of `window` batches (set_sibling_rule; default 1, 8): the
 * children of one cut mostly see the same facet of the upper image, so their LPs return the same cut -- the reference's
 * sequential loop never solves them, because the first copy of the cut removes the siblings (bslv_algs.c:1030-1080) */
int  bslv_benson_set_policy(bslv_benson *h, int policy);
int  bslv_benson_set_sibling_rule(bslv_benson *h, int cap, int window);
/* policy 4: K depth-first fronts (a vertex belongs to the front of the cut that created it, a cut to the front of the vertex
 * whose LP returned it; newest first inside a front, at most sib_cap children of one cut per front and batch) */
int  bslv_benson_set_fronts(bslv_benson *h, int nfronts, int sib_cap);
/* policy 5: the children of the newest cuts first (by the dual slot of the cut, not by the element's own slot);
 * policy 6: whole families -- all unprocessed children of a cut -- of parents chosen among the cuts of the last `batches` outer
 * iterations: mode 0 newest cuts first, 1 pseudo-random, 2 far apart (farthest-point sampling on the cuts' normals) */
int  bslv_benson_set_families(bslv_benson *h, int mode, int batches);
/* Thin rounds at the end of a chunk of cuts (its cliques, one cut each per pass over the polyhedron) may hand their cuts back
 * (bslv_poly_set_defer(min_cuts)); bslv_benson_apply keeps them and hands them in again in front of the next batch's cuts,
 * bslv_benson_collect applies whatever is waiting before it reports "nothing left".  0 = off (every cut of a batch is applied
 * in its own outer iteration, as bslv_algs.c:1041-1080 does).  stats: [0] handed back so far, [1] waiting, [2] flushes by
 * collect, [3] batches that one family would have filled (taken newest first). */
int  bslv_benson_set_defer(bslv_benson *h, int min_cuts);
int  bslv_benson_defer_stats(const bslv_benson *h, long out[4]);
/* tuning hooks (no counterpart in the reference): the caller chooses the batch itself -- elements with their coordinates and
 * parent facets as bslv_poly_unprocessed2 returns them -- and reads, per LP of this rank's last solve_local, the warm-start slot
 * (0 = root tableau), the pivots and the generation of the new slot; returns the number of entries written */
int  bslv_benson_collect_given(bslv_benson *h, int n, const int *idx, const double *val, const int *parent, int rank, int world, int *n_local, int *n_total);
int  bslv_benson_last_local(bslv_benson *h, int max_out, int *src, int *pivots, int *gen);
/* tableau pool: out[0] free slots, [1] resident warm-start sources, [2] held by a batch in flight, [3] pool size */
int  bslv_benson_pool_stats(bslv_benson *h, long out[4]);
int  bslv_benson_totals(const bslv_benson *h, long *lps, long *cuts, long *pivots);
/* warm starts: LPs whose parent's tableau was not resident on this rank (evicted, or solved on another rank) and that started
 * from the root tableau / from the resident tableau whose own vertex is nearest */
int  bslv_benson_start_stats(const bslv_benson *h, long *root_starts, long *nearest_starts);
/* size of the P2 model: M x N of init_P2 (bslv_algs.c:574-664), the model every index at the LP boundary refers to;
 * rows_folded of its rows (rows of A with a single non-zero: the hypercube rows of S-degenerate, ex/example10.m:21-24) were
 * turned into column bounds by the LP layer's presolve (bslv_lpq_create), so the tableau has M - rows_folded rows */
int  bslv_benson_lp_dims(const bslv_benson *h, int *M, int *N, int *rows_folded);
bslv_poly *bslv_benson_poly(bslv_benson *h);
bslv_lpq  *bslv_benson_lp(bslv_benson *h);

/* ------------------------------------------------------------------------------------------
 * 4a. Multi-GPU: one process per GPU, one exchange step per outer iteration, in the library.
 *     bslv_set_device(LOCAL_RANK), then either bslv_dist_init with the 128-byte ncclUniqueId of rank 0 (RCCL over xGMI; the
 *     host side distributes the id: bensolve_hip over TCP to MASTER_ADDR:MASTER_PORT, bench.py with one broadcast), or
 *     bslv_dist_init_callback with an all-gather of the caller (tests on a one-GPU box, where RCCL refuses two ranks on one
 *     device).  From then on bslv_benson_step -- and with it bslv_vlp_solve_primal and the command-line driver -- runs
 *     bslv_benson_step_dist: the vertex batch is dealt to the ranks (bslv_benson_collect), every rank solves its shard, ONE
 *     all-gather of fixed-size record blocks, every rank applies all records in ascending source slot, so the replicas of
 *     the polyhedron stay bit-identical.  The reference has no counterpart (single process, bslv_algs.c:1030-1080).
 * ------------------------------------------------------------------------------------------ */
typedef int (*bslv_allgather_fn)(const double *send, double *recv, int count_per_rank, void *ctx);   /* 0 = ok */
int  bslv_dist_unique_id(unsigned char *out, int len /* >= 128 */);          /* rank 0: ncclGetUniqueId */
int  bslv_dist_init(int rank, int world, const unsigned char *id, int len);  /* ncclCommInitRank on the current device */
int  bslv_dist_init_callback(int rank, int world, bslv_allgather_fn fn, void *ctx);
void bslv_dist_finalize(void);
int  bslv_dist_rank(void);
int  bslv_dist_world(void);
int  bslv_dist_allgather(const double *send, double *recv, int count_per_rank);   /* host buffers; recv holds world * count */
int  bslv_dist_stats(long *gathers, double *ms);
int  bslv_benson_step_dist(bslv_benson *h, int max_batch_global, long *stats /* 8 */, double *ms /* 3 */);
/* host wall clock (ms) of the phases of the last bslv_benson_step_dist on this rank: collect, this rank's LPs, all-gather of the
 * records (includes the wait for the slowest rank), application of all ranks' cuts (replicated on every rank) */
int  bslv_dist_last_phases(double out[4]);

/* ------------------------------------------------------------------------------------------
 * 4b. The callers around phase 2: ordering cone data (sol_init, bslv_vlp.c:599-864), cone_vertenum
 *     (bslv_algs.c:331-407), phase 0 (bslv_algs.c:673-800), phase 1 of the primal algorithm (bslv_algs.c:811-933) and the
 *     sequence of bslv_main.c:236-345.  Matrices of generators are q x k row-major with the generators as columns
 *     (M[j*k + i] = component j of generator i), as sol->Z / sol->R / vlp->gen are.
 * ------------------------------------------------------------------------------------------ */
typedef struct bslv_vlp_info {
    int q, o, p, r, h;            /* soltype: generators of C (o), of C* (p), of the dual recession cone (r), of the recession cone (h) */
    int c_dir;                    /* +1 c_q > 0, -1 c_q < 0 (bslv_vlp.c:690-776) */
    int negate_primal;            /* poly_trans_primal (bslv_algs.c:221-229): the result writers negate y ... */
    int negate_dual_last;         /* ... and y*_q */
    long lps, steps;              /* LPs solved in all phases, outer iterations */
    double *c, *eta, *R, *H, *Y, *Z;   /* malloc'ed: c as written to _c.sol (before the sign change of :844-853); free with bslv_vlp_info_free */
    char message[160];            /* the reference's message when the status is not "optimal" */
} bslv_vlp_info;
/* flags: PHASE1_DUAL = "-A dual" (opt->alg_phase1, bslv_main.c:283-296: phase1_dual, bslv_algs.c:1248-1371, instead of
 * phase1_primal); PREIMAGES = "-s" (opt->solution == PRE_IMG_ON; bslv_vlp_solve_primal only): the phase-2 engine keeps the
 * pre-images (BSLV_BENSON_PREIMAGES) and the directions of the upper image get theirs (bslv_algs.c:1083-1112) */
enum { BSLV_VLP_PHASE1_DUAL = 1, BSLV_VLP_PREIMAGES = 2 };
/* cone_kind 0 default (R^q_+), 1 `gen` generates C, 2 `gen` generates C* (vlp->cone_gen); c_in: q or NULL (vlp->c).
 * vlp_status (sol->status, bslv_main.h:103): 1 infeasible, 2 unbounded, 3 no vertex, 4 optimal, 5 input error.
 * With status 4 *engine_out is the finished phase-2 engine (bslv_benson_poly(engine) is the result; destroy it). */
int  bslv_vlp_solve_primal(int m, int n, int q, const double *A, const double *P,
                           const char *rtype, const double *rlb, const double *rub,
                           const char *ctype, const double *clb, const double *cub,
                           int optdir, int cone_kind, const double *gen, int n_gen, const double *c_in,
                           int bounded, int flags, double eps_phase0, double eps_phase1, double eps_benson_phase1, double eps_benson_phase2,
                           int batch, bslv_benson **engine_out, int *vlp_status, bslv_vlp_info *info /* may be NULL */);
/* the same with the DUAL algorithm in phase 2 ("-a dual": phase2_dual, bslv_algs.c:1381-1592, P1(w) LPs that differ in the
 * objective; phases 0 and 1 stay primal).  With status 4 *lower_image_out is a polyhedron whose primal side is the LOWER image
 * and whose dual side is the upper image: write it with bslv_sol_write3(..., swap = 1, ...), destroy with bslv_poly_destroy. */
int  bslv_vlp_solve_dual2(int m, int n, int q, const double *A, const double *P,
                          const char *rtype, const double *rlb, const double *rub,
                          const char *ctype, const double *clb, const double *cub,
                          int optdir, int cone_kind, const double *gen, int n_gen, const double *c_in,
                          int bounded, int flags, double eps_phase0, double eps_phase1, double eps_benson_phase1, double eps_benson_phase2,
                          int batch, bslv_poly **lower_image_out, int *vlp_status, bslv_vlp_info *info /* may be NULL */);
/* option -s with the dual algorithm (flags & BSLV_VLP_PREIMAGES in bslv_vlp_solve_dual2; phase2_dual with PRE_IMG_ON,
 * bslv_algs.c:1388-1389, 1431-1432, 1484-1497, 1508-1546): x (n values) of an element of the UPPER image = dual slot `facet` of the
 * returned polyhedron, (u, w) (m + q values) of a vertex of the LOWER image = its primal element.  0 + data, 1: nothing stored.
 * bslv_dual_preimages_free before bslv_poly_destroy. */
int  bslv_dual_preimage_x(const bslv_poly *lower_image, int facet, double *x);
int  bslv_dual_preimage_uw(const bslv_poly *lower_image, int element, double *uw);
void bslv_dual_preimages_free(const bslv_poly *lower_image);
void bslv_vlp_info_free(bslv_vlp_info *info);
/* cone_vertenum: prim = the non-redundant generators among gen (dim x n_prim), dual = generators of the dual cone
 * (dim x n_dual); malloc'ed, free with bslv_free.  rc_out 1: the cone has no interior (poly__intl_apprx failed). */
int  bslv_cone_vertenum(const double *gen, int n_in, int dim, double **prim, int *n_prim, double **dual, int *n_dual, int *rc_out);
void bslv_free(void *p);

/* ------------------------------------------------------------------------------------------
 * 5. Host side that stays C: the .vlp reader (vlp_init, bslv_vlp.c:275-588; kept file-format
 *    contract) and the result-file writers (poly_output, bslv_algs.c:50-144; formats of
 *    bslv_poly.c:341-414).  Struct bslv_vlp: bensolve_amd/csrc/host/bslv_host.h.
 * ------------------------------------------------------------------------------------------ */
struct bslv_vlp;
int  bslv_vlp_read(const char *path, struct bslv_vlp **out, int *err_line);
void bslv_vlp_free(struct bslv_vlp *v);
const char *bslv_vlp_message(const struct bslv_vlp *v);
int  bslv_sol_write(bslv_poly *poly, const char *base, const char *suffix, int optdir, long *counts /* 4, may be NULL */);
/* the same with the two sign changes of poly_trans_primal (bslv_algs.c:221-229) spelled out (bslv_vlp_info) */
int  bslv_sol_write2(bslv_poly *poly, const char *base, const char *suffix, int negate_primal, int negate_dual_last, long *counts);
/* swap != 0: the engine's primal side is the LOWER image (dual algorithm; poly_output(..., SWAP, ...), bslv_algs.c:1566-1573) */
/* <base>_pre_img_p<suffix> / _pre_img_d<suffix> (poly__primg2file, bslv_poly.c:362-380; poly_output :120-139): one row per
 * element in the order of the _img_ files: x (n values) of the upper image's elements, (u, w) (m + q values) of the lower
 * image's vertices, zeros where nothing is stored (directions of the lower image, :1114-1121).  u is multiplied by optdir
 * and w by c_dir as the reference stores them (:1068-1070). */
int  bslv_sol_write_preimages(bslv_benson *eng, const char *base, const char *suffix, int m, int n, int optdir, int c_dir);
int  bslv_sol_write3(bslv_poly *poly, const char *base, const char *suffix, int swap, int negate_upper, int negate_lower_last, long *counts);
/* the pre-image files of the dual algorithm: rows in the order of the _img_ files written by bslv_sol_write3(..., swap = 1, ...) */
int  bslv_sol_write_preimages_dual(bslv_poly *lower_image, const char *base, const char *suffix, int m, int n, int optdir, int c_dir);

#ifdef __cplusplus
}
#endif
#endif
