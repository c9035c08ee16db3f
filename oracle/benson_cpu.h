/*
 * oracle/benson_cpu.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 * Sequential restatement of the caller of the hot path: phase2_primal's loop
 * (bslv_algs.c:958-1080: one vertex -> one warm-started LP -> at most one cut), driving the
 * oracle LP (lp_dense.c) and the oracle polyhedron (poly_dd.c).  Used as the end-to-end checker
 * for the batched HIP driver and, on a bounded sample, as bench.py's cpu_baseline ("port").
 */
#ifndef ORACLE_BENSON_CPU_H
#define ORACLE_BENSON_CPU_H
#include "poly_dd.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct {
    long lps;          /* optimal LP solves (lp_num, bslv_lp.c:257) */
    long cuts;         /* poly__add_vrtx calls that cut */
    long pivots;       /* simplex pivots */
    long new_vertices; /* primal slots created by cuts */
    double secs_total, secs_lp, secs_poly;
    int status;        /* 0 ok, 1 infeasible, 2 unbounded, 3 stopped at max_lps */
    /* the same counters at the moment `warm_lps` LPs had been solved (obenson_phase2_primal_ex); zero when not reached */
    long warm_lps, warm_cuts, warm_pivots, warm_new_vertices;
    double warm_secs;
} obenson_stats;

/* Problem: min P x  s.t. row/col bounds (types 'f','l','u','d','s'), ordering-cone data R (q x r,
 * generators as columns, bslv_algs.c:599) and duality vector c (q).  A is m x n, P is q x n dense
 * row-major.  Stops after max_lps LPs when max_lps > 0.  *poly_out receives the polyhedron
 * (caller frees with opoly_free). */
int obenson_phase2_primal(int m, int n, int q, const double *A, const double *P,
                          const char *rtype, const double *rlb, const double *rub,
                          const char *ctype, const double *clb, const double *cub,
                          const double *R, int r, const double *c, double eps, long max_lps,
                          opoly **poly_out, obenson_stats *st);
/* The same with two measurement knobs (bench.py's cpu_baseline): order 0 = the reference's choice of the next vertex (lowest
 * live slot without the sltn mark, poly__get_vrtx bslv_poly.c:210-226), 1 = the NEWEST such slot (the order the batched HIP
 * driver works in; consecutive vertices are neighbours, so the warm-started dual simplex needs few pivots); warm_lps > 0:
 * the counters are snapshot into st->warm_* once that many LPs are done, so that a caller can rate the LPs after a warm-up. */
int obenson_phase2_primal_ex(int m, int n, int q, const double *A, const double *P,
                             const char *rtype, const double *rlb, const double *rub,
                             const char *ctype, const double *clb, const double *cub,
                             const double *R, int r, const double *c, double eps, long max_lps,
                             int order, long warm_lps, opoly **poly_out, obenson_stats *st);
#ifdef __cplusplus
}
#endif
#endif
