/*
 * oracle/lp_dense.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Plain-C restatement of the scalar-LP half of the hot path: the 17-function
 * `lp_*` facade of the reference (bslv_lp.h:27-105, bslv_lp.c:60-324) whose
 * arithmetic lives in GLPK (third-party, un-vendored, version un-pinned:
 * reference Makefile:3 `-lglpk`, bslv_lp.c:21 `#include <glpk.h>`).  GLPK is
 * absent from this image, so the solver below restates its published
 * algorithm -- the bounded-variable two-phase primal simplex and the
 * bounded-variable dual simplex on the model  r = A x,  l <= (r,x) <= u
 * (GLPK reference manual, "glp_simplex") -- on a dense compact tableau.
 *
 * PARITY UNPINNED by the reference: the reference ships no test, golden vector
 * or expected output for this boundary.  It is pinned
 * instead against (1) the hand-derived ex01 answer and (2) scipy/HiGHS
 * objective values committed under tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything in oracle/.  The product (bensolve_amd/) never links this file.
 */
#ifndef ORACLE_LP_DENSE_H
#define ORACLE_LP_DENSE_H

#ifdef __cplusplus
extern "C" {
#endif

/* status codes mirror lp_status_type (bslv_lp.h:47) */
enum { OLP_INFEASIBLE = 0, OLP_UNBOUNDED = 1, OLP_UNEXPECTED = 2, OLP_UNDEFINED = 3, OLP_OPTIMAL = 4 };
/* methods mirror lp_method_type (bslv_lp.h:46) */
enum { OLP_PRIMAL = 0, OLP_DUAL = 1, OLP_DUALP = 2 };

typedef struct olp olp;

olp   *olp_create(int rows, int cols);
void   olp_free(olp *lp);
/* load COO triplets, 1-based indices (lp_init, bslv_lp.c:60-70) */
void   olp_load_coo(olp *lp, int nnz, const int *ridx, const int *cidx, const double *val);
/* drop `old_extra` trailing rows/cols, append new empty ones, reset to the
 * standard basis (lp_update_extra_coeffs, bslv_lp.c:73-102) */
void   olp_resize_extra(olp *lp, int drop_rows, int drop_cols, int add_rows, int add_cols);
int    olp_rows(const olp *lp);
int    olp_cols(const olp *lp);
/* replace one row (1-based row id, 1-based column ids) (lp_set_mat_row :136-139) */
void   olp_set_mat_row(olp *lp, int row, int len, const int *cidx, const double *val);
/* bound types: 'f','l','u','d','s' (bslv_lp.c:34-43) */
void   olp_set_row_bnds(olp *lp, int row, char type, double lb, double ub);
void   olp_set_col_bnds(olp *lp, int col, char type, double lb, double ub);
/* objective: col 0 = constant shift (bslv_lp.h:33) */
void   olp_set_obj(olp *lp, int col, double val);
void   olp_std_basis(olp *lp);
int    olp_solve(olp *lp, int method);          /* returns OLP_* status */
double olp_obj_val(const olp *lp);
double olp_row_prim(const olp *lp, int row);    /* 1-based */
double olp_col_prim(const olp *lp, int col);
double olp_row_dual(const olp *lp, int row);
double olp_col_dual(const olp *lp, int col);
long   olp_iterations(const olp *lp);           /* pivots + bound flips so far */
long   olp_pivots(const olp *lp);

#ifdef __cplusplus
}
#endif
#endif
