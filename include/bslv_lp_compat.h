/*
 * bslv_lp_compat.h -- the reference's scalar-LP boundary, symbol for symbol.
 *
 * libbslv_hip.so exports the 17 lp_* functions that bslv_algs.o / bslv_main.o import from
 * bslv_lp.c (declared in the reference's bslv_lp.h:27-105; `nm -u` list in SURVEY.md section 8b) with
 * IDENTICAL names, argument types and meaning, so a reference-shaped host driver links against the
 * HIP engine unchanged (replace bslv_lp.o and -lglpk by -lbslv_hip).  The container types below have
 * the layout of bslv_lists.h:26-48 and the enums the order of bslv_lp.h:46-48 / bslv_main.h:99-101.
 *
 * Behaviour kept: one global problem (handle `i` is always 0, bslv_lp.c:31); 1-based indices; types
 * 'f','l','u','d','s' (bslv_lp.c:34-43); lp_update_extra_coeffs drops the previous extra rows/cols,
 * appends new empty ones and resets to the standard basis (:73-102); consecutive lp_solve calls warm-
 * start from the previous basis; lp_solve retries once from the standard basis (:222-227); getters
 * exit(1) on index overflow (:261-304); lp_get_num counts optimal solves (:257).
 * Difference: the engine is a dual simplex with an artificial-bounds start; the method chosen through
 * lp_set_options is accepted and ignored (any method returns the same optimal value).
 */
#ifndef BSLV_LP_COMPAT_H
#define BSLV_LP_COMPAT_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef int lp_idx;                                                     /* bslv_main.h:23 */
typedef struct { lp_idx size; lp_idx *idx; double *data; } list1d;      /* bslv_lists.h:26-31 */
typedef struct { size_t size; lp_idx *idx1; lp_idx *idx2; double *data; } list2d;            /* :33-39 */
typedef struct { lp_idx size; lp_idx *idx; double *lb; double *ub; char *type; } boundlist;  /* :41-48 */
typedef enum { PRIMAL_SIMPLEX, DUAL_SIMPLEX, DUAL_PRIMAL_SIMPLEX, LP_METHOD_AUTO } lp_method_type;                        /* bslv_lp.h:46 */
typedef enum { LP_INFEASIBLE, LP_UNBOUNDED, LP_UNEXPECTED_STATUS, LP_UNDEFINED_STATUS, LP_OPTIMAL } lp_status_type;      /* :47 */
typedef enum { PHASE0, PHASE1_PRIMAL, PHASE1_DUAL, PHASE2_PRIMAL, PHASE2_DUAL } phase_type;                             /* bslv_main.h:101 */
struct lp_opt { lp_method_type method_phase0, method_phase1, method_phase2; int message_level; };                     /* bslv_lp.h:50-53 */

void lp_init(int rows, int cols, int nnz, lp_idx *row_idx, lp_idx *col_idx, double *data);    /* bslv_lp.c:60 */
void lp_set_options(const struct lp_opt *opt, phase_type phase);                               /* :153 */
void lp_update_extra_coeffs(lp_idx n_rows, lp_idx n_cols);                                     /* :73 */
void lp_set_rows(size_t i, boundlist const *rows);                                             /* :112 */
void lp_set_rows_hom(size_t i, boundlist const *rows);                                         /* :118 */
void lp_set_cols(size_t i, boundlist const *cols);                                             /* :124 */
void lp_set_cols_hom(size_t i, boundlist const *cols);                                         /* :130 */
void lp_set_mat_row(size_t i, list1d *list, lp_idx ridx);                                      /* :136 */
void lp_clear_obj_coeffs(size_t i);                                                            /* :141 */
void lp_set_obj_coeffs(size_t i, list1d const *list);                                          /* :147 */
lp_status_type lp_solve(size_t i);                                                             /* :219 */
void lp_primal_solution_rows(size_t i, double *const x, lp_idx firstidx, lp_idx size, double sign);   /* :261 */
void lp_primal_solution_cols(size_t i, double *const x, lp_idx firstidx, lp_idx size, double sign);   /* :272 */
void lp_dual_solution_rows(size_t i, double *const u, lp_idx firstidx, lp_idx size, double sign);     /* :283 */
void lp_dual_solution_cols(size_t i, double *const u, lp_idx firstidx, lp_idx size, double sign);     /* :294 */
double lp_obj_val(size_t i);                                                                   /* :305 */
double lp_get_time(size_t i);                                                                  /* :310 */
int lp_get_num(size_t i);                                                                      /* :315 */
void lp_free(size_t i);                                                                        /* :320 */

#ifdef __cplusplus
}
#endif
#endif
