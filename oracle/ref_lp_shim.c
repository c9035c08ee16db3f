/*
 * oracle/ref_lp_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Implements the reference's LP boundary (the 17 lp_* symbols bslv_algs.o and
 * bslv_main.o import, declared in /root/reference/bslv_lp.h:27-105) on top of
 * the oracle's own dense simplex (lp_dense.c).  It replaces the translation
 * unit bslv_lp.c, which cannot be built here because GLPK is absent.  It does
 * NOT provide any glp_* symbol or glpk.h stand-in.
 *
 * Built only in this container (needs the reference headers via
 * -I/root/reference) into oracle/_ref/bensolve_hybrid together with the
 * reference's own bslv_main.c bslv_algs.c bslv_vlp.c bslv_lists.c
 * bslv_poly.c, compiled where they lie.  The result is a HYBRID: reference
 * driver + reference polyhedron engine + oracle LP.  Everything it outputs is
 * "reference" for the poly half and "oracle" for the LP half.
 *
 * Call-by-call mirror of bslv_lp.c: lp_init :60, lp_update_extra_coeffs :73,
 * lp_set_rows :112, lp_set_rows_hom :118, lp_set_cols :124, lp_set_cols_hom :130,
 * lp_set_mat_row :136, lp_clear_obj_coeffs :141, lp_set_obj_coeffs :147,
 * lp_set_options :153, lp_solve :219, getters :261-308, lp_get_num :315, lp_free :320.
 */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include "bslv_lp.h"      /* from /root/reference */
#include "lp_dense.h"

static olp *P = NULL;
static int extra_rows = 0, extra_cols = 0;
static int method = OLP_PRIMAL;
static int num_solved = 0;

void lp_init(int row_cnt, int col_cnt, int nnz, int *row_idx, int *col_idx, double *data)
{
    P = olp_create(row_cnt, col_cnt);
    olp_load_coo(P, nnz, row_idx, col_idx, data);
    extra_rows = extra_cols = 0;
}

void lp_update_extra_coeffs(lp_idx n_rows, lp_idx n_cols)
{
    olp_resize_extra(P, extra_rows, extra_cols, n_rows, n_cols);
    extra_rows = n_rows; extra_cols = n_cols;
}

/* 'd'->DB at inhom, FX(0) at hom, etc. (bslv_lp.c:34-43) */
void lp_set_rows(size_t i, const boundlist *rows)
{ for (lp_idx k = 0; k < rows->size; k++) olp_set_row_bnds(P, rows->idx[k], rows->type[k], rows->lb[k], rows->ub[k]); }
void lp_set_rows_hom(size_t i, const boundlist *rows)
{ for (lp_idx k = 0; k < rows->size; k++) olp_set_row_bnds(P, rows->idx[k], rows->type[k] == 'd' ? 's' : rows->type[k], 0.0, 0.0); }
void lp_set_cols(size_t i, const boundlist *cols)
{ for (lp_idx k = 0; k < cols->size; k++) olp_set_col_bnds(P, cols->idx[k], cols->type[k], cols->lb[k], cols->ub[k]); }
void lp_set_cols_hom(size_t i, const boundlist *cols)
{ for (lp_idx k = 0; k < cols->size; k++) olp_set_col_bnds(P, cols->idx[k], cols->type[k] == 'd' ? 's' : cols->type[k], 0.0, 0.0); }

void lp_set_mat_row(size_t i, list1d *list, lp_idx ridx)
{ olp_set_mat_row(P, ridx, list->size, list->idx, list->data); }

void lp_clear_obj_coeffs(size_t i)
{ for (int k = 0; k <= olp_cols(P); k++) olp_set_obj(P, k, 0.0); }
void lp_set_obj_coeffs(size_t i, const list1d *obj)
{ for (lp_idx k = 0; k < obj->size; k++) olp_set_obj(P, obj->idx[k], obj->data[k]); }

void lp_set_options(const struct lp_opt *opt, phase_type phase)
{
    lp_method_type m;
    switch (phase) {
    case PHASE0: m = opt->method_phase0; break;
    case PHASE1_PRIMAL: m = opt->method_phase1; method = OLP_DUAL; break;
    case PHASE1_DUAL: m = opt->method_phase1; method = OLP_PRIMAL; break;
    case PHASE2_PRIMAL: m = opt->method_phase2; method = OLP_DUAL; break;
    case PHASE2_DUAL: m = opt->method_phase2; method = OLP_PRIMAL; break;
    default: assert(0);
    }
    switch (m) {
    case LP_METHOD_AUTO: break;
    case PRIMAL_SIMPLEX: method = OLP_PRIMAL; break;
    case DUAL_SIMPLEX: method = OLP_DUAL; break;
    case DUAL_PRIMAL_SIMPLEX: method = OLP_DUALP; break;
    default: assert(0);
    }
}

lp_status_type lp_solve(size_t i)
{
    int st = olp_solve(P, method);
    if (st == OLP_UNDEFINED) {            /* bslv_lp.c:222-227: retry from the standard basis */
        printf("LP solution is undefined, try again with standard basis\n");
        olp_std_basis(P);
        st = olp_solve(P, method);
    }
    if (st == OLP_OPTIMAL) { num_solved++; return LP_OPTIMAL; }
    if (st == OLP_INFEASIBLE) return LP_INFEASIBLE;
    if (st == OLP_UNBOUNDED) return LP_UNBOUNDED;
    return LP_UNEXPECTED_STATUS;
}

static void chk(const char *who, int last, int lim)
{ if (last > lim) { printf("%s: index out of bounds.\n", who); exit(1); } }

void lp_primal_solution_rows(size_t i, double *const x, lp_idx firstidx, lp_idx size, double sign)
{ chk("lp_primal_solution_rows", firstidx + size - 1, olp_rows(P)); for (lp_idx k = 0; k < size; k++) x[k] = sign * olp_row_prim(P, k + firstidx); }
void lp_primal_solution_cols(size_t i, double *const x, lp_idx firstidx, lp_idx size, double sign)
{ chk("lp_primal_solution_cols", firstidx + size - 1, olp_cols(P)); for (lp_idx k = 0; k < size; k++) x[k] = sign * olp_col_prim(P, k + firstidx); }
void lp_dual_solution_rows(size_t i, double *const u, lp_idx firstidx, lp_idx size, double sign)
{ chk("lp_dual_solution_rows", firstidx + size - 1, olp_rows(P)); for (lp_idx k = 0; k < size; k++) u[k] = sign * olp_row_dual(P, k + firstidx); }
void lp_dual_solution_cols(size_t i, double *const u, lp_idx firstidx, lp_idx size, double sign)
{ chk("lp_dual_solution_cols", firstidx + size - 1, olp_cols(P)); for (lp_idx k = 0; k < size; k++) u[k] = sign * olp_col_dual(P, k + firstidx); }

double lp_obj_val(size_t i) { return olp_obj_val(P); }
double lp_get_time(size_t i) { return 0; }
int lp_get_num(size_t i) { return num_solved; }
void lp_free(size_t i)
{
    if (getenv("ORACLE_LP_STATS"))
        fprintf(stderr, "oracle-lp: solved=%d iterations=%ld pivots=%ld\n", num_solved, olp_iterations(P), olp_pivots(P));
    olp_free(P); P = NULL;
}
