/* bslv_lp_compat.c -- the reference's lp_* symbols on top of the batched HIP engine (batch of one,
 * in place in tableau slot 0).  See include/bslv_lp_compat.h.  Plain C host code. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "../../../include/bslv_hip.h"
#include "../../../include/bslv_lp_compat.h"

static struct {
    int M, N;                /* current rows / cols (including the extra ones) */
    int extra_rows, extra_cols;
    double *A;               /* M x N dense master copy */
    char *rtype, *ctype;
    double *rlb, *rub, *clb, *cub;
    double *cost;            /* N+1, cost[0] = shift */
    bslv_lpq *eng;           /* NULL = model changed, engine must be (re)built */
    int have_basis;          /* slot 0 holds a basis of the current model */
    int bounds_dirty;
    int num;                 /* optimal solves */
    int last_status;
} G;

static void die(const char *what)
{
    fprintf(stderr, "bslv_lp_compat: %s: %s\n", what, bslv_last_error());
    exit(3);
}

static void resize(int M1, int N1)
{
    int M0 = G.M < M1 ? G.M : M1, N0 = G.N < N1 ? G.N : N1;
    double *A = (double *)calloc((size_t)(M1 ? M1 : 1) * (N1 ? N1 : 1), sizeof(double));
    for (int i = 0; i < M0; i++) memcpy(A + (size_t)i * N1, G.A + (size_t)i * G.N, N0 * sizeof(double));
    free(G.A); G.A = A;
    G.rtype = (char *)realloc(G.rtype, M1 + 1); G.rlb = (double *)realloc(G.rlb, (M1 + 1) * 8); G.rub = (double *)realloc(G.rub, (M1 + 1) * 8);
    G.ctype = (char *)realloc(G.ctype, N1 + 1); G.clb = (double *)realloc(G.clb, (N1 + 1) * 8); G.cub = (double *)realloc(G.cub, (N1 + 1) * 8);
    G.cost = (double *)realloc(G.cost, (N1 + 2) * 8);
    for (int i = M0; i < M1; i++) { G.rtype[i] = 'f'; G.rlb[i] = G.rub[i] = 0; }      /* glp_add_rows: free rows */
    for (int j = N0; j < N1; j++) { G.ctype[j] = 's'; G.clb[j] = G.cub[j] = 0; G.cost[j + 1] = 0; }   /* glp_add_cols: fixed at 0 */
    G.M = M1; G.N = N1;
    if (G.eng) { bslv_lpq_destroy(G.eng); G.eng = NULL; }
    G.have_basis = 0;
}

void lp_init(int row_cnt, int col_cnt, int nnz, lp_idx *row_idx, lp_idx *col_idx, double *data)
{
    memset(&G, 0, sizeof G);
    G.cost = (double *)calloc(2, 8);
    resize(row_cnt, col_cnt);
    G.cost[0] = 0;
    for (int k = 0; k < nnz; k++) G.A[(size_t)(row_idx[k] - 1) * G.N + (col_idx[k] - 1)] = data[k];
}

void lp_update_extra_coeffs(lp_idx n_rows, lp_idx n_cols)
{
    int M0 = G.M - G.extra_rows, N0 = G.N - G.extra_cols;
    resize(M0, N0);                         /* drop the previous extra block */
    resize(M0 + n_rows, N0 + n_cols);       /* append new empty rows / cols; standard basis */
    G.extra_rows = n_rows; G.extra_cols = n_cols;
}

void lp_set_options(const struct lp_opt *opt, phase_type phase) { (void)opt; (void)phase; }

static void set_bnd(char *types, double *lb, double *ub, const boundlist *L, int hom)
{
    for (lp_idx k = 0; k < L->size; k++) {
        int i = L->idx[k] - 1;
        char t = L->type[k];
        if (hom) { types[i] = (t == 'd') ? 's' : t; lb[i] = 0; ub[i] = 0; }
        else { types[i] = t; lb[i] = L->lb[k]; ub[i] = L->ub[k]; }
    }
    G.bounds_dirty = 1;
}
void lp_set_rows(size_t i, const boundlist *rows) { (void)i; set_bnd(G.rtype, G.rlb, G.rub, rows, 0); }
void lp_set_rows_hom(size_t i, const boundlist *rows) { (void)i; set_bnd(G.rtype, G.rlb, G.rub, rows, 1); }
void lp_set_cols(size_t i, const boundlist *cols) { (void)i; set_bnd(G.ctype, G.clb, G.cub, cols, 0); }
void lp_set_cols_hom(size_t i, const boundlist *cols) { (void)i; set_bnd(G.ctype, G.clb, G.cub, cols, 1); }

void lp_set_mat_row(size_t i, list1d *list, lp_idx ridx)
{
    (void)i;
    double *a = G.A + (size_t)(ridx - 1) * G.N;
    memset(a, 0, G.N * sizeof(double));
    for (lp_idx k = 0; k < list->size; k++) a[list->idx[k] - 1] = list->data[k];
    if (G.eng) { bslv_lpq_destroy(G.eng); G.eng = NULL; }
    G.have_basis = 0;
}

void lp_clear_obj_coeffs(size_t i)
{
    (void)i;
    for (int k = 0; k <= G.N; k++) G.cost[k] = 0;
    if (G.eng) { bslv_lpq_destroy(G.eng); G.eng = NULL; }      /* reduced costs change: rebuild (cold start) */
    G.have_basis = 0;
}
void lp_set_obj_coeffs(size_t i, const list1d *obj)
{
    (void)i;
    for (lp_idx k = 0; k < obj->size; k++) G.cost[obj->idx[k]] = obj->data[k];
    if (G.eng) { bslv_lpq_destroy(G.eng); G.eng = NULL; }
    G.have_basis = 0;
}

static void bounds_arrays(double *lo, double *up)
{
    for (int i = 0; i < G.M; i++) {
        char t = G.rtype[i];
        lo[i] = (t == 'l' || t == 'd' || t == 's') ? G.rlb[i] : -INFINITY;
        up[i] = (t == 'u' || t == 'd') ? G.rub[i] : (t == 's' ? G.rlb[i] : INFINITY);
    }
    for (int j = 0; j < G.N; j++) {
        char t = G.ctype[j];
        lo[G.M + j] = (t == 'l' || t == 'd' || t == 's') ? G.clb[j] : -INFINITY;
        up[G.M + j] = (t == 'u' || t == 'd') ? G.cub[j] : (t == 's' ? G.clb[j] : INFINITY);
    }
}

lp_status_type lp_solve(size_t i)
{
    (void)i;
    double *lo = (double *)malloc((G.M + G.N) * 8), *up = (double *)malloc((G.M + G.N) * 8);
    bounds_arrays(lo, up);
    if (!G.eng) {
        if (bslv_lpq_create(&G.eng, G.M, G.N, G.A, lo, up, G.cost, 0, 0, 2)) die("bslv_lpq_create");
        G.have_basis = 0;
    } else if (G.bounds_dirty) {
        if (bslv_lpq_set_bounds(G.eng, lo, up)) die("bslv_lpq_set_bounds");
    }
    G.bounds_dirty = 0;
    free(lo); free(up);
    const int zero = 0;
    int st = BSLV_LP_UNDEFINED, it = 0;
    if (!G.have_basis) { if (bslv_lpq_reset_slot(G.eng, 0)) die("bslv_lpq_reset_slot"); G.have_basis = 1; }
    if (bslv_lpq_solve_batch(G.eng, 1, &zero, &zero, NULL, NULL, &st, &it)) die("bslv_lpq_solve_batch");
    if (st == BSLV_LP_UNDEFINED) {           /* bslv_lp.c:222-227: try again with the standard basis */
        if (bslv_lpq_reset_slot(G.eng, 0)) die("bslv_lpq_reset_slot");
        if (bslv_lpq_solve_batch(G.eng, 1, &zero, &zero, NULL, NULL, &st, &it)) die("bslv_lpq_solve_batch");
    }
    G.last_status = st;
    if (st == BSLV_LP_OPTIMAL) { G.num++; return LP_OPTIMAL; }
    G.have_basis = 0;                        /* a failed solve leaves no trustworthy warm start */
    if (st == BSLV_LP_INFEASIBLE) return LP_INFEASIBLE;
    if (st == BSLV_LP_UNBOUNDED) return LP_UNBOUNDED;
    return LP_UNEXPECTED_STATUS;
}

static void getv(const char *who, int dual, int base, int lim, double *const x, lp_idx firstidx, lp_idx size, double sign)
{
    if (firstidx + size - 1 > lim) { printf("%s: index out of bounds.\n", who); exit(1); }
    const int zero = 0;
    int rc = dual ? bslv_lpq_get_dual(G.eng, 1, &zero, base + firstidx - 1, size, x) : bslv_lpq_get_primal(G.eng, 1, &zero, base + firstidx - 1, size, x);
    if (rc) die(who);
    for (lp_idx k = 0; k < size; k++) x[k] *= sign;
}
void lp_primal_solution_rows(size_t i, double *const x, lp_idx f, lp_idx n, double sign) { (void)i; getv("lp_primal_solution_rows", 0, 0, G.M, x, f, n, sign); }
void lp_primal_solution_cols(size_t i, double *const x, lp_idx f, lp_idx n, double sign) { (void)i; getv("lp_primal_solution_cols", 0, G.M, G.N, x, f, n, sign); }
void lp_dual_solution_rows(size_t i, double *const u, lp_idx f, lp_idx n, double sign) { (void)i; getv("lp_dual_solution_rows", 1, 0, G.M, u, f, n, sign); }
void lp_dual_solution_cols(size_t i, double *const u, lp_idx f, lp_idx n, double sign) { (void)i; getv("lp_dual_solution_cols", 1, G.M, G.N, u, f, n, sign); }

double lp_obj_val(size_t i)
{
    (void)i;
    const int zero = 0;
    double v = 0;
    if (bslv_lpq_get_obj(G.eng, 1, &zero, &v)) die("bslv_lpq_get_obj");
    return v;
}
double lp_get_time(size_t i) { (void)i; return 0; }          /* never written in the reference either (bslv_lp.c:29,310) */
int lp_get_num(size_t i) { (void)i; return G.num; }
void lp_free(size_t i)
{
    (void)i;
    if (G.eng) bslv_lpq_destroy(G.eng);
    free(G.A); free(G.rtype); free(G.ctype); free(G.rlb); free(G.rub); free(G.clb); free(G.cub); free(G.cost);
    memset(&G, 0, sizeof G);
}
