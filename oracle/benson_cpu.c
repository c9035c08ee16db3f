/*
 * oracle/benson_cpu.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).  See benson_cpu.h.
 * Follows bslv_algs.c: init_P2 :574-664 (inhomogeneous), phase2_primal PART 1 :976-1018,
 * PART 2 :1025-1082.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include "lp_dense.h"
#include "benson_cpu.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int obenson_phase2_primal(int m, int n, int q, const double *A, const double *P,
                          const char *rtype, const double *rlb, const double *rub,
                          const char *ctype, const double *clb, const double *cub,
                          const double *R, int r, const double *c, double eps, long max_lps,
                          opoly **poly_out, obenson_stats *st)
{
    return obenson_phase2_primal_ex(m, n, q, A, P, rtype, rlb, rub, ctype, clb, cub, R, r, c, eps, max_lps, 0, 0, poly_out, st);
}

int obenson_phase2_primal_ex(int m, int n, int q, const double *A, const double *P,
                             const char *rtype, const double *rlb, const double *rub,
                             const char *ctype, const double *clb, const double *cub,
                             const double *R, int r, const double *c, double eps, long max_lps,
                             int order, long warm_lps, opoly **poly_out, obenson_stats *st)
{
    memset(st, 0, sizeof(*st));
    double t_start = now();
    /* base problem [A 0; -P I] (lp_init, bslv_main.c:258; bslv_vlp.c:376-453) + P2 extras */
    olp *lp = olp_create(m + q, n + q);
    {
        int nnz = m * n + q * n + q, k = 0;
        int *ri = (int *)malloc(nnz * sizeof(int)), *ci = (int *)malloc(nnz * sizeof(int));
        double *v = (double *)malloc(nnz * sizeof(double));
        for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) if (A[(size_t)i * n + j] != 0) { ri[k] = i + 1; ci[k] = j + 1; v[k++] = A[(size_t)i * n + j]; }
        for (int i = 0; i < q; i++) for (int j = 0; j < n; j++) if (P[(size_t)i * n + j] != 0) { ri[k] = m + i + 1; ci[k] = j + 1; v[k++] = -P[(size_t)i * n + j]; }
        for (int i = 0; i < q; i++) { ri[k] = m + i + 1; ci[k] = n + i + 1; v[k++] = 1.0; }
        olp_load_coo(lp, k, ri, ci, v);
        free(ri); free(ci); free(v);
    }
    olp_resize_extra(lp, 0, 0, r + 1, 1);
    {
        int *ci = (int *)malloc((q + 1) * sizeof(int));
        double *v = (double *)malloc((q + 1) * sizeof(double));
        for (int i = 0; i < r; i++) {
            for (int j = 0; j < q; j++) { ci[j] = n + j + 1; v[j] = R[(size_t)j * r + i]; }
            ci[q] = n + q + 1; v[q] = -1.0;
            olp_set_mat_row(lp, m + q + 1 + i, q + 1, ci, v);
        }
        olp_set_mat_row(lp, m + q + r + 1, 0, ci, v);     /* eta row: free in the inhomogeneous problem */
        free(ci); free(v);
    }
    for (int j = 0; j <= n + q + 1; j++) olp_set_obj(lp, j, 0.0);
    olp_set_obj(lp, n + q + 1, 1.0);
    for (int i = 0; i < m; i++) olp_set_row_bnds(lp, i + 1, rtype[i], rlb[i], rub[i]);
    for (int i = 0; i < q; i++) olp_set_row_bnds(lp, m + i + 1, 's', 0, 0);
    for (int i = 0; i < r; i++) olp_set_row_bnds(lp, m + q + i + 1, 'u', 0, 0);
    olp_set_row_bnds(lp, m + q + r + 1, 'f', 0, 0);
    for (int j = 0; j < n; j++) olp_set_col_bnds(lp, j + 1, ctype[j], clb[j], cub[j]);
    for (int j = 0; j <= q; j++) olp_set_col_bnds(lp, n + j + 1, 'f', 0, 0);

    opoly *up = opoly_create(q, OPOLY_LOWER2UPPER, c);
    double *val = (double *)malloc(q * sizeof(double)), *ww = (double *)malloc(q * sizeof(double));
    double *yy = (double *)malloc(q * sizeof(double));
    int rc = 0;
    /* PART 1: r weighted-sum LPs */
    for (int j = 0; j < r && !rc; j++) {
        for (int i = 0; i < r; i++) olp_set_row_bnds(lp, m + q + i + 1, i == j ? 'u' : 'f', 0, 0);
        for (int k = 0; k < q; k++) val[k] = R[(size_t)k * r + j];
        double t0 = now();
        int s = olp_solve(lp, OLP_DUAL);
        st->secs_lp += now() - t0;
        if (s != OLP_OPTIMAL) { rc = (s == OLP_INFEASIBLE) ? 1 : 2; break; }
        st->lps++;
        val[q - 1] = olp_obj_val(lp);
        opoly_add(up, val, 0);
    }
    if (!rc) {
        double t0 = now();
        opoly_init(up);
        st->secs_poly += now() - t0;
        for (int i = 0; i < r; i++) olp_set_row_bnds(lp, m + q + i + 1, 'u', 0, 0);
    }
    /* PART 2 */
    while (!rc) {
        int ideal, idx;
        if (order == 1 ? opoly_next_newest(up, val, &ideal, &idx) : opoly_next(up, val, &ideal, &idx)) break;
        if (ideal) { opoly_mark(up, idx); continue; }
        if (max_lps > 0 && st->lps >= max_lps) { rc = 3; break; }
        if (warm_lps > 0 && st->lps >= warm_lps && st->warm_lps == 0) {      /* end of the untimed-by-the-caller warm-up */
            st->warm_lps = st->lps; st->warm_cuts = st->cuts; st->warm_pivots = olp_pivots(lp);
            st->warm_new_vertices = opoly_new_vertices(up); st->warm_secs = now() - t_start;
        }
        for (int j = 0; j < r; j++) {
            double ub = 0;
            for (int k = 0; k < q; k++) ub += R[(size_t)k * r + j] * val[k];
            olp_set_row_bnds(lp, m + q + j + 1, 'u', 0, ub);
        }
        double t0 = now();
        int s = olp_solve(lp, OLP_DUAL);
        st->secs_lp += now() - t0;
        if (s != OLP_OPTIMAL) { rc = (s == OLP_INFEASIBLE) ? 1 : 2; break; }
        st->lps++;
        for (int k = 0; k < q; k++) { ww[k] = olp_row_dual(lp, m + k + 1); yy[k] = olp_col_prim(lp, n + k + 1); }
        double z = olp_obj_val(lp), last = 0;
        for (int k = 0; k < q - 1; k++) val[k] = ww[k];
        for (int k = 0; k < q; k++) last += yy[k] * ww[k];
        val[q - 1] = last;
        if (z > eps) {
            t0 = now();
            if (opoly_add(up, val, 0) == 0) st->cuts++;
            st->secs_poly += now() - t0;
        } else
            opoly_mark(up, idx);
    }
    st->pivots = olp_pivots(lp);
    st->new_vertices = opoly_new_vertices(up);
    st->status = rc;
    st->secs_total = now() - t_start;
    free(val); free(ww); free(yy);
    olp_free(lp);
    if (poly_out) *poly_out = up; else opoly_free(up);
    return rc;
}
