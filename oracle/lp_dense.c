/*
 * oracle/lp_dense.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).  See lp_dense.h.
 *
 * Model (GLPK's, as used through bslv_lp.c): auxiliary variables r = A x,
 * every variable (aux 1..M, structural 1..N) has a bound type f/l/u/d/s,
 * minimise c.x + c0.  The basis is kept as a dense compact tableau
 *     x_B = T x_N          (T is M x N, initially T = A: all aux basic)
 * with reduced costs d (objective = d . x_N).  Row duals / column duals are
 * the reduced costs of the aux / structural variables, which is the sign
 * convention the reference's getters rely on (bslv_lp.c:283-303,
 * SURVEY.md section 8a row L6).
 *
 * PARITY UNPINNED by the reference (no GLPK here, no golden vectors there).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#include "lp_dense.h"

#define NS_L 0   /* nonbasic at lower bound */
#define NS_U 1   /* nonbasic at upper bound */
#define NS_F 2   /* nonbasic free, value 0  */
#define NS_S 3   /* nonbasic fixed          */

#define TOL_BND 1e-9   /* primal feasibility (scaled by 1+|bound|) */
#define TOL_DJ  1e-9   /* dual feasibility */
#define TOL_PIV 1e-9   /* smallest admissible pivot magnitude (relative to row/col max) */

struct olp {
    int M, N;
    double *A;            /* M x N row-major master copy */
    char *rtype, *ctype;
    double *rlb, *rub, *clb, *cub;
    double *c;            /* N+1, c[0] = shift */
    /* basis */
    int valid;
    double *T;            /* M x N */
    int *bh, *nh;         /* variable ids: 0..M-1 aux, M..M+N-1 structural */
    int *nstat;           /* per column */
    double *xN, *beta, *d;
    int *posB, *posN;     /* per variable: row if basic else -1 / col if nonbasic else -1 */
    double obj;
    int status;
    long iters, pivots;
    unsigned char *art;   /* per variable: bit0 artificial lower bound -BIG, bit1 artificial upper bound +BIG */
};
#define ART_BIG 1e7

static double lbv(const olp *lp, int k)
{
    char t = k < lp->M ? lp->rtype[k] : lp->ctype[k - lp->M];
    double lb = k < lp->M ? lp->rlb[k] : lp->clb[k - lp->M];
    if (t == 'l' || t == 'd' || t == 's') return lb;
    return (lp->art && (lp->art[k] & 1)) ? -ART_BIG : -INFINITY;
}
static double ubv(const olp *lp, int k)
{
    char t = k < lp->M ? lp->rtype[k] : lp->ctype[k - lp->M];
    double lb = k < lp->M ? lp->rlb[k] : lp->clb[k - lp->M];
    double ub = k < lp->M ? lp->rub[k] : lp->cub[k - lp->M];
    if (t == 's') return lb;
    if (t == 'u' || t == 'd') return ub;
    return (lp->art && (lp->art[k] & 2)) ? ART_BIG : INFINITY;
}
static double cost(const olp *lp, int k) { return k < lp->M ? 0.0 : lp->c[k - lp->M + 1]; }

olp *olp_create(int rows, int cols)
{
    olp *lp = (olp *)calloc(1, sizeof(olp));
    lp->M = 0; lp->N = 0;
    olp_resize_extra(lp, 0, 0, rows, cols);
    return lp;
}

static void free_basis(olp *lp)
{
    free(lp->T); free(lp->bh); free(lp->nh); free(lp->nstat); free(lp->xN);
    free(lp->beta); free(lp->d); free(lp->posB); free(lp->posN);
    lp->T = NULL; lp->bh = lp->nh = lp->nstat = lp->posB = lp->posN = NULL;
    lp->xN = lp->beta = lp->d = NULL;
}

void olp_free(olp *lp)
{
    if (!lp) return;
    free_basis(lp);
    free(lp->A); free(lp->rtype); free(lp->ctype); free(lp->rlb); free(lp->rub);
    free(lp->clb); free(lp->cub); free(lp->c); free(lp->art);
    free(lp);
}

int olp_rows(const olp *lp) { return lp->M; }
int olp_cols(const olp *lp) { return lp->N; }
long olp_iterations(const olp *lp) { return lp->iters; }
long olp_pivots(const olp *lp) { return lp->pivots; }

void olp_resize_extra(olp *lp, int drop_rows, int drop_cols, int add_rows, int add_cols)
{
    int M0 = lp->M - drop_rows, N0 = lp->N - drop_cols;
    int M1 = M0 + add_rows, N1 = N0 + add_cols;
    double *A = (double *)calloc((size_t)(M1 > 0 ? M1 : 1) * (N1 > 0 ? N1 : 1), sizeof(double));
    for (int i = 0; i < M0; i++)
        for (int j = 0; j < N0; j++)
            A[(size_t)i * N1 + j] = lp->A[(size_t)i * lp->N + j];
    free(lp->A); lp->A = A;
    lp->rtype = (char *)realloc(lp->rtype, M1 + 1);
    lp->rlb = (double *)realloc(lp->rlb, (M1 + 1) * sizeof(double));
    lp->rub = (double *)realloc(lp->rub, (M1 + 1) * sizeof(double));
    for (int i = M0; i < M1; i++) { lp->rtype[i] = 'f'; lp->rlb[i] = lp->rub[i] = 0; }   /* new rows: free */
    lp->ctype = (char *)realloc(lp->ctype, N1 + 1);
    lp->clb = (double *)realloc(lp->clb, (N1 + 1) * sizeof(double));
    lp->cub = (double *)realloc(lp->cub, (N1 + 1) * sizeof(double));
    lp->c = (double *)realloc(lp->c, (N1 + 2) * sizeof(double));
    if (lp->N == 0 && drop_cols == 0 && N0 == 0) lp->c[0] = 0;
    for (int j = N0; j < N1; j++) { lp->ctype[j] = 's'; lp->clb[j] = lp->cub[j] = 0; lp->c[j + 1] = 0; } /* new cols: fixed 0 */
    lp->M = M1; lp->N = N1;
    free_basis(lp);
    lp->valid = 0;
    lp->status = OLP_UNDEFINED;
}

void olp_load_coo(olp *lp, int nnz, const int *ridx, const int *cidx, const double *val)
{
    memset(lp->A, 0, (size_t)lp->M * lp->N * sizeof(double));
    for (int k = 0; k < nnz; k++)
        lp->A[(size_t)(ridx[k] - 1) * lp->N + (cidx[k] - 1)] = val[k];
    lp->valid = 0;
}

void olp_set_mat_row(olp *lp, int row, int len, const int *cidx, const double *val)
{
    double *a = lp->A + (size_t)(row - 1) * lp->N;
    memset(a, 0, lp->N * sizeof(double));
    for (int k = 0; k < len; k++) a[cidx[k] - 1] = val[k];
    lp->valid = 0;   /* tableau must be rebuilt; callers do this right after a std-basis reset */
}

void olp_set_row_bnds(olp *lp, int row, char type, double lb, double ub)
{ lp->rtype[row - 1] = type; lp->rlb[row - 1] = lb; lp->rub[row - 1] = ub; }
void olp_set_col_bnds(olp *lp, int col, char type, double lb, double ub)
{ lp->ctype[col - 1] = type; lp->clb[col - 1] = lb; lp->cub[col - 1] = ub; }
void olp_set_obj(olp *lp, int col, double val) { lp->c[col] = val; }
void olp_std_basis(olp *lp) { lp->valid = 0; }

static void build_std_basis(olp *lp)
{
    int M = lp->M, N = lp->N;
    free_basis(lp);
    lp->T = (double *)malloc((size_t)(M ? M : 1) * (N ? N : 1) * sizeof(double));
    memcpy(lp->T, lp->A, (size_t)M * N * sizeof(double));
    lp->bh = (int *)malloc((M + 1) * sizeof(int));
    lp->nh = (int *)malloc((N + 1) * sizeof(int));
    lp->nstat = (int *)malloc((N + 1) * sizeof(int));
    lp->xN = (double *)calloc(N + 1, sizeof(double));
    lp->beta = (double *)calloc(M + 1, sizeof(double));
    lp->d = (double *)calloc(N + 1, sizeof(double));
    lp->posB = (int *)malloc((M + N + 1) * sizeof(int));
    lp->posN = (int *)malloc((M + N + 1) * sizeof(int));
    for (int i = 0; i < M; i++) { lp->bh[i] = i; lp->posB[i] = i; lp->posN[i] = -1; }
    for (int j = 0; j < N; j++) { lp->nh[j] = M + j; lp->posN[M + j] = j; lp->posB[M + j] = -1; lp->nstat[j] = NS_L; }
    lp->valid = 1;
}

/* make nonbasic statuses consistent with the current bound types, set xN */
static void sanitize(olp *lp)
{
    for (int j = 0; j < lp->N; j++) {
        int k = lp->nh[j];
        double lo = lbv(lp, k), up = ubv(lp, k);
        int st = lp->nstat[j];
        if (lo == up) st = NS_S;
        else if (isinf(lo) && isinf(up)) st = NS_F;
        else if (isinf(lo)) st = NS_U;
        else if (isinf(up)) st = NS_L;
        else if (st != NS_L && st != NS_U) st = NS_L;
        lp->nstat[j] = st;
        lp->xN[j] = (st == NS_F) ? 0.0 : (st == NS_U ? up : lo);
    }
}

static void refresh_beta(olp *lp)
{
    int M = lp->M, N = lp->N;
    for (int i = 0; i < M; i++) {
        const double *t = lp->T + (size_t)i * N;
        double s = 0;
        for (int j = 0; j < N; j++) s += t[j] * lp->xN[j];
        lp->beta[i] = s;
    }
}

static void refresh_d(olp *lp)
{
    int M = lp->M, N = lp->N;
    for (int j = 0; j < N; j++) lp->d[j] = cost(lp, lp->nh[j]);
    for (int i = 0; i < M; i++) {
        double cb = cost(lp, lp->bh[i]);
        if (cb == 0.0) continue;
        const double *t = lp->T + (size_t)i * N;
        for (int j = 0; j < N; j++) lp->d[j] += cb * t[j];
    }
}

static void pivot(olp *lp, int r, int q)
{
    int M = lp->M, N = lp->N;
    double *tr = lp->T + (size_t)r * N;
    double p = 1.0 / tr[q];
    {
        double f = lp->d[q] * p;
        if (f != 0.0) for (int j = 0; j < N; j++) lp->d[j] -= f * tr[j];
        lp->d[q] = f;
    }
    for (int i = 0; i < M; i++) {
        if (i == r) continue;
        double *ti = lp->T + (size_t)i * N;
        double f = ti[q] * p;
        if (f != 0.0) for (int j = 0; j < N; j++) ti[j] -= f * tr[j];
        ti[q] = f;
    }
    for (int j = 0; j < N; j++) tr[j] = -tr[j] * p;
    tr[q] = p;
    int kb = lp->bh[r], kn = lp->nh[q];
    lp->bh[r] = kn; lp->nh[q] = kb;
    lp->posB[kn] = r; lp->posN[kn] = -1;
    lp->posN[kb] = q; lp->posB[kb] = -1;
    lp->pivots++;
}

static double btol(double b) { return TOL_BND * (1.0 + fabs(b)); }

/* returns -1 below lower, +1 above upper, 0 feasible */
static int infeas_sign(const olp *lp, int i)
{
    int k = lp->bh[i];
    double lo = lbv(lp, k), up = ubv(lp, k), b = lp->beta[i];
    if (!isinf(lo) && b < lo - btol(lo)) return -1;
    if (!isinf(up) && b > up + btol(up)) return +1;
    return 0;
}

static int dual_feasible(const olp *lp)
{
    for (int j = 0; j < lp->N; j++) {
        double dj = lp->d[j];
        switch (lp->nstat[j]) {
        case NS_L: if (dj < -TOL_DJ) return 0; break;
        case NS_U: if (dj > TOL_DJ) return 0; break;
        case NS_F: if (fabs(dj) > TOL_DJ) return 0; break;
        default: break;
        }
    }
    return 1;
}

static void leave_to(olp *lp, int q, int leaving_var, int upper)
{
    double lo = lbv(lp, leaving_var), up = ubv(lp, leaving_var);
    if (lo == up) { lp->nstat[q] = NS_S; lp->xN[q] = lo; }
    else if (upper) { lp->nstat[q] = NS_U; lp->xN[q] = up; }
    else { lp->nstat[q] = NS_L; lp->xN[q] = lo; }
}

static int primal_simplex(olp *lp)
{
    int M = lp->M, N = lp->N;
    long maxit = 200L * (M + N) + 1000;
    double *d1 = (double *)malloc((N + 1) * sizeof(double));
    int *sig = (int *)malloc((M + 1) * sizeof(int));
    int ret = OLP_UNDEFINED;
    long since_refresh = 0;
    int verified = 1;        /* numbers are fresh (olp_solve refreshed them) */
    for (long it = 0; it < maxit; it++) {
        if (++since_refresh >= 100) { refresh_beta(lp); refresh_d(lp); since_refresh = 0; }
        int ninf = 0;
        for (int i = 0; i < M; i++) { sig[i] = infeas_sign(lp, i); if (sig[i]) ninf++; }
        const double *dj = lp->d;
        if (ninf) {
            memset(d1, 0, N * sizeof(double));
            for (int i = 0; i < M; i++) if (sig[i]) {
                const double *t = lp->T + (size_t)i * N;
                double s = sig[i];
                for (int j = 0; j < N; j++) d1[j] += s * t[j];
            }
            dj = d1;
        }
        /* pricing (Dantzig) */
        int q = -1, dir = 0; double best = 0;
        for (int j = 0; j < N; j++) {
            int st = lp->nstat[j];
            double v = dj[j], sc = 0; int dd = 0;
            if (st == NS_S) continue;
            if (st == NS_L) { if (v < -TOL_DJ) { sc = -v; dd = +1; } }
            else if (st == NS_U) { if (v > TOL_DJ) { sc = v; dd = -1; } }
            else { if (fabs(v) > TOL_DJ) { sc = fabs(v); dd = v < 0 ? +1 : -1; } }
            if (sc > best) { best = sc; q = j; dir = dd; }
        }
        if (q < 0) {
            /* re-check with freshly recomputed numbers before concluding */
            if (!verified) { refresh_beta(lp); refresh_d(lp); since_refresh = 0; verified = 1; continue; }
            ret = ninf ? OLP_INFEASIBLE : OLP_OPTIMAL;
            break;
        }
        /* Harris two-pass ratio test */
        int kq = lp->nh[q];
        double gap = ubv(lp, kq) - lbv(lp, kq);       /* inf unless boxed */
        double colmax = 0;
        for (int i = 0; i < M; i++) { double a = fabs(lp->T[(size_t)i * N + q]); if (a > colmax) colmax = a; }
        double ptol = TOL_PIV * (1.0 + colmax);
        double tmax = gap;                              /* pass 1: relaxed bound */
        for (int i = 0; i < M; i++) {
            double a = lp->T[(size_t)i * N + q] * dir;
            if (fabs(a) < ptol) continue;
            int k = lp->bh[i];
            double lo = lbv(lp, k), up = ubv(lp, k), b = lp->beta[i], t = INFINITY;
            if (sig[i] == 0) {
                if (a > 0 && !isinf(up)) t = (up + btol(up) - b) / a;
                else if (a < 0 && !isinf(lo)) t = (lo - btol(lo) - b) / a;
            } else if (sig[i] < 0) { if (a > 0) t = (lo + btol(lo) - b) / a; }
            else { if (a < 0) t = (up - btol(up) - b) / a; }
            if (t < tmax) tmax = t;
        }
        if (isinf(tmax)) { ret = ninf ? OLP_UNEXPECTED : OLP_UNBOUNDED; break; }
        int r = -1, hit_upper = 0; double amax = 0, tstep = gap;
        for (int i = 0; i < M; i++) {                   /* pass 2: largest pivot within tmax */
            double a = lp->T[(size_t)i * N + q] * dir;
            if (fabs(a) < ptol) continue;
            int k = lp->bh[i];
            double lo = lbv(lp, k), up = ubv(lp, k), b = lp->beta[i], t = INFINITY; int hu = 0;
            if (sig[i] == 0) {
                if (a > 0 && !isinf(up)) { t = (up - b) / a; hu = 1; }
                else if (a < 0 && !isinf(lo)) { t = (lo - b) / a; hu = 0; }
            } else if (sig[i] < 0) { if (a > 0) { t = (lo - b) / a; hu = 0; } }
            else { if (a < 0) { t = (up - b) / a; hu = 1; } }
            if (isinf(t)) continue;
            if (t <= tmax && fabs(a) > amax) { amax = fabs(a); r = i; hit_upper = hu; tstep = t < 0 ? 0 : t; }
        }
        lp->iters++;
        verified = 0;
        if (r < 0 || (!isinf(gap) && gap <= tstep)) {
            /* bound flip of the entering variable */
            double t = gap;
            for (int i = 0; i < M; i++) lp->beta[i] += lp->T[(size_t)i * N + q] * dir * t;
            lp->nstat[q] = (lp->nstat[q] == NS_L) ? NS_U : NS_L;
            lp->xN[q] = (lp->nstat[q] == NS_U) ? ubv(lp, kq) : lbv(lp, kq);
            continue;
        }
        double enter_val = lp->xN[q] + dir * tstep;
        for (int i = 0; i < M; i++) if (i != r) lp->beta[i] += lp->T[(size_t)i * N + q] * dir * tstep;
        int kleave = lp->bh[r];
        pivot(lp, r, q);
        leave_to(lp, q, kleave, hit_upper);
        lp->beta[r] = enter_val;
    }
    free(d1); free(sig);
    return ret;
}

static int dual_simplex(olp *lp)
{
    int M = lp->M, N = lp->N;
    long maxit = 200L * (M + N) + 1000;
    int ret = OLP_UNDEFINED;
    long since_refresh = 0;
    int verified = 1;
    for (long it = 0; it < maxit; it++) {
        if (++since_refresh >= 100) { refresh_beta(lp); refresh_d(lp); since_refresh = 0; }
        /* leaving row: largest bound violation */
        int r = -1, below = 0; double worst = 0;
        for (int i = 0; i < M; i++) {
            int k = lp->bh[i];
            double lo = lbv(lp, k), up = ubv(lp, k), b = lp->beta[i];
            if (!isinf(lo) && lo - b > btol(lo) && lo - b > worst) { worst = lo - b; r = i; below = 1; }
            if (!isinf(up) && b - up > btol(up) && b - up > worst) { worst = b - up; r = i; below = 0; }
        }
        if (r < 0) {
            if (!verified) { refresh_beta(lp); refresh_d(lp); since_refresh = 0; verified = 1; continue; }
            ret = OLP_OPTIMAL; break;
        }
        const double *tr = lp->T + (size_t)r * N;
        double sgn = below ? 1.0 : -1.0;
        double rowmax = 0;
        for (int j = 0; j < N; j++) { double a = fabs(tr[j]); if (a > rowmax) rowmax = a; }
        double ptol = TOL_PIV * (1.0 + rowmax);
        double thmax = INFINITY;
        for (int j = 0; j < N; j++) {
            int st = lp->nstat[j]; if (st == NS_S) continue;
            double a = sgn * tr[j];
            if (fabs(a) < ptol) continue;
            if ((a > 0 && (st == NS_L || st == NS_F)) || (a < 0 && (st == NS_U || st == NS_F))) {
                double th = (fabs(lp->d[j]) + TOL_DJ) / fabs(a);
                if (th < thmax) thmax = th;
            }
        }
        if (isinf(thmax)) { ret = OLP_INFEASIBLE; break; }
        int q = -1; double amax = 0;
        for (int j = 0; j < N; j++) {
            int st = lp->nstat[j]; if (st == NS_S) continue;
            double a = sgn * tr[j];
            if (fabs(a) < ptol) continue;
            if ((a > 0 && (st == NS_L || st == NS_F)) || (a < 0 && (st == NS_U || st == NS_F))) {
                double th = fabs(lp->d[j]) / fabs(a);
                if (th <= thmax && fabs(a) > amax) { amax = fabs(a); q = j; }
            }
        }
        lp->iters++;
        verified = 0;
        int kleave = lp->bh[r];
        double target = below ? lbv(lp, kleave) : ubv(lp, kleave);
        double theta = (target - lp->beta[r]) / tr[q];
        double enter_val = lp->xN[q] + theta;
        for (int i = 0; i < M; i++) if (i != r) lp->beta[i] += lp->T[(size_t)i * N + q] * theta;
        pivot(lp, r, q);
        leave_to(lp, q, kleave, !below);
        lp->beta[r] = enter_val;
    }
    return ret;
}

int olp_solve(olp *lp, int method)
{
    if (!lp->valid) build_std_basis(lp);
    sanitize(lp);
    refresh_beta(lp);
    refresh_d(lp);
    int st;
    if (method != OLP_PRIMAL && !dual_feasible(lp)) {
        /* artificial-bounds start of the dual simplex: a nonbasic column whose reduced cost has the
         * wrong sign for lack of a bound gets a temporary bound of +-1e7 on that side, which makes the
         * basis dual feasible; if such a bound is still active at the end the LP is unbounded and
         * the primal simplex below decides */
        free(lp->art);
        lp->art = (unsigned char *)calloc(lp->M + lp->N + 1, 1);
        for (int j = 0; j < lp->N; j++) {
            int k = lp->nh[j];
            double dj = lp->d[j], lo = lbv(lp, k), up = ubv(lp, k);
            if (lo == up) continue;
            if (dj > TOL_DJ && isinf(lo)) lp->art[k] |= 1;
            if (dj < -TOL_DJ && isinf(up)) lp->art[k] |= 2;
        }
        for (int j = 0; j < lp->N; j++) {
            int k = lp->nh[j];
            double dj = lp->d[j], lo = lbv(lp, k), up = ubv(lp, k);
            if (lo == up || (isinf(lo) && isinf(up))) continue;
            if (dj > TOL_DJ && !isinf(lo)) lp->nstat[j] = NS_L;
            else if (dj < -TOL_DJ && !isinf(up)) lp->nstat[j] = NS_U;
        }
        sanitize(lp);
        refresh_beta(lp);
        if (dual_feasible(lp)) {
            st = dual_simplex(lp);
            int active = 0;
            for (int j = 0; j < lp->N; j++) {
                int k = lp->nh[j];
                if (((lp->art[k] & 1) && lp->nstat[j] == NS_L) || ((lp->art[k] & 2) && lp->nstat[j] == NS_U)) active = 1;
            }
            free(lp->art); lp->art = NULL;
            if (st == OLP_OPTIMAL && !active) goto done;
            sanitize(lp); refresh_beta(lp); refresh_d(lp);
        } else { free(lp->art); lp->art = NULL; sanitize(lp); refresh_beta(lp); }
        st = primal_simplex(lp);
    } else if (method != OLP_PRIMAL) {
        st = dual_simplex(lp);
        if (st == OLP_UNDEFINED) st = primal_simplex(lp);
    } else
        st = primal_simplex(lp);
done:;
    /* objective from the structural values */
    double z = lp->c[0];
    for (int j = 0; j < lp->N; j++) z += lp->c[j + 1] * olp_col_prim(lp, j + 1);
    lp->obj = z;
    lp->status = st;
    return st;
}

double olp_obj_val(const olp *lp) { return lp->obj; }

static double var_prim(const olp *lp, int k)
{ return lp->posB[k] >= 0 ? lp->beta[lp->posB[k]] : lp->xN[lp->posN[k]]; }
static double var_dual(const olp *lp, int k)
{ return lp->posB[k] >= 0 ? 0.0 : lp->d[lp->posN[k]]; }

double olp_row_prim(const olp *lp, int row) { return var_prim(lp, row - 1); }
double olp_col_prim(const olp *lp, int col) { return var_prim(lp, lp->M + col - 1); }
double olp_row_dual(const olp *lp, int row) { return var_dual(lp, row - 1); }
double olp_col_dual(const olp *lp, int col) { return var_dual(lp, lp->M + col - 1); }
