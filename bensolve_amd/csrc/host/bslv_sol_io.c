/* bslv_sol_io.c -- result files in the reference's on-disk formats.
 *
 * Restates poly_output (bslv_algs.c:50-144) over the engine's dump getters:
 *   <base>_img_p.sol / _img_d.sol   poly__vrtx2file  (bslv_poly.c:341-360): "1 " point | "0 " direction, then "%.14g" coords
 *   <base>_adj_p.sol / _adj_d.sol   poly__adj2file   (:382-397): row i = compacted ids of the neighbours of vertex i
 *   <base>_inc_p.sol / _inc_d.sol   poly__inc2file   (:399-414): row f = compacted ids of the vertices on facet f
 * with the transforms applied before writing: sign flips for max problems (poly_trans_primal,
 * bslv_algs.c:223-231, c_q > 0 case), poly_chop |x| < 1e-10 -> 0 (:186-208), directions scaled to
 * inf-norm 1 (poly_normalize_dir :244-279).  Live slots are compacted in ascending slot order
 * (poly__initialise_permutation :314-330).  Lists are written in ascending id order (the reference
 * writes them in list order, which depends on the cut history). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "../../../include/bslv_hip.h"

static void chop_norm(double *x, int d, int ideal)
{
    for (int k = 0; k < d; k++) if (fabs(x[k]) < 1e-10) x[k] = 0.0;
    if (ideal) {
        double mx = 0;
        for (int k = 0; k < d; k++) if (fabs(x[k]) > mx) mx = fabs(x[k]);
        for (int k = 0; k < d; k++) x[k] = mx > 1e-9 ? x[k] / mx : 0.0;
    }
}

static int write_img(const char *path, int n, int d, const unsigned char *used, const unsigned char *ideal, double *X)
{
    FILE *f = fopen(path, "w");
    if (!f) return 1;
    for (int i = 0; i < n; i++) {
        if (!used[i]) continue;
        fprintf(f, "%d", 1 - (int)ideal[i]);
        for (int k = 0; k < d; k++) fprintf(f, " %.14g", X[(size_t)i * d + k]);
        fprintf(f, "\n");
    }
    fclose(f);
    return 0;
}

static int cmp_int(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }

/* rows[r] = sorted list of `other` ids for each live `row` id; pairs given as (a,b) with a on the row side when !swap */
static int write_lists(const char *path, int nrow, const int *rowmap, const int *colmap, long npairs, const int *pairs, int swap, int both)
{
    int *cnt = (int *)calloc(nrow + 1, sizeof(int));
    for (long k = 0; k < npairs; k++) {
        int a = pairs[2 * k + (swap ? 1 : 0)], b = pairs[2 * k + (swap ? 0 : 1)];
        if (rowmap[a] >= 0 && colmap[b] >= 0) cnt[rowmap[a]]++;
        if (both && rowmap[b] >= 0 && colmap[a] >= 0) cnt[rowmap[b]]++;
    }
    long *off = (long *)calloc(nrow + 2, sizeof(long));
    for (int r = 0; r < nrow; r++) off[r + 1] = off[r] + cnt[r];
    int *buf = (int *)malloc((off[nrow] + 1) * sizeof(int));
    memset(cnt, 0, (nrow + 1) * sizeof(int));
    for (long k = 0; k < npairs; k++) {
        int a = pairs[2 * k + (swap ? 1 : 0)], b = pairs[2 * k + (swap ? 0 : 1)];
        if (rowmap[a] >= 0 && colmap[b] >= 0) buf[off[rowmap[a]] + cnt[rowmap[a]]++] = colmap[b];
        if (both && rowmap[b] >= 0 && colmap[a] >= 0) buf[off[rowmap[b]] + cnt[rowmap[b]]++] = colmap[a];
    }
    FILE *f = fopen(path, "w");
    if (!f) { free(cnt); free(off); free(buf); return 1; }
    for (int r = 0; r < nrow; r++) {
        qsort(buf + off[r], cnt[r], sizeof(int), cmp_int);
        for (int k = 0; k < cnt[r]; k++) fprintf(f, k ? " %d" : "%d", buf[off[r] + k]);
        fprintf(f, "\n");
    }
    fclose(f);
    free(cnt); free(off); free(buf);
    return 0;
}

/* optdir: 1 min, -1 max (then y -> -y on the primal side and y*_q -> -y*_q on the dual side).
 * counts (may be NULL): [0] primal points, [1] primal directions, [2] dual points, [3] dual directions */
int bslv_sol_write(bslv_poly *poly, const char *base, const char *suffix, int optdir, long *counts)
{
    return bslv_sol_write2(poly, base, suffix, optdir == -1, optdir == -1, counts);
}

/* negate_primal / negate_dual_last: the sign changes of poly_trans_primal (bslv_algs.c:221-229) for max problems and c_q < 0 */
int bslv_sol_write2(bslv_poly *poly, const char *base, const char *suffix, int negate_primal, int negate_dual_last, long *counts)
{
    return bslv_sol_write3(poly, base, suffix, 0, negate_primal, negate_dual_last, counts);
}

/* swap != 0: the polyhedron engine holds the LOWER image on its primal side and the upper image on its dual side (the dual
 * algorithm, phase2_dual: poly_output(..., SWAP, ...), bslv_algs.c:1566-1573 with poly_trans_dual :232-240): the "_p" files
 * are written from the dual side and the "_d" files from the primal side.  negate_upper: y -> -y on the upper image;
 * negate_lower_last: y*_q -> -y*_q on the lower image. */
int bslv_sol_write3(bslv_poly *poly, const char *base, const char *suffix, int swap, int negate_primal, int negate_dual_last, long *counts)
{
    const int d = bslv_poly_dim(poly), nv = bslv_poly_nprimal(poly), nf = bslv_poly_ndual(poly);
    int rc = bslv_poly_dual_adjacency(poly);                    /* bslv_algs.c:1144 */
    if (rc) return rc;
    unsigned char *pu = (unsigned char *)malloc(nv + 1), *pi = (unsigned char *)malloc(nv + 1);
    unsigned char *du = (unsigned char *)malloc(nf + 1), *di = (unsigned char *)malloc(nf + 1);
    double *X = (double *)malloc((size_t)(nv + 1) * d * sizeof(double)), *Y = (double *)malloc((size_t)(nf + 1) * d * sizeof(double));
    if ((rc = bslv_poly_get_primal(poly, pu, pi, NULL, X)) || (rc = bslv_poly_get_dual(poly, du, di, Y))) return rc;
    long ne = bslv_poly_nedges(poly), ni = bslv_poly_ninc(poly), nde = bslv_poly_ndual_edges(poly);
    int *E = (int *)malloc((2 * ne + 2) * sizeof(int)), *I = (int *)malloc((2 * ni + 2) * sizeof(int)), *DE = (int *)malloc((2 * nde + 2) * sizeof(int));
    if ((rc = bslv_poly_get_edges(poly, E)) || (rc = bslv_poly_get_inc(poly, I)) || (rc = bslv_poly_get_dual_edges(poly, DE))) return rc;
    int *pmap = (int *)malloc((nv + 1) * sizeof(int)), *dmap = (int *)malloc((nf + 1) * sizeof(int));
    int np = 0, nd = 0;
    long cnt[4] = {0, 0, 0, 0};
    for (int i = 0; i < nv; i++) { pmap[i] = pu[i] ? np++ : -1; if (pu[i]) cnt[pi[i] ? 1 : 0]++; }
    for (int f = 0; f < nf; f++) { dmap[f] = du[f] ? nd++ : -1; if (du[f]) cnt[di[f] ? 3 : 2]++; }
    for (int i = 0; i < nv; i++) {
        if (!pu[i]) continue;
        if (!swap && negate_primal) for (int k = 0; k < d; k++) X[(size_t)i * d + k] = -X[(size_t)i * d + k];
        if (swap && negate_dual_last) X[(size_t)i * d + d - 1] = -X[(size_t)i * d + d - 1];
        chop_norm(X + (size_t)i * d, d, pi[i]);
    }
    for (int f = 0; f < nf; f++) {
        if (!du[f]) continue;
        if (!swap && negate_dual_last) Y[(size_t)f * d + d - 1] = -Y[(size_t)f * d + d - 1];
        if (swap && negate_primal) for (int k = 0; k < d; k++) Y[(size_t)f * d + k] = -Y[(size_t)f * d + k];
        chop_norm(Y + (size_t)f * d, d, di[f]);
    }
    char path[1024];
    int err = 0;
#define P(sfx) (snprintf(path, sizeof path, "%s%s%s", base, sfx, suffix), path)
    if (!swap) {
        err |= write_img(P("_img_p"), nv, d, pu, pi, X);
        err |= write_img(P("_img_d"), nf, d, du, di, Y);
        err |= write_lists(P("_adj_p"), np, pmap, pmap, ne, E, 0, 1);
        err |= write_lists(P("_adj_d"), nd, dmap, dmap, nde, DE, 0, 1);
        err |= write_lists(P("_inc_p"), nd, dmap, pmap, ni, I, 1, 0);     /* row = facet, entries = vertices on it */
        err |= write_lists(P("_inc_d"), np, pmap, dmap, ni, I, 0, 0);     /* row = vertex, entries = facets through it */
    } else {
        err |= write_img(P("_img_p"), nf, d, du, di, Y);
        err |= write_img(P("_img_d"), nv, d, pu, pi, X);
        err |= write_lists(P("_adj_p"), nd, dmap, dmap, nde, DE, 0, 1);
        err |= write_lists(P("_adj_d"), np, pmap, pmap, ne, E, 0, 1);
        err |= write_lists(P("_inc_p"), np, pmap, dmap, ni, I, 0, 0);
        err |= write_lists(P("_inc_d"), nd, dmap, pmap, ni, I, 1, 0);
        { long t0 = cnt[0], t1 = cnt[1]; cnt[0] = cnt[2]; cnt[1] = cnt[3]; cnt[2] = t0; cnt[3] = t1; }
    }
#undef P
    if (counts) memcpy(counts, cnt, sizeof cnt);
    free(pu); free(pi); free(du); free(di); free(X); free(Y); free(E); free(I); free(DE); free(pmap); free(dmap);
    return err ? BSLV_E_ARG : 0;
}

int bslv_sol_write_preimages(bslv_benson *eng, const char *base, const char *suffix, int m, int n, int optdir, int c_dir)
{
    bslv_poly *poly = bslv_benson_poly(eng);
    const int d = bslv_poly_dim(poly), nv = bslv_poly_nprimal(poly), nf = bslv_poly_ndual(poly);
    unsigned char *pu = (unsigned char *)malloc(nv + 1), *pi = (unsigned char *)malloc(nv + 1);
    unsigned char *du = (unsigned char *)malloc(nf + 1), *di = (unsigned char *)malloc(nf + 1);
    double *X = (double *)malloc((size_t)(nv + 1) * d * sizeof(double)), *Y = (double *)malloc((size_t)(nf + 1) * d * sizeof(double));
    double *row = (double *)malloc((size_t)(m + n + d + 1) * sizeof(double));
    FILE *f = NULL;
    int rc = BSLV_E_NOMEM;
    char path[1024];
    if (!pu || !pi || !du || !di || !X || !Y || !row) goto out;           /* (one way out: every buffer and the open file are released there) */
    if ((rc = bslv_poly_get_primal(poly, pu, pi, NULL, X)) || (rc = bslv_poly_get_dual(poly, du, di, Y))) goto out;
    rc = BSLV_E_ARG;
    snprintf(path, sizeof path, "%s_pre_img_p%s", base, suffix);
    if (!(f = fopen(path, "w"))) goto out;
    for (int i = 0; i < nv; i++) {
        if (!pu[i]) continue;
        if (bslv_benson_preimage_p(eng, i, row)) for (int k = 0; k < n; k++) row[k] = 0.0;
        for (int k = 0; k < n; k++) fprintf(f, k ? " %.14g" : "%.14g", row[k]);
        fprintf(f, "\n");
    }
    fclose(f);
    snprintf(path, sizeof path, "%s_pre_img_d%s", base, suffix);
    if (!(f = fopen(path, "w"))) goto out;
    for (int k2 = 0; k2 < nf; k2++) {
        if (!du[k2]) continue;
        if (di[k2] || bslv_benson_preimage_d(eng, k2, row)) for (int k = 0; k < m + d; k++) row[k] = 0.0;
        else { for (int k = 0; k < m; k++) row[k] *= optdir; for (int k = 0; k < d; k++) row[m + k] *= c_dir; }
        for (int k = 0; k < m + d; k++) fprintf(f, k ? " %.14g" : "%.14g", row[k]);
        fprintf(f, "\n");
    }
    fclose(f);
    f = NULL;
    rc = 0;
out:
    if (f) fclose(f);
    free(pu); free(pi); free(du); free(di); free(X); free(Y); free(row);
    return rc;
}

/* option -s with the dual algorithm (poly_output(..., SWAP, ...) with PRE_IMG_ON, bslv_algs.c:1566-1573): the polyhedron holds the
 * lower image on its primal side.  <base>_pre_img_p: x of every element of the upper image (dual slots, ascending);
 * <base>_pre_img_d: (u, w) of every vertex of the lower image (primal elements, ascending), zeros for its directions (:1540-1546). */
int bslv_sol_write_preimages_dual(bslv_poly *poly, const char *base, const char *suffix, int m, int n, int optdir, int c_dir)
{
    const int d = bslv_poly_dim(poly), nv = bslv_poly_nprimal(poly), nf = bslv_poly_ndual(poly);
    unsigned char *pu = (unsigned char *)malloc(nv + 1), *pi = (unsigned char *)malloc(nv + 1);
    unsigned char *du = (unsigned char *)malloc(nf + 1), *di = (unsigned char *)malloc(nf + 1);
    double *row = (double *)malloc((size_t)(m + n + d + 1) * sizeof(double));
    FILE *f = NULL;
    int rc = BSLV_E_NOMEM;
    char path[1024];
    if (!pu || !pi || !du || !di || !row) goto out;
    if ((rc = bslv_poly_get_primal(poly, pu, pi, NULL, NULL)) || (rc = bslv_poly_get_dual(poly, du, di, NULL))) goto out;
    rc = BSLV_E_ARG;
    snprintf(path, sizeof path, "%s_pre_img_p%s", base, suffix);
    if (!(f = fopen(path, "w"))) goto out;
    for (int k2 = 0; k2 < nf; k2++) {
        if (!du[k2]) continue;
        if (bslv_dual_preimage_x(poly, k2, row)) for (int k = 0; k < n; k++) row[k] = 0.0;
        for (int k = 0; k < n; k++) fprintf(f, k ? " %.14g" : "%.14g", row[k]);
        fprintf(f, "\n");
    }
    fclose(f);
    snprintf(path, sizeof path, "%s_pre_img_d%s", base, suffix);
    if (!(f = fopen(path, "w"))) goto out;
    for (int i = 0; i < nv; i++) {
        if (!pu[i]) continue;
        if (pi[i] || bslv_dual_preimage_uw(poly, i, row)) for (int k = 0; k < m + d; k++) row[k] = 0.0;
        else { for (int k = 0; k < m; k++) row[k] *= optdir; for (int k = 0; k < d; k++) row[m + k] *= c_dir; }
        for (int k = 0; k < m + d; k++) fprintf(f, k ? " %.14g" : "%.14g", row[k]);
        fprintf(f, "\n");
    }
    fclose(f);
    f = NULL;
    rc = 0;
out:
    if (f) fclose(f);
    free(pu); free(pi); free(du); free(di); free(row);
    return rc;
}
