/*
 * oracle/poly_dd.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).  See poly_dd.h.
 *
 * One cut (= poly__add_vrtx on an initialised polyhedron, bslv_poly.c:104-151) is restated as:
 *   pass 1  classify every live primal element against the new halfspace hp.y >= alpha
 *           (alpha = 0 for directions): PLUS  s > a+EPS, ZERO a-EPS < s <= a+EPS, MINUS otherwise.
 *           These are the three bands of poly__cut (bslv_poly.c:573,596,666-675; POLY_EPS 1e-9).
 *           The reference's projection sub-band (:666-674: a neighbour with s in (a + 1e-2 EPS, a + EPS]
 *           is moved onto the hyperplane before it is treated as lying on it; coordinates move by < 1e-9)
 *           is restated behind opoly_set_snap(p, 1) and OFF by default, as it is in the HIP engine
 *           (bslv_poly_set_snap: the engine's rounds of independent cuts classify ahead of the cuts, a
 *           moved element would invalidate that, so the band costs the batching -- DESIGN.md section 8).
 *           With the switch on, the oracle is pinned against the compiled bslv_poly.c on crafted cases
 *           (tests/test_oracle_poly.py::test_snap_band_*), and the engine against both
 *           (tests/test_poly_gpu.py::test_snap_band_*).
 *           No MINUS element -> the cut is redundant, its dual slot is left unused (:130-136).
 *   pass 2  edges: a MINUS-PLUS edge creates a new vertex on the hyperplane (:597-627), which
 *           inherits inc(minus) & inc(plus) plus the new facet (:634-665) and is adjacent to the
 *           PLUS end (:628-633); edges with a MINUS end disappear; ZERO-ZERO edges are dropped
 *           and re-found by pass 4 (the reference relocates ZERO vertices to fresh slots that
 *           keep only their PLUS neighbours, :573-588,633).
 *   pass 3  a ZERO element joins the new facet and keeps the facets it shares with a PLUS
 *           neighbour (:634-652 with smpl==0).
 *   pass 4  all pairs of elements of the new facet go through the combinatorial adjacency test
 *           edge_test (:138-143, 467-512).
 * The order in which new slots and edges are produced (edge order, then pair order) is part of
 * this restatement's definition: the HIP engine reproduces it with prefix sums.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#include "poly_dd.h"

#define POLY_EPS 1e-9

typedef struct { int *a; int n, cap; } ivec;

static void iv_push(ivec *v, int x)
{
    if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 8; v->a = (int *)realloc(v->a, v->cap * sizeof(int)); }
    v->a[v->n++] = x;
}

struct opoly {
    int d, v2h;
    double *c;
    /* primal */
    int nv, capv;
    double *X;
    unsigned char *used, *ideal, *sltn;
    ivec *inc;
    /* dual */
    int nf, capf;
    double *Y, *hp;
    unsigned char *fused, *fideal;
    /* edges */
    int *E; long ne, cape;
    int *DE; long nde, capde;
    int initialised;
    ivec queue;
    long pair_tests, new_vertices;
    int snap;                             /* opoly_set_snap: the projection sub-band of poly__cut (bslv_poly.c:666-674) */
    long snapped;
};

/* ---- vertex -> halfspace maps ---- */
static void v2h(const opoly *p, const double *v, int is_dir, double *hp)
{
    int d = p->d;
    switch (p->v2h) {
    case OPOLY_CONE_POLAR:                          /* bslv_poly.c:30-39 */
        for (int j = 0; j < d; j++) hp[j] = v[j];
        hp[d] = is_dir ? 0.0 : -1.0;
        break;
    case OPOLY_LOWER2UPPER:                         /* bslv_algs.c:287-305 */
        if (is_dir) { for (int j = 0; j < d; j++) hp[j] = 0.0; hp[d] = -1.0; }
        else {
            hp[d - 1] = 1.0;
            for (int j = 0; j < d - 1; j++) { hp[j] = v[j]; hp[d - 1] -= p->c[j] * hp[j]; }
            hp[d] = v[d - 1];
        }
        break;
    default:                                        /* bslv_algs.c:307-313 */
        hp[d - 1] = is_dir ? 0.0 : -1.0;
        for (int j = 0; j < d - 1; j++) hp[j] = v[j] - v[d - 1] * p->c[j];
        hp[d] = -v[d - 1];
        break;
    }
}

static int new_primal(opoly *p)
{
    if (p->nv == p->capv) {
        int cap = p->capv ? 2 * p->capv : 64;
        p->X = (double *)realloc(p->X, (size_t)cap * p->d * sizeof(double));
        p->used = (unsigned char *)realloc(p->used, cap);
        p->ideal = (unsigned char *)realloc(p->ideal, cap);
        p->sltn = (unsigned char *)realloc(p->sltn, cap);
        p->inc = (ivec *)realloc(p->inc, cap * sizeof(ivec));
        memset(p->inc + p->capv, 0, (cap - p->capv) * sizeof(ivec));
        p->capv = cap;
    }
    int i = p->nv++;
    p->used[i] = 1; p->ideal[i] = 0; p->sltn[i] = 0; p->inc[i].n = 0;
    return i;
}

static int new_dual(opoly *p, const double *val, int ideal)
{
    if (p->nf == p->capf) {
        int cap = p->capf ? 2 * p->capf : 64;
        p->Y = (double *)realloc(p->Y, (size_t)cap * p->d * sizeof(double));
        p->hp = (double *)realloc(p->hp, (size_t)cap * (p->d + 1) * sizeof(double));
        p->fused = (unsigned char *)realloc(p->fused, cap);
        p->fideal = (unsigned char *)realloc(p->fideal, cap);
        p->capf = cap;
    }
    int f = p->nf++;
    memcpy(p->Y + (size_t)f * p->d, val, p->d * sizeof(double));
    p->fused[f] = 1; p->fideal[f] = ideal ? 1 : 0;
    v2h(p, val, ideal, p->hp + (size_t)f * (p->d + 1));
    return f;
}

static void push_edge(opoly *p, int a, int b)
{
    if (p->ne == p->cape) { p->cape = p->cape ? 2 * p->cape : 256; p->E = (int *)realloc(p->E, 2 * p->cape * sizeof(int)); }
    p->E[2 * p->ne] = a; p->E[2 * p->ne + 1] = b; p->ne++;
}

opoly *opoly_create(int dim, int v2h_kind, const double *c)
{
    opoly *p = (opoly *)calloc(1, sizeof(opoly));
    p->d = dim; p->v2h = v2h_kind;
    p->c = (double *)calloc(dim, sizeof(double));
    if (c) memcpy(p->c, c, dim * sizeof(double));
    /* dual slot 0: the "facet at infinity", ideal point (0,..,0,-1) (bslv_poly.c:83-92) */
    double *z = (double *)calloc(dim, sizeof(double));
    z[dim - 1] = -1.0;
    new_dual(p, z, 1);
    free(z);
    return p;
}

void opoly_set_snap(opoly *p, int on) { p->snap = on != 0; }
long opoly_snapped(const opoly *p) { return p->snapped; }

void opoly_free(opoly *p)
{
    if (!p) return;
    for (int i = 0; i < p->capv; i++) free(p->inc[i].a);
    free(p->inc); free(p->X); free(p->used); free(p->ideal); free(p->sltn);
    free(p->Y); free(p->hp); free(p->fused); free(p->fideal);
    free(p->E); free(p->DE); free(p->queue.a); free(p->c);
    free(p);
}

void opoly_dual0_apex(opoly *p)
{
    p->fideal[0] = 0;
    p->Y[p->d - 1] = 0.0;
    v2h(p, p->Y, 0, p->hp);
}

/* sorted-list helpers */
static int isect(const ivec *a, const ivec *b, int *out)
{
    int i = 0, j = 0, n = 0;
    while (i < a->n && j < b->n) {
        if (a->a[i] < b->a[j]) i++;
        else if (a->a[i] > b->a[j]) j++;
        else { out[n++] = a->a[i]; i++; j++; }
    }
    return n;
}
static int subset(const int *m, int nm, const ivec *b)
{
    int j = 0;
    for (int i = 0; i < nm; i++) {
        while (j < b->n && b->a[j] < m[i]) j++;
        if (j == b->n || b->a[j] != m[i]) return 0;
        j++;
    }
    return 1;
}

static double dotf(const double *h, const double *x, int d)
{
    double s = 0.0;
    for (int j = 0; j < d; j++) s = fma(h[j], x[j], s);
    return s;
}

/* combinatorial adjacency test (edge_test, bslv_poly.c:467-512) with the candidate pool given */
static int adjacent(opoly *p, int v1, int v2, const int *pool, int npool, int *scratch)
{
    p->pair_tests++;
    int nm = isect(&p->inc[v1], &p->inc[v2], scratch);
    if (p->d == 1) return 1;
    if (nm < p->d - 1) return 0;
    for (int k = 0; k < npool; k++) {
        int w = pool[k];
        if (w == v1 || w == v2) continue;
        if (subset(scratch, nm, &p->inc[w])) return 0;
    }
    return 1;
}

static int do_cut(opoly *p, int f)
{
    const int d = p->d, nv0 = p->nv;
    const double *hp = p->hp + (size_t)f * (d + 1);
    const double alpha = hp[d];
    signed char *cls = (signed char *)malloc(nv0 ? nv0 : 1);
    int nminus = 0, maxinc = 0;
    for (int i = 0; i < nv0; i++) {
        if (!p->used[i]) { cls[i] = 2; continue; }
        double s = dotf(hp, p->X + (size_t)i * d, d);
        double a = p->ideal[i] ? 0.0 : alpha;
        cls[i] = (s > a + POLY_EPS) ? 1 : (s > a - POLY_EPS ? 0 : -1);
        if (cls[i] < 0) nminus++;
        if (p->inc[i].n > maxinc) maxinc = p->inc[i].n;
    }
    if (!nminus) { p->fused[f] = 0; free(cls); return 1; }
    if (p->snap)                          /* only a cut that removes something walks the polyhedron (bslv_poly.c:130-131) */
        for (int i = 0; i < nv0; i++) {
            if (cls[i] != 0) continue;
            double *x = p->X + (size_t)i * d;
            double s = dotf(hp, x, d), a = p->ideal[i] ? 0.0 : alpha;
            if (!(s > a + 1.0e-2 * POLY_EPS)) continue;
            double mu = s - a, nn = 0.0;      /* bslv_poly.c:666-674: x -= (s - a) hp / |hp|^2, then the element lies on the hyperplane */
            for (int j = 0; j < d; j++) nn += hp[j] * hp[j];
            mu /= nn;
            for (int j = 0; j < d; j++) x[j] -= mu * hp[j];
            p->snapped++;
        }

    int *scratch = (int *)malloc((maxinc + 2) * sizeof(int));
    /* pass 2: edges */
    long ne0 = p->ne;
    int *oldE = p->E;
    p->E = NULL; p->ne = 0; p->cape = 0;
    ivec cross = {0};                     /* (minus, plus) pairs */
    for (long e = 0; e < ne0; e++) {
        int a = oldE[2 * e], b = oldE[2 * e + 1], ca = cls[a], cb = cls[b];
        if (ca == -1 && cb == 1) { iv_push(&cross, a); iv_push(&cross, b); }
        else if (ca == 1 && cb == -1) { iv_push(&cross, b); iv_push(&cross, a); }
        else if (ca >= 0 && cb >= 0 && !(ca == 0 && cb == 0)) push_edge(p, a, b);
    }
    /* pass 3 (marks): a ZERO element keeps the facets it shares with a PLUS neighbour */
    unsigned char **keep = (unsigned char **)calloc(nv0 ? nv0 : 1, sizeof(unsigned char *));
    for (long e = 0; e < ne0; e++) {
        int a = oldE[2 * e], b = oldE[2 * e + 1];
        int z = -1, pl = -1;
        if (cls[a] == 0 && cls[b] == 1) { z = a; pl = b; }
        else if (cls[a] == 1 && cls[b] == 0) { z = b; pl = a; }
        if (z < 0) continue;
        if (!keep[z]) keep[z] = (unsigned char *)calloc(p->inc[z].n + 1, 1);
        int i = 0, j = 0;
        const ivec *A = &p->inc[z], *B = &p->inc[pl];
        while (i < A->n && j < B->n) {
            if (A->a[i] < B->a[j]) i++;
            else if (A->a[i] > B->a[j]) j++;
            else { keep[z][i] = 1; i++; j++; }
        }
    }
    free(oldE);
    /* new vertices, in crossing-edge order (bslv_poly.c:597-627) */
    int ncross = cross.n / 2;
    for (int k = 0; k < ncross; k++) {
        int mi = cross.a[2 * k], pl = cross.a[2 * k + 1];
        int w = new_primal(p);
        const double *xm = p->X + (size_t)mi * d, *xp = p->X + (size_t)pl * d;
        double *xw = p->X + (size_t)w * d;
        int im = p->ideal[mi], ip = p->ideal[pl];
        const double *base = ip ? xm : xp, *dirv = ip ? xp : xm;
        double hb = 0.0, hd = 0.0, a2 = alpha;
        if (ip && im) {                  /* direction/direction: new direction */
            a2 = 0.0;
            for (int j = 0; j < d; j++) { double dj = xp[j] - xm[j]; hd = fma(hp[j], dj, hd); }
            hb = dotf(hp, base, d);
            double mu = (a2 - hb) / hd;
            for (int j = 0; j < d; j++) xw[j] = fma(mu, xp[j] - xm[j], base[j]);
            p->ideal[w] = 1;
        } else if (!ip && !im) {         /* point/point */
            for (int j = 0; j < d; j++) { double dj = xm[j] - xp[j]; hd = fma(hp[j], dj, hd); }
            hb = dotf(hp, base, d);
            double mu = (a2 - hb) / hd;
            for (int j = 0; j < d; j++) xw[j] = fma(mu, xm[j] - xp[j], base[j]);
        } else {                         /* point + direction */
            hd = dotf(hp, dirv, d);
            hb = dotf(hp, base, d);
            double mu = (a2 - hb) / hd;
            for (int j = 0; j < d; j++) xw[j] = fma(mu, dirv[j], base[j]);
        }
        int nm = isect(&p->inc[mi], &p->inc[pl], scratch);
        for (int j = 0; j < nm; j++) iv_push(&p->inc[w], scratch[j]);
        iv_push(&p->inc[w], f);
        p->new_vertices++;
    }
    /* pass 3 (rebuild) + member list: ZERO elements in slot order, then the new vertices */
    ivec mem = {0};
    for (int i = 0; i < nv0; i++) {
        if (cls[i] != 0) continue;
        ivec *L = &p->inc[i];
        int n = 0;
        for (int j = 0; j < L->n; j++) if (keep[i] && keep[i][j]) L->a[n++] = L->a[j];
        L->n = n;
        iv_push(L, f);
        iv_push(&mem, i);
    }
    for (int k = 0; k < ncross; k++) iv_push(&mem, nv0 + k);
    for (int i = 0; i < nv0; i++) { free(keep[i]); if (cls[i] == -1) p->used[i] = 0; }
    free(keep);
    /* new-vertex edges, then pass 4 */
    for (int k = 0; k < ncross; k++) push_edge(p, nv0 + k, cross.a[2 * k + 1]);
    for (int i = 0; i < mem.n; i++)
        for (int j = i + 1; j < mem.n; j++)
            if (adjacent(p, mem.a[i], mem.a[j], mem.a, mem.n, scratch)) push_edge(p, mem.a[i], mem.a[j]);
    free(mem.a); free(cross.a); free(scratch); free(cls);
    return 0;
}

int opoly_add(opoly *p, const double *val, int ideal)
{
    int f = new_dual(p, val, ideal);
    if (!p->initialised) { iv_push(&p->queue, f); return 0; }
    return do_cut(p, f);
}

/* modified Gram-Schmidt step (bslv__normalise, bslv_poly.c:1030-1060) */
static double gs_step(const double *x, double *H, double *R, int k, int n)
{
    double nrm_in = 0, scl;
    for (int l = 0; l < n; l++) nrm_in += x[l] * x[l];
    nrm_in = sqrt(nrm_in);
    double *h = H + (size_t)k * n;
    memcpy(h, x, n * sizeof(double));
    for (int j = 0; j < k; j++) {
        double s = 0;
        for (int l = 0; l < n; l++) s += H[(size_t)j * n + l] * h[l];
        for (int l = 0; l < n; l++) h[l] -= s * H[(size_t)j * n + l];
    }
    scl = 0;
    for (int l = 0; l < n; l++) scl += h[l] * h[l];
    scl = sqrt(scl);
    if (scl < 1.0e-6) return 0;
    for (int l = 0; l < n; l++) h[l] /= scl;
    for (int j = 0; j <= k; j++) {
        double s = 0;
        for (int l = 0; l < n; l++) s += H[(size_t)j * n + l] * x[l];
        R[k * (k + 1) / 2 + j] = s;
    }
    return scl / nrm_in;
}

int opoly_init(opoly *p)
{
    const int d = p->d;
    int qn = p->queue.n;
    if (qn < d) return 1;
    double *hpq = (double *)malloc((size_t)qn * (d + 1) * sizeof(double));
    for (int k = 0; k < qn; k++) memcpy(hpq + (size_t)k * (d + 1), p->hp + (size_t)p->queue.a[k] * (d + 1), (d + 1) * sizeof(double));
    double *H = (double *)calloc((size_t)d * d, sizeof(double));
    double *R = (double *)calloc((size_t)d * (d + 1) / 2, sizeof(double));
    double *alph = (double *)calloc(d, sizeof(double));
    int *perm = (int *)calloc(d + 1, sizeof(int));
    int g = 0;
    perm[0] = 0;
    /* greedy choice of d independent halfspaces by largest relative residual (bslv_poly.c:167-185) */
    while (g < d) {
        double best = 0; int bi = -1;
        for (int k = 0; k < qn; k++) {
            double s = gs_step(hpq + (size_t)k * (d + 1), H, R, g, d);
            if (best < s) { best = s; bi = k; }
        }
        if (best < 1.0e-10) { free(hpq); free(H); free(R); free(alph); free(perm); return 1; }
        gs_step(hpq + (size_t)bi * (d + 1), H, R, g, d);
        alph[g] = hpq[(size_t)bi * (d + 1) + d];
        perm[++g] = p->queue.a[bi];
        qn--;
        memcpy(hpq + (size_t)bi * (d + 1), hpq + (size_t)qn * (d + 1), (d + 1) * sizeof(double));
        p->queue.a[bi] = p->queue.a[qn];
    }
    /* initial simplex cone (poly__poly_initialise, bslv_poly.c:711-787): x_k = sum_j R[k][j] H_j,
     * vertex y solves x_k . y = alph_k ; direction k solves x_i . dir = delta_ik */
#define RM(k, j) R[(k) * ((k) + 1) / 2 + (j)]
    double *t = (double *)calloc(d, sizeof(double));
    int v0 = new_primal(p);
    for (int k = 0; k < d; k++) {
        double s = alph[k];
        for (int j = 0; j < k; j++) s -= RM(k, j) * t[j];
        t[k] = s / RM(k, k);
    }
    for (int j = 0; j < d; j++) {
        double s = 0;
        for (int l = 0; l < d; l++) s += H[(size_t)l * d + j] * t[l];
        p->X[(size_t)v0 * d + j] = s;
    }
    for (int k = 0; k < d; k++) {
        int w = new_primal(p);
        p->ideal[w] = 1;
        for (int i = 0; i < d; i++) t[i] = 0;
        t[k] = 1.0 / RM(k, k);
        for (int i = k + 1; i < d; i++) {
            double s = 0;
            for (int j = k; j < i; j++) s += RM(i, j) * t[j];
            t[i] = -s / RM(i, i);
        }
        for (int j = 0; j < d; j++) {
            double s = 0;
            for (int l = 0; l < d; l++) s += H[(size_t)l * d + j] * t[l];
            p->X[(size_t)w * d + j] = s;
        }
    }
#undef RM
    free(t);
    /* incidence: primal j lies on facet perm[k] for every k != j ; complete adjacency (:769-780) */
    for (int j = 0; j <= d; j++) {
        int *tmp = (int *)malloc((d + 1) * sizeof(int)), n = 0;
        for (int k = 0; k <= d; k++) if (k != j) tmp[n++] = perm[k];
        for (int a = 1; a < n; a++) { int x = tmp[a], b = a - 1; while (b >= 0 && tmp[b] > x) { tmp[b + 1] = tmp[b]; b--; } tmp[b + 1] = x; }
        for (int a = 0; a < n; a++) iv_push(&p->inc[j], tmp[a]);
        free(tmp);
    }
    for (int k = 0; k <= d; k++) for (int j = k + 1; j <= d; j++) push_edge(p, k, j);
    p->initialised = 1;
    /* the halfspaces not chosen are re-added as NEW dual slots; the originals stay unused (:190-197) */
    int rest = qn;
    int *ids = (int *)malloc((rest + 1) * sizeof(int));
    memcpy(ids, p->queue.a, rest * sizeof(int));
    for (int k = 0; k < rest; k++) p->fused[ids[k]] = 0;
    for (int k = 0; k < rest; k++) {
        double *val = (double *)malloc(d * sizeof(double));
        memcpy(val, p->Y + (size_t)ids[k] * d, d * sizeof(double));
        opoly_add(p, val, p->fideal[ids[k]]);
        free(val);
    }
    free(ids); free(hpq); free(H); free(R); free(alph); free(perm);
    p->queue.n = 0;
    return 0;
}

int opoly_next(opoly *p, double *val, int *ideal, int *idx)
{
    for (int i = 0; i < p->nv; i++)
        if (p->used[i] && !p->sltn[i]) {
            memcpy(val, p->X + (size_t)i * p->d, p->d * sizeof(double));
            *ideal = p->ideal[i]; *idx = i;
            return 0;
        }
    return 1;
}

/* measurement variant (not the reference's rule): the NEWEST live slot without the sltn mark */
int opoly_next_newest(opoly *p, double *val, int *ideal, int *idx)
{
    for (int i = p->nv - 1; i >= 0; i--)
        if (p->used[i] && !p->sltn[i]) {
            memcpy(val, p->X + (size_t)i * p->d, p->d * sizeof(double));
            *ideal = p->ideal[i]; *idx = i;
            return 0;
        }
    return 1;
}

void opoly_mark(opoly *p, int idx) { p->sltn[idx] = 1; }

/* a dual slot is live iff it was applied and some live primal element still lies on it
 * (the reference clears dual.used when the facet's vertex list empties, bslv_poly.c:686-687,697-705) */
static void facet_lists(const opoly *p, ivec *verts, unsigned char *live)
{
    for (int f = 0; f < p->nf; f++) { verts[f].a = NULL; verts[f].n = verts[f].cap = 0; live[f] = 0; }
    for (int i = 0; i < p->nv; i++) {
        if (!p->used[i]) continue;
        for (int j = 0; j < p->inc[i].n; j++) { int f = p->inc[i].a[j]; iv_push(&verts[f], i); if (p->fused[f]) live[f] = 1; }
    }
}

void opoly_dual_adjacency(opoly *p)
{
    const int d = p->d;
    ivec *verts = (ivec *)malloc((p->nf + 1) * sizeof(ivec));
    unsigned char *live = (unsigned char *)malloc(p->nf + 1);
    facet_lists(p, verts, live);
    p->nde = 0;
    int maxn = 1;
    for (int f = 0; f < p->nf; f++) if (verts[f].n > maxn) maxn = verts[f].n;
    int *m = (int *)malloc((maxn + 1) * sizeof(int));
    for (int f1 = 0; f1 < p->nf; f1++) {
        if (!live[f1]) continue;
        for (int f2 = f1 + 1; f2 < p->nf; f2++) {
            if (!live[f2]) continue;
            p->pair_tests++;
            int nm = isect(&verts[f1], &verts[f2], m);
            int adj;
            if (d == 1) adj = 1;
            else if (nm < d - 1) adj = 0;
            else {
                adj = 1;
                const ivec *pool = &p->inc[m[0]];   /* facets through the first mutual vertex (:487-491) */
                for (int k = 0; k < pool->n && adj; k++) {
                    int w = pool->a[k];
                    if (w == f1 || w == f2 || !live[w]) continue;
                    if (subset(m, nm, &verts[w])) adj = 0;
                }
            }
            if (adj) {
                if (p->nde == p->capde) { p->capde = p->capde ? 2 * p->capde : 256; p->DE = (int *)realloc(p->DE, 2 * p->capde * sizeof(int)); }
                p->DE[2 * p->nde] = f1; p->DE[2 * p->nde + 1] = f2; p->nde++;
            }
        }
    }
    for (int f = 0; f < p->nf; f++) free(verts[f].a);
    free(verts); free(live); free(m);
}

int opoly_dim(const opoly *p) { return p->d; }
int opoly_nprimal(const opoly *p) { return p->nv; }
int opoly_ndual(const opoly *p) { return p->nf; }
long opoly_nedges(const opoly *p) { return p->ne; }
long opoly_ndual_edges(const opoly *p) { return p->nde; }
long opoly_pair_tests(const opoly *p) { return p->pair_tests; }
long opoly_new_vertices(const opoly *p) { return p->new_vertices; }

long opoly_ninc(const opoly *p)
{
    long n = 0;
    for (int i = 0; i < p->nv; i++) if (p->used[i]) n += p->inc[i].n;
    return n;
}

void opoly_get_primal(const opoly *p, unsigned char *used, unsigned char *ideal, unsigned char *sltn, double *coords)
{
    memcpy(used, p->used, p->nv); memcpy(ideal, p->ideal, p->nv); memcpy(sltn, p->sltn, p->nv);
    memcpy(coords, p->X, (size_t)p->nv * p->d * sizeof(double));
}

void opoly_get_dual(const opoly *p, unsigned char *used, unsigned char *ideal, double *coords)
{
    ivec *verts = (ivec *)malloc((p->nf + 1) * sizeof(ivec));
    unsigned char *live = (unsigned char *)malloc(p->nf + 1);
    if (p->initialised) facet_lists(p, verts, live);
    else for (int f = 0; f < p->nf; f++) { live[f] = p->fused[f]; verts[f].a = NULL; }
    memcpy(used, live, p->nf);
    memcpy(ideal, p->fideal, p->nf);
    memcpy(coords, p->Y, (size_t)p->nf * p->d * sizeof(double));
    for (int f = 0; f < p->nf; f++) free(verts[f].a);
    free(verts); free(live);
}

void opoly_get_edges(const opoly *p, int *ab) { memcpy(ab, p->E, 2 * p->ne * sizeof(int)); }
void opoly_get_dual_edges(const opoly *p, int *ab) { memcpy(ab, p->DE, 2 * p->nde * sizeof(int)); }

void opoly_get_inc(const opoly *p, int *pairs)
{
    long n = 0;
    for (int i = 0; i < p->nv; i++) {
        if (!p->used[i]) continue;
        for (int j = 0; j < p->inc[i].n; j++) { pairs[2 * n] = i; pairs[2 * n + 1] = p->inc[i].a[j]; n++; }
    }
}
