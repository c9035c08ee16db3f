/*
 * oracle/ref_poly_driver.c -- TEST INFRASTRUCTURE ONLY.  OUR code (no reference text).
 *
 * Thin dump driver around the reference's polyhedron engine.  It is compiled together with the
 * UNMODIFIED /root/reference/bslv_poly.c (where it lies) into oracle/_ref/libref_poly.so and
 * exposes the reference's poly__* entry points (bslv_poly.h:90-118) behind the same flat API
 * as oracle/poly_dd.h, so tests can run one cut sequence through both and compare.
 * The reference keeps file-scope state (fnc_dim, bslv_poly.c:28): one object at a time.
 */
#include <stdlib.h>
#include <string.h>
#include "bslv_poly.h"     /* from /root/reference */

typedef struct { poly_args args; int v2h; double *c; } rpoly;

/* our own statement of the two V->H maps that live as static functions in bslv_algs.c:287-313 */
static struct { size_t dim; double *ip; } prm;
static void lower2upper(double *v, int is_dir, double *hp)
{
    size_t d = prm.dim;
    if (is_dir) { for (size_t j = 0; j < d; j++) hp[j] = 0; hp[d] = -1.0; return; }
    hp[d - 1] = 1.0;
    for (size_t j = 0; j + 1 < d; j++) { hp[j] = v[j]; hp[d - 1] -= prm.ip[j] * hp[j]; }
    hp[d] = v[d - 1];
}
static void upper2lower(double *v, int is_dir, double *hp)
{
    size_t d = prm.dim;
    hp[d - 1] = is_dir ? 0 : -1.0;
    for (size_t j = 0; j + 1 < d; j++) hp[j] = v[j] - v[d - 1] * prm.ip[j];
    hp[d] = -v[d - 1];
}

void *rpoly_create(int dim, int v2h, const double *c)
{
    rpoly *r = (rpoly *)calloc(1, sizeof(rpoly));
    r->v2h = v2h;
    r->c = (double *)calloc(dim, sizeof(double));
    if (c) memcpy(r->c, c, dim * sizeof(double));
    poly__set_default_args(&r->args, dim);
    r->args.eps = 1e-9;
    if (v2h == 1) r->args.dualV2primalH = &lower2upper;
    if (v2h == 2) r->args.dualV2primalH = &upper2lower;
    prm.dim = dim; prm.ip = r->c;
    poly__initialise(&r->args);
    return r;
}
void rpoly_free(void *h) { rpoly *r = (rpoly *)h; poly__kill(&r->args); free(r->c); free(r); }
void rpoly_dual0_apex(void *h)
{
    rpoly *r = (rpoly *)h;
    UNST_BT(r->args.dual.ideal, 0);
    r->args.dual.data[r->args.dim - 1] = 0;
}
int rpoly_add(void *h, const double *val, int ideal)
{
    rpoly *r = (rpoly *)h;
    memcpy(r->args.val, val, r->args.dim * sizeof(double));
    r->args.ideal = ideal ? 1 : 0;
    return poly__add_vrtx(&r->args);
}
int rpoly_init(void *h) { return poly__intl_apprx(&((rpoly *)h)->args); }
int rpoly_next(void *h, double *val, int *ideal, int *idx)
{
    rpoly *r = (rpoly *)h;
    if (poly__get_vrtx(&r->args)) return 1;
    memcpy(val, r->args.val, r->args.dim * sizeof(double));
    *ideal = r->args.ideal; *idx = (int)r->args.idx;
    return 0;
}
void rpoly_mark(void *h, int idx) { rpoly *r = (rpoly *)h; ST_BT(r->args.primal.sltn, (size_t)idx); }
void rpoly_dual_adjacency(void *h) { poly__update_adjacence(&((rpoly *)h)->args.dual); }

int rpoly_dim(void *h) { return (int)((rpoly *)h)->args.dim; }
int rpoly_nprimal(void *h) { return (int)((rpoly *)h)->args.primal.cnt; }
int rpoly_ndual(void *h) { return (int)((rpoly *)h)->args.dual.cnt; }

static long count_lists(polytope *p, poly_list *lists)
{
    long n = 0;
    for (size_t i = 0; i < p->cnt; i++) if (IS_ELEM(p->used, i)) n += lists[i].cnt;
    return n;
}
long rpoly_nedges(void *h) { rpoly *r = (rpoly *)h; return count_lists(&r->args.primal, r->args.primal.adjacence) / 2; }
long rpoly_ninc(void *h) { rpoly *r = (rpoly *)h; return count_lists(&r->args.primal, r->args.primal.incidence); }
long rpoly_ndual_edges(void *h) { rpoly *r = (rpoly *)h; return count_lists(&r->args.dual, r->args.dual.adjacence) / 2; }

static void get_side(polytope *p, unsigned char *used, unsigned char *ideal, unsigned char *sltn, double *coords)
{
    for (size_t i = 0; i < p->cnt; i++) {
        used[i] = IS_ELEM(p->used, i); ideal[i] = IS_ELEM(p->ideal, i);
        if (sltn) sltn[i] = IS_ELEM(p->sltn, i);
    }
    memcpy(coords, p->data, p->cnt * p->dim * sizeof(double));
}
void rpoly_get_primal(void *h, unsigned char *used, unsigned char *ideal, unsigned char *sltn, double *coords)
{ get_side(&((rpoly *)h)->args.primal, used, ideal, sltn, coords); }
void rpoly_get_dual(void *h, unsigned char *used, unsigned char *ideal, double *coords)
{ get_side(&((rpoly *)h)->args.dual, used, ideal, NULL, coords); }

static void get_adj(polytope *p, int *ab)
{
    long n = 0;
    for (size_t i = 0; i < p->cnt; i++) {
        if (!IS_ELEM(p->used, i)) continue;
        for (size_t k = 0; k < p->adjacence[i].cnt; k++) {
            size_t j = p->adjacence[i].data[k];
            if (i < j) { ab[2 * n] = (int)i; ab[2 * n + 1] = (int)j; n++; }
        }
    }
}
void rpoly_get_edges(void *h, int *ab) { get_adj(&((rpoly *)h)->args.primal, ab); }
void rpoly_get_dual_edges(void *h, int *ab) { get_adj(&((rpoly *)h)->args.dual, ab); }
void rpoly_get_inc(void *h, int *pairs)
{
    polytope *p = &((rpoly *)h)->args.primal;
    long n = 0;
    for (size_t i = 0; i < p->cnt; i++) {
        if (!IS_ELEM(p->used, i)) continue;
        for (size_t k = 0; k < p->incidence[i].cnt; k++) { pairs[2 * n] = (int)i; pairs[2 * n + 1] = (int)p->incidence[i].data[k]; n++; }
    }
}
