/*
 * oracle/poly_dd.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Plain-C restatement of the polyhedron half of the hot path: the online double-description /
 * vertex-enumeration engine of the reference (bslv_poly.c:104-226, 467-512, 562-787, 992-1060;
 * bslv_poly.h:49-118).  It keeps the same objects -- a primal polyhedron (points + directions)
 * and its dual (one dual vertex per halfspace), incidence both ways, primal adjacency maintained
 * incrementally, dual adjacency rebuilt once at the end -- but stores them the way the HIP engine
 * does (sorted sparse incidence lists, an edge list) and processes one cut as four data-parallel
 * passes (classify / edges / on-plane incidence / pair tests) instead of a recursive graph walk.
 * Results are equal to the reference's as SETS (SURVEY.md section 8c comparison rule); this is
 * checked against the reference's own bslv_poly.c compiled into oracle/_ref/libref_poly.so.
 */
#ifndef ORACLE_POLY_DD_H
#define ORACLE_POLY_DD_H

#ifdef __cplusplus
extern "C" {
#endif

/* vertex -> halfspace maps (function pointers in the reference: bslv_poly.h:79-80) */
enum {
    OPOLY_CONE_POLAR = 0,    /* bslv_poly.c:30-39   */
    OPOLY_LOWER2UPPER = 1,   /* bslv_algs.c:287-305 */
    OPOLY_UPPER2LOWER = 2    /* bslv_algs.c:307-313 */
};

typedef struct opoly opoly;

opoly *opoly_create(int dim, int v2h, const double *c /* dim values, may be NULL for CONE_POLAR */);
void   opoly_free(opoly *p);
/* cone_vertenum's tweak: dual slot 0 becomes the apex (0,..,0), non-ideal (bslv_algs.c:338-339) */
void   opoly_dual0_apex(opoly *p);
/* poly__add_vrtx (bslv_poly.c:104-151): 0 = added (or queued before initialisation),
 * 1 = redundant (no primal vertex violates the halfspace; dual slot left unused) */
int    opoly_add(opoly *p, const double *val, int ideal);
/* poly__intl_apprx (bslv_poly.c:153-208): 0 ok, 1 = fewer than dim independent halfspaces */
int    opoly_init(opoly *p);
/* poly__get_vrtx (bslv_poly.c:210-226): lowest live primal slot without the sltn mark;
 * returns 1 when none is left */
int    opoly_next(opoly *p, double *val, int *ideal, int *idx);
int    opoly_next_newest(opoly *p, double *val, int *ideal, int *idx);   /* measurement variant: the newest such slot */
void   opoly_mark(opoly *p, int idx);                 /* ST_BT(primal.sltn, idx) */
/* poly__update_adjacence on the dual side (bslv_poly.c:992-1010) */
void   opoly_dual_adjacency(opoly *p);

/* ---- dumps (slot-indexed; caller canonicalises) ---- */
int    opoly_dim(const opoly *p);
int    opoly_nprimal(const opoly *p);                 /* primal slots ever created */
int    opoly_ndual(const opoly *p);
long   opoly_nedges(const opoly *p);
long   opoly_ninc(const opoly *p);                    /* total incidence pairs over live primal slots */
long   opoly_ndual_edges(const opoly *p);
/* used/ideal flags (1 byte each) and coordinates (n x dim) */
void   opoly_get_primal(const opoly *p, unsigned char *used, unsigned char *ideal, unsigned char *sltn, double *coords);
void   opoly_get_dual(const opoly *p, unsigned char *used, unsigned char *ideal, double *coords);
void   opoly_get_edges(const opoly *p, int *ab /* 2 per edge */);
void   opoly_get_inc(const opoly *p, int *pairs /* (primal slot, dual slot) per pair */);
void   opoly_get_dual_edges(const opoly *p, int *ab);
/* work counters */
long   opoly_pair_tests(const opoly *p);              /* edge_test calls so far */
long   opoly_new_vertices(const opoly *p);            /* primal slots created by cuts */
/* the projection sub-band of poly__cut (bslv_poly.c:666-674), off by default (see poly_dd.c's header); elements projected so far */
void   opoly_set_snap(opoly *p, int on);
long   opoly_snapped(const opoly *p);

#ifdef __cplusplus
}
#endif
#endif
