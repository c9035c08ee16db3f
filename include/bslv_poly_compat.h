/*
 * bslv_poly_compat.h -- the reference's polyhedron boundary, symbol for symbol.
 *
 * libbslv_hip.so exports the poly__* functions that bslv_algs.o imports from bslv_poly.c (declared in the reference's
 * bslv_poly.h:90-118; `nm -u` list in SURVEY.md section 8b) with IDENTICAL names, argument types and meaning, over structs
 * with the layout of bslv_poly.h:49-88, so the reference's driver links against the HIP polyhedron engine unchanged (replace
 * bslv_poly.o by -lbslv_hip).  bslv_algs.c does not only call these functions, it reads and writes the structs directly
 * (bslv_algs.c:67-72, 162-183, 188-279, 338-339, 355-394, 868, 893, 909-918, 1038, 1076-1078, 1100-1122 ...): `polytope` is
 * therefore kept as a HOST MIRROR of the engine's state that is coherent at every return from a poly__* call --
 *   data / cnt / used / ideal / sltn of both sides are refreshed from the engine after every call that changes it;
 *   the caller's writes are picked up at the next call: ST_BT(primal.sltn, idx) becomes bslv_poly_mark, the apex tweak of
 *   cone_vertenum (UNST_BT(dual.ideal, 0); dual.data[dim-1] = 0, bslv_algs.c:338-339) becomes bslv_poly_dual0_apex;
 *   data_primg lives only in the mirror (the engine never looks at pre-images);
 *   adjacence / incidence are filled when they are asked for (poly__initialise_permutation, poly__update_adjacence).
 * The V->H callback (a function pointer without context, bslv_poly.h:79-80) is identified by probing it: cone_polar,
 * lowerV2upperH and upperV2lowerH (bslv_poly.c:30-39, bslv_algs.c:287-313) are the engine's three built-in maps, the
 * parameter c is read off the probe.  Any other callback (the plot transforms) is refused: plotting (poly__plot, poly__swap)
 * is outside the accelerated path.
 */
#ifndef BSLV_POLY_COMPAT_H
#define BSLV_POLY_COMPAT_H
#include <stddef.h>
#include <limits.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef size_t btstrg;                                   /* bslv_poly.h:40-45: bit sets of 64-bit words */
typedef btstrg vrtx_strg;
typedef struct poly_list_strct { size_t cnt; size_t blcks; size_t *data; } poly_list;                 /* :49-53 */
typedef struct polytope_strct {                                                                        /* :55-69 */
    size_t dim, dim_primg;
    size_t cnt;
    size_t blcks;
    double *ip;
    double *data;
    double *data_primg;
    poly_list *adjacence;
    poly_list *incidence;
    vrtx_strg *ideal;
    vrtx_strg *used;
    vrtx_strg *sltn;
    struct polytope_strct *dual;
    void (*v2h)(double *, int, double *);
} polytope;
typedef struct {                                                                                       /* :71-82 */
    size_t dim, dim_primg_prml, dim_primg_dl;
    unsigned int ideal : 1;
    size_t idx;
    double *val, *val_primg_prml, *val_primg_dl;
    double eps;
    polytope primal;
    polytope dual;
    void (*primalV2dualH)();
    void (*dualV2primalH)();
    struct { double *H, *R, *alph; poly_list queue, gnrtrs; unsigned int intlsd : 1; } init_data;
} poly_args;
typedef struct { size_t cnt; size_t *data; size_t *inv; } permutation;                                 /* :84-88 */

void poly__set_default_args(poly_args *args, size_t dim);                                              /* bslv_poly.c:41-56 */
void poly__initialise(poly_args *args);                                                                /* :58-102 */
void poly__kill(poly_args *args);                                                                      /* :258-312 */
int  poly__add_vrtx(poly_args *args);                                                                  /* :104-151 */
int  poly__intl_apprx(poly_args *args);                                                                /* :153-208 */
int  poly__get_vrtx(poly_args *args);                                                                  /* :210-226 */
void poly__update_adjacence(polytope *poly);                                                           /* :992-1010 */
void poly__initialise_permutation(polytope *poly, permutation *prm);                                   /* :314-330 */
void poly__kill_permutation(permutation *prm);                                                         /* :332-339 */
void poly__vrtx2file(polytope *poly, permutation *prm, const char *fname, const char *frmt);           /* :341-360 */
void poly__primg2file(polytope *poly, permutation *prm, const char *fname, const char *frmt);          /* :362-380 */
void poly__adj2file(polytope *poly, permutation *prm, const char *fname, const char *frmt);            /* :382-397 */
void poly__inc2file(polytope *poly, permutation *prm, permutation *prm_dual, const char *fname, const char *frmt);   /* :399-414 */
void poly__swap(poly_args *a, poly_args *b);          /* plotting only (bslv_algs.c:1126-1134): refused */
void poly__plot(polytope *poly, const char *fname);   /* plotting only: refused */
void poly__polyck(poly_args *args);                   /* POLY_TEST builds only: no-op */

#ifdef __cplusplus
}
#endif
#endif
