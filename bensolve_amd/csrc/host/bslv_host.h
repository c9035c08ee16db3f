/* bslv_host.h -- plain-C host side that stays C (north star): the .vlp reader and the result
 * writers of the reference, re-written from the format contract.
 * Part of libbslv_hip.so's C ABI (declared again in include/bslv_hip.h section 5). */
#ifndef BSLV_HOST_H
#define BSLV_HOST_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { BSLV_CONE_DEFAULT = 0, BSLV_CONE_PRIMAL = 1, BSLV_CONE_DUAL = 2 };

/* in-memory VLP, dense (vlptype of bslv_vlp.h:47-64 with A_ext = [A 0; -P I] split into A and P) */
typedef struct bslv_vlp {
    int m, n, q;
    int optdir;            /* 1 min, -1 max */
    int cone_gen;          /* BSLV_CONE_* */
    int n_gen;
    long nz, nzobj;
    double *A;             /* m x n row-major */
    double *P;             /* q x n row-major, as written in the file (not negated) */
    char *rtype, *ctype;   /* 'f','l','u','d','s'; defaults rows 'f', cols 's' (bslv_vlp.c:567-574) */
    double *rlb, *rub, *clb, *cub;
    double *gen;           /* q x n_gen row-major: gen[n_gen*(i-1)+(j-1)] (bslv_vlp.c:482), NULL for the default cone */
    double *c;             /* q, zero unless given by 'k i 0 val' lines */
    int warnings;
    char msg[256];         /* last error / warning text */
} bslv_vlp;

/* vlp_init (bslv_vlp.c:275-588).  Returns 0 ok, 1 error (msg + line number in *err_line). */
int  bslv_vlp_read(const char *path, bslv_vlp **out, int *err_line);
void bslv_vlp_free(bslv_vlp *v);
const char *bslv_vlp_message(const bslv_vlp *v);

#ifdef __cplusplus
}
#endif
#endif
