/* bslv_vlp_io.c -- reader for the reference's .vlp problem files (kept contract).
 *
 * Written from the format description (parser bslv_vlp.c:275-588, writer
 * ex/prob2vlp.m:105-182), not from the reference's character-level scanner: this reader works line
 * by line with strtol/strtod.  Same accept/reject behaviour for well-formed files:
 *   p vlp (min|max) m n nz q nzobj [(cone|dualcone) n_gen nzgen]
 *   a i j val | o k j val | k i j val (j = 0 sets c_i) | i row type [lb] [ub] | j col type [lb] [ub] | e
 *   'c ...' comment lines; empty lines are ignored with a warning; missing rows default to 'f',
 *   missing columns to 's' (fixed at 0!); duplicate i/j descriptors and surplus a/o/k lines are errors.
 * Deviation (reference quirk): the component index of a 'k' line is range-checked
 * against q (the reference checks it against n_gen and can write out of bounds).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <errno.h>
#include <math.h>
#include "bslv_host.h"

#define MAXTOK 11

static int fail(bslv_vlp *v, const char *m) { snprintf(v->msg, sizeof v->msg, "%s", m); return 1; }

static int tok_int(const char *s, int *out)
{
    char *e; errno = 0;
    long x = strtol(s, &e, 10);
    if (e == s || *e != '\0' || errno || x < -2147483647L || x > 2147483647L) return 1;
    *out = (int)x;
    return 0;
}
static int tok_num(const char *s, double *out)
{
    char *e; errno = 0;
    if (!(isdigit((unsigned char)s[0]) || s[0] == '+' || s[0] == '-' || s[0] == '.')) return 1;   /* no inf/nan/hex */
    double x = strtod(s, &e);
    if (e == s || *e != '\0' || errno == ERANGE || !isfinite(x)) return 1;
    for (const char *p = s; *p; p++) if (*p == 'x' || *p == 'X') return 1;
    *out = x;
    return 0;
}
static int bound_type(const char *s, char *t)
{
    if (strlen(s) != 1 || !strchr("fluds", s[0])) return 1;
    *t = s[0];
    return 0;
}

void bslv_vlp_free(bslv_vlp *v)
{
    if (!v) return;
    free(v->A); free(v->P); free(v->rtype); free(v->ctype); free(v->rlb); free(v->rub);
    free(v->clb); free(v->cub); free(v->gen); free(v->c);
    free(v);
}
const char *bslv_vlp_message(const bslv_vlp *v) { return v ? v->msg : ""; }

int bslv_vlp_read(const char *path, bslv_vlp **out, int *err_line)
{
    bslv_vlp *v = (bslv_vlp *)calloc(1, sizeof *v);
    *out = v;
    if (err_line) *err_line = 0;
    FILE *fp = fopen(path, "r");
    if (!fp) return fail(v, "file not found or unable to open");
    char *line = NULL; size_t cap = 0; ssize_t len;
    int lineno = 0, rc = 0, have_p = 0, ended = 0, nzgen = 0;
    long na = 0, no = 0, nk = 0;
    while (!ended && (len = getline(&line, &cap, fp)) >= 0) {
        lineno++;
        char *tok[MAXTOK + 1]; int nt = 0;
        for (char *p = line; *p; p++) if (iscntrl((unsigned char)*p) && !isspace((unsigned char)*p)) { rc = fail(v, "invalid control character"); goto done; }
        if (line[0] == 'c' && (line[1] == '\0' || isspace((unsigned char)line[1]))) continue;     /* comment */
        for (char *p = strtok(line, " \t\r\n\v\f"); p; p = strtok(NULL, " \t\r\n\v\f")) {
            if (nt == MAXTOK) { rc = fail(v, "too many data fields"); goto done; }
            if (strlen(p) > 255) { rc = fail(v, "data field too long"); goto done; }
            tok[nt++] = p;
        }
        if (nt == 0) { if (!v->warnings++) snprintf(v->msg, sizeof v->msg, "empty line ignored"); continue; }
#define NEED(k, m) do { if (nt != (k)) { rc = fail(v, m); goto done; } } while (0)
        if (!have_p) {
            if (strcmp(tok[0], "p")) { rc = fail(v, "problem line missing or invalid"); goto done; }
            if (nt < 3 || strcmp(tok[1], "vlp")) { rc = fail(v, "wrong problem designator"); goto done; }
            if (!strcmp(tok[2], "min")) v->optdir = 1; else if (!strcmp(tok[2], "max")) v->optdir = -1;
            else { rc = fail(v, "objective sense missing or invalid"); goto done; }
            int nzi, nzo;
            if (nt != 8 && nt != 11) { rc = fail(v, "problem line: wrong number of fields"); goto done; }
            if (tok_int(tok[3], &v->m) || v->m < 0) { rc = fail(v, "number of rows missing or invalid"); goto done; }
            if (tok_int(tok[4], &v->n) || v->n < 0) { rc = fail(v, "number of columns missing or invalid"); goto done; }
            if (tok_int(tok[5], &nzi) || nzi < 0) { rc = fail(v, "number of nonzeros missing or invalid"); goto done; }
            if (tok_int(tok[6], &v->q) || v->q < 1) { rc = fail(v, "number of objectives missing or invalid"); goto done; }
            if (tok_int(tok[7], &nzo) || nzo < 0) { rc = fail(v, "number of objective matrix nonzeros missing or invalid"); goto done; }
            v->nz = nzi; v->nzobj = nzo;
            if (nt == 11) {
                if (!strcmp(tok[8], "cone")) v->cone_gen = BSLV_CONE_PRIMAL;
                else if (!strcmp(tok[8], "dualcone")) v->cone_gen = BSLV_CONE_DUAL;
                else { rc = fail(v, "type of cone generators missing or invalid"); goto done; }
                if (tok_int(tok[9], &v->n_gen) || v->n_gen < 0) { rc = fail(v, "number of cone generating vectors missing or invalid"); goto done; }
                if (tok_int(tok[10], &nzgen) || nzgen < 0) { rc = fail(v, "number of cone generator non-zeros missing or invalid"); goto done; }
            }
            /* the header of an untrusted file sizes every allocation: bound each count before anything is allocated (q <= 16 is
             * what the polyhedron engine supports; 2^28 dense entries is the limit of the dense path), sizes in size_t */
            if (v->q > 16) { rc = fail(v, "more than 16 objectives are not supported"); goto done; }
            if (v->m > 100000000 || v->n > 100000000 || v->n_gen > 1000000) { rc = fail(v, "problem dimensions out of range"); goto done; }
            if ((double)v->m * v->n > 268435456.0 || (double)v->q * v->n > 268435456.0) { rc = fail(v, "problem too large for the dense path (sparse-A path is SURVEY 8f rank 4)"); goto done; }
            size_t mn = (size_t)(v->m ? v->m : 1) * (size_t)(v->n ? v->n : 1), qn = (size_t)v->q * (size_t)(v->n ? v->n : 1);
            v->A = (double *)calloc(mn, sizeof(double)); v->P = (double *)calloc(qn, sizeof(double));
            v->rtype = (char *)malloc((size_t)v->m + 1); v->ctype = (char *)malloc((size_t)v->n + 1);
            v->rlb = (double *)calloc((size_t)v->m + 1, 8); v->rub = (double *)calloc((size_t)v->m + 1, 8);
            v->clb = (double *)calloc((size_t)v->n + 1, 8); v->cub = (double *)calloc((size_t)v->n + 1, 8);
            v->c = (double *)calloc((size_t)v->q, 8);
            if (v->cone_gen != BSLV_CONE_DEFAULT) v->gen = (double *)calloc((size_t)v->q * (size_t)(v->n_gen ? v->n_gen : 1), 8);
            if (!v->A || !v->P || !v->rtype || !v->ctype || !v->rlb || !v->rub || !v->clb || !v->cub || !v->c || (v->cone_gen != BSLV_CONE_DEFAULT && !v->gen)) {
                rc = fail(v, "out of memory"); goto done;
            }
            memset(v->rtype, 'x', (size_t)v->m + 1); memset(v->ctype, 'x', (size_t)v->n + 1);
            have_p = 1;
            continue;
        }
        int i, j; double x;
        if (!strcmp(tok[0], "a")) {
            NEED(4, "constraint coefficient descriptor: wrong number of fields");
            if (na == v->nz) { rc = fail(v, "too many constraint coefficient descriptors"); goto done; }
            if (tok_int(tok[1], &i)) { rc = fail(v, "constraint coefficient row number missing or invalid"); goto done; }
            if (i < 1 || i > v->m) { rc = fail(v, "constraint coefficient row number out of range"); goto done; }
            if (tok_int(tok[2], &j)) { rc = fail(v, "constraint coefficient column number missing or invalid"); goto done; }
            if (j < 1 || j > v->n) { rc = fail(v, "constraint coefficient column number out of range"); goto done; }
            if (tok_num(tok[3], &x)) { rc = fail(v, "constraint coefficient missing or invalid"); goto done; }
            v->A[(size_t)(i - 1) * v->n + (j - 1)] = x; na++;
        } else if (!strcmp(tok[0], "o")) {
            NEED(4, "objective coefficient descriptor: wrong number of fields");
            if (no == v->nzobj) { rc = fail(v, "too many objective coefficient descriptors"); goto done; }
            if (tok_int(tok[1], &i)) { rc = fail(v, "objective coefficient row number missing or invalid"); goto done; }
            if (i < 1 || i > v->q) { rc = fail(v, "objective coefficient row number out of range"); goto done; }
            if (tok_int(tok[2], &j)) { rc = fail(v, "objective coefficient column number missing or invalid"); goto done; }
            if (j < 1 || j > v->n) { rc = fail(v, "objective coefficient column number out of range"); goto done; }
            if (tok_num(tok[3], &x)) { rc = fail(v, "objective coefficient missing or invalid"); goto done; }
            v->P[(size_t)(i - 1) * v->n + (j - 1)] = x; no++;
        } else if (!strcmp(tok[0], "k")) {
            if (v->cone_gen == BSLV_CONE_DEFAULT) { rc = fail(v, "invalid designator k"); goto done; }
            NEED(4, "cone generator descriptor: wrong number of fields");
            if (tok_int(tok[1], &i)) { rc = fail(v, "cone generator coefficient row number missing or invalid"); goto done; }
            if (i < 1 || i > v->q) { rc = fail(v, "cone generator coefficient row number out of range"); goto done; }
            if (tok_int(tok[2], &j)) { rc = fail(v, "cone generator coefficient column number missing or invalid"); goto done; }
            if (j < 0 || j > v->n_gen) { rc = fail(v, "cone generator coefficient column number out of range"); goto done; }
            if (tok_num(tok[3], &x)) { rc = fail(v, "cone generator coefficient missing or invalid"); goto done; }
            if (j == 0) v->c[i - 1] = x;
            else {
                if (nk == nzgen) { rc = fail(v, "too many cone generator coefficient descriptors"); goto done; }
                v->gen[(size_t)v->n_gen * (i - 1) + (j - 1)] = x; nk++;
            }
        } else if (!strcmp(tok[0], "i") || !strcmp(tok[0], "j")) {
            const int isrow = tok[0][0] == 'i';
            const int lim = isrow ? v->m : v->n;
            char *types = isrow ? v->rtype : v->ctype; double *lb = isrow ? v->rlb : v->clb, *ub = isrow ? v->rub : v->cub;
            char t;
            if (nt < 3) { rc = fail(v, isrow ? "row type missing or invalid" : "column type missing or invalid"); goto done; }
            if (tok_int(tok[1], &i)) { rc = fail(v, isrow ? "row number missing or invalid" : "column number missing or invalid"); goto done; }
            if (i < 1 || i > lim) { rc = fail(v, isrow ? "row number out of range" : "column descriptor out of range"); goto done; }
            if (types[i - 1] != 'x') { rc = fail(v, isrow ? "duplicate row descriptor" : "duplicate column descriptor"); goto done; }
            if (bound_type(tok[2], &t)) { rc = fail(v, isrow ? "row type missing or invalid" : "column type missing or invalid"); goto done; }
            int need = 3 + ((t == 'l' || t == 'd' || t == 's') ? 1 : 0) + ((t == 'u' || t == 'd') ? 1 : 0), k = 3;
            NEED(need, isrow ? "row descriptor: wrong number of fields" : "column descriptor: wrong number of fields");
            if (t == 'l' || t == 'd' || t == 's') if (tok_num(tok[k++], &lb[i - 1])) { rc = fail(v, "lower bound missing or invalid"); goto done; }
            if (t == 'u' || t == 'd') if (tok_num(tok[k++], &ub[i - 1])) { rc = fail(v, "upper bound missing or invalid"); goto done; }
            types[i - 1] = t;
        } else if (!strcmp(tok[0], "e")) {
            ended = 1;
        } else { rc = fail(v, "line designator missing or invalid"); goto done; }
    }
    if (!have_p) { rc = fail(v, "problem line missing or invalid"); goto done; }
    if (!ended) { rc = fail(v, "unexpected end of file (missing 'e' line)"); goto done; }
    for (int i = 0; i < v->m; i++) if (v->rtype[i] == 'x') v->rtype[i] = 'f';
    for (int j = 0; j < v->n; j++) if (v->ctype[j] == 'x') v->ctype[j] = 's';
done:
    if (rc && err_line) *err_line = lineno;
    free(line);
    fclose(fp);
    return rc;
}
