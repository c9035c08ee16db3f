/* bensolve_hip -- C command-line driver over libbslv_hip.so: .vlp in, the reference's .sol files out.
 *
 * Covers what SURVEY.md section 8 rows a-e need from bslv_main.c / bslv_algs.c: bounded problems
 * with the default ordering cone ("-b": phase 2 only, R := Z = I, bslv_algs.c:943-956).  Phases 0/1 and
 * non-default cones (section 8f rank 1) are not built yet: the program says so and exits with code 2.
 * Timing mirrors the reference: starts after the file is loaded (bslv_main.c:236), stops before the
 * result files are written (bslv_algs.c:1140). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "../../../include/bslv_hip.h"
#include "bslv_host.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void usage(void)
{
    printf("Usage: bensolve_hip file.vlp -b [options]\n"
           "  -b, --bounded            assume the problem is bounded: run phase 2 only (required in this version)\n"
           "  -e, --eps_phase2 EPS     epsilon of Benson's algorithm in phase 2 (default 1e-7)\n"
           "  -o, --output_filename F  base name of the result files (default: input name up to the first '.')\n"
           "  -m, --message_level N    0-3 (default 1)\n"
           "  -B, --batch N            LPs per outer iteration (default 1024)\n"
           "  -S, --slots N            tableau pool slots (default 4*batch+64)\n");
}

int main(int argc, char **argv)
{
    if (argc < 2 || argv[1][0] == '-') { usage(); return 1; }
    const char *file = argv[1];
    int bounded = 0, msg = 1, batch = 1024, slots = 0;
    double eps = 1e-7;
    char base[1024] = "";
    for (int a = 2; a < argc; a++) {
        const char *o = argv[a];
#define ARG() (a + 1 < argc ? argv[++a] : (usage(), exit(1), ""))
        if (!strcmp(o, "-b") || !strcmp(o, "--bounded")) bounded = 1;
        else if (!strcmp(o, "-e") || !strcmp(o, "--eps_phase2")) { eps = atof(ARG()); if (!(eps > 0)) { printf("option --eps_phase2 (-e): invalid argument\n"); return 1; } }
        else if (!strcmp(o, "-o") || !strcmp(o, "--output_filename")) snprintf(base, sizeof base, "%s", ARG());
        else if (!strcmp(o, "-m") || !strcmp(o, "--message_level")) msg = atoi(ARG());
        else if (!strcmp(o, "-B") || !strcmp(o, "--batch")) batch = atoi(ARG());
        else if (!strcmp(o, "-S") || !strcmp(o, "--slots")) slots = atoi(ARG());
        else if (!strcmp(o, "-h") || !strcmp(o, "--help")) { usage(); return 1; }
        else { printf("invalid option %s\n", o); return 1; }
    }
    if (!base[0]) { snprintf(base, sizeof base, "%s", file); char *dot = strchr(base, '.'); if (dot && dot != base) *dot = 0; }
    if (batch < 1) batch = 1;
    if (slots < 4 * batch + 64) slots = 4 * batch + 64;

    bslv_vlp *v = NULL;
    int line = 0;
    if (msg >= 1) printf("loading ... \n");
    if (bslv_vlp_read(file, &v, &line)) {
        printf("Error while reading %s: line %d: %s\n", file, line, bslv_vlp_message(v));
        bslv_vlp_free(v);
        return 1;
    }
    if (v->warnings) printf("Warning occurred while reading %s: %s\n", file, bslv_vlp_message(v));
    if (msg >= 1) printf("done: %d rows, %d columns, %ld non-zero matrix coefficients\n", v->m, v->n, v->nz);
    if (v->cone_gen != BSLV_CONE_DEFAULT || !bounded) {
        printf("this version covers bounded problems with the default ordering cone only (run with -b); "
               "phases 0/1 and cone/dualcone problems are not built yet\n");
        bslv_vlp_free(v);
        return 2;
    }
    double t0 = now();
    const int q = v->q;
    /* sol_init (bslv_vlp.c:661-683, 775-792, 856-861): Y = Z = I, c = (1..1), P negated for max */
    double *R = (double *)calloc((size_t)q * q, sizeof(double)), *c = (double *)malloc(q * sizeof(double));
    for (int k = 0; k < q; k++) { R[(size_t)k * q + k] = 1.0; c[k] = 1.0; }
    if (v->optdir == -1) for (size_t k = 0; k < (size_t)q * v->n; k++) v->P[k] = -v->P[k];
    bslv_benson *h = NULL;
    int rc = bslv_benson_create(&h, v->m, v->n, q, v->A, v->P, v->rtype, v->rlb, v->rub, v->ctype, v->clb, v->cub, R, q, c, eps, slots);
    if (rc) { printf("engine error %d: %s\n", rc, bslv_last_error()); return 3; }
    if (msg >= 1) printf("running ... \n");
    int st = 0;
    if ((rc = bslv_benson_start(h, &st))) { printf("engine error %d: %s\n", rc, bslv_last_error()); return 3; }
    if (st == 1) { printf("VLP is infeasible\n"); return 1; }
    if (st == 2) { printf("VLP is not bounded, re-run without option -b\n"); return 1; }
    long stats[8]; double ms[3]; long steps = 0;
    do {
        if ((rc = bslv_benson_step(h, batch, stats, ms))) { printf("engine error %d: %s\n", rc, bslv_last_error()); return 3; }
        steps++;
        if (msg >= 3) printf("step %ld: %ld LPs, %ld cuts, %ld confirmed, %ld left, %.2f ms (LP %.2f, poly %.2f)\n", steps, stats[0], stats[1], stats[3], stats[7], ms[2], ms[0], ms[1]);
    } while (stats[0] > 0 || stats[7] > 0);
    double elapsed = now() - t0;
    long cnt[4], lps = 0, cuts = 0, piv = 0;
    char cfile[1100];
    snprintf(cfile, sizeof cfile, "%s_c.sol", base);
    FILE *cf = fopen(cfile, "w");                                         /* bslv_vlp.c:833-842 */
    if (cf) { for (int k = 0; k < q; k++) fprintf(cf, k ? " %.14g" : "%.14g", c[k]); fprintf(cf, "\n"); fclose(cf); }
    if ((rc = bslv_sol_write(bslv_benson_poly(h), base, ".sol", v->optdir, cnt))) { printf("writing results failed (%d): %s\n", rc, bslv_last_error()); return 3; }
    bslv_benson_totals(h, &lps, &cuts, &piv);
    if (msg >= 1) {
        printf("CPU time            : %.4g %s.\n", elapsed >= 1 ? elapsed : elapsed * 1e3, elapsed >= 1 ? "s" : "ms");
        printf("Number of LPs solved: %ld.\n", lps);
    }
    if (msg >= 2) printf("outer iterations %ld, cuts %ld, pivots %ld; upper image: %ld points, %ld directions; lower image: %ld points, %ld directions\n",
                         steps, cuts, piv, cnt[0], cnt[1], cnt[2], cnt[3]);
    bslv_benson_destroy(h);
    bslv_vlp_free(v);
    free(R); free(c);
    return 0;
}
