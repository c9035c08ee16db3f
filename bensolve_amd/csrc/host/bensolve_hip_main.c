/* bensolve_hip -- C command-line driver over libbslv_hip.so: .vlp in, the reference's .sol files out.
 *
 * The primal algorithm of bslv_main.c:236-345: ordering cone data (default cone, `cone` and `dualcone` problems),
 * phases 0 and 1 unless "-b" is given (then R := Z, bslv_algs.c:943-956), the batched phase 2, the result files.
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
    printf("Usage: bensolve_hip file.vlp [options]\n"
           "  -b, --bounded            assume the problem is bounded: skip phases 0 and 1 (R := Z)\n"
           "  -s, --solution           write the solutions (pre-images) to <name>_pre_img_p.sol / _pre_img_d.sol (primal algorithm in phase 2)\n"
           "  -A, --alg_phase1 ALG     primal (default) or dual: the algorithm of phase 1\n"
           "  -a, --alg_phase2 ALG     primal (default) or dual: Benson's algorithm or its dual variant in phase 2\n"
           "  -E, --eps_phase1 EPS     epsilon of Benson's algorithm in phase 1 (default 1e-7)\n"
           "  -e, --eps_phase2 EPS     epsilon of Benson's algorithm in phase 2 (default 1e-7)\n"
           "  -o, --output_filename F  base name of the result files (default: input name up to the first '.')\n"
           "  -m, --message_level N    0-3 (default 1)\n"
           "  -B, --batch N            LPs per outer iteration (default 1024)\n"
           "  -k/-L/-l METHOD, -M N, -f FORMAT, -p, -t   accepted as in the reference's command line; the LP method is recorded in the\n"
           "                           .log (every LP is solved by the engine's dual simplex), no graphics file is written\n"
           );
}

/* Multi-GPU launch (one process per GPU; RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT as torchrun, mpirun wrappers
 * or a shell loop set them): rank 0 creates the RCCL id and hands it to the other ranks over TCP (port MASTER_PORT + 17, or
 * BSLV_BOOT_PORT), then every rank joins the communicator.  Returns the world size, or -1 on failure. */
#include <sys/socket.h>
#include <netinet/in.h>
#include <arpa/inet.h>
#include <netdb.h>
#include <unistd.h>
#include <poll.h>
static int env_int(const char *k, int dflt) { const char *v = getenv(k); return v && *v ? atoi(v) : dflt; }
static int dist_bootstrap(int *rank_out)
{
    const int world = env_int("WORLD_SIZE", 1), rank = env_int("RANK", 0), local = env_int("LOCAL_RANK", rank);
    *rank_out = rank;
    if (world <= 1) return 1;
    if (bslv_set_device(local)) { printf("rank %d: cannot use GPU %d: %s\n", rank, local, bslv_last_error()); return -1; }
    const char *addr = getenv("MASTER_ADDR") ? getenv("MASTER_ADDR") : "127.0.0.1";
    const int port = getenv("BSLV_BOOT_PORT") ? atoi(getenv("BSLV_BOOT_PORT")) : env_int("MASTER_PORT", 29500) + 17;
    unsigned char id[128];
    if (rank == 0) {
        if (bslv_dist_unique_id(id, 128)) { printf("rank 0: %s\n", bslv_last_error()); return -1; }
        int ls = socket(AF_INET, SOCK_STREAM, 0), one = 1;
        setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        struct sockaddr_in sa; memset(&sa, 0, sizeof sa);
        sa.sin_family = AF_INET; sa.sin_addr.s_addr = htonl(INADDR_ANY); sa.sin_port = htons((unsigned short)port);
        if (ls < 0 || bind(ls, (struct sockaddr *)&sa, sizeof sa) || listen(ls, world)) { printf("rank 0: cannot listen on port %d\n", port); return -1; }
        for (int k = 1; k < world; k++) {
            /* a peer that died before it connected (no GPU, bad LOCAL_RANK) must not leave rank 0 waiting for ever: two minutes */
            struct pollfd pf; pf.fd = ls; pf.events = POLLIN; pf.revents = 0;
            if (poll(&pf, 1, 120000) <= 0) { printf("rank 0: only %d of %d ranks asked for the communicator id within two minutes\n", k, world); close(ls); return -1; }
            int c = accept(ls, NULL, NULL);
            if (c < 0 || write(c, id, 128) != 128) { printf("rank 0: handing out the communicator id failed\n"); return -1; }
            close(c);
        }
        close(ls);
    } else {
        struct addrinfo hints, *res = NULL; memset(&hints, 0, sizeof hints);
        hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
        char ps[16]; snprintf(ps, sizeof ps, "%d", port);
        if (getaddrinfo(addr, ps, &hints, &res) || !res) { printf("rank %d: cannot resolve %s\n", rank, addr); return -1; }
        int c = -1, got = 0;
        for (int tries = 0; tries < 600 && got != 128; tries++) {          /* rank 0 may not be listening yet: retry for a minute */
            c = socket(AF_INET, SOCK_STREAM, 0);
            if (c >= 0 && connect(c, res->ai_addr, res->ai_addrlen) == 0) {
                got = 0;
                while (got < 128) { ssize_t r = read(c, id + got, 128 - got); if (r <= 0) break; got += (int)r; }
            }
            if (c >= 0) close(c);
            if (got != 128) usleep(100000);
        }
        freeaddrinfo(res);
        if (got != 128) { printf("rank %d: no communicator id from %s:%d\n", rank, addr, port); return -1; }
    }
    if (bslv_dist_init(rank, world, id, 128)) { printf("rank %d: %s\n", rank, bslv_last_error()); return -1; }
    return world;
}

int main(int argc, char **argv)
{
    if (argc < 2 || argv[1][0] == '-') { usage(); return 1; }
    const char *file = argv[1];
    int bounded = 0, msg = 1, batch = 1024, dual2 = 0, dual1 = 0, presol = 0, plot_asked = 0;
    int lp_method[3] = {0, 0, 0};                  /* per phase: 0 as the reference's default, 1 primal_simplex, 2 dual_simplex, 3 dual_primal_simplex */
    double eps = 1e-7, eps1 = 1e-7;
    char base[1024] = "";
    for (int a = 2; a < argc; a++) {
        const char *o = argv[a];
#define ARG() (a + 1 < argc ? argv[++a] : (usage(), exit(1), ""))
        if (!strcmp(o, "-b") || !strcmp(o, "--bounded")) bounded = 1;
        else if (!strcmp(o, "-e") || !strcmp(o, "--eps_phase2")) { eps = atof(ARG()); if (!(eps > 0)) { printf("option --eps_phase2 (-e): invalid argument\n"); return 1; } }
        else if (!strcmp(o, "-o") || !strcmp(o, "--output_filename")) snprintf(base, sizeof base, "%s", ARG());
        else if (!strcmp(o, "-m") || !strcmp(o, "--message_level")) msg = atoi(ARG());
        else if (!strcmp(o, "-B") || !strcmp(o, "--batch")) batch = atoi(ARG());
        else if (!strcmp(o, "-a") || !strcmp(o, "--alg_phase2")) {
            const char *v2 = ARG();
            if (!strcmp(v2, "dual")) dual2 = 1; else if (!strcmp(v2, "primal")) dual2 = 0; else { printf("option --alg_phase2 (-a): invalid argument\n"); return 1; }
        }
        else if (!strcmp(o, "-s") || !strcmp(o, "--solution")) presol = 1;
        else if (!strcmp(o, "-A") || !strcmp(o, "--alg_phase1")) {
            const char *v1 = ARG();
            if (!strcmp(v1, "dual")) dual1 = 1; else if (!strcmp(v1, "primal")) dual1 = 0; else { printf("option --alg_phase1 (-A): invalid argument\n"); return 1; }
        }
        else if (!strcmp(o, "-E") || !strcmp(o, "--eps_phase1")) { eps1 = atof(ARG()); if (!(eps1 > 0)) { printf("option --eps_phase1 (-E): invalid argument\n"); return 1; } }
        /* options of the reference's command line that change nothing here: accepted with the reference's own argument check
         * (bslv_main.c:112-170), so that a documented command line such as ex/example09.m's
         * "ex09.vlp -e 1e-2 -m 3 -L primal_simplex -l primal_simplex -p" runs as it stands.  The LP method is recorded in the .log;
         * every LP is solved by the engine's dual simplex; no graphics file is written. */
        else if (!strcmp(o, "-k") || !strcmp(o, "--lp_method_phase0") || !strcmp(o, "-L") || !strcmp(o, "--lp_method_phase1") || !strcmp(o, "-l") || !strcmp(o, "--lp_method_phase2")) {
            const int ph = (!strcmp(o, "-k") || !strcmp(o, "--lp_method_phase0")) ? 0 : (!strcmp(o, "-L") || !strcmp(o, "--lp_method_phase1")) ? 1 : 2;
            const char *v = ARG();
            if (!strcmp(v, "primal_simplex")) lp_method[ph] = 1;
            else if (!strcmp(v, "dual_simplex")) lp_method[ph] = 2;
            else if (!strcmp(v, "dual_primal_simplex")) lp_method[ph] = 3;
            else if (ph > 0 && !strcmp(v, "auto")) lp_method[ph] = 0;
            else { printf("option --lp_method_phase%d (-%c): invalid argument\n", ph, ph == 0 ? 'k' : ph == 1 ? 'L' : 'l'); return 1; }
        }
        else if (!strcmp(o, "-M") || !strcmp(o, "--lp_message_level")) { const int v = atoi(ARG()); if (v < 0 || v > 3) { printf("option --lp_message_level (-M): invalid argument\n"); return 1; } }
        else if (!strcmp(o, "-f") || !strcmp(o, "--format")) { const char *v = ARG(); if (strcmp(v, "auto") && strcmp(v, "long") && strcmp(v, "short")) { printf("option --format (-f): invalid argument\n"); return 1; } }
        else if (!strcmp(o, "-p") || !strcmp(o, "--plot")) plot_asked = 1;
        else if (!strcmp(o, "-t") || !strcmp(o, "--test")) { /* the reference's integrity tests of its own polytope lists: nothing to test here */ }
        else if (!strcmp(o, "-h") || !strcmp(o, "--help")) { usage(); return 1; }
        else { printf("invalid option %s\n", o); return 1; }
    }
    if (!base[0]) { snprintf(base, sizeof base, "%s", file); char *dot = strchr(base, '.'); if (dot && dot != base) *dot = 0; }
    if (batch < 1) batch = 1;
    int rank = 0;
    const int world = dist_bootstrap(&rank);
    if (world < 0) return 3;
    if (world > 1) {
        if (presol) { if (rank == 0) printf("option -s is not available with more than one rank (the pre-images stay with the rank that solved the LP)\n"); return 1; }
        if (rank != 0) {                              /* one rank talks; the others write their (identical) replica under <base>.rank<k> */
            msg = 0;
            char tmp[1024];
            snprintf(tmp, sizeof tmp, "%s", base);
            snprintf(base, sizeof base, "%.1000s.rank%d", tmp, rank);
        }
    }

    bslv_vlp *v = NULL;
    int line = 0;
    if (plot_asked && msg >= 1) printf("option --plot (-p): graphics files are not written by this driver\n");
    if (msg >= 1) printf("loading ... \n");
    if (bslv_vlp_read(file, &v, &line)) {
        printf("Error while reading %s: line %d: %s\n", file, line, bslv_vlp_message(v));
        bslv_vlp_free(v);
        return 1;
    }
    if (v->warnings) printf("Warning occurred while reading %s: %s\n", file, bslv_vlp_message(v));
    if (msg >= 1) printf("done: %d rows, %d columns, %ld non-zero matrix coefficients\n", v->m, v->n, v->nz);
    double t0 = now();
    const int q = v->q;
    /* sol_init, phases 0 and 1 (unless -b), phase 2: bslv_main.c:236-345 */
    bslv_benson *h = NULL;
    bslv_vlp_info info;
    int st = 0;
    if (msg >= 1) printf("running ... \n");
    bslv_poly *lower = NULL;
    const int vflags = (dual1 ? BSLV_VLP_PHASE1_DUAL : 0) | (presol ? BSLV_VLP_PREIMAGES : 0);
    int rc = dual2 ? bslv_vlp_solve_dual2(v->m, v->n, q, v->A, v->P, v->rtype, v->rlb, v->rub, v->ctype, v->clb, v->cub,
                                          v->optdir, v->cone_gen, v->gen, v->n_gen, v->c, bounded, vflags, 1e-8, 1e-8, eps1, eps, batch, &lower, &st, &info)
                   : bslv_vlp_solve_primal(v->m, v->n, q, v->A, v->P, v->rtype, v->rlb, v->rub, v->ctype, v->clb, v->cub,
                                           v->optdir, v->cone_gen, v->gen, v->n_gen, v->c, bounded, vflags, 1e-8, 1e-8, eps1, eps, batch, &h, &st, &info);
    if (rc) { printf("engine error %d: %s\n", rc, bslv_last_error()); return 3; }
    char cfile[1100];
    if (info.c) {                                                         /* bslv_vlp.c:833-842 */
        snprintf(cfile, sizeof cfile, "%s_c.sol", base);
        FILE *cf = fopen(cfile, "w");
        if (cf) { for (int k = 0; k < q; k++) fprintf(cf, k ? " %.14g" : "%.14g", info.c[k]); fprintf(cf, "\n"); fclose(cf); }
    }
    if (st != 4) { printf("%s\n", info.message); bslv_vlp_info_free(&info); bslv_vlp_free(v); return 1; }
    if (msg >= 2 && !bounded) { printf("Result of phase 0: eta =\n "); for (int k = 0; k < q; k++) printf(" %.6g", info.eta[k]); printf("\n"); }
    double elapsed = now() - t0;
    long cnt[4], lps = 0, cuts = 0, piv = 0;
    if ((rc = bslv_sol_write3(dual2 ? lower : bslv_benson_poly(h), base, ".sol", dual2, info.negate_primal, info.negate_dual_last, cnt))) { printf("writing results failed (%d): %s\n", rc, bslv_last_error()); return 3; }
    if (presol && h && (rc = bslv_sol_write_preimages(h, base, ".sol", v->m, v->n, v->optdir, info.c_dir))) { printf("writing the pre-images failed (%d): %s\n", rc, bslv_last_error()); return 3; }
    if (presol && lower && (rc = bslv_sol_write_preimages_dual(lower, base, ".sol", v->m, v->n, v->optdir, info.c_dir))) { printf("writing the pre-images failed (%d): %s\n", rc, bslv_last_error()); return 3; }
    if (h) bslv_benson_totals(h, &lps, &cuts, &piv);
    {   /* <name>.log, fields and layout of bslv_main.c:346-397 */
        char lfile[1100];
        snprintf(lfile, sizeof lfile, "%s.log", base);
        FILE *lf = fopen(lfile, "w");
        if (!lf) { printf("unable to open file %s", lfile); return 1; }
        fprintf(lf, "BENSOLVE: VLP solver, bensolve_hip (MI355X engine behind the BENSOLVE 2.0.1 file formats)\n");
        fprintf(lf, "Problem parameters\n");
        fprintf(lf, "  problem file:      %s\n", file);      /* (the reference prints the log's own name here: a shadowed variable, bslv_main.c:348,367) */
        fprintf(lf, "  problem rows:      %7d\n", v->m);
        fprintf(lf, "  problem columns:   %7d\n", v->n);
        fprintf(lf, "  matrix non-zeros:  %7ld\n", v->nz);
        fprintf(lf, "  primal generators: %7d\n", info.o);
        fprintf(lf, "  dual generators:   %7d\n", info.p);
        fprintf(lf, "Options\n");
        fprintf(lf, "  bounded:            %s\n", bounded ? "yes (run phase 2 only)" : "no (run phases 0 to 2)");
        fprintf(lf, "  solution:           %s\n", presol ? "on (solutions (pre-image) written to files)" : "off (no solution output)");
        fprintf(lf, "  format:             %s\n", "long");
        {
            static const char *const mname[4] = {NULL, "primal_simplex", "dual_simplex", "dual_primal_simplex (dual simplex, if not succesful, primal simplex)"};
            const char *m0 = lp_method[0] ? mname[lp_method[0]] : mname[3];
            const char *m1 = lp_method[1] ? mname[lp_method[1]] : mname[3];
            const char *m2 = lp_method[2] ? mname[lp_method[2]] : (dual2 ? mname[1] : mname[3]);
            fprintf(lf, "  lp_method_phase0:   %s\n", m0);
            fprintf(lf, "  lp_method_phase1:   %s\n", m1);
            fprintf(lf, "  lp_method_phase2:   %s\n", m2);
        }
        fprintf(lf, "  message_level:      %d\n", msg);
        fprintf(lf, "  lp_message_level:   %d\n", 0);
        fprintf(lf, "  alg_phase1:         %s\n", dual1 ? "dual" : "primal");
        fprintf(lf, "  alg_phase2:         %s\n", dual2 ? "dual" : "primal");
        fprintf(lf, "  eps_benson_phase1:  %g\n", eps1);
        fprintf(lf, "  eps_benson_phase2:  %g\n", eps);
        fprintf(lf, "  eps_phase0:         %g\n", 1e-8);
        fprintf(lf, "  eps_phase1:         %g\n", 1e-8);
        fprintf(lf, "Computational results\n");
        fprintf(lf, "  CPU time (ms):      %g\n", elapsed * 1e3);
        fprintf(lf, "  # LPs:              %ld\n", info.lps);
        fprintf(lf, "Solution properties\n");
        fprintf(lf, "  # primal solution points:     %7ld\n", cnt[0]);
        fprintf(lf, "  # primal solution directions: %7ld\n", cnt[1]);
        fprintf(lf, "  # dual solution points:       %7ld\n", cnt[2]);
        fprintf(lf, "  # dual solution directions:   %7ld\n", cnt[3]);
        fclose(lf);
    }
    if (msg >= 1) {
        printf("CPU time            : %.4g %s.\n", elapsed >= 1 ? elapsed : elapsed * 1e3, elapsed >= 1 ? "s" : "ms");
        printf("Number of LPs solved: %ld.\n", info.lps);
    }
    if (msg >= 2) printf("outer iterations %ld, phase-2 cuts %ld, phase-2 pivots %ld; upper image: %ld points, %ld directions; lower image: %ld points, %ld directions\n",
                         info.steps, cuts, piv, cnt[0], cnt[1], cnt[2], cnt[3]);
    if (h) bslv_benson_destroy(h);
    if (lower) { bslv_dual_preimages_free(lower); bslv_poly_destroy(lower); }
    bslv_vlp_info_free(&info);
    bslv_vlp_free(v);
    if (world > 1) bslv_dist_finalize();
    return 0;
}
