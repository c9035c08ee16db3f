// vlp_phases.hip -- the callers around the batched phase-2 driver: ordering-cone data
// (sol_init, bslv_vlp.c:599-864), cone_vertenum (bslv_algs.c:331-407), phase 0 (bslv_algs.c:673-800) and phase 1 of the
// primal algorithm (bslv_algs.c:811-933), and the sequence bslv_main.c:236-345 runs them in.  Host code only: the LPs go
// through the batched LP engine (one LP at a time in phase 0, batches in phase 1), the cones through the polyhedron engine.
//
// Storage conventions are the reference's: a q x k matrix of k generators is stored row-major with the generators as COLUMNS,
// M[j*k + i] = component j of generator i (sol->Z, sol->Y, sol->R, sol->H, vlp->gen).
#include "common.h"
#include <vector>
#include <deque>
#include <unordered_map>
#include <mutex>
#include <algorithm>
#include <cmath>

namespace bslv {
void set_error(const char *fmt, ...);

// ---- cone_vertenum (bslv_algs.c:331-407): `gen` = n_in generators of a cone K in R^dim.  prim: the non-redundant ones among
//      them (dual slots that are used and ideal, slot order); dual: generators of the dual cone (primal ideal elements in the
//      order poly__get_vrtx hands them out = ascending slot).  fail = 1 when the initial approximation fails (K not solid). ----
static int cone_vertenum(const double *gen, int n_in, int dim, std::vector<double> &prim, int &n_prim,
                         std::vector<double> &dual, int &n_dual, int &fail)
{
    fail = 0; n_prim = n_dual = 0;
    bslv_poly *cp = nullptr;
    int rc = bslv_poly_create(&cp, dim, 0 /* cone_polar */, nullptr);
    if (rc) return rc;
    auto done = [&](int r) { bslv_poly_destroy(cp); return r; };
    if ((rc = bslv_poly_dual0_apex(cp))) return done(rc);                         // :338-339
    std::vector<double> val(dim);
    for (int k = 0; k < n_in; k++) {
        for (int j = 0; j < dim; j++) val[j] = gen[(size_t)j * n_in + k];
        int prc;
        if ((rc = bslv_poly_add(cp, val.data(), 1, &prc))) return done(rc);       // :341-349
    }
    int irc = 0;
    if ((rc = bslv_poly_init(cp, &irc))) return done(rc);
    if (irc) { fail = 1; return done(0); }                                        // :350-351
    const int nd = bslv_poly_ndual(cp), np = bslv_poly_nprimal(cp);
    std::vector<unsigned char> du(nd), di(nd), pu(np), pi(np), ps(np);
    std::vector<double> dc((size_t)nd * dim), pc((size_t)np * dim);
    if ((rc = bslv_poly_get_dual(cp, du.data(), di.data(), dc.data()))) return done(rc);
    if ((rc = bslv_poly_get_primal(cp, pu.data(), pi.data(), ps.data(), pc.data()))) return done(rc);
    std::vector<int> a, b;
    for (int k = 0; k < nd; k++) if (du[k] && di[k]) a.push_back(k);              // :353-372
    for (int k = 0; k < np; k++) if (pu[k] && pi[k]) b.push_back(k);              // :374-392
    n_prim = (int)a.size(); n_dual = (int)b.size();
    prim.assign((size_t)dim * n_prim, 0.0);
    dual.assign((size_t)dim * n_dual, 0.0);
    for (int i = 0; i < n_prim; i++) for (int j = 0; j < dim; j++) prim[(size_t)j * n_prim + i] = dc[(size_t)a[i] * dim + j];
    for (int i = 0; i < n_dual; i++) for (int j = 0; j < dim; j++) dual[(size_t)j * n_dual + i] = pc[(size_t)b[i] * dim + j];
    return done(0);
}

struct Sol {                       // the part of soltype (bslv_vlp.h) phases 0-2 pass on
    int q = 0, o = 0, p = 0, r = 0, h = 0;
    std::vector<double> Y, Z, c, R, H, eta;
    int c_dir = 1;                 // +1 C_DIR_POS, -1 C_DIR_NEG
    bool negate_P = false;         // bslv_vlp.c:856-861
};

// sol_init (bslv_vlp.c:599-864).  status: 0 ok, 5 input error (VLP_INPUTERROR) with the reference's message in `msg`.
static int sol_init(Sol &S, int q, int cone_kind, const double *gen, int n_gen, const double *c_in, int optdir, int *status, char *msg, size_t msg_len)
{
    const double EPS_C = 1e-7;
    *status = 0;
    S.q = q;
    S.eta.assign(q, 0.0);
    auto input_error = [&](const char *text) { snprintf(msg, msg_len, "%s", text); *status = 5; return 0; };
    int rc, fail = 0;
    if (cone_kind == 1) {                                                         // generators of C are given (:641-656)
        if ((rc = cone_vertenum(gen, n_gen, q, S.Y, S.o, S.Z, S.p, fail))) return rc;
        if (fail) return input_error("Input error: Ordering cone has empty interior (1)");
        if (S.p < q || S.o < q) return input_error("Input error: Ordering cone is not pointed (2)");
    } else if (cone_kind == 2) {                                                  // generators of C^* are given (:657-672)
        if ((rc = cone_vertenum(gen, n_gen, q, S.Z, S.p, S.Y, S.o, fail))) return rc;
        if (fail) return input_error("Input error: Ordering cone is not pointed (1)");
        if (S.p < q || S.o < q) return input_error("Input error: Ordering cone has empty interior (2)");
    } else {                                                                      // R^q_+ (:673-684)
        S.Y.assign((size_t)q * q, 0.0); S.Z.assign((size_t)q * q, 0.0);
        for (int k = 0; k < q; k++) S.Y[(size_t)k * q + k] = S.Z[(size_t)k * q + k] = 1.0;
        S.p = S.o = q;
    }
    S.c.assign(q, 0.0);
    if (cone_kind == 0) { for (int j = 0; j < q; j++) S.c[j] = 1.0; S.c_dir = 1; }
    else {
        for (int k = 0; k < S.o; k++) {                                           // columns of Y to 2-norm 1 (:697-706)
            double t = 0;
            for (int j = 0; j < q; j++) t += S.Y[k + (size_t)j * S.o] * S.Y[k + (size_t)j * S.o];
            for (int j = 0; j < q; j++) S.Y[k + (size_t)j * S.o] /= sqrt(t);
        }
        if (c_in && fabs(c_in[q - 1]) > EPS_C) {                                  // given c, scaled to |c_q| = 1 (:708-713)
            for (int i = 0; i < q; i++) S.c[i] = c_in[i] / fabs(c_in[q - 1]);
            S.c_dir = c_in[q - 1] > 0 ? 1 : -1;
        } else {                                                                  // generated c (:714-776)
            std::vector<double> t1(q, 0.0), t2(q, 0.0);
            double mx = 0, mn = 0;
            int k1 = 0, k2 = 0;
            for (int i = 0; i < S.o; i++) {
                const double last = S.Y[(size_t)(q - 1) * S.o + i];
                if (last > 0) { mx = std::max(mx, last); for (int j = 0; j < q; j++) t1[j] += S.Y[(size_t)j * S.o + i]; k1++; }
                else { mn = std::min(mn, last); for (int j = 0; j < q; j++) t2[j] += S.Y[(size_t)j * S.o + i]; k2++; }
            }
            if (k1 == 0 && mn < EPS_C) { S.c_dir = -1; for (int i = 0; i < q; i++) S.c[i] = t2[i] / fabs(t2[q - 1]); }
            else if (k2 == 0 && mx > EPS_C) { S.c_dir = 1; for (int i = 0; i < q; i++) S.c[i] = t1[i] / fabs(t1[q - 1]); }
            else if (mn < -EPS_C || mx > EPS_C) {
                double lambda;
                if (-mn > mx) { S.c_dir = -1; lambda = 0.2 * (-mn / (mx - mn)); }
                else { S.c_dir = 1; lambda = 0.8 - 0.2 * mn / (mx - mn); }
                for (int i = 0; i < q; i++) S.c[i] = lambda * t1[i] / k1 + (1 - lambda) * t2[i] / k2;
                const double s = fabs(S.c[q - 1]);
                for (int i = 0; i < q; i++) S.c[i] /= s;
            } else return input_error("Input error: ordering cone is not solid (3)");
        }
    }
    for (int k = 0; k < S.p; k++) {                                               // Z' c = (1..1) (:779-795)
        double t = 0;
        for (int j = 0; j < q; j++) t += S.Z[k + (size_t)j * S.p] * S.c[j];
        if (t < 1e-8) return input_error("Input error: c does not belong to interior of ordering cone");
        for (int j = 0; j < q; j++) S.Z[k + (size_t)j * S.p] /= t;
    }
    if (cone_kind != 0) {                                                         // further tests (:798-830)
        std::vector<double> sy(q, 0.0), sz(q, 0.0);
        for (int j = 0; j < q; j++) { for (int k = 0; k < S.o; k++) sy[j] += S.Y[(size_t)S.o * j + k]; for (int k = 0; k < S.p; k++) sz[j] += S.Z[(size_t)S.p * j + k]; }
        for (int k = 0; k < S.p; k++) { double t = 0; for (int j = 0; j < q; j++) t += S.Z[(size_t)S.p * j + k] * sy[j]; if (t < 1e-8) return input_error("Input error: ordering cone is not solid (4)"); }
        for (int k = 0; k < S.o; k++) { double t = 0; for (int j = 0; j < q; j++) t += S.Y[(size_t)S.o * j + k] * sz[j]; if (t < 1e-8) return input_error("Input error: ordering cone is not pointed (4)"); }
    }
    return 0;
}
// second half of sol_init, after c has been written out (:844-861): standard problem "min, c_q > 0"
static void sol_normalise(Sol &S, int optdir)
{
    if (S.c_dir < 0) { for (double &v : S.Y) v = -v; for (double &v : S.Z) v = -v; for (double &v : S.c) v = -v; }
    S.negate_P = (S.c_dir < 0 && optdir == 1) || (S.c_dir > 0 && optdir == -1);
}

struct Problem { int m, n, q; const double *A, *P; const char *rtype; const double *rlb, *rub; const char *ctype; const double *clb, *cub; };

// one LP in slot 0, in place, with the reference's retry (bslv_lp.c:222-227: undefined -> standard basis -> again)
static int solve0(bslv_lpq *lp, int p, const double *ub, int *st)
{
    const int zero = 0;
    std::vector<double> vlo(p, -INFINITY);
    int it, rc;
    if ((rc = bslv_lpq_solve_batch(lp, 1, &zero, &zero, vlo.data(), ub, st, &it))) return rc;
    if (*st == BSLV_LP_UNDEFINED) {
        if ((rc = bslv_lpq_reset_slot(lp, 0))) return rc;
        if ((rc = bslv_lpq_solve_batch(lp, 1, &zero, &zero, vlo.data(), ub, st, &it))) return rc;
    }
    if (*st == BSLV_LP_UNDEFINED && !bslv_lpq_get_extended(lp)) {
        // still nothing from the standard basis: a dual simplex stalling in degenerate pivots.  The extended selection (cost
        // perturbation, primal clean-up) from here on, for every LP of this engine
        if ((rc = bslv_lpq_set_extended(lp, 1))) return rc;
        if ((rc = bslv_lpq_reset_slot(lp, 0))) return rc;
        if ((rc = bslv_lpq_solve_batch(lp, 1, &zero, &zero, vlo.data(), ub, st, &it))) return rc;
    }
    return 0;
}

// replaces column i of the d x d matrix C (C[k*d + i]) by a unit vector orthogonal to columns 0..i-1 (which are orthogonal
// to each other): Gram-Schmidt on e_i, e_{i+1}, ... until something of length > 1e-3/2 is left (orthogonal_vector,
// bslv_lists.c:113-143)
static void orthogonal_column(std::vector<double> &C, int d, int i)
{
    double len2 = 0;
    for (int t = 0; t < d; t++) {
        for (int k = 0; k < d; k++) C[(size_t)k * d + i] = 0;
        C[(size_t)((i + t) % d) * d + i] = 1;
        for (int j = 0; j < i; j++) {
            double s = 0, s1 = 0;
            for (int k = 0; k < d; k++) { s += C[(size_t)k * d + j] * C[(size_t)k * d + i]; s1 += C[(size_t)k * d + j] * C[(size_t)k * d + j]; }
            for (int k = 0; k < d; k++) C[(size_t)k * d + i] -= s / s1 * C[(size_t)k * d + j];
        }
        len2 = 0;
        for (int k = 0; k < d; k++) len2 += C[(size_t)k * d + i] * C[(size_t)k * d + i];
        if (len2 > 1e-3) break;
    }
    for (int k = 0; k < d; k++) C[(size_t)k * d + i] /= sqrt(len2);
}

// ---- phase 0 (bslv_algs.c:673-800): eta in int(D* + K) with c.eta = 1.  status: 0 ok, 2 VLP_UNBOUNDED (totally unbounded),
//      3 VLP_NOVERTEX. ----
static int phase0(const Problem &pb, Sol &S, double eps_phase0, int *status, long *lps)
{
    *status = 0;
    const int q = pb.q, p = S.p, d = q - 1;
    bslv_benson *h = nullptr;
    int rc = bslv_benson_create_ex(&h, pb.m, pb.n, q, pb.A, pb.P, pb.rtype, pb.rlb, pb.rub, pb.ctype, pb.clb, pb.cub,
                                   S.Z.data(), p, S.c.data(), nullptr /* eta = 0 */, 1, 0, 1e-7, 4);
    if (rc) return rc;
    auto done = [&](int r) { bslv_benson_destroy(h); return r; };
    bslv_lpq *lp = bslv_benson_lp(h);
    int M, N, folded;
    bslv_benson_lp_dims(h, &M, &N, &folded);
    const int yrow0 = M - 1 - p - q;                // first of the q rows -P x + y = 0 (after the presolve)
    const int zero = 0;
    if ((rc = bslv_lpq_reset_slot(lp, 0))) return done(rc);
    std::vector<double> ub(p, 0.0), z(d), wred(d), V((size_t)d * d, 0.0), C((size_t)d * d, 0.0);
    int st;
    if ((rc = solve0(lp, p, ub.data(), &st))) return done(rc);                                          // :689-698
    if (st == BSLV_LP_UNBOUNDED) { *status = 2; return done(0); }
    if (st != BSLV_LP_OPTIMAL) { set_error("phase 0: first LP has status %d (the reference asserts optimality, bslv_algs.c:697)", st); return done(BSLV_E_STATE); }
    ++*lps;
    if ((rc = bslv_lpq_get_dual(lp, 1, &zero, yrow0, d, z.data()))) return done(rc);                     // :699
    auto solve_dir = [&](int i, double sign) -> int {                                                   // P_2(+-[C(i);0]) (:702-718, 731-747)
        for (int j = 0; j < p; j++) {
            double s = 0;
            for (int k = 0; k < d; k++) s += S.Z[(size_t)k * p + j] * C[(size_t)k * d + i];
            ub[j] = sign * s;
        }
        int r2 = solve0(lp, p, ub.data(), &st);
        if (r2) return r2;
        if (st != BSLV_LP_OPTIMAL) { set_error("phase 0: LP %d has status %d (the reference asserts optimality, bslv_algs.c:712)", i, st); return BSLV_E_STATE; }
        ++*lps;
        if ((r2 = bslv_lpq_get_dual(lp, 1, &zero, yrow0, d, wred.data()))) return r2;
        for (int k = 0; k < d; k++) V[(size_t)k * d + i] = wred[k] - z[k];
        return 0;
    };
    auto dotCV = [&](int i) { double t = 0; for (int k = 0; k < d; k++) t += C[(size_t)k * d + i] * V[(size_t)k * d + i]; return t; };
    for (int i = 0; i < d; i++) {
        orthogonal_column(C, d, i);
        if ((rc = solve_dir(i, 1.0))) return done(rc);
        if (fabs(dotCV(i)) < eps_phase0) if ((rc = solve_dir(i, -1.0))) return done(rc);                // :724-748
        if (fabs(dotCV(i)) < eps_phase0) { *status = 3; return done(0); }                               // :750-760
        // C(i) = V(i) - sum_j <C(j),V(i)>/<C(j),C(j)> C(j)   (:761-780)
        std::vector<double> e(d, 0.0);
        for (int j = 0; j < i; j++) {
            double t1 = 0, t2 = 0;
            for (int k = 0; k < d; k++) { t1 += C[(size_t)k * d + j] * V[(size_t)k * d + i]; t2 += C[(size_t)k * d + j] * C[(size_t)k * d + j]; }
            for (int k = 0; k < d; k++) e[k] -= t1 / t2 * C[(size_t)k * d + j];
        }
        for (int k = 0; k < d; k++) C[(size_t)k * d + i] = e[k] + V[(size_t)k * d + i];
    }
    // mean of 0, V(0..q-2), plus z; last component from c.eta = 1   (:782-798)
    S.eta.assign(q, 0.0);
    for (int i = 0; i < d; i++) for (int k = 0; k < d; k++) S.eta[k] += V[(size_t)k * d + i];
    for (int k = 0; k < d; k++) S.eta[k] = S.eta[k] / q + z[k];
    S.eta[q - 1] = 1.0;
    for (int k = 0; k < d; k++) S.eta[q - 1] -= S.c[k] * S.eta[k];
    return done(0);
}

// runs a started engine to termination (poly__get_vrtx returns "none left")
static int run_engine(bslv_benson *h, int batch, long *lps, long *steps)
{
    long st[8]; double ms[3];
    int rc;
    do {
        if ((rc = bslv_benson_step(h, batch, st, ms))) return rc;
        ++*steps;
    } while (st[0] > 0 || st[7] > 0);
    long a = 0, b = 0, c = 0;
    bslv_benson_totals(h, &a, &b, &c);
    *lps += a;
    return 0;
}

// ---- phase 1, primal algorithm (bslv_algs.c:811-933): Benson on the homogeneous problem, then R = generators of the dual of
//      the recession cone of the upper image (dual vertices with last component 0), H = generators of that cone. ----
static int phase1_primal(const Problem &pb, Sol &S, double eps_phase1, double eps_benson, int batch, long *lps, long *steps)
{
    const int q = pb.q;
    bslv_benson *h = nullptr;
    int rc = bslv_benson_create_ex(&h, pb.m, pb.n, q, pb.A, pb.P, pb.rtype, pb.rlb, pb.rub, pb.ctype, pb.clb, pb.cub,
                                   S.Z.data(), S.p, S.c.data(), S.eta.data(), 1, 0, eps_benson, std::max(4 * batch + 64, 64));
    if (rc) return rc;
    auto done = [&](int r) { bslv_benson_destroy(h); return r; };
    int vst = 0;
    if ((rc = bslv_benson_start(h, &vst))) return done(rc);                                             // PART 1 (:829-851)
    if (vst) { set_error("phase 1: an initialisation LP is not optimal (the reference asserts, bslv_algs.c:844)"); return done(BSLV_E_STATE); }
    if ((rc = run_engine(h, batch, lps, steps))) return done(rc);                                       // PART 2 (:853-899)
    bslv_poly *poly = bslv_benson_poly(h);                                                              // PART 3 (:901-928)
    const int nd = bslv_poly_ndual(poly);
    std::vector<unsigned char> du(nd), di(nd);
    std::vector<double> dc((size_t)nd * q);
    if ((rc = bslv_poly_get_dual(poly, du.data(), di.data(), dc.data()))) return done(rc);
    std::vector<int> sel;
    for (int l = 0; l < nd; l++) if (du[l] && !di[l] && fabs(dc[(size_t)l * q + q - 1]) < eps_phase1) sel.push_back(l);
    const int pp = (int)sel.size();
    std::vector<double> arr((size_t)q * pp);
    for (int i = 0; i < pp; i++) {
        double last = 1.0;
        for (int j = 0; j < q - 1; j++) { const double v = dc[(size_t)sel[i] * q + j]; arr[(size_t)j * pp + i] = v; last -= S.c[j] * v; }
        arr[(size_t)(q - 1) * pp + i] = last;
    }
    int fail = 0;
    if ((rc = cone_vertenum(arr.data(), pp, q, S.R, S.r, S.H, S.h, fail))) return done(rc);
    if (fail || S.r < 1) { set_error("phase 1: the recession cone data could not be enumerated (%d candidate generators)", pp); return done(BSLV_E_STATE); }
    return done(0);
}


// ---- phase 2, dual algorithm (bslv_algs.c:1381-1592): outer approximation of the LOWER image.  Each vertex y* of the current
//      approximation gives the weights w(y*) = (y*_1..y*_{q-1}, 1 - sum c_i y*_i) of P1(w): min w.y, y = Px, x feasible
//      (init_P1, :1186-1238); its optimal y cuts y* off when y*_q - w.y > eps.  The LPs of a batch differ only in the
//      objective: bslv_lpq_solve_batch_obj, all warm-started from the first solved basis (slot 0).
//      status: 0 ok, 1 VLP_INFEASIBLE, 2 VLP_UNBOUNDED.  The polyhedron (primal side = lower image) is handed to the caller. ----
//      hom != 0: the homogeneous problem of phase1_dual (:1248-1371; init_P1(..., HOMOGENEOUS): bounds zeroed, one more row
//      eta.y <= 1), started with the mean of the columns of Z and the generators Y of the ordering cone as directions.
// option -s with the dual algorithm (phase2_dual with opt->solution == PRE_IMG_ON, bslv_algs.c:1388-1389, 1431-1432, 1488-1546):
// x (n values) of every vertex of the UPPER image -- a cut of the lower image, i.e. a dual slot of the polyhedron -- and
// (u, w) (m + q values) of every confirmed vertex of the LOWER image (a primal element)
struct DualPreimg {
    int m = 0, n = 0, q = 0;
    std::unordered_map<int, std::vector<double>> x_by_facet, uw_by_element;
    std::vector<std::pair<std::vector<double>, std::vector<double>>> start;       // (y, x) of PART 1, until the dual slots exist
};
static std::mutex g_dpre_mu;
static std::unordered_map<const bslv_poly *, DualPreimg *> g_dpre;

static int dual_benson(const Problem &pb, const Sol &S, int hom, double eps, int batch, bslv_poly **poly_out, int *status, long *lps, long *steps,
                       DualPreimg *store = nullptr)
{
    *status = 0; *poly_out = nullptr;
    const int m = pb.m, n = pb.n, q = pb.q, M = m + q + (hom ? 1 : 0), N = n + q;
    const std::vector<double> &W0 = hom ? S.Z : S.R, &D0 = hom ? S.Y : S.H;
    const int nw0 = hom ? S.p : S.r, nd0 = hom ? S.o : S.h;
    std::vector<double> L((size_t)M * N, 0.0), lo(M + N), up(M + N), cost(N + 1, 0.0);
    for (int i = 0; i < m; i++) memcpy(&L[(size_t)i * N], pb.A + (size_t)i * n, n * sizeof(double));
    for (int k = 0; k < q; k++) {
        for (int j = 0; j < n; j++) L[(size_t)(m + k) * N + j] = -pb.P[(size_t)k * n + j];
        L[(size_t)(m + k) * N + n + k] = 1.0;
    }
    auto bnd = [hom](char t, double lb, double ub, double *l, double *u) {
        if (hom) { lb = ub = 0.0; if (t == 'd') t = 's'; }                           // lp_set_rows_hom / lp_set_cols_hom (bslv_lp.c:118-134)
        *l = (t == 'l' || t == 'd' || t == 's') ? lb : -INFINITY;
        *u = (t == 'u' || t == 'd') ? ub : (t == 's' ? lb : INFINITY);
    };
    for (int i = 0; i < m; i++) bnd(pb.rtype[i], pb.rlb ? pb.rlb[i] : 0, pb.rub ? pb.rub[i] : 0, &lo[i], &up[i]);
    for (int k = 0; k < q; k++) lo[m + k] = up[m + k] = 0.0;
    if (hom) { for (int k = 0; k < q; k++) L[(size_t)(m + q) * N + n + k] = S.eta[k]; lo[m + q] = -INFINITY; up[m + q] = 1.0; }   // :1201-1226
    for (int j = 0; j < n; j++) bnd(pb.ctype[j], pb.clb ? pb.clb[j] : 0, pb.cub ? pb.cub[j] : 0, &lo[M + j], &up[M + j]);
    for (int k = 0; k < q; k++) { lo[M + n + k] = -INFINITY; up[M + n + k] = INFINITY; }
    bslv_lpq *lp = nullptr;
    bslv_poly *poly = nullptr;
    const int pool = 3 * batch + 8;      // slot 0: the first optimal basis; the others: tableaux of the LPs whose cuts are still young
    int rc = bslv_lpq_create(&lp, M, N, L.data(), lo.data(), up.data(), cost.data(), 0, 0, pool);
    if (rc) return rc;
    auto done = [&](int r) { if (lp) bslv_lpq_destroy(lp); if (poly && r) { bslv_poly_destroy(poly); poly = nullptr; } return r; };
    if ((rc = bslv_poly_create(&poly, q, 2 /* upperV2lowerH */, S.c.data()))) return done(rc);
    const int zero = 0;
    int st, it;
    // a primal feasible basis first (zero objective: every basis is dual feasible, the dual simplex repairs the bounds) ...
    if ((rc = bslv_lpq_reset_slot(lp, 0))) return done(rc);
    if ((rc = bslv_lpq_solve_batch(lp, 1, &zero, &zero, nullptr, nullptr, &st, &it))) return done(rc);
    if (st == BSLV_LP_INFEASIBLE) { *status = 1; bslv_poly_destroy(poly); poly = nullptr; return done(0); }
    if (st != BSLV_LP_OPTIMAL) { set_error("phase 2 (dual): the feasibility LP has status %d", st); return done(BSLV_E_STATE); }
    // ... then PART 1 (:1397-1443): w = mean of the columns of R
    std::vector<double> w(q, 0.0), y(q), val(q);
    for (int i = 0; i < q; i++) { for (int j = 0; j < nw0; j++) w[i] += W0[(size_t)i * nw0 + j]; w[i] /= nw0; }
    if ((rc = bslv_lpq_solve_batch_obj(lp, 1, &zero, &zero, nullptr, nullptr, M + n, q, w.data(), &st, &it))) return done(rc);
    if (st != BSLV_LP_OPTIMAL) { *status = st == BSLV_LP_INFEASIBLE ? 1 : 2; bslv_poly_destroy(poly); poly = nullptr; return done(0); }
    ++*lps;
    if ((rc = bslv_lpq_get_primal(lp, 1, &zero, M + n, q, y.data()))) return done(rc);
    if (store) {                                                                     // x of the first vertex (:1431-1432)
        std::vector<double> x(n);
        if ((rc = bslv_lpq_get_primal(lp, 1, &zero, M, n, x.data()))) return done(rc);
        store->start.emplace_back(y, x);
    }
    int prc;
    if ((rc = bslv_poly_add(poly, y.data(), 0, &prc))) return done(rc);
    for (int j = 0; j < nd0; j++) {                                                  // the (recession | ordering) cone's generators as directions
        for (int i = 0; i < q; i++) val[i] = D0[(size_t)i * nd0 + j];
        if ((rc = bslv_poly_add(poly, val.data(), 1, &prc))) return done(rc);
    }
    int irc = 0;
    if ((rc = bslv_poly_init(poly, &irc))) return done(rc);
    if (irc) { set_error("phase 2 (dual): initial outer approximation failed (bslv_poly.c:174)"); return done(BSLV_E_STATE); }
    if (store) {
        // poly__intl_apprx re-adds queued halfspaces as new dual slots (bslv_poly.c:190-197): find the first vertex by its coordinates
        const int nd = bslv_poly_ndual(poly);
        std::vector<unsigned char> du(nd), di(nd);
        std::vector<double> dc((size_t)nd * q);
        if ((rc = bslv_poly_get_dual(poly, du.data(), di.data(), dc.data()))) return done(rc);
        for (int f = 0; f < nd; f++) {
            if (!du[f] || di[f]) continue;
            for (auto &sp : store->start) {
                double dd = 0;
                for (int k = 0; k < q; k++) dd = std::max(dd, std::fabs(sp.first[k] - dc[(size_t)f * q + k]));
                if (dd <= 1e-12 * (1.0 + std::fabs(sp.first[q - 1]))) { store->x_by_facet[f] = sp.second; break; }
            }
        }
    }
    // PART 2 (:1445-1500), batched.  Warm starts: a vertex y* of the lower image was created by a cut, i.e. by the optimal y of
    // an earlier P1(w'); w(y*) is close to w', so that LP's basis (kept in a tableau slot as long as the pool allows, oldest
    // evicted first, at most 64 generations deep) is the start -- a handful of primal pivots instead of hundreds from the root.
    std::vector<int> idx(batch), ideal(batch), parent(batch), src(batch, 0), dst(batch), stv(batch), itv(batch), rcv(batch), marks;
    std::vector<double> vals((size_t)batch * q), W((size_t)batch * q), Y((size_t)batch * q), obj(batch), cuts;
    std::vector<int> free_slots, gen(pool, 0);
    for (int sl = pool - 1; sl >= 1; sl--) free_slots.push_back(sl);
    std::unordered_map<int, int> facet_slot;
    std::deque<std::pair<int, int>> young;                                           // (facet, slot), oldest first
    std::vector<char> is_src(pool, 0);
    for (;;) {
        int cnt = 0;
        if ((rc = bslv_poly_unprocessed2(poly, batch, 0, idx.data(), vals.data(), ideal.data(), parent.data(), &cnt))) return done(rc);
        const int nb = std::min(cnt, batch);
        if (nb == 0) break;
        ++*steps;
        marks.clear();
        std::vector<int> pts;                                                        // positions of the points of this batch
        for (int k = 0; k < nb; k++) { if (ideal[k]) marks.push_back(idx[k]); else pts.push_back(k); }   // :1456-1459
        const int np = (int)pts.size();
        std::fill(is_src.begin(), is_src.end(), 0);
        for (int t = 0; t < np; t++) {                                               // w(y*) (:1461-1467)
            const double *v = &vals[(size_t)pts[t] * q];
            double last = 1.0;
            for (int i = 0; i < q - 1; i++) { W[(size_t)t * q + i] = v[i]; last -= v[i] * S.c[i]; }
            W[(size_t)t * q + q - 1] = last;
            auto it2 = facet_slot.find(parent[pts[t]]);
            src[t] = (it2 != facet_slot.end() && gen[it2->second] < 64) ? it2->second : 0;
            is_src[src[t]] = 1;
        }
        for (int t = 0; t < np; t++) {
            while (free_slots.empty()) {                                             // evict the oldest tableau that is not a start of this batch
                if (young.empty()) { set_error("phase 2 (dual): tableau pool exhausted"); return done(BSLV_E_NOMEM); }
                auto pr = young.front(); young.pop_front();
                auto it2 = facet_slot.find(pr.first);
                if (it2 != facet_slot.end() && it2->second == pr.second) facet_slot.erase(it2);
                if (is_src[pr.second]) young.emplace_back(-1, pr.second);            // still needed by this batch: comes up again later
                else free_slots.push_back(pr.second);
            }
            dst[t] = free_slots.back(); free_slots.pop_back();
            gen[dst[t]] = gen[src[t]] + 1;
        }
        if (np > 0) {
            if ((rc = bslv_lpq_solve_batch_obj(lp, np, src.data(), dst.data(), nullptr, nullptr, M + n, q, W.data(), stv.data(), itv.data()))) return done(rc);
            for (int t = 0; t < np; t++) if (stv[t] != BSLV_LP_OPTIMAL) {
                if (stv[t] == BSLV_LP_UNBOUNDED) { *status = 2; bslv_poly_destroy(poly); poly = nullptr; return done(0); }      // :1472-1477
                set_error("phase 2 (dual): LP status %d (the reference asserts unboundedness here, bslv_algs.c:1474)", stv[t]);
                return done(BSLV_E_STATE);
            }
            *lps += np;
            if ((rc = bslv_lpq_get_primal(lp, np, dst.data(), M + n, q, Y.data()))) return done(rc);
            if ((rc = bslv_lpq_get_obj(lp, np, dst.data(), obj.data()))) return done(rc);
            cuts.clear();
            std::vector<int> cut_src, cut_slot, cut_t;
            // (u, w) of a confirmed vertex: u = duals of the m rows of A (the reference reads lp_dual_solution_COLS 1..m here,
            // :1493 -- reduced costs of the first m columns, which is u only by accident of the indices and runs past the
            // columns when m > n + q; the rows are what the file is documented to hold), w = the weights of its LP
            auto store_uw = [&](int t, int element) -> int {
                std::vector<double> uw(m + q);
                const int sl = dst[t];
                int r2 = bslv_lpq_get_dual(lp, 1, &sl, 0, m, uw.data());
                if (r2) return r2;
                for (int k = 0; k < q; k++) uw[m + k] = W[(size_t)t * q + k];
                store->uw_by_element[element] = uw;
                return 0;
            };
            for (int t = 0; t < np; t++) {
                const double opt_val = vals[(size_t)pts[t] * q + q - 1];
                if (opt_val - obj[t] > eps) { cuts.insert(cuts.end(), &Y[(size_t)t * q], &Y[(size_t)t * q] + q); cut_src.push_back(idx[pts[t]]); cut_slot.push_back(dst[t]); cut_t.push_back(t); }   // :1479-1486
                else {                                                                                                                  // :1488-1497
                    marks.push_back(idx[pts[t]]);
                    if (store && (rc = store_uw(t, idx[pts[t]]))) return done(rc);
                    free_slots.push_back(dst[t]);
                }
            }
            const int nc = (int)cut_src.size();
            if (nc > 0) {
                const int f0 = bslv_poly_ndual(poly);                                // facet ids of the new cuts: f0, f0 + 1, ...
                if ((rc = bslv_poly_add_cuts(poly, nc, cuts.data(), nullptr, rcv.data()))) return done(rc);
                for (int k = 0; k < nc; k++) {
                    if (rcv[k]) {                                                    // nothing was cut off: the vertex stays, processed
                        marks.push_back(cut_src[k]);
                        if (store && (rc = store_uw(cut_t[k], cut_src[k]))) return done(rc);
                        free_slots.push_back(cut_slot[k]);
                    } else {
                        facet_slot[f0 + k] = cut_slot[k]; young.emplace_back(f0 + k, cut_slot[k]);
                        if (store) {                                                 // x of the new vertex of the upper image (:1484-1485)
                            std::vector<double> x(n);
                            const int sl = cut_slot[k];
                            if ((rc = bslv_lpq_get_primal(lp, 1, &sl, M, n, x.data()))) return done(rc);
                            store->x_by_facet[f0 + k] = x;
                        }
                    }
                }
            }
        }
        if (!marks.empty() && (rc = bslv_poly_mark(poly, (int)marks.size(), marks.data()))) return done(rc);
    }
    *poly_out = poly;
    return done(0);
}

static int phase2_dual(const Problem &pb, const Sol &S, double eps, int batch, bslv_poly **poly_out, int *status, long *lps, long *steps, DualPreimg *store = nullptr)
{
    return dual_benson(pb, S, 0, eps, batch, poly_out, status, lps, steps, store);
}

// ---- phase 1, dual algorithm (bslv_algs.c:1248-1371): the dual variant on the homogeneous problem; R from the vertices of
//      the lower image with last component 0 (now on the PRIMAL side of the polyhedron), H their dual cone. ----
static int phase1_dual(const Problem &pb, Sol &S, double eps_phase1, double eps_benson, int batch, long *lps, long *steps)
{
    const int q = pb.q;
    bslv_poly *poly = nullptr;
    int st = 0;
    int rc = dual_benson(pb, S, 1, eps_benson, batch, &poly, &st, lps, steps);
    if (rc) return rc;
    if (st || !poly) { set_error("phase 1 (dual): an LP of the homogeneous problem is not optimal (the reference asserts, bslv_algs.c:1282)"); return BSLV_E_STATE; }
    auto done = [&](int r) { bslv_poly_destroy(poly); return r; };
    const int np = bslv_poly_nprimal(poly);
    std::vector<unsigned char> pu(np), pi(np), ps(np);
    std::vector<double> pc((size_t)np * q);
    if ((rc = bslv_poly_get_primal(poly, pu.data(), pi.data(), ps.data(), pc.data()))) return done(rc);
    std::vector<int> sel;
    for (int l = 0; l < np; l++) if (pu[l] && !pi[l] && fabs(pc[(size_t)l * q + q - 1]) < eps_phase1) sel.push_back(l);     // :1344-1357
    const int pp = (int)sel.size();
    std::vector<double> arr((size_t)q * pp);
    for (int i = 0; i < pp; i++) {
        double last = 1.0;
        for (int j = 0; j < q - 1; j++) { const double v = pc[(size_t)sel[i] * q + j]; arr[(size_t)j * pp + i] = v; last -= S.c[j] * v; }
        arr[(size_t)(q - 1) * pp + i] = last;
    }
    int fail = 0;
    if ((rc = cone_vertenum(arr.data(), pp, q, S.R, S.r, S.H, S.h, fail))) return done(rc);
    if (fail || S.r < 1) { set_error("phase 1 (dual): the recession cone data could not be enumerated (%d candidate generators)", pp); return done(BSLV_E_STATE); }
    return done(0);
}

}  // namespace bslv

using namespace bslv;

extern "C" {

void bslv_free(void *p) { free(p); }

static double *dup_vec(const std::vector<double> &v)
{
    double *p = (double *)malloc(std::max<size_t>(v.size(), 1) * sizeof(double));
    if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(double));
    return p;
}

int bslv_cone_vertenum(const double *gen, int n_in, int dim, double **prim, int *n_prim, double **dual, int *n_dual, int *rc_out)
{
    if (!gen || n_in < 1 || dim < 2 || !prim || !n_prim || !dual || !n_dual || !rc_out) { set_error("bslv_cone_vertenum: bad argument"); return BSLV_E_ARG; }
    std::vector<double> a, b;
    int fail = 0;
    int rc = cone_vertenum(gen, n_in, dim, a, *n_prim, b, *n_dual, fail);
    if (rc) return rc;
    *rc_out = fail;
    *prim = fail ? nullptr : dup_vec(a);
    *dual = fail ? nullptr : dup_vec(b);
    return 0;
}

// pre-images of the extreme DIRECTIONS of the upper image (bslv_algs.c:1083-1112): x of the homogeneous P2 with the bound
// Z'd on the cone rows and the eta row switched off, one LP per direction
static int direction_preimages(const Problem &pb, const Sol &S, bslv_benson *eng, long *lps)
{
    const int q = pb.q, p = S.p;
    bslv_poly *poly = bslv_benson_poly(eng);
    const int np = bslv_poly_nprimal(poly);
    std::vector<unsigned char> pu(np), pi(np), ps(np);
    std::vector<double> pc((size_t)np * q);
    int rc;
    if ((rc = bslv_poly_get_primal(poly, pu.data(), pi.data(), ps.data(), pc.data()))) return rc;
    bslv_benson *hh = nullptr;
    if ((rc = bslv_benson_create_ex(&hh, pb.m, pb.n, q, pb.A, pb.P, pb.rtype, pb.rlb, pb.rub, pb.ctype, pb.clb, pb.cub,
                                    S.Z.data(), p, S.c.data(), nullptr, 1, 0, 1e-7, 4))) return rc;
    auto done = [&](int r) { bslv_benson_destroy(hh); return r; };
    bslv_lpq *lp = bslv_benson_lp(hh);
    int M, N, folded;
    bslv_benson_lp_dims(hh, &M, &N, &folded);
    if ((rc = bslv_lpq_reset_slot(lp, 0))) return done(rc);
    const int zero = 0;
    std::vector<double> ub(p), x(pb.n);
    for (int i = 0; i < np; i++) {
        if (!pu[i] || !pi[i]) continue;
        for (int j = 0; j < p; j++) { double s = 0; for (int k = 0; k < q; k++) s += S.Z[(size_t)k * p + j] * pc[(size_t)i * q + k]; ub[j] = s; }
        int st;
        if ((rc = solve0(lp, p, ub.data(), &st))) return done(rc);
        if (st != BSLV_LP_OPTIMAL) { set_error("pre-image of a direction: LP status %d (the reference asserts optimality, bslv_algs.c:1107)", st); return done(BSLV_E_STATE); }
        ++*lps;
        if ((rc = bslv_lpq_get_primal(lp, 1, &zero, M, pb.n, x.data()))) return done(rc);
        if ((rc = bslv_benson_set_preimage_p(eng, i, x.data()))) return done(rc);
    }
    return done(0);
}

// Shared front of the two entry points: sol_init, the sign normalisation, phases 0 and 1 (primal algorithm) unless bounded.
// Returns with *vlp_status != 0 when the run ends here (input error, totally unbounded, no vertex).
static int front(int m, int n, int q, const double *A, const double *P, const char *rtype, const double *rlb, const double *rub,
                 const char *ctype, const double *clb, const double *cub, int optdir, int cone_kind, const double *gen, int n_gen,
                 const double *c_in, int bounded, double eps_phase0, double eps_phase1, double eps_benson_phase1, int batch, int dual1,
                 Sol &S, std::vector<double> &Pn, long *lps, long *steps, int *vlp_status, bslv_vlp_info *info)
{
    *vlp_status = 0;
    if (info) memset(info, 0, sizeof *info);
    char msg[200] = "";
    int rc, st = 0;
    if ((rc = sol_init(S, q, cone_kind, gen, n_gen, c_in, optdir, &st, msg, sizeof msg))) return rc;
    if (st) { *vlp_status = st; if (info) snprintf(info->message, sizeof info->message, "%s", msg); return 0; }
    if (info) { info->q = q; info->c_dir = S.c_dir; info->c = dup_vec(S.c); }      // c as written to _c.sol: before the sign change
    sol_normalise(S, optdir);
    Pn.assign(P, P + (size_t)q * n);
    if (S.negate_P) for (double &v : Pn) v = -v;
    Problem pb{m, n, q, A, Pn.data(), rtype, rlb, rub, ctype, clb, cub};
    if (bounded) { S.R = S.Z; S.r = S.p; S.H = S.Y; S.h = S.o; }                     // phase2_init (bslv_algs.c:943-956)
    else {
        if ((rc = phase0(pb, S, eps_phase0, &st, lps))) return rc;
        if (st) { *vlp_status = st; if (info) { info->lps = *lps; snprintf(info->message, sizeof info->message, "%s", st == 2 ? "VLP is totally unbounded, there is no solution" : "upper image of VLP has no vertex (this case is not covered by this version)"); } return 0; }
        if ((rc = dual1 ? phase1_dual(pb, S, eps_phase1, eps_benson_phase1, batch, lps, steps)
                        : phase1_primal(pb, S, eps_phase1, eps_benson_phase1, batch, lps, steps))) return rc;
    }
    return 0;
}
static void fill_info(bslv_vlp_info *info, const Sol &S, int optdir, long lps, long steps)
{
    if (!info) return;
    info->lps = lps; info->steps = steps;
    info->o = S.o; info->p = S.p; info->r = S.r; info->h = S.h;
    info->eta = dup_vec(S.eta); info->R = dup_vec(S.R); info->H = dup_vec(S.H); info->Y = dup_vec(S.Y); info->Z = dup_vec(S.Z);
    // poly_trans_primal / poly_trans_dual (bslv_algs.c:221-240): what the writers have to undo
    info->negate_primal = (S.c_dir > 0 && optdir == -1) || (S.c_dir < 0 && optdir == 1);
    info->negate_dual_last = (optdir == -1);
}
static void phase2_failure(bslv_vlp_info *info, int vst, int bounded, long lps)
{
    if (!info) return;
    info->lps = lps;                                                                 // bslv_main.c:311-329
    snprintf(info->message, sizeof info->message, "%s", vst == 1 ? "VLP is infeasible" : bounded ? "VLP is not bounded, re-run without option -b" : "LP in phase 2 is not bounded, probably by inaccuracy in phase 1");
}

// The whole primal algorithm, bslv_main.c:236-345 without the file I/O.  P as written in the file (not negated).
// On success with *vlp_status == 4 (VLP_OPTIMAL) *engine_out holds the finished phase-2 engine (its polyhedron is the
// result; the caller destroys it) and info (may be NULL) the ordering-cone data.
int bslv_vlp_solve_primal(int m, int n, int q, const double *A, const double *P,
                          const char *rtype, const double *rlb, const double *rub,
                          const char *ctype, const double *clb, const double *cub,
                          int optdir, int cone_kind, const double *gen, int n_gen, const double *c_in,
                          int bounded, int flags, double eps_phase0, double eps_phase1, double eps_benson_phase1, double eps_benson_phase2,
                          int batch, bslv_benson **engine_out, int *vlp_status, bslv_vlp_info *info)
{
    if (!engine_out || !vlp_status || m < 1 || n < 1 || q < 2 || !A || !P || !rtype || !ctype || batch < 1 || (cone_kind != 0 && (!gen || n_gen < 1))) {
        set_error("bslv_vlp_solve_primal: bad argument");
        return BSLV_E_ARG;
    }
    *engine_out = nullptr;
    Sol S;
    std::vector<double> Pn;
    long lps = 0, steps = 0;
    int rc;
    if ((rc = front(m, n, q, A, P, rtype, rlb, rub, ctype, clb, cub, optdir, cone_kind, gen, n_gen, c_in, bounded, eps_phase0, eps_phase1,
                    eps_benson_phase1, batch, (flags & BSLV_VLP_PHASE1_DUAL) != 0, S, Pn, &lps, &steps, vlp_status, info))) return rc;
    if (*vlp_status) return 0;
    bslv_benson *h = nullptr;
    if ((rc = bslv_benson_create_ex(&h, m, n, q, A, Pn.data(), rtype, rlb, rub, ctype, clb, cub, S.R.data(), S.r, S.c.data(), S.eta.data(), 0, (flags & BSLV_VLP_PREIMAGES) ? BSLV_BENSON_PREIMAGES : 0,
                                    eps_benson_phase2, std::max(4 * batch + 64, 64)))) return rc;
    int vst = 0;
    if ((rc = bslv_benson_start(h, &vst))) { bslv_benson_destroy(h); return rc; }
    if (vst) {
        bslv_benson_destroy(h);
        *vlp_status = vst;                                                           // 1 VLP_INFEASIBLE, 2 VLP_UNBOUNDED
        phase2_failure(info, vst, bounded, lps);
        return 0;
    }
    if ((rc = run_engine(h, batch, &lps, &steps))) { bslv_benson_destroy(h); return rc; }
    if (flags & BSLV_VLP_PREIMAGES) {
        Problem pb{m, n, q, A, Pn.data(), rtype, rlb, rub, ctype, clb, cub};
        if ((rc = direction_preimages(pb, S, h, &lps))) { bslv_benson_destroy(h); return rc; }
    }
    *engine_out = h;
    *vlp_status = 4;
    fill_info(info, S, optdir, lps, steps);
    return 0;
}

// The same with the DUAL algorithm in phase 2 ("-a dual": phase2_dual, bslv_algs.c:1381-1592; phases 0 and 1 stay primal, the
// reference's default).  With status 4 *lower_image_out is a polyhedron whose PRIMAL side is the lower image and whose dual
// side is the upper image (write it with bslv_sol_write3(..., swap = 1, ...)); destroy it with bslv_poly_destroy.
int bslv_vlp_solve_dual2(int m, int n, int q, const double *A, const double *P,
                         const char *rtype, const double *rlb, const double *rub,
                         const char *ctype, const double *clb, const double *cub,
                         int optdir, int cone_kind, const double *gen, int n_gen, const double *c_in,
                         int bounded, int flags, double eps_phase0, double eps_phase1, double eps_benson_phase1, double eps_benson_phase2,
                         int batch, bslv_poly **lower_image_out, int *vlp_status, bslv_vlp_info *info)
{
    if (!lower_image_out || !vlp_status || m < 1 || n < 1 || q < 2 || !A || !P || !rtype || !ctype || batch < 1 || (cone_kind != 0 && (!gen || n_gen < 1))) {
        set_error("bslv_vlp_solve_dual2: bad argument");
        return BSLV_E_ARG;
    }
    *lower_image_out = nullptr;
    Sol S;
    std::vector<double> Pn;
    long lps = 0, steps = 0;
    int rc;
    if ((rc = front(m, n, q, A, P, rtype, rlb, rub, ctype, clb, cub, optdir, cone_kind, gen, n_gen, c_in, bounded, eps_phase0, eps_phase1,
                    eps_benson_phase1, batch, (flags & BSLV_VLP_PHASE1_DUAL) != 0, S, Pn, &lps, &steps, vlp_status, info))) return rc;
    if (*vlp_status) return 0;
    Problem pb{m, n, q, A, Pn.data(), rtype, rlb, rub, ctype, clb, cub};
    int vst = 0;
    bslv_poly *poly = nullptr;
    DualPreimg *store = (flags & BSLV_VLP_PREIMAGES) ? new DualPreimg() : nullptr;
    if (store) { store->m = m; store->n = n; store->q = q; }
    if ((rc = phase2_dual(pb, S, eps_benson_phase2, batch, &poly, &vst, &lps, &steps, store))) { delete store; return rc; }
    if (vst) { delete store; *vlp_status = vst; phase2_failure(info, vst, bounded, lps); return 0; }
    if (store) {
        // pre-images of the extreme DIRECTIONS of the upper image (dual slots that are directions; bslv_algs.c:1508-1535): x of
        // the homogeneous P2 with the bound Z'd on the cone rows and the eta row switched off, one LP per direction.  (The
        // reference indexes Z with the stride of R there, sol->Z[k*sol->r+j] at :1533 against k*sol->p+j at :1106: right only
        // when p = r; the stride used here is p.)
        const int p = S.p, nd = bslv_poly_ndual(poly);
        std::vector<unsigned char> du(nd), di(nd);
        std::vector<double> dc((size_t)nd * q);
        if ((rc = bslv_poly_get_dual(poly, du.data(), di.data(), dc.data()))) { delete store; bslv_poly_destroy(poly); return rc; }
        bslv_benson *hh = nullptr;
        if ((rc = bslv_benson_create_ex(&hh, pb.m, pb.n, q, pb.A, pb.P, pb.rtype, pb.rlb, pb.rub, pb.ctype, pb.clb, pb.cub,
                                        S.Z.data(), p, S.c.data(), nullptr, 1, 0, 1e-7, 4))) { delete store; bslv_poly_destroy(poly); return rc; }
        bslv_lpq *lp = bslv_benson_lp(hh);
        int M2, N2, folded;
        bslv_benson_lp_dims(hh, &M2, &N2, &folded);
        rc = bslv_lpq_reset_slot(lp, 0);
        const int zero = 0;
        std::vector<double> ub(p), x(n);
        for (int f = 0; f < nd && !rc; f++) {
            if (!du[f] || !di[f]) continue;
            for (int j = 0; j < p; j++) { double sdot = 0; for (int k = 0; k < q; k++) sdot += S.Z[(size_t)k * p + j] * dc[(size_t)f * q + k]; ub[j] = sdot; }
            int st2;
            if ((rc = solve0(lp, p, ub.data(), &st2))) break;
            if (st2 != BSLV_LP_OPTIMAL) { set_error("pre-image of a direction: LP status %d (the reference asserts optimality, bslv_algs.c:1537)", st2); rc = BSLV_E_STATE; break; }
            ++lps;
            if ((rc = bslv_lpq_get_primal(lp, 1, &zero, M2, n, x.data()))) break;
            store->x_by_facet[f] = x;
        }
        bslv_benson_destroy(hh);
        if (rc) { delete store; bslv_poly_destroy(poly); return rc; }
        std::lock_guard<std::mutex> lk(g_dpre_mu);
        g_dpre[poly] = store;
    }
    *lower_image_out = poly;
    *vlp_status = 4;
    fill_info(info, S, optdir, lps, steps);
    return 0;
}

// pre-images kept by bslv_vlp_solve_dual2 with BSLV_VLP_PREIMAGES: 0 + data, or 1 when nothing is stored
int bslv_dual_preimage_x(const bslv_poly *lower_image, int facet, double *x)
{
    std::lock_guard<std::mutex> lk(g_dpre_mu);
    auto it = g_dpre.find(lower_image);
    if (it == g_dpre.end() || !x) return 1;
    auto jt = it->second->x_by_facet.find(facet);
    if (jt == it->second->x_by_facet.end()) return 1;
    memcpy(x, jt->second.data(), jt->second.size() * sizeof(double));
    return 0;
}
int bslv_dual_preimage_uw(const bslv_poly *lower_image, int element, double *uw)
{
    std::lock_guard<std::mutex> lk(g_dpre_mu);
    auto it = g_dpre.find(lower_image);
    if (it == g_dpre.end() || !uw) return 1;
    auto jt = it->second->uw_by_element.find(element);
    if (jt == it->second->uw_by_element.end()) return 1;
    memcpy(uw, jt->second.data(), jt->second.size() * sizeof(double));
    return 0;
}
void bslv_dual_preimages_free(const bslv_poly *lower_image)
{
    std::lock_guard<std::mutex> lk(g_dpre_mu);
    auto it = g_dpre.find(lower_image);
    if (it != g_dpre.end()) { delete it->second; g_dpre.erase(it); }
}

void bslv_vlp_info_free(bslv_vlp_info *info)
{
    if (!info) return;
    free(info->c); free(info->eta); free(info->R); free(info->H); free(info->Y); free(info->Z);
    info->c = info->eta = info->R = info->H = info->Y = info->Z = nullptr;
}

}  // extern "C"
