// benson_driver.hip -- host side of the re-shaped hot loop: phase2_primal (bslv_algs.c:958-1082)
// turned from "one vertex -> one LP -> one cut" into
//     collect unprocessed vertices -> batched LPs (lp_engine) -> gather cut records -> apply cuts
// (poly_engine).  Pure host code (no kernels here); the two engines own the device work.
//
// Multi-GPU: every rank holds a replica of the polyhedron and the same vertex
// list; collect() deals the batch to ranks (affinity to the rank that holds the parent tableau,
// else least loaded), each rank solves its shard, the fixed-size records are all-gathered by the
// caller (RCCL, one collective per outer iteration) and every rank applies ALL records in
// ascending source-slot order, so the replicas stay identical without further traffic.
#include "common.h"
#include <vector>
#include <algorithm>
#include <chrono>
#include <deque>
#include <unordered_map>
#include <mutex>
#include <unordered_set>
#include <string>

extern "C" {
int bslv_poly_unprocessed2(bslv_poly *h, int max_out, int from_end, int *idx, double *val, int *ideal, int *parent, int *count);
int bslv_poly_children_hist(bslv_poly *h, int first_facet, int n, int *counts, int *total, int *older);
int bslv_poly_children_of(bslv_poly *h, int nfacets, const int *facets, int max_out, int *idx, double *val, int *ideal, int *parent, int *n_out);
}

using namespace bslv;
using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

struct bslv_benson {
    int m = 0, n = 0, q = 0, r = 0, M = 0, N = 0;     // m: rows of A left after the presolve
    int rows_folded = 0;                              // singleton rows of A turned into column bounds
    bool hom = false;                                 // homogeneous problem (phases 0 and 1)
    long tot_retries = 0;                             // LPs solved again from the root tableau / the standard basis
    long tot_root_starts = 0, tot_nearest_starts = 0; // LPs whose parent's tableau was not resident here: started from the root tableau / from the nearest resident one
    bool nearest_start = true;
    // option -s (opt->solution == PRE_IMG_ON): x of every confirmed vertex, (u, w) of every cut (bslv_algs.c:1064-1079)
    bool want_primg = false;
    int m_orig = 0;                                   // rows of A as given (pre-images need every row: no presolve)
    std::unordered_map<int, std::vector<double>> primg_p, primg_d;      // by primal element / by facet (dual slot)
    std::vector<std::pair<std::vector<double>, std::vector<double>>> start_primg;   // PART 1: (y*, (u, w)) until the slots exist
    std::vector<double> eta;                          // q: coefficients of the last row (eta.y <= 1 when hom)
    double eps = 1e-7;
    std::vector<double> R, c;           // q x r (generators as columns), q
    bslv_lpq *lp = nullptr;
    bslv_poly *poly = nullptr;
    int pool_slots = 0;
    int batch_cap = 1 << 30;            // LPs per outer iteration and rank the pool of tableaux can serve
    // tableau slots: 0 = root (kept).  facet -> slot of the LP that produced the cut
    std::vector<int> free_slots;
    std::deque<std::pair<int, int>> parents;          // (facet, slot) in creation order
    std::unordered_map<int, int> facet_slot;
    std::vector<int> facet_owner;                     // by facet id: rank that solved its LP (-1 unknown)
    // policy 4: K depth-first fronts.  A vertex belongs to the front of the cut that created it; a cut to the front of the
    // vertex whose LP returned it.  A front that runs dry is re-seeded with the older half of the richest front's facets.
    std::vector<int> facet_front;                     // by facet id (missing / -1: front 0)
    int nfronts = 8;
    long front_splits = 0;
    // policy 6: the batch is made of whole FAMILIES -- all unprocessed children of a cut -- of parents chosen among the cuts of
    // the last fam_batches outer iterations: fam_mode 0 newest cuts first, 1 pseudo-random, 2 far apart (farthest-point sampling on
    // the cuts' normals), so that the families of one batch act on different neighbourhoods of the polyhedron
    int fam_mode = 3, fam_batches = 1, fam_cap = 0;
    // cuts that bslv_poly_add_cuts handed back (thin rounds at the end of a chunk, bslv_poly_set_defer): kept with what their
    // bookkeeping needs and handed in again IN FRONT of the next batch's cuts; collect applies them all before it reports
    // "nothing left".  Identical on every rank (the cut phase is replicated); `slot` is the tableau of the LP on its owner.
    struct Deferred { std::vector<double> ys; double z; int owner, slot, front; };
    std::vector<Deferred> deferred;
    int defer_thr = getenv("BSLV_DEFER") ? std::max(0, atoi(getenv("BSLV_DEFER"))) : 0, defer_max = 512;
    long tot_deferred = 0, defer_flushes = 0;
    long fam_fallbacks = 0;                           // batches that one family would have filled: taken newest first
    long dir_only_windows = 0;                        // policy 6: chosen families that held nothing but extreme directions (chosen again at once)
    double facet_z0 = getenv("BSLV_FACET_Z0") ? atof(getenv("BSLV_FACET_Z0")) : (double)INFINITY;
    std::deque<int> batch_f0;                         // first dual slot of each of the last outer iterations
    std::vector<double> facet_normal;                 // q per dual slot (zeros where unknown)
    std::vector<double> facet_z;                      // per dual slot: optimal value z of the LP that returned the cut (how deep it cuts; 0 where unknown)
    long collect_seq = 0;
    bool started = false;
    // batch contexts (after collect).  Two of them so that the LPs of batch k can run (on the LP engine's
    // stream, from a second host thread) while the cuts of batch k-1 are applied (polyhedron engine)
    struct BatchCtx {
        std::vector<int> b_idx, b_parent, b_owner, b_front;    // whole batch (all ranks)
        std::vector<double> b_val;
        std::vector<int> l_pos, l_slot;               // local shard: position in batch, dst slot
    } ctx[2];
    std::mutex slot_mu;                               // tableau-slot bookkeeping is shared by solve_local and apply
    int mark_at_collect = 0;                          // pipelined mode: batch members get the sltn mark when collected
    int rank = 0, world = 1;
    int unprocessed_left = 0;
    int policy = 6;                                   // 6 (default since round 3): whole families, the children of the shallowest cuts of the last batch first; 1: newest vertices first (rounds 1-2), 2: spread over the whole queue, 3: newest first, few siblings, 4: fronts, 5: children of the newest cuts
    int sib_cap = 1, sib_window = 8;                  // policy 3: children of one cut per batch; depth of the window in batches
    std::vector<double> slot_src;                     // per slot: vertex its LP was solved for (pool_slots x q)
    std::vector<char> slot_valid;
    std::vector<int> slot_gen;                        // generations of warm starts between the root tableau and this slot
    std::vector<int> last_src, last_piv, last_gen;    // per LP of the last solve_local (tuning: bslv_benson_last_local)
    // totals
    long tot_lps = 0, tot_cuts = 0, tot_pivots = 0;
};

static int rec_len(const bslv_benson *h) { return h->q + 5; }

extern "C" {

int bslv_benson_record_len(const bslv_benson *h) { return h ? rec_len(h) : 0; }
bslv_poly *bslv_benson_poly(bslv_benson *h) { return h ? h->poly : nullptr; }
bslv_lpq *bslv_benson_lp(bslv_benson *h) { return h ? h->lp : nullptr; }

void bslv_benson_destroy(bslv_benson *h)
{
    if (!h) return;
    if (h->lp) bslv_lpq_destroy(h->lp);
    if (h->poly) bslv_poly_destroy(h->poly);
    delete h;
}

static void bounds_of(char t, double lb, double ub, double *lo, double *up)
{
    // 'f','l','u','d','s' -> [lo,up]  (bslv_lp.c:34-43; 's' fixes at lb)
    *lo = (t == 'l' || t == 'd' || t == 's') ? lb : -INFINITY;
    *up = (t == 'u' || t == 'd') ? ub : (t == 's' ? lb : INFINITY);
}

// Problem in the reference's normal form "min, c_q > 0" (sol_init has already flipped P for
// max / c_q < 0, bslv_vlp.c:845-861).  R: q x r, generators as columns (bslv_algs.c:599).
int bslv_benson_create(bslv_benson **out, int m, int n, int q, const double *A, const double *P,
                       const char *rtype, const double *rlb, const double *rub,
                       const char *ctype, const double *clb, const double *cub,
                       const double *R, int r, const double *c, double eps, int pool_slots)
{
    return bslv_benson_create_ex(out, m, n, q, A, P, rtype, rlb, rub, ctype, clb, cub, R, r, c, nullptr, 0, 0, eps, pool_slots);
}

// hom != 0: the HOMOGENEOUS problem of phases 0 and 1 (init_P2(..., HOMOGENEOUS), bslv_algs.c:574-664): every bound of the
// VLP becomes 0 and double-bounded turns into fixed (lp_set_rows_hom / lp_set_cols_hom, bslv_lp.c:34-43,118-134), the columns
// of R are the generators Z of the dual ordering cone, and the last row reads eta.y <= 1 (eta may be NULL = 0: phase 0).
int bslv_benson_create_ex(bslv_benson **out, int m, int n, int q, const double *A, const double *P,
                          const char *rtype, const double *rlb, const double *rub,
                          const char *ctype, const double *clb, const double *cub,
                          const double *R, int r, const double *c, const double *eta, int hom, int flags, double eps, int pool_slots)
{
    if (!out || m < 1 || n < 1 || q < 2 || r < 1 || !A || !P || !rtype || !ctype || !R || !c || pool_slots < 4) {
        set_error("bslv_benson_create: bad argument");
        return BSLV_E_ARG;
    }
    bslv_benson *h = new bslv_benson();
    // (Singleton rows of A -- the hypercube rows of S-degenerate, ex/example10.m:21-24 -- are folded into column bounds by the LP
    // layer itself since round 3, bslv_lpq_create: every index below is an index of the model as the reference builds it.)
    auto vlp_bounds = [hom](char t, double lb, double ub, double *lo, double *up) {
        if (hom) bounds_of(t == 'd' ? 's' : t, 0.0, 0.0, lo, up); else bounds_of(t, lb, ub, lo, up);
    };
    std::vector<double> clo(n), cup(n);
    for (int j = 0; j < n; j++) vlp_bounds(ctype[j], clb ? clb[j] : 0, cub ? cub[j] : 0, &clo[j], &cup[j]);
    std::vector<int> keep(m);
    for (int i = 0; i < m; i++) keep[i] = i;
    int singletons = 0;                                   // (only to size the pool: rows with one non-zero leave the tableau)
    if (!getenv("BSLV_NO_PRESOLVE"))
        for (int i = 0; i < m; i++) {
            int nz = 0;
            for (int j = 0; j < n && nz < 2; j++) if (A[(size_t)i * n + j] != 0.0) nz++;
            singletons += nz == 1;
        }
    singletons = std::min(singletons, m - 1);
    h->want_primg = (flags & BSLV_BENSON_PREIMAGES) != 0;
    h->m_orig = m;
    h->m = m; h->n = n; h->q = q; h->r = r; h->eps = eps;
    h->R.assign(R, R + (size_t)q * r);
    h->c.assign(c, c + q);
    // init_P2 (bslv_algs.c:574-664), inhomogeneous: rows m (A x), q (-P x + y = 0), r (R_j.y - z <= ub_j),
    // 1 free eta row; cols n (x), q (y free), 1 (z free, cost 1)
    const int M = m + q + r + 1, N = n + q + 1;
    h->M = M; h->N = N;
    std::vector<double> L((size_t)M * N, 0.0), lo(M + N), up(M + N), cost(N + 1, 0.0);
    for (int i = 0; i < m; i++) memcpy(&L[(size_t)i * N], A + (size_t)keep[i] * n, n * sizeof(double));
    for (int k = 0; k < q; k++) {
        for (int j = 0; j < n; j++) L[(size_t)(m + k) * N + j] = -P[(size_t)k * n + j];
        L[(size_t)(m + k) * N + n + k] = 1.0;
    }
    for (int i = 0; i < r; i++) {
        for (int k = 0; k < q; k++) L[(size_t)(m + q + i) * N + n + k] = R[(size_t)k * r + i];
        L[(size_t)(m + q + i) * N + n + q] = -1.0;
    }
    for (int i = 0; i < m; i++) vlp_bounds(rtype[keep[i]], rlb ? rlb[keep[i]] : 0, rub ? rub[keep[i]] : 0, &lo[i], &up[i]);
    for (int k = 0; k < q; k++) { lo[m + k] = 0; up[m + k] = 0; }
    for (int i = 0; i < r; i++) { lo[m + q + i] = -INFINITY; up[m + q + i] = 0; }
    lo[m + q + r] = -INFINITY; up[m + q + r] = hom ? 1.0 : INFINITY;            // eta.y <= 1 | free (bslv_algs.c:636-649)
    if (eta) for (int k = 0; k < q; k++) L[(size_t)(m + q + r) * N + n + k] = eta[k];
    h->hom = hom != 0;
    h->eta.assign(q, 0.0);
    if (eta) h->eta.assign(eta, eta + q);
    for (int j = 0; j < n; j++) { lo[M + j] = clo[j]; up[M + j] = cup[j]; }
    for (int j = n; j < N; j++) { lo[M + j] = -INFINITY; up[M + j] = INFINITY; }
    cost[N] = 1.0;
    {
        // The pool has to fit: a slot is the whole (M + 1) x N tableau (ex09 of the reference's suite: 4612 x 36 942 doubles = 1.36 GB).
        // The number of slots is cut to 70 % of the free device memory, and the LPs per outer iteration follow (a quarter of the
        // pool: a child needs its parent's slot and its own) -- 288 GB hold 140 such tableaux.
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
            const size_t ld = ((size_t)N + 2 + 1) & ~(size_t)1;
            const size_t Mt = (size_t)(M - singletons);         // rows of the tableau
            const size_t slot = (Mt + 2) * ld * sizeof(double) + (Mt * 2 + (size_t)N * 4 + ld) * 8;
            const size_t fit = (size_t)(0.7 * (double)fr) / std::max<size_t>(slot, 1);
            if ((size_t)pool_slots > fit) {
                if (fit < 8) { set_error("bslv_benson_create: a tableau of %zu MB does not fit 8 times into the %zu MB of free device memory", slot >> 20, fr >> 20); bslv_benson_destroy(h); return BSLV_E_CAPACITY; }
                pool_slots = (int)fit;
                h->batch_cap = std::max(1, (pool_slots - 4) / 4);       // (only where memory cut the pool: callers that sized it themselves keep their batch)
            }
        } else (void)hipGetLastError();                                  // (no figure, no cut; the error must not surface at the next launch check)
    }
    int rc = bslv_lpq_create(&h->lp, M, N, L.data(), lo.data(), up.data(), cost.data(), m + q, r, pool_slots);
    if (rc) { bslv_benson_destroy(h); return rc; }
    h->rows_folded = bslv_lpq_rows_folded(h->lp);
    rc = bslv_poly_create(&h->poly, q, 1 /* lowerV2upperH */, c);
    if (rc) { bslv_benson_destroy(h); return rc; }
    h->pool_slots = pool_slots;
    if (const char *e = getenv("BSLV_RESERVE")) {        // opt-in: capacity of the polyhedron ahead of need, "elements[:edges[:pool words]]" (bslv_poly_reserve)
        long a = 0, b = 0, c3 = 0;
        if (sscanf(e, "%ld:%ld:%ld", &a, &b, &c3) >= 1 && a > 0) {
            if (b <= 0) b = 4 * a;
            if (c3 <= 0) c3 = 16 * a;
            if ((rc = bslv_poly_reserve(h->poly, a, b, c3))) { bslv_benson_destroy(h); return rc; }
        }
    }
    if (const char *e = getenv("BSLV_POLICY")) {         // tuning: "policy[:a[:b]]" -- 3:cap:window, 4:fronts:cap, 6:mode:batches
        int pol = 0, a = -1, b = -1, c4 = -1;
        if (sscanf(e, "%d:%d:%d:%d", &pol, &a, &b, &c4) >= 1 && pol >= 1 && pol <= 6) {
            h->policy = pol;
            if (pol == 3 && a > 0) { h->sib_cap = a; if (b > 0) h->sib_window = b; }
            if (pol == 4 && a > 0) { h->nfronts = a; if (b > 0) h->sib_cap = b; }
            if (pol == 6 && a >= 0 && a <= 5) { h->fam_mode = a; if (b > 0) h->fam_batches = b; if (c4 >= 0) h->fam_cap = c4; }
        }
    }
    for (int s = pool_slots - 1; s >= 1; s--) h->free_slots.push_back(s);
    h->slot_src.assign((size_t)pool_slots * q, 0.0);
    h->slot_valid.assign(pool_slots, 0);
    h->slot_gen.assign(pool_slots, 0);
    *out = h;
    return 0;
}

// PART 1 of phase2_primal (bslv_algs.c:976-1018): r weighted-sum LPs from a cold start, then the
// initial outer approximation.  *vlp_status: 0 ok, 1 infeasible, 2 unbounded.
int bslv_benson_start(bslv_benson *h, int *vlp_status)
{
    if (!h || !vlp_status) return BSLV_E_ARG;
    const int q = h->q, r = h->r;
    int rc;
    *vlp_status = 0;
    if ((rc = bslv_lpq_reset_slot(h->lp, 0))) return rc;
    std::vector<double> vlo(r, -INFINITY), vup(r), val(q);
    const int zero = 0;
    for (int j = 0; j < r; j++) {
        for (int i = 0; i < r; i++) vup[i] = (i == j) ? 0.0 : INFINITY;     // row j 'u' (ub 0), others 'f'
        int st, it;
        // each weighted-sum LP starts cold: freeing row j-1 leaves the previous basis dual infeasible,
        // which the dual-only engine reports as UNDEFINED (GLPK would switch to its primal phase)
        if (j > 0 && (rc = bslv_lpq_reset_slot(h->lp, 0))) return rc;
        if ((rc = bslv_lpq_solve_batch(h->lp, 1, &zero, &zero, vlo.data(), vup.data(), &st, &it))) return rc;
        h->tot_pivots += it;
        if (st != BSLV_LP_OPTIMAL) { *vlp_status = (st == BSLV_LP_INFEASIBLE) ? 1 : 2; return 0; }
        h->tot_lps++;
        double obj;
        if ((rc = bslv_lpq_get_obj(h->lp, 1, &zero, &obj))) return rc;
        for (int k = 0; k < q; k++) val[k] = h->R[(size_t)k * r + j];
        val[q - 1] = obj;
        if (h->want_primg) {                                                         // (u, w) of the weighted-sum LP (:1001-1006)
            std::vector<double> uw(h->m + q);
            if ((rc = bslv_lpq_get_dual(h->lp, 1, &zero, 0, h->m + q, uw.data()))) return rc;
            h->start_primg.emplace_back(val, uw);
        }
        int prc;
        if ((rc = bslv_poly_add(h->poly, val.data(), 0, &prc))) return rc;
    }
    int irc;
    if ((rc = bslv_poly_init(h->poly, &irc))) return rc;
    if (irc) { set_error("initial outer approximation failed (rank-deficient start, bslv_poly.c:174)"); return BSLV_E_STATE; }
    if (h->want_primg) {
        // poly__intl_apprx re-adds the queued halfspaces as new dual slots (bslv_poly.c:190-197): find them by their coordinates
        const int nd = bslv_poly_ndual(h->poly);
        std::vector<unsigned char> du(nd), di(nd);
        std::vector<double> dc((size_t)nd * q);
        if ((rc = bslv_poly_get_dual(h->poly, du.data(), di.data(), dc.data()))) return rc;
        for (int f = 0; f < nd; f++) {
            if (!du[f] || di[f]) continue;
            for (auto &sp : h->start_primg) {
                double dd = 0;
                for (int k = 0; k < q; k++) dd = std::max(dd, std::fabs(sp.first[k] - dc[(size_t)f * q + k]));
                if (dd <= 1e-12 * (1.0 + std::fabs(sp.first[q - 1]))) { h->primg_d[f] = sp.second; break; }
            }
        }
    }
    h->facet_owner.assign(bslv_poly_ndual(h->poly), -1);
    h->started = true;
    return 0;
}

// Select the next batch (the newest max_batch unprocessed elements; directions are only marked,
// bslv_algs.c:1036-1040) and deal it to ranks.  n_local = LPs this rank will solve.
int bslv_benson_collect_ctx(bslv_benson *h, int ctx, int max_batch, int rank, int world, int *n_local, int *n_total);
static int deal_batch(bslv_benson *h, bslv_benson::BatchCtx &B, int rank, int world, int *n_local, int *n_total);
int bslv_benson_collect(bslv_benson *h, int max_batch, int rank, int world, int *n_local, int *n_total)
{
    return bslv_benson_collect_ctx(h, 0, max_batch, rank, world, n_local, n_total);
}
static int collect_impl(bslv_benson *h, int ctx, int max_batch, int rank, int world, int *n_local, int *n_total);
int bslv_benson_apply_ctx(bslv_benson *h, int ctx, int nrec, const double *records, long *stats);
int bslv_benson_collect_ctx(bslv_benson *h, int ctx, int max_batch, int rank, int world, int *n_local, int *n_total)
{
    int rc = collect_impl(h, ctx, max_batch, rank, world, n_local, n_total);
    // "nothing left" is only true once the cuts that were handed back (bslv_benson::deferred) are applied: they may create vertices
    for (int guard = 0; !rc && *n_total == 0 && !h->deferred.empty() && guard < 64; guard++) {
        long st[5];
        h->defer_flushes++;
        if ((rc = bslv_benson_apply_ctx(h, ctx, 0, nullptr, st))) return rc;
        rc = collect_impl(h, ctx, max_batch, rank, world, n_local, n_total);
    }
    return rc;
}
static int collect_impl(bslv_benson *h, int ctx, int max_batch, int rank, int world, int *n_local, int *n_total)
{
    if (!h || !h->started || max_batch < 1 || world < 1 || rank < 0 || rank >= world || !n_local || !n_total || ctx < 0 || ctx > 1) {
        set_error("bslv_benson_collect: bad argument / not started");
        return BSLV_E_ARG;
    }
    const int q = h->q;
    bslv_benson::BatchCtx &B = h->ctx[ctx];
    h->rank = rank; h->world = world;
    max_batch = (int)std::min<long long>(max_batch, (long long)h->batch_cap * world);          // (what the pool of tableaux can serve, see bslv_benson_create_ex)
    int rc, cnt = 0;
    if (h->policy == 6) for (int again = 0;; again++) {
        // (again: the families chosen held nothing but extreme directions -- marked as processed below -- while vertices are still
        // waiting elsewhere: choose again, as the generic path does; a caller that stops on an empty batch would otherwise stop early)
        B.b_idx.clear(); B.b_val.clear(); B.b_parent.clear(); B.b_front.clear();
        const int nf = bslv_poly_ndual(h->poly);
        h->collect_seq++;
        int total = 0;
        // window: the cuts of the last fam_batches outer iterations; widened while it cannot fill the batch
        int f_lo = h->batch_f0.empty() ? 0 : h->batch_f0[std::max(0, (int)h->batch_f0.size() - h->fam_batches)];
        std::vector<int> counts, chosen;
        for (;;) {
            counts.assign((size_t)std::max(0, nf - f_lo), 0);
            if ((rc = bslv_poly_children_hist(h->poly, f_lo, (int)counts.size(), counts.data(), &total, nullptr))) return rc;
            long in_window = 0;
            for (int c : counts) in_window += h->fam_cap > 0 ? std::min(c, h->fam_cap) : c;
            if (in_window >= max_batch || f_lo == 0) break;
            f_lo = std::max(0, f_lo - std::max(64, 2 * (nf - f_lo)));
        }
        std::vector<int> cand;
        for (int k = 0; k < (int)counts.size(); k++) if (counts[k] > 0) cand.push_back(f_lo + k);
        long have = 0, full = 0;
        if (h->fam_mode == 2 && (int)h->facet_normal.size() >= nf * q) {
            // far apart: the first parent is the newest cut, every further one the candidate farthest (in the angle of the normals) from
            // all parents taken so far
            std::vector<double> best(cand.size(), 1e300);
            std::vector<char> used(cand.size(), 0);
            int cur = cand.empty() ? -1 : (int)cand.size() - 1;
            while (cur >= 0 && have < max_batch) {
                used[cur] = 1; chosen.push_back(cand[cur]); have += counts[cand[cur] - f_lo]; full += counts[cand[cur] - f_lo];
                const double *nc = &h->facet_normal[(size_t)cand[cur] * q];
                int nxt = -1; double far = -1;
                for (size_t t = 0; t < cand.size(); t++) {
                    if (used[t]) continue;
                    const double *nt = &h->facet_normal[(size_t)cand[t] * q];
                    double dd = 0;
                    for (int k = 0; k < q; k++) dd += (nt[k] - nc[k]) * (nt[k] - nc[k]);
                    if (dd < best[t]) best[t] = dd;
                    if (best[t] > far) { far = best[t]; nxt = (int)t; }
                }
                cur = nxt;
            }
        } else {
            std::vector<int> ord(cand.size());
            for (size_t t = 0; t < ord.size(); t++) ord[t] = (int)t;
            if (h->fam_mode >= 3) {
                // by the depth z of the parent cut: 3 shallowest first (the finest scale: children next to their parent, few pivots), 4 deepest
                // first, 5 pseudo-random among the shallower half
                auto zz = [&](int f) { return f < (int)h->facet_z.size() ? h->facet_z[f] : h->facet_z0; };
                auto hsh = [&](int f) { unsigned long long z = (unsigned long long)f * 0x9E3779B97F4A7C15ull + (unsigned long long)h->collect_seq * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
                std::sort(ord.begin(), ord.end(), [&](int a, int b) { const double za = zz(cand[a]), zb = zz(cand[b]); if (za != zb) return h->fam_mode == 4 ? za > zb : za < zb; return cand[a] > cand[b]; });
                if (h->fam_mode == 5) {
                    const size_t half = std::max<size_t>(1, ord.size() / 2);
                    std::sort(ord.begin(), ord.begin() + half, [&](int a, int b) { return hsh(cand[a]) < hsh(cand[b]); });
                }
            } else if (h->fam_mode == 1) {
                auto hsh = [&](int f) { unsigned long long z = (unsigned long long)f * 0x9E3779B97F4A7C15ull + (unsigned long long)h->collect_seq * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
                std::sort(ord.begin(), ord.end(), [&](int a, int b) { return hsh(cand[a]) < hsh(cand[b]); });
            } else std::reverse(ord.begin(), ord.end());
            const int cap = h->fam_cap > 0 ? h->fam_cap : INT_MAX;
            for (int t : ord) { if (have >= max_batch) break; chosen.push_back(cand[t]); have += std::min(cap, counts[cand[t] - f_lo]); full += counts[cand[t] - f_lo]; }
        }
        // A batch that ONE family fills (the pool serves few LPs at a time -- S-degenerate: 64 -- or the family is huge) has nothing of
        // what the rule is for: it would take the first slots of the shallowest cut's children, step after step.  Such a batch is
        // taken newest first, as rounds 1-2 did (S-degenerate q = 10, three steps: 17 -> 134 LPs/s; BSLV_FAM_FALLBACK=0: never)
        static const bool fam_fallback = !(getenv("BSLV_FAM_FALLBACK") && atoi(getenv("BSLV_FAM_FALLBACK")) == 0);
        const bool one_family = fam_fallback && !chosen.empty() && counts[chosen[0] - f_lo] >= max_batch;
        if (one_family) { h->fam_fallbacks++; break; }       // (the generic path below)
        {
        // (with a cap on the children of one cut the device hands over the whole families and the host thins them out)
        const int fetch = h->fam_cap > 0 ? (int)std::min<long long>(std::max<long long>(full, 1), 16LL * max_batch) : max_batch;
        std::vector<int> idx(fetch), ideal(fetch), parent(fetch);
        std::vector<double> val((size_t)fetch * q);
        int n = 0;
        if (!chosen.empty() && (rc = bslv_poly_children_of(h->poly, (int)chosen.size(), chosen.data(), fetch, idx.data(), val.data(), ideal.data(), parent.data(), &n))) return rc;
        std::vector<char> keep;
        if (h->fam_cap > 0) {
            // at most fam_cap children of one cut, evenly spaced in the order of their slots (neighbours on the parent's facet have
            // neighbouring slots); the others stay unprocessed: by the time their turn comes the cuts of their siblings have removed many
            keep.assign(n, 0);
            std::unordered_map<int, int> seen;
            int kept = 0;
            for (int k = 0; k < n; k++) {
                if (ideal[k]) { keep[k] = 1; continue; }
                const int f = parent[k], cnt_f = (f >= f_lo && f - f_lo < (int)counts.size()) ? counts[f - f_lo] : 1, j = seen[f]++;
                const bool take = cnt_f <= h->fam_cap || (long long)(j + 1) * h->fam_cap / cnt_f != (long long)j * h->fam_cap / cnt_f;
                if (take && kept < max_batch) { keep[k] = 1; kept++; }
            }
        }
        int nkept = 0;
        std::vector<int> dirs;
        for (int k = 0; k < n; k++) {
            if (!keep.empty() && !keep[k]) continue;
            nkept++;
            if (ideal[k]) { dirs.push_back(idx[k]); continue; }
            B.b_idx.push_back(idx[k]);
            B.b_parent.push_back(parent[k]);
            B.b_val.insert(B.b_val.end(), &val[(size_t)k * q], &val[(size_t)(k + 1) * q]);
        }
        if (!dirs.empty() && (rc = bslv_poly_mark(h->poly, (int)dirs.size(), dirs.data()))) return rc;
        h->unprocessed_left = total - nkept;
        if (B.b_idx.empty() && !dirs.empty() && h->unprocessed_left > 0 && again < 256) { h->dir_only_windows++; continue; }
        return deal_batch(h, B, rank, world, n_local, n_total);
        }
    }
    for (;;) {
        const int pol = h->policy == 5 ? 3 : h->policy >= 3 ? 1 : h->policy;
        if ((rc = bslv_poly_unprocessed2(h->poly, 0, pol, nullptr, nullptr, nullptr, nullptr, &cnt))) return rc;
        // policy 3 looks at a window of the newest unprocessed elements several batches deep and takes at most `sib_cap`
        // children of one cut from it (see bslv_benson_set_policy)
        const long long want = h->policy == 4 ? (1LL << 30) : h->policy == 3 ? (long long)max_batch * h->sib_window : max_batch;
        int nb = (int)std::min<long long>(cnt, want);
        std::vector<int> idx(nb), ideal(nb), parent(nb);
        std::vector<double> val((size_t)nb * q);
        if (nb && (rc = bslv_poly_unprocessed2(h->poly, nb, pol, idx.data(), val.data(), ideal.data(), parent.data(), &cnt))) return rc;
        std::vector<int> dirs;
        B.b_idx.clear(); B.b_val.clear(); B.b_parent.clear(); B.b_front.clear();
        int taken = 0;
        if (h->policy == 4) {
            // K fronts, newest first inside each (at most sib_cap children of one cut per front and batch)
            const int K = std::max(1, h->nfronts);
            auto front_of = [&](int f) { return (f >= 0 && f < (int)h->facet_front.size() && h->facet_front[f] >= 0) ? h->facet_front[f] % K : 0; };
            std::vector<std::vector<int>> byf(K);                       // positions in the window, ascending slot order
            for (int k = 0; k < nb; k++) {
                if (ideal[k]) { dirs.push_back(idx[k]); taken++; continue; }
                byf[front_of(parent[k])].push_back(k);
            }
            const int quota = std::max(1, max_batch / K);
            // re-seed fronts that cannot fill their quota from the richest front: the facets of its older half move over
            for (int guard = 0; guard < 4 * K; guard++) {
                int poor = -1, rich = 0;
                for (int f = 0; f < K; f++) { if ((int)byf[f].size() < quota && (poor < 0 || byf[f].size() < byf[poor].size())) poor = f; if (byf[f].size() > byf[rich].size()) rich = f; }
                if (poor < 0 || (int)byf[rich].size() < 4 * quota || !byf[poor].empty()) break;
                const int half = (int)byf[rich].size() / 2;
                const int fsplit = parent[byf[rich][half]];              // facets below this id (older cuts) change front
                if ((int)h->facet_front.size() < bslv_poly_ndual(h->poly)) h->facet_front.resize(bslv_poly_ndual(h->poly), -1);
                std::vector<int> stay;
                for (int k : byf[rich]) {
                    const int f = parent[k];
                    if (f >= 0 && f < fsplit) { h->facet_front[f] = poor; byf[poor].push_back(k); } else stay.push_back(k);
                }
                if (byf[poor].empty()) break;
                byf[rich].swap(stay);
                h->front_splits++;
            }
            std::vector<int> pick;
            std::vector<int> want(K, quota);
            int spare = 0;
            for (int pass = 0; pass < 2; pass++) {
                for (int f = 0; f < K; f++) {
                    if (pass == 1) { if (spare <= 0) break; want[f] = spare; }
                    std::unordered_map<int, int> per_parent;
                    int got = 0;
                    std::vector<int> &L = byf[f];
                    std::vector<int> rest;
                    for (int t = (int)L.size() - 1; t >= 0; t--) {
                        const int k = L[t];
                        if (got >= want[f] || (parent[k] >= 0 && ++per_parent[parent[k]] > h->sib_cap)) { rest.push_back(k); continue; }
                        pick.push_back(k); B.b_front.push_back(f); got++;
                    }
                    std::reverse(rest.begin(), rest.end());
                    L.swap(rest);
                    if (pass == 0) spare += want[f] - got; else spare -= got;
                }
            }
            std::vector<int> ord(pick.size());
            for (size_t t = 0; t < ord.size(); t++) ord[t] = (int)t;
            std::sort(ord.begin(), ord.end(), [&](int a, int b) { return pick[a] < pick[b]; });
            std::vector<int> fr2;
            for (int t : ord) {
                const int k = pick[t];
                B.b_idx.push_back(idx[k]);
                B.b_parent.push_back(parent[k]);
                fr2.push_back(B.b_front[t]);
                B.b_val.insert(B.b_val.end(), &val[(size_t)k * q], &val[(size_t)(k + 1) * q]);
            }
            B.b_front.swap(fr2);
            taken += (int)pick.size();
        } else if (h->policy == 3) {
            // newest first; the chosen elements are handed on in ascending slot order, as with the other policies
            std::unordered_map<int, int> per_parent;
            std::vector<int> pick;
            for (int k = nb - 1; k >= 0; k--) {
                if (ideal[k]) { dirs.push_back(idx[k]); taken++; continue; }
                if ((int)pick.size() >= max_batch) continue;
                if (parent[k] >= 0 && ++per_parent[parent[k]] > h->sib_cap) continue;
                pick.push_back(k);
            }
            std::reverse(pick.begin(), pick.end());
            for (int k : pick) {
                B.b_idx.push_back(idx[k]);
                B.b_parent.push_back(parent[k]);
                B.b_val.insert(B.b_val.end(), &val[(size_t)k * q], &val[(size_t)(k + 1) * q]);
            }
            taken += (int)pick.size();
        } else {
            for (int k = 0; k < nb; k++) {
                if (ideal[k]) { dirs.push_back(idx[k]); continue; }
                B.b_idx.push_back(idx[k]);
                B.b_parent.push_back(parent[k]);
                B.b_val.insert(B.b_val.end(), &val[(size_t)k * q], &val[(size_t)(k + 1) * q]);
            }
            taken = nb;
        }
        if (!dirs.empty() && (rc = bslv_poly_mark(h->poly, (int)dirs.size(), dirs.data()))) return rc;
        h->unprocessed_left = cnt - taken;
        if (!B.b_idx.empty() || dirs.empty()) break;     // only directions in this window: look again
    }
    return deal_batch(h, B, rank, world, n_local, n_total);
}
// the batch is chosen (B.b_idx, b_val, b_parent): deal it to the ranks
static int deal_batch(bslv_benson *h, bslv_benson::BatchCtx &B, int rank, int world, int *n_local, int *n_total)
{
    int rc;
    const int nb = (int)B.b_idx.size();
    // pipelined mode: the batch is 'being processed' from now on, so the next collect skips it (a vertex that gets a
    // cut is removed by that cut; one that does not is confirmed -- either way the mark is final)
    if (h->mark_at_collect && nb && (rc = bslv_poly_mark(h->poly, nb, B.b_idx.data()))) return rc;
    // deal to ranks: owner of the parent cut if known and not overloaded, else least loaded
    B.b_owner.assign(nb, 0);
    std::vector<int> load(world, 0);
    const int cap = (nb + world - 1) / world + std::max(1, nb / (4 * world));
    for (int k = 0; k < nb; k++) {
        int f = B.b_parent[k];
        int o = (f >= 0 && f < (int)h->facet_owner.size()) ? h->facet_owner[f] : -1;
        if (o < 0 || o >= world || load[o] >= cap) o = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        B.b_owner[k] = o;
        load[o]++;
    }
    B.l_pos.clear();
    for (int k = 0; k < nb; k++) if (B.b_owner[k] == rank) B.l_pos.push_back(k);
    *n_local = (int)B.l_pos.size();
    *n_total = nb;
    return 0;
}

// Tuning hook: the caller chooses the batch itself (elements idx[] with their coordinates and parent facets, as
// bslv_poly_unprocessed2 returns them); directions among them must have been marked by the caller.
int bslv_benson_collect_given(bslv_benson *h, int n, const int *idx, const double *val, const int *parent, int rank, int world, int *n_local, int *n_total)
{
    if (!h || !h->started || n < 0 || world < 1 || rank < 0 || rank >= world || !n_local || !n_total || (n && (!idx || !val || !parent))) {
        set_error("bslv_benson_collect_given: bad argument / not started");
        return BSLV_E_ARG;
    }
    bslv_benson::BatchCtx &B = h->ctx[0];
    h->rank = rank; h->world = world;
    B.b_idx.assign(idx, idx + n);
    B.b_parent.assign(parent, parent + n);
    B.b_val.assign(val, val + (size_t)n * h->q);
    return deal_batch(h, B, rank, world, n_local, n_total);
}
// per LP of the last solve_local of this rank: warm-start slot (0 = root tableau), pivots, generation of the new slot
int bslv_benson_last_local(bslv_benson *h, int max_out, int *src, int *pivots, int *gen)
{
    if (!h) return BSLV_E_ARG;
    const int n = (int)std::min<size_t>(h->last_src.size(), (size_t)std::max(max_out, 0));
    for (int k = 0; k < n; k++) { if (src) src[k] = h->last_src[k]; if (pivots) pivots[k] = h->last_piv[k]; if (gen) gen[k] = h->last_gen[k]; }
    return n;
}

static int gen_max()      // BSLV_GEN_MAX: test hook
{
    static const int v = getenv("BSLV_GEN_MAX") ? std::max(1, atoi(getenv("BSLV_GEN_MAX"))) : 64;
    return v;
}
static int take_slot(bslv_benson *h)      // caller holds slot_mu
{
    if (h->free_slots.empty()) {
        // evict the oldest parents (FIFO); their children fall back to the root tableau
        int want = std::max(1, h->pool_slots / 8);
        while (want-- > 0 && !h->parents.empty()) {
            auto pr = h->parents.front();
            h->parents.pop_front();
            // (-2, s): a slot that was evicted while it still served as a warm-start source of a batch (solve_local re-queues
            // it under this key); nothing refers to it any more, so it goes back to the free list unconditionally
            if (pr.first == -2) { h->free_slots.push_back(pr.second); continue; }
            auto it = h->facet_slot.find(pr.first);
            if (it != h->facet_slot.end() && it->second == pr.second) { h->facet_slot.erase(it); h->free_slots.push_back(pr.second); }
        }
        if (h->free_slots.empty()) return -1;
    }
    int s = h->free_slots.back();
    h->free_slots.pop_back();
    return s;
}

// Solve this rank's shard.  records: n_local x (q+5) doubles
//   [source slot, LP status, add (z > eps), z, y*_1..y*_q, owner rank]   
int bslv_benson_solve_local_ctx(bslv_benson *h, int ctx, double *records, int *pivots_out, int *lockstep_out);
int bslv_benson_solve_local(bslv_benson *h, double *records, int *pivots_out, int *lockstep_out)
{
    return bslv_benson_solve_local_ctx(h, 0, records, pivots_out, lockstep_out);
}
int bslv_benson_solve_local_ctx(bslv_benson *h, int ctx, double *records, int *pivots_out, int *lockstep_out)
{
    if (!h || !h->started || ctx < 0 || ctx > 1) { set_error("bslv_benson_solve_local: not started"); return BSLV_E_STATE; }
    bslv_benson::BatchCtx &B = h->ctx[ctx];
    const int q = h->q, r = h->r, nl = (int)B.l_pos.size(), RL = rec_len(h);
    if (pivots_out) *pivots_out = 0;
    if (lockstep_out) *lockstep_out = 0;
    if (nl == 0) return 0;
    if (!records) return BSLV_E_ARG;
    if (nl > h->pool_slots - 1) { set_error("batch shard (%d) larger than the tableau pool (%d)", nl, h->pool_slots - 1); return BSLV_E_CAPACITY; }
    std::vector<int> src(nl), dst(nl);
    std::vector<double> vlo((size_t)nl * r, -INFINITY), vup((size_t)nl * r);
    B.l_slot.assign(nl, -1);
    std::unique_lock<std::mutex> lk(h->slot_mu);
    // sources first (eviction below must not take a slot we are about to read)
    // warm-start source: the tableau of the LP whose cut created the vertex; if that was evicted, the
    // resident tableau whose own vertex is nearest (every optimal tableau is dual feasible for every v)
    std::vector<int> cand;
    // a vertex whose parent's tableau is not here -- evicted, or (several ranks) solved on another rank and dealt to this one
    // because its owner was full -- starts from the resident tableau whose own vertex is nearest, not from the root tableau: any
    // optimal tableau is dual feasible for every v, and a neighbour's is a few pivots away where the root's is hundreds
    bool cand_built = false;
    auto build_cand = [&]() {                         // (only when a parent is missing: the walk over the resident parents is not free)
        cand_built = true;
        if (!(h->policy == 2 || h->world > 1 || h->nearest_start)) return;
        const int step = std::max(1, (int)h->parents.size() / 1024);
        int c = 0;
        for (auto &pr : h->parents) {
            if ((c++ % step) != 0 || pr.first < 0 || !h->slot_valid[pr.second]) continue;
            auto fs = h->facet_slot.find(pr.first);
            if (fs != h->facet_slot.end() && fs->second == pr.second) cand.push_back(pr.second);
        }
    };
    for (int k = 0; k < nl; k++) {
        int f = B.b_parent[B.l_pos[k]];
        auto it = h->facet_slot.find(f);
        src[k] = (it != h->facet_slot.end()) ? it->second : 0;
        if (it == h->facet_slot.end() && !cand_built) build_cand();
        if (it == h->facet_slot.end() && !cand.empty()) {
            const double *v = &B.b_val[(size_t)B.l_pos[k] * q];
            double best = INFINITY; int bs = 0;
            for (int s : cand) {
                const double *u = &h->slot_src[(size_t)s * q];
                double dd = 0;
                for (int kk = 0; kk < q; kk++) dd += (u[kk] - v[kk]) * (u[kk] - v[kk]);
                if (dd < best) { best = dd; bs = s; }
            }
            src[k] = bs;
            h->tot_nearest_starts++;
        } else if (it == h->facet_slot.end()) h->tot_root_starts++;
    }
    std::vector<char> is_src(h->pool_slots, 0);
    for (int k = 0; k < nl; k++) is_src[src[k]] = 1;
    for (int k = 0; k < nl; k++) {
        int s = -1;
        for (int tries = 0; tries < h->pool_slots + 8; tries++) {
            s = take_slot(h);
            if (s < 0) break;
            if (!is_src[s]) break;
            // an evicted parent that is still a source of this batch: keep it alive for this batch
            h->parents.emplace_back(-2, s);
            s = -1;
        }
        if (s < 0) { set_error("tableau pool exhausted (%d slots)", h->pool_slots); return BSLV_E_NOMEM; }
        // a tableau is never refactorised: bound the chain of rank-1 updates a slot carries by restarting from the root
        // tableau (slot 0) every 64 generations (gen_max()) -- a few more pivots for that LP, rounding drift that cannot pile up
        if (h->slot_gen[src[k]] >= gen_max()) src[k] = 0;
        h->slot_gen[s] = h->slot_gen[src[k]] + 1;
        dst[k] = s;
        B.l_slot[k] = s;
        const double *v = &B.b_val[(size_t)B.l_pos[k] * q];
        memcpy(&h->slot_src[(size_t)s * q], v, q * sizeof(double));
        h->slot_valid[s] = 1;
        for (int j = 0; j < r; j++) {                   // rows->ub[j] = R_j . v   (bslv_algs.c:1041-1046)
            double ub = 0;
            for (int kk = 0; kk < q; kk++) ub += h->R[(size_t)kk * r + j] * v[kk];
            vup[(size_t)k * r + j] = ub;
        }
    }
    lk.unlock();
    std::vector<int> st(nl), it(nl);
    int rc;
    {   // lazy tableaux (bslv_lpq_set_lazy): the slots of this batch get their tableau only where apply() keeps them as parents.  Not in
        // the pipelined mode, where the next batch is solved before this one's cuts are known.  BSLV_LP_LAZY=0: every LP writes its tableau
        static const bool lazy_ok = !(getenv("BSLV_LP_LAZY") && atoi(getenv("BSLV_LP_LAZY")) == 0);
        if ((rc = bslv_lpq_set_lazy(h->lp, (lazy_ok && !h->mark_at_collect) ? 1 : 0))) return rc;
    }
    if ((rc = bslv_lpq_solve_batch(h->lp, nl, src.data(), dst.data(), vlo.data(), vup.data(), st.data(), it.data()))) return rc;
    {
        // The reference's retry (bslv_lp.c:222-227: undefined -> standard basis -> solve again), in two stages.  A tableau is
        // handed down from parent to child without ever being refactorised; after hundreds of generations (ex07: 3400 outer
        // iterations) the reduced costs of a parent can have drifted past the dual feasibility tolerance, which k_prep reports
        // as UNDEFINED.  Such LPs start again from the root tableau (slot 0: the first optimal basis, one generation old), and
        // if that fails too, from the standard basis.
        std::vector<int> redo;
        if (getenv("BSLV_FORCE_RETRY")) for (int k = 1; k < nl; k += 2) st[k] = BSLV_LP_UNDEFINED;      // test hook: every other LP goes through the retry
        for (int k = 0; k < nl; k++) if (st[k] == BSLV_LP_UNDEFINED) redo.push_back(k);
        for (int stage = 0; stage < 3 && !redo.empty(); stage++) {
            // stage 2: still no result from the standard basis -- a dual simplex that stalls in degenerate pivots until the iteration
            // limit (ex09).  The extended selection (cost perturbation + primal clean-up, lp_engine.hip) is switched on for this
            // engine and stays on: a problem that needed it once needs it for its other LPs too.
            if (stage == 2) { if ((rc = bslv_lpq_set_extended(h->lp, 1))) return rc; }
            const int nr = (int)redo.size();
            std::vector<int> s2(nr), d2(nr), st2(nr), it2(nr);
            std::vector<double> lo2((size_t)nr * r, -INFINITY), up2((size_t)nr * r);
            for (int t = 0; t < nr; t++) {
                d2[t] = dst[redo[t]];
                s2[t] = stage == 0 ? 0 : d2[t];
                memcpy(&up2[(size_t)t * r], &vup[(size_t)redo[t] * r], r * sizeof(double));
                if (stage >= 1 && (rc = bslv_lpq_reset_slot(h->lp, d2[t]))) return rc;
            }
            if ((rc = bslv_lpq_solve_batch(h->lp, nr, s2.data(), d2.data(), lo2.data(), up2.data(), st2.data(), it2.data()))) return rc;
            std::vector<int> still;
            for (int t = 0; t < nr; t++) {
                st[redo[t]] = st2[t]; it[redo[t]] += it2[t];
                if (st2[t] == BSLV_LP_UNDEFINED) still.push_back(redo[t]);
            }
            h->tot_retries += nr;
            redo.swap(still);
        }
    }
    h->last_src = src; h->last_piv = it; h->last_gen.resize(nl);
    for (int k = 0; k < nl; k++) h->last_gen[k] = h->slot_gen[dst[k]];
    std::vector<double> ww((size_t)nl * q), yy((size_t)nl * q), zz(nl);
    if ((rc = bslv_lpq_get_dual(h->lp, nl, dst.data(), h->m, q, ww.data()))) return rc;                 // bslv_algs.c:1050
    if ((rc = bslv_lpq_get_primal(h->lp, nl, dst.data(), h->M + h->n, q, yy.data()))) return rc;        // :1055
    if ((rc = bslv_lpq_get_obj(h->lp, nl, dst.data(), zz.data()))) return rc;                           // :1056
    std::vector<double> alpha;
    if (h->hom) {                                                                                       // dual of the eta row (:879)
        alpha.resize(nl);
        if ((rc = bslv_lpq_get_dual(h->lp, nl, dst.data(), h->m + q + r, 1, alpha.data()))) return rc;
    }
    long piv = 0;
    for (int k = 0; k < nl; k++) {
        double *rec = records + (size_t)k * RL;
        rec[0] = B.b_idx[B.l_pos[k]];
        rec[1] = st[k];
        rec[3] = zz[k];
        double last = 0;
        if (h->hom) {                                                                                   // phase 1: y* = (w + alpha eta, alpha) (:880-883)
            for (int kk = 0; kk < q - 1; kk++) rec[4 + kk] = ww[(size_t)k * q + kk] + alpha[k] * h->eta[kk];
            last = alpha[k];
        } else {
            for (int kk = 0; kk < q - 1; kk++) rec[4 + kk] = ww[(size_t)k * q + kk];
            for (int kk = 0; kk < q; kk++) last += yy[(size_t)k * q + kk] * ww[(size_t)k * q + kk];     // y*_q = w.Px (:1059-1062)
        }
        rec[4 + q - 1] = last;
        rec[2] = (st[k] == BSLV_LP_OPTIMAL && zz[k] > h->eps) ? 1.0 : 0.0;                              // :1063
        rec[4 + q] = h->rank;
        piv += it[k];
    }
    h->tot_pivots += piv;
    if (pivots_out) *pivots_out = (int)piv;
    int ls = 0;
    bslv_lpq_last_stats(h->lp, &ls, nullptr, nullptr, nullptr);
    if (lockstep_out) *lockstep_out = ls;
    return 0;
}

// Apply ALL ranks' records (any order in; applied in ascending source slot).  stats (may be NULL):
// [0] LPs, [1] cuts applied, [2] redundant cuts, [3] confirmed vertices, [4] LP failures
int bslv_benson_apply_ctx(bslv_benson *h, int ctx, int nrec, const double *records, long *stats);
int bslv_benson_apply(bslv_benson *h, int nrec, const double *records, long *stats)
{
    return bslv_benson_apply_ctx(h, 0, nrec, records, stats);
}
int bslv_benson_apply_ctx(bslv_benson *h, int ctx, int nrec, const double *records, long *stats)
{
    if (!h || !h->started || nrec < 0 || (nrec > 0 && !records) || ctx < 0 || ctx > 1) return BSLV_E_ARG;
    bslv_benson::BatchCtx &B = h->ctx[ctx];
    const int q = h->q, RL = rec_len(h);
    std::vector<int> order(nrec);
    for (int k = 0; k < nrec; k++) order[k] = k;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return records[(size_t)a * RL] < records[(size_t)b * RL]; });
    std::vector<int> confirmed, cut_src;
    std::vector<double> cuts;
    std::unordered_set<std::string> dedupe;
    long nfail = 0, nduplicate = 0, nstatus[5] = {0, 0, 0, 0, 0};
    auto key_of = [&](const double *ys) {
        double sc = 1.0;
        for (int kk = 0; kk < q; kk++) sc = std::max(sc, std::fabs(ys[kk]));
        std::string key((size_t)q * sizeof(long long), '\0');
        for (int kk = 0; kk < q; kk++) { long long v = std::llround(ys[kk] / sc * 1e11); memcpy(&key[(size_t)kk * sizeof(long long)], &v, sizeof v); }
        return key;
    };
    for (const auto &dfr : h->deferred) dedupe.insert(key_of(dfr.ys.data()));       // (a new copy of a cut that is waiting is a duplicate too)
    for (int k : order) {
        const double *rec = records + (size_t)k * RL;
        if ((int)rec[1] != BSLV_LP_OPTIMAL) { nfail++; nstatus[std::min(std::max((int)rec[1], 0), 4)]++; continue; }
        if (rec[2] != 0.0) {
            // sibling vertices (children of one cut) mostly see the SAME facet of the upper image: an exact duplicate of
            // a cut already in this batch has no violating vertex once the first copy is applied (the reference's
            // poly__add_vrtx returns EXIT_FAILURE for it, bslv_poly.c:130-136); it is dropped here, on the host, instead
            // of paying a classification pass.  Key: coordinates relative to max(1,|y*|), rounded to 1e-11.
            if (!dedupe.insert(key_of(rec + 4)).second) { nduplicate++; continue; }
            cuts.insert(cuts.end(), rec + 4, rec + 4 + q); cut_src.push_back(k);
        }
        else confirmed.push_back((int)rec[0]);                                                           // :1074-1079
    }
    int rc;
    if (nfail) {
        set_error("%ld LP(s) of the batch did not reach optimality: %ld infeasible, %ld unbounded, %ld undefined (iteration limit or no dual feasible start) "
                  "(the reference asserts here, bslv_algs.c:1049)%s", nfail, nstatus[0], nstatus[1], nstatus[2] + nstatus[3], h->hom ? " [homogeneous problem, phase 1]" : "");
        return BSLV_E_STATE;
    }
    if (!confirmed.empty() && (rc = bslv_poly_mark(h->poly, (int)confirmed.size(), confirmed.data()))) return rc;
    if (h->want_primg && !confirmed.empty()) {                                       // x of the confirmed vertices (:1078)
        std::unordered_map<int, int> slot_of;
        for (size_t k = 0; k < B.l_pos.size(); k++) slot_of[B.b_idx[B.l_pos[k]]] = B.l_slot[k];
        std::vector<int> ids, slots;
        for (int v : confirmed) { auto it = slot_of.find(v); if (it != slot_of.end()) { ids.push_back(v); slots.push_back(it->second); } }
        std::vector<double> X((size_t)ids.size() * h->n);
        if (!ids.empty() && (rc = bslv_lpq_get_primal(h->lp, (int)ids.size(), slots.data(), h->M, h->n, X.data()))) return rc;
        for (size_t k = 0; k < ids.size(); k++) h->primg_p[ids[k]].assign(X.begin() + k * h->n, X.begin() + (k + 1) * h->n);
    }
    // the cuts of this call: those handed back by earlier calls first, then the new ones
    struct CutInfo { double z; int owner, slot, front, src; };
    std::vector<CutInfo> info;
    {
        std::vector<double> all;
        all.reserve(h->deferred.size() * q + cuts.size());
        for (const auto &dfr : h->deferred) { all.insert(all.end(), dfr.ys.begin(), dfr.ys.end()); info.push_back(CutInfo{dfr.z, dfr.owner, dfr.slot, dfr.front, -1}); }
        for (size_t c = 0; c < cut_src.size(); c++) {
            const double *rec = records + (size_t)cut_src[c] * RL;
            info.push_back(CutInfo{rec[3], (int)rec[4 + q], -1, 0, (int)rec[0]});
        }
        all.insert(all.end(), cuts.begin(), cuts.end());
        cuts.swap(all);
        h->deferred.clear();
    }
    const int ncut = (int)info.size();
    std::vector<int> prc(ncut, 0);
    const int f0 = bslv_poly_ndual(h->poly);
    // thin rounds hand their cuts back only while a later call is certain (new LPs in this one) and the backlog is small
    // (and the batch is large -- small batches have nothing but thin rounds; every waiting cut also holds a tableau of the pool)
    const int thr = (nrec >= 512 && h->world == 1 && !h->mark_at_collect && ncut - (int)cut_src.size() < std::min(h->defer_max, h->pool_slots / 8)) ? h->defer_thr : 0;
    if ((rc = bslv_poly_set_defer(h->poly, thr))) return rc;
    if (ncut) {       // the depth of every cut, for the order of the rounds (BSLV_R2_ORDER; unused by default)
        std::vector<double> zs(ncut);
        for (int c = 0; c < ncut; c++) zs[c] = info[c].z;
        if ((rc = bslv_poly_set_cut_priorities(h->poly, ncut, zs.data()))) return rc;
    }
    if (ncut && (rc = bslv_poly_add_cuts(h->poly, ncut, cuts.data(), nullptr, prc.data()))) return rc;
    // bookkeeping: facet ids f0.. were assigned in this order on every rank
    std::lock_guard<std::mutex> lk(h->slot_mu);
    h->facet_owner.resize(f0 + ncut, -1);
    h->batch_f0.push_back(f0);
    while (h->batch_f0.size() > 64) h->batch_f0.pop_front();
    h->facet_normal.resize((size_t)(f0 + ncut) * q, 0.0);
    h->facet_z.resize((size_t)(f0 + ncut), h->facet_z0);
    for (int c = 0; c < ncut; c++) h->facet_z[f0 + c] = info[c].z;
    for (int c = 0; c < ncut; c++) {                  // normal of the cut y*: (y*_1 .. y*_{q-1}, 1 - c.y*) (lowerV2upperH, bslv_algs.c:287-305), scaled to length 1
        double *nn = &h->facet_normal[(size_t)(f0 + c) * q];
        const double *ys = &cuts[(size_t)c * q];
        double last = 1.0, len = 0;
        for (int k = 0; k < q - 1; k++) { nn[k] = ys[k]; last -= h->c[k] * ys[k]; }
        nn[q - 1] = last;
        for (int k = 0; k < q; k++) len += nn[k] * nn[k];
        len = std::sqrt(std::max(len, 1e-300));
        for (int k = 0; k < q; k++) nn[k] /= len;
    }
    if (h->policy == 4) {
        h->facet_front.resize(f0 + ncut, -1);
        std::unordered_map<int, int> front_of_src;
        for (size_t k = 0; k < B.b_idx.size() && k < B.b_front.size(); k++) front_of_src[B.b_idx[k]] = B.b_front[k];
        for (int c = 0; c < ncut; c++) if (info[c].src >= 0) { auto it = front_of_src.find(info[c].src); info[c].front = it != front_of_src.end() ? it->second : 0; }
    }
    long applied = 0, handed_back = 0;
    std::vector<int> keep_slots;          // slots of this batch that later LPs will start from: the only ones that need their tableau (lazy)
    // local slot of a record: position in this rank's shard
    std::unordered_map<int, int> slot_of_src;
    for (size_t k = 0; k < B.l_pos.size(); k++) slot_of_src[B.b_idx[B.l_pos[k]]] = B.l_slot[k];
    for (int c = 0; c < ncut; c++) {
        CutInfo &ci = info[c];
        const int f = f0 + c;
        h->facet_owner[f] = ci.owner;
        if (h->policy == 4) h->facet_front[f] = ci.front;
        if (ci.src >= 0 && ci.owner == h->rank) {       // a new cut of this rank: the tableau of its LP
            auto it = slot_of_src.find(ci.src);
            if (it != slot_of_src.end()) { ci.slot = it->second; slot_of_src.erase(it); }
        }
        if (prc[c] == 0) {
            applied++;
            if (ci.slot >= 0) {
                if (h->want_primg) {                                 // (u, w) of the cut (:1066-1071)
                    std::vector<double> uw(h->m + q);
                    const int sl = ci.slot;
                    if ((rc = bslv_lpq_get_dual(h->lp, 1, &sl, 0, h->m + q, uw.data()))) return rc;
                    h->primg_d[f] = uw;
                }
                h->facet_slot[f] = ci.slot; h->parents.emplace_back(f, ci.slot);
                keep_slots.push_back(ci.slot);
            }
        } else if (prc[c] == 2) {                        // handed back: its tableau stays reserved
            handed_back++;
            if (ci.slot >= 0) keep_slots.push_back(ci.slot);
            h->deferred.push_back(bslv_benson::Deferred{std::vector<double>(&cuts[(size_t)c * q], &cuts[(size_t)(c + 1) * q]), ci.z, ci.owner, ci.slot, ci.front});
        } else if (ci.slot >= 0) h->free_slots.push_back(ci.slot);
    }
    h->tot_deferred += handed_back;
    // (lazy tableaux) the parents-to-be get their tableau now, the slots of every other LP of the batch never do
    if ((rc = bslv_lpq_materialise(h->lp, (int)keep_slots.size(), keep_slots.data()))) return rc;
    if ((rc = bslv_lpq_discard_pending(h->lp))) return rc;
    for (auto &kv : slot_of_src) h->free_slots.push_back(kv.second);    // confirmed vertices: tableau not needed again
    B.l_pos.clear(); B.l_slot.clear();
    h->tot_lps += nrec;
    h->tot_cuts += applied;
    if (stats) { stats[0] = nrec; stats[1] = applied; stats[2] = ncut - applied - handed_back + nduplicate; stats[3] = (long)confirmed.size(); stats[4] = nfail; }
    return 0;
}

// single-process outer iteration.  stats: as bslv_benson_apply, plus [5] pivots, [6] lock-step
// iterations, [7] unprocessed elements left; ms: [0] LP, [1] poly, [2] total
int bslv_benson_step(bslv_benson *h, int max_batch, long *stats, double *ms)
{
    if (!h) return BSLV_E_ARG;
    // more than one rank (bslv_dist_init): the batch is dealt to the ranks and the records are exchanged (dist.hip)
    if (bslv_dist_world() > 1) return bslv_benson_step_dist(h, max_batch, stats, ms);
    auto t0 = clk::now();
    int nl = 0, nt = 0, rc;
    if ((rc = bslv_benson_collect(h, max_batch, 0, 1, &nl, &nt))) return rc;
    auto t1 = clk::now();
    std::vector<double> rec((size_t)std::max(1, nl) * rec_len(h));
    int piv = 0, ls = 0;
    if ((rc = bslv_benson_solve_local(h, rec.data(), &piv, &ls))) return rc;
    double ms_lp = ms_since(t1);
    auto t2 = clk::now();
    long st[5] = {0, 0, 0, 0, 0};
    if ((rc = bslv_benson_apply(h, nl, rec.data(), st))) return rc;
    if (stats) { for (int k = 0; k < 5; k++) stats[k] = st[k]; stats[5] = piv; stats[6] = ls; stats[7] = h->unprocessed_left; }
    if (ms) { ms[0] = ms_lp; ms[1] = ms_since(t2) + std::chrono::duration<double, std::milli>(t1 - t0).count(); ms[2] = ms_since(t0); }
    return 0;
}

int bslv_benson_unprocessed_left(const bslv_benson *h) { return h ? h->unprocessed_left : 0; }
int bslv_benson_set_pipelined(bslv_benson *h, int on)
{
    if (!h) return BSLV_E_ARG;
    h->mark_at_collect = on ? 1 : 0;
    return 0;
}
int bslv_benson_set_policy(bslv_benson *h, int policy)
{
    if (!h || policy < 1 || policy > 6) return BSLV_E_ARG;
    h->policy = policy;
    return 0;
}
int bslv_benson_set_families(bslv_benson *h, int mode, int batches)
{
    if (!h || mode < 0 || mode > 5 || batches < 1 || batches > 64) return BSLV_E_ARG;
    h->fam_mode = mode; h->fam_batches = batches;
    return 0;
}
// min_cuts > 0: the polyhedron engine may hand the cuts of thin rounds back (bslv_poly_set_defer); they go in again with the next batch.
int bslv_benson_set_defer(bslv_benson *h, int min_cuts)
{
    if (!h || min_cuts < 0) return BSLV_E_ARG;
    h->defer_thr = min_cuts;
    return 0;
}
// [0] cuts handed back so far (a cut counts every time), [1] cuts waiting now, [2] times collect had to apply the waiting cuts before it
// could answer, [3] batches that one family would have filled (taken newest first)
int bslv_benson_defer_stats(const bslv_benson *h, long out[4])
{
    if (!h || !out) return BSLV_E_ARG;
    out[0] = h->tot_deferred; out[1] = (long)h->deferred.size(); out[2] = h->defer_flushes; out[3] = h->fam_fallbacks;
    return 0;
}
int bslv_benson_set_fronts(bslv_benson *h, int nfronts, int sib_cap)
{
    if (!h || nfronts < 1 || nfronts > 1024 || sib_cap < 1) return BSLV_E_ARG;
    h->nfronts = nfronts; h->sib_cap = sib_cap;
    return 0;
}
int bslv_benson_set_sibling_rule(bslv_benson *h, int cap, int window)
{
    if (!h || cap < 1 || window < 1) return BSLV_E_ARG;
    h->sib_cap = cap; h->sib_window = window;
    return 0;
}
// slots of the tableau pool: [0] free, [1] resident parents (warm-start sources), [2] held by the batch contexts, [3] pool size.
// free + resident + held = pool size - 1 (slot 0 is the root tableau) whenever no batch is between solve_local and apply.
int bslv_benson_pool_stats(bslv_benson *h, long out[4])
{
    if (!h || !out) return BSLV_E_ARG;
    std::lock_guard<std::mutex> lk(h->slot_mu);
    long resident = 0;
    for (auto &pr : h->parents) {
        if (pr.first == -2) { resident++; continue; }
        auto it = h->facet_slot.find(pr.first);
        if (it != h->facet_slot.end() && it->second == pr.second) resident++;
    }
    long held = 0;
    for (int c = 0; c < 2; c++) for (int s : h->ctx[c].l_slot) if (s >= 0) held++;
    out[0] = (long)h->free_slots.size(); out[1] = resident; out[2] = held; out[3] = h->pool_slots;
    return 0;
}
// pre-images (option -s): x[n] of a vertex of the upper image / (u[m], w[q]) of a vertex of the lower image.  Returns 0 and
// fills out, or 1 when nothing was stored for that element (directions; elements that were never confirmed).
int bslv_benson_preimage_p(const bslv_benson *h, int element, double *x)
{
    if (!h || !x) return BSLV_E_ARG;
    auto it = h->primg_p.find(element);
    if (it == h->primg_p.end()) return 1;
    memcpy(x, it->second.data(), it->second.size() * sizeof(double));
    return 0;
}
int bslv_benson_preimage_d(const bslv_benson *h, int facet, double *uw)
{
    if (!h || !uw) return BSLV_E_ARG;
    auto it = h->primg_d.find(facet);
    if (it == h->primg_d.end()) return 1;
    memcpy(uw, it->second.data(), it->second.size() * sizeof(double));
    return 0;
}
int bslv_benson_set_preimage_p(bslv_benson *h, int element, const double *x)
{
    if (!h || !x) return BSLV_E_ARG;
    h->primg_p[element].assign(x, x + h->n);
    return 0;
}
int bslv_benson_lp_dims(const bslv_benson *h, int *M, int *N, int *rows_folded)
{
    if (!h) return BSLV_E_ARG;
    if (M) *M = h->M;                                // (rows of the model as given: what every index at the LP boundary refers to; the tableau has rows_folded fewer)
    if (N) *N = h->N;
    if (rows_folded) *rows_folded = h->rows_folded;
    return 0;
}
// LPs whose parent's tableau was not resident on this rank: started from the root tableau / from the nearest resident tableau
int bslv_benson_start_stats(const bslv_benson *h, long *root_starts, long *nearest_starts)
{
    if (!h) return BSLV_E_ARG;
    if (root_starts) *root_starts = h->tot_root_starts;
    if (nearest_starts) *nearest_starts = h->tot_nearest_starts;
    return 0;
}
int bslv_benson_totals(const bslv_benson *h, long *lps, long *cuts, long *pivots)
{
    if (!h) return BSLV_E_ARG;
    if (lps) *lps = h->tot_lps;
    if (cuts) *cuts = h->tot_cuts;
    if (pivots) *pivots = h->tot_pivots;
    return 0;
}

}  // extern "C"
