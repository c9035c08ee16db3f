// lp_engine.hip -- batched bounded dual simplex on HBM-resident dense tableaux (gfx950).
//
// Replaces, for whole batches of right-hand sides, what the reference does one LP at a time
// through GLPK: lp_set_rows + lp_solve + getters (bslv_lp.c:112-116, 219-259, 261-308) as driven
// by phase2_primal's loop (bslv_algs.c:1041-1062).
//
// Data layout in HBM (one "slot" per LP):
//   T     (M+1) x ld doubles, row-major, ld = N rounded up to 16 doubles (128-B rows);
//         rows 0..M-1: x_B = T x_N ; row M: reduced costs d (objective = d . x_N)
//   beta  M+1 doubles: values of the basic variables, beta[M] = objective value (without shift)
//   xN    ld doubles: values of the nonbasic variables (padding = 0)
//   bh[M], nh[N] basis heads (variable ids: 0..M-1 aux, M..M+N-1 structural)
//   nstat[N] nonbasic status, pos[M+N] (row if basic, -1-col if nonbasic)
// One ROUND of the lock-step loop = KP x k_select (one workgroup per LP: leaving row, Harris ratio test on the tableau as it
// would be after the pivots still pending -- rebuilt from the stored tableau and the pending (pivot row, multipliers) pairs)
// + k_flush (persistent grid over (LP, row tile): pure HBM streaming, every tableau element read once, all pending pivots
// applied to it in order, written once -> 16 B per element per ROUND instead of per pivot).
#include "common.h"
#include <vector>
#include <algorithm>
#include <chrono>

namespace bslv {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// debugging aids of common.h (malloc0 / grow): fill byte of fresh device memory, allocation log
int debug_fill()
{
    static const int v = [] { const char *e = getenv("BSLV_FILL"); return e ? (int)(strtol(e, nullptr, 0) & 0xFF) : 0; }();
    return v;
}
void debug_note_alloc(const char *name, const void *p, size_t bytes, const char *file, int line)
{
    static const char *path = getenv("BSLV_ALLOC_LOG");
    if (!path) return;
    FILE *f = fopen(path, "a");
    if (!f) return;
    const char *base = strrchr(file, '/');
    fprintf(f, "%s %p %zu %s:%d\n", name, p, bytes, base ? base + 1 : file, line);
    fclose(f);
}

constexpr int NS_L = 0, NS_U = 1, NS_F = 2, NS_S = 3;
constexpr int ST_RUNNING = -1;
constexpr int MODE_NONE = 0, MODE_PIVOT = 1, MODE_REFRESH = 2;
constexpr int PRIMAL_STALL = 500;           // consecutive degenerate steps of a primal clean-up before Bland's rule takes over (until a step has a length again)
constexpr int PF_PERT = 1, PF_PRIMAL = 2, PF_PERT_PENDING = 4, PF_USES_SHIFT = 8;   // BatchView::pflags
constexpr int STALL_LIMIT = 3;         // consecutive degenerate pivots (dual step <= 1e-11) after which the costs are perturbed
constexpr int PERT_MAX_USES = 3;       // perturbations per solve
#ifndef BSLV_KP
#define BSLV_KP 6
#endif
constexpr int KP = BSLV_KP;                  // pivots selected between two passes over the tableau (delayed update; 4: 3.4 ms, 6: 3.0, 8: 3.1 ms per S-mid batch)
constexpr int REFRESH_AFTER = 32;      // pivots of one solve after which optimality is only declared on a recomputed beta
constexpr double TOL_BND = 1e-9, TOL_DJ = 1e-9, TOL_PIV = 1e-9;
constexpr double BIG = 1e7;   // artificial bound for dual-infeasible free columns
constexpr int TR = 32;        // tableau rows per workgroup in k_flush / k_init
constexpr int NT = 256;       // threads per workgroup
constexpr int FLIP_INCR_MAX = 48;   // bound switches of one iteration that beta follows by a vector update (more: recomputed in the pass)
constexpr int NT_BIG = 1024;  // k_flush where its LDS footprint leaves room for one or two workgroups per CU

struct PivDesc { int r, q; double p, pbeta, enter_val; };

struct LpView {
    int M, N, ld, Mp1, Mp1p, vfirst, vcnt, maxit, bland_after, trace, stall_limit;
    int objmode, cfirst, ccnt;  // solve_batch_obj: the LPs of the batch differ in the cost of variables cfirst .. cfirst+ccnt-1 (BatchView::cvals)
    size_t slotT;
    double pert_scale;          // multiplies the cost perturbation (1; tests raise it to force the clean-up paths)
    double *T, *beta, *xN;
    int *bh, *nh, *nstat, *pos;
    const double *lb, *ub;
    const unsigned char *art;   // bit0: lb is artificial, bit1: ub is artificial
    // REVISED FORM (round 4; bslv_lpq_create chooses it for wide sparse problems -- ex07 / ex09 of the reference's suite, which hands A
    // to GLPK as COO, bslv_lp.c:60-70): the matrix of a slot is the BASIS INVERSE B^-1 (M x ldt) instead of the tableau
    // T = -B^-1 N ((M+1) x ld), A is kept once, as CSC and CSR, for the whole pool.  With K = [I | -A] (column k of K belongs to
    // variable k: e_k for the auxiliary variable of row k, -A_j for structural j) a basis is B = K[:, bh] and
    //   row r of the tableau     T[r][j] = -rho . K_nh[j],  rho = row r of B^-1     (one sparse dot product per nonbasic column)
    //   column q of the tableau  T[:, q] = -B^-1 K_nh[q]                            (a few columns of B^-1)
    // and a pivot updates B^-1 by the SAME row operations it applies to the tableau -- row r := -rho p, row i -= f_i rho -- so the
    // delayed update, k_flush and its roofline carry over; only the tableau's column swap (entry q of every row) has no counterpart.
    // ex09: 171 MB per slot instead of 1.36 GB.  rev == 0: ldt = ld, mrows = M + 1 and everything is as before.
    int rho_off;                // rev: byte offset of the LDS copy of rho (row of B^-1 the tableau row is built from) in k_select's dynamic LDS, or -1 (does not fit: gathers from global memory)
    int helpers, launch_id;      // revised form: workgroups per LP in k_select (1 = none besides the LP's own) that share the sparse products of a tableau row; a number per launch for their mailbox
    int rev, ldt, mrows, probe;  // probe: BSLV_REV_PROBE, timing experiments only (parts of the revised selection skipped: results are WRONG)
    double *dsl;                // rev: [slots][ld] reduced costs of each slot (the tableau form keeps them as row M of T)
    const int *cptr, *cidx; const double *cval;   // rev: CSC of A
    const int *rptr, *ridx; const double *rval;   // rev: CSR of A
};
struct BatchView {
    const int *src, *dst;
    const double *vlo, *vup;
    int *status, *iters, *mode, *verified;    // verified: bit 0 = beta is fresh, bit 1 = the solve started with a variable on an artificial bound
    // Delayed update: up to KP pivots of an LP are SELECTED on vectors only (the pivot row and the entering column of the
    // tableau as it would be after the pending pivots, the reduced-cost row dcur, beta) and then applied to the tableau in
    // ONE pass (k_flush): a solve of <= KP pivots reads and writes its tableau once instead of once per pivot.
    int *npend, *flushed;       // pending pivots of the LP; has the tableau of this solve been written to its own slot yet
    PivDesc *desc;              // [B][KP]
    double *prow;               // [B][KP][ld]   pivot rows as they were when chosen
    double *pcol;               // [B][KP][Mp1p] multipliers f_i = (entering column)_i * p of every row i (0 for the pivot row)
    double *dcur;               // [B][ld]       reduced-cost row of the LP, up to date
    // Extended selection (k_select<true>): cost perturbation against dual degenerate stalling, primal clean-up afterwards
    double *dper;               // [B][ld]       perturbed reduced costs (the ratio tests use them while PF_PERT is set)
    int *pflags, *stall;        // PF_* bits; consecutive degenerate pivots
    const double *cvals;        // [B][ccnt] objective coefficients (objmode)
    int *xstat;                 // [5] of the batch: iterations with bound switches, perturbations, primal steps, removals that left wrong signs, switch iterations carried into beta without a pass
    int *work, *nwork;      // LPs whose tableau k_flush passes over in a round (k_list_pending), their number per round
    // revised form
    double *trow;           // [B][ld]   the tableau row of the selection at hand (prow then holds rows of B^-1: [B][KP][ldt])
    double *uvec;           // [B][ldt]  -K_N x_N: beta = B^-1 uvec (k_rev_u; k_init and the refresh pass of k_flush multiply by it)
    double *xfull;          // [B][N]    scratch of k_rev_u: values of the nonbasic structurals by column
    int *hmail;             // [B][8]    revised form, helper workgroups of k_select: request word, slices done, pending count, slice ticket
    int lazy;               // bslv_lpq_set_lazy: an LP that is finished when its pass would be due keeps its pending pivots; its slot gets the tableau only when asked for (bslv_lpq_materialise)
    unsigned long long *dbg; // BSLV_REV_PROBE & 8: 100 MHz clock ticks per phase of the dual selection of LP 0 (timing experiments)
};

__device__ __forceinline__ double LO(const LpView &L, const BatchView &Bv, int b, int k)
{
    int j = k - L.vfirst;
    return (j >= 0 && j < L.vcnt) ? Bv.vlo[(size_t)b * L.vcnt + j] : L.lb[k];
}
__device__ __forceinline__ double UP(const LpView &L, const BatchView &Bv, int b, int k)
{
    int j = k - L.vfirst;
    return (j >= 0 && j < L.vcnt) ? Bv.vup[(size_t)b * L.vcnt + j] : L.ub[k];
}
__device__ __forceinline__ double btol(double bnd) { return TOL_BND * (1.0 + fabs(bnd)); }

// ---- k_prep: copy the small per-slot arrays src -> dst, sanitise nonbasic statuses against the
//      new bounds and set the nonbasic values (oracle/lp_dense.c sanitize()) ----
__global__ __launch_bounds__(NT) void k_prep(LpView L, BatchView Bv, int B)
{
    int b = blockIdx.x;
    if (b >= B) return;
    int src = Bv.src[b], dst = Bv.dst[b];
    const int *bh_s = L.bh + (size_t)src * L.M, *nh_s = L.nh + (size_t)src * L.N;
    const int *ns_s = L.nstat + (size_t)src * L.N, *pos_s = L.pos + (size_t)src * (L.M + L.N);
    int *bh_d = L.bh + (size_t)dst * L.M, *nh_d = L.nh + (size_t)dst * L.N;
    int *ns_d = L.nstat + (size_t)dst * L.N, *pos_d = L.pos + (size_t)dst * (L.M + L.N);
    double *xN_d = L.xN + (size_t)dst * L.ld;
    const double *drow_s = L.rev ? L.dsl + (size_t)src * L.ld : L.T + (size_t)src * L.slotT + (size_t)L.M * L.ld;
    if (src != dst) {
        for (int i = threadIdx.x; i < L.M; i += NT) bh_d[i] = bh_s[i];
        for (int i = threadIdx.x; i < L.M + L.N; i += NT) pos_d[i] = pos_s[i];
    }
    int dual_infeasible = 0, bigm = 0;     // bigm: a nonbasic variable sits on an artificial (+-1e7) bound
    for (int j = threadIdx.x; j < L.ld; j += NT) {
        if (j >= L.N) { xN_d[j] = 0.0; continue; }
        int k = nh_s[j];
        double lo = LO(L, Bv, b, k), up = UP(L, Bv, b, k);
        int st = ns_s[j];
        if (lo == up) st = NS_S;
        else if (isinf(lo) && isinf(up)) st = NS_F;
        else if (isinf(lo)) st = NS_U;
        else if (isinf(up)) st = NS_L;
        else {
            // boxed: sit at the bound that keeps the reduced cost dual feasible
            double dj = L.objmode ? 0.0 : drow_s[j];       // (new objective: stay where the parent was, that is primal feasible)
            if (dj < -TOL_DJ) st = NS_U;
            else if (dj > TOL_DJ) st = NS_L;
            else if (st != NS_L && st != NS_U) st = NS_L;
        }
        // the dual simplex needs a dual feasible start: a bound that vanished under a non-zero
        // reduced cost cannot be repaired by a flip (the reference's GLPK would run its primal phase)
        {
            double dj = drow_s[j];
            if (!L.objmode && ((st == NS_F && fabs(dj) > 1e-7) || (st == NS_L && dj < -1e-7) || (st == NS_U && dj > 1e-7))) dual_infeasible = 1;
        }
        { const unsigned char a = L.art[k]; if ((st == NS_L && (a & 1)) || (st == NS_U && (a & 2))) bigm = 1; }
        nh_d[j] = k;
        ns_d[j] = st;
        xN_d[j] = (st == NS_F) ? 0.0 : (st == NS_U ? up : lo);
    }
    {   // working copy of the reduced-cost row (k_select keeps it up to date between passes over the tableau)
        double *dc = Bv.dcur + (size_t)b * L.ld;
        if (!L.objmode) for (int j = threadIdx.x; j < L.ld; j += NT) dc[j] = j < L.N ? drow_s[j] : 0.0;
        else {
            // new objective: d_j = c_{nh[j]} + sum over the basic cost-carrying variables c_k T[row of k][j], from the parent's
            // tableau; it also becomes row M of the new slot (k_init takes beta_M = d . x_N from there)
            const double *Ts = L.T + (size_t)src * L.slotT;
            double *rowM = L.T + (size_t)dst * L.slotT + (size_t)L.M * L.ld;
            const double *cv = Bv.cvals + (size_t)b * L.ccnt;
            for (int j = threadIdx.x; j < L.ld; j += NT) {
                double v = 0.0;
                if (j < L.N) {
                    const int kj = nh_s[j] - L.cfirst;
                    if (kj >= 0 && kj < L.ccnt) v = cv[kj];
                    for (int t = 0; t < L.ccnt; t++) { const int pr = pos_s[L.cfirst + t]; if (pr >= 0) v = fma(cv[t], Ts[(size_t)pr * L.ld + j], v); }
                }
                dc[j] = v;
                rowM[j] = v;
            }
        }
    }
    dual_infeasible = __syncthreads_or(dual_infeasible);
    bigm = __syncthreads_or(bigm);
    if (threadIdx.x == 0) {
        Bv.status[b] = dual_infeasible ? BSLV_LP_UNDEFINED : ST_RUNNING;
        Bv.iters[b] = 0;
        Bv.mode[b] = MODE_NONE;
        Bv.verified[b] = 1 | (bigm ? 2 : 0);   // k_init recomputes beta from scratch
        Bv.npend[b] = 0;
        Bv.pflags[b] = L.objmode ? PF_PRIMAL : 0;       // a new objective on a primal feasible basis: primal simplex steps
        Bv.stall[b] = 0;
        Bv.flushed[b] = (src == dst) || L.objmode;      // (objmode: the tableau is copied up front, see solve_batch)            // (in place: the slot already holds the tableau)
    }
}

// ---- k_init: beta_dst = T_src . xN_dst, one wave per row, and the reduced-cost row T_dst[M] = T_src[M].  The rest of the
//      parent's tableau is NOT copied here: the first pivot of the solve reads the parent and writes the new slot
//      (k_flush), which saves one write and one read of the tableau per LP; solves without a pivot are copied by
//      k_copy_unpivoted at the end. ----
__global__ __launch_bounds__(NT) void k_init(LpView L, BatchView Bv, int B)
{
    int b = blockIdx.y;
    if (b >= B) return;
    int src = Bv.src[b], dst = Bv.dst[b];
    const double *Ts = L.T + (size_t)src * L.slotT;
    double *Td = L.T + (size_t)dst * L.slotT;
    const double *xN = L.rev ? Bv.uvec + (size_t)b * L.ldt : L.xN + (size_t)dst * L.ld;      // (rev: beta = B^-1 uvec, k_rev_u; beta[M] comes from there too)
    double *beta = L.beta + (size_t)dst * L.Mp1p;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int ld2 = L.ldt >> 1;
    bool copy = (src != dst) && !L.rev;
    for (int rr = wave; rr < TR; rr += NT / WAVE) {
        int i = blockIdx.x * TR + rr;
        if (i >= L.mrows) break;
        const double2 *s = reinterpret_cast<const double2 *>(((L.objmode && i == L.M) ? Td : Ts) + (size_t)i * L.ldt);   // (objmode: k_prep wrote the new row M)
        double2 *d = reinterpret_cast<double2 *>(Td + (size_t)i * L.ldt);
        const double2 *x2 = reinterpret_cast<const double2 *>(xN);
        double acc = 0.0;
        for (int j2 = lane; j2 < ld2; j2 += WAVE) {
            double2 v = s[j2];
            double2 x = x2[j2];
            if (copy && i == L.M && !L.objmode) d[j2] = v;      // only the reduced-cost row: the first pass streams the rest from the parent (k_flush)
            acc = fma(v.x, x.x, acc);
            acc = fma(v.y, x.y, acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) beta[i] = acc;
    }
}

// block-wide helpers (for the block they are called from: 4 waves with NT threads, 16 with NT_BIG)
__device__ __forceinline__ ValIdx block_argmax(ValIdx x, double *sv, int *si)
{
    x = wave_argmax(x);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) { sv[wave] = x.v; si[wave] = x.i; }
    __syncthreads();
    ValIdx r{sv[0], si[0]};
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) r = better_max(r, ValIdx{sv[w], si[w]});
    return r;
}
__device__ __forceinline__ double block_max(double v, double *sv)
{
    v = wave_max(v);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sv[wave] = v;
    __syncthreads();
    double r = sv[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) r = fmax(r, sv[w]);
    return r;
}
__device__ __forceinline__ double block_min(double v, double *sv)
{
    v = wave_min(v);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sv[wave] = v;
    __syncthreads();
    double r = sv[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); w++) r = fmin(r, sv[w]);
    return r;
}

// ---- k_select: dual simplex choice of (leaving row r, entering column q) for each running LP.
//      Same rules as oracle/lp_dense.c dual_simplex(): largest bound violation, Harris two-pass
//      ratio test with the largest |pivot| among the ties. ----
// entry (i, j) of the tableau as it is after the pending pivots 0..np-1, given its value v0 in the stored tableau
// (same operations, in the same order, as k_flush applies them -- and as one rank-1 update per pivot did)
__device__ __forceinline__ double virt_entry(double v, int i, int j, int np, const PivDesc *pd, const double *prow, const double *pcol, int ld, int Mp1p)
{
    for (int s = 0; s < np; s++) {
        const PivDesc d = pd[s];
        if (i == d.r) v = j == d.q ? d.p : -prow[(size_t)s * ld + j] * d.p;
        else { const double f = pcol[(size_t)s * Mp1p + i]; v = j == d.q ? f : fma(-f, prow[(size_t)s * ld + j], v); }
    }
    return v;
}

// revised form: entry (i, c) of B^-1 as it is after the pending pivots, given its stored value (no column swap: see LpView)
__device__ __forceinline__ double virt_entry_b(double v, int i, int c, int np, const PivDesc *pd, const double *prow, const double *pcol, int ldt, int Mp1p)
{
    for (int s = 0; s < np; s++) {
        const PivDesc d = pd[s];
        if (i == d.r) v = -prow[(size_t)s * ldt + c] * d.p;
        else v = fma(-pcol[(size_t)s * Mp1p + i], prow[(size_t)s * ldt + c], v);
    }
    return v;
}
// revised form: uvec = -K_N x_N of LP b (one workgroup per LP), and beta[M] = d . x_N.  which: nullptr = every LP of the batch,
// else the work list of round `it` (only the LPs waiting for a refresh of beta are done).
__global__ __launch_bounds__(NT) void k_rev_u(LpView L, BatchView Bv, int B, const int *which, int it)
{
    __shared__ double sv[NT / WAVE];
    int b = blockIdx.x;
    if (which) { if (b >= Bv.nwork[it]) return; b = Bv.work[b]; if (Bv.mode[b] != MODE_REFRESH) return; }
    else if (b >= B) return;
    const int slot = Bv.dst[b], M = L.M, N = L.N;
    const int *nh = L.nh + (size_t)slot * N, *pos = L.pos + (size_t)slot * (M + N);
    const double *xN = L.xN + (size_t)slot * L.ld;
    double *xf = Bv.xfull + (size_t)b * N, *u = Bv.uvec + (size_t)b * L.ldt;
    for (int j = threadIdx.x; j < N; j += NT) xf[j] = 0.0;
    __syncthreads();
    double acc = 0.0;
    const double *dc = Bv.dcur + (size_t)b * L.ld;
    for (int j = threadIdx.x; j < N; j += NT) {
        const int k = nh[j];
        const double v = xN[j];
        if (k >= M) xf[k - M] = v;
        acc = fma(dc[j], v, acc);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L.ldt; i += NT) {
        double ui = 0.0;
        if (i < M) {
            const int p = pos[i];
            if (p < 0) ui = xN[-1 - p];                        // the auxiliary variable of row i is nonbasic: + e_i x_i
            for (int t = L.rptr[i]; t < L.rptr[i + 1]; t++) ui = fma(-L.rval[t], xf[L.ridx[t]], ui);
        }
        u[i] = -ui;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < NT / WAVE; w++) t += sv[w]; L.beta[(size_t)slot * L.Mp1p + M] = t; }
}
// revised form: the reduced costs of every LP of the batch go to its slot (the tableau form has them in row M, which k_flush updates)
__global__ void k_rev_store_d(LpView L, BatchView Bv, int B)
{
    const int b = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && j < L.ld) L.dsl[(size_t)Bv.dst[b] * L.ld + j] = Bv.dcur[(size_t)b * L.ld + j];
}
__global__ void k_rev_identity(LpView L, int slot, const double *cost)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < (size_t)L.M * L.ldt) { const int i = (int)(k / L.ldt), c = (int)(k % L.ldt); L.T[(size_t)slot * L.slotT + k] = i == c ? 1.0 : 0.0; }
    if (k < (size_t)L.ld) L.dsl[(size_t)slot * L.ld + k] = k < (size_t)L.N ? cost[k + 1] : 0.0;
}

// EXT = the extended selection, compiled in when the LP has boxed variables (two finite, non-artificial bounds):
//  * bound flipping ("long step") ratio test: the dual step passes the breakpoints of boxed candidates -- they switch to
//    their other bound instead of entering the basis -- for as long as the leaving row stays infeasible.  Without it every
//    boxed column with a zero reduced cost costs one degenerate pivot (hypercube rows of S-degenerate: hundreds of thousands);
//  * cost perturbation after STALL_LIMIT consecutive degenerate pivots (the ratio tests then use dper, every nonbasic reduced
//    cost moved 5e-7..1e-6 away from zero on its feasible side; the true reduced costs dcur are carried along);
//  * when the perturbed problem is solved the perturbation is taken away: boxed columns whose true reduced cost has the wrong
//    sign switch bound (dual simplex goes on), any other wrong sign is repaired by PRIMAL simplex pivots from the primal
//    feasible basis at hand (Dantzig pricing, Harris ratio test on the entering column) -- oracle/lp_dense.c does the same
//    with its primal_simplex().
// cap2 = capacity of the candidate arrays in dynamic LDS.
__device__ __forceinline__ double hash01(int k)
{
    unsigned x = (unsigned)k * 2654435761u;
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13;
    return (double)(x >> 8) * (1.0 / 16777216.0);
}
// ---- revised form: the tableau row as sparse products, by slices that several workgroups take ----
// One sparse dot product per nonbasic column j in [j0, j1): row[j] = -rho[k] for a slack k = nh[j] < M, else -(rho . A_k) ... as the
// caller's sign convention has it (K = [I | -A]: the column of structural k is -A_k; cval holds -A).  The non-zeros of a column are fetched
// EIGHT at a time with independent loads (index, value, then the gathers from rho) and two columns are in flight per thread: entry after
// entry, each a chain of three dependent loads, cost 0.7 ms per selection on ex09 (37 000 columns, one workgroup).
// Slice t of ns takes the columns t, t + ns, t + 2 ns, ...: the few dense columns sit next to each other in the list and would all fall to one slice of a contiguous split.
__device__ __forceinline__ void rev_row_slice(const LpView &L, const double *brow, const int *nh, double *row, int t_sl, int ns_sl)
{
    const int tid = threadIdx.x, NT = (int)blockDim.x, M = L.M, N = L.N;
    constexpr int RU = 8;
    auto dot8 = [&](int beg, int end) -> double {
        double v = 0.0;
        for (int t0 = beg; t0 < end; t0 += RU) {
            int ix[RU]; double va[RU], rh[RU];
#pragma unroll
            for (int u = 0; u < RU; u++) { const bool in = t0 + u < end; ix[u] = in ? L.cidx[t0 + u] : 0; va[u] = in ? L.cval[t0 + u] : 0.0; }
#pragma unroll
            for (int u = 0; u < RU; u++) rh[u] = brow[ix[u]];
#pragma unroll
            for (int u = 0; u < RU; u++) v = fma(rh[u], va[u], v);
        }
        return v;
    };
    // A column with many non-zeros (ex09: 75 of its 36 939 columns hold 512 or 1024, the others 2 or 4) is not one thread's to sum -- 128
    // rounds of dependent loads, 150 us, while the other 1023 threads of the workgroup wait: it goes to a queue in LDS and a whole wave
    // takes it, 64 non-zeros per round.
    constexpr int HEAVY = 32, HQ = 512;
    __shared__ int hq[HQ];
    __shared__ int hq_n;
    if (tid == 0) hq_n = 0;
    __syncthreads();
    const int j1 = L.ld;
    for (int i = tid; t_sl + ns_sl * i < j1; i += 2 * NT) {
        const int j = t_sl + ns_sl * i, j2 = t_sl + ns_sl * (i + NT);
        const int k1 = j < N ? nh[j] : -1, k2 = (j2 < j1 && j2 < N) ? nh[j2] : -1;
        int b1 = k1 >= M ? L.cptr[k1 - M] : 0, e1 = k1 >= M ? L.cptr[k1 - M + 1] : 0;
        int b2 = k2 >= M ? L.cptr[k2 - M] : 0, e2 = k2 >= M ? L.cptr[k2 - M + 1] : 0;
        bool q1 = false, q2 = false;
        if (e1 - b1 > HEAVY) { const int s = atomicAdd(&hq_n, 1); if (s < HQ) { hq[s] = j; q1 = true; e1 = b1; } }
        if (e2 - b2 > HEAVY) { const int s = atomicAdd(&hq_n, 1); if (s < HQ) { hq[s] = j2; q2 = true; e2 = b2; } }
        double v1 = (k1 >= 0 && k1 < M) ? -brow[k1] : 0.0, v2 = (k2 >= 0 && k2 < M) ? -brow[k2] : 0.0;
        if (e1 - b1 <= RU && e2 - b2 <= RU) {             // (the usual case: both columns in one round of loads)
            int ix[2 * RU]; double va[2 * RU], rh[2 * RU];
#pragma unroll
            for (int u = 0; u < RU; u++) {
                const bool i1 = b1 + u < e1, i2 = b2 + u < e2;
                ix[u] = i1 ? L.cidx[b1 + u] : 0; va[u] = i1 ? L.cval[b1 + u] : 0.0;
                ix[RU + u] = i2 ? L.cidx[b2 + u] : 0; va[RU + u] = i2 ? L.cval[b2 + u] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 2 * RU; u++) rh[u] = brow[ix[u]];
#pragma unroll
            for (int u = 0; u < RU; u++) { v1 = fma(rh[u], va[u], v1); v2 = fma(rh[RU + u], va[RU + u], v2); }
        } else { v1 += dot8(b1, e1); v2 += dot8(b2, e2); }
        if (!q1) row[j] = v1;
        if (j2 < j1 && !q2) row[j2] = v2;
    }
    __syncthreads();
    {
        const int nq = min(hq_n, HQ), lane = tid & (WAVE - 1), wv = tid / WAVE, nwv = NT / WAVE;
        for (int s = wv; s < nq; s += nwv) {
            const int j = hq[s], k = nh[j];
            const int beg = L.cptr[k - M], end = L.cptr[k - M + 1];
            double v = 0.0;
            for (int t0 = beg + lane; t0 < end; t0 += 4 * WAVE) {       // (four rounds in flight)
                int ix[4]; double va[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int t = t0 + u * WAVE; const bool in = t < end; ix[u] = in ? L.cidx[t] : 0; va[u] = in ? L.cval[t] : 0.0; }
#pragma unroll
                for (int u = 0; u < 4; u++) v = fma(brow[ix[u]], va[u], v);
            }
#pragma unroll
            for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
            if (lane == 0) row[j] = v;
        }
    }
    __syncthreads();
}
// The slices of a row are dealt by a ticket (hmail[3]) to whoever asks: the LP's own workgroup and its helpers -- workgroups of the same
// k_select launch (blockIdx.y > 0) that do nothing but wait for a request.  ONE workgroup per LP is what a selection is, and on ex09 the
// products over 36 865 columns were 300 of its 450 us.  Nothing waits for a workgroup that is not running: a slice nobody else took is
// taken by the LP's own workgroup, so helpers that the chip has no room for (or that gave up waiting) only cost their share.
// hmail[0]: request word = launch_id << 8 | n (n-th request of this launch; 0xFF: the launch is over), [1]: slices done, [2]: pending
// pivots of the request (which row of prow holds rho), [3]: slice ticket.  Device-scope release / acquire around every hand-over: the
// workgroups of one LP sit on different XCDs, each with its own L2.
constexpr int REV_SLICE = 2048;                    // columns per slice (two per thread of a 1024-thread workgroup)
__device__ __forceinline__ int rev_nslices(const LpView &L) { return (L.ld + REV_SLICE - 1) / REV_SLICE; }
__device__ void rev_take_slices(const LpView &L, int *mb, const int n, const double *brow, const int *nh, double *row, unsigned long long *cnt = nullptr)
{
    __shared__ int s_t;
    const int ns = rev_nslices(L);
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            // a ticket of request n and of no other: a helper that comes back late from request n - 1 must not take (or use up) one of n's
            int v = __hip_atomic_load(&mb[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), t = -1;
            while ((v >> 16) == n && (v & 0xFFFF) < ns) {
                if (__hip_atomic_compare_exchange_strong(&mb[3], &v, v + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { t = v & 0xFFFF; break; }
            }
            s_t = t;
        }
        __syncthreads();
        const int t = s_t;
        if (t < 0) break;
        if (cnt && threadIdx.x == 0) *cnt += 1;
        rev_row_slice(L, brow, nh, row, t, ns);
        if (!(L.probe & 16)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // (the slice is out before it is counted)
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&mb[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ void rev_helper(const LpView &L, const BatchView &Bv, const int b)
{
    extern __shared__ unsigned char dyn_sel[];
    __shared__ int s_req;
    int *mb = Bv.hmail + (size_t)b * 8;
    const int tid = threadIdx.x, NT = (int)blockDim.x;
    const int slot = Bv.dst[b];
    const int *nh = L.nh + (size_t)slot * L.N;
    double *srho = L.rho_off >= 0 ? reinterpret_cast<double *>(dyn_sel + L.rho_off) : nullptr;
    int last = L.launch_id << 8;
    for (long spins = 0; spins < 20000000L; spins++) {          // (bounded: ~10 s; a helper that leaves is not missed, see above)
        if (tid == 0) s_req = __hip_atomic_load(&mb[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (relaxed: an acquire here invalidates the L2 of the XCD on every poll, for everyone on it)
        __syncthreads();
        const int req = s_req;
        __syncthreads();
        if ((req >> 8) != L.launch_id || req == last) { __builtin_amdgcn_s_sleep(8); continue; }
        if ((req & 0xFF) == 0xFF) return;
        last = req;
        if (!(L.probe & 32)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (once per request: rho, the pending count and the heads as the LP's workgroup left them)
        const int np = __hip_atomic_load(&mb[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double *brow_g = Bv.prow + (size_t)b * KP * L.ldt + (size_t)np * L.ldt;
        if (srho) { for (int c = tid; c < L.ldt; c += NT) srho[c] = brow_g[c]; __syncthreads(); }
        if (!(L.probe & 64)) rev_take_slices(L, mb, req & 0xFF, srho ? srho : brow_g, nh, Bv.trow + (size_t)b * L.ld);
    }
}
// Workgroup: NT threads; NT_BIG for rows of 1536 columns and more (S-degenerate: 2011, ex09: 36 939) -- an LP's selection is a chain
// of passes over N entries by ONE workgroup, and four times the threads shorten every pass.
// ONE selection of LP b by the calling workgroup (every `return` below is taken by the whole workgroup).  Returns false when the LP
// cannot select again before the next pass over its tableau (finished, waiting for a refresh, KP pivots pending).
template <bool EXT>
__device__ bool select_once(const LpView &L, const BatchView &Bv, const int b, const int cap2)
{
    __shared__ double sv[NT_BIG / WAVE];
    __shared__ int si[NT_BIG / WAVE];
    const int NT = (int)blockDim.x;
    __shared__ PivDesc s_d;
    __shared__ int s_cnt, s_nboxed, s_stop;
    extern __shared__ unsigned char dyn_sel[];
    double *skey = reinterpret_cast<double *>(dyn_sel);          // [cap2] breakpoints |d_j| / |alpha_j| of the candidates
    int *sidx = reinterpret_cast<int *>(skey + cap2);             // [cap2] their columns
    unsigned char *sflag = reinterpret_cast<unsigned char *>(sidx + cap2);   // [N] 1 = column switches bound in this iteration
    if (Bv.status[b] != ST_RUNNING || Bv.mode[b] == MODE_REFRESH) return false;
    const int np = Bv.npend[b];
    if (np >= KP) return false;                // waits for the pass over its tableau
    const int tid = threadIdx.x;
    int slot = Bv.dst[b];
    // the stored tableau of this solve: the parent's slot until the first pass has written its own (see k_init)
    const double *T0 = L.T + (size_t)(Bv.flushed[b] ? slot : Bv.src[b]) * L.slotT;
    double *beta = L.beta + (size_t)slot * L.Mp1p;
    double *xN = L.xN + (size_t)slot * L.ld;
    int *bh = L.bh + (size_t)slot * L.M, *nh = L.nh + (size_t)slot * L.N;
    int *nstat = L.nstat + (size_t)slot * L.N, *pos = L.pos + (size_t)slot * (L.M + L.N);
    const int M = L.M, N = L.N, ld = L.ld;
    const PivDesc *pd = Bv.desc + (size_t)b * KP;
    const int ldt = L.ldt;
    double *prow0 = Bv.prow + (size_t)b * KP * ldt;
    double *pcol0 = Bv.pcol + (size_t)b * KP * L.Mp1p;
    double *drow = Bv.dcur + (size_t)b * ld;
    double *dwork = drow;                      // the reduced costs the dual ratio test works with
    // the pivot row as it is after the pending pivots, where k_flush will read it (revised form: k_flush reads the row of B^-1, brow,
    // from there, and the tableau row lives in a scratch vector of the LP)
    double *row = L.rev ? Bv.trow + (size_t)b * ld : prow0 + (size_t)np * ld;
    double *pc = pcol0 + (size_t)np * L.Mp1p;  // the multipliers of the rows (primal selection: first the entering column itself)
    // row r / column q of the tableau as it is after the pending pivots, by the whole workgroup (barriers inside)
    auto fetch_row = [&](const int r) {
        if (!L.rev) {
            for (int j = tid; j < ld; j += NT) row[j] = j < N ? virt_entry(T0[(size_t)r * ld + j], r, j, np, pd, prow0, pcol0, ld, L.Mp1p) : 0.0;
        } else {
            double *brow_g = prow0 + (size_t)np * ldt;
            // rho also goes to LDS when it fits (ex09: 37 KB): the sparse products below gather from it ~200 000 times per selection
            double *srho = L.rho_off >= 0 ? reinterpret_cast<double *>(dyn_sel + L.rho_off) : nullptr;
            for (int c = tid; c < ldt; c += NT) {
                const double v = c < M ? ((L.probe & 4) ? T0[(size_t)r * ldt + c] : virt_entry_b(T0[(size_t)r * ldt + c], r, c, np, pd, prow0, pcol0, ldt, L.Mp1p)) : 0.0;
                brow_g[c] = v;
                if (srho) srho[c] = v;
            }
            __syncthreads();
            const double *brow = srho ? srho : brow_g;
            if (L.probe & 1) { for (int j = tid; j < ld; j += NT) row[j] = (j < N && nh[j] < M) ? -brow[nh[j]] : (j < N ? 1e-3 : 0.0); __syncthreads(); return; }
            unsigned long long tf = ((L.probe & 8) && b == 0) ? wall_clock64() : 0ull;
#define ROW_PHASE(k) do { if ((L.probe & 8) && b == 0) { __syncthreads(); if (tid == 0) { const unsigned long long tn = wall_clock64(); Bv.dbg[k] += tn - tf; tf = tn; } } } while (0)
            if (L.helpers > 1) {
                // hand the row out in slices (rev_take_slices): rho is in global memory (brow_g), the request goes out, this workgroup takes
                // slices like everyone else and then waits for the ones others took
                int *mb = Bv.hmail + (size_t)b * 8;
                __threadfence();
                __syncthreads();
                ROW_PHASE(8);
                __shared__ int s_n, s_late;
                if (tid == 0) {
                    const int old = __hip_atomic_load(&mb[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int n = ((old >> 8) == L.launch_id ? (old & 0xFF) : 0) + 1;      // (at most 2 * KP requests per launch: far from 0xFF)
                    s_n = n;
                    __hip_atomic_store(&mb[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&mb[2], np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&mb[3], n << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&mb[0], (L.launch_id << 8) | n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                ROW_PHASE(9);
                rev_take_slices(L, mb, s_n, brow, nh, row, ((L.probe & 8) && b == 0) ? &Bv.dbg[13] : nullptr);
                ROW_PHASE(10);
                if (tid == 0) {
                    const int ns = rev_nslices(L);
                    long w = 0;
                    while (__hip_atomic_load(&mb[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ns && ++w < 50000000L) __builtin_amdgcn_s_sleep(2);      // (every slice counted here was taken by a workgroup that is running)
                    s_late = w >= 50000000L;
                }
                __syncthreads();
                ROW_PHASE(11);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // (the slices the others wrote)
                ROW_PHASE(12);
                if (s_late) { rev_row_slice(L, brow, nh, row, 0, 1); __syncthreads(); }      // (never seen; the row is this workgroup's to deliver either way)
            } else rev_row_slice(L, brow, nh, row, 0, 1);
        }
        __syncthreads();
    };
    auto fetch_col = [&](const int q) {
        if (!L.rev) {
            for (int i = tid; i < M; i += NT) pc[i] = virt_entry(T0[(size_t)i * ld + q], i, q, np, pd, prow0, pcol0, ld, L.Mp1p);
        } else {
            const int kq = nh[q];
            if (L.probe & 2) { for (int i = tid; i < M; i += NT) pc[i] = i == 0 ? 1.0 : 1e-3; __syncthreads(); return; }
            const int cb = kq >= M ? L.cptr[kq - M] : 0, ce = kq >= M ? L.cptr[kq - M + 1] : 0;
            for (int i0 = tid; i0 < M; i0 += 2 * NT) {    // (B^-1 as stored) K_kq: two rows per thread, the gathers of eight non-zeros of each in flight together
                const int i1 = i0 + NT;
                const double *B0 = T0 + (size_t)i0 * ldt, *B1 = T0 + (size_t)(i1 < M ? i1 : i0) * ldt;
                double v0 = 0.0, v1 = 0.0;
                if (kq < M) { v0 = B0[kq]; v1 = B1[kq]; }
                else for (int t0 = cb; t0 < ce; t0 += 8) {
                    double g0[8], g1[8], cv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { const bool in = t0 + u < ce; const int c = in ? L.cidx[t0 + u] : 0; cv[u] = in ? L.cval[t0 + u] : 0.0; g0[u] = B0[c]; g1[u] = B1[c]; }
#pragma unroll
                    for (int u = 0; u < 8; u++) { v0 = fma(-cv[u], g0[u], v0); v1 = fma(-cv[u], g1[u], v1); }
                }
                pc[i0] = v0;
                if (i1 < M) pc[i1] = v1;
            }
            for (int sp = 0; sp < np; sp++) {              // ... through the pending pivots, in order
                __syncthreads();
                const PivDesc d = pd[sp];
                const double vr = pc[d.r];
                __syncthreads();
                for (int i = tid; i < M; i += NT) pc[i] = i == d.r ? -d.p * vr : fma(-pcol0[(size_t)sp * L.Mp1p + i], vr, pc[i]);
            }
            __syncthreads();
            for (int i = tid; i < M; i += NT) pc[i] = -pc[i];      // T[:, q] = -B^-1 K_kq
        }
        __syncthreads();
    };
    int pf = 0;
    if constexpr (EXT) {
        pf = Bv.pflags[b];
        double *dp = Bv.dper + (size_t)b * ld;
        if (pf & PF_PERT_PENDING) {
            for (int j = tid; j < ld; j += NT) {
                double v = j < N ? drow[j] : 0.0;
                if (j < N) {
                    const int st = nstat[j];
                    const double eps = 5e-7 * L.pert_scale * (1.0 + hash01(nh[j]));
                    if (st == NS_L) v = fmax(v, 0.0) + eps;
                    else if (st == NS_U) v = fmin(v, 0.0) - eps;
                }
                dp[j] = v;
            }
            __syncthreads();
            pf = (pf & ~PF_PERT_PENDING) | PF_PERT;
            if (tid == 0) { Bv.pflags[b] = pf; atomicAdd(&Bv.xstat[1], 1); }
        }
        if (pf & PF_PERT) dwork = dp;
    }
    // anti-cycling: after `bland_after` pivots (a healthy solve needs far fewer) switch to Bland's rule --
    // smallest variable id among the infeasible rows, exact minimum ratio with smallest id among ties
    // (and while a primal clean-up is in a streak of PRIMAL_STALL degenerate steps: ex09 in the revised form met clean-ups that cycled on a
    //  degenerate face for 166 000 pivots, until bland_after ended them within 500.  Bland's rule from the first clean-up step on is no
    //  answer -- 2 M pivots, it crawls while there is progress to be made -- so it holds only as long as the steps have length zero)
    const bool bland = (Bv.iters[b] >= L.bland_after || (EXT && (pf & PF_PRIMAL) && Bv.stall[b] >= PRIMAL_STALL)) && !(pf & PF_PERT);     // (perturbed costs break the ties themselves)
    int r = -1, q = -1, nflip = 0;
    bool incr = false;                       // bound switches of this iteration already carried into beta
    bool below = false, have_col = false;
    double pstep = 1.0;                      // length of a primal step (clean-up)

    if (EXT && (pf & PF_PRIMAL)) {
        // ---- primal simplex step on the true reduced costs (clean-up after a perturbation) ----
        ValIdx ent{0.0, -1};
        for (int j = tid; j < N; j += NT) {
            const int st = nstat[j];
            if (st == NS_S) continue;
            const double v = drow[j];
            double sc = 0.0;
            if (st == NS_L) { if (v < -TOL_DJ) sc = -v; }
            else if (st == NS_U) { if (v > TOL_DJ) sc = v; }
            else if (fabs(v) > TOL_DJ) sc = fabs(v);
            if (sc > 0.0) ent = better_max(ent, ValIdx{bland ? (double)(L.M + L.N - nh[j]) : sc, j});
        }
        ent = block_argmax(ent, sv, si);
        if (ent.i < 0) {
            // dual feasible: the dual selection takes over again (it concludes, or repairs what rounding left infeasible)
            if (tid == 0) { Bv.pflags[b] = pf & ~PF_PRIMAL; Bv.stall[b] = 0; }
            return true;
        }
        if (Bv.iters[b] >= L.maxit) {
            if (tid == 0) { Bv.status[b] = BSLV_LP_UNDEFINED; Bv.mode[b] = MODE_NONE; }
            return true;
        }
        q = ent.i;
        const int stq = nstat[q], kq = nh[q];
        const double dq = drow[q];
        const double dir = (stq == NS_U || (stq == NS_F && dq > 0.0)) ? -1.0 : 1.0;
        fetch_col(q);
        double cmax = 0.0;
        for (int i = tid; i < M; i += NT) cmax = fmax(cmax, fabs(pc[i]));
        cmax = block_max(cmax, sv);
        const double ptol = TOL_PIV * (1.0 + cmax);
        const double gap = UP(L, Bv, b, kq) - LO(L, Bv, b, kq);     // inf unless both bounds are finite
        double tmax = gap;
        for (int i = tid; i < M; i += NT) {
            const double a = pc[i] * dir;
            if (fabs(a) < ptol) continue;
            const int k = bh[i];
            const double bt = beta[i];
            if (a > 0) { const double up = UP(L, Bv, b, k); if (!isinf(up)) tmax = fmin(tmax, fmax(up + (bland ? 0.0 : btol(up)) - bt, 0.0) / a); }
            else { const double lo = LO(L, Bv, b, k); if (!isinf(lo)) tmax = fmin(tmax, fmax(bt - lo + (bland ? 0.0 : btol(lo)), 0.0) / -a); }
        }
        tmax = block_min(tmax, sv);
        if (isinf(tmax)) {
            if (tid == 0) { Bv.status[b] = BSLV_LP_UNBOUNDED; Bv.mode[b] = MODE_NONE; }
            return true;
        }
        ValIdx lv{0.0, -1};
        for (int i = tid; i < M; i += NT) {
            const double a = pc[i] * dir;
            if (fabs(a) < ptol) continue;
            const int k = bh[i];
            const double bt = beta[i];
            if (a > 0) { const double up = UP(L, Bv, b, k); if (!isinf(up) && (up - bt) / a <= tmax) lv = better_max(lv, ValIdx{bland ? (double)(L.M + L.N - k) : a, 2 * i + 1}); }
            else { const double lo = LO(L, Bv, b, k); if (!isinf(lo) && (bt - lo) / -a <= tmax) lv = better_max(lv, ValIdx{bland ? (double)(L.M + L.N - k) : -a, 2 * i}); }
        }
        lv = block_argmax(lv, sv, si);
        double tstep = INFINITY;
        if (lv.i >= 0) {
            const int i = lv.i >> 1, k = bh[i];
            const double a = pc[i] * dir;
            tstep = fmax(((lv.i & 1) ? UP(L, Bv, b, k) - beta[i] : beta[i] - LO(L, Bv, b, k)) / fabs(a), 0.0);
        }
        if (lv.i < 0 || gap <= tstep) {
            // the entering variable reaches its own other bound first: no pivot
            __syncthreads();
            for (int i = tid; i < M; i += NT) beta[i] = fma(pc[i], dir * gap, beta[i]);
            if (tid == 0) {
                beta[M] = fma(dq, dir * gap, beta[M]);
                if (stq == NS_L) { nstat[q] = NS_U; xN[q] = UP(L, Bv, b, kq); } else { nstat[q] = NS_L; xN[q] = LO(L, Bv, b, kq); }
                if (L.trace == b) printf("lp %d it %d primal: column %d (var %d) d %.3e switches bound\n", b, Bv.iters[b], q, kq, dq);
                Bv.verified[b] &= 2;
                Bv.iters[b] += 1;
                atomicAdd(&Bv.xstat[2], 1);
            }
            return true;
        }
        r = lv.i >> 1;
        below = !(lv.i & 1);                   // the leaving variable goes to its lower bound
        pstep = fabs(dq) * tstep / (1.0 + fabs(beta[M]));      // (what the step moves the objective by, relative: the ratio test's tolerance gives a degenerate step a length of 1e-9, not 0)
        fetch_row(r);
        have_col = true;
    } else {
    // ---- dual simplex step ----
    unsigned long long tk = (L.probe & 8) ? wall_clock64() : 0ull;
#define SEL_PHASE(k) do { if ((L.probe & 8) && b == 0) { __syncthreads(); if (tid == 0) { const unsigned long long tn = wall_clock64(); Bv.dbg[k] += tn - tk; tk = tn; } } } while (0)
    // Phase A: leaving row = largest bound violation; id = 2*i + (below ? 1 : 0)
    ValIdx best{0.0, -1};
    for (int i = tid; i < M; i += NT) {
        int k = bh[i];
        double lo = LO(L, Bv, b, k), up = UP(L, Bv, b, k), bt = beta[i];
        if (!isinf(lo)) { double v = lo - bt; if (v > btol(lo)) best = better_max(best, ValIdx{bland ? (double)(L.M + L.N - k) : v, 2 * i + 1}); }
        if (!isinf(up)) { double v = bt - up; if (v > btol(up)) best = better_max(best, ValIdx{bland ? (double)(L.M + L.N - k) : v, 2 * i}); }
    }
    best = block_argmax(best, sv, si);
    if (best.i < 0) {
        if (EXT && (pf & PF_PERT)) {
            // the perturbed problem is solved: take the perturbation away and look at the true reduced costs
            int wrong1 = 0, wrongb = 0;
            for (int j = tid; j < N; j += NT) {
                const int st = nstat[j];
                const double v = drow[j];
                if ((st == NS_L && v < -TOL_DJ) || (st == NS_U && v > TOL_DJ) || (st == NS_F && fabs(v) > TOL_DJ)) {
                    const int k = nh[j];
                    if (st != NS_F && !isinf(LO(L, Bv, b, k)) && !isinf(UP(L, Bv, b, k)) && !L.art[k]) wrongb = 1; else wrong1 = 1;
                }
            }
            wrong1 = __syncthreads_or(wrong1);
            wrongb = __syncthreads_or(wrongb);
            pf &= ~PF_PERT;
            if (tid == 0 && (wrong1 || wrongb)) atomicAdd(&Bv.xstat[3], 1);
            if (wrong1) {
                if (tid == 0) { Bv.pflags[b] = pf | PF_PRIMAL; Bv.stall[b] = 0; if (L.trace == b) printf("lp %d it %d perturbation off -> primal clean-up\n", b, Bv.iters[b]); }
                return true;
            }
            if (tid == 0) Bv.pflags[b] = pf;
            if (wrongb) {
                for (int j = tid; j < N; j += NT) {
                    const int st = nstat[j], k = nh[j];
                    const double v = drow[j];
                    if (st == NS_L && v < -TOL_DJ) { nstat[j] = NS_U; xN[j] = UP(L, Bv, b, k); }
                    else if (st == NS_U && v > TOL_DJ) { nstat[j] = NS_L; xN[j] = LO(L, Bv, b, k); }
                }
                if (tid == 0) { Bv.mode[b] = MODE_REFRESH; Bv.verified[b] &= 2; if (L.trace == b) printf("lp %d it %d perturbation off -> bound switches\n", b, Bv.iters[b]); }
                return true;
            }
            dwork = drow;
        }
        // recompute beta from scratch before concluding -- unless this solve made only a few pivots since k_init computed
        // it from scratch: the rank-1 updates of beta then carry ~1e-15 of error against tolerances of 1e-9, and the
        // refresh is a full read of the tableau (4 MB per LP on S-mid)
        // (Not when the solve started on an artificial bound: values of 1e7 leave rounding debris of 1e-9 in beta.)
        if (!(Bv.verified[b] & 1) && (Bv.iters[b] > REFRESH_AFTER || (Bv.verified[b] & 2))) {
            if (tid == 0) Bv.mode[b] = MODE_REFRESH;          // k_flush applies what is pending and recomputes beta
            return true;
        }
        // optimal for the bounded problem; unbounded if an artificial bound is active
        double flag = 0.0;
        for (int j = tid; j < N; j += NT) {
            int st = nstat[j];
            unsigned char a = L.art[nh[j]];
            if (((st == NS_L && (a & 1)) || (st == NS_U && (a & 2))) && fabs(drow[j]) > TOL_DJ) flag = 1.0;
        }
        flag = block_max(flag, sv);
        if (tid == 0) { Bv.status[b] = flag > 0.0 ? BSLV_LP_UNBOUNDED : BSLV_LP_OPTIMAL; Bv.mode[b] = MODE_NONE; }
        return true;
    }
    if (Bv.iters[b] >= L.maxit) {
        if (tid == 0) { Bv.status[b] = BSLV_LP_UNDEFINED; Bv.mode[b] = MODE_NONE; }
        return true;
    }
    r = best.i >> 1;
    below = best.i & 1;
    const double sgn = below ? 1.0 : -1.0;
    SEL_PHASE(0);
    fetch_row(r);
    SEL_PHASE(1);

    // pass 0: row scale for the relative pivot tolerance
    double rmax = 0.0;
    for (int j = tid; j < N; j += NT) rmax = fmax(rmax, fabs(row[j]));
    rmax = block_max(rmax, sv);
    const double ptol = TOL_PIV * (1.0 + rmax);
    if constexpr (EXT) {
        // candidates with their breakpoints; anything to flip at all?
        if (tid == 0) { s_cnt = 0; s_nboxed = 0; s_stop = 0; }
        for (int j = tid; j < N; j += NT) sflag[j] = 0;
        __syncthreads();
        if (cap2 > 0) for (int j = tid; j < N; j += NT) {
            int st = nstat[j];
            if (st == NS_S) continue;
            double a = sgn * row[j];
            if (fabs(a) < ptol) continue;
            if ((a > 0 && (st == NS_L || st == NS_F)) || (a < 0 && (st == NS_U || st == NS_F))) {
                const int at = atomicAdd(&s_cnt, 1);
                skey[at] = fabs(dwork[j]) / fabs(a);
                sidx[at] = j;
                const int k = nh[j];
                const double lo = LO(L, Bv, b, k), up = UP(L, Bv, b, k);
                if (st != NS_F && !isinf(lo) && !isinf(up) && !L.art[k]) atomicAdd(&s_nboxed, 1);
            }
        }
        __syncthreads();
        const int C = s_cnt;
        if (s_nboxed > 0) {
            int n2 = 2;
            while (n2 < C) n2 <<= 1;
            for (int i = C + tid; i < n2; i += NT) { skey[i] = INFINITY; sidx[i] = 0x7fffffff; }
            __syncthreads();
            // bitonic sort by (breakpoint, column): the order, and with it every decision below, is unique
            for (int kk = 2; kk <= n2; kk <<= 1)
                for (int jj = kk >> 1; jj > 0; jj >>= 1) {
                    for (int i = tid; i < n2; i += NT) {
                        const int x = i ^ jj;
                        if (x > i) {
                            const double ka = skey[i], kb2 = skey[x];
                            const int ia = sidx[i], ib = sidx[x];
                            const bool gt = ka > kb2 || (ka == kb2 && ia > ib);
                            if (((i & kk) == 0) == gt) { skey[i] = kb2; skey[x] = ka; sidx[i] = ib; sidx[x] = ia; }
                        }
                    }
                    __syncthreads();
                }
            if (tid == 0) {
                // walk the breakpoints: a boxed candidate switches bound while the row stays infeasible after its switch
                const int kb0 = bh[r];
                double slope = below ? LO(L, Bv, b, kb0) - beta[r] : beta[r] - UP(L, Bv, b, kb0);
                int k = 0;
                for (; k < C; k++) {
                    const int j = sidx[k], kv = nh[j];
                    const double lo = LO(L, Bv, b, kv), up = UP(L, Bv, b, kv);
                    if (nstat[j] == NS_F || isinf(lo) || isinf(up) || L.art[kv]) break;
                    const double dec = (up - lo) * fabs(row[j]);
                    if (slope - dec < 0.0) break;
                    slope -= dec;
                    sflag[j] = 1;
                }
                s_stop = k;
            }
            __syncthreads();
            nflip = s_stop;
        }
    }
    // pass 1: Harris bound on the dual step
    SEL_PHASE(2);
    // (both passes fetch FOUR columns per thread and iteration with all loads issued first: on wide problems -- ex09: 37 000 columns,
    // 36 per thread -- a column after the other, its reduced cost loaded behind two branches, was a chain of ~100 memory latencies;
    // min and the (value, index) arg-max do not depend on the order)
    constexpr int PU = 4;
    double th = INFINITY;
    for (int j0 = tid; j0 < N; j0 += PU * NT) {
        int stv[PU]; double av[PU], dv[PU]; bool skip[PU];
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const int j = j0 + u * NT;
            const bool in = j < N;
            stv[u] = in ? nstat[j] : NS_S; av[u] = in ? sgn * row[j] : 0.0; dv[u] = in ? dwork[j] : 0.0;
            skip[u] = !in || (EXT && sflag[in ? j : 0]);
        }
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const int st = stv[u];
            const double a = av[u];
            if (skip[u] || st == NS_S || fabs(a) < ptol) continue;
            if ((a > 0 && (st == NS_L || st == NS_F)) || (a < 0 && (st == NS_U || st == NS_F)))
                th = fmin(th, (fabs(dv[u]) + (bland ? 0.0 : TOL_DJ)) / fabs(a));
        }
    }
    th = block_min(th, sv);
    SEL_PHASE(3);
    if (isinf(th)) {                      // no entering candidate: primal infeasible ...
        if (!(Bv.verified[b] & 1)) {      // ... unless the violation is rounding debris in beta: recompute it first
            if (tid == 0) Bv.mode[b] = MODE_REFRESH;
            return true;
        }
        if (tid == 0) { Bv.status[b] = BSLV_LP_INFEASIBLE; Bv.mode[b] = MODE_NONE; }
        return true;
    }
    // pass 2: largest |pivot| within the bound
    ValIdx piv{0.0, -1};
    for (int j0 = tid; j0 < N; j0 += PU * NT) {
        int stv[PU]; double av[PU], dv[PU]; bool skip[PU];
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const int j = j0 + u * NT;
            const bool in = j < N;
            stv[u] = in ? nstat[j] : NS_S; av[u] = in ? sgn * row[j] : 0.0; dv[u] = in ? dwork[j] : 0.0;
            skip[u] = !in || (EXT && sflag[in ? j : 0]);
        }
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const int st = stv[u], j = j0 + u * NT;
            const double a = av[u];
            if (skip[u] || st == NS_S || fabs(a) < ptol) continue;
            if ((a > 0 && (st == NS_L || st == NS_F)) || (a < 0 && (st == NS_U || st == NS_F)))
                if (fabs(dv[u]) / fabs(a) <= th) piv = better_max(piv, ValIdx{bland ? (double)(L.M + L.N - nh[j]) : fabs(a), j});
        }
    }
    piv = block_argmax(piv, sv, si);
    SEL_PHASE(4);
    q = piv.i;
    if constexpr (EXT) {
        // the switches: other bound, other status.  beta no longer matches xN: this pivot is applied at once and beta
        // recomputed from the new tableau (MODE_REFRESH below), before the LP selects again
        for (int k = tid; k < nflip; k += NT) {
            const int j = sidx[k], kv = nh[j];
            const double lo = LO(L, Bv, b, kv), up = UP(L, Bv, b, kv);
            if (nstat[j] == NS_L) { nstat[j] = NS_U; xN[j] = up; skey[k] = up - lo; }
            else { nstat[j] = NS_L; xN[j] = lo; skey[k] = lo - up; }
        }
        // A few switches: beta follows them as a vector update, beta_i += sum_k T_i,j(k) * delta_k over the columns j(k) of the
        // tableau as it is after the pending pivots (row M = the reduced costs, row r = the row at hand) -- nflip strided column
        // reads instead of a pass over the whole tableau, and the LP keeps selecting (up to KP pivots per pass as without
        // switches).  Many switches (a cold start walks hundreds of breakpoints): the pass with MODE_REFRESH as before.
        // beta is recomputed from the tableau before any status is reported (verified), so the update cannot end in a result.
        if (nflip > 0 && nflip <= FLIP_INCR_MAX && !L.rev) {      // (revised form: a column of the tableau is a product, not a gather -- the refresh pass)
            __syncthreads();
            for (int i = tid; i <= M; i += NT) {
                double acc = 0.0;
                for (int k = 0; k < nflip; k++) {
                    const int j = sidx[k];
                    const double t = i == M ? drow[j] : (i == r ? row[j] : virt_entry(T0[(size_t)i * ld + j], i, j, np, pd, prow0, pcol0, ld, L.Mp1p));
                    acc = fma(t, skey[k], acc);
                }
                beta[i] += acc;
            }
            incr = true;
            __syncthreads();
        }
    }
    }
    bool col_ready = have_col;                                       // pc[] holds the entering column (primal steps fetch it first)
    unsigned long long tk2 = (L.probe & 8) ? wall_clock64() : 0ull;
    if (L.rev && !col_ready) { fetch_col(q); col_ready = true; }    // (the tableau form gathers its column in Phase D)
    if ((L.probe & 8) && b == 0) { __syncthreads(); if (tid == 0) { const unsigned long long tn = wall_clock64(); Bv.dbg[5] += tn - tk2; tk2 = tn; } }
    // Phase C: the descriptor, the basis heads
    if (tid == 0) {
        int kb = bh[r], kn = nh[q];
        double lo = LO(L, Bv, b, kb), up = UP(L, Bv, b, kb);
        double target = below ? lo : up;
        double trq = row[q];
        double br = beta[r];
        // revised form: the pivot element comes out of TWO products -- rho K_q (the row) and the column B^-1 K_q -- which agree as long as
        // B^-1 is accurate.  Nothing refactorises it; when the two drift apart the LP is given up as UNDEFINED and the caller's retry
        // (bslv_lp.c:222-227: from the standard basis, an exact identity) takes over instead of a solve on a corrupted inverse
        if (L.rev && !(fabs(trq - pc[r]) <= 1e-8 * (1.0 + fabs(trq)))) {
            Bv.status[b] = BSLV_LP_UNDEFINED; Bv.mode[b] = MODE_NONE;
            if (L.trace == b) printf("lp %d it %d: pivot element from the row %.17g, from the column %.17g: B^-1 has drifted\n", b, Bv.iters[b], trq, pc[r]);
            s_d.r = -1;
        } else {
        PivDesc d;
        d.r = r; d.q = q; d.p = 1.0 / trq;
        d.pbeta = br - target;
        d.enter_val = xN[q] + (target - br) / trq;
        Bv.desc[(size_t)b * KP + np] = d;
        s_d = d;
        bh[r] = kn; nh[q] = kb;
        pos[kn] = r; pos[kb] = -1 - q;
        if (lo == up) { nstat[q] = NS_S; xN[q] = lo; }
        else if (below) { nstat[q] = NS_L; xN[q] = lo; }
        else { nstat[q] = NS_U; xN[q] = up; }
        if (L.trace == b && (Bv.iters[b] < 300 || Bv.iters[b] % 997 == 0)) printf("lp %d it %d%s r %d (var %d, %s by %.3e) q %d (var %d) alpha %.3e d %.3e step %.3e flips %d obj %.12g%s%s\n", b, Bv.iters[b], have_col ? " primal" : "", r, kb, below ? "below" : "above",
                                 below ? lo - br : br - up, q, kn, trq, dwork[q], fabs(dwork[q] / trq), nflip, beta[M], bland ? " bland" : "", (pf & PF_PERT) ? " perturbed" : "");
        if constexpr (EXT) {
            if (nflip > 0) atomicAdd(&Bv.xstat[0], 1);
            if (incr) atomicAdd(&Bv.xstat[4], 1);
            if (have_col) { atomicAdd(&Bv.xstat[2], 1); Bv.stall[b] = pstep <= 1e-7 ? Bv.stall[b] + 1 : 0; }      // (streak of degenerate steps of this primal clean-up)
            if (!have_col) {
                // dual degenerate stalling: perturb the costs from the next selection on
                int stl = fabs(dwork[q] / trq) <= 1e-11 ? Bv.stall[b] + 1 : 0;
                if (stl >= L.stall_limit && !(pf & PF_PERT) && (pf >> PF_USES_SHIFT) < PERT_MAX_USES) {
                    Bv.pflags[b] = (pf | PF_PERT_PENDING) + (1 << PF_USES_SHIFT);
                    stl = 0;
                }
                Bv.stall[b] = stl;
            }
        }
        Bv.mode[b] = (nflip > 0 && !incr) ? MODE_REFRESH : MODE_PIVOT;
        Bv.verified[b] &= 2;
        Bv.iters[b] += 1;
        }
    }
    __syncthreads();
    const PivDesc d = s_d;
    if (d.r < 0) return true;                      // (given up: see above)
    // Phase D: the entering column as it is after the pending pivots -> multipliers of all rows, beta; the reduced-cost row
    for (int i = tid; i <= M; i += NT) {
        if (i == r) { pc[i] = 0.0; beta[i] = d.enter_val; continue; }
        const double f = (i == M ? drow[q] : (col_ready ? pc[i] : virt_entry(T0[(size_t)i * ld + q], i, q, np, pd, prow0, pcol0, ld, L.Mp1p))) * d.p;
        pc[i] = f;
        beta[i] = fma(-f, d.pbeta, beta[i]);
    }
    __syncthreads();
    {
        const double fM = pc[M];
        for (int j = tid; j < N; j += NT) drow[j] = j == q ? fM : fma(-fM, row[j], drow[j]);
        if constexpr (EXT) {
            if (pf & PF_PERT) {
                double *dp = Bv.dper + (size_t)b * ld;
                const double fP = dp[q] * d.p;
                __syncthreads();
                for (int j = tid; j < N; j += NT) dp[j] = j == q ? fP : fma(-fP, row[j], dp[j]);
            }
        }
    }
    if (tid == 0) Bv.npend[b] = np + 1;
    if ((L.probe & 8) && b == 0) { __syncthreads(); if (tid == 0) { Bv.dbg[6] += wall_clock64() - tk2; Bv.dbg[7] += 1; } }
    return true;
}
// One launch selects up to nsel pivots per LP, one after the other, by the same workgroup: the KP selections between two passes
// over the tableau depend only on the LP's own vectors (beta, the reduced-cost row, the pending pivot rows and multipliers) -- no
// grid-wide dependency asks for a launch each (rounds 1-3 launched this kernel KP times per pass, 17 % of all GPU time in
// launches of 25 us; BSLV_SELECT_FUSE=0 brings that form back: the results are the same bit for bit, tests/test_lp_gpu.py).
template <bool EXT>
__global__ __launch_bounds__(NT_BIG) void k_select(LpView L, BatchView Bv, const int *active, int nact, int cap2, int nsel)
{
    if ((int)blockIdx.x >= nact) return;
    const int b = active[blockIdx.x];          // compacted list of the LPs still running
    if (blockIdx.y > 0) { rev_helper(L, Bv, b); return; }      // (revised form: helps with the sparse products of LP b's tableau rows until told to leave)
    for (int sdx = 0; sdx < nsel; sdx++) {
        if (sdx) __syncthreads();              // (what thread 0 / every thread wrote for the LP -- status, mode, pending count, beta, reduced costs -- is read by all)
        if (!select_once<EXT>(L, Bv, b, cap2)) break;
    }
    if (L.helpers > 1) {                       // every path of the LP's own workgroup ends here: the helpers may go
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&Bv.hmail[(size_t)b * 8], (L.launch_id << 8) | 0xFF, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}


// ---- which LPs need a pass over their tableau: pending pivots to apply, or beta to recompute ----
__global__ void k_list_pending(BatchView Bv, const int *active, int nact, int it)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nact) return;
    const int b = active[k];
    if (Bv.npend[b] > 0 || Bv.mode[b] == MODE_REFRESH) {
        // lazy: a pass over the tableau is for LPs that go on pivoting (or work in place); one that is finished keeps its <= KP pending
        // pivots -- values, duals and objective are all in its vectors -- and most such tableaux are never looked at again
        if (Bv.lazy && Bv.status[b] != ST_RUNNING && Bv.src[b] != Bv.dst[b] && Bv.mode[b] != MODE_REFRESH) return;
        Bv.work[atomicAdd(&Bv.nwork[it], 1)] = b;
    }
}
// lazy: the LPs list[0..n) (batch indices) whose slot still lacks its tableau -> work list of counter slot cnt_slot
__global__ void k_list_given(BatchView Bv, const int *list, int n, int cnt_slot)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int b = list[k];
    if (Bv.npend[b] > 0 || !Bv.flushed[b]) Bv.work[atomicAdd(&Bv.nwork[cnt_slot], 1)] = b;
}
// lazy: the reduced costs of every LP go to row M of its slot (k_flush would have left them there; the getters read them from there)
__global__ void k_store_d(LpView L, BatchView Bv, int B)
{
    const int b = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && j < L.ld) L.T[(size_t)Bv.dst[b] * L.slotT + (size_t)L.M * L.ld + j] = Bv.dcur[(size_t)b * L.ld + j];
}
__global__ void k_after_flush(BatchView Bv, int it)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Bv.nwork[it]) return;
    const int b = Bv.work[k];
    Bv.npend[b] = 0;
    Bv.flushed[b] = 1;
    if (Bv.mode[b] == MODE_REFRESH) { Bv.verified[b] |= 1; Bv.mode[b] = MODE_NONE; }
}

// ---- k_flush: the HBM-bound kernel.  Persistent grid over (LP of the work list x row tile).  Every row of the stored
//      tableau (the parent's slot on the first pass of a solve) is read once, the pending pivots are applied to it in order
//        row r_s:  T[r][j] = -prow_s[j] * p_s (j != q_s),  T[r][q_s] = p_s
//        others:   T[i][j] -= f_si * prow_s[j] (j != q_s), T[i][q_s] = f_si          (f_si from k_select)
//      and the row is written to the LP's own slot; with MODE_REFRESH beta_i = T_i . xN is recomputed from the finished row.
//      Algorithmic traffic of one pass: one read + one write of the tableau, whatever the number of pending pivots. ----
//      wide != 0 (rows of more than ~3000 columns: KP pivot rows do not fit in LDS): the pivot rows are read from global
//      memory instead -- every row tile of an LP reads the same KP rows, which the L2 / MALL serve after the first tile. ----
//      Workgroup size: NT threads, or NT_BIG where the KP pivot rows take so much LDS that fewer than 3 workgroups fit on a CU
//      (rows of more than ~850 columns): with 4 waves per CU the pass is latency-bound (S-degenerate, 2011 columns: 1.8 TB/s);
//      16 waves share one copy of the rows instead.
template <bool WIDE>
__global__ __launch_bounds__(NT_BIG) void k_flush(LpView L, BatchView Bv, int it, int tiles, int tr /* rows per work item: 8 .. 128 */)
{
    const int NT = (int)blockDim.x;
    extern __shared__ double s_rows[];           // KP pivot rows
    __shared__ PivDesc s_pd[KP];
    const int nitems = Bv.nwork[it] * tiles;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ld = L.ldt, ld2 = ld >> 1;         // (row length of the slot matrix: the tableau, or B^-1 in the revised form)
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int b = Bv.work[item / tiles], tile = item % tiles;
        const int np = Bv.npend[b];
        const bool refresh = Bv.mode[b] == MODE_REFRESH;
        const int slot = Bv.dst[b];
        const double *Tin = L.T + (size_t)(Bv.flushed[b] ? slot : Bv.src[b]) * L.slotT;
        double *T = L.T + (size_t)slot * L.slotT;
        double *beta = L.beta + (size_t)slot * L.Mp1p;
        const double *pcol0 = Bv.pcol + (size_t)b * KP * L.Mp1p;
        const double *rowsg = Bv.prow + (size_t)b * KP * ld;
        {
            const double2 *g = reinterpret_cast<const double2 *>(Bv.prow + (size_t)b * KP * ld);
            double2 *s2 = reinterpret_cast<double2 *>(s_rows);
            if (!WIDE) for (int j2 = threadIdx.x; j2 < np * ld2; j2 += NT) s2[j2] = g[j2];
            if (threadIdx.x < np) s_pd[threadIdx.x] = Bv.desc[(size_t)b * KP + threadIdx.x];
        }
        __syncthreads();
        const double2 *x2 = reinterpret_cast<const double2 *>(L.rev ? Bv.uvec + (size_t)b * ld : L.xN + (size_t)slot * ld);      // (rev: beta = B^-1 uvec, k_rev_u)
        const bool same = Tin == T;
        for (int rr = wave; rr < tr; rr += NT / WAVE) {
            const int i = tile * tr + rr;
            if (i >= L.mrows) break;
            double f[KP];
            bool isr[KP], any = false;
#pragma unroll
            for (int s = 0; s < KP; s++) {
                isr[s] = s < np && i == s_pd[s].r;
                f[s] = (s < np && !isr[s]) ? pcol0[(size_t)s * L.Mp1p + i] : 0.0;
                any |= isr[s] || f[s] != 0.0;
            }
            if (!any && same && !refresh) continue;               // row untouched by the pending pivots and already in place
            const double2 *t_in = reinterpret_cast<const double2 *>(Tin + (size_t)i * ld);
            double2 *t_out = reinterpret_cast<double2 *>(T + (size_t)i * ld);
            double acc = 0.0;
            for (int j2 = lane; j2 < ld2; j2 += WAVE) {
                double2 v = t_in[j2];
#pragma unroll
                for (int s = 0; s < KP; s++) {
                    if (s >= np) break;
                    const double2 pr = WIDE ? reinterpret_cast<const double2 *>(rowsg + (size_t)s * ld)[j2] : reinterpret_cast<const double2 *>(s_rows + (size_t)s * ld)[j2];
                    const int q2 = L.rev ? -1 : s_pd[s].q >> 1, qodd = s_pd[s].q & 1;       // (B^-1 has no column swap)
                    if (isr[s]) { const double p = s_pd[s].p; v.x = -pr.x * p; v.y = -pr.y * p; if (j2 == q2) { if (qodd) v.y = p; else v.x = p; } }
                    else { const double fs = f[s]; v.x = fma(-fs, pr.x, v.x); v.y = fma(-fs, pr.y, v.y); if (j2 == q2) { if (qodd) v.y = fs; else v.x = fs; } }
                }
                t_out[j2] = v;
                if (refresh) { const double2 x = x2[j2]; acc = fma(v.x, x.x, acc); acc = fma(v.y, x.y, acc); }
            }
            if (refresh) { acc = wave_sum(acc); if (lane == 0) beta[i] = acc; }
        }
        __syncthreads();          // the pivot rows in LDS are reused by the next work item
    }
}

// ---- solves that ended without a pivot never left the parent's slot (see k_init): give them their own copy ----
__global__ void k_list_unpivoted(BatchView Bv, int B, int slot_of_count)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && (slot_of_count < 0 || !Bv.flushed[b]) && Bv.src[b] != Bv.dst[b]) Bv.work[atomicAdd(&Bv.nwork[slot_of_count < 0 ? -slot_of_count : slot_of_count], 1)] = b;
}
__global__ __launch_bounds__(NT) void k_copy_unpivoted(LpView L, BatchView Bv, int slot_of_count, int tiles)
{
    const int nitems = Bv.nwork[slot_of_count] * tiles;
    const int ld2 = L.ldt >> 1;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int b = Bv.work[item / tiles], tile = item % tiles;
        const double2 *s = reinterpret_cast<const double2 *>(L.T + (size_t)Bv.src[b] * L.slotT);
        double2 *d = reinterpret_cast<double2 *>(L.T + (size_t)Bv.dst[b] * L.slotT);
        const int i0 = tile * TR, i1 = min(i0 + TR, L.M);            // row M is already there
        for (size_t k = (size_t)i0 * ld2 + threadIdx.x; k < (size_t)i1 * ld2; k += NT) d[k] = s[k];
    }
}

// ---- getters ----
__global__ void k_get(LpView L, const int *slots, int B, int first, int cnt, int what, double *out)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * cnt) return;
    int b = idx / cnt, k = first + idx % cnt;
    int slot = slots[b];
    int p = L.pos[(size_t)slot * (L.M + L.N) + k];
    double v;
    if (what == 0) v = p >= 0 ? L.beta[(size_t)slot * L.Mp1p + p] : L.xN[(size_t)slot * L.ld + (-1 - p)];
    else v = p >= 0 ? 0.0 : (L.rev ? L.dsl[(size_t)slot * L.ld + (-1 - p)] : L.T[(size_t)slot * L.slotT + (size_t)L.M * L.ld + (-1 - p)]);
    out[idx] = v;
}
__global__ void k_get_obj(LpView L, const int *slots, int B, double c0, double *out)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    out[b] = L.beta[(size_t)slots[b] * L.Mp1p + L.M] + c0;
}
__global__ void k_std_heads(LpView L, int slot)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L.M) { L.bh[(size_t)slot * L.M + i] = i; L.pos[(size_t)slot * (L.M + L.N) + i] = i; }
    if (i < L.N) {
        L.nh[(size_t)slot * L.N + i] = L.M + i;
        L.pos[(size_t)slot * (L.M + L.N) + L.M + i] = -1 - i;
        L.nstat[(size_t)slot * L.N + i] = NS_L;
    }
    if (i < L.ld) L.xN[(size_t)slot * L.ld + i] = 0.0;
    if (i < L.Mp1p) L.beta[(size_t)slot * L.Mp1p + i] = 0.0;
}

}  // namespace bslv

using namespace bslv;

struct bslv_lpq {
    LpView L{};
    int slots = 0;
    double c0 = 0;
    hipStream_t stream = nullptr;
    double *Tstd = nullptr;           // (M+1) x ld image of [A ; cost] (tableau form)
    // revised form: A once as CSC and CSR, the cost vector, per-slot reduced costs, per-LP scratch
    int *cptr_d = nullptr, *cidx_d = nullptr, *rptr_d = nullptr, *ridx_d = nullptr; double *cval_d = nullptr, *rval_d = nullptr, *cost_d = nullptr, *dsl_d = nullptr;
    unsigned long long *dbg_d = nullptr;
    // LAZY tableaux (bslv_lpq_set_lazy; the Benson driver's mode): see bslv_lpq_materialise
    bool lazy = false, lazy_open = false;          // lazy_open: the last batch left slots without their tableau
    std::vector<int> last_dst;                     // dst slots of the last batch (host copy)
    int *list_d = nullptr; int listcap = 0;
    long lazy_skipped = 0, lazy_materialised = 0;  // LPs whose pass was skipped / asked for afterwards (totals)
    double lazy_ms = 0;                            // host wall clock spent in bslv_lpq_materialise (total)
    double *trow_d = nullptr, *uvec_d = nullptr, *xfull_d = nullptr;
    int *hmail_d = nullptr; int launch_seq = 0;      // revised form: mailboxes of k_select's helper workgroups; launches so far
    long nnzA = 0;
    double *lb_d = nullptr, *ub_d = nullptr;
    unsigned char *art_d = nullptr;
    std::vector<double> cost;         // N+1
    // batch buffers
    int Bcap = 0;
    int *qslot_d = nullptr;           // slots of a getter call (batch-sized, like src_d)
    int *src_d = nullptr, *dst_d = nullptr, *status_d = nullptr, *iters_d = nullptr, *mode_d = nullptr, *ver_d = nullptr;
    int *work_d = nullptr, *nwork_d = nullptr; int nworkcap = 0;
    int *npend_d = nullptr, *flushed_d = nullptr; double *pcol_d = nullptr, *dcur_d = nullptr;     // delayed update (see BatchView)
    double *dper_d = nullptr; int *pflags_d = nullptr, *stall_d = nullptr, *xstat_d = nullptr;
    double *cvals_d = nullptr; size_t cvals_cap = 0;     // objective coefficients of solve_batch_obj
    long last_ext[5] = {0, 0, 0, 0, 0};
    long last_passes = 0;              // (LP, pass) pairs of the last batch: how many tableaux k_flush read and wrote
    long last_launches = 0;            // k_flush launches of the last batch: one per lock-step round + one per pass made on request (bslv_lpq_materialise)
    size_t select_lds_max = 64 * 1024; // dynamic LDS of k_select<true> (candidate sort of the bound flipping ratio test)
    size_t select0_lds_max = 64 * 1024; // ... of k_select<false> (revised form: rho)
    bool has_boxed = false;            // some variable outside the per-LP range has two finite, non-artificial bounds
    bool force_ext = false;            // extended selection (perturbation, primal clean-up) also without a boxed variable: bslv_lpq_set_extended,
                                       // and by itself for tableaux of 1 GiB and more (every pivot costs a millisecond there: no stalling)
    size_t flush_lds_max = 64 * 1024;  // dynamic LDS k_flush may use (raised to 144 KB at create when the runtime allows)
    int upd_grid = 32768;             // workgroups of the persistent k_flush (BSLV_UPD_GRID; 1024..32768 measured equal within 2 %)
    int *active_d = nullptr, *active_h = nullptr;       // compacted indices of the LPs still running (device / pinned)
    double *vlo_d = nullptr, *vup_d = nullptr, *prow_d = nullptr, *out_d = nullptr;
    size_t out_cap = 0;
    PivDesc *desc_d = nullptr;
    int *status_h = nullptr;          // pinned
    // stats
    int last_iters = 0;
    long last_pivots = 0;
    double last_update_ms = 0, last_total_ms = 0;
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evpool;
    // PRESOLVE at the boundary (round 3): a row of A with a single non-zero outside the per-LP range is a bound on that column
    // (the hypercube rows `d 0 1` of ex/example10.m:21-24 and of S-degenerate); it is folded into the column's bounds and leaves the
    // tableau.  The callers keep the model they handed over (GLPK's, bslv_lp.c:60-70,219-308): every index that crosses the
    // boundary is an index of THAT model, the primal value of a folded row is a_ij x_j, and its dual is the column's reduced cost
    // over a_ij whenever the bound the column sits on is the row's and not its own.
    struct Presolve {
        int M0 = 0, N0 = 0, nfold = 0;                 // the model as given; rows folded
        bool empty_box = false;                         // bounds set later made the box of a folded row's column empty: every LP is infeasible (the row would have said so)
        std::vector<int> row_in;                       // given row -> row of the engine's model, or -1 (folded)
        std::vector<int> fold_col;                     // given row -> column it bounds (folded rows)
        std::vector<double> fold_a;                    //              its coefficient
        std::vector<double> lb0, ub0;                  // bounds as given (M0 + N0), kept for set_bounds
        std::vector<int> lo_src, up_src;               // per column: the folded row whose bound is the tighter one, or -1 (its own)
        std::vector<double> clo, cup;                  // per column: folded bounds
        int map_var(int v) const { return v < M0 ? row_in[v] : (M0 - nfold) + (v - M0); }
    } ps;
};

static int materialise_indices(bslv_lpq *h, const int *list, int n);
static int ensure_batch(bslv_lpq *h, int B)
{
    if (B <= h->Bcap) return 0;
    if (h->lazy_open) { const int rc = materialise_indices(h, nullptr, 0); if (rc) return rc; h->lazy_open = false; }      // (the buffers below hold the pending pivots of the last batch)
    int cap = std::max(B, h->Bcap * 2);
    // every pointer is cleared as it is freed: when one of the allocations below fails, destroy() and a later ensure_batch()
    // see nullptr for what is gone instead of freeing it a second time
    auto fr = [](auto *&p) { if (p) (void)hipFree(p); p = nullptr; };
    fr(h->src_d); fr(h->dst_d); fr(h->status_d); fr(h->iters_d); fr(h->mode_d); fr(h->ver_d); fr(h->active_d); fr(h->work_d); fr(h->qslot_d);
    fr(h->vlo_d); fr(h->vup_d); fr(h->prow_d); fr(h->desc_d); fr(h->npend_d); fr(h->flushed_d); fr(h->pcol_d); fr(h->dcur_d); fr(h->dper_d); fr(h->pflags_d); fr(h->stall_d);
    fr(h->trow_d); fr(h->uvec_d); fr(h->xfull_d); fr(h->hmail_d);
    if (h->status_h) { (void)hipHostFree(h->status_h); h->status_h = nullptr; }
    if (h->active_h) { (void)hipHostFree(h->active_h); h->active_h = nullptr; }
    h->Bcap = 0;
    HIP_TRY(malloc0(&h->src_d, cap * sizeof(int)));
    HIP_TRY(malloc0(&h->qslot_d, cap * sizeof(int)));
    HIP_TRY(malloc0(&h->dst_d, cap * sizeof(int)));
    HIP_TRY(malloc0(&h->status_d, cap * sizeof(int)));
    HIP_TRY(malloc0(&h->iters_d, cap * sizeof(int)));
    HIP_TRY(malloc0(&h->mode_d, cap * sizeof(int)));
    HIP_TRY(malloc0(&h->work_d, cap * sizeof(int)));
    HIP_TRY(malloc0(&h->ver_d, cap * sizeof(int)));
    size_t vc = (size_t)std::max(1, h->L.vcnt);
    HIP_TRY(malloc0(&h->vlo_d, cap * vc * sizeof(double)));
    HIP_TRY(malloc0(&h->vup_d, cap * vc * sizeof(double)));
    HIP_TRY(malloc0(&h->prow_d, (size_t)cap * KP * h->L.ldt * sizeof(double)));
    if (h->L.rev) {
        HIP_TRY(malloc0(&h->trow_d, (size_t)cap * h->L.ld * sizeof(double)));
        HIP_TRY(malloc0(&h->uvec_d, (size_t)cap * h->L.ldt * sizeof(double)));
        HIP_TRY(malloc0(&h->xfull_d, (size_t)cap * h->L.N * sizeof(double)));
        HIP_TRY(malloc0(&h->hmail_d, (size_t)cap * 8 * sizeof(int)));
    }
    HIP_TRY(malloc0(&h->desc_d, (size_t)cap * KP * sizeof(PivDesc)));
    HIP_TRY(malloc0(&h->pcol_d, (size_t)cap * KP * h->L.Mp1p * sizeof(double)));
    HIP_TRY(malloc0(&h->dcur_d, (size_t)cap * h->L.ld * sizeof(double)));
    HIP_TRY(malloc0(&h->dper_d, (size_t)cap * h->L.ld * sizeof(double)));
    HIP_TRY(malloc0(&h->pflags_d, cap * sizeof(int)));
    HIP_TRY(malloc0(&h->stall_d, cap * sizeof(int)));
    if (!h->xstat_d) HIP_TRY(malloc0(&h->xstat_d, 8 * sizeof(int)));
    if (!h->dbg_d) HIP_TRY(malloc0(&h->dbg_d, 16 * sizeof(unsigned long long)));
    HIP_TRY(malloc0(&h->npend_d, cap * sizeof(int)));
    HIP_TRY(malloc0(&h->flushed_d, cap * sizeof(int)));
    HIP_TRY(hipHostMalloc(&h->status_h, cap * sizeof(int)));
    HIP_TRY(malloc0(&h->active_d, cap * sizeof(int)));
    HIP_TRY(hipHostMalloc(&h->active_h, cap * sizeof(int)));
    h->Bcap = cap;
    return 0;
}

static BatchView bview(bslv_lpq *h)
{
    BatchView v;
    v.src = h->src_d; v.dst = h->dst_d; v.vlo = h->vlo_d; v.vup = h->vup_d;
    v.status = h->status_d; v.iters = h->iters_d; v.mode = h->mode_d; v.verified = h->ver_d;
    v.desc = h->desc_d; v.prow = h->prow_d; v.pcol = h->pcol_d; v.dcur = h->dcur_d; v.npend = h->npend_d; v.flushed = h->flushed_d;
    v.work = h->work_d; v.nwork = h->nwork_d;
    v.dper = h->dper_d; v.pflags = h->pflags_d; v.stall = h->stall_d; v.xstat = h->xstat_d;
    v.trow = h->trow_d; v.uvec = h->uvec_d; v.xfull = h->xfull_d; v.dbg = h->dbg_d; v.hmail = h->hmail_d;
    v.lazy = (h->lazy && !h->L.rev) ? 1 : 0;
    return v;
}

static int upload_bounds(bslv_lpq *h, const double *lb, const double *ub)
{
    int M = h->L.M, N = h->L.N;
    std::vector<double> lo(lb, lb + M + N), up(ub, ub + M + N);
    std::vector<unsigned char> art(M + N, 0);
    // artificial bounds: a structural column with non-zero cost and no bound on the side its
    // reduced cost needs would be dual infeasible in the standard basis (oracle: primal phase 1)
    for (int j = 0; j < N; j++) {
        double c = h->cost[j + 1];
        int k = M + j;
        if (c > 0 && std::isinf(lo[k])) { lo[k] = -BIG; art[k] |= 1; }
        if (c < 0 && std::isinf(up[k])) { up[k] = BIG; art[k] |= 2; }
    }
    h->has_boxed = false;
    for (int k = 0; k < M + N; k++) if (!art[k] && std::isfinite(lo[k]) && std::isfinite(up[k]) && lo[k] < up[k]) h->has_boxed = true;
    HIP_TRY(hipMemcpyAsync(h->lb_d, lo.data(), (M + N) * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->ub_d, up.data(), (M + N) * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->art_d, art.data(), (M + N), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" {

const char *bslv_last_error(void) { return g_err; }

int bslv_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bslv_set_device(int device)
{
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) { set_error("bslv_set_device: device %d of %d", device, n); return BSLV_E_ARG; }
    HIP_TRY(hipSetDevice(device));
    return 0;
}

int bslv_device_info(char *name, int name_len, int *cus, size_t *mem_bytes)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) { strncpy(name, p.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    if (cus) *cus = p.multiProcessorCount;
    if (mem_bytes) *mem_bytes = p.totalGlobalMem;
    return 0;
}

static int raw_create(bslv_lpq **out, int M, int N, const double *A, const double *lb, const double *ub,
                      const double *cost, int var_first, int var_cnt, int pool_slots)
{
    if (!out || M < 1 || N < 1 || !A || !lb || !ub || !cost || pool_slots < 1 || var_cnt < 0 ||
        (var_cnt > 0 && (var_first < 0 || var_first + var_cnt > M + N))) {
        set_error("bslv_lpq_create: bad argument");
        return BSLV_E_ARG;
    }
    if (bslv_device_count() < 1) { set_error("no HIP device available"); return BSLV_E_NODEVICE; }
    bslv_lpq *h = new bslv_lpq();
    LpView &L = h->L;
    L.M = M; L.N = N;
    L.ld = (N + 15) / 16 * 16;
    L.Mp1 = M + 1;
    L.Mp1p = (M + 1 + 15) / 16 * 16;
    L.vfirst = var_first; L.vcnt = var_cnt;
    L.maxit = 50 * (M + N) + 1000;
    L.bland_after = 4 * (M + N) + 200;
    // tableau or revised form?  The revised form pays a sparse product per tableau row / column it looks at and moves M x M instead of
    // (M + 1) x N doubles per pass: for wide (N >= 2 M) and sparse (< 2 % non-zeros) problems.  BSLV_LP_REV=0/1 forces it.
    long nnz = 0;
    for (size_t k = 0; k < (size_t)M * N; k++) nnz += A[k] != 0.0;
    h->nnzA = nnz;
    // (by itself only where tableaux are out of reach -- 4 GiB and more each; ex09's 1.36 GB tableaux still fit 140 times into 288 GB and
    // the tableau form is the faster one for its one-LP-at-a-time phases and needs no refactorisation: DESIGN.md section 5)
    bool rev = N >= 2 * M && nnz * 50 < (long)M * N && (size_t)(M + 1) * (size_t)L.ld * sizeof(double) >= ((size_t)4 << 30);
    if (const char *e = getenv("BSLV_LP_REV")) rev = atoi(e) != 0;
    L.rev = rev ? 1 : 0;
    L.ldt = rev ? (M + 15) / 16 * 16 : L.ld;
    L.mrows = rev ? M : L.Mp1;
    L.slotT = (size_t)L.mrows * L.ldt;
    h->slots = pool_slots;
    h->cost.assign(cost, cost + N + 1);
    h->c0 = cost[0];
    int rc = 0;
    auto fail = [&](int code) { bslv_lpq_destroy(h); return code; };
    // BSLV_LP_CUMASK=<hex word>: the engine's stream may only use the CUs whose bit is set in the word (repeated over all CUs), e.g.
    // 77777777 = three of every four.  For the pipelined driver: the tableau passes saturate HBM from fewer CUs than the chip
    // has, and the small kernels of the cut phase, on their own stream, find free CUs instead of queueing behind them.
    if (const char *cm = getenv("BSLV_LP_CUMASK")) {
        const unsigned word = (unsigned)strtoul(cm, nullptr, 16);
        unsigned mask[16];
        for (int k = 0; k < 16; k++) mask[k] = word;
        if (hipExtStreamCreateWithCUMask(&h->stream, 16, mask) != hipSuccess) { (void)hipGetLastError(); h->stream = nullptr; }
    }
    if (!h->stream && hipStreamCreate(&h->stream) != hipSuccess) { set_error("hipStreamCreate failed"); return fail(BSLV_E_NODEVICE); }
#define TRYF(e) do { hipError_t _e = (e); if (_e != hipSuccess) { set_error("%s failed: %s", #e, hipGetErrorString(_e)); return fail(_e == hipErrorOutOfMemory ? BSLV_E_NOMEM : BSLV_E_NODEVICE); } } while (0)
    TRYF(malloc0(&L.T, (size_t)pool_slots * L.slotT * sizeof(double)));
    TRYF(malloc0(&L.beta, (size_t)pool_slots * L.Mp1p * sizeof(double)));
    TRYF(malloc0(&L.xN, (size_t)pool_slots * L.ld * sizeof(double)));
    TRYF(malloc0(&L.bh, (size_t)pool_slots * M * sizeof(int)));
    TRYF(malloc0(&L.nh, (size_t)pool_slots * N * sizeof(int)));
    TRYF(malloc0(&L.nstat, (size_t)pool_slots * N * sizeof(int)));
    TRYF(malloc0(&L.pos, (size_t)pool_slots * (M + N) * sizeof(int)));
    if (!rev) TRYF(malloc0(&h->Tstd, L.slotT * sizeof(double)));
    else {
        TRYF(malloc0(&h->dsl_d, (size_t)pool_slots * L.ld * sizeof(double)));
        TRYF(malloc0(&h->cost_d, (size_t)(N + 1) * sizeof(double)));
        TRYF(malloc0(&h->cptr_d, (size_t)(N + 1) * sizeof(int)));
        TRYF(malloc0(&h->rptr_d, (size_t)(M + 1) * sizeof(int)));
        TRYF(malloc0(&h->cidx_d, (size_t)std::max(nnz, 1L) * sizeof(int)));
        TRYF(malloc0(&h->ridx_d, (size_t)std::max(nnz, 1L) * sizeof(int)));
        TRYF(malloc0(&h->cval_d, (size_t)std::max(nnz, 1L) * sizeof(double)));
        TRYF(malloc0(&h->rval_d, (size_t)std::max(nnz, 1L) * sizeof(double)));
    }
    TRYF(malloc0(&h->lb_d, (M + N) * sizeof(double)));
    TRYF(malloc0(&h->ub_d, (M + N) * sizeof(double)));
    TRYF(malloc0(&h->art_d, (M + N)));
#undef TRYF
    L.lb = h->lb_d; L.ub = h->ub_d; L.art = h->art_d;
    L.dsl = h->dsl_d; L.cptr = h->cptr_d; L.cidx = h->cidx_d; L.cval = h->cval_d; L.rptr = h->rptr_d; L.ridx = h->ridx_d; L.rval = h->rval_d;
    if (rev) {
        std::vector<int> cptr(N + 1, 0), rptr(M + 1, 0), cidx((size_t)nnz), ridx((size_t)nnz);
        std::vector<double> cval((size_t)nnz), rval((size_t)nnz);
        size_t t = 0;
        for (int i = 0; i < M; i++) { rptr[i] = (int)t; for (int j = 0; j < N; j++) { const double a = A[(size_t)i * N + j]; if (a != 0.0) { ridx[t] = j; rval[t] = a; t++; cptr[j + 1]++; } } }
        rptr[M] = (int)t;
        for (int j = 0; j < N; j++) cptr[j + 1] += cptr[j];
        std::vector<int> fill(cptr.begin(), cptr.end() - 1);
        for (int i = 0; i < M; i++) for (int u = rptr[i]; u < rptr[i + 1]; u++) { const int j = ridx[u]; cidx[fill[j]] = i; cval[fill[j]] = rval[u]; fill[j]++; }
        bool okc = hipMemcpy(h->cptr_d, cptr.data(), (N + 1) * sizeof(int), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(h->rptr_d, rptr.data(), (M + 1) * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
                   hipMemcpy(h->cost_d, cost, (N + 1) * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
        if (nnz) okc = okc && hipMemcpy(h->cidx_d, cidx.data(), nnz * sizeof(int), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(h->ridx_d, ridx.data(), nnz * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
                             hipMemcpy(h->cval_d, cval.data(), nnz * sizeof(double), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(h->rval_d, rval.data(), nnz * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
        if (!okc) { set_error("upload of A (CSC / CSR) failed"); return fail(BSLV_E_NODEVICE); }
    } else {
        std::vector<double> img(L.slotT, 0.0);
        for (int i = 0; i < M; i++) memcpy(&img[(size_t)i * L.ld], A + (size_t)i * N, N * sizeof(double));
        for (int j = 0; j < N; j++) img[(size_t)M * L.ld + j] = cost[j + 1];
        if (hipMemcpy(h->Tstd, img.data(), L.slotT * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("upload of A failed");
            return fail(BSLV_E_NODEVICE);
        }
    }
    if ((rc = upload_bounds(h, lb, ub))) return fail(rc);
    if ((rc = ensure_batch(h, 64))) return fail(rc);
    h->force_ext = L.slotT * sizeof(double) >= ((size_t)1 << 30) || (rev && (size_t)(M + 1) * L.ld * sizeof(double) >= ((size_t)1 << 30));     // (large problems are degenerate problems: no stalling at a millisecond per pivot)
    {   // k_flush stages KP pivot rows in LDS: wide problems (N > ~1300) need more than the default 64 KB
        const size_t want = (size_t)KP * h->L.ldt * sizeof(double);
        if (want > h->flush_lds_max) {
            if (want <= 144 * 1024 && hipFuncSetAttribute((const void *)k_flush<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) == hipSuccess) h->flush_lds_max = want;
            else (void)hipGetLastError();
        }
    }
    *out = h;
    return 0;
}

void bslv_lpq_destroy(bslv_lpq *h)
{
    if (!h) return;
    auto fr = [](void *p) { if (p) (void)hipFree(p); };
    fr(h->L.T); fr(h->L.beta); fr(h->L.xN); fr(h->L.bh); fr(h->L.nh); fr(h->L.nstat); fr(h->L.pos);
    fr(h->Tstd); fr(h->lb_d); fr(h->ub_d); fr(h->art_d);
    fr(h->dbg_d); fr(h->list_d); fr(h->cptr_d); fr(h->cidx_d); fr(h->rptr_d); fr(h->ridx_d); fr(h->cval_d); fr(h->rval_d); fr(h->cost_d); fr(h->dsl_d); fr(h->trow_d); fr(h->uvec_d); fr(h->xfull_d); fr(h->hmail_d);
    fr(h->src_d); fr(h->dst_d); fr(h->status_d); fr(h->iters_d); fr(h->mode_d); fr(h->ver_d); fr(h->qslot_d);
    fr(h->vlo_d); fr(h->vup_d); fr(h->prow_d); fr(h->desc_d); fr(h->out_d); fr(h->active_d); fr(h->work_d); fr(h->nwork_d); fr(h->npend_d); fr(h->flushed_d); fr(h->pcol_d); fr(h->dcur_d); fr(h->dper_d); fr(h->pflags_d); fr(h->stall_d); fr(h->xstat_d); fr(h->cvals_d);
    if (h->status_h) (void)hipHostFree(h->status_h);
    if (h->active_h) (void)hipHostFree(h->active_h);
    for (auto &e : h->evpool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int bslv_lpq_pool_slots(const bslv_lpq *h) { return h ? h->slots : 0; }
size_t bslv_lpq_slot_bytes(const bslv_lpq *h)
{
    if (!h) return 0;
    const LpView &L = h->L;
    return (L.slotT + L.Mp1p + L.ld + (L.rev ? L.ld : 0)) * sizeof(double) + (size_t)(2 * (L.M + L.N) + L.N) * sizeof(int);
}

// bounds of the engine's model from the bounds of the model as given: the folded rows tighten their columns
static void fold_bounds(bslv_lpq *h, const double *lb, const double *ub, std::vector<double> &lo, std::vector<double> &up)
{
    bslv_lpq::Presolve &P = h->ps;
    const int M0 = P.M0, N0 = P.N0, Mi = M0 - P.nfold;
    P.lb0.assign(lb, lb + M0 + N0); P.ub0.assign(ub, ub + M0 + N0);
    lo.assign((size_t)Mi + N0, 0.0); up.assign((size_t)Mi + N0, 0.0);
    P.clo.assign(lb + M0, lb + M0 + N0); P.cup.assign(ub + M0, ub + M0 + N0);
    P.lo_src.assign(N0, -1); P.up_src.assign(N0, -1);
    for (int i = 0; i < M0; i++) {
        if (P.row_in[i] >= 0) { lo[P.row_in[i]] = lb[i]; up[P.row_in[i]] = ub[i]; continue; }
        const int j = P.fold_col[i];
        const double a = P.fold_a[i];
        const double l = a > 0 ? lb[i] / a : ub[i] / a, u = a > 0 ? ub[i] / a : lb[i] / a;
        if (l > P.clo[j]) { P.clo[j] = l; P.lo_src[j] = i; }
        if (u < P.cup[j]) { P.cup[j] = u; P.up_src[j] = i; }
    }
    // new bounds may leave a folded row no room (the fold set is fixed at create time): the model as given is infeasible -- reported as
    // such by solve_batch; the engine itself gets a consistent (degenerate) box
    P.empty_box = false;
    for (int j = 0; j < N0; j++) {
        if (P.clo[j] > P.cup[j] + TOL_BND * (1.0 + fabs(P.cup[j]))) { P.empty_box = true; P.cup[j] = P.clo[j]; }
        lo[Mi + j] = P.clo[j]; up[Mi + j] = P.cup[j];
    }
}
int bslv_lpq_set_bounds(bslv_lpq *h, const double *lb, const double *ub)
{
    if (!h || !lb || !ub) { set_error("bslv_lpq_set_bounds: bad argument"); return BSLV_E_ARG; }
    if (h->ps.nfold == 0) { h->ps.lb0.assign(lb, lb + h->ps.M0 + h->ps.N0); h->ps.ub0.assign(ub, ub + h->ps.M0 + h->ps.N0); return upload_bounds(h, lb, ub); }
    std::vector<double> lo, up;
    fold_bounds(h, lb, ub, lo, up);
    return upload_bounds(h, lo.data(), up.data());
}
int bslv_lpq_rows_folded(const bslv_lpq *h) { return h ? h->ps.nfold : 0; }
int bslv_lpq_is_revised(const bslv_lpq *h) { return h ? h->L.rev : 0; }

int bslv_lpq_create(bslv_lpq **out, int M, int N, const double *A, const double *lb, const double *ub,
                    const double *cost, int var_first, int var_cnt, int pool_slots)
{
    if (!out || M < 1 || N < 1 || !A || !lb || !ub || !cost || pool_slots < 1 || var_cnt < 0 ||
        (var_cnt > 0 && (var_first < 0 || var_first + var_cnt > M + N))) {
        set_error("bslv_lpq_create: bad argument");
        return BSLV_E_ARG;
    }
    bslv_lpq::Presolve P;
    P.M0 = M; P.N0 = N;
    P.row_in.assign(M, 0); P.fold_col.assign(M, -1); P.fold_a.assign(M, 0.0);
    std::vector<double> clo(lb + M, lb + M + N), cup(ub + M, ub + M + N);
    const bool off = getenv("BSLV_NO_PRESOLVE") != nullptr;
    int kept = 0;
    for (int i = 0; i < M; i++) {
        bool fold = false;
        const bool per_lp = var_cnt > 0 && i >= var_first && i < var_first + var_cnt;
        if (!off && !per_lp && M - P.nfold > 1) {
            int nz = 0, jj = -1;
            const double *row = A + (size_t)i * N;
            for (int j = 0; j < N && nz < 2; j++) if (row[j] != 0.0) { nz++; jj = j; }
            // (a column whose bounds are given per LP keeps its rows: solve_batch would overwrite the folded bound)
            const bool col_per_lp = nz == 1 && var_cnt > 0 && M + jj >= var_first && M + jj < var_first + var_cnt;
            if (nz == 1 && !col_per_lp) {
                const double a = row[jj];
                const double l = std::max(clo[jj], a > 0 ? lb[i] / a : ub[i] / a), u = std::min(cup[jj], a > 0 ? ub[i] / a : lb[i] / a);
                if (l <= u) { fold = true; clo[jj] = l; cup[jj] = u; P.fold_col[i] = jj; P.fold_a[i] = a; }      // (an empty box stays a row: the LP reports it)
            }
        }
        if (fold) { P.row_in[i] = -1; P.nfold++; } else P.row_in[i] = kept++;
    }
    if (P.nfold == 0) {
        const int rc = raw_create(out, M, N, A, lb, ub, cost, var_first, var_cnt, pool_slots);
        if (rc) return rc;
        (*out)->ps = P;
        (*out)->ps.lb0.assign(lb, lb + M + N); (*out)->ps.ub0.assign(ub, ub + M + N);
        return 0;
    }
    const int Mi = M - P.nfold;
    std::vector<double> Ai((size_t)Mi * N);
    for (int i = 0; i < M; i++) if (P.row_in[i] >= 0) memcpy(&Ai[(size_t)P.row_in[i] * N], A + (size_t)i * N, (size_t)N * sizeof(double));
    // the per-LP range keeps its place among the rows that stay (no row inside it is folded); a range over columns moves with them
    int vf = var_first;
    if (var_cnt > 0) vf = var_first < M ? P.row_in[var_first] : Mi + (var_first - M);
    std::vector<double> lo0((size_t)Mi + N, 0.0), up0((size_t)Mi + N, 0.0);
    for (int i = 0; i < M; i++) if (P.row_in[i] >= 0) { lo0[P.row_in[i]] = lb[i]; up0[P.row_in[i]] = ub[i]; }
    for (int j = 0; j < N; j++) { lo0[Mi + j] = clo[j]; up0[Mi + j] = cup[j]; }
    const int rc = raw_create(out, Mi, N, Ai.data(), lo0.data(), up0.data(), cost, vf, var_cnt, pool_slots);
    if (rc) return rc;
    (*out)->ps = P;
    std::vector<double> lo, up;
    fold_bounds(*out, lb, ub, lo, up);                    // (fills lo_src / up_src and the record of the bounds as given)
    return 0;
}

int bslv_lpq_set_profile(bslv_lpq *h, int on)
{
    if (!h) return BSLV_E_ARG;
    h->profile = on != 0;
    return 0;
}

int bslv_lpq_reset_slot(bslv_lpq *h, int slot)
{
    if (!h || slot < 0 || slot >= h->slots) { set_error("bslv_lpq_reset_slot: bad slot %d", slot); return BSLV_E_ARG; }
    if (h->lazy_open) { const int rc = materialise_indices(h, nullptr, 0); if (rc) return rc; h->lazy_open = false; }      // (a pending pass must not land on the slot after it has been reset)
    LpView &L = h->L;
    if (L.rev) {
        const size_t nk = std::max((size_t)L.M * L.ldt, (size_t)L.ld);
        hipLaunchKernelGGL(k_rev_identity, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, h->stream, L, slot, (const double *)h->cost_d);
    } else HIP_TRY(hipMemcpyAsync(L.T + (size_t)slot * L.slotT, h->Tstd, L.slotT * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    int n = std::max(std::max(L.M, L.N), std::max(L.ld, L.Mp1p));
    hipLaunchKernelGGL(k_std_heads, dim3((n + 255) / 256), dim3(256), 0, h->stream, L, slot);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// ---- lazy tableaux -----------------------------------------------------------------------------------------------------------
// The first pass of a solve streams the parent's tableau into the LP's own slot (k_flush): 8 MB written per LP on S-mid, 17 GB per
// batch of 2048 -- the largest single item of the LP phase -- and for nothing where nobody reads the slot again: only the LP of a
// vertex that yields a NEW cut becomes the parent of later LPs (28 % of a batch on S-mid; the rest confirm their vertex or return a
// cut a sibling delivered).  Values, duals and the objective of a finished LP are all in its vectors.  With bslv_lpq_set_lazy(h, 1)
// an LP that is finished when its pass would be due keeps its pending pivots (<= KP; one that needs more passes as before), and the
// caller names the slots it will use as parents: bslv_lpq_materialise(h, n, slots) gives those their tableau -- the same pass, the
// same arithmetic -- and bslv_lpq_discard_pending(h) drops the rest.  A batch that is started while slots are still open gives all
// of them their tableau first (the retry batches of the driver).  Slots that were not materialised must not be used as `src`.
static int flush_list(bslv_lpq *h, int cnt_slot, int upper)
{
    LpView &L = h->L;
    hipStream_t s = h->stream;
    BatchView bv = bview(h);
    const int wide = (size_t)KP * L.ldt * sizeof(double) > h->flush_lds_max;
    const size_t lds = wide ? 0 : (size_t)KP * L.ldt * sizeof(double);
    const bool big_flush = getenv("BSLV_FLUSH_NT") ? atoi(getenv("BSLV_FLUSH_NT")) > NT : lds > 53 * 1024;
    const int tiles = (L.mrows + TR - 1) / TR;
    int tr = upper * tiles >= 2048 ? 32 : (upper * tiles * 2 >= 2048 ? 16 : (upper * tiles * 4 >= 2048 ? 8 : 4));
    if (big_flush) { const long rows = (long)upper * L.mrows; tr = rows >= 2048L * 128 ? 128 : rows >= 2048L * 64 ? 64 : rows >= 2048L * 32 ? 32 : 16; }
    const int ntile = (L.mrows + tr - 1) / tr, fnt = big_flush ? NT_BIG : NT;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->profile) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventRecord(e0, s)); }
    if (wide) hipLaunchKernelGGL(k_flush<true>, dim3(std::min(upper * ntile, h->upd_grid)), dim3(fnt), 0, s, L, bv, cnt_slot, ntile, tr);
    else hipLaunchKernelGGL(k_flush<false>, dim3(std::min(upper * ntile, h->upd_grid)), dim3(fnt), lds, s, L, bv, cnt_slot, ntile, tr);
    if (h->profile) HIP_TRY(hipEventRecord(e1, s));
    hipLaunchKernelGGL(k_after_flush, dim3((upper + 255) / 256), dim3(256), 0, s, bv, cnt_slot);
    HIP_TRY(hipGetLastError());
    int n = 0;
    HIP_TRY(hipMemcpyAsync(&n, h->nwork_d + cnt_slot, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    h->last_passes += n;
    h->last_launches += 1;
    h->lazy_materialised += n;
    if (h->profile) { float t = 0; (void)hipEventElapsedTime(&t, e0, e1); h->last_update_ms += t; (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
    return 0;
}
// batch indices list[0..n) of the last batch (nullptr: all of it): their slots get their tableau
static int materialise_indices(bslv_lpq *h, const int *list, int n)
{
    if (!h->lazy_open || (list && n == 0)) return 0;
    hipStream_t s = h->stream;
    BatchView bv = bview(h);
    const int B = (int)h->last_dst.size();
    std::vector<int> all;
    if (!list) { all.resize(B); for (int b = 0; b < B; b++) all[b] = b; list = all.data(); n = B; }
    if (n > h->listcap) { if (h->list_d) (void)hipFree(h->list_d); h->list_d = nullptr; HIP_TRY(malloc0s(&h->list_d, (size_t)std::max(n, 1024) * sizeof(int), s)); h->listcap = std::max(n, 1024); }
    const int cnt_slot = h->L.maxit + 42;
    HIP_TRY(hipMemcpyAsync(h->list_d, list, (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(h->nwork_d + cnt_slot, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_list_given, dim3((n + 255) / 256), dim3(256), 0, s, bv, (const int *)h->list_d, n, cnt_slot);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));              // (`all` / the caller's list may go out of scope)
    int rc = flush_list(h, cnt_slot, n);
    if (rc) return rc;
    // row M again: the pass has applied the pending pivots to the row it found -- for an LP that had passed once before and finished
    // with pivots pending that was ALREADY the final row (k_store_d at the end of the solve), now updated twice; the vector is the truth
    hipLaunchKernelGGL(k_store_d, dim3((h->L.ld + 255) / 256, B), dim3(256), 0, s, h->L, bv, B);
    HIP_TRY(hipGetLastError());
    return 0;
}
static int solve_batch_impl(bslv_lpq *h, int B, const int *src, const int *dst, const double *vlo, const double *vup,
                            int cfirst, int ccnt, const double *cvals, int *status, int *iters);
int bslv_lpq_set_lazy(bslv_lpq *h, int on)
{
    if (!h) return BSLV_E_ARG;
    if (!on && h->lazy_open) { int rc = materialise_indices(h, nullptr, 0); if (rc) return rc; h->lazy_open = false; }
    h->lazy = on != 0;
    return 0;
}
int bslv_lpq_materialise(bslv_lpq *h, int n, const int *slots)
{
    if (!h || n < 0 || (n && !slots)) { set_error("bslv_lpq_materialise: bad argument"); return BSLV_E_ARG; }
    if (!h->lazy_open || n == 0) return 0;
    std::vector<int> idx;
    const auto t0 = std::chrono::steady_clock::now();
    struct Clock { bslv_lpq *h; std::chrono::steady_clock::time_point t; ~Clock() { h->lazy_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); } } clock{h, t0};
    {
        std::vector<int> where(h->slots, -1);
        for (size_t b = 0; b < h->last_dst.size(); b++) where[h->last_dst[b]] = (int)b;
        for (int k = 0; k < n; k++) if (slots[k] >= 0 && slots[k] < h->slots && where[slots[k]] >= 0) idx.push_back(where[slots[k]]);      // (slots of earlier batches have theirs)
    }
    return materialise_indices(h, idx.data(), (int)idx.size());
}
int bslv_lpq_discard_pending(bslv_lpq *h)
{
    if (!h) return BSLV_E_ARG;
    h->lazy_open = false;
    return 0;
}
int bslv_lpq_lazy_stats(const bslv_lpq *h, long out[3])
{
    if (!h || !out) return BSLV_E_ARG;
    out[0] = h->lazy_skipped; out[1] = h->lazy_materialised; out[2] = (long)(h->lazy_ms * 1000.0);
    return 0;
}
int bslv_lpq_solve_batch(bslv_lpq *h, int B, const int *src, const int *dst, const double *vlo,
                         const double *vup, int *status, int *iters)
{
    return solve_batch_impl(h, B, src, dst, vlo, vup, 0, 0, nullptr, status, iters);
}
// The LPs of the batch differ in their OBJECTIVE (lp_set_obj_coeffs + lp_solve, bslv_lp.c:141-151,219: what phase2_dual
// does per vertex, bslv_algs.c:1469-1477): cost costs[b*cost_cnt + t] on variable cost_first + t, 0 elsewhere (the engine's
// own cost vector must be zero).  Bounds: those of the last solve_batch / set_bounds for the per-LP range (vlo/vup may be
// NULL when the engine has no such range).  LP b starts from the basis of slot src[b], which must be primal feasible for
// these bounds (an optimal slot of any objective is), and runs primal simplex steps.
int bslv_lpq_solve_batch_obj(bslv_lpq *h, int B, const int *src, const int *dst, const double *vlo, const double *vup,
                             int cost_first, int cost_cnt, const double *costs, int *status, int *iters)
{
    if (!h || cost_cnt < 1 || cost_first < 0 || cost_first + cost_cnt > h->ps.M0 + h->ps.N0 || !costs) { set_error("bslv_lpq_solve_batch_obj: bad argument"); return BSLV_E_ARG; }
    if (h->ps.nfold) {      // indices of the model as given -> the engine's (a cost on a folded row would be a cost on its column: not asked for by any caller)
        for (int t = 0; t < cost_cnt; t++) if (h->ps.map_var(cost_first + t) != h->ps.map_var(cost_first) + t) { set_error("bslv_lpq_solve_batch_obj: the cost range covers rows the presolve folded into column bounds"); return BSLV_E_ARG; }
        cost_first = h->ps.map_var(cost_first);
    }
    for (size_t j = 0; j < h->cost.size(); j++) if (h->cost[j] != 0.0) { set_error("bslv_lpq_solve_batch_obj: the engine was created with a non-zero cost vector"); return BSLV_E_STATE; }
    return solve_batch_impl(h, B, src, dst, vlo, vup, cost_first, cost_cnt, costs, status, iters);
}
static int solve_batch_impl(bslv_lpq *h, int B, const int *src, const int *dst, const double *vlo, const double *vup,
                            int cfirst, int ccnt, const double *cvals, int *status, int *iters)
{
    if (h && h->ps.empty_box && B > 0 && status) {        // (bslv_lpq_set_bounds left a folded row no room: fold_bounds)
        for (int b = 0; b < B; b++) { status[b] = BSLV_LP_INFEASIBLE; if (iters) iters[b] = 0; }
        h->last_iters = 0; h->last_pivots = 0; h->last_passes = 0; h->last_launches = 0;
        return 0;
    }
    if (!h || B < 0 || (B > 0 && (!src || !dst)) || (B > 0 && h->L.vcnt > 0 && (!vlo || !vup))) {
        set_error("bslv_lpq_solve_batch: bad argument");
        return BSLV_E_ARG;
    }
    if (B == 0) return 0;
    for (int b = 0; b < B; b++)
        if (src[b] < 0 || src[b] >= h->slots || dst[b] < 0 || dst[b] >= h->slots) {
            set_error("bslv_lpq_solve_batch: slot out of range at %d (src %d dst %d, pool %d)", b, src[b], dst[b], h->slots);
            return BSLV_E_ARG;
        }
    int rc;
    if (h->lazy_open) {        // a batch while slots of the last one are still without their tableau (a retry of the driver): all of them get it now
        if ((rc = materialise_indices(h, nullptr, 0))) return rc;
        h->lazy_open = false;
    }
    if ((rc = ensure_batch(h, B))) return rc;
    LpView &L = h->L;
    const int wide = (size_t)KP * L.ldt * sizeof(double) > h->flush_lds_max;      // pivot rows from global memory in k_flush
    auto t0 = std::chrono::steady_clock::now();
    hipStream_t s = h->stream;
    HIP_TRY(hipMemcpyAsync(h->src_d, src, B * sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->dst_d, dst, B * sizeof(int), hipMemcpyHostToDevice, s));
    if (L.vcnt > 0) {
        HIP_TRY(hipMemcpyAsync(h->vlo_d, vlo, (size_t)B * L.vcnt * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(h->vup_d, vup, (size_t)B * L.vcnt * sizeof(double), hipMemcpyHostToDevice, s));
    }
    L.objmode = cvals ? 1 : 0; L.cfirst = cfirst; L.ccnt = ccnt;
    if (cvals && L.rev) { set_error("bslv_lpq_solve_batch_obj: not available in the revised form (BSLV_LP_REV=0 forces the tableau form)"); return BSLV_E_STATE; }
    if (cvals) {
        const size_t need = (size_t)B * ccnt;
        if (need > h->cvals_cap) { if (h->cvals_d) (void)hipFree(h->cvals_d); h->cvals_d = nullptr; HIP_TRY(malloc0s(&h->cvals_d, need * sizeof(double), s)); h->cvals_cap = need; }
        HIP_TRY(hipMemcpyAsync(h->cvals_d, cvals, need * sizeof(double), hipMemcpyHostToDevice, s));
    }
    if (const char *e = getenv("BSLV_UPD_GRID")) h->upd_grid = std::max(64, atoi(e));
    {   // one work-list length per lock-step iteration, zeroed here: no reset between iterations
        const int need = L.maxit + 64;
        if (need > h->nworkcap) { if (h->nwork_d) (void)hipFree(h->nwork_d); h->nwork_d = nullptr; HIP_TRY(malloc0s(&h->nwork_d, need * sizeof(int), s)); h->nworkcap = need; }
        HIP_TRY(hipMemsetAsync(h->nwork_d, 0, need * sizeof(int), s));
    }
    HIP_TRY(hipMemsetAsync(h->xstat_d, 0, 8 * sizeof(int), s));
    BatchView bv = bview(h);
    bv.cvals = h->cvals_d;
    const int tiles = (L.mrows + TR - 1) / TR;
    hipLaunchKernelGGL(k_prep, dim3(B), dim3(NT), 0, s, L, bv, B);
    if (L.rev) hipLaunchKernelGGL(k_rev_u, dim3(B), dim3(NT), 0, s, L, bv, B, (const int *)nullptr, 0);      // beta = B^-1 uvec (k_init)
    if (L.objmode) {      // new objective: the tableau rows are copied up front (the reduced-cost row is rebuilt by k_prep, not streamed from the parent)
        const int cnt_slot = L.maxit + 41;
        hipLaunchKernelGGL(k_list_unpivoted, dim3((B + 255) / 256), dim3(256), 0, s, bv, B, -cnt_slot);
        hipLaunchKernelGGL(k_copy_unpivoted, dim3(std::min(B * tiles, 2048)), dim3(NT), 0, s, L, bv, cnt_slot, tiles);
    }
    hipLaunchKernelGGL(k_init, dim3(tiles, B), dim3(NT), 0, s, L, bv, B);
    HIP_TRY(hipGetLastError());
    const size_t lds = wide ? 0 : (size_t)KP * L.ldt * sizeof(double);
    // (160 KB of LDS per CU: three workgroups of NT threads need lds <= ~53 KB)
    const bool big_flush = getenv("BSLV_FLUSH_NT") ? atoi(getenv("BSLV_FLUSH_NT")) > NT : lds > 53 * 1024;
    // bound flipping ratio test only where a variable has two finite, non-artificial bounds
    bool bfrt = h->has_boxed || L.objmode || h->force_ext;        // (the primal steps live in the extended selection)
    if (!bfrt && L.vcnt > 0)
        for (size_t k = 0; k < (size_t)B * L.vcnt && !bfrt; k++) bfrt = std::isfinite(vlo[k]) && std::isfinite(vup[k]) && vlo[k] < vup[k];
    if (getenv("BSLV_LP_EXT")) bfrt = atoi(getenv("BSLV_LP_EXT")) != 0;      // test hook: force the extended selection on / off
    int cap2 = 2;
    while (cap2 < L.N) cap2 <<= 1;
    size_t sel_lds = (size_t)cap2 * (sizeof(double) + sizeof(int)) + (size_t)L.N;
    if (bfrt && sel_lds > 144 * 1024) { cap2 = 0; sel_lds = (size_t)L.N; }     // rows too long for the in-LDS sort: extended selection without the long-step part
    if (bfrt && sel_lds > h->select_lds_max) {
        if (sel_lds <= 144 * 1024 && hipFuncSetAttribute((const void *)k_select<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sel_lds) == hipSuccess) h->select_lds_max = sel_lds;
        else bfrt = false;
    }
    if (L.objmode && !bfrt) { set_error("bslv_lpq_solve_batch_obj: rows too long for the extended selection (N=%d)", L.N); return BSLV_E_CAPACITY; }
    // revised form: rho (a row of B^-1) in LDS behind the selection's own arrays, when there is room
    size_t sel_lds_launch = bfrt ? sel_lds : 0;
    L.rho_off = -1;
    if (L.rev) {
        const size_t off = bfrt ? (sel_lds + 15) / 16 * 16 : 0, want = off + (size_t)L.ldt * sizeof(double);
        size_t &lim = bfrt ? h->select_lds_max : h->select0_lds_max;
        bool ok = want <= lim;
        if (!ok && want <= 144 * 1024) {
            ok = (bfrt ? hipFuncSetAttribute((const void *)k_select<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want)
                       : hipFuncSetAttribute((const void *)k_select<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want)) == hipSuccess;
            if (ok) lim = want; else (void)hipGetLastError();
        }
        if (ok) { L.rho_off = (int)off; sel_lds_launch = want; }
    }
    L.trace = getenv("BSLV_LP_TRACE") ? atoi(getenv("BSLV_LP_TRACE")) : -1;
    L.stall_limit = getenv("BSLV_STALL_LIMIT") ? atoi(getenv("BSLV_STALL_LIMIT")) : STALL_LIMIT;
    L.pert_scale = getenv("BSLV_PERT_SCALE") ? atof(getenv("BSLV_PERT_SCALE")) : 1.0;
    // One ROUND = KP lock-step selections on vectors, then one pass over the tableaux of the LPs that have something
    // pending (k_flush).  The status vector is read back every 1, 2, 4, ... rounds.
    // (measured, BSLV_SELECT_NT: S-degenerate, 2011 columns, 64 LPs per step: LP phase 67.6 / 52.0 / 46.6 ms with 256 / 512 / 1024 threads;
    //  S-degenerate-q4 to termination with 256 LPs per step 12.1 -> 10.4 s; same pivots)
    const int sel_nt = getenv("BSLV_SELECT_NT") ? atoi(getenv("BSLV_SELECT_NT")) : (L.N >= 1536 ? NT_BIG : NT);
    const int sel_per_launch = (getenv("BSLV_SELECT_FUSE") && atoi(getenv("BSLV_SELECT_FUSE")) == 0) ? 1 : KP;     // selections per k_select launch
    int it = 0, chunk = 1, running = B;
    for (int b = 0; b < B; b++) h->active_h[b] = b;
    HIP_TRY(hipMemcpyAsync(h->active_d, h->active_h, B * sizeof(int), hipMemcpyHostToDevice, s));
    size_t nev = 0;
    h->last_update_ms = 0;
    static const int max_rounds = getenv("BSLV_LP_MAXROUNDS") ? atoi(getenv("BSLV_LP_MAXROUNDS")) : 0;      // (timing experiments)
    L.probe = getenv("BSLV_REV_PROBE") ? atoi(getenv("BSLV_REV_PROBE")) : 0;
    while (running > 0 && it < L.maxit + 8 && !(max_rounds && it >= max_rounds)) {
        for (int c = 0; c < chunk; c++, it++) {
            for (int lev = 0; lev < KP; lev += sel_per_launch) {
                // revised form: helper workgroups for the sparse products of a tableau row (rev_helper), as many per LP as the chip holds
                // beside the LPs' own workgroups without anyone waiting for a place
                int helpers = 1;
                if (L.rev && sel_nt == NT_BIG) {
                    static const int hmax = getenv("BSLV_REV_HELPERS") ? std::max(1, atoi(getenv("BSLV_REV_HELPERS"))) : 32;
                    helpers = std::max(1, std::min(std::min(hmax, (L.ld + 2047) / 2048), 256 / std::max(1, running)));
                }
                L.helpers = helpers;
                L.launch_id = (++h->launch_seq) & 0x3FFFFF;
                if (bfrt) hipLaunchKernelGGL(k_select<true>, dim3(running, helpers), dim3(sel_nt), sel_lds_launch, s, L, bv, h->active_d, running, cap2, sel_per_launch);
                else hipLaunchKernelGGL(k_select<false>, dim3(running, helpers), dim3(sel_nt), sel_lds_launch, s, L, bv, h->active_d, running, 0, sel_per_launch);
            }
            hipLaunchKernelGGL(k_list_pending, dim3((running + 255) / 256), dim3(256), 0, s, bv, h->active_d, running, it);
            if (h->profile) {
                if (nev == h->evpool.size()) {
                    hipEvent_t a, b2;
                    HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b2));
                    h->evpool.emplace_back(a, b2);
                }
                HIP_TRY(hipEventRecord(h->evpool[nev].first, s));
            }
            // few LPs left: smaller row tiles keep >= ~2k workgroups in flight
            int tr = running * tiles >= 2048 ? 32 : (running * tiles * 2 >= 2048 ? 16 : (running * tiles * 4 >= 2048 ? 8 : 4));      // (4: one row per wave -- a single LP of a few thousand rows)
            if (big_flush) {                                            // 16 waves per workgroup: at least one row per wave, more where the batch still fills the chip
                const long rows = (long)running * L.mrows;
                tr = rows >= 2048L * 128 ? 128 : rows >= 2048L * 64 ? 64 : rows >= 2048L * 32 ? 32 : 16;
            }
            const int ntile = (L.mrows + tr - 1) / tr;
            const int fnt = big_flush ? NT_BIG : NT;
            if (L.rev) hipLaunchKernelGGL(k_rev_u, dim3(running), dim3(NT), 0, s, L, bv, B, (const int *)h->work_d, it);      // (only the LPs that asked for a refresh of beta)
            if (wide) hipLaunchKernelGGL(k_flush<true>, dim3(std::min(running * ntile, h->upd_grid)), dim3(fnt), 0, s, L, bv, it, ntile, tr);
            else hipLaunchKernelGGL(k_flush<false>, dim3(std::min(running * ntile, h->upd_grid)), dim3(fnt), lds, s, L, bv, it, ntile, tr);
            if (h->profile) { HIP_TRY(hipEventRecord(h->evpool[nev].second, s)); nev++; }
            hipLaunchKernelGGL(k_after_flush, dim3((running + 255) / 256), dim3(256), 0, s, bv, it);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h->status_h, h->status_d, B * sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        running = 0;
        for (int b = 0; b < B; b++) if (h->status_h[b] == ST_RUNNING) h->active_h[running++] = b;
        if (running) HIP_TRY(hipMemcpyAsync(h->active_d, h->active_h, running * sizeof(int), hipMemcpyHostToDevice, s));
        if (chunk < 16) chunk *= 2;
    }
    {   // tableau passes of this batch: sum of the work-list lengths of the rounds
        std::vector<int> nw(std::max(it, 1), 0);
        if (it > 0) HIP_TRY(hipMemcpy(nw.data(), h->nwork_d, (size_t)it * sizeof(int), hipMemcpyDeviceToHost));
        long passes = 0;
        for (int k = 0; k < it; k++) passes += nw[k];
        h->last_passes = passes;
    }
    if (L.rev) hipLaunchKernelGGL(k_rev_store_d, dim3((L.ld + 255) / 256, B), dim3(256), 0, s, L, bv, B);       // the reduced costs of every LP go to its slot
    if (bv.lazy) {
        // the slots keep what they hold; the reduced costs go to row M (the getters read them there), the rest waits for bslv_lpq_materialise
        hipLaunchKernelGGL(k_store_d, dim3((L.ld + 255) / 256, B), dim3(256), 0, s, L, bv, B);
        HIP_TRY(hipGetLastError());
        h->last_dst.assign(dst, dst + B);
        h->lazy_open = true;
        h->lazy_skipped += B - std::min<long>(B, h->last_passes);
    } else {   // tableaux of the solves that made no pivot
        const int cnt_slot = L.maxit + 40;
        hipLaunchKernelGGL(k_list_unpivoted, dim3((B + 255) / 256), dim3(256), 0, s, bv, B, cnt_slot);
        hipLaunchKernelGGL(k_copy_unpivoted, dim3(std::min(B * tiles, 2048)), dim3(NT), 0, s, L, bv, cnt_slot, tiles);
        HIP_TRY(hipGetLastError());
    }
    if (status) for (int b = 0; b < B; b++) status[b] = h->status_h[b] == ST_RUNNING ? BSLV_LP_UNDEFINED : h->status_h[b];
    {
        std::vector<int> itv(B);
        HIP_TRY(hipMemcpy(itv.data(), h->iters_d, B * sizeof(int), hipMemcpyDeviceToHost));
        long piv = 0;
        for (int b = 0; b < B; b++) piv += itv[b];
        h->last_pivots = piv;
        if (iters) memcpy(iters, itv.data(), B * sizeof(int));
    }
    h->last_iters = it;
    h->last_launches = it;
    { int xs[5]; HIP_TRY(hipMemcpy(xs, h->xstat_d, sizeof xs, hipMemcpyDeviceToHost)); for (int k = 0; k < 5; k++) h->last_ext[k] = xs[k]; }
    if (h->profile) {
        double ms = 0;
        for (size_t e = 0; e < nev; e++) { float t = 0; (void)hipEventElapsedTime(&t, h->evpool[e].first, h->evpool[e].second); ms += t; }
        h->last_update_ms = ms;
    }
    h->last_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (L.probe & 8) {
        unsigned long long dg[16];
        HIP_TRY(hipMemcpy(dg, h->dbg_d, sizeof dg, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(h->dbg_d, 0, sizeof dg));
        const double n = (double)std::max<unsigned long long>(dg[7], 1) * 100.0;       // ticks of 10 ns -> us per selection
        fprintf(stderr, "lp select phases (LP 0, %llu selections, us each): leaving row %.1f | tableau row %.1f | row scale + candidates %.1f | Harris pass %.1f | pivot choice %.1f | column %.1f | descriptor + vector updates %.1f\n",
                dg[7], dg[0] / n, dg[1] / n, dg[2] / n, dg[3] / n, dg[4] / n, dg[5] / n, dg[6] / n);
        if (L.rev) fprintf(stderr, "   tableau row with helpers: rho + release fence %.1f | request %.1f | own slices %.1f | wait for the others %.1f | acquire fence %.1f; slices taken by the LP's own workgroup %.2f of %d\n", dg[8] / n, dg[9] / n, dg[10] / n, dg[11] / n, dg[12] / n, dg[13] * 100.0 / n, (L.ld + 2047) / 2048);
    }
    {
        static const bool tm = getenv("BSLV_LP_TIMING") != nullptr;
        if (tm) fprintf(stderr, "lp solve_batch: %s form %d x %d, B %d, %d lock-step rounds, %ld pivots, %ld passes, %.1f ms\n", L.rev ? "revised" : "tableau", L.M, L.N, B, it, h->last_pivots, h->last_passes, h->last_total_ms);
    }
    return 0;
}

static int ensure_out(bslv_lpq *h, size_t n)
{
    if (n <= h->out_cap) return 0;
    if (h->out_d) (void)hipFree(h->out_d);
    h->out_d = nullptr; h->out_cap = 0;
    HIP_TRY(malloc0(&h->out_d, n * sizeof(double)));
    h->out_cap = n;
    return 0;
}

static int get_common(bslv_lpq *h, int B, const int *slot, int first, int cnt, int what, double *out)
{
    if (!h || B < 0 || cnt < 0 || !slot || !out || first < 0 || first + cnt > h->L.M + h->L.N) { set_error("bslv_lpq_get: bad argument"); return BSLV_E_ARG; }
    if (B == 0 || cnt == 0) return 0;
    for (int b = 0; b < B; b++) if (slot[b] < 0 || slot[b] >= h->slots) { set_error("bslv_lpq_get: bad slot"); return BSLV_E_ARG; }
    int rc;
    if ((rc = ensure_batch(h, B))) return rc;
    if ((rc = ensure_out(h, (size_t)B * cnt))) return rc;
    HIP_TRY(hipMemcpyAsync(h->qslot_d, slot, B * sizeof(int), hipMemcpyHostToDevice, h->stream));      // (not src_d: the batch arrays stay what the last solve left, bslv_lpq_materialise reads them)
    int n = B * cnt;
    hipLaunchKernelGGL(k_get, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->L, h->qslot_d, B, first, cnt, what, h->out_d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, h->out_d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

// values of the variables first .. first + cnt - 1 OF THE MODEL AS GIVEN.  A range without folded rows is one call into the
// engine; otherwise the folded rows are rebuilt from their columns: primal a_ij x_j; dual d_j / a_ij when the column sits on the
// bound the row gave it (the column's own reduced cost is then zero: it would be basic in the model as given), else 0.
static int get_mapped(bslv_lpq *h, int B, const int *slot, int first, int cnt, int what, double *out)
{
    if (!h) { set_error("bslv_lpq_get: bad argument"); return BSLV_E_ARG; }
    const bslv_lpq::Presolve &P = h->ps;
    if (first < 0 || cnt < 0 || first + cnt > P.M0 + P.N0) { set_error("bslv_lpq_get: bad argument"); return BSLV_E_ARG; }
    if (P.nfold == 0) return get_common(h, B, slot, first, cnt, what, out);
    if (B == 0 || cnt == 0) return 0;
    bool plain = true;
    for (int t = 0; t < cnt && plain; t++) { const int v = first + t; if (v < P.M0 && P.row_in[v] < 0) plain = false; }
    // (columns whose bound comes from a folded row need the split of their reduced cost as well)
    if (plain && what == 1) for (int t = 0; t < cnt && plain; t++) { const int v = first + t; if (v >= P.M0 && (P.lo_src[v - P.M0] >= 0 || P.up_src[v - P.M0] >= 0)) plain = false; }
    if (plain) {
        // contiguous in the engine's model too: rows keep their order, columns follow the rows that stay
        const int f2 = P.map_var(first);
        bool contiguous = true;
        for (int t = 0; t < cnt && contiguous; t++) if (P.map_var(first + t) != f2 + t) contiguous = false;
        if (contiguous) return get_common(h, B, slot, f2, cnt, what, out);
    }
    const int Mi = P.M0 - P.nfold, NV = Mi + P.N0;
    std::vector<double> prim((size_t)B * NV), dual;
    int rc;
    if ((rc = get_common(h, B, slot, 0, NV, 0, prim.data()))) return rc;
    if (what == 1) { dual.resize((size_t)B * NV); if ((rc = get_common(h, B, slot, 0, NV, 1, dual.data()))) return rc; }
    for (int b = 0; b < B; b++) {
        const double *x = &prim[(size_t)b * NV], *d = what == 1 ? &dual[(size_t)b * NV] : nullptr;
        // which bound a column sits on: the nearer one (a basic column has d = 0 and the answer does not matter)
        auto on_row_bound = [&](int j) -> int {            // folded row whose bound column j sits on, or -1
            if (!(d[Mi + j] != 0.0)) return -1;
            const double xl = std::fabs(x[Mi + j] - P.clo[j]), xu = std::fabs(x[Mi + j] - P.cup[j]);
            const bool at_lo = !(xu < xl);
            const int src = at_lo ? P.lo_src[j] : P.up_src[j];
            if (src < 0) return -1;
            // (the row's bound and the column's own may coincide: then the column keeps the reduced cost -- either split is a dual solution)
            const double own = at_lo ? P.lb0[P.M0 + j] : P.ub0[P.M0 + j], folded = at_lo ? P.clo[j] : P.cup[j];
            return own == folded ? -1 : src;
        };
        for (int t = 0; t < cnt; t++) {
            const int v = first + t;
            double val;
            if (v < P.M0 && P.row_in[v] >= 0) val = what == 0 ? x[P.row_in[v]] : d[P.row_in[v]];
            else if (v < P.M0) {
                const int j = P.fold_col[v];
                if (what == 0) val = P.fold_a[v] * x[Mi + j];
                else val = on_row_bound(j) == v ? d[Mi + j] / P.fold_a[v] : 0.0;
            } else {
                const int j = v - P.M0;
                if (what == 0) val = x[Mi + j];
                else val = on_row_bound(j) >= 0 ? 0.0 : d[Mi + j];
            }
            out[(size_t)b * cnt + t] = val;
        }
    }
    return 0;
}
int bslv_lpq_get_primal(bslv_lpq *h, int B, const int *slot, int first, int cnt, double *out) { return get_mapped(h, B, slot, first, cnt, 0, out); }
int bslv_lpq_get_dual(bslv_lpq *h, int B, const int *slot, int first, int cnt, double *out) { return get_mapped(h, B, slot, first, cnt, 1, out); }

int bslv_lpq_get_obj(bslv_lpq *h, int B, const int *slot, double *out)
{
    if (!h || B < 0 || !slot || !out) { set_error("bslv_lpq_get_obj: bad argument"); return BSLV_E_ARG; }
    if (B == 0) return 0;
    for (int b = 0; b < B; b++) if (slot[b] < 0 || slot[b] >= h->slots) { set_error("bslv_lpq_get_obj: bad slot"); return BSLV_E_ARG; }
    int rc;
    if ((rc = ensure_batch(h, B))) return rc;
    if ((rc = ensure_out(h, (size_t)B))) return rc;
    HIP_TRY(hipMemcpyAsync(h->qslot_d, slot, B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_get_obj, dim3((B + 255) / 256), dim3(256), 0, h->stream, h->L, h->qslot_d, B, h->c0, h->out_d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, h->out_d, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

long bslv_lpq_last_passes(const bslv_lpq *h) { return h ? h->last_passes : 0; }
long bslv_lpq_last_launches(const bslv_lpq *h) { return h ? h->last_launches : 0; }
long bslv_lpq_last_flip_updates(const bslv_lpq *h) { return h ? h->last_ext[4] : 0; }
// The extended selection for every LP of this engine from now on (on != 0) or only where a variable is boxed (0, the default
// below 1 GiB per tableau).  It changes the pivots taken, not the optimal value: used by the callers' retry when the plain
// dual simplex runs into its iteration limit on a degenerate LP.
int bslv_lpq_set_extended(bslv_lpq *h, int on)
{
    if (!h) { set_error("bslv_lpq_set_extended: bad argument"); return BSLV_E_ARG; }
    h->force_ext = on != 0;
    return 0;
}
int bslv_lpq_get_extended(const bslv_lpq *h) { return h && h->force_ext; }
int bslv_lpq_last_ext_stats(const bslv_lpq *h, long out[4])
{
    if (!h || !out) return BSLV_E_ARG;
    for (int k = 0; k < 4; k++) out[k] = h->last_ext[k];
    return 0;
}
int bslv_lpq_last_stats(const bslv_lpq *h, int *lockstep_iters, long *pivots, double *update_ms, double *total_ms)
{
    if (!h) return BSLV_E_ARG;
    if (lockstep_iters) *lockstep_iters = h->last_iters;
    if (pivots) *pivots = h->last_pivots;
    if (update_ms) *update_ms = h->last_update_ms;
    if (total_ms) *total_ms = h->last_total_ms;
    return 0;
}

}  // extern "C"
