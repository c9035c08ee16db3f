// poly_compat.hip -- the reference's poly__* symbols over the HIP polyhedron engine (include/bslv_poly_compat.h).
// Host code only: every call goes through the engine's C ABI (bslv_poly_*), the structs are a host mirror of its state.
#include "common.h"
#include "../../include/bslv_poly_compat.h"
#include <vector>
#include <map>
#include <algorithm>
#include <string>

namespace {

constexpr size_t BT = CHAR_BIT * sizeof(btstrg);
inline bool bit(const btstrg *l, size_t i) { return (l[i / BT] >> (i % BT)) & (btstrg)1; }
inline void setbit(btstrg *l, size_t i, bool v) { if (v) l[i / BT] |= (btstrg)1 << (i % BT); else l[i / BT] &= ~((btstrg)1 << (i % BT)); }

struct Side { size_t cap = 0; };           // slots the mirror arrays of one polytope hold
struct Shadow {
    bslv_poly *eng = nullptr;
    poly_args *args = nullptr;
    Side P, D;
    std::vector<btstrg> sltn_sent;         // primal sltn bits already handed to the engine
    bool apex_checked = false, lists_fresh = false;
    int nv_seen = 0;                       // primal slots whose coordinates the mirror already holds (refresh)
    long moved_seen = 0;                   // elements the snap band had moved at the last refresh: one more and every slot is fetched again
};
std::map<const void *, Shadow *> g_by_ptr;       // poly_args*, &args->primal, &args->dual  ->  shadow

[[noreturn]] void die(const char *what)
{
    fprintf(stderr, "bslv_poly_compat: %s: %s\n", what, bslv_last_error());
    exit(3);
}
Shadow *shadow_of(const void *p)
{
    auto it = g_by_ptr.find(p);
    if (it == g_by_ptr.end()) { fprintf(stderr, "bslv_poly_compat: unknown polytope %p (poly__initialise was not called on it)\n", p); exit(3); }
    return it->second;
}

// the built-in cone_polar (bslv_poly.c:30-39): set_default_args installs it, the engine's map 0
void compat_cone_polar(double *v, int is_dir, double *hp)
{
    // (never called: the engine applies its own copy; the address identifies the default)
    (void)v; (void)is_dir; (void)hp;
}

void grow_side(polytope *t, Side &s, size_t need)
{
    if (need <= s.cap) return;
    const size_t ncap = std::max(need + 64, 2 * s.cap), words0 = s.cap ? s.cap / BT + 1 : 0, words1 = ncap / BT + 1;
    t->data = (double *)realloc(t->data, ncap * t->dim * sizeof(double));
    if (t->dim_primg) {
        t->data_primg = (double *)realloc(t->data_primg, ncap * t->dim_primg * sizeof(double));
        memset(t->data_primg + s.cap * t->dim_primg, 0, (ncap - s.cap) * t->dim_primg * sizeof(double));
    }
    for (vrtx_strg **l : {&t->ideal, &t->used, &t->sltn}) {
        *l = (vrtx_strg *)realloc(*l, words1 * sizeof(btstrg));
        memset(*l + words0, 0, (words1 - words0) * sizeof(btstrg));
    }
    t->adjacence = (poly_list *)realloc(t->adjacence, ncap * sizeof(poly_list));
    t->incidence = (poly_list *)realloc(t->incidence, ncap * sizeof(poly_list));
    for (size_t i = s.cap; i < ncap; i++) { t->adjacence[i] = poly_list{0, 0, nullptr}; t->incidence[i] = poly_list{0, 0, nullptr}; }
    s.cap = ncap;
    t->blcks = ncap;
}

// engine -> mirror (coordinates, flags, counts of both sides)
void refresh(Shadow *S)
{
    poly_args *a = S->args;
    const int d = (int)a->dim, nv = bslv_poly_nprimal(S->eng), nf = bslv_poly_ndual(S->eng);
    grow_side(&a->primal, S->P, (size_t)std::max(nv, 1));        // (never empty: the bit sets and arrays exist from the start)
    grow_side(&a->dual, S->D, (size_t)std::max(nf, 1));
    std::vector<unsigned char> u(std::max(nv, nf) + 1), id(std::max(nv, nf) + 1), sl(nv + 1);
    // flags of every slot (a byte each), coordinates only of the slots that are new since the last call: a slot's coordinates do not
    // change (the whole array again on every poly__add_vrtx made the reference's driver quadratic in the number of slots)
    if (nv > 0 && bslv_poly_get_primal(S->eng, u.data(), id.data(), sl.data(), nullptr)) die("bslv_poly_get_primal");
    if (nv < S->nv_seen) S->nv_seen = 0;
    {   // ... unless the snap band moved one (bslv_poly_set_snap): then all of them again
        long moved = 0;
        if (bslv_poly_snapped(S->eng, &moved)) die("bslv_poly_snapped");
        if (moved != S->moved_seen) { S->nv_seen = 0; S->moved_seen = moved; }
    }
    if (nv > S->nv_seen && bslv_poly_get_primal_range(S->eng, S->nv_seen, nv - S->nv_seen, a->primal.data + (size_t)S->nv_seen * d)) die("bslv_poly_get_primal_range");
    S->nv_seen = nv;
    for (int i = 0; i < nv; i++) { setbit(a->primal.used, i, u[i]); setbit(a->primal.ideal, i, id[i]); setbit(a->primal.sltn, i, sl[i]); }
    a->primal.cnt = (size_t)nv;
    S->sltn_sent.assign(a->primal.sltn, a->primal.sltn + nv / BT + 1);
    if (bslv_poly_get_dual(S->eng, u.data(), id.data(), a->dual.data)) die("bslv_poly_get_dual");
    for (int f = 0; f < nf; f++) { setbit(a->dual.used, f, u[f]); setbit(a->dual.ideal, f, id[f]); }
    a->dual.cnt = (size_t)nf;
    (void)d;
    S->lists_fresh = false;
}

// mirror -> engine: what the caller wrote into the structs since the last call
void push_caller_writes(Shadow *S)
{
    poly_args *a = S->args;
    if (!S->apex_checked) {
        // cone_vertenum turns dual slot 0 into the apex (0,..,0), not ideal (bslv_algs.c:338-339)
        if (a->dual.cnt > 0 && !bit(a->dual.ideal, 0)) { if (bslv_poly_dual0_apex(S->eng)) die("bslv_poly_dual0_apex"); }
        S->apex_checked = true;
    }
    std::vector<int> marks;
    const size_t nv = a->primal.cnt;
    for (size_t i = 0; i < nv; i++)
        if (bit(a->primal.sltn, i) && !(i / BT < S->sltn_sent.size() && bit(S->sltn_sent.data(), i))) marks.push_back((int)i);
    if (!marks.empty()) {
        if (bslv_poly_mark(S->eng, (int)marks.size(), marks.data())) die("bslv_poly_mark");
        S->sltn_sent.resize(nv / BT + 1, 0);
        for (int i : marks) setbit(S->sltn_sent.data(), (size_t)i, true);
    }
}

void set_list(poly_list *l, const std::vector<size_t> &v)
{
    l->data = (size_t *)realloc(l->data, std::max<size_t>(v.size(), 1) * sizeof(size_t));
    if (!v.empty()) memcpy(l->data, v.data(), v.size() * sizeof(size_t));
    l->cnt = v.size(); l->blcks = v.size();
}
// adjacency and incidence lists of both sides (what the writers print)
void fill_lists(Shadow *S)
{
    if (S->lists_fresh) return;
    poly_args *a = S->args;
    const size_t nv = a->primal.cnt, nf = a->dual.cnt;
    std::vector<std::vector<size_t>> padj(nv), pinc(nv), dadj(nf), dinc(nf);
    const long ne = bslv_poly_nedges(S->eng), ni = bslv_poly_ninc(S->eng), nde = bslv_poly_ndual_edges(S->eng);
    std::vector<int> E(2 * ne + 2), I(2 * ni + 2), DE(2 * nde + 2);
    if (bslv_poly_get_edges(S->eng, E.data()) || bslv_poly_get_inc(S->eng, I.data()) || bslv_poly_get_dual_edges(S->eng, DE.data())) die("bslv_poly_get_edges / _inc / _dual_edges");
    for (long e = 0; e < ne; e++) { padj[E[2 * e]].push_back((size_t)E[2 * e + 1]); padj[E[2 * e + 1]].push_back((size_t)E[2 * e]); }
    for (long k = 0; k < ni; k++) { const int v = I[2 * k], f = I[2 * k + 1]; if (bit(a->dual.used, (size_t)f)) { pinc[v].push_back((size_t)f); dinc[f].push_back((size_t)v); } }
    for (long e = 0; e < nde; e++) { dadj[DE[2 * e]].push_back((size_t)DE[2 * e + 1]); dadj[DE[2 * e + 1]].push_back((size_t)DE[2 * e]); }
    for (size_t i = 0; i < nv; i++) { set_list(&a->primal.adjacence[i], padj[i]); set_list(&a->primal.incidence[i], pinc[i]); }
    for (size_t f = 0; f < nf; f++) { set_list(&a->dual.adjacence[f], dadj[f]); set_list(&a->dual.incidence[f], dinc[f]); }
    S->lists_fresh = true;
}

// which of the engine's V->H maps is this callback, and with which parameter c?
int identify_callback(void (*fn)(double *, int, double *), int d, std::vector<double> &c)
{
    c.assign(d, 0.0);
    if (!fn || fn == compat_cone_polar) return 0;
    std::vector<double> v(d, 0.0), hp(d + 1, 0.0), hp2(d + 1, 0.0);
    fn(v.data(), 0, hp.data());
    const bool zero_head = std::all_of(hp.begin(), hp.begin() + d - 1, [](double x) { return x == 0.0; });
    if (zero_head && hp[d - 1] == 0.0 && hp[d] == -1.0) {                       // cone_polar: (v | -1)
        v[0] = 1.0; fn(v.data(), 0, hp2.data());
        if (hp2[0] == 1.0) return 0;
    } else if (zero_head && hp[d - 1] == 1.0 && hp[d] == 0.0) {                 // lowerV2upperH: (v_1..v_{q-1}, 1 - sum c_j v_j | v_q)
        for (int j = 0; j < d - 1; j++) { std::fill(v.begin(), v.end(), 0.0); v[j] = 1.0; fn(v.data(), 0, hp2.data()); c[j] = 1.0 - hp2[d - 1]; }
        return 1;
    } else if (zero_head && hp[d - 1] == -1.0 && hp[d] == 0.0) {                // upperV2lowerH: (v_j - v_q c_j, -1 | -v_q)
        std::fill(v.begin(), v.end(), 0.0); v[d - 1] = 1.0; fn(v.data(), 0, hp2.data());
        for (int j = 0; j < d - 1; j++) c[j] = -hp2[j];
        return 2;
    }
    fprintf(stderr, "bslv_poly_compat: the dualV2primalH callback is none of cone_polar / lowerV2upperH / upperV2lowerH (plot transforms are not supported)\n");
    exit(3);
}

FILE *open_out(const char *fname) { FILE *f = fname ? fopen(fname, "w") : stdout; if (!f) { fprintf(stderr, "bslv_poly_compat: cannot open %s\n", fname); exit(3); } return f; }
// one output row: the items separated by what the format string itself carries (the reference's formats end in a blank,
// which it removes again at the end of the row with fseek; here the last item is printed without its trailing blank)
void put_row(FILE *f, const std::vector<std::string> &items)
{
    for (size_t k = 0; k < items.size(); k++) {
        std::string t = items[k];
        if (k + 1 == items.size()) while (!t.empty() && t.back() == ' ') t.pop_back();
        fputs(t.c_str(), f);
    }
    fputc('\n', f);
}
std::string fmt_double(const char *frmt, double x) { char b[128]; snprintf(b, sizeof b, frmt ? frmt : "%g ", x); return b; }
std::string fmt_index(const char *frmt, unsigned v) { char b[64]; snprintf(b, sizeof b, frmt ? frmt : "%u ", v); return b; }

}  // namespace

extern "C" {

void poly__set_default_args(poly_args *args, size_t dim)
{
    memset(args, 0, sizeof *args);
    args->dim = dim;
    args->eps = 1e-9;                                           // (set and never read in the reference either: POLY_EPS is the tolerance)
    args->dualV2primalH = (void (*)())compat_cone_polar;
}

void poly__initialise(poly_args *args)
{
    const int d = (int)args->dim;
    Shadow *S = new Shadow();
    S->args = args;
    std::vector<double> c;
    const int mode = identify_callback((void (*)(double *, int, double *))args->dualV2primalH, d, c);
    if (bslv_poly_create(&S->eng, d, mode, c.data())) die("bslv_poly_create");
    // poly__add_vrtx hands in one cut at a time, so the projection sub-band of poly__cut (bslv_poly.c:666-674) costs nothing here: on,
    // and the reference's driver sees the coordinates its own engine would have left (BSLV_POLY_SNAP=0 switches it off)
    { const char *e = getenv("BSLV_POLY_SNAP"); if (!(e && atoi(e) == 0) && bslv_poly_set_snap(S->eng, 1)) die("bslv_poly_set_snap"); }
    args->val = (double *)calloc((size_t)d, sizeof(double));
    args->val_primg_prml = args->dim_primg_prml ? (double *)calloc(args->dim_primg_prml, sizeof(double)) : nullptr;
    args->val_primg_dl = args->dim_primg_dl ? (double *)calloc(args->dim_primg_dl, sizeof(double)) : nullptr;
    for (polytope *t : {&args->primal, &args->dual}) {
        memset(t, 0, sizeof *t);
        t->dim = (size_t)d;
        t->v2h = nullptr;
    }
    args->primal.dim_primg = args->dim_primg_prml; args->dual.dim_primg = args->dim_primg_dl;
    args->primal.dual = &args->dual; args->dual.dual = &args->primal;
    args->init_data.intlsd = 0;
    g_by_ptr[args] = S; g_by_ptr[&args->primal] = S; g_by_ptr[&args->dual] = S;
    refresh(S);                                                 // dual slot 0: the facet at infinity (bslv_poly.c:83-92)
}

void poly__kill(poly_args *args)
{
    auto it = g_by_ptr.find(args);
    if (it == g_by_ptr.end()) return;
    Shadow *S = it->second;
    bslv_poly_destroy(S->eng);
    for (polytope *t : {&args->primal, &args->dual}) {
        const size_t cap = t == &args->primal ? S->P.cap : S->D.cap;
        for (size_t i = 0; i < cap; i++) { free(t->adjacence[i].data); free(t->incidence[i].data); }
        free(t->adjacence); free(t->incidence); free(t->data); free(t->data_primg); free(t->ideal); free(t->used); free(t->sltn);
        memset(t, 0, sizeof *t);
    }
    free(args->val); free(args->val_primg_prml); free(args->val_primg_dl);
    args->val = args->val_primg_prml = args->val_primg_dl = nullptr;
    g_by_ptr.erase(args); g_by_ptr.erase(&args->primal); g_by_ptr.erase(&args->dual);
    delete S;
}

int poly__add_vrtx(poly_args *args)
{
    Shadow *S = shadow_of(args);
    push_caller_writes(S);
    int rc = 0;
    if (bslv_poly_add(S->eng, args->val, (int)args->ideal, &rc)) die("bslv_poly_add");
    const int f = bslv_poly_ndual(S->eng) - 1;                  // the dual slot this call created
    refresh(S);
    if (args->dual.dim_primg && args->val_primg_dl) memcpy(args->dual.data_primg + (size_t)f * args->dual.dim_primg, args->val_primg_dl, args->dual.dim_primg * sizeof(double));
    return rc;                                                  // EXIT_SUCCESS / EXIT_FAILURE (redundant: no primal vertex violates)
}

int poly__intl_apprx(poly_args *args)
{
    Shadow *S = shadow_of(args);
    push_caller_writes(S);
    const size_t nf0 = args->dual.cnt;
    int rc = 0;
    if (bslv_poly_init(S->eng, &rc)) die("bslv_poly_init");
    refresh(S);
    if (rc == 0) {
        args->init_data.intlsd = 1;
        // the halfspaces that were not chosen for the initial cone are added again as NEW dual slots (bslv_poly.c:190-197):
        // their pre-images go with them
        const size_t q = args->dim, dp = args->dual.dim_primg;
        for (size_t f = nf0; dp && f < args->dual.cnt; f++)
            for (size_t g = 0; g < nf0; g++)
                if (!memcmp(args->dual.data + f * q, args->dual.data + g * q, q * sizeof(double)) && bit(args->dual.ideal, f) == bit(args->dual.ideal, g)) {
                    memcpy(args->dual.data_primg + f * dp, args->dual.data_primg + g * dp, dp * sizeof(double));
                    break;
                }
    }
    return rc;
}

int poly__get_vrtx(poly_args *args)
{
    Shadow *S = shadow_of(args);
    push_caller_writes(S);
    int ideal = 0, idx = 0, rc = 0;
    if (bslv_poly_next(S->eng, args->val, &ideal, &idx, &rc)) die("bslv_poly_next");
    if (rc == 0) { args->ideal = ideal ? 1u : 0u; args->idx = (size_t)idx; }
    return rc;                                                  // EXIT_FAILURE: no vertex left
}

void poly__update_adjacence(polytope *poly)
{
    Shadow *S = shadow_of(poly);
    if (poly != &S->args->dual) { fprintf(stderr, "bslv_poly_compat: poly__update_adjacence is implemented for the dual side (its only use, bslv_algs.c:398,1144,1569)\n"); exit(3); }
    push_caller_writes(S);
    if (bslv_poly_dual_adjacency(S->eng)) die("bslv_poly_dual_adjacency");
    S->lists_fresh = false;
    fill_lists(S);
}

void poly__initialise_permutation(polytope *poly, permutation *prm)
{
    fill_lists(shadow_of(poly));
    prm->cnt = 0;
    prm->data = (size_t *)malloc(std::max<size_t>(poly->cnt, 1) * sizeof(size_t));
    prm->inv = (size_t *)malloc(std::max<size_t>(poly->cnt, 1) * sizeof(size_t));
    for (size_t i = 0; i < poly->cnt; i++)
        if (bit(poly->used, i)) { prm->inv[i] = prm->cnt; prm->data[prm->cnt++] = i; }
}
void poly__kill_permutation(permutation *prm) { free(prm->data); free(prm->inv); prm->data = prm->inv = nullptr; }

void poly__vrtx2file(polytope *poly, permutation *prm, const char *fname, const char *frmt)
{
    FILE *f = open_out(fname);
    for (size_t k = 0; k < prm->cnt; k++) {
        const size_t i = prm->data[k];
        std::vector<std::string> row{bit(poly->ideal, i) ? "0 " : "1 "};
        for (size_t j = 0; j < poly->dim; j++) row.push_back(fmt_double(frmt, poly->data[i * poly->dim + j]));
        put_row(f, row);
    }
    if (fname) fclose(f);
}
void poly__primg2file(polytope *poly, permutation *prm, const char *fname, const char *frmt)
{
    FILE *f = open_out(fname);
    for (size_t k = 0; k < prm->cnt; k++) {
        const size_t i = prm->data[k];
        if (!bit(poly->sltn, i)) continue;
        std::vector<std::string> row;
        for (size_t j = 0; j < poly->dim_primg; j++) row.push_back(fmt_double(frmt, poly->data_primg[i * poly->dim_primg + j]));
        put_row(f, row);
    }
    if (fname) fclose(f);
}
void poly__adj2file(polytope *poly, permutation *prm, const char *fname, const char *frmt)
{
    FILE *f = open_out(fname);
    for (size_t k = 0; k < prm->cnt; k++) {
        const poly_list &l = poly->adjacence[prm->data[k]];
        std::vector<std::string> row;
        for (size_t t = 0; t < l.cnt; t++) row.push_back(fmt_index(frmt, (unsigned)prm->inv[l.data[t]]));
        put_row(f, row);
    }
    if (fname) fclose(f);
}
void poly__inc2file(polytope *poly, permutation *prm, permutation *prm_dual, const char *fname, const char *frmt)
{
    FILE *f = open_out(fname);
    for (size_t k = 0; k < prm_dual->cnt; k++) {
        const poly_list &l = poly->dual->incidence[prm_dual->data[k]];
        std::vector<std::string> row;
        for (size_t t = 0; t < l.cnt; t++) row.push_back(fmt_index(frmt, (unsigned)prm->inv[l.data[t]]));
        put_row(f, row);
    }
    if (fname) fclose(f);
}

void poly__swap(poly_args *a, poly_args *b) { (void)a; (void)b; fprintf(stderr, "bslv_poly_compat: poly__swap serves the plot option only, which is not supported\n"); exit(3); }
void poly__plot(polytope *poly, const char *fname) { (void)poly; (void)fname; fprintf(stderr, "bslv_poly_compat: poly__plot (option -p) is not supported\n"); exit(3); }
void poly__polyck(poly_args *args) { (void)args; }

}  // extern "C"
