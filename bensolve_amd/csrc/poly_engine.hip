// poly_engine.hip -- double-description vertex enumeration on the GPU (gfx950).
//
// Replaces the polyhedron half of the hot path: poly__add_vrtx / poly__cut / edge_test /
// poly__intl_apprx / poly__get_vrtx / poly__update_adjacence (bslv_poly.c:104-226, 467-512,
// 562-787, 992-1010).  One cut = four data-parallel passes with prefix-sum ordering (the same
// definition as oracle/poly_dd.c, which is checked set-wise against the compiled reference):
//   K1  k_classify      every live element against hp.y >= alpha  -> PLUS / ZERO / MINUS
//   E   k_edge_flags / k_edge_emit   MINUS-PLUS edges create vertices, survivors are compacted
//   Z   k_vert_flags / k_vert_emit   ZERO elements join the new facet (incidence rebuilt)
//   K2  k_pair_flags / k_pair_emit   adjacency prune over all pairs of the new facet:
//       sorted-list intersection prefilter (|inc_i & inc_j| >= d-1) per lane, then a wave-
//       cooperative superset scan (ballot over the facet's members) for the surviving pairs.
// HBM layout: coordinates SoA X[k*cap + i] (coalesced K1 loads), flags 1 B/element, incidence as
// sorted facet-id lists in one pool (inc_off/inc_len), edges as int2 pairs (ping-pong buffers).
#include "common.h"
#include <vector>
#include <chrono>
#include <algorithm>
#include <cmath>
#include <thread>
#include <atomic>
#include <cstddef>

namespace bslv {

constexpr int MAXD = 16;
constexpr double POLY_EPS = 1e-9;      // bslv_poly.h:47
constexpr int PB = 256;                // threads per workgroup in the poly kernels
constexpr unsigned char F_USED = 1, F_IDEAL = 2, F_SLTN = 4;
constexpr int CRING = 1024;            // ring of per-cut classify counters
constexpr int CSTRIDE = 8;             // ints per ring slot: #MINUS, #ZERO, list bound of the ZERO elements, #long ZERO elements, ticket of k_flags2
constexpr int LCAP = 16;
constexpr int LONGN = 64;      // lists longer than this are processed by a whole wave
constexpr int ZMAX = 8;        // on-plane elements with long lists that get a facet-stamp row per cut (see ZMarks)

struct Hp { double h[MAXD + 1]; };
struct Tri { int a, b, c; };

struct PolyView {
    int d, cap;
    double *X;              // d x cap
    unsigned char *flag;    // cap
    signed char *cls;       // cap
    unsigned *inc_off;      // cap
    int *inc_len;           // cap
    int *pool;              // poolcap
    unsigned char *keep;    // poolcap (all zero between cuts)
    // Long lists (extreme directions: tens of thousands of facets on covering problems) are rebuilt WITH SLACK, and while the slack lasts
    // an on-plane element's list is compacted and extended where it lies instead of being copied to the end of the pool -- the copy of
    // a 50 000-entry list by one wave was 150 us of every round.  capx[i] = (offset of the list the room belongs to) << 32 | room;
    // valid only while inc_off[i] still is that offset (zero: none).
    unsigned long long *capx;   // cap
    // hot mode (a chunk of cuts whose batched classification is known): the per-cut element passes only visit the
    // elements some cut of the chunk does not leave strictly inside (hv, ascending ids) and those created since
    // the chunk began (ids >= nv_base); everything else is PLUS for every cut of the chunk.  hv == nullptr: all.
    const int *hv;
    int nhv, nv_base;
    // hot mode, membership bitmaps of the elements with long incidence lists (extreme directions): lslot[v] = row of
    // lbits (lstride words, one bit per facet rank) or -1.  An edge to such an element then costs one independent
    // load per facet of its short end instead of a binary search (a chain of ~10 dependent loads) per facet.
    int *lslot;
    unsigned *lbits;
    int lstride;
};
// pool entries the rebuilt list of on-plane element i needs when k facets are appended: 0 = it stays where it is (zero_room_inplace)
__device__ __forceinline__ bool zero_room_inplace(const PolyView &P, int i, int n, int k)
{
    if (n <= 64) return false;                                  // (LONGN: only lists that take the wave-cooperative path)
    const unsigned long long x = P.capx[i];
    return (unsigned)(x >> 32) == P.inc_off[i] && (long long)(x & 0xFFFFFFFFull) >= (long long)n + k;
}
__device__ __forceinline__ int zero_room(const PolyView &P, int i, int k)
{
    const int n = P.inc_len[i];
    if (n <= 64) return n + k;
    if (zero_room_inplace(P, i, n, k)) return 0;
    return n + k + (n >> 2 > 64 ? n >> 2 : 64);
}
__device__ __host__ inline int vm_count(const PolyView &P, int nv) { return P.hv ? P.nhv + (nv - P.nv_base) : nv; }
__device__ __forceinline__ int vm_id(const PolyView &P, int idx) { return P.hv ? (idx < P.nhv ? P.hv[idx] : P.nv_base + (idx - P.nhv)) : idx; }

// ---------------- scans ----------------
__device__ __forceinline__ Tri tri_add(Tri x, Tri y) { return Tri{x.a + y.a, x.b + y.b, x.c + y.c}; }
__device__ __forceinline__ Tri tri_shfl_up(Tri x, int o)
{
    return Tri{__shfl_up(x.a, o, WAVE), __shfl_up(x.b, o, WAVE), __shfl_up(x.c, o, WAVE)};
}
// exclusive scan over the workgroup (blockDim.x multiple of 64, <= 1024); total returned in *tot
__device__ Tri block_exscan(Tri v, Tri *tot, Tri *lds /* >= 16 */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    Tri inc = v;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        Tri y = tri_shfl_up(inc, o);
        if (lane >= o) inc = tri_add(inc, y);
    }
    __syncthreads();
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    Tri base{0, 0, 0}, total{0, 0, 0};
    for (int w = 0; w < nw; w++) {
        if (w < wave) base = tri_add(base, lds[w]);
        total = tri_add(total, lds[w]);
    }
    *tot = total;
    Tri ex = tri_add(base, inc);
    ex.a -= v.a; ex.b -= v.b; ex.c -= v.c;
    return ex;
}
// sum of a Tri over the workgroup (result in every thread)
__device__ __forceinline__ Tri block_sum(Tri v, Tri *lds /* >= 16 */)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { v.a += __shfl_xor(v.a, o, WAVE); v.b += __shfl_xor(v.b, o, WAVE); v.c += __shfl_xor(v.c, o, WAVE); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    Tri r{0, 0, 0};
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) r = tri_add(r, lds[w]);
    __syncthreads();
    return r;
}
// scan of per-block sums by one workgroup; sums[] becomes exclusive prefixes, totals[0] the grand total
// host mailbox in mapped pinned memory: the scan tail publishes totals (and the classify counters)
// straight to the host, which spins on `seq` instead of paying a stream synchronise per readback
// One mailbox = ONE 64-byte line of mapped pinned memory (48 bytes of payload): the fields are stored, a system fence, then seq.
// (A state spanning several lines was once read with the new seq and the old content: poly_rounds2_kernels.inc, RState.)
struct alignas(64) Mail { volatile int seq; int cnt[4]; Tri t; Tri k2t; int k2seq; unsigned chk, chk2; };    // k2t/k2seq: result of the prune that was in flight, forwarded by round A
// The fields are stored, a system fence, then seq -- and still the host was once seen to read a new seq with the old content
// of a mailbox (RState, poly_rounds2_kernels.inc).  So the content carries checksums the reader verifies: chk over (seq, cnt, t),
// written by every publisher, and chk2 over (k2t, k2seq), which only round A's publisher writes -- left alone, that triple stays
// the consistent one of an earlier publish, which the reader then simply does not match with the prune it waits for.
__host__ __device__ __forceinline__ unsigned mail_sum(const int *cnt, const Tri &t, int seq)
{
    unsigned c = 0x9E3779B9u;
    for (int k = 0; k < 4; k++) c = c * 31u + (unsigned)cnt[k];
    c = c * 31u + (unsigned)t.a; c = c * 31u + (unsigned)t.b; c = c * 31u + (unsigned)t.c;
    return c + (unsigned)seq * 2654435761u;
}
__host__ __device__ __forceinline__ unsigned mail_sum2(const Tri &k2t, int k2seq)
{
    unsigned c = 0x85EBCA6Bu;
    c = c * 31u + (unsigned)k2t.a; c = c * 31u + (unsigned)k2t.b; c = c * 31u + (unsigned)k2t.c;
    return c * 31u + (unsigned)k2seq;
}
// one thread: cnt (4 ints, or nullptr = zeros) and t, then the fence, then seq
__device__ __forceinline__ void mail_publish(Mail *mail, const int *cnt, const Tri &t, int seq)
{
    int c4[4];
    for (int k = 0; k < 4; k++) { c4[k] = cnt ? cnt[k] : 0; mail->cnt[k] = c4[k]; }
    mail->t = t;
    mail->chk = mail_sum(c4, t, seq);
    __threadfence_system();
    mail->seq = seq;
}
// host: one attempt to take the mailbox content for `seq`.  true: *out is a consistent copy; false: not there yet, or (torn
// counted) seq was there before its content
static inline bool mail_try_read(const volatile Mail *m, int seq, Mail *out, long *torn)
{
    if (m->seq != seq) return false;
    __sync_synchronize();
    Mail c;
    memcpy((void *)&c, (const void *)m, sizeof c);
    if (c.seq == seq && c.chk == mail_sum(c.cnt, c.t, seq) && c.chk2 == mail_sum2(c.k2t, c.k2seq)) { *out = c; return true; }
    if (torn) ++*torn;
    return false;
}
__global__ __launch_bounds__(1024) void k_scan_blocks(Tri *sums, int nb, Tri *totals, Mail *mail = nullptr, const int *counters = nullptr, int seq = 0)
{
    __shared__ Tri lds[16];
    Tri carry{0, 0, 0};
    for (int base = 0; base < nb; base += 1024) {
        int i = base + threadIdx.x;
        Tri v = i < nb ? sums[i] : Tri{0, 0, 0};
        Tri tot;
        Tri ex = block_exscan(v, &tot, lds);
        if (i < nb) sums[i] = tri_add(ex, carry);
        carry = tri_add(carry, tot);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        totals[0] = carry;
        if (mail) mail_publish(mail, counters, carry, seq);
    }
}

// large arrays of block sums (a new facet with 10^4-10^5 elements has 10^6-10^7 pair blocks; one workgroup walks them in
// milliseconds): chunks of 1024 sums are scanned by one workgroup each, the chunk totals by k_scan_blocks, and the chunk
// prefixes added back
__global__ __launch_bounds__(1024) void k_scan_chunks(Tri *sums, int nb, Tri *aux)
{
    __shared__ Tri lds[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    Tri v = i < nb ? sums[i] : Tri{0, 0, 0};
    Tri tot;
    Tri ex = block_exscan(v, &tot, lds);
    if (i < nb) sums[i] = ex;
    if (threadIdx.x == 0) aux[blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void k_scan_add(Tri *sums, int nb, const Tri *aux)
{
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < nb) sums[i] = tri_add(sums[i], aux[blockIdx.x]);
}

// ---------------- K1: classify ----------------
__device__ __forceinline__ signed char classify_one(const PolyView &P, const Hp &hp, int i, unsigned char fl)
{
    double s = 0.0;
#pragma unroll 4
    for (int k = 0; k < P.d; k++) s = fma(hp.h[k], P.X[(size_t)k * P.cap + i], s);
    double a = (fl & F_IDEAL) ? 0.0 : hp.h[P.d];
    return (s > a + POLY_EPS) ? 1 : (s > a - POLY_EPS ? 0 : -1);
}
// counters[0] = #MINUS, counters[1] = #ZERO
// Keep marks of on-plane elements with LONG incidence lists (the extreme directions of an upper image: hundreds
// to thousands of facets, and a thousand edges each).  Marking "facet shared with a PLUS neighbour" per edge in
// the element's own keep[] bytes sends thousands of requests to the same few cache lines, which the L2 serves
// one at a time (measured 40-100 us per cut).  Instead each such element z gets a row of facet stamps:
// an edge (z, p) stamps row[g] for every facet g of its PLUS end p -- spread over all facets, no searching --
// and z keeps the facets of its list whose stamp is current.  Same rule as bslv_poly.c:634-652.
struct ZMarks {
    const int *zlist;      // up to ZMAX long on-plane elements of this cut (k_classify), count in counters[3]
    int *rows;             // ZMAX x stride stamps
    int stride, stamp;
};
__device__ __forceinline__ int zmarks_find(const ZMarks &Z, const int *counters, int v)
{
    const int nz = counters[3] < ZMAX ? counters[3] : ZMAX;
    for (int k = 0; k < nz; k++) if (Z.zlist[k] == v) return k;
    return -1;
}
// classification of element number idx of the (hot) element map; whole waves call it
__device__ __forceinline__ void classify_body(const PolyView &P, const Hp &hp, int nv, int *counters, int *zlist, int idx)
{
    int isminus = 0, iszero = 0, zlen = 0;
    if (idx < vm_count(P, nv)) {
        const int i = vm_id(P, idx);
        unsigned char fl = P.flag[i];
        signed char c = 2;
        if (fl & F_USED) { c = classify_one(P, hp, i, fl); isminus = c < 0; iszero = c == 0; }
        if (iszero) {
            zlen = zero_room(P, i, 1);             // room of its rebuilt incidence list (0: a long list with slack is rebuilt where it lies)
            if (P.inc_len[i] > LONGN) { const int k = atomicAdd(&counters[3], 1); if (k < ZMAX) zlist[k] = i; }
        }
        P.cls[i] = c;
    }
    unsigned long long bm = __ballot(isminus), bz = __ballot(iszero);
    if (bz) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) zlen += __shfl_xor(zlen, o, WAVE);
    }
    if ((threadIdx.x & 63) == 0) {
        if (bm) atomicAdd(&counters[0], __popcll(bm));
        if (bz) { atomicAdd(&counters[1], __popcll(bz)); atomicAdd(&counters[2], zlen); }
    }
}
__global__ __launch_bounds__(PB) void k_classify(PolyView P, Hp hp, int nv, int *counters, int *zlist, const int *nv_dev = nullptr)
{
    if (nv_dev) nv = *nv_dev;          // queued before the host knew how many elements the previous cut adds
    classify_body(P, hp, nv, counters, zlist, blockIdx.x * PB + threadIdx.x);
}

// The projection sub-band of poly__cut (bslv_poly.c:666-674; bslv_poly_set_snap): an on-plane element that lies more than 1e-2 POLY_EPS
// ABOVE the hyperplane of a cut that removes something is moved onto it, x -= (s - a) hp / |hp|^2, before it is treated as lying on it.
// Runs behind k_classify of the same cut (classes and counters[0] = #MINUS are final), in front of everything that reads coordinates.
// nn = |hp|^2 summed on the host in the reference's order; products and differences rounded separately as the reference's are.
__global__ __launch_bounds__(PB) void k_snap(PolyView P, Hp hp, double nn, int nv, const int *counters, unsigned long long *snapped)
{
    if (counters[0] == 0) return;                       // a redundant cut walks nothing (bslv_poly.c:130-136)
    const int idx = blockIdx.x * PB + threadIdx.x;
    if (idx >= vm_count(P, nv)) return;
    const int i = vm_id(P, idx);
    if (!(P.flag[i] & F_USED) || P.cls[i] != 0) return;
    double s = 0.0;
    for (int k = 0; k < P.d; k++) s = fma(hp.h[k], P.X[(size_t)k * P.cap + i], s);
    const double a = (P.flag[i] & F_IDEAL) ? 0.0 : hp.h[P.d];
    if (!(s > a + 1.0e-2 * POLY_EPS)) return;
    const double mu = __ddiv_rn(__dsub_rn(s, a), nn);
    for (int k = 0; k < P.d; k++) P.X[(size_t)k * P.cap + i] = __dsub_rn(P.X[(size_t)k * P.cap + i], __dmul_rn(mu, hp.h[k]));
    atomicAdd(snapped, 1ull);
}

// Batched incidence kernel: classes of nv elements against B halfspaces, 2 bits
// each (0 dead, 1 MINUS, 2 ZERO, 3 PLUS), 32 halfspaces per 64-bit word, out[w*cap + i];
// anyminus[w] gets bit bb set iff some live element violates halfspace 32 w + bb.  Algorithmic bytes:
// 8 d nv (coords) + nv (flags) + 8 (d+1) B (halfspaces) + nv B / 4 (classes).
// hps holds (d+3) doubles per halfspace: normal, alpha, alpha+EPS, alpha-EPS (thresholds precomputed on the host:
// the scalar unit has no fp64 add).  The halfspace coefficients are wave-uniform: they are read through scalar loads (no LDS, no VGPR
// traffic) and enter the fp64 FMAs as scalar operands; coordinates stay in registers (D is a
// template parameter).  `touch`/`first` (optional) = number of non-PLUS halfspaces per element and the
// first of them, for the multi-cut path.
// spread the 32 bits of x to the even bit positions of a 64-bit word
__device__ __forceinline__ unsigned long long spread32(unsigned x)
{
    unsigned long long v = x;
    v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
    v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}
__device__ __forceinline__ unsigned compress32(unsigned long long v)      // inverse of spread32: the even bits of v
{
    v &= 0x5555555555555555ull;
    v = (v | (v >> 1)) & 0x3333333333333333ull;
    v = (v | (v >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v >> 4)) & 0x00FF00FF00FF00FFull;
    v = (v | (v >> 8)) & 0x0000FFFF0000FFFFull;
    v = (v | (v >> 16)) & 0x00000000FFFFFFFFull;
    return (unsigned)v;
}
// The element-major words of the conflict-matrix rounds (RoundsV2::hw) hold the 32 cuts of a class word as two masks -- low half: the
// cuts the element is not PLUS for (MINUS or ZERO), high half: the cuts it is MINUS for -- not as 32 two-bit classes: every kernel of a
// round asks for these two masks, k_r2_minit several times per edge, and pulling the even bits out of a 64-bit word is ~30 instructions
// each time (the kernel is bound by the instructions it issues).  r2_pack: from the two-bit classes (01 MINUS, 10 ZERO, 11 PLUS).
__device__ __forceinline__ unsigned long long r2_pack(unsigned long long raw) { return (unsigned long long)compress32(raw ^ (raw >> 1)) | ((unsigned long long)compress32(raw & ~(raw >> 1)) << 32); }
__device__ __forceinline__ unsigned r2_touch32(unsigned long long packed) { return (unsigned)packed; }
__device__ __forceinline__ unsigned r2_minus32(unsigned long long packed) { return (unsigned)(packed >> 32); }
// ---- owners of on-plane elements in a round of independent cuts (poly_rounds2_kernels.inc, "Elements shared by the cuts of a round") ----
// What the flags / emit passes and the prunes need to know about the round: the element-major class words of the chunk (hw, wm:
// which words hold a touch), the selected cuts as a bit mask (selw) and their number in the selection (selmap), the owner code of
// every element (cutof: s >= 0 one cut, -1 - k: ZERO for k >= 2 selected cuts).  cutof == nullptr: the one-cut pipeline.
struct R2Own { const int *cutof; const unsigned long long *hw; const unsigned *wm; const unsigned *selw; const int *selmap; int nw; };
// owners of element i among the selected cuts, ascending: f(o) for every owner o (index in the chunk)
template <class F>
__device__ __forceinline__ void r2_for_owners(const unsigned long long *__restrict__ hw, const unsigned *__restrict__ wm, int nw, const unsigned *selw, int i, F f)
{
    unsigned m = wm[i];
    while (m) {
        const int w = __ffs((int)m) - 1;
        m &= m - 1;
        if (!selw[w]) continue;
        unsigned t = r2_touch32(hw[(size_t)i * nw + w]) & selw[w];
        while (t) { const int bit = __ffs((int)t) - 1; t &= t - 1; f(w * 32 + bit); }
    }
}
// do two ZERO elements have a selected cut in common?
__device__ __forceinline__ bool r2_common_owner(const unsigned long long *__restrict__ hw, const unsigned *__restrict__ wm, int nw, const unsigned *selw, int x, int y)
{
    unsigned m = wm[x] & wm[y];
    while (m) {
        const int w = __ffs((int)m) - 1;
        m &= m - 1;
        if (selw[w] && (r2_touch32(hw[(size_t)x * nw + w]) & r2_touch32(hw[(size_t)y * nw + w]) & selw[w])) return true;
    }
    return false;
}
// acc = 2 acc + (s > t), t wave-uniform: the compare leaves its mask in VCC and ONE add-with-carry shifts the bit in (the compiler
// builds the word from v_cndmask + v_or3 + shifts: two vector instructions per bit instead of one)
__device__ __forceinline__ void shift_in_gt(unsigned &acc, double s, double t)
{
    asm("v_cmp_lt_f64 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(s), "s"(t) : "vcc");
}
// one element per lane: coordinates and flag byte in registers (classify_load), then its classes against the B halfspaces
template <int D>
__device__ __forceinline__ void classify_load(const PolyView &P, int nv, int i, double (&x)[D > 0 ? D : MAXD], unsigned char &fl)
{
    const int d = D > 0 ? D : P.d;
    fl = 0;
#pragma unroll
    for (int k = 0; k < (D > 0 ? D : MAXD); k++) x[k] = 0.0;
    if (i < nv) {
        fl = P.flag[i];
#pragma unroll
        for (int k = 0; k < (D > 0 ? D : MAXD); k++) if (k < d) x[k] = P.X[(size_t)k * P.cap + i];
    }
}
template <int D, bool TOUCH>
__device__ __forceinline__ void classify_elem(const PolyView &P, const double *__restrict__ hps, int B, int nv,
                                              unsigned long long *__restrict__ out, unsigned *__restrict__ anyminus,
                                              int *__restrict__ tc, int *__restrict__ t1, const int i,
                                              const double (&x)[D > 0 ? D : MAXD], const unsigned char fl)
{
    const int d = D > 0 ? D : P.d;
    const bool live = fl & F_USED, ideal = fl & F_IDEAL;
    const bool wave_has_ideal = __ballot(ideal) != 0ull;      // directions are rare: scalar thresholds on the fast path
    const int nw = (B + 31) / 32;
    int touch = 0, first = -1;
    for (int w = 0; w < nw; w++) {
        const int bend = min(32, B - w * 32);
        const unsigned valid = bend == 32 ? 0xFFFFFFFFu : ((1u << bend) - 1u);
        unsigned plus = 0, notminus = 0;       // bit bb: s > a + EPS ; s > a - EPS
        if (!wave_has_ideal && bend == 32) {
            // full word, compile-time trip count: the scalar loads of several halfspaces are batched ahead
#pragma unroll
            for (int bb = 0; bb < 32; bb++) {
                const double *h = hps + (size_t)(w * 32 + bb) * (d + 3);      // wave-uniform address -> scalar loads
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < (D > 0 ? D : MAXD); k++) if (k < d) s = fma(h[k], x[k], s);
                const double hi = h[d + 1], lo = h[d + 2];                    // alpha +- EPS, precomputed on the host
                shift_in_gt(plus, s, hi);
                shift_in_gt(notminus, s, lo);
            }
        } else if (!wave_has_ideal) {
            for (int bb = 0; bb < bend; bb++) {
                const double *h = hps + (size_t)(w * 32 + bb) * (d + 3);
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < (D > 0 ? D : MAXD); k++) if (k < d) s = fma(h[k], x[k], s);
                const double hi = h[d + 1], lo = h[d + 2];
                shift_in_gt(plus, s, hi);
                shift_in_gt(notminus, s, lo);
            }
        } else {
            for (int bb = 0; bb < bend; bb++) {
                const double *h = hps + (size_t)(w * 32 + bb) * (d + 3);
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < (D > 0 ? D : MAXD); k++) if (k < d) s = fma(h[k], x[k], s);
                const double a = ideal ? 0.0 : h[d];
                plus = plus + plus + (unsigned)(s > a + POLY_EPS);
                notminus = notminus + notminus + (unsigned)(s > a - POLY_EPS);
            }
        }
        // (the bits were shifted in from the right -- one add-with-carry per compare: halfspace 0 sits at bit bend - 1)
        plus = __brev(plus) >> (32 - bend); notminus = __brev(notminus) >> (32 - bend);
        // class = 1 MINUS (01), 2 ZERO (10), 3 PLUS (11): low bit = plus | !notminus, high bit = notminus
        unsigned lowb = (plus | ~notminus) & valid, highb = notminus & valid;
        unsigned minusbits = ~notminus & valid;
        if (!live) { lowb = 0; highb = 0; minusbits = 0; }
        if (i < nv) out[(size_t)w * P.cap + i] = spread32(lowb) | (spread32(highb) << 1);
        if (TOUCH) {
            const unsigned nonplus = live ? (~plus & valid) : 0u;
            if (nonplus) { if (touch == 0) first = w * 32 + (__ffs((int)nonplus) - 1); touch += __popc(nonplus); }
        }
        if (anyminus) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) minusbits |= __shfl_xor(minusbits, o, WAVE);
            // bits already published by another wave need no atomic (relaxed read; a stale value only costs an atomic)
            if ((threadIdx.x & 63) == 0 && (minusbits & ~__hip_atomic_load(&anyminus[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
                atomicOr(&anyminus[w], minusbits);
        }
    }
    if (TOUCH && i < nv) { tc[i] = touch; t1[i] = first; }
}
template <int D, bool TOUCH>
__device__ __forceinline__ void classify_batch_body(const PolyView &P, const double *__restrict__ hps, int B, int nv,
                                                    unsigned long long *__restrict__ out, unsigned *__restrict__ anyminus,
                                                    int *__restrict__ tc, int *__restrict__ t1, const int i)
{
    double x[D > 0 ? D : MAXD]; unsigned char fl;
    classify_load<D>(P, nv, i, x, fl);
    classify_elem<D, TOUCH>(P, hps, B, nv, out, anyminus, tc, t1, i, x, fl);
}
// Grid-stride over the elements with the next element's coordinates requested before the current one is classified.  Measured
// (scripts/probe/k1_small_b.py, q = 5, 8 M elements): a grid capped at 4..16 workgroups per CU is 3-15 % SLOWER at B = 2..32 than
// one workgroup per 256 elements, so the launch is not capped and the loop runs once; it stays for nv beyond one grid.  Unrolling
// the partial-word loop by 2, 4 or 8 changes nothing (same box, +-2 %).
constexpr int K1_MAX_BLOCKS = 1 << 22;
template <int D, bool TOUCH>
__global__ __launch_bounds__(PB) void k_classify_batch_t(PolyView P, const double *__restrict__ hps, int B, int nv,
                                                         unsigned long long *__restrict__ out, unsigned *__restrict__ anyminus,
                                                         int *__restrict__ tc, int *__restrict__ t1)
{
    const int stride = (int)gridDim.x * PB;
    int i = (int)(blockIdx.x * PB + threadIdx.x);
    if (i - (int)(threadIdx.x & 63) >= nv) return;                     // (wave-uniform)
    double x[D > 0 ? D : MAXD]; unsigned char fl;
    classify_load<D>(P, nv, i, x, fl);
    for (;;) {
        const int inext = i + stride;
        const bool more = inext - (int)(threadIdx.x & 63) < nv;       // (wave-uniform; no overflow: nv + stride < 2^31 by the launch)
        double xn[D > 0 ? D : MAXD]; unsigned char fln = 0;
        if (more) classify_load<D>(P, nv, inext, xn, fln);
        classify_elem<D, TOUCH>(P, hps, B, nv, out, anyminus, tc, t1, i, x, fl);
        if (!more) break;
        i = inext; fl = fln;
        for (int k = 0; k < (D > 0 ? D : MAXD); k++) x[k] = xn[k];
    }
}

// ---- K1 on the matrix pipe (round 2) -------------------------------------------------------------------------------------
// The scalar kernel above is VALU-bound from B ~ 16 on: per (element, halfspace) 5 fp64 FMAs, 2 fp64 compares and ~4 integer
// operations for the bit packing share the vector ALU (24 TFLOP/s useful, 30-36 % of the HBM peak at B = 16..32).  Here the dot
// products go to v_mfma_f64_16x16x4 -- one tile = 16 elements x 16 halfspaces, K = the coordinates in steps of 4 (zero padded) --
// and the compares come out of the vector ALU as WAVE MASKS (v_cmp writes an SGPR pair: the "ballot" is free), so no lane packs
// bits.  (fp64 matrix peak = fp64 vector peak on gfx950: the gain is the vector ALU freed of the FMAs, not a higher peak.)
//   A[i][k] = X[k][element i]   lane l holds A[l & 15][l >> 4]              (one f64 per lane and K step)
//   B[k][j] = h_j[k]            lane l holds B[l >> 4][l & 15]
//   D[i][j] = h_j . x_i         lane l holds rows (l >> 4) + 4 r, r = 0..3, of column l & 15
// A wave owns 64 elements = 4 tiles and keeps their A operands in registers; per 16 halfspaces it runs 4 independent MFMA chains.
// Mask (t, r) of tile t, result register r: bits 16 g .. 16 g + 15 are the 16 halfspace bits of element 16 t + 4 r + g.  Read as a
// string of 16-bit fields in the order (t, r, g) the field number IS the element's lane: the 32 mask dwords are dropped into lanes
// 0..31 of one register (v_writelane), lane L fetches dword L >> 1 (one ds_bpermute) and keeps half L & 1.
// Same result bit for bit as the scalar fma chain: the K steps of one MFMA are accumulated in ascending k with one rounding per
// step, chained MFMAs continue the chain, the padding adds fma(0, 0, s) = s (bslv_k1_mfma_selftest; and the class words against
// the scalar kernel on random data and inside the +-1e-9 bands: tests/test_poly_gpu.py).
typedef double d4_t __attribute__((ext_vector_type(4)));
// The 8 wave masks of one tile (4 result registers x {> alpha + EPS, > alpha - EPS}) and their hand-off into lanes 8 t .. 8 t + 7
// of mp / mn.  Written as two blocks of assembly so that the ORDER is fixed: v_writelane_b32 fetches its scalar source ahead of the
// vector pipeline -- issued right behind the v_cmp_f64 that produces the mask it picked up the register's previous content now and
// then (seen as lost PLUS bits against the scalar kernel).  Here every mask is at least 8 vector instructions old when it is read.
template <int T0>
__device__ __forceinline__ void tile_masks(const d4_t &acc, double hi, double lo, int &mp, int &mn)
{
    unsigned long long p0, p1, p2, p3, n0, n1, n2, n3;
    asm volatile("v_cmp_gt_f64 %0, %8, %12\n\tv_cmp_gt_f64 %1, %9, %12\n\tv_cmp_gt_f64 %2, %10, %12\n\tv_cmp_gt_f64 %3, %11, %12\n\t"
                 "v_cmp_gt_f64 %4, %8, %13\n\tv_cmp_gt_f64 %5, %9, %13\n\tv_cmp_gt_f64 %6, %10, %13\n\tv_cmp_gt_f64 %7, %11, %13"
                 : "=&s"(p0), "=&s"(p1), "=&s"(p2), "=&s"(p3), "=&s"(n0), "=&s"(n1), "=&s"(n2), "=&s"(n3)
                 : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(hi), "v"(lo));
    asm volatile("v_writelane_b32 %0, %2, %18\n\tv_writelane_b32 %0, %3, %19\n\tv_writelane_b32 %0, %4, %20\n\tv_writelane_b32 %0, %5, %21\n\t"
                 "v_writelane_b32 %0, %6, %22\n\tv_writelane_b32 %0, %7, %23\n\tv_writelane_b32 %0, %8, %24\n\tv_writelane_b32 %0, %9, %25\n\t"
                 "v_writelane_b32 %1, %10, %18\n\tv_writelane_b32 %1, %11, %19\n\tv_writelane_b32 %1, %12, %20\n\tv_writelane_b32 %1, %13, %21\n\t"
                 "v_writelane_b32 %1, %14, %22\n\tv_writelane_b32 %1, %15, %23\n\tv_writelane_b32 %1, %16, %24\n\tv_writelane_b32 %1, %17, %25"
                 : "+v"(mp), "+v"(mn)
                 : "s"((unsigned)p0), "s"((unsigned)(p0 >> 32)), "s"((unsigned)p1), "s"((unsigned)(p1 >> 32)),
                   "s"((unsigned)p2), "s"((unsigned)(p2 >> 32)), "s"((unsigned)p3), "s"((unsigned)(p3 >> 32)),
                   "s"((unsigned)n0), "s"((unsigned)(n0 >> 32)), "s"((unsigned)n1), "s"((unsigned)(n1 >> 32)),
                   "s"((unsigned)n2), "s"((unsigned)(n2 >> 32)), "s"((unsigned)n3), "s"((unsigned)(n3 >> 32)),
                   "n"(8 * T0), "n"(8 * T0 + 1), "n"(8 * T0 + 2), "n"(8 * T0 + 3), "n"(8 * T0 + 4), "n"(8 * T0 + 5), "n"(8 * T0 + 6), "n"(8 * T0 + 7));
}
template <int I, int N> struct StaticFor {
    template <class F> static __device__ __forceinline__ void run(F &&f) { f(std::integral_constant<int, I>{}); StaticFor<I + 1, N>::run(f); }
};
template <int N> struct StaticFor<N, N> { template <class F> static __device__ __forceinline__ void run(F &&) {} };
template <int D, bool TOUCH>
__global__ __launch_bounds__(PB) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_classify_batch_mfma(PolyView P, const double *__restrict__ hps, int B, int nv,
                                                            unsigned long long *__restrict__ out, unsigned *__restrict__ anyminus,
                                                            int *__restrict__ tc, int *__restrict__ t1)
{
    constexpr int KS = (D + 3) / 4;                                   // K steps
    constexpr int T = 4;                                              // tiles per wave
    const int lane = threadIdx.x & 63, col = lane & 15, kq = lane >> 4;
    const int base = (int)((blockIdx.x * PB + threadIdx.x) >> 6) * 64;
    if (base >= nv) return;                                           // (wave-uniform)
    double a[T][KS];
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int k = 4 * ks + kq, e = base + 16 * t + col;
            a[t][ks] = (e < nv && k < D) ? P.X[(size_t)k * P.cap + e] : 0.0;
        }
    const int e = base + lane;                                        // the element this lane writes
    const bool mine = e < nv;
    const unsigned char fl = mine ? P.flag[e] : 0;
    const bool live = fl & F_USED;
    // directions are measured against 0 (bslv_poly.c:126), i.e. with thresholds per ROW of a tile: a wave that holds one (rare)
    // takes the scalar path for its 64 elements
    if (__ballot(fl & F_IDEAL)) { classify_batch_body<D, TOUCH>(P, hps, B, nv, out, anyminus, tc, t1, e); return; }
    const int nw = (B + 31) / 32, nh = (B + 15) / 16;
    // operand B and the thresholds of the first 16 halfspaces (each half is fetched while the previous one is multiplied)
    double bq[KS], hi, lo;
    {
        const bool hv = col < B;
        const double *h = hps + (size_t)(hv ? col : 0) * (D + 3);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) { const int k = 4 * ks + kq; bq[ks] = (hv && k < D) ? h[k] : 0.0; }
        hi = hv ? h[D + 1] : INFINITY; lo = hv ? h[D + 2] : INFINITY; // alpha +- EPS (an absent halfspace: no bit)
    }
    int touch = 0, first = -1;
    for (int w = 0; w < nw; w++) {
        unsigned plus32 = 0, nm32 = 0;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int hx = 2 * w + half;
            double bc[KS]; const double chi = hi, clo = lo;
#pragma unroll
            for (int ks = 0; ks < KS; ks++) bc[ks] = bq[ks];
            if (hx + 1 < nh) {                                        // prefetch
                const int hs = (hx + 1) * 16 + col;
                const bool hv = hs < B;
                const double *h = hps + (size_t)(hv ? hs : 0) * (D + 3);
#pragma unroll
                for (int ks = 0; ks < KS; ks++) { const int k = 4 * ks + kq; bq[ks] = (hv && k < D) ? h[k] : 0.0; }
                hi = hv ? h[D + 1] : INFINITY; lo = hv ? h[D + 2] : INFINITY;
            }
            if (hx >= nh) break;                                      // (B <= 16 (mod 32): the second half is empty)
            d4_t acc[T];
#pragma unroll
            for (int t = 0; t < T; t++) acc[t] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KS; ks++)
#pragma unroll
                for (int t = 0; t < T; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][ks], bc[ks], acc[t], 0, 0, 0);
            int mp = 0, mn = 0;                                       // lanes 0..31: the mask dwords in the order (t, r, low | high)
            // the compares below are assembly, which the compiler's hazard recogniser does not look into: all MFMAs first, then the
            // wait states a vector read of a 16-pass MFMA result needs (19), once
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
            StaticFor<0, T>::run([&](auto ic) { tile_masks<decltype(ic)::value>(acc[decltype(ic)::value], chi, clo, mp, mn); });
            const unsigned fp = (unsigned)__builtin_amdgcn_ds_bpermute((lane >> 1) << 2, mp), fn = (unsigned)__builtin_amdgcn_ds_bpermute((lane >> 1) << 2, mn);
            plus32 |= ((fp >> (16 * (lane & 1))) & 0xFFFFu) << (16 * half);
            nm32 |= ((fn >> (16 * (lane & 1))) & 0xFFFFu) << (16 * half);
        }
        const int bend = min(32, B - w * 32);
        const unsigned valid = bend == 32 ? 0xFFFFFFFFu : ((1u << bend) - 1u);
        unsigned lowb = (plus32 | ~nm32) & valid, highb = nm32 & valid, minusbits = ~nm32 & valid;
        if (!live) { lowb = 0; highb = 0; minusbits = 0; }
        if (mine) out[(size_t)w * P.cap + e] = spread32(lowb) | (spread32(highb) << 1);
        if (TOUCH) {
            const unsigned nonplus = live ? (~plus32 & valid) : 0u;
            if (nonplus) { if (touch == 0) first = w * 32 + (__ffs((int)nonplus) - 1); touch += __popc(nonplus); }
        }
        if (anyminus) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) minusbits |= __shfl_xor(minusbits, o, WAVE);
            if (lane == 0 && (minusbits & ~__hip_atomic_load(&anyminus[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) atomicOr(&anyminus[w], minusbits);
        }
    }
    if (TOUCH && mine) { tc[e] = touch; t1[e] = first; }
}

// self-test of the claim above: the dot products of random 16 x 16 tiles by chained MFMAs against the scalar fma chain, bit for bit
template <int D>
__global__ void k_k1_mfma_selftest(const double *__restrict__ X /* D x 16 per tile */, const double *__restrict__ H /* 16 x D per tile */, int ntiles, unsigned long long *mismatch)
{
    constexpr int KS = (D + 3) / 4;
    const int lane = threadIdx.x & 63, col = lane & 15, kq = lane >> 4;
    const int tile = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (tile >= ntiles) return;
    const double *x = X + (size_t)tile * D * 16, *h = H + (size_t)tile * 16 * D;
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        const int k = 4 * ks + kq;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(k < D ? x[k * 16 + col] : 0.0, k < D ? h[col * D + k] : 0.0, acc, 0, 0, 0);
    }
    int bad = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int i = kq + 4 * r;                       // element row of this result, column = halfspace col
        double sref = 0.0;
        for (int k = 0; k < D; k++) sref = fma(h[col * D + k], x[k * 16 + i], sref);
        bad += __double_as_longlong(sref) != __double_as_longlong(acc[r]);
    }
    if (bad) atomicAdd(mismatch, (unsigned long long)bad);
}

// Which of the two runs: the scalar kernel.  Measured (profiles/r02_f64_pipes.json, scripts/probe/f64_pipes.hip): on gfx950 a wave's
// v_fma_f64 / v_cmp_f64 make 6-9 % of their stand-alone progress while another wave of the same SIMD issues f64 MFMAs back to back --
// the f64 MFMA occupies the SIMD's fp64 datapath instead of running beside it, so the dot products cost the same fp64 cycles either
// way, K padded from 5 to 8 costs 60 % more of them, and the compares + mask hand-off come on top (21.7 against 23.2 TFLOP/s useful
// at q = 5, B = 512).  BSLV_K1_MFMA=1 or bslv_poly_debug_set key 9 select the matrix kernel (kept: bit-identical, tested).
static bool g_k1_mfma = getenv("BSLV_K1_MFMA") != nullptr;
static void launch_classify_batch(hipStream_t s, PolyView P, const double *hps, int B, int nv, unsigned long long *out, unsigned *anyminus,
                                  int *tc, int *t1)
{
    if (g_k1_mfma && B >= 16 && P.d >= 2 && P.d <= 10) {
        dim3 g((unsigned)((nv + PB - 1) / PB)), b(PB);                    // 64 elements per wave
        static const bool check = getenv("BSLV_K1_CHECK") != nullptr;    // DEBUG: the scalar kernel beside it, outputs compared on the host
        const int nw = (B + 31) / 32;
        unsigned long long *out2 = nullptr; unsigned *any2 = nullptr; int *tt2 = nullptr;
        if (check) {
            (void)malloc0(&out2, (size_t)nw * P.cap * 8); (void)malloc0(&any2, nw * 4); (void)malloc0(&tt2, (size_t)2 * (nv + 1) * 4);
            if (anyminus) (void)hipMemcpyAsync(any2, anyminus, nw * 4, hipMemcpyDeviceToDevice, s);
        }
        switch (P.d) {
#define CASE(D) case D: if (tc) hipLaunchKernelGGL((k_classify_batch_mfma<D, true>), g, b, 0, s, P, hps, B, nv, out, anyminus, tc, t1); \
                       else hipLaunchKernelGGL((k_classify_batch_mfma<D, false>), g, b, 0, s, P, hps, B, nv, out, anyminus, tc, t1); break;
            CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
#undef CASE
        }
        if (check) {
            g_k1_mfma = false;
            launch_classify_batch(s, P, hps, B, nv, out2, anyminus ? any2 : nullptr, tc ? tt2 : nullptr, tc ? tt2 + nv : nullptr);
            g_k1_mfma = true;
            (void)hipStreamSynchronize(s);
            std::vector<unsigned long long> wa((size_t)nw * P.cap), wb((size_t)nw * P.cap);
            std::vector<unsigned> aa(nw), ab(nw); std::vector<int> ta(2 * (size_t)nv), tb(2 * (size_t)nv);
            (void)hipMemcpy(wa.data(), out, wa.size() * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(wb.data(), out2, wb.size() * 8, hipMemcpyDeviceToHost);
            if (anyminus) { (void)hipMemcpy(aa.data(), anyminus, nw * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(ab.data(), any2, nw * 4, hipMemcpyDeviceToHost); }
            if (tc) { (void)hipMemcpy(ta.data(), tc, (size_t)nv * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(ta.data() + nv, t1, (size_t)nv * 4, hipMemcpyDeviceToHost);
                      (void)hipMemcpy(tb.data(), tt2, (size_t)2 * nv * 4, hipMemcpyDeviceToHost); }
            long bw = 0, bt = 0, ba = 0; int fw = -1, fi = -1;
            for (int w = 0; w < nw; w++) for (int i = 0; i < nv; i++) if (wa[(size_t)w * P.cap + i] != wb[(size_t)w * P.cap + i]) { if (!bw) { fw = w; fi = i; } bw++; }
            if (tc) for (size_t i = 0; i < 2 * (size_t)nv; i++) bt += ta[i] != tb[i];
            if (anyminus) for (int w = 0; w < nw; w++) ba += aa[w] != ab[w];
            fprintf(stderr, "K1 check: d %d B %d nv %d cap %d touch %d: words differ %ld (first w %d i %d: %016llx vs %016llx), tc/t1 differ %ld, anyminus differ %ld\n",
                    P.d, B, nv, P.cap, tc != nullptr, bw, fw, fi, fw >= 0 ? wa[(size_t)fw * P.cap + fi] : 0ull, fw >= 0 ? wb[(size_t)fw * P.cap + fi] : 0ull, bt, ba);
            if (bw) {
                std::vector<unsigned char> fl(nv); (void)hipMemcpy(fl.data(), P.flag, nv, hipMemcpyDeviceToHost);
                std::vector<double> hh((size_t)B * (P.d + 3)); (void)hipMemcpy(hh.data(), hps, hh.size() * 8, hipMemcpyDeviceToHost);
                fprintf(stderr, "  flags:");
                for (int i = 0; i < nv && i < 64; i++) fprintf(stderr, " %d:%02x", i, fl[i]);
                fprintf(stderr, "\n  differing elements:");
                int shown = 0;
                for (int i = 0; i < nv && shown < 12; i++) {
                    bool df = false;
                    for (int w = 0; w < nw; w++) df |= wa[(size_t)w * P.cap + i] != wb[(size_t)w * P.cap + i];
                    if (!df) continue;
                    shown++;
                    double x[MAXD]; for (int k = 0; k < P.d; k++) (void)hipMemcpy(&x[k], P.X + (size_t)k * P.cap + i, 8, hipMemcpyDeviceToHost);
                    double s0 = 0; for (int k = 0; k < P.d; k++) s0 = fma(hh[k], x[k], s0);
                    fprintf(stderr, " [%d fl %02x w0 %016llx vs %016llx s(h0) %.3e alpha0 %.3e]", i, fl[i], wa[i], wb[i], s0, hh[P.d]);
                }
                fprintf(stderr, "\n");
            }
            (void)hipFree(out2); (void)hipFree(any2); (void)hipFree(tt2);
        }
        return;
    }
    dim3 g(std::min((nv + PB - 1) / PB, K1_MAX_BLOCKS)), b(PB);
    switch (P.d) {
#define CASE(D) case D: if (tc) hipLaunchKernelGGL((k_classify_batch_t<D, true>), g, b, 0, s, P, hps, B, nv, out, anyminus, tc, t1); \
                       else hipLaunchKernelGGL((k_classify_batch_t<D, false>), g, b, 0, s, P, hps, B, nv, out, anyminus, tc, t1); break;
        CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
#undef CASE
    default: if (tc) hipLaunchKernelGGL((k_classify_batch_t<0, true>), g, b, 0, s, P, hps, B, nv, out, anyminus, tc, t1);
             else hipLaunchKernelGGL((k_classify_batch_t<0, false>), g, b, 0, s, P, hps, B, nv, out, anyminus, tc, t1); break;
    }
}

// ---------------- sorted-list helpers ----------------
__device__ __forceinline__ int isect_count(const int *a, int na, const int *b, int nb)
{
    int i = 0, j = 0, n = 0;
    while (i < na && j < nb) {
        int x = a[i], y = b[j];
        n += (x == y);
        i += (x <= y);
        j += (y <= x);
    }
    return n;
}
__device__ __forceinline__ int isect3_count(const int *a, int na, const int *b, int nb, const int *c, int nc)
{
    int i = 0, j = 0, k = 0, n = 0;
    while (i < na && j < nb && k < nc) {
        int x = a[i], y = b[j], z = c[k];
        int m = max(x, max(y, z));
        if (x == m && y == m && z == m) { n++; i++; j++; k++; }
        else { i += (x < m); j += (y < m); k += (z < m); }
    }
    return n;
}

// Short lists (the common case: a vertex of a simple polytope lies on d facets) are fetched with 16
// independent predicated loads -- ONE memory latency instead of a dependent chain of loads through the
// merge loop -- and intersected in registers.  Longer lists take the merge loop.
// The loads are unconditional (the pool is allocated with LCAP entries of slack behind its capacity, so reading
// past the end of a list never leaves the allocation) and masked afterwards: predicated loads compiled to one
// exec-masked load + wait per entry, i.e. 16 memory latencies in a row instead of one.
__device__ __forceinline__ void load_list(const int *p, int n, int (&out)[LCAP])
{
    int raw[LCAP];
#pragma unroll
    for (int k = 0; k < LCAP; k++) raw[k] = p[k];
#pragma unroll
    for (int k = 0; k < LCAP; k++) out[k] = k < n ? raw[k] : 0x7FFFFFFF;
}
// bit a of the result: A[a] occurs in B (both sorted, padded with INT_MAX which never matches a < n entry)
__device__ __forceinline__ unsigned match_mask(const int (&A)[LCAP], int na, const int (&B)[LCAP])
{
    unsigned m = 0;
#pragma unroll
    for (int a = 0; a < LCAP; a++) {
        bool hit = false;
#pragma unroll
        for (int b = 0; b < LCAP; b++) hit |= (A[a] == B[b]);
        m |= (unsigned)(hit && a < na) << a;
    }
    return m;
}
// position of x in the sorted list b[0..nb), or -1 (binary search: log2(nb) dependent loads)
__device__ __forceinline__ int find_sorted(const int *b, int nb, int x)
{
    int lo = 0, hi = nb - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1, y = b[mid];
        if (y == x) return mid;
        if (y < x) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}
// bit a: A[a] occurs in the LONG sorted list b.  The extreme directions of an upper image lie on every facet
// with a zero weight, so their incidence lists grow to thousands of entries; an edge to a direction must not
// walk such a list linearly (it cost 100-200 us per cut): each of the <= 16 short entries does its own
// binary search, the 16 searches overlap.
__device__ __forceinline__ unsigned match_mask_long(const int (&A)[LCAP], int na, const int *b, int nb, int (&posB)[LCAP])
{
    unsigned m = 0;
#pragma unroll
    for (int a = 0; a < LCAP; a++) {
        int p = a < na ? find_sorted(b, nb, A[a]) : -1;
        posB[a] = p;
        m |= (unsigned)(p >= 0) << a;
    }
    return m;
}
// bit a: facet A[a] is in the membership bitmap row (see PolyView::lslot); 16 independent loads
__device__ __forceinline__ unsigned match_mask_bits(const int (&A)[LCAP], int na, const unsigned *row)
{
    unsigned w[LCAP];
#pragma unroll
    for (int a = 0; a < LCAP; a++) w[a] = row[(a < na ? A[a] : 0) >> 5];
    unsigned m = 0;
#pragma unroll
    for (int a = 0; a < LCAP; a++) m |= (unsigned)(a < na && ((w[a] >> (A[a] & 31)) & 1u)) << a;
    return m;
}
// the bitmap row of element v, or nullptr
__device__ __forceinline__ const unsigned *lrow(const PolyView &P, int v)
{
    if (!P.lslot) return nullptr;
    const int sl = P.lslot[v];
    return sl >= 0 ? P.lbits + (size_t)sl * P.lstride : nullptr;
}
// An element that got a membership bitmap at the start of the chunk (its list was long then) and whose list has since become
// short is rebuilt by ONE thread (the short path of the emit passes): its bitmap must follow, an edge may still consult it.
__device__ __forceinline__ void relist_bitmap(const PolyView &P, int v, unsigned off, int n)
{
    unsigned *lb = const_cast<unsigned *>(lrow(P, v));
    if (!lb) return;
    for (int w = 0; w < P.lstride; w++) lb[w] = 0u;
    for (int j = 0; j < n; j++) { const int g = P.pool[off + j]; lb[g >> 5] |= 1u << (g & 31); }
}
// Intersection of two sorted lists of very different lengths: every entry of the short one is looked up in the long one by binary
// search -- ns log2(nl) loads instead of the ns + nl of the merge (a point on 20..60 facets against an extreme direction on thousands,
// when the direction has no membership bitmap: the merge was hundreds of microseconds of dependent loads per edge on covering problems).
// f(value, position in a, position in b) for every common entry, in ascending order.
template <class F>
__device__ __forceinline__ void isect_each(const int *a, int na, const int *b, int nb, F f)
{
    if ((long long)na * 12 < nb) { for (int i = 0; i < na; i++) { const int x = a[i]; const int p = find_sorted(b, nb, x); if (p >= 0) f(x, i, p); } return; }
    if ((long long)nb * 12 < na) { for (int j = 0; j < nb; j++) { const int y = b[j]; const int p = find_sorted(a, na, y); if (p >= 0) f(y, p, j); } return; }
    int i = 0, j = 0;
    while (i < na && j < nb) {
        const int x = a[i], y = b[j];
        if (x == y) f(x, i, j);
        i += (x <= y);
        j += (y <= x);
    }
}
__device__ __forceinline__ int isect_count_fast(const int *a, int na, const int *b, int nb)
{
    if (na <= LCAP && nb <= LCAP) {
        int A[LCAP], B[LCAP];
        load_list(a, na, A); load_list(b, nb, B);
        return __popc(match_mask(A, na, B));
    }
    if (na <= LCAP || nb <= LCAP) {
        const bool ashort = na <= nb;
        int S[LCAP], pos[LCAP];
        load_list(ashort ? a : b, ashort ? na : nb, S);
        return __popc(match_mask_long(S, ashort ? na : nb, ashort ? b : a, ashort ? nb : na, pos));
    }
    int n = 0;
    isect_each(a, na, b, nb, [&](int, int, int) { n++; });
    return n;
}

// ---------------- E: edges ----------------
// eflag: 0 dropped, 1 survives, 4 survives and joins a ZERO to a PLUS element, 2 crossing with a MINUS (b PLUS), 3 crossing with b MINUS
// per element triple: (survive, cross, new incidence-list length)
__device__ __forceinline__ Tri edge_triple(const PolyView &P, const int2 e, unsigned char *fl)
{
    signed char ca = P.cls[e.x], cb = P.cls[e.y];
    Tri t{0, 0, 0};
    unsigned char f = 0;
    if ((ca == -1 && cb == 1) || (ca == 1 && cb == -1)) {
        f = (ca == -1) ? 2 : 3;
        t.b = 1;
        t.c = isect_count_fast(P.pool + P.inc_off[e.x], P.inc_len[e.x], P.pool + P.inc_off[e.y], P.inc_len[e.y]) + 1;
    } else if (ca >= 0 && cb >= 0 && !(ca == 0 && cb == 0)) { f = (ca == 0 || cb == 0) ? 4 : 1; t.a = 1; }
    *fl = f;
    return t;
}
__global__ __launch_bounds__(PB) void k_edge_flags(PolyView P, const int2 *E, int ne, unsigned char *eflag, Tri *bsum)
{
    __shared__ Tri lds[16];
    int e = blockIdx.x * PB + threadIdx.x;
    Tri t{0, 0, 0};
    if (e < ne) { unsigned char f; t = edge_triple(P, E[e], &f); eflag[e] = f; }
    Tri tot;
    (void)block_exscan(t, &tot, lds);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
// writes survivors to Enew[0..nsurv), creates vertex nv0+crossidx with its incidence list at
// pool[pool0 + off), its edge at Enew[nsurv + crossidx], and marks keep[] for ZERO-PLUS edges
// D = compile-time dimension (0 = run-time P.d): keeps the coordinate arrays in registers -- with a run-time
// bound they are indexed dynamically and land in scratch memory (the kernel ran 10x slower)
template <int D>
__global__ __launch_bounds__(PB) void k_edge_emit(PolyView P, Hp hp, int facet, const int2 *E, int ne, const unsigned char *eflag,
                                                   const Tri *bpre, const Tri *totals, int2 *Enew, int nv0, unsigned pool0)
{
    __shared__ Tri lds[16];
    int e = blockIdx.x * PB + threadIdx.x;
    Tri t{0, 0, 0};
    unsigned char f = 0;
    int2 ed{0, 0};
    if (e < ne) {
        f = eflag[e]; ed = E[e];
        if (f == 1 || f == 4) t.a = 1;
        else if (f >= 2) {
            t.b = 1;
            t.c = isect_count_fast(P.pool + P.inc_off[ed.x], P.inc_len[ed.x], P.pool + P.inc_off[ed.y], P.inc_len[ed.y]) + 1;
        }
    }
    Tri tot;
    Tri ex = block_exscan(t, &tot, lds);
    if (e >= ne) return;
    ex = tri_add(ex, bpre[blockIdx.x]);
    const int d = D > 0 ? D : P.d;
    if (f == 1) Enew[ex.a] = ed;          // plain survivor: a streaming copy, no class lookups
    else if (f == 4) {
        Enew[ex.a] = ed;
        // ZERO element keeps the facets it shares with a PLUS neighbour (bslv_poly.c:634-652)
        signed char ca = P.cls[ed.x], cb = P.cls[ed.y];
        int z = -1, pl = -1;
        if (ca == 0 && cb == 1) { z = ed.x; pl = ed.y; }
        else if (ca == 1 && cb == 0) { z = ed.y; pl = ed.x; }
        if (z >= 0) {
            const int *A = P.pool + P.inc_off[z], *Bp = P.pool + P.inc_off[pl];
            unsigned char *K = P.keep + P.inc_off[z];
            int na = P.inc_len[z], nb = P.inc_len[pl];
            if (na <= LCAP && nb <= LCAP) {
                int RA[LCAP], RB[LCAP];
                load_list(A, na, RA); load_list(Bp, nb, RB);
                unsigned m = match_mask(RA, na, RB);
                while (m) { int a = __ffs((int)m) - 1; m &= m - 1; K[a] = 1; }
            } else if (na <= LCAP) {                 // ZERO element short, PLUS neighbour long
                int RA[LCAP], pos[LCAP];
                load_list(A, na, RA);
                unsigned m = match_mask_long(RA, na, Bp, nb, pos);
                while (m) { int a = __ffs((int)m) - 1; m &= m - 1; K[a] = 1; }
            } else if (nb <= LCAP) {                 // ZERO element long (a direction), PLUS neighbour short
                int RB[LCAP], pos[LCAP];
                load_list(Bp, nb, RB);
                unsigned m = match_mask_long(RB, nb, A, na, pos);
#pragma unroll
                for (int b2 = 0; b2 < LCAP; b2++) if ((m >> b2) & 1u) K[pos[b2]] = 1;
            } else {
                int i = 0, j = 0;
                while (i < na && j < nb) {
                    int x = A[i], y = Bp[j];
                    if (x == y) K[i] = 1;
                    i += (x <= y);
                    j += (y <= x);
                }
            }
        }
    } else if (f == 2 || f == 3) {
        const int mi = (f == 2) ? ed.x : ed.y, pl = (f == 2) ? ed.y : ed.x;
        const int w = nv0 + ex.b;
        const bool im = P.flag[mi] & F_IDEAL, ip = P.flag[pl] & F_IDEAL;
        constexpr int DD = D > 0 ? D : MAXD;
        double xm[DD], xp[DD], base[DD], dirv[DD];
#pragma unroll
        for (int k = 0; k < DD; k++) { xm[k] = k < d ? P.X[(size_t)k * P.cap + mi] : 0.0; xp[k] = k < d ? P.X[(size_t)k * P.cap + pl] : 0.0; }
        double hb = 0.0, hd = 0.0, a2 = hp.h[d];
        unsigned char nf = F_USED;
        // new vertex on the edge (bslv_poly.c:597-627); same operation order as oracle/poly_dd.c:
        // base + mu * dir with (base, dir) = (minus, plus - minus) for two directions, (plus, minus - plus) for two
        // points, (the point, the direction) otherwise
        if (ip && im) {
            a2 = 0.0; nf |= F_IDEAL;
#pragma unroll
            for (int k = 0; k < DD; k++) { base[k] = xm[k]; dirv[k] = xp[k] - xm[k]; }
        } else if (!ip && !im) {
#pragma unroll
            for (int k = 0; k < DD; k++) { base[k] = xp[k]; dirv[k] = xm[k] - xp[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < DD; k++) { base[k] = ip ? xm[k] : xp[k]; dirv[k] = ip ? xp[k] : xm[k]; }
        }
#pragma unroll
        for (int k = 0; k < DD; k++) if (k < d) hd = fma(hp.h[k], dirv[k], hd);
#pragma unroll
        for (int k = 0; k < DD; k++) if (k < d) hb = fma(hp.h[k], base[k], hb);
        const double mu = (a2 - hb) / hd;
#pragma unroll
        for (int k = 0; k < DD; k++) if (k < d) P.X[(size_t)k * P.cap + w] = fma(mu, dirv[k], base[k]);
        P.flag[w] = nf;
        P.cls[w] = 0;
        // incidence = inc(minus) & inc(plus) + new facet (bslv_poly.c:634-665)
        const unsigned off = pool0 + (unsigned)ex.c;
        int *out = P.pool + off;
        const int *A = P.pool + P.inc_off[mi], *Bp = P.pool + P.inc_off[pl];
        int na = P.inc_len[mi], nb = P.inc_len[pl], n = 0;
        if (na <= LCAP && nb <= LCAP) {
            int RA[LCAP], RB[LCAP];
            load_list(A, na, RA); load_list(Bp, nb, RB);
            const unsigned m = match_mask(RA, na, RB);
#pragma unroll
            for (int a = 0; a < LCAP; a++) if ((m >> a) & 1u) out[n++] = RA[a];
        } else if (na <= LCAP || nb <= LCAP) {       // one end is a direction with a long list
            const bool ashort = na <= nb;
            int S[LCAP], pos[LCAP];
            load_list(ashort ? A : Bp, ashort ? na : nb, S);
            const unsigned m = match_mask_long(S, ashort ? na : nb, ashort ? Bp : A, ashort ? nb : na, pos);
#pragma unroll
            for (int a = 0; a < LCAP; a++) if ((m >> a) & 1u) out[n++] = S[a];
        } else {
            int i = 0, j = 0;
            while (i < na && j < nb) {
                int x = A[i], y = Bp[j];
                if (x == y) out[n++] = x;
                i += (x <= y);
                j += (y <= x);
            }
        }
        out[n++] = facet;
        P.inc_off[w] = off;
        P.inc_len[w] = n;
        Enew[totals[0].a + ex.b] = int2{w, pl};
    }
}

// ---------------- Z: on-plane elements ----------------
// triple: (is ZERO, 0, new list length = kept + 1)
__global__ __launch_bounds__(PB) void k_vert_flags(PolyView P, int nv0, Tri *bsum)
{
    __shared__ Tri lds[16];
    int i = blockIdx.x * PB + threadIdx.x;
    Tri t{0, 0, 0};
    const int lane = threadIdx.x & 63;
    unsigned off = 0;
    int n = 0;
    bool zero = false;
    if (i < nv0 && P.cls[i] == 0) {
        zero = true;
        off = P.inc_off[i];
        n = P.inc_len[i];
        const unsigned char *K = P.keep + off;
        int kept = 0;
        if (n <= LCAP) {
#pragma unroll
            for (int j = 0; j < LCAP; j++) kept += (j < n) ? K[j] : 0;      // independent loads
        } else if (n <= LONGN)
            for (int j = 0; j < n; j++) kept += K[j];
        t.a = 1; t.c = kept + 1;
    }
    // long lists (extreme directions): the wave counts the kept entries together
    unsigned long long todo = __ballot(zero && n > LONGN);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const unsigned oo = __shfl(off, src, WAVE);
        const int nn = __shfl(n, src, WAVE);
        int cnt = 0;
        for (int j = lane; j < nn; j += WAVE) cnt += P.keep[oo + j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, WAVE);
        if (lane == src) t.c = cnt + 1;
    }
    Tri tot;
    (void)block_exscan(t, &tot, lds);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(PB) void k_vert_emit(PolyView P, int facet, int nv0, const Tri *bpre, int *members, unsigned pool0)
{
    __shared__ Tri lds[16];
    int i = blockIdx.x * PB + threadIdx.x;
    Tri t{0, 0, 0};
    signed char c = 2;
    unsigned keptmask = 0;
    int lst[LCAP];
    int n = 0;
    unsigned off_old = 0;
    if (i < nv0) {
        c = P.cls[i];
        if (c == 0) {
            off_old = P.inc_off[i];
            n = P.inc_len[i];
            const unsigned char *K = P.keep + off_old;
            int kept = 0;
            if (n <= LCAP) {
                load_list(P.pool + off_old, n, lst);
#pragma unroll
                for (int j = 0; j < LCAP; j++) { unsigned k = (j < n) ? K[j] : 0; keptmask |= (k & 1u) << j; }
                kept = __popc(keptmask);
            } else if (n <= LONGN)
                for (int j = 0; j < n; j++) kept += K[j];
            t.a = 1; t.c = kept + 1;
        }
    }
    const int lane = threadIdx.x & 63;
    const bool longz = (c == 0) && n > LONGN;
    {   // long lists: kept count by the whole wave (same value k_vert_flags produced)
        unsigned long long todo = __ballot(longz);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const unsigned oo = __shfl(off_old, src, WAVE);
            const int nn = __shfl(n, src, WAVE);
            int cnt = 0;
            for (int j = lane; j < nn; j += WAVE) cnt += P.keep[oo + j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, WAVE);
            if (lane == src) t.c = cnt + 1;
        }
    }
    Tri tot;
    Tri ex = block_exscan(t, &tot, lds);
    ex = tri_add(ex, bpre[blockIdx.x]);
    const unsigned off_new = pool0 + (unsigned)ex.c;
    {   // long lists: ordered compaction by the whole wave (ballot ranks), then the new facet
        unsigned long long todo = __ballot(longz);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const unsigned oo = __shfl(off_old, src, WAVE), on = __shfl(off_new, src, WAVE);
            const int nn = __shfl(n, src, WAVE), vv = __shfl(i, src, WAVE);
            int base = 0;
            for (int j0 = 0; j0 < nn; j0 += WAVE) {
                const int j = j0 + lane;
                const bool k = j < nn && P.keep[oo + j];
                const unsigned long long bm = __ballot(k);
                if (k) { P.pool[on + base + __popcll(bm & ((1ull << lane) - 1ull))] = P.pool[oo + j]; P.keep[oo + j] = 0; }
                base += __popcll(bm);
            }
            if (lane == 0) { P.pool[on + base] = facet; P.inc_off[vv] = on; P.inc_len[vv] = base + 1; }
        }
    }
    if (i >= nv0) return;
    if (c == -1) { P.flag[i] &= ~F_USED; return; }
    if (c != 0) return;
    members[ex.a] = i;
    if (longz) return;
    int m = 0;
    if (n <= LCAP) {
#pragma unroll
        for (int j = 0; j < LCAP; j++) if ((keptmask >> j) & 1u) { P.pool[off_new + m++] = lst[j]; P.keep[off_old + j] = 0; }
    } else
        for (int j = 0; j < n; j++)
            if (P.keep[off_old + j]) { P.pool[off_new + m++] = P.pool[off_old + j]; P.keep[off_old + j] = 0; }
    P.pool[off_new + m++] = facet;
    P.inc_off[i] = off_new;
    P.inc_len[i] = m;
}
__global__ void k_iota_members(int *members, int nzero, int nv0, int ncross)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < ncross) members[nzero + k] = nv0 + k;
}

// ---------------- K2: adjacency prune over the members of the new facet ----------------
// grid.x enumerates (row i, chunk of 256 columns j > i) in lexicographic order via rowblk[]:
// block g handles row rowof[g], columns j0[g] .. j0[g]+255.
struct PairBlk { int i, j0; };
// closed form of the lexicographic (row i, 256-column chunk) enumeration over i < j < nm:
// G(K) = sum_{k=1..K} ceil(k/256); blocks before row i: S(i) = G(nm-1) - G(nm-1-i)
__host__ __device__ inline long long pair_G(long long K)
{
    long long f = K / PB;
    return (long long)PB * f * (f + 1) / 2 + (K - PB * f) * (f + 1);
}
__device__ __forceinline__ PairBlk pair_block(int nm, long long g)
{
    const long long L = nm - 1, GL = pair_G(L);
    int lo = 0, hi = nm - 2;                      // largest i with S(i) <= g
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (GL - pair_G(L - mid) <= g) lo = mid; else hi = mid - 1;
    }
    long long before = GL - pair_G(L - lo);
    return PairBlk{lo, lo + 1 + (int)(g - before) * PB};
}
__global__ __launch_bounds__(PB) void k_pair_flags(PolyView P, const int *members, int nm, const PairBlk *blks,
                                                    unsigned char *pflag, Tri *bsum)
{
    __shared__ Tri lds[16];
    __shared__ int s_row[1024];
    const PairBlk pb = blks ? blks[blockIdx.x] : pair_block(nm, blockIdx.x);
    const int vi = members[pb.i];
    const int ni = P.inc_len[vi];
    const int *Li = P.pool + P.inc_off[vi];
    const bool cached = ni <= 1024;
    if (cached) for (int t = threadIdx.x; t < ni; t += PB) s_row[t] = Li[t];
    __syncthreads();
    const int *A = cached ? s_row : Li;
    const int j = pb.j0 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int vj = -1, nj = 0, nmut = 0;
    const int *Lj = nullptr;
    bool cand = false;
    if (j < nm) {
        vj = members[j];
        nj = P.inc_len[vj];
        Lj = P.pool + P.inc_off[vj];
        nmut = isect_count(A, ni, Lj, nj);
        cand = (P.d == 1) || (nmut >= P.d - 1);        // edge_test, bslv_poly.c:482-485
    }
    // wave-cooperative superset scan: for each candidate pair of this wave, all 64 lanes sweep the
    // members w and test  |inc_i & inc_j & inc_w| == |inc_i & inc_j|  (bslv_poly.c:487-505)
    bool adj = cand;
    unsigned long long todo = __ballot(cand && P.d > 1);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int cj = __shfl(vj, src, WAVE);
        const int cn = __shfl(nmut, src, WAVE);
        const int cnj = P.inc_len[cj];
        const int *cL = P.pool + P.inc_off[cj];
        bool found = false;
        for (int base = 0; base < nm; base += WAVE) {
            int w = base + lane;
            bool hit = false;
            if (w < nm) {
                int vw = members[w];
                if (vw != vi && vw != cj) {
                    int nw_ = P.inc_len[vw];
                    if (nw_ >= cn) hit = isect3_count(A, ni, cL, cnj, P.pool + P.inc_off[vw], nw_) == cn;
                }
            }
            if (__ballot(hit)) { found = true; break; }
        }
        if (lane == src) adj = !found;
    }
    Tri t{adj ? 1 : 0, 0, 0};
    pflag[(size_t)blockIdx.x * PB + threadIdx.x] = adj ? 1 : 0;
    Tri tot;
    (void)block_exscan(t, &tot, lds);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// ---- K2 on a local incidence bit matrix -------------------------------------------------------
// The members of the new facet only touch a few hundred facets.  Those get local ids (first-touch
// atomics; any bijection gives the same result) and each member a bit column: bits[w*nm + m],
// 64 local facets per word, transposed so that a wave reading word w of 64 consecutive members is
// one coalesced 512-B load.  Pair prefilter = popcount of ANDed columns; superset sweep = ANDN.
// 8 lanes per member: list entries and bit words are spread over the lanes, so the dependent
// global accesses of one member overlap instead of forming one serial chain
constexpr int LPM = 8;
// members with long lists (an extreme direction lies on thousands of facets) are swept by the whole wave
__global__ __launch_bounds__(PB) void k_local_ids(PolyView P, const int *members, int nm, int W, int stamp, int *fstamp, int *flocal,
                                                   int *nlocal, unsigned long long *bits)
{
    const int t = blockIdx.x * PB + threadIdx.x, lane = threadIdx.x & 63;
    const int m = t / LPM, sub = t % LPM;
    const bool valid = m < nm;
    int v = -1, n = 0;
    if (valid) {
        for (int w = sub; w < W; w += LPM) bits[(size_t)w * nm + m] = 0ull;
        v = members[m];
        n = P.inc_len[v];
    }
    const bool longm = valid && n > LONGN;
    if (valid && !longm) {
        const int *L = P.pool + P.inc_off[v];
        for (int j = sub; j < n; j += LPM) {
            int g = L[j];
            if (__hip_atomic_load(&fstamp[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= stamp) continue;
            int old = atomicMax(&fstamp[g], stamp);
            if (old < stamp) flocal[g] = atomicAdd(nlocal, 1);
        }
    }
    unsigned long long todo = __ballot(longm && sub == 0);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int vv = __shfl(v, src, WAVE), nn = __shfl(n, src, WAVE);
        const int *L = P.pool + P.inc_off[vv];
        for (int j = lane; j < nn; j += WAVE) {
            int g = L[j];
            if (__hip_atomic_load(&fstamp[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= stamp) continue;
            int old = atomicMax(&fstamp[g], stamp);
            if (old < stamp) flocal[g] = atomicAdd(nlocal, 1);
        }
    }
}
__global__ __launch_bounds__(PB) void k_build_bits(PolyView P, const int *members, int nm, int W, const int *flocal, unsigned long long *bits)
{
    const int t = blockIdx.x * PB + threadIdx.x, lane = threadIdx.x & 63;
    const int m = t / LPM, sub = t % LPM;
    const bool valid = m < nm;
    int v = -1, n = 0;
    if (valid) { v = members[m]; n = P.inc_len[v]; }
    const bool longm = valid && n > LONGN;
    if (valid && !longm) {
        const int *L = P.pool + P.inc_off[v];
        for (int j = sub; j < n; j += LPM) {
            int id = flocal[L[j]];
            atomicOr(&bits[(size_t)(id >> 6) * nm + m], 1ull << (id & 63));
        }
    }
    unsigned long long todo = __ballot(longm && sub == 0);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int vv = __shfl(v, src, WAVE), nn = __shfl(n, src, WAVE), mm = __shfl(m, src, WAVE);
        const int *L = P.pool + P.inc_off[vv];
        for (int j = lane; j < nn; j += WAVE) {
            int id = flocal[L[j]];
            atomicOr(&bits[(size_t)(id >> 6) * nm + mm], 1ull << (id & 63));
        }
    }
}
// Facet-major member lists of the local bit matrix (large facets): cnt[f] elements carry local facet f, list[off[f] ..] are
// their positions.  The order inside a list is whatever the atomics give; it is only used for an existence test.
__global__ __launch_bounds__(PB) void k_fm_count(const unsigned long long *bits, int nm, int W, int *cnt)
{
    const long long t = (long long)blockIdx.x * PB + threadIdx.x;
    if (t >= (long long)W * nm) return;
    unsigned long long x = bits[t];
    const int w = (int)(t / nm);
    while (x) { const int b = __ffsll((long long)x) - 1; x &= x - 1; atomicAdd(&cnt[w * 64 + b], 1); }
}
__global__ __launch_bounds__(1024) void k_fm_scan(const int *cnt, int *off, int *cur, int n)
{
    __shared__ int s_part[1024];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n ? cnt[i] : 0;
        s_part[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            int a = threadIdx.x >= o ? s_part[threadIdx.x - o] : 0;
            __syncthreads();
            s_part[threadIdx.x] += a;
            __syncthreads();
        }
        const int ex = s_part[threadIdx.x] - v + s_carry;
        if (i < n) { off[i] = ex; cur[i] = ex; }
        __syncthreads();
        if (threadIdx.x == 1023) s_carry += s_part[1023];
        __syncthreads();
    }
}
__global__ __launch_bounds__(PB) void k_fm_fill(const unsigned long long *bits, int nm, int W, int *cur, int *list)
{
    const long long t = (long long)blockIdx.x * PB + threadIdx.x;
    if (t >= (long long)W * nm) return;
    unsigned long long x = bits[t];
    const int w = (int)(t / nm), m = (int)(t % nm);
    while (x) { const int b = __ffsll((long long)x) - 1; x &= x - 1; list[atomicAdd(&cur[w * 64 + b], 1)] = m; }
}
// the pair test of one virtual pair block (row pb.i, columns pb.j0 + threadIdx.x) by the calling workgroup: is my pair adjacent?
__device__ __forceinline__ bool pair_block_test(int d, const unsigned long long *bits, int nm, int W, const PairBlk pb,
                                                const int *fm_cnt, const int *fm_off, const int *fm_list /* NULL: scan all elements */, unsigned long long *s_m)
{
    for (int w = threadIdx.x; w < W; w += PB) s_m[w] = bits[(size_t)w * nm + pb.i];
    __syncthreads();
    const int j = pb.j0 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long *Mw = s_m + W + (size_t)wave * W;
    int nmut = 0;
    if (j < nm)
        for (int w = 0; w < W; w++) nmut += __popcll(s_m[w] & bits[(size_t)w * nm + j]);
    bool cand = (j < nm) && ((d == 1) || (nmut >= d - 1));        // edge_test, bslv_poly.c:482-485
    bool adj = cand;
    unsigned long long todo = __ballot(cand && d > 1);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int cj = pb.j0 + (threadIdx.x - lane) + src;
        // M = column i & column cj, shared by the wave through LDS
        for (int w = lane; w < W; w += WAVE) Mw[w] = s_m[w] & bits[(size_t)w * nm + cj];
        __builtin_amdgcn_wave_barrier();
        bool found = false;
        if (fm_list) {
            // a third element on all mutual facets lies in particular on the mutual facet with the fewest elements: scan that list
            int bc = 0x7fffffff, bf = -1;
            for (int w = 0; w < W; w++) if ((Mw[w] >> lane) & 1ull) { const int c = fm_cnt[w * 64 + lane]; if (c < bc) { bc = c; bf = w * 64 + lane; } }
            for (int o = 32; o > 0; o >>= 1) {
                const int oc = __shfl_xor(bc, o, WAVE), of = __shfl_xor(bf, o, WAVE);
                if (oc < bc || (oc == bc && of < bf)) { bc = oc; bf = of; }
            }
            const int *Lf = fm_list + fm_off[bf];
            for (int base = 0; base < bc; base += WAVE) {
                bool hit = false;
                if (base + lane < bc) {
                    const int wv = Lf[base + lane];
                    if (wv != pb.i && wv != cj) {
                        hit = true;
                        for (int w = 0; w < W; w++)
                            if (Mw[w] & ~bits[(size_t)w * nm + wv]) { hit = false; break; }
                    }
                }
                if (__ballot(hit)) { found = true; break; }
            }
        } else
        for (int base = 0; base < nm; base += WAVE) {
            const int wv = base + lane;
            bool hit = false;
            if (wv < nm && wv != pb.i && wv != cj) {
                hit = true;
                for (int w = 0; w < W; w++)
                    if (Mw[w] & ~bits[(size_t)w * nm + wv]) { hit = false; break; }
            }
            if (__ballot(hit)) { found = true; break; }           // some other element lies on all mutual facets
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == src) adj = !found;
    }
    return adj;
}
__global__ __launch_bounds__(PB) void k_pair_flags_bits(int d, const unsigned long long *bits, int nm, int W, unsigned char *pflag, Tri *bsum,
                                                         const int *fm_cnt, const int *fm_off, const int *fm_list /* NULL: scan all elements */,
                                                         int *nzlist /* NULL | blocks with an adjacent pair */, int *nzcount)
{
    __shared__ Tri lds[16];
    extern __shared__ unsigned long long s_m[];      // W words of row i, then 4 x W words (one M per wave)
    const PairBlk pb = pair_block(nm, blockIdx.x);
    const bool adj = pair_block_test(d, bits, nm, W, pb, fm_cnt, fm_off, fm_list, s_m);
    Tri t{adj ? 1 : 0, 0, 0};
    pflag[(size_t)blockIdx.x * PB + threadIdx.x] = adj ? 1 : 0;
    Tri tot;
    (void)block_exscan(t, &tot, lds);
    if (threadIdx.x == 0) {
        bsum[blockIdx.x] = tot;
        if (nzlist && tot.a > 0) nzlist[atomicAdd(nzcount, 1)] = blockIdx.x;
    }
}
// Emission WITHOUT a flag per pair (facets of several 10^5 elements: a flag byte per pair was 163 GB at 571 084 elements, the stop of
// S-degenerate at q = 10): the pair blocks that hold an adjacent pair are known (nzlist, a few per element) and so is where each
// block's pairs go (bpre: the scanned block sums of the flags pass); the listed blocks are simply tested AGAIN and write their pairs.
__global__ __launch_bounds__(PB) void k_pair_retest_emit(int d, const unsigned long long *bits, int nm, int W, const int *fm_cnt, const int *fm_off, const int *fm_list,
                                                          const int *members, const int *nzlist, const int *nzcount, const Tri *bpre, int2 *E, int ebase, int *EP)
{
    __shared__ Tri lds[16];
    extern __shared__ unsigned long long s_m[];
    const int n = *nzcount;
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        const int blk = nzlist[k];
        const PairBlk pb = pair_block(nm, blk);
        const bool adj = pair_block_test(d, bits, nm, W, pb, fm_cnt, fm_off, fm_list, s_m);
        Tri t{adj ? 1 : 0, 0, 0};
        Tri tot;
        const Tri ex = block_exscan(t, &tot, lds);
        if (adj) {
            E[ebase + bpre[blk].a + ex.a] = int2{members[pb.i], members[pb.j0 + threadIdx.x]};
            if (EP) EP[ebase + bpre[blk].a + ex.a] = -1;
        }
        __syncthreads();          // (s_m is reused by the next listed block)
    }
}

// The same for large facets, PTI rows per workgroup: the rows' words and a window of PB + PTI - 1 columns are staged in LDS once and
// serve PTI x PB pairs.  Row i of the tile works on columns i + 1 + PB c + t, i.e. on window position r + t.  pflag / bsum are
// indexed by the same virtual block id (row i, chunk c) as k_pair_flags_bits uses, so emission and edge order do not change.
// Grid: x = chunk, y = row group -- which also keeps the launch inside the 2^32 work-items per grid dimension that a
// one-dimensional grid of pair blocks exceeds from ~92 700 elements on (the runtime wraps such a grid without an error).
constexpr int PTI = 8;
__global__ __launch_bounds__(PB) void k_pair_flags_tiled(int d, const unsigned long long *bits, int nm, int W, unsigned char *pflag, Tri *bsum,
                                                          const int *fm_cnt, const int *fm_off, const int *fm_list /* NULL: scan all elements */,
                                                          int *nzlist, int *nzcount, int gy0 /* first row group of this launch (a rank's share of the pair space) */)
{
    extern __shared__ unsigned long long s_t[];       // PTI x W row words | W x (PB + PTI) column words | 4 x W (one M per wave)
    const int i0 = ((int)blockIdx.y + gy0) * PTI, c = blockIdx.x;
    const int jbase = i0 + 1 + c * PB;
    if (i0 >= nm - 1 || jbase >= nm) return;
    const int CW = PB + PTI;
    unsigned long long *rowsw = s_t, *colsw = s_t + (size_t)PTI * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long *Mw = colsw + (size_t)W * CW + (size_t)wave * W;
    for (int x = threadIdx.x; x < PTI * W; x += PB) { const int r = x / W, w = x % W, i = i0 + r; rowsw[x] = i < nm ? bits[(size_t)w * nm + i] : 0ull; }
    for (int x = threadIdx.x; x < W * CW; x += PB) { const int w = x / CW, k = x % CW, j = jbase + k; colsw[x] = j < nm ? bits[(size_t)w * nm + j] : 0ull; }
    __syncthreads();
    const long long L1 = nm - 1, GL = pair_G(L1);
    for (int r = 0; r < PTI; r++) {
        const int i = i0 + r, j0 = i + 1 + c * PB;
        if (i >= nm - 1 || j0 >= nm) break;                       // (uniform: later rows have fewer columns)
        const unsigned long long *s_m = rowsw + (size_t)r * W;
        const int j = j0 + threadIdx.x, k = r + threadIdx.x;
        int nmut = 0;
        if (j < nm)
            for (int w = 0; w < W; w++) nmut += __popcll(s_m[w] & colsw[(size_t)w * CW + k]);
        bool cand = (j < nm) && ((d == 1) || (nmut >= d - 1));        // edge_test, bslv_poly.c:482-485
        bool adj = cand;
        unsigned long long todo = __ballot(cand && d > 1);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int cj = j0 + (threadIdx.x - lane) + src;
            for (int w = lane; w < W; w += WAVE) Mw[w] = s_m[w] & bits[(size_t)w * nm + cj];
            __builtin_amdgcn_wave_barrier();
            bool found = false;
            if (fm_list) {
                int bc = 0x7fffffff, bf = -1;
                for (int w = 0; w < W; w++) if ((Mw[w] >> lane) & 1ull) { const int cc = fm_cnt[w * 64 + lane]; if (cc < bc) { bc = cc; bf = w * 64 + lane; } }
                for (int o = 32; o > 0; o >>= 1) {
                    const int oc = __shfl_xor(bc, o, WAVE), of = __shfl_xor(bf, o, WAVE);
                    if (oc < bc || (oc == bc && of < bf)) { bc = oc; bf = of; }
                }
                const int *Lf = fm_list + fm_off[bf];
                for (int base = 0; base < bc; base += WAVE) {
                    bool hit = false;
                    if (base + lane < bc) {
                        const int wv = Lf[base + lane];
                        if (wv != i && wv != cj) {
                            hit = true;
                            for (int w = 0; w < W; w++)
                                if (Mw[w] & ~bits[(size_t)w * nm + wv]) { hit = false; break; }
                        }
                    }
                    if (__ballot(hit)) { found = true; break; }
                }
            } else
            for (int base = 0; base < nm; base += WAVE) {
                const int wv = base + lane;
                bool hit = false;
                if (wv < nm && wv != i && wv != cj) {
                    hit = true;
                    for (int w = 0; w < W; w++)
                        if (Mw[w] & ~bits[(size_t)w * nm + wv]) { hit = false; break; }
                }
                if (__ballot(hit)) { found = true; break; }
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == src) adj = !found;
        }
        const long long vb = GL - pair_G(L1 - i) + c;           // virtual block (row i, chunk c), as pair_block enumerates them
        if (pflag) pflag[(size_t)vb * PB + threadIdx.x] = adj ? 1 : 0;       // (nullptr: the listed blocks are tested again at emission, k_pair_retest_emit)
        const int cnt = __syncthreads_count(adj);
        if (threadIdx.x == 0) {
            bsum[vb] = Tri{cnt, 0, 0};
            if (nzlist && cnt > 0) nzlist[atomicAdd(nzcount, 1)] = (int)vb;
        }
    }
}

// emission for large facets: only the pair blocks that hold an adjacent pair (k_pair_flags_bits listed them, in any order:
// every block writes to its own offset), a fixed grid walking the list
__global__ __launch_bounds__(PB) void k_pair_emit_list(const int *members, int nm, const int *nzlist, const int *nzcount, const unsigned char *pflag,
                                                        const Tri *bpre, int2 *E, int ebase, int *EP)
{
    __shared__ Tri lds[16];
    const int n = *nzcount;
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        const int blk = nzlist[k];
        const PairBlk pb = pair_block(nm, blk);
        unsigned char f = pflag[(size_t)blk * PB + threadIdx.x];
        Tri t{f, 0, 0};
        Tri tot;
        Tri ex = block_exscan(t, &tot, lds);
        if (f) {
            E[ebase + bpre[blk].a + ex.a] = int2{members[pb.i], members[pb.j0 + threadIdx.x]};
            if (EP) EP[ebase + bpre[blk].a + ex.a] = -1;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(PB) void k_pair_emit(const int *members, int nm, const PairBlk *blks, const unsigned char *pflag,
                                                   const Tri *bpre, int2 *E, int ebase, int *EP = nullptr)
{
    __shared__ Tri lds[16];
    const PairBlk pb = blks ? blks[blockIdx.x] : pair_block(nm, blockIdx.x);
    unsigned char f = pflag[(size_t)blockIdx.x * PB + threadIdx.x];
    if (!__syncthreads_or(f)) return;          // (nearly every block of a large facet: no adjacent pair in it)
    Tri t{f, 0, 0};
    Tri tot;
    Tri ex = block_exscan(t, &tot, lds);
    if (!f) return;
    E[ebase + bpre[blockIdx.x].a + ex.a] = int2{members[pb.i], members[pb.j0 + threadIdx.x]};
    if (EP) EP[ebase + bpre[blockIdx.x].a + ex.a] = -1;
}


// =====================================================================================================
// Single-cut pipeline in five launches (was fourteen): every kernel boundary is ~3-5 us of latency, far
// more than the work of a cut on a q=5 upper image, so passes that do not depend on each other share a
// launch (edge blocks first, vertex blocks behind them) and the adjacency prune runs in one workgroup.
//   k_classify | k_flags2 | k_scan2 -> host | k_emit2 | k2_fused -> host
// =====================================================================================================

// a ZERO element keeps the facets it shares with a PLUS neighbour (bslv_poly.c:634-652): marks keep[]
// thousands of edges of one extreme direction mark the same few hundred bytes: the stores to one cache line
// serialise in L2 (measured ~90 us per cut with an on-plane direction), so look first (L2-coherent load) and
// store only what is not yet marked
__device__ __forceinline__ void set_keep(unsigned char *K, int idx)
{
    if (!__hip_atomic_load(&K[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) K[idx] = 1;
}
// (z: the on-plane element, pl: the other end, PLUS for z's cut)
__device__ __forceinline__ void mark_keep_za(const PolyView &P, const int z, const int pl, const ZMarks &Z, const int *counters)
{
    const int *A = P.pool + P.inc_off[z], *Bp = P.pool + P.inc_off[pl];
    unsigned char *K = P.keep + P.inc_off[z];
    const int na = P.inc_len[z], nb = P.inc_len[pl];
    if (na > LONGN) {
        const int zid = zmarks_find(Z, counters, z);
        if (zid >= 0) {                          // stamp the facets of the PLUS end in z's row
            int *row = Z.rows + (size_t)zid * Z.stride;
            if (nb <= LCAP) {
                int RB[LCAP];
                load_list(Bp, nb, RB);
#pragma unroll
                for (int b2 = 0; b2 < LCAP; b2++) if (b2 < nb) row[RB[b2]] = Z.stamp;
            } else
                for (int b2 = 0; b2 < nb; b2++) row[Bp[b2]] = Z.stamp;
            return;
        }
    }
    if (na <= LCAP && nb <= LCAP) {
        int RA[LCAP], RB[LCAP];
        load_list(A, na, RA); load_list(Bp, nb, RB);
        unsigned m = match_mask(RA, na, RB);
        while (m) { int a = __ffs((int)m) - 1; m &= m - 1; set_keep(K, a); }
    } else if (na <= LCAP) {                 // ZERO element short, PLUS neighbour long
        int RA[LCAP], pos[LCAP];
        load_list(A, na, RA);
        unsigned m = match_mask_long(RA, na, Bp, nb, pos);
        while (m) { int a = __ffs((int)m) - 1; m &= m - 1; set_keep(K, a); }
    } else if (nb <= LCAP) {                 // ZERO element long (a direction), PLUS neighbour short
        int RB[LCAP], pos[LCAP];
        load_list(Bp, nb, RB);
        unsigned m = match_mask_long(RB, nb, A, na, pos);
#pragma unroll
        for (int b2 = 0; b2 < LCAP; b2++) if ((m >> b2) & 1u) set_keep(K, pos[b2]);
    } else isect_each(A, na, Bp, nb, [&](int, int i, int) { set_keep(K, i); });
}

// What the host decides after round A, decided on the device as well, so that round B can be queued before the
// host has seen the result (speculative launch): go = the cut removes something, every capacity suffices and the
// adjacency prune of the previous cut did not ask for its fallback (abort flag).
constexpr int CROSS_UB = 4096;         // new vertices a speculatively queued classification of the next cut covers (default of bslv_poly::cross_ub)
struct CutDev { int go, nminus, nzero, zero_ub, nsurv, ncross, newlen, ne0, nv_new, ebase; unsigned pool_z; int pad; };
// both scans in one launch: workgroup 0 the edge sums (-> totals[0], mail[0] with the classify counters and,
// in cnt[3], the exact edge count this cut saw), workgroup 1 the vertex sums (-> totals[1])
struct ScanArgs {
    Tri *ebsum; int nbe; Tri *vbsum; int nbv; Tri *totals; Mail *mail; const int *counters; int ne_ub; const int *ne_dev; int seq;
    CutDev *cd; const int *abort_flag; int nv0, vcap; unsigned poolused, poolcap; int cross_ub;
    const Mail *k2src;     // device-side result of the adjacency prune still unread by the host (or nullptr)
};
// scan of one array of block sums by the calling workgroup (any size): sums[] becomes exclusive prefixes
__device__ __forceinline__ Tri scan_sums(Tri *sums, int nb, Tri *lds)
{
    Tri carry{0, 0, 0};
    for (int base = 0; base < nb; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const Tri v = i < nb ? sums[i] : Tri{0, 0, 0};
        Tri tot;
        Tri ex = block_exscan(v, &tot, lds);
        if (i < nb) sums[i] = tri_add(ex, carry);
        carry = tri_add(carry, tot);
        __syncthreads();
    }
    return carry;
}
// totals, the device's verdict (CutDev) and the host mailbox of round A; called by one thread
__device__ __forceinline__ void publish_round_a(const ScanArgs &A, Tri te)
{
    A.totals[0] = te;
    const int ne0 = A.ne_dev ? *A.ne_dev : A.ne_ub;
    if (A.cd) {
        CutDev c;
        c.nminus = A.counters[0]; c.nzero = A.counters[1]; c.zero_ub = A.counters[2];
        c.nsurv = te.a; c.ncross = te.b; c.newlen = te.c; c.ne0 = ne0;
        c.go = c.nminus > 0 && !*A.abort_flag && c.ncross <= A.cross_ub && A.nv0 + c.ncross <= A.vcap &&
               (unsigned long long)A.poolused + (unsigned)c.newlen + (unsigned)c.zero_ub <= A.poolcap;
        c.nv_new = c.go ? A.nv0 + c.ncross : A.nv0;
        c.ebase = c.go ? c.nsurv + c.ncross : ne0;
        c.pool_z = A.poolused + (unsigned)c.newlen;
        c.pad = 0;
        *A.cd = c;
    }
    if (A.k2src) {
        const Tri kt = A.k2src->t;
        const int ks = A.k2src->seq;
        A.mail->k2t = kt; A.mail->k2seq = ks; A.mail->chk2 = mail_sum2(kt, ks);
    }
    const int c4[4] = {A.counters[0], A.counters[1], A.counters[2], ne0};
    mail_publish(A.mail, c4, te, A.seq);
}
// both scans of round A in one launch
__global__ __launch_bounds__(1024) void k_scan2(ScanArgs A)
{
    __shared__ Tri lds[16];
    if (blockIdx.x == 0) {             // the edge sums carry everything the verdict needs
        const Tri te = scan_sums(A.ebsum, A.nbe, lds);
        if (threadIdx.x == 0) publish_round_a(A, te);
    } else {
        const Tri tv = scan_sums(A.vbsum, A.nbv, lds);
        if (threadIdx.x == 0) A.totals[1] = tv;
    }
}

// flags pass.  Edge blocks: eflag + (survive, cross, new list length) sums, and -- once the cut is known to
// remove something (counters[0] = #MINUS, final since k_classify) -- the keep marks of ZERO-PLUS edges.
// Vertex blocks: (is ZERO, 0, old list length + 1): the rebuilt list of an on-plane element is allocated at
// its upper bound, so its place in the pool does not wait for the keep marks.
// ne_dev != nullptr: the edge count of the previous cut is still on the device (its adjacent pairs were
// appended without a host round trip); ne_ub then only sizes the grid.
struct ZLocal { int n; int v[ZMAX]; };      // the long on-plane elements of a cut (ZMarks), fetched once per workgroup: uniform addresses
__device__ __forceinline__ ZLocal zlocal_load(const ZMarks &Z, const int *counters)
{
    ZLocal z;
    z.n = counters[3] < ZMAX ? counters[3] : ZMAX;
#pragma unroll
    for (int k = 0; k < ZMAX; k++) z.v[k] = Z.zlist[k];
    return z;
}
// One edge of the flags pass: its flag (eflag), the list length of the new vertex of a crossing edge (ecount), the keep
// marks / ZMarks stamps of a ZERO-PLUS edge (only once the cut is known to remove something: counters[0] = #MINUS).
// Returns (survives, crosses, list length of the new vertex).  Classes are read through P.cls.
struct LongStamp { const int *list; int n; int *zrow; };       // facets of a long PLUS end still to be stamped in a ZMarks row (by the whole wave)
// One job of an edge in the flags pass: the intersection of the incidence lists of its ends -- a crossing edge needs its size (the
// list of the new vertex), an edge from an on-plane element ia to an element ib that is PLUS for ia's cut needs the positions in
// ia's list (its keep marks / ZMarks stamps): one shared code path, so that a wave holding edges of both kinds does not walk two
// chains one after the other.  Returns the list length of the new vertex (cross) or 0.
__device__ __forceinline__ int edge_job(const PolyView &P, const int2 ed, int ia, int ib, bool cross, const int *counters, const ZLocal &zl, const ZMarks &Z, LongStamp &ls)
{
    const bool mark = !cross;
    int tc = 0;
    const int na = P.inc_len[ia], nb = P.inc_len[ib];
    int zid = -1;
    if (mark && na > LONGN) {
#pragma unroll
        for (int k = 0; k < ZMAX; k++) if (k < zl.n && zl.v[k] == ia) zid = k;
    }
    if (zid >= 0 && nb <= LONGN) {
        // ZERO element with a long list, PLUS end of up to 64 facets: stamp the facets of the PLUS end in its ZMarks row,
        // 16 at a time (independent loads; one entry after the other cost 10 us per edge on points with 17..64 facets)
        int *zrow = Z.rows + (size_t)zid * Z.stride;
        for (int c0 = 0; c0 < nb; c0 += LCAP) {
            const int nc = nb - c0 < LCAP ? nb - c0 : LCAP;
            int RB[LCAP];
            load_list(P.pool + P.inc_off[ib] + c0, nc, RB);
#pragma unroll
            for (int b2 = 0; b2 < LCAP; b2++) if (b2 < nc) zrow[RB[b2]] = Z.stamp;
        }
    } else if (zid >= 0) {
        // both ends are extreme directions: hundreds of facets to stamp -- left to the whole wave (flags_block); one lane
        // walking the list cost 15-40 us on every cut that has a direction on the hyperplane
        ls.list = P.pool + P.inc_off[ib]; ls.n = nb; ls.zrow = Z.rows + (size_t)zid * Z.stride;
    } else if (na <= LCAP && nb <= LCAP) {
        int RA[LCAP], RB[LCAP];
        load_list(P.pool + P.inc_off[ia], na, RA); load_list(P.pool + P.inc_off[ib], nb, RB);
        unsigned m = match_mask(RA, na, RB);
        if (cross) tc = __popc(m) + 1;
        else {
            unsigned char *K = P.keep + P.inc_off[ia];
            while (m) { const int a = __ffs((int)m) - 1; m &= m - 1; K[a] = 1; }      // short list: plain stores, nothing to wait for
        }
    } else {
        // one short list, one long list with a membership bitmap (hot mode): a bit test per short entry
        const bool ashort = na <= nb;
        const int is = ashort ? ia : ib, il = ashort ? ib : ia, ns = ashort ? na : nb;
        const unsigned *row = ns <= LONGN ? lrow(P, il) : nullptr;
        if (row && (cross || ashort)) {              // (a marking edge needs positions in the ZERO element's list: A short)
            int cnt = 0;
            for (int c0 = 0; c0 < ns; c0 += LCAP) {  // 16 entries at a time (a point on 17..64 facets takes a few rounds)
                const int nc = ns - c0 < LCAP ? ns - c0 : LCAP;
                int RS[LCAP];
                load_list(P.pool + P.inc_off[is] + c0, nc, RS);
                unsigned m = match_mask_bits(RS, nc, row);
                if (cross) cnt += __popc(m);
                else {
                    unsigned char *K = P.keep + P.inc_off[ia] + c0;
                    while (m) { const int a = __ffs((int)m) - 1; m &= m - 1; K[a] = 1; }      // plain stores, nothing to wait for
                }
            }
            if (cross) tc = cnt + 1;
        } else if (cross) tc = isect_count_fast(P.pool + P.inc_off[ia], na, P.pool + P.inc_off[ib], nb) + 1;
        else mark_keep_za(P, ia, ib, Z, counters);
    }
    return tc;
}
// One edge of the flags pass: its flag (eflag), the list length of the new vertex of a crossing edge (ecount), the keep
// marks / ZMarks stamps of an edge from an on-plane element to an element that is PLUS for its cut (only once the cut is known to
// remove something: counters[0] = #MINUS).  Returns (survives, crosses, list length of the new vertex).  Classes are read through
// P.cls; own (rounds of independent cuts): which cut(s) an on-plane element belongs to.
//   eflag: 0 dropped, 1 survives, 2 / 3 crossing (x / y MINUS), 4 survives with one end on a plane, 5 survives with both ends on
//   planes of DIFFERENT cuts (rounds only)
__device__ __forceinline__ Tri flag_edge(const PolyView &P, const int2 ed, int e, const int *counters, const ZLocal &zl, unsigned char *eflag, int *ecount,
                                         const ZMarks &Z, LongStamp (&ls)[2], const R2Own &own)
{
    ls[0].n = 0; ls[1].n = 0;
    Tri t{0, 0, 0};
    const signed char ca = P.cls[ed.x], cb = P.cls[ed.y];
    unsigned char f = 0;
    bool mark_x = false, mark_y = false;       // x (y) is on the plane of ONE cut for which the other end is PLUS: it keeps the facets the two share
    if ((ca == -1 && cb == 1) || (ca == 1 && cb == -1)) { f = (ca == -1) ? 2 : 3; t.b = 1; }
    else if (ca >= 0 && cb >= 0) {
        if (ca == 0 && cb == 0) {
            if (own.cutof) {
                const int cx = own.cutof[ed.x], cy = own.cutof[ed.y];
                const bool common = (cx >= 0 && cy >= 0) ? cx == cy : r2_common_owner(own.hw, own.wm, own.nw, own.selw, ed.x, ed.y);
                if (!common) { f = 5; t.a = 1; mark_x = cx >= 0; mark_y = cy >= 0; }
            }
        } else if (ca == 0 || cb == 0) {
            f = 4; t.a = 1;
            const int z = ca == 0 ? ed.x : ed.y;
            const bool single = !own.cutof || own.cutof[z] >= 0;       // (an element shared by several cuts of a round keeps its whole list: no marks)
            mark_x = single && ca == 0; mark_y = single && cb == 0;
        } else { f = 1; t.a = 1; }
    }
    eflag[e] = f;
    const bool cross = f == 2 || f == 3;
    if (!cross && counters[0] <= 0) return t;
    // at most two jobs (both only for f == 5), ONE copy of the job's code: job 0 = the crossing edge, or x marked by y; job 1 = y marked by x
    const int first = (cross || mark_x) ? 0 : 1, last = (!cross && mark_y) ? 1 : 0;
#pragma unroll 1
    for (int job = first; job <= last; job++) {
        LongStamp l{nullptr, 0, nullptr};
        const int tc = edge_job(P, ed, job ? ed.y : ed.x, job ? ed.x : ed.y, cross, counters, zl, Z, l);
        if (job) ls[1] = l; else ls[0] = l;
        if (cross) { t.c = tc; ecount[e] = tc; }
    }
    return t;
}
// Virtual workgroup vb of the (nbe + nbv) workgroups of a flags pass, executed by the calling workgroup (blockDim.x edges or
// elements each).  Blocks run back to front: the edges and elements the recent cuts created -- where the next cut acts, with
// the long dependent chains -- sit at the end of the arrays and must not wait for a second wave of workgroups.
// accumulate: the edge sums are ADDED to ebsum (zeroed beforehand): another workgroup may own part of the same block.
__device__ __forceinline__ void flags_block(const PolyView &P, const int2 *E, int ne, int nbe, int nbv, int nv0, const int *counters, unsigned char *eflag,
                                            int *ecount, Tri *ebsum, Tri *vbsum, const ZMarks &Z, int vb, Tri *lds, bool accumulate, const R2Own &own)
{
    const int BS = blockDim.x;
    Tri t{0, 0, 0};
    Tri tot;
    if (vb < nbe) {
        const int eb = nbe - 1 - vb;
        const ZLocal zl = zlocal_load(Z, counters);
        const int e = eb * BS + threadIdx.x;
        LongStamp ls[2] = {{nullptr, 0, nullptr}, {nullptr, 0, nullptr}};
        if (e < ne) t = flag_edge(P, E[e], e, counters, zl, eflag, ecount, Z, ls, own);
        {   // long lists to stamp: all lanes of the wave share each one
            const int lane = threadIdx.x & 63;
#pragma unroll
            for (int side = 0; side < 2; side++) {
                unsigned long long todo = __ballot(ls[side].n > 0);
                while (todo) {
                    const int src = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const int *L = (const int *)__shfl((unsigned long long)(uintptr_t)ls[side].list, src, WAVE);
                    int *zr = (int *)__shfl((unsigned long long)(uintptr_t)ls[side].zrow, src, WAVE);
                    const int n = __shfl(ls[side].n, src, WAVE);
                    for (int j = lane; j < n; j += WAVE) zr[L[j]] = Z.stamp;
                }
            }
        }
        (void)block_exscan(t, &tot, lds);
        if (threadIdx.x == 0) {
            if (accumulate) { if (tot.a) atomicAdd(&ebsum[eb].a, tot.a); if (tot.b) atomicAdd(&ebsum[eb].b, tot.b); if (tot.c) atomicAdd(&ebsum[eb].c, tot.c); }
            else ebsum[eb] = tot;
        }
    } else {
        const int b = nbe + nbv - 1 - vb, idx = b * BS + threadIdx.x;
        if (idx < vm_count(P, nv0)) {
            const int i = vm_id(P, idx);
            // (the rebuilt list of an on-plane element: its old entries + one facet per cut it belongs to)
            if (P.cls[i] == 0) { t.a = 1; const int co = own.cutof ? own.cutof[i] : 0; t.c = zero_room(P, i, co < -1 ? -1 - co : 1); }
        }
        (void)block_exscan(t, &tot, lds);
        if (threadIdx.x == 0) vbsum[b] = tot;
    }
}
// nv_dev (rounds queued ahead of the host): the element count is still on the device, as ne_dev
// halt (rounds, poly_rounds2_kernels.inc): RState::halt, halt2.  A round queued BEHIND one that did not go ahead (a capacity was
// short, nothing was selected, ...) was given the edge buffer that round WOULD have written: every kernel of it must return at
// once.  This one did not until round 4 -- it walked a buffer nobody had written with the old edge count and took its "edges" for
// element numbers: P.cls[garbage].  Harmless on zero pages (edge (0, 0): nothing flagged, nothing marked), a memory access fault
// on recycled memory that happened to hold large values (round 3; tests/test_fill_gpu.py).
__global__ __launch_bounds__(1024) void k_flags2(PolyView P, const int2 *E, int ne_ub, const int *ne_dev, int nbe, int nv0, const int *counters,
                                                 unsigned char *eflag, int *ecount, Tri *ebsum, Tri *vbsum, ZMarks Z, const int *nv_dev = nullptr,
                                                 const int *halt = nullptr, R2Own own = R2Own{nullptr, nullptr, nullptr, nullptr, nullptr, 0})
{
    __shared__ Tri lds[16];
    __shared__ unsigned s_selw[32];
    if (halt && (halt[0] != 0 || halt[1] != 0)) return;
    if (own.cutof) {       // (the selection of the round as a bit mask: read by every edge whose ends are both on a plane)
        if (threadIdx.x < 32) s_selw[threadIdx.x] = (int)threadIdx.x < own.nw ? own.selw[threadIdx.x] : 0u;
        __syncthreads();
        own.selw = s_selw;
    }
    flags_block(P, E, ne_dev ? *ne_dev : ne_ub, nbe, (int)gridDim.x - nbe, nv_dev ? *nv_dev : nv0, counters, eflag, ecount, ebsum, vbsum, Z, (int)blockIdx.x, lds, false, own);
}
// emit pass.  Edge blocks: survivors -> Enew[0..nsurv), one new vertex per crossing edge (coordinates, flags,
// incidence list, its edge to the PLUS end at Enew[nsurv + crossidx]).  Vertex blocks: MINUS elements leave,
// ZERO elements get their kept facets + the new one at pool[pool_z + prefix) and become members[0..nzero).
template <int D>
__global__ __launch_bounds__(PB) void k_emit2(PolyView P, Hp hp, int facet, const int2 *E, int ne_ub, const int *ne_dev, int nbe,
                                              const unsigned char *eflag, const int *ecount, const Tri *ebpre, const Tri *vbpre, const Tri *totals,
                                              int2 *Enew, int nv0, unsigned pool_e, unsigned pool_z, int *members, ZMarks Z, const int *counters, const CutDev *cd,
                                              const int *EPold, int *EPnew, int own_scan, ScanArgs A)
{
    __shared__ Tri lds[16];
    Tri pre_e{0, 0, 0}, pre_v{0, 0, 0};
    int nsurv_all = 0;
    if (own_scan) {
        // small grid (hot mode): no k_scan2 launch in between -- every workgroup sums the block sums of k_flags2 itself
        // (its own prefix and the totals), reaches the verdict of round A from them, and workgroup 0 publishes it
        const bool eblock = (int)blockIdx.x < nbe;
        const int mye = eblock ? nbe - 1 - (int)blockIdx.x : 0, myv = eblock ? 0 : (int)gridDim.x - 1 - (int)blockIdx.x;
        Tri pe{0, 0, 0}, te{0, 0, 0}, pv{0, 0, 0};
        for (int i = threadIdx.x; i < A.nbe; i += PB) { const Tri v = A.ebsum[i]; te = tri_add(te, v); if (eblock && i < mye) pe = tri_add(pe, v); }
        if (!eblock) for (int i = threadIdx.x; i < myv; i += PB) pv = tri_add(pv, A.vbsum[i]);
        te = block_sum(te, lds);
        if (eblock) pre_e = block_sum(pe, lds); else pre_v = block_sum(pv, lds);
        if (blockIdx.x == 0 && threadIdx.x == 0) publish_round_a(A, te);
        const bool go = A.counters[0] > 0 && !*A.abort_flag && te.b <= A.cross_ub && A.nv0 + te.b <= A.vcap &&
                        (unsigned long long)A.poolused + (unsigned)te.c + (unsigned)A.counters[2] <= A.poolcap;
        if (!go) return;
        pool_z = A.poolused + (unsigned)te.c;
        nsurv_all = te.a;
    } else if (cd) {                   // speculative launch behind k_scan2: the device's own verdict on round A
        if (!cd->go) return;
        pool_z = cd->pool_z;
    }
    Tri t{0, 0, 0};
    Tri tot;
    if ((int)blockIdx.x < nbe) {              // back to front, as k_flags2
        const int eb = nbe - 1 - (int)blockIdx.x;
        const int ne = ne_dev ? *ne_dev : ne_ub;
        const int e = eb * PB + threadIdx.x;
        unsigned char f = 0;
        int2 ed{0, 0};
        if (e < ne) {
            f = eflag[e]; ed = E[e];
            if (f == 1 || f == 4) t.a = 1;
            else if (f >= 2) { t.b = 1; t.c = ecount[e]; }        // list length of the new vertex, from k_flags2
        }
        Tri ex = block_exscan(t, &tot, lds);
        if (e >= ne) return;
        ex = tri_add(ex, own_scan ? pre_e : ebpre[eb]);
        if (!own_scan) nsurv_all = totals[0].a;
        const int d = D > 0 ? D : P.d;
        if (f == 1 || f == 4) { Enew[ex.a] = ed; if (EPold) EPnew[ex.a] = EPold[e]; }      // (hot mode: where the edge sat before the chunk)
        else if (f == 2 || f == 3) {
            const int mi = (f == 2) ? ed.x : ed.y, pl = (f == 2) ? ed.y : ed.x;
            const int w = nv0 + ex.b;
            const bool im = P.flag[mi] & F_IDEAL, ip = P.flag[pl] & F_IDEAL;
            constexpr int DD = D > 0 ? D : MAXD;
            double xm[DD], xp[DD], base[DD], dirv[DD];
#pragma unroll
            for (int k = 0; k < DD; k++) { xm[k] = k < d ? P.X[(size_t)k * P.cap + mi] : 0.0; xp[k] = k < d ? P.X[(size_t)k * P.cap + pl] : 0.0; }
            double hb = 0.0, hd = 0.0, a2 = hp.h[d];
            unsigned char nf = F_USED;
            // new vertex on the edge (bslv_poly.c:597-627); same operation order as oracle/poly_dd.c
            if (ip && im) {
                a2 = 0.0; nf |= F_IDEAL;
#pragma unroll
                for (int k = 0; k < DD; k++) { base[k] = xm[k]; dirv[k] = xp[k] - xm[k]; }
            } else if (!ip && !im) {
#pragma unroll
                for (int k = 0; k < DD; k++) { base[k] = xp[k]; dirv[k] = xm[k] - xp[k]; }
            } else {
#pragma unroll
                for (int k = 0; k < DD; k++) { base[k] = ip ? xm[k] : xp[k]; dirv[k] = ip ? xp[k] : xm[k]; }
            }
#pragma unroll
            for (int k = 0; k < DD; k++) if (k < d) hd = fma(hp.h[k], dirv[k], hd);
#pragma unroll
            for (int k = 0; k < DD; k++) if (k < d) hb = fma(hp.h[k], base[k], hb);
            const double mu = (a2 - hb) / hd;
#pragma unroll
            for (int k = 0; k < DD; k++) if (k < d) P.X[(size_t)k * P.cap + w] = fma(mu, dirv[k], base[k]);
            P.flag[w] = nf;
            P.cls[w] = 0;
            // incidence = inc(minus) & inc(plus) + new facet (bslv_poly.c:634-665)
            const unsigned off = pool_e + (unsigned)ex.c;
            int *out = P.pool + off;
            const int *A = P.pool + P.inc_off[mi], *Bp = P.pool + P.inc_off[pl];
            int na = P.inc_len[mi], nb = P.inc_len[pl], n = 0;
            if (na <= LCAP && nb <= LCAP) {
                int RA[LCAP], RB[LCAP];
                load_list(A, na, RA); load_list(Bp, nb, RB);
                const unsigned m = match_mask(RA, na, RB);
#pragma unroll
                for (int a = 0; a < LCAP; a++) if ((m >> a) & 1u) out[n++] = RA[a];
            } else if (na <= LCAP || nb <= LCAP) {       // one end is a direction with a long list
                const bool ashort = na <= nb;
                int S[LCAP], pos[LCAP];
                load_list(ashort ? A : Bp, ashort ? na : nb, S);
                const unsigned *row = lrow(P, ashort ? pl : mi);
                const unsigned m = row ? match_mask_bits(S, ashort ? na : nb, row)
                                       : match_mask_long(S, ashort ? na : nb, ashort ? Bp : A, ashort ? nb : na, pos);
#pragma unroll
                for (int a = 0; a < LCAP; a++) if ((m >> a) & 1u) out[n++] = S[a];
            } else {
                const bool ashort = na <= nb;
                const int ns = ashort ? na : nb;
                const unsigned *row = ns <= LONGN ? lrow(P, ashort ? pl : mi) : nullptr;
                if (row) {                           // 17..64 facets against a membership bitmap: rounds of 16 bit tests
                    const int *S = ashort ? A : Bp;
                    for (int c0 = 0; c0 < ns; c0 += LCAP) {
                        const int nc = ns - c0 < LCAP ? ns - c0 : LCAP;
                        int RS[LCAP];
                        load_list(S + c0, nc, RS);
                        const unsigned m = match_mask_bits(RS, nc, row);
#pragma unroll
                        for (int a = 0; a < LCAP; a++) if ((m >> a) & 1u) out[n++] = RS[a];
                    }
                } else isect_each(A, na, Bp, nb, [&](int x, int, int) { out[n++] = x; });
            }
            out[n++] = facet;
            P.inc_off[w] = off;
            P.inc_len[w] = n;
            Enew[nsurv_all + ex.b] = int2{w, pl};
            if (EPold) EPnew[nsurv_all + ex.b] = -1;
        }
        return;
    }
    // ---- vertex blocks ----
    const int b = (int)gridDim.x - 1 - (int)blockIdx.x, idx = b * PB + threadIdx.x, lane = threadIdx.x & 63;
    const bool valid = idx < vm_count(P, nv0);
    const int i = valid ? vm_id(P, idx) : -1;
    signed char c = 2;
    int n = 0;
    unsigned off_old = 0;
    if (valid) {
        c = P.cls[i];
        if (c == 0) { off_old = P.inc_off[i]; n = P.inc_len[i]; t.a = 1; t.c = zero_room(P, i, 1); }
    }
    const int room = t.c;                      // (0: a long list with slack, rebuilt where it lies)
    Tri ex = block_exscan(t, &tot, lds);
    ex = tri_add(ex, own_scan ? pre_v : vbpre[b]);
    const unsigned off_new = (c == 0 && room == 0) ? off_old : pool_z + (unsigned)ex.c;
    const bool longz = (c == 0) && n > LONGN;
    {   // long lists (extreme directions): ordered compaction by the whole wave (ballot ranks), then the new facet
        unsigned long long todo = __ballot(longz);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const unsigned oo = __shfl(off_old, src, WAVE), on = __shfl(off_new, src, WAVE);
            const int nn = __shfl(n, src, WAVE), vv = __shfl(i, src, WAVE), rm = __shfl(room, src, WAVE);
            const int zid = zmarks_find(Z, counters, vv);
            const int *row = zid >= 0 ? Z.rows + (size_t)zid * Z.stride : nullptr;
            // its membership bitmap (hot mode) is rebuilt with the list: nobody reads it in this launch (crossing
            // edges join MINUS and PLUS elements, this one is ZERO)
            unsigned *lb = const_cast<unsigned *>(lrow(P, vv));
            if (lb) {
                for (int w = lane; w < P.lstride; w += WAVE) lb[w] = 0u;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
            int base = 0;
            constexpr int LIT = 16;                     // lists of up to 1024 facets: all entries and all marks are fetched first
            if (nn <= LIT * WAVE) {                     // (two memory latencies in all instead of two per 64 entries)
                int g[LIT];
                bool k[LIT];
#pragma unroll
                for (int it = 0; it < LIT; it++) { const int j = it * WAVE + lane; g[it] = j < nn ? P.pool[oo + j] : 0; }
#pragma unroll
                for (int it = 0; it < LIT; it++) { const int j = it * WAVE + lane; k[it] = j < nn && (row ? row[g[it]] == Z.stamp : P.keep[oo + j] != 0); }
#pragma unroll
                for (int it = 0; it < LIT; it++) {
                    if (it * WAVE >= nn) break;
                    const unsigned long long bm = __ballot(k[it]);
                    if (k[it]) {
                        P.pool[on + base + __popcll(bm & ((1ull << lane) - 1ull))] = g[it];
                        if (!row) P.keep[oo + it * WAVE + lane] = 0;
                        if (lb) atomicOr(&lb[g[it] >> 5], 1u << (g[it] & 31));
                    }
                    base += __popcll(bm);
                }
            } else
            for (int j0 = 0; j0 < nn; j0 += WAVE) {
                const int j = j0 + lane;
                const int g = j < nn ? P.pool[oo + j] : 0;
                const bool k = j < nn && (row ? row[g] == Z.stamp : P.keep[oo + j] != 0);
                const unsigned long long bm = __ballot(k);
                if (k) {
                    P.pool[on + base + __popcll(bm & ((1ull << lane) - 1ull))] = g;
                    if (!row) P.keep[oo + j] = 0;
                    if (lb) atomicOr(&lb[g >> 5], 1u << (g & 31));
                }
                base += __popcll(bm);
            }
            if (lane == 0) {
                P.pool[on + base] = facet; P.inc_off[vv] = on; P.inc_len[vv] = base + 1; if (lb) atomicOr(&lb[facet >> 5], 1u << (facet & 31));
                if (rm > 0) P.capx[vv] = ((unsigned long long)on << 32) | (unsigned)rm;       // (moved: the room reserved for it, slack included)
            }
        }
    }
    if (!valid) return;
    if (c == -1) { P.flag[i] &= ~F_USED; return; }
    if (c != 0) return;
    members[ex.a] = i;
    if (longz) return;
    int m = 0;
    if (n <= LCAP) {
        int lst[LCAP];
        unsigned keptmask = 0;
        const unsigned char *K = P.keep + off_old;
        load_list(P.pool + off_old, n, lst);
#pragma unroll
        for (int j = 0; j < LCAP; j++) { unsigned k = (j < n) ? K[j] : 0; keptmask |= (k & 1u) << j; }
#pragma unroll
        for (int j = 0; j < LCAP; j++) if ((keptmask >> j) & 1u) { P.pool[off_new + m++] = lst[j]; P.keep[off_old + j] = 0; }
    } else
        for (int j = 0; j < n; j++)
            if (P.keep[off_old + j]) { P.pool[off_new + m++] = P.pool[off_old + j]; P.keep[off_old + j] = 0; }
    P.pool[off_new + m++] = facet;
    P.inc_off[i] = off_new;
    P.inc_len[i] = m;
    relist_bitmap(P, i, off_new, m);
}

// ---- K2 in ONE workgroup: the adjacency prune over the members of the new facet (bslv_poly.c:482-540) ----
// Facets that occur in fewer than two member lists can never be mutual, so only the others get a local id:
// the bit matrix then fits in LDS even though the extreme directions carry lists of thousands of facets.
//   P1 count the member lists of each facet; the second arrival assigns the local id (fcount, zeroed again below)
//   P3 bit matrix columns (per member) and rows (per local facet) in LDS
//   P4 pairs: >= d-1 mutual facets and no third member on all of them (row intersection)
//   P6 ordered emission at E[ebase..)
// Falls back (mail t.b = 1) when the bit matrix does not fit; the host then runs the multi-kernel path.
constexpr int K2T = 1024, K2_MAXNM = 512, K2_MAXLONG = 64;
constexpr int K2_HASH_LOG = 13, K2_HASH = 1 << K2_HASH_LOG;      // LDS hash table of the facets in the member lists (64 KB)
__device__ __forceinline__ void pair_decode(long long p, int nm, int &i, int &j)
{
    const double b2 = 2.0 * nm - 1.0;
    int ii = (int)((b2 - sqrt(b2 * b2 - 8.0 * (double)p)) * 0.5);
    ii = ii < 0 ? 0 : (ii > nm - 2 ? nm - 2 : ii);
    while ((long long)ii * (2 * nm - ii - 1) / 2 > p) ii--;
    while ((long long)(ii + 1) * (2 * nm - ii - 2) / 2 <= p) ii++;
    i = ii;
    j = ii + 1 + (int)(p - (long long)ii * (2 * nm - ii - 1) / 2);
}
// same for nm <= K2_MAXNM (p < 2^17): single-precision root, exact integer fix-up
__device__ __forceinline__ void pair_decode32(int p, int nm, int &i, int &j)
{
    const float b2 = 2.0f * nm - 1.0f;
    int ii = (int)((b2 - sqrtf(b2 * b2 - 8.0f * (float)p)) * 0.5f);
    ii = ii < 0 ? 0 : (ii > nm - 2 ? nm - 2 : ii);
    while (ii * (2 * nm - ii - 1) / 2 > p) ii--;
    while ((ii + 1) * (2 * nm - ii - 2) / 2 <= p) ii++;
    i = ii;
    j = ii + 1 + (p - ii * (2 * nm - ii - 1) / 2);
}
template <class F>
__device__ __forceinline__ void k2_for_entries(const int *pool, int nm, const unsigned *s_off, const int *s_len, const unsigned char *s_islong,
                                               const int *s_long, int nlong, F f)
{
    for (int m = threadIdx.x / LPM; m < nm; m += K2T / LPM) {
        if (s_islong[m]) continue;
        const int *L = pool + s_off[m];
        const int n = s_len[m];
        for (int j = threadIdx.x % LPM; j < n; j += LPM) f(m, L[j]);
    }
    for (int k = 0; k < nlong; k++) {
        const int m = s_long[k];
        const int *L = pool + s_off[m];
        const int n = s_len[m];
        for (int j = threadIdx.x; j < n; j += K2T) f(m, L[j]);
    }
}
// P4+P5 of k2_fused, one pair per thread: a candidate (>= d-1 mutual facets, edge_test bslv_poly.c:482-485) is
// adjacent unless another member lies on all its mutual facets, i.e. unless the intersection of the member sets
// (rows) of its mutual facets holds more than the pair itself: |M| short independent row reads instead of a
// sweep over all members (the sweep, one dependent LDS read per member, cost 25 us per cut).
// NWC = compile-time bound of the row length in words, so the accumulator stays in registers.
template <int NWC>
__device__ __forceinline__ void k2_pairs(int nm, int d, int W, int NW, const unsigned long long *bits, const unsigned long long *rows, unsigned *adj_bits,
                                         int (*queue)[128], const unsigned long long *later = nullptr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the row intersection of one candidate
    auto sweep = [&](int p) {
        int i, j;
        pair_decode32(p, nm, i, j);
        unsigned long long acc[NWC];
#pragma unroll
        for (int k = 0; k < NWC; k++) acc[k] = ~0ull;
        for (int w = 0; w < W; w++) {
            unsigned long long M = bits[w * nm + i] & bits[w * nm + j];
            while (M) {
                const int f = (w << 6) + __ffsll((long long)M) - 1;
                M &= M - 1;
#pragma unroll
                for (int k = 0; k < NWC; k++) acc[k] &= (NWC == 1 || k < NW) ? rows[f * NW + k] : 0ull;
            }
        }
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < NWC; k++) cnt += __popcll(acc[k]);
        if (cnt == 2) atomicOr(&adj_bits[p >> 5], 1u << (p & 31));       // i and j themselves lie on all their mutual facets
    };
    // Only a fraction of the pairs are candidates; sweeping them where they are found would leave most lanes of
    // the wave idle, so each wave queues its candidates (LDS) and sweeps 64 at a time.
    // Pairs are visited row by row (wave w takes rows w, w+16, ...; lanes run over j): column i is one broadcast
    // read per row and no pair index has to be decoded -- the one workgroup is instruction-bound here.
    int qn = 0;                                    // wave-uniform
    int *q = queue[wave];
    for (int i = wave; i < nm - 1; i += K2T / WAVE) {
        const int rowbase = i * (2 * nm - i - 1) / 2 - (i + 1);          // pair index of (i, j) = rowbase + j
        for (int j0 = i + 1; j0 < nm; j0 += WAVE) {
            const int j = j0 + lane;
            bool cand = false;
            if (j < nm) {
                int nmut = 0;
                unsigned long long lat = 0ull;       // (rounds) a mutual facet of a LATER cut of the round: that cut's prune has the final word on this pair
                for (int w = 0; w < W; w++) { const unsigned long long x = bits[w * nm + i] & bits[w * nm + j]; nmut += __popcll(x); if (later) lat |= x & later[w]; }
                if (d == 1) atomicOr(&adj_bits[(rowbase + j) >> 5], 1u << ((rowbase + j) & 31));
                else cand = nmut >= d - 1 && lat == 0ull;
            }
            const unsigned long long bm = __ballot(cand);
            if (bm == 0ull) continue;
            if (cand) q[qn + __popcll(bm & ((1ull << lane) - 1ull))] = rowbase + j;
            qn += __popcll(bm);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (qn >= WAVE) {
                qn -= WAVE;
                sweep(q[qn + lane]);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
        }
    }
    if (lane < qn) sweep(q[lane]);
}
// V2 (rounds of independent cuts, poly_rounds2_kernels.inc): one workgroup per SELECTED cut s = blockIdx.x.  Its members are
// the on-plane elements and new vertices of the round that belong to cut s; the result -- the adjacency bitmap over the pairs
// of members, the members, their number, the number of adjacent pairs (-1: needs the multi-kernel prune) -- goes to global
// memory and k_r2_k2emit writes the edges of all cuts in order.  The global facet counters (fcount / flocal) would be shared
// by concurrent workgroups, so a prune that cannot use its LDS hash table asks for the fallback instead.
struct RState;
struct K2V2 {
    const RState *st; const int *cutof; int *mem_g; unsigned *adj_g; int *cnt_g; int *nm_g; int adjw_cap; long long *pair_tests;
    // prunes whose member lists hold more entries than the LDS hash table takes (extreme directions among the members: lists of
    // thousands of facets) count in global memory, as the single-cut pipeline does -- each in a slice of its own, handed out by a
    // ticket per round (fc_ticket[round & 1]; workgroup 0 clears the other one for the next round)
    int *fc2, *fl2, *fc_ticket; int fc_slices, fc_stride;
};
__device__ __forceinline__ int k2v2_round(const RState *st);
__device__ __forceinline__ int k2v2_rank_base(const RState *st);
// is element v (owner code co = cutof[v]) a member of the new facet of rank `want`?  A shared on-plane element (co = -1 - k) carries the
// ranks of its k cuts at the end of its rebuilt list, ascending.
__device__ __forceinline__ bool k2v2_member(const PolyView &P, int v, int co, int vs, int want)
{
    if (co >= -1) return co == vs;
    const int k = -1 - co;
    const int *tail = P.pool + P.inc_off[v] + P.inc_len[v] - k;
    bool mine = false;
    for (int t = 0; t < k; t++) mine |= tail[t] == want;
    return mine;
}
__device__ __forceinline__ void k2v2_sizes(const RState *st, int &S, int &go, int &nzero, int &nv0, int &ncross);
__device__ void k2v2_check_members(const K2V2 &V, int vs, const int *s_mem, int nm, const int *members, int nzero);
// (a device function: the kernel of the single-cut pipeline, k2_fused_t<false>, and the prune launch of a round, k_r2_k2, call it)
template <bool V2>
__device__ void k2_fused_body(PolyView P, int *members, int nzero, int nv0, int ncross, int *fcount, int *flocal,
                              int lds_words, int2 *E, int ebase, int *ne_dev, Tri *totals, Mail *mail, int seq, unsigned long long *dbg,
                              const CutDev *cd, int *abort_flag, int *EP, const Hp &hn, int *counters_n, int *zlist_n, const K2V2 &V)
{
    extern __shared__ unsigned long long k2_dyn[];
    if (V2) {
        int S, go;
        k2v2_sizes(V.st, S, go, nzero, nv0, ncross);
        if (blockIdx.x == 0 && threadIdx.x == 0 && V.fc_ticket) V.fc_ticket[(k2v2_round(V.st) + 1) & 1] = 0;
        if (!go || (int)blockIdx.x >= S) return;
        cd = nullptr; ne_dev = nullptr;      // (dbg, BSLV_K2_DEBUG: every prune workgroup of a round adds its phases)
    }
    if (!V2 && blockIdx.x > 0) {
        // workgroups 1.. ride along: they classify the elements against the NEXT halfspace (the prune reads no
        // classes), which saves that launch
        classify_body(P, hn, cd->nv_new, counters_n, zlist_n, (blockIdx.x - 1) * K2T + threadIdx.x);
        return;
    }
    if (cd) {                          // speculative launch: sizes from the device, nothing to do unless the cut goes ahead
        nzero = cd->nzero; ncross = cd->ncross; ebase = cd->ebase;
        const int nm_ = nzero + ncross;
        if (!cd->go || nm_ < 2 || nm_ > K2_MAXNM) {
            if (threadIdx.x == 0) {
                const bool big = cd->go && nm_ > K2_MAXNM;       // needs the multi-kernel prune
                Tri r{0, big ? 1 : 2, 0};
                if (big) *abort_flag = 1;
                totals[0] = r;
                if (ne_dev) *ne_dev = ebase;
                mail->t = r;
                mail->seq = seq;
            }
            return;
        }
    }
    unsigned long long t_prev = dbg ? wall_clock64() : 0ull;
#define K2_PHASE(k) do { if (dbg && threadIdx.x == 0) { unsigned long long t_now = wall_clock64(); if (V2) atomicAdd(&dbg[k], t_now - t_prev); else dbg[k] += t_now - t_prev; t_prev = t_now; } } while (0)
    __shared__ Tri lds[16];
    __shared__ unsigned s_off[K2_MAXNM];
    __shared__ int s_len[K2_MAXNM], s_mem[K2_MAXNM], s_long[K2_MAXLONG];
    __shared__ unsigned char s_islong[K2_MAXNM];
    __shared__ int s_nlong, s_nloc, s_carry;
    __shared__ int s_queue[K2T / WAVE][128];      // candidate pairs per wave (k2_pairs)
    const int tid = threadIdx.x, d = P.d;
    int nm = nzero + ncross;
    const int vs = (int)blockIdx.x;                 // V2: the selected cut of this workgroup
    const int myrank = V2 ? k2v2_rank_base(V.st) + vs : 0x7FFFFFFF;      // ... and the rank of its facet
    // V2: fallback request / empty result of cut vs (uniform callers)
    auto v2_result = [&](int cnt, int n) { if (tid == 0) { V.cnt_g[vs] = cnt; V.nm_g[vs] = n; } };
    if (V2) {
        // P0a: the members of cut vs among the on-plane elements (members[0..nzero), slot order) and the new vertices
        __shared__ int s_base;
        if (tid == 0) s_base = 0;
        __syncthreads();
        const int ncand = nzero + ncross;
        // every thread takes CPT CONSECUTIVE candidates (so one scan per CPT * K2T candidates keeps them in order) and has the loads of
        // all of them in flight together: a round has 3-4 thousand candidates, which used to be four dependent load-scan-barrier trips
        constexpr int CPT = 8;
        for (int c0 = 0; c0 < ncand; c0 += K2T * CPT) {
            const int cb = c0 + tid * CPT;
            int vv[CPT], co[CPT], cnt = 0;
#pragma unroll
            for (int u = 0; u < CPT; u++) { const int c = cb + u; vv[u] = c < ncand ? (c < nzero ? members[c] : nv0 + (c - nzero)) : -1; }
#pragma unroll
            for (int u = 0; u < CPT; u++) co[u] = vv[u] >= 0 ? V.cutof[vv[u]] : 0;
#pragma unroll
            for (int u = 0; u < CPT; u++) { if (vv[u] >= 0 && !k2v2_member(P, vv[u], co[u], vs, myrank)) vv[u] = -1; cnt += vv[u] >= 0; }
            Tri t{cnt, 0, 0};
            Tri tot;
            const Tri ex = block_exscan(t, &tot, lds);
            int pos = s_base + ex.a;
#pragma unroll
            for (int u = 0; u < CPT; u++) if (vv[u] >= 0) { if (pos < K2_MAXNM) s_mem[pos] = vv[u]; pos++; }
            __syncthreads();
            if (tid == 0) s_base += tot.a;
            __syncthreads();
        }
        nm = s_base;
        K2_PHASE(10);
        if (nm > K2_MAXNM) { v2_result(-1, nm); return; }          // too large for one workgroup: multi-kernel prune (the negative count says why: -1 members, -2 pair bitmap, -3 hash table, -4 bit matrix)
#ifdef BSLV_R2_CHECK_MEMBERS
        k2v2_check_members(V, vs, s_mem, nm, members, nzero);          // debugging aid, O(nm^2): a member list never holds an element twice
#endif
        if (tid == 0 && nm >= 2) atomicAdd((unsigned long long *)V.pair_tests, (unsigned long long)((long long)nm * (nm - 1) / 2));
        if (nm < 2) { v2_result(0, nm); return; }
    }
    const long long npairs = (long long)nm * (nm - 1) / 2;
    const int nadjw = (int)((npairs + 31) / 32);
    unsigned *adj_bits = (unsigned *)k2_dyn;
    unsigned long long *bits = k2_dyn + (nadjw + 1) / 2;       // (behind the hash table when that is used, see P1)
    const long long bits_cap = (long long)lds_words - (nadjw + 1) / 2;
    if (bits_cap < 0) {                 // not even the pair bitmap fits (uniform): multi-kernel prune
        if (V2) { v2_result(-2, nm); return; }
        if (threadIdx.x == 0) { Tri r{0, 1, 0}; totals[0] = r; if (ne_dev) *ne_dev = ebase; if (abort_flag) *abort_flag = 1; mail->t = r; mail->seq = seq; }
        return;
    }
    // P0: members (ZERO elements from k_emit2, then the new vertices), their lists
    for (int m = tid; m < nm; m += K2T) {
        int v;
        if (V2) v = s_mem[m];
        else if (m < nzero) v = members[m]; else { v = nv0 + (m - nzero); members[m] = v; }
        s_mem[m] = v; s_off[m] = P.inc_off[v]; s_len[m] = P.inc_len[v]; s_islong[m] = 0;
    }
    for (int w = tid; w < nadjw; w += K2T) adj_bits[w] = 0u;
    if (tid == 0) { s_nlong = 0; s_nloc = 0; s_carry = 0; }
    __syncthreads();
    for (int m = tid; m < nm; m += K2T)
        if (s_len[m] > LONGN) { int k = atomicAdd(&s_nlong, 1); if (k < K2_MAXLONG) { s_long[k] = m; s_islong[m] = 1; } }
    __syncthreads();
    const int nlong = s_nlong < K2_MAXLONG ? s_nlong : K2_MAXLONG;
    K2_PHASE(0);
    // P1-P3: which facets occur in at least two member lists (only those can be mutual) -> local ids -> bit matrix.
    // Few list entries in all (the usual case): an open-addressing hash table in LDS, keyed by facet rank -- LDS atomics
    // instead of returning atomics on global memory (two memory latencies per phase).  Otherwise the counters in
    // global memory (fcount / flocal, zero between cuts).
    __shared__ int s_total;
    if (tid == 0) s_total = 0;
    __syncthreads();
    {
        int part = 0;
        for (int m = tid; m < nm; m += K2T) part += s_len[m];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, WAVE);
        if ((tid & 63) == 0 && part) atomicAdd(&s_total, part);
    }
    __syncthreads();
    const bool use_hash = 2 * s_total <= K2_HASH && (long long)K2_HASH <= bits_cap;        // load factor <= 1/2 (uniform)
    if (V2 && !use_hash) {
        __shared__ int s_slice;
        if (tid == 0) s_slice = V.fc_slices > 0 ? atomicAdd(&V.fc_ticket[k2v2_round(V.st) & 1], 1) : V.fc_slices;
        __syncthreads();
        if (s_slice >= V.fc_slices) { v2_result(-3, nm); return; }
        fcount = V.fc2 + (size_t)s_slice * V.fc_stride;
        flocal = V.fl2 + (size_t)s_slice * V.fc_stride;
    }
    int *hkey = (int *)bits, *hval = hkey + K2_HASH;                 // the table sits in front of the bit matrix
    if (use_hash) bits += K2_HASH;                                   // (K2_HASH ints of keys + K2_HASH of values = K2_HASH 64-bit words)
    const long long bcap = use_hash ? bits_cap - K2_HASH : bits_cap;
    auto hslot = [&](int g) {                                        // slot of key g (inserted if absent)
        unsigned sl = ((unsigned)g * 2654435761u) >> (32 - K2_HASH_LOG);
        for (;;) {
            const int k = atomicCAS(&hkey[sl], -1, g);
            if (k == -1 || k == g) return (int)sl;
            sl = (sl + 1) & (K2_HASH - 1);
        }
    };
    if (use_hash) {
        for (int w = tid; w < K2_HASH; w += K2T) { hkey[w] = -1; hval[w] = 0; }
        __syncthreads();
        k2_for_entries(P.pool, nm, s_off, s_len, s_islong, s_long, nlong, [&](int, int g) { atomicAdd(&hval[hslot(g)], 1); });
        __syncthreads();
        for (int w = tid; w < K2_HASH; w += K2T) hval[w] = hval[w] >= 2 ? (0x40000000 | atomicAdd(&s_nloc, 1)) : 0;
        __syncthreads();
    } else {
        k2_for_entries(P.pool, nm, s_off, s_len, s_islong, s_long, nlong, [&](int, int g) {
            if (atomicAdd(&fcount[g], 1) == 1) __hip_atomic_store(&flocal[g], atomicAdd(&s_nloc, 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        });
        __threadfence_block();
        __syncthreads();
    }
    K2_PHASE(1);
    K2_PHASE(2);
    const int W = (s_nloc + 63) >> 6, NW = (nm + 63) >> 6;
    unsigned long long *rows = bits + W * nm;        // rows[f * NW + k]: members on local facet f
    unsigned long long *later = rows + W * 64 * NW;  // (V2) later[w]: local facets of LATER cuts of this round (shared on-plane members carry them)
    if (V2 && (long long)W * nm + (long long)W * 64 * NW + W > bcap) {
        if (!use_hash) k2_for_entries(P.pool, nm, s_off, s_len, s_islong, s_long, nlong, [&](int, int g) { fcount[g] = 0; });
        v2_result(-4, nm);
        return;
    }
    if ((long long)W * nm + (long long)W * 64 * NW > bcap) {             // uniform: every thread sees the same s_nloc
        if (!use_hash) k2_for_entries(P.pool, nm, s_off, s_len, s_islong, s_long, nlong, [&](int, int g) { fcount[g] = 0; });
        if (tid == 0) { Tri r{0, 1, 0}; totals[0] = r; if (ne_dev) *ne_dev = ebase; if (abort_flag) *abort_flag = 1; mail->t = r; mail->seq = seq; }
        return;
    }
    for (int w = tid; w < W * nm + W * 64 * NW + (V2 ? W : 0); w += K2T) bits[w] = 0ull;
    __syncthreads();
    // P3
    if (use_hash)
        k2_for_entries(P.pool, nm, s_off, s_len, s_islong, s_long, nlong, [&](int m, int g) {
            const int v = hval[hslot(g)];
            if (!(v & 0x40000000)) return;
            const int id = v & 0x3FFFFFFF;
            atomicOr(&bits[(id >> 6) * nm + m], 1ull << (id & 63));
            atomicOr(&rows[id * NW + (m >> 6)], 1ull << (m & 63));
            if (V2 && g > myrank) atomicOr(&later[id >> 6], 1ull << (id & 63));
        });
    else
        k2_for_entries(P.pool, nm, s_off, s_len, s_islong, s_long, nlong, [&](int m, int g) {
            if (__hip_atomic_load(&fcount[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 2) return;
            const int id = __hip_atomic_load(&flocal[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(&bits[(id >> 6) * nm + m], 1ull << (id & 63));
            atomicOr(&rows[id * NW + (m >> 6)], 1ull << (m & 63));
            if (V2 && g > myrank) atomicOr(&later[id >> 6], 1ull << (id & 63));
        });
    __syncthreads();
    // the global counts go back to zero for the next cut (plain stores, nothing waits for them)
    if (!use_hash) k2_for_entries(P.pool, nm, s_off, s_len, s_islong, s_long, nlong, [&](int, int g) { fcount[g] = 0; });
    K2_PHASE(3);
    // P4+P5 (k2_pairs)
    const unsigned long long *lat = V2 ? later : nullptr;
    if (NW <= 1) k2_pairs<1>(nm, d, W, NW, bits, rows, adj_bits, s_queue, lat);
    else if (NW <= 2) k2_pairs<2>(nm, d, W, NW, bits, rows, adj_bits, s_queue, lat);
    else if (NW <= 4) k2_pairs<4>(nm, d, W, NW, bits, rows, adj_bits, s_queue, lat);
    else k2_pairs<K2_MAXNM / 64>(nm, d, W, NW, bits, rows, adj_bits, s_queue, lat);
    __syncthreads();
    K2_PHASE(4);
    K2_PHASE(5);
    if (V2) {
        // the bitmap, the members and the number of adjacent pairs go to global memory; k_r2_k2emit writes the edges
        unsigned *ag = V.adj_g + (size_t)vs * V.adjw_cap;
        int *mg = V.mem_g + (size_t)vs * K2_MAXNM;
        int part = 0;
        for (int w = tid; w < nadjw; w += K2T) { const unsigned x = adj_bits[w]; ag[w] = x; part += __popc(x); }
        for (int m = tid; m < nm; m += K2T) mg[m] = s_mem[m];
        const Tri tt = block_sum(Tri{part, 0, 0}, lds);
        v2_result(tt.a, nm);
        K2_PHASE(6);
        if (dbg && tid == 0) { atomicAdd(&dbg[7], 1ull); atomicAdd(&dbg[8], (unsigned long long)W); atomicAdd(&dbg[9], (unsigned long long)s_nloc); atomicAdd(&dbg[11], (unsigned long long)nm); atomicMax(&dbg[12], (unsigned long long)nm); }
        return;
    }
    // P6: adjacent pairs in lexicographic order; the 32 pairs of a bitmap word are consecutive, so (i, j) is
    // decoded once per word and stepped
    int last_tot = 0;
    for (int w0 = 0; w0 < nadjw; w0 += K2T) {
        const int ww = w0 + tid;
        unsigned word = ww < nadjw ? adj_bits[ww] : 0u;
        Tri t{(int)__popc(word), 0, 0};
        Tri tot;
        Tri ex = block_exscan(t, &tot, lds);
        int at = ebase + s_carry + ex.a;
        while (word) {                      // (a few set bits per word: decode each)
            const int bbit = __ffs((int)word) - 1;
            word &= word - 1;
            int i, j;
            pair_decode32(ww * 32 + bbit, nm, i, j);
            E[at] = int2{s_mem[i], s_mem[j]};
            if (EP) EP[at] = -1;
            at++;
        }
        if (w0 + K2T < nadjw) {             // (not after the last chunk: a barrier would wait for the edge stores)
            __syncthreads();
            if (tid == 0) s_carry += tot.a;
            __syncthreads();
        } else last_tot = tot.a;
    }
    K2_PHASE(6);
    if (tid == 0) {
        if (dbg) { dbg[7] += 1; dbg[8] += (unsigned long long)W; dbg[9] += (unsigned long long)s_nloc; }
        Tri r{s_carry + last_tot, 0, 0};
        totals[0] = r;
        if (ne_dev) *ne_dev = ebase + r.a;
        mail->t = r;
        mail->seq = seq;
    }
}


template <bool V2>
__global__ __launch_bounds__(K2T) void k2_fused_t(PolyView P, int *members, int nzero, int nv0, int ncross, int *fcount, int *flocal,
                                                  int lds_words, int2 *E, int ebase, int *ne_dev, Tri *totals, Mail *mail, int seq, unsigned long long *dbg,
                                                  const CutDev *cd, int *abort_flag, int *EP, Hp hn, int *counters_n, int *zlist_n, K2V2 V)
{
    k2_fused_body<V2>(P, members, nzero, nv0, ncross, fcount, flocal, lds_words, E, ebase, ne_dev, totals, mail, seq, dbg, cd, abort_flag, EP, hn, counters_n, zlist_n, V);
}

// the result of a prune (device memory: a write to host memory at the very end of the kernel kept the NEXT launch
// waiting for the PCIe acknowledgement, 5-6 us per cut) normally reaches the host with the mailbox of the next round A;
// at the end of a sequence of cuts this kernel forwards it
__global__ void k_forward_mail(const Mail *src, Mail *dst)
{
    const Tri t = src->t;
    mail_publish(dst, nullptr, t, src->seq);
}

// ---------------- hot mode: set-up and merge (once per chunk of cuts) ----------------
// elements some cut of the chunk touches (tc > 0, from k_classify_batch) -> hv (ascending); the others are PLUS for all
__global__ __launch_bounds__(PB) void k_hotv_flags(PolyView P, const int *tc, int nv, Tri *bsum)
{
    __shared__ Tri lds[16];
    const int i = blockIdx.x * PB + threadIdx.x;
    Tri t{0, 0, 0};
    if (i < nv) t.a = (P.flag[i] & F_USED) && tc[i] > 0;
    Tri tot;
    (void)block_exscan(t, &tot, lds);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(PB) void k_hotv_emit(PolyView P, const int *tc, int nv, const Tri *bpre, int *hv)
{
    __shared__ Tri lds[16];
    const int i = blockIdx.x * PB + threadIdx.x;
    Tri t{0, 0, 0};
    if (i < nv) { t.a = (P.flag[i] & F_USED) && tc[i] > 0; if (!t.a) P.cls[i] = 1; }
    Tri tot;
    Tri ex = block_exscan(t, &tot, lds);
    if (t.a) hv[bpre[blockIdx.x].a + ex.a] = i;
}
// membership bitmaps of the hot elements with long lists: one wave per hot element
__global__ __launch_bounds__(PB) void k_lbits_build(PolyView P, int nhv, int maxslots, int *nslots)
{
    const int idx = blockIdx.x * (PB / WAVE) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= nhv) return;
    const int v = P.hv[idx], n = P.inc_len[v];
    if (n <= LONGN) return;
    int sl = 0;
    if (lane == 0) sl = atomicAdd(nslots, 1);
    sl = __shfl(sl, 0, WAVE);
    if (sl >= maxslots) return;                 // (table full: this element keeps the binary search)
    unsigned *row = P.lbits + (size_t)sl * P.lstride;
    const int *L = P.pool + P.inc_off[v];
    for (int j = lane; j < n; j += WAVE) { const int g = L[j]; atomicOr(&row[g >> 5], 1u << (g & 31)); }
    if (lane == 0) P.lslot[v] = sl;
}
__global__ void k_lbits_reset(PolyView P, int nhv)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < nhv) P.lslot[P.hv[idx]] = -1;
}
// edges with a touched end -> EH (order kept) with their position in E; alive[e] = 1 for the others
__global__ __launch_bounds__(PB) void k_hote_flags(const int2 *E, int ne, const int *tc, Tri *bsum)
{
    __shared__ Tri lds[16];
    const int e = blockIdx.x * PB + threadIdx.x;
    Tri t{0, 0, 0};
    if (e < ne) { const int2 ed = E[e]; t.a = tc[ed.x] > 0 || tc[ed.y] > 0; }
    Tri tot;
    (void)block_exscan(t, &tot, lds);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(PB) void k_hote_emit(const int2 *E, int ne, const int *tc, const Tri *bpre, int2 *EH, int *EP, unsigned char *alive)
{
    __shared__ Tri lds[16];
    const int e = blockIdx.x * PB + threadIdx.x;
    Tri t{0, 0, 0};
    int2 ed{0, 0};
    if (e < ne) { ed = E[e]; t.a = tc[ed.x] > 0 || tc[ed.y] > 0; alive[e] = t.a ? 0 : 1; }
    Tri tot;
    Tri ex = block_exscan(t, &tot, lds);
    if (t.a) { const int pos = bpre[blockIdx.x].a + ex.a; EH[pos] = ed; EP[pos] = e; }
}
// end of the chunk: the hot edges that are still there revive their old positions; (count, 0, 0) = how many
__global__ __launch_bounds__(PB) void k_hot_revive(const int *EP, int neh, unsigned char *alive, Tri *bsum)
{
    __shared__ Tri lds[16];
    const int k = blockIdx.x * PB + threadIdx.x;
    Tri t{0, 0, 0};
    if (k < neh) { const int p = EP[k]; if (p >= 0) { alive[p] = 1; t.a = 1; } }
    Tri tot;
    (void)block_exscan(t, &tot, lds);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(PB) void k_alive_flags(const unsigned char *alive, int ne, Tri *bsum)
{
    __shared__ Tri lds[16];
    const int e = blockIdx.x * PB + threadIdx.x;
    Tri t{0, 0, 0};
    if (e < ne) t.a = alive[e];
    Tri tot;
    (void)block_exscan(t, &tot, lds);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(PB) void k_alive_emit(const int2 *E, const unsigned char *alive, int ne, const Tri *bpre, int2 *Enew)
{
    __shared__ Tri lds[16];
    const int e = blockIdx.x * PB + threadIdx.x;
    Tri t{0, 0, 0};
    if (e < ne) t.a = alive[e];
    Tri tot;
    Tri ex = block_exscan(t, &tot, lds);
    if (t.a) Enew[bpre[blockIdx.x].a + ex.a] = E[e];
}

// unprocessed = used && !sltn (bslv_poly.c:214-216): triple (flag, 0, 0)
__global__ __launch_bounds__(PB) void k_unproc_flags(PolyView P, int nv, Tri *bsum)
{
    __shared__ Tri lds[16];
    int i = blockIdx.x * PB + threadIdx.x;
    Tri t{0, 0, 0};
    if (i < nv) { unsigned char fl = P.flag[i]; t.a = (fl & F_USED) && !(fl & F_SLTN); }
    Tri tot;
    (void)block_exscan(t, &tot, lds);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(PB) void k_unproc_emit(PolyView P, int nv, const Tri *bpre, int skip, int maxout, int total, int *idx, double *val,
                                                     unsigned char *fl_out, int *parent)
{
    __shared__ Tri lds[16];
    int i = blockIdx.x * PB + threadIdx.x;
    Tri t{0, 0, 0};
    unsigned char fl = 0;
    if (i < nv) { fl = P.flag[i]; t.a = (fl & F_USED) && !(fl & F_SLTN); }
    Tri tot;
    Tri ex = block_exscan(t, &tot, lds);
    if (i >= nv || !t.a) return;
    int pos = bpre[blockIdx.x].a + ex.a;
    if (total > 0) {              // strided sample: maxout elements spread evenly over the `total` unprocessed ones
        long long a = (long long)pos * maxout / total, b = (long long)(pos + 1) * maxout / total;
        if (b == a) return;
        pos = (int)a;
    } else pos -= skip;
    if (pos < 0 || pos >= maxout) return;
    idx[pos] = i;
    fl_out[pos] = fl;
    parent[pos] = P.inc_len[i] > 0 ? P.pool[P.inc_off[i] + P.inc_len[i] - 1] : -1;   // newest facet through it
    for (int k = 0; k < P.d; k++) val[(size_t)pos * P.d + k] = P.X[(size_t)k * P.cap + i];
}
// "children of the newest cuts first" (bslv_poly_unprocessed2, from_end == 3): the unprocessed elements are ranked by the dual
// slot of the cut that created them (the newest facet through them), not by their own slot -- slot numbers follow the order in
// which the cuts of a batch happened to be applied (chunks, the shuffle of the rounds), facet ids the order in which they were found.
// r2f: dual slot of every facet rank.  hist[f] = unprocessed elements whose parent is facet f.
__device__ __forceinline__ int parent_facet(const PolyView &P, int i, const int *__restrict__ r2f)
{
    return P.inc_len[i] > 0 ? r2f[P.pool[P.inc_off[i] + P.inc_len[i] - 1]] : 0;
}
__global__ __launch_bounds__(PB) void k_unproc_hist(PolyView P, int nv, const int *__restrict__ r2f, int *hist)
{
    const int i = blockIdx.x * PB + threadIdx.x;
    if (i >= nv) return;
    const unsigned char fl = P.flag[i];
    if (!(fl & F_USED) || (fl & F_SLTN)) return;
    atomicAdd(&hist[parent_facet(P, i, r2f)], 1);
}
// one workgroup walks the histogram from the newest facet down until `want` elements are covered: out[0] = the lowest facet
// taken, out[1] = elements with a parent facet >= out[0]
__global__ __launch_bounds__(1024) void k_unproc_threshold(const int *__restrict__ hist, int nf, int want, int *out)
{
    __shared__ Tri lds[16];
    __shared__ int s_carry, s_done;
    if (threadIdx.x == 0) { s_carry = 0; s_done = 0; }
    __syncthreads();
    for (int top = nf; top > 0; top -= 1024) {
        const int f = top - 1 - (int)threadIdx.x;              // thread 0 takes the newest facet of the window
        Tri t{f >= 0 ? hist[f] : 0, 0, 0};
        Tri tot;
        const Tri ex = block_exscan(t, &tot, lds);
        const int before = s_carry + ex.a;                     // elements of newer facets
        if (f >= 0 && before < want && before + t.a >= want) { out[0] = f; out[1] = before + t.a; s_done = 1; }
        __syncthreads();
        if (s_done) return;
        if (threadIdx.x == 0) s_carry += tot.a;
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = 0; out[1] = s_carry; }  // fewer than `want` in all
}
__global__ __launch_bounds__(PB) void k_unproc_flags_f(PolyView P, int nv, const int *__restrict__ r2f, const int *__restrict__ thr, Tri *bsum)
{
    __shared__ Tri lds[16];
    const int i = blockIdx.x * PB + threadIdx.x, fmin = thr[0];
    Tri t{0, 0, 0};
    if (i < nv) { const unsigned char fl = P.flag[i]; t.b = (fl & F_USED) && !(fl & F_SLTN); t.a = t.b && parent_facet(P, i, r2f) >= fmin; }      // (.b: all unprocessed elements)
    Tri tot;
    (void)block_exscan(t, &tot, lds);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
// the newest maxout (by slot) of the elements with a parent facet >= thr[0], ascending slot order; parent[] = FACET ids
__global__ __launch_bounds__(PB) void k_unproc_emit_f(PolyView P, int nv, const int *__restrict__ r2f, const int *__restrict__ thr, const Tri *bpre, int skip, int maxout,
                                                       int *idx, double *val, unsigned char *fl_out, int *parent)
{
    __shared__ Tri lds[16];
    const int i = blockIdx.x * PB + threadIdx.x, fmin = thr[0];
    Tri t{0, 0, 0};
    unsigned char fl = 0;
    int pf = -1;
    if (i < nv) { fl = P.flag[i]; if ((fl & F_USED) && !(fl & F_SLTN)) { pf = parent_facet(P, i, r2f); t.a = pf >= fmin; } }
    Tri tot;
    const Tri ex = block_exscan(t, &tot, lds);
    if (i >= nv || !t.a) return;
    const int pos = bpre[blockIdx.x].a + ex.a - skip;
    if (pos < 0 || pos >= maxout) return;
    idx[pos] = i;
    fl_out[pos] = fl;
    parent[pos] = P.inc_len[i] > 0 ? pf : -1;
    for (int k = 0; k < P.d; k++) val[(size_t)pos * P.d + k] = P.X[(size_t)k * P.cap + i];
}
// the unprocessed elements whose parent facet is marked in chosen[] (one byte per dual slot >= f0; older facets: not chosen)
__global__ __launch_bounds__(PB) void k_unproc_flags_c(PolyView P, int nv, const int *__restrict__ r2f, const unsigned char *__restrict__ chosen, int f0, Tri *bsum)
{
    __shared__ Tri lds[16];
    const int i = blockIdx.x * PB + threadIdx.x;
    Tri t{0, 0, 0};
    if (i < nv) {
        const unsigned char fl = P.flag[i];
        t.b = (fl & F_USED) && !(fl & F_SLTN);
        if (t.b) { const int f = parent_facet(P, i, r2f); t.a = f >= f0 && chosen[f - f0]; }
    }
    Tri tot;
    (void)block_exscan(t, &tot, lds);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(PB) void k_unproc_emit_c(PolyView P, int nv, const int *__restrict__ r2f, const unsigned char *__restrict__ chosen, int f0, const Tri *bpre, int skip, int maxout,
                                                       int *idx, double *val, unsigned char *fl_out, int *parent)
{
    __shared__ Tri lds[16];
    const int i = blockIdx.x * PB + threadIdx.x;
    Tri t{0, 0, 0};
    unsigned char fl = 0;
    int pf = -1;
    if (i < nv) { fl = P.flag[i]; if ((fl & F_USED) && !(fl & F_SLTN)) { pf = parent_facet(P, i, r2f); t.a = pf >= f0 && chosen[pf - f0]; } }
    Tri tot;
    const Tri ex = block_exscan(t, &tot, lds);
    if (i >= nv || !t.a) return;
    const int pos = bpre[blockIdx.x].a + ex.a - skip;
    if (pos < 0 || pos >= maxout) return;
    idx[pos] = i;
    fl_out[pos] = fl;
    parent[pos] = pf;
    for (int k = 0; k < P.d; k++) val[(size_t)pos * P.d + k] = P.X[(size_t)k * P.cap + i];
}
__global__ void k_mark(PolyView P, const int *idx, int n, unsigned char bit)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) P.flag[idx[k]] |= bit;
}

// ---------------- dual-side adjacency (poly__update_adjacence on the dual, bslv_poly.c:992-1010) ----------------
// items = live facets, lists = vertices on the facet (sorted); candidate pool of a pair = facets
// through its first mutual vertex (bslv_poly.c:487-491)
struct DualView {
    const int *ids;          // live facet ids, ascending
    const unsigned *foff;    // per facet id
    const int *flen;
    const int *fpool;        // vertex ids
    const unsigned char *flive;
};
__device__ __forceinline__ int isect_count_first(const int *a, int na, const int *b, int nb, int *first)
{
    int i = 0, j = 0, n = 0, f = -1;
    while (i < na && j < nb) {
        int x = a[i], y = b[j];
        if (x == y) { if (n == 0) f = x; n++; }
        i += (x <= y);
        j += (y <= x);
    }
    *first = f;
    return n;
}
__global__ __launch_bounds__(PB) void k_dpair_flags(PolyView P, DualView D, int nm, const PairBlk *blks, unsigned char *pflag, Tri *bsum)
{
    __shared__ Tri lds[16];
    __shared__ int s_row[1024];
    const PairBlk pb = blks[blockIdx.x];
    const int fi = D.ids[pb.i];
    const int ni = D.flen[fi];
    const int *Li = D.fpool + D.foff[fi];
    const bool cached = ni <= 1024;
    if (cached) for (int t = threadIdx.x; t < ni; t += PB) s_row[t] = Li[t];
    __syncthreads();
    const int *A = cached ? s_row : Li;
    const int j = pb.j0 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int fj = -1, nmut = 0, first = -1;
    bool cand = false;
    if (j < nm) {
        fj = D.ids[j];
        nmut = isect_count_first(A, ni, D.fpool + D.foff[fj], D.flen[fj], &first);
        cand = (P.d == 1) || (nmut >= P.d - 1);
    }
    bool adj = cand;
    unsigned long long todo = __ballot(cand && P.d > 1);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int cj = __shfl(fj, src, WAVE);
        const int cn = __shfl(nmut, src, WAVE);
        const int cv = __shfl(first, src, WAVE);
        const int *cL = D.fpool + D.foff[cj];
        const int cnj = D.flen[cj];
        const int *pool = P.pool + P.inc_off[cv];
        const int npool = P.inc_len[cv];
        bool found = false;
        for (int base = 0; base < npool; base += WAVE) {
            int k = base + lane;
            bool hit = false;
            if (k < npool) {
                int w = pool[k];
                if (w != fi && w != cj && D.flive[w] && D.flen[w] >= cn)
                    hit = isect3_count(A, ni, cL, cnj, D.fpool + D.foff[w], D.flen[w]) == cn;
            }
            if (__ballot(hit)) { found = true; break; }
        }
        if (lane == src) adj = !found;
    }
    Tri t{adj ? 1 : 0, 0, 0};
    pflag[(size_t)blockIdx.x * PB + threadIdx.x] = adj ? 1 : 0;
    Tri tot;
    (void)block_exscan(t, &tot, lds);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}


// which facet ranks have a live element on them (the reference clears dual.used when a facet's vertex list empties, bslv_poly.c:686-705):
// one wave per live element walks its incidence list
__global__ __launch_bounds__(PB) void k_live_ranks(PolyView P, int nv, unsigned char *live)
{
    const int i = blockIdx.x * (PB / WAVE) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= nv || !(P.flag[i] & F_USED)) return;
    const int *L = P.pool + P.inc_off[i];
    const int n = P.inc_len[i];
    for (int j = lane; j < n; j += WAVE) live[L[j]] = 1;
}
// measurement helper: overwrite the first nv element slots with synthetic live points (splitmix-like hash)
__global__ void k_bench_fill(PolyView P, int nv, unsigned long long seed)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    for (int k = 0; k < P.d; k++) {
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)i * P.d + k + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        P.X[(size_t)k * P.cap + i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
    }
    P.flag[i] = F_USED;
    P.inc_len[i] = 0;
    P.inc_off[i] = 0;
}
#include "poly_rounds_kernels.inc"
#include "poly_rounds2_kernels.inc"
// debugging aid: a member list never holds an element twice
__device__ void k2v2_check_members(const K2V2 &V, int vs, const int *s_mem, int nm, const int *members, int nzero)
{
    RState *st = const_cast<RState *>(V.st);
    for (int m = threadIdx.x; m < nm; m += blockDim.x)
        for (int m2 = m + 1; m2 < nm; m2++)
            if (s_mem[m] == s_mem[m2] && atomicCAS(&st->err, 0, 4) == 0) {
                int c1 = -1, c2 = -1;
                for (int c = 0; c < nzero; c++) if (members[c] == s_mem[m]) { if (c1 < 0) c1 = c; else c2 = c; }
                st->err_info[0] = vs; st->err_info[1] = m; st->err_info[2] = m2; st->err_info[3] = s_mem[m]; st->err_info[4] = c1; st->err_info[5] = c2; st->err_info[6] = nzero;
            }
}
__device__ __forceinline__ int k2v2_round(const RState *st) { return st->round; }
__device__ __forceinline__ int k2v2_rank_base(const RState *st) { return st->rank_base; }
__device__ __forceinline__ void k2v2_sizes(const RState *st, int &S, int &go, int &nzero, int &nv0, int &ncross)
{
    S = st->S; go = st->go && !r2_halted(st); nzero = st->nzero; nv0 = st->nv; ncross = st->ncross;
}


}  // namespace bslv

using namespace bslv;

struct RoundsBuf;
struct Rounds2Buf;
struct bslv_poly {
    RoundsBuf *rounds = nullptr;      // scratch of the multi-cut path
    Rounds2Buf *rounds2 = nullptr;    // scratch of the device-selected rounds inside a hot chunk (poly_rounds2_host.inc)
    bool rounds2_enabled = true;      // BSLV_NO_ROUNDS2=1 / bslv_poly_debug_set(h, 6, 0): hot chunks go through the single-cut pipeline
    int r2_fuse = 0;                  // (default 0: measured fastest) 1: the classification of a round's new vertices rides in the launch of its prunes (extra workgroups); 2: and the last prune workgroup to finish writes the adjacent pairs (a ticket; measured slower); 0: three launches (BSLV_R2_FUSE / debug_set key 13)
    int r2_fork = 0;                  // 1: the classification of a round's new vertices runs on a second stream BESIDE the round's prunes (they are independent until k_r2_k2emit: the prune reads no class words, the classification writes only the class words of the new vertices), joined by an event before k_r2_k2emit (BSLV_R2_FORK / debug_set key 17)
    bool r2_spec = true;              // rounds are queued one ahead of the host (BSLV_R2_SPEC=0 / debug_set key 12: the host reads every round's mailbox before it queues the next)
    long r2_spec_void = 0;            // rounds that were queued ahead and found the device halted (bslv_poly_rounds2_stats)
    bool r2_share = true;             // (round 4) the cuts of a round may share elements that lie ON their planes: only a MINUS element makes two cuts conflict (poly_rounds2_kernels.inc, "Elements shared by the cuts of a round"); BSLV_R2_SHARE=0 / debug_set key 15: an element belongs to one cut of a round (rounds 2-3)
    bool r2_mis = true;               // rounds take a MAXIMAL independent set from a conflict matrix of the chunk (round 3); BSLV_R2_MIS=0 / debug_set key 11: the local minima of one random order (round 2)
    int chunk_cuts = 1024;             // cuts classified and applied together (bslv_poly_debug_set(h, 7, n); at most 4096)
    int r2_defer = 0;                 // rounds of a chunk stop when one holds fewer cuts than this; what is left is handed BACK to the caller (rc 2) -- see bslv_poly_set_defer
    bool r2_deferred = false;         // (the last run_rounds2 stopped for that reason)
    long r2_deferred_cuts = 0, r2_doomed = 0;
    bool r2_defer_mark = !(getenv("BSLV_DEFER_MARK") && atoi(getenv("BSLV_DEFER_MARK")) == 0);   // elements a handed-back cut will remove are marked processed
    bool snap = false;                // bslv_poly_set_snap / BSLV_POLY_SNAP: the projection sub-band of poly__cut; cuts are then applied one at a time
    unsigned long long *snapped_d = nullptr;
    int r2_rule = 0;                  // 0: average over the rounds of the chunk so far, 1: over the last four rounds (BSLV_R2_RULE)
    int r2_min_cuts = 0;              // rounds go on while they hold at least this many cuts on average (debug_set key 8; 0: until the rounds hold one cut each; -1: always)
    long r2_rounds = 0, r2_cuts = 0, r2_fallback_prunes = 0, r2_declined = 0, r2_chunks = 0, shuffle_seq = 0, r2_late_left = 0, r2_torn_reads = 0;
    long r2_fb_reason[5] = {0, 0, 0, 0, 0}, r2_fb_nm[20] = {0};     // fallback prunes by cause and by log2 of the member count (BSLV_R2_REPORT)
    int batch_mode = 1;               // 0: one cut at a time, 1: rounds of independent cuts
    long rounds_run = 0, conf_pairs = 0, conf_cuts = 0;
    int dense_streak = 0, dense_skip = 0;   // adaptive skipping of the conflict pass (apply_cuts_rounds)
    int d = 0, v2h = 0;
    std::vector<double> c;
    hipStream_t stream = nullptr;
    PolyView P{};
    int nv = 0;                       // primal slots in use
    unsigned poolcap = 0, poolused = 0;
    int2 *E[2] = {nullptr, nullptr};
    int ecap = 0, ne = 0, ecur = 0;
    unsigned char *eflag = nullptr;   // ecap
    int *ecount = nullptr; int ecountcap = 0;   // per crossing edge: list length of its new vertex (k_flags2 -> k_emit2)
    // hot mode (see PolyView): E/ecap/ne/ecur/eflag above then describe the HOT edge list and EP its positions in the
    // full list, which waits in `full` until hot_end() merges the two
    int *EP[2] = {nullptr, nullptr};
    bool hot = false, hot_enabled = true;     // BSLV_NO_HOT=1 turns hot mode off
    struct EdgeSet { int2 *E[2] = {nullptr, nullptr}; int *EP[2] = {nullptr, nullptr}; unsigned char *eflag = nullptr; int ecap = 0, ne = 0, ecur = 0; } full, hotbuf;
    unsigned char *alive = nullptr; int alivecap = 0;     // per edge of the full list: still there at the end of the chunk
    int *hv_d = nullptr; int hvcap = 0;                   // hot elements
    int *lslot_d = nullptr; int lslotcap = 0;             // per element: row of its membership bitmap or -1 (all -1 outside hot mode)
    unsigned *lbits_d = nullptr; size_t lbitscap = 0; int *lnslots_d = nullptr;
    long hot_chunks = 0, hot_elems = 0, hot_edges = 0;
    long n_spec = 0, n_declined = 0, n_k2_fallback = 0, n_single = 0;     // bslv_poly_path_stats
    double tm_hot_begin = 0, tm_seq = 0, tm_hot_end = 0, tm_add_cuts = 0; long tm_seq_cuts = 0;
    double tm_realloc = 0; long n_realloc = 0;                     // ms / count (BSLV_TIMING): element, edge and pool arrays that had to be re-allocated at twice their size
    double tm_prep = 0, tm_classify_wait = 0, tm_newdual = 0;      // ms (BSLV_TIMING): a chunk's host preparation up to its batched classification, the wait for that classification, the dual slots of a batch
    double tm_launch[4] = {0, 0, 0, 0};      // us: queueing round A, k_emit2, all of round B, waiting for the mailbox   // host wall clock (ms), printed at destroy with BSLV_TIMING
    int *members = nullptr;           // cap
    Tri *bsum = nullptr; int bsumcap = 0;
    Tri *bsum2 = nullptr; size_t bsum2cap = 0;
    int *fm_cnt = nullptr, *fm_list = nullptr; size_t fmcap = 0, fmlistcap = 0; int fm_min = 4096; long n_fm = 0; bool member_lists = true;
    // multi-GPU: pair space of large facets dealt to the ranks (k2_multi).  Below ~3e4 elements the two all-gathers cost more than the pair tests
    int largest_facet = 0;            // members of the largest new facet pruned so far (any path)
    std::vector<double> cut_prio;     // bslv_poly_set_cut_priorities: one number per cut of the NEXT bslv_poly_add_cuts (the rounds prefer small values); cleared by that call
    int r2_order = getenv("BSLV_R2_ORDER") ? atoi(getenv("BSLV_R2_ORDER")) : 0;      // priority of a chunk's cuts in the rounds: 0 pseudo-random shuffle, 1 ascending / 2 descending cut priority (where the caller gave one)
    bool k2_noflags = getenv("BSLV_K2_NOFLAGS") != nullptr; long n_noflag_prunes = 0;     // large-facet prunes emit by testing the listed pair blocks again instead of keeping a flag per pair
    int shard_min = 32768; long n_sharded = 0; int2 *shard_e = nullptr; size_t shardcap = 0;
    int *nzlist = nullptr; size_t nzcap = 0;         // pair blocks with an adjacent pair (+ their count behind the list)      // facet-major member lists of a large new facet (k_fm_*)       // chunk totals of the two-level scan (k_scan_chunks)
    Tri *totals = nullptr;            // device, 4 entries
    int *counters = nullptr;          // device, 4 ints
    Tri *totals_h = nullptr; int *counters_h = nullptr;   // pinned
    Mail *mail_h = nullptr, *mail_d = nullptr;          // mapped pinned mailbox (4 entries: round A, -, prune x2)
    Mail mail_v[4];                                     // verified copies (wait_mail): what the host reads
    long mail_torn_reads = 0;
    Mail *k2mail_d = nullptr;                           // device memory: where k2_fused leaves its result (2 slots)
    int *fstamp = nullptr, *flocal = nullptr, *nlocal = nullptr; int fcap = 0;   // local facet ids of the cut in flight
    unsigned long long *bits = nullptr; size_t bitscap = 0;                     // local incidence bit matrix
    int mailseq = 0;
    long cutseq = 0;
    size_t k2_lds = 0;                // dynamic LDS bytes granted to k2_fused
    int *ne_dev = nullptr;            // edge count as the device knows it (written by k2_fused)
    bool pend_k2 = false;             // a k2_fused is in flight: ne is an upper bound
    int pend_seq = 0, pend_slot = 2, pend_ebase = 0, pend_nm = 0, pend_stamp = 0, pend_nzero = 0, pend_nv0 = 0, pend_ncross = 0; long long pend_len_ub = 0;
    int k2flip = 0;                   // the prune mailbox alternates between two slots: one result may wait while the next is queued
    bool speculate = true;            // queue round B before the host has seen round A (BSLV_NO_SPEC=1 turns it off)
    int cross_ub = CROSS_UB;          // (BSLV_CROSS_UB=n: test hook, small values force the decline-and-rerun path)
    CutDev *cutdev = nullptr;         // CRING verdicts of k_scan2
    int *abort_d = nullptr;           // set by a k2_fused that needs its fallback: everything queued behind it declines
    int *fcount = nullptr;            // per facet rank: member lists it occurs in (k2_fused; zero between cuts)
    int *zlist = nullptr, *zrows = nullptr;     // ZMarks: CRING x ZMAX element ids, ZMAX x fcap facet stamps
    int pre_f = -1, pre_slot = 0, pre_nv = 0, pre_seq = 0;   // halfspace already classified (queued behind the previous cut's k_emit2)
    unsigned long long *k2dbg = nullptr;   // BSLV_K2_DEBUG=1: per-phase clock sums of k2_fused (100 MHz ticks), printed at destroy
    FILE *cutlog = nullptr;           // BSLV_CUT_LOG=<file>: one line per cut (nv ne nminus nzero zero_ub nsurv ncross newlen), profiling aid
    PairBlk *blks = nullptr; int blkcap = 0;
    unsigned char *pflag = nullptr; size_t pflagcap = 0;
    // dual side (host)
    std::vector<double> Y, hp;        // nf x d, nf x (d+1)
    std::vector<unsigned char> fapplied, fideal;
    // incidence lists hold RANKS (order of application), so appending the newest cut keeps them sorted even
    // when cuts of a batch are applied out of index order; facet_of_rank maps back to dual slot ids
    std::vector<int> facet_of_rank;
    int nf = 0;
    bool initialised = false;
    std::vector<int> queue;
    std::vector<int> dual_edges;      // pairs, filled by dual_adjacency
    // stats
    long pair_tests = 0, new_vertices = 0, cuts_applied = 0;
    // scratch for batched classify
    double *hps_d = nullptr; int hpscap = 0; std::vector<double> hps_stage;
    unsigned long long *clsw = nullptr; size_t clswcap = 0;
    unsigned *anyminus = nullptr; int anycap = 0;      // bit b%32 of word b/32: some element violates halfspace b
    int *idx_d = nullptr; double *val_d = nullptr; unsigned char *fl_d = nullptr; int outcap = 0;
    int *par_d = nullptr; int parcap = 0;
    int *r2f_d = nullptr; int r2fcap = 0, r2f_n = 0;   // dual slot of every facet rank (device copy of facet_of_rank, extended on demand)
    int *fhist_d = nullptr; int fhistcap = 0;           // unprocessed elements per parent facet + 4 ints of threshold
    unsigned char *chosen_d = nullptr; size_t chosencap = 0;   // bslv_poly_children_of: one byte per dual slot from the oldest chosen one on
};

static void v2h_map(const bslv_poly *h, const double *v, int is_dir, double *hp)
{
    const int d = h->d;
    switch (h->v2h) {
    case 0:   // cone_polar, bslv_poly.c:30-39
        for (int j = 0; j < d; j++) hp[j] = v[j];
        hp[d] = is_dir ? 0.0 : -1.0;
        break;
    case 1:   // lowerV2upperH, bslv_algs.c:287-305
        if (is_dir) { for (int j = 0; j < d; j++) hp[j] = 0.0; hp[d] = -1.0; }
        else {
            hp[d - 1] = 1.0;
            for (int j = 0; j < d - 1; j++) { hp[j] = v[j]; hp[d - 1] -= h->c[j] * hp[j]; }
            hp[d] = v[d - 1];
        }
        break;
    default:  // upperV2lowerH, bslv_algs.c:307-313
        hp[d - 1] = is_dir ? 0.0 : -1.0;
        for (int j = 0; j < d - 1; j++) hp[j] = v[j] - v[d - 1] * h->c[j];
        hp[d] = -v[d - 1];
        break;
    }
}

// *p (oldn elements in use) becomes an array of newn elements; what is not copied is filled like a fresh allocation (zeros;
// BSLV_FILL: see malloc0 in common.h).  zero_tail marks the arrays whose kernels RELY on the zeros (stamps, counters, keep marks).
template <typename T>
static int grow_impl(const char *name, int line, T **p, size_t oldn, size_t newn, hipStream_t s, bool zero_tail = false)
{
    T *q = nullptr;
    HIP_TRY(hipMalloc((void **)&q, newn * sizeof(T)));
    debug_note_alloc(name, (const void *)q, newn * sizeof(T), __FILE__, line);
    const size_t keep = (*p && oldn) ? oldn : 0;
    if (keep) HIP_TRY(hipMemcpyAsync(q, *p, keep * sizeof(T), hipMemcpyDeviceToDevice, s));
    if (newn > keep) HIP_TRY(hipMemsetAsync(q + keep, zero_tail ? 0 : debug_fill(), (newn - keep) * sizeof(T), s));
    HIP_TRY(hipStreamSynchronize(s));
    if (*p) (void)hipFree(*p);
    *p = q;
    return 0;
}
#define grow(p, ...) grow_impl(#p, __LINE__, p, __VA_ARGS__)

struct ReallocClock {
    bslv_poly *h; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit ReallocClock(bslv_poly *hh) : h(hh) {}
    ~ReallocClock();
};
ReallocClock::~ReallocClock() { h->tm_realloc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); h->n_realloc++; }
static int ensure_vcap(bslv_poly *h, int need)
{
    if (need < 0 || need > 0x7FFFFF00 / 2) { set_error("polyhedron too large: more than 2^30 elements"); return BSLV_E_CAPACITY; }
    if (need <= h->P.cap) return 0;
    ReallocClock clock(h);
    int ncap = (int)std::min<long long>(std::max<long long>(need, std::max<long long>(1024, 2ll * h->P.cap)), 0x7FFFFF00 / 2);
    PolyView &P = h->P;
    // SoA coordinates: re-stride
    double *X = nullptr;
    HIP_TRY(malloc0(&X, (size_t)h->d * ncap * sizeof(double)));
    for (int k = 0; k < h->d && P.X && h->nv; k++)
        HIP_TRY(hipMemcpyAsync(X + (size_t)k * ncap, P.X + (size_t)k * P.cap, (size_t)h->nv * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (P.X) (void)hipFree(P.X);
    P.X = X;
    int rc;
    if ((rc = grow(&P.flag, h->nv, ncap, h->stream, true))) return rc;
    if ((rc = grow(&P.cls, h->nv, ncap, h->stream))) return rc;   // classes of the cut in flight survive a re-allocation
    if ((rc = grow(&P.inc_off, h->nv, ncap, h->stream))) return rc;
    if ((rc = grow(&P.inc_len, h->nv, ncap, h->stream))) return rc;
    if ((rc = grow(&P.capx, h->nv, ncap, h->stream, true))) return rc;            // (zero = no slack recorded: relied upon)
    if ((rc = grow(&h->members, (size_t)P.cap, ncap, h->stream))) return rc;     // kept: a prune waiting for its fallback still needs its members
    {   // membership-bitmap slots: -1 everywhere except on the long elements of a hot chunk
        const int old = h->lslotcap;
        if ((rc = grow(&h->lslot_d, (size_t)old, (size_t)ncap, h->stream))) return rc;
        HIP_TRY(hipMemsetAsync(h->lslot_d + old, 0xFF, (size_t)(ncap - old) * sizeof(int), h->stream));
        h->lslotcap = ncap;
        if (P.lslot) P.lslot = h->lslot_d;
    }
    P.cap = ncap;
    return 0;
}
static int ensure_pool(bslv_poly *h, size_t need)
{
    if (need <= h->poolcap) return 0;
    if (need > 0xF0000000ull) { set_error("incidence pool exceeds 32-bit offsets"); return BSLV_E_CAPACITY; }
    ReallocClock clock(h);
    size_t ncap = std::min<size_t>(0xF0000000ull, std::max(need, std::max<size_t>(1 << 16, (size_t)h->poolcap * 2)));
    int rc;
    if ((rc = grow(&h->P.pool, h->poolused, ncap + 64, h->stream))) return rc;      // + slack for load_list
    if ((rc = grow(&h->P.keep, h->poolused, ncap, h->stream, true))) return rc;
    h->poolcap = (unsigned)ncap;
    return 0;
}
static int ensure_ecap(bslv_poly *h, int need)
{
    // edge, element and pool indices are 32-bit: refuse instead of wrapping around (a 10-dimensional degenerate image gets there)
    if (need < 0 || need > 0x7FFFFF00 / 2) { set_error("polyhedron too large: more than 2^30 edges"); return BSLV_E_CAPACITY; }
    if (need <= h->ecap) return 0;
    ReallocClock clock(h);
    int ncap = (int)std::min<long long>(std::max<long long>(need, std::max<long long>(4096, 2ll * h->ecap)), 0x7FFFFF00 / 2);
    int rc;
    if ((rc = grow(&h->E[h->ecur], h->ne, ncap, h->stream))) return rc;
    if ((rc = grow(&h->E[1 - h->ecur], 0, ncap, h->stream))) return rc;
    if ((rc = grow(&h->eflag, 0, ncap, h->stream))) return rc;
    if (h->hot) {
        if ((rc = grow(&h->EP[h->ecur], h->ne, ncap, h->stream))) return rc;
        if ((rc = grow(&h->EP[1 - h->ecur], 0, ncap, h->stream))) return rc;
    }
    h->ecap = ncap;
    return 0;
}
static int ensure_bsum(bslv_poly *h, int nb)
{
    if (nb <= h->bsumcap) return 0;
    int ncap = std::max(nb, std::max(1024, h->bsumcap * 2));
    int rc;
    if ((rc = grow(&h->bsum, 0, ncap, h->stream))) return rc;
    h->bsumcap = ncap;
    return 0;
}

// run the three-kernel scan tail: bsum -> exclusive prefixes, totals -> host
static int scan_totals(bslv_poly *h, int nb, Tri *tot_out)
{
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, h->stream, h->bsum, nb, h->totals);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h->totals_h, h->totals, sizeof(Tri), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *tot_out = h->totals_h[0];
    return 0;
}

// upload B halfspaces (host layout (d+1): normal, alpha) in the kernel's layout (d+3): + alpha+EPS, alpha-EPS
static int upload_hps(bslv_poly *h, const double *hp_host, int B)
{
    const int d = h->d;
    int rc;
    if (B > h->hpscap) { int nc = std::max(B, std::max(512, h->hpscap * 2)); if ((rc = grow(&h->hps_d, 0, (size_t)nc * (MAXD + 3), h->stream))) return rc; h->hpscap = nc; }
    h->hps_stage.resize((size_t)B * (d + 3));
    for (int b = 0; b < B; b++) {
        const double *src = hp_host + (size_t)b * (d + 1);
        double *dst = &h->hps_stage[(size_t)b * (d + 3)];
        for (int k = 0; k <= d; k++) dst[k] = src[k];
        dst[d + 1] = src[d] + POLY_EPS;
        dst[d + 2] = src[d] - POLY_EPS;
    }
    HIP_TRY(hipMemcpyAsync(h->hps_d, h->hps_stage.data(), h->hps_stage.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return 0;
}

static int new_dual(bslv_poly *h, const double *val, int ideal)
{
    int f = h->nf++;
    h->Y.insert(h->Y.end(), val, val + h->d);
    h->hp.resize((size_t)h->nf * (h->d + 1));
    v2h_map(h, val, ideal, &h->hp[(size_t)f * (h->d + 1)]);
    h->fapplied.push_back(1);
    h->fideal.push_back(ideal ? 1 : 0);
    return f;
}

// spin on the host mailbox (the scan tail writes it through mapped pinned memory); falls back to a
// stream synchronise so that a device fault surfaces as an error instead of a hang
// The verified copy lands in h->mail_v[slot]: nothing reads the mapped memory itself.
static int wait_mail(bslv_poly *h, int slot, int seq)
{
    volatile Mail *m = h->mail_h + slot;
    long torn = 0, drained = 0;
    for (long spin = 0; !mail_try_read(m, seq, &h->mail_v[slot], &torn); spin++) {
        if ((spin & 0xFFFF) == 0xFFFF) {
            hipError_t e = hipStreamQuery(h->stream);
            if (e == hipSuccess && ++drained > 64) {            // the stream has long drained: nothing more will arrive
                set_error("poly engine: mailbox %d never reached a consistent state for seq %d (seq there: %d, %ld torn reads)", slot, seq, (int)m->seq, torn);
                return BSLV_E_STATE;
            }
            if (e != hipSuccess && e != hipErrorNotReady) { set_error("poly engine: stream error %s", hipGetErrorString(e)); return BSLV_E_NODEVICE; }
        }
    }
    h->mail_torn_reads += torn;
    if (torn) { static const bool rep = getenv("BSLV_R2_REPORT") != nullptr; if (rep) fprintf(stderr, "poly: mailbox %d seq %d arrived before its content: %ld reads repeated\n", slot, seq, torn); }
    return 0;
}

// the multi-kernel adjacency prune (facets too large for k2_fused, or whose bit matrix does not fit in LDS);
// members[0..nm) complete.  Appends the adjacent pairs at E[ecur][h->ne..) and advances h->ne.
// stamp: fstamp value of this prune, above every state k2_fused may have left for the same cut.
static int k2_multi(bslv_poly *h, int nm, long long len_ub, int stamp)
{
    hipStream_t s = h->stream;
    int rc;
    const long long nbp = pair_G(nm - 1);
    h->largest_facet = std::max(h->largest_facet, nm);
    if (nbp > 0x7FFFFFF0ll) { set_error("new facet has too many elements (%d): more than 2^31 blocks of 256 pairs (block indices are 32-bit)", nm); return BSLV_E_CAPACITY; }
    // one flag byte per pair between the test and the ordered emission -- unless that array is out of reach (k_pair_retest_emit below)
    auto ensure_pflag = [&]() -> int {
        if ((size_t)nbp * PB > h->pflagcap) { size_t nc = std::max((size_t)nbp * PB, h->pflagcap * 2); if ((rc = grow(&h->pflag, 0, nc, s))) return rc; h->pflagcap = nc; }
        return 0;
    };
    bool noflags = false;
    if ((rc = ensure_bsum(h, (int)nbp + 1))) return rc;
    // local incidence bit matrix: at most one local id per list entry of the members
    const int nranks = (int)h->facet_of_rank.size();
    const int W = (int)((std::min<long long>(len_ub, nranks) + 63) / 64);
    if ((size_t)W * nm > h->bitscap) { size_t nc = std::max((size_t)W * nm, h->bitscap * 2); if ((rc = grow(&h->bits, 0, nc, s))) return rc; h->bitscap = nc; }
    const size_t lds_bits = (size_t)W * 5 * sizeof(unsigned long long);
    bool used_list = false, shard = false;
    if (lds_bits <= 48 * 1024) {
        HIP_TRY(hipMemsetAsync(h->nlocal, 0, sizeof(int), s));
        const int nbm = (nm * LPM + PB - 1) / PB;
        hipLaunchKernelGGL(k_local_ids, dim3(nbm), dim3(PB), 0, s, h->P, h->members, nm, W, stamp, h->fstamp, h->flocal, h->nlocal, h->bits);
        hipLaunchKernelGGL(k_build_bits, dim3(nbm), dim3(PB), 0, s, h->P, h->members, nm, W, h->flocal, h->bits);
        // large facets: the row-tiled pair kernel (2-D grid), the list of pair blocks that hold an edge, and member lists by
        // facet, so that confirming an edge scans one facet's elements instead of all nm
        const size_t lds_tiled = ((size_t)PTI * W + (size_t)W * (PB + PTI) + 4 * (size_t)W) * sizeof(unsigned long long);
        const int ngroups = (nm - 1 + PTI - 1) / PTI;
        const bool tiled = nm >= h->fm_min && h->d > 1 && lds_tiled <= 48 * 1024;      // (more than 65535 row groups: several launches, below)
        const bool fm = tiled && h->member_lists && len_ub <= (1ll << 30);
        // large facets: no flag array at all from 4 GiB of flags on (bslv_poly_debug_set key 16 / BSLV_K2_NOFLAGS=1: always, for the tests);
        // not when the pair space is dealt to the ranks (each rank's share of the flags is a fraction)
        noflags = tiled && !(bslv_dist_world() > 1 && nm >= h->shard_min) && (h->k2_noflags || (size_t)nbp * PB >= ((size_t)4 << 30));
        if (!noflags && (rc = ensure_pflag())) return rc;
        if (tiled) {
            if ((size_t)nbp > h->nzcap) { size_t nc = std::max((size_t)nbp, h->nzcap * 2); if (h->nzlist) (void)hipFree(h->nzlist); h->nzlist = nullptr; HIP_TRY(malloc0s(&h->nzlist, (nc + 1) * sizeof(int), s)); h->nzcap = nc; }
            HIP_TRY(hipMemsetAsync(h->nzlist + h->nzcap, 0, sizeof(int), s));
        }
        if (fm) {
            h->n_fm++;
            const int nf = W * 64;
            if ((size_t)(3 * nf) > h->fmcap) { size_t nc = std::max((size_t)(3 * nf), h->fmcap * 2); if ((rc = grow(&h->fm_cnt, 0, nc, s))) return rc; h->fmcap = nc; }
            if ((size_t)len_ub + 64 > h->fmlistcap) { size_t nc = std::max((size_t)len_ub + 64, h->fmlistcap * 2); if ((rc = grow(&h->fm_list, 0, nc, s))) return rc; h->fmlistcap = nc; }
            HIP_TRY(hipMemsetAsync(h->fm_cnt, 0, (size_t)nf * sizeof(int), s));
            const long long nwords = (long long)W * nm;
            const unsigned nbw = (unsigned)((nwords + PB - 1) / PB);
            hipLaunchKernelGGL(k_fm_count, dim3(nbw), dim3(PB), 0, s, (const unsigned long long *)h->bits, nm, W, h->fm_cnt);
            hipLaunchKernelGGL(k_fm_scan, dim3(1), dim3(1024), 0, s, (const int *)h->fm_cnt, h->fm_cnt + nf, h->fm_cnt + 2 * nf, nf);
            hipLaunchKernelGGL(k_fm_fill, dim3(nbw), dim3(PB), 0, s, (const unsigned long long *)h->bits, nm, W, h->fm_cnt + 2 * nf, h->fm_list);
        }
        if (tiled) {
            // Multi-GPU: the PAIR SPACE of a large facet is dealt to the ranks -- contiguous row groups of equal pair
            // count -- every rank tests its share, and the adjacent pairs found (a few per element, not the nm^2 / 2 flags) are
            // all-gathered and appended in rank order = in the order of the virtual pair blocks, i.e. the edge list is the one a
            // single GPU writes.  Everything else of the cut stays replicated.
            int g0 = 0, g1 = ngroups;
            const int world = bslv_dist_world(), rank = bslv_dist_rank();
            shard = world > 1 && nm >= h->shard_min;
            if (shard) {
                const double total = 0.5 * (double)(nm - 1) * (double)nm;
                auto before = [&](int g) { const double r = std::min((double)g * PTI, (double)(nm - 1)); return r * (double)(nm - 1) - 0.5 * r * (r - 1.0); };   // pairs in the rows of groups < g
                auto bound = [&](int k) { if (k <= 0) return 0; if (k >= world) return ngroups; int lo = 0, hi = ngroups; const double want = total * k / world;
                                          while (lo < hi) { const int mid = (lo + hi) / 2; if (before(mid) < want) lo = mid + 1; else hi = mid; } return lo; };
                g0 = bound(rank); g1 = bound(rank + 1);
                HIP_TRY(hipMemsetAsync(h->bsum, 0, (size_t)nbp * sizeof(Tri), s));          // the blocks of the other ranks: no pair
                h->n_sharded++;
            }
            for (int ga = g0; ga < g1; ga += 65535)          // (gridDim.y holds 65535 row groups: facets of more than 65535 * PTI elements take several launches)
                hipLaunchKernelGGL(k_pair_flags_tiled, dim3((unsigned)((nm - 1 + PB - 1) / PB), (unsigned)std::min(65535, g1 - ga)), dim3(PB), lds_tiled, s, h->d, h->bits, nm, W, noflags ? (unsigned char *)nullptr : h->pflag, h->bsum,
                                   fm ? (const int *)h->fm_cnt : (const int *)nullptr, fm ? (const int *)(h->fm_cnt + W * 64) : (const int *)nullptr, fm ? (const int *)h->fm_list : (const int *)nullptr,
                                   h->nzlist, h->nzlist + h->nzcap, ga);
        }
        else {
            // one-dimensional grid of pair blocks: at most 2^32 work-items (the runtime wraps a larger grid silently)
            if ((long long)nbp * PB >= (1ll << 32)) { set_error("new facet has too many elements (%d) for the one-dimensional pair launch", nm); return BSLV_E_CAPACITY; }
            hipLaunchKernelGGL(k_pair_flags_bits, dim3((unsigned)nbp), dim3(PB), lds_bits, s, h->d, h->bits, nm, W, h->pflag, h->bsum,
                               (const int *)nullptr, (const int *)nullptr, (const int *)nullptr, (int *)nullptr, (int *)nullptr);
        }
        used_list = tiled;
    } else {    // enormous local facet sets: sorted-list version
        if ((rc = ensure_pflag())) return rc;
        if ((long long)nbp * PB >= (1ll << 32)) { set_error("new facet has too many elements (%d) for the one-dimensional pair launch", nm); return BSLV_E_CAPACITY; }
        hipLaunchKernelGGL(k_pair_flags, dim3((unsigned)nbp), dim3(PB), 0, s, h->P, h->members, nm, (const PairBlk *)nullptr, h->pflag, h->bsum);
    }
    const int seq = ++h->mailseq;
    if (nbp <= (1 << 16))
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, s, h->bsum, (int)nbp, h->totals + 2, h->mail_d + 2, (const int *)nullptr, seq);
    else {
        const int nch = (int)((nbp + 1023) / 1024);
        if ((size_t)nch > h->bsum2cap) { size_t nc = std::max((size_t)nch, h->bsum2cap * 2); if ((rc = grow(&h->bsum2, 0, nc, s))) return rc; h->bsum2cap = nc; }
        hipLaunchKernelGGL(k_scan_chunks, dim3(nch), dim3(1024), 0, s, h->bsum, (int)nbp, h->bsum2);
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, s, h->bsum2, nch, h->totals + 2, h->mail_d + 2, (const int *)nullptr, seq);
        hipLaunchKernelGGL(k_scan_add, dim3(nch), dim3(1024), 0, s, h->bsum, (int)nbp, (const Tri *)h->bsum2);
    }
    HIP_TRY(hipGetLastError());
    if ((rc = wait_mail(h, 2, seq))) return rc;
    const Tri tp = h->mail_v[2].t;
    if (shard) {
        // this rank's pairs into a staging list, counts and pairs all-gathered (8 bytes per pair, carried as the bit pattern of a double)
        const int world = bslv_dist_world();
        std::vector<double> cnt_all((size_t)world), mine((size_t)std::max(tp.a, 1));
        // what only this rank does before the exchange; a failure here still enters the all-gather, with a negative count as the
        // status word, so that the other ranks return an error instead of waiting for ever
        auto local = [&]() -> int {
            if ((size_t)tp.a > h->shardcap) { const size_t nc = std::max<size_t>((size_t)tp.a, std::max<size_t>(4096, h->shardcap * 2)); if ((rc = grow(&h->shard_e, 0, nc, s))) return rc; h->shardcap = nc; }
            if (tp.a > 0)
                hipLaunchKernelGGL(k_pair_emit_list, dim3(4096), dim3(PB), 0, s, h->members, nm, (const int *)h->nzlist, (const int *)(h->nzlist + h->nzcap), (const unsigned char *)h->pflag,
                                   (const Tri *)h->bsum, h->shard_e, 0, (int *)nullptr);
            HIP_TRY(hipGetLastError());
            if (tp.a > 0) HIP_TRY(hipMemcpyAsync(mine.data(), h->shard_e, (size_t)tp.a * sizeof(int2), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            return 0;
        };
        const int rc_local = local();
        const double my_cnt = rc_local ? -(double)rc_local : (double)tp.a;
        if ((rc = bslv_dist_allgather(&my_cnt, cnt_all.data(), 1))) return rc;
        for (int r = 0; r < world; r++)
            if (cnt_all[r] < 0) {
                if (r != bslv_dist_rank()) set_error("rank %d reported error %d in its share of a sharded adjacency prune; this rank stops with it", r, (int)-cnt_all[r]);
                return r == bslv_dist_rank() ? rc_local : BSLV_E_STATE;
            }
        long long total = 0; int maxc = 0;
        for (int r = 0; r < world; r++) { total += (long long)cnt_all[r]; maxc = std::max(maxc, (int)cnt_all[r]); }
        if (total > 0) {
            if ((long long)h->ne + total > 0x3FFFFF00ll) { set_error("polyhedron too large: more than 2^30 edges"); return BSLV_E_CAPACITY; }
            mine.resize((size_t)maxc, 0.0);
            std::vector<double> all((size_t)maxc * world), packed((size_t)total);
            if ((rc = bslv_dist_allgather(mine.data(), all.data(), maxc))) return rc;
            size_t at = 0;
            for (int r = 0; r < world; r++) { const size_t c = (size_t)cnt_all[r]; memcpy(packed.data() + at, all.data() + (size_t)r * maxc, c * sizeof(double)); at += c; }
            if ((rc = ensure_ecap(h, (int)(h->ne + total)))) return rc;
            HIP_TRY(hipMemcpyAsync(h->E[h->ecur] + h->ne, packed.data(), (size_t)total * sizeof(int2), hipMemcpyHostToDevice, s));
            if (h->EP[h->ecur]) HIP_TRY(hipMemsetAsync(h->EP[h->ecur] + h->ne, 0xFF, (size_t)total * sizeof(int), s));     // (-1: no parent, as k_pair_emit_list writes)
            HIP_TRY(hipStreamSynchronize(s));                                                                                 // (the host vectors go out of scope)
            h->ne += (int)total;
        }
        return 0;
    }
    if (tp.a > 0) {
        if ((long long)h->ne + tp.a > 0x3FFFFF00ll) { set_error("polyhedron too large: more than 2^30 edges"); return BSLV_E_CAPACITY; }
        if ((rc = ensure_ecap(h, h->ne + tp.a))) return rc;
        if (noflags) {
            const int Wn = (int)((std::min<long long>(len_ub, (long long)h->facet_of_rank.size()) + 63) / 64);
            const bool fmn = h->member_lists && len_ub <= (1ll << 30);
            hipLaunchKernelGGL(k_pair_retest_emit, dim3(4096), dim3(PB), (size_t)Wn * 5 * sizeof(unsigned long long), s, h->d, (const unsigned long long *)h->bits, nm, Wn,
                               fmn ? (const int *)h->fm_cnt : (const int *)nullptr, fmn ? (const int *)(h->fm_cnt + Wn * 64) : (const int *)nullptr, fmn ? (const int *)h->fm_list : (const int *)nullptr,
                               (const int *)h->members, (const int *)h->nzlist, (const int *)(h->nzlist + h->nzcap), (const Tri *)h->bsum, h->E[h->ecur], h->ne, h->EP[h->ecur]);
            h->n_noflag_prunes++;
        } else if (used_list)
            hipLaunchKernelGGL(k_pair_emit_list, dim3(4096), dim3(PB), 0, s, h->members, nm, (const int *)h->nzlist, (const int *)(h->nzlist + h->nzcap), (const unsigned char *)h->pflag,
                               (const Tri *)h->bsum, h->E[h->ecur], h->ne, h->EP[h->ecur]);
        else
        hipLaunchKernelGGL(k_pair_emit, dim3((unsigned)nbp), dim3(PB), 0, s, h->members, nm, (const PairBlk *)nullptr, h->pflag, h->bsum,
                           h->E[h->ecur], h->ne, h->EP[h->ecur]);
        HIP_TRY(hipGetLastError());
        h->ne += tp.a;
    }
    return 0;
}

// waits for the adjacency prune still in flight (k2_fused of the previous cut) and books its edges; when the
// kernel reported that its bit matrix did not fit (or the facet is too large for it), runs the multi-kernel
// prune instead.  *redo (may be NULL) is set in that case: whatever was launched behind the failed kernel saw an
// incomplete edge list (round B of the next cut, if queued speculatively, skipped itself: abort flag).
static int settle_k2(bslv_poly *h, bool *redo = nullptr)
{
    if (redo) *redo = false;
    if (!h->pend_k2) return 0;
    int rc;
    Tri tp;
    if (h->mail_v[0].k2seq == h->pend_seq) tp = h->mail_v[0].k2t;          // came with the mailbox of the round A behind it
    else {
        hipLaunchKernelGGL(k_forward_mail, dim3(1), dim3(1), 0, h->stream, (const Mail *)(h->k2mail_d + (h->pend_slot - 2)), h->mail_d + h->pend_slot);
        HIP_TRY(hipGetLastError());
        if ((rc = wait_mail(h, h->pend_slot, h->pend_seq))) return rc;
        tp = h->mail_v[h->pend_slot].t;
    }
    h->pend_k2 = false;
    h->ne = h->pend_ebase;
    if (tp.b == 1) {
        h->n_k2_fallback++;
        if (redo) *redo = true;
        HIP_TRY(hipMemsetAsync(h->abort_d, 0, sizeof(int), h->stream));
        if (h->pend_ncross > 0)
            hipLaunchKernelGGL(k_iota_members, dim3((h->pend_ncross + 255) / 256), dim3(256), 0, h->stream, h->members, h->pend_nzero, h->pend_nv0, h->pend_ncross);
        return k2_multi(h, h->pend_nm, h->pend_len_ub, h->pend_stamp);
    }
    if (tp.b == 0) h->ne += tp.a;          // (2: the kernel had nothing to do)
    return 0;
}
static int next_counter_slot(bslv_poly *h)
{
    const int cslot = (int)(h->cutseq % CRING);
    // entering a quarter of the ring clears the quarter two ahead for its next use: the slots of the cuts in flight (the current
    // one and the pre-classified next one) are among the last few handed out, at least a quarter away from what is cleared
    constexpr int QR = CRING / 4;
    if (cslot % QR == 0 && h->cutseq > 0) {
        int *ahead = h->counters + (size_t)(((cslot / QR + 2) % 4) * QR) * CSTRIDE;
        if (hipMemsetAsync(ahead, 0, (size_t)QR * CSTRIDE * sizeof(int), h->stream) != hipSuccess) return -1;
    }
    h->cutseq++;
    return cslot;
}
template <class... A>
static void launch_emit2(int d, dim3 grid, hipStream_t s, A... a)
{
    switch (d) {
#define CASE(D) case D: hipLaunchKernelGGL(k_emit2<D>, grid, dim3(PB), 0, s, a...); break;
        CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
#undef CASE
    default: hipLaunchKernelGGL(k_emit2<0>, grid, dim3(PB), 0, s, a...); break;
    }
}

// one cut on the device; *rc = 0 cut applied, 1 redundant.  Five launches:
//   [k_classify] k_flags2 k_scan2 | k_emit2 [k_classify of next_f] k2_fused
// Round B is queued BEFORE the host has read the result of round A (speculative launch): k_scan2 leaves the
// verdict -- redundant? capacities? -- on the device (CutDev), the kernels of round B read their sizes from there and
// return at once when the cut does not go ahead; the host reads the same numbers from its mailbox, reaches the same
// verdict, and only when a capacity was short grows it and launches round B again with host arguments.
// The adjacency prune stays in flight: its pairs land behind the edges of this cut on the device, the next cut
// reads the edge count from there (ne_dev) and the host books it at its next wait (settle_k2).  The classification
// of the next halfspace is queued between k_emit2 and k2_fused, where its launch latency hides.
// Callers finish a sequence of cuts with settle_k2().
static int do_cut(bslv_poly *h, int f, int *rc_out, int next_f = -1)
{
    const int d = h->d, nv0 = h->nv;
    hipStream_t s = h->stream;
    if (h->snap) {                       // every classification sees the coordinates the cuts before it left: none ahead of its turn
        next_f = -1;
        if (!h->snapped_d) { int rc0 = grow(&h->snapped_d, 0, 1, s, true); if (rc0) return rc0; }
    }
    Hp hp, hn;
    memset(&hp, 0, sizeof(hp));
    memcpy(hp.h, &h->hp[(size_t)f * (d + 1)], (d + 1) * sizeof(double));
    memset(&hn, 0, sizeof(hn));
    if (next_f >= 0) memcpy(hn.h, &h->hp[(size_t)next_f * (d + 1)], (d + 1) * sizeof(double));
    int rc;
    {   // arrays indexed by facet rank: this cut may add one
        const int nranks = (int)h->facet_of_rank.size() + 1;
        if (nranks > h->fcap) {
            int nc = std::max(nranks + 1024, h->fcap * 2);
            if ((rc = grow(&h->fstamp, (size_t)h->fcap, (size_t)nc, s, true))) return rc;
            if ((rc = grow(&h->flocal, 0, (size_t)nc, s))) return rc;
            if ((rc = grow(&h->zrows, 0, (size_t)ZMAX * nc, s, true))) return rc;      // fresh zeros: below every stamp
            if ((rc = grow(&h->fcount, 0, (size_t)nc, s, true))) return rc;
            if (!h->nlocal && (rc = grow(&h->nlocal, 0, 4, s, true))) return rc;
            h->fcap = nc;
            h->pre_f = -1;                           // (its stamps went with the old rows)
        }
    }
    bool spec = h->speculate;
    if (spec) {   // head-room that lets the device say yes: CROSS_UB new elements with their lists
        if ((rc = ensure_vcap(h, nv0 + h->cross_ub))) return rc;
        if ((rc = ensure_pool(h, (size_t)h->poolused + (size_t)h->cross_ub * 64))) return rc;
    }
    bool classified = h->pre_f == f && h->pre_nv == nv0;
    int cslot = classified ? h->pre_slot : next_counter_slot(h);
    int cut_id = classified ? h->pre_seq : (int)h->cutseq;       // unique per classification: stamp of the ZMarks rows
    h->pre_f = -1;
    if (cslot < 0) { set_error("hipMemsetAsync failed"); return BSLV_E_NODEVICE; }
    const int nbv = (vm_count(h->P, nv0) + PB - 1) / PB;      // hot mode: only the elements the chunk can touch
    const int rank = (int)h->facet_of_rank.size();
    int nbe, seqB = 0, slotB = 0, spec_ns = -1;
    Tri *ebsum, *vbsum;
    for (;;) {
        int *counters = h->counters + CSTRIDE * cslot;
        const int ne_ub = h->ne;                   // exact unless a prune is in flight
        const int *ne_dev = h->pend_k2 ? h->ne_dev : nullptr;
        nbe = std::max(1, (ne_ub + PB - 1) / PB);
        if ((rc = ensure_bsum(h, nbe + nbv + 2))) return rc;
        ebsum = h->bsum; vbsum = h->bsum + nbe + 1;
        // room for the survivors + crossing edges (<= ne) and every pair of a small facet (k2_fused emits in place)
        if ((rc = ensure_ecap(h, ne_ub + K2_MAXNM * (K2_MAXNM - 1) / 2 + 1))) return rc;
        if (h->ecap > h->ecountcap) { if ((rc = grow(&h->ecount, 0, (size_t)h->ecap, s))) return rc; h->ecountcap = h->ecap; }
        // ---- round A ----
        const ZMarks Z{h->zlist + ZMAX * cslot, h->zrows, h->fcap, cut_id};
        CutDev *cd = h->cutdev + cslot;
        auto tl0 = std::chrono::steady_clock::now();
        if (!classified) hipLaunchKernelGGL(k_classify, dim3(nbv), dim3(PB), 0, s, h->P, hp, nv0, counters, h->zlist + ZMAX * cslot, (const int *)nullptr);
        if (h->snap && !classified) {
            double nn = 0.0;
            for (int k = 0; k < d; k++) nn += hp.h[k] * hp.h[k];
            hipLaunchKernelGGL(k_snap, dim3(nbv), dim3(PB), 0, s, h->P, hp, nn, nv0, (const int *)counters, h->snapped_d);
        }
        const int seqA = ++h->mailseq;
        const ScanArgs SA{ebsum, nbe, vbsum, nbv, h->totals + 0, h->mail_d + 0, counters, ne_ub, ne_dev, seqA, cd, h->abort_d, nv0, h->P.cap, h->poolused, h->poolcap, h->cross_ub,
                          h->pend_k2 ? (const Mail *)(h->k2mail_d + (h->pend_slot - 2)) : (const Mail *)nullptr};
        // (letting the last workgroup of k_flags2 do the scans -- ticket + fences -- was measured SLOWER than this
        // second launch: an agent-scope fence per workgroup writes the L2 back)
        hipLaunchKernelGGL(k_flags2, dim3(nbe + nbv), dim3(PB), 0, s, h->P, h->E[h->ecur], ne_ub, ne_dev, nbe, nv0, counters, h->eflag, h->ecount, ebsum, vbsum, Z);
        const bool own_scan = spec && nbe + nbv <= 1024;      // few workgroups: k_emit2 sums the block sums itself, no scan launch
        if (!own_scan) hipLaunchKernelGGL(k_scan2, dim3(2), dim3(1024), 0, s, SA);
        auto tl1 = std::chrono::steady_clock::now();
        h->tm_launch[0] += std::chrono::duration<double, std::micro>(tl1 - tl0).count();
        if (spec) {
            // ---- round B, queued on the device's own verdict ----
            launch_emit2(d, dim3(nbe + nbv), s, h->P, hp, rank, (const int2 *)h->E[h->ecur], ne_ub, ne_dev, nbe, (const unsigned char *)h->eflag, (const int *)h->ecount, (const Tri *)ebsum, (const Tri *)vbsum,
                         (const Tri *)(h->totals + 0), h->E[1 - h->ecur], nv0, h->poolused, 0u, h->members, Z, (const int *)counters, (const CutDev *)cd,
                         (const int *)h->EP[h->ecur], h->EP[1 - h->ecur], own_scan ? 1 : 0, SA);
            auto tl2 = std::chrono::steady_clock::now();
            h->tm_launch[1] += std::chrono::duration<double, std::micro>(tl2 - tl1).count();
            spec_ns = -1;
            int ncb = 0;                       // workgroups of the prune launch that classify the next halfspace
            if (next_f >= 0) {
                if ((spec_ns = next_counter_slot(h)) < 0) { set_error("hipMemsetAsync failed"); return BSLV_E_NODEVICE; }
                ncb = (vm_count(h->P, nv0) + h->cross_ub + K2T - 1) / K2T;
                h->pre_seq = (int)h->cutseq;
            }
            seqB = ++h->mailseq;
            slotB = 2 + (h->k2flip ^= 1);
            hipLaunchKernelGGL(k2_fused_t<false>, dim3(1 + ncb), dim3(K2T), h->k2_lds, s, h->P, h->members, 0, nv0, 0, h->fcount, h->flocal, (int)(h->k2_lds / 8), h->E[1 - h->ecur], 0,
                               h->ne_dev, h->totals + 2, h->k2mail_d + (slotB - 2), seqB, h->k2dbg, (const CutDev *)cd, h->abort_d, h->EP[1 - h->ecur],
                               hn, h->counters + CSTRIDE * std::max(spec_ns, 0), h->zlist + ZMAX * std::max(spec_ns, 0), K2V2{});
        }
        auto tl3 = std::chrono::steady_clock::now();
        HIP_TRY(hipGetLastError());
        if ((rc = wait_mail(h, 0, seqA))) return rc;
        auto tl4 = std::chrono::steady_clock::now();
        h->tm_launch[2] += std::chrono::duration<double, std::micro>(tl3 - tl1).count();
        h->tm_launch[3] += std::chrono::duration<double, std::micro>(tl4 - tl3).count();
        bool redo;
        if ((rc = settle_k2(h, &redo))) return rc;
        if (!redo && spec && h->mail_v[0].cnt[0] > 0) {
            // did the device decline round B for want of capacity?  Then its classification of the NEXT halfspace has
            // already overwritten the classes of this one: book the declined prune, grow, and run the cut again
            // without speculation
            const Tri t0 = h->mail_v[0].t;
            const int zub = h->mail_v[0].cnt[2];
            if (!(t0.b <= h->cross_ub && nv0 + t0.b <= h->P.cap && (unsigned long long)h->poolused + (unsigned)t0.c + (unsigned)zub <= h->poolcap)) {
                h->pend_k2 = true; h->pend_seq = seqB; h->pend_slot = slotB; h->pend_ebase = h->ne; h->pend_ncross = 0;
                if ((rc = settle_k2(h))) return rc;
                if ((rc = ensure_vcap(h, nv0 + t0.b))) return rc;
                if ((rc = ensure_pool(h, (size_t)h->poolused + t0.c + zub))) return rc;
                spec = false;
                redo = true;
                h->n_declined++;
            }
        }
        if (!redo) break;
        // the previous prune was redone by the multi-kernel path (everything queued behind it skipped itself or was
        // free of side effects), or this cut runs again without speculation: classify and flag again
        classified = false;
        if ((cslot = next_counter_slot(h)) < 0) { set_error("hipMemsetAsync failed"); return BSLV_E_NODEVICE; }
        cut_id = (int)h->cutseq;
    }
    const ZMarks Z{h->zlist + ZMAX * cslot, h->zrows, h->fcap, cut_id};
    const int *counters = h->counters + CSTRIDE * cslot;
    const int nminus = h->mail_v[0].cnt[0], nzero = h->mail_v[0].cnt[1], zero_ub = h->mail_v[0].cnt[2];
    const Tri te = h->mail_v[0].t;
    const int ne0 = h->ne;
    if (h->mail_v[0].cnt[3] != ne0) { set_error("internal: edge count on the device %d, on the host %d", h->mail_v[0].cnt[3], ne0); return BSLV_E_STATE; }
    if (h->cutlog) fprintf(h->cutlog, "%d %d %d %d %d %d %d %d\n", nv0, ne0, nminus, nzero, zero_ub, te.a, te.b, te.c);
    const int nsurv = te.a, ncross = te.b;
    const int nm = nzero + ncross;
    const long long len_ub = (long long)te.c + zero_ub;
    h->n_single++;
    if (spec) h->n_spec++;
    const bool went = spec && nminus > 0;         // (capacities were checked in the loop, with the verdict of k_scan2)
    if (spec) {   // the queued prune reports in any case (nothing to do / pairs / fallback): book it at the next wait
        h->pend_k2 = true; h->pend_seq = seqB; h->pend_slot = slotB;
        h->pend_ebase = went ? nsurv + ncross : ne0;
        h->pend_nm = nm; h->pend_len_ub = len_ub; h->pend_nzero = nzero; h->pend_nv0 = nv0; h->pend_ncross = ncross;
        if (spec_ns >= 0 && (went || nminus == 0)) { h->pre_f = next_f; h->pre_slot = spec_ns; h->pre_nv = went ? nv0 + ncross : nv0; }
    }
    if (nminus == 0) { h->fapplied[f] = 0; *rc_out = 1; return 0; }
    h->facet_of_rank.push_back(f);
    const unsigned pool_e = h->poolused, pool_z = h->poolused + (unsigned)te.c;
    if (!went) {
        // ---- round B with host arguments (no speculation, or a capacity was short and the device declined) ----
        if ((rc = ensure_vcap(h, nv0 + ncross))) return rc;
        if ((rc = ensure_pool(h, (size_t)h->poolused + te.c + zero_ub))) return rc;
        launch_emit2(d, dim3(nbe + nbv), s, h->P, hp, rank, (const int2 *)h->E[h->ecur], ne0, (const int *)nullptr, nbe, (const unsigned char *)h->eflag, (const int *)h->ecount, (const Tri *)ebsum, (const Tri *)vbsum,
                     (const Tri *)(h->totals + 0), h->E[1 - h->ecur], nv0, pool_e, pool_z, h->members, Z, counters, (const CutDev *)nullptr,
                     (const int *)h->EP[h->ecur], h->EP[1 - h->ecur], 0, ScanArgs{});
    }
    h->poolused += (unsigned)te.c + (unsigned)zero_ub;
    h->nv = nv0 + ncross;
    h->ne = nsurv + ncross;
    h->ecur = 1 - h->ecur;
    h->new_vertices += ncross;
    h->pend_stamp = (int)(4 * h->cutseq + 3);     // fstamp value of a fallback prune for this cut (monotone)
    if (nm >= 2) h->pair_tests += (long)nm * (nm - 1) / 2;
    if (went) {
        if (nm >= 2 && nm <= K2_MAXNM) h->ne += (int)((long long)nm * (nm - 1) / 2);      // upper bound until settle_k2
    } else {
        if (next_f >= 0) {
            // classify the next halfspace now (the new vertices exist once k_emit2 has run; k2 does not read classes)
            const int ns = next_counter_slot(h);
            if (ns < 0) { set_error("hipMemsetAsync failed"); return BSLV_E_NODEVICE; }
            hipLaunchKernelGGL(k_classify, dim3((vm_count(h->P, h->nv) + PB - 1) / PB), dim3(PB), 0, s, h->P, hn, h->nv, h->counters + CSTRIDE * ns, h->zlist + ZMAX * ns, (const int *)nullptr);
            h->pre_f = next_f; h->pre_slot = ns; h->pre_nv = h->nv; h->pre_seq = (int)h->cutseq;
        }
        if (nm >= 2) {
            if (nm <= K2_MAXNM) {
                const int sq = ++h->mailseq, sl = 2 + (h->k2flip ^= 1);
                hipLaunchKernelGGL(k2_fused_t<false>, dim3(1), dim3(K2T), h->k2_lds, s, h->P, h->members, nzero, nv0, ncross, h->fcount, h->flocal,
                                   (int)(h->k2_lds / 8), h->E[h->ecur], h->ne, h->ne_dev, h->totals + 2, h->k2mail_d + (sl - 2), sq, h->k2dbg, (const CutDev *)nullptr, h->abort_d, h->EP[h->ecur], hn, (int *)nullptr, (int *)nullptr, K2V2{});
                HIP_TRY(hipGetLastError());
                h->pend_k2 = true; h->pend_seq = sq; h->pend_slot = sl; h->pend_ebase = h->ne;
                h->pend_nm = nm; h->pend_len_ub = len_ub; h->pend_nzero = nzero; h->pend_nv0 = nv0; h->pend_ncross = ncross;
                h->ne += (int)((long long)nm * (nm - 1) / 2);          // upper bound until settle_k2
            } else {
                if (ncross > 0) hipLaunchKernelGGL(k_iota_members, dim3((ncross + 255) / 256), dim3(256), 0, s, h->members, nzero, nv0, ncross);
                if ((rc = k2_multi(h, nm, len_ub, h->pend_stamp))) return rc;
            }
        }
    }
    HIP_TRY(hipGetLastError());
    h->cuts_applied++;
    *rc_out = 0;
    return 0;
}

// ---- hot mode ----
// Most of a large polyhedron lies strictly inside every halfspace of a chunk of cuts (the batched classification
// says which elements do not: tc > 0).  hot_begin() moves the elements and edges a cut of the chunk can touch into
// short lists; the per-cut passes then run over those and over what the chunk itself creates, instead of over the
// whole polyhedron, and hot_end() merges the edge lists again.  Orders are preserved throughout: slot numbers and
// the edge order stay exactly those of the one-list pipeline (and of oracle/poly_dd.c).
static int hot_begin(bslv_poly *h, const int *tc)
{
    if (h->hot || h->pend_k2) { set_error("internal: hot_begin in the wrong state"); return BSLV_E_STATE; }
    hipStream_t s = h->stream;
    const int nv = h->nv, ne = h->ne, nbv = (nv + PB - 1) / PB, nbe = std::max(1, (ne + PB - 1) / PB);
    int rc;
    if ((rc = ensure_bsum(h, std::max(nbv, nbe) + 1))) return rc;
    Tri t;
    hipLaunchKernelGGL(k_hotv_flags, dim3(nbv), dim3(PB), 0, s, h->P, tc, nv, h->bsum);
    if ((rc = scan_totals(h, nbv, &t))) return rc;
    const int nhv = t.a;
    if (nhv > h->hvcap) { int nc = std::max(nhv, std::max(4096, h->hvcap * 2)); if ((rc = grow(&h->hv_d, 0, (size_t)nc, s))) return rc; h->hvcap = nc; }
    hipLaunchKernelGGL(k_hotv_emit, dim3(nbv), dim3(PB), 0, s, h->P, tc, nv, h->bsum, h->hv_d);
    hipLaunchKernelGGL(k_hote_flags, dim3(nbe), dim3(PB), 0, s, (const int2 *)h->E[h->ecur], ne, tc, h->bsum);
    if ((rc = scan_totals(h, nbe, &t))) return rc;
    const int neh = t.a;
    bslv_poly::EdgeSet &H = h->hotbuf;
    const int need = neh + K2_MAXNM * (K2_MAXNM - 1) / 2 + 1;
    if (need > H.ecap) {
        const int nc = std::max(need, std::max(1 << 16, H.ecap * 2));
        for (int k = 0; k < 2; k++) { if ((rc = grow(&H.E[k], 0, (size_t)nc, s))) return rc; if ((rc = grow(&H.EP[k], 0, (size_t)nc, s))) return rc; }
        if ((rc = grow(&H.eflag, 0, (size_t)nc, s))) return rc;
        H.ecap = nc;
    }
    if (ne > h->alivecap) { int nc = std::max(ne, std::max(1 << 16, h->alivecap * 2)); if ((rc = grow(&h->alive, 0, (size_t)nc, s))) return rc; h->alivecap = nc; }
    hipLaunchKernelGGL(k_hote_emit, dim3(nbe), dim3(PB), 0, s, (const int2 *)h->E[h->ecur], ne, tc, (const Tri *)h->bsum, H.E[0], H.EP[0], h->alive);
    HIP_TRY(hipGetLastError());
    // the full list steps aside
    bslv_poly::EdgeSet &F = h->full;
    F.E[0] = h->E[0]; F.E[1] = h->E[1]; F.eflag = h->eflag; F.ecap = h->ecap; F.ne = ne; F.ecur = h->ecur;
    h->E[0] = H.E[0]; h->E[1] = H.E[1]; h->EP[0] = H.EP[0]; h->EP[1] = H.EP[1]; h->eflag = H.eflag; h->ecap = H.ecap; h->ne = neh; h->ecur = 0;
    h->P.hv = h->hv_d; h->P.nhv = nhv; h->P.nv_base = nv;
    {   // membership bitmaps of the hot elements with long lists; ranks up to those this chunk can add
        constexpr int LMAX = 64;
        const int stride = ((int)h->facet_of_rank.size() + h->chunk_cuts + 1024 + 63) / 32;
        const size_t need = (size_t)LMAX * stride;
        if (need > h->lbitscap) { const size_t nc = std::max(need, h->lbitscap * 2); if ((rc = grow(&h->lbits_d, 0, nc, s))) return rc; h->lbitscap = nc; }
        if (!h->lnslots_d && (rc = grow(&h->lnslots_d, 0, 4, s))) return rc;
        HIP_TRY(hipMemsetAsync(h->lbits_d, 0, need * sizeof(unsigned), s));
        HIP_TRY(hipMemsetAsync(h->lnslots_d, 0, sizeof(int), s));
        h->P.lslot = h->lslot_d; h->P.lbits = h->lbits_d; h->P.lstride = stride;
        if (nhv > 0) hipLaunchKernelGGL(k_lbits_build, dim3((nhv + PB / WAVE - 1) / (PB / WAVE)), dim3(PB), 0, s, h->P, nhv, LMAX, h->lnslots_d);
        HIP_TRY(hipGetLastError());
    }
    h->hot = true;
    h->pre_f = -1;
    h->hot_chunks++; h->hot_elems += nhv; h->hot_edges += neh;
    if (getenv("BSLV_HOT_DEBUG")) fprintf(stderr, "hot chunk: %d of %d elements, %d of %d edges\n", nhv, nv, neh, ne);
    return 0;
}
static int hot_end(bslv_poly *h)
{
    if (!h->hot) return 0;
    int rc;
    if ((rc = settle_k2(h))) return rc;
    hipStream_t s = h->stream;
    // the hot buffers may have grown: keep them for the next chunk
    bslv_poly::EdgeSet &H = h->hotbuf, &F = h->full;
    H.E[0] = h->E[0]; H.E[1] = h->E[1]; H.EP[0] = h->EP[0]; H.EP[1] = h->EP[1]; H.eflag = h->eflag; H.ecap = h->ecap;
    const int neh = h->ne;
    const int2 *EH = h->E[h->ecur];
    const int *EPc = h->EP[h->ecur];
    h->E[0] = F.E[0]; h->E[1] = F.E[1]; h->EP[0] = h->EP[1] = nullptr; h->eflag = F.eflag; h->ecap = F.ecap; h->ne = F.ne; h->ecur = F.ecur;
    if (h->P.nhv > 0) hipLaunchKernelGGL(k_lbits_reset, dim3((h->P.nhv + 255) / 256), dim3(256), 0, s, h->P, h->P.nhv);
    h->P.lslot = nullptr; h->P.lbits = nullptr; h->P.lstride = 0;
    h->P.hv = nullptr; h->P.nhv = 0; h->P.nv_base = 0;
    h->hot = false;
    h->pre_f = -1;
    const int ne = h->ne, nbe = std::max(1, (ne + PB - 1) / PB), nbh = std::max(1, (neh + PB - 1) / PB);
    if ((rc = ensure_bsum(h, std::max(nbe, nbh) + 1))) return rc;
    Tri t;
    hipLaunchKernelGGL(k_hot_revive, dim3(nbh), dim3(PB), 0, s, EPc, neh, h->alive, h->bsum);
    if ((rc = scan_totals(h, nbh, &t))) return rc;
    const int nold = t.a;                          // hot edges of the old list that are still there: a prefix of EH
    hipLaunchKernelGGL(k_alive_flags, dim3(nbe), dim3(PB), 0, s, (const unsigned char *)h->alive, ne, h->bsum);
    if ((rc = scan_totals(h, nbe, &t))) return rc;
    const int nalive = t.a, nnew = neh - nold;
    if ((rc = ensure_ecap(h, nalive + nnew + 1))) return rc;
    hipLaunchKernelGGL(k_alive_emit, dim3(nbe), dim3(PB), 0, s, (const int2 *)h->E[h->ecur], (const unsigned char *)h->alive, ne, (const Tri *)h->bsum, h->E[1 - h->ecur]);
    HIP_TRY(hipGetLastError());
    if (nnew > 0) HIP_TRY(hipMemcpyAsync(h->E[1 - h->ecur] + nalive, EH + nold, (size_t)nnew * sizeof(int2), hipMemcpyDeviceToDevice, s));
    h->ecur = 1 - h->ecur;
    h->ne = nalive + nnew;
    return 0;
}

#include "poly_rounds_host.inc"
#include "poly_rounds2_host.inc"

static int upload_initial(bslv_poly *h, const std::vector<double> &X /* (d+1) x d */, const std::vector<std::vector<int>> &inc)
{
    const int d = h->d, n = d + 1;
    int rc;
    if ((rc = ensure_vcap(h, 1024))) return rc;
    size_t tot = 0;
    for (auto &l : inc) tot += l.size();
    if ((rc = ensure_pool(h, tot))) return rc;
    if ((rc = ensure_ecap(h, 4096))) return rc;
    std::vector<double> soa((size_t)d * n);
    for (int i = 0; i < n; i++) for (int k = 0; k < d; k++) soa[(size_t)k * n + i] = X[(size_t)i * d + k];
    for (int k = 0; k < d; k++)
        HIP_TRY(hipMemcpy(h->P.X + (size_t)k * h->P.cap, &soa[(size_t)k * n], n * sizeof(double), hipMemcpyHostToDevice));
    std::vector<unsigned char> fl(n, F_USED);
    for (int i = 1; i < n; i++) fl[i] |= F_IDEAL;
    HIP_TRY(hipMemcpy(h->P.flag, fl.data(), n, hipMemcpyHostToDevice));
    std::vector<unsigned> off(n);
    std::vector<int> len(n), pool;
    for (int i = 0; i < n; i++) { off[i] = (unsigned)pool.size(); len[i] = (int)inc[i].size(); pool.insert(pool.end(), inc[i].begin(), inc[i].end()); }
    HIP_TRY(hipMemcpy(h->P.inc_off, off.data(), n * sizeof(unsigned), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->P.inc_len, len.data(), n * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->P.pool, pool.data(), pool.size() * sizeof(int), hipMemcpyHostToDevice));
    h->poolused = (unsigned)pool.size();
    std::vector<int2> E;
    for (int k = 0; k <= d; k++) for (int j = k + 1; j <= d; j++) E.push_back(int2{k, j});
    HIP_TRY(hipMemcpy(h->E[h->ecur], E.data(), E.size() * sizeof(int2), hipMemcpyHostToDevice));
    h->ne = (int)E.size();
    h->nv = n;
    h->facet_of_rank.resize(h->nf);
    for (int f = 0; f < h->nf; f++) h->facet_of_rank[f] = f;
    return 0;
}

// modified Gram-Schmidt step (bslv__normalise, bslv_poly.c:1030-1060)
static double gs_step(const double *x, double *H, double *R, int k, int n)
{
    double nrm_in = 0, scl = 0;
    for (int l = 0; l < n; l++) nrm_in += x[l] * x[l];
    nrm_in = std::sqrt(nrm_in);
    double *hr = H + (size_t)k * n;
    for (int l = 0; l < n; l++) hr[l] = x[l];
    for (int j = 0; j < k; j++) {
        double s = 0;
        for (int l = 0; l < n; l++) s += H[(size_t)j * n + l] * hr[l];
        for (int l = 0; l < n; l++) hr[l] -= s * H[(size_t)j * n + l];
    }
    for (int l = 0; l < n; l++) scl += hr[l] * hr[l];
    scl = std::sqrt(scl);
    if (scl < 1.0e-6) return 0;
    for (int l = 0; l < n; l++) hr[l] /= scl;
    for (int j = 0; j <= k; j++) {
        double s = 0;
        for (int l = 0; l < n; l++) s += H[(size_t)j * n + l] * x[l];
        R[k * (k + 1) / 2 + j] = s;
    }
    return scl / nrm_in;
}

extern "C" {

int bslv_poly_create(bslv_poly **out, int dim, int v2h, const double *c)
{
    if (!out || dim < 2 || dim > MAXD || v2h < 0 || v2h > 2) { set_error("bslv_poly_create: bad argument (2 <= dim <= %d)", MAXD); return BSLV_E_ARG; }
    if (bslv_device_count() < 1) { set_error("no HIP device available"); return BSLV_E_NODEVICE; }
    bslv_poly *h = new bslv_poly();
    h->d = dim; h->v2h = v2h;
    h->c.assign(dim, 0.0);
    if (c) h->c.assign(c, c + dim);
    h->P.d = dim;
    auto fail = [&](int code) { bslv_poly_destroy(h); return code; };
    if (hipStreamCreate(&h->stream) != hipSuccess) { set_error("hipStreamCreate failed"); return fail(BSLV_E_NODEVICE); }
    if (malloc0(&h->totals, 4 * sizeof(Tri)) != hipSuccess || malloc0(&h->counters, CRING * CSTRIDE * sizeof(int)) != hipSuccess || malloc0(&h->ne_dev, 4 * sizeof(int)) != hipSuccess || malloc0(&h->k2mail_d, 4 * sizeof(Mail)) != hipSuccess || hipMemset(h->k2mail_d, 0, 4 * sizeof(Mail)) != hipSuccess || malloc0(&h->cutdev, CRING * sizeof(CutDev)) != hipSuccess ||
        malloc0(&h->abort_d, 4 * sizeof(int)) != hipSuccess || hipMemset(h->abort_d, 0, 4 * sizeof(int)) != hipSuccess || malloc0(&h->zlist, CRING * ZMAX * sizeof(int)) != hipSuccess ||
        hipHostMalloc(&h->totals_h, 4 * sizeof(Tri)) != hipSuccess || hipHostMalloc(&h->counters_h, 4 * sizeof(int)) != hipSuccess ||
        hipHostMalloc(&h->mail_h, 4 * sizeof(Mail), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void **)&h->mail_d, h->mail_h, 0) != hipSuccess ||
        hipMemset(h->counters, 0, CRING * CSTRIDE * sizeof(int)) != hipSuccess) {
        set_error("allocation of scan scratch failed");
        return fail(BSLV_E_NOMEM);
    }
    memset((void *)h->mail_h, 0, 4 * sizeof(Mail));
    memset((void *)h->mail_v, 0, sizeof h->mail_v);
    for (int k = 0; k < 4; k++) h->mail_h[k].chk2 = mail_sum2(Tri{0, 0, 0}, 0);      // (the prune triple of a mailbox nobody has written yet is consistent)
    // k2_fused keeps the local incidence bit matrix in LDS: ask for most of the CU's 160 KB, settle for 48 KB
    h->k2_lds = 128 * 1024;
    if (hipFuncSetAttribute((const void *)k2_fused_t<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->k2_lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)k2_fused_t<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->k2_lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_r2_k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->k2_lds) != hipSuccess) {
        (void)hipGetLastError();
        h->k2_lds = 48 * 1024;
    }
    if (const char *cl = getenv("BSLV_CUT_LOG")) h->cutlog = fopen(cl, "w");
    if (getenv("BSLV_NO_SPEC")) h->speculate = false;
    if (getenv("BSLV_NO_HOT")) h->hot_enabled = false;
    if (getenv("BSLV_NO_ROUNDS2")) h->rounds2_enabled = false;
    if (const char *e = getenv("BSLV_R2_MIS")) h->r2_mis = atoi(e) != 0;
    if (const char *e = getenv("BSLV_R2_SHARE")) h->r2_share = atoi(e) != 0;
    if (const char *e = getenv("BSLV_R2_SPEC")) h->r2_spec = atoi(e) != 0;
    if (const char *e = getenv("BSLV_R2_FUSE")) h->r2_fuse = std::min(2, std::max(0, atoi(e)));
    if (const char *e = getenv("BSLV_R2_FORK")) h->r2_fork = atoi(e) != 0;
    if (const char *e = getenv("BSLV_CHUNK_CUTS")) h->chunk_cuts = std::min(4096, std::max(32, atoi(e)));
    if (const char *e = getenv("BSLV_R2_MIN_CUTS")) h->r2_min_cuts = std::max(-1, atoi(e));
    if (const char *e = getenv("BSLV_R2_RULE")) h->r2_rule = atoi(e) ? 1 : 0;
    if (const char *e = getenv("BSLV_POLY_SNAP")) h->snap = atoi(e) != 0;
    if (const char *e = getenv("BSLV_CROSS_UB")) h->cross_ub = std::max(0, atoi(e));
    if (const char *e = getenv("BSLV_K2_LDS")) h->k2_lds = (size_t)std::max(64, atoi(e));      // test hook: a small value forces the multi-kernel prune
    if (getenv("BSLV_K2_DEBUG") && malloc0(&h->k2dbg, 16 * sizeof(unsigned long long)) == hipSuccess) (void)hipMemset(h->k2dbg, 0, 16 * sizeof(unsigned long long));
    h->rounds = new RoundsBuf();
    // dual slot 0: "facet at infinity", ideal point (0,..,0,-1)  (bslv_poly.c:83-92)
    std::vector<double> z(dim, 0.0);
    z[dim - 1] = -1.0;
    new_dual(h, z.data(), 1);
    *out = h;
    return 0;
}

void bslv_poly_destroy(bslv_poly *h)
{
    if (!h) return;
    if (getenv("BSLV_R2_REPORT") && h->r2_fallback_prunes) {
        fprintf(stderr, "r2: %ld prunes of rounds went through the multi-kernel path: %ld members > %d, %ld pair bitmap, %ld hash table, %ld bit matrix | members by power of two:", h->r2_fallback_prunes,
                h->r2_fb_reason[1], K2_MAXNM, h->r2_fb_reason[2], h->r2_fb_reason[3], h->r2_fb_reason[4]);
        for (int k = 0; k < 20; k++) if (h->r2_fb_nm[k]) fprintf(stderr, " 2^%d: %ld", k, h->r2_fb_nm[k]);
        fprintf(stderr, "\n");
    }
    if (getenv("BSLV_TIMING"))
        fprintf(stderr, "poly timing: add_cuts %.1f ms | hot_begin %.1f, sequences %.1f (%ld cuts, %.1f us each), hot_end %.1f ms | %ld hot chunks, %.0f elements, %.0f edges on average\n", h->tm_add_cuts,
                h->tm_hot_begin, h->tm_seq, h->tm_seq_cuts, h->tm_seq_cuts ? h->tm_seq * 1e3 / h->tm_seq_cuts : 0.0, h->tm_hot_end, h->hot_chunks,
                h->hot_chunks ? (double)h->hot_elems / h->hot_chunks : 0.0, h->hot_chunks ? (double)h->hot_edges / h->hot_chunks : 0.0),
        fprintf(stderr, "poly timing, per batch of cuts: dual slots %.2f ms, chunk preparation %.2f ms, waiting for the batched classification %.2f ms (totals); %ld re-allocations of element / edge / pool arrays, %.2f ms\n", h->tm_newdual, h->tm_prep, h->tm_classify_wait, h->n_realloc, h->tm_realloc),
        fprintf(stderr, "poly host per cut (us): queue round A %.1f, k_emit2 %.1f, round B in all %.1f, mailbox wait %.1f\n", h->tm_launch[0] / std::max(1L, h->n_single), h->tm_launch[1] / std::max(1L, h->n_single),
                h->tm_launch[2] / std::max(1L, h->n_single), h->tm_launch[3] / std::max(1L, h->n_single));
    if (h->cutlog) fclose(h->cutlog);
    if (h->k2dbg) {
        unsigned long long t[16];
        if (hipMemcpy(t, h->k2dbg, sizeof(t), hipMemcpyDeviceToHost) == hipSuccess && t[7])
            fprintf(stderr, "k2_fused phases (us/launch over %llu launches; rounds: per prune workgroup): P0a %.2f P0 %.2f P1 %.2f P2 %.2f P3 %.2f P4 %.2f P5 %.2f P6 %.2f | W %.1f nloc %.0f | members mean %.0f max %llu\n", t[7],
                    t[10] * 0.01 / t[7], t[0] * 0.01 / t[7], t[1] * 0.01 / t[7], t[2] * 0.01 / t[7], t[3] * 0.01 / t[7], t[4] * 0.01 / t[7], t[5] * 0.01 / t[7], t[6] * 0.01 / t[7], (double)t[8] / t[7], (double)t[9] / t[7],
                    (double)t[11] / t[7], t[12]);
        (void)hipFree(h->k2dbg);
    }
    auto fr = [](void *p) { if (p) (void)hipFree(p); };
    fr(h->P.X); fr(h->P.flag); fr(h->P.cls); fr(h->P.inc_off); fr(h->P.inc_len); fr(h->P.capx); fr(h->P.pool); fr(h->P.keep);
    fr(h->E[0]); fr(h->E[1]); fr(h->eflag); fr(h->members); fr(h->bsum); fr(h->bsum2); fr(h->fm_cnt); fr(h->fm_list); fr(h->nzlist); fr(h->totals); fr(h->counters); fr(h->ne_dev); fr(h->k2mail_d); fr(h->zlist); fr(h->zrows); fr(h->fcount); fr(h->cutdev); fr(h->abort_d);
    for (int k = 0; k < 2; k++) { fr(h->hotbuf.E[k]); fr(h->hotbuf.EP[k]); }
    fr(h->snapped_d);
    fr(h->hotbuf.eflag); fr(h->alive); fr(h->hv_d); fr(h->ecount); fr(h->lslot_d); fr(h->lbits_d); fr(h->lnslots_d);
    fr(h->shard_e); fr(h->blks); fr(h->pflag); fr(h->fstamp); fr(h->flocal); fr(h->nlocal); fr(h->bits); fr(h->hps_d); fr(h->clsw); fr(h->anyminus); fr(h->idx_d); fr(h->val_d); fr(h->fl_d); fr(h->par_d); fr(h->r2f_d); fr(h->fhist_d); fr(h->chosen_d);
    if (h->rounds) { rounds_free(*h->rounds); delete h->rounds; }
    if (h->rounds2) { rounds2_free(*h->rounds2); delete h->rounds2; }
    if (h->totals_h) (void)hipHostFree(h->totals_h);
    if (h->counters_h) (void)hipHostFree(h->counters_h);
    if (h->mail_h) (void)hipHostFree((void *)h->mail_h);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int bslv_poly_dual0_apex(bslv_poly *h)
{
    if (!h) return BSLV_E_ARG;
    h->fideal[0] = 0;
    h->Y[h->d - 1] = 0.0;
    v2h_map(h, h->Y.data(), 0, h->hp.data());
    return 0;
}

int bslv_poly_add(bslv_poly *h, const double *val, int ideal, int *rc_out)
{
    if (!h || !val || !rc_out) { set_error("bslv_poly_add: bad argument"); return BSLV_E_ARG; }
    int f = new_dual(h, val, ideal);
    if (!h->initialised) { h->queue.push_back(f); *rc_out = 0; return 0; }
    int rc = do_cut(h, f, rc_out);
    return rc ? rc : settle_k2(h);
}

int bslv_poly_init(bslv_poly *h, int *rc_out)
{
    if (!h || !rc_out) return BSLV_E_ARG;
    if (h->initialised) { set_error("bslv_poly_init: already initialised"); return BSLV_E_STATE; }
    const int d = h->d;
    int qn = (int)h->queue.size();
    if (qn < d) { *rc_out = 1; return 0; }
    std::vector<double> hpq((size_t)qn * (d + 1));
    for (int k = 0; k < qn; k++) memcpy(&hpq[(size_t)k * (d + 1)], &h->hp[(size_t)h->queue[k] * (d + 1)], (d + 1) * sizeof(double));
    std::vector<double> H((size_t)d * d, 0.0), R((size_t)d * (d + 1) / 2, 0.0), alph(d, 0.0);
    std::vector<int> perm(d + 1, 0);
    int g = 0;
    // greedy choice of d independent halfspaces (poly__intl_apprx, bslv_poly.c:167-185)
    while (g < d) {
        double best = 0; int bi = -1;
        for (int k = 0; k < qn; k++) {
            double s = gs_step(&hpq[(size_t)k * (d + 1)], H.data(), R.data(), g, d);
            if (best < s) { best = s; bi = k; }
        }
        if (best < 1.0e-10) { *rc_out = 1; return 0; }
        gs_step(&hpq[(size_t)bi * (d + 1)], H.data(), R.data(), g, d);
        alph[g] = hpq[(size_t)bi * (d + 1) + d];
        perm[++g] = h->queue[bi];
        qn--;
        memcpy(&hpq[(size_t)bi * (d + 1)], &hpq[(size_t)qn * (d + 1)], (d + 1) * sizeof(double));
        h->queue[bi] = h->queue[qn];
    }
    // initial simplex cone (poly__poly_initialise, bslv_poly.c:711-787)
    auto RM = [&](int k, int j) -> double & { return R[k * (k + 1) / 2 + j]; };
    std::vector<double> X((size_t)(d + 1) * d, 0.0), t(d, 0.0);
    for (int k = 0; k < d; k++) {
        double s = alph[k];
        for (int j = 0; j < k; j++) s -= RM(k, j) * t[j];
        t[k] = s / RM(k, k);
    }
    for (int j = 0; j < d; j++) { double s = 0; for (int l = 0; l < d; l++) s += H[(size_t)l * d + j] * t[l]; X[j] = s; }
    for (int k = 0; k < d; k++) {
        std::fill(t.begin(), t.end(), 0.0);
        t[k] = 1.0 / RM(k, k);
        for (int i = k + 1; i < d; i++) {
            double s = 0;
            for (int j = k; j < i; j++) s += RM(i, j) * t[j];
            t[i] = -s / RM(i, i);
        }
        for (int j = 0; j < d; j++) { double s = 0; for (int l = 0; l < d; l++) s += H[(size_t)l * d + j] * t[l]; X[(size_t)(k + 1) * d + j] = s; }
    }
    std::vector<std::vector<int>> inc(d + 1);
    for (int j = 0; j <= d; j++) {
        for (int k = 0; k <= d; k++) if (k != j) inc[j].push_back(perm[k]);
        std::sort(inc[j].begin(), inc[j].end());
    }
    int rc;
    if ((rc = upload_initial(h, X, inc))) return rc;
    h->initialised = true;
    // halfspaces not chosen are re-added as NEW dual slots; originals stay unused (bslv_poly.c:190-197)
    std::vector<int> rest(h->queue.begin(), h->queue.begin() + qn);
    h->queue.clear();
    for (int f : rest) h->fapplied[f] = 0;
    for (int f : rest) {
        std::vector<double> val(h->Y.begin() + (size_t)f * d, h->Y.begin() + (size_t)(f + 1) * d);
        int r2;
        if ((rc = bslv_poly_add(h, val.data(), h->fideal[f], &r2))) return rc;
    }
    *rc_out = 0;
    return 0;
}

// Batched poly__add_vrtx: B dual vertices in order.  A batched incidence pass (k_classify_batch)
// first finds the halfspaces no live element violates -- by convexity they stay redundant
// whatever the earlier cuts of the batch do -- and only the others run the per-cut pipeline.
int bslv_poly_add_cuts(bslv_poly *h, int B, const double *val, const int *ideal, int *rc_out)
{
    if (!h || B < 0 || (B > 0 && (!val || !rc_out))) { set_error("bslv_poly_add_cuts: bad argument"); return BSLV_E_ARG; }
    if (!h->initialised) {
        for (int b = 0; b < B; b++) { int r; int rc = bslv_poly_add(h, val + (size_t)b * h->d, ideal ? ideal[b] : 0, &r); if (rc) return rc; rc_out[b] = r; }
        return 0;
    }
    const int d = h->d;
    std::vector<int> fids(B);
    const auto tnd = std::chrono::steady_clock::now();
    for (int b = 0; b < B; b++) fids[b] = new_dual(h, val + (size_t)b * d, ideal ? ideal[b] : 0);
    h->tm_newdual += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tnd).count();
    if (h->snap) {                       // the reference's order exactly: a moved element meets the next cut where the last one left it
        h->cut_prio.clear();
        for (int b = 0; b < B; b++) { int r, rc = do_cut(h, fids[b], &r, -1); if (rc) return rc; rc_out[b] = r; }
        return settle_k2(h);
    }
    if (h->batch_mode == 1 && B >= 2) {
        auto t0 = std::chrono::steady_clock::now();
        if ((int)h->cut_prio.size() != B) h->cut_prio.clear();
        int rc = apply_cuts_rounds(h, fids, rc_out);
        h->cut_prio.clear();
        h->tm_add_cuts += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return rc;
    }
    std::vector<int> anym(B, 1);
    if (B >= 2) {
        int rc;
        const int nv = h->nv;
        const int chunkB = 512;        // halfspaces per launch: (d+1)*8*512 B of LDS
        for (int b0 = 0; b0 < B; b0 += chunkB) {
            int nb_ = std::min(chunkB, B - b0);
            if ((rc = upload_hps(h, &h->hp[(size_t)fids[b0] * (d + 1)], nb_))) return rc;
            if (nb_ > h->anycap) { if ((rc = grow(&h->anyminus, 0, (size_t)chunkB, h->stream))) return rc; h->anycap = chunkB; }
            size_t need = (size_t)((nb_ + 31) / 32) * h->P.cap;
            if (need > h->clswcap) { if ((rc = grow(&h->clsw, 0, need, h->stream))) return rc; h->clswcap = need; }
            HIP_TRY(hipMemsetAsync(h->anyminus, 0, ((nb_ + 31) / 32) * sizeof(unsigned), h->stream));
            launch_classify_batch(h->stream, h->P, h->hps_d, nb_, nv, h->clsw, h->anyminus, nullptr, nullptr);
            HIP_TRY(hipGetLastError());
            std::vector<unsigned> bits((nb_ + 31) / 32);
            HIP_TRY(hipMemcpy(bits.data(), h->anyminus, bits.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
            for (int k = 0; k < nb_; k++) anym[b0 + k] = (bits[k >> 5] >> (k & 31)) & 1u;
        }
    }
    std::vector<int> todo;
    for (int b = 0; b < B; b++) {
        if (!anym[b]) { h->fapplied[fids[b]] = 0; rc_out[b] = 1; } else todo.push_back(b);
    }
    for (size_t k = 0; k < todo.size(); k++) {
        int r, rc = do_cut(h, fids[todo[k]], &r, k + 1 < todo.size() ? fids[todo[k + 1]] : -1);
        if (rc) return rc;
        rc_out[todo[k]] = r;
    }
    return settle_k2(h);
}

// Stand-alone batched incidence kernel for tests and the roofline measurement: classes of the
// current elements against B arbitrary halfspaces (hps: B x (dim+1), normal then alpha).
// words_out (host, may be NULL): ceil(B/32) x nv 64-bit words; anyminus_out (host, may be NULL): B ints.
int bslv_poly_classify_batch(bslv_poly *h, int B, const double *hps, unsigned long long *words_out, int *anyminus_out, int repeats, float *ms_out)
{
    if (!h || B < 1 || !hps || !h->initialised) { set_error("bslv_poly_classify_batch: bad argument"); return BSLV_E_ARG; }
    const int nv = h->nv;
    if (B > 4096) { set_error("classify_batch: at most 4096 halfspaces per call"); return BSLV_E_ARG; }
    int rc;
    if ((rc = upload_hps(h, hps, B))) return rc;
    if (B > h->anycap) { if ((rc = grow(&h->anyminus, 0, (size_t)B, h->stream))) return rc; h->anycap = B; }
    size_t need = (size_t)((B + 31) / 32) * h->P.cap;
    if (need > h->clswcap) { if ((rc = grow(&h->clsw, 0, need, h->stream))) return rc; h->clswcap = need; }
    HIP_TRY(hipMemsetAsync(h->anyminus, 0, ((B + 31) / 32) * sizeof(unsigned), h->stream));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    if (repeats < 1) repeats = 1;
    // one untimed launch, then `repeats` timed ones
    for (int it = 0; it <= repeats; it++) {
        if (it == 1) HIP_TRY(hipEventRecord(e0, h->stream));
        launch_classify_batch(h->stream, h->P, h->hps_d, B, nv, h->clsw, h->anyminus, nullptr, nullptr);
    }
    HIP_TRY(hipEventRecord(e1, h->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / repeats;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (words_out)
        for (int w = 0; w < (B + 31) / 32; w++)
            HIP_TRY(hipMemcpy(words_out + (size_t)w * nv, h->clsw + (size_t)w * h->P.cap, (size_t)nv * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (anyminus_out) {
        std::vector<unsigned> bits((B + 31) / 32);
        HIP_TRY(hipMemcpy(bits.data(), h->anyminus, bits.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
        for (int k = 0; k < B; k++) anyminus_out[k] = (bits[k >> 5] >> (k & 31)) & 1u;
    }
    return 0;
}

// TEST: the same kernel in the form the chunked cut application launches it (with touch counts): tc_out[i] = halfspaces of
// the batch that element i is not strictly inside, t1_out[i] = the first of them (-1: none); nv ints each
int bslv_poly_classify_batch_touch(bslv_poly *h, int B, const double *hps, unsigned long long *words_out, int *tc_out, int *t1_out)
{
    if (!h || B < 1 || B > 4096 || !hps || !h->initialised || !tc_out || !t1_out) { set_error("bslv_poly_classify_batch_touch: bad argument"); return BSLV_E_ARG; }
    const int nv = h->nv;
    int rc;
    if ((rc = upload_hps(h, hps, B))) return rc;
    if (B > h->anycap) { if ((rc = grow(&h->anyminus, 0, (size_t)B, h->stream))) return rc; h->anycap = B; }
    size_t need = (size_t)((B + 31) / 32) * h->P.cap;
    if (need > h->clswcap) { if ((rc = grow(&h->clsw, 0, need, h->stream))) return rc; h->clswcap = need; }
    int *tt = nullptr;
    HIP_TRY(malloc0s(&tt, (size_t)2 * std::max(nv, 1) * sizeof(int), h->stream));
    HIP_TRY(hipMemsetAsync(h->anyminus, 0, ((B + 31) / 32) * sizeof(unsigned), h->stream));
    launch_classify_batch(h->stream, h->P, h->hps_d, B, nv, h->clsw, h->anyminus, tt, tt + nv);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(tc_out, tt, (size_t)nv * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(t1_out, tt + nv, (size_t)nv * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(tt);
    if (words_out)
        for (int w = 0; w < (B + 31) / 32; w++)
            HIP_TRY(hipMemcpy(words_out + (size_t)w * nv, h->clsw + (size_t)w * h->P.cap, (size_t)nv * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

// all unprocessed elements (used && !sltn) in ascending slot order: the set poly__get_vrtx
// iterates (bslv_poly.c:214-216).  *count = how many exist; at most max_out are written.
int bslv_poly_unprocessed2(bslv_poly *h, int max_out, int from_end, int *idx, double *val, int *ideal, int *parent, int *count);
static int sync_r2f(bslv_poly *h);
int bslv_poly_unprocessed(bslv_poly *h, int max_out, int *idx, double *val, int *ideal, int *count)
{
    return bslv_poly_unprocessed2(h, max_out, 0, idx, val, ideal, nullptr, count);
}
// from_end == 1: the max_out NEWEST unprocessed elements (highest slots); == 2: max_out elements spread
// evenly over all unprocessed ones (every total/max_out-th); == 3: the children of the newest cuts first (by the
// dual slot of the newest facet through them; fewer than max_out may come back while count says more exist --
// never zero); parent[k] = newest facet
// through element k (the cut that created it), -1 if none
int bslv_poly_unprocessed2(bslv_poly *h, int max_out, int from_end, int *idx, double *val, int *ideal, int *parent, int *count)
{
    if (!h || !count || max_out < 0) { set_error("bslv_poly_unprocessed: bad argument"); return BSLV_E_ARG; }
    *count = 0;
    if (!h->initialised || h->nv == 0) return 0;
    int rc;
    const int nv = h->nv, nb = (nv + PB - 1) / PB;
    if ((rc = ensure_bsum(h, nb + 1))) return rc;
    if (from_end == 3) {
        // children of the newest cuts first: at most max_out elements, those whose parent facet is newest (whole families of
        // siblings; of the oldest family taken, the newest members by slot).  *count = all unprocessed elements, as in the other modes.
        hipStream_t s = h->stream;
        const int nf = h->nf;
        if ((rc = sync_r2f(h))) return rc;
        if (nf + 4 > h->fhistcap) { const int nc = std::max(nf + 4 + 4096, h->fhistcap * 2); if ((rc = grow(&h->fhist_d, 0, (size_t)nc, s))) return rc; h->fhistcap = nc; }
        int *thr = h->fhist_d + h->fhistcap - 4;
        HIP_TRY(hipMemsetAsync(h->fhist_d, 0, (size_t)h->fhistcap * sizeof(int), s));
        hipLaunchKernelGGL(k_unproc_hist, dim3(nb), dim3(PB), 0, s, h->P, nv, (const int *)h->r2f_d, h->fhist_d);
        // (max_out == 0: only the count is wanted -- the threshold facet 0 covers everything)
        hipLaunchKernelGGL(k_unproc_threshold, dim3(1), dim3(1024), 0, s, (const int *)h->fhist_d, nf, max_out > 0 ? max_out : (1 << 30), thr);
        hipLaunchKernelGGL(k_unproc_flags_f, dim3(nb), dim3(PB), 0, s, h->P, nv, (const int *)h->r2f_d, (const int *)thr, h->bsum);
        Tri t;
        if ((rc = scan_totals(h, nb, &t))) return rc;
        *count = t.b;
        if (max_out == 0 || !idx) return 0;
        // everything from the threshold facet up, capped at 2 max_out + 4096 (newest slots); the host keeps the max_out best
        const int cap = 2 * max_out + 4096, n = std::min(t.a, cap);
        if (n > h->outcap) {
            int nc = std::max(n, h->outcap * 2);
            if ((rc = grow(&h->idx_d, 0, (size_t)nc, s))) return rc;
            if ((rc = grow(&h->val_d, 0, (size_t)nc * h->d, s))) return rc;
            if ((rc = grow(&h->fl_d, 0, (size_t)nc, s))) return rc;
            h->outcap = nc;
        }
        if (n > h->parcap) { if ((rc = grow(&h->par_d, 0, (size_t)std::max(n, h->parcap * 2), s))) return rc; h->parcap = std::max(n, h->parcap * 2); }
        hipLaunchKernelGGL(k_unproc_emit_f, dim3(nb), dim3(PB), 0, s, h->P, nv, (const int *)h->r2f_d, (const int *)thr, (const Tri *)h->bsum, t.a - n, n, h->idx_d, h->val_d, h->fl_d, h->par_d);
        HIP_TRY(hipGetLastError());
        std::vector<int> ti(n), tp(n);
        std::vector<double> tv((size_t)n * h->d);
        std::vector<unsigned char> fl(n);
        HIP_TRY(hipMemcpyAsync(ti.data(), h->idx_d, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(tp.data(), h->par_d, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(tv.data(), h->val_d, (size_t)n * h->d * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(fl.data(), h->fl_d, (size_t)n, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        std::vector<int> ord(n);
        for (int k = 0; k < n; k++) ord[k] = k;
        // newest parent facet first, within a family the newest slot first; ideal elements first of all (they cost no LP)
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) {
            const bool ia = fl[a] & F_IDEAL, ib = fl[b] & F_IDEAL;
            if (ia != ib) return ia;
            if (tp[a] != tp[b]) return tp[a] > tp[b];
            return ti[a] > ti[b];
        });
        const int m = std::min(n, max_out);
        ord.resize(m);
        std::sort(ord.begin(), ord.end(), [&](int a, int b) { return ti[a] < ti[b]; });      // handed on in ascending slot order, as in the other modes
        for (int k = 0; k < m; k++) {
            const int o = ord[k];
            idx[k] = ti[o];
            if (val) memcpy(val + (size_t)k * h->d, &tv[(size_t)o * h->d], h->d * sizeof(double));
            if (ideal) ideal[k] = (fl[o] & F_IDEAL) ? 1 : 0;
            if (parent) parent[k] = tp[o];
        }
        return 0;
    }
    hipLaunchKernelGGL(k_unproc_flags, dim3(nb), dim3(PB), 0, h->stream, h->P, nv, h->bsum);
    Tri t;
    if ((rc = scan_totals(h, nb, &t))) return rc;
    *count = t.a;
    int n = std::min(t.a, max_out);
    if (n == 0 || !idx) return 0;
    if (n > h->outcap) {
        int nc = std::max(n, h->outcap * 2);
        if ((rc = grow(&h->idx_d, 0, (size_t)nc, h->stream))) return rc;
        if ((rc = grow(&h->val_d, 0, (size_t)nc * h->d, h->stream))) return rc;
        if ((rc = grow(&h->fl_d, 0, (size_t)nc, h->stream))) return rc;
        h->outcap = nc;
    }
    if (n > h->parcap) { if ((rc = grow(&h->par_d, 0, (size_t)std::max(n, h->parcap * 2), h->stream))) return rc; h->parcap = std::max(n, h->parcap * 2); }
    hipLaunchKernelGGL(k_unproc_emit, dim3(nb), dim3(PB), 0, h->stream, h->P, nv, h->bsum, from_end == 1 ? t.a - n : 0, n, from_end == 2 ? t.a : 0,
                       h->idx_d, h->val_d, h->fl_d, h->par_d);
    HIP_TRY(hipGetLastError());
    std::vector<unsigned char> fl(n);
    if (parent) HIP_TRY(hipMemcpyAsync(parent, h->par_d, n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(idx, h->idx_d, n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (val) HIP_TRY(hipMemcpyAsync(val, h->val_d, (size_t)n * h->d * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(fl.data(), h->fl_d, n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (ideal) for (int k = 0; k < n; k++) ideal[k] = (fl[k] & F_IDEAL) ? 1 : 0;
    if (parent) for (int k = 0; k < n; k++) if (parent[k] >= 0) parent[k] = h->facet_of_rank[parent[k]];
    return 0;
}

static int sync_r2f(bslv_poly *h)
{
    hipStream_t s = h->stream;
    int rc;
    const int nr = (int)h->facet_of_rank.size();
    if (nr > h->r2fcap) { const int nc = std::max(nr + 4096, h->r2fcap * 2); if ((rc = grow(&h->r2f_d, (size_t)h->r2f_n, (size_t)nc, s))) return rc; h->r2fcap = nc; }
    if (nr > h->r2f_n) { HIP_TRY(hipMemcpyAsync(h->r2f_d + h->r2f_n, h->facet_of_rank.data() + h->r2f_n, (size_t)(nr - h->r2f_n) * sizeof(int), hipMemcpyHostToDevice, s)); h->r2f_n = nr; }
    return 0;
}
// Batch selection by FAMILIES (the unprocessed children of one cut; bslv_benson policy 6).  counts[k] = unprocessed elements whose
// parent -- the newest facet through them -- is dual slot first_facet + k, k < n; *total = all unprocessed elements,
// *older = those whose parent is older than first_facet.
int bslv_poly_children_hist(bslv_poly *h, int first_facet, int n, int *counts, int *total, int *older)
{
    if (!h || first_facet < 0 || n < 0 || (n > 0 && !counts)) { set_error("bslv_poly_children_hist: bad argument"); return BSLV_E_ARG; }
    if (total) *total = 0;
    if (older) *older = 0;
    for (int k = 0; k < n; k++) counts[k] = 0;
    if (!h->initialised || h->nv == 0) return 0;
    hipStream_t s = h->stream;
    int rc;
    const int nv = h->nv, nb = (nv + PB - 1) / PB, nf = h->nf;
    if ((rc = sync_r2f(h))) return rc;
    if (nf + 4 > h->fhistcap) { const int nc = std::max(nf + 4 + 4096, h->fhistcap * 2); if ((rc = grow(&h->fhist_d, 0, (size_t)nc, s))) return rc; h->fhistcap = nc; }
    HIP_TRY(hipMemsetAsync(h->fhist_d, 0, (size_t)nf * sizeof(int), s));
    hipLaunchKernelGGL(k_unproc_hist, dim3(nb), dim3(PB), 0, s, h->P, nv, (const int *)h->r2f_d, h->fhist_d);
    HIP_TRY(hipGetLastError());
    const int m = std::max(0, std::min(n, nf - first_facet));
    if (m > 0) HIP_TRY(hipMemcpyAsync(counts, h->fhist_d + first_facet, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, s));
    if (total || older) {
        // (the histogram itself is small: summed on the host)
        std::vector<int> all(nf);
        HIP_TRY(hipMemcpyAsync(all.data(), h->fhist_d, (size_t)nf * sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        long t = 0, o = 0;
        for (int f = 0; f < nf; f++) { t += all[f]; if (f < first_facet) o += all[f]; }
        if (total) *total = (int)t;
        if (older) *older = (int)o;
    } else HIP_TRY(hipStreamSynchronize(s));
    return 0;
}
// the unprocessed children of the facets facets[0..nfacets) (dual slots, all >= the smallest of them), at most max_out (the newest
// slots), in ascending slot order; parent[k] = dual slot of the parent; *n_out = how many were written
int bslv_poly_children_of(bslv_poly *h, int nfacets, const int *facets, int max_out, int *idx, double *val, int *ideal, int *parent, int *n_out)
{
    if (!h || nfacets < 0 || (nfacets > 0 && !facets) || max_out < 0 || !n_out || (max_out > 0 && !idx)) { set_error("bslv_poly_children_of: bad argument"); return BSLV_E_ARG; }
    *n_out = 0;
    if (!h->initialised || h->nv == 0 || nfacets == 0 || max_out == 0) return 0;
    hipStream_t s = h->stream;
    int rc;
    const int nv = h->nv, nb = (nv + PB - 1) / PB, nf = h->nf;
    int f0 = nf;
    for (int k = 0; k < nfacets; k++) { if (facets[k] < 0 || facets[k] >= nf) { set_error("bslv_poly_children_of: facet %d out of range", facets[k]); return BSLV_E_ARG; } f0 = std::min(f0, facets[k]); }
    std::vector<unsigned char> chosen((size_t)(nf - f0), 0);
    for (int k = 0; k < nfacets; k++) chosen[facets[k] - f0] = 1;
    if ((rc = sync_r2f(h))) return rc;
    if ((size_t)(nf - f0) > h->chosencap) { const size_t nc = std::max((size_t)(nf - f0) + 4096, h->chosencap * 2); if ((rc = grow(&h->chosen_d, 0, nc, s))) return rc; h->chosencap = nc; }
    HIP_TRY(hipMemcpyAsync(h->chosen_d, chosen.data(), chosen.size(), hipMemcpyHostToDevice, s));
    if ((rc = ensure_bsum(h, nb + 1))) return rc;
    hipLaunchKernelGGL(k_unproc_flags_c, dim3(nb), dim3(PB), 0, s, h->P, nv, (const int *)h->r2f_d, (const unsigned char *)h->chosen_d, f0, h->bsum);
    Tri t;
    if ((rc = scan_totals(h, nb, &t))) return rc;          // (synchronises: `chosen` may go)
    const int n = std::min(t.a, max_out);
    if (n == 0) return 0;
    if (n > h->outcap) {
        int nc = std::max(n, h->outcap * 2);
        if ((rc = grow(&h->idx_d, 0, (size_t)nc, s))) return rc;
        if ((rc = grow(&h->val_d, 0, (size_t)nc * h->d, s))) return rc;
        if ((rc = grow(&h->fl_d, 0, (size_t)nc, s))) return rc;
        h->outcap = nc;
    }
    if (n > h->parcap) { if ((rc = grow(&h->par_d, 0, (size_t)std::max(n, h->parcap * 2), s))) return rc; h->parcap = std::max(n, h->parcap * 2); }
    hipLaunchKernelGGL(k_unproc_emit_c, dim3(nb), dim3(PB), 0, s, h->P, nv, (const int *)h->r2f_d, (const unsigned char *)h->chosen_d, f0, (const Tri *)h->bsum, t.a - n, n,
                       h->idx_d, h->val_d, h->fl_d, h->par_d);
    HIP_TRY(hipGetLastError());
    std::vector<unsigned char> fl(n);
    if (parent) HIP_TRY(hipMemcpyAsync(parent, h->par_d, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(idx, h->idx_d, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
    if (val) HIP_TRY(hipMemcpyAsync(val, h->val_d, (size_t)n * h->d * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(fl.data(), h->fl_d, (size_t)n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (ideal) for (int k = 0; k < n; k++) ideal[k] = (fl[k] & F_IDEAL) ? 1 : 0;
    *n_out = n;
    return 0;
}

// poly__get_vrtx (bslv_poly.c:210-226): *rc = 1 when nothing is left
int bslv_poly_next(bslv_poly *h, double *val, int *ideal, int *idx, int *rc_out)
{
    if (!h || !val || !ideal || !idx || !rc_out) return BSLV_E_ARG;
    int cnt = 0;
    int rc = bslv_poly_unprocessed(h, 1, idx, val, ideal, &cnt);
    if (rc) return rc;
    *rc_out = cnt > 0 ? 0 : 1;
    return 0;
}

// ST_BT(primal.sltn, idx) for a list of slots
int bslv_poly_mark(bslv_poly *h, int n, const int *idx)
{
    if (!h || n < 0 || (n > 0 && !idx)) return BSLV_E_ARG;
    if (n == 0) return 0;
    for (int k = 0; k < n; k++) if (idx[k] < 0 || idx[k] >= h->nv) { set_error("bslv_poly_mark: slot %d out of range", idx[k]); return BSLV_E_ARG; }
    int rc;
    if (n > h->outcap) {
        int nc = std::max(n, h->outcap * 2);
        if ((rc = grow(&h->idx_d, 0, (size_t)nc, h->stream))) return rc;
        if ((rc = grow(&h->val_d, 0, (size_t)nc * h->d, h->stream))) return rc;
        if ((rc = grow(&h->fl_d, 0, (size_t)nc, h->stream))) return rc;
        h->outcap = nc;
    }
    HIP_TRY(hipMemcpyAsync(h->idx_d, idx, n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_mark, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->P, h->idx_d, n, F_SLTN);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}

int bslv_poly_set_batch_mode(bslv_poly *h, int mode)
{
    if (!h || mode < 0 || mode > 1) return BSLV_E_ARG;
    h->batch_mode = mode;
    return 0;
}
long bslv_poly_rounds_run(const bslv_poly *h) { return h ? h->rounds_run : 0; }
// Rounds of a chunk end when one holds fewer than min_cuts cuts (0: never -- every cut handed in is applied or found redundant);
// bslv_poly_add_cuts then reports the cuts still alive with rc 2: NOT applied, their dual slots stay behind unused like those of
// redundant cuts, the caller hands the same halfspaces in again with a later batch (bslv_benson_apply does).
int bslv_poly_set_defer(bslv_poly *h, int min_cuts)
{
    if (!h || min_cuts < 0) return BSLV_E_ARG;
    h->r2_defer = min_cuts;
    return 0;
}
int bslv_poly_debug_set(bslv_poly *h, int key, long value)
{
    if (!h) return BSLV_E_ARG;
    switch (key) {
    case 0: h->k2_lds = (size_t)std::max(64L, value); return 0;       /* dynamic LDS of k2_fused (small: force the multi-kernel prune) */
    case 1: h->speculate = value != 0; return 0;
    case 2: h->hot_enabled = value != 0; return 0;
    case 3: h->cross_ub = (int)std::max(0L, value); return 0;
    case 4: h->fm_min = (int)std::max(2L, value); return 0;
    case 6: h->rounds2_enabled = value != 0; return 0;                  /* device-selected rounds of independent cuts inside a hot chunk */
    case 7: h->chunk_cuts = (int)std::min(4096L, std::max(32L, value)); return 0;    /* cuts classified and applied together */
    case 10: h->shard_min = (int)std::max(2L, value); return 0;           /* multi-GPU: facets from this size on have their pair space dealt to the ranks */
    case 13: h->r2_fuse = (int)std::min(2L, std::max(0L, value)); return 0;                           /* one launch for a round's prunes + classification + pair emission (1) / three (0) */
    case 17: h->r2_fork = value != 0; return 0;                           /* rounds: classification of the new vertices on a second stream beside the prunes (1) / one stream (0) */
    case 12: h->r2_spec = value != 0; return 0;                           /* rounds queued one ahead of the host (1) / mailbox read before every round (0) */
    case 16: h->k2_noflags = value != 0; return 0;                        /* multi-kernel prune of large facets without a flag byte per pair (forced; by itself from 4 GiB of flags on) */
    case 15: h->r2_share = value != 0; return 0;                          /* rounds: cuts may share on-plane elements (1) / every element belongs to one cut of a round (0) */
    case 11: h->r2_mis = value != 0; return 0;                            /* rounds: maximal independent set from the conflict matrix (1) / local minima of one order (0) */
    case 9: g_k1_mfma = value != 0; return 0;                            /* incidence kernel K1 on the matrix pipe from 16 halfspaces on (1) or the scalar kernel (0, default); process-wide */
    case 14: h->r2_defer = (int)std::max(0L, value); return 0;           /* see bslv_poly_set_defer */
    case 8: h->r2_min_cuts = (int)std::max(-1L, value); return 0;        /* rounds go on while they average at least this many cuts (0: until every round holds one cut, -1: always) */
    case 5: h->member_lists = value != 0; return 0;                     /* edges of large facets confirmed through member lists (1) or against all elements (0) */           /* facets from this size on confirm edges through the facet-major member lists (4096) */
    default: return BSLV_E_ARG;
    }
}
long bslv_poly_sharded_prunes(const bslv_poly *h) { return h ? h->n_sharded : 0; }
// priorities for the cuts of the NEXT bslv_poly_add_cuts call (prio[b] for cut b; any order-inducing number, e.g. the depth z of the cut):
// with BSLV_R2_ORDER = 1 / 2 the rounds of independent cuts give the cuts of a chunk their priority in ascending / descending order of it
int bslv_poly_set_cut_priorities(bslv_poly *h, int n, const double *prio)
{
    if (!h || n < 0 || (n && !prio)) return BSLV_E_ARG;
    h->cut_prio.assign(prio, prio + n);
    return 0;
}
// Capacity ahead of need: element, edge and incidence-pool arrays double when they fill up, and every doubling is a hipMalloc + copy +
// hipFree with a stream synchronisation in the middle of a batch of cuts (~0.5 ms each, 4.5 % of S-mid's cut phase: DESIGN.md 4e item 10).
// A caller that knows it is about to run thousands of steps reserves once.  Not while a chunk is open (the rounds hold views of the arrays).
// Zeros leave that capacity alone; nothing ever shrinks.
// The projection sub-band of poly__cut (bslv_poly.c:666-674), see k_snap.  Off by default: the rounds of independent cuts classify a chunk's
// elements against all of its cuts before the first is applied, which a moved element would invalidate; with the band on, cuts are applied
// one at a time in the order handed in, as the reference applies them.  *moved (may be NULL) receives the elements moved so far.
int bslv_poly_set_snap(bslv_poly *h, int on)
{
    if (!h) return BSLV_E_ARG;
    if (h->hot) { set_error("bslv_poly_set_snap: a chunk of cuts is open"); return BSLV_E_ARG; }
    int rc = settle_k2(h);
    if (rc) return rc;
    h->pre_f = -1;                       // (a classification made ahead of its turn is not used across the switch)
    h->snap = on != 0;
    return 0;
}
int bslv_poly_snapped(bslv_poly *h, long *moved)
{
    if (!h || !moved) return BSLV_E_ARG;
    unsigned long long v = 0;
    if (h->snapped_d) { HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipMemcpy(&v, h->snapped_d, sizeof(v), hipMemcpyDeviceToHost)); }
    *moved = (long)v;
    return 0;
}
int bslv_poly_reserve(bslv_poly *h, long elements, long edges, long pool_words)
{
    if (!h || elements < 0 || edges < 0 || pool_words < 0) return BSLV_E_ARG;
    if (h->hot) { set_error("bslv_poly_reserve: a chunk of cuts is open"); return BSLV_E_ARG; }
    if (elements > 0x7FFFFF00 / 2 || edges > 0x7FFFFF00 / 2 || (unsigned long)pool_words > 0xF0000000ul) { set_error("bslv_poly_reserve: beyond the 32-bit indices of the engine"); return BSLV_E_CAPACITY; }
    int rc;
    if (elements > h->P.cap && (rc = ensure_vcap(h, (int)elements))) return rc;
    if (edges > h->ecap && (rc = ensure_ecap(h, (int)edges))) return rc;
    if ((size_t)pool_words > h->poolcap && (rc = ensure_pool(h, (size_t)pool_words))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return 0;
}
int bslv_poly_largest_facet(const bslv_poly *h) { return h ? h->largest_facet : 0; }      // members of the largest new facet that went through the multi-kernel prune
long bslv_poly_noflag_prunes(const bslv_poly *h) { return h ? h->n_noflag_prunes : 0; }      // large-facet prunes that emitted by testing the listed pair blocks again (no flag byte per pair)
long bslv_poly_rounds2_late_left(const bslv_poly *h) { return h ? h->r2_late_left : 0; }
long bslv_poly_rounds2_torn_reads(const bslv_poly *h) { return h ? h->r2_torn_reads + h->mail_torn_reads : 0; }      // (both kinds of mailbox)

// Host-only self-test of the two mailbox readers (no GPU): a writer thread publishes `rounds` messages the WRONG way round --
// the sequence number first, the content afterwards, word by word -- into plain host memory, the reader takes them with
// mail_try_read / rstate_try_read.  Every message's content is a function of its sequence number, so a reader that accepted
// old content under a new number is caught.  which: 0 = Mail (one-cut pipeline), 1 = RState (rounds).  Returns 0 and the number
// of reads that had to be repeated in *torn_out, or -1 - (number of messages accepted with foreign content).
int bslv_selftest_mailbox(int which, int rounds, long *torn_out)
{
    if (rounds < 1 || which < 0 || which > 1) return BSLV_E_ARG;
    long torn = 0, bad = 0;
    std::atomic<int> ack{0};
    if (which == 0) {
        Mail *box = new Mail();
        memset((void *)box, 0, sizeof *box);
        box->chk2 = mail_sum2(Tri{0, 0, 0}, 0);
        volatile Mail *m = box;
        std::thread writer([&] {
            for (int q = 1; q <= rounds; q++) {
                while (ack.load(std::memory_order_acquire) != q - 1) { }
                const int c4[4] = {q, 2 * q, 3 * q, 4 * q};
                const Tri t{5 * q, 6 * q, 7 * q}, k2{8 * q, 9 * q, 10 * q};
                m->seq = q;                                             // the number first ...
                __sync_synchronize();
                for (int k = 0; k < 4; k++) { m->cnt[k] = c4[k]; for (volatile int d = 0; d < 50; d++) { } }
                const_cast<Mail *>(box)->t = t; const_cast<Mail *>(box)->k2t = k2; m->k2seq = 11 * q;
                for (volatile int d = 0; d < 200; d++) { }
                m->chk2 = mail_sum2(k2, 11 * q);
                m->chk = mail_sum(c4, t, q);                            // ... the checksum last
            }
        });
        for (int q = 1; q <= rounds; q++) {
            Mail got;
            while (!mail_try_read(m, q, &got, &torn)) { }
            if (got.cnt[0] != q || got.cnt[3] != 4 * q || got.t.a != 5 * q || got.t.c != 7 * q || got.k2t.b != 9 * q || got.k2seq != 11 * q) bad++;
            ack.store(q, std::memory_order_release);
        }
        writer.join();
        delete box;
    } else {
        RState *box = new RState();
        memset((void *)box, 0, sizeof *box);
        volatile RState *m = box;
        std::thread writer([&] {
            for (int q = 1; q <= rounds; q++) {
                while (ack.load(std::memory_order_acquire) != q - 1) { }
                RState st;
                memset((void *)&st, 0, sizeof st);
                st.round = q; st.S = 3 * q; st.nalive = 5 * q; st.napplied = 7 * q; st.ne_next = 11 * q; st.err_d[11] = 13.0 * q;
                st.chk = rstate_sum(st, q);
                m->seq = q;                                             // the number first ...
                __sync_synchronize();
                const unsigned *src = reinterpret_cast<const unsigned *>(&st);
                volatile unsigned *dst = reinterpret_cast<volatile unsigned *>(box);
                const size_t seq_word = offsetof(RState, seq) / 4, chk_word = offsetof(RState, chk) / 4;
                for (size_t w = 0; w < sizeof(RState) / 4; w++) { if (w == seq_word || w == chk_word) continue; dst[w] = src[w]; for (volatile int d = 0; d < 10; d++) { } }
                dst[chk_word] = st.chk;                                 // ... the checksum last
            }
        });
        for (int q = 1; q <= rounds; q++) {
            RState got;
            while (!rstate_try_read(m, q, &got, &torn)) { }
            if (got.round != q || got.S != 3 * q || got.nalive != 5 * q || got.napplied != 7 * q || got.ne_next != 11 * q || got.err_d[11] != 13.0 * q) bad++;
            ack.store(q, std::memory_order_release);
        }
        writer.join();
        delete box;
    }
    if (torn_out) *torn_out = torn;
    return bad ? (int)(-1 - bad) : 0;
}
int bslv_poly_path_stats(const bslv_poly *h, long out[6])
{
    if (!h || !out) return BSLV_E_ARG;
    out[0] = h->hot_chunks; out[1] = h->n_spec; out[2] = h->n_declined; out[3] = h->n_k2_fallback; out[4] = h->n_single; out[5] = h->n_fm;
    return 0;
}
long bslv_poly_conflict_pairs(const bslv_poly *h) { return h ? h->conf_pairs : 0; }
// TEST: ntiles random 16 x 16 tiles of dot products of length dim, chained v_mfma_f64_16x16x4 against the scalar fma chain;
// *mismatches = results that differ in any bit (0 expected: the MFMA accumulates its K steps in ascending order, one rounding each)
int bslv_k1_mfma_selftest(int dim, int ntiles, unsigned long long seed, long *mismatches)
{
    if (dim < 2 || dim > 10 || ntiles < 1 || !mismatches) return BSLV_E_ARG;
    std::vector<double> X((size_t)ntiles * dim * 16), H((size_t)ntiles * 16 * dim);
    unsigned long long z = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() { z += 0x9E3779B97F4A7C15ull; unsigned long long t = z; t = (t ^ (t >> 30)) * 0xBF58476D1CE4E5B9ull; t = (t ^ (t >> 27)) * 0x94D049BB133111EBull; t ^= t >> 31;
                       return ((double)(t >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0) * (1.0 + (double)(t & 7)); };
    for (double &v : X) v = rnd();
    for (double &v : H) v = rnd();
    double *Xd = nullptr, *Hd = nullptr; unsigned long long *md = nullptr;
    HIP_TRY(malloc0(&Xd, X.size() * 8)); HIP_TRY(malloc0(&Hd, H.size() * 8)); HIP_TRY(malloc0(&md, 8));
    HIP_TRY(hipMemcpy(Xd, X.data(), X.size() * 8, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(Hd, H.data(), H.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(md, 0, 8));
    const dim3 g((unsigned)(((size_t)ntiles * 64 + 255) / 256)), b(256);
    switch (dim) {
#define CASE(D) case D: hipLaunchKernelGGL(k_k1_mfma_selftest<D>, g, b, 0, 0, (const double *)Xd, (const double *)Hd, ntiles, md); break;
        CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
#undef CASE
    }
    HIP_TRY(hipGetLastError());
    unsigned long long m = 0;
    HIP_TRY(hipMemcpy(&m, md, 8, hipMemcpyDeviceToHost));
    (void)hipFree(Xd); (void)hipFree(Hd); (void)hipFree(md);
    *mismatches = (long)m;
    return 0;
}
// device-selected rounds inside hot chunks: out[0] rounds, [1] cuts applied in them, [2] chunks, [3] prunes that went through the
// multi-kernel path, [4] rounds taken back for want of capacity
int bslv_poly_rounds2_stats(const bslv_poly *h, long out[5])
{
    if (!h || !out) return BSLV_E_ARG;
    out[0] = h->r2_rounds; out[1] = h->r2_cuts; out[2] = h->r2_chunks; out[3] = h->r2_fallback_prunes; out[4] = h->r2_declined;
    return 0;
}
// MEASUREMENT ONLY (bench / profiles): turns the engine into nv synthetic live points so that the
// batched incidence kernel can be timed at sizes beyond the caches.  The polyhedron is destroyed.
int bslv_poly_bench_fill(bslv_poly *h, int nv, unsigned long long seed)
{
    if (!h || nv < 1) return BSLV_E_ARG;
    int rc;
    if ((rc = ensure_vcap(h, nv))) return rc;
    hipLaunchKernelGGL(k_bench_fill, dim3((nv + 255) / 256), dim3(256), 0, h->stream, h->P, nv, seed);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->nv = nv; h->ne = 0; h->initialised = true;
    if (h->facet_of_rank.empty()) h->facet_of_rank.push_back(0);
    return 0;
}

int bslv_poly_dim(const bslv_poly *h) { return h ? h->d : 0; }
int bslv_poly_nprimal(const bslv_poly *h) { return h ? h->nv : 0; }
int bslv_poly_ndual(const bslv_poly *h) { return h ? h->nf : 0; }
long bslv_poly_nedges(const bslv_poly *h) { return h ? h->ne : 0; }
long bslv_poly_pair_tests(const bslv_poly *h) { return h ? h->pair_tests : 0; }
long bslv_poly_new_vertices(const bslv_poly *h) { return h ? h->new_vertices : 0; }

// slot-indexed dumps (host buffers sized by the counts above)
// coordinates of the element slots first .. first + count - 1 only (row-major count x dim): what a caller that mirrors the polyhedron
// needs after a cut -- the coordinates of a slot never change, only new slots appear (poly_compat.hip)
int bslv_poly_get_primal_range(bslv_poly *h, int first, int count, double *coords)
{
    if (!h || first < 0 || count < 0 || first + count > h->nv || (count && !coords)) { set_error("bslv_poly_get_primal_range: bad argument"); return BSLV_E_ARG; }
    if (count == 0) return 0;
    const int d = h->d;
    std::vector<double> col(count);
    for (int k = 0; k < d; k++) {
        HIP_TRY(hipMemcpy(col.data(), h->P.X + (size_t)k * h->P.cap + first, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < count; i++) coords[(size_t)i * d + k] = col[i];
    }
    return 0;
}
int bslv_poly_get_primal(bslv_poly *h, unsigned char *used, unsigned char *ideal, unsigned char *sltn, double *coords)
{
    if (!h) return BSLV_E_ARG;
    const int nv = h->nv, d = h->d;
    if (nv == 0) return 0;
    std::vector<unsigned char> fl(nv);
    HIP_TRY(hipMemcpy(fl.data(), h->P.flag, nv, hipMemcpyDeviceToHost));
    for (int i = 0; i < nv; i++) {
        if (used) used[i] = (fl[i] & F_USED) ? 1 : 0;
        if (ideal) ideal[i] = (fl[i] & F_IDEAL) ? 1 : 0;
        if (sltn) sltn[i] = (fl[i] & F_SLTN) ? 1 : 0;
    }
    if (coords) {
        std::vector<double> col(nv);
        for (int k = 0; k < d; k++) {
            HIP_TRY(hipMemcpy(col.data(), h->P.X + (size_t)k * h->P.cap, (size_t)nv * sizeof(double), hipMemcpyDeviceToHost));
            for (int i = 0; i < nv; i++) coords[(size_t)i * d + k] = col[i];
        }
    }
    return 0;
}

static int fetch_inc(bslv_poly *h, std::vector<unsigned char> &fl, std::vector<unsigned> &off, std::vector<int> &len, std::vector<int> &pool)
{
    const int nv = h->nv;
    fl.resize(nv); off.resize(nv); len.resize(nv); pool.resize(h->poolused);
    if (nv == 0) return 0;
    HIP_TRY(hipMemcpy(fl.data(), h->P.flag, nv, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(off.data(), h->P.inc_off, nv * sizeof(unsigned), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(len.data(), h->P.inc_len, nv * sizeof(int), hipMemcpyDeviceToHost));
    if (h->poolused) HIP_TRY(hipMemcpy(pool.data(), h->P.pool, (size_t)h->poolused * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

long bslv_poly_ninc(bslv_poly *h)
{
    if (!h || h->nv == 0) return 0;
    std::vector<unsigned char> fl(h->nv);
    std::vector<int> len(h->nv);
    if (hipMemcpy(fl.data(), h->P.flag, h->nv, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (hipMemcpy(len.data(), h->P.inc_len, h->nv * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    long n = 0;
    for (int i = 0; i < h->nv; i++) if (fl[i] & F_USED) n += len[i];
    return n;
}

int bslv_poly_get_inc(bslv_poly *h, int *pairs)
{
    if (!h || !pairs) return BSLV_E_ARG;
    std::vector<unsigned char> fl; std::vector<unsigned> off; std::vector<int> len, pool;
    int rc = fetch_inc(h, fl, off, len, pool);
    if (rc) return rc;
    long n = 0;
    for (int i = 0; i < h->nv; i++) {
        if (!(fl[i] & F_USED)) continue;
        for (int j = 0; j < len[i]; j++) { pairs[2 * n] = i; pairs[2 * n + 1] = h->facet_of_rank[pool[off[i] + j]]; n++; }
    }
    return 0;
}

int bslv_poly_get_edges(bslv_poly *h, int *ab)
{
    if (!h || !ab) return BSLV_E_ARG;
    if (h->ne) HIP_TRY(hipMemcpy(ab, h->E[h->ecur], (size_t)h->ne * sizeof(int2), hipMemcpyDeviceToHost));
    return 0;
}

// a dual slot is live iff it was applied and some live element lies on it
int bslv_poly_get_dual(bslv_poly *h, unsigned char *used, unsigned char *ideal, double *coords)
{
    if (!h) return BSLV_E_ARG;
    const int nf = h->nf;
    std::vector<unsigned char> live(nf, 0);
    if (h->initialised) {
        // liveness of every facet rank on the device (one byte per rank comes back, not the incidence pool)
        int rc;
        if ((rc = settle_k2(h))) return rc;
        const int nr = (int)h->facet_of_rank.size();
        if (nr > 0) {
            unsigned char *live_d = nullptr;
            HIP_TRY(malloc0s(&live_d, (size_t)nr, h->stream));
            HIP_TRY(hipMemsetAsync(live_d, 0, (size_t)nr, h->stream));
            if (h->nv > 0) hipLaunchKernelGGL(k_live_ranks, dim3((h->nv + PB / WAVE - 1) / (PB / WAVE)), dim3(PB), 0, h->stream, h->P, h->nv, live_d);
            std::vector<unsigned char> lr((size_t)nr);
            hipError_t e = hipMemcpyAsync(lr.data(), live_d, (size_t)nr, hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            (void)hipFree(live_d);
            HIP_TRY(e);
            for (int rk = 0; rk < nr; rk++) if (lr[rk]) { const int f = h->facet_of_rank[rk]; if (h->fapplied[f]) live[f] = 1; }
        }
    } else
        for (int f = 0; f < nf; f++) live[f] = h->fapplied[f];
    if (used) memcpy(used, live.data(), nf);
    if (ideal) memcpy(ideal, h->fideal.data(), nf);
    if (coords) memcpy(coords, h->Y.data(), (size_t)nf * h->d * sizeof(double));
    return 0;
}

// poly__update_adjacence on the dual side (bslv_poly.c:992-1010, called bslv_algs.c:398,1144,1569):
// all pairs of live facets through the same pair kernel, on the transposed incidence.
int bslv_poly_dual_adjacency(bslv_poly *h)
{
    if (!h) return BSLV_E_ARG;
    h->dual_edges.clear();
    if (!h->initialised) return 0;
    std::vector<unsigned char> fl; std::vector<unsigned> off; std::vector<int> len, pool;
    int rc = fetch_inc(h, fl, off, len, pool);
    if (rc) return rc;
    const int nf = (int)h->facet_of_rank.size();      // rank space (incidence lists hold ranks)
    std::vector<int> flen(nf, 0);
    std::vector<unsigned char> live(nf, 0);
    for (int i = 0; i < h->nv; i++) if (fl[i] & F_USED) for (int j = 0; j < len[i]; j++) { int f = pool[off[i] + j]; flen[f]++; if (h->fapplied[h->facet_of_rank[f]]) live[f] = 1; }
    std::vector<unsigned> foff(nf + 1, 0);
    for (int f = 0; f < nf; f++) foff[f + 1] = foff[f] + flen[f];
    std::vector<int> fpool(foff[nf] ? foff[nf] : 1), fill(nf, 0), ids;
    for (int i = 0; i < h->nv; i++) if (fl[i] & F_USED) for (int j = 0; j < len[i]; j++) { int f = pool[off[i] + j]; fpool[foff[f] + fill[f]++] = i; }
    for (int f = 0; f < nf; f++) if (live[f]) ids.push_back(f);
    const int nm = (int)ids.size();
    if (nm < 2) return 0;
    hipStream_t s = h->stream;
    int *ids_d = nullptr, *flen_d = nullptr, *fpool_d = nullptr; unsigned *foff_d = nullptr; unsigned char *live_d = nullptr;
    int2 *out_d = nullptr;
    auto cleanup = [&]() { for (void *p : {(void *)ids_d, (void *)flen_d, (void *)fpool_d, (void *)foff_d, (void *)live_d, (void *)out_d}) if (p) (void)hipFree(p); };
#define TRYC(e) do { hipError_t _e = (e); if (_e != hipSuccess) { set_error("%s failed: %s", #e, hipGetErrorString(_e)); cleanup(); return BSLV_E_NODEVICE; } } while (0)
    TRYC(malloc0(&ids_d, nm * sizeof(int)));
    TRYC(malloc0(&flen_d, nf * sizeof(int)));
    TRYC(malloc0(&foff_d, (nf + 1) * sizeof(unsigned)));
    TRYC(malloc0(&fpool_d, fpool.size() * sizeof(int)));
    TRYC(malloc0(&live_d, nf));
    TRYC(hipMemcpy(ids_d, ids.data(), nm * sizeof(int), hipMemcpyHostToDevice));
    TRYC(hipMemcpy(flen_d, flen.data(), nf * sizeof(int), hipMemcpyHostToDevice));
    TRYC(hipMemcpy(foff_d, foff.data(), (nf + 1) * sizeof(unsigned), hipMemcpyHostToDevice));
    TRYC(hipMemcpy(fpool_d, fpool.data(), fpool.size() * sizeof(int), hipMemcpyHostToDevice));
    TRYC(hipMemcpy(live_d, live.data(), nf, hipMemcpyHostToDevice));
    DualView D{ids_d, foff_d, flen_d, fpool_d, live_d};
    // rows are processed in slabs so the block table and flag array stay bounded
    const long max_blocks = 1 << 20;
    int row = 0;
    while (row + 1 < nm) {
        std::vector<PairBlk> blks;
        int r1 = row;
        while (r1 + 1 < nm && (long)blks.size() + (nm - r1 - 1 + PB - 1) / PB <= max_blocks) {
            for (int j0 = r1 + 1; j0 < nm; j0 += PB) blks.push_back(PairBlk{r1, j0});
            r1++;
        }
        if (r1 == row) { set_error("dual adjacency: a single row exceeds the block budget"); cleanup(); return BSLV_E_CAPACITY; }
        const int nbp = (int)blks.size();
        for (int i = row; i < r1; i++) h->pair_tests += nm - 1 - i;
        if (nbp > h->blkcap) { if ((rc = grow(&h->blks, 0, (size_t)nbp, s))) { cleanup(); return rc; } h->blkcap = nbp; }
        if ((size_t)nbp * PB > h->pflagcap) { if ((rc = grow(&h->pflag, 0, (size_t)nbp * PB, s))) { cleanup(); return rc; } h->pflagcap = (size_t)nbp * PB; }
        if ((rc = ensure_bsum(h, nbp + 1))) { cleanup(); return rc; }
        TRYC(hipMemcpyAsync(h->blks, blks.data(), (size_t)nbp * sizeof(PairBlk), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_dpair_flags, dim3(nbp), dim3(PB), 0, s, h->P, D, nm, h->blks, h->pflag, h->bsum);
        Tri tp{0, 0, 0};
        if ((rc = scan_totals(h, nbp, &tp))) { cleanup(); return rc; }
        if (tp.a > 0) {
            if (out_d) { (void)hipFree(out_d); out_d = nullptr; }
            TRYC(malloc0(&out_d, (size_t)tp.a * sizeof(int2)));
            hipLaunchKernelGGL(k_pair_emit, dim3(nbp), dim3(PB), 0, s, ids_d, nm, h->blks, h->pflag, h->bsum, out_d, 0);
            size_t base = h->dual_edges.size();
            h->dual_edges.resize(base + 2 * (size_t)tp.a);
            TRYC(hipMemcpyAsync(&h->dual_edges[base], out_d, (size_t)tp.a * sizeof(int2), hipMemcpyDeviceToHost, s));
            TRYC(hipStreamSynchronize(s));
            for (size_t k = base; k < h->dual_edges.size(); k += 2) {          // ranks -> dual slot ids
                int a = h->facet_of_rank[h->dual_edges[k]], b = h->facet_of_rank[h->dual_edges[k + 1]];
                h->dual_edges[k] = std::min(a, b); h->dual_edges[k + 1] = std::max(a, b);
            }
        }
        row = r1;
    }
#undef TRYC
    cleanup();
    return 0;
}

long bslv_poly_ndual_edges(const bslv_poly *h) { return h ? (long)h->dual_edges.size() / 2 : 0; }
int bslv_poly_get_dual_edges(bslv_poly *h, int *ab)
{
    if (!h || !ab) return BSLV_E_ARG;
    memcpy(ab, h->dual_edges.data(), h->dual_edges.size() * sizeof(int));
    return 0;
}

}  // extern "C"
