// dist.hip -- the one exchange step of the multi-GPU path, in the library: one process per GPU, the vertex
// batch of an outer iteration is dealt to the ranks, every rank solves its LPs, ONE all-gather of the fixed-size cut records
// over RCCL (xGMI), every rank applies all records in the same order to its replica of the polyhedron.
//
// RCCL is loaded on first use (dlopen of librccl.so.1: the library itself has no link-time dependency on it, so a
// single-GPU host without RCCL still loads libbslv_hip.so).  The communicator is created from a 128-byte ncclUniqueId that
// the host side distributes (bensolve_hip: TCP to MASTER_ADDR; bench.py: one torch.distributed broadcast).  For tests on a
// one-GPU box, where RCCL refuses two ranks on one device, the same step runs over a caller-supplied all-gather.
#include "common.h"
#include <dlfcn.h>
#include <vector>
#include <chrono>
#include <string>

using namespace bslv;

namespace {
// the handful of RCCL entry points used (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclAllGather, ncclCommDestroy)
typedef struct { char internal[128]; } uid_t128;
typedef void *comm_t;
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(uid_t128 *) = nullptr;
    int (*CommInitRank)(comm_t *, int, uid_t128, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, comm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
struct Dist {
    int rank = 0, world = 1;
    Rccl R;
    comm_t comm = nullptr;
    bslv_allgather_fn cb = nullptr;
    void *ctx = nullptr;
    hipStream_t stream = nullptr;
    double *send_d = nullptr, *recv_d = nullptr;
    size_t cap = 0;                 // doubles per rank the device buffers hold
    long gathers = 0;
    double gather_ms = 0;
} g;

int load_rccl()
{
    if (g.R.lib) return 0;
    // a copy that the process has already loaded comes first (PyTorch ships its own librccl.so: two copies of RCCL in one process
    // end in a double free when the process exits)
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) { set_error("RCCL not found (dlopen librccl.so.1): %s", dlerror()); return BSLV_E_NODEVICE; }
    g.R.GetUniqueId = (int (*)(uid_t128 *))dlsym(lib, "ncclGetUniqueId");
    g.R.CommInitRank = (int (*)(comm_t *, int, uid_t128, int))dlsym(lib, "ncclCommInitRank");
    g.R.AllGather = (int (*)(const void *, void *, size_t, int, comm_t, hipStream_t))dlsym(lib, "ncclAllGather");
    g.R.CommDestroy = (int (*)(comm_t))dlsym(lib, "ncclCommDestroy");
    g.R.GetErrorString = (const char *(*)(int))dlsym(lib, "ncclGetErrorString");
    if (!g.R.GetUniqueId || !g.R.CommInitRank || !g.R.AllGather || !g.R.CommDestroy) { set_error("RCCL: entry points missing in librccl"); dlclose(lib); return BSLV_E_NODEVICE; }
    g.R.lib = lib;
    return 0;
}
int nccl_fail(const char *what, int code)
{
    set_error("%s failed: %s", what, g.R.GetErrorString ? g.R.GetErrorString(code) : "RCCL error");
    return BSLV_E_NODEVICE;
}
constexpr int NCCL_DOUBLE = 8;      // ncclFloat64 (rccl.h: ncclDataType_t)
}  // namespace

extern "C" {

int bslv_dist_rank(void) { return g.rank; }
int bslv_dist_world(void) { return g.world; }

int bslv_dist_unique_id(unsigned char *out, int len)
{
    if (!out || len < 128) { set_error("bslv_dist_unique_id: the buffer must hold 128 bytes"); return BSLV_E_ARG; }
    int rc = load_rccl();
    if (rc) return rc;
    uid_t128 id;
    int e = g.R.GetUniqueId(&id);
    if (e) return nccl_fail("ncclGetUniqueId", e);
    memcpy(out, id.internal, 128);
    return 0;
}

void bslv_dist_finalize(void)
{
    if (g.comm && g.R.CommDestroy) (void)g.R.CommDestroy(g.comm);
    g.comm = nullptr; g.cb = nullptr; g.ctx = nullptr;
    if (g.send_d) (void)hipFree(g.send_d);
    if (g.recv_d) (void)hipFree(g.recv_d);
    g.send_d = g.recv_d = nullptr; g.cap = 0;
    if (g.stream) (void)hipStreamDestroy(g.stream);
    g.stream = nullptr;
    g.rank = 0; g.world = 1;
}

// communicator over the calling thread's current HIP device (bslv_set_device(LOCAL_RANK) first)
int bslv_dist_init(int rank, int world, const unsigned char *id, int len)
{
    if (world < 1 || rank < 0 || rank >= world || !id || len < 128) { set_error("bslv_dist_init: bad argument"); return BSLV_E_ARG; }
    bslv_dist_finalize();
    int rc = load_rccl();
    if (rc) return rc;
    uid_t128 uid;
    memcpy(uid.internal, id, 128);
    int e = g.R.CommInitRank(&g.comm, world, uid, rank);
    if (e) { g.comm = nullptr; return nccl_fail("ncclCommInitRank", e); }
    HIP_TRY(hipStreamCreate(&g.stream));
    g.rank = rank; g.world = world;
    return 0;
}

int bslv_dist_init_callback(int rank, int world, bslv_allgather_fn fn, void *ctx)
{
    if (world < 1 || rank < 0 || rank >= world || !fn) { set_error("bslv_dist_init_callback: bad argument"); return BSLV_E_ARG; }
    bslv_dist_finalize();
    g.cb = fn; g.ctx = ctx; g.rank = rank; g.world = world;
    return 0;
}

// every rank contributes `count` doubles (host memory) and receives world * count, in rank order
int bslv_dist_allgather(const double *send, double *recv, int count)
{
    if (!send || !recv || count < 1) return BSLV_E_ARG;
    auto t0 = std::chrono::steady_clock::now();
    if (g.world == 1 && !g.comm && !g.cb) { memcpy(recv, send, (size_t)count * sizeof(double)); return 0; }
    if (g.cb) {
        int rc = g.cb(send, recv, count, g.ctx);
        if (rc) { set_error("the all-gather callback failed (%d)", rc); return BSLV_E_STATE; }
    } else {
        if (!g.comm) { set_error("bslv_dist_allgather: no communicator"); return BSLV_E_STATE; }
        if ((size_t)count > g.cap) {
            if (g.send_d) (void)hipFree(g.send_d);
            if (g.recv_d) (void)hipFree(g.recv_d);
            g.send_d = g.recv_d = nullptr;
            g.cap = 0;
            const size_t nc = std::max<size_t>((size_t)count, 4096);
            HIP_TRY(malloc0(&g.send_d, nc * sizeof(double)));
            HIP_TRY(malloc0(&g.recv_d, nc * g.world * sizeof(double)));
            g.cap = nc;
        }
        HIP_TRY(hipMemcpyAsync(g.send_d, send, (size_t)count * sizeof(double), hipMemcpyHostToDevice, g.stream));
        int e = g.R.AllGather(g.send_d, g.recv_d, (size_t)count, NCCL_DOUBLE, g.comm, g.stream);
        if (e) return nccl_fail("ncclAllGather", e);
        HIP_TRY(hipMemcpyAsync(recv, g.recv_d, (size_t)count * g.world * sizeof(double), hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
    }
    g.gathers++;
    g.gather_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}
int bslv_dist_stats(long *gathers, double *ms)
{
    if (gathers) *gathers = g.gathers;
    if (ms) *ms = g.gather_ms;
    return 0;
}

// One outer iteration over all ranks: the batch (max_batch_global vertices) is dealt to the ranks, each rank solves its shard,
// the records are all-gathered as fixed-size blocks [count ; records ; zero padding], every rank applies all of them.
// stats / ms as bslv_benson_step (stats[0] = LPs of ALL ranks); additionally ms[1] includes the exchange.
static double g_phase[4] = {0, 0, 0, 0};
int bslv_benson_step_dist(bslv_benson *h, int max_batch_global, long *stats, double *ms)
{
    if (!h || max_batch_global < 1) return BSLV_E_ARG;
    auto t0 = std::chrono::steady_clock::now();
    const int world = g.world, rank = g.rank, RL = bslv_benson_record_len(h);
    int nl = 0, nt = 0, rc;
    if ((rc = bslv_benson_collect(h, max_batch_global, rank, world, &nl, &nt))) return rc;
    // upper bound of one rank's shard (the dealing rule of bslv_benson_collect)
    const int cap = (nt + world - 1) / world + std::max(1, nt / (4 * world)) + 1;
    if (nl > cap) { set_error("internal: shard of %d LPs exceeds the dealing rule's bound %d", nl, cap); return BSLV_E_STATE; }
    std::vector<double> block((size_t)(cap + 1) * RL, 0.0), all((size_t)(cap + 1) * RL * world);
    int piv = 0, ls = 0;
    auto t1 = std::chrono::steady_clock::now();
    // A failure that only this rank sees (its tableau pool exhausted, a device error in its LPs) must not keep it out of the
    // collective -- the others would wait in ncclAllGather for ever.  The header row of the block carries a status word: a
    // failing rank sends an empty block with its error code, and every rank returns an error when any rank reports one.
    const int rc_local = bslv_benson_solve_local(h, block.data() + RL, &piv, &ls);
    std::string local_err = rc_local ? bslv_last_error() : "";
    block[0] = rc_local ? 0 : nl;
    block[1] = rc_local;
    auto t2 = std::chrono::steady_clock::now();
    if ((rc = bslv_dist_allgather(block.data(), all.data(), (cap + 1) * RL))) return rc;
    auto tg = std::chrono::steady_clock::now();
    for (int r = 0; r < world; r++) {
        const int st_r = (int)all[(size_t)r * (cap + 1) * RL + 1];
        if (st_r) {
            if (r == rank) set_error("rank %d: %s", rank, local_err.c_str());
            else set_error("rank %d reported error %d in its LP phase (its own message names the cause); this rank stops with it", r, st_r);
            return r == rank ? rc_local : BSLV_E_STATE;
        }
    }
    std::vector<double> rec;
    rec.reserve((size_t)std::max(nt, 1) * RL);
    int total = 0;
    for (int r = 0; r < world; r++) {
        const double *b = &all[(size_t)r * (cap + 1) * RL];
        const int n = (int)b[0];
        if (n < 0 || n > cap) { set_error("internal: rank %d announced %d records (bound %d)", r, n, cap); return BSLV_E_STATE; }
        rec.insert(rec.end(), b + RL, b + (size_t)(1 + n) * RL);
        total += n;
    }
    if (total != nt) { set_error("internal: %d records gathered, %d vertices dealt", total, nt); return BSLV_E_STATE; }
    long st[5] = {0, 0, 0, 0, 0};
    if ((rc = bslv_benson_apply(h, total, rec.data(), st))) return rc;
    auto t3 = std::chrono::steady_clock::now();
    if (stats) { for (int k = 0; k < 5; k++) stats[k] = st[k]; stats[5] = piv; stats[6] = ls; stats[7] = bslv_benson_unprocessed_left(h); }
    if (ms) {
        ms[0] = std::chrono::duration<double, std::milli>(t2 - t1).count();
        ms[1] = std::chrono::duration<double, std::milli>(t3 - t2).count() + std::chrono::duration<double, std::milli>(t1 - t0).count();
        ms[2] = std::chrono::duration<double, std::milli>(t3 - t0).count();
    }
    // the four phases of this rank's step, for the scaling report (bslv_dist_last_phases): the all-gather includes the wait for the slowest rank's LPs
    g_phase[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
    g_phase[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
    g_phase[2] = std::chrono::duration<double, std::milli>(tg - t2).count();
    g_phase[3] = std::chrono::duration<double, std::milli>(t3 - tg).count();
    return 0;
}
// host wall clock of the phases of the last bslv_benson_step_dist on this rank, ms: [0] collect (choice and dealing of the batch),
// [1] this rank's LPs, [2] the all-gather of the cut records (waits for the slowest rank), [3] application of ALL ranks' cuts (replicated)
int bslv_dist_last_phases(double out[4])
{
    if (!out) return BSLV_E_ARG;
    for (int k = 0; k < 4; k++) out[k] = g_phase[k];
    return 0;
}

}  // extern "C"
