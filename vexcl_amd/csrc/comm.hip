// RCCL transport behind the C ABI:
//   vexhip_comm_*            communicator over the devices of ONE process (the reference's model: one vex::Context
//                            drives every GPU) or one rank of a one-process-per-GPU job;
//   vexhip_halo_exchange     the ghost exchange of SpMat::apply as ONE grouped ncclSend / ncclRecv step over xGMI --
//                            replaces the two PCIe hops and four finish() fences of vexcl/spmat.hpp:125-183 and
//                            sparse/distributed.hpp:347-428;
//   vexhip_allreduce_scalar  the Reductor's final combine (reductor.hpp:412-436 folds D x 8*CU partials on the host);
//   vexhip_allgather         the scan carry (scan.hpp:445-457);
//   vexhip_dist_spmv_*       one rank's whole product step -- pack, exchange, local part, remote part -- issued from
//                            C++ on two streams (optionally replayed from a hipGraph): the per-step host cost is a
//                            handful of launches instead of a Python loop over torch.distributed requests.
// librccl is loaded on first use (dlopen): a single-GPU user of libvexhip.so never pays for it.
#include "common.hpp"
#include "halo.hpp"

#include <rccl/rccl.h>
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

namespace vexhip {
// spmat.hip: the stored strip of a rank as the operand of the one-launch step (halo.hpp)
int spmat_halo_geometry(const vexhip_spmat *h, int *planes, int *lines_per_plane, int *line_length, int *value_type);
int spmat_halo_general(const vexhip_spmat *h, int64_t halo, int64_t rows_ext, int *reach, int *value_type);
int spmat_device(const vexhip_spmat *h, int *dev);
int spmat_apply_halo(const vexhip_spmat *h, hipStream_t s, double alpha, int append, const void *x, void *y, const halo_dev &H);
}
namespace vexhip {
namespace {

struct rccl_api {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;          // optional (diagnostics)
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    std::string error;
};

rccl_api &rccl() {
    static rccl_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) { api.error = std::string("cannot load librccl: ") + dlerror(); return; }
#define SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name)); \
        if (!api.field) { api.error = std::string("librccl lacks ") + name; return; }
        SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommInitAll, "ncclCommInitAll")
        SYM(CommDestroy, "ncclCommDestroy") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
        SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(AllReduce, "ncclAllReduce") SYM(AllGather, "ncclAllGather")
        SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.lib, "ncclCommCount"));
        api.CommCuDevice = reinterpret_cast<decltype(api.CommCuDevice)>(dlsym(api.lib, "ncclCommCuDevice"));
        api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(dlsym(api.lib, "ncclCommUserRank"));
    });
    return api;
}

int nccl_fail(ncclResult_t r, const char *file, int line) {
    const rccl_api &a = rccl();
    return fail(file, line, std::string("RCCL: ") + (a.GetErrorString ? a.GetErrorString(r) : "error"));
}
#define NCCL_TRY(expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) return nccl_fail(_r, __FILE__, __LINE__); } while (0)
#define RCCL_READY() do { if (!rccl().error.empty()) return fail(__FILE__, __LINE__, rccl().error); } while (0)

size_t type_bytes(int dtype) { return (dtype == VEXHIP_F32 || dtype == VEXHIP_I32 || dtype == VEXHIP_U32) ? 4 : 8; }

struct comm {
    int world = 0;                 // ranks in the communicator
    std::vector<int> devs;         // local devices
    std::vector<int> ranks;        // rank of every local device
    std::vector<ncclComm_t> comms; // one per local device (RCCL transport)
    // PEER transport (single process only): the same exchange as direct device-to-device copies ordered by events --
    // what a process that sees every buffer can do without a communicator, and the only option when two logical
    // devices share one GPU (the reference's own test fixture, tests/context_setup.hpp:24-39; RCCL refuses that)
    bool peer = false;
    std::vector<hipEvent_t> ready, done;
    // One communicator is shared by every object that works on its device list (vexcl/exchange.hpp make_comm): calls from
    // different host threads are serialised, so that the record / wait pairs of one exchange (PEER) and the operations of
    // one RCCL group are never interleaved with another call's.
    std::mutex mx;
};

bool distinct(const std::vector<int> &v) {
    for (size_t a = 0; a < v.size(); ++a) for (size_t b = a + 1; b < v.size(); ++b) if (v[a] == v[b]) return false;
    return true;
}

// The exchange of vexhip_halo_exchange by copies: every device's send buffer is ready in ITS stream's order (as for
// ncclSend); the consumer's stream waits for that and pulls its share; the owner's stream then waits for the pulls,
// so that -- as after ncclSend -- work issued on it afterwards may overwrite the send buffer.
int peer_exchange(comm *c, int dtype, const void *const *send_bufs, const int64_t *send_counts,
        void *const *recv_bufs, const int64_t *recv_counts, void *const *streams)
{
    const int nd = (int)c->devs.size();
    const size_t b = type_bytes(dtype);
    for (int o = 0; o < nd; ++o) { VEXHIP_SET_DEVICE(c->devs[o]); VEXHIP_TRY(hipEventRecord(c->ready[o], as_stream(streams[o]))); }
    for (int d = 0; d < nd; ++d) {
        VEXHIP_SET_DEVICE(c->devs[d]);
        int64_t ro = 0;
        for (int o = 0; o < nd; ++o) {
            const int64_t n = recv_counts[d * nd + o];
            if (n) {
                VEXHIP_REQUIRE(send_counts[o * nd + d] == n, "halo exchange: send and receive counts differ");
                int64_t so = 0;
                for (int p = 0; p < d; ++p) so += send_counts[o * nd + p];
                if (o != d) VEXHIP_TRY(hipStreamWaitEvent(as_stream(streams[d]), c->ready[o], 0));
                const char *src = static_cast<const char *>(send_bufs[o]) + so * b;
                char *dst = static_cast<char *>(recv_bufs[d]) + ro * b;
                if (c->devs[o] == c->devs[d]) VEXHIP_TRY(hipMemcpyAsync(dst, src, (size_t)n * b, hipMemcpyDeviceToDevice, as_stream(streams[d])));
                else VEXHIP_TRY(hipMemcpyPeerAsync(dst, c->devs[d], src, c->devs[o], (size_t)n * b, as_stream(streams[d])));
            }
            ro += n;
        }
        VEXHIP_TRY(hipEventRecord(c->done[d], as_stream(streams[d])));
    }
    for (int o = 0; o < nd; ++o) {
        VEXHIP_SET_DEVICE(c->devs[o]);
        for (int d = 0; d < nd; ++d)
            if (d != o && send_counts[o * nd + d]) VEXHIP_TRY(hipStreamWaitEvent(as_stream(streams[o]), c->done[d], 0));
    }
    return 0;
}

int nccl_type(int dtype, ncclDataType_t *t) {
    switch (dtype) {
        case VEXHIP_F64: *t = ncclFloat64; return 0;
        case VEXHIP_F32: *t = ncclFloat32; return 0;
        case VEXHIP_I32: *t = ncclInt32; return 0;
        case VEXHIP_U32: *t = ncclUint32; return 0;
        case VEXHIP_I64: *t = ncclInt64; return 0;
        case VEXHIP_U64: *t = ncclUint64; return 0;
    }
    return fail(__FILE__, __LINE__, "unknown dtype");
}


// ---- IPC transport: peer-mapped ghost windows ------------------------------------------------------------------
// One window per rank, allocated UNCACHED (hipDeviceMallocUncached: neither this GPU's L2 nor its L1 keeps a line of
// it, so what a peer writes over xGMI -- which lands in HBM without probing this GPU's caches -- is what the next
// load returns) and exported with hipIpcGetMemHandle:
//     [ arrive[world] | consumed[world] | pad to 256 B | ghost values ]
// arrive[o]   (written by owner o):    number of the latest product whose share owner o has written into this window;
// consumed[p] (written by consumer p): number of the latest product for which p has finished reading what THIS rank
//                                      wrote into p's window -- this rank may overwrite it then.
// The flags are 64-bit step numbers that only grow: nothing is ever reset, a late reader can not miss an update.
// How long a flag wait may last before the product is declared failed: wall_clock64() counts at 100 MHz.  20 s by default
// (VEXHIP_IPC_TIMEOUT_MS): ranks of one job may be seconds apart (a JIT compile, I/O, an unbalanced phase between two
// products), and a wait that gives up corrupts the product -- so a timeout is an ERROR, not a fallback: the waiting kernel
// overwrites the ghost values with NaN (the remote part then produces NaN instead of numbers from stale ghosts), the error
// flag lives in pinned host memory, and vexhip_dist_spmv_apply / _profile fail from then on.
inline unsigned long long spin_ticks() {
    static const unsigned long long t = [] {
        const char *e = env(ENV_VEXHIP_IPC_TIMEOUT_MS);
        const long long ms = e ? std::atoll(e) : 20000;
        return (unsigned long long)(ms > 0 ? ms : 20000) * 100000ull;
    }();
    return t;
}

struct ipc_window {
    int dev = 0, rank = 0, world = 1;
    int64_t data_bytes = 0;
    size_t bytes = 0;
    bool uncached = true;                      // hipDeviceMallocUncached (the default): no cache ever holds a line of the window
    char *base = nullptr;
    std::vector<char *> peer;                  // base address of every opened window (own pointer for this rank)
    std::vector<char> opened;                  // mapped with hipIpcOpenMemHandle (to be closed)
    // products issued over this window so far, IN DEVICE MEMORY: the flags carry these numbers, so a second vexhip_dist_spmv on
    // the same windows continues the count (every rank issues the same products in the same order).  The first kernel of a step
    // increments it and the kernels of the step read it -- no kernel argument changes from one product to the next, which is
    // what lets a whole step be captured in a hipGraph and replayed (round 4).
    unsigned long long *d_step = nullptr;
};
inline size_t window_header(int world) { return ((size_t)world * 16 + 255) / 256 * 256; }
inline unsigned long long *window_arrive(char *base, int o) { return reinterpret_cast<unsigned long long *>(base) + o; }
inline unsigned long long *window_consumed(char *base, int world, int p) { return reinterpret_cast<unsigned long long *>(base) + world + p; }

struct push_peer {
    void *dst;                                 // where this rank's share starts in the destination's window
    unsigned long long *arrive;                // destination's arrive[me]
    const unsigned long long *consumed;        // my consumed[destination]
    long long first, count, direct_first, blk0;
};
// Elements one block of the push kernel writes (VEXHIP_IPC_PUSH_PER_BLOCK; a multiple of 256).  Stores into the uncached window
// are slow (about 75 GB/s from a full grid on the one-GPU box, where the "peer" is the GPU itself) and a wide push kernel slows
// the local part that runs beside it; at the 1/8 strip of the 512^3 problem (2 x 262 144 ghosts, tools/r04_dist_step.py) 2048 /
// 16384 / 65536 / 262144 elements per block give 110 / 95 / 202 / 647 us per step (local + remote part alone: 64 us).
inline int push_per_block() {
    static const int v = [] { const char *e = env(ENV_VEXHIP_IPC_PUSH_PER_BLOCK); int k = e ? std::atoi(e) : 16384; k = k < 256 ? 256 : k; return k / 256 * 256; }();
    return v;
}

// Every owner writes, per destination, exactly the values that destination needs straight into its window (idx: the
// packed order; NULL: the share is one run of x starting at direct_first), then raises arrive[me] there.
// first kernel of a step (compute stream): this product's number on the window, and this plan's launch count
__global__ void ipc_begin_kernel(unsigned long long *step, unsigned long long *launches) { ++*step; ++*launches; }

template <typename T>
__global__ __launch_bounds__(256)
void ipc_push_kernel(const push_peer *__restrict__ peers, int npeers, const unsigned long long *step_p, const int32_t *__restrict__ idx,
        const T *__restrict__ x, unsigned long long *done, const unsigned long long *launches_p, int *err, unsigned long long ticks, int per_block)
{
    const unsigned long long step = *step_p, launches = *launches_p;
    int j = 0;
    while (j + 1 < npeers && (long long)blockIdx.x >= peers[j + 1].blk0) ++j;
    const push_peer P = peers[j];
    __shared__ int s_ok;
    if (threadIdx.x == 0) s_ok = spin_until(P.consumed, step - 1, err, ticks) ? 1 : 0;     // the destination has read the previous product's share
    __syncthreads();
    if (!s_ok) return;                        // uniform: a destination that does not answer is not written to, and `arrive` is not raised
    T *dst = static_cast<T *>(P.dst);
    const long long i0 = ((long long)blockIdx.x - P.blk0) * per_block;
    // a share that is one run of x (plane partitions) and lies 16-byte aligned on both sides travels as 16-byte pieces: half
    // (fp64) or a quarter (fp32) of the store instructions into the uncached window
    constexpr int VN = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(16 / sizeof(T))));
    const T *src = x + P.direct_first;
    if (!idx && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0 && per_block % (256 * VN) == 0) {       // uniform
#pragma unroll 2
        for (int k = 0; k < per_block / (256 * VN); ++k) {
            const long long i = i0 + ((long long)k * 256 + threadIdx.x) * VN;
            if (i + VN <= P.count) *reinterpret_cast<vec_t *>(dst + i) = *reinterpret_cast<const vec_t *>(src + i);
            else for (long long q = i; q < P.count; ++q) dst[q] = src[q];
        }
    } else {
#pragma unroll 4
        for (int k = 0; k < per_block / 256; ++k) {
            const long long i = i0 + k * 256 + threadIdx.x;
            if (i < P.count) dst[i] = idx ? x[idx[P.first + i]] : x[P.direct_first + i];
        }
    }
    // The window is UNCACHED memory: its stores bypass the L2, and a wave's stores have been performed at the destination once
    // the wave's store counter is back at zero -- a workgroup-scope release (s_waitcnt vmcnt(0)) per lane, not the system-scope
    // fence every lane used to execute here: that one writes back and invalidates the WHOLE L2 65 000 times per product, which
    // doubled the time of the local part running beside it (profiles/r04_dist_step.json: 59 -> 129 us).  One system-scope
    // release remains: by the lane that raises the flag.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long nblk = (unsigned long long)((P.count + per_block - 1) / per_block);
        const unsigned long long old = __hip_atomic_fetch_add(&done[j], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        // `done` counts THIS plan's blocks and `launches` this plan's products: several plans may share a window (the flags carry the
        // window's step numbers), each sees its own launches only
        if (old + 1 == nblk * launches) {                                           // ... the last block of this destination raises the flag
            __threadfence_system();
            __hip_atomic_store(P.arrive, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// one wave: lane k waits for owner k's share; if any wait fails, every ghost value becomes NaN (all bits set) -- the remote part
// that follows must not turn stale ghosts into plausible numbers
__global__ __launch_bounds__(64)
void ipc_wait_kernel(const unsigned long long *const *flags, int n, const unsigned long long *step_p, int *err, unsigned long long ticks,
        unsigned long long *ghost_words, long long nwords) {
    const unsigned long long step = *step_p;
    bool ok = true;
    if ((int)threadIdx.x < n) ok = spin_until(flags[threadIdx.x], step, err, ticks);
    if (__builtin_amdgcn_ballot_w64(!ok))
        for (long long i = threadIdx.x; i < nwords; i += 64) ghost_words[i] = ~0ull;
}

__global__ __launch_bounds__(64)
void ipc_signal_kernel(unsigned long long *const *flags, int n, const unsigned long long *step_p) {
    const unsigned long long step = *step_p;
    if ((int)threadIdx.x < n) __hip_atomic_store(flags[threadIdx.x], step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- one rank's product step -----------------------------------------------------------------------------------
struct dist_spmv {
    comm *c = nullptr;
    int dev = 0, dtype = VEXHIP_F64;
    const vexhip_spmat *loc = nullptr;         // local part (may be NULL: no local entries)
    int64_t rows = 0;
    int64_t rem_rows = 0;                      // remote part: row-subset CSR (vexhip_spmv_csr_rows_*), may be 0
    const int32_t *rows_idx = nullptr, *rem_ptr = nullptr, *rem_col = nullptr; const void *rem_val = nullptr;
    int64_t nsend = 0, nghost = 0;
    const int32_t *send_idx = nullptr; void *send_buf = nullptr, *ghost_buf = nullptr;
    std::vector<int64_t> send_counts, recv_counts;       // per peer rank
    // every peer's share of x is ONE ascending run of consecutive elements (plane partitions: a boundary plane):
    // send_first[peer] = its first element, the sends read x itself and no pack kernel runs
    bool direct = false;
    std::vector<int64_t> send_first;
    hipStream_t comm_stream = nullptr;
    hipEvent_t packed = nullptr, received = nullptr;
    // optional replay of the whole step from a hipGraph (same x, y, alpha, append, stream as at capture)
    bool use_graph = false;
    hipGraphExec_t exec = nullptr;
    const void *gx = nullptr; void *gy = nullptr; double galpha = 0; int gappend = 0; hipStream_t gstream = nullptr;
    // IPC transport (peer-mapped ghost windows): the owners WRITE their shares into the consumer's window
    ipc_window *win = nullptr;
    unsigned long long *d_launches = nullptr;  // products issued through THIS plan, in device memory (the `done` counters count its blocks)
    push_peer *d_push = nullptr; int npush = 0; int64_t push_blocks = 0;
    unsigned long long *d_done = nullptr;      // per destination: blocks of the push kernel that have finished (monotonic)
    int *d_err = nullptr;                      // sticky, in pinned host memory mapped into the device: a flag was not raised in time
    const unsigned long long **d_arrive = nullptr; unsigned long long **d_consumed = nullptr; int nown = 0;
    hipEvent_t pushed = nullptr;
    // one-launch step (halo.hpp): the stored strip with its ghost planes, the kernel's view of the windows
    bool halo = false;
    const vexhip_spmat *ext = nullptr;
    halo_dev hd;
    unsigned *d_halo_done = nullptr;
    unsigned long long *d_halo_debug = nullptr;
    int pull = 0;                              // 0 pushed shares; 1 the neighbours' planes of x read in place behind flags; 2 ... ordered by the host's events
    bool has_lo = false, has_hi = false;
    unsigned long long *d_own_step = nullptr;  // pull == 2 without a window: the (unused) step word
    const void *glo = nullptr, *ghi = nullptr; // pull: the ghost planes of the captured step
    // optional phase timing of one step (vexhip_dist_spmv_profile)
    hipEvent_t *prof = nullptr;                // [0..3] compute stream: start, local done, ghosts here, end; [4..6] comm stream: start, packed, exchanged
};

int exchange_one(comm *c, int slot, int dtype, const void *send, const int64_t *scount, void *recv, const int64_t *rcount, hipStream_t s,
        const int64_t *send_first = nullptr) {
    rccl_api &a = rccl();
    ncclDataType_t t;
    if (int rc = nccl_type(dtype, &t)) return rc;
    const size_t b = type_bytes(dtype);
    const int me = c->ranks[slot];
    int64_t so = 0, ro = 0;
    for (int peer = 0; peer < c->world; ++peer) {
        const int64_t ns = scount[peer], nr = rcount[peer];
        static const bool self_over_rccl = env(ENV_VEXHIP_RCCL_SELF) != nullptr;      // tests: force ncclSend/ncclRecv to self
        // send_first: `send` is the vector itself and this peer's share starts at element send_first[peer] (no packed buffer)
        const char *src = static_cast<const char *>(send) + (send_first ? send_first[peer] : so) * (int64_t)b;
        if (peer == me && ns == nr && ns > 0 && !self_over_rccl) {
            // a rank's own share never crosses a link: device copy instead of a send/recv pair to itself
            VEXHIP_TRY(hipMemcpyAsync(static_cast<char *>(recv) + ro * b, src, (size_t)ns * b, hipMemcpyDeviceToDevice, s));
        } else {
            if (ns) NCCL_TRY(a.Send(src, (size_t)ns, t, peer, c->comms[slot], s));
            if (nr) NCCL_TRY(a.Recv(static_cast<char *>(recv) + ro * b, (size_t)nr, t, peer, c->comms[slot], s));
        }
        so += ns; ro += nr;
    }
    return 0;
}

#define PROF(k, stream) do { if (D->prof) VEXHIP_TRY(hipEventRecord(D->prof[k], stream)); } while (0)

int remote_part(dist_spmv *D, hipStream_t s, double alpha, void *y) {
    if (!D->rem_rows) return 0;
    return D->dtype == VEXHIP_F64
        ? vexhip_spmv_csr_rows_f64_i32(D->dev, s, D->rem_rows, alpha, D->rows_idx, D->rem_ptr, D->rem_col, static_cast<const double *>(D->rem_val),
                                       static_cast<const double *>(D->ghost_buf), static_cast<double *>(y))
        : vexhip_spmv_csr_rows_f32_i32(D->dev, s, D->rem_rows, (float)alpha, D->rows_idx, D->rem_ptr, D->rem_col, static_cast<const float *>(D->rem_val),
                                       static_cast<const float *>(D->ghost_buf), static_cast<float *>(y));
}

int local_part(dist_spmv *D, hipStream_t s, double alpha, int append, const void *x, void *y) {
    if (D->loc)
        return D->dtype == VEXHIP_F64 ? vexhip_spmat_apply_f64(D->loc, s, alpha, append, static_cast<const double *>(x), static_cast<double *>(y))
                                      : vexhip_spmat_apply_f32(D->loc, s, (float)alpha, append, static_cast<const float *>(x), static_cast<float *>(y));
    if (!append && D->rows) VEXHIP_TRY(hipMemsetAsync(y, 0, (size_t)D->rows * type_bytes(D->dtype), s));          // csr.inl:196-199
    return 0;
}

// The step over peer-mapped windows: no communicator, no pack buffer, no receive --
//   compute stream s:  [x ready] ...... local part ...... wait kernel (arrive flags) -> remote part -> signal kernel (consumed flags) -> wait(pushed)
//   comm stream:       wait(x ready) -> push kernel: per destination wait for consumed >= step-1, write the share into ITS window, raise arrive = step
int issue_step_ipc(dist_spmv *D, hipStream_t s, double alpha, int append, const void *x, void *y) {
    if (D->d_err && *static_cast<volatile int *>(D->d_err))
        return fail(__FILE__, __LINE__, "an earlier product of this plan timed out waiting for a peer's ghost flag (IPC transport, VEXHIP_IPC_TIMEOUT_MS): "
                                        "its result and every later one are invalid");
    const unsigned long long ticks = spin_ticks();
    const unsigned long long *step = D->win->d_step, *launches = D->d_launches;
    PROF(0, s);
    ipc_begin_kernel<<<1, 1, 0, s>>>(D->win->d_step, D->d_launches);
    VEXHIP_LAUNCH_CHECK();
    if (D->npush) {
        VEXHIP_TRY(hipEventRecord(D->packed, s));
        VEXHIP_TRY(hipStreamWaitEvent(D->comm_stream, D->packed, 0));
        PROF(4, D->comm_stream); PROF(5, D->comm_stream);
        const int32_t *idx = D->direct ? nullptr : D->send_idx;
        if (D->dtype == VEXHIP_F64)
            ipc_push_kernel<double><<<(unsigned)D->push_blocks, 256, 0, D->comm_stream>>>(D->d_push, D->npush, step, idx, static_cast<const double *>(x), D->d_done, launches, D->d_err, ticks, push_per_block());
        else
            ipc_push_kernel<float><<<(unsigned)D->push_blocks, 256, 0, D->comm_stream>>>(D->d_push, D->npush, step, idx, static_cast<const float *>(x), D->d_done, launches, D->d_err, ticks, push_per_block());
        VEXHIP_LAUNCH_CHECK();
        PROF(6, D->comm_stream);
        VEXHIP_TRY(hipEventRecord(D->pushed, D->comm_stream));
    } else if (D->prof) { PROF(4, s); PROF(5, s); PROF(6, s); }
    if (int rc = local_part(D, s, alpha, append, x, y)) return rc;
    PROF(1, s);
    if (D->nown) {
        ipc_wait_kernel<<<1, 64, 0, s>>>(D->d_arrive, D->nown, step, D->d_err, ticks, static_cast<unsigned long long *>(D->ghost_buf),
                                         (long long)((D->nghost * (int64_t)type_bytes(D->dtype) + 7) / 8));      // rounded UP: an odd number of fp32 ghosts ends half-way into a word (the window is padded)
        VEXHIP_LAUNCH_CHECK();
    }
    PROF(2, s);
    if (int rc = remote_part(D, s, alpha, y)) return rc;
    if (D->nown) {
        ipc_signal_kernel<<<1, 64, 0, s>>>(D->d_consumed, D->nown, step);
        VEXHIP_LAUNCH_CHECK();
    }
    if (D->npush) VEXHIP_TRY(hipStreamWaitEvent(s, D->pushed, 0));       // x may be overwritten by what the caller issues next
    PROF(3, s);
    return 0;
}

// The whole step as ONE product launch (+ the one-thread kernel that raises `consumed`): push workgroups, ghost planes read by the
// plane product itself (halo.hpp, plane.hip).  No second stream, no event, nothing to capture: two launches per product.
int issue_step_halo(dist_spmv *D, hipStream_t s, double alpha, int append, const void *x, void *y) {
    if (D->d_err && *static_cast<volatile int *>(D->d_err))
        return fail(__FILE__, __LINE__, "an earlier product of this plan timed out waiting for a peer's ghost flag (IPC transport, VEXHIP_IPC_TIMEOUT_MS): "
                                        "its result and every later one are invalid");
    PROF(0, s); PROF(4, s); PROF(5, s); PROF(6, s);
    if (int rc = spmat_apply_halo(D->ext, s, alpha, append, x, y, D->hd)) return rc;
    PROF(1, s); PROF(2, s); PROF(3, s);
    return 0;
}

int issue_step(dist_spmv *D, hipStream_t s, double alpha, int append, const void *x, void *y) {
    if (D->halo) return issue_step_halo(D, s, alpha, append, x, y);
    if (D->win) return issue_step_ipc(D, s, alpha, append, x, y);
    const bool f64 = D->dtype == VEXHIP_F64;
    const bool exch = D->nsend > 0 || D->nghost > 0;
    PROF(0, s);
    if (exch) {
        // The pack kernel and the exchange run on the SECOND stream, beside the local part (which needs neither):
        //   compute stream s:   [x ready] ............ local part ............ wait(received) remote part [consumed]
        //   comm stream:        wait(x ready, consumed of the previous step)  pack -> send/recv  [received]
        // Ordering: the pack re-uses send_buf after the previous step's sends (same stream, in order); the receive
        // overwrites ghost_buf only after the previous step's remote part has read it (`consumed`, recorded on s).
        VEXHIP_TRY(hipEventRecord(D->packed, s));                         // "x is ready" (and, transitively, `consumed`)
        VEXHIP_TRY(hipStreamWaitEvent(D->comm_stream, D->packed, 0));
        PROF(4, D->comm_stream);
        if (D->nsend && !D->direct) {
            int rc = f64 ? vexhip_gather_f64_i32(D->dev, D->comm_stream, D->nsend, D->send_idx, static_cast<const double *>(x), static_cast<double *>(D->send_buf))
                         : vexhip_gather_f32_i32(D->dev, D->comm_stream, D->nsend, D->send_idx, static_cast<const float *>(x), static_cast<float *>(D->send_buf));
            if (rc) return rc;
        }
        PROF(5, D->comm_stream);
        NCCL_TRY(rccl().GroupStart());
        // (direct: the sends read x; x stays untouched until `received`, which follows sends and receives alike)
        int rc = D->direct ? exchange_one(D->c, 0, D->dtype, x, D->send_counts.data(), D->ghost_buf, D->recv_counts.data(), D->comm_stream, D->send_first.data())
                           : exchange_one(D->c, 0, D->dtype, D->send_buf, D->send_counts.data(), D->ghost_buf, D->recv_counts.data(), D->comm_stream);
        ncclResult_t ge = rccl().GroupEnd();
        if (rc) return rc;
        NCCL_TRY(ge);
        PROF(6, D->comm_stream);
        VEXHIP_TRY(hipEventRecord(D->received, D->comm_stream));
    } else if (D->prof) { PROF(4, s); PROF(5, s); PROF(6, s); }
    // local part, overlapped with pack + exchange
    if (int rc = local_part(D, s, alpha, append, x, y)) return rc;
    PROF(1, s);
    if (exch) VEXHIP_TRY(hipStreamWaitEvent(s, D->received, 0));
    PROF(2, s);
    if (int rc = remote_part(D, s, alpha, y)) return rc;
    PROF(3, s);
    return 0;
}
#undef PROF

// are the shares single runs of consecutive elements?  (one read of the index list at set-up)
hipError_t detect_runs(dist_spmv *D, const int32_t *send_idx, int world) {
    std::vector<int32_t> h((size_t)D->nsend);
    hipError_t e = hipMemcpy(h.data(), send_idx, sizeof(int32_t) * (size_t)D->nsend, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return e;
    D->send_first.assign(world, 0);
    bool runs = true;
    int64_t o = 0;
    for (int p = 0; p < world && runs; ++p) {
        const int64_t k = D->send_counts[p];
        if (k) {
            D->send_first[p] = h[(size_t)o];
            for (int64_t j = 1; j < k && runs; ++j) runs = h[(size_t)(o + j)] == h[(size_t)o] + (int32_t)j;
            runs = runs && h[(size_t)o] >= 0 && (int64_t)h[(size_t)o] + k <= D->rows;
        }
        o += k;
    }
    D->direct = runs;
    return hipSuccess;
}

template <typename T>
void fold(int op, std::vector<char> &host, size_t nd, size_t count) {
    T *v = reinterpret_cast<T *>(host.data());
    for (size_t k = 0; k < count; ++k) {
        T r = v[k];
        for (size_t d = 1; d < nd; ++d) {
            const T o = v[d * count + k];
            r = (op == VEXHIP_MIN) ? (o < r ? o : r) : (op == VEXHIP_MAX) ? (o > r ? o : r) : (T)(r + o);
        }
        v[k] = r;
    }
}

} // namespace
} // namespace vexhip

using namespace vexhip;

extern "C" {

int vexhip_comm_unique_id(void *id128) {
    VEXHIP_REQUIRE(id128, "NULL argument");
    RCCL_READY();
    ncclUniqueId id;
    NCCL_TRY(rccl().GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return 0;
}

int vexhip_comm_init(int ndev, const int *devs, int transport, vexhip_comm **out) {
    reload_env();
    VEXHIP_REQUIRE(out && ndev >= 1 && devs, "bad argument");
    VEXHIP_REQUIRE(transport >= VEXHIP_COMM_AUTO && transport <= VEXHIP_COMM_PEER, "unknown transport");
    *out = nullptr;
    comm *c = new (std::nothrow) comm;
    VEXHIP_REQUIRE(c, "out of host memory");
    c->world = ndev;
    c->devs.assign(devs, devs + ndev);
    c->ranks.resize(ndev);
    for (int d = 0; d < ndev; ++d) c->ranks[d] = d;
    // RCCL needs one GPU per rank; one device, or logical devices that share a GPU, exchange by copies
    bool want_rccl = transport == VEXHIP_COMM_RCCL || (transport == VEXHIP_COMM_AUTO && ndev > 1 && distinct(c->devs));
    if (transport == VEXHIP_COMM_AUTO && env(ENV_VEXHIP_COMM_PEER)) want_rccl = false;
    if (want_rccl) {
        if (!rccl().error.empty()) {
            if (transport == VEXHIP_COMM_RCCL) { delete c; return fail(__FILE__, __LINE__, rccl().error); }
            want_rccl = false;
        } else {
            c->comms.assign(ndev, nullptr);
            ncclResult_t r = rccl().CommInitAll(c->comms.data(), ndev, devs);
            if (r != ncclSuccess) {
                if (transport == VEXHIP_COMM_RCCL) { delete c; return nccl_fail(r, __FILE__, __LINE__); }
                c->comms.clear(); want_rccl = false;
            }
        }
    }
    if (!want_rccl) {
        c->peer = true;
        c->ready.assign(ndev, nullptr); c->done.assign(ndev, nullptr);
        for (int d = 0; d < ndev; ++d) {
            hipError_t e = hipSetDevice(devs[d]);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready[d], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done[d], hipEventDisableTiming);
            if (e != hipSuccess) { vexhip_comm_destroy(reinterpret_cast<vexhip_comm *>(c)); return check(e, __FILE__, __LINE__); }
        }
    }
    *out = reinterpret_cast<vexhip_comm *>(c);
    return 0;
}

int vexhip_comm_init_rank(int dev, int rank, int world, const void *id128, vexhip_comm **out) {
    reload_env();
    VEXHIP_REQUIRE(out && id128 && world >= 1 && rank >= 0 && rank < world, "bad argument");
    *out = nullptr;
    RCCL_READY();
    VEXHIP_SET_DEVICE(dev);
    comm *c = new (std::nothrow) comm;
    VEXHIP_REQUIRE(c, "out of host memory");
    c->world = world; c->devs.assign(1, dev); c->ranks.assign(1, rank); c->comms.assign(1, nullptr);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclResult_t r = rccl().CommInitRank(&c->comms[0], world, id, rank);
    if (r != ncclSuccess) { delete c; return nccl_fail(r, __FILE__, __LINE__); }
    *out = reinterpret_cast<vexhip_comm *>(c);
    return 0;
}

int vexhip_comm_destroy(vexhip_comm *h) {
    comm *c = reinterpret_cast<comm *>(h);
    if (!c) return 0;
    for (size_t d = 0; d < c->comms.size(); ++d)
        if (c->comms[d]) { (void)hipSetDevice(c->devs[d]); (void)rccl().CommDestroy(c->comms[d]); }
    for (size_t d = 0; d < c->ready.size(); ++d) {
        (void)hipSetDevice(c->devs[d]);
        if (c->ready[d]) (void)hipEventDestroy(c->ready[d]);
        if (c->done[d]) (void)hipEventDestroy(c->done[d]);
    }
    delete c;
    return 0;
}

int vexhip_comm_size(const vexhip_comm *h, int *world, int *nlocal, int *transport) {
    const comm *c = reinterpret_cast<const comm *>(h);
    VEXHIP_REQUIRE(c, "NULL communicator");
    if (world) *world = c->world;
    if (nlocal) *nlocal = (int)c->devs.size();
    if (transport) *transport = c->peer ? VEXHIP_COMM_PEER : VEXHIP_COMM_RCCL;
    return 0;
}

int vexhip_halo_exchange(vexhip_comm *h, int dtype, const void *const *send_bufs, const int64_t *send_counts,
        void *const *recv_bufs, const int64_t *recv_counts, void *const *streams)
{
    comm *c = reinterpret_cast<comm *>(h);
    VEXHIP_REQUIRE(c && send_bufs && send_counts && recv_bufs && recv_counts && streams, "NULL argument");
    std::lock_guard<std::mutex> serialise(c->mx);
    if (c->peer) return peer_exchange(c, dtype, send_bufs, send_counts, recv_bufs, recv_counts, streams);
    RCCL_READY();
    NCCL_TRY(rccl().GroupStart());
    int rc = 0;
    for (size_t d = 0; d < c->devs.size() && !rc; ++d) {
        if (hipSetDevice(c->devs[d]) != hipSuccess) { rc = fail(__FILE__, __LINE__, "hipSetDevice failed"); break; }
        rc = exchange_one(c, (int)d, dtype, send_bufs[d], send_counts + d * c->world, recv_bufs[d], recv_counts + d * c->world, as_stream(streams[d]));
    }
    ncclResult_t ge = rccl().GroupEnd();
    if (rc) return rc;
    NCCL_TRY(ge);
    return 0;
}

int vexhip_allreduce_scalar(vexhip_comm *h, int op, int dtype, void *const *bufs, int64_t count, void *const *streams) {
    comm *c = reinterpret_cast<comm *>(h);
    VEXHIP_REQUIRE(c && bufs && streams && count >= 1, "bad argument");
    std::lock_guard<std::mutex> serialise(c->mx);
    VEXHIP_REQUIRE(op == VEXHIP_SUM || op == VEXHIP_SUM_KAHAN || op == VEXHIP_MIN || op == VEXHIP_MAX, "unsupported reduction for the all-reduce");
    if (c->peer) {
        // one process sees every device: read the D partial results, fold on the host in device order (what the reference
        // does with its 8 x CU partials per device, reductor.hpp:420-436), write the result back
        const size_t nd = c->devs.size(), b = type_bytes(dtype);
        std::vector<char> host(nd * (size_t)count * b);
        for (size_t d = 0; d < nd; ++d) {
            VEXHIP_SET_DEVICE(c->devs[d]);
            VEXHIP_TRY(hipMemcpyAsync(host.data() + d * count * b, bufs[d], (size_t)count * b, hipMemcpyDeviceToHost, as_stream(streams[d])));
        }
        for (size_t d = 0; d < nd; ++d) { VEXHIP_SET_DEVICE(c->devs[d]); VEXHIP_TRY(hipStreamSynchronize(as_stream(streams[d]))); }
        switch (dtype) {
            case VEXHIP_F64: fold<double>(op, host, nd, (size_t)count); break;
            case VEXHIP_F32: fold<float>(op, host, nd, (size_t)count); break;
            case VEXHIP_I32: fold<int32_t>(op, host, nd, (size_t)count); break;
            case VEXHIP_U32: fold<uint32_t>(op, host, nd, (size_t)count); break;
            case VEXHIP_I64: fold<int64_t>(op, host, nd, (size_t)count); break;
            case VEXHIP_U64: fold<uint64_t>(op, host, nd, (size_t)count); break;
            default: return fail(__FILE__, __LINE__, "unknown dtype");
        }
        for (size_t d = 0; d < nd; ++d) {
            VEXHIP_SET_DEVICE(c->devs[d]);
            VEXHIP_TRY(hipMemcpyAsync(bufs[d], host.data(), (size_t)count * b, hipMemcpyHostToDevice, as_stream(streams[d])));
            VEXHIP_TRY(hipStreamSynchronize(as_stream(streams[d])));
        }
        return 0;
    }
    RCCL_READY();
    ncclDataType_t t;
    if (int rc = nccl_type(dtype, &t)) return rc;
    const ncclRedOp_t ro = op == VEXHIP_MIN ? ncclMin : op == VEXHIP_MAX ? ncclMax : ncclSum;
    NCCL_TRY(rccl().GroupStart());
    ncclResult_t r = ncclSuccess;
    for (size_t d = 0; d < c->devs.size() && r == ncclSuccess; ++d) {
        (void)hipSetDevice(c->devs[d]);
        r = rccl().AllReduce(bufs[d], bufs[d], (size_t)count, t, ro, c->comms[d], as_stream(streams[d]));
    }
    ncclResult_t ge = rccl().GroupEnd();
    NCCL_TRY(r);
    NCCL_TRY(ge);
    return 0;
}

int vexhip_allgather(vexhip_comm *h, int dtype, const void *const *send, void *const *recv, int64_t count, void *const *streams) {
    comm *c = reinterpret_cast<comm *>(h);
    VEXHIP_REQUIRE(c && send && recv && streams && count >= 0, "bad argument");
    std::lock_guard<std::mutex> serialise(c->mx);
    if (count == 0) return 0;
    if (c->peer) {
        const size_t nd = c->devs.size(), b = type_bytes(dtype);
        for (size_t o = 0; o < nd; ++o) { VEXHIP_SET_DEVICE(c->devs[o]); VEXHIP_TRY(hipEventRecord(c->ready[o], as_stream(streams[o]))); }
        for (size_t d = 0; d < nd; ++d) {
            VEXHIP_SET_DEVICE(c->devs[d]);
            for (size_t o = 0; o < nd; ++o) {
                if (o != d) VEXHIP_TRY(hipStreamWaitEvent(as_stream(streams[d]), c->ready[o], 0));
                char *dst = static_cast<char *>(recv[d]) + o * (size_t)count * b;
                if (c->devs[o] == c->devs[d]) VEXHIP_TRY(hipMemcpyAsync(dst, send[o], (size_t)count * b, hipMemcpyDeviceToDevice, as_stream(streams[d])));
                else VEXHIP_TRY(hipMemcpyPeerAsync(dst, c->devs[d], send[o], c->devs[o], (size_t)count * b, as_stream(streams[d])));
            }
            VEXHIP_TRY(hipEventRecord(c->done[d], as_stream(streams[d])));
        }
        for (size_t o = 0; o < nd; ++o) {
            VEXHIP_SET_DEVICE(c->devs[o]);
            for (size_t d = 0; d < nd; ++d) if (d != o) VEXHIP_TRY(hipStreamWaitEvent(as_stream(streams[o]), c->done[d], 0));
        }
        return 0;
    }
    RCCL_READY();
    ncclDataType_t t;
    if (int rc = nccl_type(dtype, &t)) return rc;
    NCCL_TRY(rccl().GroupStart());
    ncclResult_t r = ncclSuccess;
    for (size_t d = 0; d < c->devs.size() && r == ncclSuccess; ++d) {
        (void)hipSetDevice(c->devs[d]);
        r = rccl().AllGather(send[d], recv[d], (size_t)count, t, c->comms[d], as_stream(streams[d]));
    }
    ncclResult_t ge = rccl().GroupEnd();
    NCCL_TRY(r);
    NCCL_TRY(ge);
    return 0;
}

// ---- one rank's product step ---------------------------------------------------------------------------------------
int vexhip_dist_spmv_create(vexhip_comm *hc, int dtype, int64_t rows, const vexhip_spmat *local,
        int64_t rem_rows, const int32_t *rows_idx, const int32_t *rem_ptr, const int32_t *rem_col, const void *rem_val,
        int64_t nsend, const int32_t *send_idx, void *send_buf, const int64_t *send_counts,
        int64_t nghost, void *ghost_buf, const int64_t *recv_counts, vexhip_dist_spmv **out)
{
    reload_env();
    comm *c = reinterpret_cast<comm *>(hc);
    VEXHIP_REQUIRE(out, "NULL output");
    *out = nullptr;
    VEXHIP_REQUIRE(c && c->devs.size() == 1 && !c->peer, "vexhip_dist_spmv needs a one-device RCCL communicator (vexhip_comm_init_rank)");
    VEXHIP_REQUIRE(dtype == VEXHIP_F64 || dtype == VEXHIP_F32, "value type must be f64 or f32");
    VEXHIP_REQUIRE(rows >= 0 && rem_rows >= 0 && nsend >= 0 && nghost >= 0, "negative size");
    VEXHIP_REQUIRE((!nsend || (send_idx && send_buf && send_counts)) && (!nghost || (ghost_buf && recv_counts)), "NULL exchange buffers");
    VEXHIP_REQUIRE(!rem_rows || (rows_idx && rem_ptr && rem_col && rem_val && ghost_buf), "NULL remote part");
    dist_spmv *D = new (std::nothrow) dist_spmv;
    VEXHIP_REQUIRE(D, "out of host memory");
    D->c = c; D->dev = c->devs[0]; D->dtype = dtype; D->loc = local; D->rows = rows;
    D->rem_rows = rem_rows; D->rows_idx = rows_idx; D->rem_ptr = rem_ptr; D->rem_col = rem_col; D->rem_val = rem_val;
    D->nsend = nsend; D->send_idx = send_idx; D->send_buf = send_buf; D->nghost = nghost; D->ghost_buf = ghost_buf;
    D->send_counts.assign(c->world, 0); D->recv_counts.assign(c->world, 0);
    int64_t ts = 0, tr = 0;
    for (int p = 0; p < c->world; ++p) {
        if (nsend) D->send_counts[p] = send_counts[p];
        if (nghost) D->recv_counts[p] = recv_counts[p];
        ts += D->send_counts[p]; tr += D->recv_counts[p];
    }
    if (ts != nsend || tr != nghost) { delete D; return fail(__FILE__, __LINE__, "exchange counts do not add up to the buffer sizes"); }
    hipError_t e = hipSetDevice(D->dev);
    if (e == hipSuccess) {
        // highest priority: the pack kernel and the RCCL kernels are small and must not queue behind the 32 768
        // workgroups of the local part they are meant to overlap with
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
        e = hipStreamCreateWithPriority(&D->comm_stream, hipStreamNonBlocking, hi);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&D->packed, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&D->received, hipEventDisableTiming);
    if (e == hipSuccess && nsend > 0 && !env(ENV_VEXHIP_DIST_PACK)) e = detect_runs(D, send_idx, c->world);
    if (e != hipSuccess) { vexhip_dist_spmv_destroy(reinterpret_cast<vexhip_dist_spmv *>(D)); return check(e, __FILE__, __LINE__); }
    *out = reinterpret_cast<vexhip_dist_spmv *>(D);
    return 0;
}

int vexhip_dist_spmv_destroy(vexhip_dist_spmv *h) {
    dist_spmv *D = reinterpret_cast<dist_spmv *>(h);
    if (!D) return 0;
    (void)hipSetDevice(D->dev);
    if (D->exec) (void)hipGraphExecDestroy(D->exec);
    if (D->comm_stream) { (void)hipStreamSynchronize(D->comm_stream); (void)hipStreamDestroy(D->comm_stream); }
    if (D->packed) (void)hipEventDestroy(D->packed);
    if (D->received) (void)hipEventDestroy(D->received);
    if (D->pushed) (void)hipEventDestroy(D->pushed);
    if (D->prof) { for (int k = 0; k < 7; ++k) if (D->prof[k]) (void)hipEventDestroy(D->prof[k]); delete[] D->prof; }
    if (D->d_push) (void)hipFree(D->d_push);
    if (D->d_done) (void)hipFree(D->d_done);
    if (D->d_launches) (void)hipFree(D->d_launches);
    if (D->d_err) (void)hipHostFree(D->d_err);
    if (D->d_arrive) (void)hipFree(D->d_arrive);
    if (D->d_consumed) (void)hipFree(D->d_consumed);
    if (D->d_halo_done) (void)hipFree(D->d_halo_done);
    if (D->d_halo_debug) (void)hipFree(D->d_halo_debug);
    if (D->d_own_step) (void)hipFree(D->d_own_step);
    delete D;
    return 0;
}

int vexhip_dist_spmv_set_graph(vexhip_dist_spmv *h, int enable) {
    dist_spmv *D = reinterpret_cast<dist_spmv *>(h);
    VEXHIP_REQUIRE(D, "NULL argument");
    D->use_graph = enable != 0;
    if (!enable && D->exec) { (void)hipGraphExecDestroy(D->exec); D->exec = nullptr; }
    return 0;
}

static int dist_spmv_apply_impl(dist_spmv *D, void *stream, double alpha, int append, const void *x, void *y);

int vexhip_dist_spmv_apply(vexhip_dist_spmv *h, void *stream, double alpha, int append, const void *x, void *y) {
    dist_spmv *D = reinterpret_cast<dist_spmv *>(h);
    VEXHIP_REQUIRE(D && (x || !D->rows) && (y || !D->rows), "NULL argument");
    VEXHIP_REQUIRE(!D->pull, "a pull plan takes the neighbours' planes with every product: vexhip_dist_spmv_apply_pull");
    return dist_spmv_apply_impl(D, stream, alpha, append, x, y);
}

// PULL (round 6): x_below / x_above = the lower neighbour's LAST plane and the upper neighbour's FIRST plane of ITS segment of the
// same vector (`halo` elements each, readable from this device: peer access), NULL where the plan has no neighbour.
int vexhip_dist_spmv_apply_pull(vexhip_dist_spmv *h, void *stream, double alpha, int append, const void *x, void *y,
        const void *x_below, const void *x_above) {
    dist_spmv *D = reinterpret_cast<dist_spmv *>(h);
    VEXHIP_REQUIRE(D && x && y, "NULL argument");
    VEXHIP_REQUIRE(D->halo && D->pull, "not a pull plan (vexhip_dist_spmv_create_halo_pull)");
    VEXHIP_REQUIRE((x_below != nullptr) == D->has_lo && (x_above != nullptr) == D->has_hi, "the neighbours' planes do not match the plan's neighbours");
    VEXHIP_REQUIRE(((reinterpret_cast<uintptr_t>(x_below) | reinterpret_cast<uintptr_t>(x_above)) & 15) == 0, "the neighbours' planes must be 16-byte aligned");
    if (D->exec && (D->glo != x_below || D->ghi != x_above)) { (void)hipGraphExecDestroy(D->exec); D->exec = nullptr; }
    D->glo = x_below; D->ghi = x_above;
    D->hd.lo = static_cast<const double *>(x_below); D->hd.hi = static_cast<const double *>(x_above);
    return dist_spmv_apply_impl(D, stream, alpha, append, x, y);
}

static int dist_spmv_apply_impl(dist_spmv *D, void *stream, double alpha, int append, const void *x, void *y) {
    if (!D->win && !D->halo && (D->nsend || D->nghost)) RCCL_READY();
    VEXHIP_SET_DEVICE(D->dev);
    hipStream_t s = as_stream(stream);
    // Steps with an RCCL exchange are always issued directly: capturing ncclSend / ncclRecv into a hipGraph crashes in this RCCL
    // (2.26.6, measured with tools/r02_dist_step.py).  The IPC step is kernels and events only -- its step numbers live in
    // device memory -- and replays like a step without an exchange (tools/r04_dist_step.py: host time per product).
    if (!D->use_graph || (!D->win && !D->halo && (D->nsend || D->nghost))) return issue_step(D, s, alpha, append, x, y);
    if (D->d_err && *static_cast<volatile int *>(D->d_err))
        return fail(__FILE__, __LINE__, "an earlier product of this plan timed out waiting for a peer's ghost flag (IPC transport, VEXHIP_IPC_TIMEOUT_MS): "
                                        "its result and every later one are invalid");
    // replay: the captured step is valid for exactly these operands
    if (D->exec && (D->gx != x || D->gy != y || D->galpha != alpha || D->gappend != append || D->gstream != s)) {
        (void)hipGraphExecDestroy(D->exec); D->exec = nullptr;
    }
    if (!D->exec) {
        VEXHIP_REQUIRE(s != nullptr, "graph replay needs an explicit (non-default) stream");
        hipGraph_t g = nullptr;
        VEXHIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int rc = issue_step(D, s, alpha, append, x, y);
        hipError_t e = hipStreamEndCapture(s, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        VEXHIP_TRY(e);
        e = hipGraphInstantiate(&D->exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        VEXHIP_TRY(e);
        D->gx = x; D->gy = y; D->galpha = alpha; D->gappend = append; D->gstream = s;
    }
    VEXHIP_TRY(hipGraphLaunch(D->exec, s));
    return 0;
}

// ---- IPC windows + the product step over them --------------------------------------------------------------------
int vexhip_ipc_window_create(int dev, int rank, int world, int64_t data_bytes, vexhip_ipc_window **out) {
    reload_env();
    VEXHIP_REQUIRE(out && world >= 1 && rank >= 0 && rank < world && data_bytes >= 0, "bad argument");
    *out = nullptr;
    VEXHIP_SET_DEVICE(dev);
    ipc_window *w = new (std::nothrow) ipc_window;
    VEXHIP_REQUIRE(w, "out of host memory");
    w->dev = dev; w->rank = rank; w->world = world; w->data_bytes = data_bytes;
    w->bytes = window_header(world) + (((size_t)data_bytes + 255) / 256 * 256) + 256;
    w->peer.assign(world, nullptr); w->opened.assign(world, 0);
    void *p = nullptr;
    // (VEXHIP_IPC_WINDOW_MEM=finegrained | default: diagnostics on ONE device only -- tools/r05_dist_step.py measures what the
    //  uncached mapping costs the reader; between two devices only the uncached window is known to show a peer's writes)
    unsigned kind = hipDeviceMallocUncached;
    if (const char *m = env(ENV_VEXHIP_IPC_WINDOW_MEM)) {
        if (std::string(m) == "finegrained") kind = hipDeviceMallocFinegrained;
        else if (std::string(m) == "default") kind = hipDeviceMallocDefault;
    }
    w->uncached = kind == hipDeviceMallocUncached;
    hipError_t e = hipExtMallocWithFlags(&p, w->bytes, kind);
    if (e == hipSuccess) e = hipMemset(p, 0, w->bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { if (p) (void)hipFree(p); delete w; return check(e, __FILE__, __LINE__); }
    w->base = static_cast<char *>(p);
    w->peer[rank] = w->base;
    e = hipMalloc(reinterpret_cast<void **>(&w->d_step), sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(w->d_step, 0, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(p); if (w->d_step) (void)hipFree(w->d_step); delete w; return check(e, __FILE__, __LINE__); }
    *out = reinterpret_cast<vexhip_ipc_window *>(w);
    return 0;
}

int vexhip_ipc_window_export(const vexhip_ipc_window *h, void *handle64) {
    const ipc_window *w = reinterpret_cast<const ipc_window *>(h);
    VEXHIP_REQUIRE(w && handle64, "NULL argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    VEXHIP_SET_DEVICE(w->dev);
    hipIpcMemHandle_t mh;
    VEXHIP_TRY(hipIpcGetMemHandle(&mh, w->base));
    std::memcpy(handle64, &mh, sizeof(mh));
    return 0;
}

int vexhip_ipc_window_open(vexhip_ipc_window *h, int peer, const void *handle64) {
    ipc_window *w = reinterpret_cast<ipc_window *>(h);
    VEXHIP_REQUIRE(w && handle64 && peer >= 0 && peer < w->world, "bad argument");
    if (w->peer[peer]) return 0;
    VEXHIP_SET_DEVICE(w->dev);
    hipIpcMemHandle_t mh;
    std::memcpy(&mh, handle64, sizeof(mh));
    void *p = nullptr;
    VEXHIP_TRY(hipIpcOpenMemHandle(&p, mh, hipIpcMemLazyEnablePeerAccess));
    w->peer[peer] = static_cast<char *>(p); w->opened[peer] = 1;
    return 0;
}

// One process driving several devices (vex::Context): the peer's window is an address this process already has -- no handle to
// export or open; distinct GPUs get peer access in both directions (which covers every allocation of the peer, the vectors the
// pull step reads in place included).
int vexhip_ipc_window_attach(vexhip_ipc_window *h, int peer, const vexhip_ipc_window *hp) {
    ipc_window *w = reinterpret_cast<ipc_window *>(h);
    const ipc_window *pw = reinterpret_cast<const ipc_window *>(hp);
    VEXHIP_REQUIRE(w && pw && peer >= 0 && peer < w->world && pw->rank == peer && pw->world == w->world, "bad argument");
    if (w->dev != pw->dev) {
        int can = 0;
        VEXHIP_TRY(hipDeviceCanAccessPeer(&can, w->dev, pw->dev));
        VEXHIP_REQUIRE(can, "the devices cannot access each other's memory (no peer access)");
        VEXHIP_SET_DEVICE(w->dev);
        hipError_t e = hipDeviceEnablePeerAccess(pw->dev, 0);
        if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
        VEXHIP_TRY(e);
    }
    w->peer[peer] = pw->base; w->opened[peer] = 0;
    return 0;
}

// Any device allocation made visible to another process (the pull step between processes reads the neighbours' x in place): the handle of
// the ALLOCATION that holds ptr and ptr's offset in it (what torch does for tensors shared between processes).
int vexhip_ipc_export(int dev, const void *ptr, void *handle64, int64_t *offset) {
    VEXHIP_REQUIRE(ptr && handle64 && offset, "NULL argument");
    VEXHIP_SET_DEVICE(dev);
    hipDeviceptr_t base = nullptr; size_t size = 0;
    VEXHIP_TRY(hipMemGetAddressRange(&base, &size, const_cast<void *>(ptr)));
    hipIpcMemHandle_t mh;
    VEXHIP_TRY(hipIpcGetMemHandle(&mh, base));
    std::memcpy(handle64, &mh, sizeof(mh));
    *offset = (int64_t)(static_cast<const char *>(ptr) - static_cast<const char *>(base));
    return 0;
}
int vexhip_ipc_open(int dev, const void *handle64, void **base) {
    VEXHIP_REQUIRE(handle64 && base, "NULL argument");
    *base = nullptr;
    VEXHIP_SET_DEVICE(dev);
    hipIpcMemHandle_t mh;
    std::memcpy(&mh, handle64, sizeof(mh));
    VEXHIP_TRY(hipIpcOpenMemHandle(base, mh, hipIpcMemLazyEnablePeerAccess));
    return 0;
}
int vexhip_ipc_close(int dev, void *base) {
    if (!base) return 0;
    VEXHIP_SET_DEVICE(dev);
    VEXHIP_TRY(hipIpcCloseMemHandle(base));
    return 0;
}

int vexhip_ipc_window_data(const vexhip_ipc_window *h, void **data) {
    const ipc_window *w = reinterpret_cast<const ipc_window *>(h);
    VEXHIP_REQUIRE(w && data, "NULL argument");
    *data = w->base + window_header(w->world);
    return 0;
}

int vexhip_ipc_window_destroy(vexhip_ipc_window *h) {
    ipc_window *w = reinterpret_cast<ipc_window *>(h);
    if (!w) return 0;
    (void)hipSetDevice(w->dev);
    (void)hipDeviceSynchronize();
    for (int p = 0; p < w->world; ++p) if (w->opened[p] && w->peer[p]) (void)hipIpcCloseMemHandle(w->peer[p]);
    if (w->base) (void)hipFree(w->base);
    if (w->d_step) (void)hipFree(w->d_step);
    delete w;
    return 0;
}

int vexhip_dist_spmv_create_ipc(vexhip_ipc_window *hw, int dtype, int64_t rows, const vexhip_spmat *local,
        int64_t rem_rows, const int32_t *rows_idx, const int32_t *rem_ptr, const int32_t *rem_col, const void *rem_val,
        int64_t nsend, const int32_t *send_idx, const int64_t *send_counts, const int64_t *dst_offsets,
        int64_t nghost, const int64_t *recv_counts, vexhip_dist_spmv **out)
{
    reload_env();
    ipc_window *w = reinterpret_cast<ipc_window *>(hw);
    VEXHIP_REQUIRE(out, "NULL output");
    *out = nullptr;
    VEXHIP_REQUIRE(w, "NULL window");
    VEXHIP_REQUIRE(dtype == VEXHIP_F64 || dtype == VEXHIP_F32, "value type must be f64 or f32");
    VEXHIP_REQUIRE(rows >= 0 && rem_rows >= 0 && nsend >= 0 && nghost >= 0, "negative size");
    VEXHIP_REQUIRE((!nsend || (send_idx && send_counts && dst_offsets)) && (!nghost || recv_counts), "NULL exchange plan");
    VEXHIP_REQUIRE(!rem_rows || (rows_idx && rem_ptr && rem_col && rem_val), "NULL remote part");
    VEXHIP_REQUIRE((int64_t)type_bytes(dtype) * nghost <= w->data_bytes, "the window is smaller than the ghost set");
    VEXHIP_REQUIRE(w->world <= 64, "more than 64 ranks");
    dist_spmv *D = new (std::nothrow) dist_spmv;
    VEXHIP_REQUIRE(D, "out of host memory");
    D->win = w; D->dev = w->dev; D->dtype = dtype; D->loc = local; D->rows = rows;
    D->rem_rows = rem_rows; D->rows_idx = rows_idx; D->rem_ptr = rem_ptr; D->rem_col = rem_col; D->rem_val = rem_val;
    D->nsend = nsend; D->send_idx = send_idx; D->nghost = nghost;
    D->ghost_buf = w->base + window_header(w->world);
    D->send_counts.assign(w->world, 0); D->recv_counts.assign(w->world, 0);
    int64_t ts = 0, tr = 0;
    for (int p = 0; p < w->world; ++p) {
        if (nsend) D->send_counts[p] = send_counts[p];
        if (nghost) D->recv_counts[p] = recv_counts[p];
        ts += D->send_counts[p]; tr += D->recv_counts[p];
    }
    auto bail = [&](int rc) { vexhip_dist_spmv_destroy(reinterpret_cast<vexhip_dist_spmv *>(D)); return rc; };
    if (ts != nsend || tr != nghost) return bail(fail(__FILE__, __LINE__, "exchange counts do not add up to the buffer sizes"));
    hipError_t e = hipSetDevice(D->dev);
    if (e == hipSuccess) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
        e = hipStreamCreateWithPriority(&D->comm_stream, hipStreamNonBlocking, hi);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&D->packed, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&D->pushed, hipEventDisableTiming);
    if (e == hipSuccess && nsend > 0 && !env(ENV_VEXHIP_DIST_PACK)) e = detect_runs(D, send_idx, w->world);
    if (e != hipSuccess) return bail(check(e, __FILE__, __LINE__));
    // the push plan: one entry per destination with a share, blocks of kPushPerBlock elements
    std::vector<push_peer> plan;
    const size_t b = type_bytes(dtype);
    int64_t first = 0, blk = 0;
    for (int p = 0; p < w->world; ++p) {
        const int64_t k = D->send_counts[p];
        if (k) {
            if (!w->peer[p]) return bail(fail(__FILE__, __LINE__, "the window of a rank this rank sends to has not been opened (vexhip_ipc_window_open)"));
            push_peer q;
            q.dst = w->peer[p] + window_header(w->world) + (size_t)dst_offsets[p] * b;
            q.arrive = window_arrive(w->peer[p], w->rank);
            q.consumed = window_consumed(w->base, w->world, p);
            q.first = first; q.count = k; q.direct_first = D->direct ? D->send_first[p] : 0; q.blk0 = blk;
            plan.push_back(q);
            blk += (k + push_per_block() - 1) / push_per_block();
        }
        first += k;
    }
    D->npush = (int)plan.size(); D->push_blocks = blk;
    // the owners this rank waits for, and where it tells them that their share has been read
    std::vector<const unsigned long long *> arr; std::vector<unsigned long long *> con;
    for (int o = 0; o < w->world; ++o)
        if (D->recv_counts[o]) {
            if (!w->peer[o]) return bail(fail(__FILE__, __LINE__, "the window of a rank this rank receives from has not been opened (vexhip_ipc_window_open)"));
            arr.push_back(window_arrive(w->base, o));
            con.push_back(window_consumed(w->peer[o], w->world, w->rank));
        }
    D->nown = (int)arr.size();
    e = hipHostMalloc(reinterpret_cast<void **>(&D->d_err), sizeof(int), hipHostMallocMapped);       // host-visible without a copy
    if (e == hipSuccess) *D->d_err = 0;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&D->d_launches), sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(D->d_launches, 0, sizeof(unsigned long long));
    if (e == hipSuccess && D->npush) {
        e = hipMalloc(reinterpret_cast<void **>(&D->d_push), sizeof(push_peer) * plan.size());
        if (e == hipSuccess) e = hipMemcpy(D->d_push, plan.data(), sizeof(push_peer) * plan.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&D->d_done), sizeof(unsigned long long) * plan.size());
        if (e == hipSuccess) e = hipMemset(D->d_done, 0, sizeof(unsigned long long) * plan.size());
    }
    if (e == hipSuccess && D->nown) {
        e = hipMalloc(reinterpret_cast<void **>(&D->d_arrive), sizeof(void *) * arr.size());
        if (e == hipSuccess) e = hipMemcpy(D->d_arrive, arr.data(), sizeof(void *) * arr.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&D->d_consumed), sizeof(void *) * con.size());
        if (e == hipSuccess) e = hipMemcpy(D->d_consumed, con.data(), sizeof(void *) * con.size(), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) return bail(check(e, __FILE__, __LINE__));
    *out = reinterpret_cast<vexhip_dist_spmv *>(D);
    return 0;
}

// The one-launch step.  `ext` is the rank's strip stored as a grid matrix of (has_lower + planes + has_upper) planes: `halo`
// empty rows (one plane) in front of the rank's rows when it has a lower neighbour, one plane of empty rows behind them when it
// has an upper one, columns counted from the first element of the lower ghost plane.  The window holds [lower ghost plane | upper
// ghost plane]; the neighbours' windows must have been opened.  The plan owns the window's step counter (no other plan on it).
static int create_halo_impl(ipc_window *w, const vexhip_spmat *ext, int64_t rows, int64_t halo, int lower, int upper, int pull, vexhip_dist_spmv **out);

int vexhip_dist_spmv_create_halo(vexhip_ipc_window *hw, const vexhip_spmat *ext, int64_t rows, int64_t halo, int lower, int upper,
        vexhip_dist_spmv **out)
{
    return create_halo_impl(reinterpret_cast<ipc_window *>(hw), ext, rows, halo, lower, upper, 0, out);
}

// The PULL form of the one-launch step (round 6; one process drives every GPU): no share is copied -- the planes next to a ghost plane
// read the neighbours' boundary planes of x where they lie (vexhip_dist_spmv_apply_pull passes them with every product).
// order = VEXHIP_PULL_FLAGS: the windows (vexhip_ipc_window_create with data_bytes 0 + vexhip_ipc_window_attach) carry the flags
// ("x is final" at launch start, `consumed` behind the launch; the kernel behind the launch also waits for the neighbours'
// `consumed`).  order = VEXHIP_PULL_EVENTS: no flag is raised or waited for (`win` may be NULL): the CALLER orders the devices'
// streams with events -- x of the neighbours final before this launch, this launch finished before they overwrite x.
int vexhip_dist_spmv_create_halo_pull(vexhip_ipc_window *hw, const vexhip_spmat *ext, int64_t rows, int64_t halo, int lower, int upper,
        int order, vexhip_dist_spmv **out)
{
    VEXHIP_REQUIRE(order == VEXHIP_PULL_FLAGS || order == VEXHIP_PULL_EVENTS, "order must be VEXHIP_PULL_FLAGS or VEXHIP_PULL_EVENTS");
    return create_halo_impl(reinterpret_cast<ipc_window *>(hw), ext, rows, halo, lower, upper, order, out);
}

static int create_halo_impl(ipc_window *w, const vexhip_spmat *ext, int64_t rows, int64_t halo, int lower, int upper, int pull, vexhip_dist_spmv **out)
{
    reload_env();
    VEXHIP_REQUIRE(out, "NULL output");
    *out = nullptr;
    VEXHIP_REQUIRE((w || pull == 2) && ext, "NULL argument");
    VEXHIP_REQUIRE(rows > 0 && halo > 0 && halo < (1ll << 31) && rows % halo == 0, "the rank's rows must be whole planes of `halo` elements");
    const int world = w ? w->world : (1 << 30);
    VEXHIP_REQUIRE(lower >= -1 && lower < world && upper >= -1 && upper < world, "bad neighbour");
    VEXHIP_REQUIRE(pull || 2 * halo * 8 <= w->data_bytes, "the window is smaller than two ghost planes");
    int planes = 0, ny = 0, nx = 0, vtype = VEXHIP_F64;
    if (int rc = spmat_halo_geometry(ext, &planes, &ny, &nx, &vtype)) return rc;
    const int has_lo = lower >= 0 ? 1 : 0, has_hi = upper >= 0 ? 1 : 0;
    // round 6: not a grid matrix, but stored with diagonal codes (a general banded operator) whose diagonals stay within one ghost range:
    // the pair product's role (sell8.hip), pull form only
    int reach = 0;
    if (planes == 0 && pull) { if (int rc = spmat_halo_general(ext, halo, (int64_t)(has_lo + has_hi) * halo + rows, &reach, &vtype)) return rc; }
    VEXHIP_REQUIRE(reach > 0 || (planes > 0 && (int64_t)ny * nx == halo && planes == has_lo + rows / halo + has_hi && (pull || (nx == 512 && vtype == VEXHIP_F64))),
                   "the stored strip is not a plane-product matrix (pull: or a grid-product matrix, or a matrix with diagonal codes whose diagonals stay within one ghost range) "
                   "of (lower ghost plane +) the rank's planes (+ upper ghost plane)");
    VEXHIP_REQUIRE(!w || ((lower < 0 || w->peer[lower]) && (upper < 0 || w->peer[upper])), "a neighbour's window has not been opened (vexhip_ipc_window_open / _attach)");
    int ext_dev = 0;
    if (int rc = spmat_device(ext, &ext_dev)) return rc;
    VEXHIP_REQUIRE(!w || w->dev == ext_dev, "the window and the stored strip live on different devices");
    dist_spmv *D = new (std::nothrow) dist_spmv;
    VEXHIP_REQUIRE(D, "out of host memory");
    D->win = w; D->dev = ext_dev; D->dtype = vtype; D->rows = rows; D->halo = true; D->ext = ext; D->direct = true;
    D->pull = pull; D->has_lo = has_lo != 0; D->has_hi = has_hi != 0;
    D->nsend = (has_lo + has_hi) * halo; D->nghost = (has_lo + has_hi) * halo;
    if (w) {
        D->send_counts.assign(w->world, 0); D->recv_counts.assign(w->world, 0);
        if (has_lo) { D->send_counts[lower] += halo; D->recv_counts[lower] += halo; }
        if (has_hi) { D->send_counts[upper] += halo; D->recv_counts[upper] += halo; }
    }
    auto bail = [&](int rc) { vexhip_dist_spmv_destroy(reinterpret_cast<vexhip_dist_spmv *>(D)); return rc; };
    hipError_t e = hipSetDevice(D->dev);
    if (e == hipSuccess && !w) {
        e = hipMalloc(reinterpret_cast<void **>(&D->d_own_step), sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemset(D->d_own_step, 0, sizeof(unsigned long long));
    }
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&D->d_err), sizeof(int), hipHostMallocMapped);
    if (e == hipSuccess) *D->d_err = 0;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&D->d_halo_done), 4 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(D->d_halo_done, 0, 4 * sizeof(unsigned));
    const unsigned long long one = 1ull;                       // products are numbered from 1; the flags start at 0
    if (e == hipSuccess) e = hipMemcpy(w ? w->d_step : D->d_own_step, &one, sizeof(one), hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail(check(e, __FILE__, __LINE__));
    halo_dev &H = D->hd;
    H = halo_dev();
    H.pull = pull;
    if (!w) {
        // ordered by the caller's events: no window, no flag; the ghost planes come with every product
        H.step = D->d_own_step; H.done = D->d_halo_done; H.err = D->d_err; H.ticks = spin_ticks(); H.push_blocks = 0;
        H.halo = (int)halo; H.z0 = has_lo; H.z1 = has_lo + (int)(rows / halo);
        if (reach > 0) { H.lo_planes = reach; H.hi_planes = -1; }       // the pair product's role (spmat.hip spmat_apply_halo)
        *out = reinterpret_cast<vexhip_dist_spmv *>(D);
        return 0;
    }
    const size_t hdr = window_header(w->world);
    double *mine = reinterpret_cast<double *>(w->base + hdr);
    // window layout: [lower ghost plane | upper ghost plane]
    if (has_lo) {
        H.lo = mine;
        H.arrive_lo = window_arrive(w->base, lower);
        H.consumed_lo = window_consumed(w->peer[lower], w->world, w->rank);
        H.dst_lo = reinterpret_cast<double *>(w->peer[lower] + hdr) + halo;          // my first plane is the lower neighbour's UPPER ghost plane
        H.peer_arrive_lo = window_arrive(w->peer[lower], w->rank);
        H.sent_lo = window_consumed(w->base, w->world, lower);
    }
    if (has_hi) {
        H.hi = mine + halo;
        H.arrive_hi = window_arrive(w->base, upper);
        H.consumed_hi = window_consumed(w->peer[upper], w->world, w->rank);
        H.dst_hi = reinterpret_cast<double *>(w->peer[upper] + hdr);                 // my last plane is the upper neighbour's LOWER ghost plane
        H.peer_arrive_hi = window_arrive(w->peer[upper], w->rank);
        H.sent_hi = window_consumed(w->base, w->world, upper);
        if (upper == lower) {
            // one rank exchanging with itself (tools/r05_dist_step.py: the cost of a step on the one-GPU box): both sides would
            // share arrive[rank] / consumed[rank]; the upper side takes two spare words of the window's header
            if (upper != w->rank || 2 * (size_t)w->world + 2 > hdr / 8) return bail(fail(__FILE__, __LINE__, "lower and upper neighbour are the same rank"));
            unsigned long long *spare = reinterpret_cast<unsigned long long *>(w->base) + 2 * w->world;
            H.arrive_hi = spare; H.peer_arrive_hi = spare; H.consumed_hi = spare + 1; H.sent_hi = spare + 1;
        }
    }
    H.step = w->d_step; H.done = D->d_halo_done; H.err = D->d_err; H.ticks = spin_ticks();
    if (pull) { H.lo = H.hi = nullptr; H.dst_lo = H.dst_hi = nullptr; }       // nothing is copied: lo / hi arrive with every product
    H.push_blocks = 16;
    if (const char *pb = env(ENV_VEXHIP_HALO_PUSH_BLOCKS)) H.push_blocks = std::max(0, std::min(1024, std::atoi(pb)));       // 0: the product workgroups push (plane.hip)
    H.halo = (int)halo; H.z0 = has_lo; H.z1 = has_lo + (int)(rows / halo); H.lo_planes = 0; H.hi_planes = 0;
    if (reach > 0) { H.lo_planes = reach; H.hi_planes = -1; }           // the pair product's role (spmat.hip spmat_apply_halo)
    H.debug = nullptr;
    if (env(ENV_VEXHIP_HALO_DEBUG)) {                  // diagnostics: 6 words per workgroup of the LAST launch (tools/r05_halo_timeline.py reads them through vexhip_dist_spmv_debug)
        if (hipMalloc(reinterpret_cast<void **>(&D->d_halo_debug), 6 * 8 * 4096) == hipSuccess) { (void)hipMemset(D->d_halo_debug, 0, 6 * 8 * 4096); H.debug = D->d_halo_debug; }
    }
    H.lo_two_pass = 0;
    if (const char *tp = env(ENV_VEXHIP_HALO_TWO_PASS)) H.lo_two_pass = std::atoi(tp) != 0;
    H.acquire = 0;                                           // the ghost planes live in uncached memory (halo.hpp, spin_until)
    if (const char *aq = env(ENV_VEXHIP_HALO_ACQUIRE)) H.acquire = std::max(0, std::min(2, std::atoi(aq)));
    // A window in CACHED memory (VEXHIP_IPC_WINDOW_MEM=finegrained | default: a one-device diagnostic) gets the model-correct form
    // whatever was asked for: the reader invalidates at system scope behind the flag, the writers release at system scope in front of
    // it (H.release, plane.hip) -- the waitcnt-only hand-off rests on stores that go past every cache (advisor, round 5)
    H.release = 0;
    if (!w->uncached) { H.acquire = 2; H.release = 1; }
    // PULL reads the neighbours' x itself -- ordinary cached memory: behind the flag the workgroup's first lane invalidates what this
    // device may still hold of it (system scope: the planes may lie on another GPU)
    if (pull && !env(ENV_VEXHIP_HALO_ACQUIRE)) H.acquire = 2;
    H.one_launch = 1;
    if (const char *ol = env(ENV_VEXHIP_HALO_TWO_LAUNCHES)) H.one_launch = std::atoi(ol) ? 0 : 1;      // (A/B: the one-thread kernel behind the launch, round 5)
    if (env(ENV_VEXHIP_HALO_NO_PUSH)) {
        // diagnostics (tools/r05_dist_step.py): nobody pushes, the flags this rank waits for are raised once and for all -- what the
        // product with ghost planes costs when the exchange costs nothing
        const unsigned long long big = ~0ull >> 2;
        if (H.arrive_lo) (void)hipMemcpy(const_cast<unsigned long long *>(H.arrive_lo), &big, 8, hipMemcpyHostToDevice);
        if (H.arrive_hi) (void)hipMemcpy(const_cast<unsigned long long *>(H.arrive_hi), &big, 8, hipMemcpyHostToDevice);
        H.dst_lo = H.dst_hi = nullptr; H.consumed_lo = H.consumed_hi = nullptr;
        if (const char *g = env(ENV_VEXHIP_HALO_NO_GHOST)) {          // ... and nobody reads a ghost plane either (wrong numbers: the cost of the chunking alone); 1 both, 2 lower only, 3 upper only
            const int k = std::atoi(g);
            if (k == 1 || k == 2) H.lo = nullptr;
            if (k == 1 || k == 3) H.hi = nullptr;
        }
    }
    *out = reinterpret_cast<vexhip_dist_spmv *>(D);
    return 0;
}

/* diagnostics (VEXHIP_HALO_DEBUG=1 at creation): 6 x 64-bit words per workgroup of the last one-launch step -- start, ghost flag seen,
 * first ghost line in registers, end (100 MHz ticks), first plane, end plane (push workgroups: ~0, side) -- copied to `out` (up to
 * 4096 workgroups = 196 608 bytes); returns the number of workgroups' records the plan holds, 0 without the switch.                 */
int vexhip_dist_spmv_debug(vexhip_dist_spmv *h, void *out, int64_t bytes) {
    dist_spmv *D = reinterpret_cast<dist_spmv *>(h);
    VEXHIP_REQUIRE(D && out, "NULL argument");
    if (!D->d_halo_debug) return 0;
    VEXHIP_SET_DEVICE(D->dev);
    VEXHIP_TRY(hipDeviceSynchronize());
    VEXHIP_TRY(hipMemcpy(out, D->d_halo_debug, (size_t)std::min<int64_t>(bytes, 6 * 8 * 4096), hipMemcpyDeviceToHost));
    return 0;
}

int vexhip_dist_spmv_status(vexhip_dist_spmv *h, int *timed_out, int *transport, int *direct) {
    dist_spmv *D = reinterpret_cast<dist_spmv *>(h);
    VEXHIP_REQUIRE(D, "NULL argument");
    if (timed_out) {
        *timed_out = 0;
        if (D->d_err) *timed_out = *static_cast<volatile int *>(D->d_err);
    }
    if (transport) *transport = D->halo ? VEXHIP_COMM_HALO : D->win ? VEXHIP_COMM_IPC : VEXHIP_COMM_RCCL;
    if (direct) *direct = D->direct ? 1 : 0;
    return 0;
}

int vexhip_dist_spmv_profile(vexhip_dist_spmv *h, void *stream, double alpha, int append, const void *x, void *y, float *ms6) {
    dist_spmv *D = reinterpret_cast<dist_spmv *>(h);
    VEXHIP_REQUIRE(D && ms6 && (x || !D->rows) && (y || !D->rows), "NULL argument");
    if (!D->win && (D->nsend || D->nghost)) RCCL_READY();
    VEXHIP_SET_DEVICE(D->dev);
    hipStream_t s = as_stream(stream);
    hipEvent_t *ev = new (std::nothrow) hipEvent_t[7]();
    VEXHIP_REQUIRE(ev, "out of host memory");
    hipError_t e = hipSuccess;
    for (int k = 0; k < 7 && e == hipSuccess; ++k) e = hipEventCreate(&ev[k]);
    int rc = 0;
    if (e == hipSuccess) {
        D->prof = ev;
        rc = issue_step(D, s, alpha, append, x, y);
        D->prof = nullptr;
        if (!rc) e = hipStreamSynchronize(s);
        if (!rc && e == hipSuccess && D->comm_stream) e = hipStreamSynchronize(D->comm_stream);
        // total, local part, wait for the ghosts after the local part, remote part, pack, exchange
        const int pair[6][2] = {{0, 3}, {0, 1}, {1, 2}, {2, 3}, {4, 5}, {5, 6}};
        for (int k = 0; k < 6 && !rc && e == hipSuccess; ++k) e = hipEventElapsedTime(&ms6[k], ev[pair[k][0]], ev[pair[k][1]]);
    }
    for (int k = 0; k < 7; ++k) if (ev[k]) (void)hipEventDestroy(ev[k]);
    delete[] ev;
    if (rc) return rc;
    VEXHIP_TRY(e);
    return 0;
}

int vexhip_comm_rccl_info(const vexhip_comm *h, int *nranks, int *device, int *user_rank) {
    const comm *c = reinterpret_cast<const comm *>(h);
    VEXHIP_REQUIRE(c, "NULL communicator");
    if (nranks) *nranks = c->world;
    if (device) *device = c->devs.empty() ? -1 : c->devs[0];
    if (user_rank) *user_rank = c->ranks.empty() ? -1 : c->ranks[0];
    if (c->peer || c->comms.empty() || !c->comms[0]) return 0;
    RCCL_READY();
    rccl_api &a = rccl();
    if (nranks && a.CommCount) NCCL_TRY(a.CommCount(c->comms[0], nranks));            // what RCCL itself says
    if (device && a.CommCuDevice) NCCL_TRY(a.CommCuDevice(c->comms[0], device));
    if (user_rank && a.CommUserRank) NCCL_TRY(a.CommUserRank(c->comms[0], user_rank));
    return 0;
}

} // extern "C"

VEXHIP_WARM_TU(comm)
