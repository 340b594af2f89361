// The product step of one rank in ONE launch (round 5; reference: the five phases of vexcl/spmat.hpp:120-185 -- gather the
// boundary values, ship them, local part, wait, remote part).  For a matrix whose remote columns are the plane below the rank's
// first plane and the plane above its last one (a plane partition of a 7-point operator), the rank's strip is stored as ONE grid
// matrix that includes the entries reaching into the neighbours' planes, and the plane product (plane.hip) reads those two ghost
// planes straight from the rank's peer-mapped window (comm.hip) -- no remote part, no wait kernel, no second stream:
//   workgroups 0 .. 2 * push_blocks - 1 : copy the rank's first / last plane of x into the neighbours' windows and raise
//                                         `arrive` there (dispatched first: the shares are on their way before the product starts);
//   the other workgroups                : the plane product; a workgroup checks `arrive` when its walk first needs a line of a
//                                         ghost plane -- only the SHORT chunks next to the two ghost planes ever do (dispatched
//                                         behind the main chunks, which stream the bulk of the strip meanwhile);
//   a one-thread kernel behind it       : raises `consumed` at the owners of the ghost planes and advances the step number.
// A launch waits only for launches of OTHER ranks, and a rank runs its products one after the other: with a GPU per rank nothing
// a launch waits for competes with it for CUs.  Ranks that SHARE a GPU (the one-device stand-ins of the tests and of
// `bench.py --one-device`) do: the waiting workgroups of all ranks must leave room for the pushes they wait for -- up to 256 per
// ghost plane at 512 lines per plane against 768 resident workgroups; two ranks always fit, three only with shorter planes.
#pragma once
#include "common.hpp"

// The flag hand-offs of comm.hip and plane.hip wait for a wave's STORES with `s_waitcnt vmcnt(0)`: on the gfx9 family (gfx90a,
// gfx942, gfx950) vmcnt counts loads and stores alike; gfx10 and later count stores in vscnt and would need `s_waitcnt_vscnt`.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "libvexhip's inter-GPU flag protocol waits for stores with s_waitcnt vmcnt(0): gfx9-family targets only (ARCH in csrc/Makefile)"
#endif

namespace vexhip {

struct halo_dev {
    const double *lo, *hi;                              // this rank's ghost planes (in its window); NULL: no neighbour on that side
    const unsigned long long *arrive_lo, *arrive_hi;    // in this rank's window: the owner has written the share of product s
    unsigned long long *consumed_lo, *consumed_hi;      // in the owners' windows: this rank has read the share of product s
    double *dst_lo, *dst_hi;                            // where this rank's first / last plane goes: the lower neighbour's upper ghost plane, the upper neighbour's lower one
    unsigned long long *peer_arrive_lo, *peer_arrive_hi;
    const unsigned long long *sent_lo, *sent_hi;        // in this rank's window: the neighbour has read this rank's previous share
    unsigned long long *step;                           // device word: the number of the product this launch computes (advanced by its last workgroup)
    unsigned *done;                                     // [0]: workgroups of the launch that have finished; [1], [2]: push workgroups per side
    int *err;                                           // sticky, pinned host memory: a flag was not raised in time
    unsigned long long ticks;                           // bound of a flag wait (100 MHz ticks)
    int push_blocks;                                    // workgroups per side that copy a plane
    int halo;                                           // elements of a ghost plane
    int z0, z1;                                         // planes of the stored grid this launch computes: [z0, z1)
    int lo_planes, hi_planes;                           // planes of the short chunks next to the lower / upper ghost plane (0: none); the pair product's role (sell8.hip): lo_planes = reach of the diagonals in rows
    unsigned long long *debug;                          // diagnostics (VEXHIP_HALO_DEBUG): per workgroup {start, ghost flag seen, first ghost line in registers, end} in 100 MHz ticks
    int lo_two_pass;                                    // the lower chunk walks its planes above the first one first and its first plane (the one that needs the ghost plane) last
    int acquire;                                        // behind a ghost flag: 0 no cache invalidate (the window is uncached), 1 agent scope, 2 system scope
    int release;                                        // in front of a raised flag: 0 s_waitcnt only (uncached window: the stores go past every cache), 1 system-scope release fence
    // PULL (round 6; one process drives every GPU -- vex::Context, vexcl/spmat.hpp): lo / hi point at the NEIGHBOURS' boundary planes
    // of x themselves (peer access: hipDeviceEnablePeerAccess), nothing is copied.  1 = flags: the launch's first workgroup raises
    // `arrive` at the neighbours ("my x is final": everything in front of this launch in the stream has finished), the workgroups
    // next to a ghost plane wait for the neighbours' flags, and the kernel behind the launch raises `consumed` AND waits for the
    // neighbours' `consumed` before the stream goes on (a later kernel may overwrite the plane they read).  2 = no flags at all:
    // the host orders the streams with events (logical devices sharing one GPU may share a hardware queue, where a launch that
    // waits for a LATER launch never ends).
    int pull;
    int one_launch;                                     // round 6: the launch's LAST workgroup raises `consumed` and advances the step number (no second launch)
};

// returns false when the flag was not raised in time (err, in pinned host memory, is set then and stays set: the products that
// are already queued fail fast instead of waiting `ticks` each; the host refuses further ones)
// acquire: 2 = system-scope acquire once the flag is there (invalidates this CU's L1 AND the XCD's L2 lines that are not kept coherent:
// the right thing in front of loads from CACHED memory another device wrote), 1 = agent scope, 0 = none -- for loads from the UNCACHED
// window, which no cache ever holds: the flag's value has returned before the first data load is issued, and that is all the order
// an uncached read needs.
__device__ inline bool spin_until(const unsigned long long *flag, unsigned long long want, int *err, unsigned long long ticks, int acquire = 2) {
    // relaxed polls (the flags are uncached: every poll reads memory), ONE acquire once the flag is there.  The common case -- the flag
    // is already up -- costs one read: the sticky error word (pinned HOST memory: a trip over PCIe) and the clock are looked at only
    // once a poll has missed
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return false;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > ticks) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return false; }
        }
    }
    if (acquire == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    else if (acquire == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return true;
}

// PULL, order flags: the launch's first workgroup tells the neighbours that this rank's x is final (stream order: its writers were
// earlier KERNELS, whose end wrote the caches back).  A RELAXED store: nothing of THIS launch is published, and a release fence here
// writes back an L2 that the launch's other workgroups keep filling with y -- the flag left 65 us late (profiles/r06_dist_step_first.json:
// 117 us a step against 49 without flags); H.release asks for the fence anyway.
__device__ inline void halo_announce(const halo_dev &H, unsigned long long step) {
    if (H.pull == 1 && blockIdx.x == 0 && threadIdx.x == 0) {
        if (H.release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        if (H.peer_arrive_lo) __hip_atomic_store(H.peer_arrive_lo, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (H.peer_arrive_hi) __hip_atomic_store(H.peer_arrive_hi, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// What used to be a second launch (halo_signal_kernel, round 5) is done by the workgroup that finishes LAST: every workgroup's reads of the
// ghost planes have returned before it is counted (s_waitcnt + barrier; a relaxed count: an agent-scope release here would write back an
// L2 full of this launch's y, once per workgroup), so the last one may tell the owners that their planes have been read, wait -- PULL: the
// planes are the owners' x itself -- until the neighbours say the same of this rank's, and advance the step number.  The next launch
// of the stream starts behind this one: it reads the new number.  Called by EVERY workgroup of the launch, idle ones included.
// (`counted`, `target`: a launch of MANY short workgroups of which few read a ghost range -- the pair product's role, sell8.hip --
// counts only those and workgroup 0, which has raised the flags: the wait for the stores' acknowledgements in front of the count
// costs a workgroup that lives 12 us a quarter of its life, and the others have nothing the neighbours wait for.)
__device__ inline void halo_finish(const halo_dev &H, unsigned long long step, bool counted = true, unsigned target = 0) {
    if (!H.one_launch || !counted) return;                                // (uniform)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(H.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == (target ? target : gridDim.x)) {
            __hip_atomic_store(H.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (H.consumed_lo) __hip_atomic_store(H.consumed_lo, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (H.consumed_hi) __hip_atomic_store(H.consumed_hi, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (H.pull == 1) {
                if (H.sent_lo) (void)spin_until(H.sent_lo, step, H.err, H.ticks, 0);
                if (H.sent_hi) (void)spin_until(H.sent_hi, step, H.err, H.ticks, 0);
            }
            *H.step = step + 1ull;
        }
    }
}

// A workgroup that will read a ghost plane waits here, once, before its walk (the walks of grid.hip / plane32.hip / grid32.hip that
// touch a ghost plane are short and dispatched last; plane.hip waits when the walk first needs a ghost line).  false: the owner's flag
// did not come (the sticky error is set; the caller fills its ghost values with NaN).  flag: one LDS word of the caller.
__device__ inline bool halo_wait(const halo_dev &H, unsigned long long step, bool need_lo, bool need_hi, int *flag) {
    if (H.pull == 2 || !(need_lo || need_hi)) return true;         // (uniform)
    if (threadIdx.x == 0) {
        bool ok = true;
        if (need_lo) ok = spin_until(H.arrive_lo, step, H.err, H.ticks, H.acquire) && ok;
        if (need_hi) ok = spin_until(H.arrive_hi, step, H.err, H.ticks, H.acquire) && ok;
        *flag = ok ? 1 : 0;
    }
    __syncthreads();
    const bool ok = *flag != 0;
    __syncthreads();
    return ok;
}

} // namespace vexhip
