// Diagnostic (tools/ only): how fast is a RE-READ of a working set that fits the 256 MiB Infinity Cache, and a read of what a
// scatter-like kernel has just written?  Decides whether a chunked sort pass (histogram of a chunk, then its scatter re-reading the
// chunk from the memory-side cache) can take the histogram reads off the HBM.
//   hipcc -O3 --offload-arch=gfx950 tools/r06_mall_probe.hip -o tools/build/r06_mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u4 *__restrict__ p, long long nv, unsigned *sink) {
    unsigned acc = 0;
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < nv; i += stride) {
        u4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const long long j = i + u * 256; q[u] = j < nv ? (NT ? __builtin_nontemporal_load(p + j) : p[j]) : u4{0, 0, 0, 0}; }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += q[u].x ^ q[u].y ^ q[u].z ^ q[u].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void write_kernel(u4 *p, long long nv) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride) p[i] = u4{(unsigned)i, 1u, 2u, 3u};
}
int main() {
    const size_t total = 4ull << 30;
    char *buf; unsigned *sink; CK(hipMalloc(&buf, total)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, total));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t sizes[] = {32ull << 20, 64ull << 20, 100ull << 20, 128ull << 20, 192ull << 20, 256ull << 20, 384ull << 20, 1ull << 30, 4ull << 30};
    std::printf("[\n");
    for (int nt = 0; nt < 2; ++nt)
    for (size_t sz : sizes) {
        const long long nv = (long long)(sz / 16);
        const unsigned grid = (unsigned)std::min<long long>((nv + 1023) / 1024, 256 * 16);
        // (a) re-read of the same working set, back to back
        for (int w = 0; w < 2; ++w) { if (nt) read_kernel<true><<<grid, 256>>>((const u4 *)buf, nv, sink); else read_kernel<false><<<grid, 256>>>((const u4 *)buf, nv, sink); }
        const int reps = sz >= (1ull << 30) ? 5 : 40;
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) { if (nt) read_kernel<true><<<grid, 256>>>((const u4 *)buf, nv, sink); else read_kernel<false><<<grid, 256>>>((const u4 *)buf, nv, sink); }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // (b) write the set, then read it (one pair timed as a whole, the write alone timed too)
        float msw = 0, mswr = 0;
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) write_kernel<<<grid, 256>>>((u4 *)buf, nv);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&msw, e0, e1));
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) { write_kernel<<<grid, 256>>>((u4 *)buf, nv); if (nt) read_kernel<true><<<grid, 256>>>((const u4 *)buf, nv, sink); else read_kernel<false><<<grid, 256>>>((const u4 *)buf, nv, sink); }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&mswr, e0, e1));
        std::printf(" {\"nontemporal_loads\": %d, \"MiB\": %zu, \"reread_TBps\": %.2f, \"write_TBps\": %.2f, \"read_after_write_TBps\": %.2f},\n", nt, sz >> 20,
                    (double)sz * reps / ms / 1e9, (double)sz * reps / msw / 1e9, (double)sz * reps / (mswr - msw) / 1e9);
    }
    std::printf(" {}\n]\n");
    return 0;
}
