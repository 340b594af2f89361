// YARDSTICK ONLY -- never linked into libvexhip.so, never called by the product (tools/README.md).
// rocprim::radix_sort_keys (rocPRIM's keys-only sort: onesweep on this part) on the same 1e9 hashed u32 keys the bench sorts
// (bench.py `sort u32 keys n=1e9`; vex::sort reference: /root/reference/vexcl/sort.hpp:1716-1870), timed with HIP events and --
// under `rocprofv3 --kernel-trace --stats` -- traced per kernel.  Built on the GPU box by tools/r06_gpu1.sh:
//   hipcc -O3 --offload-arch=gfx950 tools/r06_rocprim_sort.hip -o gpurun_out/r06_rocprim_sort
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

// the counter hash of vexcl_amd/csrc/misc.hip fill_hash (splitmix-style finaliser), restated: same distribution, full range
__global__ void fill(unsigned *k, long long n, unsigned seed) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned long long z = (unsigned long long)i + 0x9e3779b97f4a7c15ull * (seed + 1ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; z ^= z >> 31;
    k[i] = (unsigned)z;
}
__global__ void check_sorted(const unsigned *k, long long n, unsigned long long *bad) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i + 1 < n && k[i] > k[i + 1]) atomicAdd(bad, 1ull);
}

int main(int argc, char **argv) {
    const long long n = argc > 1 ? (long long)std::atof(argv[1]) : 1000000000ll;
    const int reps = argc > 2 ? std::atoi(argv[2]) : 5;
    unsigned *in, *out; unsigned long long *bad;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    size_t tmp_bytes = 0;
    CK(rocprim::radix_sort_keys(nullptr, tmp_bytes, in, out, (size_t)n, 0, 32, 0));
    void *tmp; CK(hipMalloc(&tmp, tmp_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int r = 0; r < reps; ++r) {
        fill<<<(unsigned)((n + 255) / 256), 256>>>(in, n, 42u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        CK(rocprim::radix_sort_keys(tmp, tmp_bytes, in, out, (size_t)n, 0, 32, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    check_sorted<<<(unsigned)((n + 255) / 256), 256>>>(out, n, bad);
    unsigned long long hbad = 0; CK(hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost));
    std::sort(ms.begin(), ms.end());
    std::printf("{\"what\": \"rocprim::radix_sort_keys u32 (yardstick, not the product)\", \"n\": %lld, \"tmp_bytes\": %zu, \"best_ms\": %.3f, \"median_ms\": %.3f, "
                "\"gkeys_per_s\": %.1f, \"inversions\": %llu}\n", n, tmp_bytes, ms.front(), ms[ms.size() / 2], n / ms.front() / 1e6, hbad);
    return hbad ? 2 : 0;
}
