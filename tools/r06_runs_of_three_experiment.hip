// Round 6 experiment, NOT part of the library (never compiled by csrc/Makefile; the form that WAS kept -- entries decoded at set-up -- is
// sell8v_runs_kernel in csrc/sell8.hip): "runs of three diagonals" for value-coded slices wider
// than nine columns (27-point stencils) in its LAST form -- triples found once per distinct slice at set-up (sell8v_runs_plan), the
// product persistent with the codes of a dictionary block kept in registers across the slices that share it.  Four forms were measured,
// all bit-identical (tests/test_gpu_spmv.py), none kept (profiles/r06_runs_of_three.md):
//   27-point constant coefficients   any-width kernel (kept)   triples found per wave,    ... all requests   triples from the set-up,   ... persistent, codes
//                                                              one at a time              first              one workgroup per slice    in registers (this text)
//   256^3                            0.24 ms                   0.285                      0.58               0.227                      0.261
//   320^3                            0.46 - 0.52 ms            0.55                       1.14               0.438                      0.518
// 54 gathers became 18 requests, the codes' 54 bytes per row no longer cross the L2 -- and the time did not move: the product of such
// slices is bound by what it does PER ENTRY (extract two codes, test for padding, read the value from the LDS table, select, multiply: ~10
// instructions, 3 456 entries per wave and slice), not by its requests.  A faster 27-point product needs the values decoded once per
// class of lines and kept -- the walk of the grid storage -- not fewer loads.
// ---------------------------------------------------------------------------
// RUNS of three diagonals (round 6): value-coded slices wider than nine columns whose distinct slices live in the dictionary -- 19- and
// 27-point stencils, 9-point operators in 2-D, dense bands.  Such a row holds its entries in triples on consecutive diagonals d-1, d,
// d+1 (the three x-neighbours of one (y, z) neighbour): the six elements of x a lane's two rows need for a triple are x[i+d-1 ..
// i+d+2] -- ONE 16-byte request per lane (x[i+d], x[i+d+1]), the neighbour lanes' halves by DPP, and the two ends of the wave by one
// more request (every lane asks for the element in front of the wave, lane 63 for the one behind it: two cache lines).  27 points: 9 + 9
// requests instead of 54 gathers, each of which costs the any-width kernel ~20 cycles of the CU's address path (0.46 ms at 320^3).
// WHICH columns of a wave's 128 rows are such a triple is found ONCE per distinct slice, at set-up (sell8v_runs_plan_kernel: columns
// 3k .. 3k+2, every entry of each on one diagonal or padding, the diagonals consecutive) and handed to the product as a mask and the
// centre diagonals per wave -- found per wave inside the product from the codes themselves, the same idea was SLOWER than the gathers
// (profiles/r06_runs_of_three.md: 0.55 / 1.14 ms; for the one triple of a seven-point row it lost in round 2 already, see the pair
// kernels below).  The product asks for every triple's elements first, then forms the products in column order: those of the
// any-width kernel, bit for bit.  Groups that are no triple (the first columns of a wave that holds boundary rows) gather.
constexpr int RUNS_GROUPS = 11;            // groups of three columns: ELL widths up to 33
constexpr int RUNS_STRIDE = 12;            // ints per (block, wave): the mask, then the centre diagonal of every group

__global__ __launch_bounds__(256)
void sell8v_runs_plan_kernel(const char *__restrict__ pool, int w, const int *__restrict__ deltas, int *__restrict__ desc, int *__restrict__ total)
{
    __shared__ int s_delta[256];
    s_delta[threadIdx.x] = deltas[threadIdx.x];
    __syncthreads();
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wp = (w + 1) / 2;
    const unsigned *cw = reinterpret_cast<const unsigned *>(pool + (long long)blockIdx.x * ((long long)wp * 2048)) + t;
    // column j of this wave: every entry on one diagonal?  (-> that diagonal; INT_MIN: no, or no entry at all)
    auto diagonal_of = [&](int j) -> int {
        const unsigned word = cw[(j >> 1) * 256] >> (16 * (j & 1));
        const unsigned c0 = word & 255u, c1 = (word >> 8) & 255u;
        const bool e0 = c0 < S8_FIRST_PAD, e1 = c1 < S8_FIRST_PAD;
        const unsigned long long any = __builtin_amdgcn_ballot_w64(e0 || e1);
        if (!any) return INT_MIN;
        const unsigned cu = (unsigned)__shfl((int)(e0 ? c0 : c1), __builtin_ctzll(any), 64);
        if (__builtin_amdgcn_ballot_w64((e0 && c0 != cu) || (e1 && c1 != cu))) return INT_MIN;
        return s_delta[cu];
    };
    int *out = desc + ((long long)blockIdx.x * 4 + wave) * RUNS_STRIDE;
    unsigned mask = 0;
    for (int k = 0; k < RUNS_GROUPS; ++k) {
        int dc = 0;
        if (3 * k + 2 < w) {
            const int d0 = diagonal_of(3 * k), d1 = diagonal_of(3 * k + 1), d2 = diagonal_of(3 * k + 2);
            if (d0 != INT_MIN && d1 != INT_MIN && d2 != INT_MIN && (long long)d1 == (long long)d0 + 1 && (long long)d2 == (long long)d0 + 2) { mask |= 1u << k; dc = d1; }
        }
        if (lane == 0) out[1 + k] = dc;
    }
    if (lane == 0) { out[0] = (int)mask; if (mask) atomicAdd(total, __popc(mask)); }
}

// The product: workgroups stay and take slices s, s + gridDim.x, ...; the launch's stride is a multiple of the PERIOD of the matrix in
// slices (sell8v_runs_plan: blocks[s] == blocks[s + p] almost everywhere -- 320-point lines against 512-row slices repeat every five),
// so a workgroup meets the same dictionary block slice after slice and keeps its 28 words of codes per lane IN REGISTERS: launched per
// slice the product re-read them from the L2 every time (54 bytes per row against the 16 of x and y) behind two dependent requests
// (block number -> codes); what is left per slice is one round of requests for x and the store.
template <typename V>
__global__ __launch_bounds__(256)
void sell8v_runs_kernel(long long n, long long nslices, V alpha, int append, int w,
        const char *__restrict__ pool, const int *__restrict__ deltas, const V *__restrict__ values,
        const int *__restrict__ csr_ptr, const int *__restrict__ csr_col, const V *__restrict__ csr_val,
        const V *__restrict__ x, V *__restrict__ y, trav_dev trav, const int *__restrict__ blocks, const int *__restrict__ desc, long long x_last)
{
    constexpr int G = RUNS_GROUPS;
    typedef typename vec2<V>::type V2;
    __shared__ int s_delta[256];
    __shared__ V s_value[256];
    s_delta[threadIdx.x] = deltas[threadIdx.x];
    s_value[threadIdx.x] = values[threadIdx.x];
    __syncthreads();

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wp = (w + 1) / 2;
    long long cur = -1;                                  // the dictionary block whose codes the registers hold
    unsigned c[(3 * G + 1) / 2], vc[(3 * G + 1) / 2];
    unsigned tmask0 = 0;
    int dcs[G];
#pragma unroll
    for (int jp = 0; jp < (3 * G + 1) / 2; ++jp) { c[jp] = 0xffffffffu; vc[jp] = 0u; }
#pragma unroll
    for (int k = 0; k < G; ++k) dcs[k] = 0;

    for (long long s = blockIdx.x; s < nslices; s += gridDim.x) {
        const long long sb = (long long)__builtin_amdgcn_readfirstlane(blocks[s]);
        if (sb != cur) {                                                               // uniform
            const unsigned *cw = reinterpret_cast<const unsigned *>(pool + sb * ((long long)wp * 2048)) + t;
            const unsigned *vw = cw + wp * 256;
#pragma unroll
            for (int jp = 0; jp < (3 * G + 1) / 2; ++jp) { c[jp] = jp < wp ? cw[jp * 256] : 0xffffffffu; vc[jp] = jp < wp ? vw[jp * 256] : 0u; }
            const int *dw = desc + (sb * 4 + wave) * RUNS_STRIDE;
            tmask0 = (unsigned)__builtin_amdgcn_readfirstlane(dw[0]);
#pragma unroll
            for (int k = 0; k < G; ++k) dcs[k] = __builtin_amdgcn_readfirstlane(dw[1 + k]);
            cur = sb;
        }
        const long long i = s * S8_ROWS + 2 * t;
        const long long r_lo = s * S8_ROWS + 128 * wave, r_hi = r_lo + 127;     // the wave's rows
        unsigned tmask = tmask0;
        // every triple's elements of x, requested before anything is used
        V2 Pk[G];
        V Ek[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            if ((tmask >> k) & 1u) {                                                   // uniform
                const long long dc = (long long)dcs[k];
                if (r_lo + dc - 1 >= 0 && r_hi + dc + 1 <= x_last) {
                    __builtin_memcpy(&Pk[k], x + (i + dc), sizeof(V2));               // (4-byte alignment is enough for the wide load)
                    Ek[k] = x[lane == 63 ? i + dc + 2 : r_lo + dc - 1];
                } else tmask &= ~(1u << k);                                            // the first / last waves of the matrix: gathers
            }
        }
        V sum[2] = {V(0), V(0)};
#pragma unroll
        for (int k = 0; k < G; ++k) {
            if (3 * k >= w) continue;                                                  // uniform
            V xs[3][2] = {{V(0), V(0)}, {V(0), V(0)}, {V(0), V(0)}};
            const bool triple = (tmask >> k) & 1u;                                     // uniform
            if (triple) {
                const V2 P = Pk[k]; const V E = Ek[k];
                xs[0][0] = shift_from_lower_lane(P.y, E); xs[0][1] = P.x;              // diagonal dc - 1: x[i + dc - 1], x[i + dc]
                xs[1][0] = P.x; xs[1][1] = P.y;                                        // diagonal dc
                xs[2][0] = P.y; xs[2][1] = shift_from_upper_lane(P.x, E);              // diagonal dc + 1: x[i + dc + 1], x[i + dc + 2]
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int j = 3 * k + u;
                if (j >= w) continue;                                                  // uniform
                const int sh = 16 * (j & 1);
                const unsigned c0 = (c[j >> 1] >> sh) & 255u, c1 = (c[j >> 1] >> (sh + 8)) & 255u;
                const bool e0 = c0 < S8_FIRST_PAD, e1 = c1 < S8_FIRST_PAD;
                V x0 = xs[u][0], x1 = xs[u][1];
                if (!triple) {
                    if (e0) x0 = x[i + s_delta[c0]];
                    if (e1) x1 = x[i + 1 + s_delta[c1]];
                }
                if (e0) sum[0] += s_value[(vc[j >> 1] >> sh) & 255u] * x0;
                if (e1) sum[1] += s_value[(vc[j >> 1] >> (sh + 8)) & 255u] * x1;
            }
        }
        if (csr_ptr) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (i + q < n)
                    for (int j = csr_ptr[i + q], e = csr_ptr[i + q + 1]; j < e; ++j) sum[q] += csr_val[j] * x[csr_col[j]];
        }
        store_pair<V>(n, i, alpha, append, sum, y, trav);
    }
}


// ---- runs of three diagonals (above): the plan of a dictionary's blocks, and the product (spmat.hip) ----
// desc_out: device array of nblocks x 4 waves x RUNS_STRIDE ints (hipFree), NULL when no wave of any block holds a triple;
// period_out: p <= 64 with blocks[s] == blocks[s + p] for 19 slices in 20 of the matrix's middle (0: none)
int sell8v_runs_plan(int dev, void *stream, const void *pool, int64_t nblocks, int64_t w, const int *deltas, const int *blocks, int64_t nslices, int **desc_out, int *period_out)
{
    VEXHIP_REQUIRE(desc_out && period_out, "NULL output");
    *desc_out = nullptr; *period_out = 0;
    if (!pool || !deltas || !blocks || nblocks < 1 || w < 10 || w > 3 * RUNS_GROUPS) return 0;
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    int *desc = nullptr, *total = nullptr;
    VEXHIP_TRY(hipMalloc(reinterpret_cast<void **>(&desc), sizeof(int) * (size_t)(nblocks * 4 * RUNS_STRIDE + 1)));
    total = desc + nblocks * 4 * RUNS_STRIDE;
    hipError_t e = hipMemsetAsync(total, 0, sizeof(int), s);
    if (e == hipSuccess) { sell8v_runs_plan_kernel<<<(unsigned)nblocks, 256, 0, s>>>(static_cast<const char *>(pool), (int)w, deltas, desc, total); e = hipGetLastError(); }
    int found = 0;
    std::vector<int> id((size_t)nslices);
    if (e == hipSuccess) e = hipMemcpyAsync(&found, total, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(id.data(), blocks, sizeof(int) * (size_t)nslices, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess || found == 0) { (void)hipFree(desc); return e == hipSuccess ? 0 : check(e, __FILE__, __LINE__); }
    const int64_t lo = nslices / 4, hi = std::min<int64_t>(nslices - 64, lo + 20000);
    for (int p = 1; p <= 64 && hi > lo; ++p) {
        int64_t same = 0;
        for (int64_t k = lo; k < hi; ++k) same += id[(size_t)k] == id[(size_t)(k + p)];
        if (same * 20 >= (hi - lo) * 19) { *period_out = p; break; }
    }
    *desc_out = desc;
    return 0;
}

template <typename V>
static int sell8v_runs_apply_impl(int dev, void *stream, int64_t n, V alpha, int append, int64_t w, const void *pool, const int *blocks,
        const int *deltas, const V *values, const int *cp, const int *cc, const V *cv, const V *x, V *y, const int *desc, long long x_last, int period)
{
    VEXHIP_REQUIRE(n >= 0 && w >= 10 && w <= 3 * RUNS_GROUPS && pool && blocks && deltas && values && desc && x_last >= 0, "bad arguments of the runs product");
    if (n == 0) return 0;
    VEXHIP_REQUIRE(x && y, "NULL vector");
    VEXHIP_SET_DEVICE(dev);
    hipStream_t s = as_stream(stream);
    const long long ns = (n + S8_ROWS - 1) / S8_ROWS;
    // workgroups that stay: as many as the device holds at once, their number a multiple of the matrix's period in slices
    static thread_local int per_cu = 0;
    if (per_cu == 0 && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sell8v_runs_kernel<V>, 256, 0) != hipSuccess) per_cu = 0;
    long long grid = std::min<long long>(ns, (long long)std::max(1, per_cu) * std::max(1, info(dev).cus));
    if (period > 1 && grid > period) grid -= grid % period;
    trav_dev t8 = with_addend(trav_dev{nullptr, 0, 0, 0});
    sell8v_runs_kernel<V><<<(unsigned)grid, 256, 0, s>>>(n, ns, alpha, append, (int)w, static_cast<const char *>(pool), deltas, values, cp, cc, cv, x, y, t8, blocks, desc, x_last);
    VEXHIP_LAUNCH_CHECK();
    return 0;
}
int sell8v_runs_apply(int dev, void *stream, int64_t n, double alpha, int append, int64_t w, const void *pool, const int *blocks, const int *deltas, const double *values,
        const int *cp, const int *cc, const double *cv, const double *x, double *y, const int *desc, long long x_last, int period)
{ return sell8v_runs_apply_impl<double>(dev, stream, n, alpha, append, w, pool, blocks, deltas, values, cp, cc, cv, x, y, desc, x_last, period); }
int sell8v_runs_apply(int dev, void *stream, int64_t n, float alpha, int append, int64_t w, const void *pool, const int *blocks, const int *deltas, const float *values,
        const int *cp, const int *cc, const float *cv, const float *x, float *y, const int *desc, long long x_last, int period)
{ return sell8v_runs_apply_impl<float>(dev, stream, n, alpha, append, w, pool, blocks, deltas, values, cp, cc, cv, x, y, desc, x_last, period); }


