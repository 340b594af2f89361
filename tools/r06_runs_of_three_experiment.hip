// Round 6 experiment, NOT part of the library (never compiled by csrc/Makefile): the kernel text of "runs of three diagonals" for value-coded
// slices wider than nine columns (27-point stencils), as it stood when it was measured and dropped.  It replaced the default launch of
// spmv_sell8v (sell8.hip) for w in 10 .. 32 when spmat.hip had said how far x reaches; bit-identical (tests/test_gpu_spmv.py), and SLOWER:
//   27-point constant coefficients      any-width kernel (kept)     one triple at a time     all requests first (this text)
//   256^3                               0.24 ms                     0.285 ms                 0.58 ms
//   320^3                               0.46 - 0.52 ms              0.55 ms                  1.14 ms
// One triple at a time: nine dependent round trips to memory per wave and slice (the any-width kernel makes four).  All requests first:
// 151 registers (three waves per SIMD), 8 000 instructions, 350 exec-mask regions -- the per-wave analysis of the codes (is a column on one
// diagonal?) does not compile into uniform control flow.  What would pay is the analysis done ONCE per distinct slice at set-up (a list of
// (column, kind, diagonal) per wave of a dictionary block, read by scalar loads) -- or the walk in registers that the grid storage uses.
// profiles/r06_runs_of_three.md.
// ---------------------------------------------------------------------------
// RUNS of three diagonals (round 6): value-coded slices wider than eight columns -- 19- and 27-point stencils, 9-point operators in
// 2-D.  Such a row holds its entries in triples on consecutive diagonals d-1, d, d+1 (the three x-neighbours of one (y, z)
// neighbour): the six elements of x a lane's two rows need for a triple are x[i+d-1 .. i+d+2] -- ONE 16-byte request per lane
// (x[i+d], x[i+d+1]) plus the neighbour lanes' halves by DPP, the two ends of the wave by one more request (every lane asks for the
// element in front of the wave, lane 63 for the one behind it: two cache lines).  27 points: 9 + 9 requests instead of 54 gathers
// (the any-width kernel above spends ~20 cycles of the CU's address path on each: 0.46 ms at 320^3).
// Whether columns j, j+1, j+2 of a wave's 128 rows are such a triple is found per wave from the codes themselves: every entry of a
// column on ONE diagonal (or padding), the three diagonals consecutive, every request inside x.  Anything else -- the columns of a
// wave that holds boundary rows, ragged ends -- takes the guarded 8-byte gathers.  (For SEVEN-point rows the same idea lost to its
// bookkeeping in round 2, see the pair kernels below: one triple in seven columns; here it is nine in nine.)
// The products and their order are those of the any-width kernel: bit-identical.
template <typename V>
__global__ __launch_bounds__(256)
void sell8v_runs_kernel(long long n, long long nslices, V alpha, int append, int w,
        const char *__restrict__ buf, const int *__restrict__ deltas, const V *__restrict__ values,
        const int *__restrict__ csr_ptr, const int *__restrict__ csr_col, const V *__restrict__ csr_val,
        const V *__restrict__ x, V *__restrict__ y, trav_dev trav, const int *__restrict__ blocks, long long x_last)
{
    constexpr int MAXW = 32;
    typedef typename vec2<V>::type V2;
    __shared__ int s_delta[256];
    __shared__ V s_value[256];
    s_delta[threadIdx.x] = deltas[threadIdx.x];
    s_value[threadIdx.x] = values[threadIdx.x];
    __syncthreads();

    const long long s = traversal_block(trav, nslices);
    if (s < 0) return;
    const int t = threadIdx.x, lane = t & 63;
    const long long i = s * S8_ROWS + 2 * t;
    const long long r_lo = s * S8_ROWS + 128 * (t >> 6), r_hi = r_lo + 127;        // the wave's rows (uniform)
    const int wp = (w + 1) / 2;
    const long long sb = blocks ? (long long)blocks[s] : s;
    const unsigned *cw = reinterpret_cast<const unsigned *>(buf + sb * ((long long)wp * 2048)) + t;
    const unsigned *vw = cw + wp * 256;
    unsigned c[MAXW / 2], vc[MAXW / 2];
#pragma unroll
    for (int jp = 0; jp < MAXW / 2; ++jp) { c[jp] = jp < wp ? cw[jp * 256] : 0xffffffffu; vc[jp] = jp < wp ? vw[jp * 256] : 0u; }

    // column j of this wave: every entry on one diagonal?  (-> that diagonal; INT_MIN: no, or no entry at all)
    auto diagonal_of = [&](int j) -> int {
        const unsigned c0 = (c[j >> 1] >> (16 * (j & 1))) & 255u, c1 = (c[j >> 1] >> (16 * (j & 1) + 8)) & 255u;
        const bool e0 = c0 < S8_FIRST_PAD, e1 = c1 < S8_FIRST_PAD;
        const unsigned long long any = __builtin_amdgcn_ballot_w64(e0 || e1);
        if (!any) return INT_MIN;
        const unsigned cu = (unsigned)__builtin_amdgcn_readlane((int)(e0 ? c0 : c1), __builtin_ctzll(any));
        if (__builtin_amdgcn_ballot_w64((e0 && c0 != cu) || (e1 && c1 != cu))) return INT_MIN;
        return __builtin_amdgcn_readfirstlane(s_delta[cu]);              // (a scalar: thirty-two of them would fill as many vector registers)
    };

    // 1. the diagonal of every column (uniform), 2. the triples, greedily from the left, and their requests -- all of them in flight
    // before the first product (a triple that starts at column j keeps its registers in slot j / 3: disjoint runs of three columns start
    // in different aligned groups of three), 3. the products in column order; columns outside a triple gather there.
    int dg[MAXW];
#pragma unroll
    for (int j = 0; j < MAXW; ++j) dg[j] = j < w ? diagonal_of(j) : INT_MIN;
    V2 Pk[(MAXW + 2) / 3];
    V Ek[(MAXW + 2) / 3];
    unsigned starts = 0;                                 // bit j: a triple starts at column j (uniform)
    {
        int busy = 0;
#pragma unroll
        for (int j = 0; j + 2 < MAXW; ++j) {
            const long long dc = (long long)dg[j] + 1;
            const bool triple = busy == 0 && j + 2 < w && dg[j] != INT_MIN && dg[j + 1] != INT_MIN && dg[j + 2] != INT_MIN
                             && (long long)dg[j + 1] == dc && (long long)dg[j + 2] == dc + 1 && r_lo + dc - 1 >= 0 && r_hi + dc + 1 <= x_last;
            if (triple) {
                __builtin_memcpy(&Pk[j / 3], x + (i + dc), sizeof(V2));                 // (4-byte alignment is enough for the wide load)
                Ek[j / 3] = x[lane == 63 ? i + dc + 2 : r_lo + dc - 1];
                starts |= 1u << j;
                busy = 3;
            }
            busy = busy > 0 ? busy - 1 : 0;
        }
    }
    V sum[2] = {V(0), V(0)};
    V xa[2] = {V(0), V(0)}, xb[2] = {V(0), V(0)};
    int pend = 0;                                        // columns of the current triple still to come (uniform)
#pragma unroll
    for (int j = 0; j < MAXW; ++j) {
        if (j >= w) continue;                            // uniform (no break: the loop must unroll, c[] and vc[] are registers)
        const int sh = 16 * (j & 1);
        const unsigned c0 = (c[j >> 1] >> sh) & 255u, c1 = (c[j >> 1] >> (sh + 8)) & 255u;
        const bool e0 = c0 < S8_FIRST_PAD, e1 = c1 < S8_FIRST_PAD;
        V x0 = V(0), x1 = V(0);
        if (pend == 2) { x0 = xa[0]; x1 = xa[1]; pend = 1; }
        else if (pend == 1) { x0 = xb[0]; x1 = xb[1]; pend = 0; }
        else if ((starts >> j) & 1u) {
            const V2 P = Pk[j / 3]; const V E = Ek[j / 3];
            x0 = shift_from_lower_lane(P.y, E); x1 = P.x;                               // diagonal dc - 1: x[i + dc - 1], x[i + dc]
            xa[0] = P.x; xa[1] = P.y;                                                   // diagonal dc
            xb[0] = P.y; xb[1] = shift_from_upper_lane(P.x, E);                         // diagonal dc + 1: x[i + dc + 1], x[i + dc + 2]
            pend = 2;
        } else {
            if (e0) x0 = x[i + s_delta[c0]];
            if (e1) x1 = x[i + 1 + s_delta[c1]];
        }
        if (e0) sum[0] += s_value[(vc[j >> 1] >> sh) & 255u] * x0;
        if (e1) sum[1] += s_value[(vc[j >> 1] >> (sh + 8)) & 255u] * x1;
    }
    if (csr_ptr) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (i + q < n)
                for (int j = csr_ptr[i + q], e = csr_ptr[i + q + 1]; j < e; ++j) sum[q] += csr_val[j] * x[csr_col[j]];
    }
    store_pair<V>(n, i, alpha, append, sum, y, trav);
}

