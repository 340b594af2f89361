// Round-4 prototype (diagnostic, not a product path): the "plane march" data flow for the 512-wide 7-point grid, and hand
// copy kernels in the same geometries.  A lane owns rows 2t, 2t+1 of TY consecutive grid lines and marches through the
// planes: the +-n and +-n^2 neighbours of its rows are ITS OWN earlier / later loads (registers), the +-1 neighbours come
// from the adjacent lanes (DPP wave shifts; the two values beyond a wave's 128 rows are scalar loads).  No LDS, no barrier.
// Compiled by hiprtc from tools/r04_pm_proto.py; `hipcc -c` of this file is the register / spill check.
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#pragma clang fp contract(off)   // products and sums stay separate: every bit must match the CSR loop
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double shr1(double v, double edge) {       // lane i <- lane i - 1, lane 0 <- edge
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shl1(double v, double edge) {       // lane i <- lane i + 1, lane 63 <- edge
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x130, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double masked_hi(double v, unsigned long long lanes) {
    unsigned hi = (unsigned)__double2hiint(v), rhi;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(rhi) : "v"(hi), "s"(lanes));
    return __hiloint2double((int)rhi, __double2loint(v));
}

template <int TY, bool COMPUTE>
__device__ __forceinline__ void pm_impl(const double *__restrict__ x, double *__restrict__ y, int n, int LZ, int flags, double alpha)
{
    const int t = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned b = blockIdx.x, xcd = b & 7u, q = b >> 3;
    const int YT = n / TY, ytx = YT / 8, ZC = n / LZ;
    int zc, tyl;
    if (flags & 2) { tyl = q / ZC; zc = q - tyl * ZC; } else { zc = q / ytx; tyl = q - zc * ytx; }
    const int y0 = (int)(xcd * ytx + tyl) * TY;
    const bool down = (flags & 1) && (zc & 1);
    const int dz = down ? -1 : 1;
    int z = down ? (zc + 1) * LZ - 1 : zc * LZ;
    const long long nlines = (long long)n * n, N = nlines * 512;
    const double h2 = (double)(n - 1) * (double)(n - 1);

    auto line_of = [&](int zz, int l) -> long long {           // clamped: an out-of-range line is never referenced by an entry
        long long li = (long long)zz * n + (y0 - 1 + l);
        li = li < 0 ? 0 : li; li = li >= nlines ? nlines - 1 : li;
        return li;
    };
    auto ld = [&](int zz, int l) -> d2 {
        const double *p = x + line_of(zz, l) * 512;
        unsigned lb = 16u * (unsigned)t;
        return *reinterpret_cast<const d2 *>(reinterpret_cast<const char *>(p) + lb);
    };
    auto edge = [&](int zz, int l, int side) -> double {
        long long i = line_of(zz, l) * 512 + wv * 128 + (side ? 128 : -1);
        i = i < 0 ? 0 : i; i = i >= N ? N - 1 : i;
        return x[i];
    };

    // decoded "interior line" block: values and validity of the lane's two rows at the 7 diagonal positions
    double a[7][2]; unsigned long long m[7][2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = 2 * t + r; const bool xb = row == 0 || row == 511;
#pragma unroll
        for (int p = 0; p < 7; ++p) {
            a[p][r] = xb ? (p == 3 ? 1.0 : 0.0) : (p == 3 ? 6 * h2 : -h2);
            m[p][r] = __builtin_amdgcn_ballot_w64(!xb || p == 3);
        }
    }

    d2 prev[TY], cur[TY + 2], nxt[TY], ph[2], pc[TY];
    double eL[TY], eR[TY], peL[TY], peR[TY];
#pragma unroll
    for (int l = 0; l < TY; ++l) { prev[l] = ld(z - dz, l + 1); nxt[l] = ld(z + dz, l + 1); pc[l] = ld(z + 2 * dz, l + 1); }
#pragma unroll
    for (int l = 0; l < TY + 2; ++l) cur[l] = ld(z, l);
    ph[0] = ld(z + dz, 0); ph[1] = ld(z + dz, TY + 1);
#pragma unroll
    for (int l = 0; l < TY; ++l) { eL[l] = edge(z, l + 1, 0); eR[l] = edge(z, l + 1, 1); peL[l] = edge(z + dz, l + 1, 0); peR[l] = edge(z + dz, l + 1, 1); }

    for (int k = 0; k < LZ; ++k) {
        // ---- plane z ----
#pragma unroll
        for (int l = 0; l < TY; ++l) {
            const int yl = y0 + l;
            const bool ident = yl == 0 || yl == n - 1 || z == 0 || z == n - 1;     // uniform
            const d2 c = cur[l + 1];
            d2 o;
            if (!COMPUTE) { o.x = c.x + prev[l].x + nxt[l].x + cur[l].x + cur[l + 2].x + eL[l]; o.y = c.y + prev[l].y + nxt[l].y + cur[l].y + cur[l + 2].y + eR[l]; }
            else if (ident) { double s0 = 0, s1 = 0; s0 += 1.0 * c.x; s1 += 1.0 * c.y; o.x = alpha * s0; o.y = alpha * s1; }
            else {
                const double xs0[7] = {prev[l].x, cur[l].x, shr1(c.y, eL[l]), c.x, c.y, cur[l + 2].x, nxt[l].x};
                const double xs1[7] = {prev[l].y, cur[l].y, c.x, c.y, shl1(c.x, eR[l]), cur[l + 2].y, nxt[l].y};
                double s0 = 0, s1 = 0;
#pragma unroll
                for (int p = 0; p < 7; ++p) { s0 += a[p][0] * masked_hi(xs0[p], m[p][0]); s1 += a[p][1] * masked_hi(xs1[p], m[p][1]); }
                o.x = alpha * s0; o.y = alpha * s1;
            }
            double *yp = y + ((long long)z * n + yl) * 512;
            __builtin_nontemporal_store(o, reinterpret_cast<d2 *>(reinterpret_cast<char *>(yp) + 16u * (unsigned)t));
        }
        // ---- rotate; request the halos of plane z + 2 dz and the centre lines of plane z + 3 dz ----
#pragma unroll
        for (int l = 0; l < TY; ++l) { prev[l] = cur[l + 1]; cur[l + 1] = nxt[l]; nxt[l] = pc[l]; eL[l] = peL[l]; eR[l] = peR[l]; }
        cur[0] = ph[0]; cur[TY + 1] = ph[1];
        z += dz;
        if (k + 1 < LZ) {
            ph[0] = ld(z + dz, 0); ph[1] = ld(z + dz, TY + 1);
#pragma unroll
            for (int l = 0; l < TY; ++l) { pc[l] = ld(z + 2 * dz, l + 1); peL[l] = edge(z + dz, l + 1, 0); peR[l] = edge(z + dz, l + 1, 1); }
        }
    }
}

extern "C" __global__ __launch_bounds__(256) void pm2(const double *x, double *y, int n, int LZ, int flags, double alpha) { pm_impl<2, true>(x, y, n, LZ, flags, alpha); }
extern "C" __global__ __launch_bounds__(256) void pm4(const double *x, double *y, int n, int LZ, int flags, double alpha) { pm_impl<4, true>(x, y, n, LZ, flags, alpha); }
extern "C" __global__ __launch_bounds__(256) void pm1(const double *x, double *y, int n, int LZ, int flags, double alpha) { pm_impl<1, true>(x, y, n, LZ, flags, alpha); }
extern "C" __global__ __launch_bounds__(256) void pm2_nocompute(const double *x, double *y, int n, int LZ, int flags, double alpha) { pm_impl<2, false>(x, y, n, LZ, flags, alpha); }
extern "C" __global__ __launch_bounds__(256) void pm4_nocompute(const double *x, double *y, int n, int LZ, int flags, double alpha) { pm_impl<4, false>(x, y, n, LZ, flags, alpha); }

// ---- hand copy kernels (1 read : 1 write of N doubles) ----
// one 16-byte pair per lane, no loop
extern "C" __global__ __launch_bounds__(256) void copy_1(const double *__restrict__ x, double *__restrict__ y, long long npairs, int nt) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < npairs) { d2 v = reinterpret_cast<const d2 *>(x)[i]; if (nt) __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(y) + i); else reinterpret_cast<d2 *>(y)[i] = v; }
}
// U pairs per lane, all loads first
template <int U, bool NTL, bool NTS>
__device__ __forceinline__ void copy_u(const double *__restrict__ x, double *__restrict__ y, long long npairs) {
    const long long base = (long long)blockIdx.x * (256 * U) + threadIdx.x;
    d2 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) { const long long i = base + k * 256; if (i < npairs) v[k] = NTL ? __builtin_nontemporal_load(reinterpret_cast<const d2 *>(x) + i) : reinterpret_cast<const d2 *>(x)[i]; }
#pragma unroll
    for (int k = 0; k < U; ++k) { const long long i = base + k * 256; if (i < npairs) { if (NTS) __builtin_nontemporal_store(v[k], reinterpret_cast<d2 *>(y) + i); else reinterpret_cast<d2 *>(y)[i] = v[k]; } }
}
extern "C" __global__ __launch_bounds__(256) void copy_u2(const double *x, double *y, long long np) { copy_u<2, false, true>(x, y, np); }
extern "C" __global__ __launch_bounds__(256) void copy_u4(const double *x, double *y, long long np) { copy_u<4, false, true>(x, y, np); }
extern "C" __global__ __launch_bounds__(256) void copy_u8(const double *x, double *y, long long np) { copy_u<8, false, true>(x, y, np); }
extern "C" __global__ __launch_bounds__(256) void copy_u4_plain(const double *x, double *y, long long np) { copy_u<4, false, false>(x, y, np); }
extern "C" __global__ __launch_bounds__(256) void copy_u4_ntl(const double *x, double *y, long long np) { copy_u<4, true, true>(x, y, np); }
// grid-stride (resident grid)
extern "C" __global__ __launch_bounds__(256) void copy_gs(const double *__restrict__ x, double *__restrict__ y, long long npairs, int nt) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x, g = (long long)gridDim.x * 256; i < npairs; i += g) {
        d2 v = reinterpret_cast<const d2 *>(x)[i];
        if (nt) __builtin_nontemporal_store(v, reinterpret_cast<d2 *>(y) + i); else reinterpret_cast<d2 *>(y)[i] = v;
    }
}
// the march product's geometry: a workgroup walks `run` consecutive 512-row slices (one pair per lane and slice, the next
// slice requested before this one is stored, a barrier per slice), runs dealt to XCDs in strips of 64 slices per plane
extern "C" __global__ __launch_bounds__(256) void copy_march(const double *__restrict__ x, double *__restrict__ y, long long nslices, int run, int barrier) {
    const unsigned mb = blockIdx.x, k = mb & 7u, qq = mb >> 3;
    const unsigned chunk = 64, planes = 512, plane_blocks = 512, rpc = chunk / (unsigned)run;
    const unsigned r = qq / rpc, ri = qq - r * rpc;
    const unsigned tile = r / planes, p = r - tile * planes;
    const long long l = (long long)tile * (8 * chunk) + k * chunk + ri * (unsigned)run;
    const long long first = (long long)p * plane_blocks + l;
    if (first >= nslices) return;
    const d2 *xs = reinterpret_cast<const d2 *>(x) + first * 256 + threadIdx.x;
    d2 *ys = reinterpret_cast<d2 *>(y) + first * 256 + threadIdx.x;
    d2 c = xs[0];
    for (int s = 0; s < run; ++s) {
        d2 nx = c;
        if (s + 1 < run) nx = xs[(s + 1) * 256];
        __builtin_nontemporal_store(c, ys + s * 256);
        c = nx;
        if (barrier) __syncthreads();
    }
}
// the plane-march geometry without the halo: TY centre lines per plane step, 1 : 1
template <int TY>
__device__ __forceinline__ void copy_pm_impl(const double *__restrict__ x, double *__restrict__ y, int n, int LZ) {
    const unsigned b = blockIdx.x, xcd = b & 7u, q = b >> 3;
    const int YT = n / TY, ytx = YT / 8;
    const int zc = q / ytx, tyl = q - zc * ytx;
    const int y0 = (int)(xcd * ytx + tyl) * TY;
    long long li = ((long long)zc * LZ * n + y0) * 256 + threadIdx.x;     // in pairs
    const d2 *xs = reinterpret_cast<const d2 *>(x); d2 *ys = reinterpret_cast<d2 *>(y);
    d2 c[TY];
#pragma unroll
    for (int l = 0; l < TY; ++l) c[l] = xs[li + l * 256];
    for (int k = 0; k < LZ; ++k) {
        d2 nx[TY];
        const long long lj = li + (long long)n * 256;
#pragma unroll
        for (int l = 0; l < TY; ++l) nx[l] = (k + 1 < LZ) ? xs[lj + l * 256] : c[l];
#pragma unroll
        for (int l = 0; l < TY; ++l) __builtin_nontemporal_store(c[l], ys + li + l * 256);
#pragma unroll
        for (int l = 0; l < TY; ++l) c[l] = nx[l];
        li = lj;
    }
}
extern "C" __global__ __launch_bounds__(256) void copy_pm2(const double *x, double *y, int n, int LZ) { copy_pm_impl<2>(x, y, n, LZ); }
extern "C" __global__ __launch_bounds__(256) void copy_pm4(const double *x, double *y, int n, int LZ) { copy_pm_impl<4>(x, y, n, LZ); }
