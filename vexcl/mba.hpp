#ifndef VEXCL_MBA_HPP
#define VEXCL_MBA_HPP
// vex::mba<NDIM, real>: scattered data interpolation with multilevel B-splines
// (S. Lee, G. Wolberg, S. Y. Shin, 1997; reference: vexcl/mba.hpp:155-480 for the fit,
// :484-795 for the kernel terminal).
//
//   vex::mba<2> surf(ctx, lo, hi, points, values, grid [, levels, tol]);
//   z = sin(surf(x, y));          // evaluated inside the fused kernel
//
// The fit is a hierarchy of control lattices, each fitted to the residual of the previous ones
// and folded into the next finer lattice by B-spline refinement.  The reference fits on the
// host; here the data is uploaded once and every level is fitted in HBM (vexhip_mba_fit:
// one lane per data point with hardware floating-point atomics into the lattice, refinement
// gathered per fine node), for 1 to 3 dimensions; more dimensions, or VEXCL_MBA_HOST_FIT in the
// environment, take the host loop below (the same algorithm; the two agree to rounding).
// What the kernels of user expressions see is the final lattice, one read-only copy per queue,
// and the evaluation: a device function doing the 4^NDIM-point tensor-product cubic B-spline
// sum around the cell of (x0, x1, ...).  The coordinate operands are arbitrary expressions.
#include <array>
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <numeric>
#include <vector>
#include "operations.hpp"
#include "vector.hpp"

namespace vex {

namespace detail {

/// N^M at compile time.
template <size_t N, size_t M> struct power : std::integral_constant<size_t, N * power<N, M - 1>::value> {};
template <size_t N> struct power<N, 0> : std::integral_constant<size_t, 1> {};

/// Digits of `flat` in a mixed-radix system (last digit fastest).
template <size_t M>
inline std::array<size_t, M> unflatten(size_t flat, const std::array<size_t, M> &radix) {
    std::array<size_t, M> d;
    for (size_t k = M; k-- > 0;) { d[k] = flat % radix[k]; flat /= radix[k]; }
    return d;
}
template <size_t M>
inline std::array<size_t, M> unflatten(size_t flat, size_t radix) {
    std::array<size_t, M> r; r.fill(radix);
    return unflatten<M>(flat, r);
}

template <class MBA, class... X> struct mba_interp;

} // namespace detail

template <size_t NDIM, typename real = double>
class mba {
    public:
        typedef real value_type;
        typedef std::array<real, NDIM> point;
        typedef std::array<size_t, NDIM> index;
        static const size_t ndim = NDIM;

        std::vector<backend::command_queue> queue;
        std::vector<backend::device_vector<real>> phi;      // the control lattice, one copy per queue
        point xmin, hinv;
        index n, stride;

        /// `cmin` / `cmax`: the domain; `coo` / `val`: the data; `grid`: initial control grid.  The
        /// hierarchy has at most `levels` levels and stops once the residual falls below `tol`.
        mba(const std::vector<backend::command_queue> &queue, const point &cmin, const point &cmax,
            const std::vector<point> &coo, std::vector<real> val, std::array<size_t, NDIM> grid,
            size_t levels = 8, real tol = 1e-8)
            : queue(queue)
        { init(cmin, cmax, coo.begin(), coo.end(), val.begin(), grid, levels, tol); }

        template <class CooIter, class ValIter>
        mba(const std::vector<backend::command_queue> &queue, const point &cmin, const point &cmax,
            CooIter coo_begin, CooIter coo_end, ValIter val_begin, std::array<size_t, NDIM> grid,
            size_t levels = 8, real tol = 1e-8)
            : queue(queue)
        {
            // the residuals are updated in place: work on a copy of the values
            std::vector<real> val(val_begin, val_begin + (coo_end - coo_begin));
            init(cmin, cmax, coo_begin, coo_end, val.begin(), grid, levels, tol);
        }

        /// Interpolated values at the given coordinates (one expression per dimension).
        template <class... Expr>
        detail::mba_interp<mba, detail::as_expr_t<Expr>...> operator()(const Expr &...expr) const {
            static_assert(sizeof...(Expr) == NDIM, "Wrong number of parameters");
            return detail::mba_interp<mba, detail::as_expr_t<Expr>...>(*this, detail::as_expr<Expr>::get(expr)...);
        }

        /// The four cubic B-spline basis functions on [0, 1).
        static real B(size_t k, real t) {
            switch (k) {
                case 0:  return (t * (t * (-t + 3) - 3) + 1) / 6;
                case 1:  return (t * t * (3 * t - 6) + 4) / 6;
                case 2:  return (t * (t * (-3 * t + 3) + 3) + 1) / 6;
                default: return t * t * t / 6;
            }
        }

    private:
        /// One level of the hierarchy: a control lattice fitted to (coo, val).
        struct lattice {
            point xmin, hinv;
            index n, stride;
            std::vector<real> phi;

            template <class CooIter, class ValIter>
            lattice(const point &cmin, const point &cmax, const index &grid, CooIter coo_begin, CooIter coo_end, ValIter val_begin)
                : xmin(cmin), n(grid)
            {
                // one extra control point on each side of the domain
                for (size_t d = 0; d < NDIM; ++d) {
                    hinv[d] = (grid[d] - 1) / (cmax[d] - cmin[d]);
                    xmin[d] -= 1 / hinv[d];
                    n[d] += 2;
                }
                stride[NDIM - 1] = 1;
                for (size_t d = NDIM - 1; d-- > 0;) stride[d] = stride[d + 1] * n[d + 1];
                const size_t total = n[0] * stride[0];
                std::vector<real> num(total, 0), den(total, 0);

                const size_t NW = detail::power<4, NDIM>::value;
                std::array<real, NW> w;
                ValIter v = val_begin;
                for (CooIter p = coo_begin; p != coo_end; ++p, ++v) {
                    if (!inside(cmin, cmax, *p)) continue;
                    index cell; point s;
                    locate(*p, cell, s);
                    real sw2 = 0;
                    for (size_t t = 0; t < NW; ++t) {
                        const auto d = detail::unflatten<NDIM>(t, 4);
                        real prod = 1;
                        for (size_t k = 0; k < NDIM; ++k) prod *= B(d[k], s[k]);
                        w[t] = prod;
                        sw2 += prod * prod;
                    }
                    // every data point proposes (*v) w / sum w^2 for its 4^NDIM control points; proposals are
                    // averaged with weights w^2
                    for (size_t t = 0; t < NW; ++t) {
                        const auto d = detail::unflatten<NDIM>(t, 4);
                        size_t at = 0;
                        for (size_t k = 0; k < NDIM; ++k) at += (cell[k] + d[k]) * stride[k];
                        const real w2 = w[t] * w[t];
                        num[at] += w2 * ((*v) * w[t] / sw2);
                        den[at] += w2;
                    }
                }
                phi.resize(total);
                for (size_t i = 0; i < total; ++i) phi[i] = std::fabs(den[i]) < 1e-32 ? real(0) : num[i] / den[i];
            }

            /// Value of this level's spline at p.
            real operator()(const point &p) const {
                index cell; point s;
                locate(p, cell, s);
                real f = 0;
                for (size_t t = 0; t < detail::power<4, NDIM>::value; ++t) {
                    const auto d = detail::unflatten<NDIM>(t, 4);
                    real wgt = 1; size_t at = 0; bool in = true;
                    for (size_t k = 0; k < NDIM; ++k) {
                        wgt *= B(d[k], s[k]);
                        const size_t j = cell[k] + d[k];
                        if (j >= n[k]) { in = false; break; }
                        at += j * stride[k];
                    }
                    if (in) f += wgt * phi[at];
                }
                return f;
            }

            /// Replace the data by its residual with respect to this level; returns the squared residual.
            template <class CooIter, class ValIter>
            real take_residual(CooIter coo_begin, CooIter coo_end, ValIter val_begin) const {
                real res = 0;
                ValIter v = val_begin;
                for (CooIter c = coo_begin; c != coo_end; ++c, ++v) { *v -= (*this)(*c); res += (*v) * (*v); }
                return res;
            }

            /// Add the coarser lattice `r`, refined onto this one (cubic B-spline subdivision: masks 1/8 (1 4 6 4 1)).
            void add_refined(const lattice &r) {
                static const real mask[5] = {real(0.125), real(0.5), real(0.75), real(0.5), real(0.125)};
                const size_t rtotal = r.n[0] * r.stride[0];
                for (size_t flat = 0; flat < rtotal; ++flat) {
                    const auto i = detail::unflatten<NDIM>(flat, r.n);
                    const real f = r.phi[flat];
                    for (size_t t = 0; t < detail::power<5, NDIM>::value; ++t) {
                        const auto d = detail::unflatten<NDIM>(t, 5);
                        size_t at = 0; real c = 1; bool in = true;
                        for (size_t k = 0; k < NDIM; ++k) {
                            const size_t j = 2 * i[k] + d[k] - 3;       // wraps for negative positions: caught below
                            if (j >= n[k]) { in = false; break; }
                            at += j * stride[k];
                            c *= mask[d[k]];
                        }
                        if (in) phi[at] += f * c;
                    }
                }
            }

            private:
                void locate(const point &p, index &cell, point &s) const {
                    for (size_t d = 0; d < NDIM; ++d) {
                        const real u = (p[d] - xmin[d]) * hinv[d];
                        cell[d] = static_cast<size_t>(std::floor(u) - 1);
                        s[d] = u - std::floor(u);
                    }
                }
                static bool inside(const point &lo, const point &hi, const point &x) {
                    const real eps = real(1e-12);
                    for (size_t d = 0; d < NDIM; ++d)
                        if (x[d] - eps < lo[d] || x[d] + eps >= hi[d]) return false;
                    return true;
                }
        };

        /// The fit in HBM (vexhip_mba_fit: atomics-based accumulation, gathered refinement; 1 to 3 dimensions): the data
        /// is uploaded once, every level runs on the first queue's device, the host reads one residual per level.
        template <class CooIter, class ValIter>
        void fit_on_device(const point &cmin, const point &cmax, CooIter coo_begin, CooIter coo_end, ValIter val_begin,
                           const std::array<size_t, NDIM> &grid, size_t levels, real tol)
        {
            const size_t np = (size_t)(coo_end - coo_begin);
            std::vector<real> flat(np * NDIM), vals(np);
            { size_t i = 0; ValIter v = val_begin; for (CooIter c = coo_begin; c != coo_end; ++c, ++v, ++i) { for (size_t d = 0; d < NDIM; ++d) flat[i * NDIM + d] = (*c)[d]; vals[i] = *v; } }
            const backend::command_queue &q = queue[0];
            backend::device_vector<real> dcoo(q, flat.size(), flat.data()), dval(q, vals.size(), vals.data());
            double lo[NDIM], hi[NDIM], xm[NDIM], hv[NDIM]; size_t g[NDIM], nn[NDIM], st[NDIM];
            for (size_t d = 0; d < NDIM; ++d) { lo[d] = cmin[d]; hi[d] = cmax[d]; g[d] = grid[d]; }
            void *p = nullptr; size_t elems = 0;
            backend::check(vexhip_mba_fit(q.device_ordinal(), q.raw(), std::is_same<real, float>::value ? VEXHIP_F32 : VEXHIP_F64,
                    (int)NDIM, lo, hi, np ? dcoo.raw() : nullptr, np ? dval.raw() : nullptr, (int64_t)np, g, (int)levels, (double)tol,
                    xm, hv, nn, st, &p, &elems));
            for (size_t d = 0; d < NDIM; ++d) { xmin[d] = (real)xm[d]; hinv[d] = (real)hv[d]; n[d] = nn[d]; stride[d] = st[d]; }
            phi.reserve(queue.size());
            phi.push_back(backend::device_vector<real>::adopt(q, static_cast<real *>(p), elems));
            if (queue.size() > 1) {                      // one read-only copy per further queue
                std::vector<real> host(elems);
                phi[0].read(q, 0, elems, host.data(), true);
                for (size_t k = 1; k < queue.size(); ++k)
                    phi.push_back(backend::device_vector<real>(queue[k], elems, host.data(), backend::MEM_READ_ONLY));
            }
        }

        template <class CooIter, class ValIter>
        void init(const point &cmin, const point &cmax, CooIter coo_begin, CooIter coo_end, ValIter val_begin,
                  std::array<size_t, NDIM> grid, size_t levels, real tol)
        {
            for (size_t k = 0; k < NDIM; ++k) precondition(grid[k] > 1, "mba: the control grid needs at least 2 points per dimension");
            if (NDIM <= 3 && !queue.empty() && !std::getenv("VEXCL_MBA_HOST_FIT")) {
                fit_on_device(cmin, cmax, coo_begin, coo_end, val_begin, grid, levels, tol);
                return;
            }
            double res0 = 0;
            { ValIter v = val_begin; for (CooIter c = coo_begin; c != coo_end; ++c, ++v) res0 += (*v) * (*v); }

            std::unique_ptr<lattice> psi(new lattice(cmin, cmax, grid, coo_begin, coo_end, val_begin));
            double res = psi->take_residual(coo_begin, coo_end, val_begin);
            for (size_t k = 1; res > res0 * tol && k < levels; ++k) {
                for (size_t d = 0; d < NDIM; ++d) grid[d] = 2 * grid[d] - 1;
                std::unique_ptr<lattice> f(new lattice(cmin, cmax, grid, coo_begin, coo_end, val_begin));
                res = f->take_residual(coo_begin, coo_end, val_begin);
                f->add_refined(*psi);
                psi = std::move(f);
            }
            xmin = psi->xmin; hinv = psi->hinv; n = psi->n; stride = psi->stride;
            phi.reserve(queue.size());
            for (const auto &q : queue)
                phi.push_back(backend::device_vector<real>(q, psi->phi.size(), psi->phi.data(), backend::MEM_READ_ONLY));
        }
};

namespace detail {

/// surf(x0, x1, ...): a terminal whose value is the device function `<prm>_mba` applied to the coordinate
/// expressions (mba.hpp:484-795 of the reference).
template <class MBA, class... X>
struct mba_interp : expression_base {
    typedef typename MBA::value_type value_type;
    typedef value_type real;
    static const size_t NDIM = MBA::ndim;
    const MBA &cloud;
    std::tuple<X...> coord;
    mba_interp(const MBA &m, const X &...x) : cloud(m), coord(x...) {}

    static std::string xname(const std::string &n, size_t k) { return n + "_x" + std::to_string(k); }

    void preamble(gen_context &c) const {
        const std::string n = c.next();
        tuple_for_each(coord, [&](const auto &x, size_t k) { gen_context i(c, xname(n, k)); x.preamble(i); });
        auto &src = c.src;
        const std::string R = type_name<real>();
        src.template begin_function<real>(n + "_mba");
        src.begin_function_parameters();
        for (size_t k = 0; k < NDIM; ++k) src.template parameter<real>("x" + std::to_string(k));
        for (size_t k = 0; k < NDIM; ++k) {
            src.template parameter<real>("c" + std::to_string(k));
            src.template parameter<real>("h" + std::to_string(k));
            src.template parameter<size_t>("n" + std::to_string(k));
            src.template parameter<size_t>("m" + std::to_string(k));
        }
        src.template parameter<global_ptr<const real>>("phi");
        src.end_function_parameters();
        // per dimension: the cell and the four basis values at the position inside it, in scalars (no indexed arrays:
        // they would live in scratch memory)
        for (size_t k = 0; k < NDIM; ++k) {
            src.new_line() << R << " w" << k << "_0, w" << k << "_1, w" << k << "_2, w" << k << "_3; ulong i" << k << ";";
            src.open("{");
            src.new_line() << "const " << R << " u = (x" << k << " - c" << k << ") * h" << k << ";";
            src.new_line() << "const " << R << " fl = floor(u), t = u - fl;";
            src.new_line() << "i" << k << " = (ulong)(long)(fl - 1);";
            src.new_line() << "w" << k << "_0 = (t * (t * (-t + 3) - 3) + 1) / 6;";
            src.new_line() << "w" << k << "_1 = (t * t * (3 * t - 6) + 4) / 6;";
            src.new_line() << "w" << k << "_2 = (t * (t * (-3 * t + 3) + 3) + 1) / 6;";
            src.new_line() << "w" << k << "_3 = t * t * t / 6;";
            src.close("}");
        }
        src.new_line() << R << " f = 0;";
        // the 4^NDIM terms, written out; a lane whose whole stencil is inside the lattice (the usual case) takes the
        // branch-free sum
        src.new_line() << "if (";
        for (size_t k = 0; k < NDIM; ++k) src << (k ? " && " : "") << "i" << k << " + 3 < n" << k;
        src << ")";
        src.open("{");
        src.new_line() << "const ulong base = ";
        for (size_t k = 0; k < NDIM; ++k) src << (k ? " + " : "") << "i" << k << " * m" << k;
        src << ";";
        size_t combos = 1; for (size_t k = 0; k < NDIM; ++k) combos *= 4;
        for (size_t t = 0; t < combos; ++t) {
            const auto d = unflatten<NDIM>(t, 4);
            src.new_line() << "f += ";
            for (size_t k = 0; k < NDIM; ++k) src << "w" << k << "_" << d[k] << " * ";
            src << "phi[base";
            for (size_t k = 0; k < NDIM; ++k) if (d[k]) src << " + " << d[k] << " * m" << k;
            src << "];";
        }
        src.close("}");
        src.new_line() << "else";
        src.open("{");
        for (size_t t = 0; t < combos; ++t) {
            const auto d = unflatten<NDIM>(t, 4);
            src.new_line() << "if (";
            for (size_t k = 0; k < NDIM; ++k) src << (k ? " && " : "") << "i" << k << " + " << d[k] << " < n" << k;
            src << ") f += ";
            for (size_t k = 0; k < NDIM; ++k) src << "w" << k << "_" << d[k] << " * ";
            src << "phi[";
            for (size_t k = 0; k < NDIM; ++k) src << (k ? " + " : "") << "(i" << k << " + " << d[k] << ") * m" << k;
            src << "];";
        }
        src.close("}");
        src.new_line() << "return f;";
        src.end_function();
    }
    void params(gen_context &c) const {
        const std::string n = c.next();
        tuple_for_each(coord, [&](const auto &x, size_t k) { gen_context i(c, xname(n, k)); x.params(i); });
        for (size_t k = 0; k < NDIM; ++k) {
            c.src.template parameter<real>(n + "_c" + std::to_string(k));
            c.src.template parameter<real>(n + "_h" + std::to_string(k));
            c.src.template parameter<size_t>(n + "_n" + std::to_string(k));
            c.src.template parameter<size_t>(n + "_m" + std::to_string(k));
        }
        c.src.template parameter<global_ptr<const real>>(n + "_phi");
    }
    void local_init(gen_context &c) const {
        const std::string n = c.next();
        tuple_for_each(coord, [&](const auto &x, size_t k) { gen_context i(c, xname(n, k)); x.local_init(i); });
    }
    void emit(gen_context &c) const {
        const std::string n = c.next();
        c.src << n << "_mba( ";
        tuple_for_each(coord, [&](const auto &x, size_t k) { if (k) c.src << ", "; gen_context i(c, xname(n, k)); x.emit(i); });
        for (size_t k = 0; k < NDIM; ++k)
            c.src << ", " << n << "_c" << k << ", " << n << "_h" << k << ", " << n << "_n" << k << ", " << n << "_m" << k;
        c.src << ", " << n << "_phi )";
    }
    void set_args(arg_context &a) const {
        a.next();
        tuple_for_each(coord, [&](const auto &x, size_t) { arg_context i(a); x.set_args(i); });
        for (size_t k = 0; k < NDIM; ++k) {
            a.krn.push_arg(cloud.xmin[k]);
            a.krn.push_arg(cloud.hinv[k]);
            a.krn.push_arg(cloud.n[k]);
            a.krn.push_arg(cloud.stride[k]);
        }
        a.krn.push_arg(cloud.phi[a.device]);
    }
    void get_props(prop_context &p) const {
        tuple_for_each(coord, [&](const auto &x, size_t) { x.get_props(p); });
    }
};

} // namespace detail
} // namespace vex
#endif
