#ifndef VEXCL_TENSORDOT_HPP
#define VEXCL_TENSORDOT_HPP
// vex::tensordot(slice(a), slice(b), axes_pairs(...)) -- contraction of two n-D arrays along
// pairs of axes (reference: vexcl/tensordot.hpp:54-396; numpy.tensordot semantics).
//
// A terminal of the fused kernel: output element idx runs row-major over the free
// dimensions of the left operand followed by the free dimensions of the right one; the
// contracted dimensions are an in-kernel loop nest accumulating lhs * rhs, each operand
// being an arbitrary expression evaluated at the mapped position (the block re-declares
// `idx`).  The geometry (lengths and strides, free dimensions first) is passed as value
// parameters, so one cached kernel serves every shape and every choice of axes of the
// same ranks.  Single-device, like every view.
#include <array>
#include <numeric>
#include "vector_view.hpp"

namespace vex {

template <class L, size_t LDIM, class R, size_t RDIM, size_t CDIM>
struct tensordot_expr : detail::expression_base {
    static_assert(CDIM <= LDIM && CDIM <= RDIM, "more contracted axes than dimensions");
    typedef typename detail::binary_result<detail::tag::multiplies, typename L::value_type, typename R::value_type>::type value_type;
    L lhs; R rhs;
    gslice<LDIM> ls; gslice<RDIM> rs;

    tensordot_expr(const L &l, const gslice<LDIM> &lsl, const R &r, const gslice<RDIM> &rsl,
                   const std::array<std::array<size_t, 2>, CDIM> &axes)
        : lhs(l), rhs(r), ls(lsl), rs(rsl)
    {
        for (size_t k = 0; k < CDIM; ++k) {
            precondition(axes[k][0] < LDIM && axes[k][1] < RDIM, "tensordot: axis out of range");
            precondition(ls.length[axes[k][0]] == rs.length[axes[k][1]], "Incompatible common dimensions in tensordot");
        }
        rearrange<0>(ls, axes);
        rearrange<1>(rs, axes);
    }

    size_t size() const {
        size_t n = 1;
        for (size_t d = 0; d < LDIM - CDIM; ++d) n *= ls.length[d];
        for (size_t d = 0; d < RDIM - CDIM; ++d) n *= rs.length[d];
        return n;
    }

    static std::string lname(const std::string &n) { return n + "_lhs"; }
    static std::string rname(const std::string &n) { return n + "_rhs"; }

    void preamble(detail::gen_context &c) const {
        const std::string n = c.next();
        { detail::gen_context i(c, lname(n)); lhs.preamble(i); }
        { detail::gen_context i(c, rname(n)); rhs.preamble(i); }
    }
    void params(detail::gen_context &c) const {
        const std::string n = c.next();
        { detail::gen_context i(c, lname(n)); lhs.params(i); }
        ls.params(c.src, lname(n));
        { detail::gen_context i(c, rname(n)); rhs.params(i); }
        rs.params(c.src, rname(n));
    }
    void local_init(detail::gen_context &c) const {
        const std::string n = c.next(), ln = lname(n), rn = rname(n);
        const std::string T = type_name<value_type>();
        auto &src = c.src;
        src.new_line() << T << " " << n << "_sum = 0;";
        src.open("{");
        src.new_line() << "ulong vex_lp0 = " << ln << "_start, vex_rp0 = " << rn << "_start;";
        src.open("{");
        src.new_line() << "ulong vex_pos = idx;";
        for (size_t i = RDIM - CDIM; i-- > 0;) {
            src.new_line() << "vex_rp0 += (long)( vex_pos % " << rn << "_length" << i << " ) * " << rn << "_stride" << i << "; "
                           << "vex_pos /= " << rn << "_length" << i << ";";
        }
        for (size_t i = LDIM - CDIM; i-- > 0;) {
            src.new_line() << "vex_lp0 += (long)( vex_pos % " << ln << "_length" << i << " ) * " << ln << "_stride" << i << "; "
                           << "vex_pos /= " << ln << "_length" << i << ";";
        }
        src.close("}");
        for (size_t i = 1, il = LDIM - CDIM, ir = RDIM - CDIM; i <= CDIM; ++i, ++il, ++ir) {
            src.new_line() << "for(ulong vex_c" << i << " = 0, vex_lp" << i << " = vex_lp" << i - 1 << ", vex_rp" << i << " = vex_rp" << i - 1 << "; "
                           << "vex_c" << i << " < " << ln << "_length" << il << "; ++vex_c" << i << ", "
                           << "vex_lp" << i << " += " << ln << "_stride" << il << ", vex_rp" << i << " += " << rn << "_stride" << ir << ")";
            src.open("{");
        }
        src.new_line() << T << " vex_prod;";
        src.new_line() << "const ulong vex_lpos = vex_lp" << CDIM << ", vex_rpos = vex_rp" << CDIM << ";";
        src.open("{");
        src.new_line() << "const ulong idx = vex_lpos;";
        { detail::gen_context i(c, ln); lhs.local_init(i); }
        src.new_line() << "vex_prod = ";
        { detail::gen_context i(c, ln); lhs.emit(i); }
        src << ";";
        src.close("}");
        src.open("{");
        src.new_line() << "const ulong idx = vex_rpos;";
        { detail::gen_context i(c, rn); rhs.local_init(i); }
        src.new_line() << "vex_prod *= ";
        { detail::gen_context i(c, rn); rhs.emit(i); }
        src << ";";
        src.close("}");
        src.new_line() << n << "_sum += vex_prod;";
        for (size_t i = 1; i <= CDIM; ++i) src.close("}");
        src.close("}");
    }
    void emit(detail::gen_context &c) const { c.src << c.next() << "_sum"; }
    void set_args(detail::arg_context &a) const {
        a.next();
        { detail::arg_context i(a); lhs.set_args(i); }
        ls.set_args(a.krn);
        { detail::arg_context i(a); rhs.set_args(i); }
        rs.set_args(a.krn);
    }
    void get_props(detail::prop_context &p) const {
        detail::prop_context ql, qr;
        lhs.get_props(ql); rhs.get_props(qr);          // the operands' sizes differ; only their queues matter
        precondition(ql.queue.size() <= 1 && qr.queue.size() <= 1, "tensordot is only supported for single-device expressions");
        if (p.queue.empty()) p.queue = ql.queue.empty() ? qr.queue : ql.queue;
        if (p.size == 0) p.size = size();
        if (p.part.empty()) p.part = {0, p.size};
    }

    private:
        /// Free axes first (in their original order), contracted axes last (in the order of the pairs).
        template <size_t SIDE, size_t DIM>
        void rearrange(gslice<DIM> &s, const std::array<std::array<size_t, 2>, CDIM> &axes) {
            bool common[DIM];
            for (size_t i = 0; i < DIM; ++i) common[i] = false;
            for (size_t i = 0; i < CDIM; ++i) common[axes[i][SIDE]] = true;
            std::array<size_t, DIM> len; std::array<ptrdiff_t, DIM> str;
            size_t j = 0;
            for (size_t i = 0; i < DIM; ++i) if (!common[i]) { len[j] = s.length[i]; str[j] = s.stride[i]; ++j; }
            for (size_t i = 0; i < CDIM; ++i, ++j) { len[j] = s.length[axes[i][SIDE]]; str[j] = s.stride[axes[i][SIDE]]; }
            s.length = len; s.stride = str;
        }
};

/// axes_pairs(a0, b0, a1, b1, ...): axis a_k of the left operand is contracted with axis b_k of the right one.
template <class... Args>
std::array<std::array<size_t, 2>, sizeof...(Args) / 2> axes_pairs(Args... args) {
    static_assert(sizeof...(Args) % 2 == 0, "Odd number of arguments in axes_pairs");
    const size_t flat[] = {static_cast<size_t>(args)..., 0};
    std::array<std::array<size_t, 2>, sizeof...(Args) / 2> a;
    for (size_t k = 0; k < sizeof...(Args) / 2; ++k) { a[k][0] = flat[2 * k]; a[k][1] = flat[2 * k + 1]; }
    return a;
}

namespace detail {
template <class T, size_t N> vector_ref<T> sliced_operand(const vector_slice_view<T, N> &v) { return vector_ref<T>(*v.base); }
template <class E, size_t N> const E &sliced_operand(const expr_slice_view<E, N> &v) { return v.expr; }
template <class V> struct sliced_traits;
template <class T, size_t N> struct sliced_traits<vector_slice_view<T, N>> { typedef vector_ref<T> expr; static const size_t dim = N; };
template <class E, size_t N> struct sliced_traits<expr_slice_view<E, N>> { typedef E expr; static const size_t dim = N; };
}

/// Tensor dot product of two sliced expressions along the given pairs of axes.
template <class LV, class RV, size_t CDIM>
tensordot_expr<typename detail::sliced_traits<LV>::expr, detail::sliced_traits<LV>::dim,
               typename detail::sliced_traits<RV>::expr, detail::sliced_traits<RV>::dim, CDIM>
tensordot(const LV &lhs, const RV &rhs, const std::array<std::array<size_t, 2>, CDIM> &common_axes) {
    return tensordot_expr<typename detail::sliced_traits<LV>::expr, detail::sliced_traits<LV>::dim,
                          typename detail::sliced_traits<RV>::expr, detail::sliced_traits<RV>::dim, CDIM>(
            detail::sliced_operand(lhs), lhs.slice, detail::sliced_operand(rhs), rhs.slice, common_axes);
}

} // namespace vex
#endif
