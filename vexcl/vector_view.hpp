#ifndef VEXCL_VECTOR_VIEW_HPP
#define VEXCL_VECTOR_VIEW_HPP
// Views of vectors and vector expressions (reference: vexcl/vector_view.hpp):
//   vex::permutation(index_expr)(base)            :602-700   base[ index_expr ], an lvalue too
//   vex::gslice<N>, vex::slicer<N>, range, _      :264-600   n-D strided slices, lvalues on vectors
//   vex::reduce<RDC>(slice | extents, expr, dims) :700-1010  reduction along some dimensions, in-kernel loop
//   vex::reshape(expr, dst_dims, src_dims)        :1012-1130 dimension permutation
// Single-device only.  A slice of a VECTOR indexes it in place: `prm_k_base[ <index of idx> ]`.
// A slice of an EXPRESSION evaluates the expression at the mapped position by the
// same trick the reference's sparse products use: the inner expression is generated
// inside a block in which `idx` is re-declared as the mapped position, into a
// temporary, and the outer expression reads the temporary.
#include <array>
#include <cstdlib>
#include <functional>
#include <numeric>
#include "vector.hpp"
#include "reductor.hpp"

namespace vex {

template <class T, class Index>
struct permutation_view : detail::expression_base {
    typedef T value_type;
    const vector<T> *base; Index index;
    permutation_view(const vector<T> &b, const Index &i) : base(&b), index(i) {
        precondition(b.nparts() == 1, "permutation is only supported for single-device vectors");
    }
    void preamble(detail::gen_context &c) const { c.next(); index.preamble(c); }
    void params(detail::gen_context &c) const { c.src.template parameter<global_ptr<T>>(c.next()); index.params(c); }
    void local_init(detail::gen_context &c) const { c.next(); index.local_init(c); }
    void emit(detail::gen_context &c) const { c.src << c.next() << "[ "; index.emit(c); c.src << " ]"; }
    void set_args(detail::arg_context &a) const { a.next(); a.krn.push_arg((*base)(a.device)); index.set_args(a); }
    void get_props(detail::prop_context &p) const {
        index.get_props(p);
        if (p.queue.empty()) { p.queue = base->queue_list(); }
        if (p.part.empty() && p.size) { p.part = {0, p.size}; }
    }

    /// Address of the designated element, for atomics.
    const detail::address_expr<permutation_view> operator&() const { return detail::address_expr<permutation_view>(*this); }

#define VEXCL_VIEW_ASSIGN(op, tag)                                                                      \
    template <class Expr>                                                                               \
    typename std::enable_if<detail::is_operand<Expr>::value, const permutation_view &>::type            \
    operator op(const Expr &expr) const {                                                               \
        detail::prop_context p; get_props(p);                                                           \
        detail::assign_expression<assign::tag>(*this, detail::as_expr<Expr>::get(expr), p.queue, p.part); \
        return *this;                                                                                   \
    }
    VEXCL_VIEW_ASSIGN(=, SET) VEXCL_VIEW_ASSIGN(+=, ADD) VEXCL_VIEW_ASSIGN(-=, SUB)
    VEXCL_VIEW_ASSIGN(*=, MUL) VEXCL_VIEW_ASSIGN(/=, DIV)
#undef VEXCL_VIEW_ASSIGN
    /// view = view copies ELEMENTS.
    const permutation_view &operator=(const permutation_view &other) const {
        detail::prop_context p; get_props(p);
        detail::assign_expression<assign::SET>(*this, other, p.queue, p.part);
        return *this;
    }
    permutation_view(const permutation_view &) = default;
};

/// permutation(index)(EXPRESSION): the expression evaluated at position index[idx] (an rvalue).  The
/// expression is generated inside a block that re-declares `idx` as the mapped position.
template <class E, class Index>
struct expr_permutation_view : detail::expression_base {
    typedef typename E::value_type value_type;
    E expr; Index index;
    expr_permutation_view(const E &e, const Index &i) : expr(e), index(i) {}
    static std::string inner(const std::string &n) { return n + "_e"; }
    static std::string where(const std::string &n) { return n + "_i"; }
    void preamble(detail::gen_context &c) const {
        const std::string n = c.next();
        { detail::gen_context j(c, where(n)); index.preamble(j); }
        { detail::gen_context i(c, inner(n)); expr.preamble(i); }
    }
    void params(detail::gen_context &c) const {
        const std::string n = c.next();
        { detail::gen_context j(c, where(n)); index.params(j); }
        { detail::gen_context i(c, inner(n)); expr.params(i); }
    }
    void local_init(detail::gen_context &c) const {
        const std::string n = c.next();
        c.src.new_line() << type_name<value_type>() << " " << n << "_val;";
        c.src.open("{");
        { detail::gen_context j(c, where(n)); index.local_init(j); }
        c.src.new_line() << "const ulong vex_pos = ";
        { detail::gen_context j(c, where(n)); index.emit(j); }
        c.src << ";";
        c.src.open("{");
        c.src.new_line() << "const ulong idx = vex_pos;";
        { detail::gen_context i(c, inner(n)); expr.local_init(i); }
        c.src.new_line() << n << "_val = ";
        { detail::gen_context i(c, inner(n)); expr.emit(i); }
        c.src << ";";
        c.src.close("}");
        c.src.close("}");
    }
    void emit(detail::gen_context &c) const { c.src << c.next() << "_val"; }
    void set_args(detail::arg_context &a) const {
        a.next();
        { detail::arg_context j(a); index.set_args(j); }
        { detail::arg_context i(a); expr.set_args(i); }
    }
    void get_props(detail::prop_context &p) const {
        index.get_props(p);
        detail::prop_context q; expr.get_props(q);
        precondition(q.queue.size() <= 1, "permutation is only supported for single-device expressions");
        if (p.queue.empty()) p.queue = q.queue;
        if (p.part.empty() && p.size) p.part = {0, p.size};
    }
};

template <class Index>
struct permutation_builder {
    Index index;
    template <class T>
    permutation_view<T, Index> operator()(const vector<T> &base) const { check_overrun(base); return permutation_view<T, Index>(base, index); }
    template <class Expr>
    typename std::enable_if<detail::is_expr<Expr>::value && !detail::has_ref_type<Expr>::value,
        expr_permutation_view<detail::as_expr_t<Expr>, Index>>::type
    operator()(const Expr &e) const { check_overrun(e); return expr_permutation_view<detail::as_expr_t<Expr>, Index>(detail::as_expr<Expr>::get(e), index); }

    /// VEXCL_CHECK_SIZES > 1: the largest index (a device reduction) must exist in the base (vector_view.hpp:659-677).
    template <class Base> void check_overrun(const Base &base) const {
#if defined(VEXCL_CHECK_SIZES) && (VEXCL_CHECK_SIZES > 1)
        detail::prop_context b; detail::as_expr<Base>::get(base).get_props(b);
        precondition(!b.queue.empty(), "Can not permute stateless expression");
        detail::prop_context i; index.get_props(i);
        if (i.size == 0 && b.size == 0) return;
        Reductor<size_t, MAX> max(b.queue);
        precondition(max(index) < b.size, "Permutation will result in array overrun");
#else
        (void)base;
#endif
    }
};

/// permutation(index_expression)(vector)  (vector_view.hpp:602-700)
template <class Expr>
typename std::enable_if<detail::is_expr<Expr>::value, permutation_builder<detail::as_expr_t<Expr>>>::type
permutation(const Expr &index) { return permutation_builder<detail::as_expr_t<Expr>>{detail::as_expr<Expr>::get(index)}; }

// ---- n-D slices ---------------------------------------------------------------------------------
/// start:stride:stop along one dimension (vector_view.hpp:420-446); range() = everything.
struct range {
    ptrdiff_t start, stride, stop;
    range() : start(0), stride(0), stop(0) {}
    range(ptrdiff_t i) : start(i), stride(1), stop(i + 1) {}
    range(ptrdiff_t start, ptrdiff_t stride, ptrdiff_t stop) : start(start), stride(stride), stop(stop) {}
    range(ptrdiff_t start, ptrdiff_t stop) : start(start), stride(1), stop(stop) {}
    bool empty() const { return !(start || stride || stop); }
    size_t length() const { return (size_t)((std::abs(stop - start) + std::abs(stride) - 1) / std::abs(stride)); }
};
/// Placeholder for "all elements of this dimension".
static const range _;

/// extents[a][b][c] (vector_view.hpp:449-468).
template <size_t NDIM>
struct extent_gen {
    std::array<size_t, NDIM> dim;
    extent_gen() {}
    extent_gen<NDIM + 1> operator[](size_t new_dim) const {
        extent_gen<NDIM + 1> e;
        std::copy(dim.begin(), dim.end(), e.dim.begin());
        e.dim.back() = new_dim;
        return e;
    }
    size_t size() const { return std::accumulate(dim.begin(), dim.end(), size_t(1), std::multiplies<size_t>()); }
};
static const extent_gen<0> extents;

template <class T, class... Tail>
std::array<T, 1 + sizeof...(Tail)> make_array(T t, Tail... tail) {
    std::array<T, 1 + sizeof...(Tail)> a = {{t, static_cast<T>(tail)...}};
    return a;
}

/// indices[i][range][_]...: one index or range per dimension (vector_view.hpp:470-503).  `Dims` records
/// WHICH positions were given as ranges -- the dimensions a view of the result still has (multi_array.hpp).
template <size_t NR, class Dims = std::index_sequence<>> struct index_gen;
template <size_t NR, size_t... D>
struct index_gen<NR, std::index_sequence<D...>> {
    std::array<range, NR> ranges;
    index_gen() {}
    index_gen<NR + 1, std::index_sequence<D..., NR>> operator[](const range &r) const { return append<std::index_sequence<D..., NR>>(r); }
    index_gen<NR + 1, std::index_sequence<D...>> operator[](size_t i) const { return append<std::index_sequence<D...>>(range((ptrdiff_t)i)); }
    private:
        template <class Next> index_gen<NR + 1, Next> append(const range &r) const {
            index_gen<NR + 1, Next> idx;
            std::copy(ranges.begin(), ranges.end(), idx.ranges.begin());
            idx.ranges.back() = r;
            return idx;
        }
};
static const index_gen<0> indices;

template <class E, size_t NDIM> struct expr_slice_view;
template <class T, size_t NDIM> struct vector_slice_view;

/// Generalized slice: start, length and (signed) stride per dimension, last dimension fastest
/// (vector_view.hpp:264-410).
template <size_t NDIM>
struct gslice {
    static_assert(NDIM > 0, "a slice has at least one dimension");
    size_t start;
    std::array<size_t, NDIM> length;
    std::array<ptrdiff_t, NDIM> stride;

    gslice() : start(0) {}
    template <class T1, class T2>
    gslice(size_t start, const T1 *len, const T2 *str) : start(start) {
        for (size_t d = 0; d < NDIM; ++d) { length[d] = (size_t)len[d]; stride[d] = (ptrdiff_t)str[d]; }
    }
    template <class T1, class T2>
    gslice(size_t start, const std::array<T1, NDIM> &len, const std::array<T2, NDIM> &str) : gslice(start, len.data(), str.data()) {}

    size_t size() const { return std::accumulate(length.begin(), length.end(), size_t(1), std::multiplies<size_t>()); }

    // ---- code generation: geometry is passed as value parameters, never baked into the source
    void params(backend::source_generator &src, const std::string &name) const {
        src.template parameter<size_t>(name + "_start");
        for (size_t d = 0; d < NDIM; ++d) {
            src.template parameter<size_t>(name + "_length" + std::to_string(d));
            src.template parameter<ptrdiff_t>(name + "_stride" + std::to_string(d));
        }
    }
    void set_args(backend::kernel &k) const {
        k.push_arg(start);
        for (size_t d = 0; d < NDIM; ++d) { k.push_arg(length[d]); k.push_arg(stride[d]); }
    }
    /// Position of element `idx` of the slice, over the dimensions selected by `dims` (row-major, last fastest).
    static std::string position(const std::string &name, const std::string &idx, const std::vector<size_t> &dims, bool with_start = true) {
        std::string pos = with_start ? name + "_start" : std::string("0");
        std::string rem = idx;
        for (size_t k = dims.size(); k-- > 0;) {
            const std::string d = std::to_string(dims[k]);
            const std::string digit = k ? "( " + rem + " % " + name + "_length" + d + " )" : rem;
            pos += " + (long)" + digit + " * " + name + "_stride" + d;
            if (k) rem = "( " + rem + " / " + name + "_length" + d + " )";
        }
        return "(ulong)( " + pos + " )";
    }
    static std::vector<size_t> all_dims() { std::vector<size_t> d(NDIM); std::iota(d.begin(), d.end(), size_t(0)); return d; }

    /// slice(vector): indexes the vector in place, an lvalue; slice(expression): an rvalue.
    template <class T> vector_slice_view<T, NDIM> operator()(const vector<T> &base) const { check_overrun(base); return vector_slice_view<T, NDIM>(base, *this); }
    template <class Expr>
    typename std::enable_if<detail::is_expr<Expr>::value && !detail::has_ref_type<typename std::decay<Expr>::type>::value,
        expr_slice_view<detail::as_expr_t<Expr>, NDIM>>::type
    operator()(const Expr &e) const { check_overrun(e); return expr_slice_view<detail::as_expr_t<Expr>, NDIM>(detail::as_expr<Expr>::get(e), *this); }

    /// VEXCL_CHECK_SIZES > 0: the last element of the slice must exist in the sliced expression (vector_view.hpp:398-409).
    template <class Expr> void check_overrun(const Expr &e) const {
#if defined(VEXCL_CHECK_SIZES) && (VEXCL_CHECK_SIZES > 0)
        ptrdiff_t lo = (ptrdiff_t)start, hi = (ptrdiff_t)start;
        bool empty = false;
        for (size_t d = 0; d < NDIM; ++d) {
            if (!length[d]) empty = true;
            const ptrdiff_t span = ((ptrdiff_t)length[d] - 1) * stride[d];
            if (span > 0) hi += span; else lo += span;
        }
        detail::prop_context p; detail::as_expr<Expr>::get(e).get_props(p);
        precondition(empty || p.size == 0 || (lo >= 0 && (size_t)hi < p.size), "Slice will result in array overrun");
#else
        (void)e;
#endif
    }
};

template <class T, size_t NDIM>
struct vector_slice_view : detail::expression_base {
    typedef T value_type;
    const vector<T> *base; gslice<NDIM> slice;
    vector_slice_view(const vector<T> &b, const gslice<NDIM> &s) : base(&b), slice(s) {
        precondition(b.nparts() <= 1, "slices are only supported for single-device vectors");
    }
    void preamble(detail::gen_context &c) const { c.next(); }
    void params(detail::gen_context &c) const {
        const std::string n = c.next();
        c.src.template parameter<global_ptr<T>>(n + "_base");
        slice.params(c.src, n);
    }
    void local_init(detail::gen_context &c) const { c.next(); }
    void emit(detail::gen_context &c) const {
        const std::string n = c.next();
        c.src << n << "_base[ " << gslice<NDIM>::position(n, "idx", gslice<NDIM>::all_dims()) << " ]";
    }
    void set_args(detail::arg_context &a) const { a.next(); a.krn.push_arg((*base)(a.device)); slice.set_args(a.krn); }
    void get_props(detail::prop_context &p) const {
        if (p.queue.empty()) p.queue = base->queue_list();
        if (p.size == 0) p.size = slice.size();
        if (p.part.empty()) p.part = {0, p.size};
    }
#define VEXCL_SLICE_ASSIGN(op, tag)                                                                     \
    template <class Expr>                                                                               \
    typename std::enable_if<detail::is_operand<Expr>::value, const vector_slice_view &>::type           \
    operator op(const Expr &expr) const {                                                               \
        std::vector<size_t> part = {0, slice.size()};                                                   \
        detail::assign_expression<assign::tag>(*this, detail::as_expr<Expr>::get(expr), base->queue_list(), part); \
        return *this;                                                                                   \
    }
    VEXCL_SLICE_ASSIGN(=, SET) VEXCL_SLICE_ASSIGN(+=, ADD) VEXCL_SLICE_ASSIGN(-=, SUB)
    VEXCL_SLICE_ASSIGN(*=, MUL) VEXCL_SLICE_ASSIGN(/=, DIV)
#undef VEXCL_SLICE_ASSIGN
    /// slice = slice copies ELEMENTS (y(indices[_][_][i]).vec() = x(indices[i][_][_]).vec(), tests/multi_array.cpp:63).
    const vector_slice_view &operator=(const vector_slice_view &other) const {
        std::vector<size_t> part = {0, slice.size()};
        detail::assign_expression<assign::SET>(*this, other, base->queue_list(), part);
        return *this;
    }
    vector_slice_view(const vector_slice_view &) = default;
};

template <class E, size_t NDIM>
struct expr_slice_view : detail::expression_base {
    typedef typename E::value_type value_type;
    E expr; gslice<NDIM> slice;
    expr_slice_view(const E &e, const gslice<NDIM> &s) : expr(e), slice(s) {}
    static std::string inner(const std::string &n) { return n + "_e"; }
    void preamble(detail::gen_context &c) const { const std::string n = c.next(); detail::gen_context i(c, inner(n)); expr.preamble(i); }
    void params(detail::gen_context &c) const {
        const std::string n = c.next();
        slice.params(c.src, n);
        detail::gen_context i(c, inner(n)); expr.params(i);
    }
    void local_init(detail::gen_context &c) const {
        const std::string n = c.next();
        c.src.new_line() << type_name<value_type>() << " " << n << "_val;";
        c.src.open("{");
        c.src.new_line() << "const ulong vex_pos = " << gslice<NDIM>::position(n, "idx", gslice<NDIM>::all_dims()) << ";";
        c.src.open("{");
        c.src.new_line() << "const ulong idx = vex_pos;";
        { detail::gen_context i(c, inner(n)); expr.local_init(i); }
        c.src.new_line() << n << "_val = ";
        { detail::gen_context i(c, inner(n)); expr.emit(i); }
        c.src << ";";
        c.src.close("}");
        c.src.close("}");
    }
    void emit(detail::gen_context &c) const { c.src << c.next() << "_val"; }
    void set_args(detail::arg_context &a) const { a.next(); slice.set_args(a.krn); detail::arg_context i(a); expr.set_args(i); }
    void get_props(detail::prop_context &p) const {
        detail::prop_context q; expr.get_props(q);
        precondition(q.queue.size() <= 1, "slices are only supported for single-device expressions");
        if (p.queue.empty()) p.queue = q.queue;
        if (p.size == 0) p.size = slice.size();
        if (p.part.empty()) p.part = {0, p.size};
    }
};

/// Shape of an n-D array stored row-major in a vector; slicer[range][range]...(x) (vector_view.hpp:511-600).
template <size_t NR>
struct slicer {
    std::array<size_t, NR> dim;
    std::array<size_t, NR> stride;
    template <class T> slicer(const std::array<T, NR> &d) { init(d.data()); }
    template <class T> slicer(const T *d) { init(d); }
    slicer(const extent_gen<NR> &e) { init(e.dim.data()); }

    template <size_t C>
    struct slice : gslice<NR> {
        const slicer &parent;
        slice(const slicer &p, const range &r) : parent(p) {          // C == 0
            this->start = 0;
            for (size_t d = 0; d < NR; ++d) { this->length[d] = p.dim[d]; this->stride[d] = (ptrdiff_t)p.stride[d]; }
            apply(0, r);
        }
        slice(const slice<(C > 0 ? C - 1 : 0)> &prev, const range &r, int) : gslice<NR>(prev), parent(prev.parent) { apply(C, r); }
        typename std::conditional<(C + 1 < NR), slice<C + 1>, void>::type
        operator[](const range &r) const {
            static_assert(C + 1 < NR, "too many indices for this slicer");
            return slice<C + 1>(*this, r.empty() ? range(0, (ptrdiff_t)parent.dim[C + 1]) : r, 0);
        }
        private:
            void apply(size_t d, const range &r) {
                this->start += (size_t)(r.start * (ptrdiff_t)parent.stride[d]);
                this->length[d] = r.length();
                this->stride[d] = r.stride * (ptrdiff_t)parent.stride[d];
            }
    };
    slice<0> operator[](const range &r) const { return slice<0>(*this, r.empty() ? range(0, (ptrdiff_t)dim[0]) : r); }
    /// slicer(indices[...][...]): the slice in one call (vector_view.hpp:532-547).
    template <class Dims>
    gslice<NR> operator()(const index_gen<NR, Dims> &idx) const {
        size_t start = 0;
        std::array<size_t, NR> len; std::array<ptrdiff_t, NR> str;
        for (size_t i = 0; i < NR; ++i) {
            const range r = idx.ranges[i].empty() ? range(0, (ptrdiff_t)dim[i]) : idx.ranges[i];
            start += (size_t)(r.start * (ptrdiff_t)stride[i]);
            len[i] = r.length();
            str[i] = r.stride * (ptrdiff_t)stride[i];
        }
        return gslice<NR>(start, len, str);
    }

    private:
        template <class T> void init(const T *d) {
            for (size_t i = 0; i < NR; ++i) dim[i] = (size_t)d[i];
            stride.back() = 1;
            for (size_t i = NR - 1; i-- > 0;) stride[i] = stride[i + 1] * dim[i + 1];
        }
};

// ---- reduction along dimensions of a slice (vector_view.hpp:700-1010) ------------------------------
namespace detail {
template <class RDC, class T> struct reduce_op;
template <class T> struct reduce_op<SUM, T> { static std::string init() { return literal(T()); } static std::string apply(const std::string &a, const std::string &b) { return a + " + " + b; } };
template <class T> struct reduce_op<MAX, T> { static std::string init() { return literal(std::numeric_limits<T>::lowest()); } static std::string apply(const std::string &a, const std::string &b) { return MAX::impl<T>::device(a, b); } };
template <class T> struct reduce_op<MIN, T> { static std::string init() { return literal(std::numeric_limits<T>::max()); } static std::string apply(const std::string &a, const std::string &b) { return MIN::impl<T>::device(a, b); } };
}

/// Element idx of the result runs over the kept dimensions (row-major); the reduced dimensions
/// are an in-kernel loop.  Nests: reduce<MAX>(s2[_], reduce<SUM>(s3[_], sin(x), 2), 1).
template <class E, size_t NDIM, size_t NR, class RDC>
struct reduced_view : detail::expression_base {
    typedef typename E::value_type value_type;
    E expr; gslice<NDIM> slice; std::array<size_t, NR> rdims;
    reduced_view(const E &e, const gslice<NDIM> &s, const std::array<size_t, NR> &d) : expr(e), slice(s), rdims(d) {
        for (size_t r : rdims) precondition(r < NDIM, "reduce: dimension out of range");
    }
    std::vector<size_t> kept() const {
        std::vector<size_t> k;
        for (size_t d = 0; d < NDIM; ++d) if (std::find(rdims.begin(), rdims.end(), d) == rdims.end()) k.push_back(d);
        return k;
    }
    size_t size() const { size_t n = 1; for (size_t d : kept()) n *= slice.length[d]; return n; }
    static std::string inner(const std::string &n) { return n + "_e"; }
    // WHICH dimensions are reduced is a run-time value while kernels are cached per expression type:
    // the source only knows "NDIM - NR kept dimensions, then NR reduced ones"; the geometry is passed
    // in that order (set_args), so one kernel serves every choice of dimensions.
    static const size_t NK = NDIM - NR;
    void preamble(detail::gen_context &c) const { const std::string n = c.next(); detail::gen_context i(c, inner(n)); expr.preamble(i); }
    void params(detail::gen_context &c) const {
        const std::string n = c.next();
        c.src.template parameter<size_t>(n + "_start");
        for (size_t k = 0; k < NK; ++k) { c.src.template parameter<size_t>(n + "_klen" + std::to_string(k)); c.src.template parameter<ptrdiff_t>(n + "_kstr" + std::to_string(k)); }
        for (size_t k = 0; k < NR; ++k) { c.src.template parameter<size_t>(n + "_rlen" + std::to_string(k)); c.src.template parameter<ptrdiff_t>(n + "_rstr" + std::to_string(k)); }
        detail::gen_context i(c, inner(n)); expr.params(i);
    }
    void local_init(detail::gen_context &c) const {
        const std::string n = c.next();
        typedef detail::reduce_op<RDC, value_type> op;
        c.src.new_line() << type_name<value_type>() << " " << n << "_val = " << op::init() << ";";
        c.src.open("{");
        {   // position of the first element of this output's run: row-major over the kept dimensions
            std::string pos = n + "_start", rem = "idx";
            for (size_t k = NK; k-- > 0;) {
                const std::string d = std::to_string(k);
                const std::string digit = k ? "( " + rem + " % " + n + "_klen" + d + " )" : rem;
                pos += " + (long)" + digit + " * " + n + "_kstr" + d;
                if (k) rem = "( " + rem + " / " + n + "_klen" + d + " )";
            }
            c.src.new_line() << "const ulong vex_base = (ulong)( " << pos << " );";
        }
        for (size_t k = 0; k < NR; ++k)
            c.src.new_line() << "for(ulong vex_r" << k << " = 0; vex_r" << k << " < " << n << "_rlen" << k << "; ++vex_r" << k << ")";
        c.src.open("{");
        c.src.new_line() << "const ulong vex_pos = (ulong)( (long)vex_base";
        for (size_t k = 0; k < NR; ++k) c.src << " + (long)vex_r" << k << " * " << n << "_rstr" << k;
        c.src << " );";
        c.src.open("{");
        c.src.new_line() << "const ulong idx = vex_pos;";
        { detail::gen_context i(c, inner(n)); expr.local_init(i); }
        c.src.new_line() << type_name<value_type>() << " vex_v = ";
        { detail::gen_context i(c, inner(n)); expr.emit(i); }
        c.src << ";";
        c.src.new_line() << n << "_val = " << op::apply(n + "_val", "vex_v") << ";";
        c.src.close("}");
        c.src.close("}");
        c.src.close("}");
    }
    void emit(detail::gen_context &c) const { c.src << c.next() << "_val"; }
    void set_args(detail::arg_context &a) const {
        a.next();
        a.krn.push_arg(slice.start);
        for (size_t d : kept()) { a.krn.push_arg(slice.length[d]); a.krn.push_arg(slice.stride[d]); }
        for (size_t d : rdims) { a.krn.push_arg(slice.length[d]); a.krn.push_arg(slice.stride[d]); }
        detail::arg_context i(a); expr.set_args(i);
    }
    void get_props(detail::prop_context &p) const {
        detail::prop_context q; expr.get_props(q);
        precondition(q.queue.size() <= 1, "reduce over a slice is only supported for single-device expressions");
        if (p.queue.empty()) p.queue = q.queue;
        if (p.size == 0) p.size = size();
        if (p.part.empty()) p.part = {0, p.size};
    }
};

template <class RDC, class Expr, size_t NDIM, size_t NR>
typename std::enable_if<detail::is_operand<Expr>::value, reduced_view<detail::as_expr_t<Expr>, NDIM, NR, RDC>>::type
reduce(const gslice<NDIM> &slice, const Expr &expr, const std::array<size_t, NR> &dims) {
    return reduced_view<detail::as_expr_t<Expr>, NDIM, NR, RDC>(detail::as_expr<Expr>::get(expr), slice, dims);
}
template <class RDC, class Expr, size_t NDIM>
typename std::enable_if<detail::is_operand<Expr>::value, reduced_view<detail::as_expr_t<Expr>, NDIM, 1, RDC>>::type
reduce(const gslice<NDIM> &slice, const Expr &expr, size_t dim) {
    std::array<size_t, 1> d = {{dim}};
    return reduce<RDC>(slice, expr, d);
}
template <class RDC, class Expr, size_t NDIM, size_t NR>
auto reduce(const extent_gen<NDIM> &ext, const Expr &expr, const std::array<size_t, NR> &dims)
    -> decltype(reduce<RDC>(gslice<NDIM>(), expr, dims)) { return reduce<RDC>(gslice<NDIM>(slicer<NDIM>(ext)[_]), expr, dims); }
template <class RDC, class Expr, size_t NDIM, size_t NR>
auto reduce(const extent_gen<NDIM> &ext, const Expr &expr, const extent_gen<NR> &dims)
    -> decltype(reduce<RDC>(gslice<NDIM>(), expr, dims.dim)) { return reduce<RDC>(gslice<NDIM>(slicer<NDIM>(ext)[_]), expr, dims.dim); }
template <class RDC, class Expr, size_t NDIM>
auto reduce(const extent_gen<NDIM> &ext, const Expr &expr, size_t dim)
    -> decltype(reduce<RDC>(gslice<NDIM>(), expr, dim)) { return reduce<RDC>(gslice<NDIM>(slicer<NDIM>(ext)[_]), expr, dim); }
/// reduce<SUM>(slice(x), dims): the slice and the vector it views.
template <class RDC, class T, size_t NDIM, size_t NR>
reduced_view<detail::vector_ref<T>, NDIM, NR, RDC> reduce(const vector_slice_view<T, NDIM> &v, const std::array<size_t, NR> &dims) {
    return reduced_view<detail::vector_ref<T>, NDIM, NR, RDC>(detail::vector_ref<T>(*v.base), v.slice, dims);
}
template <class RDC, class T, size_t NDIM>
reduced_view<detail::vector_ref<T>, NDIM, 1, RDC> reduce(const vector_slice_view<T, NDIM> &v, size_t dim) {
    std::array<size_t, 1> d = {{dim}};
    return reduce<RDC>(v, d);
}

/// reshape(expr, dst_dims, src_dims): an expression shaped `dst_dims` made from one shaped
/// `dst_dims[src_dims]` (row-major); `src_dims` are indices into `dst_dims`
/// (vector_view.hpp:1003-1124).  Destination dimensions that `src_dims` does not name are
/// broadcast (stride 0) -- `reshape(b, extents[N][N], extents[1])` repeats the vector b along
/// dimension 0.  A slice with permuted strides.
template <class Expr, size_t Nout, size_t Nin>
auto reshape(const Expr &expr, const std::array<size_t, Nout> &dst_dims, const std::array<size_t, Nin> &src_dims)
    -> decltype(gslice<Nout>()(expr))
{
    static_assert(Nin >= 1 && Nin <= Nout, "reshape: the source has more dimensions than the destination");
    std::array<size_t, Nin> src_stride;
    for (size_t k = 0; k < Nin; ++k) precondition(src_dims[k] < Nout, "reshape: bad source dimension");
    src_stride[Nin - 1] = 1;
    for (size_t k = Nin - 1; k-- > 0;) src_stride[k] = src_stride[k + 1] * dst_dims[src_dims[k + 1]];
    std::array<ptrdiff_t, Nout> stride;
    stride.fill(0);
    for (size_t k = 0; k < Nin; ++k) stride[src_dims[k]] = (ptrdiff_t)src_stride[k];
    return gslice<Nout>(0, dst_dims, stride)(expr);
}
template <class Expr, size_t Nout, size_t Nin>
auto reshape(const Expr &expr, const extent_gen<Nout> &dst_dims, const extent_gen<Nin> &src_dims)
    -> decltype(reshape(expr, dst_dims.dim, src_dims.dim)) { return reshape(expr, dst_dims.dim, src_dims.dim); }

} // namespace vex
#endif