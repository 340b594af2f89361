#ifndef VEXCL_MULTI_ARRAY_HPP
#define VEXCL_MULTI_ARRAY_HPP
// vex::multi_array<T, NR>: a vector together with its n-D shape (reference: vexcl/multi_array.hpp:40-143).
//   x.vec()                      the whole array as a vector expression / lvalue
//   x(indices[i][_][range(..)])  a view; .vec() is the sliced vector (an lvalue), size<d>() the extent of the
//                                d-th dimension that was given as a range
//   reduce<SUM>(x, dims)         reduction along dimensions
// Single-device, like the slices it is made of.
#include "vector_view.hpp"

namespace vex {

template <typename T, size_t NR, class Dims> class multi_array_view;

template <typename T, size_t NR, size_t... D>
class multi_array_view<T, NR, std::index_sequence<D...>> {
    public:
        typedef std::integral_constant<size_t, sizeof...(D)> ndim;
        typedef vector_slice_view<T, NR> base_type;

        multi_array_view(vector<T> &data, const gslice<NR> &slice) : data(data), slice(slice) {}

        vector_slice_view<T, NR> vec() const { return slice(data); }

        template <size_t d>
        size_t size() const {
            static_assert(d < sizeof...(D), "Wrong dimension!");
            const size_t kept[] = {D..., 0};
            return slice.length[kept[d]];
        }
    private:
        vector<T> &data;
        gslice<NR> slice;
};

template <typename T, size_t NR>
class multi_array {
    public:
        typedef std::integral_constant<size_t, NR> ndim;
        typedef vector<T> base_type;

        multi_array(const std::vector<backend::command_queue> &queue, const extent_gen<NR> &ext)
            : data(queue, ext.size()), slice(ext)
        {
            precondition(queue.size() == 1, "Multi-arrays are restricted to single-device contexts");
        }

        const vector<T> &vec() const { return data; }
        vector<T> &vec() { return data; }

        template <class Dims>
        multi_array_view<T, NR, Dims> operator()(const index_gen<NR, Dims> &idx) const {
            return multi_array_view<T, NR, Dims>(const_cast<vector<T> &>(data), slice(idx));
        }

        template <size_t d>
        size_t size() const {
            static_assert(d < NR, "Wrong dimension!");
            return slice.dim[d];
        }

    private:
        vector<T> data;
    public:
        slicer<NR> slice;
};

/// Reduce a multi_array along the given dimensions.
template <class RDC, typename T, size_t NDIM, size_t NR>
reduced_view<detail::vector_ref<T>, NDIM, NR, RDC> reduce(const multi_array<T, NDIM> &m, const std::array<size_t, NR> &dims) {
    return reduced_view<detail::vector_ref<T>, NDIM, NR, RDC>(detail::vector_ref<T>(m.vec()), m.slice[_], dims);
}
template <class RDC, typename T, size_t NDIM>
reduced_view<detail::vector_ref<T>, NDIM, 1, RDC> reduce(const multi_array<T, NDIM> &m, size_t dim) {
    std::array<size_t, 1> d = {{dim}};
    return reduce<RDC>(m, d);
}

} // namespace vex
#endif
