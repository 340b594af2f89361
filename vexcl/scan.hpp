#ifndef VEXCL_SCAN_HPP
#define VEXCL_SCAN_HPP
// vex::inclusive_scan / vex::exclusive_scan (reference: vexcl/scan.hpp:419-518;
// device algorithm :66-414).  Per device: libvexhip's reduce-then-scan kernels;
// across devices the carry of the preceding partitions is added with one fused
// elementwise kernel per device, as the reference does (:445-457, :489-506).
#include <numeric>
#include "vector.hpp"
#include "function.hpp"

namespace vex {

/// The default scan operator (scan.hpp:419-424 of the reference): the library's look-back /
/// reduce-then-scan kernels.  Any other operator -- a struct with VEX_DUAL_FUNCTOR, as in the
/// reference -- goes through the generated segmented-scan kernels of scan_by_key.hpp with one
/// segment (generic_scan below).
template <class T> struct plus {
    VEX_DUAL_FUNCTOR(T, (T, x)(T, y), return x + y;)
};

namespace detail {
    template <class T> struct prim_dtype;
    template <> struct prim_dtype<double> { static const int value = VEXHIP_F64; };
    template <> struct prim_dtype<float> { static const int value = VEXHIP_F32; };
    template <> struct prim_dtype<int> { static const int value = VEXHIP_I32; };
    template <> struct prim_dtype<unsigned> { static const int value = VEXHIP_U32; };
    template <> struct prim_dtype<long> { static const int value = VEXHIP_I64; };
    template <> struct prim_dtype<unsigned long> { static const int value = VEXHIP_U64; };
    template <> struct prim_dtype<long long> { static const int value = VEXHIP_I64; };
    template <> struct prim_dtype<unsigned long long> { static const int value = VEXHIP_U64; };

    template <class T>
    void scan_impl(const vector<T> &input, vector<T> &output, bool exclusive, T init) {
        precondition(input.size() == output.size() && input.nparts() == output.nparts(), "scan: incompatible vectors");
        const auto &queue = input.queue_list();
        const unsigned nd = static_cast<unsigned>(queue.size());
        std::vector<T> tail(nd, T());        // sum of each partition's input
        for (unsigned d = 0; d < nd; ++d) {
            size_t n = input.part_size(d);
            if (!n) continue;
            int dev = queue[d].device_ordinal();
            backend::device_vector<char> tmp = scratch_pool::instance().get(queue[d], 3, vexhip_scan_tmp_bytes(prim_dtype<T>::value, (int64_t)n));
            // multi-device: remember the last input element before an in-place scan overwrites it
            T last_in = T();
            if (nd > 1 && exclusive) input(d).read(queue[d], n - 1, 1, &last_in, true);
            T zero = T();
            backend::check(vexhip_scan(dev, queue[d].raw(), prim_dtype<T>::value, exclusive ? 1 : 0,
                        d == 0 ? &init : &zero, input(d).raw(), output(d).raw(), (int64_t)n, tmp.raw()));
            if (nd > 1) {
                T last_out; output(d).read(queue[d], n - 1, 1, &last_out, true);
                tail[d] = exclusive ? static_cast<T>(last_out - (d == 0 ? init : T()) + last_in) : last_out;
            }
        }
        if (nd > 1) {
            T carry = exclusive ? init : T();
            for (unsigned d = 0; d < nd; ++d) {
                if (d > 0 && input.part_size(d)) {
                    std::vector<backend::command_queue> q1(1, queue[d]);
                    vector<T> seg(queue[d], output(d));
                    seg += carry;
                }
                carry = static_cast<T>(carry + tail[d]);
            }
        }
    }
}

namespace detail {
    template <class T, class Oper>
    void generic_scan(const vector<T> &input, vector<T> &output, bool exclusive, T init, Oper oper);   // scan_by_key.hpp
}

/// output[i] = input[0] + ... + input[i]; in-place allowed (scan.hpp:461-469).
template <class T> void inclusive_scan(const vector<T> &input, vector<T> &output) {
    detail::scan_impl(input, output, false, T());
}
template <class T, class Oper> void inclusive_scan(const vector<T> &input, vector<T> &output, T init, Oper oper) {
    if constexpr (std::is_same<Oper, plus<T>>::value) detail::scan_impl(input, output, false, T());
    else detail::generic_scan(input, output, false, init, oper);
}
/// output[i] = init + input[0] + ... + input[i-1] (scan.hpp:510-518).
template <class T> void exclusive_scan(const vector<T> &input, vector<T> &output, T init = T()) {
    detail::scan_impl(input, output, true, init);
}
template <class T, class Oper> void exclusive_scan(const vector<T> &input, vector<T> &output, T init, Oper oper) {
    if constexpr (std::is_same<Oper, plus<T>>::value) detail::scan_impl(input, output, true, init);
    else detail::generic_scan(input, output, true, init, oper);
}

} // namespace vex

#include "scan_by_key.hpp"
#endif
