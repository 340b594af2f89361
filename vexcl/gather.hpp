#ifndef VEXCL_GATHER_HPP
#define VEXCL_GATHER_HPP
// vex::gather / vex::scatter: selected elements of a (multi-device) vector to / from a
// host array (reference API: vexcl/gather.hpp:90-165; tests/vector_copy.cpp:72-109).
//
// MI355X design.  The reference maps every device partition to the host in full and
// picks the elements there.  Here the selection runs on the GPUs: at construction the
// requested positions are bucketed by owning device and uploaded as one index vector
// per device; a call then launches ONE kernel per device (`packed[i] = part[idx[i]]`
// for gather, the indexed store for scatter -- both are the permutation view of
// vector_view.hpp) and only the selected elements cross PCIe.
#include <algorithm>
#include <numeric>
#include "vector.hpp"
#include "vector_view.hpp"

namespace vex {
namespace detail {

/// Where each requested element lives: for device d, `where[d]` lists positions inside
/// that device's partition (on the device) and `slot[d]` the places of those elements in
/// the user's host array, in the same order.
class selection {
    public:
        selection(const std::vector<backend::command_queue> &queue, size_t size, const std::vector<size_t> &wanted)
            : queue(queue), slot(queue.size()), where(queue.size()), total(wanted.size())
        {
            const std::vector<size_t> bounds = partition(size, queue);
            std::vector<std::vector<size_t>> local(queue.size());
            for (size_t k = 0; k < wanted.size(); ++k) {
                const size_t g = wanted[k];
                precondition(g < size, "gather / scatter: index out of range");
                const size_t d = column_owner(g, bounds);
                local[d].push_back(g - bounds[d]);
                slot[d].push_back(k);
            }
            for (unsigned d = 0; d < queue.size(); ++d)
                if (!local[d].empty()) where[d] = vector<size_t>(one(d), local[d]);
        }
    protected:
        std::vector<backend::command_queue> one(unsigned d) const { return std::vector<backend::command_queue>(1, queue[d]); }
        size_t count(unsigned d) const { return slot[d].size(); }

        std::vector<backend::command_queue> queue;
        std::vector<std::vector<size_t>> slot;
        std::vector<vector<size_t>> where;
        size_t total;
};

} // namespace detail

/// get(device_vector, host_array): host_array[k] = device_vector[indices[k]].
class gather : protected detail::selection {
    public:
        gather(const std::vector<backend::command_queue> &q, size_t size, const std::vector<size_t> &indices)
            : detail::selection(q, size, indices) {}

        template <class T, class HostVector>
        void operator()(const vex::vector<T> &src, HostVector &dst) {
            std::vector<std::vector<T>> landed(queue.size());
            std::vector<vector<T>> packed(queue.size());
            for (unsigned d = 0; d < queue.size(); ++d) {
                if (!count(d)) continue;
                vector<T> segment(queue[d], src(d));             // this device's partition, viewed as a vector
                packed[d] = vector<T>(one(d), count(d));
                packed[d] = permutation(where[d])(segment);      // the pack kernel
                landed[d].resize(count(d));
                packed[d](0).read(queue[d], 0, count(d), landed[d].data(), false);
            }
            for (unsigned d = 0; d < queue.size(); ++d) {
                if (!count(d)) continue;
                queue[d].finish();
                for (size_t i = 0; i < count(d); ++i) dst[slot[d][i]] = landed[d][i];
            }
        }
};

/// put(host_array, device_vector): device_vector[indices[k]] = host_array[k].
class scatter : protected detail::selection {
    public:
        scatter(const std::vector<backend::command_queue> &q, size_t size, const std::vector<size_t> &indices)
            : detail::selection(q, size, indices) {}

        template <class HostVector, class T>
        void operator()(const HostVector &src, vex::vector<T> &dst) {
            for (unsigned d = 0; d < queue.size(); ++d) {
                if (!count(d)) continue;
                std::vector<T> staged(count(d));
                for (size_t i = 0; i < count(d); ++i) staged[i] = src[slot[d][i]];
                vector<T> segment(queue[d], dst(d));
                vector<T> values(one(d), staged);
                permutation(where[d])(segment) = values;         // the indexed-store kernel
            }
            for (unsigned d = 0; d < queue.size(); ++d) if (count(d)) queue[d].finish();
        }
};

} // namespace vex
#endif
