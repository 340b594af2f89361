#ifndef VEXCL_GATHER_HPP
#define VEXCL_GATHER_HPP
// vex::gather / vex::scatter: selected elements of a (multi-device) vector to / from a
// host array (reference: vexcl/gather.hpp:44-165; tests/vector_copy.cpp:72-109).
// The reference maps every partition to the host in full; here each device packs the
// requested elements with one kernel (`dst[i] = src[idx[i]]`, resp. the indexed store)
// and only those cross PCIe.
#include <algorithm>
#include <numeric>
#include "vector.hpp"
#include "vector_view.hpp"

namespace vex {
namespace detail {

/// Splits the indices by owning device (gather.hpp:44-88): idx holds positions local to
/// the owner's partition, ptr[d]..ptr[d+1] is device d's share, ord maps back when the
/// indices had to be sorted first.
class index_partition {
    public:
        index_partition(const std::vector<backend::command_queue> &q, size_t size, const std::vector<size_t> &indices)
            : queue(q), ptr(q.size() + 1, 0)
        {
            std::vector<size_t> sorted;
            if (queue.size() > 1 && !std::is_sorted(indices.begin(), indices.end())) {
                ord.resize(indices.size());
                std::iota(ord.begin(), ord.end(), size_t(0));
                std::sort(ord.begin(), ord.end(), [&indices](size_t i, size_t j) { return indices[i] < indices[j]; });
                sorted.resize(indices.size());
                for (size_t i = 0; i < indices.size(); ++i) sorted[i] = indices[ord[i]];
            } else {
                sorted = indices;
            }
            const std::vector<size_t> part = partition(size, queue);
            std::vector<std::vector<size_t>> local(queue.size());
            for (size_t g : sorted) {
                precondition(g < size, "gather / scatter index out of range");
                size_t d = column_owner(g, part);
                local[d].push_back(g - part[d]);
                ++ptr[d + 1];
            }
            std::partial_sum(ptr.begin(), ptr.end(), ptr.begin());
            for (unsigned d = 0; d < queue.size(); ++d)
                if (!local[d].empty()) idx.push_back(vector<size_t>(std::vector<backend::command_queue>(1, queue[d]), local[d]));
                else idx.push_back(vector<size_t>());
        }
    protected:
        std::vector<backend::command_queue> queue;
        std::vector<size_t> ptr, ord;
        std::vector<vector<size_t>> idx;      // per device, on the device
};

} // namespace detail

class gather : protected detail::index_partition {
    public:
        gather(const std::vector<backend::command_queue> &q, size_t size, const std::vector<size_t> &indices)
            : detail::index_partition(q, size, indices) {}

        template <class T, class HostVector>
        void operator()(const vex::vector<T> &src, HostVector &dst) {
            std::vector<T> packed(ptr.back());
            std::vector<vector<T>> buf(queue.size());
            for (unsigned d = 0; d < queue.size(); ++d) {
                const size_t n = ptr[d + 1] - ptr[d];
                if (!n) continue;
                std::vector<backend::command_queue> q1(1, queue[d]);
                vector<T> part(queue[d], src(d));                // this device's partition as a vector of its own
                buf[d] = vector<T>(q1, n);
                buf[d] = permutation(idx[d])(part);              // one gather kernel
                buf[d](0).read(queue[d], 0, n, packed.data() + ptr[d], false);
            }
            for (unsigned d = 0; d < queue.size(); ++d) if (ptr[d + 1] > ptr[d]) queue[d].finish();
            if (ord.empty()) for (size_t i = 0; i < packed.size(); ++i) dst[i] = packed[i];
            else for (size_t i = 0; i < packed.size(); ++i) dst[ord[i]] = packed[i];
        }
};

class scatter : protected detail::index_partition {
    public:
        scatter(const std::vector<backend::command_queue> &q, size_t size, const std::vector<size_t> &indices)
            : detail::index_partition(q, size, indices) {}

        template <class HostVector, class T>
        void operator()(const HostVector &src, vex::vector<T> &dst) {
            std::vector<T> packed(ptr.back());
            if (ord.empty()) for (size_t i = 0; i < packed.size(); ++i) packed[i] = src[i];
            else for (size_t i = 0; i < packed.size(); ++i) packed[i] = src[ord[i]];
            for (unsigned d = 0; d < queue.size(); ++d) {
                const size_t n = ptr[d + 1] - ptr[d];
                if (!n) continue;
                std::vector<backend::command_queue> q1(1, queue[d]);
                vector<T> part(queue[d], dst(d));
                vector<T> vals(q1, n, packed.data() + ptr[d]);
                permutation(idx[d])(part) = vals;                // one indexed-store kernel
            }
            for (unsigned d = 0; d < queue.size(); ++d) if (ptr[d + 1] > ptr[d]) queue[d].finish();
        }
};

} // namespace vex
#endif
