#ifndef VEXCL_UTIL_HPP
#define VEXCL_UTIL_HPP
// Small helpers of the vex:: API (reference: vexcl/util.hpp:67-152).
#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>
#include <algorithm>

namespace vex {

/// Throws std::runtime_error when the condition is false (util.hpp:67-77).
inline void precondition(bool cond, const std::string &msg) {
    if (!cond) throw std::runtime_error(msg);
}

/// Next power of two >= n (util.hpp nextpow2).
inline size_t nextpow2(size_t n) {
    size_t p = 1;
    while (p < n) p <<= 1;
    return p;
}

/// Rounds n up to a multiple of m (util.hpp:91-93, default 16).
inline size_t alignup(size_t n, size_t m = 16U) { return (n + m - 1) / m * m; }

/// Owner of column c in a partitioning (util.hpp:143-152).
inline size_t column_owner(size_t c, const std::vector<size_t> &part) {
    return std::upper_bound(part.begin(), part.end(), c) - part.begin() - 1;
}

} // namespace vex
#endif
