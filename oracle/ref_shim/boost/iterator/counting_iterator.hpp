// shim: boost::counting_iterator<T> -- a random-access iterator over consecutive integers
#ifndef VEX_REF_SHIM_COUNTING_ITERATOR_HPP
#define VEX_REF_SHIM_COUNTING_ITERATOR_HPP
#include <cstddef>
#include <iterator>
namespace boost {
template <class T> class counting_iterator {
    public:
        typedef std::random_access_iterator_tag iterator_category;
        typedef T value_type; typedef std::ptrdiff_t difference_type; typedef const T *pointer; typedef T reference;
        counting_iterator() : v() {}
        explicit counting_iterator(T v) : v(v) {}
        T operator*() const { return v; }
        T operator[](difference_type k) const { return v + k; }
        counting_iterator &operator++() { ++v; return *this; }
        counting_iterator operator++(int) { counting_iterator t(*this); ++v; return t; }
        counting_iterator &operator--() { --v; return *this; }
        counting_iterator operator--(int) { counting_iterator t(*this); --v; return t; }
        counting_iterator &operator+=(difference_type k) { v += k; return *this; }
        counting_iterator &operator-=(difference_type k) { v -= k; return *this; }
        friend counting_iterator operator+(counting_iterator a, difference_type k) { return a += k; }
        friend counting_iterator operator+(difference_type k, counting_iterator a) { return a += k; }
        friend counting_iterator operator-(counting_iterator a, difference_type k) { return a -= k; }
        friend difference_type operator-(const counting_iterator &a, const counting_iterator &b) { return (difference_type)a.v - (difference_type)b.v; }
        friend bool operator==(const counting_iterator &a, const counting_iterator &b) { return a.v == b.v; }
        friend bool operator!=(const counting_iterator &a, const counting_iterator &b) { return a.v != b.v; }
        friend bool operator<(const counting_iterator &a, const counting_iterator &b) { return a.v < b.v; }
        friend bool operator>(const counting_iterator &a, const counting_iterator &b) { return a.v > b.v; }
        friend bool operator<=(const counting_iterator &a, const counting_iterator &b) { return a.v <= b.v; }
        friend bool operator>=(const counting_iterator &a, const counting_iterator &b) { return a.v >= b.v; }
    private:
        T v;
};
}
#endif
