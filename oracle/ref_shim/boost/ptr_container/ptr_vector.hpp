// shim: boost::ptr_vector<T> -- owns the pointers pushed into it, indexes to references
#ifndef VEX_REF_SHIM_PTR_VECTOR_HPP
#define VEX_REF_SHIM_PTR_VECTOR_HPP
#include <memory>
#include <vector>
namespace boost {
template <class T> class ptr_vector {
    public:
        void push_back(T *p) { items.emplace_back(p); }
        T &operator[](size_t i) { return *items[i]; }
        const T &operator[](size_t i) const { return *items[i]; }
        size_t size() const { return items.size(); }
    private:
        std::vector<std::unique_ptr<T>> items;
};
}
#endif
