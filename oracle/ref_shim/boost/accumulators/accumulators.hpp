// shim: the one accumulator the reference's tests use -- a Kahan-compensated sum
#ifndef VEX_REF_SHIM_ACCUMULATORS_HPP
#define VEX_REF_SHIM_ACCUMULATORS_HPP
namespace boost { namespace accumulators {
namespace tag { struct sum_kahan {}; }
template <class... Tags> struct stats {};
template <class T, class Stats> class accumulator_set {
    public:
        accumulator_set() : sum(0), comp(0) {}
        void operator()(T v) { const T y = v - comp; const T t = sum + y; comp = (t - sum) - y; sum = t; }
        T kahan() const { return sum; }
    private:
        T sum, comp;
};
template <class T, class Stats> T sum_kahan(const accumulator_set<T, Stats> &a) { return a.kahan(); }
} }
#endif
