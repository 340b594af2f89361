// Stand-in for <boost/io/ios_state.hpp>: ios_all_saver restores a stream's formatting state on scope exit.
#ifndef VEX_REF_SHIM_BOOST_IOS_STATE_HPP
#define VEX_REF_SHIM_BOOST_IOS_STATE_HPP
#include <ios>
namespace boost { namespace io {
struct ios_all_saver {
    std::ios_base &s; std::ios_base::fmtflags f; std::streamsize p, w;
    explicit ios_all_saver(std::ios_base &s) : s(s), f(s.flags()), p(s.precision()), w(s.width()) {}
    ~ios_all_saver() { s.flags(f); s.precision(p); s.width(w); }
};
} }
#endif
