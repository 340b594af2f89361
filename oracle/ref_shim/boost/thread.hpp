// shim: boost::thread is std::thread
#ifndef VEX_REF_SHIM_THREAD_HPP
#define VEX_REF_SHIM_THREAD_HPP
#include <thread>
namespace boost { typedef std::thread thread; }
#endif
