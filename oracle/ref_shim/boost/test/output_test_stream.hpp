// shim: boost::test_tools::output_test_stream -- a string stream that can be compared with a literal
#ifndef VEX_REF_SHIM_OUTPUT_TEST_STREAM_HPP
#define VEX_REF_SHIM_OUTPUT_TEST_STREAM_HPP
#include <sstream>
#include <string>
namespace boost { namespace test_tools {
class output_test_stream : public std::ostringstream {
    public:
        bool is_equal(const std::string &expected, bool flush = true) {
            const bool same = str() == expected;
            if (flush) str("");
            return same;
        }
        bool is_empty(bool flush = true) { return is_equal("", flush); }
};
} }
#endif
