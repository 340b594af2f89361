
#include <vexcl/vexcl.hpp>
#include <vexcl/scan_by_key.hpp>
#include <vexcl/reduce_by_key.hpp>

#include <fstream>
using namespace vex;
VEX_FUNCTION(bool, keys_equal, (int, a1)(long, a2)(int, b1)(long, b2), return a1 == b1 && a2 == b2;);
VEX_FUNCTION(double, dplus, (double, x)(double, y), return x + y;);
int main(int argc, char **argv) {
    using namespace detail::sbk;
    backend::command_queue q;
    const std::string dir = argc > 1 ? argv[1] : ".";
    int i = 0;
    for (scan_mode m : {INCLUSIVE, EXCLUSIVE, REDUCE}) {
        std::ofstream(dir + "/d2_" + std::to_string(i) + ".hip") << source<double, decltype(keys_equal), decltype(dplus)>(q, {"int", "long"}, m);
        std::ofstream(dir + "/d1_" + std::to_string(i) + ".hip") << source<double, equal_fn<int>, plus_fn<double>>(q, {"int"}, m);
        std::ofstream(dir + "/f1_" + std::to_string(i) + ".hip") << source<float, equal_fn<int>, plus_fn<float>>(q, {"int"}, m);
        ++i;
    }
    return 0;
}
