#ifndef VEXCL_RANDOM_THREEFRY_HPP
#define VEXCL_RANDOM_THREEFRY_HPP
// vex::random::threefry lives in random.hpp (reference: vexcl/random/threefry.hpp).
#include "../random.hpp"
#endif
