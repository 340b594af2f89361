#ifndef VEXCL_RANDOM_PHILOX_HPP
#define VEXCL_RANDOM_PHILOX_HPP
// vex::random::philox lives in random.hpp (reference: vexcl/random/philox.hpp).
#include "../random.hpp"
#endif
