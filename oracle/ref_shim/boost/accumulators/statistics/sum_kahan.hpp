#include "../accumulators.hpp"
