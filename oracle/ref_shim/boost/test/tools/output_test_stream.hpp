#include "../output_test_stream.hpp"
