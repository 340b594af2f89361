#ifndef VEXCL_VEXCL_HPP
#define VEXCL_VEXCL_HPP
// Umbrella header (reference: vexcl/vexcl.hpp).  MI355X-native implementation
// of VexCL's vector-expression hot path; see DESIGN.md for what is in scope.
#include "util.hpp"
#include "types.hpp"
#include "backend.hpp"
#include "cache.hpp"
#include "devlist.hpp"
#include "profiler.hpp"
#include "operations.hpp"
#include "function.hpp"
#include "vector.hpp"
#include "element_index.hpp"
#include "tagged_terminal.hpp"
#include "vector_pointer.hpp"
#include "vector_view.hpp"
#include "eval.hpp"
#include "constants.hpp"
#include "reductor.hpp"
#include "spmat.hpp"
#include "stencil.hpp"
#include "sparse/csr.hpp"
#include "sparse/ell.hpp"
#include "sparse/matrix.hpp"
#include "sparse/distributed.hpp"
#include "scan.hpp"
#include "sort.hpp"
#endif
