// Kernel translation unit of libspamtree_hip.so: factor_generic.hpp, chol_blocked.hpp, factor_quad.hpp (definitions).
#define ST_DEFS_FACTOR_QUAD 1   // this translation unit compiles the kernels of that family; the other headers give structures and prototypes
#define ST_STAMP_SUFFIX 
#include "factor_generic.hpp"
#include "chol_blocked.hpp"
#include "factor_quad.hpp"
