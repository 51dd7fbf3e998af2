// Kernel translation unit of libspamtree_hip.so: factor_generic.hpp (definitions).
#define ST_DEFS_FACTOR_GENERIC 1   // this translation unit compiles the kernels of that family; the other headers give structures and prototypes
#define ST_STAMP_SUFFIX _generic
#include "st_device.hpp"
#include "chol_blocked.hpp"
#include "factor_generic.hpp"
