// Kernel translation unit of libspamtree_hip.so: factor_generic.hpp, factor_big.hpp, factor_wide.hpp, factor_lchain.hpp (definitions).
#define ST_DEFS_FACTOR_WIDE 1   // this translation unit compiles the kernels of that family; the other headers give structures and prototypes
#define ST_STAMP_SUFFIX _wide
#include "factor_generic.hpp"
#include "factor_big.hpp"
#include "factor_wide.hpp"
#include "factor_lchain.hpp"
