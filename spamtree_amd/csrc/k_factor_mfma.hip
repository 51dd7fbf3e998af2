// Kernel translation unit of libspamtree_hip.so: factor_mfma.hpp (definitions).
#define ST_DEFS_FACTOR_MFMA 1   // this translation unit compiles the kernels of that family; the other headers give structures and prototypes
#define ST_STAMP_SUFFIX _mfma
#include "factor_mfma.hpp"
