// Kernel translation unit of libspamtree_hip.so: sample_kernels.hpp (definitions).
#define ST_DEFS_SAMPLE 1   // this translation unit compiles the kernels of that family; the other headers give structures and prototypes
#define ST_STAMP_SUFFIX _sample
#include "sample_kernels.hpp"
