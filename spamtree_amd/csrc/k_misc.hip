// Kernel translation unit of libspamtree_hip.so: misc_kernels.hpp (definitions).
#define ST_DEFS_MISC 1   // this translation unit compiles the kernels of that family; the other headers give structures and prototypes
#include "misc_kernels.hpp"
