// k_dense_wide.hip -- dense_kernel for responses of five to eight columns run as ONE filter (a translation unit of its own for build time)
#define SSDE_DENSE_WIDE_TU 1
#define SSDE_DENSE_NOUNROLL 1
#include "k_dense.hip"
