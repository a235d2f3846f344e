/* tests/r_stub/R.h -- TEST-ONLY stand-in, NOT R's header (see README.md in this directory): syntax check of the .Call shim only */
#ifndef SSDE_TEST_R_STUB_R_H
#define SSDE_TEST_R_STUB_R_H
#include <stddef.h>
#include <stdint.h>
typedef enum { FALSE = 0, TRUE } Rboolean;
char *R_alloc(size_t n, int size);
#endif
