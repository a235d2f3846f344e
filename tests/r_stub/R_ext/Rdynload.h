/* tests/r_stub/R_ext/Rdynload.h -- TEST-ONLY stand-in, NOT R's header (see ../README.md) */
#ifndef SSDE_TEST_R_STUB_RDYNLOAD_H
#define SSDE_TEST_R_STUB_RDYNLOAD_H
#include "../R.h"
typedef void *(*DL_FUNC)(void);
typedef struct _DllInfo DllInfo;
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef R_CallMethodDef R_ExternalMethodDef;
typedef struct { const char *name; DL_FUNC fun; int numArgs; void *types; } R_CMethodDef;
typedef R_CMethodDef R_FortranMethodDef;
int R_registerRoutines(DllInfo *, const R_CMethodDef *, const R_CallMethodDef *, const R_FortranMethodDef *, const R_ExternalMethodDef *);
Rboolean R_useDynamicSymbols(DllInfo *, Rboolean);
#endif
