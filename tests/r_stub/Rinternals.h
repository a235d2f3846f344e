/* tests/r_stub/Rinternals.h -- TEST-ONLY stand-in, NOT R's header (see README.md in this directory) */
#ifndef SSDE_TEST_R_STUB_RINTERNALS_H
#define SSDE_TEST_R_STUB_RINTERNALS_H
#include "R.h"
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef unsigned int SEXPTYPE;
#define LGLSXP 10
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
extern SEXP R_NilValue, R_NamesSymbol;
typedef void (*R_CFinalizer_t)(SEXP);
SEXP Rf_getAttrib(SEXP, SEXP);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
R_xlen_t Rf_xlength(SEXP);
int Rf_nrows(SEXP);
int Rf_ncols(SEXP);
int Rf_asInteger(SEXP);
const char *CHAR(SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
SEXP VECTOR_ELT(SEXP, R_xlen_t);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
double *REAL(SEXP);
int *INTEGER(SEXP);
int *LOGICAL(SEXP);
SEXP Rf_allocVector(SEXPTYPE, R_xlen_t);
SEXP Rf_allocMatrix(SEXPTYPE, int, int);
SEXP Rf_mkChar(const char *);
SEXP Rf_duplicate(SEXP);
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
void Rf_error(const char *, ...) __attribute__((noreturn, format(printf, 1, 2)));
void *R_ExternalPtrAddr(SEXP);
void R_ClearExternalPtr(SEXP);
SEXP R_MakeExternalPtr(void *, SEXP, SEXP);
void R_RegisterCFinalizerEx(SEXP, R_CFinalizer_t, Rboolean);
#endif
