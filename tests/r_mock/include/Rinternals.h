/* TEST DOUBLE, not R: the handful of R C-API declarations that r-package/src/r_glue.c uses, so that the glue (our own
 * code) can be compiled and EXECUTED in an image without R (tests/test_gpu_r_glue.py).  Semantics follow "Writing R
 * Extensions" for these calls only: vectors with a names attribute, external pointers with a C finalizer, Rf_error as
 * a non-local exit.  Nothing of the reference is built with this. */
#ifndef T3MI_R_MOCK_RINTERNALS_H
#define T3MI_R_MOCK_RINTERNALS_H
#include <stddef.h>

typedef ptrdiff_t R_xlen_t;
typedef struct r_mock_sexp* SEXP;
typedef enum { FALSE = 0, TRUE = 1 } Rboolean;
typedef void (*R_CFinalizer_t)(SEXP);

#define NILSXP 0
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
#define EXTPTRSXP 22
#define CHARSXP 9

extern SEXP R_NilValue, R_NamesSymbol;

int TYPEOF(SEXP x);
int Rf_length(SEXP x);
R_xlen_t Rf_xlength(SEXP x);
int* INTEGER(SEXP x);
double* REAL(SEXP x);
const char* CHAR(SEXP x);
SEXP STRING_ELT(SEXP x, R_xlen_t i);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP Rf_getAttrib(SEXP x, SEXP name);
SEXP Rf_allocVector(unsigned type, R_xlen_t n);
int Rf_asInteger(SEXP x);
double Rf_asReal(SEXP x);
Rboolean Rf_isReal(SEXP x);
Rboolean Rf_isInteger(SEXP x);
SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)
void Rf_error(const char* fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
void Rf_warning(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
SEXP R_MakeExternalPtr(void* p, SEXP tag, SEXP prot);
void* R_ExternalPtrAddr(SEXP s);
void R_ClearExternalPtr(SEXP s);
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit);
#endif
