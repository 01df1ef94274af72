/* TEST DOUBLE, see ../Rinternals.h. */
#ifndef T3MI_R_MOCK_RDYNLOAD_H
#define T3MI_R_MOCK_RDYNLOAD_H
#include <Rinternals.h>
typedef void* (*DL_FUNC)(void);
typedef struct { const char* name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct r_mock_dll DllInfo;
int R_registerRoutines(DllInfo* info, const void* c, const R_CallMethodDef* call, const void* f, const void* e);
Rboolean R_useDynamicSymbols(DllInfo* info, Rboolean value);
#endif
