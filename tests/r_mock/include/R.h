/* TEST DOUBLE, see Rinternals.h in this directory. */
#ifndef T3MI_R_MOCK_R_H
#define T3MI_R_MOCK_R_H
#include <stdlib.h>
#endif
