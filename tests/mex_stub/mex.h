/* Declarations-only stand-in for MATLAB's mex.h, used by tests/test_host_logic.py to type-check matlab/kp_mex.c against
 * include/koopman_hip.h (gcc -fsyntax-only).  Nothing here is linked or executed; MATLAB is absent from the image. */
#ifndef MEX_STUB_H
#define MEX_STUB_H
#include <stdbool.h>
#include <stddef.h>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef enum { mxDOUBLE_CLASS, mxUINT64_CLASS } mxClassID;
typedef enum { mxREAL, mxCOMPLEX } mxComplexity;
bool mxIsUint64(const mxArray*); bool mxIsUint8(const mxArray*); bool mxIsInt32(const mxArray*); bool mxIsDouble(const mxArray*);
bool mxIsComplex(const mxArray*); bool mxIsEmpty(const mxArray*);
size_t mxGetNumberOfElements(const mxArray*); size_t mxGetM(const mxArray*); size_t mxGetN(const mxArray*);
void* mxGetData(const mxArray*); double* mxGetPr(const mxArray*); double mxGetScalar(const mxArray*);
int mxGetString(const mxArray*, char*, mwSize);
mxArray* mxGetField(const mxArray*, mwSize, const char*);
mxArray* mxCreateNumericMatrix(mwSize, mwSize, mxClassID, mxComplexity);
mxArray* mxCreateNumericArray(mwSize, const mwSize*, mxClassID, mxComplexity);
mxArray* mxCreateDoubleMatrix(mwSize, mwSize, mxComplexity);
mxArray* mxCreateDoubleScalar(double);
void mxDestroyArray(mxArray*);
void mexErrMsgIdAndTxt(const char*, const char*, ...);
void mexWarnMsgIdAndTxt(const char*, const char*, ...);
bool mexIsLocked(void); void mexLock(void); int mexAtExit(void (*)(void));
#endif
