/* Functional stand-in for MATLAB's mex.h / matrix.h (test infrastructure; MATLAB is absent from the image).
 *
 * matlab/kp_mex.c is compiled UNCHANGED against this header and linked with mex_shim.c into tests/mex_shim/kp_mex_shim.so, so
 * that tests/ can EXECUTE mexFunction - every gateway command, on the GPU box - and compare what comes back with the direct
 * C-ABI result (tests/test_mex_gateway.py).  Only the subset of the MEX API the gateway uses exists here, with MATLAB's
 * documented semantics: column-major storage, mxGetN = product of the trailing dimensions, mxGetScalar converts the first
 * element of any numeric class, mexErrMsgIdAndTxt does not return (longjmp back into shim_call). */
#ifndef KP_MEX_SHIM_H
#define KP_MEX_SHIM_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum {
  mxUNKNOWN_CLASS = 0, mxCELL_CLASS, mxSTRUCT_CLASS, mxLOGICAL_CLASS, mxCHAR_CLASS, mxVOID_CLASS, mxDOUBLE_CLASS, mxSINGLE_CLASS,
  mxINT8_CLASS, mxUINT8_CLASS, mxINT16_CLASS, mxUINT16_CLASS, mxINT32_CLASS, mxUINT32_CLASS, mxINT64_CLASS, mxUINT64_CLASS
} mxClassID;
typedef enum { mxREAL, mxCOMPLEX } mxComplexity;

bool mxIsUint64(const mxArray*); bool mxIsUint8(const mxArray*); bool mxIsInt32(const mxArray*); bool mxIsDouble(const mxArray*);
bool mxIsComplex(const mxArray*); bool mxIsEmpty(const mxArray*); bool mxIsChar(const mxArray*); bool mxIsStruct(const mxArray*);
bool mxIsNumeric(const mxArray*);
mxClassID mxGetClassID(const mxArray*);
size_t mxGetNumberOfElements(const mxArray*); size_t mxGetM(const mxArray*); size_t mxGetN(const mxArray*);
mwSize mxGetNumberOfDimensions(const mxArray*); const mwSize* mxGetDimensions(const mxArray*);
void* mxGetData(const mxArray*); double* mxGetPr(const mxArray*); double mxGetScalar(const mxArray*);
int mxGetString(const mxArray*, char*, mwSize);
mxArray* mxGetField(const mxArray*, mwIndex, const char*);
mxArray* mxCreateNumericMatrix(mwSize, mwSize, mxClassID, mxComplexity);
mxArray* mxCreateNumericArray(mwSize, const mwSize*, mxClassID, mxComplexity);
mxArray* mxCreateDoubleMatrix(mwSize, mwSize, mxComplexity);
mxArray* mxCreateDoubleScalar(double);
mxArray* mxCreateString(const char*);
void mxDestroyArray(mxArray*);
void mexErrMsgIdAndTxt(const char*, const char*, ...) __attribute__((noreturn));
void mexWarnMsgIdAndTxt(const char*, const char*, ...);
bool mexIsLocked(void); void mexLock(void); void mexUnlock(void); int mexAtExit(void (*)(void));

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);

/* ---- harness side (not part of MATLAB's API): what tests/mexshim.py calls through ctypes ---- */
mxArray* shim_new(int class_id, int ndim, const size_t* dims);      /* zero-filled numeric / char array */
mxArray* shim_new_struct(void);                                     /* 1 x 1 struct without fields */
int shim_set_field(mxArray* s, const char* name, mxArray* value);   /* the struct takes ownership of value */
int shim_call(int nlhs, mxArray** plhs, int nrhs, mxArray** prhs);  /* 0, or 1 after mexErrMsgIdAndTxt */
const char* shim_error_id(void);
const char* shim_error_msg(void);
const char* shim_warning_id(void);                                  /* most recent mexWarnMsgIdAndTxt ("" if none since the last call) */
const char* shim_warning_msg(void);
int shim_locked(void);
void shim_run_at_exit(void);                                        /* what MATLAB does at `clear mex` / exit */
int shim_live_arrays(void);                                         /* arrays allocated and not yet destroyed (leak check) */

#ifdef __cplusplus
}
#endif
#endif
