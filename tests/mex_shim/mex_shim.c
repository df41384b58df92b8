/* Implementation of the functional mex.h stand-in (see mex.h).  Test infrastructure only. */
#include "mex.h"

#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXDIM 4
#define MAXFIELD 32
struct mxArray_tag {
  mxClassID cls;
  mwSize ndim;
  mwSize dims[MAXDIM];
  void* data;
  int nfields;
  char* fname[MAXFIELD];
  mxArray* fval[MAXFIELD];
};

static jmp_buf g_jmp;
static int g_in_call = 0;
static char g_err_id[128], g_err_msg[1024], g_warn_id[128], g_warn_msg[1024];
static int g_locked = 0, g_live = 0;
static void (*g_at_exit)(void) = NULL;
/* arrays created during the running call: MATLAB frees them when the call ends in an error */
static mxArray* g_made[256];
static int g_nmade = 0;

static size_t elem_size(mxClassID c) {
  switch (c) {
    case mxDOUBLE_CLASS: case mxINT64_CLASS: case mxUINT64_CLASS: return 8;
    case mxSINGLE_CLASS: case mxINT32_CLASS: case mxUINT32_CLASS: return 4;
    case mxINT16_CLASS: case mxUINT16_CLASS: case mxCHAR_CLASS: return 2;      /* MATLAB chars are 16 bit */
    case mxINT8_CLASS: case mxUINT8_CLASS: case mxLOGICAL_CLASS: return 1;
    default: return 0;
  }
}
static size_t numel(const mxArray* a) {
  size_t n = 1;
  for (mwSize i = 0; i < a->ndim; ++i) n *= a->dims[i];
  return n;
}
static mxArray* make(mxClassID cls, mwSize ndim, const mwSize* dims) {
  mxArray* a = (mxArray*)calloc(1, sizeof *a);
  a->cls = cls;
  a->ndim = ndim < 2 ? 2 : ndim;
  if (a->ndim > MAXDIM) { fprintf(stderr, "mex_shim: more than %d dimensions\n", MAXDIM); abort(); }
  for (mwSize i = 0; i < a->ndim; ++i) a->dims[i] = i < ndim ? dims[i] : 1;
  while (a->ndim > 2 && a->dims[a->ndim - 1] == 1) --a->ndim;               /* MATLAB drops trailing singleton dimensions */
  const size_t bytes = numel(a) * elem_size(cls);
  a->data = bytes ? calloc(1, bytes) : NULL;
  ++g_live;
  if (g_in_call && g_nmade < 256) g_made[g_nmade++] = a;
  return a;
}

bool mxIsUint64(const mxArray* a) { return a->cls == mxUINT64_CLASS; }
bool mxIsUint8(const mxArray* a) { return a->cls == mxUINT8_CLASS; }
bool mxIsInt32(const mxArray* a) { return a->cls == mxINT32_CLASS; }
bool mxIsDouble(const mxArray* a) { return a->cls == mxDOUBLE_CLASS; }
bool mxIsComplex(const mxArray* a) { (void)a; return false; }
bool mxIsChar(const mxArray* a) { return a->cls == mxCHAR_CLASS; }
bool mxIsStruct(const mxArray* a) { return a->cls == mxSTRUCT_CLASS; }
bool mxIsNumeric(const mxArray* a) { return a->cls >= mxDOUBLE_CLASS; }
bool mxIsEmpty(const mxArray* a) { return numel(a) == 0; }
mxClassID mxGetClassID(const mxArray* a) { return a->cls; }
size_t mxGetNumberOfElements(const mxArray* a) { return numel(a); }
size_t mxGetM(const mxArray* a) { return a->dims[0]; }
size_t mxGetN(const mxArray* a) {
  size_t n = 1;
  for (mwSize i = 1; i < a->ndim; ++i) n *= a->dims[i];
  return n;
}
mwSize mxGetNumberOfDimensions(const mxArray* a) { return a->ndim; }
const mwSize* mxGetDimensions(const mxArray* a) { return a->dims; }
void* mxGetData(const mxArray* a) { return a->data; }
double* mxGetPr(const mxArray* a) { return a->cls == mxDOUBLE_CLASS ? (double*)a->data : NULL; }
double mxGetScalar(const mxArray* a) {
  if (numel(a) == 0) return 0.0;                                            /* MATLAB: undefined; never relied on */
  const void* p = a->data;
  switch (a->cls) {
    case mxDOUBLE_CLASS: return *(const double*)p;
    case mxSINGLE_CLASS: return *(const float*)p;
    case mxINT8_CLASS: return *(const int8_t*)p;
    case mxUINT8_CLASS: case mxLOGICAL_CLASS: return *(const uint8_t*)p;
    case mxINT16_CLASS: return *(const int16_t*)p;
    case mxUINT16_CLASS: case mxCHAR_CLASS: return *(const uint16_t*)p;
    case mxINT32_CLASS: return *(const int32_t*)p;
    case mxUINT32_CLASS: return *(const uint32_t*)p;
    case mxINT64_CLASS: return (double)*(const int64_t*)p;
    case mxUINT64_CLASS: return (double)*(const uint64_t*)p;
    default: return 0.0;
  }
}
int mxGetString(const mxArray* a, char* buf, mwSize buflen) {
  if (a->cls != mxCHAR_CLASS || buflen == 0) return 1;
  const size_t n = numel(a);
  const uint16_t* s = (const uint16_t*)a->data;
  const size_t k = n < buflen - 1 ? n : buflen - 1;
  for (size_t i = 0; i < k; ++i) buf[i] = (char)s[i];
  buf[k] = 0;
  return n > buflen - 1;                                                    /* 1: truncated, as MATLAB reports it */
}
mxArray* mxGetField(const mxArray* s, mwIndex index, const char* name) {
  if (s->cls != mxSTRUCT_CLASS || index != 0) return NULL;
  for (int i = 0; i < s->nfields; ++i)
    if (!strcmp(s->fname[i], name)) return s->fval[i];
  return NULL;
}
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID cls, mxComplexity c) { (void)c; return make(cls, ndim, dims); }
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity c) {
  const mwSize d[2] = {m, n};
  (void)c;
  return make(cls, 2, d);
}
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c) { return mxCreateNumericMatrix(m, n, mxDOUBLE_CLASS, c); }
mxArray* mxCreateDoubleScalar(double v) {
  mxArray* a = mxCreateDoubleMatrix(1, 1, mxREAL);
  *(double*)a->data = v;
  return a;
}
mxArray* mxCreateString(const char* s) {
  const mwSize d[2] = {1, strlen(s)};
  mxArray* a = make(mxCHAR_CLASS, 2, d);
  for (size_t i = 0; i < d[1]; ++i) ((uint16_t*)a->data)[i] = (unsigned char)s[i];
  return a;
}
void mxDestroyArray(mxArray* a) {
  if (!a) return;
  for (int i = 0; i < a->nfields; ++i) { free(a->fname[i]); mxDestroyArray(a->fval[i]); }
  for (int i = 0; i < g_nmade; ++i)
    if (g_made[i] == a) g_made[i] = NULL;
  free(a->data);
  free(a);
  --g_live;
}

void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err_msg, sizeof g_err_msg, fmt, ap);
  va_end(ap);
  snprintf(g_err_id, sizeof g_err_id, "%s", id);
  if (!g_in_call) { fprintf(stderr, "mexErrMsgIdAndTxt outside shim_call: %s: %s\n", g_err_id, g_err_msg); abort(); }
  longjmp(g_jmp, 1);
}
void mexWarnMsgIdAndTxt(const char* id, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_warn_msg, sizeof g_warn_msg, fmt, ap);
  va_end(ap);
  snprintf(g_warn_id, sizeof g_warn_id, "%s", id);
}
bool mexIsLocked(void) { return g_locked > 0; }
void mexLock(void) { ++g_locked; }
void mexUnlock(void) { if (g_locked > 0) --g_locked; }
int mexAtExit(void (*f)(void)) { g_at_exit = f; return 0; }

/* ---- harness side ---- */
mxArray* shim_new(int class_id, int ndim, const size_t* dims) { return make((mxClassID)class_id, (mwSize)ndim, dims); }
mxArray* shim_new_struct(void) {
  const mwSize d[2] = {1, 1};
  return make(mxSTRUCT_CLASS, 2, d);
}
int shim_set_field(mxArray* s, const char* name, mxArray* value) {
  if (!s || s->cls != mxSTRUCT_CLASS || s->nfields >= MAXFIELD) return 1;
  s->fname[s->nfields] = strdup(name);
  s->fval[s->nfields] = value;
  ++s->nfields;
  return 0;
}
int shim_call(int nlhs, mxArray** plhs, int nrhs, mxArray** prhs) {
  g_err_id[0] = g_err_msg[0] = g_warn_id[0] = g_warn_msg[0] = 0;
  g_nmade = 0;
  g_in_call = 1;
  if (setjmp(g_jmp)) {                       /* mexErrMsgIdAndTxt: MATLAB frees what the call had allocated */
    g_in_call = 0;
    for (int i = 0; i < g_nmade; ++i)
      if (g_made[i]) { mxArray* a = g_made[i]; g_made[i] = NULL; mxDestroyArray(a); }
    g_nmade = 0;
    return 1;
  }
  mexFunction(nlhs, plhs, nrhs, (const mxArray**)prhs);
  g_in_call = 0;
  g_nmade = 0;
  return 0;
}
const char* shim_error_id(void) { return g_err_id; }
const char* shim_error_msg(void) { return g_err_msg; }
const char* shim_warning_id(void) { return g_warn_id; }
const char* shim_warning_msg(void) { return g_warn_msg; }
int shim_locked(void) { return g_locked; }
void shim_run_at_exit(void) {
  if (g_at_exit) g_at_exit();
  g_at_exit = NULL;
  g_locked = 0;
}
int shim_live_arrays(void) { return g_live; }
