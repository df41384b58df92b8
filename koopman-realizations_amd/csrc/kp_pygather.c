/* _kp_gather: host-side marshalling helper of the Python mirror (NOT part of the C ABI, no GPU code).
 *
 * evaluate_rand_models.m:45-59 hands the sweep a cell array of data4sysid structs - tens of thousands of small trial
 * arrays (1024 systems x 11 trials x {t, y, u}).  The library wants one block per quantity (kp_traj_upload).  In a MEX
 * gateway that gather is a loop over mxGetPr pointers; from Python, np.concatenate spends ~1.7 us of interpreter-side
 * set-up per 8 KB piece and holds the GIL throughout (60 ms for 270 MB, threads do not help).  Here: the buffer protocol
 * gives the pointers (~0.1 us each, GIL held), then the pieces are copied by a few threads with the GIL released.
 *
 *   gather(seq, dst_address, dst_bytes, nthreads) -> (bytes copied, all pieces equally long)
 *     seq: sequence of C-contiguous float64 buffers; they land back to back at dst_address (a page-locked block of the
 *     context, sweep._stack_raw).  Raises TypeError / ValueError for anything else - the caller then takes the numpy path.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const char* src; char* dst; size_t len; } piece_t;
typedef struct { const piece_t* p; size_t lo, hi; } job_t;

static void* copy_range(void* arg) {
  const job_t* j = (const job_t*)arg;
  for (size_t i = j->lo; i < j->hi; ++i) memcpy(j->p[i].dst, j->p[i].src, j->p[i].len);
  return NULL;
}

static PyObject* gather(PyObject* self, PyObject* args) {
  PyObject* seq_in;
  unsigned long long dst_addr;
  Py_ssize_t dst_bytes;
  int nthreads = 4;
  if (!PyArg_ParseTuple(args, "OKn|i", &seq_in, &dst_addr, &dst_bytes, &nthreads)) return NULL;
  PyObject* seq = PySequence_Fast(seq_in, "gather: a sequence of buffers is expected");
  if (!seq) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
  Py_buffer* views = (Py_buffer*)malloc((size_t)(n > 0 ? n : 1) * sizeof(Py_buffer));
  piece_t* pieces = (piece_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(piece_t));
  if (!views || !pieces) { free(views); free(pieces); Py_DECREF(seq); return PyErr_NoMemory(); }
  Py_ssize_t got = 0;
  size_t off = 0;
  int ok = 1, same = 1;
  for (; got < n; ++got) {
    PyObject* it = PySequence_Fast_GET_ITEM(seq, got);
    if (PyObject_GetBuffer(it, &views[got], PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) { ok = 0; break; }
    const Py_buffer* v = &views[got];
    if (v->itemsize != 8 || !v->format || strcmp(v->format[0] == '<' || v->format[0] == '=' ? v->format + 1 : v->format, "d") != 0) {
      PyErr_SetString(PyExc_TypeError, "gather: float64 buffers expected");
      ++got;               /* this view is held too */
      ok = 0;
      break;
    }
    pieces[got].src = (const char*)v->buf;
    pieces[got].dst = (char*)(uintptr_t)dst_addr + off;
    pieces[got].len = (size_t)v->len;
    if (got > 0 && pieces[got].len != pieces[0].len) same = 0;
    off += (size_t)v->len;
  }
  if (ok && off > (size_t)dst_bytes) {
    PyErr_SetString(PyExc_ValueError, "gather: the pieces do not fit the destination");
    ok = 0;
  }
  if (ok && n > 0) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 16) nthreads = 16;
    if (off < ((size_t)8 << 20)) nthreads = 1;          /* small gathers: thread start-up costs more than it saves */
    Py_BEGIN_ALLOW_THREADS
    pthread_t th[16];
    job_t jobs[16];
    /* ranges of (about) equal bytes */
    size_t lo = 0, acc = 0;
    int started = 0;
    for (int t = 0; t < nthreads; ++t) {
      const size_t target = off / (size_t)nthreads * (size_t)(t + 1);
      size_t hi = lo;
      while (hi < (size_t)n && (t == nthreads - 1 || acc + pieces[hi].len <= target)) acc += pieces[hi++].len;
      jobs[t].p = pieces; jobs[t].lo = lo; jobs[t].hi = hi;
      lo = hi;
      if (t == nthreads - 1 || pthread_create(&th[started], NULL, copy_range, &jobs[t]) != 0) copy_range(&jobs[t]);
      else ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    Py_END_ALLOW_THREADS
  }
  for (Py_ssize_t i = 0; i < got; ++i) PyBuffer_Release(&views[i]);
  free(views);
  free(pieces);
  Py_DECREF(seq);
  if (!ok) return NULL;
  return Py_BuildValue("(nO)", (Py_ssize_t)off, same ? Py_True : Py_False);
}

static PyMethodDef methods[] = {{"gather", gather, METH_VARARGS, "gather(seq, dst_address, dst_bytes, nthreads=4) -> (bytes copied, all pieces equally long)"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_kp_gather", "host-side gather of many small float64 buffers", -1, methods};
PyMODINIT_FUNC PyInit__kp_gather(void) { return PyModule_Create(&moddef); }
