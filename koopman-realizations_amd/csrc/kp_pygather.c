/* _kp_gather: host-side marshalling helper of the Python mirror (NOT part of the C ABI, no GPU code).
 *
 * evaluate_rand_models.m:45-59 hands the sweep a cell array of data4sysid structs - tens of thousands of small trial
 * arrays (1024 systems x 11 trials x {t, y, u}).  The library wants one block per quantity (kp_traj_upload).  In a MEX
 * gateway that gather is a loop over mxGetPr pointers; from Python, np.concatenate spends ~1.7 us of interpreter-side
 * set-up per 8 KB piece and holds the GIL throughout (60 ms for 270 MB, threads do not help).  Here: the buffer protocol
 * gives the pointers (~0.1 us each, GIL held), then the pieces are copied by a few threads with the GIL released.
 *
 *   gather(seq, dst_address, dst_bytes, nthreads=4, key=None) -> (bytes copied, all pieces equally long)
 *     seq: sequence of C-contiguous float64 buffers; they land back to back at dst_address (a page-locked block of the
 *     context, sweep._stack_raw).  With `key`: seq is a sequence of sequences of dicts (systems -> trials) and the pieces are
 *     trial[key] in that order (the list comprehension that flattened it cost as much as the copy).  Raises TypeError /
 *     ValueError for anything else - the caller then takes the numpy path.
 *   trials_increasing(seq, k, nthreads=4, key=None) -> (seams exactly at the trial joins, all pieces equally long)
 *     the seam test of get_snapshotPairs on the time vectors, read-only (below).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#ifdef KP_HAVE_NUMPY            /* the Makefile sets it when numpy's headers are installed: exact ndarrays then skip the */
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION   /* buffer protocol (68 ns per array, 0.7 ms per quantity of the sweep) */
#include <numpy/arrayobject.h>
#endif
#include <pthread.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const char* src; char* dst; size_t len; } piece_t;

/* The pieces of a call: buffer views held until release_pieces. */
typedef struct {
  Py_buffer* views;                                     /* obj == NULL: the piece is held through owners[] instead */
  PyObject** owners;
  piece_t* pieces;
  Py_ssize_t n, cap;
  size_t total;
  int same;
} pieces_t;

#ifdef KP_HAVE_NUMPY
static int have_numpy = 0;
#endif

static void release_pieces(pieces_t* P) {
  for (Py_ssize_t i = 0; i < P->n; ++i) {
    if (P->owners[i]) Py_DECREF(P->owners[i]);
    else PyBuffer_Release(&P->views[i]);
  }
  free(P->views);
  free(P->owners);
  free(P->pieces);
  P->views = NULL; P->owners = NULL; P->pieces = NULL; P->n = 0;
}

static int add_piece(pieces_t* P, PyObject* it, const char* who) {
  if (P->n == P->cap) {
    const Py_ssize_t cap = P->cap ? 2 * P->cap : 1024;
    Py_buffer* v = (Py_buffer*)realloc(P->views, (size_t)cap * sizeof(Py_buffer));
    if (v) P->views = v;
    piece_t* p = (piece_t*)realloc(P->pieces, (size_t)cap * sizeof(piece_t));
    if (p) P->pieces = p;
    PyObject** o = (PyObject**)realloc(P->owners, (size_t)cap * sizeof(PyObject*));
    if (o) P->owners = o;
    if (!v || !p || !o) { PyErr_NoMemory(); return -1; }
    P->cap = cap;
  }
#ifdef KP_HAVE_NUMPY
  if (have_numpy && PyArray_CheckExact(it)) {
    PyArrayObject* a = (PyArrayObject*)it;
    if (PyArray_TYPE(a) == NPY_DOUBLE && PyArray_IS_C_CONTIGUOUS(a) && PyArray_ISALIGNED(a) && PyArray_ISNOTSWAPPED(a)) {
      Py_INCREF(it);
      P->owners[P->n] = it;
      piece_t* p = &P->pieces[P->n];
      ++P->n;
      p->src = (const char*)PyArray_DATA(a);
      p->dst = NULL;
      p->len = (size_t)PyArray_NBYTES(a);
      if (P->n > 1 && p->len != P->pieces[0].len) P->same = 0;
      P->total += p->len;
      return 0;
    }
  }
#endif
  Py_buffer* v = &P->views[P->n];
  if (PyObject_GetBuffer(it, v, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) return -1;
  P->owners[P->n] = NULL;
  ++P->n;                                               /* held from here on */
  if (v->itemsize != 8 || !v->format || strcmp(v->format[0] == '<' || v->format[0] == '=' ? v->format + 1 : v->format, "d") != 0) {
    PyErr_Format(PyExc_TypeError, "%s: float64 buffers expected", who);
    return -1;
  }
  piece_t* p = &P->pieces[P->n - 1];
  p->src = (const char*)v->buf;
  p->dst = NULL;
  p->len = (size_t)v->len;
  if (P->n > 1 && p->len != P->pieces[0].len) P->same = 0;
  P->total += p->len;
  return 0;
}

/* seq: flat sequence of buffers (key NULL / None), or sequence of sequences of dicts whose [key] are the buffers */
static int acquire(pieces_t* P, PyObject* seq_in, PyObject* key, const char* who) {
  memset(P, 0, sizeof(*P));
  P->same = 1;
  PyObject* seq = PySequence_Fast(seq_in, "a sequence is expected");
  if (!seq) return -1;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
  int rc = 0;
  if (!key || key == Py_None) {
    for (Py_ssize_t i = 0; i < n && rc == 0; ++i) rc = add_piece(P, PySequence_Fast_GET_ITEM(seq, i), who);
  } else {
    for (Py_ssize_t i = 0; i < n && rc == 0; ++i) {
      PyObject* inner = PySequence_Fast(PySequence_Fast_GET_ITEM(seq, i), "a sequence of trials is expected");
      if (!inner) { rc = -1; break; }
      const Py_ssize_t m = PySequence_Fast_GET_SIZE(inner);
      for (Py_ssize_t j = 0; j < m && rc == 0; ++j) {
        PyObject* d = PySequence_Fast_GET_ITEM(inner, j);
        if (!PyDict_Check(d)) { PyErr_Format(PyExc_TypeError, "%s: trials must be dicts", who); rc = -1; break; }
        PyObject* it = PyDict_GetItemWithError(d, key);             /* borrowed */
        if (!it) { if (!PyErr_Occurred()) PyErr_SetObject(PyExc_KeyError, key); rc = -1; break; }
        rc = add_piece(P, it, who);
      }
      Py_DECREF(inner);
    }
  }
  Py_DECREF(seq);
  if (rc != 0) release_pieces(P);
  return rc;
}

static int clamp_threads(int nthreads, size_t total) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 16) nthreads = 16;
  if (total < ((size_t)8 << 20)) nthreads = 1;          /* small jobs: thread start-up costs more than it saves */
  return nthreads;
}

typedef struct { const piece_t* p; size_t lo, hi; } job_t;

static void* copy_range(void* arg) {
  const job_t* j = (const job_t*)arg;
  for (size_t i = j->lo; i < j->hi; ++i) memcpy(j->p[i].dst, j->p[i].src, j->p[i].len);
  return NULL;
}

static PyObject* gather(PyObject* self, PyObject* args) {
  PyObject *seq_in, *key = NULL;
  unsigned long long dst_addr;
  Py_ssize_t dst_bytes;
  int nthreads = 4;
  if (!PyArg_ParseTuple(args, "OKn|iO", &seq_in, &dst_addr, &dst_bytes, &nthreads, &key)) return NULL;
  pieces_t P;
  if (acquire(&P, seq_in, key, "gather") != 0) return NULL;
  if (P.total > (size_t)dst_bytes) {
    release_pieces(&P);
    PyErr_SetString(PyExc_ValueError, "gather: the pieces do not fit the destination");
    return NULL;
  }
  size_t off = 0;
  for (Py_ssize_t i = 0; i < P.n; ++i) { P.pieces[i].dst = (char*)(uintptr_t)dst_addr + off; off += P.pieces[i].len; }
  if (P.n > 0) {
    nthreads = clamp_threads(nthreads, P.total);
    const size_t n = (size_t)P.n;
    Py_BEGIN_ALLOW_THREADS
    pthread_t th[16];
    job_t jobs[16];
    /* ranges of (about) equal bytes */
    size_t lo = 0, acc = 0;
    int started = 0;
    for (int t = 0; t < nthreads; ++t) {
      const size_t target = P.total / (size_t)nthreads * (size_t)(t + 1);
      size_t hi = lo;
      while (hi < n && (t == nthreads - 1 || acc + P.pieces[hi].len <= target)) acc += P.pieces[hi++].len;
      jobs[t].p = P.pieces; jobs[t].lo = lo; jobs[t].hi = hi;
      lo = hi;
      if (t == nthreads - 1 || pthread_create(&th[started], NULL, copy_range, &jobs[t]) != 0) copy_range(&jobs[t]);
      else ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    Py_END_ALLOW_THREADS
  }
  const size_t total = P.total;
  const int same = P.same;
  release_pieces(&P);
  return Py_BuildValue("(nO)", (Py_ssize_t)total, same ? Py_True : Py_False);
}

/* trials_increasing: the time vectors of the trials, k consecutive ones per system.  This is get_snapshotPairs' seam test
 * (Ksysid.m:948: a pair is dropped where before.t >= after.t) asked of the whole population at once: the device forms the
 * pairs itself and only needs to know that seams lie exactly at the trial joins - every trial strictly increasing, the
 * clock NOT running on from a trial to the next one of the same system.  Read-only, threads, GIL released. */
typedef struct { const piece_t* p; size_t lo, hi, k; int bad; } chk_t;

/* non-zero when some entry is not followed by a larger one (NaN counts as such) */
static long long count_not_increasing(const double* __restrict t, size_t n) {
  long long c = 0;
  size_t q = 0;
#if defined(__SSE2__)
  int all = 3;                                           /* AND of the compare masks: one test at the end */
  for (; q + 9 <= n; q += 8) {
    const int m0 = _mm_movemask_pd(_mm_cmplt_pd(_mm_loadu_pd(t + q), _mm_loadu_pd(t + q + 1)));
    const int m1 = _mm_movemask_pd(_mm_cmplt_pd(_mm_loadu_pd(t + q + 2), _mm_loadu_pd(t + q + 3)));
    const int m2 = _mm_movemask_pd(_mm_cmplt_pd(_mm_loadu_pd(t + q + 4), _mm_loadu_pd(t + q + 5)));
    const int m3 = _mm_movemask_pd(_mm_cmplt_pd(_mm_loadu_pd(t + q + 6), _mm_loadu_pd(t + q + 7)));
    all &= m0 & m1 & m2 & m3;
  }
  c += all != 3;
#endif
  for (; q + 1 < n; ++q) c += (long long)!(t[q] < t[q + 1]);
  return c;
}

static void* check_range(void* arg) {
  chk_t* j = (chk_t*)arg;
  int bad = 0;
  for (size_t i = j->lo; i < j->hi && !bad; ++i) {
    const double* t = (const double*)j->p[i].src;
    const size_t n = j->p[i].len / 8;
    bad |= count_not_increasing(t, n) != 0;
    if ((i % j->k) != j->k - 1 && n > 0) {
      const double* tn = (const double*)j->p[i + 1].src;
      if (j->p[i + 1].len >= 8 && t[n - 1] < tn[0]) bad = 1;
    }
  }
  j->bad = bad;
  return NULL;
}

static PyObject* trials_increasing(PyObject* self, PyObject* args) {
  PyObject *seq_in, *key = NULL;
  Py_ssize_t k;
  int nthreads = 4;
  if (!PyArg_ParseTuple(args, "On|iO", &seq_in, &k, &nthreads, &key)) return NULL;
  pieces_t P;
  if (acquire(&P, seq_in, key, "trials_increasing") != 0) return NULL;
  if (k < 1 || P.n % k != 0) {
    release_pieces(&P);
    PyErr_SetString(PyExc_ValueError, "trials_increasing: k trials per system expected");
    return NULL;
  }
  int bad = 0;
  if (P.n > 0) {
    nthreads = clamp_threads(nthreads, P.total);
    const size_t n = (size_t)P.n;
    Py_BEGIN_ALLOW_THREADS
    pthread_t th[16];
    chk_t jobs[16];
    int started = 0;
    for (int t = 0; t < nthreads; ++t) {
      jobs[t].p = P.pieces; jobs[t].k = (size_t)k; jobs[t].bad = 0;
      jobs[t].lo = n * (size_t)t / (size_t)nthreads;
      jobs[t].hi = n * (size_t)(t + 1) / (size_t)nthreads;
      if (t == nthreads - 1 || pthread_create(&th[started], NULL, check_range, &jobs[t]) != 0) check_range(&jobs[t]);
      else ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    for (int t = 0; t < nthreads; ++t) bad |= jobs[t].bad;
    Py_END_ALLOW_THREADS
  }
  const int same = P.same;
  release_pieces(&P);
  return Py_BuildValue("(OO)", bad ? Py_False : Py_True, same ? Py_True : Py_False);
}

static PyMethodDef methods[] = {{"gather", gather, METH_VARARGS,
                                 "gather(seq, dst_address, dst_bytes, nthreads=4, key=None) -> (bytes copied, all pieces equally long)"},
                                {"trials_increasing", trials_increasing, METH_VARARGS,
                                 "trials_increasing(seq, k, nthreads=4, key=None) -> (seams exactly at the trial joins, all pieces equally long)"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_kp_gather", "host-side gather of many small float64 buffers", -1, methods};
PyMODINIT_FUNC PyInit__kp_gather(void) {
#ifdef KP_HAVE_NUMPY
  if (_import_array() == 0) have_numpy = 1;             /* no numpy at run time: the buffer protocol serves everything */
  else PyErr_Clear();
#endif
  PyObject* m = PyModule_Create(&moddef);
#ifdef KP_HAVE_NUMPY
  if (m) PyModule_AddIntConstant(m, "numpy_fast_path", have_numpy);
#else
  if (m) PyModule_AddIntConstant(m, "numpy_fast_path", 0);
#endif
  return m;
}
