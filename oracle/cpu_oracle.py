"""oracle/cpu_oracle.py -- numpy front end of the CPU restatement.

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg.  The product package
(mms_answer_selection_amd) never imports this module.  PARITY UNPINNED -- see
the header of oracle/mms_oracle.c.

Each wrapper takes/returns C-contiguous numpy arrays and calls the C function
of the same name in libmms_oracle.so (built by oracle/Makefile).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmms_oracle.so")
_lib = None


def build(force=False):
    """Compile libmms_oracle.so with gcc/g++ (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in
            ("mms_oracle.c", "mms_oracle_impl.h", "mms_oracle_rank.cpp", "Makefile")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.oracle_time_simcross_fwd_bwd_f32.restype = C.c_double
        for n in ("oracle_map", "oracle_mrr", "oracle_auc", "oracle_rank_accuracy"):
            for sfx, rt in (("_f32", C.c_float), ("_f64", C.c_double)):
                if hasattr(_lib, n + sfx):
                    getattr(_lib, n + sfx).restype = rt
    return _lib


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f32", C.c_float
    if dtype == np.float64:
        return "_f64", C.c_double
    raise TypeError("oracle instantiates float and double only (common.hpp:41-44)")


def _p(x):
    return None if x is None else x.ctypes.data_as(C.c_void_p)


def _c(x, dtype):
    return None if x is None else np.ascontiguousarray(x, dtype=dtype)


def simcross_forward(mode, q, a, W=None, bias=None):
    """-> (top (N,M|1,W1,W2), norm0 (N,W1), norm1 (N,W2)).  sim_cross_layer.cpp:83-163."""
    dt = q.dtype
    sfx, _ = _sfx(dt)
    q, a, W, bias = _c(q, dt), _c(a, dt), _c(W, dt), _c(bias, dt)
    N, W1, D = q.shape
    W2 = a.shape[1]
    M = W.shape[0] if mode == 2 else 1
    top = np.zeros((N, M, W1, W2), dt)
    n0 = np.zeros((N, W1), dt)
    n1 = np.zeros((N, W2), dt)
    getattr(lib(), "oracle_simcross_forward" + sfx)(
        C.c_int(mode), N, W1, W2, D, M, _p(q), _p(a), _p(W), _p(bias),
        _p(top), _p(n0), _p(n1))
    return top, n0, n1


def simcross_backward(mode, q, a, top, top_diff, W=None, bias_term=False,
                      norm0=None, norm1=None, dbias_in=None,
                      propagate_down=(True, True)):
    """-> (dq, da, dW, dbias).  sim_cross_layer.cpp:166-307."""
    dt = q.dtype
    sfx, _ = _sfx(dt)
    q, a, W = _c(q, dt), _c(a, dt), _c(W, dt)
    top, top_diff = _c(top, dt), _c(top_diff, dt)
    norm0, norm1 = _c(norm0, dt), _c(norm1, dt)
    N, W1, D = q.shape
    W2 = a.shape[1]
    M = W.shape[0] if mode == 2 else 1
    dq = np.full(q.shape, np.nan, dt)   # backward must zero them itself
    da = np.full(a.shape, np.nan, dt)
    dW = np.full((M, D, D), np.nan, dt) if mode == 2 else None
    dbias = None
    if mode == 2 and bias_term:
        dbias = (np.zeros((M, W1, W2), dt) if dbias_in is None
                 else np.array(dbias_in, dtype=dt, copy=True))
    getattr(lib(), "oracle_simcross_backward" + sfx)(
        C.c_int(mode), N, W1, W2, D, M, _p(q), _p(a), _p(W), int(bool(bias_term)),
        _p(top), _p(top_diff), _p(norm0), _p(norm1),
        int(bool(propagate_down[0])), int(bool(propagate_down[1])),
        _p(dq), _p(da), _p(dW), _p(dbias))
    return dq, da, dW, dbias


def simmatrix_forward(q, a, W):
    """-> (top (N,1), scratch (N,K2) = what lands in bottom[1].diff).  sim_matrix_layer.cpp:53-65."""
    dt = q.dtype
    sfx, _ = _sfx(dt)
    q, a, W = _c(q, dt), _c(a, dt), _c(W, dt)
    N = q.shape[0]
    K1 = int(np.prod(q.shape[1:]))
    K2 = int(np.prod(a.shape[1:]))
    top = np.zeros((N, 1), dt)
    scratch = np.zeros((N, K2), dt)
    getattr(lib(), "oracle_simmatrix_forward" + sfx)(
        N, K1, K2, _p(q), _p(a), _p(W), _p(top), _p(scratch))
    return top, scratch


def simmatrix_backward(q, a, W, top_diff, dW_in=None, param_propagate_down=True,
                       propagate_down=(True, True)):
    """-> (dq, da, dW).  sim_matrix_layer.cpp:68-95 (dW accumulates)."""
    dt = q.dtype
    sfx, _ = _sfx(dt)
    q, a, W, top_diff = _c(q, dt), _c(a, dt), _c(W, dt), _c(top_diff, dt)
    N = q.shape[0]
    K1 = int(np.prod(q.shape[1:]))
    K2 = int(np.prod(a.shape[1:]))
    dq = np.zeros((N, K1), dt)
    da = np.zeros((N, K2), dt)
    dW = np.zeros((K1, K2), dt) if dW_in is None else np.array(dW_in, dtype=dt, copy=True)
    getattr(lib(), "oracle_simmatrix_backward" + sfx)(
        N, K1, K2, _p(q), _p(a), _p(W), _p(top_diff), int(bool(param_propagate_down)),
        int(bool(propagate_down[0])), int(bool(propagate_down[1])),
        _p(dq), _p(da), _p(dW))
    return dq, da, dW


def pairrank_forward(a, b, y, margin=1.0):
    """-> (loss scalar, ordered, similar).  pair_rank_loss_layer.cpp:26-52."""
    dt = a.dtype
    sfx, ct = _sfx(dt)
    a, b, y = _c(a, dt), _c(b, dt), _c(y, dt)
    ordered = np.zeros(a.shape, dt)
    similar = np.zeros(a.shape, dt)
    loss = np.zeros((1,), dt)
    getattr(lib(), "oracle_pairrank_forward" + sfx)(
        int(a.size), ct(margin), _p(a), _p(b), _p(y), _p(ordered), _p(similar), _p(loss))
    return loss[0], ordered, similar


def pairrank_backward(y, ordered, similar, top_diff=1.0, propagate_down=(True, True)):
    """-> (da, db).  pair_rank_loss_layer.cpp:55-84."""
    dt = y.dtype
    sfx, ct = _sfx(dt)
    y, ordered, similar = _c(y, dt), _c(ordered, dt), _c(similar, dt)
    da = np.zeros(y.shape, dt)
    db = np.zeros(y.shape, dt)
    getattr(lib(), "oracle_pairrank_backward" + sfx)(
        int(y.size), ct(top_diff), _p(y), _p(ordered), _p(similar),
        int(bool(propagate_down[0])), int(bool(propagate_down[1])), _p(da), _p(db))
    return da, db


def rank_accuracy(a, b, label):
    dt = a.dtype
    sfx, _ = _sfx(dt)
    a, b, label = _c(a, dt), _c(b, dt), _c(label, dt)
    return getattr(lib(), "oracle_rank_accuracy" + sfx)(int(a.size), _p(a), _p(b), _p(label))


def map_score(prob, label, group, fixed_axis=1):
    """prob (n, fixed_axis+1).  -> (MAP, effective groups).  map_layer.cpp:41-100."""
    dt = prob.dtype
    sfx, _ = _sfx(dt)
    prob, label, group = _c(prob, dt), _c(label, dt), _c(group, dt)
    eff = C.c_int(0)
    v = getattr(lib(), "oracle_map" + sfx)(
        int(label.size), int(fixed_axis), _p(prob), _p(label), _p(group), C.byref(eff))
    return dt.type(v), eff.value


def mrr_score(prob, label, group, fixed_axis=1):
    """-> (MRR, effective groups).  mrr_layer.cpp:38-79."""
    dt = prob.dtype
    sfx, _ = _sfx(dt)
    prob, label, group = _c(prob, dt), _c(label, dt), _c(group, dt)
    eff = C.c_int(0)
    v = getattr(lib(), "oracle_mrr" + sfx)(
        int(label.size), int(fixed_axis), _p(prob), _p(label), _p(group), C.byref(eff))
    return dt.type(v), eff.value


def auc_score(prob, label, fixed_axis=1):
    """prob (n, dim) float32.  auc_layer.cpp:47-136."""
    prob, label = _c(prob, np.float32), _c(label, np.float32)
    return np.float32(lib().oracle_auc_f32(
        int(label.size), int(prob.shape[1]), int(fixed_axis), _p(prob), _p(label)))


def auc_score_nd(prob, label, axis=1, fixed_axis=1, ignore_label=None):
    """prob with its class axis at `axis`, label of the remaining shape.  auc_layer.cpp:47-136, general indexing."""
    prob, label = _c(prob, np.float32), _c(label, np.float32)
    axis = axis % prob.ndim
    outer = int(np.prod(prob.shape[:axis])) if axis else 1
    inner = int(np.prod(prob.shape[axis + 1:])) if axis + 1 < prob.ndim else 1
    f = lib().oracle_auc_nd_f32
    f.restype = C.c_float
    return np.float32(f(outer, int(prob.shape[axis]), inner, int(fixed_axis), _p(prob), _p(label),
                        int(ignore_label is not None), int(ignore_label or 0)))


def time_simcross_fwd_bwd(mode, q, a, top_diff, W=None, bias=None, iters=1):
    """Seconds for `iters` forward+backward passes, one thread (cpu_baseline leg)."""
    dt = np.float32
    q, a, W, bias, top_diff = _c(q, dt), _c(a, dt), _c(W, dt), _c(bias, dt), _c(top_diff, dt)
    N, W1, D = q.shape
    W2 = a.shape[1]
    M = W.shape[0] if mode == 2 else 1
    top = np.zeros((N, M, W1, W2), dt)
    n0 = np.zeros((N, W1), dt)
    n1 = np.zeros((N, W2), dt)
    dq = np.zeros_like(q)
    da = np.zeros_like(a)
    dW = np.zeros((M, D, D), dt)
    dbias = np.zeros((M, W1, W2), dt)
    return lib().oracle_time_simcross_fwd_bwd_f32(
        C.c_int(mode), N, W1, W2, D, M, _p(q), _p(a), _p(W), _p(bias), _p(top_diff),
        _p(top), _p(n0), _p(n1), _p(dq), _p(da), _p(dW), _p(dbias), int(iters))


def embed_forward(index, weight, bias=None):
    """index (...) float, weight (K,N) -> top (..., N).  embed_layer.cpp:135-152."""
    dt = weight.dtype
    sfx, _ = _sfx(dt)
    index, weight, bias = _c(index, dt), _c(weight, dt), _c(bias, dt)
    M, N = int(index.size), int(weight.shape[1])
    top = np.zeros(index.shape + (N,), dt)
    getattr(lib(), "oracle_embed_forward" + sfx)(M, N, _p(index), _p(weight), _p(bias), _p(top))
    return top


def embed_backward(index, top_diff, weight_diff_in, bias_diff_in=None):
    """-> (weight_diff, bias_diff), both ACCUMULATED into copies of the inputs.  embed_layer.cpp:155-180."""
    dt = top_diff.dtype
    sfx, _ = _sfx(dt)
    index, top_diff = _c(index, dt), _c(top_diff, dt)
    wd = np.array(weight_diff_in, dtype=dt, copy=True)
    bd = None if bias_diff_in is None else np.array(bias_diff_in, dtype=dt, copy=True)
    getattr(lib(), "oracle_embed_backward" + sfx)(int(index.size), int(wd.shape[1]), _p(index),
                                                  _p(top_diff), _p(wd), _p(bd))
    return wd, bd


def time_simcross_fwd_bwd_mt(mode, q, a, top_diff, iters=1, threads=1):
    """Seconds for `iters` fwd+bwd passes with the pairs dealt to `threads` OpenMP threads
    (modes 0 / 1; courtesy upper bound next to the single-threaded reference loop)."""
    dt = np.float32
    q, a, top_diff = _c(q, dt), _c(a, dt), _c(top_diff, dt)
    N, W1, D = q.shape
    W2 = a.shape[1]
    top = np.zeros((N, 1, W1, W2), dt)
    n0 = np.zeros((N, W1), dt)
    n1 = np.zeros((N, W2), dt)
    dq = np.zeros_like(q)
    da = np.zeros_like(a)
    f = lib().oracle_time_simcross_fwd_bwd_mt_f32
    f.restype = C.c_double
    return f(C.c_int(mode), N, W1, W2, D, _p(q), _p(a), _p(top_diff), _p(top), _p(n0), _p(n1),
             _p(dq), _p(da), int(iters), int(threads))
