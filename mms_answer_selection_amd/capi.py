"""ctypes binding of libmms_hip.so (include/mms.h) for torch device tensors.

torch is plumbing here: it owns device memory and the HIP stream; every
computation goes through the C ABI.  There is NO fallback: if the library is
missing or a tensor is not on the GPU the call raises.
"""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmms_hip.so")

MMS_OK = 0
_lib = None

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

_SIGNATURES = {
    "mms_version": (C.c_int, []),
    "mms_error_string": (C.c_char_p, [_i]),
    "mms_simcross_workspace_bytes": (_sz, [_i] * 6),
    "mms_simcross_forward_f32": (_i, [_i] * 6 + [_vp] * 8 + [_sz, _vp]),
    "mms_simcross_backward_f32": (_i, [_i] * 6 + [_vp] * 3 + [_i] + [_vp] * 4 + [_i, _i] + [_vp] * 5 + [_sz, _vp]),
    "mms_simcross_forward_backward_f32": (_i, [_i] * 6 + [_vp] * 13 + [_sz, _vp]),
    "mms_simmatrix_workspace_bytes": (_sz, [_i] * 3),
    "mms_embed_simcross_forward_f32": (_i, [_i] * 6 + [_vp] * 8),
    "mms_embed_simcross_bilinear_forward_f32": (_i, [_i] * 6 + [_vp] * 8),
    "mms_simmatrix_forward_f32": (_i, [_i] * 3 + [_vp] * 6),
    "mms_simmatrix_forward_ws_f32": (_i, [_i] * 3 + [_vp] * 6 + [_sz, _vp]),
    "mms_simmatrix_forward_f16": (_i, [_i] * 3 + [_vp] * 5 + [_sz, _vp]),
    "mms_simmatrix_forward_train_f16": (_i, [_i] * 3 + [_vp] * 6 + [_sz, _vp]),
    "mms_simmatrix_backward_f16": (_i, [_i] * 3 + [_vp] * 9 + [_sz, _vp]),
    "mms_set_matrix_mode": (_i, [_i]),
    "mms_get_matrix_mode": (_i, []),
    "mms_simmatrix_backward_f32": (_i, [_i] * 3 + [_vp] * 4 + [_i] * 3 + [_vp] * 4 + [_sz, _vp]),
    "mms_simmatrix_backward_cached_f32": (_i, [_i] * 3 + [_vp] * 5 + [_i] * 3 + [_vp] * 4 + [_sz, _vp]),
    "mms_pairrank_workspace_bytes": (_sz, [_i]),
    "mms_pairrank_forward_f32": (_i, [_i, _f] + [_vp] * 7 + [_sz, _vp]),
    "mms_pairrank_backward_f32": (_i, [_i, _f] + [_vp] * 3 + [_i, _i] + [_vp] * 3),
    "mms_triplet_workspace_bytes": (_sz, [_i]),
    "mms_triplet_workspace_init": (_i, [_vp, _sz, _vp]),
    "mms_triplet_simmatrix_workspace_bytes": (_sz, [_i, _i, _i]),
    "mms_triplet_simmatrix_step_f32": (_i, [_i, _i, _i, _f, _f] + [_vp] * 12 + [_vp, _sz, _vp]),
    "mms_triplet_euclid_step_f32": (_i, [_i, _i, _f, _f] + [_vp] * 11 + [_sz, _vp]),
    "mms_simcross_euclid_forward_f16": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    "mms_simcross_euclid_forward_backward_f16": (_i, [_i, _i] + [_vp] * 7),
    "mms_simcross_cosine_forward_f16": (_i, [_i, _i] + [_vp] * 5 + [_vp]),
    "mms_simcross_cosine_forward_backward_f16": (_i, [_i, _i] + [_vp] * 8 + [_vp]),
    "mms_embed_workspace_bytes": (_sz, [_i, _i]),
    "mms_embed_forward_f32": (_i, [_i, _i, _i] + [_vp] * 5),
    "mms_embed_backward_f32": (_i, [_i, _i, _i] + [_vp] * 5 + [_sz, _vp]),
    "mms_embed_backward_pair_f32": (_i, [_i, _i, _i, _i] + [_vp] * 6 + [_vp, _sz, _vp]),
    "mms_embed_pair_index_supported": (_i, [_i, _i, _i]),
    "mms_embed_forward_pair_f32": (_i, [_i, _i, _i, _i] + [_vp] * 6 + [_vp, _sz, _vp]),
    "mms_embed_backward_pair_indexed_f32": (_i, [_i, _i, _i, _i] + [_vp] * 6 + [_vp, _sz, _vp]),
    "mms_feed_gather_rows_f32": (_i, [_i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "mms_simcross_workspace_bytes_f64": (_sz, [_i] * 6),
    "mms_simcross_forward_f64": (_i, [_i] * 6 + [_vp] * 8 + [_sz, _vp]),
    "mms_simcross_backward_f64": (_i, [_i] * 6 + [_vp] * 3 + [_i] + [_vp] * 4 + [_i, _i] + [_vp] * 5 + [_sz, _vp]),
    "mms_simmatrix_forward_f64": (_i, [_i] * 3 + [_vp] * 6),
    "mms_simmatrix_backward_f64": (_i, [_i] * 3 + [_vp] * 4 + [_i] * 3 + [_vp] * 4),
    "mms_pairrank_forward_f64": (_i, [_i, C.c_double] + [_vp] * 7),
    "mms_pairrank_backward_f64": (_i, [_i, C.c_double] + [_vp] * 3 + [_i, _i] + [_vp] * 3),
    "mms_set_euclid_backward_mode": (_i, [_i]),
    "mms_get_euclid_backward_mode": (_i, []),
    "mms_null_launch": (_i, [_i, _vp]),
    "mms_simcross_forward_block_f32": (_i, [_vp, _vp]),
    "mms_simcross_backward_block_f32": (_i, [_vp, _vp]),
    "mms_dot_f32": (_i, [_i, _vp, _vp, _vp, _vp]),
    "mms_split_backward_f32": (_i, [_i, _i, _vp, _vp, _vp]),
    "mms_dot_f64": (_i, [_i, _vp, _vp, _vp, _vp]),
    "mms_set_f16_distance_mode": (_i, [_i]),
    "mms_get_f16_distance_mode": (_i, []),
    "mms_set_rank_tie_mode": (_i, [_i]),
    "mms_get_rank_tie_mode": (_i, []),
    "mms_set_loss_sum_mode": (_i, [_i]),
    "mms_get_loss_sum_mode": (_i, []),
    "mms_set_triplet_finish_mode": (_i, [_i]),
    "mms_get_triplet_finish_mode": (_i, []),
    "mms_set_pairrank_hinge_mode": (_i, [_i]),
    "mms_get_pairrank_hinge_mode": (_i, []),
    "mms_rank_workspace_bytes": (_sz, [_i]),
    "mms_rank_map_mrr_f32": (_i, [_i, _i] + [_vp] * 7 + [_sz, _vp]),
    "mms_rank_auc_f32": (_i, [_i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "mms_rank_auc_nd_f32": (_i, [_i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "mms_rank_accuracy_f32": (_i, [_i] + [_vp] * 5 + [_sz, _vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class MMSError(RuntimeError):
    pass


MMS_VERSION = 212      # include/mms.h


def lib():
    """The loaded C-ABI library; raises (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MMSError(
                "libmms_hip.so is not built (%s). Run `python -m mms_answer_selection_amd.build` "
                "or __graft_entry__.build(); there is no CPU fallback." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the header and library diverge
            fn.restype = res
            fn.argtypes = args
        if l.mms_version() != MMS_VERSION:            # include/mms.h: a host built against another header must refuse
            raise MMSError("libmms_hip.so reports ABI version %d, this binding was written for %d: rebuild "
                           "(python -m mms_answer_selection_amd.build --force)" % (l.mms_version(), MMS_VERSION))
        _lib = l
    return _lib


def check(rc, what):
    if rc != MMS_OK:
        raise MMSError("%s failed: %s (code %d)" % (what, lib().mms_error_string(rc).decode(), rc))


def _ptr(t, name, allow_none=False, dtype=torch.float32):
    if t is None:
        if allow_none:
            return None
        raise MMSError("%s: tensor required" % name)
    if not t.is_cuda:
        raise MMSError("%s must live in GPU memory (got %s); the HIP path has no CPU fallback"
                       % (name, t.device))
    if t.dtype != dtype:
        raise MMSError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise MMSError("%s must be contiguous" % name)
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class Workspace:
    """Grow-only device scratch, owned by the caller of the C ABI (a layer)."""

    def __init__(self):
        self.buf = None
        self._retired = []          # superseded buffers stay alive: a captured hipGraph or work still queued on
                                    # another stream may hold their address (they are freed with the Workspace)

    def get(self, nbytes, device):
        if nbytes == 0:
            return None, 0
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            if self.buf is not None:
                self._retired.append(self.buf)
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self.buf.data_ptr(), self.buf.numel()


# Calls that are not given a Workspace share this one.  It only grows, and what it outgrows is retired, not freed;
# callers that capture graphs or use several streams should still pass their own (`ws=`): the library keeps
# intermediates in it BETWEEN its own launches of one call, so two calls in flight must not share a workspace.
_default_ws = Workspace()


def simcross_workspace_bytes(mode, N, W1, W2, D, M):
    return lib().mms_simcross_workspace_bytes(mode, N, W1, W2, D, M)


def simcross_forward(mode, q, a, top, W=None, bias=None, norm0=None, norm1=None, ws=None):
    N, W1, D = q.shape
    W2 = a.shape[1]
    M = W.shape[0] if mode == 2 else 1
    wsp, wsb = (ws or _default_ws).get(simcross_workspace_bytes(mode, N, W1, W2, D, M), q.device)
    check(lib().mms_simcross_forward_f32(
        mode, N, W1, W2, D, M, _ptr(q, "q"), _ptr(a, "a"), _ptr(W, "W", True),
        _ptr(bias, "bias", True), _ptr(top, "top"), _ptr(norm0, "norm0", True),
        _ptr(norm1, "norm1", True), wsp, wsb, _stream()), "mms_simcross_forward_f32")


def simcross_backward(mode, q, a, top, top_diff, dq, da, W=None, bias_term=False, norm0=None,
                      norm1=None, dW=None, dbias=None, propagate_down=(True, True), ws=None):
    N, W1, D = q.shape
    W2 = a.shape[1]
    M = W.shape[0] if mode == 2 else 1
    wsp, wsb = (ws or _default_ws).get(simcross_workspace_bytes(mode, N, W1, W2, D, M), q.device)
    check(lib().mms_simcross_backward_f32(
        mode, N, W1, W2, D, M, _ptr(q, "q"), _ptr(a, "a"), _ptr(W, "W", True), int(bool(bias_term)),
        _ptr(top, "top"), _ptr(top_diff, "top_diff"), _ptr(norm0, "norm0", True),
        _ptr(norm1, "norm1", True), int(bool(propagate_down[0])), int(bool(propagate_down[1])),
        _ptr(dq, "dq"), _ptr(da, "da"), _ptr(dW, "dW", True), _ptr(dbias, "dbias", True),
        wsp, wsb, _stream()), "mms_simcross_backward_f32")


def simcross_forward_backward(mode, q, a, top_diff, top, dq, da, W=None, bias=None, norm0=None,
                              norm1=None, dW=None, dbias=None, ws=None):
    N, W1, D = q.shape
    W2 = a.shape[1]
    M = W.shape[0] if mode == 2 else 1
    wsp, wsb = (ws or _default_ws).get(simcross_workspace_bytes(mode, N, W1, W2, D, M), q.device)
    check(lib().mms_simcross_forward_backward_f32(
        mode, N, W1, W2, D, M, _ptr(q, "q"), _ptr(a, "a"), _ptr(W, "W", True),
        _ptr(bias, "bias", True), _ptr(top_diff, "top_diff"), _ptr(top, "top"),
        _ptr(norm0, "norm0", True), _ptr(norm1, "norm1", True), _ptr(dq, "dq"), _ptr(da, "da"),
        _ptr(dW, "dW", True), _ptr(dbias, "dbias", True), wsp, wsb, _stream()),
        "mms_simcross_forward_backward_f32")


def embed_simcross_forward(mode, index_q, index_a, weight, top, norm0=None, norm1=None, embed_bias=None):
    """top = SimCross(Embed(index_q), Embed(index_a)), dist_mode 0 / 1, the gather fused into the loads;
    embed_bias (D) = the Embed layers' bias blob, if they have one."""
    N, W1 = index_q.shape[0], index_q.shape[1]
    W2 = index_a.shape[1]
    K, D = weight.shape
    check(lib().mms_embed_simcross_forward_f32(
        mode, N, W1, W2, D, K, _ptr(index_q, "index_q"), _ptr(index_a, "index_a"), _ptr(weight, "weight"),
        _ptr(embed_bias, "embed_bias", True), _ptr(top, "top"), _ptr(norm0, "norm0", True),
        _ptr(norm1, "norm1", True), _stream()), "mms_embed_simcross_forward_f32")


def embed_simcross_bilinear_forward(index_q, index_a, weight, W, bias, top, embed_bias=None):
    """top (N,M,W1,W2) = SimCross dist_mode 2 of (Embed(index_q), Embed(index_a)), one launch (word grids)."""
    N, W1 = index_q.shape[0], index_q.shape[1]
    W2 = index_a.shape[1]
    K, D = weight.shape
    M = W.shape[0]
    check(lib().mms_embed_simcross_bilinear_forward_f32(
        N, W1, W2, D, M, K, _ptr(index_q, "index_q"), _ptr(index_a, "index_a"), _ptr(weight, "weight"),
        _ptr(embed_bias, "embed_bias", True), _ptr(W, "W"), _ptr(bias, "bias", True), _ptr(top, "top"), _stream()),
        "mms_embed_simcross_bilinear_forward_f32")


def simmatrix_forward(q, a, W, top, qw_scratch, ws=None, use_workspace=True):
    """With a workspace (the default one unless use_workspace=False) the call is mms_simmatrix_forward_ws_f32: Q.W on
    the bf16 matrix pipe at fp32 accuracy for N >= 2048 (include/mms.h); without, mms_simmatrix_forward_f32."""
    N = q.shape[0]
    K1, K2 = W.shape
    if not use_workspace:
        check(lib().mms_simmatrix_forward_f32(
            N, K1, K2, _ptr(q, "q"), _ptr(a, "a"), _ptr(W, "W"), _ptr(top, "top"),
            _ptr(qw_scratch, "qw_scratch"), _stream()), "mms_simmatrix_forward_f32")
        return
    wsp, wsb = (ws or _default_ws).get(lib().mms_simmatrix_workspace_bytes(N, K1, K2), q.device)
    check(lib().mms_simmatrix_forward_ws_f32(
        N, K1, K2, _ptr(q, "q"), _ptr(a, "a"), _ptr(W, "W"), _ptr(top, "top"),
        _ptr(qw_scratch, "qw_scratch"), wsp, wsb, _stream()), "mms_simmatrix_forward_ws_f32")


def simmatrix_forward_f16(q, a, W, top, ws=None):
    """fp16-storage scoring: q, a half tensors, W and top float32 (include/mms.h: mms_simmatrix_forward_f16)."""
    N = q.shape[0]
    K1, K2 = W.shape
    h = torch.float16
    wsp, wsb = (ws or _default_ws).get(lib().mms_simmatrix_workspace_bytes(N, K1, K2), q.device)
    check(lib().mms_simmatrix_forward_f16(N, K1, K2, _ptr(q, "q", dtype=h), _ptr(a, "a", dtype=h), _ptr(W, "W"),
                                          _ptr(top, "top"), wsp, wsb, _stream()), "mms_simmatrix_forward_f16")


def simmatrix_forward_train_f16(q, a, W, top, qw_scratch, ws=None):
    N = q.shape[0]
    K1, K2 = W.shape
    h = torch.float16
    wsp, wsb = (ws or _default_ws).get(lib().mms_simmatrix_workspace_bytes(N, K1, K2), q.device)
    check(lib().mms_simmatrix_forward_train_f16(N, K1, K2, _ptr(q, "q", dtype=h), _ptr(a, "a", dtype=h), _ptr(W, "W"),
                                                _ptr(top, "top"), _ptr(qw_scratch, "qw_scratch"), wsp, wsb, _stream()),
          "mms_simmatrix_forward_train_f16")


def simmatrix_backward_f16(q, a, W, qw, top_diff, dq, da, dW, ws=None):
    """dq, da: half tensors or None; dW: float32 (accumulated into) or None."""
    N = q.shape[0]
    K1, K2 = W.shape
    h = torch.float16
    wsp, wsb = (ws or _default_ws).get(lib().mms_simmatrix_workspace_bytes(N, K1, K2), q.device)
    check(lib().mms_simmatrix_backward_f16(N, K1, K2, _ptr(q, "q", dtype=h), _ptr(a, "a", dtype=h), _ptr(W, "W"),
                                           _ptr(qw, "qw", True), _ptr(top_diff, "top_diff"), _ptr(dq, "dq", True, h),
                                           _ptr(da, "da", True, h), _ptr(dW, "dW", True), wsp, wsb, _stream()),
          "mms_simmatrix_backward_f16")


def set_matrix_mode(mode):
    """"bf16x3" (default: exact three-way bf16 splits on the bf16 pipe) or "fp32" (fp32 MFMA); include/mms.h."""
    check(lib().mms_set_matrix_mode({"bf16x3": 0, "fp32": 1}[mode]), "mms_set_matrix_mode")


def get_matrix_mode():
    return ("bf16x3", "fp32")[lib().mms_get_matrix_mode()]


def simmatrix_backward(q, a, W, top_diff, dq, da, dW, param_propagate_down=True,
                       propagate_down=(True, True), ws=None, qw=None):
    """qw: the forward's qw_scratch (same q, W; may be `da` itself) -> the cached entry point."""
    N = q.shape[0]
    K1, K2 = W.shape
    wsp, wsb = (ws or _default_ws).get(lib().mms_simmatrix_workspace_bytes(N, K1, K2), q.device)
    if qw is not None:
        check(lib().mms_simmatrix_backward_cached_f32(
            N, K1, K2, _ptr(q, "q"), _ptr(a, "a"), _ptr(W, "W"), _ptr(qw, "qw"), _ptr(top_diff, "top_diff"),
            int(bool(param_propagate_down)), int(bool(propagate_down[0])), int(bool(propagate_down[1])),
            _ptr(dq, "dq", True), _ptr(da, "da", True), _ptr(dW, "dW", True), wsp, wsb, _stream()),
            "mms_simmatrix_backward_cached_f32")
        return
    check(lib().mms_simmatrix_backward_f32(
        N, K1, K2, _ptr(q, "q"), _ptr(a, "a"), _ptr(W, "W"), _ptr(top_diff, "top_diff"),
        int(bool(param_propagate_down)), int(bool(propagate_down[0])), int(bool(propagate_down[1])),
        _ptr(dq, "dq", True), _ptr(da, "da", True), _ptr(dW, "dW", True), wsp, wsb, _stream()),
        "mms_simmatrix_backward_f32")


def pairrank_forward(a, b, y, ordered, similar, loss, margin=1.0, ws=None):
    count = a.numel()
    wsp, wsb = (ws or _default_ws).get(lib().mms_pairrank_workspace_bytes(count), a.device)
    check(lib().mms_pairrank_forward_f32(
        count, float(margin), _ptr(a, "a"), _ptr(b, "b"), _ptr(y, "y"), _ptr(ordered, "ordered"),
        _ptr(similar, "similar"), _ptr(loss, "loss"), wsp, wsb, _stream()),
        "mms_pairrank_forward_f32")


def pairrank_backward(y, ordered, similar, da, db, top_diff=1.0, propagate_down=(True, True)):
    check(lib().mms_pairrank_backward_f32(
        y.numel(), float(top_diff), _ptr(y, "y"), _ptr(ordered, "ordered"),
        _ptr(similar, "similar"), int(bool(propagate_down[0])), int(bool(propagate_down[1])),
        _ptr(da, "da", True), _ptr(db, "db", True), _stream()), "mms_pairrank_backward_f32")


class TripletWorkspace:
    """The fused step's own scratch: its head holds the arrival words of the in-launch loss sum, which must be zero
    between launches (include/mms.h), so it is never shared with other entry points' scratch.  Grow-only; a buffer
    it outgrows is retired, not freed (a captured hipGraph may hold its address); every new buffer is initialised
    with mms_triplet_workspace_init on the current stream.  `reset()` re-initialises after a failed launch."""

    def __init__(self):
        self.buf = None
        self._retired = []

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            if self.buf is not None:
                self._retired.append(self.buf)
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self.reset()
        return self.buf.data_ptr(), self.buf.numel()

    def reset(self):
        if self.buf is not None:
            check(lib().mms_triplet_workspace_init(self.buf.data_ptr(), self.buf.numel(), _stream()),
                  "mms_triplet_workspace_init")


_default_triplet_ws = {}


def triplet_euclid_step(q, a_pos, a_neg, y, s_pos, s_neg, loss, dq, da_pos, da_neg, margin=1.0,
                        loss_weight=1.0, ws=None):
    """`ws`: a TripletWorkspace; calls that may be in flight together (streams, concurrently replayed graphs) must
    not share one.  Default: one per device."""
    N, D = q.shape[0], q.shape[-1]
    if ws is None:
        ws = _default_triplet_ws.setdefault(q.device, TripletWorkspace())
    if not isinstance(ws, TripletWorkspace):
        raise TypeError("triplet_euclid_step needs a capi.TripletWorkspace (its arrival words must stay zero "
                        "between launches; a general Workspace is overwritten by other calls)")
    wsp, wsb = ws.get(lib().mms_triplet_workspace_bytes(N), q.device)
    check(lib().mms_triplet_euclid_step_f32(
        N, D, float(margin), float(loss_weight), _ptr(q, "q"), _ptr(a_pos, "a_pos"),
        _ptr(a_neg, "a_neg"), _ptr(y, "y"), _ptr(s_pos, "s_pos"), _ptr(s_neg, "s_neg"),
        _ptr(loss, "loss", True), _ptr(dq, "dq"), _ptr(da_pos, "da_pos"), _ptr(da_neg, "da_neg"),
        wsp, wsb, _stream()), "mms_triplet_euclid_step_f32")


def triplet_simmatrix_step(q, a_pos, a_neg, y, W, s_pos, s_neg, loss, dq, da_pos, da_neg, dW, margin=1.0,
                           loss_weight=1.0, ws=None):
    """The fused learned-metric step (include/mms.h: mms_triplet_simmatrix_step_f32).  dW is ACCUMULATED into."""
    N, K1 = q.shape
    K2 = a_pos.shape[1]
    wsp, wsb = (ws or _default_ws).get(lib().mms_triplet_simmatrix_workspace_bytes(N, K1, K2), q.device)
    check(lib().mms_triplet_simmatrix_step_f32(
        N, K1, K2, float(margin), float(loss_weight), _ptr(q, "q"), _ptr(a_pos, "a_pos"), _ptr(a_neg, "a_neg"),
        _ptr(y, "y"), _ptr(W, "W"), _ptr(s_pos, "s_pos"), _ptr(s_neg, "s_neg"), _ptr(loss, "loss", True),
        _ptr(dq, "dq"), _ptr(da_pos, "da_pos"), _ptr(da_neg, "da_neg"), _ptr(dW, "dW"), wsp, wsb, _stream()),
        "mms_triplet_simmatrix_step_f32")


def rank_map_mrr(prob, label, group, fixed_axis=1, ws=None):
    """-> (MAP, MRR, effective groups) as Python numbers (device -> host copy of 3 scalars)."""
    n = label.numel()
    wsp, wsb = (ws or _default_ws).get(lib().mms_rank_workspace_bytes(n), prob.device)
    out = torch.empty(2, dtype=torch.float32, device=prob.device)
    eff = torch.empty(1, dtype=torch.int32, device=prob.device)
    check(lib().mms_rank_map_mrr_f32(
        n, int(fixed_axis), _ptr(prob, "prob"), _ptr(label, "label"), _ptr(group, "group"),
        out.data_ptr(), out.data_ptr() + 4, eff.data_ptr(), wsp, wsb, _stream()), "mms_rank_map_mrr_f32")
    o = out.cpu().numpy()
    return o[0], o[1], int(eff.item())


def rank_map_mrr_device(prob, label, group, out, eff, fixed_axis=1, ws=None):
    """The same with the results left on the device: out (2 floats: MAP, MRR), eff (1 int32); no host sync."""
    n = label.numel()
    wsp, wsb = (ws or _default_ws).get(lib().mms_rank_workspace_bytes(n), prob.device)
    check(lib().mms_rank_map_mrr_f32(
        n, int(fixed_axis), _ptr(prob, "prob"), _ptr(label, "label"), _ptr(group, "group"),
        out.data_ptr(), out.data_ptr() + 4, eff.data_ptr(), wsp, wsb, _stream()), "mms_rank_map_mrr_f32")


def rank_auc(prob, label, fixed_axis=1, ignore_label=None, ws=None):
    n = label.numel()
    wsp, wsb = (ws or _default_ws).get(lib().mms_rank_workspace_bytes(n), prob.device)
    out = torch.empty(1, dtype=torch.float32, device=prob.device)
    check(lib().mms_rank_auc_f32(
        n, int(prob.shape[1]), int(fixed_axis), _ptr(prob, "prob"), _ptr(label, "label"),
        int(ignore_label is not None), int(ignore_label or 0), out.data_ptr(), wsp, wsb, _stream()),
        "mms_rank_auc_f32")
    return out.cpu().numpy()[0]


def rank_auc_nd(prob, label, axis=1, fixed_axis=1, ignore_label=None, ws=None):
    """AUC layer for any label axis: prob (..., C, ...) with `axis` the class axis, label of the remaining shape."""
    shape = tuple(prob.shape)
    axis = axis % len(shape)
    outer = int(np.prod(shape[:axis])) if axis else 1
    inner = int(np.prod(shape[axis + 1:])) if axis + 1 < len(shape) else 1
    n = outer * inner
    wsp, wsb = (ws or _default_ws).get(lib().mms_rank_workspace_bytes(n), prob.device)
    out = torch.empty(1, dtype=torch.float32, device=prob.device)
    check(lib().mms_rank_auc_nd_f32(
        outer, int(shape[axis]), inner, int(fixed_axis), _ptr(prob, "prob"), _ptr(label, "label"),
        int(ignore_label is not None), int(ignore_label or 0), out.data_ptr(), wsp, wsb, _stream()),
        "mms_rank_auc_nd_f32")
    return out.cpu().numpy()[0]


def rank_accuracy(a, b, label, ws=None):
    n = a.numel()
    wsp, wsb = (ws or _default_ws).get(max(4096, lib().mms_rank_workspace_bytes(1)), a.device)
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    check(lib().mms_rank_accuracy_f32(n, _ptr(a, "a"), _ptr(b, "b"), _ptr(label, "label"),
                                      out.data_ptr(), wsp, wsb, _stream()), "mms_rank_accuracy_f32")
    return out.cpu().numpy()[0]


def simcross_euclid_forward_f16(q, a, top):
    """fp16-storage scoring (cfg 5): q, a half tensors (N,1,D); top float32."""
    N, D = q.shape[0], q.shape[-1]
    h = torch.float16
    check(lib().mms_simcross_euclid_forward_f16(N, D, _ptr(q, "q", dtype=h), _ptr(a, "a", dtype=h),
                                                _ptr(top, "top"), _stream()),
          "mms_simcross_euclid_forward_f16")


def simcross_euclid_forward_backward_f16(q, a, top_diff, top, dq, da):
    N, D = q.shape[0], q.shape[-1]
    h = torch.float16
    check(lib().mms_simcross_euclid_forward_backward_f16(
        N, D, _ptr(q, "q", dtype=h), _ptr(a, "a", dtype=h), _ptr(top_diff, "top_diff"),
        _ptr(top, "top"), _ptr(dq, "dq", dtype=h), _ptr(da, "da", dtype=h), _stream()),
        "mms_simcross_euclid_forward_backward_f16")


def simcross_cosine_forward_backward_f16(q, a, top_diff, top, dq, da, norm0=None, norm1=None):
    N, D = q.shape[0], q.shape[-1]
    h = torch.float16
    check(lib().mms_simcross_cosine_forward_backward_f16(
        N, D, _ptr(q, "q", dtype=h), _ptr(a, "a", dtype=h), _ptr(top_diff, "top_diff"), _ptr(top, "top"),
        _ptr(norm0, "norm0", True), _ptr(norm1, "norm1", True), _ptr(dq, "dq", dtype=h), _ptr(da, "da", dtype=h),
        _stream()), "mms_simcross_cosine_forward_backward_f16")


def simcross_cosine_forward_f16(q, a, top, norm0=None, norm1=None):
    N, D = q.shape[0], q.shape[-1]
    h = torch.float16
    check(lib().mms_simcross_cosine_forward_f16(N, D, _ptr(q, "q", dtype=h), _ptr(a, "a", dtype=h), _ptr(top, "top"),
                                                _ptr(norm0, "norm0", True), _ptr(norm1, "norm1", True), _stream()),
          "mms_simcross_cosine_forward_f16")


def embed_forward(index, weight, top, bias=None):
    M, (K, N) = index.numel(), weight.shape
    check(lib().mms_embed_forward_f32(M, N, K, _ptr(index, "index"), _ptr(weight, "weight"),
                                      _ptr(bias, "bias", True), _ptr(top, "top"), _stream()),
          "mms_embed_forward_f32")


def embed_backward(index, top_diff, weight_diff, bias_diff=None, ws=None):
    M, (K, N) = index.numel(), weight_diff.shape
    wsp, wsb = (ws or _default_ws).get(lib().mms_embed_workspace_bytes(M, N), index.device)
    check(lib().mms_embed_backward_f32(M, N, K, _ptr(index, "index"), _ptr(top_diff, "top_diff"),
                                       _ptr(weight_diff, "weight_diff", True),
                                       _ptr(bias_diff, "bias_diff", True), wsp, wsb, _stream()),
          "mms_embed_backward_f32")


def embed_backward_pair(index0, index1, top_diff0, top_diff1, weight_diff, bias_diff=None, ws=None):
    """Backward of two Embed layers over one table in one pass: layer 0's rows, then layer 1's (include/mms.h)."""
    M0, M1, (K, N) = index0.numel(), index1.numel(), weight_diff.shape
    wsp, wsb = (ws or _default_ws).get(lib().mms_embed_workspace_bytes(M0 + M1, N), index0.device)
    check(lib().mms_embed_backward_pair_f32(M0, M1, N, K, _ptr(index0, "index0"), _ptr(top_diff0, "top_diff0"),
                                            _ptr(index1, "index1"), _ptr(top_diff1, "top_diff1"),
                                            _ptr(weight_diff, "weight_diff", True), _ptr(bias_diff, "bias_diff", True),
                                            wsp, wsb, _stream()), "mms_embed_backward_pair_f32")


class EmbedPairIndex:
    """The inverted index of two Embed layers' word ids, built by embed_forward_pair and consumed by
    embed_backward_pair(index=...): its own buffer (nothing else may write it between the two calls)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self.buf.data_ptr(), self.buf.numel()


def embed_forward_pair(index0, index1, weight, top0, top1, bias=None, index=None):
    """top0 = Embed(index0), top1 = Embed(index1) in one launch; `index` (an EmbedPairIndex): also build the inverted
    index of the ids for embed_backward_pair.  Returns True if the index was built."""
    M0, M1, (K, N) = index0.numel(), index1.numel(), weight.shape
    wsp, wsb, built = None, 0, False
    if index is not None and lib().mms_embed_pair_index_supported(M0, M1, K):
        wsp, wsb = index.get(lib().mms_embed_workspace_bytes(M0 + M1, N), index0.device)
        built = True
    check(lib().mms_embed_forward_pair_f32(M0, M1, N, K, _ptr(index0, "index0"), _ptr(index1, "index1"),
                                           _ptr(weight, "weight"), _ptr(bias, "bias", True), _ptr(top0, "top0"),
                                           _ptr(top1, "top1"), wsp, wsb, _stream()), "mms_embed_forward_pair_f32")
    return built


def embed_backward_pair_indexed(index0, index1, top_diff0, top_diff1, weight_diff, index, bias_diff=None):
    """embed_backward_pair with the index embed_forward_pair(index0, index1, ..., index=index) built (same order)."""
    M0, M1, (K, N) = index0.numel(), index1.numel(), weight_diff.shape
    if index.buf is None:
        raise MMSError("embed_backward_pair_indexed: no index was built (embed_forward_pair returned False); use "
                       "embed_backward_pair")
    check(lib().mms_embed_backward_pair_indexed_f32(M0, M1, N, K, _ptr(index0, "index0"), _ptr(top_diff0, "top_diff0"),
                                                    _ptr(index1, "index1"), _ptr(top_diff1, "top_diff1"),
                                                    _ptr(weight_diff, "weight_diff", True), _ptr(bias_diff, "bias_diff", True),
                                                    index.buf.data_ptr(), index.buf.numel(), _stream()),
          "mms_embed_backward_pair_indexed_f32")


def feed_gather_rows(src, first, rows, dst, perm=None):
    """dst[i] = src[perm[first+i]] (perm int32 on the device, or None = identity)."""
    src_rows = src.shape[0]
    row_elems = src.numel() // max(src_rows, 1)
    check(lib().mms_feed_gather_rows_f32(rows, row_elems, src_rows, _ptr(src, "src"),
                                         _ptr(perm, "perm", True, dtype=torch.int32), first,
                                         _ptr(dst, "dst"), _stream()),
          "mms_feed_gather_rows_f32")


class SimCrossArgs(C.Structure):
    """include/mms.h: mms_simcross_args_f32 -- the arguments of the SimCross forward / backward in one block."""
    _fields_ = [("dist_mode", C.c_int), ("N", C.c_int), ("W1", C.c_int), ("W2", C.c_int), ("D", C.c_int), ("M", C.c_int),
                ("q", C.c_void_p), ("a", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p),
                ("top", C.c_void_p), ("norm0", C.c_void_p), ("norm1", C.c_void_p),
                ("bias_term", C.c_int), ("propagate_down0", C.c_int), ("propagate_down1", C.c_int),
                ("top_diff", C.c_void_p), ("dq", C.c_void_p), ("da", C.c_void_p), ("dW", C.c_void_p), ("dbias", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


def null_launch(workgroups=256):
    """An empty kernel of `workgroups` x 512 threads on the current stream (measurement aid: the launch floor)."""
    check(lib().mms_null_launch(int(workgroups), _stream()), "mms_null_launch")


EUCLID_BWD_FP32, EUCLID_BWD_REFERENCE = 0, 1


def set_euclid_backward_mode(mode):
    """'fp32' (default: <= 2 ulp from the reference) or 'reference' (the reference's bits)."""
    m = {"fp32": 0, "reference": 1}.get(mode, mode)
    check(lib().mms_set_euclid_backward_mode(int(m)), "mms_set_euclid_backward_mode")


def set_pairrank_hinge_mode(mode):
    """'cpu' (default): `ordered > 0` as PairRankLossLayer::Backward_cpu; 'gpu': `ordered >= 0` as the
    reference's .cu kernel.  Per calling thread."""
    m = {"cpu": 0, "gpu": 1}[mode] if isinstance(mode, str) else int(mode)
    check(lib().mms_set_pairrank_hinge_mode(m), "mms_set_pairrank_hinge_mode")


def set_f16_distance_mode(mode):
    """'ordered' (default): the reference's d-ascending fp32 sum, bit-identical scores; 'tree': fixed tree sum,
    ~1e-6 relative, no ordered chain.  Per calling thread; fp16-storage entry points only."""
    m = {"ordered": 0, "tree": 1}[mode] if isinstance(mode, str) else int(mode)
    check(lib().mms_set_f16_distance_mode(m), "mms_set_f16_distance_mode")


def set_rank_tie_mode(mode):
    """'input' (default): equal scores keep the input order; 'libstdcxx': the order libstdc++'s std::sort leaves them in
    (include/mms.h: MMS_RANK_TIES_*)."""
    m = {"input": 0, "libstdcxx": 1}[mode] if isinstance(mode, str) else int(mode)
    check(lib().mms_set_rank_tie_mode(m), "mms_set_rank_tie_mode")


def set_loss_sum_mode(mode):
    """'fast' (default): order-free sum of the loss terms; 'reference': Forward_cpu's running fp32 sum, bit for bit
    (include/mms.h: MMS_LOSS_SUM_*)."""
    m = {"fast": 0, "reference": 1}[mode] if isinstance(mode, str) else int(mode)
    check(lib().mms_set_loss_sum_mode(m), "mms_set_loss_sum_mode")


def set_triplet_finish_mode(mode):
    """'inlaunch' (default): the loss is summed inside the step's one launch; 'launch': a second launch sums the terms
    (include/mms.h: MMS_TRIPLET_FINISH_*)."""
    m = {"launch": 0, "inlaunch": 1}[mode] if isinstance(mode, str) else int(mode)
    check(lib().mms_set_triplet_finish_mode(m), "mms_set_triplet_finish_mode")


def get_pairrank_hinge_mode():
    return "gpu" if lib().mms_get_pairrank_hinge_mode() == 1 else "cpu"


def get_euclid_backward_mode():
    return "reference" if lib().mms_get_euclid_backward_mode() == 1 else "fp32"


# ---- double instantiation (functional kernels; tests/test_gpu_f64.py) -------------------------
_D = torch.float64


def simcross_forward_f64(mode, q, a, top, W=None, bias=None, norm0=None, norm1=None, ws=None):
    N, W1, D = q.shape
    W2 = a.shape[1]
    M = W.shape[0] if mode == 2 else 1
    wsp, wsb = (ws or _default_ws).get(lib().mms_simcross_workspace_bytes_f64(mode, N, W1, W2, D, M), q.device)
    check(lib().mms_simcross_forward_f64(
        mode, N, W1, W2, D, M, _ptr(q, "q", dtype=_D), _ptr(a, "a", dtype=_D), _ptr(W, "W", True, _D),
        _ptr(bias, "bias", True, _D), _ptr(top, "top", dtype=_D), _ptr(norm0, "norm0", True, _D),
        _ptr(norm1, "norm1", True, _D), wsp, wsb, _stream()), "mms_simcross_forward_f64")


def simcross_backward_f64(mode, q, a, top, top_diff, dq, da, W=None, bias_term=False, norm0=None,
                          norm1=None, dW=None, dbias=None, propagate_down=(True, True), ws=None):
    N, W1, D = q.shape
    W2 = a.shape[1]
    M = W.shape[0] if mode == 2 else 1
    wsp, wsb = (ws or _default_ws).get(lib().mms_simcross_workspace_bytes_f64(mode, N, W1, W2, D, M), q.device)
    check(lib().mms_simcross_backward_f64(
        mode, N, W1, W2, D, M, _ptr(q, "q", dtype=_D), _ptr(a, "a", dtype=_D), _ptr(W, "W", True, _D),
        int(bool(bias_term)), _ptr(top, "top", dtype=_D), _ptr(top_diff, "top_diff", dtype=_D),
        _ptr(norm0, "norm0", True, _D), _ptr(norm1, "norm1", True, _D), int(bool(propagate_down[0])),
        int(bool(propagate_down[1])), _ptr(dq, "dq", dtype=_D), _ptr(da, "da", dtype=_D),
        _ptr(dW, "dW", True, _D), _ptr(dbias, "dbias", True, _D), wsp, wsb, _stream()),
        "mms_simcross_backward_f64")


def simmatrix_forward_f64(q, a, W, top, qw_scratch):
    N = q.shape[0]
    K1, K2 = W.shape
    check(lib().mms_simmatrix_forward_f64(
        N, K1, K2, _ptr(q, "q", dtype=_D), _ptr(a, "a", dtype=_D), _ptr(W, "W", dtype=_D),
        _ptr(top, "top", dtype=_D), _ptr(qw_scratch, "qw_scratch", dtype=_D), _stream()),
        "mms_simmatrix_forward_f64")


def simmatrix_backward_f64(q, a, W, top_diff, dq, da, dW, param_propagate_down=True,
                           propagate_down=(True, True)):
    N = q.shape[0]
    K1, K2 = W.shape
    check(lib().mms_simmatrix_backward_f64(
        N, K1, K2, _ptr(q, "q", dtype=_D), _ptr(a, "a", dtype=_D), _ptr(W, "W", dtype=_D),
        _ptr(top_diff, "top_diff", dtype=_D), int(bool(param_propagate_down)),
        int(bool(propagate_down[0])), int(bool(propagate_down[1])), _ptr(dq, "dq", True, _D),
        _ptr(da, "da", True, _D), _ptr(dW, "dW", True, _D), _stream()), "mms_simmatrix_backward_f64")


def pairrank_forward_f64(a, b, y, ordered, similar, loss, margin=1.0):
    check(lib().mms_pairrank_forward_f64(
        a.numel(), float(margin), _ptr(a, "a", dtype=_D), _ptr(b, "b", dtype=_D), _ptr(y, "y", dtype=_D),
        _ptr(ordered, "ordered", dtype=_D), _ptr(similar, "similar", dtype=_D),
        _ptr(loss, "loss", dtype=_D), _stream()), "mms_pairrank_forward_f64")


def pairrank_backward_f64(y, ordered, similar, da, db, top_diff=1.0, propagate_down=(True, True)):
    check(lib().mms_pairrank_backward_f64(
        y.numel(), float(top_diff), _ptr(y, "y", dtype=_D), _ptr(ordered, "ordered", dtype=_D),
        _ptr(similar, "similar", dtype=_D), int(bool(propagate_down[0])), int(bool(propagate_down[1])),
        _ptr(da, "da", True, _D), _ptr(db, "db", True, _D), _stream()), "mms_pairrank_backward_f64")
