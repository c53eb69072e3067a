"""Python face of the C++ Caffe-API mirror (libmms_caffe.so, include/mms_layer.h).

Plays the role pycaffe plays for the reference (python/caffe/_caffe.cpp:212-340,
python/caffe/net_spec.py:31-79): build a layer from its prototxt message or
from NetSpec-style keyword arguments, wire Blobs, SetUp / Forward / Backward.

    q, a, top = Blob((50, 40, 50)), Blob((50, 40, 50)), Blob()
    sim = SimCross(dist_mode=2, mesure_count=4)            # like L.SimCross(q, a, ...)
    sim.SetUp([q, a], [top]); sim.Forward([q, a], [top])

All compute happens in the HIP library; this module only moves pointers.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmms_caffe.so")
_lib = None

_vp, _i, _ip, _fp = C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)
_bpp = C.POINTER(C.c_void_p)

_SIGNATURES = {
    "mms_blob_create": (_vp, [_ip, _i]),
    "mms_blob_destroy": (None, [_vp]),
    "mms_blob_reshape": (None, [_vp, _ip, _i]),
    "mms_blob_num_axes": (_i, [_vp]),
    "mms_blob_shape": (_i, [_vp, _i]),
    "mms_blob_count": (_i, [_vp]),
    "mms_blob_cpu": (_fp, [_vp, _i]),
    "mms_blob_mutable_cpu": (_fp, [_vp, _i]),
    "mms_blob_gpu": (_vp, [_vp, _i]),
    "mms_blob_mutable_gpu": (_vp, [_vp, _i]),
    "mms_layer_create": (_vp, [C.c_char_p, C.c_char_p, _i]),
    "mms_layer_destroy": (None, [_vp]),
    "mms_layer_type": (C.c_char_p, [_vp]),
    "mms_layer_setup": (None, [_vp, _bpp, _i, _bpp, _i]),
    "mms_layer_forward": (C.c_float, [_vp, _bpp, _i, _bpp, _i]),
    "mms_layer_backward": (None, [_vp, _bpp, _i, _ip, _bpp, _i]),
    "mms_layer_num_param_blobs": (_i, [_vp]),
    "mms_layer_param_blob": (_vp, [_vp, _i]),
    "mms_layer_set_param_propagate_down": (None, [_vp, _i, _i]),
    "mms_layer_set_option": (_i, [_vp, C.c_char_p, _i]),
    "mms_net_create": (_vp, [C.c_char_p, _i, C.c_char_p, _i]),
    "mms_net_destroy": (None, [_vp]),
    "mms_net_set_option": (_i, [_vp, C.c_char_p, _i]),
    "mms_net_num_fused": (_i, [_vp]),
    "mms_net_num_splits": (_i, [_vp]),
    "mms_net_name": (C.c_char_p, [_vp]),
    "mms_net_num_layers": (_i, [_vp]),
    "mms_net_layer_name": (C.c_char_p, [_vp, _i]),
    "mms_net_layer_type": (C.c_char_p, [_vp, _i]),
    "mms_net_layer_supported": (_i, [_vp, _i]),
    "mms_net_layer_runnable": (_i, [_vp, _i]),
    "mms_net_layer_why_not": (C.c_char_p, [_vp, _i]),
    "mms_net_layer": (_vp, [_vp, _i]),
    "mms_net_num_blobs": (_i, [_vp]),
    "mms_net_blob_name": (C.c_char_p, [_vp, _i]),
    "mms_net_blob": (_vp, [_vp, C.c_char_p]),
    "mms_net_setup": (_i, [_vp]),
    "mms_net_forward": (C.c_float, [_vp]),
    "mms_net_backward": (None, [_vp]),
    "mms_caffe_set_mode": (None, [_i]),
    "mms_caffe_set_random_seed": (None, [C.c_uint]),
    "mms_layer_registry_types": (C.c_char_p, []),
    "mms_layer_run_f64": (_i, [C.c_char_p, _i, _ip, _ip, _vp, _i, _vp, _vp, _ip, _vp, C.c_longlong, _ip, _ip, _vp, _vp,
                               C.c_char_p, _i]),
    "mms_h5_open": (_vp, [C.c_char_p, C.c_char_p, _i]),
    "mms_h5_close": (None, [_vp]),
    "mms_h5_num_datasets": (_i, [_vp]),
    "mms_h5_dataset_name": (C.c_char_p, [_vp, _i]),
    "mms_h5_dataset_info": (_i, [_vp, C.c_char_p, C.POINTER(C.c_longlong), _i, _ip, _ip, C.c_char_p, _i]),
    "mms_h5_read_float": (_i, [_vp, C.c_char_p, _fp, C.c_longlong, C.c_char_p, _i]),
    "mms_h5_writer_create": (_vp, []),
    "mms_h5_writer_destroy": (None, [_vp]),
    "mms_h5_writer_add": (None, [_vp, C.c_char_p, C.POINTER(C.c_longlong), _i, _i, C.POINTER(C.c_double)]),
    "mms_h5_writer_save": (_i, [_vp, C.c_char_p, C.c_char_p, _i]),
    "mms_snapshot_open": (_vp, [C.c_char_p, C.c_char_p, _i]),
    "mms_snapshot_close": (None, [_vp]),
    "mms_snapshot_net_name": (C.c_char_p, [_vp]),
    "mms_snapshot_num_layers": (_i, [_vp]),
    "mms_snapshot_layer_name": (C.c_char_p, [_vp, _i]),
    "mms_snapshot_layer_type": (C.c_char_p, [_vp, _i]),
    "mms_snapshot_num_blobs": (_i, [_vp, _i]),
    "mms_snapshot_blob_shape": (_i, [_vp, _i, _i, _ip, _i]),
    "mms_snapshot_blob_count": (_i, [_vp, _i, _i]),
    "mms_snapshot_blob_data": (_fp, [_vp, _i, _i]),
    "mms_layer_copy_from_snapshot": (_i, [_vp, _vp, C.c_char_p]),
    "mms_snapshot_writer_save_hdf5": (_i, [_vp, C.c_char_p, C.c_char_p, _i]),
    "mms_snapshot_writer_create": (_vp, [C.c_char_p]),
    "mms_snapshot_writer_destroy": (None, [_vp]),
    "mms_snapshot_writer_add_layer": (None, [_vp, C.c_char_p, C.c_char_p]),
    "mms_snapshot_writer_add_blob": (None, [_vp, _ip, _i, _fp]),
    "mms_snapshot_writer_add_from_layer": (None, [_vp, _vp, C.c_char_p]),
    "mms_snapshot_writer_save": (_i, [_vp, C.c_char_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libmms_caffe.so is not built (%s); run __graft_entry__.build()" % LIB_PATH)
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def registered_layer_types():
    return lib().mms_layer_registry_types().decode().split(",")


def set_mode_gpu():
    lib().mms_caffe_set_mode(1)


def set_mode_cpu():
    lib().mms_caffe_set_mode(0)


def set_random_seed(seed):
    lib().mms_caffe_set_random_seed(int(seed))


def _ints(seq):
    seq = [int(x) for x in seq]
    return (C.c_int * max(1, len(seq)))(*seq), len(seq)


class Blob:
    """caffe::Blob<float>.  `.data` / `.diff` are numpy views of the HOST side
    (they move the SyncedMemory head to the CPU, like net.blobs[...].data)."""

    def __init__(self, shape=(), _handle=None):
        self._owned = _handle is None
        if _handle is None:
            arr, n = _ints(shape)
            _handle = lib().mms_blob_create(arr, n)
        self._h = _handle

    def __del__(self):
        if getattr(self, "_owned", False) and self._h and _lib is not None:
            _lib.mms_blob_destroy(self._h)
            self._h = None

    @property
    def shape(self):
        return tuple(lib().mms_blob_shape(self._h, i) for i in range(lib().mms_blob_num_axes(self._h)))

    @property
    def count(self):
        return lib().mms_blob_count(self._h)

    def reshape(self, *shape):
        arr, n = _ints(shape)
        lib().mms_blob_reshape(self._h, arr, n)

    def _view(self, which, mutable):
        f = lib().mms_blob_mutable_cpu if mutable else lib().mms_blob_cpu
        p = f(self._h, which)
        n = self.count
        if n == 0:
            return np.zeros(self.shape, np.float32)
        return np.ctypeslib.as_array(p, shape=(n,)).reshape(self.shape)

    @property
    def data(self):
        return self._view(0, True)

    @property
    def diff(self):
        return self._view(1, True)

    def gpu_data_ptr(self):
        return lib().mms_blob_gpu(self._h, 0)

    def gpu_diff_ptr(self):
        return lib().mms_blob_gpu(self._h, 1)


def _handles(blobs):
    arr = (C.c_void_p * max(1, len(blobs)))(*[b._h for b in blobs])
    return arr, len(blobs)


def _text_value(v):
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, str):
        return '"%s"' % v
    return repr(v)


def _text_message(d, indent=2):
    """dict -> protobuf text format (the job of python/caffe/net_spec.py:56-79)."""
    out = []
    pad = " " * indent
    for k, v in d.items():
        for item in (v if isinstance(v, (list, tuple)) else [v]):
            if isinstance(item, dict):
                out.append("%s%s {\n%s%s}\n" % (pad, k, _text_message(item, indent + 2), pad))
            else:
                out.append("%s%s: %s\n" % (pad, k, _text_value(item)))
    return "".join(out)


class Layer:
    """caffe::Layer<float> created through LayerRegistry::CreateLayer."""

    def __init__(self, prototxt):
        err = C.create_string_buffer(512)
        self.prototxt = prototxt
        self._h = lib().mms_layer_create(prototxt.encode(), err, 512)
        if not self._h:
            raise ValueError("prototxt: " + err.value.decode())

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None and not getattr(self, "_borrowed", False):
            _lib.mms_layer_destroy(self._h)
        self._h = None

    @property
    def type(self):
        return lib().mms_layer_type(self._h).decode()

    def SetUp(self, bottom, top):
        b, nb = _handles(bottom)
        t, nt = _handles(top)
        lib().mms_layer_setup(self._h, b, nb, t, nt)

    def Forward(self, bottom, top):
        b, nb = _handles(bottom)
        t, nt = _handles(top)
        return lib().mms_layer_forward(self._h, b, nb, t, nt)

    def Backward(self, top, propagate_down, bottom):
        b, nb = _handles(bottom)
        t, nt = _handles(top)
        pd, _ = _ints([1 if x else 0 for x in propagate_down])
        lib().mms_layer_backward(self._h, t, nt, pd, b, nb)

    @property
    def blobs(self):
        n = lib().mms_layer_num_param_blobs(self._h)
        return [Blob(_handle=lib().mms_layer_param_blob(self._h, i)) for i in range(n)]

    def set_param_propagate_down(self, i, v):
        lib().mms_layer_set_param_propagate_down(self._h, int(i), 1 if v else 0)

    def set_option(self, key, value):
        """Per-layer switch of this implementation (include/mms_layer.h: mms_layer_set_option)."""
        if lib().mms_layer_set_option(self._h, key.encode(), int(value)) != 0:
            raise KeyError("%s layer has no option %r" % (self.type, key))


class Net:
    """A generated net file (NetParameter text), unmodified: the layers this library implements are
    instantiated and wired by blob name, the others are listed and skipped (include/mms_layer.h: mms_net_*).
    phase: "TRAIN" or "TEST"."""

    def __init__(self, prototxt, phase="TEST"):
        err = C.create_string_buffer(512)
        self._h = lib().mms_net_create(prototxt.encode(), 1 if phase in ("TEST", 1) else 0, err, 512)
        if not self._h:
            raise ValueError("net prototxt: " + err.value.decode())

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mms_net_destroy(self._h)
        self._h = None

    @property
    def name(self):
        return lib().mms_net_name(self._h).decode()

    @property
    def layers(self):
        """[(name, type, supported, runnable, why_not)] in file order, after phase filtering."""
        l, h = lib(), self._h
        return [(l.mms_net_layer_name(h, i).decode(), l.mms_net_layer_type(h, i).decode(),
                 bool(l.mms_net_layer_supported(h, i)), bool(l.mms_net_layer_runnable(h, i)),
                 l.mms_net_layer_why_not(h, i).decode()) for i in range(l.mms_net_num_layers(h))]

    @property
    def blob_names(self):
        return [lib().mms_net_blob_name(self._h, i).decode() for i in range(lib().mms_net_num_blobs(self._h))]

    def blob(self, name):
        b = lib().mms_net_blob(self._h, name.encode())
        if not b:
            raise KeyError(name)
        bl = Blob(_handle=b)
        bl._keep = self                 # the net owns the blob
        return bl

    def layer(self, name):
        for i, (n, _, sup, _, _) in enumerate(self.layers):
            if n == name:
                if not sup:
                    raise KeyError("%s is not a layer type of this library" % name)
                lay = Layer.__new__(Layer)
                lay._h = lib().mms_net_layer(self._h, i)
                lay._borrowed = True
                lay._keep = self
                return lay
        raise KeyError(name)

    def SetUp(self):
        return lib().mms_net_setup(self._h)

    def set_option(self, key, value):
        """'fuse_embed_scoring': score straight from word ids where a SimCross layer is fed by two Embed layers
        sharing one table (forward-only use; include/mms_layer.h: mms_net_set_option)."""
        if lib().mms_net_set_option(self._h, key.encode(), int(value)):
            raise KeyError(key)

    @property
    def num_fused(self):
        return lib().mms_net_num_fused(self._h)

    @property
    def num_splits(self):
        """Blobs that feed several back-propagating layers and therefore got Split semantics (insert_splits.cpp)."""
        return lib().mms_net_num_splits(self._h)

    def Forward(self):
        return float(lib().mms_net_forward(self._h))

    def Backward(self):
        lib().mms_net_backward(self._h)


def _make(type_name, param_field, name=None, loss_weight=None, top=None, **kwargs):
    d = {"name": name or type_name.lower(), "type": type_name}
    if top is not None:
        d["top"] = list(top)
    if loss_weight is not None:
        d["loss_weight"] = loss_weight
    if kwargs:
        d[param_field] = kwargs
    return Layer("layer {\n%s}\n" % _text_message(d))


def SimCross(**kw):
    """L.SimCross(q, a, dist_mode=2, mesure_count=4, ...) -> sim_cross_param {...}
    (do_trec_qa_clean.py:468; field names and defaults caffe.proto:471-477)."""
    return _make("SimCross", "sim_cross_param", **kw)


def SimMatrix(**kw):
    return _make("SimMatrix", "sim_matrix_param", **kw)


def PairRankLoss(**kw):
    return _make("PairRankLoss", "pair_rank_loss_param", **kw)


def Embed(**kw):
    """L.Embed(question, input_dim=V, num_output=Dw, weight_filler=..., weight_source=...)
    (do_trec_qa_clean.py:461-466)."""
    return _make("Embed", "embed_param", **kw)


def HDF5Data(top, **kw):
    """L.HDF5Data(batch_size=B, source=list_file, shuffle=0, ntop=5) with the tops named after the
    datasets, as NetSpec names them from the assignment targets (do_trec_qa_clean.py:380)."""
    kw.pop("ntop", None)
    if "shuffle" in kw:
        kw["shuffle"] = bool(kw["shuffle"])
    return _make("HDF5Data", "hdf5_data_param", top=top, **kw)


def MAP(**kw):
    """L.MAP(prob, label, group)  (do_trec_qa_clean.py:495)."""
    return _make("MAP", "map_param", **kw)


def MRR(**kw):
    return _make("MRR", "mrr_param", **kw)


def AUC(**kw):
    return _make("AUC", "auc_param", **kw)


def RankAccuracy(**kw):
    return _make("RankAccuracy", "rank_accuracy_param", **kw)


class Snapshot:
    """A snapshot opened for reading -- .caffemodel (binary NetParameter, current or V1/V0 layer lists) or an
    HDF5-format snapshot: {layer name: [numpy blobs]}."""

    def __init__(self, path):
        err = C.create_string_buffer(512)
        self._h = lib().mms_snapshot_open(str(path).encode(), err, 512)
        if not self._h:
            raise IOError(err.value.decode())

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mms_snapshot_close(self._h)
            self._h = None

    @property
    def net_name(self):
        return lib().mms_snapshot_net_name(self._h).decode()

    def layers(self):
        out = []
        for i in range(lib().mms_snapshot_num_layers(self._h)):
            blobs = []
            for j in range(lib().mms_snapshot_num_blobs(self._h, i)):
                shp = (C.c_int * 32)()
                n = lib().mms_snapshot_blob_shape(self._h, i, j, shp, 32)
                cnt = lib().mms_snapshot_blob_count(self._h, i, j)
                data = np.ctypeslib.as_array(lib().mms_snapshot_blob_data(self._h, i, j), shape=(cnt,)).copy() \
                    if cnt else np.zeros(0, np.float32)
                blobs.append(data.reshape([shp[a] for a in range(n)]) if n else data)
            out.append((lib().mms_snapshot_layer_name(self._h, i).decode(),
                        lib().mms_snapshot_layer_type(self._h, i).decode(), blobs))
        return out

    def copy_into(self, layer, layer_name):
        """Net::CopyTrainedLayersFrom for one layer; raises on count / shape mismatch."""
        rc = lib().mms_layer_copy_from_snapshot(layer._h, self._h, layer_name.encode())
        if rc == 1:
            return False
        if rc:
            raise ValueError("snapshot layer %r does not fit the target layer (%s)"
                             % (layer_name, {2: "blob count", 3: "blob shape"}[rc]))
        return True


def save_snapshot(path, net_name, named_layers=(), raw_layers=(), hdf5=False):
    """Write a .caffemodel (or, with hdf5=True, an HDF5-format snapshot): `named_layers` = [(name, Layer)],
    `raw_layers` = [(name, type, [numpy blobs])]."""
    w = lib().mms_snapshot_writer_create(net_name.encode())
    try:
        for name, layer in named_layers:
            lib().mms_snapshot_writer_add_from_layer(w, layer._h, name.encode())
        for name, typ, blobs in raw_layers:
            lib().mms_snapshot_writer_add_layer(w, name.encode(), typ.encode())
            for b in blobs:
                b = np.ascontiguousarray(b, np.float32)
                shp, n = _ints(b.shape)
                lib().mms_snapshot_writer_add_blob(w, shp, n, b.ctypes.data_as(_fp))
        if hdf5:
            err = C.create_string_buffer(512)
            if lib().mms_snapshot_writer_save_hdf5(w, str(path).encode(), err, 512):
                raise IOError(err.value.decode())
        elif lib().mms_snapshot_writer_save(w, str(path).encode()):
            raise IOError("cannot write %s" % path)
    finally:
        lib().mms_snapshot_writer_destroy(w)


class H5File:
    """An HDF5 file as the HDF5Data layer sees it: root-group datasets, read whole as float32."""

    def __init__(self, path):
        err = C.create_string_buffer(512)
        self._h = lib().mms_h5_open(str(path).encode(), err, 512)
        if not self._h:
            raise IOError(err.value.decode())

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.mms_h5_close(self._h)
            self._h = None

    def keys(self):
        return [lib().mms_h5_dataset_name(self._h, i).decode() for i in range(lib().mms_h5_num_datasets(self._h))]

    def info(self, name):
        """-> (shape, type_class 0 int / 1 float, element bytes)"""
        dims = (C.c_longlong * 32)()
        cls, es = C.c_int(), C.c_int()
        err = C.create_string_buffer(512)
        n = lib().mms_h5_dataset_info(self._h, name.encode(), dims, 32, C.byref(cls), C.byref(es), err, 512)
        if n < 0:
            raise KeyError(err.value.decode())
        return tuple(dims[a] for a in range(n)), cls.value, es.value

    def __getitem__(self, name):
        shape, _, _ = self.info(name)
        out = np.empty(shape, np.float32)
        err = C.create_string_buffer(512)
        if lib().mms_h5_read_float(self._h, name.encode(), out.ctypes.data_as(_fp), out.size, err, 512):
            raise IOError(err.value.decode())
        return out


def write_h5(path, datasets):
    """{name: float32 / float64 ndarray} -> contiguous root-group datasets (what h5py's `f[name] = arr` writes)."""
    w = lib().mms_h5_writer_create()
    try:
        for name, arr in datasets.items():
            arr = np.asarray(arr)
            es = 4 if arr.dtype == np.float32 else 8
            vals = np.ascontiguousarray(arr, np.float64)
            dims = (C.c_longlong * max(1, arr.ndim))(*arr.shape)
            lib().mms_h5_writer_add(w, name.encode(), dims, arr.ndim, es, vals.ctypes.data_as(C.POINTER(C.c_double)))
        err = C.create_string_buffer(512)
        if lib().mms_h5_writer_save(w, str(path).encode(), err, 512):
            raise IOError(err.value.decode())
    finally:
        lib().mms_h5_writer_destroy(w)


def run_layer_f64(prototxt, bottoms, top_diff=None, params=None, param_diffs=None, propagate_down=None,
                  top_capacity=1 << 22):
    """Layer<double> end to end (include/mms_layer.h: mms_layer_run_f64): returns
    (top, [bottom diffs], [parameter diffs]) as float64 arrays."""
    f = lib().mms_layer_run_f64
    dp = C.POINTER(C.c_double)
    bots = [np.ascontiguousarray(b, np.float64) for b in bottoms]
    axes = (C.c_int * len(bots))(*[b.ndim for b in bots])
    dims = [d for b in bots for d in b.shape]
    dims_c = (C.c_int * max(1, len(dims)))(*dims)
    bptr = (dp * len(bots))(*[b.ctypes.data_as(dp) for b in bots])
    params = [np.ascontiguousarray(p, np.float64) for p in (params or [])]
    pdiffs = [np.array(d, np.float64, copy=True) for d in (param_diffs or [np.zeros_like(p) for p in params])]
    pptr = (dp * max(1, len(params)))(*[p.ctypes.data_as(dp) for p in params])
    pdptr = (dp * max(1, len(params)))(*[d.ctypes.data_as(dp) for d in pdiffs])
    bdiffs = [np.full(b.shape, np.nan) for b in bots]
    bdptr = (dp * len(bots))(*[d.ctypes.data_as(dp) for d in bdiffs])
    td = None if top_diff is None else np.ascontiguousarray(top_diff, np.float64)
    pd = None if propagate_down is None else (C.c_int * len(bots))(*[1 if x else 0 for x in propagate_down])
    top = np.empty(top_capacity, np.float64)
    tdims, taxes = (C.c_int * 8)(), C.c_int(0)
    err = C.create_string_buffer(512)
    rc = f(prototxt.encode(), len(bots), axes, dims_c, bptr, len(params), pptr,
           None if td is None else td.ctypes.data_as(dp), pd, top.ctypes.data_as(dp), C.c_longlong(top_capacity),
           tdims, C.byref(taxes), bdptr, pdptr, err, 512)
    if rc:
        raise RuntimeError("mms_layer_run_f64: %s (code %d)" % (err.value.decode(), rc))
    shape = tuple(tdims[a] for a in range(taxes.value))
    n = int(np.prod(shape)) if shape else 1
    return top[:n].reshape(shape).copy(), bdiffs, pdiffs
