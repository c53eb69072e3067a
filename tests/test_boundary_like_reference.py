"""The boundary types of the plugin API exercised the way the reference's own unit tests
exercise them -- each case names the gtest it restates:

    src/caffe/test/test_blob.cpp            BlobSimpleTest
    src/caffe/test/test_syncedmem.cpp       SyncedMemoryTest (through Blob, which owns the SyncedMemory)
    src/caffe/test/test_filler.cpp          Constant / Uniform / Gaussian fillers
    src/caffe/test/test_layer_factory.cpp   LayerFactoryTest.TestCreateLayer
    src/caffe/test/test_embed_layer.cpp     EmbedLayerTest TestSetUp / TestForward / TestForwardWithBias

against the C++ mirror in csrc/caffe_api.hpp / caffe_layers.cpp (handle API of include/mms_layer.h)."""
import ctypes

import numpy as np
import pytest


@pytest.fixture(scope="module")
def L():
    from mms_answer_selection_amd import layers
    layers.lib()
    return layers


# ---- test_blob.cpp:26-59 ---------------------------------------------------------------------------
def test_blob_initialization_and_reshape(L):
    blob, pre = L.Blob(), L.Blob((2, 3, 4, 5))
    assert pre.shape == (2, 3, 4, 5) and pre.count == 120           # TestInitialization (:26-36)
    assert blob.shape == () and blob.count == 0
    blob.reshape(2, 3, 4, 5)                                         # TestReshape (:45-52)
    assert blob.shape == (2, 3, 4, 5) and blob.count == 120
    blob.reshape(0, 5)                                               # TestReshapeZero (:54-60)
    assert blob.count == 0
    # legacy accessors (blob.hpp:132-151): SimCross reads D as height() of a 3-axis blob
    b3 = L.Blob((7, 40, 50))
    lib = L.lib()
    assert [lib.mms_blob_shape(b3._h, i) for i in range(3)] == [7, 40, 50]


@pytest.mark.gpu                                                     # host memory is pinned in GPU mode (syncedmem.hpp:15-26)
def test_blob_pointers_first_touch_and_capacity(L):
    L.set_mode_gpu()
    lib = L.lib()
    pre = L.Blob((2, 3, 4, 5))
    for f in (lib.mms_blob_gpu, lib.mms_blob_cpu, lib.mms_blob_mutable_gpu, lib.mms_blob_mutable_cpu):
        assert f(pre._h, 0)                                          # TestPointersCPUGPU (:38-43)
    # capacity never shrinks (blob.cpp:23-45): the host pointer survives a smaller reshape
    p0 = ctypes.addressof(lib.mms_blob_mutable_cpu(pre._h, 0).contents)
    pre.reshape(2, 3)
    assert ctypes.addressof(lib.mms_blob_mutable_cpu(pre._h, 0).contents) == p0
    b = L.Blob((3, 5))                                               # syncedmem.cpp:28-29: zero-filled on first touch
    assert (b.data == 0).all() and (b.diff == 0).all()
    b.data[...] = 1.5
    assert (b.diff == 0).all() and (b.data == 1.5).all()


# ---- test_layer_factory.cpp:22-47 ------------------------------------------------------------------
def test_layer_factory_creates_every_registered_type(L, tmp_path):
    types = L.registered_layer_types()
    assert {"SimCross", "SimMatrix", "PairRankLoss"} <= set(types)
    for t in types:
        extra = ""
        if t == "HDF5Data":                                          # data layers expect a source (:30-41)
            src = tmp_path / "list.txt"
            src.write_text("")
            extra = ' top: "x" hdf5_data_param { source: "%s" batch_size: 1 }' % src
        if t == "Embed":
            extra = " embed_param { num_output: 3 input_dim: 2 }"
        lay = L.Layer('layer { name: "l" type: "%s"%s }' % (t, extra))
        assert lay.type == t                                         # EXPECT_EQ(iter->first, layer->type())


# ---- test_filler.cpp:27-41, 56-66, 122-142 (fillers as the layers invoke them in LayerSetUp) --------
def _filled(L, filler):
    L.set_mode_gpu()
    lay = L.SimMatrix(weight_filler=filler)
    q, a, top = L.Blob((4, 30)), L.Blob((4, 40)), L.Blob()
    lay.SetUp([q, a], [top])
    return lay.blobs[0].data.copy()


@pytest.mark.gpu
def test_constant_uniform_gaussian_fillers(L):
    L.set_random_seed(1701)
    w = _filled(L, dict(type="constant", value=10.0))
    assert w.shape == (30, 40) and (w == 10.0).all()                 # ConstantFillerTest.TestFill
    w = _filled(L, dict(type="uniform", min=1.0, max=2.0))
    assert w.min() >= 1.0 and w.max() <= 2.0 and w.std() > 0.2       # UniformFillerTest.TestFill
    mean, std = 3.0, 0.1
    w = _filled(L, dict(type="gaussian", mean=mean, std=std)).astype(np.float64)
    m = w.mean()
    var = ((w - mean) ** 2).mean()
    assert mean - 5 * std <= m <= mean + 5 * std                     # GaussianFillerTest.TestFill ("very loose")
    assert std * std / 5 <= var <= std * std * 5
    # the seed reproduces the fill (Caffe::set_random_seed, common.cpp:98-104)
    L.set_random_seed(7)
    w1 = _filled(L, dict(type="gaussian", mean=0.0, std=1.0))
    L.set_random_seed(7)
    w2 = _filled(L, dict(type="gaussian", mean=0.0, std=1.0))
    assert (w1 == w2).all()


# ---- test_syncedmem.cpp:53-120, test_embed_layer.cpp:38-135 (need the device) ------------------------
@pytest.mark.gpu
def test_syncedmem_cpu_write_gpu_read_gpu_write_cpu_read(L):
    import torch
    L.set_mode_gpu()
    b = L.Blob((10, 4))
    b.data[...] = 1.0                                                # TestCPUWrite: head at CPU
    dev = L.lib().mms_blob_gpu(b._h, 0)                              # TestGPURead: to_gpu(), SYNCED
    out = torch.empty(40, device="cuda")
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip.hipMemcpy(out.data_ptr(), ctypes.cast(dev, ctypes.c_void_p), 160, 3) == 0
    assert (out.cpu().numpy() == 1.0).all()
    two = torch.full((40,), 2.0, device="cuda")                      # TestGPUWrite: mutable_gpu_data(), head at GPU
    mdev = L.lib().mms_blob_mutable_gpu(b._h, 0)
    assert hip.hipMemcpy(ctypes.cast(mdev, ctypes.c_void_p), two.data_ptr(), 160, 3) == 0
    torch.cuda.synchronize()
    assert (b.data == 2.0).all()                                     # cpu_data() syncs back


@pytest.mark.gpu
@pytest.mark.parametrize("bias_term", [False, True])
def test_embed_layer_setup_and_forward_like_the_reference_tests(L, bias_term):
    L.set_mode_gpu()
    L.set_random_seed(1701)
    k_out, k_in = 10, 5
    kw = dict(num_output=k_out, input_dim=k_in, bias_term=bias_term,
              weight_filler=dict(type="uniform", min=-10.0, max=10.0))
    if bias_term:
        kw["bias_filler"] = dict(type="uniform", min=-10.0, max=10.0)
    lay = L.Embed(**kw)
    bottom, top = L.Blob((4, 1, 1, 1)), L.Blob()
    lay.SetUp([bottom], [top])
    assert top.shape == (4, 1, 1, 1, k_out)                          # TestSetUp (:38-52)
    assert len(lay.blobs) == (2 if bias_term else 1)
    assert lay.blobs[0].shape == (k_in, k_out)
    idx = np.random.default_rng(3).integers(0, k_in, 4)
    bottom.data[...] = idx.reshape(4, 1, 1, 1).astype(np.float32)
    lay.Forward([bottom], [top])
    w = lay.blobs[0].data
    want = w[idx]
    if bias_term:
        want = w[idx] + lay.blobs[1].data.reshape(1, k_out)          # TestForwardWithBias (:93-135): EXPECT_EQ, exact
    assert (top.data.reshape(4, k_out) == want).all()                # TestForward (:54-91)
