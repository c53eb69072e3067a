"""CPU suite, part 2: the C-ABI library builds for gfx950, loads, and exports
every symbol include/mms.h declares (no compute calls without a GPU)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "mms.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mms_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(hiplib):
    from mms_answer_selection_amd import capi
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(hiplib, n), "include/mms.h declares %s but libmms_hip.so lacks it" % n
    assert set(names) == set(capi.EXPORTED_SYMBOLS), "capi.py binding table out of sync with mms.h"


def test_version_and_error_strings(hiplib):
    assert hiplib.mms_version() == 212          # include/mms.h MMS_VERSION: bumped with every incompatible change
    assert hiplib.mms_error_string(0) == b"ok"
    assert b"workspace" in hiplib.mms_error_string(3)


def test_workspace_queries_are_host_only(hiplib):
    assert hiplib.mms_simcross_workspace_bytes(1, 4096, 1, 1, 300, 1) == 0
    assert hiplib.mms_simcross_workspace_bytes(0, 4096, 1, 1, 300, 1) == 0
    need = hiplib.mms_simcross_workspace_bytes(2, 16384, 1, 1, 300, 1)
    assert need >= 2 * 16384 * 300 * 4
    assert hiplib.mms_simcross_workspace_bytes(7, 1, 1, 1, 1, 1) == 0       # bad mode
    assert hiplib.mms_pairrank_workspace_bytes(4096) in (0, 16)
    assert hiplib.mms_triplet_workspace_bytes(4096) == 1056 * 8 + 4096 * 4   # arrival words, then one term per triplet
    assert hiplib.mms_triplet_workspace_init(None, 0, None) == 3             # MMS_ERR_WORKSPACE, nothing enqueued
    assert hiplib.mms_simmatrix_workspace_bytes(16384, 300, 300) > 16384 * 300 * 4
    # ... and holds the operand image of the bf16 pipe: 19 k-steps x 10 column tiles x 3 planes x 1 KB for W (300, 300),
    # next to the split-K slabs of dW (64 chunks at 16384 pairs) and the fp32 path's W^T
    sm = hiplib.mms_simmatrix_workspace_bytes(16384, 300, 300)
    assert sm >= 16384 * 300 * 4 + 64 * 300 * 300 * 4 + 300 * 300 * 4 + 19 * 10 * 3 * 1024
    assert hiplib.mms_simmatrix_workspace_bytes(16384, 304, 300) >= sm          # grows with the shape
    # entry points that need a workspace say so without one (host-side checks only; no launch)
    assert hiplib.mms_simmatrix_forward_ws_f32(8, 4, 4, 1, 1, 1, 1, 1, None, 0, None) == 3
    assert hiplib.mms_set_matrix_mode(7) == 1 and hiplib.mms_get_matrix_mode() == 0


def test_code_object_is_gfx950_only():
    """Single code path: every embedded code object targets gfx950."""
    so = os.path.join(ROOT, "mms_answer_selection_amd", "libmms_hip.so")
    blob = open(so, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_product_does_not_touch_the_oracle():
    """The product package must not import or link oracle/ (tier rule 3)."""
    pkg = os.path.join(ROOT, "mms_answer_selection_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                src = open(os.path.join(dp, f)).read()
                assert "cpu_oracle" not in src and "mms_oracle" not in src, os.path.join(dp, f)
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mms_answer_selection_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(capi.MMSError):
        capi.lib()
