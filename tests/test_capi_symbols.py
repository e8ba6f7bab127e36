"""CPU: the C-ABI library loads without a GPU and exports every symbol include/pcdet_mi355.h declares."""
import ctypes
import os
import re

from openpcdet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pcdet_mi355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pvx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    names = _declared_symbols()
    assert len(names) >= 10
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "missing export " + n
        assert n in _lib.SIGNATURES, "no ctypes signature for " + n
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_argument_errors_without_gpu():
    L = _lib.lib()
    assert b"gfx950" in L.pvx_version()
    assert L.pvx_boxes_iou_bev(None, -1, None, 0, None, None) == -1
    assert L.pvx_nms(None, -5, 0.5, 0, None, None, None, 0, None) == -1
    assert L.pvx_nms_workspace_bytes(500, 6) >= 6 * 500 * 8 * 8
