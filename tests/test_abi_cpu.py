"""CPU-only: the C-ABI library loads and exports every symbol include/maavss.h declares."""
import ctypes
import os

import pytest

from maavss_amd import _lib


def test_library_exports_every_declared_symbol():
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    protos = _lib.parse_header()
    assert len(protos) >= 5
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(cdll, name), f"{name} declared in include/maavss.h but not exported"
    L = _lib.lib()
    assert L.cdll.maavss_arch() == b"gfx950"
    assert L.cdll.maavss_version() >= 100


def test_no_cpu_fallback():
    import torch
    import maavss_amd
    st = maavss_amd.STFT(512, 66, device="cpu")
    with pytest.raises(_lib.MaavssError):
        st(torch.zeros(1, 4224))
