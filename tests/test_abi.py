"""The C-ABI library loads and exports every symbol include/freddie_seg.h declares (no compute)."""
import os
import re

import pytest

from freddie_amd import _lib, build


def header_functions():
    text = open(os.path.join(build.INCLUDE, "freddie_seg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fseg_[a-z_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    L = _lib.load()
    declared = header_functions()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(L, name), "libfreddie_seg.so does not export %s" % name
    assert sorted(_lib.EXPORTS) == declared
    assert L.fseg_abi_version() == 2
    assert L.fseg_source_hash().decode() == build.embedded_hash(build.SEG_SO) == build.seg_hash()
    assert L.fseg_n_stages() >= 8


def test_host_library_exports_every_declared_symbol():
    from freddie_amd import _host
    text = open(os.path.join(build.INCLUDE, "freddie_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(fhost_[a-z_]+)\s*\(", text)))
    assert len(declared) >= 14
    L = _host.load()
    for name in declared:
        assert hasattr(L, name), "libfreddie_host.so does not export %s" % name


def test_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.SegError, match="no CPU fallback"):
        _lib.Context(0)


def test_product_does_not_import_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs may use oracle/: nothing in the package, the
    drop-in scripts or the tools does."""
    root = os.path.dirname(build.INCLUDE)
    for sub in ("freddie_amd", "py", "tools"):
        for dirpath, _, files in os.walk(os.path.join(root, sub)):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".h", ".c", ".sh")):
                    src = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), os.path.join(dirpath, f)
                    assert "libfreddie_oracle" not in src, os.path.join(dirpath, f)
