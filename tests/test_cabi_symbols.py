"""CPU checks of the boundary: the shared library loads and exports every symbol include/sfgwas_hip.h declares,
and fails loudly (no CPU fallback) when no GPU is present."""
import os
import re
import ctypes as C
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "sfgwas_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(sfg_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from sfgwas_amd import capi
    lib = capi.lib()
    names = declared_symbols()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/sfgwas_hip.h but not exported: {missing}"
    assert set(names) == set(capi.EXPORTS), "capi.py signature table out of sync with the header"


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sfgwas_amd import capi
    import oracle_lib as ol
    with pytest.raises(capi.SfgError, match="no HIP device|hip"):
        capi.Context(ol.Q_PN14, ol.P_PN14)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sfgwas_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in txt and "liboracle" not in txt and "sfgwas_oracle" not in txt, f"{f} references the oracle"
