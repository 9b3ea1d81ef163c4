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


def test_go_shim_calls_only_declared_entry_points_with_the_declared_number_of_arguments():
    """integration/go/ cannot be compiled here (no Go toolchain): at least every C.sfg_* call in it must name a function include/sfgwas_hip.h declares and pass as
    many arguments as the declaration has, and every C.SFG_* constant must be one of the header's."""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "sfgwas_hip.h")).read(), flags=re.S)
    nargs = {}
    for m in re.finditer(r"\b(sfg_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        params = m.group(2).strip()
        nargs[m.group(1)] = 0 if params in ("", "void") else params.count(",") + 1
    consts = set(re.findall(r"\b(SFG_[A-Z0-9_]+)\b", hdr))

    def call_args(txt, start):                       # number of top-level arguments of the call whose '(' is at `start`
        depth, n, i, seen = 0, 0, start, False
        while i < len(txt):
            c = txt[i]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
                if depth == 0:
                    return n + (1 if seen else 0)
            elif c == "," and depth == 1:
                n += 1
            elif depth >= 1 and not c.isspace():
                seen = True
            i += 1
        raise AssertionError("unbalanced call")

    calls = 0
    for dirpath, _, files in os.walk(os.path.join(ROOT, "integration", "go")):
        for f in files:
            if not f.endswith(".go"):
                continue
            txt = open(os.path.join(dirpath, f)).read()
            code = re.sub(r"//[^\n]*", "", txt)
            for m in re.finditer(r"\bC\.(sfg_[a-z0-9_]+)\s*\(", code):
                name = m.group(1)
                assert name in nargs, f"{f}: C.{name} is not declared in include/sfgwas_hip.h"
                got = call_args(code, m.end() - 1)
                assert got == nargs[name], f"{f}: C.{name} called with {got} arguments, declared with {nargs[name]}"
                calls += 1
            for c in re.findall(r"\bC\.(SFG_[A-Z0-9_]+)\b", code):
                assert c in consts, f"{f}: C.{c} is not a constant of the header"
    assert calls >= 20


def test_go_shim_files_call_only_methods_the_hip_package_defines_with_matching_arity():
    """the four shim files (gwas/, crypto/, mpc/) reach the library through package hip: every `hip.X(...)` function and every `h.X(...)` / `hip.Default.X(...)`
    method they call must be defined in integration/go/hip/hip.go with that many parameters (no Go toolchain here to tell us otherwise)"""
    src = re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "integration", "go", "hip", "hip.go")).read())

    def count_params(sig):
        sig = sig.strip()
        if not sig:
            return 0
        n, depth, groups = 0, 0, []
        cur = ""
        for c in sig:
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
            if c == "," and depth == 0:
                groups.append(cur); cur = ""
            else:
                cur += c
        groups.append(cur)
        # Go lets names share a type ("s, inLevel, maxLevel int"): every comma-separated item is one parameter either way
        return len(groups)

    funcs, methods = {}, {}
    for m in re.finditer(r"^func\s+(\(\s*\w+\s+\*?(\w+)\s*\)\s*)?(\w+)\s*\(", src, flags=re.M):
        start = m.end()
        depth, i = 1, start
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        n = count_params(src[start:i - 1])
        (methods if m.group(1) else funcs)[m.group(3)] = n
    assert "Init" in funcs and funcs["Init"] == 4 and "MatmulResident" in methods and "CMultDev" in methods

    def call_args(txt, start):
        depth, n, i, seen = 0, 0, start, False
        while i < len(txt):
            c = txt[i]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
                if depth == 0:
                    return n + (1 if seen else 0)
            elif c == "," and depth == 1:
                n += 1
            elif depth >= 1 and not c.isspace():
                seen = True
            i += 1
        raise AssertionError("unbalanced call")

    checked = 0
    for sub in ("gwas", "crypto", "mpc"):
        d = os.path.join(ROOT, "integration", "go", sub)
        for f in os.listdir(d):
            if not f.endswith(".go"):
                continue
            code = re.sub(r"//[^\n]*", "", open(os.path.join(d, f)).read())
            for m in re.finditer(r"\bhip\.(?:Default\.)?([A-Z]\w*)\s*\(", code):
                name = m.group(1)
                table = methods if "Default." in m.group(0) else funcs
                assert name in table, f"{f}: hip.{name} is not defined in hip/hip.go"
                assert call_args(code, m.end() - 1) == table[name], f"{f}: hip.{name} called with {call_args(code, m.end() - 1)} arguments, defined with {table[name]}"
                checked += 1
            for m in re.finditer(r"\bh\.([A-Z]\w*)\s*\(", code):      # h := hip.Default / a fork
                name = m.group(1)
                assert name in methods, f"{f}: method {name} is not defined on hip.Ctx"
                assert call_args(code, m.end() - 1) == methods[name], f"{f}: h.{name} called with {call_args(code, m.end() - 1)} arguments, defined with {methods[name]}"
                checked += 1
    assert checked >= 25


def test_library_shard_arithmetic_matches_sharding_py():
    """sfg_mgpu_shard is host arithmetic (no device): the SNP-block windows the multi-GPU engine gives its ranks are those of sfgwas_amd/sharding.py"""
    from sfgwas_amd import capi
    from sfgwas_amd.sharding import snp_block_range
    lib = capi.lib()
    for m_snp in (1, 8192, 8193, 100_000, 1_000_000):
        for world in (1, 2, 3, 8):
            for r in range(world):
                v = [C.c_size_t() for _ in range(4)]
                assert lib.sfg_mgpu_shard(world, m_snp, r, *[C.byref(x) for x in v]) == 0
                assert tuple(x.value for x in v) == snp_block_range(m_snp, r, world)
    assert lib.sfg_mgpu_shard(2, 100, 2, None, None, None, None) != 0 and lib.sfg_mgpu_shard(0, 100, 0, None, None, None, None) != 0


def test_multi_gpu_engine_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sfgwas_amd import capi
    import oracle_lib as ol
    with pytest.raises(capi.SfgError, match="no HIP device|hip|device"):
        capi.MultiGpu(ol.Q_PN14, ol.P_PN14, devices=[0, 0])
