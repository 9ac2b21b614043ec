"""CPU tier: the C-ABI library builds for gfx950, loads, and exports every symbol include/stedm_hip.h declares.
No compute call is made (no GPU here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "stedm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(stedm_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from stedm_amd import build, _lib
    build.build(verbose=False)
    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for s in declared:
        assert hasattr(L, s), f"{s} declared in include/stedm_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in stedm_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == declared
    assert L.stedm_abi_version() == _lib.ABI_VERSION


def test_conv_args_struct_layout():
    """ctypes mirror of struct stedm_conv_args must match the C layout (checked against a gcc-compiled probe)."""
    import ctypes
    import subprocess
    import tempfile
    from stedm_amd import _lib
    fields = [f[0] for f in _lib.ConvArgs._fields_]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "stedm_hip.h"\nint main(){printf("%zu", sizeof(stedm_conv_args));' + \
        "".join(f'printf(" %zu", offsetof(stedm_conv_args, {f}));' for f in fields) + "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "p.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "p")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        vals = [int(v) for v in subprocess.check_output([exe]).split()]
    assert vals[0] == ctypes.sizeof(_lib.ConvArgs)
    for f, off in zip(fields, vals[1:]):
        assert getattr(_lib.ConvArgs, f).offset == off, f


def test_product_never_imports_oracle():
    """The product package must not import, call or link anything under oracle/."""
    pkg = os.path.join(ROOT, "stedm_amd")
    for dp, _dn, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{fn} imports oracle"
                assert "oracle/" not in txt and "oracle." not in txt.replace("the oracle.", ""), f"{fn} references oracle"


def test_ops_fail_loudly_without_gpu():
    import pytest
    import torch
    from stedm_amd import _lib, ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.StedmHipError):
        ops.gn_scale_shift(torch.zeros(1, 2, 2, 32), None, torch.ones(32), torch.zeros(32), 1e-5,
                           torch.zeros(1, 32), torch.zeros(1, 32))
