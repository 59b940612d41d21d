"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/mbpls_hip.h declares; the ctypes binding covers all of them.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mbpls_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mbpls_hip_\w+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from mbpls_amd import _capi
    names = declared_symbols()
    assert len(names) >= 30
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for nm in names:
        assert hasattr(lib, nm), f"{nm} declared in the header but not exported by the library"
    assert sorted(_capi.SIGNATURES) == names, set(names) ^ set(_capi.SIGNATURES)
    assert _capi.load().mbpls_hip_abi_version() == _capi.ABI_VERSION


def test_no_gpu_means_loud_failure():
    """Without a device the product must raise, never fall back to a CPU path."""
    from mbpls_amd import _capi
    if _capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    import numpy as np
    from mbpls_amd import MBPLS
    with pytest.raises(_capi.HipError):
        MBPLS().fit(np.random.rand(10, 20), np.random.rand(10))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mbpls_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("the oracle", "").replace("CPU oracle", ""), f


def test_struct_layouts_match_the_header(tmp_path):
    """The ctypes mirrors of the header's structs (mbpls_nipals_step, mbpls_copy, mbpls_unit_scale) have the size and the
    field offsets a C compiler gives them: a field added on one side only would silently shift every argument behind it."""
    import ctypes
    import shutil
    import subprocess
    from mbpls_amd import _capi
    gcc = shutil.which("gcc") or shutil.which("cc")
    if gcc is None:
        pytest.skip("no C compiler here")
    structs = {"mbpls_nipals_step": _capi.NipalsStep, "mbpls_copy": _capi.Copy, "mbpls_unit_scale": _capi.UnitScale}
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "' + os.path.join(ROOT, "include", "mbpls_hip.h") + '"', "int main(void) {"]
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([gcc, "-std=c11", str(src), "-o", str(exe)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    got = {tuple(ln.split()[:2]): int(ln.split()[2]) for ln in out if ln.strip()}
    for cname, cls in structs.items():
        assert got[(cname, "size")] == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)
