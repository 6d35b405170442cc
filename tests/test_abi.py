"""The C-ABI library loads and exports every symbol include/mer.h declares; without a GPU it fails loudly
(no CPU fallback).  No compute calls here."""
import ctypes
import os
import re
import numpy as np
import pytest
from mitsubaer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "mer.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mer_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = capi.lib()
    declared = _header_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(lib, s)]
    assert missing == []
    assert sorted(capi.SYMBOLS) == declared          # the Python binding covers the whole header
    assert lib.mer_abi_version() == 3
    chk = capi.lib(capi.CHECK_LIB_PATH)              # the bounds-checking build exports the same ABI
    assert [s for s in declared if not hasattr(chk, s)] == []


def test_struct_layouts_match_header():
    """ctypes mirrors of the POD structs have the size the C compiler gives them."""
    import subprocess, tempfile
    src = '#include "mer.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu\\n", sizeof(mer_grid_desc), sizeof(mer_scene_desc), sizeof(mer_shard));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        a, b, c = map(int, subprocess.check_output([os.path.join(d, "t")]).split())
    assert ctypes.sizeof(capi.GridDesc) == a
    assert ctypes.sizeof(capi.SceneDesc) == b
    assert ctypes.sizeof(capi.Shard) == c


def test_no_cpu_fallback():
    """Without a GPU the product refuses to run instead of silently computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.MerError, match="no HIP device"):
        capi.Context(0)


def test_product_does_not_import_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "mitsubaer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                for pat in ("import orc", "from oracle", "import oracle", "libmer_oracle", "mer_oracle.h", "orc_render", "orc."):
                    assert pat not in txt, (dirpath, f, pat)
