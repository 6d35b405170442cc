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


def test_every_option_is_documented_in_the_header():
    """the names mer_context_set_option accepts (the table in csrc/mer_api.hip) and the names include/mer.h documents are the same set"""
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    api = open(os.path.join(root, "mitsubaer_amd", "csrc", "mer_api.hip")).read()
    table = api[api.index("option_slot("):]
    table = table[:table.index("for (const auto &t : table)")]
    accepted = set(re.findall(r'\{"([a-z_]+)",\s*&o\.', table))
    assert len(accepted) >= 20
    hdr = open(os.path.join(root, "include", "mer.h")).read()
    doc = hdr[hdr.index("Scheduling / A-B options of a context"):]
    doc = doc[:doc.index("*/")]
    names = doc[doc.index("Names:"):]
    documented = set(re.findall(r"^\s{5}([a-z_]+)\s", names, re.M)) | set(re.findall(r"([a-z_]+)", names.strip().splitlines()[-1]))
    assert accepted <= documented, sorted(accepted - documented)
    # and the struct that holds them has a field per accepted name
    internal = open(os.path.join(root, "mitsubaer_amd", "csrc", "mer_internal.hpp")).read()
    fields = set(re.findall(r"int64_t ([a-z_]+) =", internal[internal.index("struct Options"):]))
    assert accepted == fields, sorted(accepted ^ fields)
