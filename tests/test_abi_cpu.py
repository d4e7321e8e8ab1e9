"""CPU: the C-ABI shared library loads and exports every symbol include/splice_hip.h declares,
and the ctypes binding covers the same set (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

from splice_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "splice_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(splice_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    syms = _header_symbols()
    assert len(syms) >= 40
    assert os.path.exists(_lib.LIB_PATH), "libsplice_hip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_binding_matches_header():
    syms = set(_header_symbols())
    bound = set(_lib.exported_symbols())
    assert bound <= syms, sorted(bound - syms)
    assert syms <= bound, sorted(syms - bound)


def test_version_and_error_string():
    lib = _lib.lib()
    assert lib.splice_version() >= 100
    assert isinstance(lib.splice_last_error(), bytes)


def _c_layout(struct, fields, tmp_path):
    """sizeof + offsetof of every field, as the C compiler lays out include/splice_hip.h (gcc, x86-64)."""
    import subprocess
    src = tmp_path / f"layout_{struct}.c"
    lines = "".join(f'    printf("{f} %zu\\n", offsetof({struct}, {f}));\n' for f in fields)
    src.write_text(f'#include <stdio.h>\n#include <stddef.h>\n#include "splice_hip.h"\nint main(void) {{\n'
                   f'    printf("sizeof %zu\\n", sizeof({struct}));\n{lines}    return 0;\n}}\n')
    exe = tmp_path / f"layout_{struct}"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    return dict(zip(out[0::2], map(int, out[1::2])))


def test_struct_layouts_match_header(tmp_path):
    """The ctypes mirrors of the public structs agree with the C compiler's layout of the header: same size, same
    offset for every field, and no field of the header missing from the binding (or vice versa)."""
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "splice_hip.h")).read(), flags=re.S)
    src = re.sub(r"\[[A-Za-z_0-9]+\]", "", src)   # array extents (int down[SPLICE_GEN_MAX_SCALES]) do not matter for the field list
    for cname, mirror in (("splice_gemm_epilogue", _lib.GemmEpilogue), ("splice_step_config", _lib.StepConfig), ("splice_gen_arch", _lib.GenArch)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        header_fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                header_fields += [re.sub(r"[*\s]", "", part).split()[-1] if " " in part.strip() else part.strip().lstrip("*")
                                  for part in re.sub(r"^(const\s+)?[A-Za-z_0-9 ]+?[ *]+(?=[A-Za-z_0-9]+\s*(,|$))", "", decl).split(",")]
        bound_fields = [f[0] for f in mirror._fields_]
        assert header_fields == bound_fields, (cname, header_fields, bound_fields)
        c = _c_layout(cname, bound_fields, tmp_path)
        assert ctypes.sizeof(mirror) == c["sizeof"], (cname, ctypes.sizeof(mirror), c["sizeof"])
        for f in bound_fields:
            assert getattr(mirror, f).offset == c[f], (cname, f, getattr(mirror, f).offset, c[f])


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No CPU fallback: without the built HIP extension the binding raises and says what to do."""
    from splice_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libsplice_hip.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_product_entry_points_refuse_a_cpu_only_box(tmp_path):
    """train_model / train_pairs never route through the oracle or torch-CPU: on a box without a GPU they stop at once."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    from splice_amd.train import train_model, train_pairs
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        train_model(str(tmp_path), cfg_overrides=dict(seed=1))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        train_pairs([str(tmp_path)], cfg_overrides=dict(seed=1))
    import glob, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    offenders = [f for f in glob.glob(os.path.join(root, "splice_amd", "*.py")) if "import oracle" in open(f).read() or "from oracle" in open(f).read()]
    assert offenders == []                       # the oracle is test infrastructure only
