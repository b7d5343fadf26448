"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports exactly what
include/sybilgpu.h declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "sybilgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(sybl_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_symbols_are_exported_and_bound():
    from sybil_amd import _native
    lib = _native.lib()
    declared = _declared_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "libsybilgpu.so does not export %s" % name
        assert name in _native.SIGNATURES, "no ctypes signature for %s" % name
    assert sorted(_native.SIGNATURES) == declared
    assert lib.sybl_abi_version() == 5


def test_header_compiles_as_plain_c(tmp_path):
    import subprocess
    c = tmp_path / "t.c"
    c.write_text('#include "sybilgpu.h"\nint main(void){ sybl_query_desc d; (void)d; return SYBL_OK; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(c),
                           "-o", str(tmp_path / "t.o")])


def test_c_example_links_against_the_library(tmp_path):
    """tools/example_query.c: the whole open -> prepare -> scan -> finalize -> rows -> encode flow from C99,
    linked against the in-tree library (what a cgo build does)."""
    import subprocess
    import sybil_amd
    libdir = os.path.dirname(os.path.abspath(sybil_amd.__file__))
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "example_query.c"), "-L", libdir, "-lsybilgpu",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(tmp_path / "example_query")])


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import sybil_amd
    with pytest.raises(sybil_amd.SyblError) as ei:
        sybil_amd.Context(0)
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)


def test_struct_layouts_match_header():
    # sizes the C compiler gives the boundary structs vs the ctypes mirrors
    import subprocess
    import tempfile
    from sybil_amd import _native as N
    prog = r'''
#include <stdio.h>
#include "sybilgpu.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(sybl_col_view), sizeof(sybl_synth_col),
  sizeof(sybl_filter), sizeof(sybl_query_desc), sizeof(sybl_agg_out), sizeof(sybl_group_row), sizeof(sybl_run_stats)); return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        out = subprocess.check_output([os.path.join(d, "s")]).split()
    sizes = [ctypes.sizeof(x) for x in (N.ColView, N.SynthCol, N.Filter, N.QueryDesc, N.AggOut, N.GroupRow, N.RunStats)]
    assert [int(x) for x in out] == sizes


def test_every_environment_switch_of_the_library_is_documented():
    """DESIGN.md section 6 lists the diagnostic switches; a switch the sources read and no document names is drift."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for path in glob.glob(os.path.join(root, "sybil_amd", "csrc", "*")):
        if path.endswith((".cpp", ".h", ".hip")):
            names |= set(re.findall(r'env\("(SYBL_[A-Z0-9_]+)"\)', open(path).read()))
    assert len(names) > 40
    docs = "".join(open(os.path.join(root, d)).read() for d in ("DESIGN.md", "INTEGRATION.md", "README.md"))
    missing = sorted(n for n in names if n not in docs)
    assert not missing, missing


def test_documents_name_files_and_tests_that_exist():
    """A `profiles/...`, `tools/...`, `tests/...` path a document quotes must be in the tree (a prefix such as `profiles/r05_`
    must match something), and a `tests/file.py::test_name` must be a test of that file."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for d in ("DESIGN.md", "BASELINE.md", "README.md", "INTEGRATION.md"):
        text = open(os.path.join(root, d)).read()
        for m in re.findall(r'`((?:profiles|tools|tests|oracle|include|sybil_amd)/[A-Za-z0-9_./\-]+)', text):
            path = m.rstrip(".,;:)")
            if path == "oracle/_ref":  # (built, git-ignored)
                continue
            if not os.path.exists(os.path.join(root, path)) and not glob.glob(os.path.join(root, path) + "*"):
                bad.append((d, path))
        for f, t in re.findall(r'`(tests/[A-Za-z0-9_]+\.py)::(test_[A-Za-z0-9_]+)[`\[]', text):
            if os.path.exists(os.path.join(root, f)) and ("def " + t + "(") not in open(os.path.join(root, f)).read():
                bad.append((d, f + "::" + t))
    assert not bad, bad


def test_the_cgo_shim_in_the_integration_notes_names_what_the_header_declares():
    """INTEGRATION.md shows the binding a maintainer of the reference would add; a C.sybl_* call or a `sybl_*` name in it that
    the header does not declare is a shim that would not compile."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    hdr = open(os.path.join(root, "include", "sybilgpu.h")).read()
    names = set(re.findall(r"C\.(sybl_[a-z0-9_]+)", doc)) | set(re.findall(r"`(sybl_[a-z0-9_]+)[`(]", doc)) | set(re.findall(r"C\.(SYBL_[A-Z0-9_]+)", doc))
    assert len(names) >= 30
    missing = sorted(n for n in names if not re.search(r"\b" + n + r"\b", hdr))
    assert not missing, missing
