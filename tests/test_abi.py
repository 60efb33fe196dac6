"""The ABI guard: CGVP_ABI_VERSION in the header, ABI_VERSION in the ctypes binding, the number INTEGRATION.md
quotes and the number the built library reports must agree, and the header's declarations may not change without
a bump (digest of the declarations pinned next to the version in gvp_hip/_lib.py).  CPU only."""
import os
import re

from conftest import REPO
from gvp_hip import _lib

HEADER = os.path.join(REPO, "include", "caster_gvp.h")


def test_versions_agree_everywhere():
    header = open(HEADER).read()
    v = int(re.search(r"#define\s+CGVP_ABI_VERSION\s+(\d+)", header).group(1))
    assert v == _lib.ABI_VERSION
    integ = open(os.path.join(REPO, "INTEGRATION.md")).read()
    assert {int(n) for n in re.findall(r"C ABI v(\d+)", integ)} == {v}
    assert _lib.lib().cgvp_abi_version() == v          # the built .so (host-side call, no GPU needed)


def test_declarations_cannot_change_without_a_bump():
    assert _lib.abi_header_digest() == _lib.ABI_HEADER_SHA256, (
        "include/caster_gvp.h declarations changed: bump CGVP_ABI_VERSION (header), ABI_VERSION and "
        "ABI_HEADER_SHA256 (gvp_hip/_lib.py) and the version INTEGRATION.md quotes")


def test_digest_ignores_comments_but_not_prototypes(tmp_path):
    text = open(HEADER).read()
    a = tmp_path / "a.h"
    a.write_text(text.replace("/* Library self-description", "/* Reworded comment: library self-description"))
    assert _lib.abi_header_digest(str(a)) == _lib.ABI_HEADER_SHA256
    b = tmp_path / "b.h"
    b.write_text(text.replace("int32_t max_workgroups, void* stream);", "int64_t max_workgroups, void* stream);"))
    assert _lib.abi_header_digest(str(b)) != _lib.ABI_HEADER_SHA256


def test_header_documents_no_global_state():
    """the GINE workgroup cap was process-global state behind a 'no global state' header."""
    assert "cgvp_gine_bwd_workgroups" not in open(HEADER).read()
    assert "cgvp_gine_bwd_workgroups" not in _lib.exported_symbols()


def test_cpp_bridge_loads_and_matches_the_abi():
    """lib/caster_gvp_torch.so (csrc/torch_bridge.cpp, built by __graft_entry__.build()): importable without a GPU,
    built against the same C ABI, exports the two encoder entry points."""
    b = _lib.bridge()
    assert b is not None, "build it: bash caster-dta_amd/csrc/build_bridge.sh"
    assert b.abi_version() == _lib.ABI_VERSION
    assert callable(b.lba_encoder) and callable(b.gine_encoder)


def test_library_never_calls_hipmemset():
    """Captured into a HIP graph, hipMemsetAsync re-zeroes only part of its range from the second replay on (ROCm 7.2 on
    this pool; tools/memset_capture_probe.py): the library zero-fills with its own kernel and must not
    import any hipMemset* symbol."""
    import subprocess
    syms = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "hipLaunchKernel" in syms or "hipModuleLaunchKernel" in syms or "__hipPushCallConfiguration" in syms
    assert "emset" not in syms, [l for l in syms.splitlines() if "emset" in l]
