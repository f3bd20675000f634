"""CPU-side checks of the drop-in boundary: the library loads and exports every symbol
include/nbx.h declares, and the ctypes table mirrors the header (no compute calls)."""

import re
from pathlib import Path

import pytest

from nbed_amd import _nbx

REPO = Path(__file__).resolve().parent.parent
HEADER = (REPO / "include" / "nbx.h").read_text()


def declared_functions():
    # "int nbx_foo(" / "size_t nbx_foo(" / "const char* nbx_foo("
    return sorted(set(re.findall(r"^(?:int|size_t|const char\*)\s+(nbx_\w+)\s*\(", HEADER, flags=re.M)))


def test_header_and_ctypes_table_agree():
    assert declared_functions() == sorted(_nbx.SIGNATURES)


def test_library_exports_every_declared_symbol():
    if not _nbx.LIB_PATH.exists():
        pytest.fail(f"{_nbx.LIB_PATH} is not built; run __graft_entry__.build()")
    lib = _nbx.load_library()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.nbx_version() == _nbx.NBX_VERSION == 2


def test_every_entry_point_cites_the_reference():
    # each block comment of the header names the reference file it replaces
    for fn in ("nbx_jk_dense", "nbx_gemm", "nbx_eigh", "nbx_svd_right", "nbx_ao2mo", "nbx_spinorb_scatter",
               "nbx_huzinaga_sym", "nbx_trace_prod", "nbx_sym_pow"):
        idx = HEADER.index(f" {fn}(")
        assert "nbed/" in HEADER[max(0, idx - 2600):idx], fn


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_nbx.NbxUnavailableError):
        _nbx.load_library(tmp_path / "libnbx.so")


def test_no_gpu_means_no_backend():
    import torch

    from nbed_amd.backend import HipBackend

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_nbx.NbxUnavailableError):
        HipBackend()


def test_streaming_jk_kernels_have_no_scratch(tmp_path):
    """jk_m4_kernel keeps the row partials of a range in the loading waves' registers, 254 of 256 of them: one register
    more and the compiler spills -- and a scratch access in that loop waits behind the whole queue of chunk loads
    (DESIGN.md section 9 (xiv)).  The cross-compiled ISA must say private_seg_size 0 for every instance (N = 100 .. 148, one or two densities)."""
    import re
    import shutil
    import subprocess
    from pathlib import Path

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        import pytest

        pytest.skip("no hipcc")
    root = Path(__file__).resolve().parent.parent
    asm = tmp_path / "jk_m4.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", f"-I{root / 'include'}", "-S", "--cuda-device-only",
                    str(root / "nbed_amd" / "csrc" / "jk_m4.hip"), "-o", str(asm)], check=True, capture_output=True, timeout=600)
    text = asm.read_text()
    sizes = re.findall(r"jk_m4_kernelILi(\d+)ELi([12])E\S*\.private_seg_size, (\d+)", text)
    assert ("37", "1") in {(nb, k) for nb, k, _ in sizes} and ("37", "2") in {(nb, k) for nb, k, _ in sizes}, sizes
    assert len(sizes) >= 2 and all(int(v) == 0 for _, _, v in sizes), f"jk_m4_kernel spills: {sizes}"
