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
    assert lib.nbx_version() == _nbx.NBX_VERSION == 3


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


def _cross_compile_isa(tmp_path, name, extra=()):
    import shutil
    import subprocess
    from pathlib import Path

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        import pytest

        pytest.skip("no hipcc")
    root = Path(__file__).resolve().parent.parent
    asm = tmp_path / f"{name}.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", f"-I{root / 'include'}", *extra, "-S", "--cuda-device-only",
                    str(root / "nbed_amd" / "csrc" / f"{name}.hip"), "-o", str(asm)], check=True, capture_output=True, timeout=900)
    return asm.read_text()


def _asm_sgpr_hazards(text):
    """Inline-asm memory instructions that read an SGPR a VALU instruction (v_readlane / v_readfirstlane / a VALU compare)
    wrote fewer than five instructions earlier: the compiler's hazard recogniser does not look into asm statements, and a
    vector-memory instruction that reads an SGPR within five wait states of a VALU write of it sees the OLD value
    (that is how jk_mx.hip's first weights loads read the previous slot's base when their scalar bases were
    spilled into VGPR lanes)."""
    import re

    lines = [ln.strip() for ln in text.splitlines()]
    code = [(i, ln) for i, ln in enumerate(lines) if ln and not ln.startswith((";", ".", "//")) and not ln.endswith(":")]
    in_asm, bad = False, []
    asm_idx = set()
    for i, ln in enumerate(lines):
        if ln.startswith(";;#ASMSTART"):
            in_asm = True
        elif ln.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm and ln and not ln.startswith(";"):
            asm_idx.add(i)
    pos = {i: k for k, (i, _) in enumerate(code)}
    for i in sorted(asm_idx):
        ln = lines[i]
        if not ln.startswith(("global_load", "buffer_load", "s_mov_b32 m0")):
            continue
        used = set()
        for a, b in re.findall(r"s\[(\d+):(\d+)\]", ln):
            used.update(range(int(a), int(b) + 1))
        used.update(int(a) for a in re.findall(r"(?<![\w\[])s(\d+)\b", ln))
        k = pos[i]
        for back in range(1, 6):
            if k - back < 0:
                break
            prev = code[k - back][1]
            m = re.match(r"(v_readlane_b32|v_readfirstlane_b32)\s+s(\d+)", prev)
            if m and int(m.group(2)) in used:
                bad.append((prev, ln))
    return bad


def test_streaming_jk_kernels_have_no_scratch_and_no_asm_sgpr_hazard(tmp_path):
    """jk_m4_kernel keeps the row partials of a range in the loading waves' registers, 254 of 256 of them: one register
    more and the compiler spills -- and a scratch access in that loop waits behind the whole queue of chunk loads
    (DESIGN.md section 9 (xiv)).  The cross-compiled ISA must say private_seg_size 0 for every instance (N = 100 .. 148
    in jk_m4.hip, N = 152 .. 288 in jk_mx.hip; one or two densities), and no inline-asm load may read a scalar register
    that a v_readlane has just written."""
    import re

    # (jk_mx.hip a second time with two of the instances of jk_mx_hi.hip: band-segment chunks, N = 304 and 384)
    # (jk_m8.hip: the 8-fold form of N = 148 -- 235 registers, its J accumulators pinned in place)
    for name, kernel, probe, extra in (("jk_m4", "jk_m4", "37", ()), ("jk_m8", "jk_m8", "37", ()), ("jk_mx", "jk_mx", "64", ()),
                                       ("jk_mx", "jk_mx", "96", ("-DNBX_MX_SIZES(X)=X(76) X(96)", "-DMX_FN(name)=name##_hi"))):
        text = _cross_compile_isa(tmp_path, name, extra)
        sizes = re.findall(kernel + r"_kernelILi(\d+)ELi([12])E\S*\.private_seg_size, (\d+)", text)
        assert (probe, "1") in {(nb, k) for nb, k, _ in sizes} and (probe, "2") in {(nb, k) for nb, k, _ in sizes}, sizes
        assert len(sizes) >= 2 and all(int(v) == 0 for _, _, v in sizes), f"{name}_kernel spills: {sizes}"
        hazards = _asm_sgpr_hazards(text)
        assert not hazards, hazards[:5]


def test_asm_sgpr_hazard_checker_sees_the_pattern():
    text = """
	v_readlane_b32 s12, v183, 24
	v_readlane_b32 s13, v183, 25
	;;#ASMSTART
	global_load_dwordx4 v[134:137], v166, s[12:13]
	;;#ASMEND
	s_add_u32 s14, s60, 0x16000
	;;#ASMSTART
	global_load_dwordx4 v[122:125], v166, s[14:15]
	;;#ASMEND
"""
    bad = _asm_sgpr_hazards(text)
    assert len(bad) == 2 and all("s[12:13]" in b[1] for b in bad)


def test_packed_sizes_are_host_arithmetic_and_additive_over_slabs():
    """nbx_jk_packed_fold / nbx_eri_packed_bytes need no GPU: the 8-fold form of N = 97 .. 148 (every integral once + the
    padding of whole chunks: 0.624 GB at N = 148 against 0.992 GB of the 4-fold blocks), the 4-fold form above, and slabs
    that add up (what nbed_amd.dist.Shards.for_packed_jk cuts the rows of a multi-GPU run by)."""
    lib = _nbx.load_library()
    assert [lib.nbx_jk_packed_fold(n) for n in (24, 96, 97, 100, 147, 148, 149, 152, 256, 400, 402)] == [4, 4, 8, 8, 8, 8, 4, 4, 4, 4, 0]
    slack = lib.nbx_eri_packed_bytes(148, 7, 7)
    whole = lib.nbx_eri_packed_bytes(148, 0, 148) - slack
    assert whole == 624329728  # (tests/native/m8_geometry_check.hip prints the same number for NB = 37, LP = 6)
    unique = 8 * (148 * 149 // 2) * (148 * 149 // 2 + 1) // 2
    assert unique < whole < 1.3 * unique
    for cut in (1, 60, 116, 147):
        assert lib.nbx_eri_packed_bytes(148, 0, cut) + lib.nbx_eri_packed_bytes(148, cut, 148) - 2 * slack == whole
    rows = [lib.nbx_eri_packed_bytes(148, p, p + 1) - slack for p in range(148)]
    assert sum(rows) == whole and all(b >= a for a, b in zip(rows[:-1], rows[1:]))
    assert rows[147] > 500 * rows[0]  # (the 8-fold form's rows grow like p^3; the 4-fold form's like p)
    rows4 = [lib.nbx_eri_packed_bytes(256, p, p + 1) - lib.nbx_eri_packed_bytes(256, 3, 3) for p in range(256)]
    assert rows4[255] == 256 * rows4[0]


def test_jk_m8_geometry(tmp_path):
    """The compile-time geometry of csrc/jk_m8.hip (the 8-fold packed tiles: chunks of whole block rows, tile lengths and
    addresses, ring / LDS / vmcnt budgets, the staging order of the Dtot' table and of the J partials and its identity with
    jk_m4.hip's index formula when a tile has four chunks) checked on the host (tests/native/m8_geometry_check.hip)."""
    import shutil
    import subprocess
    from pathlib import Path

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        import pytest

        pytest.skip("no hipcc")
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "m8_geometry_check"
    subprocess.run([hipcc, "-std=c++17", "-O1", "--offload-arch=gfx950", f"-I{root / 'nbed_amd' / 'csrc'}",
                    str(root / "tests" / "native" / "m8_geometry_check.hip"), "-o", str(exe)], check=True, capture_output=True,
                   timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("bad=0") == 16, r.stdout
    # the bench's instance: four chunks, a ring of five, every integral once + the padding of whole chunks
    assert "NB=37 LP=6 NCH=4 RING=5" in r.stdout


def test_jk_mx_chunk_tables(tmp_path):
    """The compile-time chunk tables of csrc/jk_mx.hip (whole block rows at the top of a tile's triangle, band segments
    below) for every instantiated size, checked on the host: exact cover in order, ring-buffer and LDS budgets,
    block <-> (chunk, offset) round trip (tests/native/mx_geometry_check.hip)."""
    import shutil
    import subprocess
    from pathlib import Path

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        import pytest

        pytest.skip("no hipcc")
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "mx_geometry_check"
    subprocess.run([hipcc, "-std=c++17", "-O1", "--offload-arch=gfx950", f"-I{root / 'nbed_amd' / 'csrc'}",
                    str(root / "tests" / "native" / "mx_geometry_check.hip"), "-o", str(exe)], check=True, capture_output=True,
                   timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("bad=0") == 23, r.stdout
