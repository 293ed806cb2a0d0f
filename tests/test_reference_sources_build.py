"""Drop-in check: the reference's OWN example programs and its test
driver compile UNMODIFIED against include/ and link against seq-align_amd/lib/libseqalign_hip.so.

Compile + link only (no GPU here, and the library has no CPU path); running them is the GPU tier's job
(tests/test_gpu_parity.py::test_legacy_api_known_answers replays tests.c's vectors).  Skipped where
/root/reference does not exist (the GPU box): nothing is copied from it, the sources are compiled where they lie.
"""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
LIB = ROOT / "seq-align_amd" / "lib"

SOURCES = [
    ("gcc", ["-std=c99"], REF / "examples" / "nw_example.c"),
    ("gcc", ["-std=c99"], REF / "examples" / "sw_example.c"),
    ("g++", [], REF / "examples" / "nw_example.cpp"),
    ("gcc", ["-std=c99"], REF / "src" / "tools" / "tests.c"),
]


@pytest.mark.skipif(not REF.exists(), reason="/root/reference absent (GPU box)")
@pytest.mark.parametrize("cc,flags,src", SOURCES, ids=[s[2].name for s in SOURCES])
def test_reference_program_builds_against_our_headers_and_library(cc, flags, src, tmp_path):
    assert (LIB / "libseqalign_hip.so").exists(), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    assert shutil.which(cc)
    exe = tmp_path / (src.stem + "_" + cc)
    cmd = [cc, *flags, "-O1", "-Wall", "-Wextra", "-Werror", "-I", str(ROOT / "include"), str(src), "-o", str(exe),
           "-L", str(LIB), "-lseqalign_hip", f"-Wl,-rpath,{LIB}", "-Wl,-rpath,/opt/rocm/lib"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    # every symbol the program needs from the reference library resolves inside ours
    und = subprocess.run(["nm", "-u", str(exe)], capture_output=True, text=True).stdout
    exported = subprocess.run(["nm", "-D", "--defined-only", str(LIB / "libseqalign_hip.so")],
                              capture_output=True, text=True).stdout
    defined = {ln.split()[-1] for ln in exported.splitlines() if ln.strip()}
    wanted = {ln.split()[-1] for ln in und.splitlines()
              if ln.split() and ln.split()[-1].startswith(("aligner", "alignment", "scoring", "needleman", "smith", "align_col"))}
    assert wanted and wanted <= defined, wanted - defined
