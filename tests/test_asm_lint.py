"""CPU tier: the build's lint of the hand-tracked load pipelines (seq-align_amd/tools/check_inflight_loads.py).

sa_sw_sweep.hip and sa_reduce.hip request rows with inline-asm loads and claim them with an explicit s_waitcnt; that is only sound
while the compiler never touches a destination register in between.  The Makefile runs the lint on the device assembly of both
files and fails the build on a finding.  Here: the lint itself on hand-made assembly (it must see each hazard and accept the
clean forms), and on the assembly the current build produced."""
import importlib.util
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location("check_inflight_loads", ROOT / "seq-align_amd" / "tools" / "check_inflight_loads.py")
lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(lint)

HEAD = "_Z6kernelv:\n"
TAIL = "\ts_endpgm\n\t.amdhsa_kernel _Z6kernelv\n\t\t.amdhsa_private_segment_fixed_size {scratch}\n\t.end_amdhsa_kernel\n"
LOAD = "\t;;#ASMSTART\n\t{nop}global_load_dwordx4 v[4:7], v2, s[4:5]\n\t;;#ASMEND\n"
CLAIM = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n"


def findings(tmp_path, body, scratch=0):
    f = tmp_path / "k.s"
    f.write_text(HEAD + body + TAIL.format(scratch=scratch))
    kernels, found = lint.check(str(f))
    assert kernels == 1
    return found


def test_lint_accepts_the_clean_pipeline(tmp_path):
    body = LOAD.format(nop="") + "\tv_add_u32_e32 v9, v8, v8\n" + CLAIM + "\tv_add_u32_e32 v9, v4, v5\n"
    assert findings(tmp_path, body) == []


def test_lint_sees_a_register_touched_in_flight(tmp_path):
    body = LOAD.format(nop="") + "\tv_mov_b32_e32 v9, v5\n" + CLAIM
    assert any("touches registers in flight" in f for f in findings(tmp_path, body))
    spill = LOAD.format(nop="") + "\tscratch_store_dwordx4 off, v[4:7], off offset:4\n" + CLAIM
    assert any("touches registers in flight" in f for f in findings(tmp_path, spill))


def test_lint_sees_scratch_and_overlapping_loads(tmp_path):
    body = LOAD.format(nop="") + CLAIM
    assert any("scratch" in f for f in findings(tmp_path, body, scratch=36))
    two = LOAD.format(nop="") + "\t;;#ASMSTART\n\tglobal_load_dwordx4 v[6:9], v3, s[4:5]\n\t;;#ASMEND\n" + CLAIM
    assert any("overwrites registers still in flight" in f for f in findings(tmp_path, two))


def test_lint_sees_the_sgpr_hazard_and_accepts_the_nop(tmp_path):
    """gfx9: five wait states between a VALU write of an SGPR (here the reload of a spilled base address) and a VMEM read of it;
    the compiler's hazard recogniser does not look into asm -- the statement must open with s_nop 4."""
    reload_ = "\tv_readlane_b32 s5, v93, 5\n"
    assert any("SGPR" in f for f in findings(tmp_path, reload_ + LOAD.format(nop="") + CLAIM))
    assert findings(tmp_path, reload_ + LOAD.format(nop="s_nop 4\n\t") + CLAIM) == []


def test_the_build_products_are_clean():
    """What `make` produced for the two files (kept by -save-temps=obj): no findings, and the kernels are there."""
    files = sorted((ROOT / "seq-align_amd" / "build" / "csrc").glob("sa_*-hip-amdgcn-amd-amdhsa-gfx950.s"))
    names = {f.name.split("-hip-")[0] for f in files}
    assert {"sa_sw_sweep", "sa_reduce"} <= names, "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    total = 0
    for f in files:
        kernels, found = lint.check(str(f))
        assert found == [], found
        total += kernels
    assert total >= 14          # 12 instantiations of the direction-byte sweep + 2 of the reduction
