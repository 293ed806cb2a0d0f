"""The scoring BUILDERS against the reference, once more under the gpu marker.

The GPU parity tests hand the oracle a byte copy of the product-built scoring_t (test_gpu_parity.py:
oracle_scoring_of), so a regression in scoring_init / scoring_add_* / the presets would make product and checker
wrong TOGETHER; the tests that pin the builders to the compiled reference's bytes (tests/test_host_api.py, CPU tier)
need no device -- but the GPU box is the only place the driver runs pytest at round end, so they run there too.
Reference: src/alignment_scoring.c:21-72, 307-392."""
import itertools

import pytest

import orclib as O
import seqalign_amd as S
import test_host_api as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", O.PRESETS)
def test_presets_equal_reference(name):
    H.test_presets_equal_reference(name)


def test_blosum62_export_matches_preset():
    H.test_blosum62_export_matches_preset()


def test_builders_equal_oracle_builders():
    H.test_builders_equal_oracle_builders()


def test_gpu_parity_tests_use_a_scoring_equal_to_the_checkers_own():
    """What test_gpu_parity.py relies on: for the specs it uses, the product's scoring_t bytes ARE what the checker
    would have built from the spec itself."""
    # (presets are table data, pinned by test_presets_equal_reference above; "default" is scoring_init 1/-2/-4/-1,
    # alignment_scoring.c:380-392)
    default = O.Scoring.from_buffer_copy(bytes(S.make_scoring({"preset": "default"})))
    assert O.scoring_defined_bytes(default) == O.scoring_defined_bytes(O.build_scoring({"init": [1, -2, -4, -1, 0, 0, 0, 0, 0, 0]}, "oracle"))
    specs = [{"init": [2, -2, -2, -1, 0, 0, 0, 0, 0, 0]},
             {"init": [1, -2, -4, 1, 0, 0, 0, 0, 0, 0]}, {"init": [1, -2, -4, -1, 1, 1, 0, 0, 0, 0], "mutations": [["a", "c", -2], ["c", "a", -1]]}]
    for flags in itertools.product([0, 1], repeat=5):
        specs.append({"init": [1, -2, -4, -1, *flags, 0], "wildcards": [["N", -1]]})
    for spec in specs:
        ours = O.Scoring.from_buffer_copy(bytes(S.make_scoring(spec)))
        theirs = O.build_scoring(spec, "oracle")
        assert O.scoring_defined_bytes(ours) == O.scoring_defined_bytes(theirs), spec
