"""a bounded run of tools/fuzz_rows.py: randomised scene-clipper, brute-force matcher and feature-extraction cases
against the CPU oracle (needs an MI355X)"""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_randomised_clipper_bruteforce_and_feature_cases(oracle, hip_ctx):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_rows
    out = fuzz_rows.run(24, 20200306, ctx=hip_ctx, oracle=oracle, verbose=False)
    for row, (bad, total) in out.items():
        assert bad == 0, row
    assert out["clip"][1] > 1000 and out["bruteforce"][1] > 50 and out["features"][1] > 20
