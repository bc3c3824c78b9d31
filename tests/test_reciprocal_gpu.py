"""The aligner kernels take 1.0f / x as v_rcp_f32 + one Newton step where that is the IEEE quotient (csrc/prs_device.h, recip_exact):
every one of the 2^32 operands through the function as it is built into the library, against the device's own IEEE division.
The CPU checker divides with the host's IEEE division, so this is what ties the short form to it (tools/probes/rcp_exact_probe.hip
is the stand-alone version that found the domain)."""
import pytest

from srrg2_proslam_amd import ops

pytestmark = pytest.mark.gpu


def test_every_float_operand_gives_the_ieee_reciprocal(hip_ctx):
    differ, short_form = ops.selftest_reciprocal(hip_ctx)
    assert differ == 0
    # both signs x 252 binades x 2^23 mantissas are eligible; a wave that also holds an operand outside them takes the long form
    assert short_form >= 2 * 250 * (1 << 23)
