"""A fixed range of seeds of the randomised parity soak (tests/soak.py; `python tools/soak.py` runs it for as long as it is told
to) with every GPU test run: packers, sample widths, byte orders, geometries, call sequences, entry points and the IIR stage drawn
at random, every stream and every decode against the oracle.  The two bugs the soak found in round 4 (profiles/r04_notes.md 4)
have regression tests of their own; this keeps the net itself in the suite."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", [700000, 700100, 700200])
def test_soak_seeds(first):
    import soak

    bad = []
    for seed in range(first, first + 100):
        res, desc = soak.one_case(seed)
        if res is not None and res[1]:
            bad.append(desc + ": " + "; ".join(res[1][:3]))
    assert not bad, "\n".join(bad)
