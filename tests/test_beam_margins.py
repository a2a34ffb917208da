"""The margins of the conservative ("beam") pruning are machine-checked.

`differt_amd/csrc/beam_margins.hpp` holds every constant; the kernels use nothing else, and
`oracle/studies/beam_bounds_check.py` derives worst-case rounding bounds of the reference's float32 operation sequence by
interval arithmetic and checks every inequality of the argument against those constants.  A constant changed without the
check passing fails HERE, on the CPU."""

from __future__ import annotations

import importlib.util
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "differt_amd" / "csrc"
spec = importlib.util.spec_from_file_location("beam_bounds_check", ROOT / "oracle" / "studies" / "beam_bounds_check.py")
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def test_the_constants_in_the_header_pass_the_check():
    res = chk.check()
    bad = [c for c in res["checks"] if not c["ok"]]
    assert not bad, bad
    assert len(res["checks"]) > 30


@pytest.mark.parametrize("name,value", [
    ("kKappaDefault", 16.0),        # a quarter of the unit: neither the positional nor the lateral budget holds
    ("kKappaDefault", 32.0),
    ("kLateralConst", 1.0),         # the constant share of a mirror must cover a second unfolding (255 ulp(M))
    ("kLateralSigma", 0.5),         # the sigma share must cover Moller-Trumbore's third-edge uncertainty (56 sigma)
    ("kSpreadFactor", 1.0),         # below sqrt(3)
    ("kSideEpsFactor", 1.0),        # round 5's value: kappa sigma < the positional coefficient
    ("kFaceEpsFactor", 1.0),
    ("kSideUnits", 0.25),
    ("kChildFaceOffRatio", 1.05),   # not wider than the receiver stage's
    ("kChildFaceUnits", 1.0),
    ("kChildFaceUnits", 1.1),       # round 5's value: 6 ulp(M), less than the rounding of the box's support
    ("kChildDeltaRoundUp", 1.0),
    ("kBoxExtraUnits", 0.0),        # round 5: a box test no more conservative than the point test it stands for
    ("kRhoRoundDown", 1.0),
    ("kSlopeRounding", 1e-7),
])
def test_the_check_has_teeth(name, value):
    m = chk.parse_margins()
    assert name in m
    m[name] = value
    res = chk.check(m)
    assert not res["ok"], (name, value)


def test_derived_bounds_hold_on_adversarial_inputs():
    """float32 oracle vs float64 at the top of a binade, grazing incidences, slivers: measured <= derived."""
    d = chk.check()["derived"]
    meas = chk.measure(40_000, seed=11)
    for k, v in meas.items():
        assert v <= d[k], (k, v, d[k])
    # and the measurement is not vacuous: errors of the order of an ulp(M) do occur
    assert meas["lateral_u0"] > 0.5 and meas["plane_u0"] > 0.5 and meas["image_apex_u0"] > 2.0
    assert meas["mt_first_u0"] > 0.2 and meas["mt_third_per_sigma_u0"] > 0.2


def test_kernels_read_their_margins_from_the_header():
    consts = chk.parse_margins()
    src = (CSRC / "beam.hip").read_text() + (CSRC / "mesh.hip").read_text()
    unused = [k for k in consts if f"margins::{k}" not in src]
    assert not unused, f"constants of beam_margins.hpp no kernel uses: {unused}"
    # no margin literal left behind in the code of the tests (comments aside): the literals the header replaced
    code = "\n".join(ln.split("//")[0] for ln in (CSRC / "beam.hip").read_text().splitlines())
    for lit in ("1.0101f", "0.9999f", "1.05f", "1.06f", "2.1e-4f", "2e-6f", "1.00002f", "0.50001f", "1.000001f", "64.0f"):
        assert lit not in code, lit
    assert re.search(r"#include \"beam_margins.hpp\"", (CSRC / "beam.hip").read_text())


def test_header_syntax_contract():
    """One `constexpr float kName = <literal>f;` per constant, no expressions: what the parser relies on."""
    text = (CSRC / "beam_margins.hpp").read_text()
    decl = [ln for ln in text.splitlines() if ln.strip().startswith("constexpr")]
    assert len(decl) == len(chk.parse_margins()) >= 20
    for ln in decl:
        assert re.match(r"\s*constexpr float k\w+ = [-+0-9.eE]+f;", ln), ln
