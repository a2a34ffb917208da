"""The error model behind the beam pruning's margins, checked on the CPU oracle: the float32
reflection point of the reference (_solver_image_method.py:116-135) errs ALONG its ray like 1 / cos(incidence), but its
lateral distance from the line and its distance from the mirror plane stay at the level of one ulp(M) at every incidence;
Moller-Trumbore's inside decision (_utils.py:1263-1322) is wrong only within ~1 ulp(M) / sin(ray, edge) of an edge line.
The margins in csrc/beam.hip assume bounds 5-50x above what is asserted here."""

import importlib.util
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
spec = importlib.util.spec_from_file_location("beam_error_model", ROOT / "oracle" / "studies" / "beam_error_model.py")
study = importlib.util.module_from_spec(spec)
spec.loader.exec_module(study)


@pytest.mark.parametrize("deg", [0, 60, 85, 89, 89.9])
def test_reflection_point_errors_split(deg):
    r = study.reflection_point_errors(deg, n=4000)
    assert r["samples"] > 2000
    assert r["lateral_max_u0"] < 4.0 and r["plane_distance_max_u0"] < 4.0   # bound used: 4.3 / 18 u0
    assert r["along_times_cos_max_u0"] < 4.0                                 # bound used: 13.4 u0
    if deg >= 89:
        assert r["along_max_u0"] > 5.0  # the along-ray error really is amplified: the split matters


@pytest.mark.parametrize("deg", [0, 70, 89])
def test_inside_test_uncertainty(deg):
    r = study.inside_test_uncertainty(deg, n=40000)
    assert r["samples"] > 20000
    assert r["largest_wrong_distance_times_sin_u0"] < 4.0                    # bound used: 46 u0 / sin
