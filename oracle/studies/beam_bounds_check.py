#!/usr/bin/env python3
"""Mechanical check of the margins of the conservative ("beam") pruning (DESIGN.md section 9.3 / csrc/beam_margins.hpp).

TEST INFRASTRUCTURE (lives under oracle/: only tests/ may import it; the product never does).

What it does
------------
1. DERIVES, by interval arithmetic over the float32 operation sequence of the reference -- every operation of
   `image_of_vertex_with_respect_to_mirror` (differt/src/differt/geometry/_solver_image_method.py:68-79),
   `intersection_of_ray_with_plane` (:116-135), the backward scan (:152-203), the same-side test (:443-454) and
   Moller-Trumbore (_utils.py:1263-1322), in the order oracle/differt_oracle.c restates them -- worst-case bounds of the
   rounding errors the pruning argument rests on.  Every rounding is one term: |fl(x op y) - (x op y)| <= half an ulp of
   the largest magnitude the result can have, magnitudes being intervals propagated from the inputs' bounds.  The bounds
   hold for ALL inputs in the stated domain, not for a sample.
2. CHECKS every inequality the argument needs between those bounds and the constants the kernels use, which it parses from
   differt_amd/csrc/beam_margins.hpp (the one header the kernels read them from).
3. VALIDATES the derived bounds against measurement: the C oracle (float32) next to float64 on adversarial inputs
   (magnitudes at the top of a binade, grazing incidences, slivers): every measured error must lie below its derived
   bound (a derivation that forgot a term shows up here).

Domain (what "all inputs" means): finite float32 coordinates; M = largest coordinate magnitude of mesh, transmitters and
receivers, u0 = ulp(M); the apex of a prefix within 2 M mu per coordinate, mu = mag_scale >= 1 being the factor the kernels
multiply their unit with (beam.hip: mag_scale) -- all bounds scale linearly with mu, so they are derived at mu = 1; polygon
vertices unfolded through later mirrors stay within |apex| + sqrt(3) * 3 M of the origin (a reflection is an isometry:
the unfolded polygon is as far from the unfolded apex as the original polygon from its own apex).

What the check does NOT do: it does not re-prove the geometric composition (how the per-step bounds add along a prefix:
DESIGN 9.3 items 1-4) -- it evaluates the formulas of that argument with derived numbers instead of hand-counted ones --
and it does not cover paths whose consecutive reflection points coincide within the arithmetic's resolution (DESIGN 9.8:
the reference's inside test then runs on a direction that is rounding noise; tests/golden/beam_cases/sub_ulp_segment_*).

    python oracle/studies/beam_bounds_check.py [--json] [--no-measure]
"""
from __future__ import annotations

import json
import math
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
HEADER = ROOT / "differt_amd" / "csrc" / "beam_margins.hpp"

SQ3 = math.sqrt(3.0)
EPS = 2.0 ** -24  # unit roundoff of float32 (round to nearest)


# ---------------------------------------------------------------------------------------------------------------
# the constants the kernels use
# ---------------------------------------------------------------------------------------------------------------
def parse_margins(path: Path = HEADER) -> dict[str, float]:
    """`constexpr float kName = <literal>f;` lines of beam_margins.hpp (its syntax contract: no expressions)."""
    out = {}
    for m in re.finditer(r"constexpr\s+float\s+(k\w+)\s*=\s*([-+0-9.eE]+)f\s*;", path.read_text()):
        out[m.group(1)] = float(m.group(2))
    return out


# ---------------------------------------------------------------------------------------------------------------
# interval arithmetic over roundings.  Unit of magnitude: M_top = 2^24 u0 (the top of M's binade: M < M_top, the worst
# case -- absolute roundings are fixed per binade, so the largest M of a binade makes operands as large as they get).
# Unit of error: u0 = ulp(M).
# ---------------------------------------------------------------------------------------------------------------
def hu(m: float) -> float:
    """Half an ulp (in u0) of any float32 whose magnitude is < m (m in units of M_top): the largest rounding error of
    an operation whose exact result is below m in magnitude.  m in (1/2, 1] -> 1/2, (1, 2] -> 1, (2, 4] -> 2, ..."""
    if m <= 0.0:
        return 0.0
    return 2.0 ** (math.ceil(math.log2(m)) - 1)


class Ledger:
    """Named error terms (u0) of one derivation, so that the printed report shows where a bound comes from."""

    def __init__(self, name):
        self.name, self.terms = name, []

    def add(self, what, value):
        self.terms.append((what, float(value)))
        return float(value)

    def total(self):
        return sum(v for _, v in self.terms)

    def report(self):
        return {"bound_u0": self.total(), "terms": [{"what": w, "u0": v} for w, v in self.terms]}


def dot3_rounding(a_inf, a_2, b_inf, b_2):
    """Rounding error of ((a0 b0 + a1 b1) + a2 b2) evaluated in float32 on float DATA a, b (oracle dot3): three products
    (each below a_inf b_inf), the first sum (below min(2 a_inf b_inf, a_2 b_2)), the second sum (below a_2 b_2)."""
    return 3 * hu(a_inf * b_inf) + hu(min(2 * a_inf * b_inf, a_2 * b_2)) + hu(a_2 * b_2)


N_INF, N_2 = 1.0, 1.0 + 4 * EPS  # a float32 unit normal: components <= 1, length 1 +- 4 eps (normalize: sqrt, three divisions)


def image_error(x_inf: float, r_inf: float, led: Ledger | None = None, tag="") -> float:
    """IM:68-79 / oracle image_one.  x: the point (|x|_inf < x_inf), p: plane point (< 1), n: unit normal, result < r_inf.
        inc = x - p;  c = 2 * dot3(inc, n);  r_i = x_i - c * n_i
    Returns a bound (u0, Euclidean norm) of |fl(A(x)) - A(x)|, A(x) = x - 2 <x - p, n> n evaluated exactly on the float
    inputs (A is an affine map -- it need not be an isometry, |n| != 1 in the last bits; the argument only uses that the
    kernels unfold polygon vertices with the SAME map)."""
    led = led or Ledger("image")
    inc_inf, inc_2 = x_inf + 1.0, SQ3 * (x_inf + 1.0)
    e_inc = hu(inc_inf)                                    # per component of inc
    e_dot = SQ3 * e_inc * N_2 + dot3_rounding(inc_inf, inc_2, N_INF, N_2)   # |sum n_i d(inc_i)| <= |n|_1 e_inc <= sqrt3 e_inc
    e_c = 2.0 * e_dot                                      # the factor 2 is exact
    c_mag = SQ3 * (x_inf + r_inf) * N_2                    # |c| |n| = |x - A(x)| <= |x|_2 + |A(x)|_2
    per_comp = hu(c_mag) + hu(r_inf)                       # fl(c n_i), then fl(x_i - .)
    total = e_c * N_2 + SQ3 * per_comp
    led.add(f"{tag}2 x rounding of <x - p, n> (inc, 3 products, 2 sums)", e_c * N_2)
    led.add(f"{tag}products c n_i and final subtractions (Euclidean over 3 components)", SQ3 * per_comp)
    return total


def ray_plane_errors(apex_inf: float = 2.0):
    """IM:116-135 / oracle ray_plane_one inside the backward scan (IM:152-203): o = next point (< 1), I = image (< apex_inf),
        d = I - o;  v = p - o;  un = dot3(d, n);  vn = dot3(v, n);  t = vn / un;  P_i = o_i + d_i * t
    for a step with 0 <= t <= 1 (the reflection point lies between the next point and the image: what the same-side test
    enforces on every accepted candidate whose previous point is clear of the mirror plane).
    Returns (lateral, plane): Euclidean bounds (u0) of the distance of the computed point from the EXACT line through o and I,
    and of its distance from the mirror plane <x - p, n> = 0."""
    lat = Ledger("lateral distance of a computed reflection point from the exact line (next point, image)")
    d_inf = apex_inf + 1.0
    e_d = hu(d_inf)  # d~ = fl(I - o): the computed direction is off by this per component; |t| <= 1 scales it
    lat.add("t * (d~ - d): rounding of the direction, |t| <= 1", SQ3 * e_d)
    lat.add("products d_i t (Euclidean)", SQ3 * hu(d_inf))
    lat.add("final sums o_i + . (the point is a scene point: < M)", SQ3 * hu(1.0))
    pl = Ledger("distance of a computed reflection point from the mirror plane")
    # With Q = o + t~ d~ (exact operations on the float data):  <Q - p, n> = e_v + delta (VN + e_v) - t~ e_u  EXACTLY, where
    # VN = <p - o, n>, e_v / e_u = errors of the computed vn / un against the exact dot products of (p - o) and of d~ with n,
    # delta = rounding of the division.  No small denominator appears.
    v_inf, v_2 = 2.0, 2.0 * SQ3
    e_v = SQ3 * hu(v_inf) * N_2 + dot3_rounding(v_inf, v_2, N_INF, N_2)
    e_u = dot3_rounding(d_inf, SQ3 * d_inf, N_INF, N_2)   # d~ is float data of Q: no input error
    pl.add("e_v: v = fl(p - o), dot3(v, n)", e_v * (1 + EPS))
    pl.add("|t| e_u: dot3(d~, n)", e_u)
    pl.add("rounding of the division, |VN| <= |p - o|_2", v_2 * 2.0 ** 24 * EPS * (1 + EPS))
    pl.add("P = fl(Q): products and sums, projected on n", SQ3 * (hu(d_inf) + hu(1.0)))
    return lat, pl


def side_test_error():
    """IM:443-454 / oracle same_side: sign(dot3(fl(x - p), n)) for a scene point x.  The sign is right whenever the exact
    <x - p, n> exceeds this bound in magnitude."""
    led = Ledger("uncertainty of the reference's same-side sign (distance from the plane below which it may be wrong)")
    led.add("x - p (3 components, |n|_1 <= sqrt 3)", SQ3 * hu(2.0) * N_2)
    led.add("dot3 roundings", dot3_rounding(2.0, 2.0 * SQ3, N_INF, N_2))
    return led


def moller_trumbore_uncertainty():
    """UT:1263-1322 / oracle mt_one, seen from the apex of the pyramid over the triangle.

    The three range tests are sign tests of determinants.  With X^in = the point where the tested ray meets the triangle's
    plane and q = its in-plane distance from an edge line, the determinant of that edge's test equals |d| |e| q cos(phi)
    (phi = incidence), and as seen from the apex -- distance rho from the edge line -- the point lies q cos(phi) / rho
    (an angle) outside the pyramid's face over that edge: the incidence cancels.  So each test's absolute determinant
    error E, divided by |d| |e|, is the uncertainty of (angle x rho), in u0.  Relative error model per operation
    (|fl(z) - z| <= eps |z|: magnitudes vary freely here), |s| = |o - v0| <= 2 sqrt(3) M:
        h = cross(d, e2):        |err h|_2 <= (sqrt 2 + 1) eps |d| |e2|      (two products and one difference per component)
        A = dot3(s, h):          + eps (sum |s_i h_i| + |s_z h_z|) <= 2 eps |s| |h|   near the decision A = 0
        => E_A <= (sqrt 2 + 3) eps |s| |d| |e2|;   E_B likewise with e1;   E_a likewise with |e1| in place of |s|.
    u >= 0 and v >= 0 are the signs of A and B (times the sign of a).  u + v <= 1 compares fl(fl(f A) + fl(f B)) with 1:
        |fl(u + v) - (A + B) / a| <= (E_A + E_B + E_a) / |a| + 3 eps   for u, v in [0, 1]
    and (1 - (A + B) / a) a is the determinant of the third edge e3 = e2 - e1, so its (angle x rho) uncertainty is
        (E_A + E_B + E_a + 3 eps |a|) / (|d| |e3|)  <=  eps [ (sqrt2 + 3) |s| (|e1| + |e2|) / |e3| + (sqrt2 + 3 + 3) |e1| |e2| / |e3| ]
    with (|e1| + |e2|) / |e3| <= 2 sigma and |e2| / |e3| <= sigma (law of sines; sigma = largest 1 / sin(corner)).
    Returns (c_first, c_third): bounds of (angle x rho) in u0 for the two edges at v0, and the coefficient of sigma for the
    third edge."""
    s_2 = 2.0 * SQ3                       # |o - v0|_2 in units of M_top
    k = (math.sqrt(2.0) + 3.0)
    c_first = k * s_2 * 2.0 ** 24 * EPS   # eps * M_top = 1 u0 (M_top = 2^24 u0, eps = 2^-24)
    e1_2 = 2.0 * SQ3
    c_third = (k * s_2 * 2.0 + (k + 3.0) * e1_2) * 2.0 ** 24 * EPS
    led = Ledger("Moller-Trumbore: uncertainty of (angle seen from the apex) x (distance apex - edge line)")
    led.add("edges at v0 (u >= 0, v >= 0)", c_first)
    led.add("third edge (u + v <= 1), per unit of sigma", c_third)
    return led, c_first, c_third


# ---------------------------------------------------------------------------------------------------------------
def derive() -> dict:
    """All derived bounds, in u0 (mag_scale = 1)."""
    out = {}
    lat, pl = ray_plane_errors()
    out["lateral"] = lat.report()
    out["plane"] = pl.report()
    out["side_sign"] = side_test_error().report()
    mt, c_first, c_third = moller_trumbore_uncertainty()
    out["moller_trumbore"] = mt.report()
    out["mt_first"], out["mt_third_per_sigma"] = c_first, c_third
    # images: the apex (2 M -> 2 M), a polygon vertex unfolded once (M -> 2 M + 3 sqrt3 M) and once more
    r_unf = 2.0 + 3.0 * SQ3
    la = Ledger("image of an apex (|I| < 2 M before and after)")
    out["image_apex_val"] = image_error(2.0, 2.0, la)
    out["image_apex"] = la.report()
    l1 = Ledger("polygon vertex unfolded through one later mirror")
    out["image_vertex1_val"] = image_error(1.0, r_unf, l1)
    out["image_vertex1"] = l1.report()
    l2 = Ledger("polygon vertex unfolded through a second later mirror (rounding of the second step only)")
    out["image_vertex2_val"] = image_error(r_unf, r_unf, l2)
    out["image_vertex2"] = l2.report()
    # input perturbations Moller-Trumbore works on: s = fl(o - v0), e = fl(v - v0), d = fl(P_j - P_{j-1}): each moves a
    # point by at most sqrt(3) hu(2 M)
    out["mt_input_shift"] = SQ3 * hu(2.0)
    return out


SIGMA_MIN = 2.0 / SQ3  # the sharpest corner of a triangle is at most 60 degrees: sigma = 1 / sin >= 1.1547


def check(m: dict[str, float] | None = None, d: dict | None = None) -> dict:
    """Every inequality between derived bounds and header constants.  Returns {"ok": bool, "checks": [...]}."""
    m = m or parse_margins()
    d = d or derive()
    kappa = m["kKappaDefault"]
    lam, nu = d["lateral"]["bound_u0"], d["plane"]["bound_u0"]
    shift = d["mt_input_shift"]
    c1, c3 = d["mt_first"], d["mt_third_per_sigma"]
    side = d["side_sign"]["bound_u0"]
    checks = []

    def need(name, lhs, rhs, why):
        checks.append({"name": name, "lhs": lhs, "rhs": rhs, "ok": bool(lhs <= rhs), "why": why})

    # kernel-side evaluation error of <x - I, n> with fdot on float data (|x - I|_inf < 2 + 1 at mag_scale 1; FMA: one rounding
    # per accumulate) + the subtraction x - I -- in u0; used wherever the kernels compare a plane distance with a margin
    k_eval = SQ3 * hu(3.0) + 3 * hu(3.0 * SQ3)
    need("side test, exact points (transmitter, receivers): kSideUnits * kappa covers the reference's sign uncertainty + the "
         "kernel's own evaluation", side + k_eval, m["kSideUnits"] * kappa,
         "beam_seed / receiver_rest: side_of_range(d, d, kSideUnits u)")

    # ---- positional bound: in-plane distance of a computed reflection point from its primitive, times cos(incidence) ----
    # Moller-Trumbore's third-edge uncertainty (c3 sigma), the shifts of its inputs (origin, edge vectors, direction), and the
    # distance of the computed point from the point where the tested ray meets the plane (nu / cos(phi) along the ray)
    def c_pos(sig):
        return max(c1, c3 * sig) + 3 * shift + nu

    # moving a point by r changes the face expression <w, n> + g |w|_1 by at most r (1 + sqrt3 g); the kernels store n and g
    # divided by (1 + kSpreadFactor g), which turns every comparison with a threshold T into one with T (1 + kSpreadFactor g):
    # the thresholds below need to cover the positional error itself, whatever the slope
    need("kSpreadFactor is at least sqrt(3) (|w|_1 <= sqrt(3) |w|_2)", SQ3, m["kSpreadFactor"], "pyr_face: n, g scaled by 1 / (1 + kSpreadFactor g)")
    spread = 1.0
    for sig in (SIGMA_MIN, 1.5, 2.0, 4.0, 16.0, 1e3):
        need(f"side test, computed points (sigma = {sig:.4g}): kSideEpsFactor * kappa * sigma covers the positional coefficient",
             c_pos(sig), m["kSideEpsFactor"] * kappa * sig * m["kSigmaRoundUp"],
             "prim_stage1 / box_pruned / beam_child: eps = u sigma D / h, D / h >= 1 / cos(incidence)")
        need(f"face threshold (sigma = {sig:.4g}): kFaceEpsFactor * kappa * sigma covers (1 + sqrt3 g) x the positional coefficient",
             spread * c_pos(sig), m["kFaceEpsFactor"] * kappa * sig * m["kSigmaRoundUp"],
             "prim_stage1: base = -(kFaceEpsFactor eps_c + kFaceUnits u)")
    need("side test, computed points: kSideUnits * kappa covers the plane distance of the point + the sign uncertainty + the "
         "kernel's evaluation", nu + side + k_eval, m["kSideUnits"] * kappa, "the part of the margin that incidence does not amplify")
    need("face threshold: kFaceUnits * kappa covers (1 + sqrt3 g) x the plane distance of the point + the kernel's evaluation",
         spread * nu + k_eval, m["kFaceUnits"] * kappa, "")

    # ---- lateral tolerance (slopes of the faces): delta = sum_l kappa (kLateralSigma sigma_l + kLateralConst) (x mag_scale) ----
    # pyramid over the prefix's LAST mirror: Moller-Trumbore's uncertainty, its input shifts, and the lateral distance between
    # the tested ray's crossing point and the outgoing line (2 nu sin(phi) + lateral + shift) -- against that mirror's own share
    a, b = m["kLateralSigma"], m["kLateralConst"]
    for sig in (SIGMA_MIN, 1.5, 2.0, 4.0, 16.0, 1e3):
        first = max(c1, c3 * sig) + 3 * shift + (2 * nu + lam + shift)
        need(f"lateral tolerance, last mirror (sigma = {sig:.4g})", first, kappa * (a * sig + b), "build_ctx: delta = sum u (kLateralSigma sigma + kLateralConst)")
    # every unfolding through a later mirror l adds: the reflection of the (nearly in-plane) reflection point (2 nu, stretched by
    # the map's |n|^2), the lateral error of the next step, the rounding of the apex's image and of the unfolded vertices --
    # nothing that grows with a shape factor: it goes against the smallest share a later mirror can bring
    stretch = 1.0 + 8 * EPS
    unfold1 = 2 * nu * stretch + lam + d["image_apex_val"] + d["image_vertex1_val"]
    unfold2 = 2 * nu * stretch + lam + d["image_apex_val"] + d["image_vertex2_val"]
    need("lateral tolerance, one unfolding: the later mirror's own share of delta covers it", unfold1, kappa * (a * SIGMA_MIN + b), "")
    need("lateral tolerance, second unfolding (order 3)", unfold2, kappa * (a * SIGMA_MIN + b), "")

    # ---- the kernels' own approximations ----
    # v_rcp_f32 / v_sqrt_f32: 1 ulp = 2 eps relative each
    need("kRhoRoundDown covers rho = len * rcp(el) with v_sqrt (x2), v_rcp, two products and the cross product's rounding",
         1.0 - (1.0 - 2 * EPS) ** 3 * (1.0 - EPS) ** 2 * (1.0 - 8 * EPS), 1.0 - m["kRhoRoundDown"], "pyr_face")
    need("kSlopeFactor covers delta * rcp(rho - delta) (v_rcp, product, difference)", (1 + 2 * EPS) * (1 + EPS) ** 2 * (1 + 2 * EPS),
         m["kSlopeFactor"], "pyr_face")
    need("kSlopeRounding covers the face normal's normalisation (v_sqrt, v_rcp, product: 5 eps), fdot's three roundings and |w|_1",
         (5 + 3 + 2) * EPS * 2, m["kSlopeRounding"], "pyr_face / pyramids_separate")
    need("kEpsRoundUp covers u sigma (D rcp(h)): v_rcp and three products", (1 + 2 * EPS) * (1 + EPS) ** 3, m["kEpsRoundUp"], "beam_eps")
    need("kLenRoundUp covers v_sqrt_f32 of a sum of three FMAs", (1 + 2 * EPS) * (1 + 2 * EPS), m["kLenRoundUp"] * (1 + 0.0), "margin_len")
    need("kBoxHalfExtent covers the rounding of (hi - lo) * 0.5", 0.5 * (1 + 4 * EPS), m["kBoxHalfExtent"], "box_pruned")
    need("kBoxExtraUnits * kappa covers what kBoxHalfExtent cannot -- the rounding of the box's CENTRE (half an ulp(M) per coordinate, "
         "whatever the extent) -- and the rounding of its support against a point's own value", SQ3 * hu(1.0) + k_eval + 3 * hu(1.0),
         m["kBoxExtraUnits"] * kappa, "box_pruned: thresholds (kFaceUnits + kBoxExtraUnits) u, (kSideUnits + kBoxExtraUnits) u")
    need("kSigmaRoundUp covers pm / len (two square roots, products, one division)", (1 + EPS) ** 6, m["kSigmaRoundUp"], "mesh_prepare_kernel")

    # ---- child filter of the last expansion ----
    # It builds the child's narrowest pyramid from the SAME float values as the receiver stage (image_of_vertex of the parent's
    # apex and of the parent's unfolded first mirror, then make_pyr): identical normals and edge distances, so "what the filter
    # drops, the receiver stage drops" is monotonicity in the constants plus the rounding of the box's support function.
    need("child filter: rho rounded down further than the receiver stage's", m["kChildRhoRoundDown"], m["kRhoRoundDown"] * (1 - 1e-4), "")
    need("child filter: faces switch off earlier", m["kFaceOffRatio"] * 1.005, m["kChildFaceOffRatio"], "")
    need("child filter: whole pyramid switches off earlier", m["kPlaneOffRatio"] * 1.005, m["kChildPlaneOffRatio"], "")
    need("child filter: larger relative slope allowance", m["kSlopeRounding"] * 10, m["kChildSlopeRounding"], "")
    need("child filter: its lateral tolerance is rounded up past the receiver stage's (sum of <= 3 products in another order: 4 eps)",
         1.0 + 8 * EPS, m["kChildDeltaRoundUp"], "child_misses_receivers: delta = u (kLateralSigma (sig_parent + sig_c) + kLateralConst mirrors) kChildDeltaRoundUp")
    need("child filter: the extra threshold covers the rounding of the box's centre / half extents / support against a receiver's own value",
         k_eval + 3 * hu(1.0), (m["kChildFaceUnits"] - m["kFaceUnits"]) * kappa, "child_misses_receivers: thr = -kChildFaceUnits u")
    return {"ok": all(c["ok"] for c in checks), "checks": checks, "kappa": kappa, "sigma_min": SIGMA_MIN,
            "derived": {"lateral_u0": lam, "plane_u0": nu, "side_sign_u0": side, "mt_first_u0": c1, "mt_third_per_sigma_u0": c3,
                        "image_apex_u0": d["image_apex_val"], "image_vertex_unfolded_once_u0": d["image_vertex1_val"],
                        "image_vertex_unfolded_twice_step_u0": d["image_vertex2_val"], "mt_input_shift_u0": shift,
                        }}


# ---------------------------------------------------------------------------------------------------------------
# measurement: the float32 oracle next to float64 on adversarial inputs; measured <= derived
# ---------------------------------------------------------------------------------------------------------------
def measure(n: int = 200_000, seed: int = 7) -> dict:
    import numpy as np

    sys.path.insert(0, str(ROOT))
    import oracle as orc

    rng = np.random.default_rng(seed)
    Mtop = 2.0 ** 10                       # binade [512, 1024): u0 = 2^-14
    u0 = float(np.spacing(np.float32(Mtop * 0.75)))
    f32 = np.float32

    def unit(v):
        return v / np.linalg.norm(v, axis=-1, keepdims=True)

    def top(shape, scale=1.0):
        """coordinates crowded towards +-scale * Mtop (the top of the binade) with random signs"""
        return (1.0 - rng.random(shape) ** 4 * 0.5) * scale * Mtop * rng.choice([-1.0, 1.0], shape) * (1 - 2.0 ** -20)

    out = {}
    # --- image of an apex: |x| < 2 M, result < 2 M ---
    nrm = unit(rng.normal(size=(n, 3))).astype(f32)
    p = top((n, 3)).astype(f32)
    x = top((n, 3), 2.0).astype(f32)
    got = orc.image_of_vertex_with_respect_to_mirror(x, p, nrm).astype(np.float64)
    x64, p64, n64 = (a.astype(np.float64) for a in (x, p, nrm))
    ex = x64 - 2.0 * ((x64 - p64) * n64).sum(-1, keepdims=True) * n64
    keep = np.abs(ex).max(-1) < 2 * Mtop
    out["image_apex_u0"] = float(np.linalg.norm(got - ex, axis=-1)[keep].max() / u0)
    # --- ray / plane step at incidences up to 89.99 degrees ---
    lat_max = pl_max = 0.0
    for deg in (0.0, 45.0, 80.0, 89.0, 89.9, 89.99):
        nn = unit(rng.normal(size=(n, 3)))
        t1 = unit(np.cross(nn, rng.normal(size=(n, 3))))
        phi = np.deg2rad(deg)
        dirv = np.cos(phi) * nn + np.sin(phi) * t1
        xx = top((n, 3), 0.9)
        a, b = rng.uniform(1, 0.5 * Mtop, (n, 1)), rng.uniform(1, 1.5 * Mtop, (n, 1))
        o, img = xx - dirv * a, xx + dirv * b
        k = (np.abs(o).max(-1) < Mtop) & (np.abs(img).max(-1) < 2 * Mtop)
        o32, i32, p32, n32 = (v[k].astype(f32) for v in (o, img, xx, nn))
        d32 = (i32 - o32).astype(f32)
        got = orc.intersection_of_ray_with_plane(o32, d32, p32, n32).astype(np.float64)
        o64, i64, p64, n64 = (v.astype(np.float64) for v in (o32, i32, p32, n32))
        dd = i64 - o64
        dh = unit(dd)
        e = got - o64
        lat = np.linalg.norm(e - dh * (e * dh).sum(-1, keepdims=True), axis=-1)
        pld = np.abs(((got - p64) * n64).sum(-1)) / np.linalg.norm(n64, axis=-1)
        fin = np.isfinite(lat) & np.isfinite(pld)
        lat_max, pl_max = max(lat_max, float(lat[fin].max() / u0)), max(pl_max, float(pld[fin].max() / u0))
    out["lateral_u0"], out["plane_u0"] = lat_max, pl_max
    # --- same-side sign: smallest exact distance at which the float32 sign is wrong ---
    nn = unit(rng.normal(size=(n, 3))).astype(f32)
    p = top((n, 3)).astype(f32)
    tang = unit(np.cross(nn.astype(np.float64), rng.normal(size=(n, 3))))
    xx = (p.astype(np.float64) + tang * rng.uniform(0, 1.5 * Mtop, (n, 1)) + nn.astype(np.float64) * rng.normal(size=(n, 1)) * 4 * u0)
    xx = np.clip(xx, -Mtop * (1 - 2.0 ** -20), Mtop * (1 - 2.0 ** -20)).astype(f32)
    inc = (xx - p).astype(f32)
    s32 = ((inc[:, 0] * nn[:, 0] + inc[:, 1] * nn[:, 1]).astype(f32) + inc[:, 2] * nn[:, 2]).astype(f32)
    ex = ((xx.astype(np.float64) - p.astype(np.float64)) * nn.astype(np.float64)).sum(-1)
    wrong = np.sign(s32.astype(np.float64)) != np.sign(ex)
    out["side_sign_u0"] = float(np.abs(ex[wrong]).max() / u0) if wrong.any() else 0.0
    # --- Moller-Trumbore: rays from an apex at distance rho from an edge line, passing at a known angle outside the face ---
    worst1 = worst3 = 0.0
    for _ in range(4):
        v0 = top((n, 3), 0.5)
        e1 = unit(rng.normal(size=(n, 3))) * rng.uniform(1, 0.9 * Mtop, (n, 1))
        ang = rng.uniform(np.deg2rad(3), np.deg2rad(120), (n, 1))
        e2d = unit(np.cross(np.cross(e1, rng.normal(size=(n, 3))), e1))
        e2 = (np.cos(ang) * unit(e1) + np.sin(ang) * e2d) * rng.uniform(1, 0.9 * Mtop, (n, 1))
        tv = np.stack((v0, v0 + e1, v0 + e2), axis=1)
        ok = np.abs(tv).reshape(n, -1).max(-1) < Mtop
        tv32 = tv[ok].astype(f32)
        m = tv32.shape[0]
        t64 = tv32.astype(np.float64)
        a64, b64, c64 = t64[:, 0], t64[:, 1], t64[:, 2]
        nrm = unit(np.cross(b64 - a64, c64 - a64))
        la, lb, lc = np.linalg.norm(b64 - a64, axis=-1), np.linalg.norm(c64 - b64, axis=-1), np.linalg.norm(a64 - c64, axis=-1)
        area2 = np.linalg.norm(np.cross(b64 - a64, c64 - a64), axis=-1)
        sig = np.maximum(np.maximum(la * lb, lb * lc), lc * la) / area2
        for edge, (pa, pb) in enumerate(((a64, c64), (a64, b64), (b64, c64))):  # u >= 0: edge v0-v2; v >= 0: edge v0-v1; u + v <= 1: v1-v2
            ed = unit(pb - pa)
            inward = np.cross(nrm, ed)
            third = (b64, c64, a64)[edge]
            inward *= np.sign(((third - pa) * inward).sum(-1, keepdims=True))
            # target point in the plane just outside / inside the edge, origin off the plane
            along = rng.uniform(0.05, 0.95, (m, 1))
            q = rng.normal(size=(m, 1)) * 6 * u0 * (sig[:, None] if edge == 2 else 1.0)
            tgt = pa + (pb - pa) * along - inward * q  # q > 0: outside by q
            phi = np.deg2rad(rng.choice([0, 40, 80, 88, 89.5], (m, 1)))
            tdir = unit(np.cross(nrm, rng.normal(size=(m, 3))))
            back = np.cos(phi) * nrm + np.sin(phi) * tdir
            o = tgt + back * rng.uniform(1, 0.8 * Mtop, (m, 1))
            okk = np.abs(o).max(-1) < Mtop
            o32 = o[okk].astype(f32)
            dvec = ((tgt[okk] - o[okk]) * rng.uniform(1.0, 1.5, (int(okk.sum()), 1))).astype(f32)
            t, hit = orc.ray_intersect_triangle(o32, dvec, tv32[okk])
            # exact verdict for the FLOAT inputs
            o64, d64 = o32.astype(np.float64), dvec.astype(np.float64)
            A, B, C = a64[okk], b64[okk], c64[okk]
            N = np.cross(B - A, C - A)
            tt = ((A - o64) * N).sum(-1) / (d64 * N).sum(-1)
            X = o64 + d64 * tt[:, None]
            E1, E2 = B - A, C - A
            det = np.linalg.det(np.stack((E1, E2, np.cross(E1, E2)), -1))
            w = X - A
            uu = np.linalg.det(np.stack((w, E2, np.cross(E1, E2)), -1)) / det
            vv = np.linalg.det(np.stack((E1, w, np.cross(E1, E2)), -1)) / det
            inside = (uu >= 0) & (vv >= 0) & (uu + vv <= 1) & (tt > 1.2e-6)
            wrong = (hit.astype(bool) != inside)
            if wrong.any():
                # in-plane distance from the nearest edge line x cos(incidence), in u0 (the quantity the derivation bounds)
                cosi = np.abs((unit(d64) * unit(N)).sum(-1))
                dists = []
                for (qa, qb) in ((A, C), (A, B), (B, C)):
                    e = unit(qb - qa)
                    r = X - qa
                    dists.append(np.linalg.norm(r - e * (r * e).sum(-1, keepdims=True), axis=-1))
                dmin = np.minimum(np.minimum(dists[0], dists[1]), dists[2])
                val = (dmin * cosi)[wrong] / u0
                if edge == 2:
                    worst3 = max(worst3, float((val / sig[okk][wrong]).max()))
                else:
                    worst1 = max(worst1, float(val.max()))
    out["mt_first_u0"], out["mt_third_per_sigma_u0"] = worst1, worst3
    return out


def main(argv):
    res = check()
    if "--no-measure" not in argv:
        meas = measure(60_000)
        res["measured"] = meas
        dv = res["derived"]
        pairs = (("image_apex_u0", "image_apex_u0"), ("lateral_u0", "lateral_u0"), ("plane_u0", "plane_u0"),
                 ("side_sign_u0", "side_sign_u0"), ("mt_first_u0", "mt_first_u0"), ("mt_third_per_sigma_u0", "mt_third_per_sigma_u0"))
        res["measured_below_derived"] = {a: bool(meas[a] <= dv[b]) for a, b in pairs}
        res["ok"] = res["ok"] and all(res["measured_below_derived"].values())
    if "--json" in argv:
        print(json.dumps(res, indent=1))
    else:
        print(f"kappa = {res['kappa']:g}, sigma_min = {res['sigma_min']:.4f}")
        for k, v in res["derived"].items():
            mv = res.get("measured", {}).get(k)
            print(f"  derived {k:45s} {v:10.4g}" + (f"   measured {mv:8.3g}" if mv is not None else ""))
        for c in res["checks"]:
            print(f"  [{'ok' if c['ok'] else 'FAIL'}] {c['lhs']:10.4g} <= {c['rhs']:10.4g}   {c['name']}")
        print("ALL CHECKS PASS" if res["ok"] else "CHECK FAILED")
    return 0 if res["ok"] else 1


if __name__ == "__main__":
    raise SystemExit(main(sys.argv[1:]))
