// beam_margins.hpp -- every constant the conservative ("beam") pruning's margins are made of, in ONE place.
//
// Read by two parties:
//   * the kernels of csrc/beam.hip (and the shape factor of csrc/mesh.hip), which use nothing but these names;
//   * oracle/studies/beam_bounds_check.py, which parses the `constexpr float NAME = VALUE;` lines below, derives worst-case
//     bounds of the REFERENCE's float32 arithmetic by interval arithmetic over its operation sequence
//     (differt/src/differt/geometry/_solver_image_method.py:68-79, 116-135, 152-203, 443-454 and _utils.py:1263-1322,
//     as the repository's CPU restatement orders them) and checks every inequality the pruning argument needs
//     between those bounds and these constants.  tests/test_beam_margins.py runs it on the CPU: a constant changed here
//     without the check passing fails the suite.
//
// Syntax contract for the parser: one constant per line, `constexpr float kName = <float literal>f;` -- no expressions.
#pragma once

namespace drt {
namespace margins {

// ---- the error unit -------------------------------------------------------------------------------------------------
// u = kKappaDefault * ulp(M) * mag_scale, M = largest coordinate magnitude of mesh, transmitters and receivers.
constexpr float kKappaDefault = 64.0f;
// lateral tolerance of a prefix: delta = sum over its mirrors of u * (kLateralSigma * sigma_l + kLateralConst) -- a part that
// grows with the mirror's shape factor (Moller-Trumbore's third-edge uncertainty) and a part that does not (plane distance and
// lateral error of the reflection points, rounding of the images: what every unfolding adds)
constexpr float kLateralSigma = 1.0f;
constexpr float kLateralConst = 3.0f;
// shape factor sigma = largest 1 / sin(corner) of a triangle, rounded up by this factor and clamped to >= 1 (mesh.hip)
constexpr float kSigmaRoundUp = 1.0001f;

// ---- positional bound eps = u sigma D / h ---------------------------------------------------------------------------
constexpr float kEpsRoundUp = 1.0001f;        // covers v_rcp_f32 (1 ulp) and the three products
constexpr float kLenRoundUp = 1.000001f;      // v_sqrt_f32 (1 ulp) nudged up: |w| rounded up
constexpr float kBoxFarRoundUp = 1.0001f;     // farthest box corner from the apex, rounded up
constexpr float kBoxHalfExtent = 0.50001f;    // half extents of a box, rounded up
// a box test (cluster of primitives, cluster of receivers) uses thresholds this many units wider than the point test it stands
// for: covers the rounding of the box's centre (half an ulp(M), whatever the box's size) and of its support function
constexpr float kBoxExtraUnits = 0.5f;

// ---- side tests (which side of a mirror plane a point set lies on) ---------------------------------------------------
// exact points (transmitter, receivers): margin kSideUnits * u; computed points add their positional bound
constexpr float kSideUnits = 2.0f;
constexpr float kSideEpsFactor = 2.0f;     // computed points: their positional bound eps counts this many times

// ---- pyramid faces --------------------------------------------------------------------------------------------------
// vertex x of candidate c is outside face f when <x - I, n_f> + g_f |x - I|_1 < -(kFaceEpsFactor eps_c + kFaceUnits u)
constexpr float kFaceEpsFactor = 2.0f;
constexpr float kFaceUnits = 1.0f;
// slope g_f = kSlopeFactor * delta / (rho_f - delta) + kSlopeRounding, rho_f = kRhoRoundDown * (apex to edge line)
constexpr float kSlopeFactor = 1.0101f;
constexpr float kSlopeRounding = 2e-6f;       // rounding of the face normal's normalisation and of <x - I, n_f> (fdot)
constexpr float kRhoRoundDown = 0.9999f;
// normal and slope of a face are stored divided by (1 + kSpreadFactor g_f), kSpreadFactor >= sqrt(3): a point moved by r moves the
// face expression by at most r (1 + sqrt(3) g_f), so thresholds that cover a positional error hold for steep faces too
constexpr float kSpreadFactor = 1.7321f;
// a face is OFF while rho_f <= kFaceOffRatio * delta; a whole pyramid while its apex lies within kPlaneOffRatio * delta of
// the polygon's plane (distance rounded down by kRhoRoundDown)
constexpr float kFaceOffRatio = 1.05f;
constexpr float kPlaneOffRatio = 1.05f;      // the whole pyramid: apex within this many delta of the polygon's plane

// ---- child filter of the last expansion ----------------------------------------------------------------------------
// The filter builds the child's narrowest pyramid from THE SAME float values as the receiver stage (same apex, same unfolded
// vertices, same make_pyr), so "what it drops, the receiver stage drops" is monotonicity in these constants:
constexpr float kChildDeltaRoundUp = 1.00002f;  // the two stages sum the shape factors in different orders
constexpr float kChildRhoRoundDown = 0.999f;  // < kRhoRoundDown
constexpr float kChildFaceOffRatio = 1.06f;   // > kFaceOffRatio
constexpr float kChildPlaneOffRatio = 1.06f;  // > kPlaneOffRatio
constexpr float kChildSlopeRounding = 2.1e-4f;  // > kSlopeRounding
constexpr float kChildFaceUnits = 1.5f;       // > kFaceUnits + (rounding of the box's support against a receiver's own value) / kappa

// ---- pairing pass: a coplanar pair is a convex fan quad when every corner turns by sin >= this -----------------------
constexpr float kQuadConvexSin = 1e-3f;

}  // namespace margins
}  // namespace drt
