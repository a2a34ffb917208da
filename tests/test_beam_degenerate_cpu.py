"""tests/beam_degenerate.py (the one class of paths the pruned search does not promise): the classifier
itself, on the CPU."""

import numpy as np

import beam_degenerate as BD


def test_short_segment_mask():
    ulp = BD.ulp_of_scene(np.array([[51800.0, 3.0, -7.0]]), np.array([[1.0, 2.0, 3.0]]))
    assert ulp == float(np.spacing(np.float32(51800.0))) == 0.00390625
    tx, rx = [0.0, 0.0, 10.0], [50.0, 0.0, 10.0]
    far = [[tx, [10.0, 0.0, 0.0], [30.0, 0.0, 0.0], rx]]                       # reflection points 20 m apart
    near = [[tx, [10.0, 0.0, 0.0], [10.0 + 32 * ulp, 0.0, 0.0], rx]]           # 32 ulp(M) apart: below the unit of 64
    edge = [[tx, [10.0, 0.0, 0.0], [10.0 + 65 * ulp, 0.0, 0.0], rx]]
    m = BD.short_segment_mask(np.asarray(far + near + edge), ulp)
    assert m.tolist() == [False, True, False]
    # the end segments (transmitter - first point, last point - receiver) do not count, nor do orders 0 and 1
    assert not BD.short_segment_mask(np.asarray([[tx, [0.0, 0.0, 10.0 - ulp], [30.0, 0.0, 0.0], rx]]), ulp)[0]
    assert BD.short_segment_mask(np.zeros((4, 3, 3)), ulp).tolist() == [False] * 4
    assert BD.short_segment_mask(np.zeros((0, 4, 3)), ulp).shape == (0,)
    # order 3: either inner segment
    o3 = [[tx, [10.0, 0, 0], [20.0, 0, 0], [20.0 + ulp, 0, 0], rx]]
    assert BD.short_segment_mask(np.asarray(o3, float), ulp)[0]
