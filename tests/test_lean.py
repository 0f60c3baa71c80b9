"""Lean passes of the device-resident loop: the same registration with them off, on, and
tried at every chance (so that they fail and are repeated while the cloud still moves)."""
import numpy as np
import pytest
import torch  # noqa: F401  (before libicpx.so is loaded: both bring a HIP runtime, torch's must come first)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def icpx():
    from iterativeclosestpoint_amd import icpx as m
    assert m.device_count() >= 1, "no HIP device visible"
    return m


@pytest.fixture(scope="module")
def synth():
    from iterativeclosestpoint_amd import synth as s
    return s


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_lean_passes_change_nothing(icpx, synth, metric):
    c = synth.config_c3(nu=300, nv=150, npts=120_000)
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"], c["source_normals"] if metric == 2 else None)
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks = metric, 40, len(c["source_xyz"])
    p.max_distance = c["max_dist"]
    out = {}
    for mode in (0, 1, 2, 3, 1):
        ctx.set_lean(mode)
        before = ctx.debug_stats()
        T, it, mean, st = ctx.run(p)
        after = ctx.debug_stats()
        out.setdefault(mode, []).append((T, it, mean, st,
                                         after["lean_passes"] - before["lean_passes"],
                                         after["lean_failures"] - before["lean_failures"]))
    T0, it0, m0, st0, lp0, lf0 = out[0][0]
    assert lp0 == 0 and lf0 == 0 and it0 == 40
    for mode in (1, 2, 3):
        for T, it, mean, st, lp, lf in out[mode]:
            assert np.array_equal(T, T0) and (it, st) == (it0, st0) and mean == m0, (mode, lp, lf)
    # the settled passes of the default mode were lean ones, and none had to be repeated more
    # than now and then (point-to-point still creeps towards its fixed point after 40
    # iterations on this pair: few of its passes are settled ones)
    assert out[1][0][5] <= 3 and (metric == 0 or out[1][0][4] >= 20), out[1]
    # tried at every chance, lean passes run while the cloud still moves
    assert out[2][0][4] > out[1][0][4], (out[1], out[2])
    # ... and every third one made to fail is repeated in full
    assert out[3][0][5] >= 5 and out[3][0][4] > out[3][0][5], out[3]


def test_lean_pass_stops_on_mean_distance(icpx, synth):
    """The mean-distance stop is reached inside a lean pass's solve like any other."""
    c = synth.config_c3(nu=300, nv=150, npts=120_000)
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"])
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks = 1, 60, len(c["source_xyz"])
    p.max_distance = c["max_dist"]
    ctx.set_lean(0)
    p.check_mean_distance, p.max_mean_distance = 0, 0.0
    T, it, mean, st = ctx.run(p)
    # a threshold the registration crosses late, among the settled passes
    p.check_mean_distance, p.max_mean_distance = 1, mean * 1.0000001
    res = []
    for mode in (0, 1):
        ctx.set_lean(mode)
        res.append(ctx.run(p))
    assert res[0][1] == res[1][1] and res[0][3] == res[1][3] == 1
    assert np.array_equal(res[0][0], res[1][0]) and res[0][2] == res[1][2]


def test_lean_passes_with_distance_threshold(icpx, synth):
    """Partial overlap with outliers and a distance threshold: queries with nothing within the
    threshold stand on "far" certificates, which a lean pass has to let through as well."""
    c = synth.config_c5(nu=300, nv=150, npts=150_000)
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"])
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks = 1, 40, len(c["source_xyz"])
    p.max_distance = c["max_dist"]
    res = {}
    for mode in (0, 1, 2, 3):
        ctx.set_lean(mode)
        before = ctx.debug_stats()
        T, it, mean, st = ctx.run(p)
        after = ctx.debug_stats()
        res[mode] = (T, it, mean, st, after["lean_passes"] - before["lean_passes"],
                     after["lean_failures"] - before["lean_failures"])
    for mode in (1, 2, 3):
        assert np.array_equal(res[mode][0], res[0][0]) and res[mode][1:4] == res[0][1:4], (mode, res[mode][4:])
    assert res[2][4] >= 5, res[2][4:]
    assert res[3][5] >= 1, res[3][4:]


def test_lean_failures_at_the_end_of_a_run(icpx, synth):
    """A lean pass that fails among the last passes of a run is found when the run's last records
    are awaited, and repeated: runs of every length end as they do without lean passes."""
    c = synth.config_c3(nu=300, nv=150, npts=120_000)
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"], c["source_normals"])
    p = icpx.default_params()
    p.metric, p.max_landmarks = 2, len(c["source_xyz"])
    p.max_distance = c["max_dist"]
    failures = 0
    for iters in range(6, 16):
        p.max_iterations = iters
        ctx.set_lean(0)
        T0, it0, m0, st0 = ctx.run(p)
        ctx.set_lean(3)
        before = ctx.debug_stats()["lean_failures"]
        T3, it3, m3, st3 = ctx.run(p)
        failures += ctx.debug_stats()["lean_failures"] - before
        assert np.array_equal(T3, T0) and (it3, st3, m3) == (it0, st0, m0) and it0 == iters, iters
    assert failures >= 5
