"""Host logic of the recurrence launch schedule (dss_vae_amd/ops.py::_segments / _schedule): no GPU -- the CU count and
the library's layout answer are stubbed."""
import pytest

from dss_vae_amd import ops


@pytest.fixture
def sched(monkeypatch):
    monkeypatch.setattr(ops, "n_cus", lambda: 256)
    monkeypatch.setattr(ops, "GRU_SEGMENT_MIN", 16)
    monkeypatch.setattr(ops, "GRU_SCHEDULE", True)
    monkeypatch.delenv("DSSVAE_GRU_KSPLIT", raising=False)
    monkeypatch.delenv("DSSVAE_GRU_WAVES", raising=False)
    answers = {"waves": 16, "asked": []}

    def layout(Bs, Hs, rows, backward):
        answers["asked"].append((tuple(Bs), tuple(Hs), tuple(rows), backward))
        return answers["waves"]

    monkeypatch.setattr(ops, "_layout_waves", layout)
    ops._SCHEDULE_CACHE.clear()
    yield answers
    ops._SCHEDULE_CACHE.clear()


def covered(plan, specs):
    """every recurrence runs every one of its steps exactly once, in order, over the launches of the plan"""
    at = [0] * len(specs)
    for launch in plan:
        assert len({j for j, _, _, _ in launch}) == len(launch)
        for j, t0, t1, rows in launch:
            assert t0 == at[j] and t0 < t1 <= specs[j][2] and rows in (0, 64, 128)
            at[j] = t1
    return at == [T for _, _, T in specs]


def test_decoder_layer_of_the_headline_config(sched):
    specs = [(256, 600, 31), (256, 300, 91), (256, 300, 91), (256, 300, 31), (256, 300, 31)]
    plan = ops._schedule(specs)
    assert covered(plan, specs)
    assert sorted(plan[0]) == [(0, 0, 31, 64), (1, 0, 31, 128), (2, 0, 31, 128)]
    assert sorted(plan[1]) == [(1, 31, 91, 64), (2, 31, 91, 64), (3, 0, 31, 128), (4, 0, 31, 128)]
    for launch in plan:   # one 16-wave workgroup per CU
        assert sum(-(-specs[j][0] // r) * -(-specs[j][1] // 16) for j, _, _, r in launch) <= 256
    # both directions of both launches were put to the library before the plan was adopted
    assert sorted(b for _, _, _, b in sched["asked"]) == [0, 0, 1, 1]


def test_plan_falls_back_to_the_time_cuts_when_the_library_disagrees(sched):
    sched["waves"] = 8
    specs = [(256, 600, 31), (256, 300, 91), (256, 300, 91), (256, 300, 31), (256, 300, 31)]
    plan = ops._schedule(specs)
    assert covered(plan, specs)
    assert plan == [[(j, 0, 31, 0) for j in range(5)], [(1, 31, 91, 0), (2, 31, 91, 0)]]


@pytest.mark.parametrize("specs", [
    [(256, 150, 30)] * 4,                                   # one segment: nothing to schedule
    [(256, 300, 91), (256, 300, 91), (256, 300, 31)],        # fits in 64-row cells as it is
    [(70, 30, 40), (70, 50, 12), (70, 16, 40), (70, 30, 13)],
    [(1024, 2048, 31), (1024, 1024, 91)],                    # slices beyond the LDS budget of the 16-wave layouts
    [(64, 600, 31), (64, 300, 91), (64, 300, 91), (64, 300, 31), (64, 300, 31)],
    [(256, 600, 31), (256, 300, 91)],
])
def test_every_plan_covers_every_step_once(sched, specs):
    assert covered(ops._schedule(specs), specs)
    assert covered(ops._schedule(specs), specs)   # (cached)


def test_knobs_that_pin_the_layout_switch_the_search_off(sched, monkeypatch):
    specs = [(256, 600, 31), (256, 300, 91), (256, 300, 91), (256, 300, 31), (256, 300, 31)]
    base = [[(j, 0, 31, 0) for j in range(5)], [(1, 31, 91, 0), (2, 31, 91, 0)]]
    monkeypatch.setenv("DSSVAE_GRU_KSPLIT", "0")
    assert ops._schedule(specs) == base
    monkeypatch.delenv("DSSVAE_GRU_KSPLIT")
    monkeypatch.setattr(ops, "GRU_SCHEDULE", False)
    assert ops._schedule(specs) == base
    monkeypatch.setattr(ops, "GRU_SCHEDULE", True)
    monkeypatch.setattr(ops, "GRU_SEGMENT_MIN", 0)
    assert ops._schedule(specs) == [[(0, 0, 31, 0), (1, 0, 91, 0), (2, 0, 91, 0), (3, 0, 31, 0), (4, 0, 31, 0)]]
