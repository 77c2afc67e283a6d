"""CPU: the coordinate descent of deepcut_tools.tune_in_flight on fake executors — which signatures it visits, which tiles it
tries, when it keeps one (the GPU side, dc_net_tune_report / dc_net_set_tile, is tests/test_gpu_tuning.py)."""
from deepcut_tools import tune_in_flight


class FakeNet(object):
    def __init__(self, report, log):
        self.report = [dict(r, timed=list(r["timed"])) for r in report]
        self.log = log

    def tune_report(self):
        return [dict(r) for r in self.report]

    def set_tile(self, signature, tile):
        for r in self.report:
            if r["signature"] == signature:
                assert tile in [t for t, _ in r["timed"]]
                r["tile"] = tile
                self.log.append((signature, tile))
                return
        raise KeyError(signature)


REPORT = [
    {"signature": "busy", "tile": "a", "launches": 36, "timed": [("a", 10.0), ("b", 10.5), ("c", 11.9), ("d", 30.0)]},
    {"signature": "small", "tile": "a", "launches": 1, "timed": [("a", 5.0), ("b", 5.1)]},
    {"signature": "cached", "tile": "z", "launches": 50, "timed": []},          # from a cache file: nothing to walk
    {"signature": "single", "tile": "a", "launches": 40, "timed": [("a", 9.0)]},  # one candidate only
    {"signature": "mid", "tile": "b", "launches": 10, "timed": [("a", 20.0), ("b", 20.2)]},
]
# seconds of the load as a function of the tiles in place: "b" is the in-flight winner for `busy`, "a" (alone-best) for `mid`
COST = {"busy": {"a": 1.00, "b": 0.90, "c": 0.95, "d": 0.50}, "small": {"a": 0.010, "b": 0.0099}, "mid": {"a": 0.30, "b": 0.33}}


def _setup():
    log = []
    nets = [FakeNet(REPORT, log), FakeNet(REPORT, log)]
    calls = [0]

    def load():
        calls[0] += 1
        return sum(COST[r["signature"]][r["tile"]] for r in nets[0].report if r["signature"] in COST) + 1.0

    return nets, load, log, calls


def test_descent_keeps_only_real_gains_on_every_executor():
    nets, load, log, calls = _setup()
    res = tune_in_flight(nets, load, top=10, margin=1.20, min_gain=0.004, reps=2)
    final = {r["signature"]: r["tile"] for r in nets[0].report}
    assert final == {r["signature"]: r["tile"] for r in nets[1].report}  # every executor carries the same choices
    assert final["busy"] == "b"       # 10 % better under load although 5 % slower alone
    assert final["mid"] == "a"        # the alone-best also wins under load
    assert final["small"] == "a"      # 0.004 % of the load: below min_gain, the incumbent stays
    assert final["cached"] == "z" and final["single"] == "a"
    assert ("busy", "d") not in log   # 3x slower alone: outside the margin, never tried (although it would have won)
    assert [c[0] for c in res["changed"]] == ["busy", "mid"]  # busiest first: 36 x 10 us, then 10 x 20.2 us
    assert abs(res["before"] - (1.0 + 1.00 + 0.010 + 0.33)) < 1e-9 and abs(res["after"] - (1.0 + 0.90 + 0.010 + 0.30)) < 1e-9
    assert res["runs"] == calls[0]


def test_top_limits_the_signatures_walked():
    nets, load, log, _ = _setup()
    res = tune_in_flight(nets, load, top=1, reps=1)
    assert set(s for s, _ in log) == {"busy"} and len(res["changed"]) == 1


def test_the_winograd_sibling_is_tried_even_without_isolated_timings():
    """A choice read from a DC_TUNE_CACHE file carries no timings, and such signatures are left alone — except that the two forms of
    the Winograd kernel (8 / 16 waves per workgroup) are each other's candidate by construction: which one wins IS a question of load."""
    class Wino(FakeNet):
        def set_tile(self, signature, tile):
            for r in self.report:
                if r["signature"] == signature:
                    assert tile in ("wino_f23", "wino_f23_w16")
                    r["tile"] = tile
                    self.log.append((signature, tile))

    log = []
    rep = [{"signature": "res4 3x3+w", "tile": "wino_f23_w16", "launches": 36, "timed": []},
           {"signature": "cached", "tile": "z", "launches": 50, "timed": []}]
    nets = [Wino(rep, log), Wino(rep, log)]
    res = tune_in_flight(nets, lambda: 1.0 if nets[0].report[0]["tile"] == "wino_f23_w16" else 0.95, reps=1)
    assert [c[:3] for c in res["changed"]] == [("res4 3x3+w", "wino_f23_w16", "wino_f23")] and res["skipped"] == 1
    assert all(n.report[0]["tile"] == "wino_f23" and n.report[1]["tile"] == "z" for n in nets)


def test_nothing_to_tune_is_not_an_error():
    log = []
    nets = [FakeNet([REPORT[2], REPORT[3]], log)]
    res = tune_in_flight(nets, lambda: 1.0)
    assert res["changed"] == [] and log == [] and res["before"] == res["after"] == 1.0


def test_a_failing_load_leaves_every_executor_on_the_incumbent_tile():
    """an exception from run() or set_tile must not leave the executors on mixed trial tiles."""
    import pytest

    nets, load, log, calls = _setup()

    def flaky():
        if calls[0] == 9:  # somewhere inside the trials of the first signature
            calls[0] += 1
            raise RuntimeError("device lost")
        return load()

    with pytest.raises(RuntimeError):
        tune_in_flight(nets, flaky, reps=2)
    a = {r["signature"]: r["tile"] for r in nets[0].report}
    assert a == {r["signature"]: r["tile"] for r in nets[1].report}
    assert a["busy"] in ("a", "b")  # the incumbent, or a tile that had been accepted before the failure — never a trial tile


def test_a_signature_an_executor_does_not_have_is_skipped_and_counted():
    log = []
    nets = [FakeNet(REPORT, log), FakeNet([r for r in REPORT if r["signature"] != "busy"], log)]

    def load():
        return 1.0 + sum(COST[r["signature"]][r["tile"]] for r in nets[0].report if r["signature"] in COST)

    res = tune_in_flight(nets, load, reps=1)
    assert ("busy", "b") not in log and res["skipped"] == 3  # cached, single, and busy (missing on the second executor)


def test_a_gain_that_does_not_reproduce_is_not_kept():
    nets, _, log, _ = _setup()
    seq = iter([2.0, 2.0, 1.9, 1.9] + [2.0] * 50)  # the first candidate looks 5 % faster once, then never again

    res = tune_in_flight(nets, lambda: next(seq), top=1, reps=1, min_gain=0.01)
    assert res["changed"] == [] and {r["signature"]: r["tile"] for r in nets[0].report}["busy"] == "a"
