"""Tile selection under the caller's own load.

`Net::autotune` (csrc/net.cpp) times every tile alone — and, for float32, once more inside whole forward passes —, i.e. for
the latency of ONE forward.  A service that keeps several forwards in flight (deepcut_tools.Pipeline, bench.py's `value`) wants
the tiles that maximise throughput under that load, and the two differ: a one-workgroup-per-CU tile that wins alone leaves the
other forwards no room.  `tune_in_flight` is a coordinate descent on the real objective: for the busiest GEMM signatures, in
turn, each tile that was within `margin` of the best when timed alone is put in place on every executor (dc_net_set_tile) and
the caller's workload is run; a tile is kept only if it beats the incumbent by `min_gain`.  Everything stays in this process
(set DC_TUNE_CACHE to persist the result); the reference has nothing to mirror here (one SGEMM per layer).
"""


def tune_in_flight(nets, run, top=10, margin=1.20, min_gain=0.004, reps=3, max_candidates=4, log=None):
    """nets: the executors of ONE model (a net and its clones), all at the shape to tune, each having run a forward.
    run(): enqueue the representative load on the executors, synchronise, return the wall seconds.
    Returns {"before": s, "after": s, "changed": [(signature, old tile, new tile, seconds before, seconds after)], "runs": n}."""
    report = nets[0].tune_report()
    ranked = []
    for sig in report:
        if len(sig["timed"]) < 2:
            continue  # the choice came from a cache file (no timings) or there is nothing to choose from
        alone = dict(sig["timed"])
        ranked.append((alone.get(sig["tile"], sig["timed"][0][1]) * sig["launches"], sig))
    ranked.sort(key=lambda t: -t[0])
    runs = [0]

    def measure():
        run()  # re-captures the graphs a tile change dropped; not timed
        runs[0] += reps + 1
        return min(run() for _ in range(reps))

    before = incumbent = measure()
    changed = []
    for _share, sig in ranked[:top]:
        best_alone = sig["timed"][0][1]
        cur = sig["tile"]
        best = (incumbent, cur)
        for tile, us in sig["timed"][:max_candidates + 1]:
            if tile == cur or us > margin * best_alone:
                continue
            for n in nets:
                n.set_tile(sig["signature"], tile)
            t = measure()
            if log:
                log("  %-60s %-24s %.3f ms (incumbent %.3f)" % (sig["signature"][:60], tile, t * 1e3, best[0] * 1e3))
            if t < best[0] * (1.0 - min_gain):
                best = (t, tile)
        for n in nets:
            n.set_tile(sig["signature"], best[1])
        if best[1] != cur:
            changed.append((sig["signature"], cur, best[1], incumbent, best[0]))
            incumbent = best[0]
    after = measure() if changed else incumbent
    return {"before": before, "after": after, "changed": changed, "runs": runs[0]}
