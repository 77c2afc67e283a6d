"""Tile selection under the caller's own load.

`Net::autotune` (csrc/net_tune.cpp) times every tile alone — and once more inside whole forward passes —, i.e. for
the latency of ONE forward.  A service that keeps several forwards in flight (deepcut_tools.Pipeline, bench.py's `value`) wants
the tiles that maximise throughput under that load, and the two differ: a one-workgroup-per-CU tile that wins alone leaves the
other forwards no room.  `tune_in_flight` is a coordinate descent on the real objective: for the busiest GEMM signatures, in
turn, each tile that was within `margin` of the best when timed alone is put in place on every executor (dc_net_set_tile) and
the caller's workload is run; a tile is kept only if it beats the incumbent by `min_gain` AND still does when both are measured
once more (a 1 % difference of the minimum of three wall-clock runs is inside the noise of a 40-ms load).  Whatever happens in
between — an exception from run(), a signature one executor does not have — every executor ends on one and the same tile per
signature: the incumbent's, or the accepted one's.  The choices stay in this process and go to the DC_TUNE_CACHE file when one
is set (dc_net_set_tile rewrites it); the reference has nothing to mirror here (one SGEMM per layer).
"""


# the two forms of the Winograd kernel (8 / 16 waves per workgroup, csrc/kernels.hip): wherever one is in use the other is eligible, and
# which one is faster is exactly a question of load (16 waves win a launch of at most one workgroup per CU running alone, 8 waves win as
# soon as workgroups share CUs) — so the sibling is tried even for a signature whose choice came from a cache file, without timings
_WINO_SIBLING = {"wino_f23": "wino_f23_w16", "wino_f23_w16": "wino_f23"}


def tune_in_flight(nets, run, top=6, margin=1.20, min_gain=0.01, reps=3, max_candidates=2, log=None):
    """nets: the executors of ONE model (a net and its clones), all at the shape to tune, each having run a forward.
    run(): enqueue the representative load on the executors, synchronise, return the wall seconds.
    Returns {"before": s, "after": s, "changed": [(signature, old tile, new tile, seconds before, seconds after)], "runs": n,
             "skipped": signatures left alone because no isolated timings exist for them (tiles read from a DC_TUNE_CACHE file)}."""
    runs = [0]

    def measure():
        run()  # re-captures the graphs a tile change dropped; not timed
        runs[0] += reps + 1
        return min(run() for _ in range(reps))

    before = incumbent = measure()  # first: an executor that has not met the shape yet lowers it here (its report is empty before)
    report = nets[0].tune_report()
    have = [set(s["signature"] for s in n.tune_report()) for n in nets[1:]]
    ranked, skipped = [], 0
    for sig in report:
        if len(sig["timed"]) < 2 and sig["tile"] in _WINO_SIBLING:
            sig = dict(sig, timed=[(sig["tile"], 15.0), (_WINO_SIBLING[sig["tile"]], 15.0)])  # (nominal microseconds: ranking only)
        if len(sig["timed"]) < 2:
            skipped += 1  # the choice came from a cache file (no timings) or there is nothing to choose from
            continue
        if any(sig["signature"] not in h for h in have):
            skipped += 1  # an executor sits at another shape (e.g. a partial batch): its plan has no such launch
            continue
        alone = dict(sig["timed"])
        ranked.append((alone.get(sig["tile"], sig["timed"][0][1]) * sig["launches"], sig))
    ranked.sort(key=lambda t: -t[0])
    def put(signature, tile):
        for n in nets:
            n.set_tile(signature, tile)

    changed = []
    for _share, sig in ranked[:top]:
        best_alone = sig["timed"][0][1]
        cur = sig["tile"]
        keep = cur
        try:
            for tile, us in sig["timed"][:max_candidates + 1]:
                if tile == cur or us > margin * best_alone:
                    continue
                put(sig["signature"], tile)
                t = measure()
                if log:
                    log("  %-60s %-24s %.3f ms (incumbent %.3f)" % (sig["signature"][:60], tile, t * 1e3, incumbent * 1e3))
                if t < incumbent * (1.0 - min_gain):
                    # confirm: the incumbent tile and the candidate once more, back to back
                    put(sig["signature"], keep)
                    t_old = measure()
                    put(sig["signature"], tile)
                    t_new = measure()
                    if log:
                        log("    confirm: incumbent %.3f ms, candidate %.3f ms" % (t_old * 1e3, t_new * 1e3))
                    if t_new < t_old * (1.0 - min_gain):
                        changed.append((sig["signature"], keep, tile, t_old, t_new))
                        keep, incumbent = tile, t_new
                    else:
                        incumbent = min(incumbent, t_old)
        finally:
            put(sig["signature"], keep)  # every executor on the same tile, whatever happened above
    after = measure() if changed else incumbent
    return {"before": before, "after": after, "changed": changed, "runs": runs[0], "skipped": skipped}
