"""Where does a multi-rank step on ONE shared GPU spend its time?  Reads the Chrome traces that
``bench.py --trace-dir DIR`` (or any entry script's ``--trace-dir``) wrote per rank and reports, over the
captured window: GPU-busy time of every rank (union of its kernel intervals), the union over all ranks, and
how much of a rank's idle time is covered by the OTHER ranks' kernels (= time-sharing, not communication).

usage: python bench/trace_overlap.py DIR
"""
import glob
import json
import os
import sys


def intervals(path):
    with open(path) as f:
        ev = json.load(f)["traceEvents"]
    ks = [(e["ts"], e["ts"] + e["dur"]) for e in ev if e.get("ph") == "X" and e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset")]
    host = [(e["name"], e["ts"], e["dur"]) for e in ev if e.get("ph") == "X" and e.get("cat") == "user_annotation"]
    return sorted(ks), host


def union(iv):
    out, cur_s, cur_e = 0.0, None, None
    for s, e in sorted(iv):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                out += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        out += cur_e - cur_s
    return out


def main(d):
    files = sorted(glob.glob(os.path.join(d, "trace_rank*.json")))
    per = {}
    for f in files:
        r = int(os.path.basename(f)[len("trace_rank"):-5])
        per[r] = intervals(f)
    lo = max(min(s for s, _ in ks) for ks, _ in per.values())
    hi = min(max(e for _, e in ks) for ks, _ in per.values())
    clip = lambda iv: [(max(s, lo), min(e, hi)) for s, e in iv if e > lo and s < hi]
    span = hi - lo
    print(f"window common to all ranks: {span / 1e3:.2f} ms, {len(per)} ranks")
    allk = []
    for r, (ks, host) in sorted(per.items()):
        c = clip(ks)
        allk += c
        steps = [h for h in host if h[0].startswith("Forward 0") and lo <= h[1] <= hi]
        print(f"  rank {r}: {len(c)} GPU intervals, busy {union(c) / 1e3:.2f} ms = {100 * union(c) / span:.1f} % of the window;"
              f" {len(steps)} steps in the window")
    u = union(allk)
    print(f"  all ranks: GPU busy {u / 1e3:.2f} ms = {100 * u / span:.1f} % of the window (sum of ranks "
          f"{sum(union(clip(ks)) for ks, _ in per.values()) / 1e3:.2f} ms)")
    for r, (ks, _) in sorted(per.items()):
        mine = union(clip(ks))
        others = [iv for q, (k2, _) in per.items() if q != r for iv in clip(k2)]
        both = union(clip(ks) + others)
        # idle time of r covered by the others = union(all) - mine
        print(f"  rank {r}: idle {(span - mine) / 1e3:.2f} ms, of which the other ranks' kernels cover {(both - mine) / 1e3:.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
