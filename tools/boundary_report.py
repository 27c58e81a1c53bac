"""The dependent-launch boundary made visible: under hipGraph replay rocprofv3's begin timestamp of a dispatch is
(about) the end of its predecessor, so a replayed step lists no idle time and every duration CONTAINS its boundary.  An EAGER trace
of the same step has the pure kernel durations (the host runs ahead; the gaps it lists are queue gaps, not host stalls, wherever the
host is ahead).  Joining the two per kernel name gives, per family, what a launch costs beyond its kernel.
Usage: python tools/boundary_report.py <replay timeline.txt> <eager timeline.txt> <out.txt>"""
import re
import sys


def per_kernel(path):
    rows, take = {}, False
    head = open(path).readline().strip()
    for ln in open(path):
        if ln.startswith(" calls") or ln.startswith("calls"):
            take = True
            continue
        if take:
            if not ln.strip():
                break
            if ln.startswith("---"):
                continue
            m = re.match(r"\s*(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+(.*)", ln)
            if m:
                rows[m.group(5).strip()[:70]] = (int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4)))
    return head, rows


def main():
    h1, rep = per_kernel(sys.argv[1])
    h2, eag = per_kernel(sys.argv[2])
    L = ["replay: " + h1, "eager : " + h2, "",
         "per kernel: average duration listed under replay (contains the boundary) - average pure duration in the eager trace = boundary share;",
         "the eager trace's own idle gap in front of the kernel beside it (host-issue bound where large)", "",
         f"{'calls':>5} {'replay_us':>9} {'eager_us':>8} {'diff_us':>8} {'eager_gap_us':>12} {'diff x calls (ms)':>17}  kernel", "-" * 120]
    tot = tot_calls = 0.0
    for name, (n, us, avg, _) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
        e = eag.get(name)
        if e is None or e[0] != n:
            continue
        d = avg - e[2]
        tot += d * n
        tot_calls += n
        L.append(f"{n:5d} {avg:9.2f} {e[2]:8.2f} {d:8.2f} {e[3] / n:12.2f} {d * n * 1e-3:17.3f}  {name}")
    L += ["", f"matched launches: {int(tot_calls)}; sum of (replay - eager) durations: {tot * 1e-3:.2f} ms = {tot / max(tot_calls, 1):.2f} us per launch"]
    open(sys.argv[3], "w").write("\n".join(L) + "\n")
    print("\n".join(L[:40]))
    print(L[-1])


if __name__ == "__main__":
    main()
