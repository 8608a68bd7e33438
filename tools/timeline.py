#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace [--memory-copy-trace] output directory and prints how the launches of a run lie in time:
how many render launches run at any moment, the time no render launch runs, the spans of every kernel kind and of the copies,
and a listing of a few consecutive steps (times relative to the first listed event, in microseconds).

    python tools/timeline.py <dir> [first_fraction last_fraction]      (default: the middle of the run, 0.4 .. 0.6)
"""
import csv, glob, os, sys


def rows(d, pattern):
    out = []
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def short(name):
    n = name.replace("void ", "").replace("vrt::", "")
    return n.split("(")[0][:44]


def main():
    d = sys.argv[1]
    lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.4, 0.6)
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows(d, "*kernel_trace.csv")]
    for r in rows(d, "*memory_copy_trace.csv"):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
    ev.sort()
    render = [e for e in ev if e[2].startswith("k_render")]
    if not render:
        print("no render launches in the trace"); return
    # the timed part of the run: the longest run of render launches without a gap of more than 50 ms between starts
    t_first, t_last = render[len(render) // 10][0], render[-max(len(render) // 10, 1)][1]
    win0, win1 = t_first + (t_last - t_first) * lo, t_first + (t_last - t_first) * hi
    inside = [e for e in ev if e[0] >= win0 and e[1] <= win1]
    rin = [e for e in inside if e[2].startswith("k_render")]
    span = (win1 - win0) / 1e3
    print(f"window {span:.0f} us of the run, {len(rin)} render launches -> one every {span / max(len(rin), 1):.1f} us")
    # render launches running over time
    pts = sorted([(e[0], 1) for e in rin] + [(e[1], -1) for e in rin])
    level, last, hist = 0, win0, {}
    for t, dl in pts:
        hist[level] = hist.get(level, 0) + (t - last)
        level += dl; last = t
    hist[level] = hist.get(level, 0) + (win1 - last)
    tot = sum(hist.values())
    print("render launches running at once: " + ", ".join(f"{k}: {100 * v / tot:.1f} %" for k, v in sorted(hist.items())))
    kinds = {}
    for e in inside:
        k = e[2] if not e[2].startswith("copy") else " ".join(e[2].split()[:2])
        kinds.setdefault(k, []).append((e[1] - e[0]) / 1e3)
    for k, v in sorted(kinds.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {k:46s} n={len(v):4d} mean {sum(v) / len(v):9.1f} us  max {max(v):9.1f}")
    print("listing (start, end, what), us:")
    t0 = inside[0][0] if inside else 0
    for e in inside[:40]:
        print(f"  {(e[0] - t0) / 1e3:9.1f} {(e[1] - t0) / 1e3:9.1f}  {e[2]}")


if __name__ == "__main__":
    main()
