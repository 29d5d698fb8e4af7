"""Per (kernel, grid) durations AND the idle gap in front of each launch from a rocprofv3 rocpd database: where a launch-bound
sequence (the decode token step, the denoise loop) loses its time.   python tools/rocpd_gaps.py file.db [min_calls=16]"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\[clone .*\]", "", name)
    name = re.sub(r"^void ", "", name)
    return name.strip()[:70]


def main(path, min_calls=16):
    db = sqlite3.connect(path)
    c = db.cursor()
    dcols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
    scols = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    namecol = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
    gx = "grid_size_x" if "grid_size_x" in dcols else ("grid_x" if "grid_x" in dcols else None)
    sel = f"s.{namecol}, d.start, d.end" + (f", d.{gx}" if gx else ", 0")
    rows = c.execute(f"select {sel} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    agg = defaultdict(lambda: [0, 0, 0, 0])  # calls, dur, gap, gap_calls
    prev_end = None
    for name, st, en, g in rows:
        k = (short(name), g)
        a = agg[k]
        a[0] += 1
        a[1] += en - st
        if prev_end is not None and 0 <= st - prev_end < 200000:   # ignore host-side pauses (> 0.2 ms)
            a[2] += st - prev_end
            a[3] += 1
        prev_end = en
    print(f"{'kernel':70s} {'grid':>9s} {'calls':>7s} {'avg_us':>8s} {'gap_before_us':>13s} {'total_ms':>9s}")
    for (n, g), (calls, dur, gap, gc) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        if calls < min_calls:
            continue
        print(f"{n:70s} {g:9d} {calls:7d} {dur / calls / 1e3:8.2f} {gap / max(gc, 1) / 1e3:13.2f} {(dur + gap) / 1e6:9.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 16)
