"""Per-kernel PMC averages from a rocprofv3 --pmc run (rocpd sqlite): python tools/rocpd_pmc.py file.db [substr] [--by-grid]
(--by-grid: one entry per (kernel, grid size) -- the same kernel on different problem shapes)"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
args = [a for a in sys.argv[2:] if not a.startswith("--")]
sub = args[0] if args else ""
BY_GRID = "--by-grid" in sys.argv
dcols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
gx = "grid_size_x" if "grid_size_x" in dcols else ("grid_x" if "grid_x" in dcols else None)
q = f"""select s.display_name, d.id, d.end - d.start, i.name, p.value, {"d." + gx if gx else "0"} from rocpd_kernel_dispatch d
       join rocpd_info_kernel_symbol s on d.kernel_id = s.id join rocpd_pmc_event p on p.event_id = d.event_id
       join rocpd_info_pmc i on p.pmc_id = i.id"""
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(dict)
for n, did, dt, cname, val, grid in c.execute(q):
    if sub in n:
        k = n.replace("(anonymous namespace)::", "")[:70] + (f"  grid={grid}" if BY_GRID else "")
        acc[k][cname].append(val)
        dur[k][did] = dt
for k, d in acc.items():
    ds = list(dur[k].values())
    print(f"{k}  launches={len(ds)} avg_ms={sum(ds) / len(ds) / 1e6:.3f}")
    for cn, vals in sorted(d.items()):
        print(f"    {cn:32s} {sum(vals) / len(vals):.4g}")
