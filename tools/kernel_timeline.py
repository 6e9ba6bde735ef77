#!/usr/bin/env python3
"""Print the dispatch timeline (start offset, duration, gap) of the last N kernel launches recorded by
`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python3 bench.py ...` (reads DIR/NAME_results.db)."""
import sqlite3
import sys


def main():
    db, last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = c.execute(f"select s.kernel_name, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id "
                     f"order by d.start").fetchall()[-last:]
    t0, prev_end = rows[0][1], rows[0][1]
    for name, s, e in rows:
        print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f} us  {name[:90]}")
        prev_end = e


if __name__ == "__main__":
    main()
