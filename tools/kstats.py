#!/usr/bin/env python3
"""Per-kernel average duration from a rocprofv3 --kernel-trace results database (newest *.db under the given dir)."""
import glob
import os
import sqlite3
import sys

d = sys.argv[1]
dbs = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True), key=os.path.getmtime)
c = sqlite3.connect(dbs[-1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id "
     "group by s.kernel_name order by 3 desc")
for name, n, us in list(c.execute(q))[: int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    print(f"{us:10.1f} us  x{n:<4d} {name[:110]}")
