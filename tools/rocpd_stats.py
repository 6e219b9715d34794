"""Per-kernel totals from a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats -d DIR -o NAME writes NAME_results.db):
    python tools/rocpd_stats.py gpurun_out/prof_train/t_results.db [steps] [top]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"kernel time {tot / 1e6:.2f} ms total, {tot / 1e6 / steps:.3f} ms per step ({steps} steps), {sum(r[1] for r in rows) / steps:.0f} launches per step")
for r in rows[:top]:
    print(f"{r[0][:86]:86s} {r[1] / steps:7.1f}/step {r[2] / 1e6 / steps:8.3f} ms/step {r[3] / 1e3:8.1f} us {100 * r[2] / tot:5.1f}%")
