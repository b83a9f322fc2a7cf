"""Per-kernel statistics and the last N dispatches of a rocprofv3 --kernel-trace results database (rocprofv3 writes t_results.db when run without --stats).
usage: python scripts/summarise_kernel_db.py <results.db> [n_last]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name order by 4 desc"))
tot = sum(r[3] for r in rows)
for r in rows[:30]:
    print("%-100s %7d calls %9.1f us avg %5.1f %%" % (r[0][:100], r[1], r[2] / 1e3, 100 * r[3] / tot))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if n:
    ks = list(c.execute("select name, start, end from kernels order by start"))[-n:]
    t0 = ks[0][1]
    print("\nstart_us  dur_us  kernel")
    for k in ks:
        print("%9.1f %8.1f  %s" % ((k[1] - t0) / 1e3, (k[2] - k[1]) / 1e3, k[0][:80]))
