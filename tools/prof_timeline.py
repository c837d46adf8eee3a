"""Development aid: kernel timeline of a rocprofv3 --kernel-trace database: for the last `n` kernels, start offset, duration, gap to the previous kernel."""
import sqlite3, sys, glob
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
for db in glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    rows = rows[-n:]
    t0 = rows[0][1]; prev_end = rows[0][1]
    busy = 0
    for name, st, en in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:46]
        print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((st - t0) / 1e3, (st - prev_end) / 1e3, (en - st) / 1e3, short))
        busy += en - st; prev_end = en
    print("span %.1f us, kernels busy %.1f us" % ((rows[-1][2] - t0) / 1e3, busy / 1e3))
