import sqlite3, sys, glob
for db in glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    for r in rows[:25]:
        print("%-90s n=%6d total %9.3f ms avg %9.2f us max %9.2f us" % (r[0][:90], r[1], r[2], r[3], r[4]))
