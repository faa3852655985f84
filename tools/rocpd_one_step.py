import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if "embed_gather_kernel" in r[0]]
a, b = marks[-3], marks[-2]
t0 = rows[a][1]
for n, s, e in rows[a:b+1]:
    print(f"{(s-t0)/1e3:8.1f} {(e-t0)/1e3:8.1f} {(e-s)/1e3:7.1f}  {n[:90]}")
