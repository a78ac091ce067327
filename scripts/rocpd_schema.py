import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
for name, typ in cur.execute("select name, type from sqlite_master where type in ('table','view') order by name").fetchall():
    if any(k in name.lower() for k in ("pmc", "counter", "kernel")):
        cols = [r[1] for r in cur.execute(f"pragma table_info('{name}')").fetchall()]
        try:
            n = cur.execute(f"select count(*) from '{name}'").fetchone()[0]
        except Exception as e:
            n = str(e)
        print(typ, name, n, cols)
        try:
            for row in cur.execute(f"select * from '{name}' limit 2").fetchall():
                print("    ", row)
        except Exception as e:
            print("    err", e)
