"""Summarise a rocprofv3 --pmc rocpd database: per-kernel mean of every collected counter (dev tool)."""
import glob
import sqlite3
import sys


def main(pattern, name_filter="cpi_"):
    for db in sorted(glob.glob(pattern, recursive=True)):
        con = sqlite3.connect(db)
        cur = con.cursor()
        try:
            rows = cur.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events "
                               "group by name, counter_name").fetchall()
        except Exception as ex:
            cols = [d[0] for d in cur.execute("select * from pmc_events limit 1").description]
            print(db, "schema:", cols, ex)
            continue
        for name, ctr, val, n in rows:
            if name_filter in name:
                short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                print("%-28s %-44s %-24s avg=%.6g n=%d" % (db.split("/")[-1][:28], short, ctr, val, n))


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]))
