"""Summarise rocprofv3 --pmc rocpd databases: per-kernel mean of every collected counter (dev tool).

    python tools/pmc_summary.py '/tmp/pmc/**/*.db'                      # text table
    python tools/pmc_summary.py --json BUILD_ID OUT.json "ROW|KERNEL,KERNEL=GLOB" ...   # profiles/r02_pmc.json for bench.py

--json: every ROW is "<workload>:<W>:<N>" (bench.py's key); KERNEL,... are name prefixes of the kernels one step of that
workload launches ("V1 full" is two kernels: their per-launch totals are added; kernels that only prepare the inputs of a
row are not listed); GLOB = the databases of that row's passes.  Counters are per-launch totals (instances summed,
launches averaged) and converted:
    traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024     MI355X_MICROARCH.md "HBM": the counters are KiB and gfx950's
                                                             FETCH_SIZE reports half of a wide streaming read
    fp64_flop     = 64 * (2 * FMA_F64 + MUL_F64 + ADD_F64 + TRANS_F64)   wave-level SQ instruction counters x 64 lanes
"""
import glob
import json
import sqlite3
import sys


def rows_of(db):
    """(kernel name, counter, per-dispatch TOTAL averaged over dispatches, dispatches).  rocprofv3 stores one row per counter
    instance (SQ counters: one per XCD / shader engine = 32 on MI355X): the instances of a dispatch are SUMMED, the
    dispatches averaged."""
    con = sqlite3.connect(db)
    cur = con.cursor()
    try:
        per = cur.execute("select name, counter_name, dispatch_id, sum(counter_value) from pmc_events "
                          "group by name, counter_name, dispatch_id").fetchall()
    except Exception:
        return []
    acc = {}
    for name, ctr, _, val in per:
        a = acc.setdefault((name, ctr), [0.0, 0])
        a[0] += val; a[1] += 1
    return [(name, ctr, tot / n, n) for (name, ctr), (tot, n) in sorted(acc.items())]


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main_text(pattern, name_filter="cpi_"):
    for db in sorted(glob.glob(pattern, recursive=True)):
        for name, ctr, val, n in rows_of(db):
            if name_filter in name:
                print("%-28s %-44s %-24s avg=%.6g n=%d" % (db.split("/")[-1][:28], short(name), ctr, val, n))


def main_json(build_id, out, specs):
    res = {"build_id": build_id, "rows": {},
           "how": "tools/pmc_collect.sh: one rocprofv3 --pmc pass per counter group and row (no trace domains); per-launch "
                  "means over the timed launches of tools/microbench.py; conversions in tools/pmc_summary.py"}
    for spec in specs:
        key, pattern = spec.split("=", 1)
        key, want = key.split("|", 1)
        want = [w.strip() for w in want.split(";") if w.strip()]
        acc, kernels = {}, set()
        for db in sorted(glob.glob(pattern, recursive=True)):
            for name, ctr, val, n in rows_of(db):
                sn = short(name)
                if any(sn.startswith(w) for w in want):
                    acc.setdefault(ctr, {})[sn] = val
                    kernels.add(sn)
        c = {ctr: sum(v.values()) for ctr, v in acc.items()}
        row = {"kernels": sorted(kernels), "counters": c}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            row["traffic_bytes"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        f = [c.get("SQ_INSTS_VALU_FMA_F64"), c.get("SQ_INSTS_VALU_MUL_F64"), c.get("SQ_INSTS_VALU_ADD_F64")]
        if all(x is not None for x in f):
            tr = c.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
            row["fp64_insts"] = f[0] + f[1] + f[2] + tr
            row["fp64_flop"] = 64.0 * (2.0 * f[0] + f[1] + f[2] + tr)
        if "SQ_INSTS_VALU" in c:
            row["valu_insts"] = c["SQ_INSTS_VALU"]
        res["rows"][key] = row
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in res["rows"].items()}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "--json":
        main_json(sys.argv[2], sys.argv[3], sys.argv[4:])
    else:
        main_text(sys.argv[1], *(sys.argv[2:3]))
