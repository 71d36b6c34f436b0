"""Summarise a rocprofv3 --kernel-trace rocpd database as a markdown table (dev tool).
   python tools/kernel_stats.py '/tmp/prof/**/*.db' > profiles/rNN_kernel_stats_table.md"""
import glob
import sqlite3
import sys


def main(pattern):
    rows = {}
    for db in sorted(glob.glob(pattern, recursive=True)):
        con = sqlite3.connect(db)
        cur = con.cursor()
        cols = [d[1] for d in cur.execute("pragma table_info(kernels)").fetchall()]
        get = lambda *names: next((n for n in names if n in cols), None)
        c_name, c_start, c_end = get("name", "kernel_name"), get("start", "start_timestamp"), get("end", "end_timestamp")
        c_vgpr, c_agpr, c_sgpr = get("arch_vgpr_count", "vgpr_count"), get("accum_vgpr_count"), get("sgpr_count")
        c_lds, c_scr = get("lds_size", "lds_block_size"), get("scratch_size", "private_segment_size")
        c_gx, c_wx = get("grid_x", "grid_size_x", "grid_size"), get("workgroup_x", "workgroup_size_x", "workgroup_size")
        sel = ", ".join(x if x else "NULL" for x in (c_name, c_start, c_end, c_vgpr, c_agpr, c_sgpr, c_lds, c_scr, c_gx, c_wx))
        for name, s, e, vg, ag, sg, lds, scr, gx, wx in cur.execute("select %s from kernels" % sel):
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            r = rows.setdefault(short, {"d": [], "meta": (vg, ag, sg, lds, scr, gx, wx)})
            r["d"].append((e - s) / 1e3)
    total = sum(sum(r["d"]) for r in rows.values())
    print("| kernel | calls | avg us | min us | max us | total ms | % | VGPR | AGPR | SGPR | LDS B | scratch | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for k, r in sorted(rows.items(), key=lambda kv: -sum(kv[1]["d"])):
        d = r["d"]
        vg, ag, sg, lds, scr, gx, wx = r["meta"]
        print("| %s | %d | %.2f | %.2f | %.2f | %.3f | %.1f | %s | %s | %s | %s | %s | %s | %s |" % (
            k, len(d), sum(d) / len(d), min(d), max(d), sum(d) / 1e3, 100.0 * sum(d) / total, vg, ag, sg, lds, scr, gx, wx))


if __name__ == "__main__":
    main(sys.argv[1])
