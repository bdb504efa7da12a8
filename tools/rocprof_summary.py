"""Summarise a rocprofv3 results .db (kernel trace and/or PMC) as text: per-kernel launch
statistics, and per-kernel average counter values when counters were collected."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    q = """select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start),
                  sum(d.end-d.start), max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count),
                  max(d.group_segment_size), max(d.workgroup_size_x*d.workgroup_size_y*d.workgroup_size_z),
                  max(d.grid_size_x*d.grid_size_y*d.grid_size_z)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 6 desc"""
    rows = list(cur.execute(q))
    tot = sum(r[5] for r in rows) or 1
    print("%-64s %6s %10s %10s %10s %6s %5s %5s %5s %7s %6s %9s" % (
        "kernel", "calls", "avg_ns", "min_ns", "max_ns", "pct", "vgpr", "agpr", "sgpr", "lds_B", "wg", "grid"))
    for r in rows:
        print("%-64s %6d %10.0f %10.0f %10.0f %6.2f %5s %5s %5s %7s %6s %9s" % (
            r[0][:64], r[1], r[2], r[3], r[4], 100.0 * r[5] / tot, r[6], r[7], r[8], r[9], r[10], r[11]))
    try:
        q = """select s.kernel_name, p.name, avg(e.value), count(*)
               from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
               join rocpd_kernel_dispatch d on e.event_id = d.event_id
               join rocpd_info_kernel_symbol s on d.kernel_id = s.id
               group by s.kernel_name, p.name order by s.kernel_name, p.name"""
        rows = list(cur.execute(q))
        if rows:
            print("\nper-dispatch average counter values")
            for r in rows:
                print("%-64s %-28s %16.1f  (n=%d)" % (r[0][:64], r[1], r[2], r[3]))
    except sqlite3.Error as ex:
        print("no pmc tables:", ex)


if __name__ == "__main__":
    main(sys.argv[1])
