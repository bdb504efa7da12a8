"""Print the dispatches of a rocprofv3 .db in launch order (name, duration) -- for seeing how a kernel's time moves from
call to call (e.g. the approx-EMD passes level by level).  Usage: rocprof_sequence.py <db> [name-substring] [first] [count]"""
import sqlite3
import sys


def main(path, sub="", first=0, count=80):
    cur = sqlite3.connect(path).cursor()
    q = """select s.kernel_name, d.end - d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           order by d.start"""
    rows = [r for r in cur.execute(q) if sub in r[0]]
    for name, ns in rows[first:first + count]:
        print("%-70s %9.1f us" % (name[:70], ns / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]), *map(int, sys.argv[3:5]))
