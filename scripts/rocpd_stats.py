#!/usr/bin/env python3
"""Kernel statistics (the table `rocprofv3 --kernel-trace --stats` prints) from a rocpd SQLite
database, written as CSV.  usage: rocpd_stats.py results.db out.csv"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
        "max(d.end - d.start), max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), "
        "max(d.group_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "VGPR", "AGPR",
                    "SGPR", "LDS"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.2f" % (100.0 * r[2] / total), r[4], r[5], r[6], r[7], r[8],
                        r[9]])
    return rows, total


if __name__ == "__main__":
    rows, total = main(sys.argv[1], sys.argv[2])
    for r in rows[:45]:
        print("%6.2f%%  n=%5d  avg %9.1f us  %s" % (100.0 * r[2] / total, r[1], r[3] / 1e3, r[0][:110]))
