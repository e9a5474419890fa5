#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats results .db into a small text summary
(per-kernel calls / total / average duration) fit for committing under profiles/."""
import sqlite3
import sys


def main(db_path, out_path, note=""):
    db = sqlite3.connect(db_path)
    rows = list(db.execute(
        "select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(out_path, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary ({db_path.split('/')[-1]})\n")
        if note:
            f.write(f"# {note}\n")
        f.write("# durations in microseconds\n")
        f.write(f"{'calls':>6} {'total_us':>14} {'avg_us':>14} {'pct':>7}  kernel\n")
        for name, calls, total, avg, pct in rows:
            short = name if len(name) < 100 else name[:97] + "..."
            f.write(f"{calls:6d} {total:14.3f} {avg:14.3f} {pct:7.3f}  {short}\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
