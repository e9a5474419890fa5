#!/usr/bin/env python3
"""Turn the csv output of the rocprofv3 --pmc passes of scratch/profile_r2.sh (two SQ passes,
FETCH_SIZE, WRITE_SIZE; each its own run of `bench.py --no-cpu <args>`) into the text summary
committed under profiles/.
usage: pmc_summary.py <dir with pmc_<pass>/pmc_counter_collection.csv> <out file> <steps> <note>"""
import collections
import csv
import re
import sys

SQ = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_FLAT",
      "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY",
      "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"]
KERNELS = ("mh_steps_multi_kernel<2>", "mh_steps_multi_kernel<4>", "schedule_kernel",
           "mh_steps_kernel", "forward_kernel")


def main(src, out_path, steps, note):
    tot = collections.defaultdict(float)
    seen = set()
    for name in ("SQ_WAVES", "SQ_WAIT_ANY", "FETCH_SIZE", "WRITE_SIZE"):
        try:
            rows = list(csv.DictReader(open(f"{src}/pmc_{name}/pmc_counter_collection.csv")))
        except OSError:
            continue
        for r in rows:
            m = re.search(r"(mh_steps_multi_kernel<\d>|schedule_kernel|mh_steps_kernel|forward_kernel)",
                          r["Kernel_Name"])
            # (a counter asked for in two passes is taken from the first)
            if m and (r["Counter_Name"], name) not in seen and \
                    not any(c == r["Counter_Name"] and p != name for c, p in seen):
                tot[(m.group(1), r["Counter_Name"])] += float(r["Counter_Value"])
        seen |= {(r["Counter_Name"], name) for r in rows}
    out = ["# rocprofv3 --pmc, separate passes (SQ block twice, FETCH_SIZE, WRITE_SIZE), csv output",
           f"# command per pass: rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --no-cpu {note}",
           f"# values are TOTALS over the command ({steps} MH steps of each of 1024 chains incl. warm-up);",
           "# SQ_*_CYCLES, SQ_ACTIVE_* and SQ_WAIT_* count quad-cycles",
           "#"]
    for k in KERNELS:
        if not any(kk == k for kk, _ in tot):
            continue
        out.append(k)
        for n in SQ:
            if (k, n) in tot:
                out.append("    %-22s %.4g" % (n, tot[(k, n)]))
        f, w = tot.get((k, "FETCH_SIZE"), 0.0), tot.get((k, "WRITE_SIZE"), 0.0)
        out.append("    %-22s %.4g KB   WRITE_SIZE %.4g KB   (%.3g GB per 250 MH steps of 1024 chains)"
                   % ("FETCH_SIZE", f, w, (f + w) * 1024 / 1e9 * 250 / steps))
        wc = tot.get((k, "SQ_WAVE_CYCLES"), 0.0)
        if wc:
            out.append("    per wave cycle: VALU active %.3f, any instruction active %.3f, issue stall "
                       "(WAIT_INST_ANY) %.3f, parked on s_waitcnt / barrier (WAIT_ANY) %.3f" % tuple(
                           tot.get((k, n), 0.0) / wc for n in
                           ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")))
    open(out_path, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), " ".join(sys.argv[4:]))
