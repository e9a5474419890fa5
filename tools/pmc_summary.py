#!/usr/bin/env python3
"""Turn the csv output of three rocprofv3 --pmc passes (SQ block, FETCH_SIZE, WRITE_SIZE) over
`bench.py --no-cpu --steps 500 --warmup 250` into the text summary and the traffic file that are
committed under profiles/.  usage: pmc_summary.py <dir with pmc_<pass>/pmc_counter_collection.csv> <out prefix>"""
import collections
import csv
import json
import re
import sys

SQ = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU",
      "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"]
KERNELS = ("mh_steps_multi_kernel<2>", "mh_steps_multi_kernel<4>", "schedule_kernel")
CALLS = 3  # timed calls of mhd_mh_steps in that command (500 steps, 250 per call, after one warm-up call)


def main(src, prefix):
    tot = collections.defaultdict(float)
    for name in ("SQ_WAVES", "FETCH_SIZE", "WRITE_SIZE"):
        for r in csv.DictReader(open(f"{src}/pmc_{name}/pmc_counter_collection.csv")):
            m = re.search(r"(mh_steps_multi_kernel<\d>|schedule_kernel)", r["Kernel_Name"])
            if m:
                tot[(m.group(1), r["Counter_Name"])] += float(r["Counter_Value"])
    per = {k: v / CALLS for k, v in tot.items()}
    out = ["# rocprofv3 --pmc, three separate passes (SQ block / FETCH_SIZE / WRITE_SIZE), csv output",
           "# command per pass: rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --no-cpu --steps 500 --warmup 250",
           "# values are PER CALL of mhd_mh_steps (250 MH steps x 1024 chains): the sum over the call's two launches of each",
           "# kernel (first launch: all chains up to the deadline; second: the unfinished ones), mean of the 3 timed calls",
           "#", "# kernel                    " + "  ".join("%-20s" % n for n in SQ)]
    for k in KERNELS:
        out.append("%-27s " % k + "  ".join("%-20.4g" % per.get((k, n), 0.0) for n in SQ))
    out += ["#", "# kernel                    FETCH_SIZE [KB]   WRITE_SIZE [KB]"]
    f_all = w_all = 0.0
    for k in KERNELS:
        f, w = per.get((k, "FETCH_SIZE"), 0.0), per.get((k, "WRITE_SIZE"), 0.0)
        f_all += f
        w_all += w
        out.append("%-27s %-17.4g %-17.4g" % (k, f, w))
    busy = per[(KERNELS[0], "SQ_ACTIVE_INST_VALU")] / per[(KERNELS[0], "SQ_WAVE_CYCLES")]
    out += ["#", "# Reading:",
            "# * VALU: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = %.2f per wave of the <2> kernel, i.e. with two waves per SIMD the" % busy,
            "#   FP64 pipe is busy ~%.0f %% of the time while both are resident." % (200 * busy),
            "# * Memory: FETCH_SIZE + WRITE_SIZE = %.1f GB per call against ~8 MB of algorithmic traffic (draws 20 B per" % ((f_all + w_all) * 1024 / 1e9),
            "#   chain-step, chain state twice per call).  The excess is private (scratch) memory: register save/restore around",
            "#   the out-of-line calls; it fits the Infinity Cache.  On gfx950 FETCH_SIZE under-reports wide streaming reads by",
            "#   2x; these are dword/dwordx2 scratch accesses, uncalibrated."]
    open(prefix + "_pmc.txt", "w").write("\n".join(out) + "\n")
    json.dump({"fetch_kb_per_call": f_all, "write_kb_per_call": w_all, "source": prefix.split("/")[-1] + "_pmc.txt",
               "command": "python3 bench.py --no-cpu --steps 500 --warmup 250 under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)"},
              open(prefix + "_traffic.json", "w"), indent=1)
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
