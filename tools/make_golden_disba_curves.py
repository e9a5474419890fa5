#!/usr/bin/env python3
"""Extract the disba known-answer vectors the reference embeds as numeric
literals and store them as a data fixture, tests/golden/disba_curves.json.

Source of the numbers (read as text, container only):
  /root/reference/src/plotting/plot_dispersion_curve.py
    :1512-1530  "inversion test model 4": 5 parameters + 100 phase velocities
                (numpy 8-digit print inside a docstring)
    :1533-1548  three [thickness; vp; vs; rho] 4x3 layer tables (docstring)
    :1599       the period grid, np.flip(1 / np.logspace(0, 1.1, 100))
    :1601-1925  m1/dc1, m2/dc2, m3/dc3: 5 parameters -> 100 phase velocities

They are outputs of disba 0.6.1 PhaseDispersion(...)(periods, mode=0,
wave="rayleigh") for n_layers=2, vpvs=1.75 and Gardner density, printed to 8
significant digits.  Only numbers are copied; no source text is kept.
"""
import json
import re
import sys

SRC = "/root/reference/src/plotting/plot_dispersion_curve.py"
OUT = "tests/golden/disba_curves.json"

NUM = re.compile(r"[-+]?\d+\.\d*(?:[eE][-+]?\d+)?|[-+]?\d+\.(?!\d)")


def numbers(text):
    return [float(x) for x in NUM.findall(text)]


def array_literal(lines, start):
    """Numbers of the np.array(...) literal opening on line index `start`."""
    depth, buf = 0, []
    for ln in lines[start:]:
        buf.append(ln)
        depth += ln.count("(") - ln.count(")")
        if depth <= 0 and buf:
            break
    body = "".join(buf)
    body = body[body.index("np.array") + len("np.array"):]
    return numbers(body)


def main():
    lines = open(SRC).read().splitlines(keepends=True)
    out = {
        "source": "src/plotting/plot_dispersion_curve.py:1509-1548,1599-1925 (numeric literals only)",
        "periods_expr": "np.flip(1 / np.logspace(0, 1.1, 100))",
        "n_layers": 2,
        "vpvs": 1.75,
        "curves": [],
        "layer_tables": [],
    }
    # m1..m3 / dc1..dc3
    for name in ("1", "2", "3"):
        mi = next(i for i, l in enumerate(lines) if re.match(rf"\s+m{name} = np\.array", l))
        di = next(i for i, l in enumerate(lines) if re.match(rf"\s+dc{name} = np\.array", l))
        m = array_literal(lines, mi)
        dc = array_literal(lines, di)
        assert len(m) == 5 and len(dc) == 100, (name, len(m), len(dc))
        out["curves"].append({"name": "m" + name, "params": m, "velocity": dc})
    # docstring model 4 (lines 1512-1530, 1-based)
    doc = "".join(lines[1512:1530])
    vals = numbers(doc)
    assert len(vals) == 105, len(vals)
    out["curves"].append({"name": "model4", "params": vals[:5], "velocity": vals[5:]})
    # docstring layer tables (lines 1535-1548)
    doc = "".join(lines[1534:1548])
    vals = numbers(doc)
    assert len(vals) == 36, len(vals)
    for t in range(3):
        rows = [vals[12 * t + 3 * r: 12 * t + 3 * r + 3] for r in range(4)]
        out["layer_tables"].append(
            {"thickness": rows[0], "vp": rows[1], "vs": rows[2], "rho": rows[3]}
        )
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, [len(c["velocity"]) for c in out["curves"]])


if __name__ == "__main__":
    sys.exit(main())
