"""Condense a rocprofv3 `*_kernel_stats.csv` for `profiles/`: every NUMERIC column kept, only the kernel name shortened.

rocprofv3 prints the full demangled signature (a fused-kernel row is > 400 characters); cutting lines by width - what
`scripts/gpu_round.sh` did in round 3 - lost Calls / TotalDurationNs / AverageNs of exactly the longest row, the dominant
kernel.  Here a name loses its `(anonymous namespace)::` qualifiers and its ARGUMENT list (template arguments stay: they
identify the variant) and is capped at 200 characters; the 7 numeric fields are copied verbatim.

    python scripts/kernel_stats_summary.py <rocprof kernel_stats.csv> <out.csv> [--top N]
"""

from __future__ import annotations

import csv
import re
import sys

COLUMNS = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]


def short_name(name: str, limit: int = 200) -> str:
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void\s+", "", name)
    depth = 0
    for i, c in enumerate(name):  # cut at the '(' that opens the argument list (outside template brackets)
        if c == "<":
            depth += 1
        elif c == ">":
            depth -= 1
        elif c == "(" and depth == 0 and i > 0:
            name = name[:i]
            break
    name = re.sub(r"\s+", " ", name).strip()
    return name if len(name) <= limit else name[: limit - 3] + "..."


def condense(src: str, dst: str, top: int | None = None) -> int:
    with open(src, newline="") as fh:
        rows = list(csv.DictReader(fh))
    missing = [c for c in COLUMNS if rows and c not in rows[0]]
    if missing:
        raise SystemExit(f"{src}: not a rocprofv3 kernel stats file (no column {missing[0]!r})")
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    if top:
        rows = rows[:top]
    with open(dst, "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(COLUMNS)
        for r in rows:
            w.writerow([short_name(r["Name"])] + [float(r[c]) if "." in r[c] else int(r[c]) for c in COLUMNS[1:]])
    return len(rows)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else None
    if "--top" in sys.argv:
        args = [a for a in args if a != str(top)]
    if len(args) != 2:
        raise SystemExit(__doc__)
    print(f"{condense(args[0], args[1], top)} kernels -> {args[1]}")
