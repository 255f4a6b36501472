"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_summary.py <dir> [<dir> ...]"""
import csv, glob, json, sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for d in sys.argv[1:]:
        for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(fn)):
                k = row["Kernel_Name"]
                c = acc[k][row["Counter_Name"]]
                c[0] += 1
                c[1] += float(row["Counter_Value"])
    out = {k: {c: {"calls": v[0], "total": v[1], "per_call": v[1] / max(v[0], 1)} for c, v in cs.items()} for k, cs in acc.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
