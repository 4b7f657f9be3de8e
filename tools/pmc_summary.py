"""Per-kernel average / max of one PMC counter from a rocprofv3 counter_collection CSV.
Usage: python tools/pmc_summary.py <counter_collection.csv> <COUNTER>"""
import csv
import sys
from collections import defaultdict


def main(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name].append(float(r["Counter_Value"]))
    print("%-70s %6s %14s %14s   (%s, units as reported by rocprofv3)" % ("kernel", "calls", "avg", "max", counter))
    for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print("%-70s %6d %14.1f %14.1f" % (name[:70], len(v), sum(v) / len(v), max(v)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
