"""Print the kernel timeline of the last training step from a rocprofv3 kernel-trace CSV.
Usage: python tools/step_timeline.py <..._kernel_trace.csv>"""
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # step boundary = the small (dense-parameter) Adam launch that ends a step; a step that is not split has
    # one launch only
    adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
    small = [i for i in adam if int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) < 50000]
    if len(small) >= 2:
        adam = small
    a, b = adam[-2], adam[-1]
    t0 = int(rows[a]["End_Timestamp"])
    for r in rows[a + 1:b + 1]:
        name = r["Kernel_Name"].split("(")[0].replace("void hpmn::", "").replace("hpmn::", "")[:44]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%-46s start %8.1f  dur %7.1f us  queue %s" % (name, (s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "")))
    print("step: %.1f us" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
