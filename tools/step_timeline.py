"""Print the kernel timeline of the last training step from a rocprofv3 kernel-trace CSV.
Usage: python tools/step_timeline.py <..._kernel_trace.csv>"""
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # step boundary = the LAST Adam launch in front of a forward launch (the table's and the dense parameters' updates may be
    # several launches; configurations with small tables end a step with two small ones)
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    fwd = lambda r: any(k in r["Kernel_Name"] for k in ("fwd", "input_proj", "embed_gather"))
    ends = []
    for i in adam:
        nxt = next((j for j in range(i + 1, len(rows)) if fwd(rows[j]) or "adam_kernel" in rows[j]["Kernel_Name"]), None)
        if nxt is None or fwd(rows[nxt]):
            ends.append(i)
    if len(ends) < 2:
        ends = adam
    a, b = ends[-2], ends[-1]
    t0 = int(rows[a]["End_Timestamp"])
    for r in rows[a + 1:b + 1]:
        name = r["Kernel_Name"].split("(")[0].replace("void hpmn::", "").replace("hpmn::", "")[:44]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%-46s start %8.1f  dur %7.1f us  queue %s" % (name, (s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "")))
    print("step: %.1f us" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
