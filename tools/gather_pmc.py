"""The gather probes, timed and counted, at the tables the configs really use and at a cold one.

    python tools/gather_pmc.py time   > gpurun_out/gather_time.json          # HIP-event timings, all cases
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d D -- python tools/gather_pmc.py count
    python tools/gather_pmc.py digest <counter_collection.csv> <gather_time.json> <out.json>

Cases (C3 id shape B=500, T=1001, F=2 unless noted; 4 distinct id batches rotated, table rows drawn uniformly):
  cold      4 GiB table (64 Mi rows), far beyond the 256 MiB Infinity Cache
  c3        the C3 table itself, 3 308 019 rows = 212 MB (inside the Infinity Cache)
  c1        the C1 table, 256 205 rows = 16 MB, C1's id shape (B=128, T=100, F=3)
  seq       CALIBRATION: the cold table read through the same kernels with ids 0..n-1 -- every row exactly once, in order, so
            the bytes the counters should see are known (64 B per row): FETCH_SIZE per row of this case is the counter's scale
            for THIS access width (the guide calibrates only 16 B/lane streaming reads)
Kernels: embed_gather_seq_kernel (rows materialised) and embed_gather_sum_kernel (rows consumed in place).
FETCH bytes per row = FETCH_SIZE[KB] * 1024 / rows; `x2` = with the guide's gfx950 coalesced-read correction, reported beside
the raw figure; the calibration case says which one applies to a 64-byte row."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [  # name, table rows, (B, T, F), sequential ids
    ("cold", 64 * 1024 * 1024, (500, 1001, 2), False),
    ("c3", 19002 + 3269017 + 20000, (500, 1001, 2), False),
    ("c1", 63001 + 801 + 192403, (128, 100, 3), False),
    ("seq", 64 * 1024 * 1024, (500, 1001, 2), True),
]
N_BATCH = 4
LAUNCHES = 8          # per (case, kernel) in the counted / timed region


def run(mode):
    import torch
    from hpmn_amd import build, ops
    build.build_library()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream()
    gen = torch.Generator(device=dev).manual_seed(99)
    res = {}
    tables = {}
    for name, V, (B, T, F), seq in CASES:
        if V not in tables:
            tables[V] = torch.empty(V, 16, device=dev).normal_(0.0, 0.1)
        tab = tables[V]
        n = B * T * F
        if seq:
            ids = [(torch.arange(n, device=dev, dtype=torch.int64) + k * n).to(torch.int32).view(B, T, F) for k in range(N_BATCH)]
        else:
            ids = [torch.randint(0, V, (B, T, F), device=dev, dtype=torch.int32, generator=gen) for _ in range(N_BATCH)]
        gout = torch.empty(B, T, F * 16, device=dev)
        gsum = torch.zeros(B, F * 16, device=dev)
        for kname, fn in (("embed_gather_seq_kernel", lambda i: ops.embed_gather_seq(ids[i % N_BATCH], tab, 0, False, out=gout)),
                          ("embed_gather_sum_kernel", lambda i: ops.embed_gather_sum(ids[i % N_BATCH], tab, False, out=gsum))):
            if mode == "time":
                for i in range(4):
                    fn(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for i in range(4 * LAUNCHES):
                    fn(i)
                e1.record(st)
                e1.synchronize()
                ms = e0.elapsed_time(e1) / (4 * LAUNCHES)
                alg = n * 68
                res["%s/%s" % (name, kname)] = {"ms": ms, "rows": n, "table_rows": V, "table_bytes": V * 64,
                                                "algorithmic_GBs": alg / ms / 1e6, "frac_of_8TBs": alg / ms / 1e6 / 8000.0}
            else:
                for i in range(LAUNCHES):
                    fn(i)
                torch.cuda.synchronize()
        del ids, gout, gsum
    if mode == "time":
        print(json.dumps(res, indent=1))


def digest(counter_csv, time_json, out):
    rows = [r for r in csv.DictReader(open(counter_csv)) if r.get("Counter_Name") == "FETCH_SIZE"
            and "embed_gather_s" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    times = json.load(open(time_json))
    import hashlib
    sha = hashlib.sha1(open(os.path.join(ROOT, "hpmn_amd", "csrc", "embed.hip"), "rb").read()).hexdigest()[:16]
    d = {"embed_hip_sha": sha, "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/gather_pmc.py count  (separate pass; timings from "
                    "`tools/gather_pmc.py time`, HIP events, no profiler)",
         "units": "FETCH_SIZE in KB as rocprofv3 reports it; bytes_per_row_raw = FETCH*1024/rows, _x2 = with the gfx950 coalesced-"
                  "read correction of MI355X_MICROARCH.md; the `seq` case (every row once, in order: 64 B per row are the "
                  "truth) shows which of the two the counter means for 64-byte row reads",
         "cases": {}}
    k = 0
    for name, V, (B, T, F), seq in CASES:
        for kname in ("embed_gather_seq_kernel", "embed_gather_sum_kernel"):
            mine = rows[k:k + LAUNCHES]
            k += LAUNCHES
            assert len(mine) == LAUNCHES and all(kname in r["Kernel_Name"] for r in mine), (name, kname, len(mine))
            v = [float(r["Counter_Value"]) for r in mine]
            n = B * T * F
            key = "%s/%s" % (name, kname)
            d["cases"][key] = dict(times.get(key, {}), FETCH_SIZE_KB_avg=sum(v) / len(v), FETCH_SIZE_KB_min=min(v),
                                   FETCH_SIZE_KB_max=max(v), fetch_bytes_per_row_raw=sum(v) / len(v) * 1024.0 / n,
                                   fetch_bytes_per_row_x2=2.0 * sum(v) / len(v) * 1024.0 / n)
    assert k == len(rows), (k, len(rows))
    json.dump(d, open(out, "w"), indent=1)
    for key, c in d["cases"].items():
        print("%-36s %8.1f us  %6.0f GB/s alg (%.3f)   FETCH/row raw %6.1f B  x2 %6.1f B" % (
            key, c.get("ms", 0) * 1e3, c.get("algorithmic_GBs", 0), c.get("frac_of_8TBs", 0),
            c["fetch_bytes_per_row_raw"], c["fetch_bytes_per_row_x2"]))


if __name__ == "__main__":
    if sys.argv[1] == "digest":
        digest(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        run(sys.argv[1])
