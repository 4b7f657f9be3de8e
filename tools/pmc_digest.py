"""Digest of the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, --kernel-trace only) of
    python bench.py --config C --steps S --warmup W --no-cpu-baseline --no-roofline --no-auc --no-parity-gate --no-eval
into profiles/rNN_pmc_summary.json: per kernel calls / avg / max of both counters, HBM bytes = (2 FETCH + WRITE) KB
(gfx950: FETCH_SIZE under-reports coalesced reads by exactly 2, MI355X_MICROARCH.md HBM section), the per-step
total over every kernel of the run, and the sha of the kernel sources it was taken on (bench.py quotes the digest
only while that still matches).
Usage: python tools/pmc_digest.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps+warmup>
                                  <config_id> <batch> <out.json>"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(path, counter):
    """-> ({kernel: [values of the launches of the steps]}, KB of the launches IN FRONT of the first product kernel).  The
    model's construction (torch.zeros of the flat gradient / moment buffers: three table-sized fills) is not part of any
    step; until round 4 the digest divided it into the per-step figure (+0.2 GB at S + W = 4)."""
    rows = [r for r in csv.DictReader(open(path)) if r.get("Counter_Name") == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    first = next((i for i, r in enumerate(rows) if "hpmn::" in r["Kernel_Name"]), 0)
    acc = defaultdict(list)
    for r in rows[first:]:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace(" ", "")
        name = name.replace("hpmn::", "")
        acc[name].append(float(r["Counter_Value"]))
    return acc, sum(float(r["Counter_Value"]) for r in rows[:first])


def main(fetch_csv, write_csv, nsteps, config_id, batch, out):
    import bench
    (f, f_setup), (w, w_setup) = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    kernels, total = {}, 0.0
    for name in sorted(set(f) | set(w)):
        fv, wv = f.get(name, [0.0]), w.get(name, [0.0])
        tot = (2.0 * sum(fv) + sum(wv)) * 1024.0
        total += tot
        kernels[name] = {"calls": max(len(fv), len(wv)), "FETCH_SIZE_avg_KB": sum(fv) / len(fv),
                         "FETCH_SIZE_max_KB": max(fv), "WRITE_SIZE_avg_KB": sum(wv) / len(wv),
                         "WRITE_SIZE_max_KB": max(wv),
                         "hbm_bytes_max_launch": (2.0 * max(fv) + max(wv)) * 1024.0,
                         "hbm_bytes_total_per_step": tot / nsteps}
    d = {"command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python bench.py --config %s --steps S "
                    "--warmup W --no-cpu-baseline --no-roofline --no-auc --no-parity-gate --no-eval (S+W = %d)"
                    % (config_id, nsteps),
         "units": "rocprofv3 reports KB; hbm bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE x2 correction)",
         "source_sha": bench.source_sha(), "config_id": config_id, "batch": batch, "steps_in_run": nsteps,
         "hbm_bytes_per_step": total / nsteps,
         "setup_bytes_not_in_any_step": (2.0 * f_setup + w_setup) * 1024.0, "kernels": kernels}
    json.dump(d, open(out, "w"), indent=1)
    print("hbm bytes per step: %.3f GB over %d kernels (source sha %s)" % (total / nsteps / 1e9, len(kernels), d["source_sha"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), sys.argv[6])
