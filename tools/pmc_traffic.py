"""Builds profiles/<name>.json (HBM traffic per launch of one kernel) from two rocprofv3 --pmc passes.
usage: pmc_traffic.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <kernel substring> <workload text> <command text> <out.json>
The median over the launches of the kernel is taken (the set-up launches of the same instantiation carry
different traffic); FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE is doubled (MI355X_MICROARCH.md, gfx950)."""
import csv, glob, json, os, statistics, sys
fetch_dir, write_dir, kernel, workload, command, out = sys.argv[1:7]


def per_launch(d, counter):
    vals = []
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return vals


f, w = per_launch(fetch_dir, "FETCH_SIZE"), per_launch(write_dir, "WRITE_SIZE")
fetch, write = statistics.median(f) * 1024.0 * 2.0, statistics.median(w) * 1024.0
commit = os.environ.get("CSOLVE_COMMIT")  # the commit the profiled library was built at (csolve_amd/csrc/Makefile)
if not commit:
    try:
        commit = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "csolve_amd", "csrc", "build", "COMMIT")).read().strip()
    except OSError:
        commit = None
json.dump({"kernel": kernel, "workload": workload, "command": command, "commit": commit or None,
           "note": "median over the launches of this kernel; FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)",
           "hbm_traffic_bytes_per_launch": fetch + write, "fetch_bytes_corrected": fetch, "write_bytes": write,
           "launches": {"FETCH_SIZE": len(f), "WRITE_SIZE": len(w)}}, open(out, "w"), indent=1)
print(out, fetch + write)
