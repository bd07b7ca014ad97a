#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of a profiling
workload into gpurun_out/traffic_<tag>.json (copy it to profiles/, where bench.py
looks for it): mean HBM bytes per launch of every nmpc kernel, with a hash of the
library sources the counters were collected on.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write ...csv> <tag> <workload text>

FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte
read requests as 64 bytes, so the read side is doubled.  Cross-check of the
write side: k_linearise stores 409 600 stages x (C 1440 B + c 96 B + six
15-vectors) = 924 MB per launch and the counter reports 932 MB.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

SHORT = {"k_linearise": "linearise", "k_linearise_free": "linearise_free", "k_qp_begin": "qp_begin",
         "k_newton_solve": "newton_solve", "k_newton_solve_team": "newton_solve",
         "k_newton_step": "newton_step", "k_step_hessian": "newton_step",
         "k_init_stages": "init_stages", "k_line_search_tail": "line_search_tail", "k_line_search_fast": "line_search_fast", "k_line_search_exact": "line_search_exact",
         "k_line_search_commit": "line_search_commit", "k_line_search_hessian": "line_search_hessian",
         "k_list_count": "list_count", "k_list_fill": "list_fill", "k_store_refs": "store_refs",
         "k_qp_finish": "qp_finish", "k_gather_results": "gather_results"}


# Kernels that follow another one inside the same timed interval of the library: their bytes
# count, their launches do not.
FOLLOWERS = {"k_step_hessian"}


def collect(path, counter):
    per = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            m = re.search(r"nmpc::(?:\w+::)*(k_\w+)", row["Kernel_Name"])
            name = m.group(1) if m else row["Kernel_Name"].split("(")[0]
            per[name].append(float(row["Counter_Value"]))
    return per


def main():
    fetch_csv, write_csv, tag, workload = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    fetch = collect(fetch_csv, "FETCH_SIZE")
    write = collect(write_csv, "WRITE_SIZE")
    out = {"workload": workload,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; "
                     "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, mean over the launches",
           "bytes_per_launch": {}, "read_bytes_per_launch": {}, "write_bytes_per_launch": {},
           "launches": {}}
    # per name of the library's timers: the kernels of one name together
    total = defaultdict(lambda: [0.0, 0.0, 0])
    for name in sorted(set(fetch) | set(write)):
        if name not in SHORT:
            continue
        t = total[SHORT[name]]
        t[0] += 2.0 * 1024.0 * sum(fetch.get(name, [0]))
        t[1] += 1024.0 * sum(write.get(name, [0]))
        if name not in FOLLOWERS:
            t[2] += len(fetch.get(name, []))
    for k, (r, w, n) in sorted(total.items()):
        n = max(1, n)
        out["bytes_per_launch"][k] = round((r + w) / n)
        out["read_bytes_per_launch"][k] = round(r / n)
        out["write_bytes_per_launch"][k] = round(w / n)
        out["launches"][k] = n
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import hashlib
    h = hashlib.sha1()
    src = os.path.join(root, "nmpc_amd", "csrc")
    for name in sorted(os.listdir(src)):      # the same hash as bench.py's kernel_sources_sha1
        if name.endswith((".hip", ".hpp", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(src, name), "rb").read())
    out["kernel_sources_sha1"] = h.hexdigest()[:12]
    dst = os.path.join(root, "gpurun_out", "traffic_%s.json" % tag)
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(tag, json.dumps(out["bytes_per_launch"]))


if __name__ == "__main__":
    main()
