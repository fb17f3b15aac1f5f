#!/usr/bin/env python3
"""profiles/traffic.json entry for one workload from two rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE, collected separately: MI355X_MICROARCH.md "rocprofv3 PMC slots").

    python tools/make_traffic.py <workload> <kernel-substring> <fetch.csv> <write.csv>

HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KB): on gfx950 FETCH_SIZE reports half of the bytes of
wide 16-B/lane streaming reads (same guide, section HBM); mean over the full-batch dispatches of the
kernel.  The entry is keyed by the kernel name and the sha256 over the library's sources and compile flags
(scann_rust_amd/build.py src_sha256), so bench.py emits it only for those sources (else null), wherever they were built."""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_full(path, sub, counter):
    v = {}
    for r in csv.DictReader(open(path)):
        if sub in r["Kernel_Name"] and r["Counter_Name"] == counter:
            v[r["Dispatch_Id"]] = v.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    vals = list(v.values())
    if not vals:
        raise SystemExit("no %s rows for %s in %s" % (counter, sub, path))
    big = [x for x in vals if x >= 0.5 * max(vals)]
    return sum(big) / len(big), len(big)


def main():
    workload, sub, fetch_csv, write_csv = sys.argv[1:5]
    f, nf = mean_full(fetch_csv, sub, "FETCH_SIZE")
    w, nw = mean_full(write_csv, sub, "WRITE_SIZE")
    sys.path.insert(0, ROOT)
    from scann_rust_amd import build as hip_build
    sha = hip_build.src_sha256()
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        d = json.load(open(path))
    except Exception:
        d = {}
    d["_comment"] = ("HBM bytes per launch of the dominant kernel from rocprofv3 --pmc (separate FETCH_SIZE / "
                     "WRITE_SIZE passes; mean over the full-batch dispatches): 2*FETCH_SIZE + WRITE_SIZE KB "
                     "(gfx950: FETCH_SIZE counts half of wide 16-B/lane reads, MI355X_MICROARCH.md section HBM). "
                     "Entries are valid for the named kernel of the library sources with the given src_sha256 only.")
    d[workload] = {"kernel": sub, "src_sha256": sha, "bytes": int((2.0 * f + w) * 1024),
                   "fetch_kb": f, "write_kb": w, "dispatches": [nf, nw],
                   "source": "%s + %s" % (os.path.basename(fetch_csv), os.path.basename(write_csv))}
    json.dump(d, open(path, "w"), indent=1)
    print(json.dumps(d[workload]))


if __name__ == "__main__":
    main()
