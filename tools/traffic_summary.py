"""HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, separate runs).
gfx950: FETCH_SIZE counts 16-B/lane streaming reads at half (MI355X_MICROARCH.md 'HBM') -> x2; both counters are in KiB.
The JSON records the source tree it was taken on ("head": SOFTSPOKEN_HEAD, which tools/profile_rNN.sh's caller exports -- the GPU box has
no .git --, and "tree_sha16": a hash over softspoken_amd/csrc) and every kernel name it saw: bench.py prints a traffic figure only for
a kernel the file names and whose profiled duration agrees with the run's (bench.py traffic_of).
usage: python tools/traffic_summary.py <fetch_dir> <write_dir> <windows_per_launch> <out.json>"""
import collections, csv, glob, hashlib, json, os, sys

def tree_sha16():
    h = hashlib.sha256()
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "softspoken_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]

def counters(d, name):
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return vals

def durations(d):
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return dur

def short(k):           # "void ss::conv3x3_v4_kernel<...>(ss::ConvArgs, int, int)" -> "conv3x3_v4_kernel<...>"
    k = k.replace("void ", "").replace("ss::", "")
    depth = 0
    for i, ch in enumerate(k):
        if ch == "<": depth += 1
        elif ch == ">": depth -= 1
        elif ch == "(" and depth == 0: return k[:i]
    return k

fetch, write, dur = counters(sys.argv[1], "FETCH_SIZE"), counters(sys.argv[2], "WRITE_SIZE"), durations(sys.argv[1])
nwin = int(sys.argv[3])
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on tools/run_chunks.py <precision> %d 2; "
                 "FETCH_SIZE x 1024 x 2 (gfx950 counts 16-B/lane streaming reads at half), WRITE_SIZE x 1024; MI355X_MICROARCH.md 'HBM'; "
                 "second (warm) repetition of each launch" % nwin,
       "head": os.environ.get("SOFTSPOKEN_HEAD", "unknown"), "tree_sha16": tree_sha16(),
       "windows_per_launch": nwin, "kernels": {}}
for k in sorted(fetch, key=lambda k: -sum(dur.get(k, [0]))):
    n = len(fetch[k])
    half = n // 2 if n >= 2 else 0                     # launches of the second repetition
    f = sum(fetch[k][half:]) / max(1, n - half) * 1024 * 2
    w = sum(write.get(k, [0] * n)[half:]) / max(1, n - half) * 1024
    us = sum(dur[k][half:]) / max(1, len(dur[k]) - half) if k in dur else 0.0
    out["kernels"][short(k)] = {"launches_per_pass": n - half, "fetch_bytes_per_launch": f, "write_bytes_per_launch": w,
                                "hbm_bytes_per_launch": f + w, "hbm_bytes_per_window": (f + w) / nwin, "avg_us": round(us, 2),
                                "tb_per_s": round((f + w) / us / 1e6, 2) if us else None}
out["kernel_names"] = sorted(out["kernels"])
json.dump(out, open(sys.argv[4], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k[:72]:72s} n={v['launches_per_pass']} fetch {v['fetch_bytes_per_launch']/1e6:8.1f} MB write {v['write_bytes_per_launch']/1e6:8.1f} MB {v['avg_us']:8.1f} us {v['tb_per_s']} TB/s")
