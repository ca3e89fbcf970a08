#!/bin/bash
# ON THE GPU BOX: rocprofv3 evidence for RoIAlign's HBM rate: kernel durations (kernel trace) + FETCH_SIZE / WRITE_SIZE (separate PMC
# passes) of tools/roialign_bench.py (rotating input sets larger than the Infinity Cache) -> gpurun_out/roialign_profile.txt and
# gpurun_out/roialign_hbm.json (copied to profiles/r05_roialign_hbm.json: bench.py's hbm_kernels leg quotes its counter bytes).
set -e
root=$PWD; out=$PWD/gpurun_out/roialign_prof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $root/tools/roialign_bench.py > $out/events.txt 2>&1
rocprofv3 --kernel-trace --stats -d $out/trace -o ra -- python3 $root/tools/roialign_bench.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/f -- python3 $root/tools/roialign_bench.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/w -- python3 $root/tools/roialign_bench.py > /dev/null 2>&1
python3 - "$out" <<'PY'
import csv, glob, json, math, os, sqlite3, sys
out = sys.argv[1]
def pmc(d, name):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "roi_align_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return [v for _, v in sorted(vals)]
fetch, write = pmc(os.path.join(out, "f"), "FETCH_SIZE"), pmc(os.path.join(out, "w"), "WRITE_SIZE")
db = sqlite3.connect(os.path.join(out, "trace", "ra_results.db"))
durs = [r[0] for r in db.execute("select (end - start) from kernels where name like '%roi_align_kernel%' order by start")]
cfgs, pos = [], 0
for label, key, B, R in (("B=2 R=32", "roialign_bench_size", 2, 32), ("B=16 R=32", "roialign_16_images", 16, 32), ("B=16 R=128", "roialign_16x128", 16, 128)):
    alg = B * R * 250880.0
    nsets = int(min(24, max(3, math.ceil(320e6 / (0.8 * alg)) + 1)))
    n = 5 * nsets                                    # one warm-up round + ROUNDS = 4
    cfgs.append((label, key, alg, pos + nsets, pos + n))   # skip the warm-up round
    pos += n
res = {}
with open(os.path.join(os.path.dirname(out), "roialign_profile.txt"), "w") as f:
    f.write(open(os.path.join(out, "events.txt")).read())
    f.write("\nrocprofv3, rotating input sets (nothing cache-resident): mean kernel duration, HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KB counters), achieved HBM rate\n")
    for label, key, alg, lo, hi in cfgs:
        d = sum(durs[lo:hi]) / (hi - lo) / 1e3
        fb = sum(fetch[lo:hi]) / (hi - lo) * 2 * 1024
        wb = sum(write[lo:hi]) / (hi - lo) * 1024
        f.write("%-11s %7.2f us   fetch %8.2f MB  write %7.2f MB   %6.2f TB/s HBM (counter bytes)   %6.2f TB/s algorithmic\n"
                % (label, d, fb / 1e6, wb / 1e6, (fb + wb) / d / 1e6, alg / d / 1e6))
        res[key] = {"kernel_us_rocprof": round(d, 2), "counter_bytes_per_launch": fb + wb, "fetch_bytes": fb, "write_bytes": wb, "algorithmic_bytes": alg,
                    "launches_averaged": hi - lo}
json.dump(res, open(os.path.join(os.path.dirname(out), "roialign_hbm.json"), "w"), indent=1)
print(open(os.path.join(os.path.dirname(out), "roialign_profile.txt")).read())
PY
(echo; echo "== row / embedding gathers of the decoder (tools/gather_bench.py, HIP events)"; python3 $root/tools/gather_bench.py 2>&1 | grep -v amdgpu.ids) >> $root/gpurun_out/roialign_profile.txt
tail -8 $root/gpurun_out/roialign_profile.txt
rm -rf $out
