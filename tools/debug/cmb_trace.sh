#!/bin/bash
# debug (GPU box): kernel + copy trace of the pthread harness (tests/host/threads_harness.c: 32 threads calling the
# drop-in sdrtracking()), to see what a combined launch chain spends on the device.  Noise input: the call rate does
# not depend on the signal.
set -e
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/cmbtrace
mkdir -p $OUT
python3 - <<'PY'
import numpy as np
rng = np.random.default_rng(5)
rng.integers(-60, 61, size=56 * 65536, dtype=np.int8).tofile("/tmp/if32.dat")
PY
PKG=$PWD/erlangnetwork-gnsslib-sdr_amd
gcc -O2 -o /tmp/threads_harness tests/host/threads_harness.c -L$PKG -lgnsscorr -lpthread -lm -Wl,-rpath,$PKG
GNSSCORR_CMB_PROF=1 /tmp/threads_harness /tmp/if32.dat 56 32 200 > $OUT/plain.txt 2> $OUT/plain.err
head -1 $OUT/plain.txt; cat $OUT/plain.err
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/rp -- /tmp/threads_harness /tmp/if32.dat 56 32 200 > $OUT/traced.txt 2> $OUT/traced.err
head -1 $OUT/traced.txt
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/cmbtrace/rp/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
for f in glob.glob("gpurun_out/cmbtrace/rp/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")))
rows.sort()
mid = len(rows) // 2
t0 = rows[mid][0]
with open("gpurun_out/cmbtrace/timeline.txt", "w") as f:
    for s, e, n in rows[mid:mid + 40]:
        f.write("%9.1f us  +%6.1f us  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, n))
print(open("gpurun_out/cmbtrace/timeline.txt").read())
PY
