#!/bin/bash
# debug: repeat the sweep's child runs (tools/debug/plan_sweep.py) and report every configuration whose digest differs
# between repetitions -- a result that is not a function of its inputs.
cd /root/repo
export SWEEP_SEED=${SWEEP_SEED:-31337} SWEEP_NCFG=${SWEEP_NCFG:-160} SWEEP_NEPOCH=${SWEEP_NEPOCH:-500}
i=0
for mode in default default default default verify verify replica replica; do
  i=$((i+1))
  case $mode in
    verify) export GNSSCORR_PLAN_VERIFY=1; unset GNSSCORR_TRK_ALGO;;
    replica) export GNSSCORR_TRK_ALGO=replica; unset GNSSCORR_PLAN_VERIFY;;
    *) unset GNSSCORR_PLAN_VERIFY GNSSCORR_TRK_ALGO;;
  esac
  timeout -k 10 280 python tools/debug/plan_sweep.py $mode > gpurun_out/rep_$i.log 2>&1 || { echo "run $i ($mode) failed"; tail -5 gpurun_out/rep_$i.log; exit 1; }
  echo "run $i ($mode) done"
done
python - <<'PY'
import json, collections
runs = []
for i in range(1, 9):
    runs.append({json.loads(l)["cfg"]: json.loads(l)["digest"] for l in open("gpurun_out/rep_%d.log" % i) if l.startswith("{")})
bad = 0
for k in sorted(runs[0]):
    c = collections.Counter(r.get(k) for r in runs)
    if len(c) > 1:
        bad += 1
        print("cfg", k, "digests by run:", [r.get(k, "")[:10] for r in runs])
print("configurations", len(runs[0]), "x 8 runs; configurations with differing digests:", bad)
PY
