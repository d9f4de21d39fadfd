#!/bin/bash
cd /root/repo
for e in planprof p4e1 p4e2 p4e4 p4e9 p4e15; do
  echo "== $e"
  GNSSCORR_LIB=$PWD/tools/variants/lib_$e.so NCH=1 timeout -k 10 100 python tools/debug/plan_alone.py 2>&1 | tail -2 || exit 1
done
