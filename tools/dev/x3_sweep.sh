#!/bin/bash
for env in "X3_NORES=1" "X3_NORES=1 X3_DBG=1" "X3_NORES=1 X3_DBG=2" "X3_NORES=1 X3_DBG=3" "X3_NORES=1 X3_DBG=4" "X3_NORES=1 X3_DBG=7" "X3_NORES=1 X3_ZC=4" "X3_NORES=1 X3_ZC=16" "X3_NORES=1 X3_ZC=48"; do
  echo "== $env"; env $env timeout 100 tools/dev/x3_test 2 2>&1 | grep time
done
