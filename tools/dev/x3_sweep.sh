#!/bin/bash
for env in "X3_DBG=0" "X3_DBG=1" "X3_DBG=3" "X3_DBG=7" "X3_DBG=2" "X3_DBG=4"; do
  echo "== $env"; X3_NORES=1 env $env timeout 20 tools/dev/x3_test 3 2>&1 | grep "time" | head -4
done
