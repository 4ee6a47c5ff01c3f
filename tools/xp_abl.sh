#!/bin/bash
# timing ablations of the paired-wave kernel on one layer: bash tools/xp_abl.sh "0 32 1 30" HW CIN COUT
set -uo pipefail
L="$1"; shift
for D in $L; do
  echo -n "dbg=$D  "
  SHDR_X3P_DBG=$D timeout -k 10 120 python3 tools/x3_one.py "$@" 2>&1 | tail -1
done
