#!/bin/bash
# one network's forward under several builds of the library on one box:  bash tools/ab_libs.sh "deq lin" wave noat ...
set -euo pipefail
nets=$1; shift
for n in $nets; do
  python _base/tools/net_one.py $n 2>/dev/null | sed 's/^/base  /'
  python tools/net_one.py $n 2>/dev/null | sed 's/^/head  /'
  for v in "$@"; do SHDR_LIB=$PWD/singlehdr-tf2_amd/libshdr_$v.so python tools/net_one.py $n 2>/dev/null | sed "s/^/$v  /"; done
done
