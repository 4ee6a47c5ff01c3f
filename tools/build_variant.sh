#!/bin/bash
# a kernel-experiment build next to the product one:  bash tools/build_variant.sh NAME "-DFLAG ..." file.hip [file.hip ...]
# -> singlehdr-tf2_amd/libshdr_NAME.so (the named sources recompiled with the flags, every other object taken from build/); run with
#    SHDR_LIB=singlehdr-tf2_amd/libshdr_NAME.so
set -euo pipefail
name=$1; flags=$2; shift 2
P=$(cd "$(dirname "$0")/.." && pwd)/singlehdr-tf2_amd
mkdir -p $P/build/$name
objs=""
for o in $P/build/*.o; do
  b=$(basename $o .o); use=$o
  for f in "$@"; do
    if [ "$(basename $f .hip)" = "$b" ]; then
      use=$P/build/$name/$b.o
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$P/../include -I$P/csrc -Wno-unused-function $flags -x hip -c $P/csrc/$b.hip -o $use
    fi
  done
  objs="$objs $use"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $P/libshdr_$name.so $objs
echo built $P/libshdr_$name.so
