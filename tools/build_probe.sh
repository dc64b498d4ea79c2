#!/bin/bash
# Build a what-if variant of libmrn_hip.so: one source (default conv_x3.hip) recompiled with -D<flag>, every other object reused.
# usage: bash tools/build_probe.sh MRN_PROBE_NO_MFMA [rnn.hip]  ->  tools/probe/libmrn_MRN_PROBE_NO_MFMA.so  (select it with MRN_LIB_PATH)
set -e
flag=$1
src=${2:-conv_x3.hip}
stem=${src%.hip}
cd "$(dirname "$0")/.."
mkdir -p tools/probe
python -m mrn_amd.build > /dev/null
/opt/rocm/bin/hipcc -x hip -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -fno-gpu-rdc -Wno-unused-result -D$flag -I mrn_amd/csrc \
  $EXTRA -c mrn_amd/csrc/$src -o tools/probe/${stem}_$flag.o
objs=$(ls mrn_amd/csrc/build/*.o | grep -v $src.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/probe/libmrn_$flag.so $objs tools/probe/${stem}_$flag.o
echo tools/probe/libmrn_$flag.so
