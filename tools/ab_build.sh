#!/bin/bash
# Development: variant builds of the library for A-B measurements on the GPU box.
#   tools/ab_build.sh name1 "-DE1_GRAB=16" name2 "-DE1_X=1 -DE1_Y=2" ...   ->  lz4_frame_conduit_amd/build/ab/lib_<name>.so
# (built here - hipcc cross-compiles - and shipped with the snapshot; select one with LZ4F_MI355X_LIB=<path>)
set -e
cd "$(dirname "$0")/../lz4_frame_conduit_amd/csrc"
mkdir -p ../build/ab
make -s >/dev/null
pids=()
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function $flags -c -o ../build/ab/engine_$name.o engine.hip &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -Wl,-soname,liblz4f_mi355x.so -o ../build/ab/lib_$name.so ../build/ab/engine_$name.o ../build/pipeline.o ../build/frame_host.o ../build/conduit.o &&
    rm -f ../build/ab/engine_$name.o && echo "built $name ($flags)" ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
