#!/bin/bash
# dev: E1 variants built on the box
set -o pipefail
cd lz4_frame_conduit_amd/csrc
for v in "16 4" "16 8" "8 8" "8 16" "16 16"; do
  set -- $v
  touch engine.hip
  make CXXFLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DE1_WAVES=$1 -DE1_GRAB=$2" 2>&1 | grep -E "error" 
  echo "== waves $1 grab $2"
  (cd ../.. && timeout -k 10 100 python tools/quick_bench.py 4096 2>&1 | grep -v amdgpu.ids | tail -2 | head -1)
done
