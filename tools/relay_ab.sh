#!/bin/bash
# Development (GPU box): tools/text_big_blocks.py (1 GiB of text in 4 MiB independent blocks) over variant builds of the workgroup-per-block decoder
# (tools/ab_build.sh <name> "-DRELAY_W=.. -DRELAY_S=.. -DRL_D=.. -DRL_DPOST=.. -DRL_SLEEP=.."): tools/relay_ab.sh name1 name2 ...
cd "$(dirname "$0")/.."
echo "default: $(timeout -k 10 200 python3 tools/text_big_blocks.py 8 2>&1 | tail -1)"
for v in "$@"; do
  echo "$v: $(LZ4F_MI355X_LIB=$PWD/lz4_frame_conduit_amd/build/ab/lib_$v.so timeout -k 10 200 python3 tools/text_big_blocks.py 8 2>&1 | tail -1)"
done
