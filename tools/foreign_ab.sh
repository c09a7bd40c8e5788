#!/bin/bash
# Development (GPU box): the bench's foreign leg (liblz4's bytes, bare frame) per library variant:  tools/foreign_ab.sh base seg16 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for n in "$@"; do
  if [ "$n" = base ]; then unset LZ4F_MI355X_LIB; else export LZ4F_MI355X_LIB=$R/lz4_frame_conduit_amd/build/ab/lib_$n.so; fi
  timeout -k 10 300 python3 $R/bench.py --legs foreign --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['foreign_frame']; o=f['own_frame_minus_trailer']; print('$n', 'foreign %.3f ms (walk %.3f, decode %.3f) ok=%s; own frame without trailer %.3f' % (f['ms'], f['walk_ms'], f['decode_ms'], f['roundtrip_verified'], o['ms']))"
done
