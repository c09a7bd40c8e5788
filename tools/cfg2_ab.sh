#!/bin/bash
# Development (GPU box): BASELINE configs[1] (bench.py --legs cfg2) per library variant:  tools/cfg2_ab.sh base m1 m3 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for n in "$@"; do
  if [ "$n" = base ]; then unset LZ4F_MI355X_LIB; else export LZ4F_MI355X_LIB=$R/lz4_frame_conduit_amd/build/ab/lib_$n.so; fi
  timeout -k 10 300 python3 $R/bench.py --legs cfg2 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['cfg2']; print('$n', 'cfg2 decompress %.3f ms (walk %.3f, decode %.3f) ok=%s' % (c['decompress_ms'], c['walk_ms'], c['decode_ms'], c['roundtrip_verified']))"
done
