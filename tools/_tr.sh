set -e
LINKED=1 IX_SCALE=8 timeout -k 10 120 python3 tools/ix_check.py 2048 4 text 2>&1 | grep -o "ok [A-Za-z]*\|'decode': [0-9.]*" | paste - -
IX_SCALE=8 timeout -k 10 120 python3 tools/ix_check.py 2048 7 text 2>&1 | grep -o "ok [A-Za-z]*\|'decode': [0-9.]*" | paste - -
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 600 python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
