set -e
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 300 python3 tools/soak_indexed.py 100 > gpurun_out/soak.log 2>&1
tail -1 gpurun_out/soak.log
timeout -k 10 600 python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
