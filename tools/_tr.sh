set -e
LZ4F_MI355X_TRACE_ALWAYS=1 timeout -k 10 240 python3 tools/soak_indexed.py 60 > gpurun_out/soak_tr.log 2>&1
tail -1 gpurun_out/soak_tr.log
for f in "64 4 text" "256 4 text"; do LINKED=1 IX_SCALE=8 timeout -k 10 120 python3 tools/ix_check.py $f 2>&1 | grep -o "ok [A-Za-z]*\|'decode': [0-9.]*" | paste - -; done
