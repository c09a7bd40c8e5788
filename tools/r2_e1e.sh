#!/bin/bash
# dev: phase stamps of pass E1 (diagnostic build made on the box; the in-tree library is not touched afterwards by the caller)
cd lz4_frame_conduit_amd/csrc && touch engine.hip && make CXXFLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DE1_DEBUG" 2>&1 | grep error
cd ../.. && timeout -k 10 100 python tools/quick_bench.py 1024 2>&1 | grep -E "E1 wave|parse:|ok " | tail -12
