#!/bin/bash
# one-off: how often does the process abort at exit after the native-ring tests, per teardown mode?
mkdir -p gpurun_out/r2g
for mode in destroy keep finalize; do
  for k in 1 2 3 4; do
    PICLES_RING_TEARDOWN=$mode timeout -k 10 120 python -X faulthandler -m pytest tests/test_gpu_native_ring.py -q -m gpu -p no:cacheprovider > gpurun_out/r2g/$mode.$k.log 2>&1
    echo "$mode $k rc=$? $(grep -c 'double free\|Aborted\|Fatal Python' gpurun_out/r2g/$mode.$k.log)"
  done
done
