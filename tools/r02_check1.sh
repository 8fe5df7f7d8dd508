#!/bin/bash
# round-2 GPU call 1: the whole GPU tier, the headline line, the N = 2 rehearsal
mkdir -p gpurun_out/r02
python -m pytest tests -q -m gpu -x -s $PYTEST_EXTRA > gpurun_out/r02/pytest_gpu.log 2>&1; rc=$?
tail -5 gpurun_out/r02/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 20 --warmup 5 > gpurun_out/r02/bench_n1.json 2> gpurun_out/r02/bench_n1.err || exit 1
cat gpurun_out/r02/bench_n1.json
bash tools/rehearse_multi_gpu.sh > gpurun_out/r02/rehearse.log 2>&1; rc=$?
tail -3 gpurun_out/r02/rehearse.log
exit $rc
