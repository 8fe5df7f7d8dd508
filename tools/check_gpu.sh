#!/bin/bash
# One GPU call: parity tier, the headline bench line, and one timed pass of the other BASELINE configs.
set -e
python -m pytest tests -q -m gpu -x 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline
python tools/time_configs.py
