#!/bin/bash
# quick GPU check: parity tests + the three BASELINE timing lines
set -e
python -m pytest tests -q -m gpu -x 2>&1 | tail -5
python bench.py --steps 10 --warmup 3 --no-cpu-baseline
python scratch/fan_model.py
