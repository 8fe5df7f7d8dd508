set -e
python -m pytest tests -q -m gpu -x 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline
python scratch/sg_time.py
