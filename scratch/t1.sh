python -m pytest tests/test_gpu_parity.py -q -x -k "probes" 2>&1 | tail -60
