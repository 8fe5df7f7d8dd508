python scratch/fan_model.py
python - <<'PY'
import sys,os
sys.path.insert(0,os.getcwd())
import scratch.fan_model as fm
fm.run("configs/cfg3b_solovev64k_rk4.in", 16, 400, reps=3)
fm.run("configs/cfg2_solovev1024_rk4.in", 1, None, reps=10)
PY
python scratch/sg_time.py
