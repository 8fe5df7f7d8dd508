"""Headline fan at several sizes (1x .. 16x rays, nstep_max = 400) with the current library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fan_model as fm
cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs/cfg3b_solovev64k_rk4.in")
for scale in (1, 2, 4, 16):
    fm.run(cfg, scale, 400, reps=3)
