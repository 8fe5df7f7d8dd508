import sys, os, shutil, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
orig = os.path.join(root, "rays_amd", "lib", "librays_hip.so")
shutil.copy(orig, "/tmp/orig.so")
lib = sys.argv[1]
scales = [int(x) for x in sys.argv[2].split(",")]
shutil.copy(os.path.join(root, "scratch", "libs", lib), orig)
try:
    for s in scales:
        out = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--fan-scale", str(s), "--nstep-max", "400"], cwd=root, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            print(lib, j["config"]["rays_total"], "%.4g" % j["value"], "%.3f ms" % j["ms_per_step"])
        except Exception as e:
            print(lib, "FAILED", out.stdout[-300:], out.stderr[-500:])
    t = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "tests/test_gpu_edge_and_scale.py", "-m", "gpu", "-q", "-k", "rk4_matches or full_fan_matches_oracle or refill or ragged"], cwd=root, capture_output=True, text=True)
    print(t.stdout.strip().splitlines()[-1])
finally:
    shutil.copy("/tmp/orig.so", orig)
