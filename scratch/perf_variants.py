import sys, os, shutil, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libdir = os.path.join(root, "rays_amd", "lib")
orig = os.path.join(libdir, "librays_hip.so")
shutil.copy(orig, "/tmp/orig.so")
for name in ("librays_hip_exact.so", "librays_ffpcontractfast.so", "librays_ffpcontractfastfreciprocalmath.so"):
    shutil.copy(os.path.join(root, "scratch", "libs", name), orig)
    out = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"], cwd=root, capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print(name, "value %.4g" % j["value"], "ms %.3f" % j["ms_per_step"], "steps", j["config"]["recorded_steps_per_pass"])
    except Exception as e:
        print(name, "FAILED", out.stdout[-300:], out.stderr[-300:])
    # parity of this variant vs golden
    t = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-q", "-k", "rk4_matches or full_fan_matches_oracle", "-s"], cwd=root, capture_output=True, text=True)
    print("\n".join(l for l in t.stdout.splitlines() if "rel err" in l or "passed" in l or "failed" in l or "bitwise" in l or "max abs" in l))
shutil.copy("/tmp/orig.so", orig)
