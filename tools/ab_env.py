import json, os, subprocess, sys
# usage: ab_env.py VAR=val [bench args]   -> alternates unset / set, 3 rounds
var, val = sys.argv[1].split("=")
extra = sys.argv[2:]
for rnd in range(3):
    for on in (0, 1):
        env = dict(os.environ)
        if on: env[var] = val
        out = subprocess.run([sys.executable, "bench.py", "--steps", "100", "--warmup", "20", "--no-cpu-baseline"] + extra, capture_output=True, text=True, env=env)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(("%s=%s" % (var, val)) if on else "default", round(d["ms_per_step"], 4), {k: v for k, v in list(d["kernels"].items())[:4]}, flush=True)
        except Exception:
            print("failed", out.stderr[-500:], flush=True)
