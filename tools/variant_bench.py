#!/usr/bin/env python3
"""A/B helper for kernel experiments: run bench.py against alternative builds of the library (libdvs_<name>.so next to the
shipped one, e.g. built with extra -D flags) in the same gpurun call.  The package itself only ever loads libdvs_hip.so;
this tool swaps the name before the first load, per child process.
    gpurun -- 'python tools/variant_bench.py hip stag32 stag96'"""
import json
import subprocess
import sys

CHILD = r'''
import sys, runpy
sys.path.insert(0, ".")
from dags_vae_search_amd import _lib as dl
dl.LIB_NAME = "libdvs_%s.so"
sys.argv = ["bench.py", "--steps", "100", "--warmup", "20", "--no-cpu-baseline"] + %r
runpy.run_path("bench.py", run_name="__main__")
'''
extra = []
names = []
for a in sys.argv[1:]:                   # library names first, then bench.py's own arguments from the first "--..." on
    (extra if extra or a.startswith("--") else names).append(a)
for name in names:
    out = subprocess.run([sys.executable, "-c", CHILD % (name, extra)], capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(name, round(d["ms_per_step"], 4), {k: v for k, v in list(d["kernels"].items())[:9]}, flush=True)
    except Exception:
        print(name, "failed", out.stderr[-800:], flush=True)
