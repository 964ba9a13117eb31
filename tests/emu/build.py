"""TEST INFRASTRUCTURE: build tests/emu/_build/libdvs_emu.so — the product's kernel + C-ABI sources compiled
for the host against the lock-step emulator (hip_emu.h).  Used only by the `not gpu` tests."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "dags_vae_search_amd", "csrc")
OUT = os.path.join(HERE, "_build")
CXX = os.environ.get("DVS_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
SOURCES = ["k_forward.hip", "k_heads.hip", "k_backward.hip", "k_backward_heads.hip", "k_optim.hip", "k_wide_fwd.hip", "k_wide_bwd.hip", "k_decode.hip", "k_bic.hip", "dvs_api.hip"]


def build(force=False, opt="-O2"):
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, "libdvs_emu.so")
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + \
        [os.path.join(HERE, "hip_emu.h"), os.path.join(HERE, "hip_emu.cpp"), os.path.join(REPO, "include", "dvs.h")]
    if not force and os.path.exists(lib) and all(os.path.getmtime(lib) > os.path.getmtime(d) for d in deps):
        return lib
    flags = [opt, "-g", "-std=c++17", "-fPIC", "-fno-omit-frame-pointer", "-Wno-unused-value", "-Wno-unused-function",
             "-ffp-contract=off", "-pthread"]
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [CXX, *flags, "-x", "c++", "-include", os.path.join(HERE, "hip_emu.h"), "-I", CSRC, "-c", s, "-o", o]
        procs.append((cmd, subprocess.Popen(cmd)))
    o = os.path.join(OUT, "hip_emu.o")
    objs.append(o)
    cmd = [CXX, *flags, "-c", os.path.join(HERE, "hip_emu.cpp"), "-o", o]
    procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("emu build failed: " + " ".join(cmd))
    subprocess.check_call([CXX, "-shared", "-pthread", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
