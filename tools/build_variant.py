"""Build an A/B variant of libt2v_hip.so with extra -D flags:  python tools/build_variant.py <tag> -DFOO=0 ...
-> tools/variants/libt2v_hip_<tag>.so (select with T2V_LIB_PATH; git-ignored, and never beside the
product library — a timing-only variant may compute wrong results)."""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

tag, flags = sys.argv[1], sys.argv[2:]
only = None          # --only=a.hip,b.hip: compile just these with the flags, take every other object from the product build (build/*.o)
for f in list(flags):
    if f.startswith("--only="):
        only = f[len("--only="):].split(",")
        flags.remove(f)
objdir = os.path.join(ge.PKG, "build", tag)
os.makedirs(objdir, exist_ok=True)
procs = []
for src in ge.SOURCES:
    obj = os.path.join(objdir, src.replace(".hip", ".o"))
    if only is not None and src not in only:
        base = os.path.join(ge.PKG, "build", src.replace(".hip", ".o"))
        assert os.path.exists(base), f"{base}: build the product library first"
        procs.append((base, None))
        continue
    cmd = [ge._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c"] + ge.EXTRA_FLAGS.get(src, []) + flags + \
          [os.path.join(ge.CSRC, src), "-o", obj]
    procs.append((obj, subprocess.Popen(cmd, cwd=ROOT)))
for obj, pr in procs:
    assert pr is None or pr.wait() == 0, obj
os.makedirs(os.path.join(ROOT, "tools", "variants"), exist_ok=True)
out = os.path.join(ROOT, "tools", "variants", f"libt2v_hip_{tag}.so")
subprocess.run([ge._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [o for o, _ in procs] + ["-ldl"], check=True)
print(out)
