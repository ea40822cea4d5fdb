"""Build a variant of the gfx950 library with extra compiler flags (A/B runs): python tools/build_variant.py NAME -DFOO=1 ...
-> tools/_libsed_NAME.so; use with `python bench.py --lib tools/_libsed_NAME.so`."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from desed_task_amd import build as B
name, extra = sys.argv[1], sys.argv[2:]
obj_dir = os.path.join(HERE, "_obj_" + name)
os.makedirs(obj_dir, exist_ok=True)
objs = []
only = os.environ.get("ONLY")        # ONLY=sed_mel.hip: the extra flags go to that source alone
def cc(src):
    obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
    ex = extra if (only is None or os.path.basename(src) == only) else []
    subprocess.check_call([B.HIPCC] + B.FLAGS + ex + ["-c", src, "-o", obj])
    return obj
with ThreadPoolExecutor(max_workers=8) as ex:
    objs = list(ex.map(cc, B.sources()))
out = os.path.join(HERE, "_libsed_%s.so" % name)
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
