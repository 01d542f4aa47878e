"""Build an experimental variant of libdance_hip.so for same-box A/Bs (tools/ab_variants.sh):
    python tools/build_variant.py <name> <source.hip | all> [-DMACRO=VALUE ...] [--allow-packed]
recompiles ONE translation unit with the extra defines and links it with the in-tree objects of the
others into dance_amd/_variants/lib_<name>.so (git-ignored, travels to the GPU box)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dance_amd import build as B  # noqa: E402

name, src, defs = sys.argv[1], sys.argv[2], [a for a in sys.argv[3:] if a != "--allow-packed"]
B.build()
vdir = os.path.join(B.HERE, "_variants")
os.makedirs(vdir, exist_ok=True)
srcs = [x for x, _ in B.SOURCES] if src == "all" else [src]
objs = []
for s_, extra in B.SOURCES:
    if s_ in srcs:
        obj = os.path.join(vdir, "%s_%s.o" % (name, s_.replace(".hip", "")))
        B._compile(["/opt/rocm/bin/hipcc", "-x", "hip"] + B.COMMON + extra + defs + ["-c", os.path.join(B.CSRC, s_), "-o", obj])
        objs.append(obj)
    else:
        objs.append(os.path.join(B.OBJ, s_.replace(".hip", ".o")))
out = os.path.join(vdir, "lib_%s.so" % name)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", out] + objs)
if "--allow-packed" not in sys.argv:
    B.check_no_packed_fp32(out)
print(out)
