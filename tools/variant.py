"""Build a measurement variant of the library: python tools/variant.py NAME -DSWITCH=VALUE ...  ->  prints the path of
hpmn_amd/lib/variants/libhpmn_NAME.so (objects are cached per source + flags, so a variant recompiles only the files its
flags change... every file sees the flags, so: everything once per distinct flag set, ~15 s).  Run with HPMN_LIB_PATH=<path>."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpmn_amd import build  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "hpmn_amd", "lib", "variants", "libhpmn_%s.so" % name)
print(build.build_library(out=out, flags=flags))
