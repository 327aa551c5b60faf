"""Developer tool: an A/B build of the library with extra flags on chosen sources, next to the real one.
    python tools/build_variant.py OUT.so pct_knn.hip=-DPCT_ABL_NO_SORT [pct_fit.hip=-DX ...]
    PCT_LIB=$PWD/OUT.so python bench.py --no-cpu-baseline --no-extras
Objects are cached under csrc/.obj, so only the named sources are recompiled."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
out = os.path.abspath(sys.argv[1])
extra = {}
for a in sys.argv[2:]:
    src, flags = a.split("=", 1)
    extra.setdefault(src, []).extend(flags.split())
ge.compile_and_link(out, extra)
print(out)
