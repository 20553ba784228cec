#!/usr/bin/env python
"""A/B timing builds: libkpal_hip.so variants with one KPAL_AB_* / KPAL_QUAD_* macro each, cross-compiled HERE into
build/variants/ (they travel to the GPU box with the snapshot) and timed THERE by tools/ab_run.sh through
KPAL_HIP_LIBRARY.  Variants may count wrongly on purpose (a path compiled out): only their kernel times are read.

    python tools/ab_build.py NAME=-DMACRO[,-DMACRO2] ...        e.g.  hist_nomerge=-DKPAL_AB_HIST_NO_MERGE
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

OUT = os.path.join(ROOT, 'build', 'variants')
os.makedirs(OUT, exist_ok=True)
UNITS = sorted(f[:-4] for f in os.listdir(g.CSRC) if f.endswith('.hip'))
# only these units see the macros (default: the ones that include quad_kernels.hpp; --units=kpal_vec[,...] as the first
# argument names others); the rest are taken from the regular build
QUAD_UNITS = ('kpal_quads', 'kpal_quads2')


def build(spec):
    name, flags = spec.split('=', 1)
    flags = flags.split(',')
    objs = []
    for u in UNITS:
        if u in QUAD_UNITS:
            obj = os.path.join(OUT, '%s_%s.o' % (name, u))
            subprocess.check_call(['hipcc'] + g.HIPCC_FLAGS + flags + ['-c', os.path.join(g.CSRC, u + '.hip'), '-o', obj], cwd=ROOT)
        else:
            obj = os.path.join(g.OBJ, u + '.o')
        objs.append(obj)
    lib = os.path.join(OUT, 'lib_%s.so' % name)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs + ['-ldl'], cwd=ROOT)
    return lib


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1].startswith('--units='):
        QUAD_UNITS = tuple(sys.argv.pop(1)[len('--units='):].split(','))
    g.build()
    with ThreadPoolExecutor(max_workers=4) as pool:
        for lib in pool.map(build, sys.argv[1:]):
            print(lib)
