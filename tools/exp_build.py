#!/usr/bin/env python3
"""Diagnostic variants of the library: the shipped objects with ONE source recompiled under extra -D flags.
    python tools/exp_build.py <name> <source.hip> -DFLAG[=v] [...]   ->  dcs-net_amd/lib/exp/libdcsnet_hip_<name>.so
Run a tool against it with DCS_LIB_PATH=dcs-net_amd/lib/exp/libdcsnet_hip_<name>.so (timing probes give wrong results by design)."""
import importlib.util, os, subprocess, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'dcs-net_amd')
EXP = os.path.join(PKG, 'lib', 'exp')
name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
spec = importlib.util.spec_from_file_location('dcsnet_build', os.path.join(PKG, 'build.py'))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
b.build()
os.makedirs(EXP, exist_ok=True)
hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
objs = []
for s in b._sources():
    for variant in ([''] + (['_h'] if s in b.ACT_SOURCES else [])):
        o = os.path.join(b.OBJ, s[:-4] + variant + '.o')
        if s == src:
            o2 = os.path.join(EXP, f'{s[:-4]}{variant}_{name}.o')
            extra = b._mfma_source_flags().get(s, []) + (['-DDCS_ACT_BF16'] if variant else [])
            subprocess.run([hipcc] + b.FLAGS + extra + flags + ['-c', os.path.join(b.CSRC, s), '-o', o2], check=True)
            o = o2
        objs.append(o)
lib = os.path.join(EXP, f'libdcsnet_hip_{name}.so')
subprocess.run([hipcc, '-shared', '-fPIC', f'--offload-arch={b.ARCH}', '-o', lib] + objs, check=True)
print('built', lib)
