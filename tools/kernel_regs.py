"""Register / LDS / scratch use of the kernels of an object or library whose demangled name contains a pattern (no GPU needed).
usage: python tools/kernel_regs.py <file.o|.so> <pattern>"""
import os, re, struct, subprocess, sys, tempfile
path, pat = sys.argv[1], sys.argv[2]
readelf = '/opt/rocm/lib/llvm/bin/llvm-readelf'
data = open(path, 'rb').read()
magic = b'__CLANG_OFFLOAD_BUNDLE__'
with tempfile.TemporaryDirectory() as tmp:
    n = 0
    for m in re.finditer(magic, data):
        p = m.start(); q = p + len(magic)
        cnt = struct.unpack_from('<Q', data, q)[0]; q += 8
        for _ in range(cnt):
            off, size, tl = struct.unpack_from('<QQQ', data, q); q += 24
            triple = data[q:q + tl].decode(); q += tl
            if 'gfx950' not in triple or size == 0:
                continue
            f = os.path.join(tmp, f'co{n}.o'); n += 1
            open(f, 'wb').write(data[p + off:p + off + size])
            notes = subprocess.run([readelf, '--notes', f], capture_output=True, text=True, check=True).stdout
            for blk in notes.split('- .agpr_count:')[1:]:
                g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
                name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
                if pat in name:
                    agpr = re.match(r'\s*(\d+)', blk).group(1)
                    print(f"vgpr {g('vgpr_count'):>4} agpr {agpr:>4} sgpr {g('sgpr_count'):>4} lds {g('group_segment_fixed_size'):>6} "
                          f"scratch {g('private_segment_fixed_size'):>5} wg {g('max_flat_workgroup_size'):>5}  {name[:150]}")
