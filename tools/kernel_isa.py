"""Dump the gfx950 ISA of the kernels of an object / library whose demangled name contains a pattern.
usage: python tools/kernel_isa.py <file.o|.so> <pattern> [outfile]   (no GPU needed)"""
import os, re, struct, subprocess, sys, tempfile
path, pat = sys.argv[1], sys.argv[2]
objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
data = open(path, 'rb').read()
magic = b'__CLANG_OFFLOAD_BUNDLE__'
out = []
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
            asm = subprocess.run([objdump, '-d', '--mcpu=gfx950', f], capture_output=True, text=True, check=True).stdout
            parts = re.split(r'\n[0-9a-f]+ <([^>]+)>:\n', asm)
            names = subprocess.run(['c++filt'], input='\n'.join(parts[1::2]), capture_output=True, text=True).stdout.split('\n')
            for name, body in zip(names, parts[2::2]):
                if pat in name:
                    out.append(f'=== {name}\n' + re.sub(r'\s*//.*', '', body))
text = '\n'.join(out)
if len(sys.argv) > 3:
    open(sys.argv[3], 'w').write(text)
else:
    print(text)
