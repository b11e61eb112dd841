"""Count selected instructions per kernel of the built library (llvm-objdump over its gfx950 code objects): IEEE fp32 divisions
(v_div_fmas_f32), LDS crossbar shuffles (ds_bpermute_b32), 64-bit integer division helpers, scratch accesses.
usage: python tools/isa_scan.py [lib.so]   (no GPU needed)"""
import os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'dcs-net_amd', 'lib', 'libdcsnet_hip.so')
objdump, filt = '/opt/rocm/lib/llvm/bin/llvm-objdump', 'c++filt'
data = open(lib, 'rb').read()
magic = b'__CLANG_OFFLOAD_BUNDLE__'
pats = {'div': r'v_div_fmas_f32', 'bperm': r'ds_bpermute_b32', 'scratch': r'scratch_(load|store)', 'rcp64': r'v_rcp_f64|v_div_fmas_f64',
        'exp': r'v_exp_f32', 'mfma': r'v_mfma'}
rows = []
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
            for name, body in zip(parts[1::2], parts[2::2]):
                c = {k: len(re.findall(v, body)) for k, v in pats.items()}
                c['len'] = body.count('\n')
                rows.append((name, c))
names = subprocess.run([filt], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.split('\n')
print(f'{"div":>5} {"bperm":>5} {"scr":>4} {"f64":>4} {"exp":>4} {"mfma":>5} {"instr":>6}  kernel')
for (raw, c), nm in sorted(zip(rows, names), key=lambda t: -(t[0][1]['div'] * 10 + t[0][1]['bperm'])):
    if c['div'] or c['bperm'] or c['scratch'] or c['rcp64']:
        print(f"{c['div']:5d} {c['bperm']:5d} {c['scratch']:4d} {c['rcp64']:4d} {c['exp']:4d} {c['mfma']:5d} {c['len']:6d}  {nm[:110]}")
