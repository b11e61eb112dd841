"""Per-kernel MFMA-pipe utilisation from one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace`
pass of bench.py (tools/collect_profiles.sh):
    python tools/pmc_mfma_util.py <counter_collection.csv> <label> > profiles/<tag>_pmc_mfma_util_<mode>.txt
util = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES): the share of the time a CU had work during which its four matrix
pipes were executing MFMAs (both counters are summed over CUs / XCDs by rocprofv3, so the ratio is per CU; 4 SIMDs per CU).
Time-weighted means per kernel name; the conv kernels are listed per template instantiation (= per layer shape)."""
import collections, csv, re, sys

path, label = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
mf, bu, n, dur = (collections.defaultdict(float) for _ in range(4))
seen = set()
for r in csv.DictReader(open(path)):
    k = r['Kernel_Name']
    v = float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES':
        mf[k] += v
    elif r['Counter_Name'] == 'SQ_BUSY_CU_CYCLES':
        bu[k] += v
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id'])
        n[k] += 1
        dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3


def short(k):
    k = k.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', k)[:72]


rows = [(mf[k] / (4 * bu[k]) if bu[k] else 0.0, mf[k], bu[k], n[k], dur[k], short(k)) for k in bu if mf[k] > 0]
rows.sort(key=lambda r: -r[4])
print(f'# {label}')
print('# util = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES); duration under the counter pass (slower than un-profiled)')
print(f'{"util":>6s} {"launches":>8s} {"total us":>10s}  kernel')
for u, m, b, c, d, k in rows:
    print(f'{u:6.3f} {int(c):8d} {d:10.1f}  {k}')
fam = lambda pred: (sum(m for _, m, _, _, _, k in rows if pred(k)), sum(b for _, _, b, _, _, k in rows if pred(k)))
for name, pred in (('cconv_mfma_kernel + cconv_mfma16_kernel (forward / data gradient)', lambda k: k.startswith('cconv_mfma')),
                   ('cconv_wgrad_mfma_kernel / cconv_wgrad_x6_kernel (weight gradient)', lambda k: k.startswith('cconv_wgrad_mfma') or k.startswith('cconv_wgrad_x6')),
                   ('cconv_enc0_kernel + cconv_enc0_wgrad_kernel', lambda k: k.startswith('cconv_enc0')),
                   ('all MFMA kernels', lambda k: True)):
    m, b = fam(pred)
    if b:
        print(f'# {name}: util {m / (4 * b):.3f}')
