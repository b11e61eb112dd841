"""Per-kernel wait-state / operand-path summary from the counter passes of tools/pmc_passes.sh (one rocprofv3 --pmc pass
per group, gpurun_out/<tag>/pmc_<group>.csv):
    python tools/pmc_wait_states.py gpurun_out/<tag> [name-filter ...] > profiles/<tag>_pmc_wait_states.txt
Columns (all per kernel name = per template instantiation, summed over its launches of the pass):
  mfma    SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES)         share of CU-busy time the matrix pipes execute
  wait    SQ_WAIT_ANY / SQ_WAVE_CYCLES                              waves parked (s_waitcnt / barrier)
  istall  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                         waves ready but stalled at issue (dependency / pipe)
  ilds    SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES                         ... of which on the LDS queue
  active  SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES                       waves issuing
  valu/mf SQ_INSTS_VALU (incl. MFMA) per SQ_INSTS_MFMA             vector instructions per matrix instruction
  lds/mf, vmem/mf                                                    LDS / vector-memory-read instructions per MFMA
  ldsbc   SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                  LDS bank-conflict cycles per LDS-array cycle
  coex    SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES     share of MFMA time overlapped by VALU
  tcp%    TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES               L1 (TCP) miss rate of vector loads
  l2hit   TCC_HIT / (TCC_HIT + TCC_MISS)
  tastall TCP_TCP_TA_DATA_STALL_CYCLES / TCP_GATE_EN1               texture-addresser data stall share (if collected)
Quad-cycle counters (WAVE_CYCLES, WAIT_*, ACTIVE_INST_*) are ratios of like units; MFMA_BUSY and BUSY_CU are cycles."""
import collections, csv, os, re, sys

d = sys.argv[1]
filt = sys.argv[2:]
C = collections.defaultdict(lambda: collections.defaultdict(float))
N = collections.defaultdict(int)
DUR = collections.defaultdict(float)
for g in ('wait', 'inst', 'mem', 'tcp', 'tcc', 'ta'):
    p = os.path.join(d, f'pmc_{g}.csv')
    if not os.path.exists(p):
        continue
    seen = set()
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name']
        C[k][r['Counter_Name']] += float(r['Counter_Value'])
        if g == 'wait' and r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            N[k] += 1
            DUR[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3


def short(k):
    k = k.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', k)[:60]


def ratio(c, a, b, scale=1.0):
    return c[a] / (scale * c[b]) if c.get(b) else float('nan')


rows = []
for k, c in C.items():
    if filt and not any(f in k for f in filt):
        continue
    rows.append((DUR[k], k, c))
rows.sort(key=lambda r: -r[0])
print(f'# {d}: one eager train step pass per counter group (bench.py --no-graph --steps 2 --warmup 1), sums over all launches of a kernel')
print(f'{"us":>8s} {"n":>4s} {"mfma":>5s} {"wait":>5s} {"istall":>6s} {"ilds":>5s} {"active":>6s} {"valu/mf":>7s} {"lds/mf":>6s} {"vmem/mf":>7s} {"ldsbc":>5s} {"coex":>5s} {"tcp%":>5s} {"l2hit":>5s} {"tastl":>5s}  kernel')
for dur, k, c in rows[:70]:
    mf = c.get('SQ_INSTS_MFMA', 0.0)
    f = lambda v: f'{v:5.2f}' if v == v else '    -'
    print(f'{dur:8.1f} {N[k]:4d} {f(ratio(c, "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", 4))} {f(ratio(c, "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"))} '
          f'{f(ratio(c, "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")):>6s} {f(ratio(c, "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"))} '
          f'{f(ratio(c, "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES")):>6s} '
          f'{(c["SQ_INSTS_VALU"] / mf if mf else float("nan")):7.1f} {(c["SQ_INSTS_LDS"] / mf if mf else float("nan")):6.2f} {(c["SQ_INSTS_VMEM_RD"] / mf if mf else float("nan")):7.2f} '
          f'{f(ratio(c, "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"))} {f(ratio(c, "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES"))} '
          f'{f(ratio(c, "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"))} '
          f'{f(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if c.get("TCC_HIT_sum") else float("nan"))} '
          f'{f(ratio(c, "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_GATE_EN1_sum"))}  {short(k)}')
