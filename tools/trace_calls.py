"""Per-dispatch durations of kernels whose name contains a substring, from a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 40
sel = [r for r in rows if pat in r['Kernel_Name']]
for r in sel[-last:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print(f"{r['Kernel_Name'][:60]:60s} grid=({r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) lds={r['LDS_Block_Size']} vgpr={r['VGPR_Count']}+{r['Accum_VGPR_Count']} {d:9.1f} us")
