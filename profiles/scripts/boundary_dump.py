"""Kernel timeline around the optimizer step / start of the next step of a rocprofv3 kernel trace (bench.py --steps 3 --warmup 2):
python3 profiles/scripts/boundary_dump.py <kernel_trace.csv>"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_step_kernel' in r['Kernel_Name']]
i0 = idx[3] if len(idx) > 3 else idx[-1]
prev_end = int(rows[i0 - 8]['End_Timestamp'])
t0 = int(rows[i0 - 8]['Start_Timestamp'])
for r in rows[i0 - 8:i0 + 45]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:60]
    print(f'{(st - t0) / 1e3:9.1f} us  gap {(st - prev_end) / 1e3:7.1f}  dur {(en - st) / 1e3:7.1f}  {n}')
    prev_end = max(prev_end, en)
