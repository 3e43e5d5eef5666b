"""Longest kernels and longest idle gaps of a whole rocprofv3 kernel trace (to find one-off stalls): python3 hiccup_dump.py trace.csv"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
def nm(r): return re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:70]
print('longest kernels:')
for r in sorted(rows, key=lambda r: int(r['Start_Timestamp']) - int(r['End_Timestamp']))[:12]:
    print(f"  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:8.2f} ms at {(int(r['Start_Timestamp']) - t0) / 1e6:9.1f} ms  {nm(r)}")
gaps, end, last = [], int(rows[0]['End_Timestamp']), rows[0]
for r in rows[1:]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if st > end:
        gaps.append((st - end, end, nm(last), nm(r)))
    if en > end:
        end, last = en, r
print('longest gaps:')
for g, at, a, b in sorted(gaps, reverse=True)[:14]:
    print(f'  {g / 1e6:8.2f} ms at {(at - t0) / 1e6:9.1f} ms  {a} -> {b}')
