"""Per-kernel means of the rocprofv3 --pmc CSVs written by attn_pmc.sh: one table, kernels x counters
(the last 20 dispatches of every kernel = the timed probe launches).  usage: pmc_kernels.py <dir>"""
import collections, csv, glob, os, re, sys
root = sys.argv[1]
want = ('win_attn_bwd_mfma_kernel', 'win_attn_fwd_mfma_kernel', 'token_gemm_res_kernel', 'token_gemm_wreg_kernel', 'wgrad256_kernel', 'wgrad_reduce')
tab = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(root, 'g*', '**', '*counter_collection.csv'), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
        if not any(w in k for w in want):
            continue
        per[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in per.items():
        for c, v in cs.items():
            v = v[-20:]
            tab[k][c] = sum(v) / len(v)
cols = sorted({c for v in tab.values() for c in v})
print('| kernel | ' + ' | '.join(cols) + ' |')
print('|---|' + '---:|' * len(cols))
for k in sorted(tab):
    print(f'| `{k}` | ' + ' | '.join(f'{tab[k].get(c, float("nan")):.4g}' for c in cols) + ' |')
