"""Kernel-name matching shared by the profile summarisers.

rocprofv3 prints a kernel as `void name<arg, arg, ...>(params)`.  The summarisers of rounds 1-4 compared whole strings, so a NEW
template argument (round 4 added DG as the seventh of token_gemm_wreg_kernel) silently emptied a group: the counter pass of the
priced kernel was written as `{}` / 0 and nothing complained.  Here a kernel is matched by its base name plus the template
arguments that identify the instance BY POSITION, missing trailing arguments read as their default `false`, and an empty group
is an error (`require`)."""
import re
import sys


def parse(kernel_name):
    """'void f<a, b<c, d>, e>(args)' -> ('f', ['a', 'b<c, d>', 'e'])"""
    n = re.sub(r'\(.*', '', kernel_name).replace('void ', '').strip()
    m = re.match(r'([^<]+)<(.*)>\s*$', n)
    if not m:
        return n, []
    args, depth, cur = [], 0, ''
    for ch in m.group(2):
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        if ch == ',' and depth == 0:
            args.append(cur.strip())
            cur = ''
        else:
            cur += ch
    args.append(cur.strip())
    return m.group(1).strip(), args


def clean(kernel_name):
    return re.sub(r'\(.*', '', kernel_name).replace('void ', '').strip()


# token_gemm_wreg_kernel<K, NTC, NWC, POS, ACC, GELU2, DG, ...>: flag positions (csrc/token_gemm_wreg.hip)
_TGW_FLAGS = {'POS': 3, 'ACC': 4, 'GELU2': 5, 'DG': 6}


def is_tgw(kernel_name, k=256, ntc=4, nwc=8, **flags):
    """token_gemm_wreg_kernel<k, ntc, nwc, ...> with exactly the named flags set (every other flag, present or not, false)."""
    base, a = parse(kernel_name)
    if base != 'token_gemm_wreg_kernel' or len(a) < 3 or a[:3] != [str(k), str(ntc), str(nwc)]:
        return False
    want = {_TGW_FLAGS[f]: bool(v) for f, v in flags.items()}
    for pos in range(3, max(len(a), 7)):
        have = a[pos] == 'true' if pos < len(a) else False
        if have != want.get(pos, False):
            return False
    return True


def is_tgw_plain(n):
    return is_tgw(n)


def is_tgw_gelu(n):
    return is_tgw(n, GELU2=True)


def prefix(*prefixes):
    return lambda n: any(clean(n).startswith(p) for p in prefixes)


def require(group, matched):
    if not matched:
        sys.exit(f'{sys.argv[0]}: group "{group}" matched no kernel -- a kernel was renamed or its template arguments changed; '
                 f'fix profiles/scripts/_kernels.py instead of committing an empty summary')
    return matched
