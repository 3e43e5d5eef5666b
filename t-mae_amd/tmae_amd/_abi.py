"""Signature fingerprint of the C ABI (include/tmae_hip.h <-> libtmae_hip.so <-> the ctypes table of _lib.py).

One canonical line per exported function, `name:R:ARGS`, with one letter per C type class -- P pointer (incl. hipStream_t),
I int / int32_t, L int64_t, Z size_t, F float, D double -- sorted by name; its FNV-1a hash is compiled into the library from
the HEADER (build.py passes -DTMAE_ABI_HASH, csrc/abi.hip returns it from tmae_abi_hash()) and recomputed by _lib.py from its
TABLE at import.  A library built from another header, or a table row whose argument count / order / width differs from the
prototype, fails the import instead of reaching a kernel launch with shifted arguments (ctypes accepts EXTRA arguments to a
cdecl function and passes them as 32-bit ints; a shifted stream or pointer argument is a host segfault inside the HIP runtime --
gpurun_out/r5u_tests.log of round 4 is what that looks like).  Pure Python, no torch: build.py loads this file by path."""
import re

_CLASSES = (('*', 'P'), ('hipStream_t', 'P'), ('int64_t', 'L'), ('uint64_t', 'L'), ('size_t', 'Z'), ('double', 'D'),
            ('float', 'F'), ('int32_t', 'I'), ('uint32_t', 'I'), ('unsigned', 'I'), ('int', 'I'))


def _type_class(decl):
    d = decl.strip()
    for needle, code in _CLASSES:
        if needle == '*':
            if '*' in d:
                return code
        elif re.search(r'\b' + needle + r'\b', d):
            return code
    raise ValueError(f'tmae_hip.h: cannot classify parameter "{decl}"')


def header_signatures(header_text):
    """{name: (restype code, [argument codes])} of every `tmae_*` prototype in the header text."""
    text = re.sub(r'/\*.*?\*/', ' ', header_text, flags=re.S)
    text = re.sub(r'//[^\n]*', ' ', text)
    out = {}
    for m in re.finditer(r'\b(int|size_t|int64_t|void|float|double)\s+(tmae_\w+)\s*\(([^)]*)\)\s*;', text):
        res, name, params = m.group(1), m.group(2), m.group(3).strip()
        args = [] if params in ('', 'void') else [_type_class(p) for p in params.split(',')]
        out[name] = ('V' if res == 'void' else _type_class(res), args)
    return out


def canonical(sigs):
    return '\n'.join(f'{n}:{sigs[n][0]}:{"".join(sigs[n][1])}' for n in sorted(sigs))


def fnv1a31(text):
    h = 0x811C9DC5
    for b in text.encode():
        h = ((h ^ b) * 0x01000193) & 0xFFFFFFFF
    return h & 0x7FFFFFFF


def header_hash(header_text):
    return fnv1a31(canonical(header_signatures(header_text)))
