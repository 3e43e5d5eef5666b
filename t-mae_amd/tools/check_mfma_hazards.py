"""Static check of gfx950 ISA for MFMA -> VALU / memory read-after-write hazards that the compiler left unpadded.

    python t-mae_amd/tools/check_mfma_hazards.py [file.hip | file.s ...]      (default: every t-mae_amd/csrc/*.hip)

Why this exists.  On CDNA3/4 the result registers of an MFMA are NOT interlocked against a following VALU, LDS, VMEM or FLAT
instruction that reads or overwrites them (nor against a following MFMA that reads them as SrcA / SrcB): software has to put the
wait states in between (CDNA3 ISA guide, "Manually Inserted Wait States"; LLVM's GCNHazardRecognizer::checkMAIVALUHazards).  The
compiler pads straight-line code correctly, but its backwards search over predecessor blocks marks blocks as visited ACROSS paths:
when the join block behind a wave-uniform `if` is reached first through the (long) fall-through block, the (short) path of the
TAKEN branch is never examined, and an MFMA issued right in front of `s_cbranch` meets its first consumer behind the branch with
no wait state at all.  The consumer then reads the register's PREVIOUS content.  That is the cause of round 4's "same arithmetic on
paper, wrong gradients" in the attention backward's second pass: restricting the absent-key mask to the last key
tile put a uniform branch between the dP MFMA and `ds = p * (dP - D)`; on the taken path `dP` was still the exp() argument of the
previous statement -- -inf for an absent query, hence 0 * -inf = NaN in dK and garbage in dQ, while dV (from P alone) stayed right.

What is checked: for every v_mfma, every path of instructions that follows it (both sides of conditional branches) until the
required number of wait states has passed; an instruction on such a path that names a VGPR of the MFMA's destination is reported
(a dependent MFMA is checked for SrcA / SrcB only: SrcC and back-to-back accumulation are interlocked).  A taken branch is counted
as ONE wait state, like the compiler does.  Required wait states per MFMA shape: what the compiler pads with in straight-line
code (need_states)."""
import glob
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def need_states(op):
    """Wait states the compiler itself puts between an MFMA and a dependent VALU in straight-line gfx950 code (calibrated with
    one-line kernels: `s_nop 7` behind v_mfma_f32_16x16x16_bf16 and v_mfma_f32_16x16x32_bf16, `s_nop 11` behind the 32x32
    shapes); unknown shapes get the guide's maximum."""
    m = re.match(r'v_mfma_\w+?_(\d+)x(\d+)x(\d+)', op)
    if not m:
        return 19
    mm = int(m.group(1))
    return {4: 5, 16: 8, 32: 12}.get(mm, 19)


_VREG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')


def vregs(text):
    out = set()
    for a, b, c in _VREG.findall(text):
        if c:
            out.add(int(c))
        else:
            out.update(range(int(a), int(b) + 1))
    return out


def parse_functions(asm_text):
    """{function: [(label or None, mnemonic, operand string)]}"""
    funcs, cur, name, pending = {}, None, None, None
    for line in asm_text.splitlines():
        s = line.split(';')[0].rstrip()
        if not s:
            continue
        m = re.match(r'^(_Z\w+|[A-Za-z_]\w*):\s*$', s)
        if m and not s.startswith('.'):
            name, cur = m.group(1), []
            funcs[name] = cur
            pending = None
            continue
        if cur is None:
            continue
        m = re.match(r'^(\.L\w+):\s*$', s)
        if m:
            if m.group(1).startswith('.Lfunc_end'):
                cur, name = None, None
            else:
                pending = m.group(1)
            continue
        if s.startswith('\t') and not s.strip().startswith('.'):
            parts = s.strip().split(None, 1)
            cur.append((pending, parts[0], parts[1] if len(parts) > 1 else ''))
            pending = None
    return funcs


def check_function(name, ins):
    labels = {lab: i for i, (lab, _, _) in enumerate(ins) if lab}
    problems = []
    for i, (_, op, args) in enumerate(ins):
        if not op.startswith('v_mfma'):
            continue
        dst = vregs(args.split(',')[0])
        need = need_states(op)
        stack, seen = [(i + 1, 0)], set()
        while stack:
            j, waited = stack.pop()
            while j < len(ins) and waited < need:
                if (j, waited) in seen:
                    break
                seen.add((j, waited))
                _, o2, a2 = ins[j]
                if o2 == 's_endpgm':
                    break
                if o2.startswith('v_mfma'):
                    ops = [x.strip() for x in a2.split(',')]
                    touched = vregs(','.join(ops[1:3])) & dst          # SrcA / SrcB only
                elif o2.startswith(('s_', ';')):
                    touched = set()
                else:
                    touched = vregs(a2) & dst
                if touched:
                    problems.append((name, i, op, args, j, o2, a2, waited, need))
                    break
                if o2 == 's_nop':
                    waited += int(a2.strip() or 0) + 1
                else:
                    waited += 1
                if o2 == 's_branch':
                    j = labels.get(a2.strip(), len(ins))
                    continue
                if o2.startswith('s_cbranch'):
                    t = labels.get(a2.strip())
                    if t is not None:
                        stack.append((t, waited))
                j += 1
    return problems


def check_asm(asm_text):
    out = []
    for name, ins in parse_functions(asm_text).items():
        out += check_function(name, ins)
    return out


def _per_file_flags(hip_file):
    """the flags build.py compiles this file with beyond the common ones (the ISA that is checked must be the ISA that ships)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('_tmae_build', os.path.join(PKG, 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return list(mod.PER_FILE_FLAGS.get(os.path.basename(hip_file), []))


def compile_to_asm(hip_file, out_file):
    cmd = [HIPCC, '-O3', '--offload-arch=gfx950', '-std=c++17', '-DTMAE_ABI_HASH=0', '-I', os.path.join(PKG, '..', 'include'),
           '-S', '--cuda-device-only'] + _per_file_flags(hip_file) + [hip_file, '-o', out_file]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])


def check_files(files, verbose=False):
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            if f.endswith('.s'):
                text = open(f).read()
            else:
                if 'mfma' not in open(f).read():
                    continue
                s = os.path.join(tmp, os.path.basename(f) + '.s')
                compile_to_asm(f, s)
                text = open(s).read()
            probs = check_asm(text)
            if verbose:
                n = sum(1 for fn in parse_functions(text).values() for _, o, _ in fn if o.startswith('v_mfma'))
                print(f'{os.path.basename(f)}: {n} MFMAs, {len(probs)} unpadded hazards')
            bad += [(f,) + p for p in probs]
    return bad


def main(argv):
    files = argv or sorted(glob.glob(os.path.join(PKG, 'csrc', '*.hip')))
    bad = check_files(files, verbose=True)
    for f, name, i, op, args, j, o2, a2, waited, need in bad:
        print(f'HAZARD {os.path.basename(f)} {name[:60]}: #{i} {op} {args.split(",")[0]} -> #{j} {o2} {a2}  '
              f'({waited} wait states, {need} needed)')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
