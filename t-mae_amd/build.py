"""Build libtmae_hip.so (gfx950) in-tree: hipcc per .hip file, then one link.

    python t-mae_amd/build.py [--force] [--verbose]

The shared object lands in t-mae_amd/tmae_amd/lib/ (git-ignored, but it travels with the tree to the
GPU box).  hipcc cross-compiles gfx950 without a GPU.
"""
import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIBDIR = os.path.join(HERE, 'tmae_amd', 'lib')
LIB = os.path.join(LIBDIR, 'libtmae_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
         '-fvisibility=default', '-I', os.path.join(HERE, '..', 'include')]


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src,) + tuple(extra))


def _abi_hash():
    """Fingerprint of include/tmae_hip.h's prototypes (tmae_amd/_abi.py, loaded by path: importing the package would load the
    library this script is about to build)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('_tmae_abi', os.path.join(HERE, 'tmae_amd', '_abi.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.header_hash(open(os.path.join(HERE, '..', 'include', 'tmae_hip.h')).read())


# Per-file flags.  -fno-slp-vectorize: the SLP vectoriser turns adjacent scalar f32 adds / multiplies into v_pk_*_f32, which beside
# MFMAs cost several times the two scalar instructions they replace (MI355X_MICROARCH.md, per-instruction constants); the files
# below were A/B-measured with and without it (profiles/round5_ab_no_slp.txt).  tools/check_mfma_hazards.py compiles with the same.
PER_FILE_FLAGS = {
    'attention_mfma.hip': ['-fno-slp-vectorize'],          # attention family -0.27 ms per step
    # measured and NOT set: token_gemm_wreg.hip (its GELU epilogues run 2-6 % SLOWER unpacked), wgrad.hip, spconv_igemm.hip (neutral)
}


def build(force=False, verbose=False, ab=False):
    """ab=True: the A/B debug build (-DTMAE_AB: environment switches and the retired kernel variants behind them, see
    csrc/common.h) -> t-mae_amd/build_ab/libtmae_ab.so; load it with TMAE_LIB_PATH (profiles/scripts/ab_env.sh)."""
    global OBJ, LIBDIR, LIB
    flags = FLAGS + [f'-DTMAE_ABI_HASH={_abi_hash()}']
    if ab:
        OBJ = os.path.join(HERE, 'build_ab', 'obj')
        LIBDIR = os.path.join(HERE, 'build_ab')
        LIB = os.path.join(LIBDIR, 'libtmae_ab.so')
        flags = flags + ['-DTMAE_AB']
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    hdrs = tuple(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')) + (
        os.path.join(HERE, '..', 'include', 'tmae_hip.h'), os.path.join(HERE, 'tmae_amd', '_abi.py'), os.path.abspath(__file__))
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + '.o')
        if force or _newer(src, obj, hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + flags + PER_FILE_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
        if verbose:
            cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    failed = False
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for src, rc, out in ex.map(cc, jobs):
            if rc != 0 or verbose:
                print(f'--- {os.path.basename(src)} (rc={rc})\n{out}', file=sys.stderr)
            failed |= rc != 0
    if failed:
        raise RuntimeError('hipcc failed')
    objs = [os.path.join(OBJ, s[:-4] + '.o') for s in srcs]
    if jobs or force or not os.path.exists(LIB):
        r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stdout + r.stderr, file=sys.stderr)
            raise RuntimeError('link failed')
    return LIB


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--force', action='store_true')
    ap.add_argument('--verbose', action='store_true')
    ap.add_argument('--ab', action='store_true', help='A/B debug build with -DTMAE_AB into t-mae_amd/build_ab/')
    a = ap.parse_args()
    print(build(a.force, a.verbose, a.ab))
