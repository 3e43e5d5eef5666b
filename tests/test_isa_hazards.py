"""CPU: the gfx950 ISA of every MFMA kernel is free of unpadded MFMA -> VALU / memory hazards on EVERY path (both sides of each
branch), t-mae_amd/tools/check_mfma_hazards.py.  The compiler pads straight-line code but can leave the taken side of a
wave-uniform branch bare; round 4's wrong attention gradients at the temperature clamp were exactly that."""
import importlib.util
import os

from conftest import ROOT


def _tool():
    spec = importlib.util.spec_from_file_location('check_mfma_hazards',
                                                  os.path.join(ROOT, 't-mae_amd', 'tools', 'check_mfma_hazards.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_checker_sees_a_bare_taken_branch():
    """Hand-written ISA: an MFMA in front of a conditional branch whose target reads the result at once (flagged), the same with
    the compiler's padding (clean), and a dependent accumulation through SrcC (interlocked: clean)."""
    t = _tool()
    bad = """
kern:
\tv_mfma_f32_16x16x32_bf16 v[70:73], v[30:33], v[110:113], 0
\ts_cbranch_scc1 .LBB0_2
\tv_mov_b32_e32 v1, 0
\tv_mov_b32_e32 v2, 0
\tv_mov_b32_e32 v3, 0
\tv_mov_b32_e32 v4, 0
\tv_mov_b32_e32 v5, 0
\tv_mov_b32_e32 v6, 0
\tv_mov_b32_e32 v7, 0
.LBB0_2:
\tv_pk_add_f32 v[70:71], v[70:71], v[164:165]
\ts_endpgm
.Lfunc_end0:
"""
    probs = t.check_asm(bad)
    assert len(probs) == 1 and probs[0][5] == 'v_pk_add_f32' and probs[0][7] == 1, probs
    good = bad.replace('.LBB0_2:\n', '.LBB0_2:\n\ts_nop 6\n')
    assert t.check_asm(good) == []
    chain = """
kern:
\tv_mfma_f32_16x16x32_bf16 v[70:73], v[30:33], v[110:113], 0
\tv_mfma_f32_16x16x32_bf16 v[70:73], v[34:37], v[114:117], v[70:73]
\ts_nop 7
\tv_add_f32_e32 v1, v70, v71
\ts_endpgm
.Lfunc_end0:
"""
    assert t.check_asm(chain) == []


def test_no_unpadded_mfma_hazard_in_any_kernel():
    t = _tool()
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 't-mae_amd', 'csrc', '*.hip')))
    bad = t.check_files(files)
    assert not bad, [(os.path.basename(b[0]), b[1][:50], b[3], b[6], b[8]) for b in bad[:10]]
