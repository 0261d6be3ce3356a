"""
CPU check of the built library's gfx950 code objects (tools/isa_audit.py): no packed-FP32 instruction anywhere (a wavefront
resumed after a context save can lose lanes 48-63 of such a result on this platform: csrc/poll.hip),
and no register spills to scratch in any kernel the plan can dispatch.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


@pytest.fixture(scope='module')
def kernels():
    import isa_audit
    from keras_retinanet_3D.backend import hip
    if not os.path.isfile(hip.LIB_PATH):
        hip.build()
    return isa_audit.audit(hip.LIB_PATH)


def test_no_packed_fp32_instruction_in_the_library(kernels):
    assert len(kernels) > 100
    bad = {k: v['packed_fp32'] for k, v in kernels.items() if v.get('packed_fp32')}
    assert not bad, 'packed-FP32 instructions are back (Makefile: NOPK): {}'.format(sorted(bad.items())[:5])


def test_no_kernel_spills_to_scratch(kernels):
    bad = {k: (v.get('private_segment_fixed_size', 0), v.get('vgpr_spill_count', 0)) for k, v in kernels.items()
           if v.get('private_segment_fixed_size', 0) or v.get('vgpr_spill_count', 0)}      # (SGPRs spilled to VGPR lanes use no memory)
    assert not bad, 'kernels with scratch (bytes per lane, VGPRs spilled): {}'.format(sorted(bad.items()))


def test_the_power_bound_x3_loops_accumulate_in_place(kernels):
    """ an accumulating MFMA that writes another register than the one it reads costs a power-bound loop 20 % at the same instruction count
    (HISTORY.md 4.10, profiles/r4/kws_shared_taps.txt): the three-phase x3 loops -- pipelined tiles on pre-split maps, the dual-shape and
    the mixed-height grids -- must not have one.  (The loops the compiler schedules itself have some; they belong to HBM-bound layers.) """
    import re
    pipelined_x3 = re.compile(r'conv_igemm_kernelILi[45]ELi\d+ELi\d+ELi\d+ELi\d+ELi2ELb1ELb1E|conv_igemm_mix_kernelILi[45]E|conv_igemm_dual_kernelILi[45]ELb1E')
    seen = {k: v for k, v in kernels.items() if pipelined_x3.search(k)}
    assert len(seen) >= 20 and all(v.get('mfma', 0) >= 24 for v in seen.values()), sorted(seen)[:3]
    # the headline type: none at all; bf16x3 (same source, another allocation): one tile has 5 of 72 -- bounded, not forbidden
    bad = {k: (v['mfma_out_of_place'], v['mfma']) for k, v in seen.items()
           if v.get('mfma_out_of_place') and ('ILi5E' in k or v['mfma_out_of_place'] * 10 > v['mfma'])}
    assert not bad, bad
