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
