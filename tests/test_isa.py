"""Properties of the GENERATED CODE in libe3unet.so that the kernels' performance rests on and no runtime test sees:
CPU tier -- hipcc cross-compiles gfx950 without a GPU, and tools/isa_check.py reads the embedded code objects (metadata notes + llvm-objdump).
A ROCm upgrade (or an innocent edit) that undoes one of them fails here instead of costing a few per cent silently.
"""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


@pytest.fixture(scope='module')
def kernels():
    import isa_check
    from elektronn3_amd.build import build
    if not os.path.exists(os.path.join(isa_check.LLVM, 'llvm-objdump')):
        pytest.skip('llvm-objdump of the ROCm toolchain is not installed')
    ks = isa_check.inspect(build())
    assert len(ks) > 300, len(ks)
    return ks


def _family(ks, fam):
    """Kernels whose (mangled) name holds `fam` as a whole identifier."""
    sel = [k for k in ks if re.search(r'\d' + fam + r'(I|E|P|v|\d|$)', k['mangled'])]
    assert sel, f'no kernel named {fam} in the library'
    return sel


def test_every_kernel_is_a_wave64_gfx950_kernel_with_a_sane_budget(kernels):
    for k in kernels:
        m = k['meta']
        assert m.get('vgpr_count', 0) + 0 <= 512 and m.get('max_flat_workgroup_size', 0) <= 1024, k['name']
        assert k['hist'], f'no disassembly for {k["name"]}'


def test_wino4_window_reads_stay_unpaired(kernels):
    """conv_wino4.hip: the window reads are `volatile` 8-byte LDS loads so that hipcc does NOT pair them into ds_read2_b64 -- the slot layout
    is conflict-free for ds_read_b64's 32-lane groups on 64 banks and 2-way conflicted for ds_read2_b64's 16-lane groups (+0.07 ms per step,
    DESIGN.md 3a).  Per 8-channel chunk and wave: 48 window reads, 96 MFMAs."""
    for k in _family(kernels, 'conv3_wino4_kernel'):
        h = k['hist']
        assert h.get('ds_read2_b64', 0) == 0 and h.get('ds_read2st64_b64', 0) == 0, (k['name'], 'hipcc paired the window reads again')
        assert h.get('ds_read_b64', 0) == 192, (k['name'], h.get('ds_read_b64'))      # prologue, two chunk forms, epilogue: 4 x 48
        assert h.get('v_mfma_f32_16x16x4_f32', 0) == 2 * 96, (k['name'], h.get('v_mfma_f32_16x16x4_f32'))      # two chunk forms: a brick's first chunk (zero accumulators) and the others
        assert h.get('buffer_load_dwordx4 lds', 0) == 30, (k['name'], 'the halo planes are staged with LDS-DMA: 3 x 6 pieces in the prologue, 6 per chunk form')


# family -> the one MFMA opcode it is built on (exact fp32: v_mfma_f32_32x32x2_f32 / 16x16x4_f32; the 16-bit path: 32x32x16, compiled for bf16 and f16)
MFMA_OF = {
    'conv3_wino_pkernel': {'v_mfma_f32_32x32x2_f32'}, 'conv3_wino_kernel': {'v_mfma_f32_32x32x2_f32'}, 'conv3_wino4_kernel': {'v_mfma_f32_16x16x4_f32'},
    'wgrad_wino_kernel': {'v_mfma_f32_32x32x2_f32'}, 'wgrad_wino_sk_kernel': {'v_mfma_f32_32x32x2_f32'}, 'conv2_wino_kernel': {'v_mfma_f32_32x32x2_f32'}, 'wgrad_wino2d_kernel': {'v_mfma_f32_32x32x2_f32'},
    'conv_first_mfma_kernel': {'v_mfma_f32_32x32x2_f32'}, 'upconv_gemm_kernel': {'v_mfma_f32_32x32x2_f32'}, 'upconv_fwd_persist_kernel': {'v_mfma_f32_32x32x2_f32'},
    'upconv_wgrad_kernel': {'v_mfma_f32_32x32x2_f32'},
    'conv_b16_pkernel': {'v_mfma_f32_32x32x16_bf16', 'v_mfma_f32_32x32x16_f16'}, 'conv_b16_kernel': {'v_mfma_f32_32x32x16_bf16', 'v_mfma_f32_32x32x16_f16'},
    'wgrad_b16_kernel': {'v_mfma_f32_32x32x16_bf16', 'v_mfma_f32_32x32x16_f16'}, 'wgrad_b16_sk_kernel': {'v_mfma_f32_32x32x16_bf16', 'v_mfma_f32_32x32x16_f16'},
    'upconv_fwd_b16_kernel': {'v_mfma_f32_32x32x16_bf16', 'v_mfma_f32_32x32x16_f16'},
}


@pytest.mark.parametrize('fam', sorted(MFMA_OF))
def test_matrix_kernels_use_the_expected_mfma_opcode(kernels, fam):
    import isa_check
    seen = set()
    for k in _family(kernels, fam):
        ops = set(isa_check.mfma_ops(k['hist']))
        assert len(ops) == 1 and ops <= MFMA_OF[fam], (k['name'], ops)
        seen |= ops
    assert seen == MFMA_OF[fam], (fam, seen)


# The one-wave-per-SIMD kernels hold 512 registers (256 VGPR + 256 AGPR): a spill inside their loops is a scratch round trip that waits for every
# request in flight.  Bounds = what this toolchain produces today, per template form: zero where the hot loop has none; the forms that carry a few
# spilled values keep them OUTSIDE the chunk loop (prologue constants), and the fence keeps that from growing unnoticed.
SPILL_FENCE = {
    # (family, template arguments as mangled) -> max spilled VGPRs
    ('conv3_wino4_kernel', 'ILb0ELb0ELb0ELb0EE'): 0,      # data gradient
    ('conv3_wino4_kernel', 'ILb1ELb0ELb0ELb0EE'): 0,      # eval forward
    ('conv3_wino4_kernel', 'ILb1ELb1ELb0ELb0EE'): 0,      # eval forward + pool
    ('conv3_wino4_kernel', 'ILb1ELb0ELb1ELb0EE'): 0,      # eval forward + head
    ('conv3_wino4_kernel', 'ILb0ELb0ELb0ELb1EE'): 5,      # data gradient + BatchNorm reduce
    ('conv3_wino_pkernel', None): 3,
    ('wgrad_wino_kernel', None): 11,
    ('wgrad_wino_sk_kernel', None): 18,      # (the segment loop's state; the brick loop itself is the one-layer kernel's)
}


def test_spill_fence_of_the_512_register_kernels(kernels):
    checked = 0
    for (fam, targs), bound in SPILL_FENCE.items():
        for k in _family(kernels, fam):
            if targs is not None and (fam + targs) not in k['mangled']:
                continue
            m = k['meta']
            assert m.get('vgpr_spill_count', 0) <= bound, (k['name'], m.get('vgpr_spill_count'), bound)
            if bound == 0:
                assert m.get('private_segment_fixed_size', 0) == 0, (k['name'], 'scratch in a kernel whose loop must not touch it')
                assert not any(op.startswith('scratch_') for op in k['hist']), k['name']
            checked += 1
    assert checked >= 12


def test_the_16bit_conv_kernels_fit_two_waves_per_simd(kernels):
    """conv_b16_pkernel and the 16-bit weight-gradient kernels run two 256-thread workgroups per CU: more than 256 registers would halve their occupancy, and the
    sliding-window weight gradient must not spill (a first form with two copies of the k-step loop spilled 160+ VGPRs)."""
    for k in _family(kernels, 'conv_b16_pkernel'):
        assert k['meta']['vgpr_count'] <= 256 and k['meta'].get('vgpr_spill_count', 0) == 0, (k['name'], k['meta'])
    for fam in ('wgrad_b16_kernel', 'wgrad_b16_sk_kernel'):
        for k in _family(kernels, fam):
            assert k['meta']['vgpr_count'] <= 256, (k['name'], k['meta'])
            if 'Lb1E' in k['mangled']:      # the sliding-window forms
                assert k['meta'].get('vgpr_spill_count', 0) == 0, (k['name'], k['meta'])


def test_no_wide_store_with_scalar_offset_is_followed_by_a_write_of_its_data():
    """Round 5's non-reproducible conv_first_mfma_kernel, root-caused in round 6 (tools/repro_soffset.hip, profiles/r06_soffset_hazard.md): a
    buffer_store of more than 64 bits reads its data registers after issue; hipcc inserts the wait state in front of a VALU overwrite only when the
    store's soffset is an immediate -- with an SGPR soffset (the guides' exemption) it does not, and on gfx950 the overwrite then reaches memory.
    The library must not contain that sequence."""
    import isa_check
    from elektronn3_amd.build import build
    if not os.path.exists(os.path.join(isa_check.LLVM, 'llvm-objdump')):
        pytest.skip('llvm-objdump of the ROCm toolchain is not installed')
    bad = isa_check.wide_store_hazards(build(), window=2)
    assert not bad, '\n'.join(f'{k}: {st}  ->  {ov}' for k, st, ov in bad[:10])
