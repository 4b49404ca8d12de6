"""Every kernel-selecting environment switch (README.md "Switches for A/B runs") under the parity tests: the switches are read once per
process, so each group runs the per-op and whole-network tests in a child process with the group's environment.  (only the
defaults ran in the driver's GPU tier; "suite green under every switch" was a builder claim.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FP32_TESTS = ['tests/test_ops_gpu.py', 'tests/test_unet_gpu.py', '-k',
              'conv3 or convT or train_step_matches_reference or eval_forward or full_size_properties or forward_with_loss']
# (the two largest per-op cases -- 9 s each, most of it the fp64 reference -- run under the default switches only: no switch below selects by size above them)
B16_TESTS = ['tests/test_bf16_gpu.py', 'tests/test_f16_gpu.py', '-k', 'not full_size and not 64-64-128-64-32 and not 30-125-100-32-64']
GROUPS = {
    'direct_kernels': (dict(E3_CONV_NO_WINO='1', E3_WGRAD_NO_WINO='1', E3_CONV_NO_WINO2D='1', E3_WGRAD_NO_WINO2D='1'), FP32_TESTS),
    'unfused_unbatched': (dict(E3_WINO_NO_PERSIST='1', E3_NO_SPLITK='1', E3_NO_REDUCE_BATCH='1', E3_NO_FIRST_FUSE='1', E3_UPCONV_NO_GEMM='1', E3_WINO_NO_TR='1', E3_WINO_NO_POOL='1', E3_WINO_NO_HEAD='1', E3_WINO_BOX_ALIGNED='1', E3_FIRST_WGRAD_PLANAR_VALU='1', E3_FIRST_NO_MFMA='1',
                               E3_NO_LOSS_BWD='1', E3_WGRAD_NO_DEFER='1'), FP32_TESTS),
    # (E3_WGRAD_DEFER_MAX_MB: EVERY Winograd weight gradient in the one stream-K launch at the end of the backward, not only the layers up to 80 MB)
    # (E3_WINO_BLOCK=kw,kh,kd: a forced block shape of the persistent kernels' logical brick order, here 2 x 8 x 1 bricks wherever the counts divide; the default
    # rule runs in the other groups, the plain order in 'plain_rows')
    'persistent_everywhere': (dict(E3_WINO_PERSIST_MIN='1', E3_UPCONV_NO_PERSIST='1', E3_CONV_NO_V3='1', E3_ATT_VALU='1', E3_WGRAD_DEFER_MAX_MB='100000', E3_WINO_BLOCK='1,3,0'), FP32_TESTS),
    'b16_alternatives': (dict(E3_B16_NO_SPLITK='1', E3_B16_UP_GENERIC='1', E3_B16_BD='2', E3_B16_TW='16', E3_B16_COT='1'), B16_TESTS),
    # the round-4 persistent kernels (conv_first_mfma_kernel, conv_first_b16_pkernel, conv_b16_pkernel) at EVERY size they can take: small and ragged grids,
    # workgroups without a single item, statistics records of empty workgroups (E3_B16_COL: brick columns of 2 x 4 bricks instead of the default 8 x 2)
    'persistent_kernels_on_small_grids': (dict(E3_FIRST_MFMA_MIN='1', E3_B16_FIRST_PERSIST_MIN='1', E3_B16_PERSIST_MIN='1', E3_B16_BD='4', E3_B16_TW='32', E3_B16_COT='1', E3_B16_COL='1,2',
                                               E3_B16_NO_SPLITK='1'),
                                          ['tests/test_ops_gpu.py', 'tests/test_unet_gpu.py', 'tests/test_bf16_gpu.py', 'tests/test_f16_gpu.py', 'tests/test_predictor.py', '-k',
                                           '(conv3 or first_conv or train_step_matches_reference or eval_forward or forward_with_loss or bf16 or f16 or fixture or pipelined) '
                                           'and not 64-64-128-64-32 and not 30-125-100-32-64']),
    'b16_no_persistent_conv': (dict(E3_B16_NO_PERSIST='1', E3_B16_FIRST_NO_PERSIST='1'), ['tests/test_bf16_gpu.py', '-k', 'full_size or fixture']),
    'b16_on_fp32_kernels_plain_predictor': (dict(E3_NO_BF16='1', E3_PREDICTOR_NO_PIPELINE='1'),
                                            ['tests/test_predictor.py', 'tests/test_bf16_gpu.py', '-k', 'predictor or fixture or autocast']),
    # data-parallel: overlapped bucket with a CU reserve as GradSync's default; Predictor: the runtime's pageable copies; elementwise passes: non-temporal
    # loads from 1 MB on; weights packed again for every tile  (E3_STAGE_BATCH is a compile-time constant of conv_mfma.hip, not an environment switch)
    'dp_overlap_pageable_copies': (dict(E3_DP_OVERLAP='1', E3_DP_CU_RESERVE='8', E3_PREDICTOR_NO_PINNED='1', E3_EW_NT_MB='1', E3_NO_PACK_REUSE='1'),
                                   ['tests/test_dataparallel_gpu.py', 'tests/test_predictor.py', 'tests/test_unet_gpu.py', '-k',
                                    'two_rank or pipelined or needed_region or in_place or train_step_matches_reference']),
    # F(2x2x4) Winograd tiles (conv_wino4.hip; by default the eval-mode forward and the data gradients of grids with >= 256 bricks): OFF everywhere ...
    'wino4_off': (dict(E3_WINO4='0', E3_NO_BNRED_FUSE='1'), ['tests/test_ops_gpu.py', 'tests/test_unet_gpu.py', 'tests/test_predictor.py', '-k',
                                       'conv3 or train_step_matches_reference or eval_forward or full_size_properties or forward_with_loss or forward_roi or needed_region or full_size_cfg2 or predictor']),
    # ... and on every grid that has a brick (ragged grids, workgroups without a brick, one brick per workgroup): data gradients -- with the REDUCE pass of the
    # BatchNorm backward inside them wherever a grid tiles (E3_BNRED_MIN_MB=0; default: from 32 MB tensors on, the separate pass runs in the default suite and in
    # 'wino4_off') --, the folded eval epilogue with the fused pool / head, the needed-region forward, the in-place tiles.  (Round 6: this group took over
    # 'bnred_in_the_data_gradient', which differed from it in that one flag.)
    'wino4_on_every_grid': (dict(E3_WINO4_MIN='1', E3_BNRED_MIN_MB='0'),
                            ['tests/test_ops_gpu.py', 'tests/test_unet_gpu.py', 'tests/test_predictor.py', '-k',
                             'conv3 or train_step_matches_reference or eval_forward or full_size_properties or forward_with_loss or forward_roi or needed_region or head_in_the_last or pool_in_the_conv or predictor or full_size_cfg2 or digest']),
    # ... with every tensor in plain [voxel][C] rows: the data gradients' input (E3_NO_CHUNKED=1; the chunked form is the default wherever the F(2x2x4) data gradient
    # and the Winograd weight gradient both read the tensor), the inference forwards' conv1 -> conv2 tensors and concat buffers (E3_NO_CHUNKED_FWD=1), and the
    # encoder's skip activations stored in full although the decoder has a needed region (E3_NO_STORE_BOX=1).  The groups above and the defaults run the chunked
    # forms.  (Round 6: 'wino4_plain_rows' + 'forward_plain_rows' in one child process -- the two sets of flags touch disjoint launches.)
    'plain_rows': (dict(E3_WINO4_MIN='1', E3_BNRED_MIN_MB='0', E3_NO_CHUNKED='1', E3_NO_CHUNKED_FWD='1', E3_NO_STORE_BOX='1', E3_WINO_BLOCK='0'),
                   ['tests/test_unet_gpu.py', 'tests/test_predictor.py', '-k',
                    'train_step_matches_reference or full_size_cfg2 or full_size_properties or forward_with_loss or digest or eval_forward or forward_roi or needed_region or head_in_the_last or pool_in_the_conv or predictor or cfg5']),
    # ... and for the TRAINING forward with its statistics as well (E3_WINO4=2; not a default: DESIGN.md section 3a) -- per-op parity and the property tests
    'wino4_training_forward': (dict(E3_WINO4='2', E3_WINO4_MIN='1'), ['tests/test_ops_gpu.py', 'tests/test_unet_gpu.py', '-k', 'conv3 or full_size_properties or forward_with_loss or eval_forward']),
}


@pytest.mark.parametrize('group', list(GROUPS))
def test_parity_suite_under_switch_group(group):
    env_extra, tests = GROUPS[group]
    tests = tests[:-1] + [f'({tests[-1]}) and not bench_volume']      # (the full 512x2048x2048 Predictor run stays in the default process: `-k predictor` matches the module name)
    env = dict(os.environ, **env_extra)
    env['E3_MARGINS_DIR'] = os.path.join(ROOT, 'gpurun_out', 'switch_groups', group)      # (the default run's parity-margin tables are not overwritten by a variant's)
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider', *tests], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    tail = (r.stdout or '')[-3000:] + (r.stderr or '')[-1500:]
    assert r.returncode == 0, f'{group} {env_extra}:\n{tail}'
    assert ' passed' in r.stdout and 'no tests ran' not in r.stdout, tail
