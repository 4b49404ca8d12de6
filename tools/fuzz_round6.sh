#!/bin/bash
# round 6's changed kernels under the random-configuration tools, forced onto every grid (fresh seeds): the F(2x2x4) epilogue in store order / three-unit look-ahead,
# the cross-layer stream-K weight gradients (every layer deferred), the 16-bit sliding-window weight gradient
O=gpurun_out/fuzz_r06; mkdir -p $O
E3_WINO4_MIN=1 E3_BNRED_MIN_MB=0 E3_WINO_PERSIST_MIN=1 E3_WGRAD_DEFER_MAX_MB=100000 python tools/fuzz_unet.py 40 701 > $O/unet_forced.log 2>&1; tail -2 $O/unet_forced.log
python tools/fuzz_unet.py 30 703 > $O/unet_default.log 2>&1; tail -2 $O/unet_default.log
E3_WINO4_MIN=1 E3_BNRED_MIN_MB=0 E3_WINO_PERSIST_MIN=1 python tools/fuzz_conv.py 100 705 > $O/conv_forced.log 2>&1; tail -2 $O/conv_forced.log
python tools/fuzz_eval_layouts.py 30 707 > $O/eval_layouts.log 2>&1; tail -2 $O/eval_layouts.log
python tools/fuzz_ops_bf16.py 60 709 > $O/ops_bf16.log 2>&1; tail -2 $O/ops_bf16.log
python tools/fuzz_bf16.py 20 711 > $O/bf16.log 2>&1; tail -3 $O/bf16.log
# (second half of the round: the block-structured brick order of the persistent Winograd kernels and the 16-bit conv's 8 x 2 columns on random shapes, with forced odd blocks as well)
E3_WINO4_MIN=1 E3_WINO_PERSIST_MIN=1 E3_WINO_BLOCK=2,3,1 python tools/fuzz_conv.py 60 713 > $O/conv_block231.log 2>&1; tail -2 $O/conv_block231.log
E3_B16_PERSIST_MIN=1 E3_B16_COL=2,2 python tools/fuzz_ops_bf16.py 40 715 > $O/ops_bf16_col22.log 2>&1; tail -2 $O/ops_bf16_col22.log
