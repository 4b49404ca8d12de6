#!/bin/bash
# round 6's changed kernels under the random-configuration tools, forced onto every grid (fresh seeds): the F(2x2x4) epilogue in store order / three-unit look-ahead,
# the cross-layer stream-K weight gradients (every layer deferred), the 16-bit sliding-window weight gradient
O=gpurun_out/fuzz_r06; mkdir -p $O
E3_WINO4_MIN=1 E3_BNRED_MIN_MB=0 E3_WINO_PERSIST_MIN=1 E3_WGRAD_DEFER_MAX_MB=100000 python tools/fuzz_unet.py 40 601 > $O/unet_forced.log 2>&1; tail -2 $O/unet_forced.log
python tools/fuzz_unet.py 30 603 > $O/unet_default.log 2>&1; tail -2 $O/unet_default.log
E3_WINO4_MIN=1 E3_BNRED_MIN_MB=0 E3_WINO_PERSIST_MIN=1 python tools/fuzz_conv.py 100 605 > $O/conv_forced.log 2>&1; tail -2 $O/conv_forced.log
python tools/fuzz_eval_layouts.py 30 607 > $O/eval_layouts.log 2>&1; tail -2 $O/eval_layouts.log
python tools/fuzz_ops_bf16.py 60 609 > $O/ops_bf16.log 2>&1; tail -2 $O/ops_bf16.log
python tools/fuzz_bf16.py 20 611 > $O/bf16.log 2>&1; tail -3 $O/bf16.log
