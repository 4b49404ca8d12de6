"""elektronn3_amd -- MI355X (gfx950) implementation of elektronn3's 3D U-Net hot path.

Host-side mirror of the reference interface for that path only:

* :class:`elektronn3_amd.unet.UNet`            <- ``elektronn3.models.unet.UNet``
* :class:`elektronn3_amd.resunet.UNet`         <- ``elektronn3.models.resunet.UNet`` (the same network built from residual ConvBlocks)
* :class:`elektronn3_amd.inference.Predictor`  <- ``elektronn3.inference.Predictor`` / ``tiled_apply``
* :class:`elektronn3_amd.dataparallel.GradSync` (one process per GPU, RCCL all-reduce of the flat gradient)

The arithmetic lives in ``libe3unet.so`` (hand-written HIP, ``csrc/``) behind the C ABI in ``include/e3unet.h``.
There is no CPU fallback: CPU tensors raise.
"""
__version__ = '0.1.0'
