"""vision-longformer_amd: MI355X-native (gfx950) implementation of Vision
Longformer's 2-D sliding-chunk local+global attention, behind the reference's
module API (``Long2DSCSelfAttention(x, nx, ny)``, ATTN_TYPE='longformerhand').

Layout
  csrc/            hand-written HIP kernels + the C ABI (include/vil_attn.h) -> libvilattn.so
  _lib.py          ctypes binding of the C ABI (fails loudly when the library is missing)
  ops.py           torch.autograd.Function around vil_attn_fwd / vil_attn_bwd
  longformer2d.py  drop-in Long2DSCSelfAttention module (same ctor, state-dict keys, mode RNG)
  slidingchunk_2d.py  the reference's operator-level functions (slidingchunk_2d, mask_invalid_locations) on HIP kernels
  msvit.py         host model (MsViT) assembled from stock PyTorch-ROCm blocks + the module above
  engine.py        one-process-per-GPU data-parallel training step (RCCL all-reduce via DDP)
"""
__version__ = "0.2.0"
