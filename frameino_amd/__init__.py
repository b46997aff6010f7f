"""frameino_amd -- MI355X-native (gfx950) implementation of FrameINO's denoising hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all hot-path
arithmetic runs in hand-written HIP kernels behind the C ABI of include/frameino_hip.h
(frameino_amd/lib/libframeino_hip.so).  There is no CPU / eager fallback: importing
`frameino_amd.ops` raises if the library is missing.
"""
__version__ = "0.1.0"
