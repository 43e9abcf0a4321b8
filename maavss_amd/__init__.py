"""maavss_amd -- MI355X-native (gfx950) implementation of the MAAVSS training hot path.

Everything numerical runs in hand-written HIP kernels behind the C-ABI of include/maavss.h
(libmaavss_hip.so); PyTorch-ROCm only owns device memory, streams and torch.distributed.
"""
from . import _lib  # noqa: F401
from .stft import STFT, calc_hop_size  # noqa: F401
from .avse import AV_Fusion_Model_Frames  # noqa: F401
from .avfm import AV_Fusion_Model  # noqa: F401
from .phasegram import video_phasegram  # noqa: F401
from .trainer import FusedAdam, GradSync, TrainStep, shard_batch  # noqa: F401
from .video_attention import VideoAttention  # noqa: F401
from .checkpoint import latest_file, load_checkpoint, save_checkpoint, save_model  # noqa: F401
from .pipeline import ClipPipeline  # noqa: F401

from . import attn_cache  # noqa: F401


_det_ws = None


def set_deterministic(on=True, workspace_mb=32, stream=None):
    """Process-wide: no f32-atomic accumulation in the Linear / GEMM kernels (bit-identical runs; include/maavss.h).  A device scratch
    of `workspace_mb` MB is handed to the library so that the M = batch Linear forms keep their split over K (partial sums + an ordered
    reduction) instead of streaming the weights through one slice.  The scratch is bound to `stream` (default: the current stream at the first
    call, i.e. the training stream): Linear kernels on other streams take the single-slice path.  Returns the previous setting."""
    global _det_ws
    import torch
    if on and _det_ws is None and torch.cuda.is_available():
        _det_ws = torch.empty(workspace_mb * 262144, device="cuda", dtype=torch.float32)
        st = (stream if stream is not None else torch.cuda.current_stream()).cuda_stream
        _lib.call("maavss_set_deterministic_workspace", _det_ws.data_ptr(), _det_ws.numel() * 4, st)
    return bool(_lib.query("maavss_set_deterministic", int(bool(on))))
