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
