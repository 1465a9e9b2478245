"""Host-side mirrors of the reference's hot-path modules (same class names / parameters / signatures)."""
from .attention import (BasicTransformerBlock, CrossAttention, FeedForward, GEGLU,  # noqa: F401
                        MemoryEfficientCrossAttention, MemoryEfficientInterViewAttentionTwo,
                        MemoryEfficientIntraViewAttention, SpatialTemporalTransformer)
from .controlmodel import ControlNet3D, ControlledUNetModel3D  # noqa: F401
from .openaimodel import (Downsample, ResBlock3D, TimestepBlock, TimestepEmbedSequential,  # noqa: F401
                          UNetModel3D, Upsample)
from .wrappers import IdentityWrapper, OpenAIWrapper, OpenAIWrapperControlLDM3D  # noqa: F401
