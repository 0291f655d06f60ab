"""MI355X-native hot path for lit-llama's quantized decode (drop-in for `lit_llama`'s inference API).

Exports mirror /root/reference lit_llama/__init__.py:3-4; `Tokenizer` is CPU string work outside the hot path
and is not re-implemented (the reference's SentencePiece wrapper can be used unchanged).
"""
from .model import LLaMA, LLaMAConfig, RMSNorm, apply_rope, build_rope_cache  # noqa: F401
from .generate import generate  # noqa: F401
from .utils import EmptyInitOnDevice, quantization  # noqa: F401

__all__ = ["LLaMA", "LLaMAConfig", "RMSNorm", "apply_rope", "build_rope_cache", "generate", "quantization",
           "EmptyInitOnDevice"]
