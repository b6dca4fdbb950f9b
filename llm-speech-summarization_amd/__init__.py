"""MI355X-native hot path of the speech-prompted LLM pipeline (HuBERT -> Llama), behind the reference's
Python surface.  See DESIGN.md.  Importing the package does not load the HIP library; the first op does,
and fails loudly if it is missing."""
from ._lib import SpeechLLMError, LIB_PATH, EXPORTS  # noqa: F401

__all__ = ["SpeechLLMError", "LIB_PATH", "EXPORTS"]
