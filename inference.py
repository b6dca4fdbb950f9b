#!/usr/bin/env python3
"""Drop-in entry point with the reference's CLI (ref:inference.py:140-178):

    python inference.py -c config/llama3_hubert.yaml -g 0 -p audio_encoder.pt -a utterance.wav

and the reference's import path (`from inference import LLMSpeechTextInference`).  The implementation is
the HIP-backed class in `llm-speech-summarization_amd/inference.py`.
"""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
_impl = importlib.import_module("llm-speech-summarization_amd.inference")
LLMSpeechTextInference = _impl.LLMSpeechTextInference


def load_audio_16k(path):
    """librosa.load(path, sr=16000) equivalent on scipy (librosa is not a dependency): mono float32 @ 16 kHz."""
    import numpy as np
    from scipy.io import wavfile
    from scipy.signal import resample_poly
    sr, x = wavfile.read(path)
    x = x.astype(np.float32) / (np.iinfo(x.dtype).max + 1.0) if np.issubdtype(x.dtype, np.integer) else x.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1)
    if sr != 16000:
        g = np.gcd(sr, 16000)
        x = resample_poly(x, 16000 // g, sr // g).astype(np.float32)
    return x, 16000


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('-c', '--config', type=str, help="yaml file for configuration")
    parser.add_argument('-g', '--gpu_idx', type=int, default=0, help="index of home GPU device")
    parser.add_argument('-p', '--audio_encoder_checkpoint', type=str, help="path to audio encoder checkpoint")
    parser.add_argument('-a', '--audio_file', type=str, required=True,
                        help="audio file containing speech utterance to be used in prompt")
    args = parser.parse_args()
    import torch
    torch.cuda.set_device(args.gpu_idx)      # libspeechllm launches on the current HIP device / stream
    config = importlib.import_module("llm-speech-summarization_amd.config").load_config(args.config)
    dtype = torch.float32 if str(config.get("runtime", {}).get("dtype", "bf16")) == "fp32" else torch.bfloat16
    llm_inferencer = LLMSpeechTextInference(config=config, audio_encoder_checkpoint=args.audio_encoder_checkpoint,
                                            device=torch.device(f"cuda:{args.gpu_idx}"), dtype=dtype)
    audio, sr = load_audio_16k(args.audio_file)
    print("LLM Response:\n")
    print(llm_inferencer.generate_audio_response(audio, max_new_tokens=512))
