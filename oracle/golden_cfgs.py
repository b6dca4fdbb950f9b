"""Model configurations shared by the golden-fixture generator (oracle/gen_golden.py) and the tests.

Test infrastructure.  Tiny configs keep the kernel-imposed shape rules of the HIP path (head_dim 64 for
the encoder, 128 for the LLM, every GEMM K a multiple of 8) so the same fixtures check the oracle on
CPU and the HIP path on the GPU box.
"""
from oracle.hubert_oracle import HubertCfg
from oracle.llama_oracle import LlamaCfg
from oracle.whisper_oracle import WhisperCfg

LLAMA_ID = "meta-llama/Llama-3.2-3B-Instruct"
MINICHAT_ID = "GeneZC/MiniChat-2-3B"

TINY_HUBERT = HubertCfg(conv_dim=(64,) * 7, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                        intermediate_size=256, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)
WIDE_HUBERT = HubertCfg(num_hidden_layers=2)  # full HuBERT-large width, 2 layers
TINY_LLAMA = LlamaCfg(hidden_size=256, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                      head_dim=128, intermediate_size=512, vocab_size=1000, rope_theta=500000.0,
                      rope_scaling=dict(factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                        original_max_position_embeddings=8192),
                      tie_word_embeddings=True, eos_token_ids=(5, 7), pad_token_id=5)
TINY_MHA = LlamaCfg(hidden_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                    head_dim=128, intermediate_size=384, vocab_size=777, rope_theta=10000.0, rope_scaling=None,
                    tie_word_embeddings=False, eos_token_ids=(2,), pad_token_id=2)
WIDE_LLAMA = LlamaCfg(num_hidden_layers=2, eos_token_ids=(128001, 128008, 128009), pad_token_id=128001,
                      rope_scaling=dict(factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                        original_max_position_embeddings=8192))
# tiny Whisper: 2 s chunks (200 mel frames -> 100 positions), head_dim 64
TINY_WHISPER = WhisperCfg(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256, num_mel_bins=80,
                          max_source_positions=100)
# Whisper-medium width (d_model 1024, 16 heads, FFN 4096, 80 mel bins, 1 500 positions = one 30 s window), 2 layers
WIDE_WHISPER = WhisperCfg(d_model=1024, encoder_layers=2, encoder_attention_heads=16, encoder_ffn_dim=4096, num_mel_bins=80,
                          max_source_positions=1500)
