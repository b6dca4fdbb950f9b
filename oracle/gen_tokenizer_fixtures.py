"""Generates the tokenizer fixtures under tests/golden/tokenizers/ (test infrastructure; run once in the build container).

The reference loads its tokenizers from the hub (ref:inference.py:32-37, ref:trainer.py:50-55): Llama-3's tiktoken-style
byte-level BPE and MiniChat's SentencePiece model.  Neither vocabulary is available offline, so these fixtures are
tokenizers of the SAME KIND, trained here on a toy corpus with the libraries' own trainers, small enough to commit:

* `Llama-3.2-3B-Instruct/`: tokenizer.json — byte-level BPE behind Llama-3's pre-tokenizer split pattern, the Llama-3 header /
  turn special tokens as added special tokens, a template post-processor that prepends <|begin_of_text|>; tokenizer_config.json
  as the hub repo carries it (PreTrainedTokenizerFast; `use_fast=False` falls back to it, as for the real checkpoint).
* `MiniChat-2-3B/`: tokenizer.model — a SentencePiece BPE model with byte fallback (<unk>=0, <s>=1, </s>=2) + LlamaTokenizer config.

The directory names are the hub ids' basenames, which is what `utils.prompt_template` matches a local path on.
`tokenizer_ids.json` records, from the tokenizers loaded through AutoTokenizer in this container, the ids of the reference's
four template strings and of a few prompts; tests/test_host_cpu.py replays them.  What carries over to the real vocabularies
is the STRUCTURE — `system`, `user`, `assistant` and "\n\n" are single tokens in Llama-3's vocabulary as they are here — so the
Llama-3 prefix is 9 ids (BOS first) and the suffix 6 (5 after the reference's `[:, 1:]`), i.e. a 10 s utterance (123 audio
embeddings) gives a 137-token prompt.
"""
import json
import os
import random
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden", "tokenizers")
sys.path.insert(0, REPO)

LLAMA3_SPLIT = (r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+")
LLAMA3_SPECIALS = ["<|begin_of_text|>", "<|end_of_text|>", "<|start_header_id|>", "<|end_header_id|>", "<|eot_id|>", "<|eom_id|>",
                   "<|python_tag|>", "<|finetune_right_pad_id|>"]
WORDS = ["system", "user", "assistant", "the", "speech", "summary", "please", "summarize", "audio", "what", "is", "said", "in", "this",
         "clip", "hello", "world", "a", "of", "and", "to", "User", "Assistant"]


def llama3(path):
    from tokenizers import Regex, Tokenizer, decoders, models, pre_tokenizers, processors, trainers
    random.seed(0)
    corpus = [" ".join(random.choice(WORDS) for _ in range(8)) + ".\n\n" + random.choice(WORDS) + "\n\n" for _ in range(400)]
    corpus += ["system\n\n", "user\n\n", "assistant\n\n"] * 50
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(LLAMA3_SPLIT), behavior="isolated", invert=False),
                                                 pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    tok.decoder = decoders.ByteLevel()
    tok.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=480, special_tokens=[], show_progress=False,
                                                        initial_alphabet=pre_tokenizers.ByteLevel.alphabet()))
    tok.add_special_tokens(LLAMA3_SPECIALS)
    bos = tok.token_to_id("<|begin_of_text|>")
    tok.post_processor = processors.TemplateProcessing(single="<|begin_of_text|> $A", pair="<|begin_of_text|> $A <|begin_of_text|> $B",
                                                       special_tokens=[("<|begin_of_text|>", bos)])
    os.makedirs(path, exist_ok=True)
    tok.save(os.path.join(path, "tokenizer.json"))
    json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "bos_token": "<|begin_of_text|>", "eos_token": "<|eot_id|>",
               "model_input_names": ["input_ids", "attention_mask"], "clean_up_tokenization_spaces": True, "model_max_length": 131072},
              open(os.path.join(path, "tokenizer_config.json"), "w"), indent=1)
    json.dump({"bos_token": "<|begin_of_text|>", "eos_token": "<|eot_id|>"}, open(os.path.join(path, "special_tokens_map.json"), "w"))


def minichat(path):
    import sentencepiece as spm
    random.seed(1)
    os.makedirs(path, exist_ok=True)
    corpus = os.path.join(path, "corpus.txt")
    with open(corpus, "w") as f:
        for _ in range(600):
            f.write(" ".join(random.choice(WORDS) for _ in range(8)) + ". [|User|] " + random.choice(WORDS) + " [|Assistant|]\n")
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=os.path.join(path, "tokenizer"), vocab_size=400, model_type="bpe", byte_fallback=True,
                                   unk_id=0, bos_id=1, eos_id=2, pad_id=-1, character_coverage=1.0, normalization_rule_name="identity",
                                   add_dummy_prefix=True, split_digits=True, remove_extra_whitespaces=False, minloglevel=2)
    os.remove(corpus)
    os.remove(os.path.join(path, "tokenizer.vocab"))
    json.dump({"tokenizer_class": "LlamaTokenizer", "bos_token": "<s>", "eos_token": "</s>", "unk_token": "<unk>", "add_bos_token": True,
               "add_eos_token": False, "clean_up_tokenization_spaces": False, "legacy": True, "model_max_length": 4096, "sp_model_kwargs": {}},
              open(os.path.join(path, "tokenizer_config.json"), "w"), indent=1)


def record(paths):
    import importlib
    from transformers import AutoTokenizer
    utils = importlib.import_module("llm-speech-summarization_amd.utils")
    rec = {}
    for name, path in paths.items():
        t = AutoTokenizer.from_pretrained(path, use_fast=False, padding_side="left")      # ref:inference.py:32-37
        t.pad_token = t.eos_token
        prefix, suffix = utils.prompt_template(path)
        texts = {"prefix": prefix, "suffix": suffix, "text_prompt": f"{prefix} hello world{suffix} ",      # ref:inference.py:78 (spaces kept)
                 "additional_text_prompt": "please summarize this clip", "plain": "what is said in this audio"}
        rec[name] = {"vocab_size": len(t), "bos_token_id": t.bos_token_id, "eos_token_id": t.eos_token_id, "pad_token_id": t.pad_token_id,
                     "strings": texts, "ids": {k: t(v, return_tensors="pt").input_ids[0].tolist() for k, v in texts.items()}}
        ids = rec[name]["ids"]["plain"]
        rec[name]["decoded_plain_skip_special"] = t.batch_decode([ids + [t.eos_token_id]], skip_special_tokens=True, clean_up_tokenization_spaces=True)[0]
    json.dump(rec, open(os.path.join(OUT, "tokenizer_ids.json"), "w"), indent=1)
    return rec


if __name__ == "__main__":
    paths = {"llama3": os.path.join(OUT, "Llama-3.2-3B-Instruct"), "minichat": os.path.join(OUT, "MiniChat-2-3B")}
    llama3(paths["llama3"])
    minichat(paths["minichat"])
    r = record(paths)
    for k, v in r.items():
        print(k, v["vocab_size"], {n: len(i) for n, i in v["ids"].items()}, repr(v["decoded_plain_skip_special"]))
