"""TextCache with the caller's text encoders (FluxPriorReduxPipeline.encode_prompt semantics): CLIP pooler output of
`prompt`, T5 last hidden state of `prompt_2 or prompt`, cached in memory and on disk.  Tiny random transformers models,
stub tokenizers (no vocab files offline)."""
import torch
from transformers import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel

from domain_rag_amd.engine import TextCache, encode_prompt_with


class Tok:
    def __init__(self, max_len, vocab):
        self.model_max_length, self.vocab, self.calls = max_len, vocab, []

    def __call__(self, texts, padding, max_length, truncation, return_tensors):
        self.calls.append((texts[0], max_length))
        ids = [(ord(c) % (self.vocab - 3)) + 2 for c in texts[0]][: max_length - 1] + [1]
        ids += [0] * (max_length - len(ids))
        return type("Enc", (), {"input_ids": torch.tensor([ids])})()


def test_text_cache_uses_given_encoders_and_caches(tmp_path):
    torch.manual_seed(0)
    clip = CLIPTextModel(CLIPTextConfig(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                                        max_position_embeddings=77, projection_dim=32, eos_token_id=1)).eval()
    t5 = T5EncoderModel(T5Config(vocab_size=64, d_model=48, d_kv=8, d_ff=64, num_layers=1, num_heads=2)).eval()
    tok, tok2 = Tok(77, 64), Tok(512, 64)
    cache = TextCache(str(tmp_path), False, 16, 48, 32, "cpu", encoders=(clip, t5, tok, tok2))
    e, p = cache.get("a fish", "")
    assert e.shape == (16, 48) and p.shape == (32,) and e.dtype == torch.bfloat16
    assert tok.calls == [("a fish", 77)] and tok2.calls == [("a fish", 16)]          # prompt_2 = prompt_2 or prompt
    ref_e, ref_p = encode_prompt_with(clip, t5, tok, tok2, "a fish", "", 16)
    assert torch.equal(e, ref_e.bfloat16()) and torch.equal(p, ref_p.bfloat16())
    # second object: served from the cache file, encoders untouched
    n = len(tok.calls)
    cache2 = TextCache(str(tmp_path), False, 16, 48, 32, "cpu", encoders=(clip, t5, tok, tok2))
    e2, p2 = cache2.get("a fish", "")
    assert len(tok.calls) == n and torch.equal(e2, e) and torch.equal(p2, p)
    # no encoders, no file, not synthetic -> loud error
    import pytest
    with pytest.raises(FileNotFoundError):
        TextCache(str(tmp_path), False, 16, 48, 32, "cpu").get("unseen prompt")
