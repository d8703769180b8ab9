"""A deterministic stand-in for the InternLM2 SentencePiece tokenizer (``tokenizer.model`` ships with checkpoints, none is
available offline).  Character level: the chat template's special strings map to single ids at the reference's positions
(``synth.special_ids``: <|im_end|> = V-11 ... <IMG_CONTEXT> = V-7, i.e. 92542 ... 92546 for V = 92553), ``<s>`` = 1 is
prepended, every other character c becomes 3 + ord(c) % n_plain.  It offers the small part of the HF tokenizer API the
reference's prompt code and ``chat`` methods touch (dataset.py:595-682, modeling_internvl_chat.py:533-636).  Test
infrastructure only: used by tests/ and by tests/golden/make_host_golden.py (which feeds it to the reference's code)."""
from __future__ import annotations

from typing import List, Sequence, Union

import torch


class _Encoding(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class StubTokenizer:
    def __init__(self, vocab_size: int = 92553, model_max_length: int = 4096):
        self.vocab_size = vocab_size
        self.special = {"<|im_end|>": vocab_size - 11, "<|im_start|>": vocab_size - 10, "<img>": vocab_size - 9,
                        "</img>": vocab_size - 8, "<IMG_CONTEXT>": vocab_size - 7, "<s>": 1, "</s>": 2}
        self._by_len = sorted(self.special, key=len, reverse=True)
        self._special_ids = set(self.special.values())
        self.n_plain = max(16, min(vocab_size - 20, 60000) - 3)
        self.pad_token_id = 2            # InternLM2: pad = eos = </s> (internvl_chat_eval2/config.json:38,72)
        self.eos_token_id = 2
        self.bos_token_id = 1
        self.unk_token_id = 0
        self.padding_side = "right"
        self.model_max_length = model_max_length

    # ---- encoding ------------------------------------------------------------------------------------------
    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        ids = [self.bos_token_id] if add_special_tokens else []
        i = 0
        while i < len(text):
            for s in self._by_len:
                if text.startswith(s, i):
                    ids.append(self.special[s])
                    i += len(s)
                    break
            else:
                ids.append(3 + ord(text[i]) % self.n_plain)
                i += 1
        return ids

    def convert_tokens_to_ids(self, token: Union[str, Sequence[str]]):
        if isinstance(token, str):
            return self.special[token] if token in self.special else self.encode(token, add_special_tokens=False)[0]
        return [self.convert_tokens_to_ids(t) for t in token]

    def __call__(self, text, return_tensors=None, padding=False, max_length=None, truncation=False, **kw):
        single = isinstance(text, str)
        rows = [self.encode(t) for t in ([text] if single else list(text))]
        if truncation and max_length:
            rows = [r[:max_length] for r in rows]
        if padding == "max_length":
            width = max_length or self.model_max_length
        elif padding:
            width = max(len(r) for r in rows)
        else:
            width = None
        mask = [[1] * len(r) for r in rows]
        if width is not None:
            for r, m in zip(rows, mask):
                pad = width - len(r)
                if self.padding_side == "left":
                    r[:0] = [self.pad_token_id] * pad
                    m[:0] = [0] * pad
                else:
                    r += [self.pad_token_id] * pad
                    m += [0] * pad
        if return_tensors == "pt":
            return _Encoding(input_ids=torch.tensor(rows, dtype=torch.long), attention_mask=torch.tensor(mask, dtype=torch.long))
        if single:
            return _Encoding(input_ids=rows[0], attention_mask=mask[0])
        return _Encoding(input_ids=rows, attention_mask=mask)

    # ---- decoding ------------------------------------------------------------------------------------------
    def decode(self, ids, skip_special_tokens: bool = False) -> str:
        inv = {v: k for k, v in self.special.items()}
        out = []
        for t in (ids.tolist() if hasattr(ids, "tolist") else ids):
            t = int(t)
            if t in inv:
                if not skip_special_tokens:
                    out.append(inv[t])
            elif t >= 3:
                out.append(chr(t - 3) if 32 <= t - 3 < 0x2FFFF else "?")
        return "".join(out)

    def batch_decode(self, seqs, skip_special_tokens: bool = False) -> List[str]:
        return [self.decode(s, skip_special_tokens) for s in seqs]
