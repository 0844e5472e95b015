"""A tiny HF-style tokenizer used to pin the collator against the reference's DataCollatorForCausalLM
(the Llama sentencepiece model is not available offline).  Word-level, BOS = 1, EOS = 2, pad = 0; the word
'图' maps to id 30861 like in the Llama vocabulary, the word '▁' to 29871."""
import torch


class _Enc(dict):
    @property
    def data(self):
        return self


class ToyTokenizer:
    pad_token_id, eos_token_id, bos_token_id = 0, 2, 1

    def __init__(self):
        self.padding_side = "right"
        self.vocab = {"图": 30861, "▁": 29871}

    def _encode(self, text):
        ids = [self.bos_token_id]
        for w in text.split():
            if w not in self.vocab:
                self.vocab[w] = 3 + (sum(ord(c) * (i + 1) for i, c in enumerate(w)) % 20000)
            ids.append(self.vocab[w])
        return ids

    def batch_decode(self, ids, skip_special_tokens=True, clean_up_tokenization_spaces=True):
        words = {v: k for k, v in self.vocab.items()}
        special = {self.pad_token_id, self.eos_token_id, self.bos_token_id} if skip_special_tokens else set()
        return [" ".join(words.get(int(t), f"<{int(t)}>") for t in row if int(t) not in special) for row in ids]

    def __call__(self, text, padding=None, truncation=False, max_length=None, return_tensors=None):
        single = isinstance(text, str)
        rows = [self._encode(t) for t in ([text] if single else text)]
        if truncation and max_length:
            rows = [r[:max_length] for r in rows]
        if padding == "max_length":
            width = max_length
        else:
            width = max(len(r) for r in rows)
        ids, mask = [], []
        for r in rows:
            pad = [self.pad_token_id] * (width - len(r))
            if self.padding_side == "left":
                ids.append(pad + r)
                mask.append([0] * len(pad) + [1] * len(r))
            else:
                ids.append(r + pad)
                mask.append([1] * len(r) + [0] * len(pad))
        return _Enc(input_ids=torch.tensor(ids), attention_mask=torch.tensor(mask))


def collator_instances():
    import json
    rep1 = {"image_description": {"x": 1},
            "Sentence 1": {"copied content": "the cat sits", "score": 4, "error type": "correct", "rewritten content": "the cat sits"},
            "Sentence 2": {"copied_content": "on a red mat", "score": 2, "error_type": "image_recognition_error", "rewritten_content": "on a blue mat"},
            "Added": {"rewritten content": "near a window"}}
    rep2 = {"Sentence 1": {"copied content": "two dogs", "score": 1, "error type": "image_recognition_error", "rewritten content": "one dog runs"},
            "Sentence 2": {"copied content": "", "score": 3, "error type": "language_comprehension_error", "rewritten content": ""}}
    img = torch.zeros(3, 4, 4)
    return [
        {"queries": "USER: 图 what is here ? ASSISTANT:", "images": img, "standard_response": "a cat on a mat",
         "original_generate_response": "the cat sits quietly on a red mat today", "AI_pseudo_response": "the cat sits on a blue mat near a window",
         "AI_json_report": json.dumps(rep1)},
        {"queries": "USER: 图 count the animals ASSISTANT:", "images": img + 1, "standard_response": "one dog",
         "original_generate_response": "two dogs", "AI_pseudo_response": "one dog runs", "AI_json_report": json.dumps(rep2)},
    ]


class EosAwareTokenizer(ToyTokenizer):
    """ToyTokenizer that, like the Llama sentencepiece model, turns the literal '</s>' into the single EOS id (also when it is glued
    to a word) - the length bookkeeping of the reference's preprocess_v1 (round = BOS + words, closed by one '</s>' token) relies on it.
    `tokenizer(text).input_ids` and `tokenizer(text)["input_ids"]` both work (plain Python lists for a single string)."""
    model_max_length = 64

    def _encode(self, text):
        ids = [self.bos_token_id]
        for w in text.replace("</s>", " </s> ").split():
            ids.append(self.eos_token_id if w == "</s>" else super()._encode(w)[1])
        return ids

    def __call__(self, text, **kw):
        if isinstance(text, str):
            return _Ids(self._encode(text))
        kw = {k: v for k, v in kw.items() if k in ("padding", "truncation", "max_length", "return_tensors")}
        enc = super().__call__(text, **kw)
        return _Ids(enc["input_ids"], enc["attention_mask"])


class _Ids(dict):
    def __init__(self, ids, mask=None):
        super().__init__(input_ids=ids, attention_mask=mask)
        self.input_ids, self.attention_mask = ids, mask
