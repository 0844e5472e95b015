"""Host-side data layout of the DPO stage: HF-dataset rows -> fixed-shape tensors.

Mirrors utils/data_utils_dpo.py:32-285 (DataCollatorForCausalLM and its helpers).  This is the layout contract of
SURVEY.md §3.5: queries left-padded to query_len with the "图" token (id 30861) replaced by IMAGE_TOKEN_INDEX;
responses right-padded to response_len, first pad replaced by EOS; with `detailed_report` the original / AI-pseudo
responses are re-assembled sentence by sentence from the GPT-4V JSON report so that per-sentence `score`
(4->1.0, 3->1.5, 2->2.0, 1->2.5) and `error type` (image_recognition_error->3.0, else 1.0) become per-token float
weights (0.0 on padding; the EOS cell of the AI response inherits the previous weight).  Any exception falls back to
plain tokenisation with all-zero weights, exactly like the reference.

The tokenizer is whatever HF-style callable the caller passes (Llama sentencepiece in production); nothing here
touches the GPU.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from typing import Any, Dict, List

import torch

from .dims import IMAGE_TOKEN_INDEX

IMAGE_PLACEHOLDER_ID = 30861      # "图" in the Llama vocabulary (data_utils_dpo.py:119-121)
SCORE_WEIGHT = {1: 2.5, 2: 2.0, 3: 1.5, 4: 1.0}
ERROR_TYPE_WEIGHT = {"image_recognition_error": 3.0, "correct": 1.0, "language_comprehension_error": 1.0}


def pad_and_stack(rows: List[torch.Tensor], pad_value, max_length=None) -> torch.Tensor:
    if max_length is None:
        max_length = max(t.size(0) for t in rows)
    return torch.stack([torch.cat([t, torch.full((max_length - t.size(0),), pad_value, dtype=t.dtype)]) for t in rows])


def complete_copied_content(original: str, pieces: List[str]) -> List[str]:
    """Stretch every 'copied content' snippet so that consecutive snippets tile the original response again
    (text the annotator skipped is attached to the following snippet; the tail to the last one).  If any snippet
    cannot be found in order the list is returned untouched."""
    fixed, rest = [], original
    for piece in pieces:
        t = piece.strip()
        if not t:
            fixed.append("")
            continue
        pos = rest.find(t)
        if pos == -1:
            return pieces
        fixed.append(rest[: pos + len(t)].strip(" "))
        rest = rest[pos + len(t):]
    if fixed and rest.strip():
        fixed[-1] += rest.strip()
    return fixed


def add_eos(t, pad_id: int, eos_id: int):
    """First pad cell of every right-padded row becomes EOS (rows without padding stay as they are)."""
    if hasattr(t, "data") and not isinstance(t, torch.Tensor):
        t = t.data
    if isinstance(t, dict):
        for k, v in t.items():
            t[k] = add_eos(v, pad_id, eos_id)
        return t
    if not isinstance(t, torch.Tensor):
        raise ValueError("Unsupported type for `tensor`")
    for row in t:
        idx = (row == pad_id).nonzero(as_tuple=True)[0]
        if idx.numel() > 0:
            row[idx[0]] = eos_id
    return t


def pad_eos(response: torch.Tensor, score: torch.Tensor, eos_id: int) -> torch.Tensor:
    """The EOS cell takes the weight of the token before it (1 if that weight is 0)."""
    for row_r, row_s in zip(response, score):
        idx = (row_r == eos_id).nonzero(as_tuple=True)[0]
        if idx.numel() > 0:
            prev = row_s[idx[0] - 1]
            row_s[idx[0]] = prev if prev != 0 else 1
    return score


@dataclass
class DataCollatorForCausalLM:
    tokenizer: Any
    query_len: int = 128
    response_len: int = 896
    detailed_report: bool = False

    def _fixed(self, texts: List[str], length: int):
        return self.tokenizer(texts, padding="max_length", truncation=True, max_length=length, return_tensors="pt")

    def _ids(self, text: str) -> torch.Tensor:
        return self.tokenizer(text, return_tensors="pt")["input_ids"]

    def _plain(self, texts: List[str]):
        tok = self.tokenizer
        return add_eos(self._fixed(texts, self.response_len), tok.pad_token_id, tok.eos_token_id)

    def __call__(self, instances: List[Dict]) -> Dict[str, torch.Tensor]:
        tok = self.tokenizer
        get = lambda k: [inst[k] for inst in instances]
        originals, pseudos = get("original_generate_response"), get("AI_pseudo_response")
        tok.padding_side = "left"
        q = self._fixed(get("queries"), self.query_len)
        q["input_ids"][q["input_ids"] == IMAGE_PLACEHOLDER_ID] = IMAGE_TOKEN_INDEX
        tok.padding_side = "right"
        std = self._plain(get("standard_response"))
        batch = {"queries": q["input_ids"], "queries_attention_mask": q["attention_mask"],
                 "standard_response": std["input_ids"], "standard_response_attention_mask": std["attention_mask"]}

        def plain_pair(with_zero_weights: bool):
            o, a = self._plain(originals), self._plain(pseudos)
            batch.update({"original_generate_response": o["input_ids"], "original_generate_response_attention_mask": o["attention_mask"],
                          "AI_pseudo_response": a["input_ids"], "AI_pseudo_response_attention_mask": a["attention_mask"]})
            if with_zero_weights:
                for k, src in (("original_generate_response", o), ("AI_pseudo_response", a)):
                    batch[k + "_scores"] = torch.zeros_like(src["input_ids"])
                    batch[k + "_image_relations"] = torch.zeros_like(src["input_ids"])

        if not self.detailed_report:
            plain_pair(False)
        else:
            # the reports are parsed BEFORE the try (data_utils_dpo.py:116): a malformed AI_json_report raises instead of silently
            # zero-weighting the whole batch; only failures of the alignment logic fall back (data_utils_dpo.py:259-278)
            reports = [json.loads(r) for r in get("AI_json_report")]
            try:
                batch.update(self._from_reports(reports, originals))
            except Exception as e:
                print(e)
                plain_pair(True)
        images = get("images")
        if all(x is not None and x.shape == images[0].shape for x in images):
            batch["images"] = torch.stack(images)
        else:
            batch["images"] = images
        return batch

    def _from_reports(self, reports: List[dict], originals: List[str]) -> Dict[str, torch.Tensor]:
        tok = self.tokenizer
        field = lambda sent, a, b, default="": sent.get(a, sent.get(b, default))
        reports = [{k: v for k, v in rep.items() if k not in ("image_description", "image description")} for rep in reports]
        for rep, orig in zip(reports, originals):      # re-align the snippets with the original response
            keys = list(rep.keys())
            fixed = complete_copied_content(orig, [field(rep[k], "copied content", "copied_content") or "" for k in keys])
            for k, text in zip(keys, fixed):
                if text:
                    if "copied content" in rep[k]:
                        rep[k]["copied content"] = text
                    elif "copied_content" in rep[k]:
                        rep[k]["copied_content"] = text
        cols = {n: [] for n in ("o_ids", "a_ids", "o_sc", "a_sc", "o_rel", "a_rel")}
        for rep in reports:
            row = {n: [] for n in cols}
            for count, (key, sent) in enumerate(rep.items()):
                rewritten = self._ids(field(sent, "rewritten content", "rewritten_content"))
                rewritten = None if rewritten.size(1) <= 1 else (rewritten[:, 1:] if count != 0 else rewritten)
                copied = None
                if key != "Added":
                    copied = self._ids(field(sent, "copied content", "copied_content"))
                    if copied.size(1) <= 1:
                        copied = None
                    else:
                        copied = copied[:, 1:] if count != 0 else copied
                        if copied[:, 0] == torch.tensor(29871):      # leading '' piece of the sentencepiece vocabulary
                            copied = copied[:, 1:]
                    sw = SCORE_WEIGHT.get(sent.get("score", 4), 1.0)
                    rw = ERROR_TYPE_WEIGHT.get(field(sent, "error type", "error_type", "correct"), 1.0)
                    if copied is not None:
                        row["o_ids"].append(copied)
                        row["o_sc"].append(torch.ones_like(copied) * sw)
                        row["o_rel"].append(torch.ones_like(copied) * rw)
                    if rewritten is not None:
                        row["a_sc"].append(torch.ones_like(rewritten) * sw)
                        row["a_rel"].append(torch.ones_like(rewritten) * rw)
                elif rewritten is not None:
                    row["a_sc"].append(torch.ones_like(rewritten))
                    row["a_rel"].append(torch.ones_like(rewritten))
                if rewritten is not None:
                    row["a_ids"].append(rewritten)
            for n in cols:
                cols[n].append(torch.cat(row[n], dim=1)[0])
        T, pad, eos = self.response_len, tok.pad_token_id, tok.eos_token_id
        o_ids = add_eos(pad_and_stack(cols["o_ids"], pad, T), pad, eos)
        a_ids = add_eos(pad_and_stack(cols["a_ids"], pad, T), pad, eos)
        # the original response's EOS cell keeps weight 0 (reference note: "We DONOT ADD EOS TOKEN TO ORIGINAL ...")
        return {"original_generate_response": o_ids, "original_generate_response_attention_mask": o_ids != pad,
                "AI_pseudo_response": a_ids, "AI_pseudo_response_attention_mask": a_ids != pad,
                "original_generate_response_scores": pad_and_stack(cols["o_sc"], 0.0, T),
                "AI_pseudo_response_scores": pad_eos(a_ids, pad_and_stack(cols["a_sc"], 0.0, T), eos),
                "original_generate_response_image_relations": pad_and_stack(cols["o_rel"], 0.0, T),
                "AI_pseudo_response_image_relations": pad_eos(a_ids, pad_and_stack(cols["a_rel"], 0.0, T), eos)}


QUERY_TEMPLATE_HEAD = ("<s> A chat between a curious user and an artificial intelligence assistant. The assistant gives helpful, "
                       "detailed, and polite answers to the user's questions. USER: ")
QUERY_TEMPLATE_TAIL = " ASSISTANT: "
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def format_query(question: str) -> str:
    """Prompt template of DPO_Dataset.__getitem__ (data_utils_dpo.py:292-293,344; Quirk Q11: literal '<s>' plus the
    tokenizer's own BOS): the image placeholder '<image>' becomes '图 ', which the collator maps to IMAGE_TOKEN_INDEX."""
    return QUERY_TEMPLATE_HEAD + question.replace("<image>", "图 ") + QUERY_TEMPLATE_TAIL


def preprocess_image(pil_img, size: int = 336, pad_to_square: bool = True) -> torch.Tensor:
    """SURVEY.md B1 (data_utils_dpo.py:319-341): RGB -> pad to square with the CLIP mean colour
    (image_aspect_ratio 'pad') -> CLIPImageProcessor: resize shortest edge (bicubic), centre crop, /255, normalise."""
    import numpy as np
    from PIL import Image
    img = pil_img.convert("RGB")
    if pad_to_square and img.size[0] != img.size[1]:
        w, h = img.size
        side = max(w, h)
        bg = Image.new("RGB", (side, side), tuple(int(x * 255) for x in CLIP_MEAN))
        bg.paste(img, ((side - w) // 2, (side - h) // 2))
        img = bg
    w, h = img.size
    scale = size / min(w, h)
    # CLIPImageProcessor.get_resize_output_image_size: short edge = size, long edge = int(size * long / short) (truncation)
    short, long_ = (w, h) if w <= h else (h, w)
    new_long = int(size * long_ / short)
    img = img.resize((size, new_long) if w <= h else (new_long, size), Image.BICUBIC)
    w, h = img.size
    left, top = (w - size) // 2, (h - size) // 2
    img = img.crop((left, top, left + size, top + size))
    x = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
    return (x - torch.tensor(CLIP_MEAN)[:, None, None]) / torch.tensor(CLIP_STD)[:, None, None]


class DPODataset(torch.utils.data.Dataset):
    """DPO_Dataset (data_utils_dpo.py:287-350): rows with `queries, image_bytes (base64) | images | image_id,
    standard_response, original_generate_response, AI_pseudo_response, AI_json_report`."""

    def __init__(self, rows, image_size: int = 336, image_dir: str = "", pad_to_square: bool = True):
        self.rows, self.image_size, self.image_dir, self.pad = rows, image_size, image_dir, pad_to_square

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, idx):
        import base64
        import io
        import os
        from PIL import Image
        item = self.rows[idx]
        if "images" in item or "image_id" in item:
            img = Image.open(os.path.join(self.image_dir, item.get("images", item.get("image_id"))))
        elif "image_bytes" in item:
            img = Image.open(io.BytesIO(base64.b64decode(item["image_bytes"].encode("utf-8"))))
        else:
            raise ValueError("No image found in the dataset")
        return {"queries": format_query(item["queries"]), "images": preprocess_image(img, self.image_size, self.pad),
                "standard_response": item["standard_response"], "original_generate_response": item["original_generate_response"],
                "AI_pseudo_response": item["AI_pseudo_response"], "AI_json_report": item["AI_json_report"]}
