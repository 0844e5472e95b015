"""Query construction for the rollout stage: (question, chosen answer, image) rows -> left-padded
prompt tokens + right-padded standard responses, the tensors `generate.Generator.rollout` consumes.

Reference behaviour restated (no code shared):
  * utils/data_utils_online_gpt4v.py:26-30  - a row becomes a two-turn conversation, human = '<image>\\n' + question;
  * utils/common_utils.py:190-221           - the image placeholder is moved to the front of the turn it occurs in;
  * utils/common_utils.py:336-475           - `preprocess_v1`: Vicuna-v1 two-separator template, tokenised with the image
    placeholder as IMAGE_TOKEN_INDEX (-200); on this call path `mask_target=False`, only `input_ids` are used;
  * utils/data_utils_online_gpt4v.py:41-173 - the answer turn is replaced by a newline before templating, the last three
    tokens of the prompt are dropped (newline, end-of-sequence and the blank before them, so the prompt ends where the answer
    starts), rows whose prompt exceeds `query_len` are dropped, prompts are left-padded to `query_len`, the tokenised answers
    (without BOS, plus EOS) right-padded to the longest one; `__getitem__` adds the square-padded, CLIP-normalised image, a
    data URL and the raw bytes.

PARITY UNPINNED for the template text and the placeholder tokenisation: both live in the third-party package `llava`
(`llava.conversation`, `llava.mm_utils.tokenizer_image_token`; LLaVA v1.1.x), which is absent from /root/reference, so the
reference's `preprocess_v1` cannot be imported here.  Their published behaviour is restated below; the template wording is
anchored on the reference's own hard-coded copy in utils/data_utils_dpo.py:291-292 (same system sentence, roles and blanks).

Host-only code: nothing here touches the GPU.
"""
from __future__ import annotations

import base64
import io
from typing import Dict, List, Optional, Sequence

import torch

from .data import preprocess_image

IMAGE_TOKEN_INDEX = -200
IMAGE_PLACEHOLDER = "<image>"
SYSTEM = ("A chat between a curious user and an artificial intelligence assistant. "
          "The assistant gives helpful, detailed, and polite answers to the user's questions.")
ROLES = ("USER", "ASSISTANT")
SEP, SEP2 = " ", "</s>"


def form_conversation(question: str, answer: str) -> List[dict]:
    return [{"from": "human", "value": IMAGE_PLACEHOLDER + "\n" + question}, {"from": "gpt", "value": answer}]


def move_image_placeholder_first(conversation: List[dict]) -> List[dict]:
    """In place: a turn that mentions the image starts with the placeholder on its own line (common_utils.py:199-206)."""
    for turn in conversation:
        if IMAGE_PLACEHOLDER in turn["value"]:
            body = turn["value"].replace(IMAGE_PLACEHOLDER, "").strip()
            turn["value"] = (IMAGE_PLACEHOLDER + "\n" + body).strip()
    return conversation


def render_prompt(conversation: Sequence[dict]) -> str:
    """Two-separator (Vicuna v1) layout: system, then 'ROLE: text' turns closed alternately by a blank and by '</s>'; an empty
    turn is rendered as 'ROLE:' (open for generation)."""
    turns = list(conversation)
    if turns and turns[0]["from"] != "human":
        turns = turns[1:]                                  # common_utils.py:356-358
    seps = (SEP, SEP2)
    out = SYSTEM + SEP
    for j, turn in enumerate(turns):
        role = ROLES[0] if turn["from"] == "human" else ROLES[1]
        if role != ROLES[j % 2]:
            raise AssertionError("turns must alternate human / gpt")
        out += (role + ": " + turn["value"] + seps[j % 2]) if turn["value"] else (role + ":")
    return out


def tokenize_with_image(prompt: str, tokenizer, image_token_index: int = IMAGE_TOKEN_INDEX) -> List[int]:
    """Tokenise the text between image placeholders separately and join the pieces with `image_token_index`; the BOS every piece
    starts with is kept once (published behaviour of llava.mm_utils.tokenizer_image_token)."""
    chunks = [torch.as_tensor(tokenizer(c)["input_ids"]).view(-1).tolist() for c in prompt.split(IMAGE_PLACEHOLDER)]
    ids: List[int] = []
    skip = 0
    if chunks and chunks[0] and chunks[0][0] == tokenizer.bos_token_id:
        skip = 1
        ids.append(chunks[0][0])
    for k, c in enumerate(chunks):
        if k:
            ids.append(image_token_index)
        ids.extend(c[skip:])
    return ids


IGNORE_INDEX = -100      # utils/constants.py


def preprocess_v1(sources: Sequence[Sequence[dict]], tokenizer, has_image: bool = False, mask_target: bool = True,
                  query_len: Optional[int] = None, response_len: Optional[int] = None) -> Dict:
    """`preprocess_v1` of utils/common_utils.py:336-475 (the SFT data path: data_utils_sft.py:187-214 calls it per (query, response)
    sample with mask_target=True; the rollout path with mask_target=False): conversations -> Vicuna-v1 prompt -> ids; labels = ids with,
    when `mask_target`, BOS, every instruction part (`... ASSISTANT: `, counted as its token count - 2) and everything after the last
    round set to IGNORE_INDEX; a sample whose round lengths do not add up to its token count is ignored entirely (all labels
    IGNORE_INDEX, the reference prints a warning).  `validity[c]` = the LAST round fits query_len / response_len.
    -> {input_ids [n, width], labels [n, width], validity [n]} (has_image: all prompts must tokenise to the same length, like the
    reference's torch.stack; otherwise right-padded to the longest)."""
    convs = [render_prompt(src) for src in sources]
    if has_image:
        rows = [tokenize_with_image(c, tokenizer) for c in convs]
        if len({len(r) for r in rows}) != 1:
            raise RuntimeError("stack expects each tensor to be equal size (has_image=True tokenises without padding)")
        input_ids = torch.tensor(rows, dtype=torch.long)
    else:
        rows = [torch.as_tensor(tokenizer(c)["input_ids"]).view(-1).tolist() for c in convs]
        width = min(max(len(r) for r in rows), getattr(tokenizer, "model_max_length", 1 << 30))
        input_ids = torch.tensor([(r[:width] + [tokenizer.pad_token_id] * (width - len(r[:width]))) for r in rows], dtype=torch.long)
    targets = input_ids.clone()
    validity = [True] * len(convs)
    sep = SEP + ROLES[1] + ": "
    n_tok = (lambda t: len(tokenize_with_image(t, tokenizer))) if has_image else (lambda t: len(torch.as_tensor(tokenizer(t)["input_ids"]).view(-1)))
    for c, (conversation, target) in enumerate(zip(convs, targets)):
        total_len = int(target.ne(tokenizer.pad_token_id).sum())
        cur_len = 1
        if mask_target:
            target[:cur_len] = IGNORE_INDEX
        final_query_len = final_response_len = 0
        for rou in conversation.split(SEP2):
            if rou == "":
                break
            parts = rou.split(sep)
            if len(parts) != 2:
                break
            parts[0] += sep
            round_len = n_tok(rou)
            instruction_len = n_tok(parts[0]) - 2
            if mask_target:
                target[cur_len: cur_len + instruction_len] = IGNORE_INDEX
            final_query_len, final_response_len = cur_len, round_len
            cur_len += round_len
        if final_response_len == 0:
            raise ValueError(f"Empty response: {conversation}")
        validity[c] = (query_len is None or final_query_len <= query_len) and (response_len is None or final_response_len <= response_len)
        if mask_target:
            target[cur_len:] = IGNORE_INDEX
        if cur_len < getattr(tokenizer, "model_max_length", 1 << 30) and cur_len != total_len:
            if mask_target:
                target[:] = IGNORE_INDEX
            print(f"WARNING: tokenization mismatch: {cur_len} vs. {total_len}. (ignored)")
    return dict(input_ids=input_ids, labels=targets, validity=validity)


def build_query_ids(question: str, answer: str, tokenizer) -> torch.Tensor:
    """Prompt tokens of one row: answer turn blanked to a newline, templated, tokenised, last three tokens dropped."""
    conv = move_image_placeholder_first(form_conversation(question, answer))
    conv[-1]["value"] = "\n"
    ids = tokenize_with_image(render_prompt(conv), tokenizer)
    return torch.tensor(ids, dtype=torch.long)[:-3]


def _ids_of(tokenizer, text: str) -> torch.Tensor:
    return torch.as_tensor(tokenizer(text)["input_ids"], dtype=torch.long).view(-1)


def _pad(t: torch.Tensor, width: int, value: int, left: bool) -> torch.Tensor:
    fill = torch.full((width - t.numel(),), value, dtype=t.dtype)
    return torch.cat([fill, t]) if left else torch.cat([t, fill])


class QueryResponseDataset(torch.utils.data.Dataset):
    """Left-padded prompts + right-padded standard responses (data_utils_online_gpt4v.py:41-173).  `rows`: records with
    'question', 'chosen' and 'image' = {'bytes', 'path'} (the RLAIF-V layout)."""

    def __init__(self, rows: Sequence[dict], tokenizer, query_len: int, image_size: int = 336, pad_to_square: bool = True,
                 log=print):
        rows = list(rows)
        queries = [build_query_ids(r["question"], r["chosen"], tokenizer) for r in rows]
        responses = [_ids_of(tokenizer, r["chosen"])[1:] for r in rows]                      # BOS dropped (:80-82)
        keep = [i for i, q in enumerate(queries) if q.numel() <= query_len]
        if not keep:
            raise ValueError("no prompt fits query_len")                                     # the reference's max() of nothing
        log(f"Max query length: {max(queries[i].numel() for i in keep)}")
        log(f"Filtered out {len(rows) - len(keep)} instances out of {len(rows)} that exceed length limit.")
        eos = torch.tensor([tokenizer.eos_token_id])
        responses = [torch.cat([responses[i], eos]) for i in keep]
        width = max(r.numel() for r in responses)
        self.queries = torch.stack([_pad(queries[i], query_len, tokenizer.pad_token_id, left=True) for i in keep])
        self.query_attn_masks = self.queries.ne(tokenizer.pad_token_id).long()
        self.standard_responses = torch.stack([_pad(r, width, tokenizer.pad_token_id, left=False) for r in responses])
        # Deliberate fix: the reference keeps the UNFILTERED records next to the filtered tensors (data_utils_online_gpt4v.py:127),
        # so after a dropped row item i pairs query i with the image of original row i - a different sample.  Here the records
        # are filtered with the tensors.  Identical whenever no prompt exceeds query_len (the shipped data: questions << 128 tokens).
        self.rows = [rows[i] for i in keep]
        self.image_size, self.pad_to_square = image_size, pad_to_square

    def __len__(self) -> int:
        return self.queries.shape[0]

    def __getitem__(self, idx: int) -> Dict:
        from PIL import Image
        image = self.rows[idx]["image"]
        raw = image["bytes"]
        try:
            pil = Image.open(io.BytesIO(raw)).convert("RGB")
        except Exception as e:
            raise ValueError(f"Error loading image for index {idx}") from e
        return dict(
            queries=self.queries[idx], query_attn_masks=self.query_attn_masks[idx], standard_responses=self.standard_responses[idx],
            images=preprocess_image(pil, self.image_size, self.pad_to_square), images_path=image.get("path"),
            images_url="data:image/jpeg;base64," + base64.b64encode(raw).decode("utf-8"), images_bytes=raw)


def collate_query_response(instances: Sequence[Dict]) -> Dict:
    """Tensors stacked, everything else listed (data_utils_online_gpt4v.py:32-41)."""
    out = {}
    for key in instances[0]:
        vals = [inst[key] for inst in instances]
        out[key] = torch.stack(vals) if isinstance(vals[0], torch.Tensor) else vals
    return out


def make_rollout_data_module(tokenizer, data_path: str, query_len: int, image_size: int = 336, pad_to_square: bool = True,
                             rows: Optional[Sequence[dict]] = None) -> Dict:
    """`make_rlaif_gpt4v_data_module` (:176-207): the HF dataset on disk, unshuffled, as one QueryResponseDataset."""
    if rows is None:
        import datasets
        rows = list(datasets.load_from_disk(data_path))
    return dict(train_dataset=QueryResponseDataset(rows, tokenizer, query_len, image_size, pad_to_square), eval_dataset=None,
                data_collator=collate_query_response)
