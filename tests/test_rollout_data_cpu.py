"""Rollout query construction - host logic on the toy tokenizer.  The template / placeholder tokenisation
are restated from the absent third-party `llava` package (parity unpinned, see the module header); these tests pin the layout
the reference's own files state: the hard-coded template of utils/data_utils_dpo.py:291-292 and the padding / filtering rules of
utils/data_utils_online_gpt4v.py:41-173."""
import io
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "opa-dpo_amd"))
sys.path.insert(0, os.path.dirname(__file__))
from opadpo_amd import rollout_data as rd          # noqa: E402
from opadpo_amd.data import QUERY_TEMPLATE_HEAD, QUERY_TEMPLATE_TAIL      # noqa: E402
from toy_tokenizer import ToyTokenizer            # noqa: E402


def _png(color):
    from PIL import Image
    buf = io.BytesIO()
    Image.new("RGB", (6, 4), color).save(buf, format="PNG")
    return buf.getvalue()


def _rows():
    return [
        {"question": "what is on the mat ?", "chosen": "a cat sits there", "image": {"bytes": _png((255, 0, 0)), "path": "a.png"}},
        {"question": "describe <image> the scene in detail please now", "chosen": "two dogs", "image": {"bytes": _png((0, 255, 0)), "path": "b.png"}},
        {"question": " ".join(["long"] * 80), "chosen": "too long", "image": {"bytes": _png((0, 0, 255)), "path": "c.png"}},
        {"question": "last one ?", "chosen": "yes it is the last one indeed", "image": {"bytes": _png((9, 9, 9)), "path": "d.png"}},
    ]


def test_template_matches_the_reference_hard_coded_copy():
    # data_utils_dpo.py:291-292 spells the same template out as '<s> ' + system + ' USER: ' ... ' ASSISTANT: '
    conv = [{"from": "human", "value": "Q"}, {"from": "gpt", "value": None}]
    assert "<s> " + rd.render_prompt(conv) == QUERY_TEMPLATE_HEAD + "Q" + QUERY_TEMPLATE_TAIL.rstrip(" ")
    full = rd.render_prompt(rd.form_conversation("Q", "A"))
    assert full == rd.SYSTEM + " USER: <image>\nQ ASSISTANT: A</s>"
    with pytest.raises(AssertionError):
        rd.render_prompt([{"from": "human", "value": "x"}, {"from": "human", "value": "y"}])
    # a leading non-human turn is skipped
    assert rd.render_prompt([{"from": "gpt", "value": "hi"}] + rd.form_conversation("Q", "A")) == full


def test_placeholder_moves_to_the_front_and_tokenises_once():
    conv = rd.move_image_placeholder_first(rd.form_conversation("describe <image> the scene", "ok"))
    assert conv[0]["value"] == "<image>\ndescribe  the scene"          # both placeholders removed, one re-added in front
    assert conv[0]["value"].count("<image>") == 1 and conv[0]["value"].startswith("<image>\n")
    tok = ToyTokenizer()
    ids = rd.tokenize_with_image("a b <image> c d", tok)
    assert ids[0] == tok.bos_token_id and ids.count(tok.bos_token_id) == 1
    assert ids.count(rd.IMAGE_TOKEN_INDEX) == 1 and ids.index(rd.IMAGE_TOKEN_INDEX) == 3
    assert ids == tok._encode("a b") + [rd.IMAGE_TOKEN_INDEX] + tok._encode("c d")[1:]
    assert rd.tokenize_with_image("no image here", tok) == tok._encode("no image here")


def test_dataset_layout_filter_and_padding():
    tok = ToyTokenizer()
    logs = []
    ds = rd.QueryResponseDataset(_rows(), tok, query_len=48, image_size=8, log=logs.append)
    assert len(ds) == 3 and any("Filtered out 1 instances out of 4" in m for m in logs)
    assert ds.queries.shape == (3, 48) and ds.query_attn_masks.dtype == torch.long
    for i, src in enumerate([0, 1, 3]):
        q = rd.build_query_ids(_rows()[src]["question"], _rows()[src]["chosen"], tok)
        n = q.numel()
        assert torch.equal(ds.queries[i, -n:], q) and bool((ds.queries[i, :-n] == tok.pad_token_id).all())      # left padded
        assert int(ds.query_attn_masks[i].sum()) == n
        assert int((ds.queries[i] == rd.IMAGE_TOKEN_INDEX).sum()) == 1
        r = torch.tensor(tok._encode(_rows()[src]["chosen"])[1:] + [tok.eos_token_id])
        assert torch.equal(ds.standard_responses[i, :r.numel()], r)                                               # no BOS, EOS appended
        assert bool((ds.standard_responses[i, r.numel():] == tok.pad_token_id).all())                             # right padded
    assert ds.standard_responses.shape[1] == 8          # longest answer (7 words) + EOS
    # the prompt ends where the answer starts: the full templated prompt minus its last three tokens
    conv = rd.move_image_placeholder_first(rd.form_conversation("what is on the mat ?", "x"))
    conv[-1]["value"] = "\n"
    full = rd.tokenize_with_image(rd.render_prompt(conv), tok)
    assert rd.build_query_ids("what is on the mat ?", "x", tok).tolist() == full[:-3]
    with pytest.raises(ValueError):
        rd.QueryResponseDataset(_rows()[2:3], tok, query_len=48, log=lambda *_: None)


def test_items_and_collation():
    tok = ToyTokenizer()
    mod = rd.make_rollout_data_module(tok, data_path="", query_len=48, image_size=8, rows=_rows())
    ds, collate = mod["train_dataset"], mod["data_collator"]
    assert mod["eval_dataset"] is None
    batch = collate([ds[0], ds[1]])
    assert batch["queries"].shape == (2, 48) and batch["images"].shape == (2, 3, 8, 8)
    assert batch["images_path"] == ["a.png", "b.png"] and batch["images_bytes"][0] == _rows()[0]["image"]["bytes"]
    assert batch["images_url"][1].startswith("data:image/jpeg;base64,")
    # deliberate fix of data_utils_online_gpt4v.py:127 (records kept unfiltered there): item 2 is original row 3, image included
    assert ds[2]["images_path"] == "d.png" and ds[2]["images_bytes"] == _rows()[3]["image"]["bytes"]
    with pytest.raises(ValueError):
        bad = _rows(); bad[0]["image"]["bytes"] = b"not an image"
        rd.QueryResponseDataset(bad, tok, query_len=48, image_size=8, log=lambda *_: None)[0]


def test_rollout_step_columns_and_dataset_round_trip(tmp_path):
    """rollout_step -> write_rollout_json -> build_rows: the eight columns of online_generator.py:352-362 survive the wire."""
    from opadpo_amd import online_generate as og
    from opadpo_amd.dataset_build import build_rows, write_rollout_json
    assert og.query_text("SYS USER:  \nwhat is it ? ASSISTANT:") == "what is it ?"
    assert og.query_text("no markers") == "no markers"[7:-1]                       # str.find == -1 on both sides, like the reference
    tok = ToyTokenizer()
    ds = rd.QueryResponseDataset(_rows(), tok, query_len=48, image_size=8, log=lambda *_: None)
    batches = [rd.collate_query_response([ds[0], ds[1]]), rd.collate_query_response([ds[2]])]
    canned = ["the cat is red . it sleeps .", "two dogs run .", "yes ."]
    seen = []

    def sample(q, m, img):
        assert q.shape[1] == 48 and img.shape[1:] == (3, 8, 8) and torch.equal(m, q.ne(0).long())
        rows = [tok._encode(canned[len(seen) + i])[1:] + [tok.eos_token_id] for i in range(q.shape[0])]
        seen.extend(rows)
        w = max(len(r) for r in rows)
        return torch.tensor([r + [0] * (w - len(r)) for r in rows])

    def feedback(urls, queries, responses, standard):
        assert all(u.startswith("data:image/jpeg;base64,") for u in urls)
        return {"Pseudo_response": [r + " (fixed)" for r in responses], "Generated_response": list(responses),
                "report_json": [{"Sentence 1": {"score": 4}} for _ in responses]}

    out = og.rollout_step(batches, tok, sample, feedback)
    assert list(out) == ["query", "image_id", "standard_response", "original_generate_response", "AI_generate_response",
                         "AI_pseudo_response", "AI_json_report", "image_bytes"]
    assert all(len(v) == 3 for v in out.values())
    assert out["original_generate_response"] == canned and out["standard_response"][0] == "a cat sits there"
    assert out["image_id"] == ["a.png", "b.png", "d.png"] and all(q.startswith("<image>\n") for q in out["query"])
    path = write_rollout_json(str(tmp_path), 0, out, rank=0)
    rows = build_rows([os.path.dirname(path)], log=lambda *_: None)
    assert len(rows) == 3 and rows[1]["AI_pseudo_response"] == "two dogs run . (fixed)"
    # without a feedback model every record is dropped by the builder's first filter
    out2 = og.rollout_step(batches[:1], tok, lambda q, m, i: torch.tensor([[5, 2], [6, 2]]))
    p2 = write_rollout_json(str(tmp_path / "nofb"), 0, out2, rank=0)
    assert build_rows([os.path.dirname(p2)], log=lambda *_: None) == []
    with pytest.raises(ValueError):
        og.rollout_step(batches[:1], tok, lambda q, m, i: torch.tensor([[5, 2], [6, 2]]),
                        lambda *a: {"Pseudo_response": [""], "Generated_response": [""], "report_json": [""]})


def test_eval_prompt_and_question_chunks():
    """Text side of eval_llava_rlhf_coco/model_vqa.py:33-44,153-170 (no GPU: the module is imported, nothing is launched)."""
    from opadpo_amd import eval_generate as eg
    p = eg.eval_prompt("is there a dog ?")
    assert p == rd.SYSTEM + " USER: <image>\nis there a dog ?\nAnswer the question using a single word or phrase. ASSISTANT:"
    assert eg.eval_prompt("Q", None).endswith("USER: <image>\nQ ASSISTANT:")
    qs = list(range(10))
    assert eg.question_chunk(qs, 3, 0) == [0, 1, 2, 3] and eg.question_chunk(qs, 3, 2) == [8, 9] and eg.question_chunk(qs, 1, 0) == qs
    assert sum((eg.question_chunk(qs, 4, k) for k in range(4)), []) == qs
    with pytest.raises(IndexError):
        eg.question_chunk(list(range(4)), 3, 2)          # ceil(4/3) = 2 -> only two chunks exist


def test_preprocess_v1_label_masking_against_the_reference(golden_dir):
    """rollout_data.preprocess_v1 vs the reference's own preprocess_v1 (utils/common_utils.py:336-475; SFT data path with
    mask_target=True, data_utils_sft.py:187-214), run on the same conversations in the build container
    (tests/golden/make_preprocess_golden.py: template / image tokenisation stubbed with this build's restatements)."""
    import json
    import os
    from opadpo_amd.rollout_data import preprocess_v1
    from toy_tokenizer import EosAwareTokenizer
    cases = json.load(open(os.path.join(golden_dir, "ref_preprocess_v1.json")))
    assert len(cases) == 18
    seen_masked = seen_ignored = 0
    for c in cases:
        out = preprocess_v1(c["sources"], EosAwareTokenizer(), has_image=c["has_image"], mask_target=c["mask_target"],
                            query_len=c["query_len"], response_len=c["response_len"])
        assert out["input_ids"].tolist() == c["input_ids"], c["name"]
        assert out["labels"].tolist() == c["labels"], (c["name"], c["mask_target"])
        assert [bool(v) for v in out["validity"]] == c["validity"], (c["name"], c["query_len"])
        if c["mask_target"]:
            lab = out["labels"]
            seen_masked += int(((lab == -100).any(1) & (lab != -100).any(1)).sum())
            seen_ignored += int((lab == -100).all(1).sum())
    assert seen_masked > 0, "the fixture must contain rows with instruction tokens masked and response tokens kept"
