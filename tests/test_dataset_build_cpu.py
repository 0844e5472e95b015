"""opadpo_amd/dataset_build.py against the reference's own dataset builder (tests/golden/ref_dataset_build.json was produced by
running base_operations/make_opadpo_dataset.py on the same synthetic rollout files, tests/golden/make_dataset_golden.py)."""
import base64
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "opa-dpo_amd"))
from opadpo_amd import dataset_build as DB  # noqa: E402

GOLD = json.load(open(os.path.join(REPO, "tests", "golden", "ref_dataset_build.json"), encoding="utf-8"))


def _write_inputs(tmp_path):
    dirs = []
    for i in (1, 2, 3, 4):                                   # the reference lists four directories; 2 and 4 do not exist
        d = tmp_path / "output" / f"llava7b_online_generation_subset{i}" / "rollouts"
        dirs.append(str(d))
        fl = GOLD["files"].get(f"subset{i}")
        if fl:
            d.mkdir(parents=True)
            for name, recs in fl.items():
                (d / name).write_text(json.dumps(recs, indent=4))
    return dirs


def test_rows_match_reference_builder(tmp_path):
    rows = DB.build_rows(_write_inputs(tmp_path), log=lambda *_: None)
    assert DB.opa_columns(rows) == GOLD["opa"]
    assert DB.opadpo_columns(rows) == GOLD["opadpo"]
    assert len(rows) == 5                                    # 10 records: 1 empty report, 2 degenerate repetitions, 2 without a pseudo response


def test_saved_datasets_round_trip(tmp_path):
    pytest.importorskip("datasets")
    from datasets import load_from_disk
    rows = DB.build_rows(_write_inputs(tmp_path), log=lambda *_: None)
    opa, dpo = str(tmp_path / "base_datasets" / "opa"), str(tmp_path / "base_datasets" / "opadpo")
    DB.save_datasets(rows, opa, dpo, log=lambda *_: None)
    DB.save_datasets(rows, opa, str(tmp_path / "base_datasets" / "opadpo2"), log=lambda *_: None)      # an existing OPA directory is replaced
    a, d = load_from_disk(opa), load_from_disk(dpo)
    assert {c: list(a[c]) for c in a.column_names} == GOLD["opa"]
    assert {c: list(d[c]) for c in d.column_names} == GOLD["opadpo"]


def test_filters_and_ordering():
    assert DB.rollout_file_key("step10_rank3.json") == (10, 3)
    assert sorted(["step10_rank0.json", "step2_rank1.json", "step2_rank0.json"], key=DB.rollout_file_key) == \
        ["step2_rank0.json", "step2_rank1.json", "step10_rank0.json"]
    with pytest.raises(IndexError):
        DB.rollout_file_key("notes.txt")
    assert DB.has_repeating_last_sentence("The cat sleeps. The cat sleeps. The cat sleeps.")
    assert not DB.has_repeating_last_sentence("One sentence only") and not DB.has_repeating_last_sentence("A. B. C.")
    assert DB.has_repeating_last_word("go " * 40 + "and go") and not DB.has_repeating_last_word("go " * 20 + "and go")
    assert not DB.has_repeating_last_word("x")
    rec = DB.normalize_record({"query": "sys USER:  \n<image>\nWhat?", "AI_json_report": ""})
    assert rec["query"] == "<image>\n<image>\nWhat?" and rec["AI_json_report"] == '""'


def test_rollout_json_writer(tmp_path):
    """online_generator.py:379-396: list of records, base64 image bytes, step/rank in the file name; readable by the builder."""
    cols = {"query": ["q0 USER:  \nrest", "q1"], "image_id": ["a", "b"], "standard_response": ["s0.", "s1."],
            "original_generate_response": ["First. Second.", "Other. Text."], "AI_generate_response": ["g0", "g1"],
            "AI_pseudo_response": ["p0", "p1"], "AI_json_report": [[{"sentence": "x", "score": 4}], ""],
            "image_bytes": [b"\x00\x01\xff", b"abc"]}
    path = DB.write_rollout_json(str(tmp_path), 7, cols, rank=2)
    assert path.endswith(os.path.join("rollouts", "step7_rank2.json"))
    recs = json.load(open(path))
    assert [base64.b64decode(r["image_bytes"]) for r in recs] == cols["image_bytes"] and recs[1]["query"] == "q1"
    rows = DB.build_rows([os.path.dirname(path)], log=lambda *_: None)
    assert len(rows) == 1 and rows[0]["query"] == "<image>\nrest"           # record 1 has an empty report
    with pytest.raises(ValueError):
        DB.write_rollout_json(str(tmp_path), 8, {"query": ["a"], "image_id": []}, rank=0)
    assert DB.write_rollout_json(None, 1, cols) is None


def test_stratified_subsets_match_the_reference_script(tmp_path):
    """dataset_build.stratified_subsets / make_online_generation_subsets on the synthetic pool of tests/golden/make_subsets_golden.py:
    the same four 2500-row subsets, in the same order, as the reference's base_operations/make_online_generation_dataset.py."""
    import importlib.util
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_online_subsets.json")))
    spec = importlib.util.spec_from_file_location("mk", os.path.join(os.path.dirname(__file__), "golden", "make_subsets_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    assert (mk.N_ROWS, mk.N_SHARDS, mk.SEED) == (gold["n_rows"], gold["n_shards"], gold["seed"])
    df = mk.pool()
    from opadpo_amd.dataset_build import make_online_generation_subsets, stratified_subsets
    subs = stratified_subsets(df)
    assert [list(s["idx"]) for s in subs] == gold["subsets"]
    ids = [set(s["idx"]) for s in subs]
    assert all(len(i) == 2500 for i in ids) and len(set().union(*ids)) == 10000           # disjoint
    share = [float((s["origin_dataset"] == "VQAv2").mean()) for s in subs]
    assert max(share) - min(share) < 0.01                                                  # stratified
    # through parquet shards and HF datasets on disk (smaller subsets: same chain of splits)
    d = tmp_path / "pool"
    d.mkdir()
    per = (len(df) + 3) // 4
    files = []
    for i in range(4):
        files.append(str(d / f"RLAIF-V-Dataset_{i:03d}.parquet"))
        df.iloc[i * per:(i + 1) * per].to_parquet(files[-1], index=False)
    paths = make_online_generation_subsets(files, str(tmp_path / "sub"), per_subset=2500, log=lambda *_: None)
    from datasets import load_from_disk
    assert [list(load_from_disk(p)["idx"]) for p in paths] == gold["subsets"]
