"""Wire formats between the rollout stage and the two training stages.

Reference behaviour restated (no code shared):
  * generator_models/online_generator.py:368-396 - one JSON file per rollout step and rank,
    `<output_dir>/rollouts/step{step}_rank{rank}.json`, a list of records (one per prompt) whose `image_bytes` are base64 text;
  * base_operations/make_opadpo_dataset.py:6-128 - all rollout files of the listed directories are concatenated (files ordered
    by the first two integers in their name = step, rank; missing directories are skipped with a message), every record is
    normalised (report serialised with json.dumps(ensure_ascii=False, indent=4); a query containing 'USER:  \\n' keeps what
    follows it, prefixed with '<image>\\n'), three filters run in order (empty report; generated response whose last sentence
    already occurs earlier or whose last word occurs more than 30 times; missing / empty AI pseudo response) and two HF
    datasets are written: the OPA (SFT) columns and the OPA-DPO columns.

Host-only code: nothing here touches the GPU.  `datasets` is imported lazily (save / load only).
"""
from __future__ import annotations

import argparse
import base64
import json
import os
import re
import shutil
from typing import Dict, Iterable, List, Sequence, Tuple

OPA_COLUMNS = ("queries", "image_bytes", "standard_response", "AI_pseudo_response")
OPADPO_COLUMNS = ("queries", "image_bytes", "standard_response", "original_generate_response", "AI_pseudo_response", "AI_json_report")
QUERY_MARKER = "USER:  \n"


def rollout_file_key(file_name: str) -> Tuple[int, int]:
    """(step, rank) = the first two integers of the file name (make_opadpo_dataset.py:6-8); a name with fewer than two raises
    IndexError there as well."""
    numbers = re.findall(r"\d+", file_name)
    return int(numbers[0]), int(numbers[1])


def load_rollout_records(json_dirs: Sequence[str] | str, log=print) -> List[dict]:
    """Concatenate the records of every `*.json` under the given directories, files in (step, rank) order per directory."""
    if isinstance(json_dirs, str):
        json_dirs = [json_dirs]
        must_exist = True          # a single directory is read unconditionally by the reference (:68-70)
    else:
        must_exist = False
    records: List[dict] = []
    for d in json_dirs:
        if not os.path.exists(d):
            if must_exist:
                raise FileNotFoundError(d)
            log(f"Directory {d} does not exist.")
            continue
        for name in sorted(os.listdir(d), key=rollout_file_key):
            if name.endswith(".json"):
                with open(os.path.join(d, name), "r", encoding="utf-8") as f:
                    records.extend(json.load(f))
    return records


def normalize_record(item: dict) -> dict:
    """In place, like the reference (:76-82): report -> indented JSON text, query cut after the conversation header."""
    item["AI_json_report"] = json.dumps(item["AI_json_report"], ensure_ascii=False, indent=4)
    q = item["query"]
    if QUERY_MARKER in q:
        item["query"] = "<image>\n" + q[q.find(QUERY_MARKER) + len(QUERY_MARKER):]
    return item


def has_repeating_last_sentence(text: str) -> bool:
    """The last complete sentence (text between the last two '.') already occurs in what precedes it (:19-29)."""
    sentences = text.split(".")
    if len(sentences) < 2:
        return False
    last = sentences[-2].strip()
    return last in ".".join(sentences[:-2])


def has_repeating_last_word(text: str) -> bool:
    """The last word occurs more than 30 times among all but the last two words (:31-38)."""
    words = text.split()
    if len(words) < 2:
        return False
    return words[:-2].count(words[-1].strip()) > 30


def filter_records(records: Iterable[dict], log=print) -> List[dict]:
    """The three filters of make_opadpo_dataset.py:85-98, in order, on NORMALISED records."""
    records = list(records)
    n0 = len(records)
    kept = [r for r in records if r["AI_json_report"] != '""']
    log(f"Filter1 (empty AI_json_report): {n0} -> {len(kept)}")
    n1 = len(kept)
    kept = [r for r in kept if not has_repeating_last_sentence(r["original_generate_response"])
            and not has_repeating_last_word(r["original_generate_response"])]
    log(f"Filter2 (degenerate repetition): {n1} -> {len(kept)}")
    n2 = len(kept)
    kept = [r for r in kept if isinstance(r.get("AI_pseudo_response", ""), str) and len(r.get("AI_pseudo_response", "")) > 0]
    log(f"Filter3 (empty AI_pseudo_response): {n2} -> {len(kept)}")
    return kept


def _columns(records: List[dict], names: Sequence[str]) -> Dict[str, list]:
    src = {"queries": "query"}
    return {n: [r[src.get(n, n)] for r in records] for n in names}


def opa_columns(records: List[dict]) -> Dict[str, list]:
    """Columns of the OPA (LoRA-SFT) dataset (:111-116)."""
    return _columns(records, OPA_COLUMNS)


def opadpo_columns(records: List[dict]) -> Dict[str, list]:
    """Columns of the OPA-DPO dataset (:120-127) - what data.DataCollatorForCausalLM consumes."""
    return _columns(records, OPADPO_COLUMNS)


def build_rows(json_dirs: Sequence[str] | str, log=print) -> List[dict]:
    records = load_rollout_records(json_dirs, log)
    seen = []
    for r in records:
        if r["image_id"] not in seen:
            seen.append(r["image_id"])
        normalize_record(r)
    log(f"Number of unique image_id: {len(seen)}")
    return filter_records(records, log)


def save_datasets(records: List[dict], opa_path: str, opadpo_path: str, log=print) -> None:
    """Write both HF datasets (an existing OPA directory is replaced, its parent created; :101-128)."""
    from datasets import Dataset
    if os.path.exists(opa_path):
        log(f"Removing existing file: {opa_path}")
        shutil.rmtree(opa_path)
    parent = os.path.dirname(opa_path)
    if parent and not os.path.exists(parent):
        log(f"Creating directory: {parent}")
        os.makedirs(parent)
    Dataset.from_dict(opa_columns(records)).save_to_disk(opa_path)
    Dataset.from_dict(opadpo_columns(records)).save_to_disk(opadpo_path)


def write_rollout_json(output_dir: str, step_idx: int, response_dict: Dict[str, list], rank: int | None = None) -> str | None:
    """One rollout step -> `<output_dir>/rollouts/step{step}_rank{rank}.json` (online_generator.py:379-396): the dict of
    equal-length columns becomes a list of records, `image_bytes` (raw bytes) is stored as base64 text, indent 4."""
    if output_dir is None:
        return None
    keys = list(response_dict.keys())
    n = len(response_dict[keys[0]]) if keys else 0
    if any(len(response_dict[k]) != n for k in keys):
        raise ValueError("All arrays must be of the same length")          # what the reference's pandas.DataFrame(...) raises
    rows = [{k: response_dict[k][i] for k in keys} for i in range(n)]
    if "image_bytes" in response_dict:
        for row in rows:
            row["image_bytes"] = base64.b64encode(row["image_bytes"]).decode("utf-8")
    d = os.path.join(output_dir, "rollouts")
    os.makedirs(d, exist_ok=True)
    if rank is None:
        rank = int(os.environ.get("RANK", 0))
    path = os.path.join(d, f"step{step_idx}_rank{rank}.json")
    with open(path, "w") as f:
        json.dump(rows, f, indent=4)
    return path


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description="rollout JSON -> OPA / OPA-DPO HF datasets")
    ap.add_argument("--json_dir", nargs="+", default=[f"./output/llava7b_online_generation_subset{i}/rollouts" for i in (1, 2, 3, 4)])
    ap.add_argument("--opa_out", default="./base_datasets/opa_training_data-7B")
    ap.add_argument("--opadpo_out", default="./base_datasets/opadpo_training_data-7B")
    a = ap.parse_args(argv)
    save_datasets(build_rows(a.json_dir), a.opa_out, a.opadpo_out)


if __name__ == "__main__":
    main()


# ---- rollout-stage input: four stratified subsets of the RLAIF-V pool (base_operations/make_online_generation_dataset.py:10-47) ----
def stratified_subsets(df, per_subset: int = 2500, key: str = "origin_dataset", seed: int = 42):
    """Four disjoint subsets of `per_subset` rows each, stratified by `key`: 4 * per_subset rows are drawn from the pool, halved,
    and each half halved again - the reference's chain of sklearn `train_test_split(stratify=..., random_state=42)` calls, so the
    same pool gives the same rows in the same order.  Returns [subset1, subset2, subset3, subset4] (pandas DataFrames)."""
    from sklearn.model_selection import train_test_split
    n = 4 * per_subset
    _, pool = train_test_split(df, test_size=n, stratify=df[key], random_state=seed)
    first, second = train_test_split(pool, test_size=n // 2, stratify=pool[key], random_state=seed)
    s1, s2 = train_test_split(first, test_size=n // 4, stratify=first[key], random_state=seed)
    s3, s4 = train_test_split(second, test_size=n // 4, stratify=second[key], random_state=seed)
    return [s1, s2, s3, s4]


def make_online_generation_subsets(data_files: Sequence[str], out_root: str = "./base_datasets/LLaVA-RLAIF-SubData", per_subset: int = 2500,
                                   log=print) -> List[str]:
    """Parquet shards of the RLAIF-V dataset -> `<out_root>/subset{1..4}` HF datasets (the `--data_path` of run/online_generate.sh)."""
    import pandas as pd
    from datasets import Dataset, load_dataset
    train = load_dataset("parquet", data_files=list(data_files))["train"]
    df = pd.DataFrame(train)
    log(f"pool: {len(df)} rows, origins: {dict(df['origin_dataset'].value_counts())}")
    paths = []
    for i, sub in enumerate(stratified_subsets(df, per_subset), 1):
        path = os.path.join(out_root, f"subset{i}")
        Dataset.from_pandas(sub).save_to_disk(path)
        paths.append(path)
    return paths
