#!/usr/bin/env python3
"""Golden fixture for opadpo_amd/dataset_build.py (BUILD container only): synthetic rollout JSON files are written into a
scratch directory, the REFERENCE's own base_operations/make_opadpo_dataset.py runs there as a subprocess (it is a script with
relative paths), and the two HF datasets it saves are read back.  Stored: the input files and the resulting columns - data
only.  Output: tests/golden/ref_dataset_build.json."""
import base64
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SCRIPT = "/root/reference/base_operations/make_opadpo_dataset.py"


def records():
    img = lambda i: base64.b64encode(bytes([i, i + 1, i + 2, 255 - i])).decode()
    rep = lambda s: [{"sentence": "A dog.", "score": s, "error type": "correct"}, {"sentence": "It flies.", "score": 1, "error type": "image_recognition_error"}]
    mk = lambda i, **kw: dict({"query": f"A chat. USER:  \n<image>\nDescribe picture {i}. ASSISTANT:", "image_id": f"img{i % 5}",
                               "standard_response": f"Standard answer {i}.", "original_generate_response": f"A dog sits. It looks at picture {i}. Nice.",
                               "AI_generate_response": f"gpt raw {i}", "AI_pseudo_response": f"A dog sits. It rests near picture {i}. Nice.",
                               "AI_json_report": rep(4 - i % 4), "image_bytes": img(i)}, **kw)
    files = {
        "subset1": {"step2_rank0.json": [mk(0), mk(1, AI_json_report=""), mk(2, query="<image>\nNo header here?")],
                    "step10_rank0.json": [mk(3, original_generate_response="The cat sleeps. The cat sleeps. The cat sleeps."),
                                          mk(4, original_generate_response="go " * 40 + "and go"),
                                          mk(5, AI_pseudo_response="")],
                    "step2_rank1.json": [mk(6, AI_pseudo_response=None), mk(7, AI_json_report=rep(2) + [{"sentence": "Ünïcode 图.", "score": 3, "error type": "correct"}])]},
        "subset3": {"step1_rank0.json": [mk(8, original_generate_response="One sentence only"), mk(9, original_generate_response="x")]},
    }
    return files


def main():
    from datasets import load_from_disk
    files = records()
    with tempfile.TemporaryDirectory() as tmp:
        for sub, fl in files.items():
            d = os.path.join(tmp, "output", f"llava7b_online_generation_{sub}", "rollouts")
            os.makedirs(d)
            for name, recs in fl.items():
                with open(os.path.join(d, name), "w") as f:
                    json.dump(recs, f, indent=4)
        r = subprocess.run([sys.executable, REF_SCRIPT], cwd=tmp, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        opa = load_from_disk(os.path.join(tmp, "base_datasets", "opa_training_data-7B"))
        dpo = load_from_disk(os.path.join(tmp, "base_datasets", "opadpo_training_data-7B"))
        out = {"files": files, "opa": {c: list(opa[c]) for c in opa.column_names}, "opadpo": {c: list(dpo[c]) for c in dpo.column_names}}
    with open(os.path.join(HERE, "ref_dataset_build.json"), "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print("rows kept:", len(out["opa"]["queries"]), "columns:", list(out["opa"]), list(out["opadpo"]))


if __name__ == "__main__":
    main()
