"""Launch surfaces of the two generation stages, same flag names as the reference:

  * `main_rollout`  - opadpo/online_generation_custom.py as launched by run/online_generate.sh: the RLAIF-V style dataset on disk ->
    left-padded queries -> sampled responses of the (adapter-free or LoRA) policy -> `<output_dir>/rollouts/step{N}_rank{R}.json`.
    One process per GPU, prompts strided by rank, no collective (replicas only).  The GPT-4V feedback client needs the network
    and stays out of scope: records carry empty feedback fields unless a `feedback` callable is passed to `run_rollout`.
  * `main_eval`     - eval_llava_rlhf_coco/model_vqa.py: question file -> answers JSONL from base weights + a PEFT adapter.

`--synthetic tiny|7b` swaps checkpoints, tokenizer and data for random-init stand-ins (no network in the build environment).
PPO-era / DeepSpeed flags of the reference launchers are accepted and ignored (Quirk Q9).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from typing import Callable, List, Optional


def _bool(v) -> bool:
    return str(v).lower() in ("1", "true", "yes", "y")


def rollout_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description="on-policy rollout generation (MI355X)", allow_abbrev=False)
    ap.add_argument("--cfg", default=None)
    ap.add_argument("--base_model_name", default="./base_models/llava-v1.5-7b")
    ap.add_argument("--policy_model_name_or_path", default="none")
    ap.add_argument("--data_path", default="./base_datasets/LLaVA-RLAIF-SubData/subset1")
    ap.add_argument("--output_dir", default="./output/llava7b_online_generation_subset1")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--rollout_batch_size", type=int, default=32)
    ap.add_argument("--rollout_per_device_batch_size", type=int, default=4)
    ap.add_argument("--query_len", type=int, default=128)
    ap.add_argument("--response_len", type=int, default=896)
    ap.add_argument("--model_max_length", type=int, default=2048)
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--top_k", type=int, default=30)
    ap.add_argument("--top_p", type=float, default=0.95)
    ap.add_argument("--sample_num", type=int, default=0, help="use only the first N rows (0 = all)")
    ap.add_argument("--max_step", type=int, default=0, help="stop after N rollout steps (0 = whole dataset)")
    ap.add_argument("--image_aspect_ratio", default="pad")
    ap.add_argument("--lora_rank", type=int, default=256)
    ap.add_argument("--lora_alpha", type=float, default=512.0)
    ap.add_argument("--synthetic", default=None, choices=[None, "tiny", "7b"])
    ap.add_argument("--synthetic_rows", type=int, default=8)
    return ap


def eval_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description="evaluation generation: question file -> answers file (MI355X)", allow_abbrev=False)
    ap.add_argument("--model-path", default="./base_models/llava-v1.5-7b")
    ap.add_argument("--model-base", default=None)
    ap.add_argument("--image-folder", default="")
    ap.add_argument("--question-file", default="tables/question.jsonl")
    ap.add_argument("--answers-file", default="answer.jsonl")
    ap.add_argument("--conv-mode", default="llava_v1")
    ap.add_argument("--num-chunks", type=int, default=1)
    ap.add_argument("--chunk-idx", type=int, default=0)
    ap.add_argument("--temperature", type=float, default=0.0)
    ap.add_argument("--top_p", type=float, default=None)
    ap.add_argument("--num_beams", type=int, default=1)
    ap.add_argument("--use-qlora", type=_bool, default=False)
    ap.add_argument("--qlora-path", default="")
    ap.add_argument("--short_eval", type=_bool, default=False)
    ap.add_argument("--image_aspect_ratio", default="pad")
    ap.add_argument("--test-prompt", default="\nAnswer the question using a single word or phrase.")
    ap.add_argument("--batch-size", type=int, default=1, help="questions per generation (the reference runs 1)")
    ap.add_argument("--merge-adapter", type=_bool, default=True)
    ap.add_argument("--lora_rank", type=int, default=256)
    ap.add_argument("--lora_alpha", type=float, default=512.0)
    ap.add_argument("--synthetic", default=None, choices=[None, "tiny", "7b"])
    return ap


def _engine(base_dir: str, synthetic: Optional[str], adapter_dir: Optional[str], lora_rank: int, lora_alpha: float, seed: int = 0):
    """-> (engine, frozen adapter or None, tokenizer)."""
    import torch
    from . import checkpoint_io as CK
    from .dims import LlavaDims
    from .ctx import CtxEngine
    from .model import BaseWeights, LoraAdapter
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(dev)
    if synthetic:
        from .synth import SyntheticTokenizer, init_lora, init_weights
        d = {"tiny": LlavaDims.tiny, "7b": LlavaDims.llava15_7b}[synthetic]()
        state = init_weights(d, seed=seed, device=dev)
        adapter_sd = init_lora(d, seed=seed + 1, device=dev) if adapter_dir else None
        tok = SyntheticTokenizer(d.vocab)
    else:
        from transformers import AutoTokenizer
        d = CK.dims_from_config(base_dir, lora_rank, lora_alpha)
        state = CK.load_llava_state(base_dir)
        adapter_sd = None
        if adapter_dir and os.path.exists(os.path.join(adapter_dir, "adapter_config.json")):
            adapter_sd = CK.load_adapter(adapter_dir)
        elif adapter_dir:
            print(f"No lora adapter found in {adapter_dir}, using the original model")
        tok = AutoTokenizer.from_pretrained(base_dir, use_fast=False)
        if tok.pad_token_id is None:
            tok.pad_token = tok.unk_token
    vision = {k: v for k, v in (adapter_sd or {}).items() if "vision_tower" in k or "mm_projector" in k} or None
    engine = CtxEngine(BaseWeights(d, state, dev, need_backward=False, vision_lora=vision))
    adapter = LoraAdapter(d, adapter_sd, dev, trainable=False) if adapter_sd else None
    return engine, adapter, tok


def run_rollout(args, feedback: Optional[Callable] = None, log=print) -> List[str]:
    """The loop of Online_Generator.train (online_generator.py:399-430) without the optimisation step it never runs: per step
    `rollout_batch_size / world` prompts of this rank in per-device batches -> one JSON file.  Returns the files written."""
    from .dataset_build import write_rollout_json
    from .generate import Generator
    from .online_generate import generator_sampler, rollout_step
    from .rollout_data import QueryResponseDataset, collate_query_response
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    none = args.policy_model_name_or_path in (None, "", "none")
    engine, adapter, tok = _engine(args.base_model_name, args.synthetic, None if none else args.policy_model_name_or_path,
                                   args.lora_rank, args.lora_alpha, args.seed)
    if args.synthetic:
        from .synth import synth_question_rows
        rows = synth_question_rows(args.synthetic_rows, seed=args.seed)
    else:
        import datasets
        rows = list(datasets.load_from_disk(args.data_path))
    if args.sample_num:
        rows = rows[:args.sample_num]
    ds = QueryResponseDataset(rows, tok, args.query_len, image_size=engine.d.image_size, pad_to_square=args.image_aspect_ratio == "pad", log=log)
    gen = Generator(engine, adapter, merge_adapter=True, fuse_swiglu=True)
    sample = generator_sampler(gen, response_len=args.response_len, temperature=args.temperature, top_k=args.top_k, top_p=args.top_p,
                               seed=args.seed + 1000 * rank)
    mine = list(range(rank, len(ds), world))
    per_step = max(args.rollout_per_device_batch_size, args.rollout_batch_size // world)
    files = []
    for step, s0 in enumerate(range(0, len(mine), per_step)):
        if args.max_step and step >= args.max_step:
            break
        idx = mine[s0:s0 + per_step]
        batches = [collate_query_response([ds[j] for j in idx[b0:b0 + args.rollout_per_device_batch_size]])
                   for b0 in range(0, len(idx), args.rollout_per_device_batch_size)]
        files.append(write_rollout_json(args.output_dir, step, rollout_step(batches, tok, sample, feedback), rank=rank))
        log(f"step {step}: {len(idx)} rollouts -> {files[-1]}")
    return files


def main_rollout(argv: Optional[List[str]] = None) -> None:
    argv = sys.argv[1:] if argv is None else argv
    ns, _ignored = rollout_parser().parse_known_args(argv)
    from .cli import load_yaml_defaults
    load_yaml_defaults(ns, argv)          # --cfg fills what the command line left unset
    run_rollout(ns)


def main_eval(argv: Optional[List[str]] = None) -> None:
    from .eval_generate import answer_questions, question_chunk
    ns, _ignored = eval_parser().parse_known_args(sys.argv[1:] if argv is None else argv)
    if os.path.exists(ns.answers_file):
        print(f"{ns.answers_file} already exists. Please delete it first.")
        raise SystemExit(1)
    engine, adapter, tok = _engine(ns.model_path, ns.synthetic, ns.qlora_path if ns.use_qlora else None, ns.lora_rank, ns.lora_alpha)
    questions = [json.loads(q) for q in open(os.path.expanduser(ns.question_file), "r")]
    questions = question_chunk(questions, ns.num_chunks, ns.chunk_idx)
    n = answer_questions(engine, tok, questions, ns.image_folder, os.path.expanduser(ns.answers_file), adapter=adapter,
                         model_id=os.path.basename(os.path.normpath(ns.model_path)), temperature=ns.temperature, top_p=ns.top_p,
                         short_eval=ns.short_eval, test_prompt=ns.test_prompt, image_size=engine.d.image_size,
                         pad_to_square=ns.image_aspect_ratio == "pad", merge_adapter=ns.merge_adapter, batch_size=ns.batch_size)
    print(f"{n} answers -> {ns.answers_file}")


if __name__ == "__main__":
    main_rollout()
