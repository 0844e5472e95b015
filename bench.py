#!/usr/bin/env python3
"""Headline benchmark: preference-pairs/sec, LLaVA-1.5-7B LoRA DPO, seq_len 512 (query 128 + response 384,
L = 1087 LLM positions), synthetic image+text batches, random-init weights of the 7B architecture.

    python bench.py --gpus N --steps K --warmup W          # N > 1 without a launcher environment: re-executes ITSELF through
                                                           # torch.distributed.run (one rank per GPU, RCCL, rendezvous on 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W     # same thing, launched from outside
    python bench.py --gpus 2 --dry-run                     # launcher + rendezvous + bucketed exchange self-check, no kernels (gloo on a CPU box)

One STEP = one optimizer step on every rank: `accum` (default 1) micro-batches of `pairs` (default 22) (image, chosen, rejected)
pairs -> vision encode (once per image) -> frozen-reference forward (no grad) -> policy forward (activations
resident) -> token-level DPO loss -> LoRA backward (the RCCL exchange of a finished bucket of layers starts inside it) ->
global-norm clip + AdamW -> refresh of the transposed LoRA copies.  Nothing is skipped or cached across steps; the micro-batches
cycle through a pool of 8 different synthetic batches (SURVEY.md §8d lengths).  Default engine: the sequence-level C entry points
(opadpo_ctx) on RAGGED rows - padding positions are not rows of any kernel (`--padded` keeps the reference's padded rows,
`--op-level` the Python sequencing of single-kernel calls).
Weak scaling: per-GPU work is fixed, value = total pairs of all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "opa-dpo_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs across processes on this driver

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF figure is 2:1 sparse)


def _pmc_traffic():
    """HBM-side bytes per gemm_nt launch (launch-weighted mean over the gemm_nt kernels of one optimizer step) from the committed
    rocprofv3 --pmc passes of THIS workload and THIS round's kernels (tools/pmc_bench.sh: FETCH_SIZE and WRITE_SIZE in separate
    passes, FETCH_SIZE x2 = the gfx950 correction; rocprofv3 cannot run inside the timed process).  Newest profiles/r*_pmc_traffic.json."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")))
    for f in reversed(files):
        try:
            return json.load(open(f))["gemm_nt_avg_hbm_bytes_per_launch"], os.path.basename(f)
        except Exception:
            continue
    return None, None


def _parity_record():
    """HIP-vs-oracle parity of the benchmarked path at the benchmarked DEPTH (tests/test_fulldepth_parity_gpu.py: full LLaVA-1.5-7B,
    32 layers, 2 pairs at seq512, product path = context API, packed ragged rows, merged reference adapter; per-token log-prob error
    against oracle/llava_ref.py in fp32): the newest committed profiles/r*_parity_fulldepth.json (the test writes it under
    gpurun_out/; a CPU-hours oracle cannot run inside the timed process)."""
    import glob
    for f in reversed(sorted(glob.glob(os.path.join(REPO, "profiles", "r*_parity_fulldepth.json")))):
        try:
            r = json.load(open(f))
            out = dict(r["bench_line"])
            out["source"] = os.path.basename(f)
            return out
        except Exception:
            continue
    return None


def _parity_in_run():
    """The same record PRODUCED IN THIS RUN (round 6): a child process runs tests/test_fullsize_gpu.py::test_p7_full_depth_32_layers_against_the_oracle on this
    GPU after the timed region (HIP product path vs three oracle passes - fp32, bf16-emulating, merged - all evaluated in the child; ~1 min
    with its import and weight initialisation) and its report gpurun_out/parity_fulldepth.json is read back.  The oracle is the CHECKER here, never
    part of what is timed.  Falls back to the committed record (source says which) when the child fails."""
    import subprocess
    rep = os.path.join(REPO, "gpurun_out", "parity_fulldepth.json")
    t0 = time.time()
    try:
        if os.path.exists(rep):
            os.remove(rep)
        env = dict(os.environ, GRAFT_REPO_ROOT=REPO)
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                            os.path.join(REPO, "tests", "test_fullsize_gpu.py") + "::test_p7_full_depth_32_layers_against_the_oracle"],
                           capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
        j = json.load(open(rep))
        out = dict(j["bench_line"])
        out.update({"source": "this run (child process: tests/test_fullsize_gpu.py::test_p7_full_depth_32_layers_against_the_oracle)",
                    "assertions_passed": r.returncode == 0, "oracle_device": j.get("oracle_device"), "child_wall_s": time.time() - t0,
                    "oracle_seconds": {k: j[k] for k in ("oracle_fp32_seconds", "oracle_emu_bf16_seconds", "oracle_emu_merged_seconds") if k in j}})
        return out
    except Exception as e:
        out = _parity_record() or {}
        out["in_run_error"] = repr(e)[:200]
        return out


def _cpu_full_record():
    """The oracle timed DIRECTLY at full depth on the GPU box's host cores (tools/cpu_baseline_full.py: one whole pair, 32 layers, nothing
    extrapolated; minutes of CPU work, so it is a committed record - newest profiles/r*_cpu_baseline_full.json - not part of this run)."""
    import glob
    for f in reversed(sorted(glob.glob(os.path.join(REPO, "profiles", "r*_cpu_baseline_full.json")))):
        try:
            r = json.load(open(f))
            return {"value": r["value"], "unit": r["unit"], "seconds_per_pair": r["seconds_per_pair"], "cores": r["cores"], "extrapolated": False,
                    "sample": r["sample"], "source": os.path.basename(f)}
        except Exception:
            continue
    return None


def cpu_baseline_config_p(dims_kw, n_layers=1):
    """SURVEY.md §8(d) config (1) timed DIRECTLY on the host cores: 8 pairs, query 32 + response 96 (L = 703), fp32, world 1, the whole
    path of a pair - CLIP tower + projector once per image, frozen-reference forward (no grad) and policy forward on chosen + rejected,
    token-level DPO loss, backward into the LoRA tensors - on the oracle (the parity-checked CPU restatement of the reference's
    forward), with the decoder truncated to `n_layers` of the 32 (stated; §8d allows the truncation for this config)."""
    from oracle import dpo_ref as DR
    from oracle import llava_ref as LR
    torch.set_num_threads(min(os.cpu_count(), 64))
    d = LR.LlavaDims(**{**dims_kw, "n_layers": n_layers})
    W = LR.init_weights(d, seed=0)
    lora_p = {k: v.requires_grad_(True) for k, v in LR.init_lora(d, seed=1, with_vision=False).items()}
    lora_r = LR.init_lora(d, seed=2, with_vision=False)
    g = torch.Generator().manual_seed(0)
    B, Q, T = 8, 32, 96
    images = torch.randn(B, 3, d.image_size, d.image_size, generator=g)
    queries = torch.randint(3, d.vocab, (B, Q), generator=g)
    qmask = torch.ones(B, Q, dtype=torch.bool)
    resp = {}
    for b in range(B):
        n_pad = int(torch.randint(0, Q // 2 + 1, (1,), generator=g))
        queries[b, :n_pad] = 0
        qmask[b, :n_pad] = False
        queries[b, int(torch.randint(n_pad, Q, (1,), generator=g))] = -200
    for k in ("chosen_response", "rejected_response"):
        ids = torch.randint(3, d.vocab, (B, T), generator=g)
        for b in range(B):
            ln = int(torch.randint(T // 6, T, (1,), generator=g))
            ids[b, ln] = 2
            ids[b, ln + 1:] = 0
        resp[k] = ids
    from oracle.dpo_ref import policy_head, stack_policy_inputs

    def fwd(lora, feats):          # AutoregressivePolicy.forward on precomputed image features (once per image, like the product path)
        ids, mask = stack_policy_inputs(queries, qmask, resp)
        lp, _ = policy_head(LR.llava_logits(ids, mask, None, W, lora, d, feats=feats.repeat(2, 1, 1)), ids, Q, T, 1.0)
        return {"chosen_response_logprobs": lp[:B], "rejected_response_logprobs": lp[B:]}
    t0 = time.time()
    with torch.no_grad():
        feats = LR.image_features(images, W, None, d)
        r = fwd(lora_r, feats)
    o = fwd(lora_p, feats)
    loss, _, _ = DR.plain_pair_loss(DR.DPOConfig(), o["chosen_response_logprobs"], o["rejected_response_logprobs"],
                                    r["chosen_response_logprobs"], r["rejected_response_logprobs"])
    loss.backward()
    dt = time.time() - t0
    return {"pairs": B, "query_len": Q, "response_len": T, "L": Q + T + d.n_patches - 1, "n_layers_run": n_layers, "n_layers_model": dims_kw["n_layers"],
            "seconds": dt, "pairs_per_s_truncated_model": B / dt,
            "note": "directly timed, nothing extrapolated: vision tower once per image, decoder truncated to n_layers_run, head, DPO loss, LoRA backward"}


class GpuSampler:
    """Background sampler of the GPU's clock / power / temperature during a sustained run (--sustained): `rocm-smi --json` once per
    `period` seconds (the sysfs hwmon nodes are not mounted in every container).  Host-side only; nothing touches the stream."""

    def __init__(self, period=2.0, device=0):
        import threading
        self.period, self.device, self.samples, self._stop = period, device, [], threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import subprocess
        t0 = time.time()
        while not self._stop.is_set():
            try:
                r = subprocess.run(["rocm-smi", "-d", str(self.device), "--showclocks", "--showpower", "--showtemp", "--json"],
                                   capture_output=True, text=True, timeout=10)
                js = json.loads(r.stdout)
                card = next(iter(js.values()))
                rec = {"t": time.time() - t0}
                import re
                if not self.samples:      # the raw clock fields once, so that a parsing miss on another rocm-smi build stays visible
                    rec["raw_clock_fields"] = {k: str(v) for k, v in card.items() if "clk" in k.lower()}
                for k, v in card.items():
                    kl = k.lower()
                    num = re.search(r"(\d+(?:\.\d+)?)\s*mhz", str(v).lower())
                    if "sclk" in kl and num:
                        rec["sclk_mhz"] = float(num.group(1))
                    elif "mclk" in kl and num:
                        rec["mclk_mhz"] = float(num.group(1))
                    elif "power" in kl and "(w)" in kl:
                        rec["power_w"] = float(v)
                    elif "temperature" in kl and ("junction" in kl or "hotspot" in kl):
                        rec["temp_c"] = float(v)
                self.samples.append(rec)
            except Exception as e:  # keep sampling; the record says what went wrong
                self.samples.append({"t": time.time() - t0, "error": repr(e)[:120]})
            self._stop.wait(self.period)

    def __enter__(self):
        self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._th.join(timeout=15)


def cpu_baseline(dims_kw, q_len, t_len):
    """Reported CPU baseline: the oracle (the parity-checked CPU restatement of the reference's forward, kind
    'port') on the host cores, on a bounded sample: ONE of the 32 decoder layers at 7B width, one pair =
    4 sequence forwards (2 of them under autograd + backward through the LoRA tensors) at L = 1087, run twice (best of 2); scaled by
    n_layers to a full-model pair (head + vision, <2 % of the FLOPs, not included)."""
    from oracle import llava_ref as LR
    torch.set_num_threads(min(os.cpu_count(), 64))     # torch CPU GEMMs at these sizes stop scaling (and slow down) beyond ~64 threads
    d = LR.LlavaDims(**dims_kw)
    d1 = LR.LlavaDims(**{**dims_kw, "n_layers": 1})
    g = torch.Generator().manual_seed(0)
    W, lora = {}, {}
    p = "model.layers.0."
    for lin in LR.LLM_LINEARS:
        o, i = LR.llm_linear_shape(d1, lin)
        W[p + lin + ".weight"] = torch.randn(o, i, generator=g) * 0.02
        lora[f"base_model.model.{p}{lin}.lora_A.weight"] = (torch.randn(d.lora_r, i, generator=g) * 0.01).requires_grad_(True)
        lora[f"base_model.model.{p}{lin}.lora_B.weight"] = (torch.randn(o, d.lora_r, generator=g) * 0.01).requires_grad_(True)
    W[p + "input_layernorm.weight"] = torch.ones(d.hidden)
    W[p + "post_attention_layernorm.weight"] = torch.ones(d.hidden)
    L = q_len + t_len + d.n_patches - 1
    x = torch.randn(1, L, d.hidden, generator=g)
    km = torch.ones(1, L, dtype=torch.bool)
    def one_pair():      # both sequences of the pair, each: reference forward (no grad) + policy forward + LoRA backward
        t0 = time.time()
        for _ in range(2):
            with torch.no_grad():
                LR.llama_decoder(x, km, W, {k: v.detach() for k, v in lora.items()}, d1)
            y = LR.llama_decoder(x, km, W, lora, d1)
            y.sum().backward()
            for v in lora.values():
                v.grad = None
        return time.time() - t0
    dt = min(one_pair(), one_pair())                                                   # ~10 s of CPU work in total
    pair_s = dt * d.n_layers
    return {"value": 1.0 / pair_s, "unit": "pairs/s", "cores": min(os.cpu_count(), 64), "kind": "port", "extrapolated": True,
            "sample": f"EXTRAPOLATED from 1 of {d.n_layers} decoder layers at 7B width (fp32 torch CPU oracle), L={L}, "
                      f"both sequences of a pair: reference forward + policy forward + LoRA backward = {dt:.1f} s per layer-pair (best of 2), scaled x{d.n_layers}"}


def rollout_leg(eng, d, dev, batches=(8, 64), steps=64):
    """BASELINE.json configs[4] (on-policy rollout, opadpo/generator_models/online_generator.py:292-309) in the driver's line:
    LLaVA-1.5-7B KV-cache sampling decode (top-k 30 / top-p 0.95, no LoRA = the shipped rollout config), prefill L = 703, `steps`
    decode steps per batch size.  HBM-bound: bytes per step = bf16 weights of every linear + lm_head (read once per
    step) + the KV cache read of every sequence at the mean context of the timed steps; frac = bytes / time / 8 TB/s."""
    from opadpo_amd.generate import Generator
    from opadpo_amd.synth import synth_pairs
    out = {}
    gen = Generator(eng, None, fuse_swiglu=True)
    wbytes = 2 * (d.n_layers * (4 * d.hidden ** 2 + 3 * d.hidden * d.ffn) + d.vocab * d.hidden)
    for B in batches:
        p = synth_pairs(d, B, 128, 8, seed=77, device=dev)
        feats = eng.encode_images(p["images"])
        t = {}
        for n in (2, steps + 2):
            kw = dict(image_feats=feats, max_new_tokens=n, top_k=30, top_p=0.95, suppress_eos=True)
            gen.generate(p["queries"], p["queries_attn_masks"], seed=1, **kw)            # warm (kernel attribute calls, allocator)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gen.generate(p["queries"], p["queries_attn_masks"], seed=2, **kw)
            torch.cuda.synchronize()
            t[n] = time.perf_counter() - t0
        per_step = (t[steps + 2] - t[2]) / steps
        ctx = 128 + d.n_patches - 1 + 2 + steps / 2
        kv = 2 * 2 * d.n_layers * d.hidden * ctx * B
        # the shipped rollout length (response_len 896, online_generator.py:292-309): prefill + 896 decode steps, end to end, one run
        n_full = 896
        kwf = dict(image_feats=feats, max_new_tokens=n_full, top_k=30, top_p=0.95, suppress_eos=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gen.generate(p["queries"], p["queries_attn_masks"], seed=3, **kwf)
        torch.cuda.synchronize()
        t_full = time.perf_counter() - t0
        prefill_ms = max(t[2] - 2 * per_step, 0.0) * 1e3
        out[f"b{B}"] = {"batch": B, "decode_ms_per_step": per_step * 1e3, "tokens_per_s": B / per_step, "prefill_plus_2_steps_ms": t[2] * 1e3,
                        "prefill_ms": prefill_ms, "prefill_tokens_per_s": B * (128 + d.n_patches - 1) / max(prefill_ms * 1e-3, 1e-9),
                        "end_to_end_896_new_tokens": {"seconds": t_full, "tokens_per_s": B * n_full / t_full,
                                                      "ms_per_step_mean": (t_full - prefill_ms * 1e-3) / n_full * 1e3, "ctx_final": 128 + d.n_patches - 1 + n_full},
                        "bytes_per_step_GB": (wbytes + kv) / 1e9, "weight_GB": wbytes / 1e9, "kv_GB": kv / 1e9,
                        "hbm_frac": (wbytes + kv) / per_step / 8e12, "steps_timed": steps}
        del feats, p
        torch.cuda.empty_cache()
    return {"workload": "LLaVA-1.5-7B rollout decode, query 128 -> prefill L=703, top-k 30 / top-p 0.95, no LoRA (shipped rollout config), "
                        "one C++ launch loop per token inside the context (opadpo_decode_run)", "bound": "hbm", "peak_GBps": 8000.0, **out}


def extra_config(which):
    """Sub-records of the default line for the BASELINE.json configurations the headline does not cover (measured AFTER the timed region,
    in a child process on the same GPU; never part of `value`):
      thirteen_b : configs[3]'s model at 1 GPU - LLaVA-1.5-13B LoRA DPO, 12 packed pairs per step, 6 timed steps (this file, --model 13b);
      recipe     : the reference's NATIVE unit at the shipped recipe's lengths (run/train_opa_dpo.sh:39-50: query 128, response 896, 3 responses
                   per sample + 2 on the CoPO-masked image, AncPO; DPOTrainer.step() of the product, tools/sample_bench.py) - samples/s."""
    import subprocess
    env = dict(os.environ)
    try:
        if which == "thirteen_b":
            cmd = [sys.executable, os.path.abspath(__file__), "--model", "13b", "--steps", "6", "--warmup", "1", "--batch-pool", "3",
                   "--no-cpu-baseline", "--no-rollout", "--no-side-legs", "--no-exchange-probe", "--no-extra-configs"]
        else:
            cmd = [sys.executable, os.path.join(REPO, "tools", "sample_bench.py")]
            env.update(SB_T="896", SB_BATCH=env.get("OPADPO_BENCH_RECIPE_BATCH", "4"), SB_STEPS="6")
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"rc={r.returncode}", "stderr_tail": r.stderr[-400:]}
        j = json.loads(lines[-1])
        if which == "thirteen_b":
            return {"workload": j["config"]["workload"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"],
                    "pairs_per_step": j["config"]["global_pairs_per_step"], "roofline_frac": j["roofline"]["frac"] if j.get("roofline") else None,
                    "gemm_nt_TFLOPs": j["roofline"]["achieved"] if j.get("roofline") else None,
                    "executed_flops_per_pair_TF": j["executed_flops_per_pair_TF"], "mfma_roofline_frac_end_to_end": j["mfma_roofline_frac_end_to_end"],
                    "hbm_peak_allocated_GB": j["hbm_peak_allocated_GB"], "child_wall_s": time.time() - t0}
        return {"workload": j["metric"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j.get("steps"),
                "samples_per_step": j["samples_per_step"], "response_layout": j["response_layout"], "sequence_forwards_per_sample": j["sequence_forwards_per_sample"],
                "hbm_peak_allocated_GB": j["hbm_peak_allocated_GB"], "child_wall_s": time.time() - t0}
    except Exception as e:  # a side record never costs the headline
        return {"error": repr(e)}


def exchange_probe(numel, dev, layer_numel, n_layers):
    """1-rank timing of the data-parallel exchange path of one optimizer step on THIS GPU (no multi-GPU node is
    visible to this run): the fp32 -> bf16 staging casts, the per-bucket reduce-scatter and the per-bucket all-gather of FlatAdamW
    over a 1-rank RCCL group (device-local copies at world 1 - the floor of the exchange, not an xGMI number)."""
    from opadpo_amd.optim import FlatAdamW, layer_buckets
    made = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        made = True
    prev = os.environ.get("OPADPO_FORCE_COLLECTIVES")
    os.environ["OPADPO_FORCE_COLLECTIVES"] = "1"
    try:
        master = torch.zeros(numel, dtype=torch.float32, device=dev)
        grad = torch.randn(numel, dtype=torch.float32, device=dev) * 1e-3
        work = torch.zeros(numel, dtype=torch.bfloat16, device=dev)
        opt = FlatAdamW(master, grad, work, lr=1e-6, max_grad_norm=1.0, mode="zero1", bucket_bounds=layer_buckets(layer_numel, n_layers, 4))
        res = {}
        for it in range(3):
            torch.cuda.synchronize()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            for bi in range(len(opt.buckets) - 1, -1, -1):
                opt.launch_bucket(bi)
            opt.prepare()
            e[1].record()
            opt.apply()
            e[2].record()
            torch.cuda.synchronize()
            res = {"reduce_scatter_and_norm_ms": e[0].elapsed_time(e[1]), "adamw_and_all_gather_ms": e[1].elapsed_time(e[2])}
        res.update({"world": 1, "buckets": len(opt.buckets), "wire_dtype": "bf16", "wire_bytes_per_rank_GB": 2 * numel / 1e9,
                    "note": "1-rank RCCL group on one MI355X (OPADPO_FORCE_COLLECTIVES=1): device-local floor of the exchange path, launched per bucket"})
        return res
    finally:
        if prev is None:
            os.environ.pop("OPADPO_FORCE_COLLECTIVES", None)
        else:
            os.environ["OPADPO_FORCE_COLLECTIVES"] = prev
        if made:
            dist.destroy_process_group()

def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dist_timeout():
    import datetime
    return datetime.timedelta(seconds=int(os.environ.get("OPADPO_DIST_TIMEOUT_S", "180")))


def rank_info(rank, local, use_gpu=True):
    """What THIS rank is bound to, as the runtime reports it (gathered over the process group into the line's `dist.ranks`)."""
    info = {"rank": rank, "local_rank": local, "pid": os.getpid(), "device": "cpu"}
    if use_gpu and torch.cuda.is_available():
        pr = torch.cuda.get_device_properties(local)
        # the device's identity as the runtime reports it: the UUID, else its PCI address, else the ordinal (never empty: check_dist_record counts
        # distinct identities over the ranks, and a missing attribute on some torch build must not cost the first unattended N > 1 run its line)
        uid = str(getattr(pr, "uuid", "") or "")
        if not uid:
            uid = "pci:%s:%s:%s" % (getattr(pr, "pci_domain_id", "?"), getattr(pr, "pci_bus_id", "?"), getattr(pr, "pci_device_id", "?")) \
                if hasattr(pr, "pci_bus_id") else f"ordinal:{local}"
        info.update({"device": pr.name, "uuid": uid, "arch": getattr(pr, "gcnArchName", "") or "unknown",
                     "compute_units": pr.multi_processor_count, "hbm_GB": pr.total_memory / 1e9})
    return info


DIST_FIELDS = ("backend", "collective_library", "world_size", "allreduce_of_ones", "ranks", "ms_per_step_min_over_ranks", "ms_per_step_max_over_ranks")
RANK_FIELDS = ("rank", "local_rank", "pid", "device")
RANK_FIELDS_GPU = ("uuid", "arch", "compute_units", "hbm_GB")


def check_dist_record(rec, world, on_gpu):
    """Schema of the `dist` record of an N > 1 line (real run and --dry-run print the SAME fields; tests/test_bench_launcher_cpu.py): a SCALE line must
    not come back without the all-reduce of ones (= the world size, through the collective library itself) and what every rank was bound to (on GPUs:
    the device UUIDs, all different).  Raises ValueError - rank 0 then fails the run loudly instead of printing an unverifiable line."""
    miss = [k for k in DIST_FIELDS if k not in rec]
    if miss:
        raise ValueError(f"dist record lacks {miss}")
    if rec["world_size"] != world or float(rec["allreduce_of_ones"]) != float(world):
        raise ValueError(f"dist record: world_size {rec['world_size']} / all-reduce of ones {rec['allreduce_of_ones']} for a {world}-rank launch")
    ranks = rec["ranks"]
    if len(ranks) != world or sorted(x.get("rank", -1) for x in ranks) != list(range(world)):
        raise ValueError(f"dist record: ranks {[x.get('rank') for x in ranks]} for world size {world}")
    for x in ranks:
        need = RANK_FIELDS + (RANK_FIELDS_GPU if on_gpu else ())
        m = [k for k in need if k not in x or x[k] in (None, "")]
        if m:
            raise ValueError(f"dist record: rank {x.get('rank')} lacks {m}")
    if len({x["pid"] for x in ranks}) != world:
        raise ValueError("dist record: ranks share a pid")
    if on_gpu and world > 1 and len({x["uuid"] for x in ranks}) != world and os.environ.get("OPADPO_BENCH_SHARE_DEVICE") != "1":
        raise ValueError(f"dist record: {world} ranks on {len({x['uuid'] for x in ranks})} distinct device UUID(s)")
    return True


def preflight(world, dev, backend):
    """First collective of the run, right after the rendezvous and BEFORE any model state exists: the sum of ones over the group must be the
    world size.  Bounded by the process group's timeout (OPADPO_DIST_TIMEOUT_S, default 180 s) so a rank that never arrives ends the run with a
    reason instead of a hang at the first barrier of the timed region."""
    if world <= 1:
        return 1.0
    probe = torch.ones(1, device=dev)
    dist.all_reduce(probe)
    if float(probe) != float(world):
        raise RuntimeError(f"pre-flight all-reduce of ones over {backend} returned {float(probe)} for world size {world}")
    return float(probe)


def collective_lib_version():
    try:
        v = torch.cuda.nccl.version()
        return "RCCL/NCCL " + ".".join(str(x) for x in v)
    except Exception as e:      # CPU-only build or gloo
        return None


def self_launch(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment (RANK / WORLD_SIZE unset): start the N ranks the way
    the reference's run/train_opa_dpo.sh:96-100 starts the trainer (`torchrun --nproc-per-node=$GPUS_PER_NODE`, one process per GPU) - `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port <free port> bench.py <same arguments>` - and hand its exit code back.  Rank 0's JSON line goes
    straight through to this process's stdout (the children inherit it)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    print("[bench] launching", n, "ranks:", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def dry_run(args, world, rank, local):
    """`--dry-run`: everything of the N > 1 path that is not a kernel - the launcher, the rendezvous, rank -> device binding, the
    barrier-bracketed max-over-ranks timing and the bucketed ZeRO-1 exchange of optim.FlatAdamW (reduce-scatter per bucket launched
    last bucket first like the backward's layer hook does, sharded clip + AdamW, all-gather) on a flat buffer of the model's
    layer-major LoRA layout, with torch stand-ins for the two HIP update kernels.  Self-check: every rank ends with the SAME bf16
    working copy, and it equals the 1-rank step on the rank-averaged gradient.  Runs on gloo / CPU when there are fewer GPUs than
    ranks (tests/test_bench_launcher_cpu.py), on RCCL otherwise.  Prints the bench line with "dry_run": true and value null."""
    import math
    from opadpo_amd.dims import LlavaDims, lora_param_count
    from opadpo_amd.optim import FlatAdamW, layer_buckets, torch_cast
    use_gpu = torch.cuda.is_available() and torch.cuda.device_count() >= world and args.backend != "gloo"
    dev = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(local)
    backend = "nccl" if use_gpu else "gloo"
    if world > 1:
        dist.init_process_group(backend, timeout=_dist_timeout(), **({"device_id": dev} if use_gpu else {}))
    if os.environ.get("OPADPO_BENCH_FAIL_RANK") == str(rank):      # tests: a rank that dies after the rendezvous (the others sit in the pre-flight)
        raise RuntimeError(f"injected failure on rank {rank} (OPADPO_BENCH_FAIL_RANK)")
    preflight(world, dev, backend)

    def t_sumsq(g, out):
        out += (g.double() ** 2).sum().float()

    def t_adamw(p, g, m, v, p_bf16, *, lr, beta1, beta2, eps, weight_decay, step, sumsq, max_norm, grad_div):
        scale = grad_div
        if sumsq is not None and max_norm:
            norm = math.sqrt(float(sumsq)) * grad_div
            scale *= min(1.0, max_norm / (norm + 1e-6))
        gi = g.float() * scale
        m.mul_(beta1).add_(gi, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(gi, gi, value=1 - beta2)
        p.addcdiv_(m, v.sqrt() / math.sqrt(1 - beta2 ** step) + eps, value=-lr / (1 - beta1 ** step))
        p_bf16.copy_(p.to(p_bf16.dtype))
    kw = dict(sumsq_fn=t_sumsq, adamw_fn=t_adamw, cast_fn=torch_cast)
    d = {"7b": LlavaDims.llava15_7b, "13b": LlavaDims.llava15_13b, "tiny": LlavaDims.tiny}[args.model]()
    numel = lora_param_count(d)
    layer_numel = numel // d.n_layers
    bounds = layer_buckets(layer_numel, d.n_layers, 1 if args.model == "tiny" else 4)
    g0 = torch.Generator().manual_seed(7)
    p0 = torch.randn(numel, generator=g0) * 0.02

    def grad_of(r, s):
        return torch.randn(numel, generator=torch.Generator().manual_seed(1000 * s + r)) * 0.01
    master = p0.clone().to(dev)
    grad = torch.zeros(numel, device=dev)
    work = master.to(torch.bfloat16)
    opt = FlatAdamW(master, grad, work, lr=1e-3, max_grad_norm=1.0, mode=args.optimizer_mode, bucket_bounds=bounds,
                    exchange_dtype=torch.float32, **kw)

    def sync():
        if world > 1:
            dist.barrier()
        if use_gpu:
            torch.cuda.synchronize()
    steps = max(1, args.steps)
    sync()
    t0 = time.perf_counter()
    for s in range(steps):
        grad.copy_(grad_of(rank, s))
        for bi in range(len(opt.buckets) - 1, -1, -1):        # the order the backward's layer hook launches them in
            opt.launch_bucket(bi)
        opt.step()
        opt.zero_grad()
    sync()
    dt = time.perf_counter() - t0
    # 1-rank step on the rank-averaged gradient (the parity definition of SURVEY.md §8e), on the host
    m1, w1 = p0.clone(), p0.to(torch.bfloat16)
    ref = FlatAdamW(m1, torch.zeros(numel), w1, lr=1e-3, max_grad_norm=1.0, mode="allreduce", local_only=True, **kw)
    for s in range(steps):
        ref.grad.copy_(sum(grad_of(r, s) for r in range(world)))
        ref.step(grad_accum_div=world)
    err = float((work.float().cpu() - w1.float()).abs().max())
    tt = torch.tensor([dt, -dt, err], dtype=torch.float64, device=dev)
    seen, same = [rank_info(rank, local, use_gpu)], True
    ones = 1.0
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        seen = [None] * world
        dist.all_gather_object(seen, rank_info(rank, local, use_gpu))
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)
        ones = float(probe)
        chk = [torch.empty_like(work) for _ in range(world)]
        dist.all_gather(chk, work)
        same = all(torch.equal(chk[0], c) for c in chk[1:])
    ok = same and float(tt[2]) <= 2.0 ** -7 * float(p0.abs().max()) and ones == world
    out = None
    if rank == 0:
        out = {"metric": "preference-pairs/sec LLaVA-1.5-7B LoRA DPO seq512", "value": None, "unit": "pairs/s", "n_gpus": world,
               "steps": steps, "warmup": 0, "ms_per_step": float(tt[0]) / steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic", "dry_run": True,
               "config": {"workload": f"DRY RUN (no kernels): launcher + rendezvous + bucketed {args.optimizer_mode} exchange of a {numel}-element "
                                      f"flat LoRA buffer ({args.model} layout, {len(opt.buckets)} buckets) with torch stand-ins for the update kernels",
                          "parallelism": f"dp{world}"},
               "dist": {"backend": backend if world > 1 else None, "collective_library": collective_lib_version() if use_gpu else None,
                        "world_size": world, "allreduce_of_ones": ones, "ranks": seen,
                        "ms_per_step_min_over_ranks": -float(tt[1]) / steps * 1e3, "ms_per_step_max_over_ranks": float(tt[0]) / steps * 1e3,
                        "replicas_identical_after_step": same, "max_abs_diff_vs_1_rank_step_on_averaged_gradient": float(tt[2]), "self_check": ok},
               "roofline": None, "cpu_baseline": None}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        check_dist_record(out["dist"], world, use_gpu)
        print(json.dumps(out), flush=True)
    return 0 if ok else 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    # chosen + rejected of a pair are packed on their shared image + query prefix: one row of 703 + 2*384 = 1471 positions per
    # pair and pass (the reference stacks 2 x 1087).  22 pairs = 32362 rows = 127 M-tiles of 256: every big GEMM of the step
    # then covers (almost) a whole number of 256-CU rounds (N=4096: 2032 tiles = 7.94 rounds).  --no-pack: the reference's
    # layout, 15 pairs = 30 x 1087 = 32610 rows = 128 M-tiles.
    ap.add_argument("--pairs", type=int, default=int(os.environ.get("OPADPO_BENCH_PAIRS", 0)), help="pairs per micro-batch per GPU (0: 22 packed / 15 stacked; 13b: 12 / 8)")
    ap.add_argument("--no-pack", action="store_true", help="stack chosen / rejected as separate sequences (reference layout)")
    ap.add_argument("--accum", type=int, default=int(os.environ.get("OPADPO_BENCH_ACCUM", 1)), help="micro-batches per optimizer step")
    ap.add_argument("--model", default=os.environ.get("OPADPO_BENCH_MODEL", "7b"), choices=["7b", "13b", "tiny"])
    ap.add_argument("--optimizer-mode", default="zero1", choices=["allreduce", "zero1"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch-pool", type=int, default=8, help="number of different synthetic micro-batches cycled over the steps")
    ap.add_argument("--padded", action="store_true", help="keep the reference's padded rows (padding positions computed and thrown away) instead of ragged rows")
    ap.add_argument("--op-level", action="store_true", help="sequence the kernels from Python (model.LlavaEngine) instead of the opadpo_ctx entry points")
    ap.add_argument("--ctx-flags", type=int, default=-1, help="opadpo_ctx_set_flags use_tr word for A/B runs (-1: defaults; bit 6: SwiGLU backward fused into the dgrad epilogue, bit 7: top decoder layer on every row)")
    ap.add_argument("--no-rollout", action="store_true", help="skip the rollout (decode) sub-record")
    ap.add_argument("--sustained", action="store_true", help="per-step stream time stamps + a background rocm-smi sampler (clock / power / temperature): the `sustained` record (config 2: --steps 220 = 4.8k pairs)")
    ap.add_argument("--no-side-legs", action="store_true", help="skip the dense-batch and exchange-overlap sub-records (measured after the timed region)")
    ap.add_argument("--no-exchange-probe", action="store_true", help="skip the 1-rank timing of the gradient exchange path")
    ap.add_argument("--no-merge-ref", action="store_true", help="keep the frozen reference adapter unmerged (K-concatenated LoRA in the no-grad pass too)")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the `thirteen_b` (BASELINE.json configs[3] at 1 GPU) and `recipe` (the reference's native unit: 3 responses, response_len 896, CoPO) sub-records (each runs in a child process after the 7B state is released)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous / exchange self-check without kernels (gloo on a box with fewer GPUs than ranks)")
    ap.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"], help="force the process-group backend (gloo: --dry-run on CPU, or the one-device test of the N > 1 path)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # no launcher environment: start the ranks ourselves (the driver calls `python bench.py --gpus N` the same way as for N = 1)
        if not args.dry_run and torch.cuda.device_count() < args.gpus and os.environ.get("OPADPO_BENCH_SHARE_DEVICE") != "1":
            print(f"[bench] --gpus {args.gpus} but this node shows {torch.cuda.device_count()} GPU(s): nothing to launch "
                  "(--dry-run exercises the launcher and the exchange on gloo / CPU)", file=sys.stderr)
            sys.exit(2)
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    pack = not args.no_pack
    if args.pairs <= 0:
        args.pairs = (22 if pack else 15) if args.model != "13b" else (12 if pack else 8)      # 13B: activations of 12 packed pairs fit the 288 GB

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but the launcher environment says WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        sys.stdout.flush()
        rc = dry_run(args, world, rank, local)
        sys.stdout.flush()
        os.dup2(2, 1)
        sys.exit(rc)
    share = os.environ.get("OPADPO_BENCH_SHARE_DEVICE") == "1"      # tests only: every rank on cuda:0 (RCCL refuses two ranks per device: gloo wire)
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or os.environ.get("OPADPO_FORCE_COLLECTIVES") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if share or args.backend == "gloo":
            dist.init_process_group("gloo", timeout=_dist_timeout())
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=_dist_timeout())
        if os.environ.get("OPADPO_BENCH_FAIL_RANK") == str(rank):
            raise RuntimeError(f"injected failure on rank {rank} (OPADPO_BENCH_FAIL_RANK)")
        preflight(world, dev if not (share or args.backend == "gloo") else torch.device("cpu"), dist.get_backend())

    from opadpo_amd import lib as L
    from opadpo_amd.dims import LlavaDims, lora_param_count, pair_flops, pair_flops_packed
    from opadpo_amd.losses import DPOArgs, pair_loss
    from opadpo_amd.ctx import CtxEngine
    from opadpo_amd.model import BaseWeights, LlavaEngine, LoraAdapter
    from opadpo_amd.optim import FlatAdamW, layer_buckets
    from opadpo_amd.policy import AutoregressivePolicy
    from opadpo_amd.synth import init_lora, init_weights, synth_pairs
    L.load()

    d = {"7b": LlavaDims.llava15_7b, "13b": LlavaDims.llava15_13b, "tiny": LlavaDims.tiny}[args.model]()
    q_len, t_len = (128, 384) if args.model != "tiny" else (16, 16)
    W = init_weights(d, seed=0, device=dev)
    base = BaseWeights(d, W, dev, need_backward=True)
    del W
    eng = LlavaEngine(base) if args.op_level else CtxEngine(base, ragged=not args.padded)      # default: ONE C call per pass (opadpo_seq_logprobs_fwd / _bwd)
    if args.ctx_flags >= 0 and not args.op_level:
        eng.set_flags(use_tr=args.ctx_flags)
    ragged = getattr(eng, "ragged", False)
    pol_ad = LoraAdapter(d, init_lora(d, seed=1, device=dev), dev, trainable=True)
    ref_ad = LoraAdapter(d, init_lora(d, seed=2, device=dev), dev, trainable=False)
    if not args.no_merge_ref:      # frozen adapter: s*B@A folded once into a second bf16 copy of the projections (PEFT-style merge)
        ref_ad.merge_into_base(base)
    torch.cuda.empty_cache()
    policy = AutoregressivePolicy(eng, pol_ad, t_len, pack_responses=pack)
    ref_policy = AutoregressivePolicy(eng, ref_ad, t_len, pack_responses=pack)
    opt = FlatAdamW(pol_ad.master, pol_ad.grad, pol_ad.work, lr=1e-6, max_grad_norm=1.0, mode=args.optimizer_mode,
                    bucket_bounds=layer_buckets(pol_ad.layer_numel, d.n_layers, 4))

    def make_hook(o):                # the exchange of a bucket of layers starts when the backward has left its lowest layer
        def hook(layer):
            pos = layer * pol_ad.layer_numel
            for bi, b in enumerate(o.buckets):
                if b.lo == pos:
                    o.launch_bucket(bi)
        return hook
    bucket_hook = make_hook(opt)
    largs = DPOArgs()
    # a POOL of different synthetic micro-batches, cycled over the steps: with ragged rows the cost of a step depends on the valid
    # lengths of its batch (SURVEY.md §8d draws them per pair), so one fixed batch would be a biased sample
    n_pool = max(1, args.batch_pool) * args.accum
    pool = [synth_pairs(d, args.pairs, q_len, t_len, seed=1000 * rank + i, device=dev) for i in range(n_pool)]
    main_pool, main_opt = pool, opt
    step_no = [0]

    def step(pool=None, opt=None, hook=None):
        pool = pool if pool is not None else main_pool
        opt = opt if opt is not None else main_opt
        hook = hook if hook is not None else bucket_hook
        k0 = (step_no[0] * args.accum) % len(pool)
        step_no[0] += 1
        batches = pool[k0:k0 + args.accum]
        for mi, b in enumerate(batches):
            feats = eng.encode_images(b["images"])
            # row_lead / row_lens: the batch's ragged-row plan as the (synthetic) collator produced it on the host - no pass reads
            # anything back from the device
            kw = dict(queries=b["queries"], queries_attn_masks=b["queries_attn_masks"], image_feats=feats,
                      chosen_response=b["chosen"], rejected_response=b["rejected"], row_lead=b["row_lead"], row_lens=b["row_lens"])
            # (round 6: the reference pass on a second stream beside the policy forward - the two are independent until the loss - measured 1.5 % SLOWER:
            # the passes' one-tile-per-workgroup launches interleave on the CUs and halve each other's L2 share; profiles/r06b_ab_overlap_ref.txt)
            with torch.no_grad():
                r = ref_policy(**kw)
            o = policy(**kw)
            loss, _, _ = pair_loss(largs, o["chosen_response_logprobs"], o["rejected_response_logprobs"],
                                   r["chosen_response_logprobs"], r["rejected_response_logprobs"])
            policy.layer_done_hook = hook if mi == len(batches) - 1 else None
            loss.backward()
            policy.layer_done_hook = None
        opt.step(grad_accum_div=args.accum)
        opt.zero_grad()
        pol_ad.refresh_transposed()
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    trace = os.environ.get("OPADPO_BENCH_TRACE") == "1"
    for _ in range(args.warmup):
        step()
    sync()
    L.PROFILE = [] if rank == 0 else None               # vision / adapter-refresh GEMMs launched from Python (op-level wrapper)
    if rank == 0 and hasattr(eng, "profile"):
        eng.profile(True)                               # the LLM passes: launched inside opadpo_seq_logprobs_fwd / _bwd
    sampler = GpuSampler(device=local) if (args.sustained and rank == 0) else None
    step_ev = []
    if sampler:
        sampler.__enter__()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        loss = step()
        if args.sustained:      # per-step time stamps on the stream (read after the run; nothing synchronises inside the timed region)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            step_ev.append(ev)
        if trace:      # diagnostics only (synchronises every step)
            torch.cuda.synchronize()
            print(f"[trace] step {step_no[0]}: {(time.perf_counter() - ts) * 1e3:.1f} ms", file=sys.stderr, flush=True)
    sync()
    dt = time.perf_counter() - t0
    if sampler:
        sampler.__exit__()
    prof, L.PROFILE = L.PROFILE, None
    ctx_prof = None
    if rank == 0 and hasattr(eng, "profile"):
        ctx_prof = eng.profile_read()
        eng.profile(False)
    tmax = torch.tensor([dt], device=dev)
    dist_rec = None
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # what the collective backend actually saw: every rank's (rank, local device, device name) through RCCL itself
        seen = [None] * world
        dist.all_gather_object(seen, rank_info(rank, local))
        peak = torch.tensor([torch.cuda.max_memory_allocated() / 1e9], device=dev)
        dist.all_reduce(peak, op=dist.ReduceOp.MAX)
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)                       # sum of ones over RCCL = number of ranks in the communicator
        tmin = torch.tensor([dt], device=dev)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist_rec = {"backend": dist.get_backend(), "collective_library": collective_lib_version(), "world_size": dist.get_world_size(),
                    "allreduce_of_ones": float(probe), "ranks": seen, "hbm_peak_allocated_GB_max_over_ranks": float(peak),
                    "ms_per_step_min_over_ranks": float(tmin) / args.steps * 1e3, "ms_per_step_max_over_ranks": float(tmax) / args.steps * 1e3,
                    "exchange": f"{args.optimizer_mode}: per-bucket {'reduce_scatter + all_gather' if args.optimizer_mode == 'zero1' else 'all_reduce'} of the flat LoRA gradient, launched from the backward's layer hook"}
        if not args.no_side_legs:
            # what the exchange costs a step at THIS world size: the same step on the same pool with a collective-free optimizer
            # (every rank updates its own replica, nothing on the wire), barrier-bracketed, max over ranks both ways
            try:
                def timed_all(n_warm, n, **kw):
                    for _ in range(n_warm):
                        step(**kw)
                    sync()
                    t0_ = time.perf_counter()
                    for _ in range(n):
                        step(**kw)
                    sync()
                    t_ = torch.tensor([time.perf_counter() - t0_], device=dev)
                    dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                    return float(t_) / n
                n_ab = max(4, min(8, len(main_pool)))
                step_no[0] = 0
                t_with = timed_all(1, n_ab)
                m_local = pol_ad.master if getattr(pol_ad, "master", None) is not None else pol_ad.work.float()
                opt_l = FlatAdamW(m_local, pol_ad.grad, pol_ad.work, lr=1e-6, max_grad_norm=1.0, mode="allreduce", local_only=True)
                step_no[0] = 0
                t_without = timed_all(1, n_ab, opt=opt_l, hook=lambda layer: None)
                dist_rec["exposed_exchange"] = {"steps": n_ab, "ms_per_step_with_collectives": t_with * 1e3, "ms_per_step_without": t_without * 1e3,
                                                "exposed_ms_per_step": (t_with - t_without) * 1e3, "frac_of_step": (t_with - t_without) / t_without,
                                                "wire_bytes_per_rank_GB": 2 * (world - 1) / world * 2 * pol_ad.numel / 1e9,
                                                "note": "same batches both ways; 'without' = replicated local update, no collective (the replicas diverge: timing only, measured after the timed region)"}
                del opt_l
            except Exception as e:
                dist_rec["exposed_exchange"] = {"error": repr(e)}
    dt = float(tmax)
    pairs_per_step = args.pairs * args.accum * world
    value = pairs_per_step * args.steps / dt
    if rank == 0:
        fl_ref = pair_flops(d, q_len, t_len)                      # reference formulation: 4 full sequences per pair
        merged = not args.no_merge_ref
        fl = pair_flops_packed(d, q_len, t_len, 2, ref_merged=merged) if pack else fl_ref - (2 * 2 * lora_param_count(d) * (q_len + t_len + d.n_patches - 1) if merged else 0)     # what this run executes
        rows_per_pair = (q_len + d.n_patches - 1 + 2 * t_len) if pack else 2 * (q_len + t_len + d.n_patches - 1)
        if ragged and pack:      # padding positions are not rows: FLOPs and rows from the valid lengths of this rank's batches
            from opadpo_amd.dims import pair_flops_ragged
            tot, rows_tot, n = 0.0, 0, 0
            for b in pool:
                lead = (b["queries_attn_masks"].int().cumsum(1) == 0).sum(1).tolist()
                vc = (t_len - ((b["chosen"] != 0).flip(1).int().cumsum(1) == 0).sum(1)).tolist()
                vr = (t_len - ((b["rejected"] != 0).flip(1).int().cumsum(1) == 0).sum(1)).tolist()
                for l_, c_, r_ in zip(lead, vc, vr):
                    pr = q_len + d.n_patches - 1 - l_
                    tot += pair_flops_ragged(d, pr, [c_, r_], ref_merged=merged, compact_top=not (args.ctx_flags >= 0 and args.ctx_flags & 128))
                    rows_tot += pr + c_ + r_
                    n += 1
            fl, rows_per_pair = tot / n, rows_tot / n
        roof = None
        if prof or ctx_prof:
            tot_f = sum(p[0] for p in prof or [])
            tot_ms = sum(p[1].elapsed_time(p[2]) for p in prof or [])
            n_launch = len(prof or [])
            if ctx_prof:
                tot_f, tot_ms, n_launch = tot_f + ctx_prof[0], tot_ms + ctx_prof[1], n_launch + ctx_prof[2]
            ach = tot_f / (tot_ms * 1e-3) / 1e12
            traffic, traffic_src = _pmc_traffic()
            roof = {"bound": "mfma", "kernel": "gemm_nt (256x256x64 tile, 4 waves x 128x128 with 256 AGPR accumulators, K-loop as one generated asm block with the LDS-DMA pieces at a period of 4 MFMAs - for products of >= 128 K-tiles its DEEP text: first operand's region released early, bursts at a period of 3 -, streaming persistent form for the plain products / 128x128 for skinny N; LoRA tail fused by K-concatenation; since round 5 the launches also carry the residual adds of the o / down projections and the SwiGLU backward of the down projection's dgrad in their direct epilogues - `unfused_epilogues` holds the rate without that work)",
                    "achieved": ach, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_MFMA_TFLOPS,
                    "traffic": traffic, "traffic_source": traffic_src, "launches": n_launch, "avg_launch_ms": tot_ms / n_launch,
                    # side note (round 6, profiles/r06o_mfma_power.txt): a BARE v_mfma_f32_16x16x32_bf16 loop with random bf16 operands sustains 1971 TF/s at the socket's
                    # ~1.37-kW limit (2.22 GHz) - the 2.5-PF `peak` above is reachable with zero operands only.  `frac` stays priced against the guide's peak.
                    "bare_mfma_random_operands_TFLOPs": 1971.0, "frac_of_bare_mfma_at_power_limit": ach / 1971.0,
                    "gemm_time_share_of_step": tot_ms * 1e-3 / dt}
        out = {"metric": "preference-pairs/sec LLaVA-1.5-7B LoRA DPO seq512", "value": value, "unit": "pairs/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": f"LLaVA-1.5-{args.model.upper()} LoRA(r={d.lora_r}) DPO, 1xMI355X per rank, query {q_len} + response {t_len} "
                                      f"(L={q_len + t_len + d.n_patches - 1}), random-init weights, synthetic pairs",
                          "response_layout": ("packed: chosen+rejected share one pass over the image+query prefix (segment-masked attention), "
                                              f"{q_len + d.n_patches - 1}+2x{t_len} positions per pair and pass" if pack else
                                              f"stacked: 2 x {q_len + t_len + d.n_patches - 1} positions per pair and pass (reference layout)"),
                          "rows": (f"ragged: padding positions (left pad of the query, right pad of each response) are not rows of any kernel; {rows_per_pair:.0f} rows per pair "
                                   f"and pass on average instead of {q_len + d.n_patches - 1 + 2 * t_len}; the top decoder layer's o-projection / MLP only on the rows the head reads" if (ragged and pack) else "padded: every position is a row, like the reference computes it"),
                          "reference_adapter": ("frozen adapter merged into a second bf16 copy of the LLM projections at load (no LoRA GEMMs in the no-grad pass)"
                                                " - NOT the trainer CLI's default: opadpo_train runs --merge_ref_adapter 0 out of the box, whose step is the"
                                                " `reference_unmerged` record of this line (~4 % slower); --merge_ref_adapter 1 selects this form"
                                                if not args.no_merge_ref else "unmerged (K-concatenated LoRA in the no-grad pass; the trainer CLI's default)"),
                          "headline_is_trainer_cli_default": bool(args.no_merge_ref),
                          "pairs_per_microbatch_per_gpu": args.pairs, "grad_accum": args.accum,
                          "global_pairs_per_step": pairs_per_step, "seq_len": q_len + t_len,
                          "parallelism": f"dp{world}" + ("+zero1" if args.optimizer_mode == "zero1" and world > 1 else ""),
                          "loss": float(loss.detach()) if hasattr(loss, "detach") else float(loss)},
               "executed_flops_per_pair_TF": fl / 1e12, "reference_layout_flops_per_pair_TF": fl_ref / 1e12,
               # hardware utilisation = FLOPs this run EXECUTES per second / peak.  The packed layout executes 0.66x the FLOPs of the
               # reference's stacked layout per pair (reference_layout_flops_per_pair_TF); that saving is credited in pairs/s only.
               "mfma_roofline_frac_end_to_end": value / world * fl / (PEAK_BF16_MFMA_TFLOPS * 1e12),
               "hbm_peak_allocated_GB": torch.cuda.max_memory_allocated() / 1e9,
               "roofline": roof}
        if args.sustained and len(step_ev) > 1:
            ms = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(len(step_ev) - 1)]      # step i+1's duration (event to event)
            blk = 20
            blocks = [sum(ms[i:i + blk]) / len(ms[i:i + blk]) for i in range(0, len(ms), blk)]
            ok = [x for x in (sampler.samples if sampler else []) if "error" not in x]
            def col(k):
                v = [x[k] for x in ok if k in x]
                return {"first": v[0], "last": v[-1], "min": min(v), "max": max(v), "mean": sum(v) / len(v)} if v else None
            out["sustained"] = {"steps": args.steps, "pairs": args.steps * pairs_per_step,
                                "pairs_per_s_first_20": pairs_per_step / world * 1e3 / (sum(ms[:blk]) / len(ms[:blk])),
                                "pairs_per_s_last_20": pairs_per_step / world * 1e3 / (sum(ms[-blk:]) / len(ms[-blk:])),
                                "ms_per_step_by_block_of_20": blocks, "sclk_mhz": col("sclk_mhz"), "power_w": col("power_w"), "temp_c": col("temp_c"),
                                "samples": len(ok), "sample_errors": len(sampler.samples) - len(ok) if sampler else 0,
                                "trace": ok[::max(1, len(ok) // 60)]}
        if dist_rec is not None:
            check_dist_record(dist_rec, world, True)
            out["dist"] = dist_rec
        if args.model == "7b" and (world > 1 or args.no_extra_configs):      # default N = 1 line: measured in this run, below (after the state is released)
            par = _parity_record()
            if par is not None:
                out["parity"] = par
        if world == 1 and not args.no_side_legs and not args.op_level:
            def timed(n_warm, n, **kw):
                for _ in range(n_warm):
                    step(**kw)
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for _ in range(n):
                    step(**kw)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0_) / n
            eng.release()
            torch.cuda.empty_cache()
            # (a) the seq512 shape with NOTHING to drop: every query 128 and every response 384 tokens long (no padding), same code path
            try:
                dense_pool = [synth_pairs(d, args.pairs, q_len, t_len, seed=4242 + i, device=dev, dense=True) for i in range(args.accum)]
                dt_d = timed(1, 3, pool=dense_pool)
                fl_d = pair_flops_packed(d, q_len, t_len, 2, ref_merged=not args.no_merge_ref) if pack else fl_ref
                out["dense"] = {"value": args.pairs * args.accum / dt_d, "unit": "pairs/s", "ms_per_step": dt_d * 1e3, "steps": 3,
                                "rows_per_pair": (q_len + d.n_patches - 1 + 2 * t_len) if pack else 2 * (q_len + t_len + d.n_patches - 1),
                                "executed_flops_per_pair_TF": fl_d / 1e12,
                                "mfma_roofline_frac_end_to_end": args.pairs * args.accum / dt_d * fl_d / (PEAK_BF16_MFMA_TFLOPS * 1e12),
                                "note": "all queries / responses at full length (no padding for the ragged layout to drop): the seq512-dense rate of the same kernels"}
                del dense_pool
            except Exception as e:
                out["dense"] = {"error": repr(e)}
            # (a2) the same step with the frozen reference adapter UNMERGED (the trainer CLI's default since round 5: policy and reference run
            # the same K-concatenated kernels, so their log-ratio at equal adapters is exactly 0 like in the reference, dpo_trainer.py:444-449);
            # the headline keeps the merged copy (no LoRA GEMMs in the no-grad pass) and says so in config.reference_adapter
            if not args.no_merge_ref and ref_ad.merged is not None:
                try:
                    saved_merged, ref_ad.merged = ref_ad.merged, None
                    step_no[0] = 0
                    dt_u = timed(1, 4)
                    out["reference_unmerged"] = {"value": args.pairs * args.accum / dt_u, "unit": "pairs/s", "ms_per_step": dt_u * 1e3, "steps": 4,
                                                 "note": "opadpo_train's default (--merge_ref_adapter 0): log-ratio exactly 0 at equal adapters"}
                except Exception as e:
                    out["reference_unmerged"] = {"error": repr(e)}
                finally:
                    ref_ad.merged = saved_merged
            # (a3) round 5 moved two elementwise passes into GEMM epilogues (the residual adds of the o / down projections, the SwiGLU backward of the
            # down projection's dgrad): the step gets shorter, the gemm_nt launches longer - `roofline.frac` of the headline therefore prices those
            # epilogues as GEMM time.  The same step with both passes back in their own kernels (context flag bits 13 | 14), events on: what the
            # kernel's MFMA rate is without the extra epilogue work, and what the fusion is worth per step
            if hasattr(eng, "profile") and args.ctx_flags < 0:
                try:
                    eng.set_flags(use_tr=1 | 8192 | 16384)
                    step_no[0] = 0
                    step()
                    torch.cuda.synchronize()
                    eng.profile(True)
                    dt_n = timed(0, 4)
                    f_n, ms_n, n_n = eng.profile_read()
                    eng.profile(False)
                    out["unfused_epilogues"] = {"ms_per_step": dt_n * 1e3, "value": args.pairs * args.accum / dt_n, "unit": "pairs/s", "steps": 4,
                                                "gemm_nt_roofline_frac": f_n / (ms_n * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, "gemm_nt_launches": n_n,
                                                "note": "residual adds in rmsnorm_sum_fwd and SwiGLU backward as silu_mul_bwd (the round-4 form, bit-identical results): "
                                                        "the llm-pass GEMMs alone, HIP events on; the headline runs both inside gemm_nt's direct epilogues"}
                except Exception as e:
                    out["unfused_epilogues"] = {"error": repr(e)}
                finally:
                    eng.set_flags(use_tr=-1)
            # (b) does the data-parallel exchange hide behind the backward?  Same step with a 1-rank RCCL group: every bucket's bf16 staging
            # cast + reduce_scatter is launched from the layer hook INSIDE the backward, all-gathers after AdamW (optim.FlatAdamW zero1),
            # A/B against the collective-free step measured back to back on the same pool
            if args.model == "7b" and not dist.is_initialized():
                try:
                    eng.release()
                    torch.cuda.empty_cache()
                    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                    os.environ.setdefault("MASTER_PORT", "29547")
                    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
                    os.environ["OPADPO_FORCE_COLLECTIVES"] = "1"
                    opt2 = FlatAdamW(pol_ad.master, pol_ad.grad, pol_ad.work, lr=1e-6, max_grad_norm=1.0, mode="zero1",
                                     bucket_bounds=layer_buckets(pol_ad.layer_numel, d.n_layers, 4))
                    os.environ.pop("OPADPO_FORCE_COLLECTIVES", None)
                    n_ab = max(4, min(8, len(main_pool)))
                    step_no[0] = 0
                    t_with = timed(2, n_ab, opt=opt2, hook=make_hook(opt2))
                    step_no[0] = 0
                    t_without = timed(2, n_ab)
                    if out.get("roofline"):      # what the in-step HIP-event profiling of the timed region costs: same pool, events off
                        out["roofline"]["event_profiling"] = {"events_in_timed_region_per_step": 2 * out["roofline"]["launches"] // max(1, args.steps),
                                                             "ms_per_step_same_pool_without_events": t_without * 1e3,
                                                             "note": "the timed region (ms_per_step, value) brackets every gemm_nt launch with two HIP events; "
                                                                     "the same pool timed without them afterwards"}
                    out["exchange_overlap"] = {"world": 1, "buckets": len(opt2.buckets), "steps": n_ab,
                                               "ms_per_step_with_inline_collectives": t_with * 1e3, "ms_per_step_without": t_without * 1e3,
                                               "overlapped_step_delta_ms": (t_with - t_without) * 1e3,
                                               "delta_frac_of_step": (t_with - t_without) / t_without,
                                               "note": "1-rank RCCL group on one MI355X: per bucket fp32->bf16 staging + reduce_scatter launched from the "
                                                       "backward's layer hook, sharded AdamW, per-bucket all_gather; same batches both ways"}
                    del opt2
                    dist.destroy_process_group()
                except Exception as e:
                    out["exchange_overlap"] = {"error": repr(e)}
                    os.environ.pop("OPADPO_FORCE_COLLECTIVES", None)
        if args.model == "7b" and world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(dict(hidden=d.hidden, n_layers=d.n_layers, n_heads=d.n_heads, head_dim=d.head_dim,
                                                        ffn=d.ffn, vocab=d.vocab, lora_r=d.lora_r, lora_alpha=d.lora_alpha), q_len, t_len)
                full = _cpu_full_record()
                if full is not None:
                    # `value` is the DIRECTLY MEASURED figure (one whole pair at the full 32-layer depth on the host cores, tools/cpu_baseline_full.py;
                    # minutes of CPU work, hence a committed record of an earlier box); the bounded sample timed in THIS run (one decoder layer,
                    # scaled by the depth) rides beside it as `in_run_extrapolation` - it leaves out the head, the vision tower and the memory
                    # effects of a full-depth graph and reads ~1.5 x higher
                    extrap = out["cpu_baseline"]
                    out["cpu_baseline"] = {"value": full["value"], "unit": full["unit"], "cores": full["cores"], "kind": "port", "extrapolated": False,
                                           "sample": full["sample"], "source": full["source"], "seconds_per_pair": full["seconds_per_pair"],
                                           "in_run_extrapolation": extrap}
                try:
                    out["cpu_baseline"]["config_p_direct"] = cpu_baseline_config_p(dict(
                        hidden=d.hidden, n_layers=d.n_layers, n_heads=d.n_heads, head_dim=d.head_dim, ffn=d.ffn, vocab=d.vocab, lora_r=d.lora_r,
                        lora_alpha=d.lora_alpha, v_hidden=d.v_hidden, v_layers=d.v_layers, v_heads=d.v_heads, v_ffn=d.v_ffn,
                        image_size=d.image_size, patch=d.patch))
                except Exception as e:
                    out["cpu_baseline"]["config_p_direct"] = {"error": repr(e)}
            except Exception as e:  # the baseline is a report, never a reason to lose the measurement
                out["cpu_baseline"] = {"error": repr(e)}
        if args.model == "7b" and world == 1 and not (args.no_rollout and args.no_exchange_probe):
            # side records, measured AFTER the timed region on the same GPU; the training state is released first
            numel, layer_numel = pol_ad.numel, pol_ad.layer_numel
            del opt, main_opt, main_pool, policy, ref_policy, pol_ad, ref_ad, pool, loss
            eng.release()
            torch.cuda.empty_cache()
            if not args.no_exchange_probe:
                try:
                    out["exchange_probe"] = exchange_probe(numel, dev, layer_numel, d.n_layers)
                except Exception as e:
                    out["exchange_probe"] = {"error": repr(e)}
                torch.cuda.empty_cache()
            if not args.no_rollout:
                try:
                    out["rollout"] = rollout_leg(eng, d, dev)
                except Exception as e:
                    out["rollout"] = {"error": repr(e)}
            if not args.no_extra_configs:
                # the other BASELINE.json configurations, in the driver's line: each in a CHILD process on the same GPU after this
                # process has given its memory back (a 13B model next to the 7B state does not fit; the child pays its own import + init)
                del eng, base
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                out["parity"] = _parity_in_run()
                out["thirteen_b"] = extra_config("thirteen_b")
                out["recipe"] = extra_config("recipe")
    else:
        out = None
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if out is not None:
        print(json.dumps(out), flush=True)
    # RCCL writes a version banner to stdout (fd 1) at process exit: send everything after the JSON line to stderr so that the
    # line above stays the last thing on stdout
    sys.stdout.flush()
    os.dup2(2, 1)


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:      # one line with the reason, a non-zero exit: the launcher (torch.distributed.run) then ends the other ranks
        import traceback
        traceback.print_exc()
        print(f"[bench] FAILED on rank {os.environ.get('RANK', '0')} of {os.environ.get('WORLD_SIZE', '1')}: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        os._exit(1)
