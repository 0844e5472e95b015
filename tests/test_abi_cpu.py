"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/opadpo_hip.h
declares with the argument lists the ctypes binding assumes (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "opadpo_hip.h")

CTYPE = {"int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "size_t": ctypes.c_size_t,
         "uint64_t": ctypes.c_uint64}


def parse_header():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(int|void|size_t|const char\*)\s+(opadpo_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a or a.split()[0].endswith("_fn"):       # pointers, and the typedef'd allocator function pointers
                    types.append(ctypes.c_void_p)
                else:
                    types.append(CTYPE[a.replace("const ", "").split()[0]])
        protos[name] = (ret, types)
    return protos


@pytest.fixture(scope="module")
def built_lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location("opadpo_build", os.path.join(REPO, "opa-dpo_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(verbose=False)


def test_header_declares_the_path():
    protos = parse_header()
    assert len(protos) >= 33
    for need in ("opadpo_gemm_nt", "opadpo_gemm_tn", "opadpo_attn_fwd", "opadpo_attn_bwd", "opadpo_head_fwd",
                 "opadpo_head_bwd", "opadpo_adamw", "opadpo_sample", "opadpo_attn_decode", "opadpo_rope_kv_append", "opadpo_embed_splice",
                 # sequence-level context API
                 "opadpo_ctx_create", "opadpo_ctx_destroy", "opadpo_ctx_last_error", "opadpo_vision_encode", "opadpo_seq_logprobs_fwd",
                 "opadpo_seq_logprobs_bwd", "opadpo_decode_begin", "opadpo_decode_step", "opadpo_decode_run"):
        assert need in protos


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    for name in parse_header():
        assert hasattr(lib, name), f"{name} declared in include/opadpo_hip.h but not exported"
    lib.opadpo_abi_version.restype = ctypes.c_int
    assert lib.opadpo_abi_version() == 2


def test_binding_matches_header(built_lib):
    from opadpo_amd import lib as L
    protos = parse_header()
    bound = set(L.SIGNATURES) | set(L.OTHER_SYMBOLS)
    assert bound == set(protos), bound ^ set(protos)
    for name, argtypes in L.SIGNATURES.items():
        ret, types = protos[name]
        assert ret == "int"
        assert [t for t in argtypes] == types, f"{name}: binding {argtypes} != header {types}"
    L.load()


def test_host_side_validation_needs_no_gpu(built_lib):
    """Shape validation happens before any launch: a bad call fails loudly with a message."""
    lib = ctypes.CDLL(built_lib)
    lib.opadpo_last_error.restype = ctypes.c_char_p
    fn = lib.opadpo_gemm_nt
    from opadpo_amd import lib as L
    fn.argtypes = L.SIGNATURES["opadpo_gemm_nt"]
    fn.restype = ctypes.c_int
    rc = fn(None, 64, None, 64, 64, None, 0, None, 0, 0, 0, 0, 0, 0, None, 100, 0, None, 0, 0, None, 10, 100, 1.0, 0, None)
    assert rc != 0 and b"multiple of 128" in lib.opadpo_last_error()


def test_context_lifecycle_and_errors_need_no_gpu(built_lib):
    """opadpo_ctx: creation validates the geometry, entry points fail with a message (never abort) before any launch."""
    from opadpo_amd import lib as L
    from opadpo_amd.ctx import Dims
    lib = L.load()
    good = Dims(256, 2, 2, 128, 384, 512, 1e-5, 10000.0, 128, 2, 2, 256, 56, 14, 1e-5, 128, 256.0)
    h = ctypes.c_void_p()
    assert lib.opadpo_ctx_create(ctypes.byref(good), 0, ctypes.byref(h)) == 0 and h.value
    assert lib.opadpo_ctx_set_flags(h, 4, 1) == 0
    # forward before the weights were handed over: refused with text
    rc = lib.opadpo_seq_logprobs_fwd(h, 0, None, None, None, None, None, 1, 8, 2, 1, 1.0, 0, None, None, None, None, None)
    assert rc != 0 and b"weights not set" in lib.opadpo_ctx_last_error(h)
    assert lib.opadpo_ctx_set_adapter(h, 99, None, None, None) != 0 and b"out of range" in lib.opadpo_ctx_last_error(h)
    assert lib.opadpo_decode_step(h, None) != 0 and b"no active rollout" in lib.opadpo_ctx_last_error(h)
    assert lib.opadpo_saved_release(h, ctypes.c_void_p(1234)) != 0
    assert lib.opadpo_ctx_bytes_peak(h) == 0
    lib.opadpo_ctx_destroy(h)
    bad = Dims(250, 2, 2, 128, 384, 512, 1e-5, 10000.0, 128, 2, 2, 256, 56, 14, 1e-5, 128, 256.0)     # hidden != heads * head_dim
    h2 = ctypes.c_void_p()
    assert lib.opadpo_ctx_create(ctypes.byref(bad), 0, ctypes.byref(h2)) != 0 and not h2.value


def test_decode_attention_key_range_rule(built_lib):
    """Host-side geometry of opadpo_attn_decode, read through the workspace size (B * nh * ranges * (hd + 2) floats, 0 for one range):
    up to 256 (sequence, head) pairs as many key ranges as fit ONE round of the 256 CUs; two ranges for 257-767 pairs; from 768 pairs two
    ranges only where they cut the rounds of 1024 resident workgroups by a fifth or more; never more ranges than max_ctx / 256."""
    lib = ctypes.CDLL(built_lib)
    f = lib.opadpo_attn_decode_workspace_bytes
    f.restype = ctypes.c_size_t
    f.argtypes = [ctypes.c_int] * 4
    nh, hd, ctx = 32, 128, 1024

    def ranges(B, nh=nh, max_ctx=ctx):
        b = f(B, nh, hd, max_ctx)
        assert b % (B * nh * (hd + 2) * 4) == 0
        return max(1, b // (B * nh * (hd + 2) * 4))

    assert [ranges(B) for B in (1, 2, 3, 4)] == [4, 4, 2, 2]            # 4 = the max_ctx / 256 cap
    assert [ranges(B) for B in (5, 6, 7, 8)] == [1, 1, 1, 1]            # 160-256 pairs: one round of 16-wave workgroups
    assert [ranges(B) for B in (9, 12, 16, 23)] == [2, 2, 2, 2]
    assert [ranges(B) for B in (24, 28, 32)] == [1, 1, 1]
    assert [ranges(B) for B in (33, 36, 40, 48)] == [2, 2, 2, 2]
    assert [ranges(B) for B in (49, 56, 64)] == [1, 1, 1]
    assert ranges(40, max_ctx=384) == 1 and ranges(12, max_ctx=384) == 1 and ranges(2, max_ctx=600) == 2
    assert ranges(32, nh=40) == 2 and ranges(64, nh=40) == 1            # 13B head count
    assert f(0, nh, hd, ctx) == 0


def test_generated_kloop_is_current():
    """opa-dpo_amd/csrc/w4_kloop.inc (the K-loop of gemm_nt_w4_kernel as an asm block) is generated: the committed file must be what the
    committed generator emits, and the generator's hazard checks (LDS stage reuse, fragment registers, M0, vmcnt bookkeeping) must hold."""
    import subprocess
    import sys
    gen = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opa-dpo_amd", "csrc", "w4_kloop_gen.py")
    r = subprocess.run([sys.executable, gen, "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_kloop_generator_experiment_variants(tmp_path):
    """The generator's experiment variants (tools/build_kloop_exp.sh: W4K_EXP / W4K_OUT) pass its hazard checks too, never write the committed file, and the DEEP
    texts the library ships (W4K_/W4S_TEXT_*_DEEP) are in the committed file; the MFMA walk visits every accumulator exactly once per k-half in both directions."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gen = os.path.join(root, "opa-dpo_amd", "csrc", "w4_kloop_gen.py")
    inc = open(os.path.join(root, "opa-dpo_amd", "csrc", "w4_kloop.inc")).read()
    for name in ("W4K_TEXT_BFIRST", "W4K_TEXT_AFIRST", "W4K_TEXT_BFIRST_DEEP", "W4K_TEXT_AFIRST_DEEP", "W4S_TEXT_BFIRST", "W4S_TEXT_AFIRST", "W4S_TEXT_BFIRST_DEEP",
                 "W4S_TEXT_AFIRST_DEEP"):
        assert "#define " + name + " " in inc, name
    for exp in ("noadv", "early", "p3", "eb8p3l0w92", "eb5p3l3w96", "w96", "straight"):
        out = tmp_path / (exp + ".inc")
        r = subprocess.run([sys.executable, gen], capture_output=True, text=True, env=dict(os.environ, W4K_EXP=exp, W4K_OUT=str(out)))
        assert r.returncode == 0 and out.exists() and "v_mfma_f32_16x16x32_bf16" in out.read_text(), (exp, r.stdout + r.stderr)
    r = subprocess.run([sys.executable, gen], capture_output=True, text=True, env=dict(os.environ, W4K_EXP="noadv"))      # no W4K_OUT: refused
    assert r.returncode != 0
    sys.path.insert(0, os.path.dirname(gen))
    try:
        import w4_kloop_gen as g
        seen = {}
        for k in range(128):
            kk, i, j = g.mfma_ij(k)
            seen.setdefault(kk, []).append((i, j))
        assert sorted(seen[0]) == sorted(seen[1]) == [(i, j) for i in range(8) for j in range(8)]
        assert all(k < 64 for k in range(128) if g.mfma_ij(k)[0] == 0)                      # every k-half-0 MFMA before every k-half-1 one
        assert [g.mfma_ij(k)[2] for k in range(8, 16)] == list(range(7, -1, -1))            # odd rows walk the B fragments backwards
    finally:
        sys.path.pop(0)
