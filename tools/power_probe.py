"""Diagnostic: socket power and shader clock while one GEMM shape runs in a loop (ours vs the vendor GEMM as a yardstick).
Usage: python tools/power_probe.py   (GB_M / GB_SECS optional).  Not part of the product path."""
import json, os, subprocess, sys, threading, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "opa-dpo_amd"))
from opadpo_amd import lib as L


def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10)
            j = json.loads(r.stdout)
            c = j.get("card0", {})
            pw = [v for k, v in c.items() if "ower" in k and "W" in k]
            sclk = [v for k, v in c.items() if k.startswith("sclk")]
            out.append((time.time(), pw, sclk))
        except Exception as e:      # noqa
            out.append((time.time(), repr(e), None))
        time.sleep(0.2)


def run(name, fn, secs):
    samples, stop = [], threading.Event()
    th = threading.Thread(target=poll, args=(stop, samples))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); n += 20
    dt = time.time() - t0
    stop.set(); th.join()
    print(name, "ms/iter %.3f" % (dt / n * 1e3), "samples:", [(s[1], s[2]) for s in samples[2:-1]][:8], flush=True)
    return dt / n


def main():
    L.load()
    dev = torch.device("cuda:0")
    M = int(os.environ.get("GB_M", 32362)); N, K = 22016, 4352
    secs = float(os.environ.get("GB_SECS", 4))
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    for v in (31, 17):
        L.set_flags(v, True)
        t = run("ours variant %d" % v, lambda: L.gemm_nt(a, b, out), secs)
        print("   -> %.0f TF/s" % (fl / t / 1e12))
    t = run("vendor GEMM (yardstick)", lambda: torch.matmul(a, b.t(), out=out), secs)
    print("   -> %.0f TF/s" % (fl / t / 1e12))
    z = torch.zeros_like(a)
    L.set_flags(31, True)
    t = run("ours variant 23, A = 0", lambda: L.gemm_nt(z, b, out), secs)
    print("   -> %.0f TF/s" % (fl / t / 1e12))
    t = run("vendor GEMM, A = 0", lambda: torch.matmul(z, b.t(), out=out), secs)
    print("   -> %.0f TF/s" % (fl / t / 1e12))


if __name__ == "__main__":
    main()
