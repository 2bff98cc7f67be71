"""Shader clock and board power while one kernel of the hot path runs back to back (a few seconds each): does the GEMM run at the
2.4 GHz the 2.5 PFLOP/s peak is priced at, or does the board hold a power limit by lowering the clock?

A sampling thread reads the amdgpu sysfs files of the card (pp_dpm_sclk: the active level is starred; hwmon power1_average /
power1_input in microwatts; freq1_input in Hz where present) every 20 ms while the main thread keeps the queue full.
Falls back to `rocm-smi --showclocks --showpower` once per phase when sysfs is unreadable.  Output: one line per phase."""
import glob
import os
import re
import subprocess
import sys
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cryovit_amd._lib import EPI_SWIGLU  # noqa: E402
from cryovit_amd.engine import ops  # noqa: E402


def card_dirs():
    """The sysfs directory of torch's cuda:0 (matched by PCI bus id: the box shows every card of the host) first."""
    dirs = [d for d in sorted(glob.glob("/sys/class/drm/card*/device")) if os.path.exists(d + "/pp_dpm_sclk")]
    try:
        pr = torch.cuda.get_device_properties(0)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        mine = [d for d in dirs if want in os.path.realpath(d)]
        print("cuda:0 is PCI", want, "->", mine, flush=True)
        return mine or dirs
    except Exception as e:  # noqa: BLE001
        print("no PCI id from torch:", e, flush=True)
        return dirs


def read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


class Sampler(threading.Thread):
    def __init__(self, dev_dir):
        super().__init__(daemon=True)
        self.d, self.samples, self.on = dev_dir, [], True
        hw = sorted(glob.glob(dev_dir + "/hwmon/hwmon*"))
        self.hw = hw[0] if hw else None

    def run(self):
        while self.on:
            s = read(self.d + "/pp_dpm_sclk") or ""
            m = re.search(r"(\d+)Mhz \*", s)
            sclk = int(m.group(1)) if m else None
            p = f = None
            if self.hw:
                for name in ("power1_average", "power1_input"):
                    v = read(f"{self.hw}/{name}")
                    if v and v.strip().isdigit():
                        p = int(v) / 1e6
                        break
                v = read(f"{self.hw}/freq1_input")
                if v and v.strip().isdigit():
                    f = int(v) / 1e6
            self.samples.append((time.time(), sclk, p, f))
            time.sleep(0.02)


def stats(xs):
    xs = sorted(x for x in xs if x is not None)
    if not xs:
        return "n/a"
    return f"min {xs[0]:.0f} med {xs[len(xs) // 2]:.0f} max {xs[-1]:.0f}"


def phase(name, fn, sampler, seconds=3.0):
    torch.cuda.synchronize()
    n0 = len(sampler.samples) if sampler else 0
    t0 = time.time()
    launches = 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        launches += 20
        torch.cuda.synchronize()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / max(launches, 1)
    line = f"{name:28s} {launches:5d} launches, {ms:7.3f} ms each"
    if sampler:
        ss = sampler.samples[n0 + 10:]  # skip the ramp
        line += f" | sclk MHz {stats([s[1] for s in ss])} | freq1 MHz {stats([s[3] for s in ss])} | power W {stats([s[2] for s in ss])} ({len(ss)} samples)"
    else:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True)
        line += " | " + " ; ".join(x.strip() for x in r.stdout.splitlines() if "sclk" in x or "Power" in x)
    print(line, flush=True)


def main():
    dev = torch.device("cuda:0")
    dirs = card_dirs()
    print("sysfs cards:", dirs, flush=True)
    sampler = None
    if dirs:
        sampler = Sampler(dirs[0])
        sampler.start()
    M, K, N = 128 * 1032, 1536, 8192
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn(ops.alloc_rows(M), K, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * K**-0.5).to(torch.bfloat16)
    out = torch.zeros(ops.alloc_rows(M), N // 2, dtype=torch.bfloat16, device=dev)
    bc = torch.randn(2, N, device=dev, generator=g)
    rowstat = torch.rand(ops.alloc_rows(M), 2, device=dev, generator=g)
    a0 = torch.zeros_like(a)
    w0 = torch.zeros_like(w)
    big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    big2 = torch.empty_like(big)

    def idle():
        time.sleep(0.01)

    phase("idle", idle, sampler, 1.0)
    phase("w12 GEMM (random operands)", lambda: ops.gemm(EPI_SWIGLU, a, w, out, bc, m=M, n=N, ln_rowstat=rowstat), sampler)
    phase("w12 GEMM (all-zero operands)", lambda: ops.gemm(EPI_SWIGLU, a0, w0, out, bc, m=M, n=N, ln_rowstat=rowstat), sampler)
    phase("w12 GEMM (A random, W zero)", lambda: ops.gemm(EPI_SWIGLU, a, w0, out, bc, m=M, n=N, ln_rowstat=rowstat), sampler)
    phase("w12 GEMM (A zero, W random)", lambda: ops.gemm(EPI_SWIGLU, a0, w, out, bc, m=M, n=N, ln_rowstat=rowstat), sampler)
    sgn_a = torch.sign(a.float()).to(torch.bfloat16)   # +-1: random sign bit only, exponent and mantissa constant
    sgn_w = torch.sign(w.float()).to(torch.bfloat16)
    phase("w12 GEMM (+-1 operands)", lambda: ops.gemm(EPI_SWIGLU, sgn_a, sgn_w, out, bc, m=M, n=N, ln_rowstat=rowstat), sampler)
    pow_a = torch.exp2(torch.floor(torch.log2(a.float().abs() + 1e-30))).to(torch.bfloat16) * sgn_a  # random sign + exponent, mantissa 0
    pow_w = torch.exp2(torch.floor(torch.log2(w.float().abs() + 1e-30))).to(torch.bfloat16) * sgn_w
    phase("w12 GEMM (+-2^k operands)", lambda: ops.gemm(EPI_SWIGLU, pow_a, pow_w, out, bc, m=M, n=N, ln_rowstat=rowstat), sampler)
    del sgn_a, sgn_w, pow_a, pow_w
    phase("w12 GEMM (random operands)", lambda: ops.gemm(EPI_SWIGLU, a, w, out, bc, m=M, n=N, ln_rowstat=rowstat), sampler)
    phase("1-GiB device copy", lambda: big2.copy_(big), sampler)
    for mib in (64, 16, 2):  # working sets inside the Infinity Cache (256 MiB) / inside the L2s (8 x 4 MiB): the cost of a byte by where it comes from
        n = mib << 20
        srcs = [big[i * n:(i + 1) * n] for i in range(1)]
        dsts = [big2[i * n:(i + 1) * n] for i in range(1)]
        reps = max(1, (1 << 30) // n // 8)

        def small_copy(srcs=srcs, dsts=dsts, reps=reps):
            for _ in range(reps):
                dsts[0].copy_(srcs[0])

        t0 = time.time()
        phase(f"{mib}-MiB device copy x{reps}", small_copy, sampler)
    qkv = torch.randn(ops.alloc_rows(128 * 1032), 4608, device=dev, generator=g).to(torch.bfloat16)
    o = torch.empty(ops.alloc_rows(128 * 1032), 1536, dtype=torch.bfloat16, device=dev)
    phase("attention (qkv form)", lambda: ops.attention_qkv(qkv, o, slices=128, heads=24, ntok=1029, ntp=1032), sampler)
    phase("idle", idle, sampler, 1.0)
    if sampler:
        sampler.on = False


if __name__ == "__main__":
    main()
