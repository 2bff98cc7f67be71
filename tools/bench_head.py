"""BASELINE configs[2] alone: the CryoVIT head forward + masked Dice on synthetic fp16 features [1536,128,32,32] (seed 3), labels
seed 4, head weights seed 5.  Prints ms per forward (HIP events).

    python tools/bench_head.py [reps]
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_head -o head -- python3 tools/bench_head.py 5
    python tools/bench_head.py --summarize gpurun_out/prof_head   # per-launch table in launch order, averaged over the forwards

The per-launch table is what the kernel stats cannot show: the same template runs layers of very different shapes.
"""
import csv
import glob
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def t1_layers():
    """(layer, algorithmic FLOPs, compulsory bytes in + out) of the 28 launches of one head forward at [1,1536,128,32,32] (SURVEY App. B;
    fp16 activations, fp32 probabilities, int8 labels), in launch order."""
    D, h = 128, 32
    nv = D * h * h
    L = [("layers.0 1x1x1 1536->1024 +GELU", 2.0 * nv * 1536 * 1024, nv * (1536 + 1024) * 2)]
    H = h
    for bi, (c1, c2, c3) in enumerate(((1024, 192, 128), (128, 64, 32), (32, 32, 32), (32, 16, 8))):
        n = D * H * H
        L += [(f"sb{bi + 1} GroupNorm stats", 0.0, n * c1 * 2), (f"sb{bi + 1} GroupNorm finalize", 0.0, 0.0), (f"sb{bi + 1} GroupNorm apply", 0.0, n * c1 * 4),
              (f"sb{bi + 1}.conv1 {c1}->{c2} +GELU", 2.0 * n * 27 * c1 * c2, n * (c1 + c2) * 2), (f"sb{bi + 1}.conv2 {c2}->{c2} +GELU", 2.0 * n * 27 * c2 * c2, n * c2 * 4),
              (f"sb{bi + 1}.convT {c2}->{c3} x4 +GELU", 2.0 * n * 4 * c2 * c3, n * (c2 + 4 * c3) * 2)]
        H *= 2
    n = D * H * H
    L += [("output_layer.0 8->8 +GELU", 2.0 * n * 27 * 64, n * 32), ("output_layer.2 8->1 + clip + sigmoid + Dice", 2.0 * n * 27 * 8, n * (16 + 4 + 1)),
          ("dice finalize", 0.0, 0.0)]
    return L


def summarize(d: str) -> None:
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(f)) if "cvx::" in r["Kernel_Name"] or "cvx" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    # one forward = the launches from one gemm256p/EpiBF16<1 (the 1x1x1 projection) to the next k_dice_finalize
    ends = [i for i, n in enumerate(names) if "k_dice_finalize" in n]
    if len(ends) < 2:
        raise SystemExit("need >= 2 forwards in the trace")
    per = ends[-1] - ends[-2]
    fw = [rows[e - per + 1 : e + 1] for e in ends[1:]]
    fw = [x for x in fw if len(x) == per]
    tot = 0.0
    layers = t1_layers() if per == 28 else None
    for k in range(per):
        dur = [int(x[k]["End_Timestamp"]) - int(x[k]["Start_Timestamp"]) for x in fw]
        ms = sum(dur) / len(dur) / 1e6
        tot += ms
        n = fw[0][k]["Kernel_Name"].replace("cvx::", "").split("(")[0]
        if layers:
            name, fl, by = layers[k]
            print(f"{k:3d} {ms:7.3f} ms {fl / ms / 1e9 if fl else 0:7.0f} TFLOP/s {by / ms / 1e9 if by else 0:6.2f} TB/s (compulsory)  {name:44s} {n[:70]}")
        else:
            print(f"{k:3d} {ms:8.3f} ms  {n[:110]}")
    span = [(int(x[-1]["End_Timestamp"]) - int(x[0]["Start_Timestamp"])) / 1e6 for x in fw]
    print(f"sum of kernels {tot:.3f} ms, first-start to last-end {sum(span) / len(span):.3f} ms over {len(fw)} forwards")


def main() -> None:
    if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
        return summarize(sys.argv[2])
    import torch

    import bench
    from cryovit_amd.engine import ops
    from cryovit_amd.engine.head import HeadEngine

    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda:0")
    head = HeadEngine(bench.synthetic_head_state_dict(5, dev), dev)
    D, hp, wp, C = 128, 32, 32, 1536
    g = torch.Generator(device="cpu").manual_seed(3)
    cl = torch.zeros(ops.alloc_rows(D * hp * wp), C, dtype=torch.float16, device=dev)
    for z in range(0, D, 16):
        cl[z * hp * wp : (z + 16) * hp * wp] = torch.randn(16 * hp * wp, C, generator=g).to(torch.float16).to(dev)
    labels = bench.synthetic_labels(dev, 4)
    out = head.forward(cl, D, hp, wp, labels=labels)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = head.forward(cl, D, hp, wp, labels=labels)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    vox = D * 256 * hp * wp
    print(f"head forward + Dice: {ms:.3f} ms = {vox / ms / 1e6:.2f} Gvoxel/s, {head.flops(D, hp, wp) / ms / 1e9:.0f} TFLOP/s; "
          f"dice sums {out['dice_sums'].tolist()}, fg fraction {float((out['probs'] >= 0.5).float().mean()):.4f}", flush=True)


if __name__ == "__main__":
    main()
