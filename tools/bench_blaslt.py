"""Measurement only (never on the product path): what the vendor library (hipBLASLt through torch.matmul) reaches on
the four ViT-g GEMM shapes, as a yardstick for cvx_gemm_bf16.  Prints TFLOP/s per shape."""
import torch

M = 128 * 1032
SHAPES = {"qk": (3072, 1536), "proj": (1536, 1536), "w12": (8192, 1536), "w3": (1536, 4096)}


def main():
    dev = torch.device("cuda:0")
    for name, (N, K) in SHAPES.items():
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * K**-0.5
        for _ in range(3):
            c = a @ w.T
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            c = a @ w.T
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{name:5s} M={M} N={N} K={K}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s", flush=True)
        del a, w, c


if __name__ == "__main__":
    main()
