"""BASELINE config 4: EfficientNet-B4 `extract_features` forward + backward on N x 3 x 1024 x 1024 synthetic tiles (N = 4 per
GPU), fp32 vs bf16 (bf16 tensors in HBM, v_mfma_f32_32x32x16_bf16 for the pointwise convolutions, fp32 accumulation /
statistics / parameter gradients).  One JSON line.  Not the graded headline (that is bench.py on config 2): the reference
defines no B4 U-Net, so this is the encoder benchmark SURVEY.md 8(d) C4 describes.

    python tools/bench_b4.py [--batch 4] [--size 1024] [--steps 10] [--warmup 3]

`roofline` is for the bf16 pointwise-conv GEMM kernels (88-92 % of the encoder's FLOPs): algorithmic bytes (operands read
once + result written once) and FLOPs of every launch, HIP events on the launch stream, against the guide's peaks
(HBM 8 TB/s, bf16 MFMA 2.5 PFLOP/s dense).  Their arithmetic intensity at this size is 10-400 FLOP/B against a ridge of ~310,
so the bound that applies is HBM; both fractions are printed."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ssunet_gan_amd as S  # noqa: E402

GMAC_FWD_B4_1024 = 31.325            # B4 extract_features forward at 1024^2, per image
PEAK_HBM_TBPS, PEAK_BF16_TFLOPS = 8.0, 2500.0


def run(dtype, args, dev, profile):
    torch.manual_seed(41)
    enc = S.efficientnet_pytorch.EfficientNet.from_name('efficientnet-b4').to(dev).train().set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(args.batch, 3, args.size, args.size, generator=g).to(dev)

    def step():
        enc.zero_grad(set_to_none=True)
        f = enc.extract_features(x)
        f.sum().backward()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if profile:
        S.bf16.PROFILE = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    prof, S.bf16.PROFILE = S.bf16.PROFILE, None
    del enc
    torch.cuda.empty_cache()
    return dt, prof


def run_graph(dtype, args, dev):
    """The same fwd+bwd captured ONCE into a hipGraph and replayed: at N = 4 the step is ~2 300 kernels of ~16 us, and the
    Python + ctypes launch path (~17 us per launch) -- not the GPU -- sets the eager step time once the kernels are fast
    enough.  Drop-connect's torch.rand is graph-safe (philox offsets advance per replay)."""
    torch.manual_seed(41)
    enc = S.efficientnet_pytorch.EfficientNet.from_name('efficientnet-b4').to(dev).train().set_compute_dtype(dtype)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(args.batch, 3, args.size, args.size, generator=g).to(dev)

    def step():
        f = enc.extract_features(x)
        f.sum().backward()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):                      # warm-up: packed-weight caches, lazily set kernel attributes, grad buffers
            enc.zero_grad(set_to_none=True)
            step()
    torch.cuda.current_stream(dev).wait_stream(side)
    enc.zero_grad(set_to_none=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for _ in range(args.warmup):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    finite = all(torch.isfinite(p.grad).all().item() for p in enc.parameters() if p.grad is not None)
    del graph, enc
    torch.cuda.empty_cache()
    return dt, finite


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--shapes', action='store_true', help='per-shape table of the bf16 GEMM launches on stderr')
    ap.add_argument('--only', choices=['both', 'bf16'], default='both', help="'bf16': skip the fp32 leg (for a clean kernel trace)")
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    flop_img = 3 * 2 * GMAC_FWD_B4_1024 * 1e9 * (args.size / 1024.0) ** 2
    dt32 = run(torch.float32, args, dev, False)[0] if args.only == 'both' else float('nan')
    S.bf16.PROFILE_SHAPES = args.shapes
    dt16e, prof = run(torch.bfloat16, args, dev, True)
    if args.shapes:
        tab = {}
        for label, flops, nbytes, e0, e1 in prof or []:
            a = tab.setdefault(label, [0.0, 0.0, 0.0, 0]); a[0] += flops; a[1] += nbytes; a[2] += e0.elapsed_time(e1) * 1e-3; a[3] += 1
        for k, v in sorted(tab.items(), key=lambda kv: -kv[1][2]):
            print('%-52s x%-3d %8.1f us  %6.2f TB/s %7.1f TF' % (k, v[3] // args.steps, v[2] / v[3] * 1e6, v[1] / v[2] / 1e12, v[0] / v[2] / 1e12), file=sys.stderr)
        prof = [(l.split(' P')[0], f, b, e0, e1) for l, f, b, e0, e1 in prof]
    try:
        dt16, graph_ok = run_graph(torch.bfloat16, args, dev)
        graph_note = 'hipGraph replay of the captured fwd+bwd' if graph_ok else 'graph replay produced non-finite gradients'
        if not graph_ok:
            dt16 = dt16e
    except Exception as e:                       # capture refused: report the eager number
        dt16, graph_note = dt16e, 'hipGraph capture failed (%s): eager launches' % type(e).__name__
    agg = {}
    for label, flops, nbytes, e0, e1 in prof or []:
        a = agg.setdefault(label, [0.0, 0.0, 0.0, 0])
        a[0] += flops; a[1] += nbytes; a[2] += e0.elapsed_time(e1) * 1e-3; a[3] += 1
    kern = {k: {'launches_per_step': v[3] // args.steps, 'time_frac_of_step': round(v[2] / args.steps / dt16e, 4),
                'tbps': round(v[1] / v[2] / 1e12, 3), 'tflops': round(v[0] / v[2] / 1e12, 1)} for k, v in sorted(agg.items())}
    tot = [sum(v[i] for v in agg.values()) for i in range(3)]
    line = {
        'metric': 'EfficientNet-B4 extract_features fwd+bwd images/sec (1024^2 tiles)', 'unit': 'images/sec', 'n_gpus': 1,
        'value': round(args.batch / dt16, 2), 'dtype': 'bf16', 'ms_per_step': round(dt16 * 1e3, 2), 'launch': graph_note,
        'eager_value': round(args.batch / dt16e, 2), 'eager_ms_per_step': round(dt16e * 1e3, 2),
        'fp32_value': round(args.batch / dt32, 2) if dt32 == dt32 else None, 'fp32_ms_per_step': round(dt32 * 1e3, 2) if dt32 == dt32 else None,
        'bf16_speedup': round(dt32 / dt16e, 3) if dt32 == dt32 else None,    # eager vs eager
        'steps': args.steps, 'warmup': args.warmup, 'data': 'synthetic',
        'config': {'workload': 'EfficientNet-B4 encoder (efficientnet_pytorch/model.py:202-218), train mode, default drop_connect_rate 0.2, '
                               '%d x 3x%dx%d per GPU' % (args.batch, args.size, args.size)},
        'algorithmic_tflops_bf16': round(args.batch / dt16 * flop_img / 1e12, 2),
        'roofline': {'bound': 'hbm', 'kernels': 'bf16 pointwise-conv GEMMs (gemm_bf16_kernel fwd/dgrad, gemm_wgrad_bf16_kernel)',
                     'achieved': round(tot[1] / tot[2] / 1e12, 3) if tot[2] else None, 'peak': PEAK_HBM_TBPS, 'unit': 'TB/s',
                     'frac': round(tot[1] / tot[2] / 1e12 / PEAK_HBM_TBPS, 4) if tot[2] else None,
                     'mfma_tflops': round(tot[0] / tot[2] / 1e12, 1) if tot[2] else None,
                     'mfma_frac_of_bf16_peak': round(tot[0] / tot[2] / 1e12 / PEAK_BF16_TFLOPS, 4) if tot[2] else None,
                     'time_frac_of_step': round(tot[2] / args.steps / dt16e, 4) if tot[2] else None, 'per_kernel': kern},
    }
    print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
