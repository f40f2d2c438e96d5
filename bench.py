#!/usr/bin/env python3
"""Headline benchmark: STFT-frames/sec (fwd + PIT), Conformer-1ch, batch = 32 x 400-frame chunks.

One "step" = one pass of the hot path over one synthetic batch that is already resident in
HBM:  raw audio (mix, source1, source2: 3 x [32, 102 656] f32)
      -> STFT magnitudes (csship_stft)                     [feature.py:45-82]
      -> Conformer-base forward to the ReLU masks          [conformer.py:83-104]
      -> fused mask-apply + PIT-MSE loss                   [conformer.py:103, mse.py:22-37]
Frames counted per step = B*T = 12 800 mixture frames per GPU (the two target spectrograms
are work, not frames).  N > 1: one process per GPU (torchrun), every rank runs its own batch
(weak scaling), no collective on the data path; barrier + synchronize on both
sides of the timed region, max over ranks.

Extra objects on the JSON line (task section 4):
  roofline     dominant kernel = the encoder chain (chain_kernel: FFN_out+LN -> FFN_in -> LN+QKV,
               15 launches per step, MFMA bound); achieved = algorithmic flops (2 x 4*M*256*1024 +
               2*M*256*768) / average launch duration measured with HIP events on the launch
               stream; peak = 2.5 PFLOP/s dense bf16; traffic = HBM bytes per launch from the
               rocprofv3 PMC pass committed under profiles/ (null if absent).
               roofline.stft carries the HBM-bound front end the same way (2 052 B/frame).
  cpu_baseline the oracle (fp32 CPU restatement, kind "port") timed on this box's host cores on
               ONE full batch (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, T, NBIN = 32, 400, 257
N_SAMPLES = 512 + 256 * (T - 1)  # 102 656
PEAK_BF16 = 2.5e15
FLOP_PER_FRAME = 52.57e6  # Conformer-base forward @ T=400


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fuse-conv", type=int, default=None, help="debug: csship_conformer_weights.fuse_conv override")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying the "
                    "step as one captured HIP graph")
    ap.add_argument("--blocks", type=int, default=16, help="debug only: anything but 16 is not the headline config")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    from css_amd import ops
    from css_amd.models import MODELS
    from css_amd.utils import synthetic as syn

    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=args.blocks, kernel_size=33,
                dropout_rate=0.1, relative_pos_emb=True)
    model = MODELS["Conformer"](257, 257, 2, 1, conf)
    sd = syn.fill_state_dict(syn.conformer_shapes(num_blocks=args.blocks), 0)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    if args.fuse_conv is not None:
        model.fuse_conv = args.fuse_conv

    mix, src = syn.synthetic_sources(B, N_SAMPLES, seed=rank)
    audio = torch.stack([mix, src[0], src[1]], 0).to(dev)  # [3, B, n]

    def step():
        # the metric is forward + PIT loss: inference mode (under autograd the model would run its training forward,
        # which keeps every operator's input for the backward pass)
        with torch.no_grad():
            mags = ops.stft(audio, ("mag",))["mag"]  # [3,B,T,257]
            masks = model.forward_masks(mags[0])
            loss, perm, _, _ = ops.pit_mse(masks, mags[0], [mags[1], mags[2]])
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()  # (the first collective - the barrier in fence() - comes after the capture below)
    # the step is a fixed sequence of ~70 kernel launches on one stream: capture it once, replay it per step
    # (same kernels, same inputs resident in HBM; only the host-side launch work disappears)
    graphed = False
    if not args.no_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # thread_local: other threads of this process (e.g. the RCCL watchdog when world > 1) may touch the
            # runtime while this thread captures
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                static_loss = step()
            eager_step = step

            def step():  # noqa: F811
                graph.replay()
                return static_loss
            for _ in range(2):
                step()
            graphed = True
        except Exception as e:  # noqa: BLE001  (capture unsupported: keep the eager launches)
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
            step = eager_step if "eager_step" in dir() else step
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss)

    # ---- roofline of the dominant kernel (the fused FeedForward, 32 launches per step), measured live with
    # HIP events recorded on the stream the kernel is launched on (torch's current stream)
    def launch_time(fn, reps):
        """Seconds per call of fn: `reps` calls recorded into one HIP graph, the replay timed with HIP events on the
        stream it is launched on - the host's per-call work (allocations, ctypes) cannot leak into a 20 us kernel's
        figure.  Falls back to timing the eager loop when capture is unavailable."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for _ in range(reps):
                    fn()
            g.replay()
            torch.cuda.synchronize()
            e0.record(torch.cuda.current_stream())
            g.replay()
            e1.record(torch.cuda.current_stream())
        except Exception as e:  # noqa: BLE001
            print(f"[bench] roofline capture failed ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
            torch.cuda.synchronize()
            e0.record(torch.cuda.current_stream())
            for _ in range(reps):
                fn()
            e1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    roof = None
    if rank == 0:
        M = B * T
        x = torch.randn(M, 256, device=dev)
        g1, b0 = torch.ones(256, device=dev), torch.zeros(256, device=dev)

        def ffn_w():
            return ((g1, b0), ops.pack_frag(torch.randn(1024, 256, device=dev) / 16), torch.zeros(1024, device=dev),
                    ops.pack_frag(torch.randn(256, 1024, device=dev) / 32), torch.zeros(256, device=dev))
        fa, fb = ffn_w(), ffn_w()
        wq, bq = ops.pack_frag(torch.randn(768, 256, device=dev) / 16), torch.zeros(768, device=dev)
        out = torch.empty_like(x)
        run = lambda: ops.encoder_chain(x, 1024, ffn_a=fa, final_ln=(g1, b0), ffn_b=fb, qkv=((g1, b0), wq, bq, 768), out=out)  # noqa: E731
        dur = launch_time(run, 50)
        # two FeedForwards (4*M*256*1024 each) + the q/k/v projection (2*M*256*768): SURVEY.md 8(d) per-layer figures
        flops = 2 * 4.0 * M * 256 * 1024 + 2.0 * M * 256 * 768
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath):
            for name, rec in json.load(open(tpath)).items():
                if name.startswith("chain_kernel<8, false, true, true, true>"):
                    traffic = rec["hbm_bytes_per_launch"]
        roof = {"bound": "mfma", "kernel": "chain_kernel<8,0,1,1,1> (FFN_out+LN -> FFN_in -> LN+QKV, M=12800; 15 launches/step)",
                "achieved": flops / dur / 1e12, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                "frac": flops / dur / PEAK_BF16, "traffic": traffic, "avg_launch_us": dur * 1e6,
                "flops_per_launch": flops,
                "whole_step_frac": (B * T * FLOP_PER_FRAME * args.blocks / 16) / (dt / args.steps) / PEAK_BF16}
        # HBM-bound front end for reference (north_star asks for the STFT roofline too)
        sd_ = launch_time(lambda: ops.stft(audio, ("mag",)), 20)
        nfr = 3 * B * T
        roof["stft"] = {"bound": "hbm", "achieved": nfr * 2052 / sd_ / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": nfr * 2052 / sd_ / 8.0e12, "avg_launch_us": sd_ * 1e6,
                        "bytes_per_frame": 2052}
        # mask application + PIT-MSE (pit_sse + finalize launches), SURVEY 8(d): 6 168 B/frame minus the noise mask
        with torch.no_grad():
            mags_ = ops.stft(audio, ("mag",))["mag"]
            masks_ = model.forward_masks(mags_[0])
        pd_ = launch_time(lambda: ops.pit_mse(masks_, mags_[0], [mags_[1], mags_[2]]), 20)
        pit_bytes = 5 * 257 * 4
        roof["pit"] = {"bound": "hbm", "achieved": B * T * pit_bytes / pd_ / 1e9, "peak": 8000.0, "unit": "GB/s",
                       "frac": B * T * pit_bytes / pd_ / 8.0e12, "avg_launch_us": pd_ * 1e6,
                       "bytes_per_frame": pit_bytes}

    # ---- CPU baseline: the oracle on one full batch, host cores of this box
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        cores = max(1, min(cores, int(os.environ.get("CSS_CPU_THREADS", "16"))))  # the GPU box shares its host cores
        torch.set_num_threads(cores)
        with torch.no_grad():
            c0 = time.perf_counter()
            mags = [oracle.feature.stft(x)[0].transpose(1, 2).contiguous() for x in (mix, src[0], src[1])]
            y_pred, _ = oracle.conformer.forward(mags[0], sd, num_blocks=args.blocks)
            closs = oracle.pit.mse_objective(y_pred, mags[1], mags[2])
            cdt = time.perf_counter() - c0
        cpu = {"value": B * T / cdt, "unit": "frames/s", "cores": cores, "kind": "port",
               "sample": f"1 step of the full workload (B={B} x T={T} frames, fp32, torch {torch.__version__} CPU ops)",
               "loss": float(closs), "gpu_loss": loss_val}

    if rank == 0:
        frames = world * B * T * args.steps
        line = {
            "metric": "STFT-frames/sec (fwd+PIT) Conformer-1ch", "value": frames / dt, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "Conformer-base 1ch 2-spk fwd+PIT, batch=32x400-frame chunks (configs[1])",
                       "global_batch": world * B, "frames_per_chunk": T, "num_blocks": args.blocks,
                       "mode": "eval forward (running-stat BN, no dropout)", "parallelism": f"dp{world}",
                       "launch": "hip-graph replay" if graphed else "eager"},
            "roofline": roof, "cpu_baseline": cpu, "loss": loss_val,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
