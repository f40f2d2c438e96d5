#!/usr/bin/env python3
"""Headline benchmark: STFT-frames/sec (fwd + PIT), Conformer-1ch, batch = 32 x 400-frame chunks (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one synthetic batch that is already resident in HBM:
    raw audio (mix, source1, source2: 3 x [32, 102 656] f32)
      -> STFT magnitudes (csship_stft)                     [feature.py:45-82]
      -> Conformer-base encoder forward                    [conformer.py:83-97]
      -> mask head + mask application + PIT-MSE loss, one kernel, masks never written
                                                           [conformer.py:98-103, mse.py:22-37]
Frames counted per step = B*T = 12 800 mixture frames per GPU (the two target spectrograms are work, not frames).

--mode train times the training step instead: the same forward in TRAIN mode (batch-statistics BatchNorm), PIT-MSE,
the explicit HIP backward, the gradient all-reduce over RCCL (N > 1) and the fused clip + Adam update
(css_amd.trainer.GraphedTrainStep).  Same frame count per step.

Multi-GPU: one process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI).  Either the launcher starts
the ranks (torchrun: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or, when WORLD_SIZE is not set and
--gpus N > 1, this script starts N rank processes itself (plain subprocesses, before anything touches a GPU).  Every
rank runs its own batch (weak scaling); the fwd+PIT path has no collective on the data path, the train
mode has the gradient all-reduce.  Barrier + synchronize on both sides of the timed region, max over ranks; rank 0
prints ONE JSON line with n_gpus = the world size the collective library reports and every rank's ms/step.

Extra objects on the JSON line (task section 4):
  roofline     measured INSIDE the step: one un-captured step is run with the library's launch timeline on (HIP event
               pairs directly around every kernel launch, on the launch stream); dominant kernel = the encoder chain
               (chain_s_kernel<8,1,1,1,0,1>: FFN_out+LN -> FFN_in -> LN+QKV, 15 launches per step, MFMA bound):
               achieved = algorithmic flops (2 x 4*M*256*1024 + 2*M*256*768) / average launch duration; peak = 2.5
               PFLOP/s dense bf16; traffic = HBM bytes per launch from the rocprofv3 PMC passes under profiles/.
               roofline.attention / .outproj_conv (MFMA), .stft / .pit (HBM, 2 052 and 4 108 B/frame) the same way.
  cpu_baseline the oracle (fp32 CPU restatement, kind "port") timed on this box's host cores (rank 0, N = 1 only),
               BASELINE.md section 3: 1 warm-up + 3 timed passes of the full batch on all (<= 16) cores, and a second
               leg on ONE thread over a bounded sample (4 of the 32 chunks); CPU model string reported.
  train_step   (default fwd mode) the TRAINING step of the same workload measured in the same run: train-mode forward
               (dropout, batch-statistics BatchNorm), PIT, explicit backward, gradient all-reduce (N > 1), clip + Adam,
               replayed as captured HIP graph(s); ms/step, frames/s, fraction of the bf16 MFMA peak at 3 x the forward
               FLOPs, and with N > 1 the collective backend, world size and bytes all-reduced per step.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, T, NBIN = 32, 400, 257
N_SAMPLES = 512 + 256 * (T - 1)  # 102 656
PEAK_BF16 = 2.5e15
PEAK_HBM = 8.0e12
LOSS_BUDGET = 1e-3   # PIT loss relative error, bf16 operands vs the fp32 oracle
FLOP_PER_FRAME = 52.57e6  # Conformer-base forward @ T=400
FLOP_PER_FRAME_7CH = 53.36e6  # same table, idim 1799


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=("fwd", "train"), default="fwd",
                    help="fwd: the headline metric (features + forward + PIT); train: + backward, all-reduce, Adam")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="gloo: rehearsal without RCCL")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (implies --backend gloo)")
    ap.add_argument("--buckets", type=int, default=4, help="train mode: all-reduce chunks in flight")
    ap.add_argument("--p2p", action="store_true", help="training legs: gradients through css_amd.parallel.P2PAllReduce (the "
                    "hand-written direct all-reduce over hipIpc peer mappings) instead of RCCL - the A/B for an 8-GPU node")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-step", action="store_true", help="fwd mode: skip the extra training-step measurement")
    ap.add_argument("--fuse-conv", type=int, default=None, help="debug: csship_conformer_weights.fuse_conv override")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying the "
                    "step as one captured HIP graph")
    ap.add_argument("--leg-timeout", type=float, default=300.0, help="N > 1: seconds the training legs (the only part that runs "
                    "collectives) may take before rank 0 prints the line without them and every rank leaves")
    ap.add_argument("--blocks", type=int, default=16, help="debug only: anything but 16 is not the headline config")
    ap.add_argument("--overlap-stft", action="store_true", help="debug A/B (measured SLOWER, 1.46 vs 1.43 ms): the two target "
                    "spectrograms on a second stream beside the encoder, joined before the loss kernel")
    ap.add_argument("--pad-rows", action="store_true", help="debug A/B (measured: no gain): spectrogram rows padded to 260 floats, 16-byte aligned")
    ap.add_argument("--no-q-log2e", action="store_true", help="debug A/B: attention with the explicit exponent scaling")
    ap.add_argument("--no-ln-fold", action="store_true", help="debug A/B: keep the LayerNorm affines out of the weights and "
                    "the q/k/v rows in natural order (the round-2 form of the chain kernel's inputs)")
    return ap.parse_args()


_T0 = time.time()


def progress(msg):
    """a line on stderr per phase: a long silent run is taken to be hung on the GPU box"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] t={time.time() - _T0:6.1f}s {msg}", file=sys.stderr, flush=True)


def usable_cores():
    """threads this process can really run at once: affinity mask, capped by the cgroup CPU quota (v2 cpu.max or v1 cfs)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), None))):
        try:
            with open(path) as fh:
                q, per = parse(fh.read())
            if per is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    per = fh.read().strip()
            if q not in ("max", "-1") and float(per) > 0:
                n = max(1, min(n, int(float(q) / float(per) + 0.5)))
            break
        except (OSError, ValueError):
            continue
    return n


def spawn_ranks(n):
    """Start n rank processes of this script (fresh interpreters: nothing here has touched a GPU) and wait."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r))))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main():
    args = parse_args()
    if args.p2p:
        os.environ["CSS_AMD_P2P"] = "1"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s): reporting the launcher's world size",
              file=sys.stderr)
    backend = "gloo" if args.same_device else args.backend
    dev_index = 0 if args.same_device else local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if not args.same_device and torch.cuda.device_count() < world:
            raise SystemExit(f"[bench] {world} ranks need {world} GPUs, this node shows {torch.cuda.device_count()} "
                             "(--same-device rehearses the rank logic on one GPU over gloo)")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)

    from css_amd import _lib, ops
    from css_amd.models import MODELS
    PAD_ROWS = bool(args.pad_rows)
    from css_amd.utils import synthetic as syn

    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=args.blocks, kernel_size=33,
                dropout_rate=0.1, relative_pos_emb=True)
    model = MODELS["Conformer"](257, 257, 2, 1, conf)
    sd = syn.fill_state_dict(syn.conformer_shapes(num_blocks=args.blocks), 0)
    model.load_state_dict(sd)
    model = model.to(dev)
    if args.fuse_conv is not None:
        model.fuse_conv = args.fuse_conv
    if args.no_ln_fold:
        model.no_ln_fold = True
    if args.no_q_log2e:
        model.no_q_log2e = True

    mix, src = syn.synthetic_sources(B, N_SAMPLES, seed=rank)
    audio = torch.stack([mix, src[0], src[1]], 0).to(dev)  # [3, B, n]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    graphed = False
    if args.mode == "fwd":
        model.eval()

        side_stream = torch.cuda.Stream()

        def step():
            # the metric is forward + PIT loss: inference mode (under autograd the model would run its training
            # forward, which keeps every operator's input for the backward pass).  --overlap-stft: the two TARGET
            # spectrograms feed nothing but the loss kernel at the very end, so their STFT can run on a second stream
            # beside the encoder - measured slower (the HBM-bound kernel disturbs the first chain launch more than
            # the 14 us it takes off the critical path), kept as the experiment it was.
            with torch.no_grad():
                if not args.overlap_stft:
                    mags = ops.stft(audio, ("mag",), pad_rows=PAD_ROWS)["mag"]  # [3,B,T,257]
                    loss, perm, _ = model.forward_loss(mags[0], [mags[1], mags[2]])
                    return loss
                main = torch.cuda.current_stream()
                side_stream.wait_stream(main)
                box = {}
                with torch.cuda.stream(side_stream):
                    box["t"] = ops.stft(audio[1:3], ("mag",), pad_rows=PAD_ROWS)["mag"]   # [2,B,T,257]
                mix_mag = ops.stft(audio[0], ("mag",), pad_rows=PAD_ROWS)["mag"]           # [B,T,257]
                loss, perm, _ = model.forward_loss(mix_mag, lambda: [box["t"][0], box["t"][1]],
                                                   before_head=lambda: main.wait_stream(side_stream))
            return loss
        eager_step = step
        for _ in range(args.warmup):
            loss = step()
        torch.cuda.synchronize()  # (the first collective - the barrier in fence() - comes after the capture below)
        # the step is a fixed sequence of ~55 kernel launches on one stream: capture it once, replay it per step
        # (same kernels, same inputs resident in HBM; only the host-side launch work disappears)
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

                def step():  # noqa: F811
                    graph.replay()
                    return static_loss
                for _ in range(2):
                    step()
                graphed = True
            except Exception as e:  # noqa: BLE001  (capture unsupported: keep the eager launches)
                print(f"[bench] graph capture failed ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
                step = eager_step
    else:
        from css_amd.objectives import OBJECTIVES
        from css_amd.optim import FusedAdam
        from css_amd.trainer import GraphedTrainStep
        model.train()
        objective = OBJECTIVES["MSE"].build_objective({})
        optim = FusedAdam(model.parameters(), lr=1e-4, max_norm=5.0)

        def features():
            with torch.no_grad():
                mags = ops.stft(audio, ("mag",), pad_rows=PAD_ROWS)["mag"]
            return {"mix": mags[0], "source1": mags[1], "source2": mags[2]}
        if args.no_graph:
            from css_amd import parallel

            def step():
                optim.zero_grad()
                loss = objective(model, features(), device=dev)
                loss.backward()
                parallel.allreduce_flat(optim.flat_grad, buckets=args.buckets)
                optim.step(max_norm=5.0)
                return loss.detach()
        else:
            stepper = GraphedTrainStep(model, objective, optim, features(), max_norm=5.0, device=dev,
                                       buckets=args.buckets)
            graphed = True

            def step():
                return stepper(features())[0]
        for _ in range(args.warmup):
            loss = step()

    def timed(fn, n):
        """n calls of fn between barrier + synchronize fences; (max-over-ranks seconds, per-rank ms/step, last result)"""
        fence()
        t0 = time.perf_counter()
        out = None
        for _ in range(n):
            out = fn()
        fence()
        d = time.perf_counter() - t0
        per_rank = [d / n * 1e3]
        if world > 1:
            t = torch.tensor([d], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            gathered = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(gathered, t)
            per_rank = [float(g.item()) / n * 1e3 for g in gathered]
            d = max(float(g.item()) for g in gathered)
        return d, per_rank, out

    progress(f"warm-up done ({'graph replay' if graphed else 'eager'}); timing {args.steps} steps")
    dt, per_rank_ms, loss = timed(step, args.steps)
    loss_val = float(loss)
    progress(f"headline: {dt / args.steps * 1e3:.4f} ms/step")

    # ---- fwd mode: the TRAINING legs of the same workload, same run (SURVEY 8d asks for train-mode throughput;
    # configs[1]/[2] are training configurations).  With N > 1 the training steps are the path that communicates.
    train_rec = train7_rec = train_fwd_rec = None

    def all_ranks_ok(ok):
        """A leg that failed on ONE rank must not leave the others waiting in a collective: every rank learns
        whether all ranks built the leg before any of them enters its timed region."""
        if world == 1:
            return ok
        flag = torch.tensor([1.0 if ok else 0.0], device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item() > 0.5)

    def train_leg(idim, features_fn, flop_per_frame, what, forward_only=False):
        """One training-mode leg on its own model instance: GraphedTrainStep (forward, PIT, backward, all-reduce, clip +
        Adam) or, forward_only, the train-mode forward + PIT alone (dropout, batch-statistics BatchNorm, everything the
        backward needs kept - what iteration_types.py:34-50 runs before loss.backward()), captured and replayed."""
        from css_amd.objectives import OBJECTIVES
        from css_amd.optim import FusedAdam
        from css_amd.trainer import GraphedTrainStep
        err, run, tstep, toptim = None, None, None, None
        try:
            tmodel = MODELS["Conformer"](idim, 257, 2, 1, conf)
            tmodel.load_state_dict(sd if idim == 257 else syn.fill_state_dict(syn.conformer_shapes(idim=idim, num_blocks=args.blocks), 0))
            tmodel = tmodel.to(dev).train()
            tobj = OBJECTIVES["MSE"].build_objective({})
            if forward_only:
                static = features_fn()
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):
                        tobj(tmodel, static, device=dev)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    static_loss = tobj(tmodel, static, device=dev).detach()

                def run():
                    feats = features_fn()
                    for k in static:
                        static[k].copy_(feats[k], non_blocking=True)
                    g.replay()
                    return static_loss
            else:
                toptim = FusedAdam(tmodel.parameters(), lr=1e-4, max_norm=5.0)
                tstep = GraphedTrainStep(tmodel, tobj, toptim, features_fn(), max_norm=5.0, device=dev, buckets=args.buckets)

                def run():
                    return tstep(features_fn())[0]
        except Exception as e:  # noqa: BLE001  (the headline line must not die with an extra leg)
            err = f"{type(e).__name__}: {e}"
        if not all_ranks_ok(err is None):
            return {"error": err or "the leg failed on another rank"}
        for _ in range(2):
            run()
        tsteps = max(5, args.steps // 2)
        tdt, tper_rank, tout = timed(run, tsteps)
        rec = {"metric": what, "ms_per_step": tdt / tsteps * 1e3, "value": world * B * T / (tdt / tsteps), "unit": "frames/s",
               "steps": tsteps, "idim": idim, "loss": float(tout), "per_rank_ms_per_step": tper_rank,
               "whole_step_frac": (1.0 if forward_only else 3.0) * B * T * flop_per_frame * args.blocks / 16 / (tdt / tsteps) / PEAK_BF16}
        if not forward_only:
            rec.update({"collective_backend": (backend if world > 1 else None),
                        "collective": (("p2p-direct (css_amd.parallel.P2PAllReduce)" if args.p2p else "all-reduce of the backend")
                                       if world > 1 else None),
                        "world_size": dist.get_world_size() if world > 1 else 1,
                        "allreduce_bytes_per_step": int(toptim.flat_grad.numel() * 4) if world > 1 else 0,
                        "allreduce_schedule": ([hi - lo for lo, hi in tstep.plan] if world > 1 else None)})
        return rec

    progress("in-step launch timeline")
    # ---- rooflines from INSIDE the step: one un-captured eval step with the library's launch timeline on
    roof = None
    if rank == 0:
        import ctypes as C
        lib = C.CDLL(_lib.LIB_PATH)
        lib.csship_debug_timeline.argtypes = [C.c_int]
        lib.csship_debug_timeline_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        was_training = model.training
        model.eval()
        REPS = 3

        calib = [False]

        def probe():
            with torch.no_grad():
                mags = ops.stft(audio, ("mag",), pad_rows=PAD_ROWS)["mag"]
                if calib[0]:   # empty brackets queued in the MIDDLE of a step (the stream is busy on both sides of them)
                    lib.csship_debug_timeline_calibrate(C.c_void_p(torch.cuda.current_stream().cuda_stream), 4)
                model.forward_loss(mags[0], [mags[1], mags[2]])
        probe()
        torch.cuda.synchronize()
        # two un-timed steps are queued first and the timeline is switched on while the GPU is still working on them:
        # every timed launch then runs back to back behind earlier work, as in the replayed graph (a launch on an idle
        # GPU reads twice as long: cold caches, clock ramp)
        probe()
        probe()
        lib.csship_debug_timeline_calibrate.argtypes = [C.c_void_p, C.c_int]
        lib.csship_debug_timeline(1)
        calib[0] = True
        for _ in range(REPS):
            probe()
        calib[0] = False
        torch.cuda.synchronize()
        ems, en = C.c_double(0), C.c_int(0)
        lib.csship_debug_timeline_read(11, C.byref(ems), C.byref(en))
        bracket_measured = ems.value * 1e-3 / max(en.value, 1)     # seconds per empty bracket
        # what a bracket adds to the launch it times is at most an empty bracket's own duration, and cannot be so large
        # that the kernels would cover less than 99 % of the replayed step: in a rocprofv3 kernel trace of the replayed
        # graph the kernels stand back to back (0.1-0.3 us between an end and the next begin, 54 launches; the drain of
        # one kernel and the ramp of the next are INSIDE their durations, as rocprofv3 counts them), so the per-launch
        # times must add up to the step - with the 94 % of rounds 3-4 they read ~5 % shorter than rocprofv3's
        # (profiles/r05_fwd_replay_kernel_avg.csv)
        raw_sum, n_l = 0.0, 0
        for tag in range(10):
            ms, n = C.c_double(0), C.c_int(0)
            lib.csship_debug_timeline_read(tag, C.byref(ms), C.byref(n))
            raw_sum, n_l = raw_sum + ms.value * 1e-3 / REPS, n_l + n.value // REPS
        bracket = max(0.0, min(bracket_measured, (raw_sum - 0.99 * dt / args.steps) / max(n_l, 1))) if args.mode == "fwd" else bracket_measured
        tl = {}
        for tag, name in enumerate(("chain", "chain_edge", "attention", "outproj_conv", "embed", "head", "stft", "pit_sse",
                                    "pit_finalize", "mvn")):
            ms, n = C.c_double(0), C.c_int(0)
            lib.csship_debug_timeline_read(tag, C.byref(ms), C.byref(n))
            # (seconds per launch with the bracket's own cost taken off, launches per step)
            tl[name] = (max(ms.value * 1e-3 / max(n.value, 1) - bracket, 0.0) if n.value else 0.0, n.value // REPS)
        lib.csship_debug_timeline(0)
        if was_training:
            model.train()
        M = B * T
        # HBM traffic / MFMA-busy come from a committed rocprofv3 PMC pass of THIS command (tools/profile_round.sh), not
        # from this run: the record names the profile and is dropped when the kernel sources changed since it was taken
        traffic, mfma_busy, pmc_src = {}, {}, None
        import hashlib
        hsh = hashlib.sha1()
        for fn in sorted(os.listdir(os.path.join(ROOT, "css_amd", "csrc"))):
            if fn.endswith((".hip", ".hpp", ".inc")) and fn != "fft_tables.inc":   # (the generated table is not a source)
                hsh.update(open(os.path.join(ROOT, "css_amd", "csrc", fn), "rb").read())
        csrc_sha = hsh.hexdigest()[:16]
        for tp in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tp)
            if os.path.exists(tpath):
                rec_all = json.load(open(tpath))
                meta = rec_all.pop("_meta", {})
                pmc_src = {"file": "profiles/" + tp, "csrc_sha_at_profile": meta.get("csrc_sha"), "csrc_sha_now": csrc_sha,
                           "current": meta.get("csrc_sha") == csrc_sha}
                if pmc_src["current"]:
                    for name, rec in rec_all.items():
                        if "hbm_bytes_per_launch" in rec:
                            traffic.setdefault(name, rec["hbm_bytes_per_launch"])
                        if rec.get("mfma_busy_frac") is not None:
                            mfma_busy.setdefault(name, rec["mfma_busy_frac"])
                break

        def busy_of(prefix):  # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch duration x 2.4 GHz), committed PMC pass
            return next((v for name, v in mfma_busy.items() if name.startswith(prefix)), None)

        def traffic_of(prefix):
            for name, v in traffic.items():
                if name.startswith(prefix):
                    return v
            return None
        dur, nl = tl["chain"]
        # two FeedForwards (4*M*256*1024 each) + the q/k/v projection (2*M*256*768): SURVEY.md 8(d) per-layer figures
        flops = 2 * 4.0 * M * 256 * 1024 + 2.0 * M * 256 * 768
        step_flops = B * T * FLOP_PER_FRAME * args.blocks / 16
        roof = {"bound": "mfma",
                "kernel": f"chain_s_kernel<8,1,1,1,0,1,PIPE> (FFN_out+LN -> FFN_in -> LN+QKV, static pipelined form, M=12800; {nl} launches/step)",
                "achieved": flops / dur / 1e12 if dur else None, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                "frac": flops / dur / PEAK_BF16 if dur else None, "traffic": traffic_of("chain_s_kernel<8, true, true, true, false, true"),
                "mfma_busy_pmc": busy_of("chain_s_kernel<8, true, true, true, false, true"),
                "avg_launch_us": dur * 1e6, "flops_per_launch": flops,
                "timing": "HIP event pair around every in-step launch, minus the cost of an empty pair (at most the measured one, "
                          "and such that the launches of a step add up to 99 % of the replayed step: rocprofv3 shows them back to back)",
                "bracket_overhead_us": bracket * 1e6, "empty_bracket_us": bracket_measured * 1e6,
                "pmc_source": pmc_src,
                "whole_step_frac": (step_flops * (3.0 if args.mode == "train" else 1.0)) / (dt / args.steps) / PEAK_BF16}
        ad, an = tl["attention"]
        aflops = B * 4 * (2.0 * T * T * 64 * 3)  # QK^T + Q.PE^T + P.V over the full T x T, SURVEY 8(d)
        roof["attention"] = {"bound": "mfma", "kernel": f"mha_fwd_kernel<8> ({an} launches/step)",
                             "achieved": aflops / ad / 1e12 if ad else None, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                             "frac": aflops / ad / PEAK_BF16 if ad else None, "avg_launch_us": ad * 1e6,
                             "traffic": traffic_of("mha_fwd_kernel"), "mfma_busy_pmc": busy_of("mha_fwd_kernel")}
        od, on = tl["outproj_conv"]
        oflops = 2.0 * M * 256 * 256 + 2.0 * M * 256 * 33  # output projection + depthwise taps (halo excluded)
        roof["outproj_conv"] = {"bound": "mixed", "kernel": f"outproj_conv_kernel ({on} launches/step)",
                                "achieved": oflops / od / 1e12 if od else None, "unit": "TFLOP/s", "avg_launch_us": od * 1e6,
                                "hbm_algorithmic_GBps": (M * 256 * (2 + 4 + 4)) / od / 1e9 if od else None,
                                "traffic": traffic_of("outproj_conv_kernel")}
        sdur, _ = tl["stft"]
        nfr = 3 * B * T
        roof["stft"] = {"bound": "hbm", "achieved": nfr * 2052 / sdur / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                        "frac": nfr * 2052 / sdur / PEAK_HBM, "avg_launch_us": sdur * 1e6, "bytes_per_frame": 2052,
                        "traffic": traffic_of("stft_kernel")}
        # mask head + mask application + PIT-MSE.  Fused (csship_head_pit, the default loss-only path): the masks are never
        # written; algorithmic bytes per frame = the encoder row (256 f32) + f_orig and the two targets (3 x 257 f32) =
        # 4 108 B (+ 0.26 MB of head weights per launch), timed as the fused kernel + the finalize launch.  Unfused
        # (pit_sse + finalize on materialised masks): SURVEY 8(d)'s 6 168 B/frame minus the unused noise mask.
        fused_pit = tl["pit_sse"][1] == 0
        if fused_pit:
            pdur = tl["head"][0] + tl["pit_finalize"][0]
            pit_bytes = 256 * 4 + 3 * 257 * 4
            roof["pit"] = {"bound": "hbm", "kernel": "head_pit_split_kernel (mask head + mask application + pairwise SSE) + finalize",
                           "achieved": B * T * pit_bytes / pdur / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                           "frac": B * T * pit_bytes / pdur / PEAK_HBM, "avg_launch_us": pdur * 1e6,
                           "head_pit_us": tl["head"][0] * 1e6, "finalize_us": tl["pit_finalize"][0] * 1e6,
                           "bytes_per_frame": pit_bytes, "traffic": traffic_of("head_pit_split_kernel") or traffic_of("head_pit_kernel")}
        else:
            pdur = tl["pit_sse"][0] + tl["pit_finalize"][0]
            pit_bytes = 5 * 257 * 4
            roof["pit"] = {"bound": "hbm", "achieved": B * T * pit_bytes / pdur / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                           "frac": B * T * pit_bytes / pdur / PEAK_HBM, "avg_launch_us": pdur * 1e6,
                           "sse_us": tl["pit_sse"][0] * 1e6, "finalize_us": tl["pit_finalize"][0] * 1e6,
                           "bytes_per_frame": pit_bytes, "traffic": traffic_of("pit_sse_kernel")}
        roof["step_kernel_us"] = {k: round(v[0] * v[1] * 1e6, 1) for k, v in tl.items()}
        roof["step_kernel_us_sum"] = round(sum(roof["step_kernel_us"].values()), 1)

    def emit(tf_, tr_, t7_, cpu_, parity_):
        """rank 0: the ONE JSON line"""
        frames = world * B * T * args.steps
        mode_txt = ("eval forward (running-stat BN, no dropout)" if args.mode == "fwd" else
                    "training step: train-mode forward (batch-stat BN) + PIT + backward + gradient all-reduce + clip/Adam")
        metric = "STFT-frames/sec (fwd+PIT) Conformer-1ch" if args.mode == "fwd" else \
            "STFT-frames/sec (training step: fwd+PIT+bwd+all-reduce+Adam) Conformer-1ch"
        line = {
            "metric": metric, "value": frames / dt, "unit": "frames/s",
            "n_gpus": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "Conformer-base 1ch 2-spk fwd+PIT, batch=32x400-frame chunks (configs[1])",
                       "global_batch": world * B, "frames_per_chunk": T, "num_blocks": args.blocks,
                       "mode": mode_txt, "parallelism": f"dp{world}",
                       "launch": "hip-graph replay" if graphed else "eager",
                       "streams": ("1" if (args.mode != "fwd" or not args.overlap_stft) else
                                   "2: target-spectrogram STFT beside the encoder, joined before the loss kernel"),
                       "collective_backend": (backend if world > 1 else None),
                       "ranks_launched_by": ("launcher" if "TORCHELASTIC_RUN_ID" in os.environ or "GROUP_RANK" in os.environ
                                             else ("bench.py --gpus" if world > 1 else None))},
            "per_rank_ms_per_step": per_rank_ms,
            "roofline": roof, "cpu_baseline": cpu_, "train_fwd_pit": tf_, "train_step": tr_,
            "train_step_7ch": t7_, "loss": loss_val, "loss_parity": parity_,
        }
        print(json.dumps(line))
        sys.stdout.flush()

    # The training legs run the only collectives of this program (RCCL all-reduce inside the captured step).  With N > 1 they
    # sit behind a watchdog: should a collective never return, rank 0 still prints the headline line - measured above,
    # before any collective but the barriers - with the legs marked as timed out, and every rank leaves.
    legs_done = [False]
    watchdog = None
    if world > 1 and args.mode == "fwd" and not args.no_train_step:
        import threading

        def bail():
            if legs_done[0]:
                return
            print(f"[bench] rank {rank}: the training legs did not finish within {args.leg_timeout} s; leaving without them", file=sys.stderr)
            if rank == 0:
                stuck = {"error": f"did not finish within {args.leg_timeout} s (watchdog); the headline above is unaffected"}
                emit(train_fwd_rec or stuck, train_rec or stuck, train7_rec or stuck, None, None)
            sys.stderr.flush()
            os._exit(0)
        watchdog = threading.Timer(args.leg_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
    if args.mode == "fwd" and not args.no_train_step:
        def tfeatures():
            with torch.no_grad():
                mags = ops.stft(audio, ("mag",), pad_rows=PAD_ROWS)["mag"]
            return {"mix": mags[0], "source1": mags[1], "source2": mags[2]}
        progress("leg: train-mode forward + PIT")
        train_fwd_rec = train_leg(257, tfeatures, FLOP_PER_FRAME, "STFT-frames/sec (TRAIN-mode fwd+PIT: dropout 0.1, batch-statistics "
                                  "BatchNorm, activations kept for the backward; no backward) Conformer-1ch", forward_only=True)
        progress(f"leg: training step ({train_fwd_rec.get('ms_per_step', train_fwd_rec)} ms train fwd+PIT)")
        train_rec = train_leg(257, tfeatures, FLOP_PER_FRAME, "STFT-frames/sec (training step: fwd+PIT+bwd+all-reduce+clip/Adam) Conformer-1ch")
        if train_rec and "error" not in train_rec:
            train_rec["mode"] = ("train-mode forward (dropout 0.1, batch-statistics BatchNorm) + PIT + explicit backward + "
                                 "gradient all-reduce (N > 1) + fused clip/Adam; captured HIP graph replay")
        # configs[2]: the 7-channel training step (IPD features [32, 400, 1799] from raw 7-channel audio, idim 1799)
        from css_amd.executor.feature import FeatureExtractor
        x7 = syn.synthetic_multichannel(src, n_ch=7, seed=rank).to(dev)          # [B, 7, n]
        fe7 = FeatureExtractor(512, 256, True, "1,0;2,0;3,0;4,0;5,0;6,0")

        def tfeatures7():
            with torch.no_grad():
                _, f7, _, _ = fe7.forward_tf(x7, want_spectra=False)
                tg = ops.stft(audio[1:3], ("mag",), pad_rows=PAD_ROWS)["mag"]
            return {"mix": f7, "source1": tg[0], "source2": tg[1]}
        progress(f"leg: 7-channel training step ({train_rec.get('ms_per_step', train_rec)} ms 1ch)")
        train7_rec = train_leg(257 * 7, tfeatures7, FLOP_PER_FRAME_7CH, "STFT-frames/sec (training step, 7-channel IPD features "
                               "idim 1799: features+fwd+PIT+bwd+all-reduce+clip/Adam) Conformer-7ch (configs[2])")
        del x7

    legs_done[0] = True
    if watchdog is not None:
        watchdog.cancel()

    # ---- CPU baseline: the oracle on one full batch, host cores of this box
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        cores = max(1, min(usable_cores(), int(os.environ.get("CSS_CPU_THREADS", "16"))))  # the GPU box shares its host cores
        cpu_model = "unknown"
        try:
            with open("/proc/cpuinfo") as fh:
                cpu_model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "unknown")
        except OSError:
            pass

        def cpu_pass(nb):
            """features -> Conformer forward -> PIT-MSE of the first nb chunks through the oracle; seconds"""
            with torch.no_grad():
                c0 = time.perf_counter()
                mags = [oracle.feature.stft(x[:nb])[0].transpose(1, 2).contiguous() for x in (mix, src[0], src[1])]
                y_pred, _ = oracle.conformer.forward(mags[0], sd, num_blocks=args.blocks)
                closs = oracle.pit.mse_objective(y_pred, mags[1], mags[2])
                return time.perf_counter() - c0, float(closs)
        # >= 1 warm-up + >= 3 timed iterations, all cores and one thread
        torch.set_num_threads(cores)
        progress(f"cpu baseline: oracle on {cores} threads")
        cpu_pass(B)
        runs = [cpu_pass(B) for _ in range(3)]
        progress(f"cpu baseline: {runs[0][0]:.2f} s per pass; one thread, 4 chunks")
        cdt = sorted(r[0] for r in runs)[1]   # median of 3
        nb1 = 4                                # one thread: a bounded sample (4 of the 32 chunks)
        torch.set_num_threads(1)
        cpu_pass(nb1)
        runs1 = sorted(cpu_pass(nb1)[0] for _ in range(3))
        # BASELINE.md section 3 says os.cpu_count().  What this process can actually run on is the smaller of its affinity
        # mask and its cgroup CPU quota (the GPU box gives a container a 16-core share of a 256-thread host: with 256
        # threads under that quota a pass takes 225 s instead of 2.5 s - measured, round 4); the leg above already uses
        # min(usable, 16).  An extra leg runs only when more than that is usable, behind a bounded probe (4 chunks).
        allc = None
        usable = usable_cores()
        if usable > cores and not os.environ.get("CSS_CPU_NO_ALLCORES"):
            progress(f"cpu baseline: all {usable} usable host threads (probe on 4 chunks first)")
            torch.set_num_threads(usable)
            cpu_pass(4)
            probe = cpu_pass(4)[0]
            if 4 * T / probe > 0.8 * B * T / cdt:      # not oversubscribed: time the full batch
                cpu_pass(B)
                runs_all = [cpu_pass(B)[0] for _ in range(2)]
                allc = {"value": B * T / min(runs_all), "unit": "frames/s", "cores": usable,
                        "sample": f"forward+PIT of the full batch on {usable} threads (affinity mask and cgroup quota): 1 warm-up + 2 "
                                  "timed passes, the faster one", "seconds_per_pass": [round(r, 3) for r in runs_all]}
            else:
                allc = {"value": 4 * T / probe, "unit": "frames/s", "cores": usable,
                        "sample": "4 of the 32 chunks, one pass: slower than the 16-thread leg, full batch not timed"}
        torch.set_num_threads(cores)
        cpu = {"value": B * T / cdt, "unit": "frames/s", "cores": cores, "kind": "port", "all_cores": allc,
               "sample": f"forward+PIT of the full batch (B={B} x T={T} frames, fp32, torch {torch.__version__} CPU ops): "
                         f"1 warm-up + 3 timed passes, median",
               "seconds_per_pass": [round(r[0], 3) for r in runs], "cpu_model": cpu_model, "host_cpu_count": os.cpu_count(), "usable_cores": usable,
               "single_thread": {"value": nb1 * T / runs1[1], "unit": "frames/s", "cores": 1,
                                 "sample": f"{nb1} of the {B} chunks x T={T} frames: 1 warm-up + 3 timed passes, median",
                                 "seconds_per_pass": [round(r, 3) for r in runs1]},
               "loss": runs[-1][1], "gpu_loss": loss_val if args.mode == "fwd" else None}

    parity = None
    if cpu is not None and cpu.get("gpu_loss") is not None:
        rel = abs(cpu["gpu_loss"] - cpu["loss"]) / abs(cpu["loss"])
        parity = {"gpu_loss": cpu["gpu_loss"], "cpu_loss": cpu["loss"], "rel_err": rel, "budget": LOSS_BUDGET,
                  "margin": 1.0 - rel / LOSS_BUDGET, "ok": bool(rel < LOSS_BUDGET),
                  "what": "PIT-MSE loss of the timed batch: HIP path (bf16 operands) vs the fp32 oracle, SURVEY 8d budget"}
    if rank == 0:
        emit(train_fwd_rec, train_rec, train7_rec, cpu, parity)
        if parity is not None and not parity["ok"]:
            print(f"[bench] PARITY FAILURE: loss {parity['gpu_loss']:.6f} vs oracle {parity['cpu_loss']:.6f}: relative error "
                  f"{parity['rel_err']:.2e} >= {LOSS_BUDGET:.0e}", file=sys.stderr)
            sys.exit(3)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
