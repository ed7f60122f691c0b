#!/usr/bin/env python
"""Throughput harness for the leaf-evaluation path (the counterpart of the reference's `dream_go --bench`
forward line, src/dg/bench/forward.rs:28-47, and of src/dg_tests/benches/batch_sizes.rs:41-68).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (dg_nn::forward's device work, graph.rs:134-152) over one batch of
synthetic positions that is already resident in HBM.  Workload at every N: BASELINE.json configs[2] --
19x19, 20-block / 256-filter network, batch 256 per GPU -- the configuration the metric's roofline target is
quoted on.  Multi-GPU: independent replicas, one process per GPU, units sharded across ranks with no data-path
collective ("weak" scaling); the only collective is one RCCL broadcast of the packed weight image at start-up.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP16_MFMA_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--cpu-sample", type=int, default=256, help="boards the CPU oracle evaluates for cpu_baseline (0 = skip)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the weight broadcast even with one rank (self-test of the N > 1 path)")
    return ap.parse_args()


class _DevicePointer:
    """Lets torch wrap an HBM range owned by libdgnn_hip.so (zero-copy) through __cuda_array_interface__."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def broadcast_weights(net, device, torch, dist, replicas):
    """One RCCL broadcast of the packed weight image from rank 0 over xGMI, verified by checksum."""
    ptr, nbytes = net.weight_image(device)
    try:
        image = torch.as_tensor(_DevicePointer(ptr, nbytes), device=f"cuda:{device}")
        replicas.broadcast_image(dist, image, src=0)
        torch.cuda.synchronize(device)
        mode = "zero-copy"
    except Exception:   # noqa: BLE001 -- fall back to a staging tensor + device-to-device copy
        image = torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{device}")
        if dist.get_rank() == 0:
            image.copy_(torch.from_numpy(net.weight_image_host().copy()))
        replicas.broadcast_image(dist, image, src=0)
        torch.cuda.synchronize(device)
        net.set_weight_image(device, image.data_ptr(), nbytes, src_on_device=True)
        mode = "staged"
    if not replicas.all_agree(dist, torch, net.weight_checksum_device(device), device=f"cuda:{device}"):
        raise RuntimeError("weight image differs across ranks after the broadcast")
    return mode, nbytes


def effective_cpus():
    """The cores this process may really use: the scheduler affinity, capped by the cgroup CPU quota (a GPU box of the pool
    shows 256 logical CPUs but grants 16 CPUs of time, /sys/fs/cgroup/cpu.max = "1600000 100000")."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(text, features, batch, sample):
    """The CPU oracle (a port of the reference semantics; the reference has no CPU NN path) on the host cores."""
    from oracle import oracle
    oracle.lib().dgo_set_num_threads(effective_cpus())            # one thread per usable core
    tensors, _ = oracle.load_weights(text)
    net = oracle.OracleNetwork(tensors)
    sample = min(sample, batch)
    f = features.reshape(batch, 11552)[:sample].reshape(-1)
    net.forward(f[:11552], 1)                       # warm (thread pool, page faults)
    t0 = time.perf_counter()
    value, policy = net.forward(f, sample)
    dt = time.perf_counter() - t0
    return {
        "value": sample / dt, "unit": "evals/s", "cores": int(oracle.lib().dgo_num_threads()), "kind": "port",
        "sample": f"{sample} boards of the same batch, one dgo_forward call (oracle/dgnn_oracle.c, OpenMP), {dt:.1f} s",
    }, (value, policy)


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch   # device plumbing + torch.distributed only; must be imported before libdgnn_hip.so is loaded
    import __graft_entry__ as graft
    graft.package()
    from dream_go_amd import api, replicas, synth

    dist = None
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    device = local_rank

    # ---- synthetic network + inputs in the reference's formats (no network access for real weights)
    text = synth.to_json(synth.make_tensors(args.blocks, args.channels, seed=1))
    net = api.Network.from_json(text)
    info = net.info()
    features = synth.make_features(args.batch, seed=replicas.rank_seed(synth.FEATURE_SEED, rank))

    bcast = None
    if dist is not None:
        bcast = broadcast_weights(net, device, torch, dist, replicas)

    ws = net.get_workspace(args.batch, device=device)
    ws.upload(features)                              # inputs resident in HBM before the timed region

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)
        ws.sync()

    for _ in range(args.warmup):
        ws.forward_resident()
    barrier()

    t0 = time.perf_counter()
    # K passes back to back on the workspace's stream; HIP events on that stream bracket the whole run and the
    # residual-tower launches of every pass (include/dgnn.h: dgnn_forward_timed)
    total_ms, tower_ms, tower_launches = ws.forward_timed(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0

    if dist is not None:
        elapsed = replicas.max_over_ranks(dist, torch, elapsed, device=f"cuda:{device}")

    evals = world * args.batch * args.steps
    value = evals / elapsed

    # ---- roofline of the dominant kernel: the C -> C residual convolution (2 per block)
    c = info["num_channels"]
    conv_flops = 2.0 * 9 * c * c * 361 * args.batch            # algorithmic FLOPs per convolution
    convs_per_launch = max(1, round(2 * info["num_blocks"] * args.steps / max(tower_launches, 1)))   # 1, or 2 * blocks (tower kernel)
    launch_ms = tower_ms / max(tower_launches, 1)
    achieved = conv_flops * convs_per_launch / (launch_ms * 1e-3) / 1e12 if launch_ms > 0 else 0.0
    roofline = {
        "bound": "mfma", "achieved": achieved, "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": achieved / PEAK_FP16_MFMA_TFLOPS, "traffic": None,
        "kernel": f"residual 3x3 convolution C->C ({ws.conv_kernel()})", "launch_us": launch_ms * 1e3,
        "launches_timed": tower_launches, "convolutions_per_launch": convs_per_launch,
        "flops_per_launch": conv_flops * convs_per_launch, "us_per_convolution": launch_ms * 1e3 / convs_per_launch,
    }
    # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc passes (tools/profile.sh +
    # tools/pmc_summary.py); only quoted when the profile was taken on this workload and this kernel
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file):
        try:
            t = json.load(open(traffic_file))
            if t.get("batch") == args.batch and t.get("channels") == c and t.get("kernel", "") in roofline["kernel"]:
                roofline["traffic"] = t["hbm_bytes_per_launch"]
        except Exception:   # noqa: BLE001
            pass

    out = None
    if rank == 0:
        flops_eval = synth.flops_per_eval(info["num_blocks"], c, info["num_samples"])
        out = {
            "metric": "nn_leaf_evals_per_sec", "value": value, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {
                "workload": f"19x19, {info['num_blocks']}-block/{c}-filter net, batch {args.batch} per GPU "
                            "(BASELINE.json configs[2]); Bernoulli(0.2) features, random-init weights in the reference JSON schema",
                "blocks": info["num_blocks"], "channels": c, "samples": info["num_samples"],
                "batch_per_gpu": args.batch, "global_batch": world * args.batch,
                "parallelism": f"replicas x{world} (independent leaf batches, no data-path collective)",
            },
            "roofline": roofline,
            "whole_pass": {
                "tflops": flops_eval * args.batch * args.steps / (total_ms * 1e-3) / 1e12,
                "frac_of_mfma_peak": flops_eval * args.batch * args.steps / (total_ms * 1e-3) / 1e12 / PEAK_FP16_MFMA_TFLOPS,
                "gpu_ms_per_step": total_ms / args.steps, "flops_per_eval": flops_eval,
            },
            "derived": {
                "games_per_hour_est": value * 3600.0 / (250 * 800),
                "assumption": "250 moves x 800 leaf evaluations per move, no cache hits, inputs resident; the measured "
                              "figure through board + search + pool is tools/selfplay_go.py (DESIGN.md section 8)",
            },
        }
        # the reference's own harness style (`dream_go --bench`, src/dg/bench/forward.rs:28-47): one thread, one workspace,
        # synchronous forward() calls on host buffers -- PCIe-inclusive, reported beside `value`, never as `value`
        if world == 1:
            reps = max(3, min(args.steps, 10))
            api.forward(ws, features)
            t1 = time.perf_counter()
            for _ in range(reps):
                api.forward(ws, features)
            dt1 = time.perf_counter() - t1
            out["reference_style"] = {
                "boards_per_sec": args.batch * reps / dt1, "calls": reps, "batch": args.batch,
                "what": "synchronous dgnn_forward on host buffers (H2D + pass + D2H per call), single thread",
            }
        if bcast is not None:
            out["weight_broadcast"] = {"mode": bcast[0], "bytes": bcast[1], "backend": "rccl"}
        if world == 1 and args.cpu_sample > 0:
            baseline, (ref_value, ref_policy) = cpu_baseline(text, features, args.batch, args.cpu_sample)
            out["cpu_baseline"] = baseline
            got = ws.download()
            n = min(args.cpu_sample, args.batch)
            dv = float(np.abs(got.value.astype(np.float32)[:n] - ref_value.view(np.float16).astype(np.float32)).max())
            dp = float(np.abs(got.policy.astype(np.float32)[:n * 362] - ref_policy.view(np.float16).astype(np.float32)).max())
            out["parity_vs_oracle"] = {"boards": n, "max_abs_value_diff": dv, "max_abs_policy_diff": dp, "tolerance": 1e-3}
        line = json.dumps(out)
    else:
        line = None

    ws.release()
    net.destroy()
    if dist is not None:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)   # every rank empties its C stdio (RCCL's banner) BEFORE the barrier rank 0 prints after
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        # RCCL prints a version banner through C stdio, which is block-buffered when stdout is a file and would otherwise
        # land AFTER this line at exit: flush the C streams first so that the JSON line is the last thing on stdout
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(line, flush=True)


if __name__ == "__main__":
    main()
