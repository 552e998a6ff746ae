#!/usr/bin/env python
"""Headline benchmark: PIV frame-pairs/s on a synthetic 1080p stack, 32x32 windows @ 50 % overlap.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (lspiv_piv_pairs_dev: window gather + normalise + FFT
cross-correlation + corr_max / s2n + sub-pixel peak, fused) over ONE batch of 1000 frame pairs
per GPU that is already resident in HBM (BASELINE.json configs[1]).  With N > 1 every rank owns
its own 1000-pair time block (weak scaling, BASELINE.json configs[4]); the only exchange is the
RCCL all-gather of the packed (4, t, y, x) result block, software-pipelined one step behind the kernel.

Prints ONE JSON line on rank 0 (see the task contract).  No PyTorch is needed for N = 1; for
N > 1 torch.distributed is plumbing for the rendezvous, the barrier and the all-gather only.
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pyorc_amd import _lib, window  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=1000, help="frame pairs per GPU per step")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--window", type=int, default=32)
    ap.add_argument("--overlap", type=int, default=16)
    ap.add_argument("--cpu-pairs", type=int, default=-1, help="pairs of the CPU-baseline sample (-1: auto, 0: skip)")
    ap.add_argument("--seed", type=int, default=20260927 + 2)
    return ap.parse_args()


def measured_traffic(kernel_substr: str, pairs: int, H: int, W: int, window_size: int, overlap: int):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/*_summary.json,
    written by tools/summarize_profile.py from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this very
    command).  Only returned when the profiled launch had the same shape and kernel (latest summary wins); otherwise
    null."""
    import glob

    best = None
    want = {"pairs": pairs, "H": H, "W": W, "window": window_size, "overlap": overlap}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        launch = d.get("launch", {"pairs": 1000, "H": 1080, "W": 1920, "window": 32, "overlap": 16})
        if launch != want:
            continue
        for name, k in d.get("kernels", {}).items():
            if kernel_substr in name and "hbm_traffic_bytes" in k:
                best = {"bytes": round(k["hbm_traffic_bytes"]), "source": os.path.basename(f)}
    return best


def cpu_baseline(frames_sample: np.ndarray, ws, ov, gpu_block):
    """Time the CPU oracle on a bounded sample of the same stack; also a live parity check."""
    from oracle import cpu_baseline as cb  # test infrastructure: baseline leg only

    return cb.run(frames_sample, ws, ov, gpu_block)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {a.gpus} != WORLD_SIZE {world}")

    if os.environ.get("LSPIV_BENCH_SAME_DEVICE"):  # plumbing test of the N>1 path on a 1-GPU box (not a measurement)
        local_rank = 0
    dist = torch = None
    use_dist = world > 1 or bool(os.environ.get("LSPIV_BENCH_FORCE_DIST"))  # FORCE_DIST: 1-rank plumbing test
    if use_dist:
        # torch FIRST: its wheel bundles its own libamdhip64 (SONAME libamdhip64.so.7).  Loaded first, the dynamic
        # loader hands the same copy to liblspiv_hip.so; loaded second, the process would hold two HIP runtimes and
        # the second one finds no GPU.
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
        _warm = torch.zeros(1, device="cuda")
        dist.all_reduce(_warm)  # build the RCCL communicator now, whatever --warmup says
        torch.cuda.synchronize()
    lib = _lib.load()
    _lib.require_device()
    _lib.check(lib.lspiv_set_device(local_rank))

    H, W, T = a.height, a.width, a.pairs + 1
    ws, ov = (a.window, a.window), (a.overlap, a.overlap)
    n_rows, n_cols = window.get_array_shape((H, W), ws, ov)
    n_win = n_rows * n_cols
    n_tiles = a.pairs * n_win

    # ---- device-resident synthetic stack + result block -------------------------------------
    d_frames, d_out = C.c_void_p(), C.c_void_p()
    if use_dist:
        t_frames = torch.empty(T * H * W, dtype=torch.uint8, device="cuda")
        t_out = torch.empty(4 * n_tiles, dtype=torch.float32, device="cuda")
        t_all = torch.empty(world * 4 * n_tiles, dtype=torch.float32, device="cuda")
        d_frames, d_out = C.c_void_p(t_frames.data_ptr()), C.c_void_p(t_out.data_ptr())
    else:
        _lib.check(lib.lspiv_dev_malloc(C.byref(d_frames), T * H * W))
        _lib.check(lib.lspiv_dev_malloc(C.byref(d_out), 4 * n_tiles * 4))
    _lib.check(lib.lspiv_synth_particles_dev(d_frames, T, H, W, a.seed + rank, 0.02))
    _lib.check(lib.lspiv_synchronize())  # the generator ran on the library's stream; the steps may use another one

    def launch_all():
        _lib.check(lib.lspiv_piv_pairs_dev(d_frames, 0, T, H, W, ws[0], ws[1], ov[0], ov[1], -1.0, d_out, None, None))

    if not use_dist:
        step = launch_all

        def drain():
            pass

        def sync():
            _lib.check(lib.lspiv_synchronize())

        def barrier():
            pass
    else:
        # Software pipeline over steps: step k's kernel (all 1000 pairs, one launch) runs while step k-1's result
        # block is all-gathered over RCCL on a second stream.  Two result buffers; before a buffer is overwritten
        # (two steps later) the compute stream waits for its gather.  Every gather issued inside the timed region
        # completes inside it (drain() before the closing synchronize), so K steps = K kernels + K all-gathers.
        comp = torch.cuda.Stream()   # non-default: its handle goes through the C ABI
        comm = torch.cuda.Stream()
        assert comp.cuda_stream != 0
        outs = [t_out, torch.empty_like(t_out)]
        alls = [t_all, torch.empty_like(t_all)]
        pending = [None, None]
        state = {"k": 0}

        def step():
            b = state["k"] & 1
            with torch.cuda.stream(comp):
                if pending[b] is not None:
                    pending[b].wait()            # stream-level: comp waits until gather k-2 has read outs[b]
                _lib.check(lib.lspiv_piv_pairs_dev(d_frames, 0, T, H, W, ws[0], ws[1], ov[0], ov[1], -1.0,
                                                   C.c_void_p(outs[b].data_ptr()), None, C.c_void_p(comp.cuda_stream)))
                ev = torch.cuda.Event()
                ev.record(comp)
            with torch.cuda.stream(comm):
                comm.wait_event(ev)
                pending[b] = dist.all_gather_into_tensor(alls[b], outs[b], async_op=True)
            state["k"] += 1

        def drain():
            for w in pending:
                if w is not None:
                    w.wait()
            torch.cuda.current_stream().wait_stream(comp)
            torch.cuda.current_stream().wait_stream(comm)

        def sync():
            torch.cuda.synchronize()

        def barrier():
            dist.barrier()

    for _ in range(a.warmup):
        step()
    drain(); sync(); barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain(); sync(); barrier(); sync()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    gathered = alls[(state["k"] - 1) & 1].clone() if use_dist else None

    # ---- live kernel timing with HIP events on the launch stream (roofline leg, N-independent) --
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    _lib.check(lib.lspiv_event_create(C.byref(ev0)))
    _lib.check(lib.lspiv_event_create(C.byref(ev1)))
    reps = max(3, min(a.steps, 10))
    _lib.check(lib.lspiv_synchronize())
    _lib.check(lib.lspiv_event_record(ev0))
    for _ in range(reps):
        launch_all()
    _lib.check(lib.lspiv_event_record(ev1))
    ms = C.c_float()
    _lib.check(lib.lspiv_event_elapsed_ms(ev0, ev1, C.byref(ms)))
    kernel_ms = ms.value / reps

    dist_check = None
    if use_dist:
        # this rank's slice of the all-gathered block must equal its own single-launch result bit for bit;
        # cheap, outside the timed region
        torch.cuda.synchronize()
        whole = t_out.view(-1)                          # launch_all (kernel-timing loop) wrote the same stack here
        mine = gathered.view(world, -1)[rank]
        same = (mine == whole) | (mine.isnan() & whole.isnan())
        dist_check = bool(same.all().item())
    if rank != 0:
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    b_alg_pair = 2 * H * W * 1 + 16 * n_win  # both frames read once + 4 f32 per window
    # the kernel the library dispatches this shape to (pyorc_amd/csrc/piv_fft_impl.h, launch_t): time-walking by default
    # every even window 6..64 has FFT kernels of its own (walking by default); odd ones run embedded / direct kernels
    walking = os.environ.get("LSPIV_WALK", "1") != "0" and a.window % 2 == 0 and 6 <= a.window <= 64 and a.pairs >= 3
    kernel_name = f"piv_fft_{'walk_' if walking else ''}kernel<unsigned char, {a.window}, false, false>"
    achieved = b_alg_pair * a.pairs / (kernel_ms * 1e-3) / 1e9
    pairs_per_s = world * a.pairs * a.steps / dt
    out = {
        # BASELINE.json's metric; a non-default --window / --overlap / --height / --width run says what it measured
        "metric": ("PIV frame-pairs/sec, 1080p 32x32@50% overlap (Mvectors/sec in config)"
                   if (a.window, a.overlap, H, W) == (32, 16, 1080, 1920) else
                   f"PIV frame-pairs/sec, {H}x{W} frames, {a.window}x{a.window} windows @ overlap {a.overlap} (not the BASELINE.json metric)"),
        "value": round(pairs_per_s, 2),
        "unit": "frame-pairs/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"synthetic {H}x{W} uint8 particle stack, {a.pairs} frame-pairs per GPU, "
                        f"{a.window}x{a.window} windows @ overlap {a.overlap} ("
                        + ("BASELINE.json configs[1]" if (a.window, a.overlap, H, W, a.pairs) == (32, 16, 1080, 1920, 1000)
                           else "a variation of BASELINE.json configs[1]")
                        + ("; configs[4] sharding" if world > 1 else "") + ")",
            "frame_dtype": "u8",
            "windows_per_pair": n_win,
            "mvectors_per_s": round(pairs_per_s * n_win / 1e6, 3),
            "parallelism": f"time-block shard x{world}, RCCL all-gather of (4,t,y,x) result" if use_dist else "single GPU",
            **({"allgather_matches_single_launch": dist_check} if use_dist else {}),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": kernel_name,
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": None,
            "algorithmic_bytes_per_launch": b_alg_pair * a.pairs,
            "algorithmic_bytes_per_pair": b_alg_pair,
            "kernel_ms_per_launch": round(kernel_ms, 4),
            "note": "FFT path is FP32-VALU/LDS bound, not HBM bound (DESIGN.md section 4); secondary bound below",
            "secondary": {"bound": "fp32-valu", "flop_per_pair": 0.66e9,
                          "achieved_tflops": round(0.66e9 * a.pairs / (kernel_ms * 1e-3) / 1e12, 2), "peak_tflops": 157.3},
        },
    }
    tr = measured_traffic(kernel_name.split(",")[0] + ",", a.pairs, H, W, a.window, a.overlap)
    if tr:
        out["roofline"]["traffic"] = tr["bytes"]
        out["roofline"]["traffic_source"] = f"profiles/{tr['source']} (rocprofv3 --pmc FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)"
    # ---- CPU baseline on a bounded sample of the same stack (rank 0, N = 1 only) --------------
    if world == 1 and a.cpu_pairs != 0:
        n_s = a.cpu_pairs if a.cpu_pairs > 0 else None
        from oracle import cpu_baseline as cb

        n_s = cb.default_sample_pairs(H, W, ws) if n_s is None else n_s
        n_s = min(n_s, a.pairs)
        sample = np.empty((n_s + 1, H, W), dtype=np.uint8)
        _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(sample), d_frames, sample.nbytes))
        gpu_block = np.empty((4, a.pairs, n_rows, n_cols), dtype=np.float32)
        _lib.check(lib.lspiv_memcpy_d2h(_lib.ptr(gpu_block), d_out, gpu_block.nbytes))
        base = cpu_baseline(sample, ws, ov, gpu_block[:, :n_s])
        out["cpu_baseline"] = base
        if base.get("value"):
            out["config"]["speedup_vs_cpu_baseline"] = round(pairs_per_s / base["value"], 1)
    print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
