#!/usr/bin/env python3
"""Benchmark of the hot path on BASELINE.json's metric: decoded video frames/s (720x486 STC-007).

A "step" = one pass of the binarize path (Binarizer + VideoToDigital bit-cell extraction + per-line CRC)
over one batch of synthetic NTSC frames that is already resident in HBM (workload = BASELINE.json
configs[1]: 10k-frame synthetic STC-007 NTSC batch, 1 x MI355X, binarize + bit-extract + CRC only).

  python bench.py --gpus N --steps K --warmup W
  N > 1: launched by torch.distributed.run, one rank per GPU; ONE tape of N x `--frames` frames per step is sharded
  over the ranks in contiguous frame ranges (weak scaling: per-GPU work fixed).  The only collective is the all-gather
  (RCCL) of every rank's 120-byte final chain state per step, against which each rank checks the state it started its
  range from (sdvpcmdecoder_amd/sharded.py); timing is barrier + synchronize bracketed, MAX over ranks, rank 0 prints
  ONE JSON line.

Extra objects in the JSON line: "roofline" (HBM: algorithmic bytes per launch / HIP-event kernel time,
measured live on the stream the kernel runs on) and "cpu_baseline" (the real reference, or the oracle
port when the reference build is not loadable, timed on this host on a bounded sample)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_frame(width, height):
    """luma plane read once + packed outputs written once:
    W*H luma + (H+3) x 48 B line records + 32 B frame descriptor."""
    return width * height + (height + 3) * 48 + 32


HOT_KERNEL = "sdv_k_stc007_frames_lean"
def _newest_pmc_profile():
    """profiles/rNN_pmc_<hot kernel>.json of the latest round that has one (the PMC passes are made once per round, on the sources as committed)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_%s.json" % HOT_KERNEL)))
    return os.path.relpath(found[-1], ROOT) if found else os.path.join("profiles", "r06_pmc_%s.json" % HOT_KERNEL)


PMC_PROFILE = _newest_pmc_profile()


def measured_hbm_traffic(frames_per_launch, workload):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same
    command, made by tools/pmc_to_json.py): FETCH_SIZE is in KB and, on gfx950, reports half of a wide coalesced stream
    (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE in KB.  Scaled per frame.  The profile names the sources and the
    workload it was measured on: returns (traffic, None) when they are this run's, else (None, what the profile holds) - a
    counter from another build or workload is reported as such, never as this run's traffic."""
    path = os.path.join(ROOT, PMC_PROFILE)
    try:
        from sdvpcmdecoder_amd.build import source_hash
        d = json.load(open(path))
        fetch = float(d["pmc3"]["FETCH_SIZE"]) * 1024.0 * 2.0
        write = float(d["pmc4"]["WRITE_SIZE"]) * 1024.0
        per_launch = (fetch + write) / (float(d["pmc3"].get("grid_size", 640000)) / 64.0) * frames_per_launch
        same = d.get("source_sha16") == source_hash(HOT_KERNEL) and d.get("workload") == workload and not os.environ.get("SDVPCM_LIB")
        if same:
            return per_launch, None
        return None, {"bytes_per_launch": per_launch, "profile": PMC_PROFILE, "profile_source_sha16": d.get("source_sha16"),
                      "profile_workload": d.get("workload"), "this_source_sha16": source_hash(HOT_KERNEL), "this_workload": workload}
    except Exception as ex:      # noqa: BLE001
        return None, {"error": repr(ex), "profile": PMC_PROFILE}


def cpu_baseline(luma_sample, mode):
    """Times the CPU path on this host: the real reference (oracle/_ref, VideoToDigital worker thread)
    when it loads, else the C port in oracle/.  Only used as a reported baseline."""
    import ctypes as C
    import numpy as np
    import libs
    n, h, w = luma_sample.shape
    kind = "port"
    try:
        if libs.ref_available():
            from golden.make_golden import run_ref  # noqa: F401
            kind = "reference"
    except Exception:
        kind = "port"
    if kind == "reference":
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
            mg = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mg)
            fd = os.dup(2)                        # the reference logs "[V2D] Launched..." to stderr
            devnull = os.open(os.devnull, os.O_WRONLY)
            os.dup2(devnull, 2)
            try:
                t0 = time.perf_counter()
                recs, _ = mg.run_ref(luma_sample, mode)
                dt = time.perf_counter() - t0
            finally:
                os.dup2(fd, 2)
                os.close(devnull)
                os.close(fd)
        except Exception:
            kind = "port"
    if kind == "port":
        from oracle_run import oracle_binarize
        t0 = time.perf_counter()
        recs, _ = oracle_binarize(luma_sample, mode=mode)
        dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "host_cores": os.cpu_count(), "kind": kind,
            "sample": f"first {n} frames of the same synthetic batch, {dt:.1f} s of CPU work, single worker thread "
                      f"(the reference runs the path on one thread per stage)"}, recs


def cpu_worker(sample_path, mode, ready_path, go_path):
    """One worker of the all-core CPU baseline (a fresh process, no torch, no GPU): the real reference's VideoToDigital worker (or the
    oracle port) over the frames in `sample_path`, started when `go_path` appears.  Prints one JSON line."""
    import numpy as np
    import libs
    luma = np.load(sample_path, mmap_mode="r")          # one copy of the pixels in the page cache for all workers (read only)
    kind = "port"
    run = None
    try:
        if libs.ref_available():
            import importlib.util
            spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
            mg = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mg)
            libs.load_ref()
            run = lambda: mg.run_ref(luma, mode)        # noqa: E731
            kind = "reference"
    except Exception:
        run = None
    if run is None:
        from oracle_run import oracle_binarize
        libs.load_oracle()
        run = lambda: oracle_binarize(luma, mode=mode)      # noqa: E731
    open(ready_path, "w").close()
    while not os.path.exists(go_path):
        time.sleep(0.002)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 2)                                 # the reference logs to stderr
    t0 = time.time()
    recs, _ = run()
    t1 = time.time()
    import hashlib
    print(json.dumps({"kind": kind, "frames": int(luma.shape[0]), "t0": t0, "t1": t1, "sha": hashlib.sha256(recs.tobytes()).hexdigest()}), flush=True)


def physical_cores():
    """Cores of this host, not hardware threads: distinct (physical id, core id) pairs of /proc/cpuinfo (os.cpu_count() when that cannot be read)."""
    try:
        seen = set(); phys = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), None when unlimited or unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline_all_cores(luma_sample, mode, single_core_frames_per_s=None):
    """the CPU path on every core of this host - one fresh child process per PHYSICAL core, one reference worker each (a worker is two
    threads: the reference's own VideoToDigital worker and the feeder that plays the input plugin; a process per hardware thread oversubscribed
    the host two to one and measured the scheduler).  The reference itself runs the path on a single thread per stage; this is what a host-side
    shard over all cores would reach."""
    import subprocess
    import tempfile
    import numpy as np
    phys = physical_cores()
    quota = cpu_quota()
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = os.cpu_count() or 1
    # as many workers as cores this process may really use at once: physical cores, the CPUs it is allowed on, the container's CPU-time quota
    # (the GPU boxes of the pool show 256 hardware threads and grant 16 CPUs' worth of time: 128 workers there measured the throttle - 12x one core)
    cores = max(1, int(min(phys, affinity, quota if quota else phys)))
    d = tempfile.mkdtemp(prefix="sdv_cpu_")
    try:
        sample = os.path.join(d, "sample.npy")
        np.save(sample, luma_sample)
        go = os.path.join(d, "go")
        env = dict(os.environ)
        env.pop("ROCR_VISIBLE_DEVICES", None)
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", sample, str(mode), os.path.join(d, "ready%d" % i), go],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for i in range(cores)]
        t_wait = time.time()
        while sum(os.path.exists(os.path.join(d, "ready%d" % i)) for i in range(cores)) < cores and time.time() - t_wait < 180 and all(p.poll() is None for p in procs):
            time.sleep(0.01)
        open(go, "w").close()
        res = []
        for p_ in procs:
            out, _ = p_.communicate(timeout=600)
            lines = [ln for ln in out.splitlines() if ln.startswith("{")]
            if p_.returncode == 0 and lines:
                res.append(json.loads(lines[-1]))
        if not res:
            return {"error": "no worker finished", "cores": cores}
        wall = max(r["t1"] for r in res) - min(r["t0"] for r in res)
        frames = sum(r["frames"] for r in res)
        per_worker = frames / len(res) / (sum(r["t1"] - r["t0"] for r in res) / len(res))
        return {"value": frames / wall, "unit": "frames/s", "cores": cores, "physical_cores": phys, "hardware_threads": os.cpu_count(), "cpu_quota_of_the_container": quota,
                "workers_finished": len(res), "kind": res[0]["kind"],
                "workers_rule": "min(physical cores, CPU affinity, cgroup CPU quota) since round 4; the BENCH files of rounds 1-3 ran a worker per hardware thread "
                                "(256 on the pool's boxes, 16 of them runnable at once) - compare round over round by per_worker_frames_per_s x cores",
                "per_worker_frames_per_s": per_worker,
                "per_worker_slowdown_vs_one_core_alone": (single_core_frames_per_s / per_worker) if single_core_frames_per_s else None,
                "all_workers_decoded_the_same_records": len(set(r["sha"] for r in res)) == 1,
                "sample": f"{len(res)} worker processes (one per core this container may use at once: {cores} of {phys} physical cores, {os.cpu_count()} hardware threads, "
                          f"CPU quota {quota}), each the first {res[0]['frames']} frames of the "
                          f"same synthetic batch, {wall:.1f} s of wall time from the first start to the last finish"}
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


def main():
    if len(sys.argv) >= 6 and sys.argv[1] == "--cpu-worker":
        cpu_worker(sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5])
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=10000, help="frames per GPU per step")
    ap.add_argument("--mode", type=int, default=2, help="Binarizer mode (2 = NORMAL, the reference default)")
    ap.add_argument("--noise", type=float, default=4.0)
    ap.add_argument("--cpu-frames", type=int, default=3000)
    ap.add_argument("--cpu-frames-all-cores", type=int, default=1500, help="frames per worker of the all-core CPU baseline")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-stitch", action="store_true", help="skip the extra stitch-stage measurement")
    ap.add_argument("--configs4-frames", type=int, default=100000, help="length of the one tape of BASELINE configs[4] (sharded over the ranks: strong scaling); 0 = skip that leg")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves - as fresh child processes, before this process
        # has touched torch or the GPU - and pass rank 0's JSON line through
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=env).returncode)

    import numpy as np
    import torch
    import torch.distributed as dist
    from sdvpcmdecoder_amd import Engine, synth, LINE_DTYPE

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" in os.environ and world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s): the two must agree")
    backend = os.environ.get("SDV_BENCH_BACKEND", "nccl")     # nccl = RCCL; "gloo" lets the N > 1 path be exercised on a 1-GPU box
    # SDV_BENCH_FORCE_DIST=1: run the sharded (collective) code path with a single rank too - lets the RCCL plumbing be exercised
    # on a 1-GPU box (torch.distributed.run --nproc-per-node 1)
    use_dist = world > 1 or (os.environ.get("SDV_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # before anything initialises HIP
    assert torch.cuda.is_available(), "bench.py needs a GPU (the decode engine has no CPU path)"
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        if backend == "nccl":
            dist.init_process_group(backend=backend, init_method="env://", device_id=dev)     # one rank per GPU, bound to it
        else:
            dist.init_process_group(backend=backend, init_method="env://")

    W, H = 720, 486
    n = args.frames
    # one seamless (cyclic) tape of world*n frames; this rank renders and keeps its own contiguous part of it
    luma, w9 = synth.stc007_frames_torch(n * world, seed=2, device=dev, width=W, height=H, noise_sigma=args.noise, cyclic=True,
                                         frame_range=(rank * n, (rank + 1) * n))
    w9 = w9[rank * n * 2 * 245:(rank + 1) * n * 2 * 245]
    eng = Engine(local_rank)
    eng.setBinarizationMode(args.mode)
    eng.set_profiling(True)
    nrec = n * (H + 3)
    out_lines = torch.empty((nrec + 1, 48), dtype=torch.uint8, device=dev)
    out_stats = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if use_dist:
        from sdvpcmdecoder_amd.sharded import ShardedBinarizeLoop, torch_all_gather
        loop = ShardedBinarizeLoop(eng, rank, world, torch_all_gather(dev if backend == "nccl" else None))

        def decode_batch(first_frame_no, new_file, out):
            loop.step(luma, first_frame_no + rank * n, new_file=new_file, out_lines=out, out_stats=out_stats, stream=stream)
    else:
        loop = None

        def decode_batch(first_frame_no, new_file, out):
            eng.binarize_frames(luma, first_frame_no=first_frame_no, new_file=new_file, out_lines=out, out_stats=out_stats, stream=stream)

    # first pass: start of the stream (cold chain: NEW_FILE, first frame decoded alone)
    decode_batch(1, True, out_lines if rank == 0 else out_lines[1:])
    torch.cuda.synchronize(dev)
    first_recs = out_lines[:1 + 4 * (H + 3)].cpu().numpy().view(LINE_DTYPE).reshape(-1) if rank == 0 else None
    # steady state: every step decodes the batch as the continuation of the stream
    tape_pos = 1 + n * world
    for _ in range(args.warmup):
        decode_batch(tape_pos, False, out_lines[1:])
        tape_pos += n * world
    barrier()
    kernel_ms = 0.0
    rounds = 0
    launched = 0
    general = 0             # frames that needed the full kernel (0 on a tape in steady state)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        decode_batch(tape_pos, False, out_lines[1:])
        tape_pos += n * world
        info = eng.run_info()
        kernel_ms += info.kernel_ms
        rounds += info.rounds
        launched += info.frames_launched
        general += info.frames_general
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # what the timed steps left (looked at below; the legs that follow decode other tapes into the same buffer)
    timed_host = out_lines[1:1 + 8 * (H + 3)].cpu().numpy().view(LINE_DTYPE).reshape(8, H + 3).copy()

    # N > 1, beyond BASELINE's metric: the whole path frames -> PCMSamplePair for ONE tape sharded over the ranks (ShardedDecoder: warm-up,
    # all-gather of the two workers' hand-over states, verify, repair - DESIGN.md section 7), NEW_FILE .. END_FILE, timed as one job
    sharded_full = None
    if use_dist and not args.no_stitch:
        try:
            from sdvpcmdecoder_amd.sharded import ShardedDecoder
            ns = n                                              # frames per rank of this leg: as many as in the timed loop
            total = ns * world
            dec = ShardedDecoder(eng, rank, world, torch_all_gather(dev if backend == "nccl" else None), H)
            f0, f1 = dec.frames_needed(total)
            lum_s, _ = synth.stc007_frames_torch(total, seed=3, device=dev, width=W, height=H, noise_sigma=args.noise, cyclic=True, frame_range=(f0, f1))
            ms_best = None
            for _ in range(3):                                   # the first pass pays the allocations
                barrier()
                t1 = time.perf_counter()
                s_pairs, s_frames = dec.decode(lum_s, total, first_frame_no=1)
                barrier()
                ms = (time.perf_counter() - t1) * 1e3
                ms_best = ms if ms_best is None else min(ms_best, ms)
            one_engine_ms = None
            if world == 1:      # the same job without the sharding protocol: the whole file, NEW_FILE .. END_FILE, through the fused entry of one engine (a cold start like the sharded passes)
                from sdvpcmdecoder_amd.engine import PCM_STC007
                for _ in range(3):
                    eng.reset_stream(); eng.reset_stitcher()
                    torch.cuda.synchronize(dev)
                    t1 = time.perf_counter()
                    o_pairs = eng.decode_frames(PCM_STC007, lum_s, first_frame_no=1, new_file=True, end_file=True)[0]
                    torch.cuda.synchronize(dev)
                    ms = (time.perf_counter() - t1) * 1e3
                    one_engine_ms = ms if one_engine_ms is None else min(one_engine_ms, ms)
                assert o_pairs.shape[0] == s_pairs.shape[0]
                del o_pairs
            tt = torch.tensor([ms_best, float(s_pairs.shape[0]), float(dec.stats["binarize_redo"]), float(dec.stats["stitch_redo"])], dtype=torch.float64,
                              device=dev if backend == "nccl" else "cpu")
            mx = tt.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            sm = tt.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            sharded_full = {"frames": total, "frames_per_rank": ns, "ms": float(mx[0].item()), "frames_per_s": total / (float(mx[0].item()) / 1e3),
                            "sample_pairs": int(sm[1].item()), "ranges_decoded_again": {"binarize": int(sm[2].item()), "stitch": int(sm[3].item())},
                            "one_engine_same_file_ms": one_engine_ms,
                            "note": "one synthetic NTSC tape of `frames` frames, NEW_FILE .. END_FILE, every rank its contiguous range with a 20-frame warm-up and one "
                                    "successor frame; best of three passes, wall clock between two barriers incl. the all-gather of the hand-over states (2 x 120 B + 2 x 3.8 KB per "
                                    "rank); one_engine_same_file_ms (one rank only): the same file through sdv_decode_frames of one engine, cold start as well - "
                                    "the steady-state figure of end_to_end is a continuing tape; not part of `value`"}
            del lum_s, s_pairs, s_frames
            eng.reset_stream(); eng.reset_stitcher()
        except Exception as ex:     # noqa: BLE001 - an extra figure must not take the benchmark line down
            sharded_full = {"error": repr(ex)}

    # BASELINE configs[4], literally: ONE tape of 100 000 NTSC frames, NEW_FILE .. END_FILE, frames -> PCMSamplePair, sharded over the ranks of the
    # job in contiguous frame ranges (ShardedDecoder; 12 500 frames per rank at N = 8) - strong scaling: the tape stays, the ranks share it.  Runs at
    # every N (at N = 1 through the same code, the "all-gather" being the rank's own bytes when no process group exists), so that the driver's
    # 1/2/4/8 runs give the curve north_star asks for.  Reported as absolute frames/s and as a fraction of the HBM roofline on SURVEY 8d's bytes.
    configs4 = None
    if args.configs4_frames > 0 and not args.no_stitch:
        try:
            from sdvpcmdecoder_amd.sharded import ShardedDecoder
            total = args.configs4_frames
            if use_dist:
                assert dist.get_world_size() == args.gpus == world, "the process group does not have --gpus ranks"
                gather4 = torch_all_gather(dev if backend == "nccl" else None)
            else:
                gather4 = lambda b: [b]         # noqa: E731 - one rank: what it would receive from itself
            dec4 = ShardedDecoder(eng, rank, world, gather4, H)
            g0, g1 = dec4.frames_needed(total)
            torch.cuda.synchronize(dev)
            tg = time.perf_counter()
            lum4, _ = synth.stc007_frames_torch(total, seed=5, device=dev, width=W, height=H, noise_sigma=args.noise, frame_range=(g0, g1))
            torch.cuda.synchronize(dev)
            gen_s = time.perf_counter() - tg
            eng.reset_stream(); eng.reset_stitcher()
            ms4 = []
            for _ in range(3):                                   # the first pass pays the allocations
                barrier()
                t1 = time.perf_counter()
                p4, f4 = dec4.decode(lum4, total, first_frame_no=1)
                barrier()
                ms4.append((time.perf_counter() - t1) * 1e3)
            t4 = torch.tensor([min(ms4), float(p4.shape[0]), float(dec4.stats["binarize_redo"]), float(dec4.stats["stitch_redo"]), float(g1 - g0)], dtype=torch.float64,
                              device=dev if (use_dist and backend == "nccl") else "cpu")
            mx4, sm4 = t4.clone(), t4.clone()
            if use_dist:
                dist.all_reduce(mx4, op=dist.ReduceOp.MAX); dist.all_reduce(sm4, op=dist.ReduceOp.SUM)
            best4 = float(mx4[0].item())
            E2E_BYTES4 = W * H + H * 32 + 1470 * 8      # 377 232 B per NTSC frame, frames -> PCMSamplePair
            configs4 = {"workload": "configs[4]: one %d-frame synthetic STC-007 NTSC stream, NEW_FILE .. END_FILE, frames -> PCMSamplePair, sharded over %d rank(s) in contiguous "
                                    "frame ranges with one all-gather of the hand-over states" % (total, world),
                        "scaling": "strong", "frames": total, "ranks": world, "frames_per_rank": (total + world - 1) // world,
                        "frames_rendered_per_rank_max": int(mx4[4].item()),
                        "ms": best4, "frames_per_s": total / (best4 / 1e3),
                        "roofline": {"bound": "hbm", "achieved": total * E2E_BYTES4 / best4 / 1e6, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                                     "frac": total * E2E_BYTES4 / best4 / 1e6 / (HBM_PEAK_GBS * world), "algorithmic_bytes_per_frame": E2E_BYTES4,
                                     "note": "whole-job bytes over the wall clock of the slowest rank, against the HBM peak of all ranks' GPUs"},
                        "sample_pairs": int(sm4[1].item()), "ranges_decoded_again": {"binarize": int(sm4[2].item()), "stitch": int(sm4[3].item())},
                        "all_gathers_per_pass": dec4.stats["gathers"] // 3,
                        "backend": (backend if use_dist else "none (one process)"), "process_group_ranks": (dist.get_world_size() if use_dist else 1),
                        "passes_ms": ms4, "tape_render_s": gen_s,
                        "note": "best of three passes, wall clock between two barriers (max over the ranks), cold start of a file every pass; not part of `value` "
                                "(which is BASELINE's per-GPU batch, weak scaling)"}
            del lum4, p4, f4
            eng.reset_stream(); eng.reset_stitcher()
        except Exception as ex:     # noqa: BLE001 - an extra figure must not take the benchmark line down
            configs4 = {"error": repr(ex)}

    # beyond BASELINE's metric: the stitch stage (STC007DataStitcher -> PCMSamplePair) over the records just produced, and the
    # whole path frames -> PCM, both as a continuing stream (every step continues the tape, like the binarize steps above)
    stitch = None
    if not args.no_stitch and world == 1:
        eng.reset_stitcher()
        lines_all = out_lines[:1 + nrec]                                  # NEW_FILE line + n frames
        sp = torch.empty((n * 1470 + 65536, 12), dtype=torch.uint8, device=dev)
        sf = torch.empty((n + 64, 64), dtype=torch.uint8, device=dev)
        eng.binarize_frames(luma, first_frame_no=1, new_file=True, out_lines=out_lines, out_stats=out_stats, stream=stream)
        pairs, frs = eng.stitch_frames(lines_all, out_pairs=sp, out_frames=sf, stream=stream)
        first_pairs = pairs[:4 * 1470].cpu().numpy().copy() if rank == 0 else None
        frame_no = 1 + n
        # one untimed continuing step: the stream's steady shape (the carried frame makes every later call one frame longer)
        eng.binarize_frames(luma, first_frame_no=frame_no, new_file=False, out_lines=out_lines[1:], out_stats=out_stats, stream=stream)
        eng.stitch_frames(out_lines[1:1 + nrec], out_pairs=sp, out_frames=sf, stream=stream)
        frame_no += n
        st_ms = e2e_ms = 0.0
        st_rounds = 0
        st_dev_ms = 0.0
        k_steps = max(1, min(args.steps, 5))
        for _ in range(k_steps):
            # the tape goes on: frame numbers keep increasing, the chain states of both stages carry over
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            eng.binarize_frames(luma, first_frame_no=frame_no, new_file=False, out_lines=out_lines[1:], out_stats=out_stats, stream=stream)
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            pairs, frs = eng.stitch_frames(out_lines[1:1 + nrec], out_pairs=sp, out_frames=sf, stream=stream)
            torch.cuda.synchronize(dev)
            t3 = time.perf_counter()
            st_ms += (t3 - t2) * 1e3
            e2e_ms += (t3 - t1) * 1e3
            st_rounds += eng.stitch_info().rounds
            st_dev_ms += eng.stitch_info().device_ms
            frame_no += n
        # the same tape through the fused entry with the audio stage behind it: video -> masked PCM (sdv_decode_frames, all three workers).
        # Wall clock as a caller sees it: without the engine's event pairs (sdv_set_profiling is a diagnostic; with it on the fused entry does not queue its
        # stitch kernels behind the frame kernel ahead of the host's look at that round).
        eng.set_profiling(False)
        eng.set_audio_masking(5)        # DROP_INTER_LIN_BLOCK
        full_ms = 0.0
        for _ in range(k_steps):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            eng.decode_frames(2, luma, first_frame_no=frame_no, with_audio=True, stream=stream)
            torch.cuda.synchronize(dev)
            full_ms += (time.perf_counter() - t1) * 1e3
            frame_no += n
        # frames -> PCMSamplePair through the fused entry (sdv_decode_frames without the audio stage): the path north_star names, on SURVEY 8d's bytes
        fused_ms = 0.0
        fp = torch.empty(((n + 2) * 1800 + 8192, 12), dtype=torch.uint8, device=dev)        # the caller's buffers, as in the two-call loop above
        for _ in range(k_steps):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            eng.decode_frames(2, luma, first_frame_no=frame_no, with_audio=False, stream=stream, out_pairs=fp, out_frames=sf, out_stats=out_stats)
            torch.cuda.synchronize(dev)
            fused_ms += (time.perf_counter() - t1) * 1e3
            frame_no += n
        eng.set_profiling(True)
        e2e_best_ms = min(e2e_ms, fused_ms) / k_steps
        E2E_BYTES = W * H + H * 32 + 1470 * 8           # 349 920 B luma + 486 x 32 B line records + 1470 x 8 B sample pairs = 377 232 B per NTSC frame
        end_to_end = {"workload": f"{n}-frame NTSC STC-007 batch resident in HBM, frames -> PCMSamplePair (binarize + stitch + deinterleave + P/Q ECC), continuing tape",
                      "ms_per_step": e2e_best_ms, "frames_per_s": n / e2e_best_ms * 1e3,
                      "two_calls_ms_per_step": e2e_ms / k_steps, "fused_entry_ms_per_step": fused_ms / k_steps,
                      "roofline": {"bound": "hbm", "achieved": n * E2E_BYTES / e2e_best_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": n * E2E_BYTES / e2e_best_ms / 1e6 / HBM_PEAK_GBS, "algorithmic_bytes_per_frame": E2E_BYTES,
                                   "note": "wall clock of the whole chain (several kernels and their host round trips), not one kernel's launch time"}}
        stitch = {"frames_to_masked_pcm_ms_per_step": full_ms / k_steps, "frames_to_masked_pcm_frames_per_s": n / (full_ms / k_steps) * 1e3,
                  "stitch_ms_per_step": st_ms / k_steps, "stitch_frames_per_s": n / (st_ms / k_steps) * 1e3,
                  "frames_to_pcm_ms_per_step": e2e_best_ms, "frames_to_pcm_frames_per_s": n / e2e_best_ms * 1e3,
                  "frames_to_pcm_two_calls_ms_per_step": e2e_ms / k_steps, "frames_to_pcm_fused_entry_ms_per_step": fused_ms / k_steps,
                  "sample_pairs_per_step": int(pairs.shape[0]), "rounds_per_step": st_rounds / k_steps, "stitch_device_ms_per_step": st_dev_ms / k_steps,
                  "note": "stitch = frame reassembly + CWD + deinterleave + P/Q ECC to PCMSamplePair (sdv_stitch_frames), wall clock per "
                          "batch incl. its host round trips; frames_to_pcm = the whole path frames -> PCMSamplePair, the faster of the fused entry "
                          "(sdv_decode_frames) and the two separate calls (both given); not part of `value`"}

    # BASELINE configs[2]: STC-007 PAL 720x576, Deinterleaver + P/Q error correction on - a clean tape, and the tape of SURVEY 8d C3 (every 97th
    # line of a frame lost, a bit cell inverted on one line in 53: P and Q corrections, BROKEN blocks, seam masking at work)
    pal = None
    if not args.no_stitch and world == 1:
        HP = 576
        PAL_BYTES = W * HP + HP * 32 + 1764 * 8         # 414 720 + 18 432 + 14 112 = 447 264 B per PAL frame
        npal = n
        pal = {"note": "configs[2]: synthetic STC-007 PAL 720x576 frames (294 lines per field, 288 visible) resident in HBM -> sdv_binarize_frames -> sdv_stitch_frames "
                       "with P, Q and CWD corrections on, continuing tape, wall clock per batch incl. host round trips; roofline on SURVEY 8d's 447 264 B per frame; "
                       "not part of `value`"}
        luma_p, _w = synth.stc007_frames_torch(npal, seed=7, device=dev, width=W, height=HP, lines_per_field=294, noise_sigma=args.noise, cyclic=True)
        nrec_p = npal * (HP + 3)
        ol_p = torch.empty((nrec_p + 1, 48), dtype=torch.uint8, device=dev)
        os_p = torch.empty((npal, 32), dtype=torch.uint8, device=dev)
        sp_p = torch.empty((npal * 1764 + 65536, 12), dtype=torch.uint8, device=dev)
        sf_p = torch.empty((npal + 64, 64), dtype=torch.uint8, device=dev)
        pal_keep = {}
        for tape_name in ("clean", "lost_lines_and_flipped_cells"):
            if tape_name != "clean":
                # every frame damaged, ~11 lines per frame that read at no reference level: in NORMAL mode each of them costs the reference's full level
                # sweep (170 levels x 24 marker searches + ladder: 1.3 ms per line on a CPU core) and the speculation needs tens of rounds - a fifth of
                # the batch keeps the default run within its minutes
                npal = max(1, n // 5)
                luma_p = luma_p[:npal]; nrec_p = npal * (HP + 3)
                luma_p[:, 96::97, :] = 16
                flat = luma_p.view(-1, W)
                gsel = torch.Generator(device=dev); gsel.manual_seed(53)
                rows = torch.arange(0, flat.shape[0], 53, device=dev)
                xs = 12 + (torch.randint(4, 132, rows.shape, generator=gsel, device=dev) * (W - 24)) // 137
                for dx in range(5):
                    flat[rows, xs + dx] = (230 - flat[rows, xs + dx].to(torch.int16)).clamp_(0, 255).to(torch.uint8)
            eng.setBinarizationMode(args.mode)
            eng.reset_stream(); eng.reset_stitcher()
            eng.binarize_frames(luma_p, first_frame_no=1, new_file=True, out_lines=ol_p, out_stats=os_p, stream=stream)
            pp0_, _ff0 = eng.stitch_frames(ol_p[:1 + nrec_p], out_pairs=sp_p, out_frames=sf_p, stream=stream)
            # the start of the tape for the CPU leg below: pixels, and what the GPU made of them (records and sample pairs)
            k_cpu = min(npal, 300 if tape_name == "clean" else 40)
            pal_keep[tape_name] = (luma_p[:k_cpu].cpu().numpy(), ol_p[:1 + k_cpu * (HP + 3)].cpu().numpy().copy(), pp0_[:k_cpu * 1764].cpu().numpy().copy())
            fno = 1 + npal
            eng.binarize_frames(luma_p, first_frame_no=fno, out_lines=ol_p[1:], out_stats=os_p, stream=stream)
            eng.stitch_frames(ol_p[1:1 + nrec_p], out_pairs=sp_p, out_frames=sf_p, stream=stream)
            fno += npal
            k_steps = max(1, min(args.steps, 3))
            b_ms = s_ms = 0.0; b_rounds = s_rounds = b_general = 0; b_sweeps = 0
            eng.set_profiling(False)                    # wall clock as a caller sees it: no event pair around every round of the damaged tape
            for _ in range(k_steps):
                torch.cuda.synchronize(dev); t1 = time.perf_counter()
                eng.binarize_frames(luma_p, first_frame_no=fno, out_lines=ol_p[1:], out_stats=os_p, stream=stream)
                torch.cuda.synchronize(dev); t2 = time.perf_counter()
                pp_, ff_ = eng.stitch_frames(ol_p[1:1 + nrec_p], out_pairs=sp_p, out_frames=sf_p, stream=stream)
                torch.cuda.synchronize(dev); t3 = time.perf_counter()
                b_ms += (t2 - t1) * 1e3; s_ms += (t3 - t2) * 1e3
                b_rounds += eng.run_info().rounds; b_general += eng.run_info().frames_general; b_sweeps += eng.run_info().sweeps; s_rounds += eng.stitch_info().rounds
                fno += npal
            fused_p_ms = None
            if tape_name == "clean":        # ... and the same frames through the fused entry (one call: the records stay in the engine)
                fused_p_ms = 0.0
                for r_ in range(-1, k_steps):
                    torch.cuda.synchronize(dev); t1 = time.perf_counter()
                    eng.decode_frames(2, luma_p, first_frame_no=fno, with_audio=False, stream=stream, out_pairs=sp_p, out_frames=sf_p, out_stats=os_p)
                    torch.cuda.synchronize(dev)
                    if r_ >= 0: fused_p_ms += (time.perf_counter() - t1) * 1e3
                    fno += npal
                fused_p_ms /= k_steps
            eng.set_profiling(True)
            tot = (b_ms + s_ms) / k_steps
            pr = pp_[:, :].cpu().numpy().reshape(-1).view(np.dtype([("w", "<i2", (2,)), ("fl", "u1", (2,)), ("rate", "<u2"), ("e", "u1"), ("srv", "u1"), ("_p", "<u2")]))
            pal[tape_name] = {"frames_per_step": npal, "binarize_ms_per_step": b_ms / k_steps, "stitch_ms_per_step": s_ms / k_steps, "ms_per_step": tot,
                              "frames_per_s": npal / tot * 1e3, "binarize_rounds_per_step": b_rounds / k_steps, "frames_by_full_kernel_per_step": b_general / k_steps,
                              "reference_level_sweeps_per_step": b_sweeps / k_steps, "timed_steps": k_steps,
                              "fused_entry_ms_per_step": fused_p_ms,
                              "stitch_rounds_per_step": s_rounds / k_steps, "sample_pairs_per_step": int(pp_.shape[0]), "sample_rate": int(pr["rate"][len(pr) // 2]),
                              "samples_valid_share": float(((pr["fl"] & 2) != 0).mean()),
                              "roofline": {"bound": "hbm", "achieved": npal * PAL_BYTES / tot / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": npal * PAL_BYTES / tot / 1e6 / HBM_PEAK_GBS, "algorithmic_bytes_per_frame": PAL_BYTES}}
        del luma_p, ol_p, os_p, sp_p, sf_p
        eng.reset_stream(); eng.reset_stitcher()

    # damaged tapes in the binarize stage (the speculation across frames has to repair what it predicted wrong): 16 lost lines and 16 jumps of
    # the data window per 10 000 frames - the workloads of tools/dropout_probe.py / tools/jump_probe.py, on the tape `value` was measured on
    damaged = None
    if not args.no_stitch and world == 1:
        damaged = {"note": "sdv_binarize_frames over the benchmark's tape with damage (16 lost lines / 16 jumps of the data window per 10 000 frames; the whole tape two pixels beside its coordinates; "
                           "every 97th row of every frame lost), continuing stream, wall clock per batch; rounds = launches of the frame kernels "
                           "until every frame was decoded from its predecessor's real state; frames_launched counts re-decodes; frames_by_full_kernel those that "
                           "needed the kernel with the general path; compare with `value` (clean tape, 1 round); not part of `value`"}
        per = max(1, n // 625)                          # 16 per 10 000 frames
        rng_d = np.random.default_rng(16)
        for kind in ("lost_lines", "window_jumps", "beside_coordinates", "lost_lines_in_every_frame"):
            lum = luma.clone()
            if kind == "lost_lines":
                for f_ in sorted(rng_d.choice(np.arange(50, n - 50), size=per, replace=False)):
                    lum[int(f_), int(rng_d.integers(40, 440))] = 16
            elif kind == "beside_coordinates":          # the whole tape two pixels beside the coordinates the binarizer holds: every line reads, on a later shift stage
                lum = torch.roll(luma, -2, dims=2)
            elif kind == "lost_lines_in_every_frame":   # every 97th row of every frame lost: every frame through the general kernel, several times
                lum[:, 96::97, :] = 16
            else:
                at = 0
                for f_ in sorted(rng_d.choice(np.arange(50, n - 50), size=per, replace=False)):
                    to = at
                    while to == at:
                        to = int(rng_d.integers(-8, 9))
                    lum[int(f_):] = torch.roll(luma[int(f_):], to, dims=2); at = to
            eng.setBinarizationMode(args.mode)
            eng.reset_stream()
            eng.binarize_frames(luma, first_frame_no=1, new_file=True, out_lines=out_lines, out_stats=out_stats, stream=stream)
            k_steps = max(1, min(args.steps, 3))
            d_ms = 0.0; d_kms = 0.0; d_rounds = d_launched = d_general = 0
            for r_ in range(-2, k_steps):               # (-2: once untimed - the first damaged batch of a process pays first uses: the full kernel's code, pinned buffers;
                                                        #  -1: once more untimed with the engine's event pairs on: where the kernel time comes from)
                # (the jumps leave the window displaced at the end of the batch: every step starts from the clean tape's state again)
                eng.binarize_frames(luma, first_frame_no=1 + (2 * r_ + 3) * n, out_lines=out_lines[1:], out_stats=out_stats, stream=stream)
                # the wall clock of the call as a caller sees it: without the engine's event pair around every round (sdv_set_profiling), which is
                # what the untimed first pass reads the kernel time from
                eng.set_profiling(r_ == -1)
                torch.cuda.synchronize(dev); t1 = time.perf_counter()
                eng.binarize_frames(lum, first_frame_no=1 + (2 * r_ + 4) * n, out_lines=out_lines[1:], out_stats=out_stats, stream=stream)
                torch.cuda.synchronize(dev)
                eng.set_profiling(True)
                i_ = eng.run_info()
                if r_ < 0:
                    if r_ == -1:
                        d_kms = i_.kernel_ms * k_steps
                    continue
                d_ms += (time.perf_counter() - t1) * 1e3
                d_rounds += i_.rounds; d_launched += i_.frames_launched; d_general += i_.frames_general
            damaged[kind] = {"events_per_step": per if kind in ("lost_lines", "window_jumps") else None, "frames_per_step": n, "ms_per_step": d_ms / k_steps, "frames_per_s": n / (d_ms / k_steps) * 1e3,
                             "kernel_ms_per_step": d_kms / k_steps, "kernel_ms_from": "a separate warm pass with an event pair around every round (not one of the timed passes)", "rounds_per_step": d_rounds / k_steps, "frames_launched_per_step": d_launched / k_steps,
                             "frames_by_full_kernel_per_step": d_general / k_steps}
            del lum
        eng.reset_stream()

    # the other format branch built so far: the PCM-1 back half (PCM1DataStitcher -> PCMSamplePair) over a tape of the same length
    pcm1 = None
    if not args.no_stitch and world == 1:
        from sdvpcmdecoder_amd import synth as _synth
        p1_recs = _synth.pcm1_tape(n)
        p1_dev = torch.from_numpy(p1_recs.view(np.uint8).reshape(len(p1_recs), 32)).to(dev)
        p1p = torch.empty((n * 1470 + 64, 12), dtype=torch.uint8, device=dev)
        p1f = torch.empty((n + 64, 52), dtype=torch.uint8, device=dev)
        eng.pcm1_stitch_frames(p1_dev, out_pairs=p1p, out_frames=p1f, stream=stream)
        k_steps = max(1, min(args.steps, 5))
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(k_steps):
            pp, pf = eng.pcm1_stitch_frames(p1_dev, out_pairs=p1p, out_frames=p1f, stream=stream)
        torch.cuda.synchronize(dev)
        p1_ms = (time.perf_counter() - t1) * 1e3 / k_steps
        p1_bytes = len(p1_recs) * 32 + n * (1470 * 12 + 52)
        pcm1 = {"ms_per_step": p1_ms, "frames_per_s": n / p1_ms * 1e3, "algorithmic_gb_per_s": p1_bytes / p1_ms / 1e6,
                "sample_pairs_per_step": int(pp.shape[0]),
                "note": "PCM-1 frames (490 line records each) -> trim, field split, padding, deinterleave to PCMSamplePair "
                        "(sdv_pcm1_stitch_frames), wall clock per batch incl. its host round trips; not part of `value`"}
        p1_first = pp[:3000 * 1470].cpu().numpy().copy() if rank == 0 else None

    # ... the PCM-16x0 back half (PCM16X0DataStitcher): sub-line records -> sample pairs, both interleave formats
    pcm16 = None
    if not args.no_stitch and world == 1:
        from sdvpcmdecoder_amd import synth as _synth
        pcm16 = {"note": "PCM-1630 frames (1470 sub-line records each, damaged tape with rows lost at the top and bottom of the fields) -> trim, "
                         "field split, padding detection by P-code checks over every candidate padding, deinterleave + P-code correction to "
                         "PCMSamplePair (sdv_pcm16x0_stitch_frames), wall clock per batch incl. its host round trips; not part of `value`"}
        p16_keep = {}
        for fmt, ei in (("si", False), ("ei", True)):
            recs16 = _synth.pcm16x0_tape(n, ei=ei)
            d16 = torch.from_numpy(recs16.view(np.uint8).reshape(len(recs16), 36)).to(dev)
            st16 = eng.default_pcm16x0_stitch_settings(); st16.format = 2 if ei else 1
            o16p = torch.empty((n * 1470 + 64, 12), dtype=torch.uint8, device=dev)
            o16f = torch.empty((n + 64, 56), dtype=torch.uint8, device=dev)
            eng.set_pcm16x0_stitch_settings(st16)
            eng.pcm16x0_stitch_frames(d16, out_pairs=o16p, out_frames=o16f, stream=stream)
            k_steps = max(1, min(args.steps, 5))
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(k_steps):
                eng.set_pcm16x0_stitch_settings(st16)          # a fresh stitcher: every step decodes the same tape from its start
                pp16, pf16 = eng.pcm16x0_stitch_frames(d16, out_pairs=o16p, out_frames=o16f, stream=stream)
            torch.cuda.synchronize(dev)
            ms16 = (time.perf_counter() - t1) * 1e3 / k_steps
            b16 = len(recs16) * 36 + n * (1470 * 12 + 56)
            pcm16[fmt] = {"ms_per_step": ms16, "frames_per_s": n / ms16 * 1e3, "algorithmic_gb_per_s": b16 / ms16 / 1e6, "sample_pairs_per_step": int(pp16.shape[0]),
                          "frames_with_padding_found": int((pf16[:, 54] & 32).ne(0).sum().item())}
            if rank == 0:
                p16_keep[fmt] = (recs16, pp16[:200 * 1470].cpu().numpy().copy())
            del d16, o16p, o16f

    # the consumer of every PCMSamplePair stream: AudioProcessor (dropout masking) on the pair stream of a tape of the same length,
    # in three states of wear; the output of each is packed to WAV bytes (SamplesToWAV, 8f-2)
    audio = None
    if not args.no_stitch and world == 1:
        import audio_api as _A
        audio = {"note": "PCMSamplePair stream of n frames (1470 pairs each, NEW_FILE .. END_FILE) -> AudioProcessor window automaton in linear-interpolation "
                         "mode (sdv_audio_process) -> 16-bit stereo PCM of the WAV file (sdv_wav_pack); wall clock per tape incl. the host round trips; "
                         "algorithmic bytes = 12 B read + 12 B written per pair (+ 12 + 4 for the WAV packing); not part of `value`"}
        npairs = n * 1470
        rng = np.random.default_rng(9)
        starts = np.sort(rng.integers(1000, npairs - 5000, max(1, n // 25)))
        tapes = (("clean", dict()),
                 ("dropout_every_25_frames", dict(runs=[(int(s_), int(rng.integers(1, 700)), int(rng.integers(0, 3))) for s_ in starts])),
                 ("invalid_word_in_every_window", dict(p_bad=0.01)))
        a_out = torch.empty((npairs + 1024, 12), dtype=torch.uint8, device=dev)
        a_pur = torch.empty((16, 16), dtype=torch.uint8, device=dev)
        a_pcm = torch.empty((npairs + 1024, 2), dtype=torch.int16, device=dev)
        eng.set_audio_masking(_A.DROP_INTER_LIN_WORD)
        audio_keep = {}
        for name, kw in tapes:
            tape = _A.tape(["N", _A.audio(npairs, 3, tone=False, **kw), "E"])
            d_t = torch.from_numpy(tape.view(np.uint8).reshape(len(tape), 12)).to(dev)
            eng.reset_audio(); eng.audio_process(d_t, stop=True, out_pairs=a_out, out_purges=a_pur, stream=stream)
            k_steps = max(1, min(args.steps, 5))
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(k_steps):
                eng.reset_audio()
                ao, apu, am = eng.audio_process(d_t, stop=True, out_pairs=a_out, out_purges=a_pur, stream=stream)
            torch.cuda.synchronize(dev)
            a_ms = (time.perf_counter() - t1) * 1e3 / k_steps
            t1 = time.perf_counter()
            for _ in range(k_steps):
                eng.wav_pack(ao, out=a_pcm, stream=stream)
            torch.cuda.synchronize(dev)
            w_ms = (time.perf_counter() - t1) * 1e3 / k_steps
            audio[name] = {"ms_per_step": a_ms, "frames_per_s": n / a_ms * 1e3, "pairs_per_s": npairs / a_ms * 1e3, "algorithmic_gb_per_s": 24 * npairs / a_ms / 1e6,
                           "masked_samples": int(am), "wav_pack_ms_per_step": w_ms, "wav_pack_gb_per_s": 16 * npairs / w_ms / 1e6}
            if rank == 0:
                k = min(n, 2000) * 1470
                audio_keep[name] = (tape[:k + 1], ao[:k - 600].cpu().numpy().copy())
            del d_t
        del a_out, a_pcm

    # ... and its front half: video lines -> PCM1Line records (sdv_pcm1_binarize_lines), a tape that plays (every line preset from a
    # decoded neighbour) and the cold case (nothing preset: the marker-less coordinate search on every line)
    pcm1f = None
    if not args.no_stitch and world == 1:
        from sdvpcmdecoder_amd import synth as _synth
        LPF = 490
        f_frames = min(n, 2000)
        base, base_words = _synth.pcm1_random_lines(LPF * 8, seed=21, x0=5, x1=713, noise_sigma=args.noise)
        nl = f_frames * LPF
        fl = torch.from_numpy(base).to(dev).repeat((nl + len(base) - 1) // len(base), 1)[:nl].contiguous()
        gen = torch.Generator(device=dev); gen.manual_seed(5)
        for i in range(0, nl, 1 << 18):         # every line its own pixels: +-3 on top, far inside the decision levels
            blk = fl[i:i + (1 << 18)]
            blk.copy_((blk.to(torch.int16) + torch.randint(-3, 4, blk.shape, generator=gen, device=dev, dtype=torch.int16)).clamp_(0, 255).to(torch.uint8))
        eng.setBinarizationMode(args.mode)
        cold_n = min(nl, 20 * LPF)
        rec_dt = np.dtype([("frame", "<u4"), ("line", "<u2"), ("words", "<u2", (7,)), ("crc", "<u2"), ("start", "<i2"), ("stop", "<i2"), ("lv", "u1", (5,)),
                           ("hs", "u1", (2,)), ("srv", "u1"), ("pk", "u1", (2,)), ("flags", "u1"), ("_p", "u1", (3,))])
        cold = eng.pcm1_binarize_lines(fl[:cold_n], stream=stream)
        torch.cuda.synchronize(dev)
        cr = cold.cpu().numpy().reshape(-1).view(rec_dt)
        k0 = int(np.flatnonzero((cr["flags"] & 64) != 0)[0])
        st = np.zeros(nl, dtype=np.dtype([("black", "u1"), ("white", "u1"), ("ref", "u1"), ("_p", "u1"), ("start", "<i2"), ("stop", "<i2"), ("d", "u1"), ("_p2", "u1")]))
        st["black"], st["white"], st["ref"], st["start"], st["stop"] = cr["lv"][k0][0], cr["lv"][k0][1], cr["lv"][k0][3], cr["start"][k0], cr["stop"][k0]
        d_st = torch.from_numpy(st.view(np.uint8).reshape(nl, 10)).to(dev)
        fo = torch.empty((nl, 40), dtype=torch.uint8, device=dev)
        res = {}
        k_steps = max(1, min(args.steps, 5))
        for name, a_l, a_s, cnt in (("warm", fl, d_st, nl), ("cold", fl[:cold_n], None, cold_n)):
            eng.pcm1_binarize_lines(a_l, a_s, out_lines=fo[:cnt], stream=stream)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(k_steps):
                eng.pcm1_binarize_lines(a_l, a_s, out_lines=fo[:cnt], stream=stream)
            torch.cuda.synchronize(dev)
            res[name] = (time.perf_counter() - t1) / k_steps
            if name == "warm":
                warm_host = fo[:3 * LPF * 8].cpu().numpy().reshape(-1).view(rec_dt).copy()
        words_ok1 = bool((warm_host["words"] == base_words[np.arange(len(warm_host)) % len(base)]).all()) and bool(((warm_host["flags"] & 4) != 0).all())
        pcm1f = {"lines_per_step": nl, "ms_per_step": res["warm"] * 1e3, "lines_per_s": nl / res["warm"], "frames_per_s": nl / res["warm"] / LPF,
                 "algorithmic_gb_per_s": nl * (720 + 10 + 40) / res["warm"] / 1e9, "decoded_words_match_generator": words_ok1,
                 "cold_lines": cold_n, "cold_ms": res["cold"] * 1e3, "cold_lines_per_s": cold_n / res["cold"],
                 "note": "PCM-1 video lines (720 px) -> PCM1Line records (sdv_pcm1_binarize_lines, Binarizer mode as above): `warm` = every line "
                         "preset with the levels and coordinates of a decoded line (what the frame driver hands on while a tape plays), `cold` = "
                         "nothing preset, every line runs the 25 x 25 coordinate search; not part of `value`"}
        pcm1f_sample = (fl[:3 * LPF].cpu().numpy(), st[:3 * LPF].copy(), warm_host[:3 * LPF].copy(), fl[:LPF // 2].cpu().numpy(), cr[:LPF // 2].copy()) if rank == 0 else None
        del fl, fo, d_st

    # ... and the frame drivers of the two marker-less formats (VideoToDigital::doBinarize with TYPE_PCM1 / TYPE_PCM16X0): whole frames in,
    # line records out, with the reference's per-frame coordinate prescan - BASELINE configs[3], the format dispatch
    fmt_stages = {}
    if not args.no_stitch and world == 1:
        from sdvpcmdecoder_amd import synth as _synth
        nf = min(n, 10000)         # BASELINE's batch (the frame kernels want more frames than the GPU has SIMDs: 2 000 frames leave half of them idle)
        for key, gen, call, rec_bytes, per_frame in (("pcm1_frames_stage", _synth.pcm1_frames, eng.pcm1_binarize_frames, 40, H + 3),
                                                     ("pcm16x0_frames_stage", _synth.pcm16x0_frames, eng.pcm16x0_binarize_frames, 36, 3 * H + 3)):
            base, _w = gen(8, seed=530, height=H, width=W, noise_sigma=args.noise)
            fl = torch.from_numpy(np.tile(base, ((nf + 7) // 8, 1, 1))[:nf]).to(dev)
            ol = torch.empty((nf * per_frame + 1, rec_bytes), dtype=torch.uint8, device=dev)
            osx = torch.empty((nf, 32), dtype=torch.uint8, device=dev)
            eng.setBinarizationMode(args.mode)
            eng.reset_stream()
            call(fl, first_frame_no=1, new_file=True, out_lines=ol, out_stats=osx, stream=stream)          # the start of the tape
            first = ol[:1 + 2 * per_frame].cpu().numpy().copy()
            call(fl, first_frame_no=1 + nf, out_lines=ol[1:], out_stats=osx, stream=stream)
            k_steps = max(1, min(args.steps, 3))
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter(); rnd = 0
            for r in range(k_steps):
                call(fl, first_frame_no=1 + (r + 2) * nf, out_lines=ol[1:], out_stats=osx, stream=stream)
                rnd += eng.run_info().rounds
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t1) * 1e3 / k_steps
            fmt_stages[key] = ({"frames_per_step": nf, "ms_per_step": ms, "frames_per_s": nf / ms * 1e3, "rounds_per_step": rnd / k_steps,
                                "algorithmic_gb_per_s": nf * (W * H + per_frame * rec_bytes + 32) / ms / 1e6,
                                "note": "synthetic %s frames (720x486, every row a PCM line) -> line records incl. the per-frame coordinate prescan, "
                                        "Binarizer mode as above, continuing tape, wall clock per batch; not part of `value`" % key.split("_")[0].upper()},
                               base.copy(), first)
            del fl, ol, osx
        eng.reset_stream()

    # correctness of what was timed: all lines decode to the generator's words
    host = timed_host
    lines = np.concatenate([host[:, :243], host[:, 244:487]], axis=1)
    f = np.arange(8)[:, None]
    r = np.arange(243)[None, :]
    idx = np.concatenate([f * 490 + 2 + r, f * 490 + 245 + 2 + r], axis=1)
    words_ok = bool((lines["words"] == w9.cpu().numpy()[idx].astype(np.uint16)).all())

    if rank == 0:
        total_frames = n * world * args.steps
        value = total_frames / dt
        bpf = algorithmic_bytes_per_frame(W, H)
        avg_launch_ms = kernel_ms / max(rounds, 1)
        frames_per_launch = launched / max(rounds, 1)
        achieved = (bpf * frames_per_launch) / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        traffic, stale = measured_hbm_traffic(frames_per_launch, f"frames={n},mode={args.mode},noise={args.noise},width={W},height={H}")
        out = {
            "metric": "decoded video frames/sec (720x486 STC-007), binarize+bit-extract+CRC, bit-exact vs CPU",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"configs[1]: {n}-frame synthetic STC-007 NTSC 720x486 batch per GPU resident in HBM, "
                                   f"Binarizer mode {args.mode}, noise sigma {args.noise}, binarize+bit-extract+CRC only",
                       "frames_per_gpu_per_step": n, "speculation_rounds_per_step": rounds / args.steps,
                       "sharding": ("one tape of %d frames per step in contiguous ranges, 120-byte state all-gather per step, %d range re-decodes"
                                    % (n * world, loop.redo)) if loop is not None else "single GPU",
                       "decoded_words_match_generator": words_ok},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "bytes per launch, from the rocprofv3 PMC passes committed under profiles/ (same sources, same workload)",
                         "kernel": "sdv_k_stc007_frames_lean" if general == 0 else "sdv_k_stc007_frames_lean + sdv_k_stc007_frames", "avg_launch_ms": avg_launch_ms,
                         "algorithmic_bytes_per_launch": bpf * frames_per_launch},
        }
        if stale is not None:
            out["roofline"]["traffic_from_committed_profile"] = stale
        if sharded_full is not None:
            out["sharded_full_path"] = sharded_full
        if configs4 is not None:
            out["configs4_strong"] = configs4
        if world == 1 and not args.no_stitch:
            # the boundary takes device pointers; a caller that keeps its frames in host memory pays this on top (never part of `value`)
            try:
                hb = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
                db = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
                db.copy_(hb, non_blocking=True)
                torch.cuda.synchronize(dev)
                th = time.perf_counter()
                for _ in range(4):
                    db.copy_(hb, non_blocking=True)
                torch.cuda.synchronize(dev)
                bw = 4 * (256 << 20) / (time.perf_counter() - th)
                out["host_fed"] = {"h2d_gb_per_s": bw / 1e9, "frames_per_s_bound_by_pcie": bw / (W * H),
                                   "note": "pinned host -> HBM copy rate measured here; 349 920 B of luma per frame have to cross it when the "
                                           "frames start in host memory, so that path is bound by the link, not by the kernels"}
                del hb, db
            except Exception as ex:     # noqa: BLE001 - an extra figure must not take the benchmark line down
                out["host_fed"] = {"error": repr(ex)}
        if stitch is not None:
            out["stitch_stage"] = stitch
            out["end_to_end"] = end_to_end
        if pal is not None:
            out["pal_stage"] = pal
        if damaged is not None:
            out["damaged_tape"] = damaged
        if pcm1 is not None:
            out["pcm1_stage"] = pcm1
        if pcm1f is not None:
            out["pcm1_front_stage"] = pcm1f
        if pcm16 is not None:
            out["pcm16x0_stage"] = pcm16
        if audio is not None:
            out["audio_stage"] = audio
            if not args.no_cpu:
                # the real AudioProcessor (its own processAudio loop on a thread, oracle/_ref) or the oracle port on the head of each tape
                import libs as _libs
                import audio_api as _A
                use_ref = _libs.ref_available()
                lib = _libs.load_ref() if use_ref else _libs.load_oracle()
                for name, (head, first) in audio_keep.items():
                    sample = _A.tape([head, "E"])
                    fd = os.dup(2); devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 2)       # the worker logs to stderr
                    try:
                        t0 = time.perf_counter()
                        cr = _A.run_cpu(lib, "ref_" if use_ref else "orc_", sample, _A.DROP_INTER_LIN_WORD, np.array([len(sample)], dtype=np.uint64), 1)
                        dtc = time.perf_counter() - t0
                    finally:
                        os.dup2(fd, 2); os.close(devnull); os.close(fd)
                    idle = 0.14 if use_ref else 0.0         # the driver's wait for the worker to go idle after the queue ran dry
                    nf = (len(sample) - 2) / 1470
                    audio[name]["cpu_baseline"] = {"value": nf / max(dtc - idle, 1e-6), "unit": "frames/s", "cores": 1, "kind": "reference" if use_ref else "port",
                                                   "sample": f"the first {nf:.0f} frames of the tape, {dtc - idle:.2f} s of CPU work",
                                                   "bit_exact_vs_gpu_on_overlap": bool(cr[0][:len(first)].view(np.uint8).tobytes() == first.tobytes())}
        if pal is not None and not args.no_cpu:
            # configs[2] on the CPU: the real reference's VideoToDigital worker and STC007DataStitcher (or the oracle ports) over the start of each PAL tape
            import libs as _libs
            import stitch_api as _sa
            use_ref = _libs.ref_available()
            for tape_name, (lum_c, recs_g, pairs_g) in pal_keep.items():
                try:
                    fd = os.dup(2); devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 2)       # the workers log to stderr
                    try:
                        t0 = time.perf_counter()
                        if use_ref:
                            import importlib.util
                            spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
                            mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
                            crecs, _cst = mg.run_ref(np.ascontiguousarray(lum_c), args.mode)
                        else:
                            from oracle_run import oracle_binarize
                            crecs, _cst = oracle_binarize(np.ascontiguousarray(lum_c), mode=args.mode)
                        t1 = time.perf_counter()
                        cpairs, _cf = _sa.run_cpu(_libs.load_ref() if use_ref else _libs.load_oracle(), "ref_" if use_ref else "orc_", np.ascontiguousarray(crecs), _sa.default_settings())
                        t2 = time.perf_counter()
                    finally:
                        os.dup2(fd, 2); os.close(devnull); os.close(fd)
                    idle = 0.3 if use_ref else 0.0          # the driver's wait for the stitcher thread to go idle after the queue ran dry
                    kk = min(len(cpairs), len(pairs_g))
                    pal[tape_name]["cpu_baseline"] = {
                        "value": len(lum_c) / max((t2 - t0) - idle, 1e-6), "unit": "frames/s", "cores": 1, "kind": "reference" if use_ref else "port",
                        "binarize_frames_per_s": len(lum_c) / (t1 - t0), "stitch_frames_per_s": len(lum_c) / max((t2 - t1) - idle, 1e-6),
                        "sample": f"the first {len(lum_c)} frames of the tape, {t1 - t0:.1f} s in the VideoToDigital worker + {max((t2 - t1) - idle, 0):.1f} s in the stitcher",
                        "bit_exact_vs_gpu_on_overlap": bool(crecs.tobytes() == recs_g.tobytes() and kk > 0 and
                                                            cpairs[:kk].tobytes() == pairs_g.reshape(-1).view(_sa.PAIR_DTYPE)[:kk].tobytes())}
                except Exception as ex:     # noqa: BLE001 - an extra figure must not take the benchmark line down
                    pal[tape_name]["cpu_baseline"] = {"error": repr(ex)}
        for key, (stage, _b, _f) in fmt_stages.items():
            out[key] = stage
        if not args.no_cpu and world == 1:
            # the frame drivers of PCM-1 / PCM-16x0 on the CPU: the real reference's worker (or the oracle port) on the first eight frames
            import libs as _libs
            for key, (stage, base2, first) in fmt_stages.items():
                api = __import__("pcm1_frames_api" if key.startswith("pcm1_") else "pcm16_frames_api")
                use_ref = _libs.ref_available()
                lib = _libs.load_ref() if use_ref else _libs.load_oracle()
                fd = os.dup(2); devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 2)       # the worker logs to stderr
                try:
                    t0 = time.perf_counter()
                    cr, _cs = api.run_cpu(lib, "ref_" if use_ref else "orc_", base2, args.mode, dict(new_file=True))
                    dtc = time.perf_counter() - t0
                finally:
                    os.dup2(fd, 2); os.close(devnull); os.close(fd)
                stage["cpu_baseline"] = {"value": len(base2) / dtc, "unit": "frames/s", "cores": 1, "kind": "reference" if use_ref else "port",
                                         "sample": f"the first {len(base2)} frames of the tape, {dtc:.2f} s of CPU work",
                                         "bit_exact_vs_gpu_on_overlap": bool(cr.tobytes()[:first.nbytes] == first.tobytes())}
        if not args.no_cpu and world == 1:
            ncpu = min(args.cpu_frames, n)
            sample = luma[:ncpu].cpu().numpy()
            cb, cpu_recs = cpu_baseline(sample, args.mode)
            # the CPU path and the GPU path decode the same stream start: compare what both produced
            k = min(len(first_recs), len(cpu_recs))
            cb["bit_exact_vs_gpu_on_overlap"] = bool(first_recs[:k].tobytes() == cpu_recs[:k].tobytes())
            out["cpu_baseline"] = cb
            try:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(sample[:min(args.cpu_frames_all_cores, ncpu)], args.mode, cb["value"])
            except Exception as ex:     # noqa: BLE001 - an extra figure must not take the benchmark line down
                out["cpu_baseline_all_cores"] = {"error": repr(ex), "cores": os.cpu_count()}
            if stitch is not None:
                # the stitch stage on the CPU: the real STC007DataStitcher on its own thread (or the oracle port of it) over the records of the first 1000 frames
                import libs
                import stitch_api as sa
                nst = min(1000, ncpu)
                srecs = np.ascontiguousarray(cpu_recs[:1 + nst * (H + 3)])
                use_ref_st = libs.ref_available()
                fd = os.dup(2); devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 2)       # the stitcher logs to stderr
                try:
                    t0 = time.perf_counter()
                    cp, _ = sa.run_cpu(libs.load_ref() if use_ref_st else libs.load_oracle(), "ref_" if use_ref_st else "orc_", srecs, sa.default_settings())
                    dts = time.perf_counter() - t0
                finally:
                    os.dup2(fd, 2); os.close(devnull); os.close(fd)
                kk = min(len(cp), len(first_pairs))
                stitch["cpu_baseline"] = {"value": nst / dts, "unit": "frames/s", "cores": 1, "kind": "reference" if use_ref_st else "port",
                                          "sample": f"records of the first {nst} frames, {dts:.1f} s of CPU work" + (" (incl. ~0.3 s the driver waits for the stitcher thread to go idle)" if use_ref_st else ""),
                                          "bit_exact_vs_gpu_on_overlap": bool(cp[:kk].tobytes() == first_pairs.reshape(-1).view(sa.PAIR_DTYPE)[:kk].tobytes())}
            if pcm1 is not None:
                import pcm1_api as p1a
                np1 = min(3000, n)
                end = int(np.nonzero(p1_recs["service_type"] == 5)[0][np1 - 1])
                use_ref_p1 = libs.ref_available()
                fd = os.dup(2); devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 2)
                try:
                    t0 = time.perf_counter()
                    cp1, _ = p1a.run_cpu(libs.load_ref() if use_ref_p1 else libs.load_oracle(), "ref_" if use_ref_p1 else "orc_", p1_recs[:end + 1], p1a.default_settings())
                    dt1 = time.perf_counter() - t0
                finally:
                    os.dup2(fd, 2); os.close(devnull); os.close(fd)
                pcm1["cpu_baseline"] = {"value": np1 / dt1, "unit": "frames/s", "cores": 1, "kind": "reference" if use_ref_p1 else "port",
                                        "sample": f"the first {np1} frames, {dt1:.2f} s of CPU work" + (" (incl. ~0.3 s the driver waits for the stitcher thread to go idle)" if use_ref_p1 else ""),
                                        "bit_exact_vs_gpu_on_overlap": bool(cp1.tobytes() == p1_first.reshape(-1).view(p1a.PAIR_DTYPE)[:len(cp1)].tobytes())}
            if pcm16 is not None:
                # the PCM-16x0 back half on the CPU: the real reference's stitcher thread (or the oracle port) on the first 200 frames
                import pcm16_api as p16a
                use_ref = libs.ref_available()
                lib16 = libs.load_ref() if use_ref else libs.load_oracle()
                for fmt, (recs16, got16) in p16_keep.items():
                    n16 = min(200, n)
                    end = int(np.nonzero(recs16["service_type"] == 5)[0][n16 - 1])
                    st = p16a.default_settings(format=2 if fmt == "ei" else 1)
                    fd = os.dup(2); devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 2)       # the stitcher logs to stderr
                    try:
                        t0 = time.perf_counter()
                        cp16, _ = p16a.run_cpu(lib16, "ref_" if use_ref else "orc_", recs16[:end + 1], st)
                        dt16 = time.perf_counter() - t0
                    finally:
                        os.dup2(fd, 2); os.close(devnull); os.close(fd)
                    pcm16[fmt]["cpu_baseline"] = {"value": n16 / dt16, "unit": "frames/s", "cores": 1, "kind": "reference" if use_ref else "port",
                                                  "sample": f"the first {n16} frames, {dt16:.2f} s of CPU work" + (" (incl. ~0.3 s the driver waits for the stitcher thread to go idle)" if use_ref else ""),
                                                  "bit_exact_vs_gpu_on_overlap": bool(cp16.tobytes() == got16.reshape(-1).view(p16a.PAIR_DTYPE)[:len(cp16)].tobytes())}
            if pcm1f is not None:
                import pcm1_front_api as pfa
                wl, wst, wgot, cl, cgot = pcm1f_sample
                orc = libs.load_oracle()
                t0 = time.perf_counter()
                cw = pfa.run_lines_with_states(orc, "orc_bin1_", wl, wst.view(pfa.STATE_DTYPE), mode=args.mode)
                dtw = time.perf_counter() - t0
                t0 = time.perf_counter()
                cold_st = np.zeros(len(cl), dtype=pfa.STATE_DTYPE); cold_st["start"], cold_st["stop"] = -32768, 32767
                cc = pfa.run_lines_with_states(orc, "orc_bin1_", cl, cold_st, mode=args.mode)
                dtc = time.perf_counter() - t0
                pcm1f["cpu_baseline"] = {"value": len(wl) / dtw, "unit": "lines/s", "cores": 1, "kind": "port",
                                         "sample": f"the first {len(wl)} preset lines, {dtw:.2f} s of CPU work (incl. the ctypes call per line); "
                                                   f"cold: {len(cl)} lines in {dtc:.2f} s = {len(cl) / dtc:.0f} lines/s",
                                         "bit_exact_vs_gpu_on_overlap": bool(cw.tobytes() == wgot.view(pfa.BIN1_DTYPE).tobytes()
                                                                             and cc.tobytes() == cgot.view(pfa.BIN1_DTYPE).tobytes())}
        # The driver keeps the tail of what this prints: the line of BASELINE's metric goes LAST and stays short - the headline, its roofline and CPU
        # baseline, and the other legs as plain numbers; everything else (the full objects of every leg) goes to stderr and, where the directory can be
        # made, to gpurun_out/bench_details.json.
        def pick(d, *path):
            for k_ in path:
                if not isinstance(d, dict) or k_ not in d:
                    return None
                d = d[k_]
            return d
        summary = {
            "end_to_end_ms_per_step": pick(out, "end_to_end", "ms_per_step"), "end_to_end_frac": pick(out, "end_to_end", "roofline", "frac"),
            "stitch_ms_per_step": pick(out, "stitch_stage", "ms_per_step"),
            "pal_clean_frames_per_s": pick(out, "pal_stage", "clean", "frames_per_s"), "pal_clean_fused_entry_ms_per_step": pick(out, "pal_stage", "clean", "fused_entry_ms_per_step"),
            "pal_damaged_frames_per_s": pick(out, "pal_stage", "lost_lines_and_flipped_cells", "frames_per_s"),
            "pal_damaged_binarize_ms_per_step": pick(out, "pal_stage", "lost_lines_and_flipped_cells", "binarize_ms_per_step"),
            "pal_damaged_cpu_frames_per_s": pick(out, "pal_stage", "lost_lines_and_flipped_cells", "cpu_baseline", "value"),
            "pal_bit_exact_vs_cpu": [pick(out, "pal_stage", "clean", "cpu_baseline", "bit_exact_vs_gpu_on_overlap"),
                                     pick(out, "pal_stage", "lost_lines_and_flipped_cells", "cpu_baseline", "bit_exact_vs_gpu_on_overlap")],
            "damaged_lost_lines_ms_per_step": pick(out, "damaged_tape", "lost_lines", "ms_per_step"),
            "damaged_window_jumps_ms_per_step": pick(out, "damaged_tape", "window_jumps", "ms_per_step"),
            "damaged_beside_coordinates_ms_per_step": pick(out, "damaged_tape", "beside_coordinates", "ms_per_step"),
            "damaged_lost_lines_in_every_frame_ms_per_step": pick(out, "damaged_tape", "lost_lines_in_every_frame", "ms_per_step"),
            "pcm1_frames_ms_per_step": pick(out, "pcm1_frames_stage", "ms_per_step"), "pcm16x0_frames_ms_per_step": pick(out, "pcm16x0_frames_stage", "ms_per_step"),
            "pcm1_stitch_ms_per_step": pick(out, "pcm1_stage", "ms_per_step"),
            "pcm16x0_si_ms_per_step": pick(out, "pcm16x0_stage", "si", "ms_per_step"), "pcm16x0_ei_ms_per_step": pick(out, "pcm16x0_stage", "ei", "ms_per_step"),
            "audio_worn_tape_ms_per_step": pick(out, "audio_stage", "invalid_word_in_every_window", "ms_per_step"),
            "cpu_all_cores_frames_per_s": pick(out, "cpu_baseline_all_cores", "value"), "cpu_all_cores": pick(out, "cpu_baseline_all_cores", "cores"),
            "configs4_frames": pick(out, "configs4_strong", "frames"), "configs4_ranks": pick(out, "configs4_strong", "ranks"), "configs4_scaling": pick(out, "configs4_strong", "scaling"),
            "configs4_ms": pick(out, "configs4_strong", "ms"), "configs4_frames_per_s": pick(out, "configs4_strong", "frames_per_s"),
            "configs4_frac": pick(out, "configs4_strong", "roofline", "frac"), "configs4_process_group_ranks": pick(out, "configs4_strong", "process_group_ranks"),
            "sharded_full_path_ms": pick(out, "sharded_full_path", "ms"),
            "sharded_one_engine_same_file_ms": pick(out, "sharded_full_path", "one_engine_same_file_ms"),
        }
        short = {k_: out[k_] for k_ in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                                        "config", "roofline", "cpu_baseline") if k_ in out}
        short["roofline"] = {k_: v_ for k_, v_ in out["roofline"].items() if k_ not in ("traffic_unit", "traffic_from_committed_profile")}
        short["summary"] = {k_: v_ for k_, v_ in summary.items() if v_ is not None}
        short["details"] = "the full objects of every leg: stderr of this run, gpurun_out/bench_details.json"
        full = json.dumps(out)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_details.json"), "w") as fh:
                fh.write(full + "\n")
        except OSError:
            pass
        print(full, file=sys.stderr, flush=True)
        print(json.dumps(short), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
