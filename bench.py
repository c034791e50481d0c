#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json): wavefront path tracing of a sparse voxel
octree at 1920x1080, 64 spp (4 steps of 16 spp), reported as Mrays/s (primary + secondary rays).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid-res R] [--scene dragon|rtcamp|cave|tunnel]

N > 1 without a launcher (`python bench.py --gpus N`): bench.py starts its own ranks -- `python -m torch.distributed.run --nproc-per-node N
bench.py ...` as a child process, before this process imports torch or the library -- and relays rank 0's line.  Under an explicit
torch.distributed.run (WORLD_SIZE set) it is a rank.

A "step" is one PathTracer::step (reference PathTracer.hpp:150-169): 16 spp for every pixel of the frame.
N > 1 (launched by torch.distributed.run, one rank per GPU): the frame's 256-pixel blocks are dealt
round-robin to the ranks (strong scaling: the total work is fixed), the read-only SVO is rebuilt
identically on every rank, and after the K steps the per-rank accumulation buffers are exchanged with ONE
RCCL all-gather and assembled into the full frame -- all inside the timed region.

Timed region: inputs resident in HBM (scene built, HDRI/PMJ tables uploaded, buffers allocated) before it
starts; K steps (+ gather for N > 1) between barrier + device synchronize on both sides; max over ranks.
The K steps are rendered as 64-spp FRAMES (4 steps each; the metric's unit of work): clear, 4 steps, gather, device synchronise -- so
`value` is the 64-spp figure whatever K the driver asks for (longer runs do not amortise a frame's latency floor).

The JSON line also carries
  roofline     -- traversal kernel (kPtTraceStream): algorithmic bytes per launch / mean launch time, measured in an extra
                  NON-OVERLAPPED pass after the timed region (pipeline depth 1, one step per pass: HIP events on the one stream
                  every kernel runs on, so kernel times cannot overlap), against the 8 TB/s HBM peak;
  cpu_baseline -- the CPU oracle (a port of the reference's voxRT / renderPT arithmetic, oracle/) timed on this
                  host's cores on a bounded band of the same frame (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RANDOM_LINE_CEILING_GBS = 3200.0  # measured on the box: 50 G random 64-byte lines per second (tools/calib/gather_rate.hip)
STREAM_READ_CEILING_GBS = 6100.0  # measured on the box: tools/calib/stream_read.hip, 8-32 GiB buffers (profiles/r02_stream_read.txt); SURVEY.md 8d's second denominator
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak (6.29 TB/s measured copy)


def stress(args, mv):
    """BASELINE.json configs[4]: synthetic non-DAG octree far larger than the 256 MB Infinity Cache, incoherent rays;
    the one configuration where the traversal is bound by HBM (every descent is a dependent 64-byte line fetch)."""
    svo = mv.IntersectorOctreeGPU()
    res, n_vox, n_rays = args.grid_res, int(args.voxels), int(args.rays)
    t0 = time.time()
    svo.build_synthetic(res, n_vox, seed=2024, flags=svo.BUILD_NO_DAG | svo.BUILD_NO_EMBEDDED_MASK)
    mv.synchronize()
    build_s = time.time() - t0
    info = svo.info()
    rng = np.random.default_rng(7)
    # incoherent rays: origins on a sphere around the unit cube, aimed at uniformly random interior points
    d = rng.normal(size=(n_rays, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ro = (0.5 + 1.2 * d).astype(np.float32)
    rd = (rng.random((n_rays, 3), dtype=np.float32) - ro).astype(np.float32)
    dev = [mv.DeviceArray.from_host(np.ascontiguousarray(a)) for a in (ro[:, 0], ro[:, 1], ro[:, 2], rd[:, 0], rd[:, 1], rd[:, 2])]
    t = mv.DeviceArray(n_rays, np.float32)
    nm = mv.DeviceArray(n_rays, np.int32)
    vi = mv.DeviceArray(n_rays, np.uint32)
    de = mv.DeviceArray(n_rays, np.uint32)
    for _ in range(args.warmup):
        svo.intersect_device(n_rays, *dev, None, t, nm, vi, de)
    mv.synchronize()
    tt = time.perf_counter()
    for _ in range(args.steps):
        svo.intersect_device(n_rays, *dev, None, t, nm, vi, de)
    mv.synchronize()
    el = time.perf_counter() - tt
    desc = de.to_host().astype(np.uint64)
    hits = int((t.to_host() != mv.MAX_FLOAT).sum())
    algo = n_rays * 40 + int(desc.sum()) * 8 + hits * 8
    # what the dependent pointer chase really moves: one brick (16 bytes of a 64-byte line) per two descents; sibling bricks share lines, so the
    # lines actually fetched are fewer -- taken from the committed counter passes (fabric read requests per ray) when they are there
    bricks = int(desc.sum()) // 2
    tj_stress, tj_note = committed_traffic("traffic_stress.json")
    reqs = None if tj_stress is None else tj_stress.get("read_requests_per_ray")
    line_bytes = int((reqs * n_rays if reqs else bricks) * 64)
    gbs = algo * args.steps / el / 1e9
    print(json.dumps({
        "metric": "Mrays/sec (incoherent rays, HBM-resident synthetic octree)", "value": round(n_rays * args.steps / el / 1e6, 2), "unit": "Mrays/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "synthetic %d^3 non-DAG SVO, %d random voxels, %d incoherent rays per step (trace + vIndex resolve)" % (res, n_vox, n_rays),
                   "voxels": int(info.numberOfVoxels), "nodes": int(info.numberOfNodes), "reference_node_gb": round(info.numberOfNodes * 68 / 1e9, 2),
                   "resident_structure_gb": round(svo.traversal_bytes() / 1e9, 2),
                   "embedded_mask": int(info.embeddedMask), "svo_build_s": round(build_s, 2), "hits": hits, "descents_per_ray": round(float(desc.mean()), 2)},
        "roofline": {"bound": "hbm", "kernel": "kTraceBatchStream<2> (tree flavour: two-level bricks)", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                     "measured_stream_read_gbs": STREAM_READ_CEILING_GBS,
                     "traffic": stress_traffic(n_rays), "traffic_source": tj_note,
                     "algorithmic_bytes_per_launch": algo, "line_bytes_per_launch": line_bytes,
                     "line_bytes_from": "committed fabric read requests per ray" if reqs else "bricks entered x 64 B (no counter passes of this build)",
                     "line_gbs": round(line_bytes * args.steps / el / 1e9, 1), "random_line_ceiling_gbs": RANDOM_LINE_CEILING_GBS,
                     "line_frac_of_ceiling": round(line_bytes * args.steps / el / 1e9 / RANDOM_LINE_CEILING_GBS, 3),
                     "bricks_entered_per_launch": bricks,
                     "note": "achieved counts 4-8 useful bytes per descent; line_gbs counts the 64-byte fabric reads of the committed counter passes "
                             "(profiles/traffic_stress.json; one per brick entered -- two descents -- without them); "
                             "random_line_ceiling_gbs = what the chip serves for divergent 64-byte-line gathers at 128 GiB footprint "
                             "(tools/calib/gather_rate.hip, profiles/r01_gfx950_issue_and_gather_costs.txt)"},
    }), flush=True)


def committed_traffic(name):
    """profiles/<name>: fabric-side bytes per ray from separate rocprofv3 --pmc passes (HBM-side bytes cannot be read live).  Only valid for the library
    it was measured with: the file records the source digest of that build, and a different digest means the kernel has changed since -> None."""
    p = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(p):
        return None, "no committed counter passes"
    tj = json.load(open(p))
    from massivevoxelraytracing_amd import build as B
    if tj.get("source_digest") != B.source_digest():
        return None, "profiles/%s was measured with another build of the library (source digest differs): stale, not reported" % name
    return tj, tj["_source"]


def stress_traffic(n_rays):
    """fabric-side bytes per launch of the config-5 traversal kernel, from the committed PMC passes of THIS build (else None)"""
    tj, _ = committed_traffic("traffic_stress.json")
    return None if tj is None else int(tj["traffic_bytes_per_ray"] * n_rays)


def cpu_primary(args):
    """BASELINE.json configs[0]: scenes/bunny.obj voxelized at 256^3, primary rays through the pixel centres on the CPU -- what voxRT.cpp:307-358 times
    (per-row CameraPinhole::shoot -> IntersectorOctree::intersect; serial, or ParallelFor over rows).  No GPU: the CPU oracle (oracle/, the port of
    that loop) IS the thing measured here, as BASELINE.md section 3 asks -- 1 thread (the reference's default, and what BASELINE.md section 2 measured
    on the survey container: 11.0 Mrays/s) and every core of this job."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import bunny_tris, probe_camera
    from oracle import oracle as O
    res = args.grid_res or 256
    sc = O.build_scene_from_triangles(bunny_tris(), res)
    cam = probe_camera(sc.origin, sc.dps, res)
    W, H = args.width, args.height
    cores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), args.cpu_threads)
    out = {}
    for label, th in (("1", 1), ("all", cores)):
        sc.render_primary(cam, W, H, threads=th)  # warm-up (page in the octree)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r = sc.render_primary(cam, W, H, threads=th)
        out[label] = W * H * args.steps / (time.perf_counter() - t0) / 1e6
    hits = int((r["t"] != O.MAX_FLOAT).sum())
    print(json.dumps({
        "metric": "Mrays/sec (primary) at %dx%d, CPU IntersectorOctree path" % (W, H), "value": round(out["all"], 3), "unit": "Mrays/s", "n_gpus": 0, "steps": args.steps,
        "warmup": 1, "ms_per_step": round(W * H / out["all"] / 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "bunny.obj (tests/golden/bunny_tris.f32)",
        "config": {"workload": "bunny.obj %d^3 SVO (%d voxels, %d DAG nodes), primary rays through pixel centres, CPU oracle = port of voxRT.cpp:307-358 (plumbing, no GPU)" % (res, len(sc.morton), len(sc.nodes)),
                   "hits": hits, "descents_per_ray": round(float(r["descents"].mean()), 2)},
        "cpu_baseline": {"value": round(out["all"], 3), "value_1_thread": round(out["1"], 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                         "sample": "%d full frames per figure" % args.steps,
                         "survey_container_1_thread": {256: 11.0, 1024: 8.9, 2048: 7.2}.get(res)},
        "roofline": None}), flush=True)


def self_launch_cmd(n, argv, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
            os.path.abspath(__file__)] + list(argv)


def self_launch(n):
    """`python3 bench.py --gpus N` without a launcher: one rank per GPU under torch.distributed.run on a free local port.  Nothing in THIS process
    has initialised the GPU (no torch, no library import yet), and the ranks are started with subprocess -- never an exec."""
    import socket
    import subprocess
    assert "torch" not in sys.modules and "massivevoxelraytracing_amd" not in sys.modules, "self_launch must run before anything touches the GPU"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this image
    cmd = self_launch_cmd(n, sys.argv[1:], port)
    if os.environ.get("MVRT_BENCH_PRINT_LAUNCH") == "1":  # (tests/test_multirank_cpu.py: the command, without running it)
        print(json.dumps(cmd))
        return
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for l in p.stdout.splitlines():
        if l.startswith("{") and '"metric"' in l:
            line = l
    if line is None:
        sys.stdout.write(p.stdout)
        sys.exit(p.returncode or 1)
    print(line, flush=True)
    sys.exit(p.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--build-flags", type=int, default=0, help="octree build flags (MVRT_BUILD_*); 0 = the reference's DAG with embedded masks")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=4, help="untimed steps first (default: one 64-spp frame, which also tells the library the frame length -- it merges at most half a frame into one pass)")
    ap.add_argument("--grid-res", type=int, default=0, help="0 = the scene's BASELINE size (dragon 2048, rtcamp 4096, cave 2048, tunnel 4096)")
    ap.add_argument("--scene", default="dragon", choices=["dragon", "rtcamp", "cave", "tunnel"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--frame-steps", type=int, default=4, help="steps per frame: 4 = the metric's 64 spp.  The K timed steps are rendered as frames of this many steps, the "
                    "frame buffer cleared and the device synchronised between frames, so `value` is the 64-spp figure whatever K is (0 = one frame of K steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hints", action="store_true", help="A/B: every ray starts at the root (PathTracer.set_origin_hints(False)); results are identical")
    ap.add_argument("--no-serial-pass", action="store_true", help="skip the extra non-overlapped pass the roofline numbers come from")
    ap.add_argument("--serial-only", action="store_true", help="run ONLY the non-overlapped pass (pipeline depth 1, batch 1): the command profiles/ *_serial_kernel_stats.csv is taken from")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--voxels", type=float, default=1.05e9, help="stress mode: random voxels of the synthetic octree")
    ap.add_argument("--rays", type=float, default=1.6e7, help="stress mode: incoherent rays per step")
    ap.add_argument("--mode", default="pt", choices=["pt", "primary", "stress", "cpu-primary"],
                    help="pt = wavefront path tracer (headline); primary = the render kernel of voxRTGPU (config 2); stress = config 5; cpu-primary = config 1 (CPU only)")
    ap.add_argument("--emulate-tiles", type=int, default=0, help="diagnostic: render only tile 0 of N on one GPU (predicts per-rank time of an N-GPU run)")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the frame the CPU baseline renders (0 = auto)")
    args = ap.parse_args()
    if args.mode == "cpu-primary":
        return cpu_primary(args)
    if args.grid_res == 0:
        args.grid_res = {"dragon": 2048, "rtcamp": 4096, "cave": 2048, "tunnel": 4096}[args.scene] if args.mode != "stress" else 8192

    force_dist = os.environ.get("MVRT_FORCE_DIST") == "1"
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or force_dist) and args.mode == "pt":
        # plain `python3 bench.py --gpus N`: start the N ranks ourselves -- as a CHILD process, before this process has imported torch or the
        # library or touched the GPU -- and relay rank 0's JSON line and the exit code (the explicit torch.distributed.run form keeps working)
        return self_launch(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    dist = None
    torch = None
    if world > 1 or force_dist:
        import torch  # noqa: F811  (device memory for the collective + torch.distributed over RCCL)
        import torch.distributed as dist  # noqa: F811
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    import massivevoxelraytracing_amd as mv
    from massivevoxelraytracing_amd import scenes
    mv.lib()
    mv.set_device(local_rank)

    def barrier_sync():
        mv.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if args.mode == "stress":
        return stress(args, mv)

    # ---- scene + renderer (outside the timed region) ----
    t_setup = time.time()
    verts, cols, emis = scenes.SCENES[args.scene](args.detail)
    origin, dps = scenes.bounding_grid(verts, args.grid_res)
    W, H = args.width, args.height
    pt = mv.PathTracer()
    pt.setup(None)
    pt.set_tile(rank, world) if not args.emulate_tiles else pt.set_tile(0, args.emulate_tiles)
    pt.resizeFrameBufferIfNeeded(None, W, H)
    if args.no_hints:
        pt.set_origin_hints(False)
    hdr = os.path.join(ROOT, "tests", "golden", "monks_forest_s.hdr")
    pt.loadHDRI(None, hdr, hdr)
    t_build = time.time()
    if args.build_flags:  # experiment knob: the other octree flavours on the same scene (1 no DAG, 2 plain indices, 3 two-level bricks, 4 conservative voxelization)
        pt.m_intersectorOctreeGPU.build(verts, cols, emis, None, origin, dps, args.grid_res, flags=args.build_flags)
    else:
        pt.updateScene(verts, cols, emis, None, origin, dps, args.grid_res)
    mv.synchronize()
    build_s = time.time() - t_build
    info = pt.m_intersectorOctreeGPU.info()
    lo, hi = np.array(info.lower[:]), np.array(info.upper[:])
    centre = (lo + hi) / 2
    if args.scene == "cave":
        cam = scenes.cave_camera(lo, hi)
    elif args.scene == "tunnel":
        cam = scenes.tunnel_camera(lo, hi)
    else:
        eye = centre + (np.array([2.6, 1.5, 3.1]) if args.scene == "dragon" else np.array([4.2, 2.2, 5.0]))
        cam = scenes.look_at_camera(eye, centre, 40.0, float(np.linalg.norm(eye - centre)), 0.02)
    setup_s = time.time() - t_setup

    if args.mode == "primary":
        # BASELINE.json configs[1]: primary-ray cast (the `render` kernel, voxKernel.cu:437-483) through pixel centres
        svo = pt.m_intersectorOctreeGPU
        rgba_dev = mv.DeviceArray((W * H, 4), np.uint8)
        for _ in range(args.warmup):
            svo.render_device(cam, W, H, True, rgba_dev)
        barrier_sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            svo.render_device(cam, W, H, True, rgba_dev)
        barrier_sync()
        el = time.perf_counter() - t0
        print(json.dumps({"metric": "Mrays/sec (primary) at %dx%d" % (W, H), "value": round(W * H * args.steps / el / 1e6, 2), "unit": "Mrays/s", "n_gpus": 1, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": "%s stand-in %d^3 SVO, primary-ray cast with voxel colours (render kernel), %dx%d" % (args.scene, args.grid_res, W, H),
                                                          "voxels": int(info.numberOfVoxels), "dag_nodes": int(info.numberOfNodes)}}), flush=True)
        return
    owned = pt.owned_pixels()
    gather_in = gather_out = frame = None
    if dist is not None:
        gather_in = torch.empty(owned * 4, dtype=torch.float32, device="cuda")
        gather_out = torch.empty(world * owned * 4, dtype=torch.float32, device="cuda")
        frame = torch.empty(W * H * 4, dtype=torch.float32, device="cuda")

    frame_steps = args.frame_steps if args.frame_steps > 0 else max(args.steps, 1)
    # N > 1: everything of a frame is issued on torch's CURRENT stream (its handle is passed to the library explicitly), which is also the stream
    # the collective is ordered against -- no reliance on the legacy null stream's implicit ordering
    ts = torch.cuda.current_stream().cuda_stream if dist is not None else None

    def run_steps(k, remainder_first=False):
        """k steps as frames of `frame_steps` steps (64 spp): per frame clear -> steps -> (N > 1: one RCCL all-gather of the per-rank accumulation
        buffers + assemble) -> the frame is complete on the device.  Everything of a frame is stream-ordered (step() is deferred / pipelined
        inside the library; join() makes the stream wait for it; N > 1: the d2d copy and the assembly are issued on torch's current stream,
        which the collective is ordered against), so the only host synchronisation is the one that ends the frame."""
        done = 0
        while done < k:
            n = min(frame_steps, k - done)
            if remainder_first and done == 0 and k % frame_steps:
                n = k % frame_steps  # (warmup: end on a WHOLE frame -- the library sizes its passes by the length of the caller's last frame)
            pt.clearFrameBuffer(ts)
            for _ in range(n):
                pt.step(ts, cam)
            pt.join(ts)
            if dist is not None:
                mv.memcpy_d2d(gather_in, pt.framebuffer_dev(), owned * 16, ts)
                dist.all_gather_into_tensor(gather_out, gather_in)
                mv.assemble_tiles(gather_out, world, owned, W, H, frame, ts)
            mv.synchronize()  # the 64-spp frame is finished: an application would read / resolve it now
            done += n

    def timed(k):
        pt.reset_stats()
        barrier_sync()
        t0 = time.perf_counter()
        run_steps(k)
        barrier_sync()
        return time.perf_counter() - t0, pt.stats()

    # ---- warmup, then the timed region: EXACTLY K steps (pipelined / batched as the library does by default) ----
    elapsed, st = None, None
    if not args.serial_only:
        run_steps(args.warmup, remainder_first=True)
        elapsed, st = timed(args.steps)

    # ---- the non-overlapped pass the roofline comes from: pipeline depth 1, one step per pass, no sibling passes, HIP events around every
    # kernel on the one stream everything runs on.  Kernel times cannot overlap, so sum(kernel ms) <= wall time of this pass. ----
    serial = None
    if (not args.no_serial_pass and not args.emulate_tiles) or args.serial_only:  # (N > 1: every rank runs it on its tile share; rank 0 reports its own)
        pt.set_pipeline_depth(1)
        pt.set_batch_steps(1)
        pt.set_split_small_passes(False)
        run_steps(args.warmup if args.serial_only else 1)
        pt.set_profiling(True)
        k_serial = args.steps if args.serial_only else min(args.steps, 4)
        s_el, s_st = timed(k_serial)
        pt.set_profiling(False)
        serial = (s_el, s_st, k_serial)
        if args.serial_only:
            elapsed, st = s_el, s_st
        else:  # back to the defaults for the CPU-baseline comparison below
            pt.set_pipeline_depth(3)
            pt.set_batch_steps(0)
            pt.set_split_small_passes(True)

    rays = float(st["rays"])
    gather_ok = None
    if dist is not None:
        # outside the timed region: every rank looks its own pixels up in the assembled frame of the last gather (bit for bit)
        from massivevoxelraytracing_amd import tiles
        mv.synchronize()
        full = frame.cpu().numpy().reshape(-1, 4)
        mine = pt.read_framebuffer()
        g = tiles.global_pixel_index(W, H, rank, world)
        ok = g >= 0
        flag = torch.tensor([1.0 if np.array_equal(full[g[ok]], mine[ok]) else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_ok = bool(flag[0] > 0)
        t = torch.tensor([elapsed, rays], dtype=torch.float64, device="cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays = float(t[1])

    # ---- roofline of the traversal kernel (DESIGN.md "Algorithmic bytes") ----
    # B_ray = 28 (ray in) + 12 (hit out) + D * (4 + 4 [non-shadow]) + 8 [non-shadow hit]   
    roofline = None
    if serial is not None and serial[1]["traceKernelMs"] > 0:
        s_el, ss, k_serial = serial
        algo_bytes = ss["rays"] * 40 + ss["descents"] * 8 + ss["shadowDescents"] * 4 + ss["hits"] * 8
        launches = max(int(ss["traceLaunches"]), 1)
        trace_ms = ss["traceKernelMs"]
        achieved = algo_bytes / (trace_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        # HBM-side bytes cannot be read live: separate rocprofv3 --pmc passes of `bench.py --serial-only` (tools/final_profiles.sh), valid for the build they were taken with
        tj, traffic_src = committed_traffic("traffic_latest.json")
        if tj is not None and tj.get("scene") == args.scene and tj.get("grid_res") == args.grid_res and (W, H) == (1920, 1080):
            traffic = int(tj["traffic_bytes_per_ray"] * ss["rays"] / launches)
        elif tj is not None:
            traffic_src = "profiles/traffic_latest.json is for the default workload"
        roofline = {
            "bound": "hbm", "kernel": "kPtTraceStream", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
            "measured_stream_read_gbs": STREAM_READ_CEILING_GBS, "frac_of_measured_stream_read": round(achieved / STREAM_READ_CEILING_GBS, 5),
            "measured_in": "extra non-overlapped pass after the timed region: pipeline depth 1, 1 step per pass, %d steps, HIP events around every kernel on the launch stream%s"
                           % (k_serial, "" if world == 1 else "; rank 0's tile share (1/%d of the frame)" % world),
            "algorithmic_bytes_per_launch": int(algo_bytes / launches), "avg_launch_ms": round(trace_ms / launches, 4), "launches": launches,
            "serial_pass_wall_ms": round(s_el * 1e3, 3), "sum_kernel_ms": round(ss["totalKernelMs"], 3),
            "bytes_per_ray": round(algo_bytes / max(ss["rays"], 1), 2),
            "descents_per_ray": round((ss["descents"] + ss["shadowDescents"]) / max(ss["rays"], 1), 2),
            "trace_share_of_kernel_time": round(trace_ms / max(ss["totalKernelMs"], 1e-9), 3),
            "shade_share_of_kernel_time": round(ss["shadeKernelMs"] / max(ss["totalKernelMs"], 1e-9), 3),
            "trace_kernel_mrays_per_s": round(ss["rays"] / (trace_ms * 1e-3) / 1e6, 1),
            "serial_job_mrays_per_s": round(ss["rays"] / s_el / 1e6, 1),
        }

    # ---- CPU baseline (rank 0, single GPU runs only): the oracle on a bounded band of the same frame ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.emulate_tiles:
        from oracle import oracle as O  # the checker / baseline -- never part of the measured GPU path
        nodes, attrs, _ = pt.m_intersectorOctreeGPU.download()
        sc = O.Scene(nodes.view(O.NODE_DTYPE), attrs, origin, dps, args.grid_res, info.hasEmission)
        rgba, hw, hh = O.decode_rgbe(open(hdr, "rb").read())
        Hh = O.HDRI(rgba, hw, hh, rgba, hw, hh, math_mode=1)
        # worker pool sized to this job's CPU share: a 1-GPU box grants 16 cores however many the host shows
        cores = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), args.cpu_threads)
        rows = args.cpu_rows or (H if st["rays"] / max(st["samples"], 1) < 6 else H // 4)  # ~10-30 s of CPU work on 16 cores
        y0 = (H - rows) // 2
        p0, p1 = y0 * W, (y0 + rows) * W
        fb_cpu = np.zeros((W * H, 4), np.float32)
        tc = time.perf_counter()
        _, _, cnt = sc.render_pt(Hh, cam, W, H, 0, math_mode=1, fb=fb_cpu, pixel_begin=p0, pixel_end=p1, threads=cores)
        cpu_s = time.perf_counter() - tc
        # ... and the 1-thread figure (BASELINE.md section 3 / SURVEY.md 8d ask for it beside the all-cores one): 1 / (2 * cores) of the band's rows
        rows1 = max(rows // (2 * cores), 1)
        q0 = (y0 + (rows - rows1) // 2) * W
        t1 = time.perf_counter()
        _, _, cnt1 = sc.render_pt(Hh, cam, W, H, 0, math_mode=1, fb=np.zeros((W * H, 4), np.float32), pixel_begin=q0, pixel_end=q0 + rows1 * W, threads=1)
        cpu1_s = time.perf_counter() - t1
        # the same band from the GPU frame buffer (iteration 0 only) must agree bit for bit
        pt.clearFrameBuffer(None)
        pt.step(None, cam)
        gpu_band = pt.read_framebuffer()[p0:p1]
        cpu = {
            "value": round(cnt["rays"] / cpu_s / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "value_1_thread": round(cnt1["rays"] / cpu1_s / 1e6, 3), "sample_1_thread": "%d rows in the middle of that band: %d rays in %.1f s" % (rows1, cnt1["rays"], cpu1_s),
            "sample": "rows %d-%d of the %dx%d frame, iteration 0 (16 spp): %d samples, %d rays in %.1f s" % (y0, y0 + rows - 1, W, H, cnt["samples"], cnt["rays"], cpu_s),
            "gpu_band_bit_exact": bool(np.array_equal(gpu_band, fb_cpu[p0:p1])),
        }

    if rank == 0:
        spp = 16 * frame_steps
        names = {"dragon": "xyzrgb_dragon stand-in", "rtcamp": "rtcamp9 stand-in", "cave": "closed cave (rtcamp9-class occlusion)",
                 "tunnel": "closed tunnel (sized like the reference's slide-67 scene: 4096^3, ~41 M voxels)"}
        out = {
            "metric": "Mrays/sec (primary+secondary) at %dx%d, %d spp" % (W, H, spp),
            "value": round(rays / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "n_ranks_seen": dist.get_world_size() if dist is not None else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s %d^3 SVO (procedural, pinned by tests/test_scenes.py), wavefront path trace %dx%d, %d steps of 16 spp rendered as %d-spp frames "
                            "(frame buffer cleared + device synchronised per frame), 8 bounces + IBL shadow rays%s" %
                            (names[args.scene], args.grid_res, W, H, args.steps, spp, "; SERIAL MODE (pipeline depth 1, batch 1)" if args.serial_only else ""),
                "voxels": int(info.numberOfVoxels), "dag_nodes": int(info.numberOfNodes), "octree_mb": round(info.numberOfNodes * 64 / 1e6, 1),
                "triangles": int(len(verts) // 3), "svo_build_s": round(build_s, 3), "setup_s": round(setup_s, 1),
                "parallelism": ("tile-split x%d (256-px blocks round-robin) + 1 RCCL all-gather per frame" % world) if dist is not None else
                               ("1 GPU, tile 0 of %d (emulated share)" % args.emulate_tiles if args.emulate_tiles else "1 GPU"),
                "rays_per_sample": round(rays / max(st["samples"], 1), 3) if world == 1 else None,
                "gather_ok": gather_ok,
                "device": mv.device_name(),
            },
            "rays": int(rays),
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
