"""N>1 path on CPU: two gloo ranks each fill their owned tile pixels, all_gather equal chunks and
assemble the frame exactly as bench.py does with RCCL on the GPUs."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, w, h, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from massivevoxelraytracing_amd import tiles
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = tiles.global_pixel_index(w, h, rank, world)
    mine = np.zeros((len(g), 4), np.float32)
    ok = g >= 0
    mine[ok, 0] = g[ok]            # a function of the GLOBAL pixel index only, like a path-traced sample
    mine[ok, 1] = (g[ok] * 7) % 13
    mine[ok, 3] = 16.0
    t = torch.from_numpy(mine)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    frame = tiles.assemble(np.stack([o.numpy() for o in out]), w, h)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, frame))


@pytest.mark.parametrize("w,h", [(100, 37), (640, 360)])
def test_two_rank_gather_assembles_full_frame(w, h):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, w, h, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    idx = np.arange(w * h)
    for r in range(2):
        f = res[r]
        assert np.array_equal(f[:, 0], idx.astype(np.float32))
        assert np.array_equal(f[:, 1], ((idx * 7) % 13).astype(np.float32))
        assert (f[:, 3] == 16.0).all()


def test_bench_self_launch_command():
    """`python bench.py --gpus N` with no launcher: bench.py starts `torch.distributed.run --nproc-per-node N bench.py ...` as a child, before it
    imports torch or the library (checked by an assert in self_launch); here only the command is printed (no GPU in the CPU suite)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MVRT_BENCH_PRINT_LAUNCH"] = "1"
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "8", "--warmup", "2"], env=env, timeout=120).decode()
    cmd = json.loads(out.strip().split("\n")[-1])
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "8", "--warmup", "2"]
