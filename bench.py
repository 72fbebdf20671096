#!/usr/bin/env python3
"""bench.py -- kernel-level throughput of the block-kernel hot path on MI355X, through the kvz_hip C ABI.

Metric (BASELINE.json): Mblocks/s (SAD8 / SATD8 / DCT32) per GPU.

Headline leg (`value`; BASELINE.json configs[1] + the DCT32 of the metric): the 1080p CTU grid -- per frame 32 400 8x8
luma block pairs for sad_8x8 and for satd_8x8 and 1 980 full 32x32 residual blocks for dct_32x32 -- batched over
FRAMES frames per launch so that every operand array (>= 0.5 GB) is larger than the 256 MiB Infinity Cache: the
kernels stream from HBM.  A "step" = one launch each of kvz_hip_sad_nxn_batch(8), kvz_hip_satd_nxn_batch(8),
kvz_hip_transform_batch(DCT, 32), inputs resident in HBM.  With N ranks every rank runs its own batch (independent
blocks, no collective): "scaling": "weak".

Shard leg (`shard_4k`; BASELINE.json configs[4]: 3840x2160, CTU rows sharded across the GPUs of one node, border
exchange over RCCL): ONE fixed batch for every N -- strong scaling.  kvazaar_amd/shard.py cuts the 34 CTU rows of a 4K
frame into contiguous shards; rank r owns its rows of EVERY frame of the batch.
  * kernels: sad_8x8 + satd_8x8 + dct_32x32 over the rank's blocks of FRAMES_4K frames (129 600 8x8 pairs and 8 040
    32x32 blocks per frame in total); no data-path collective; checksums of all results, summed over ranks, are the
    same for every N (reported, so the sharded result can be compared with the unsharded one).
  * search: a sequence of 4K frames, frame by frame: the rank's rows of the previous frame's reconstruction are placed
    in its extended reference buffer, the 80 boundary rows are exchanged with the ring neighbours
    (shard.exchange_halo_into: RCCL send / recv, enqueued on the kernel stream, no host sync), then every 8x8, 16x16,
    32x32 and 64x64 PU of the rank's rows is searched by kvz_hip_search_pu_batch (hexbs + fractional search, preset
    medium settings) with mv_constraint = frame-and-tile-margin on the extended buffer, i.e. no vector may read
    beyond the halo.  All inside the timed region.
Both legs: W untimed warmup steps, then exactly K steps between barrier + device sync, MAX over ranks.

Launch: python bench.py [--steps K --warmup W]            (1 GPU)
        python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BLK8_PER_FRAME = 32400          # (1920/8) * (1080/8)
BLK32_PER_FRAME = 1980          # (1920/32) * floor(1080/32)
W4K, H4K = 3840, 2160
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec (MI355X_MICROARCH.md); copy-shaped kernels measure 6.3-6.75 TB/s on this pool
BYTES = {"sad_8x8": 132, "satd_8x8": 132, "dct_32x32": 4096}   # SURVEY 8(d) algorithmic bytes per block
NAMES = ("sad_8x8", "satd_8x8", "dct_32x32")
# the order of the three launches inside a step (A/B knob for measurements of launch-to-launch effects; the work is the same)
STEP_ORDER = tuple(os.environ.get("KVZ_BENCH_STEP_ORDER", "sad_8x8,satd_8x8,dct_32x32").split(","))
assert sorted(STEP_ORDER) == sorted(NAMES)
SEED = 12345
SETTLE_S = float(os.environ.get("KVZ_BENCH_SETTLE_S", "0.03"))     # untimed launches before the warm-up steps (Env.timed)


def kernel_sources_digest():
    """sha1 over the sources of the three headline kernels (sad_nxn_kernel / satd8_kernel: picture.hip + satd_regs.h; dct32_mfma_kernel:
    dct32_mfma.hip + dct32_mfma_core.h + transform_core.h; kvz_hip_internal.h): ties a committed PMC capture (profiles/pmc_traffic.json) to the code whose
    traffic it measured -- an edit elsewhere in csrc/ cannot change what these kernels read and write"""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "kvazaar_amd", "csrc")
    for f in ("dct32_mfma.hip", "dct32_mfma_core.h", "kvz_hip_internal.h", "picture.hip", "satd_regs.h", "transform_core.h"):
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(budget_s=2.0):
    """The reference's strategies -- its best SIMD (avx2) and its generic C, both named by north_star -- or our scalar
    port if the compiled reference is absent, timed on this host's cores on a bounded sample of the same workload:
    4 frames (129 600 8x8 pairs = 16.6 MB, 7 920 32x32 blocks = 32 MB in+out), larger than L2 so the CPU also
    streams.  One harness thread per core.  `value` = the avx2 rate for the block mix of one GPU step."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    frames = 4
    n8, n32 = BLK8_PER_FRAME * frames, BLK32_PER_FRAME * frames
    try:
        cores = min(16, len(os.sched_getaffinity(0)))   # the CPU share of a 1-GPU box is 16 threads
    except AttributeError:
        cores = os.cpu_count() or 1
    g = np.random.default_rng(12345)

    def mix(rates, idx):
        t = BLK8_PER_FRAME / rates["sad_8x8"][idx] + BLK8_PER_FRAME / rates["satd_8x8"][idx] + BLK32_PER_FRAME / rates["dct_32x32"][idx]
        return (2 * BLK8_PER_FRAME + BLK32_PER_FRAME) / t / 1e6

    import ref_lib as R
    if R.available():
        L = R.lib()

        def run_threads(fn):
            out = [0.0] * cores
            def work(i):
                out[i] = fn(i)
            ts = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
            [t.start() for t in ts]; [t.join() for t in ts]
            return out

        bufs = []
        for i in range(cores):
            a = R._aligned(g.integers(0, 256, n8 * 64, dtype=np.uint8))
            b = R._aligned(np.clip(a.astype(np.int16) + g.integers(-8, 9, a.shape), 0, 255).astype(np.uint8))
            x = R._aligned(g.integers(-255, 256, n32 * 1024).astype(np.int16))
            y = R._aligned(np.zeros(n32 * 1024, np.int16))
            bufs.append((a, b, x, y))
        u8p, i16p = C.POINTER(C.c_uint8), C.POINTER(C.c_int16)
        per = {}
        for strategy in ("avx2", "generic"):
            if not R.has_strategy("satd_8x8", strategy):
                continue
            rates = {}
            for name, t in (("sad_8x8", b"sad_8x8"), ("satd_8x8", b"satd_8x8")):
                r = run_threads(lambda i: L.ref_bench_cost_nxn(t, strategy.encode(), 8, bufs[i][0].ctypes.data_as(u8p),
                                                               bufs[i][1].ctypes.data_as(u8p), n8, budget_s, None))
                rates[name] = (sum(r), r[0])
            r = run_threads(lambda i: L.ref_bench_transform(b"dct_32x32", strategy.encode(), 32, bufs[i][2].ctypes.data_as(i16p),
                                                            bufs[i][3].ctypes.data_as(i16p), n32, budget_s))
            rates["dct_32x32"] = (sum(r), r[0])
            per[strategy] = rates
        best = "avx2" if "avx2" in per else "generic"
        out = {
            "value": round(mix(per[best], 0), 3), "unit": "Mblocks/s", "cores": cores, "kind": "reference",
            "sample": "the reference's %s strategy (oracle/_ref), %d thread(s), each %.0f s per function over 4 frames of the 1080p grid "
                      "(129600 8x8 pairs, 7920 32x32 blocks; streaming working set); value = same SAD8+SATD8+DCT32 block mix as a GPU step"
                      % (best, cores, budget_s),
            "per_thread_value_under_load": round(mix(per[best], 1), 3),
        }
        for s, rates in per.items():
            out["%s_Mblocks_s" % s] = {"mix": round(mix(rates, 0), 3), **{k: round(v[0] / 1e6, 3) for k, v in rates.items()}}
        return out
    import oracle_lib as O
    rates = {}
    a = g.integers(0, 256, (400000, 64), dtype=np.uint8)
    b = g.integers(0, 256, (400000, 64), dtype=np.uint8)
    for name, k in (("sad_8x8", "sad"), ("satd_8x8", "satd")):
        t0 = time.time(); O.cost_nxn_many(k, 8, a, b, threads=1); dt = time.time() - t0
        rates[name] = (len(a) / dt, len(a) / dt)
    x = g.integers(-255, 256, (4000, 1024)).astype(np.int16)
    t0 = time.time(); O.transform_many("dct", 32, x, threads=1); dt = time.time() - t0
    rates["dct_32x32"] = (len(x) / dt, len(x) / dt)
    return {"value": round(mix(rates, 0), 3), "unit": "Mblocks/s", "cores": 1, "kind": "port",
            "sample": "oracle (scalar C restatement of generic), 1 thread: 400000 8x8 pairs, 4000 32x32 blocks",
            "generic_Mblocks_s": {"mix": round(mix(rates, 0), 3), **{k: round(v[0] / 1e6, 3) for k, v in rates.items()}}}


class Env:
    """process-wide handles shared by the legs"""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: kvazaar_amd has no CPU path")
        # rehearsal knobs for a 1-GPU box (never set by the driver): ranks share device 0 and rendezvous over gloo,
        # because RCCL refuses two ranks on one device; the halo exchange then bounces through pinned host memory
        self.backend = os.environ.get("KVZ_BENCH_BACKEND", "nccl")
        if os.environ.get("KVZ_BENCH_SHARE_DEVICE") == "1":
            local_rank = 0
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.cuda.set_device(local_rank)
            import datetime
            tmo = datetime.timedelta(seconds=180)       # a failed exchange surfaces as an error within minutes, not as a hung job
            if self.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
            else:
                dist.init_process_group(backend=self.backend, timeout=tmo)
            self.dist = dist
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        self.red_dev = self.dev if self.backend == "nccl" else torch.device("cpu")
        from kvazaar_amd import _lib
        self._lib = _lib
        self.L = _lib.init(local_rank)
        self.stream = self.L.kvz_hip_stream_create()
        self.tstream = torch.cuda.ExternalStream(int(self.stream), device=self.dev)   # torch's view of the same stream
        self.args = args

    def check(self, rc, what):
        self._lib.check(rc, what)

    def sync(self):
        self.check(self.L.kvz_hip_stream_sync(self.stream), "stream_sync")
        self.torch.cuda.synchronize()

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def timed(self, step, steps, warmup):
        """the contract's bracket: W untimed steps, then K steps between barrier + device sync on both sides, MAX over ranks"""
        # Before the contract's warm-up: ~30 ms of the same launches, untimed.  A burst that starts on an idle device runs 5-10 % slower
        # between its 3rd and 7th millisecond (clock / power management settling: profiles/r03_launch_sequence.txt lists every launch of
        # such a run in order) -- with K = 20 steps of 0.37 ms the timed region would sit exactly there, and `value` is a steady-state rate.
        if SETTLE_S > 0:
            t0 = time.perf_counter()
            for _ in range(4):
                step(None)
            self.sync()
            per = max((time.perf_counter() - t0) / 4, 1e-5)
            n = min(2000, int(SETTLE_S / per))
            if self.dist:                                # a step may hold an exchange: every rank runs the same number of them
                t = self.torch.tensor([n], dtype=self.torch.int64, device=self.red_dev)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
                n = int(t.item())
            for _ in range(n):
                step(None)
        for _ in range(warmup):
            step(None)
        self.sync(); self.barrier(); self.sync()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        self.sync(); self.barrier(); self.sync()
        dt = time.perf_counter() - t0
        if self.dist:
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.red_dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def all_sum(self, values):
        """int64 sums over ranks (checksums; outside the timed regions)"""
        if not self.dist:
            return [int(v) for v in values]
        t = self.torch.tensor([int(v) for v in values], dtype=self.torch.int64, device=self.red_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [int(v) for v in t.tolist()]

    def all_gather(self, obj):
        """[obj of rank 0, ...] on every rank (descriptions of the shards; outside the timed regions)"""
        if not self.dist:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def events(self, n):
        return [self.L.kvz_hip_event_create() for _ in range(n)]

    def elapsed(self, a, b):
        f = C.c_float()
        self.check(self.L.kvz_hip_event_elapsed_ms(a, b, C.byref(f)), "event_elapsed")
        return f.value


def kernel_stats(ms_lists, blocks):
    """per-kernel launch statistics from the HIP events recorded inside the timed region"""
    kern = {}
    for name in NAMES:
        v = sorted(ms_lists[name])
        avg = sum(v) / len(v)
        gbs = BYTES[name] * blocks[name] / (avg * 1e-3) / 1e9
        kern[name] = {"Mblocks_s": round(blocks[name] / (avg * 1e-3) / 1e6, 1), "avg_launch_ms": round(avg, 5),
                      "median_launch_ms": round(v[len(v) // 2], 5), "min_launch_ms": round(v[0], 5), "max_launch_ms": round(v[-1], 5),
                      "blocks_per_launch": blocks[name], "algorithmic_bytes_per_block": BYTES[name],
                      "achieved_GBs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
    return kern


def three_kernel_step(env, cur, ref, res, sad, satd, coef, n8, n32):
    L, st = env.L, env.stream
    launches = {
        "sad_8x8": lambda: env.check(L.kvz_hip_sad_nxn_batch(8, cur.data_ptr(), ref.data_ptr(), n8, sad.data_ptr(), st), "sad_8x8"),
        "satd_8x8": lambda: env.check(L.kvz_hip_satd_nxn_batch(8, cur.data_ptr(), ref.data_ptr(), n8, satd.data_ptr(), st), "satd_8x8"),
        "dct_32x32": lambda: env.check(L.kvz_hip_transform_batch(0, 32, res.data_ptr(), coef.data_ptr(), n32, st), "dct_32x32"),
    }
    order = [launches[n] for n in STEP_ORDER]

    def step(ev):
        if ev: L.kvz_hip_event_record(ev[0], st)
        for i, launch in enumerate(order):
            launch()
            if ev: L.kvz_hip_event_record(ev[i + 1], st)
    return step


def collect_event_ms(env, evs):
    ms = {n: [] for n in NAMES}
    for ev in evs:
        for i, name in enumerate(STEP_ORDER):
            ms[name].append(env.elapsed(ev[i], ev[i + 1]))
    return ms


def headline_leg(env, steps, warmup, frames):
    torch, dev = env.torch, env.dev
    n8, n32 = BLK8_PER_FRAME * frames, BLK32_PER_FRAME * frames
    gen = torch.Generator(device=dev); gen.manual_seed(SEED + env.rank)
    cur = torch.randint(0, 256, (n8, 64), dtype=torch.uint8, device=dev, generator=gen)
    noise = torch.randint(-8, 9, (n8, 64), dtype=torch.int16, device=dev, generator=gen)
    ref = (cur.to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8)
    del noise
    res = torch.randint(-255, 256, (n32, 1024), dtype=torch.int16, device=dev, generator=gen)
    sad = torch.empty(n8, dtype=torch.int32, device=dev)
    satd = torch.empty(n8, dtype=torch.int32, device=dev)
    coef = torch.empty_like(res)
    torch.cuda.synchronize()
    evs = [env.events(4) for _ in range(steps)]
    launch = three_kernel_step(env, cur, ref, res, sad, satd, coef, n8, n32)
    dt = env.timed(lambda k: launch(evs[k] if k is not None else None), steps, warmup)
    ms = collect_event_ms(env, evs)
    # light integrity property at full size (parity proper lives in tests/): a second launch gives identical costs
    satd2 = torch.empty_like(satd)
    env.check(env.L.kvz_hip_satd_nxn_batch(8, cur.data_ptr(), ref.data_ptr(), n8, satd2.data_ptr(), env.stream), "satd_8x8")
    env.sync()
    assert bool((satd2 == satd).all()), "non-deterministic satd"
    blocks = {"sad_8x8": n8, "satd_8x8": n8, "dct_32x32": n32}
    return dt, ms, blocks


def shard_kernel_leg(env, sh, steps, warmup, frames, partition):
    """the three block kernels over this rank's share of a fixed batch of 4K frames: a raster span of CTUs (partition "spans":
    shares equal to within one CTU) or whole CTU rows ("rows")"""
    from kvazaar_amd import shard as S
    torch, dev = env.torch, env.dev
    if partition == "spans":
        sp = S.SpanShard(sh.width, sh.height, env.world, env.rank)
        cur, ref = S.block_pairs_of_ctu_span(torch, dev, SEED, sp.ctus(), frames, 8)
        res = S.residual_blocks_of_ctu_span(torch, dev, SEED, sp.ctus(), frames, 32)
        want8, want32, descr = sp.blocks(8) * frames, sp.blocks(32) * frames, sp.describe()
    else:
        cur, ref, res = [], [], []
        for r, h in sh.ctu_row_heights():
            c, f = S.block_pairs_of_ctu_row(torch, dev, SEED, r, h, sh.width, frames, 8)
            cur.append(c); ref.append(f)
            res.append(S.residual_blocks_of_ctu_row(torch, dev, SEED, r, h, sh.width, frames, 32))
        cur, ref, res = torch.cat(cur), torch.cat(ref), torch.cat(res)
        want8, want32, descr = sh.blocks(8) * frames, sh.blocks(32) * frames, sh.describe()
    n8, n32 = cur.shape[0], res.shape[0]
    assert n8 == want8 and n32 == want32
    sad = torch.empty(n8, dtype=torch.int32, device=dev)
    satd = torch.empty(n8, dtype=torch.int32, device=dev)
    coef = torch.empty_like(res)
    torch.cuda.synchronize()
    evs = [env.events(4) for _ in range(steps)]
    launch = three_kernel_step(env, cur, ref, res, sad, satd, coef, n8, n32)
    dt = env.timed(lambda k: launch(evs[k] if k is not None else None), steps, warmup)
    ms = collect_event_ms(env, evs)
    ca, cw = S.coeff_checksum(torch, coef)
    sums = env.all_sum([S.cost_checksum(sad), S.cost_checksum(satd), ca, cw, n8, n32])
    per_rank = env.all_gather([2 * n8 + n32, descr])
    return dt, ms, {"sad_8x8": n8, "satd_8x8": n8, "dct_32x32": n32}, sums, per_rank


def _dbg(msg):
    if os.environ.get("KVZ_BENCH_DEBUG"):
        sys.stderr.write("[rank %s] %s\n" % (os.environ.get("RANK", "0"), msg)); sys.stderr.flush()


def shard_search_leg(env, sh, steps, warmup, frames):
    """frame after frame: reconstruction rows of frame f-1 -> extended reference buffer -> halo exchange -> motion search of
    every PU of the rank's rows of frame f"""
    import numpy as np
    from kvazaar_amd import shard as S
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from patterns import ME_PU, ME_RESULT, me_params          # layouts of kvz_hip_me_pu / _params / _result (plain numpy dtypes)
    torch, dev, L, st = env.torch, env.dev, env.L, env.stream
    W = sh.width
    pics = [S.shard_plane(torch, dev, sh, SEED, f, 0) for f in range(frames)]                      # source, extended layout
    recs = [S.shard_plane(torch, dev, sh, SEED, f, 1, extended=False) for f in range(frames)]      # reconstruction, own rows
    ext_ref = torch.zeros((sh.ext_rows, W), dtype=torch.uint8, device=dev)
    pus_np, spans = S.shard_pus(np, sh, (8, 16, 32, 64), ME_PU)
    # The exchange overlaps with the search: the CTU rows away from the shared edges ("interior", vectors confined to the rank's
    # own rows) are searched while the halo rows travel on a second stream; the CTU rows next to a shared edge ("boundary",
    # own rows + halo) wait for them on the device (kvz_hip_stream_wait_event).  PUs are regrouped so that every launch takes a
    # contiguous run of one size class.
    order, groups = [], []                        # groups: (name, first PU in the regrouped list, count, params)
    for name, idx, tile in S.search_groups(np, sh, pus_np):
        for hint, sizes in ((1, (8, 16)), (2, (32,)), (4, (64,))):
            sel = idx[np.isin(pus_np["width"][idx], sizes)]
            if len(sel):
                # preset medium: hexbs, early termination on, fme_level 4; mv_constraint 4: no vector may make the search or its
                # interpolation read beyond the group's rectangle
                p = me_params(lambda_cost=20, early_termination=1, fme_level=4, mv_constraint=4, tile=tile)
                p["size_classes"] = hint
                groups.append((name, len(order), len(sel), p))
                order += sel.tolist()
    pus_np = pus_np[np.asarray(order, dtype=np.int64)]
    _dbg("search leg: %d PUs, groups %s" % (len(pus_np), [(g[0], g[1], g[2]) for g in groups]))
    pus = torch.from_numpy(pus_np.view(np.uint8).reshape(-1, 64)).to(dev)
    results = torch.zeros((frames, len(pus_np), 8), dtype=torch.int32, device=dev)
    staging = {}
    xstream = L.kvz_hip_stream_create()                                   # the exchange's stream
    xt = torch.cuda.ExternalStream(int(xstream), device=dev)
    ev_rows, ev_halo = L.kvz_hip_event_create(), L.kvz_hip_event_create()
    torch.cuda.synchronize()
    evs = [[env.events(4) for _ in range(frames)] for _ in range(steps)]
    row_bytes = sh.rows * W

    def search(f, which):
        for name, first, count, p in groups:
            if name == which:
                env.check(L.kvz_hip_search_pu_batch(pics[f].data_ptr(), W, W, sh.ext_rows, ext_ref.data_ptr(), W, W, sh.ext_rows,
                                                    pus.data_ptr() + 64 * first, count, p.ctypes.data,
                                                    results[f].data_ptr() + 32 * first, st), "search_pu")

    def step(k):
        for f in range(frames):
            ev = evs[k][f] if k is not None else None
            if ev: L.kvz_hip_event_record(ev[0], st)
            prev = recs[(f - 1) % frames]
            env.check(L.kvz_hip_memcpy_d2d(ext_ref.data_ptr() + sh.top * W, prev.data_ptr(), row_bytes, st), "rec rows")
            if env.dist:
                L.kvz_hip_event_record(ev_rows, st)
                env.check(L.kvz_hip_stream_wait_event(xstream, ev_rows), "wait rows")
                _dbg("frame %d: exchange" % f)
                with torch.cuda.stream(xt):
                    S.exchange_halo_into(ext_ref, sh, env.dist, staging)
                _dbg("frame %d: exchanged" % f)
                L.kvz_hip_event_record(ev_halo, xstream)
            if ev: L.kvz_hip_event_record(ev[1], st)
            search(f, "interior")
            if ev: L.kvz_hip_event_record(ev[2], st)
            if env.dist:                                                  # also keeps the next frame's copy into ext_ref behind this frame's sends
                env.check(L.kvz_hip_stream_wait_event(st, ev_halo), "wait halo")
            search(f, "boundary")
            if ev: L.kvz_hip_event_record(ev[3], st)

    _dbg("search leg: first step")
    dt = env.timed(step, steps, warmup)
    _dbg("search leg: timed done")
    ex_ms, se_ms, bd_ms = [], [], []
    for k in range(steps):
        for f in range(frames):
            ex_ms.append(env.elapsed(evs[k][f][0], evs[k][f][1]))
            se_ms.append(env.elapsed(evs[k][f][1], evs[k][f][2]))
            bd_ms.append(env.elapsed(evs[k][f][2], evs[k][f][3]))
    r = results.cpu().numpy().view(ME_RESULT).reshape(frames, -1)
    found = int((r["cost"] != 0xFFFFFFFF).sum())
    sums = env.all_sum([len(pus_np), found, int(r["mv"].astype(np.int64).sum()), int(r["cost"].astype(np.int64).sum()),
                        int((np.abs(r["mv"][..., 0] - S.NOMINAL_MV[0]) + np.abs(r["mv"][..., 1] - S.NOMINAL_MV[1]) <= 2).sum())])
    n_boundary = sum(c for (name, _, c, _) in groups if name == "boundary")
    # the exchange's stream stays alive until the process ends: torch keeps events (pinned-memory bookkeeping, collectives' work
    # objects) that were recorded on it, and destroying it under them crashed the multi-rank run on exit from this function
    L.kvz_hip_stream_sync(xstream)
    return dt, ex_ms, se_ms, bd_ms, sums, n_boundary


def reference_workload_leg(env, budget_s=0.4):
    """The reference's OWN benchmark workload (tests/speed_tests.c) next to the random batches of the headline: the radial-gradient
    chunk set for sad_8x8 / satd_8x8 / dct_32x32 (:63-92, :116-153, :252-300) and the reg_sad loop on the 4K frame `inter_a` with the
    sparse +-6 vector grid at 8x8, 16x16, 32x32, 64x64, 64x63 and 1x1 (:94-103, :198-236, :400-403) -- the shape the encoder's motion
    search really issues.  GPU: the batched entries on the same data resident in HBM (HIP events); CPU: the reference's best
    registered strategy for each function (oracle/_ref) on one thread and on all the box's threads."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import patterns as P
    torch, dev, L, st = env.torch, env.dev, env.L, env.stream
    try:
        cores = min(16, len(os.sched_getaffinity(0)))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        import ref_lib as R
        have_ref = R.available()
    except Exception:                                  # noqa: BLE001
        have_ref = False
    RL = R.lib() if have_ref else None

    def gpu_rate(fn, calls, iters=10):
        e0, e1 = L.kvz_hip_event_create(), L.kvz_hip_event_create()
        for _ in range(2):
            fn()
        L.kvz_hip_event_record(e0, st)
        for _ in range(iters):
            fn()
        L.kvz_hip_event_record(e1, st)
        ms = env.elapsed(e0, e1)
        L.kvz_hip_event_destroy(e0); L.kvz_hip_event_destroy(e1)
        return calls * iters / (ms * 1e-3)

    def cpu_threads(fn):
        one = fn(0)
        out = [0.0] * cores

        def work(i):
            out[i] = fn(i)
        ts = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
        [t.start() for t in ts]; [t.join() for t in ts]
        return one, sum(out)

    def best_name(type_, order=("avx2", "sse41", "sse2", "generic")):
        for n in order:
            if R.has_strategy(type_, n):
                return n
        return None

    rows = {}
    bufs = P.speed_test_bufs()
    u8p, i16p = C.POINTER(C.c_uint8), C.POINTER(C.c_int16)
    # -- contiguous block kernels on the gradient set, replicated to > 256 MiB per operand so that the GPU streams from HBM
    for kind, n in (("sad", 8), ("satd", 8)):
        b1, b2 = P.speed_test_intra_pairs(bufs, n)
        rep = max(1, (300 << 20) // b1.nbytes)
        d1, d2 = torch.from_numpy(b1).to(dev).repeat(rep, 1), torch.from_numpy(b2).to(dev).repeat(rep, 1)
        cost = torch.empty(d1.shape[0], dtype=torch.int32, device=dev)
        f = L.kvz_hip_sad_nxn_batch if kind == "sad" else L.kvz_hip_satd_nxn_batch
        g = gpu_rate(lambda: env.check(f(n, d1.data_ptr(), d2.data_ptr(), d1.shape[0], cost.data_ptr(), st), kind), d1.shape[0])
        env.sync()
        row = {"gpu_Mcalls_s": round(g / 1e6, 1), "calls_per_launch": int(d1.shape[0]), "pattern": "radial gradients, first chunk of 36 against the other 35"}
        if have_ref:
            t = ("%s_%dx%d" % (kind, n, n)).encode()
            name = best_name(t.decode())
            a1, a2 = R._aligned(b1.ravel()), R._aligned(b2.ravel())
            one, all_ = cpu_threads(lambda i: RL.ref_bench_cost_nxn(t, name.encode(), n, a1.ctypes.data_as(u8p), a2.ctypes.data_as(u8p), b1.shape[0], budget_s, None))
            row.update({"cpu_strategy": name, "cpu_Mcalls_s_1_thread": round(one / 1e6, 2), "cpu_Mcalls_s_%d_threads" % cores: round(all_ / 1e6, 1),
                        "gpu_over_cpu_all_threads": round(g / all_, 1)})
        rows["%s_%dx%d" % (kind, n, n)] = row
        del d1, d2, cost
    res = P.speed_test_dct_residuals(bufs, 32)
    rep = max(1, (300 << 20) // res.nbytes)
    dr = torch.from_numpy(res).to(dev).repeat(rep, 1)
    dc = torch.empty_like(dr)
    g = gpu_rate(lambda: env.check(L.kvz_hip_transform_batch(0, 32, dr.data_ptr(), dc.data_ptr(), dr.shape[0], st), "dct"), dr.shape[0])
    env.sync()
    row = {"gpu_Mcalls_s": round(g / 1e6, 2), "calls_per_launch": int(dr.shape[0]), "pattern": "residuals of the gradient chunks (first chunk - chunk)"}
    if have_ref:
        name = best_name("dct_32x32")
        x = R._aligned(res.ravel())
        ys = [R._aligned(np.zeros(res.size, np.int16)) for _ in range(cores)]
        one, all_ = cpu_threads(lambda i: RL.ref_bench_transform(b"dct_32x32", name.encode(), 32, x.ctypes.data_as(i16p), ys[i].ctypes.data_as(i16p),
                                                               res.shape[0], budget_s))
        row.update({"cpu_strategy": name, "cpu_Mcalls_s_1_thread": round(one / 1e6, 3), "cpu_Mcalls_s_%d_threads" % cores: round(all_ / 1e6, 2),
                    "gpu_over_cpu_all_threads": round(g / all_, 1)})
    rows["dct_32x32"] = row
    del dr, dc
    # -- reg_sad on the 4K frame: descriptors of the reference's loop, 64 sweeps over the 58 x 31 inner LCUs per launch
    frame = P.speed_test_inter_frame(W4K, H4K)
    dframe = torch.from_numpy(frame).to(dev)
    if have_ref:
        RL.ref_bench_speed_inter_sad.restype = C.c_double
        RL.ref_bench_speed_inter_sad.argtypes = [C.c_char_p, C.c_void_p] + [C.c_int] * 4 + [C.c_double, C.POINTER(C.c_ulonglong)]
    for (bw, bh) in ((8, 8), (16, 16), (32, 32), (64, 64), (64, 63), (1, 1)):
        sweeps = 64 if bw * bh <= 1024 else 8
        pairs = P.speed_test_inter_pairs(bw, bh, 58 * 31 * sweeps, W4K, H4K)
        dp = torch.from_numpy(pairs).to(dev)
        cost = torch.empty(pairs.shape[0], dtype=torch.int32, device=dev)
        g = gpu_rate(lambda: env.check(L.kvz_hip_reg_sad_batch(dframe.data_ptr(), W4K, dframe.data_ptr(), W4K, dp.data_ptr(), pairs.shape[0], cost.data_ptr(), st),
                                       "reg_sad"), pairs.shape[0], iters=5)
        env.sync()
        row = {"gpu_Mcalls_s": round(g / 1e6, 1), "calls_per_launch": int(pairs.shape[0]), "gpu_checksum": int(cost.long().sum().item()),
               "pattern": "4K frame inter_a, first CU of every inner LCU x the 25 vectors {-6,-3,0,3,6}^2"}
        if have_ref:
            name = best_name("reg_sad", ("x86_asm_avx", "avx2", "sse41", "generic"))
            one, all_ = cpu_threads(lambda i: RL.ref_bench_speed_inter_sad(name.encode(), frame.ctypes.data, W4K, H4K, bw, bh, budget_s, None))
            row.update({"cpu_strategy": name, "cpu_Mcalls_s_1_thread": round(one / 1e6, 2), "cpu_Mcalls_s_%d_threads" % cores: round(all_ / 1e6, 1),
                        "gpu_over_cpu_all_threads": round(g / all_, 1)})
        rows["reg_sad_%dx%d" % (bw, bh)] = row
        del dp, cost
    return {"what": "the reference's own benchmark workload (tests/speed_tests.c): GPU batched entries vs the reference's best registered CPU strategy "
                    "on the same data; M calls/s like the reference's speed tests report", "cpu_threads": cores, "rows": rows}


def encoder_leg(frames=16, threads=16, timeout_s=280):
    """BASELINE's second half, "encoder fps 1080p medium": the compiled reference encoder (oracle/_ref, Kvazaar built from
    /root/reference by oracle/Makefile) at 1920x1080, preset medium, its own thread pool and --owf auto, timed untouched and with
    its 2Nx2N inter searches answered by the product's search service (kvz_hip_me_service_*) from all its worker threads; the
    bitstreams must be identical.  Runs tools/served_encode.py in a child process (the encoder is a host application of the
    library, not part of it); returns None when oracle/_ref is not there."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libkvzref.so")):
        return None
    cmd = [sys.executable, os.path.join(ROOT, "tools", "served_encode.py"), "--size", "1920x1080", "--frames", str(frames), "--threads", str(threads),
           "--min-size", "8,32,64", "--probe"]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": "tools/served_encode.py did not finish in %d s" % timeout_s}
    rows, probe, summary = [], None, None
    for line in r.stdout.splitlines():
        try:
            d = json.loads(line)
        except ValueError:
            continue
        if "cpu_search_us" in d:
            probe = d
        elif "summary" in d:
            summary = d
        elif "fps_served" in d:
            rows.append(d)
    if not rows or summary is None:
        return {"error": "tools/served_encode.py failed (rc %d)" % r.returncode}
    # the same host where the SAD path is the bulk of its work: --me full16 (exhaustive +-16 around the zero vector, the start vector
    # and the merge candidates, search_inter.c:886-962): every search served (the service's workgroup-wide exhaustive search), and --
    # the candidate-independent half alone -- its kvz_image_calc_sad calls answered from kvz_hip_me_service_sad_tables
    full, full_served = None, None
    try:
        r2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "served_encode.py"), "--size", "1920x1080", "--frames", "6", "--threads", str(threads),
                             "--opts", "preset=medium,qp=32,me=full16", "--min-size", "8", "--tables", "16"],
                            stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=150)
        for line in r2.stdout.splitlines():
            try:
                d = json.loads(line)
            except ValueError:
                continue
            if d.get("mode") == "sad_tables":
                full = {k: d[k] for k in ("opts", "table_range", "fps_untouched", "fps_with_tables", "identical_bitstream", "hit_rate",
                                          "sad_calls_answered_from_tables", "sad_calls_outside_the_range", "table_KB_per_ctu_and_picture", "table_MB")}
            elif "fps_served" in d:
                full_served = {k: d[k] for k in ("opts", "frames", "min_pu_served", "fps_untouched", "fps_served", "identical_bitstream", "searches_served",
                                                 "searches_left_to_cpu", "mean_wait_us", "service_setup_s", "failed")}
                full_served["speedup"] = round(d["fps_served"] / d["fps_untouched"], 3) if d["fps_untouched"] else None
    except subprocess.TimeoutExpired:
        full = {"error": "timed out"}
    # the same with twice as many encoder threads as the leg's cores: a caller that waits for the device naps, another one computes
    full_served_2x = None
    try:
        r3 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "served_encode.py"), "--size", "1920x1080", "--frames", "8", "--threads", str(2 * threads),
                             "--opts", "preset=medium,qp=32,me=full16", "--min-size", "8"],
                            stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=150)
        for line in r3.stdout.splitlines():
            try:
                d = json.loads(line)
            except ValueError:
                continue
            if "fps_served" in d:
                full_served_2x = {k: d[k] for k in ("opts", "frames", "threads", "fps_untouched", "fps_served", "identical_bitstream", "mean_wait_us", "failed")}
                full_served_2x["speedup"] = round(d["fps_served"] / d["fps_untouched"], 3) if d["fps_untouched"] else None
    except subprocess.TimeoutExpired:
        full_served_2x = {"error": "timed out"}
    out = {
        "what": "reference encoder (oracle/_ref) 1920x1080 preset medium qp 32, %d synthetic frames, threads=%d, owf auto: frames/s untouched (avx2 "
                "strategies) and with its 2Nx2N inter searches of at least `min_pu_served` pixels answered by kvz_hip_me_service_search from all "
                "worker threads (all reference pictures of a PU in parallel; resident workgroups take the units from a ring, no launch per request); the "
                "encode alone is timed both ways, creating the service (device planes, page-locked areas) once per session is service_setup_s" % (frames, threads),
        "fps_untouched_same_threads": rows[0]["fps_untouched"],
        "all_bitstreams_identical": bool(summary.get("all_identical")),
        "served": [{k: row[k] for k in ("min_pu_served", "fps_served", "searches_served", "searches_left_to_cpu", "launches", "mean_requests_per_batch",
                                        "mean_units_per_launch", "max_batch_units", "mean_wait_us", "upload_MB", "service_setup_s", "failed")} for row in rows],
        # what a served search has to beat: the reference's own kvz_search_cu_inter per CU size on this host (all reference pictures of the PU)
        "full_search_served": full_served,
        "full_search_served_twice_the_threads": full_served_2x,
        "full_search_with_sad_tables": full,
        "cpu_search_us_per_cu": probe["cpu_search_us"] if probe else None,
        "cpu_searches_per_cu_size": probe["cpu_searches"] if probe else None,
        "note": "a served search costs its caller mean_wait_us; the CPU does the same search in cpu_search_us_per_cu -- at preset medium (hexbs, early "
                "termination) that is 4-60 us, below the device's search + PCIe round trip for every size but 64x64, so the served encode trails the "
                "untouched one there; with the exhaustive search (full_search_served) the CPU needs 117-790 us per search and the served encode "
                "wins; DESIGN.md section 6 has the account",
    }
    return out


SHARD_LEG_LIMIT_S = 300


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=128, help="1080p frames per batch (per GPU), headline leg")
    ap.add_argument("--frames-4k", type=int, default=128, help="4K frames of the fixed batch of the shard leg (whole job)")
    ap.add_argument("--search-frames", type=int, default=8, help="4K frames per step of the sharded search sequence")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shard-leg", action="store_true")
    ap.add_argument("--no-encoder-leg", action="store_true", help="skip the served-encode leg (reference encoder + search service)")
    ap.add_argument("--no-reference-workload", action="store_true", help="skip the leg that runs the reference's own benchmark patterns (tests/speed_tests.c)")
    ap.add_argument("--partition", choices=("spans", "rows"), default="spans",
                    help="shard_4k block-kernel leg: raster spans of CTUs (equal to within one CTU) or whole CTU rows")
    args = ap.parse_args()
    env = Env(args)
    torch, world, rank = env.torch, env.world, env.rank
    F = args.frames

    dt, ms, blocks = headline_leg(env, args.steps, args.warmup, F)
    torch.cuda.empty_cache()
    ref_workload = None
    if world == 1 and not args.no_reference_workload:
        try:
            ref_workload = reference_workload_leg(env)
        except Exception as e:                          # noqa: BLE001 -- an extra leg must not cost the headline line
            ref_workload = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    encoder = None
    if world == 1 and not args.no_encoder_leg:
        try:
            encoder = encoder_leg()
        except Exception as e:                          # noqa: BLE001
            encoder = {"error": "%s: %s" % (type(e).__name__, e)}

    def emit(shard_out):
        """rank 0: the one JSON line (the headline numbers are final before the shard leg starts)"""
        total_blocks = (2 * blocks["sad_8x8"] + blocks["dct_32x32"]) * world * args.steps
        kern = kernel_stats(ms, blocks)
        dom = max(NAMES, key=lambda k: kern[k]["avg_launch_ms"])
        traffic, traffic_src = None, None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")      # written from separate rocprofv3 --pmc passes of this command
        if os.path.exists(tfile) and F == 128:                          # the PMC passes were taken at the default batch size
            try:
                t = json.load(open(tfile))
                cap = t.get("_captured", {})
                if cap.get("kernel_sources_sha1_16") == kernel_sources_digest():
                    traffic = t.get(dom)
                    traffic_src = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (bytes per "
                                   "launch), captured %s at commit %s on these kernel sources" % (cap.get("date"), cap.get("commit")))
                else:
                    traffic_src = "profiles/pmc_traffic.json was captured on other kernel sources (%s): not quoted" % cap.get("commit")
            except Exception:
                traffic = None
        out = {
            "metric": "Mblocks/s (SAD8/SATD8/DCT32) per GPU",
            "value": round(total_blocks / dt / 1e6, 1),
            "unit": "Mblocks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "1080p CTU grid x %d frames per launch per GPU: sad_8x8 + satd_8x8 on %d 8x8 block pairs, "
                                   "dct_32x32 on %d int16 residual blocks, via the kvz_hip C ABI (batched 'hip' strategy entries)"
                                   % (F, blocks["sad_8x8"], blocks["dct_32x32"]),
                       "frames_per_batch": F,
                       "untimed_before_warmup": "%.0f ms of the same launches (a burst on an idle device dips 5-10 %% between its 3rd and 7th "
                                                "millisecond; KVZ_BENCH_SETTLE_S=0 switches it off)" % (SETTLE_S * 1e3),
                       "parallelism": "independent batches per GPU, no data-path collective (value); "
                                                             "CTU-row shards of one fixed 4K batch + RCCL halo exchange (shard_4k)"},
            "kernels": kern,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": kern[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": kern[dom]["frac_of_hbm_peak"], "traffic": traffic, "traffic_source": traffic_src},
        }
        if shard_out is not None:
            out["shard_4k"] = shard_out
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        if ref_workload is not None:
            out["reference_workload"] = ref_workload
        if encoder is not None:
            out["encoder"] = encoder
        print(json.dumps(out), flush=True)

    # The shard leg exchanges halo rows between ranks; should a collective ever hang, the headline line is still printed: a
    # watchdog emits it (with the failure noted) and ends the process with a non-zero exit code.
    watchdog = None
    if not args.no_shard_leg and world > 1:
        import threading

        def give_up():
            sys.stderr.write("rank %d: shard_4k leg exceeded %d s, giving up on it\n" % (rank, SHARD_LEG_LIMIT_S))
            if rank == 0:
                emit({"error": "timed out after %d s" % SHARD_LEG_LIMIT_S, "note": "the shard_4k leg hung; `value` (headline leg) is unaffected"})
            os._exit(3)                 # a hung exchange is a failed run: the line above is for the record, the exit code says so
        watchdog = threading.Timer(SHARD_LEG_LIMIT_S, give_up)
        watchdog.daemon = True
        watchdog.start()

    shard_out = None
    if not args.no_shard_leg:
        from kvazaar_amd import shard as S
        sh = S.RowShard(W4K, H4K, world, rank, S.HALO_ROWS)
        F4, FS = args.frames_4k, args.search_frames
        shard_err = None
        try:
            kdt, kms, kblocks, ksums, kper_rank = shard_kernel_leg(env, sh, args.steps, args.warmup, F4, args.partition)
            torch.cuda.empty_cache()
            s_steps, s_warm = max(1, args.steps // 10), max(1, args.warmup // 10)        # a search step is FS frames, ~8 ms at N = 1
            sdt, ex_ms, se_ms, bd_ms, ssums, n_boundary = shard_search_leg(env, sh, s_steps, s_warm, FS)
            descr = env.all_gather([sh.describe(), sh.rows, n_boundary])
        except Exception as e:              # noqa: BLE001 -- the headline line must still be printed; the failure is reported in it
            import traceback
            shard_err = "%s: %s" % (type(e).__name__, e)
            sys.stderr.write("rank %d: shard_4k leg failed\n%s\n" % (rank, traceback.format_exc()))
        if shard_err is not None:
            shard_out = {"error": shard_err, "note": "the shard_4k leg failed on rank 0; `value` (headline leg) is unaffected"}
        elif rank == 0:
            tot8, tot32 = ksums[4], ksums[5]
            assert tot8 == 129600 * F4 and tot32 == 8040 * F4, (tot8, tot32)
            kern = kernel_stats(kms, kblocks)
            n_pus = ssums[0]
            shard_out = {
                "metric": "Mblocks/s (SAD8/SATD8/DCT32) over ONE fixed batch of 4K frames cut across the ranks (%s)"
                          % ("raster spans of CTUs, equal to within one CTU" if args.partition == "spans" else "whole CTU rows"),
                "value": round((2 * tot8 + tot32) * args.steps / kdt / 1e6, 1), "unit": "Mblocks/s", "scaling": "strong",
                "n_gpus": world, "steps": args.steps, "ms_per_step": round(kdt / args.steps * 1e3, 5),
                "workload": "3840x2160 x %d frames, fixed for every N: %d 8x8 pairs (sad_8x8 + satd_8x8) and %d 32x32 residual blocks "
                            "(dct_32x32) in total; rank r owns its %s of every frame; no data-path collective"
                            % (F4, tot8, tot32, "raster span of the 2040 CTUs (kvazaar_amd/shard.py SpanShard)" if args.partition == "spans"
                               else "CTU rows (kvazaar_amd/shard.py row_range over 34 rows)"),
                "partition": args.partition,
                "share_per_rank": [d for (_, d) in kper_rank],
                # what the partition itself allows: all blocks / the largest rank's blocks (whole CTU rows over 8 ranks: 34 / 5 = 6.8)
                "ideal_speedup": round(S.ideal_speedup([b for (b, _) in kper_rank]), 3),
                "rank0_kernels": kern,
                "checksums_over_all_ranks": {"sum_sad": ksums[0], "sum_satd": ksums[1], "sum_abs_coeff": ksums[2], "sum_weighted_coeff": ksums[3],
                                             "note": "partition-independent: equal for every n_gpus"},
                "search": {
                    "metric": "PU searches/s, whole job (kvz_hip_search_pu_batch: hexbs + fractional, every 8x8/16x16/32x32/64x64 PU of each frame)",
                    "value": round(n_pus * FS * s_steps / sdt / 1e6, 3), "unit": "M PUs/s", "scaling": "strong",
                    "frames_per_s": round(FS * s_steps / sdt, 1), "ms_per_frame": round(sdt / (FS * s_steps) * 1e3, 4),
                    "steps": s_steps, "frames_per_step": FS, "PUs_per_frame": n_pus,
                    "rows_per_rank": [d for (d, _, _) in descr],
                    "ideal_speedup": round(S.ideal_speedup([r for (_, r, _) in descr]), 3),
                    "rank0_ms_per_frame": {"rec_rows_copy_and_exchange_enqueue": round(sum(ex_ms) / len(ex_ms), 4),
                                           "interior_search_while_the_halo_travels": round(sum(se_ms) / len(se_ms), 4),
                                           "wait_for_halo_plus_boundary_search": round(sum(bd_ms) / len(bd_ms), 4)},
                    "boundary_PUs_per_rank": [b for (_, _, b) in descr],
                    "exchange": {"backend": env.backend if world > 1 else None, "halo_rows": S.HALO_ROWS,
                                 "bytes_per_boundary_per_frame_each_way": S.HALO_ROWS * W4K,
                                 "what": "luma rows of the previous frame's reconstruction, isend/irecv between ring neighbours on a second "
                                         "stream inside the timed region; the CTU rows away from the shared edges are searched meanwhile, the "
                                         "two CTU rows next to an edge after the halo has landed (kvz_hip_stream_wait_event)"},
                    "mv_constraint": "4 (frame and tile margin): interior CTU rows confined to the rank's own rows, boundary CTU rows to rows + halo",
                    "results": {"searched": ssums[0] * FS, "found": ssums[1], "within_half_pel_of_true_motion": ssums[4],
                                "sum_mv": ssums[2], "sum_cost": ssums[3],
                                "note": "depends on the partition (the halo bounds the vectors), like tiles in the reference"},
                },
            }

    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        emit(shard_out)
    if env.dist:
        env.dist.barrier()
        env.dist.destroy_process_group()


if __name__ == "__main__":
    main()
