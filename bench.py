#!/usr/bin/env python3
"""bench.py -- train samples/sec + logloss of the FTRL-FFM hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (H2D of the CSR block -> group -> lazy refresh + forward ->
FTRL update) over one block of synthetic libffm rows.  The metric is SURVEY.md 8(d)'s: rows / wall
time of the train loop with the rows parsed and resident in HOST memory and the H2D of every CSR
block INSIDE the timed region -- what the reference times at src/task/ftrl_offline.cpp:46-48.  The
blocks live in page-locked host memory and go through the engine's pipelined host entry points
(ffm_engine_stage_batch + ffm_engine_train_staged on one GPU; + train_forward_staged + all-reduce +
train_update_device on a sharded rank): block t+1 is DMA-ed to HBM and grouped on a side stream
while block t trains (`--host-copy`: through ffm_engine_train_batch_async, which first copies the
caller's pageable arrays into the engine's pinned staging slot).  The same loop over blocks already
resident in HBM is timed afterwards and reported as `resident` (never as `value`).

Workload at N = 1: BASELINE.json's headline configuration itself -- FFM n_fields=39 n_factors=16
n_feats=33M (247 GB of (w,n,z): it fits one MI355X), block = 8192 rows, Zipf(1.1) ids, 64 distinct
blocks.  N > 1: the same 33M-feature tensor is field-pair sharded over the ranks; every rank sees
the whole block and computes / updates the field pairs it owns; one RCCL all-reduce of n_rows
partial logits per step.  `--scaling weak` (default): the block grows with N (8192 * N rows) so
per-GPU pair work is constant; `--scaling strong`: SURVEY 8(d)'s literal C5, 8192 rows per step
globally.  PyTorch is plumbing here (the process group, device buffers of the resident leg).
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_FIELDS, N_FACTORS = 39, 16
MODEL = "FFM"
FEATS_C5 = 33_000_000
# BASELINE.json configs; only c5's single-GPU slice is the bench line, the others are for DESIGN.md
CONFIGS = {
    "c5": dict(model="FFM", fields=39, factors=16, rows=8192, feats=FEATS_C5),
    "c2": dict(model="FFM", fields=8, factors=16, rows=4096, feats=10_000),
    "c3": dict(model="FFM", fields=39, factors=4, rows=8192, feats=1_000_000),
    "c4": dict(model="FM", fields=39, factors=64, rows=8192, feats=10_000_000),
}
PEAK_HBM_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak
SMALL_MAX = 10  # occurrences per block up to which a feature is "few" (csrc/engine_types.h)


def algorithmic_bytes_per_row(nnz, k):
    """every touched state element read once + written once per row, no reuse
    credit: latent nnz(nnz-1)k*20 + linear nnz*20 + bias 20 + CSR (nnz*12+8) + outputs 12."""
    return nnz * (nnz - 1) * k * 20 + nnz * 20 + 20 + (nnz * 12 + 8) + 12


def refresh_mode(n_shards):
    """Where the engine refreshes / updates the features that occur once in a block
    (csrc/engine.hip refresh_mode; FFM_ENGINE_ROW_REFRESH overrides): 1 = refresh kernel + single
    kernel, 2 = refreshed by their row, 3 = and updated there (one shard: the row kernel has tmp_grad)."""
    env = os.environ.get("FFM_ENGINE_ROW_REFRESH", "")[:1]
    mode = {"0": 1, "2": 2, "3": 3}.get(env, 3)
    return 2 if mode == 3 and n_shards > 1 else mode


def kernel_share_bytes(kernel, blocks_feat, nnz, k, n_shards):
    """Algorithmic bytes ONE launch of `kernel` is responsible for, averaged over the bench's
    blocks.  SURVEY.md 8(d)'s per-row total (20 B per touched slot-factor: read n,z + write w,n,z)
    is apportioned to the kernels that move those bytes, so the shares add up to it:
      refresh              : read (n,z) + write w = 12 B per touched slot-factor of the block's
                             DISTINCT features (with (n,z) frozen over the block every occurrence
                             would compute the same w, so once per distinct feature is all the
                             block algorithm needs; the per-row figure of 8(d) counts it per
                             occurrence, and step_algorithmic_GBps keeps that accounting) -- the
                             refresh kernel's, except (modes 2, 3) for the features that occur once
                             in the block, whose 12 B belong to the row kernel
      row kernel           : CSR in, linear weights, logit / tmp_grad / loss out (+ the above; + in
                             mode 3 the once-only features' update)
      update               : write (n,z) = 8 B per slot-factor of the occurrences each kernel owns:
                             ffm_update_all_kernel every feature with two or more occurrences (on a
                             compact shard the few-occurrence ones have a launch of their own,
                             ffm_update_small_flat_kernel), ffm_update_single_kernel the once-only
                             features where their row could not update them (shards)
    Under field-pair sharding every rank moves 1/n_shards of the slot-factors."""
    if MODEL == "FM":
        return fm_kernel_share_bytes(kernel, blocks_feat, nnz, k)
    per_occ = (nnz - 1) * k  # slot-factors one occurrence of a feature touches
    rows = [len(f) // nnz for f in blocks_feat]
    mode = refresh_mode(n_shards)
    counts = [np.unique(f, return_counts=True)[1] for f in blocks_feat]
    once = [int((c == 1).sum()) for c in counts]
    if "refresh" in kernel:
        return float(np.mean([(len(c) - (o if mode >= 2 else 0)) * per_occ * 12 / n_shards
                              for c, o in zip(counts, once)]))
    if "row_kernel" in kernel:
        per_once = (12 if mode >= 2 else 0) + (8 if mode == 3 else 0)
        return float(np.mean([r * (nnz * 12 + 12 + (nnz * 12 + 8) + 4 + 16) + o * per_occ * per_once / n_shards
                              for r, o in zip(rows, once)]))
    if "single" in kernel and mode == 3:
        return 0.0
    flat = n_shards > 1  # (csrc/engine.hip: flat_pays -- compact shards' short records)
    shares = []
    for f, c in zip(blocks_feat, counts):
        # engine_step.h: a whole model's large update runs as three launches side by side
        # (hot | few | giant), else as one (plus, for the longest features, a second pass and a join)
        split = update_split(len(f), k, n_shards)
        occ = {"single": c[c == 1].sum(), "few": c[(c > 1) & (c <= SMALL_MAX)].sum(), "hot": c[c > SMALL_MAX].sum(),
               "giant": c[c >= GIANT_MIN].sum(), "super": c[c >= SUPER_MIN].sum()}
        if "single" in kernel:
            owned = occ["single"]
        elif "small_flat" in kernel or "<few>" in kernel:
            owned = occ["few"]
        elif "<giant>" in kernel:
            owned = occ["giant"] if split else occ["super"]
        elif split:  # the hot range's launch
            owned = occ["hot"] - occ["giant"]
        else:  # ffm_update_all_kernel / ffm_update_generic_kernel
            owned = occ["hot"] - occ["super"] + (0 if flat else occ["few"])
        shares.append(owned * per_occ * 8 / n_shards)
    return float(np.mean(shares))


GIANT_MIN, SUPER_MIN = 257, 2048  # csrc/engine_types.h: kGiantMin, FFM_SUPER_MIN


def update_split(block_nnz, k, n_shards):
    """engine_step.h's rule for the FFM update: three launches side by side for a whole model's large
    blocks (FFM_UPDATE_SPLIT overrides)."""
    env = os.environ.get("FFM_UPDATE_SPLIT")
    if env is not None:
        return int(env) != 0
    return n_shards == 1 and block_nnz * k >= (4 << 20)


def fm_kernel_share_bytes(kernel, blocks_feat, nnz, k):
    """The same apportioning for FM (nnz*k*20 + nnz*20 + 20 + nnz*8 + 8 + 12 B per
    row; a feature's record is k slot-factors).  fm_row_wave_kernel (csrc/kernels_fm.h) refreshes
    every OCCURRENCE's record (read n,z + write w = 12 B per factor), reads the CSR entries, the
    linear terms, writes logit / tmp_grad / loss / the row's k factor sums, and applies the (n, z)
    step (8 B per factor) of the features that occur once in the block; fm_update_kernel owns every
    other feature (8 B per factor-occurrence)."""
    rows = [len(f) // nnz for f in blocks_feat]
    counts = [np.unique(f, return_counts=True)[1] for f in blocks_feat]
    if "row_kernel" in kernel:
        return float(np.mean([r * (nnz * k * 12 + nnz * 12 + (nnz * 8 + 8) + 4 + 16 + 4 * k) + int((c == 1).sum()) * (k * 8 + 8)
                              for r, c in zip(rows, counts)]))
    # fm_update_kernel: every feature in more than one row of the block
    return float(np.mean([c[c > 1].sum() * k * 8 for c in counts]))


def cpu_baseline(args, gen_kwargs):
    """The reference's own multi-thread CPU path timed on this host on a bounded sample of the same
    workload: oracle/_ref (the unmodified reference model classes compiled from /root/reference by
    oracle/Makefile; the harness runs FtrlOffline::one_epoch's loop, ftrl_offline.cpp:63-91, over
    them) -- kind "reference".  Where that build is absent, the oracle's restatement -- kind "port"
    (profiles/archive/r02_cpu_baseline_validation.json: port/reference = 1.23 at 1 thread, 0.89 at 8).
    Checker code, used here only as the reported baseline."""
    from oracle import pyoracle
    from oracle.pyoracle import CpuModel
    from ftrl_ffm_amd import synth
    n_feats = N_FIELDS * 2048  # host-RAM bound sample of the same F / k / nnz / id distribution
    rows = args.cpu_rows
    g = synth.Generator(N_FIELDS, n_feats, **gen_kwargs)
    blk = g.block(rows)
    if MODEL != "FFM":
        blk.field[:] = 0  # libsvm rows
    kind = "ref" if pyoracle.have_ref() else "oracle"
    if kind == "ref":
        try:
            lib, _ = pyoracle._lib("ref")
            if not (hasattr(lib, "fr_create_sized") and hasattr(lib, "fr_train_rows_threaded")):
                kind = "oracle"
        except OSError:
            kind = "oracle"
    ncpu = os.cpu_count() or 1
    rng = np.random.default_rng(5)
    st = None
    tried = {}
    for threads in sorted({1, min(8, ncpu), ncpu}):
        m = CpuModel(kind, MODEL, n_feats, N_FIELDS if MODEL == "FFM" else 1, N_FACTORS)
        if st is None:
            st = m.zero_state()
            st["vec_w"][...] = rng.normal(0, 0.02, st["vec_w"].shape).astype(np.float32)
            st["vec_n"][...] = rng.uniform(0.05, 1.0, st["vec_n"].shape).astype(np.float32)
            st["vec_z"][...] = rng.normal(0, 0.3, st["vec_z"].shape).astype(np.float32)
        m.set_state(st)
        sec, _ = m.train_rows_threaded(blk, threads)
        tried[threads] = rows / sec
        del m
    best_t = max(tried, key=tried.get)
    out = {"value": round(tried[best_t], 1), "unit": "samples/s", "cores": best_t,
           "kind": "reference" if kind == "ref" else "port",
           "sample": "%d rows, %s F=%d k=%d nnz=%d Zipf(1.1), n_feats=%d, warm state; %s at n_threads in "
                     "{1, 8, all}: %s; best reported; host has %d cores" % (
                         rows, MODEL, N_FIELDS, N_FACTORS, N_FIELDS, n_feats,
                         "the reference's FtrlOffline::one_epoch loop over its own model "
                         "(oracle/_ref)" if kind == "ref" else "oracle fo_train_rows_threaded",
                         ", ".join("%dT=%.0f/s" % kv for kv in sorted(tried.items())), ncpu)}
    # The same loop at the largest model this host holds comfortably ("the largest
    # n_feats that fits"): its state then lives in DRAM instead of the caches, and a mutex per feature
    # is rarely contended.  Zero-filled state (the harness sizes the reference's members itself: its
    # constructor takes ~32 us per weight), so the latent part is dead as in a fresh model -- the loop
    # does the same work per row either way (no branch in ffm.cpp:90-136 depends on the values).
    if args.cpu_big_feats > 0 and MODEL == "FFM":
        try:
            import psutil
            avail = psutil.virtual_memory().available
        except Exception:  # noqa: BLE001
            avail = 0
        big = args.cpu_big_feats - args.cpu_big_feats % N_FIELDS
        need = big * (N_FIELDS * N_FACTORS + 1) * 4 * 3 * 1.3
        if avail and need > avail * 0.6:
            big = int(avail * 0.6 / ((N_FIELDS * N_FACTORS + 1) * 12 * 1.3))
            big -= big % N_FIELDS
        if big >= 10 * n_feats:
            gb = synth.Generator(N_FIELDS, big, **gen_kwargs)
            blk_b = gb.block(args.cpu_big_rows)
            mb = CpuModel(kind, MODEL, big, N_FIELDS, N_FACTORS)
            sec, _ = mb.train_rows_threaded(blk_b, best_t)
            out["large_model"] = {"value": round(args.cpu_big_rows / sec, 1), "unit": "samples/s", "cores": best_t,
                                  "n_feats": big, "state_GB": round(big * (N_FIELDS * N_FACTORS + 1) * 12 / 1e9, 1),
                                  "rows": args.cpu_big_rows,
                                  "note": "same loop, same thread count, zero-filled state (fresh model)"}
            del mb
    return out


def spawn_ranks(n):
    """One rank per GPU through torch.distributed.run, as the driver would launch them
    (rendezvous on 127.0.0.1, a free port); returns the launcher's exit code."""
    import subprocess
    # (--standalone: the launcher picks a free port itself -- no bind-then-close race)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1",
           "--nnodes=1", "--nproc-per-node", str(n), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    return subprocess.call(cmd, env=env)


def block_logloss_cost(rows):
    """What training in blocks of `rows` rows costs in logloss against the reference's per-sample
    loop (FFM 39 x 16, default hyper-parameters, block ramp 32; measured with the oracle, which the
    GPU path equals bit for bit): the committed table's entry for the largest block <= rows, from
    its longest run."""
    best = None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "**", "r*_logloss_vs_block*.json"), recursive=True)):
        try:
            d = json.load(open(fn))
        except (OSError, ValueError):
            continue
        for b, e in d.get("blocks", {}).items():
            key = (int(b), e.get("full_blocks", 0))
            if int(b) <= rows and (best is None or key > best[0]):
                best = (key, {"block_rows": int(b), "d_train_logloss": e["d_train"], "d_eval_logloss": e["d_eval"],
                              "full_blocks": e.get("full_blocks"), "shape": d.get("shape"),
                              "source": os.path.relpath(fn, ROOT)})
    return best[1] if best else None


def workload_key(config, dist, state, rows, n_feats, n_shards):
    """What identifies the block stream a kernel's counters were collected on."""
    return "%s|%s|%s|rows=%d|n_feats=%d|shards=%d" % (config, dist, state, rows, n_feats, n_shards)


def matching_pmc_summary(key):
    """The newest committed PMC summary (tools/summarize_profile.py) whose capture ran THIS workload:
    the summary records the workload key of the bench line printed under the profiler."""
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_summary*.json")), reverse=True):
        try:
            with open(fn) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get("_capture", {}).get("workload_key") == key:
            return fn, d
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n-feats", type=int, default=0, help="override total n_feats")
    ap.add_argument("--rows", type=int, default=0, help="override rows per step (global)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--dist", default="zipf", choices=["zipf", "uniform"])
    ap.add_argument("--state", default="warm", choices=["warm", "fresh"])
    ap.add_argument("--n-blocks", type=int, default=0, help="distinct synthetic blocks cycled (default 64)")
    ap.add_argument("--cpu-rows", type=int, default=100000)
    ap.add_argument("--cpu-big-feats", type=int, default=2_000_000,
                    help="second CPU-baseline leg at this many features (0: off); shrunk to fit host RAM")
    ap.add_argument("--cpu-big-rows", type=int, default=60000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-resident", action="store_true", help="skip the HBM-resident leg")
    ap.add_argument("--no-eval", action="store_true", help="skip the evaluation (predict + logloss) leg")
    ap.add_argument("--resident-blocks", type=int, default=8, help="blocks uploaded for the resident leg")
    ap.add_argument("--host-copy", action="store_true",
                    help="host leg through ffm_engine_train_batch_async (pageable caller arrays, "
                         "copied into the engine's pinned slot) instead of zero-copy staging")
    ap.add_argument("--resident-only", action="store_true",
                    help="time only the HBM-resident loop (tuning aid: `value` is then NOT the metric)")
    ap.add_argument("--no-lookahead", action="store_true", help="resident leg: group each block inline")
    ap.add_argument("--config", default="c5", choices=sorted(CONFIGS))
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend (gloo + --same-device: functional dry run of the "
                         "multi-rank path on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="every rank uses cuda:0 (dry run)")
    ap.add_argument("--emulate-shards", type=int, default=0,
                    help="tuning aid on one GPU: run ONE rank's share of an N-GPU field-pair-sharded "
                         "job (1/N of the field pairs, no all-reduce) and print what the N-GPU "
                         "job's rate would be if the exchange were free")
    ap.add_argument("--emulate-rank", type=int, default=0)
    ap.add_argument("--all-columns", action="store_true",
                    help="sharded runs: hand every rank all columns of every row (the engine drops "
                         "the ones it owns nothing of) instead of only the rank's own columns")
    ap.add_argument("--no-field-map", action="store_true",
                    help="sharded runs: do not tell the engine the per-field id ranges (every shard "
                         "then keeps full-length records and all columns)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    if n_gpus > 1 and world == 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, one fresh
        # process per GPU, BEFORE anything in this process has touched the GPU (no torch import,
        # no HIP call yet), wait for them and hand their exit code on.  Never exec from a process
        # that has initialised the GPU.
        sys.exit(spawn_ranks(n_gpus))

    import torch
    import ftrl_ffm_amd as fa
    from ftrl_ffm_amd import sharding, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and world != n_gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (n_gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    fa.build()

    global N_FIELDS, N_FACTORS, MODEL
    cfgw = CONFIGS[args.config]
    model = MODEL = cfgw["model"]
    N_FIELDS, N_FACTORS = cfgw["fields"], cfgw["factors"]
    emu = args.emulate_shards if world == 1 else 0
    n_shards = emu or world
    rows = args.rows or cfgw["rows"] * (n_shards if args.scaling == "weak" else 1)
    n_feats = args.n_feats or cfgw["feats"]
    n_feats -= n_feats % N_FIELDS
    rec_bytes = 3 * (N_FIELDS if model == "FFM" else 1) * N_FACTORS * 4
    free_b, _total_b = torch.cuda.mem_get_info()
    reduced = False
    budget = int(free_b * 0.9) - (4 << 30)
    if args.same_device:
        budget //= max(world, 1)
    compact = n_shards > 1 and model == "FFM" and not args.no_field_map
    shard_bytes = rec_bytes // n_shards * 5 // 4 if compact else rec_bytes  # compact: ~1/n of a record (+ padding)
    if n_feats * shard_bytes > budget:
        n_feats = budget // shard_bytes
        n_feats -= n_feats % N_FIELDS
        reduced = True
    gen_kwargs = dict(dist=args.dist, seed=42)

    # one side stream carries the engine's kernels and (through torch) the RCCL collective
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream

    def make_engine(nf):
        # sharded: the generator's per-field id ranges go to the engine, so every shard stores only
        # its slots (~1/n_shards of the tensor) and looks only at the columns it has pairs of
        # (one shard: the ranges only tell the grouping that it may sort every field's ids by themselves)
        known = model == "FFM" and not args.no_field_map
        fs = (np.arange(N_FIELDS + 1, dtype=np.int64) * (nf // N_FIELDS)).astype(np.int32) if known else None
        return fa.Engine(model, nf, N_FIELDS, N_FACTORS, max_batch_rows=rows,
                         max_batch_nnz=rows * N_FIELDS, device_id=local_rank, n_shards=n_shards,
                         shard_rank=args.emulate_rank if emu else rank, stream=stream, seed=42,
                         max_row_nnz=N_FIELDS, field_start=fs)

    eng = None
    while eng is None:  # an allocation that does not fit is retried 10 % smaller, never fatal
        try:
            eng = make_engine(n_feats)
        except fa.EngineError as err:
            if err.code != -3 or n_feats < 10 * N_FIELDS:
                raise
            n_feats = int(n_feats * 0.9)
            n_feats -= n_feats % N_FIELDS
            reduced = True
    if dist is not None:  # every rank must train the same model shape
        t = torch.tensor([n_feats], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) != n_feats:
            n_feats = int(t.item())
            eng.close()
            eng = make_engine(n_feats)
            reduced = True
    if args.state == "warm":
        eng.fill_state(seed=7, n_lo=0.05, n_hi=1.0, z_stddev=0.3)

    # identical synthetic blocks on every rank, parsed and resident in HOST memory
    n_blocks = args.n_blocks or max(8, min(64, (1 << 19) // rows))
    gen = synth.Generator(N_FIELDS, n_feats, **gen_kwargs)
    keep = None
    if compact and not args.all_columns:
        # a compact shard never looks at the columns of fields it owns nothing of: its loader hands
        # the engine only the kept columns (about half of them at 8 shards), which halves the
        # rank's PCIe traffic and the entries its grouping sorts
        plan = fa.shard_plan(N_FIELDS, n_shards, field_map=True)
        me = args.emulate_rank if emu else rank
        keep = (plan["pair_owner"] == me).any(axis=1) | (plan["lin_owner"] == me)
    blocks_feat = []

    def make_blocks(n_rows_block, count, note_feats=False):
        out_blocks = []
        for _ in range(count):
            b = gen.block(n_rows_block)
            if model != "FFM":
                b.field[:] = 0  # libsvm rows
            if note_feats and len(blocks_feat) < 8:
                blocks_feat.append(b.feat)
            if keep is not None:
                sel = keep[b.field]
                per_row = np.add.reduceat(sel.astype(np.int64), b.row_ptr[:-1].astype(np.int64))
                b = synth.Block(np.concatenate([[0], np.cumsum(per_row)]).astype(np.int32), b.field[sel].copy(),
                                b.feat[sel].copy(), b.val[sel].copy(), b.label)
            out_blocks.append(b)
        return out_blocks
    host_blocks = make_blocks(rows, n_blocks, note_feats=True)
    total_steps = args.steps + args.warmup + 1
    logit = torch.zeros(rows, dtype=torch.float32, device="cuda")
    loss_sum = torch.zeros(2 * total_steps, dtype=torch.float64, device="cuda")
    sharded = n_shards > 1
    sstep = sharding.ShardedStep(eng, dist, logit) if sharded else None

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- leg 1 (the metric): rows stream from host memory, H2D inside the timed region ----
    zero_copy = not args.host_copy
    pinned_keep = []
    def pin_blocks(blocks):
        # the parsed rows live in page-locked host memory (allocated pinned, as a loader that
        # targets this engine would): the DMA reads them in place
        for b in blocks:
            for name in ("row_ptr", "field", "feat", "val", "label"):
                t = torch.from_numpy(getattr(b, name)).pin_memory()
                pinned_keep.append(t)
                setattr(b, name, t.numpy())
    if zero_copy and not args.resident_only:
        pin_blocks(host_blocks)

    main_blocks = host_blocks
    STAGE_AHEAD = int(os.environ.get("BENCH_STAGE_AHEAD", "2"))  # blocks staged ahead of the one in training

    def run_host(first, count, blocks=None):
        """`count` steps; returns the sum of the steps' losses (all enqueued work is flushed)."""
        host_blocks = blocks if blocks is not None else main_blocks
        n_blocks = len(host_blocks)
        if count == 0:
            return 0.0
        if not sharded and not zero_copy:
            for i in range(count):
                eng.train_batch_async(host_blocks[(first + i) % n_blocks])
            return eng.train_flush()
        # block t+2 is staged BEFORE block t's training is enqueued: its upload then sits ahead of
        # block t's hot update on that stream (it runs when block t-1 ends) and its grouping is
        # scheduled into block t's refresh / row window with until the end of block t+1 to finish
        staged = 0
        for i in range(count):
            while staged < min(i + 1 + STAGE_AHEAD, count):
                eng.stage_batch(host_blocks[(first + staged) % n_blocks], zero_copy)
                staged += 1
            if sharded:
                sstep.train_staged(host_blocks[(first + i) % n_blocks].n_rows, loss_sum.data_ptr() + 8 * (first + i))
            else:
                eng.train_staged(None, loss_sum.data_ptr() + 8 * (first + i))
        eng.sync()
        # (every step's loss is in HBM now; adding the K doubles up on the host -- a torch reduction, a
        # second device sync and a D2H copy, ~0.2 ms of Python and runtime latency -- is reporting, not
        # the train loop: timed() does it after the closing fence)
        return lambda: float(loss_sum[first:first + count].sum().item())

    # ---- leg 2 (extra): the same blocks already resident in HBM ----
    dev_blocks = []

    def upload_resident():
        for b in host_blocks[:args.resident_blocks]:
            dev_blocks.append(dict(
                n_rows=b.n_rows, nnz=b.nnz,
                row_ptr=torch.from_numpy(b.row_ptr).cuda(), field=torch.from_numpy(b.field).cuda(),
                feat=torch.from_numpy(b.feat).cuda(), val=torch.from_numpy(b.val).cuda(),
                label=torch.from_numpy(b.label).cuda()))

    def step_resident(i, blk):
        ptr = lambda t: t.data_ptr()  # noqa: E731
        out_loss = loss_sum.data_ptr() + 8 * (total_steps + i % total_steps)
        if not sharded:
            eng.train_batch_device(blk["n_rows"], blk["nnz"], ptr(blk["row_ptr"]), ptr(blk["field"]),
                                   ptr(blk["feat"]), ptr(blk["val"]), ptr(blk["label"]),
                                   ptr(logit), out_loss)
        else:
            sstep(blk["n_rows"], blk["nnz"], ptr(blk["row_ptr"]), ptr(blk["field"]), ptr(blk["feat"]),
                  ptr(blk["val"]), ptr(blk["label"]), out_loss)

    def prepare(blk):  # the scheduler's look-ahead: group the next block beside this one's update
        ptr = lambda t: t.data_ptr()  # noqa: E731
        eng.prepare_device(blk["n_rows"], blk["nnz"], ptr(blk["row_ptr"]), ptr(blk["field"]),
                           ptr(blk["feat"]), ptr(blk["val"]))

    DEPTH = 2  # blocks grouped ahead (ffm_engine_prepare_device): two hide the grouping completely

    def run_resident(first, count):
        prepared = 0  # blocks of this run prepared so far (block i itself counts once i is prepared)
        for i in range(count):
            # the look-ahead for block i+2 goes in BEFORE block i's training (see run_host)
            while not args.no_lookahead and prepared < min(i + DEPTH + 1, count):
                if prepared >= i:
                    prepare(dev_blocks[(first + prepared) % len(dev_blocks)])
                prepared += 1
            step_resident(first + i, dev_blocks[(first + i) % len(dev_blocks)])

    import gc

    def quiet_gc():
        # the interpreter's cyclic garbage collector stays out of the timed region (BENCH_GC=1 lets it
        # run): a full collection over torch's and numpy's module graphs takes tens of milliseconds
        # -- one of them inside 20 steps of 1.1 ms would be most of the measurement.  Collected
        # BEFORE the warm-up steps, not between them and the timed ones: tens of idle milliseconds
        # there let the GPU's clocks fall, and the first timed steps paid ~1 ms for it (a 20-step run
        # read 1.116 ms per step where a 200-step run read 1.059).
        gc.collect()
        if os.environ.get("BENCH_GC", "0") != "1":
            gc.disable()

    def timed(run, first, count, warm=None):
        if warm is not None:  # (the main leg has had quiet_gc() + its warm-up steps already)
            quiet_gc()
            warm()
        fence()
        t0 = time.perf_counter()
        out = run(first, count)
        fence()
        el = time.perf_counter() - t0
        gc.enable()
        if callable(out):
            out = out()
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, out

    # warm-up with every kernel timed (the table on stderr, and which kernel dominates); the timed
    # region then carries HIP events only around that dominant kernel
    host_leg = not args.resident_only
    if args.resident_only or not args.no_resident:
        upload_resident()
    # The resident leg (an extra, not the metric) runs first: the metric's W warm-up + K timed steps
    # then start on a GPU that has been busy for a few hundred milliseconds.  (Measured back to back on
    # one box, 20-step shape: 1.085 / 1.088 / 1.085 ms with the metric first, 1.078 / 1.089 / 1.084 this
    # way -- the order does not move the metric; the resident leg itself reads 1 - 2 % better warm.)
    resident = None
    total_rows = rows * args.steps

    def measure_resident():
        # (its own warm-up is ~0.1 s of steps: timed straight after set-up this leg read 1 - 4 % slower)
        el2, _ = timed(run_resident, args.warmup, args.steps, warm=lambda: run_resident(0, 100))
        return {"value": round(total_rows / el2, 1), "unit": "samples/s",
                "ms_per_step": round(1000.0 * el2 / args.steps, 4),
                "note": "same loop over %d blocks already resident in HBM (no H2D); not the metric"
                        % len(dev_blocks)}
    if host_leg and not args.no_resident:
        resident = measure_resident()
    if not args.no_profile:
        eng.profile_enable(True)
    quiet_gc()
    # the W warm-up steps: all but the last two with every kernel timed (the table), then -- after the
    # table has been read out, a millisecond or two with the GPU idle -- the last two as the timed
    # steps will run, so that the timed region starts on a busy GPU
    main_run = run_host if host_leg else run_resident
    w_tail = min(2, max(0, args.warmup - 1)) if not args.no_profile else 0
    main_run(0, args.warmup - w_tail)
    fence()
    table = ""
    if not args.no_profile:
        table = eng.profile_dump()
        eng.profile_focus()
    if w_tail:
        main_run(args.warmup - w_tail, w_tail)
    elapsed, host_loss = timed(main_run, args.warmup, args.steps)
    kname, klaunches, kms = ("", 0, 0.0)
    if not args.no_profile:
        kname, klaunches, kms = eng.profile_read()
        eng.profile_enable(False)
    value = total_rows / elapsed
    if host_leg:
        train_loss = host_loss / total_rows
    else:
        train_loss = float(loss_sum[total_steps + args.warmup:total_steps + args.warmup + args.steps]
                           .sum().item()) / total_rows
    host_copy = None
    if host_leg and zero_copy and not sharded and not args.no_resident:
        # the same loop for a caller whose rows are NOT page-locked: every block is first copied
        # into the engine's own pinned staging slot (ffm_engine_train_batch_async), two in flight
        def run_copy(first, count):
            for i in range(count):
                eng.train_batch_async(host_blocks[(first + i) % n_blocks])
            return eng.train_flush()
        el3, _ = timed(run_copy, args.warmup, args.steps, warm=lambda: run_copy(0, min(args.warmup, 3)))
        host_copy = {"value": round(total_rows / el3, 1), "unit": "samples/s",
                     "ms_per_step": round(1000.0 * el3 / args.steps, 4),
                     "note": "rows handed over through the copying entry point (pageable memory: one host memcpy "
                             "per block into the engine's pinned slot); not the metric"}

    # ---- sharded runs: the OTHER scaling beside the one `value` is quoted on ----
    # `--scaling weak` (default) multiplies the block with the GPUs; SURVEY.md 8(d)'s literal C5 is
    # 8192 rows per step GLOBALLY -- the strong line.  Both legs on the same engines, same process.
    other_scaling = None
    if sharded and host_leg and zero_copy and not args.rows:
        o_rows = cfgw["rows"] * (1 if args.scaling == "weak" else n_shards)
        if o_rows <= rows:  # (the engines were created for `rows` rows per block)
            o_blocks = make_blocks(o_rows, max(8, min(64, (1 << 19) // o_rows)))
            pin_blocks(o_blocks)
            o_steps = args.steps
            el_o, _ = timed(lambda f, c: run_host(f, c, o_blocks), 0, o_steps,
                            warm=lambda: run_host(0, min(args.warmup, 5), o_blocks))
            o_val = o_rows * o_steps / el_o
            other_scaling = {"scaling": "strong" if args.scaling == "weak" else "weak",
                             "value": round(o_val, 1), "unit": "samples/s",
                             "rows_per_step": o_rows, "ms_per_step": round(1000.0 * el_o / o_steps, 4),
                             "block_logloss_cost": block_logloss_cost(o_rows),
                             "note": "same engines, same run: %d rows per step over all GPUs (H2D included)" % o_rows}

    # ---- evaluation: predict + logloss of the same blocks, pipelined ----
    eval_leg = None
    if (host_leg or args.resident_only) and zero_copy and not sharded and not args.no_resident and not args.no_eval:
        def run_eval(first, count):
            for i in range(count):
                eng.predict_batch_async(host_blocks[(first + i) % n_blocks], zero_copy=True)
            return eng.train_flush()

        def run_eval_resident(first, count):
            ptr = lambda t: t.data_ptr()  # noqa: E731
            for i in range(count):
                blk = dev_blocks[(first + i) % len(dev_blocks)]
                eng.predict_batch_device(blk["n_rows"], blk["nnz"], ptr(blk["row_ptr"]), ptr(blk["field"]),
                                         ptr(blk["feat"]), ptr(blk["val"]), ptr(blk["label"]), 0, ptr(logit),
                                         loss_sum.data_ptr())
        el5, _ = timed(run_eval_resident, args.warmup, args.steps, warm=lambda: run_eval_resident(0, 3))
        # (--resident-only, the counter passes of tools/pmc_sq.sh: no host blocks -- the H2D figures repeat the resident ones)
        el4, eval_loss = (timed(run_eval, args.warmup, args.steps, warm=lambda: run_eval(0, 3)) if host_leg
                          else (el5, float("nan")))
        # Rows that are one entry per field in field order (the generator's, like python/generate_data.py's)
        # may go over WITHOUT their field array (include/ffm_engine.h: field == NULL on the staged entry
        # points, validated per block; the upload kernel writes the array): a third fewer bytes over PCIe.
        lean_eval = None
        if host_leg and model == "FFM" and all(
                b.nnz == b.n_rows * N_FIELDS and np.array_equal(b.field, np.tile(np.arange(N_FIELDS, dtype=np.int32), b.n_rows))
                for b in host_blocks):
            import copy

            def without_field(b):
                c = copy.copy(b)
                c.__dict__.pop("_ffm_csr_args", None)
                c.field = None
                return c
            lean_blocks = [without_field(b) for b in host_blocks]

            def run_eval_lean(first, count):
                for i in range(count):
                    eng.predict_batch_async(lean_blocks[(first + i) % n_blocks], zero_copy=True)
                return eng.train_flush()
            el6, lean_loss = timed(run_eval_lean, args.warmup, args.steps, warm=lambda: run_eval_lean(0, 3))
            lean_eval = (el6, lean_loss)
        if not args.no_profile:  # the leg's kernels, every launch timed (after the timed regions)
            eng.profile_enable(True)
            run_eval_resident(0, 20)
            fence()
            sys.stderr.write("[eval, resident] " + eng.profile_dump() + "\n")
            eng.profile_enable(False)
        # read-only path: every touched weight once (4 B per slot-factor), CSR in, loss out
        if model == "FFM":
            eval_bytes = N_FIELDS * (N_FIELDS - 1) * N_FACTORS * 4 + N_FIELDS * 4 + 4 + (N_FIELDS * 12 + 8) + 8
        else:
            eval_bytes = N_FIELDS * N_FACTORS * 4 + N_FIELDS * 4 + 4 + (N_FIELDS * 8 + 8) + 8
        eval_leg = {"value": round(total_rows / el4, 1), "unit": "samples/s",
                    "ms_per_step": round(1000.0 * el4 / args.steps, 4),
                    "resident": round(total_rows / el5, 1),
                    "logloss": round(eval_loss / total_rows, 6),
                    "algorithmic_bytes_per_row": eval_bytes,
                    "roofline_frac": round(total_rows / el4 * eval_bytes / 1e9 / PEAK_HBM_GBPS, 4),
                    "roofline_frac_resident": round(total_rows / el5 * eval_bytes / 1e9 / PEAK_HBM_GBPS, 4),
                    "note": "predict + logloss of the same host blocks through ffm_engine_predict_batch_async "
                            "(upload on the side stream, H2D included) and of the resident blocks through "
                            "ffm_engine_predict_batch_device; not the metric"}
        if lean_eval:
            eval_leg["without_field_array"] = {
                "value": round(total_rows / lean_eval[0], 1), "ms_per_step": round(1000.0 * lean_eval[0] / args.steps, 4),
                "logloss": round(lean_eval[1] / total_rows, 6),
                "roofline_frac": round(total_rows / lean_eval[0] * eval_bytes / 1e9 / PEAK_HBM_GBPS, 4),
                "note": "the same blocks handed over with field == NULL (rows are one entry per field in field order; "
                        "the upload kernel writes the field array instead of pulling it over PCIe)"}

    if model == "FFM":
        bytes_row = algorithmic_bytes_per_row(N_FIELDS, N_FACTORS)
    else:  # FM, SURVEY.md 8(d): nnz*k*20 + nnz*20 + 20 + nnz*8 + 8 + 12
        bytes_row = N_FIELDS * N_FACTORS * 20 + N_FIELDS * 20 + 20 + N_FIELDS * 8 + 8 + 12

    if rank == 0:
        out = {
            "metric": "train samples/sec + logloss, FFM f=39 k=16" if args.config == "c5" else
                      "train samples/sec + logloss, " + args.config, "value": round(value, 1),
            "unit": "samples/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "h2d_included": bool(host_leg),
            "config": {
                "workload": "%s n_fields=%d n_factors=%d nnz=%d, %s ids, block=%d rows, n_feats=%d "
                            "(%.1f GB of w,n,z per GPU), %s state, reference default hyper-parameters, "
                            "%s" % (model, N_FIELDS, N_FACTORS, N_FIELDS,
                                    "Zipf(1.1)" if args.dist == "zipf" else "uniform", rows, n_feats,
                                    n_feats * shard_bytes / 1e9, args.state,
                                    "%d distinct blocks streamed from %s host memory (H2D in the timed region)"
                                    % (n_blocks, "page-locked" if zero_copy else "pageable")
                                    if host_leg else "blocks resident in HBM (no H2D)"),
                "workload_key": workload_key(args.config, args.dist, args.state, rows, n_feats, n_shards),
                "rows_per_step": rows, "n_feats": n_feats, "n_feats_reduced_to_fit": reduced,
                "n_blocks": n_blocks,
                "sharding": "field-pair x%d (%s), one all-reduce of %d partial logits per step"
                            % (n_shards, "compact storage, per-field id ranges" if compact else "full records",
                               rows) if n_shards > 1 else "none",
            },
            **({"emulated": "one rank's compute of a %d-GPU job on one GPU, no exchange; `value` is "
                            "what the %d-GPU job would reach if the all-reduce were free (tuning "
                            "aid, not a result)" % (emu, emu)} if emu else {}),
            "train_logloss": round(train_loss, 6),
            "step_algorithmic_GBps": round(value * bytes_row / n_gpus / 1e9, 1),
        }
        if sharded and model == "FFM":
            # what the block size of this line costs in logloss against the per-sample reference loop
            out["block_logloss_cost"] = block_logloss_cost(rows)
        if other_scaling:
            out["other_scaling"] = other_scaling
        if resident:
            out["resident"] = resident
        if host_copy:
            out["config"]["pageable_host_copy"] = host_copy
        if eval_leg:
            out["eval"] = eval_leg
        if kname:
            share = kernel_share_bytes(kname, blocks_feat, N_FIELDS, N_FACTORS, n_shards)
            avg_s = kms / 1000.0 / max(klaunches, 1)
            achieved = share / avg_s / 1e9
            traffic = None
            pmc, pmc_d = matching_pmc_summary(out["config"]["workload_key"]) if world == 1 and not emu else (None, None)
            if pmc_d:  # rocprofv3 --pmc passes of this same workload (tools/profile_round.sh)
                for name, v in pmc_d.items():
                    # (FM rows with k <= 64 run fm_row_wave_kernel; the profile's label is fm_row_kernel)
                    if not name.startswith("_") and name.split("<")[0].replace("_wave", "") == kname.split("<")[0]:
                        traffic = v["hbm_bytes_per_launch"]
            out["roofline"] = {
                "bound": "hbm", "kernel": kname, "achieved": round(achieved, 1),
                "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBPS, 4),
                "traffic": traffic, "traffic_source": os.path.basename(pmc) if traffic else None,
                "avg_launch_us": round(avg_s * 1e6, 2), "launches": klaunches,
                "algorithmic_bytes_per_launch": int(share),
                # the whole step against the same roof: SURVEY.md 8(d)'s bytes per row x rows/s
                "whole_step": {"achieved": out["step_algorithmic_GBps"],
                               "frac": round(out["step_algorithmic_GBps"] / PEAK_HBM_GBPS, 4),
                               "bytes_per_row": int(bytes_row)},
                "note": "traffic = bytes leaving the L2 (Infinity-Cache hits included, FETCH_SIZE x2 for "
                        "float4 record streams: tools/summarize_profile.py), from the committed capture of "
                        "this same workload (config.workload_key), else null; "
                        "warm-up spans of all kernels are in other_kernels",
            }
            # the other big kernels, from the fully timed warm-up launches (same accounting)
            others = []
            for line in table.splitlines():
                parts = line.split()
                nm = parts[0]
                full = {"row_kernel<train>": "fm_row_kernel<train>", "update_kernel": "fm_update_kernel"}.get(nm) \
                    if model == "FM" else \
                       {"row_kernel<train>": "ffm_row_kernel<train>",
                        "refresh_kernel": "ffm_refresh_kernel",
                        "update_single_kernel": "ffm_update_single_kernel",
                        "update_few_kernel": "ffm_update_small_flat_kernel" if n_shards > 1 else "ffm_update_all_kernel<few>",
                        "update_giant_kernel": "ffm_update_all_kernel<giant>",
                        "update_kernel": "ffm_update_all_kernel"}.get(nm)
                if not full:
                    continue
                us = float(parts[-1])
                sh = kernel_share_bytes(full, blocks_feat, N_FIELDS, N_FACTORS, n_shards)
                others.append({"kernel": full, "avg_launch_us": us,
                               "algorithmic_bytes_per_launch": int(sh),
                               "achieved": round(sh / (us * 1e-6) / 1e9, 1),
                               "frac": round(sh / (us * 1e-6) / 1e9 / PEAK_HBM_GBPS, 4)})
            out["roofline"]["other_kernels"] = others
        if n_gpus == 1 and not emu and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, gen_kwargs)
        print(json.dumps(out), flush=True)
        if table:
            sys.stderr.write(table)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
