"""Several GPUs as far as one GPU can show it: the library's own sharding -- voice ranges, routing by global
voice index, the sum of the ranges' blocks -- in both forms (one process owning the devices: knh_bank_create_multi_device;
one process per GPU: knh_bank_create_rank), the RCCL communicator with the one rank a one-GPU box allows, two real ranks in
two processes on one GPU with the sum carried by gloo, and bench.py's N = 2 control flow in rehearsal mode."""
import ctypes as C
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import assert_bit_equal, fire_all, make_gpu
from knaster_amd import _lib as L
from knaster_amd import configs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bank(knh, w, **kw):
    b = knh.VoiceBank(w.stages, w.n_voices, w.sample_type, w.out_channels, L.MIX_TREE, -1, False, **kw)
    for s, a in w.ctor.items():
        b.set_ctor_args(s, a)
    b.init(configs.SAMPLE_RATE, w.block_size)
    return b


def _script(w, launch, bank):
    v = np.arange(w.n_voices, dtype=np.uint32)
    if launch == 0:
        fire_all(bank, w.n_voices, *w.restart)
        bank.param_apply_many(v[::3], 2, 0, L.VALUE_FLOAT, 300.0 + v[::3], block_offset=1)  # SVF cutoffs, block 1 of the launch
    if launch == 1:
        bank.param_apply_many(v, w.release[0], w.release[1], L.VALUE_TRIGGER, block_offset=2)
        bank.param_apply(w.n_voices - 1, 0, 0, 777.0)
        bank.set_delay_within_block_for_param(0, 0, 0, 3)  # not a WrPreciseTiming stage: a warning, no effect, on any form


@pytest.mark.parametrize("name,n_voices,block_size,n_dev", [("C3", 1000, 64, 3), ("C4", 300, 32, 2), ("M1", 600, 64, 4), ("C3", 130, 48, 8)])
def test_multi_device_bank_on_one_device(knh, name, n_voices, block_size, n_dev):
    """devices = [0, 0, ..]: every voice range on the same GPU, through the multi-device code path.  Same ranges and the same
    sum order as a host-sharded bank, hence bit-identical to it; within the tree-mix tolerance of the plain bank."""
    w = configs.config(name, n_voices=n_voices, block_size=block_size)
    md = _bank(knh, w, devices=[0] * n_dev)
    hs = _bank(knh, w, host_threads=n_dev)
    plain = _bank(knh, w)
    assert md.ranks() == hs.ranks() == min(n_dev, (n_voices + 63) // 64) and plain.ranks() == 1
    for launch in range(3):
        for b in (md, hs, plain):
            if name != "M1":
                _script(w, launch, b)
            elif launch == 0:
                fire_all(b, w.n_voices, *w.restart)
        a, _ = md.process_blocks(4)
        c, _ = hs.process_blocks(4)
        p, _ = plain.process_blocks(4)
        assert_bit_equal(a, c, f"{name} launch {launch}: multi-device vs host-sharded")
        assert np.max(np.abs(a.astype(np.float64) - p)) <= 1e-5
        assert np.abs(p).max() > 1e-4
    np.testing.assert_array_equal(md.read_done_frames(), plain.read_done_frames())
    for b in (md, hs, plain):
        b.close()


def test_multi_device_rejects_bad_arguments(knh):
    w = configs.config("C3", n_voices=128, block_size=32)
    with pytest.raises(L.KnasterHipError):
        knh.VoiceBank(w.stages, 128, w.sample_type, 2, L.MIX_TREE, devices=[0, 99])
    with pytest.raises(L.KnasterHipError):
        knh.VoiceBank(w.stages, 128, w.sample_type, 2, L.MIX_LEFT_FOLD, devices=[0, 0])
    with pytest.raises(L.KnasterHipError):
        knh.VoiceBank(w.stages, 128, w.sample_type, 2, L.MIX_TREE, rank=2, world=2, comm_id=b"\0" * 128)
    with pytest.raises(L.KnasterHipError):  # two ranks need a communicator id or a reduce function
        knh.VoiceBank(w.stages, 128, w.sample_type, 2, L.MIX_TREE, rank=0, world=2)


@pytest.mark.parametrize("name,n_voices,block_size", [("C3", 700, 64), ("C4", 200, 32)])
def test_rank_bank_with_one_rank_is_the_plain_bank(knh, name, n_voices, block_size):
    w = configs.config(name, n_voices=n_voices, block_size=block_size)
    r = _bank(knh, w, rank=0, world=1)
    p = _bank(knh, w)
    assert r.ranks() == 1 and r.outputs() == 2
    for launch in range(3):
        _script(w, launch, r)
        _script(w, launch, p)
        a, fa = r.process_blocks(5)
        b, fb = p.process_blocks(5)
        assert_bit_equal(a, b, f"{name} launch {launch}")
        assert fa == fb
    a, fa = r.process_block()
    b, fb = p.process_block()
    assert_bit_equal(a, b, "single block")
    np.testing.assert_array_equal(r.read_done_frames(), p.read_done_frames())
    r.close()
    p.close()


def test_two_ranks_in_one_process_partition_the_voices(knh):
    """Rank 0 and rank 1 of a two-rank bank, both in this process, with a reduce function that leaves the buffers alone:
    each renders exactly its own voice range (global indices in, calls for the other rank's voices dropped), so the two
    partial mixes add up to the plain bank's mix, and each equals a plain bank of just that range."""
    n, bs = 1000, 64
    w = configs.config("C3", n_voices=n, block_size=bs)
    seen = []

    def no_reduce(_user, buf, count, sample_type, root, stream):
        seen.append((count, sample_type, root))
        return 0
    ranks = [_bank(knh, w, rank=r, world=2, reduce_fn=no_reduce) for r in range(2)]
    plain = _bank(knh, w)
    parts = []
    for r in range(2):
        lo, cnt = knh.shard_voice_range(n, r, 2)
        ws = configs.config("C3", n_voices=n, block_size=bs)
        ws.n_voices = cnt
        ws.ctor = {s: a[lo:lo + cnt] for s, a in w.ctor.items()}
        parts.append((lo, cnt, _bank(knh, ws)))
    v = np.arange(n, dtype=np.uint32)
    for launch in range(3):
        for b in ranks + [plain]:
            _script(w, launch, b)
        for lo, cnt, b in parts:  # the same events, by hand, in local indices
            vl = np.arange(cnt, dtype=np.uint32)
            if launch == 0:
                fire_all(b, cnt, *w.restart)
                g = v[::3]
                g = g[(g >= lo) & (g < lo + cnt)]
                b.param_apply_many(g - lo, 2, 0, L.VALUE_FLOAT, 300.0 + g, block_offset=1)
            if launch == 1:
                b.param_apply_many(vl, w.release[0], w.release[1], L.VALUE_TRIGGER, block_offset=2)
                if lo <= n - 1 < lo + cnt:
                    b.param_apply(n - 1 - lo, 0, 0, 777.0)
        outs = [b.process_blocks(4)[0] for b in ranks]
        want, _ = plain.process_blocks(4)
        for r in range(2):
            assert_bit_equal(outs[r], parts[r][2].process_blocks(4)[0], f"rank {r} launch {launch}: its own range, nothing else")
        assert np.max(np.abs(outs[0].astype(np.float64) + outs[1] - want)) <= 1e-5
    assert len(seen) == 6 and all(s == (4 * 2 * bs, 0, 0) for s in seen)
    with pytest.raises(L.KnasterHipError) as e:
        ranks[0].param_apply(n, 0, 0, 1.0)
    assert e.value.status == L.ERR_OUT_OF_RANGE
    ranks[0].param_apply(n - 1, 0, 0, 1.0)  # the other rank's voice: accepted, no effect here
    for b in ranks + [plain] + [p[2] for p in parts]:
        b.close()


def test_eight_ranks_in_one_process_split_c4(knh):
    """BASELINE.json's C4 split as north_star splits it -- 65 536 f64 voices, 8 ranks x 8 192 -- with all eight ranks in THIS
    process (a reduce function that leaves the buffers alone): every rank renders exactly the contiguous range
    knh_shard_voice_range gives it, from the same global parameter stream (calls for the other ranks' voices dropped), and the
    eight partial mixes add up to the one-GPU bank's mix.  What depends on the NUMBER of ranks (ranges, routing, the reduce's
    shape and root) is exercised without the hardware; the RCCL sum itself is one ncclReduce (test_rccl_communicator_with_one_rank)."""
    n, bs, world = 65536, 512, 8
    w = configs.config("C4", n_voices=n, block_size=bs)
    assert w.sample_type == L.F64
    seen = []

    def no_reduce(_user, buf, count, sample_type, root, stream):
        seen.append((count, sample_type, root))
        return 0
    ranks = [_bank(knh, w, rank=r, world=world, reduce_fn=no_reduce) for r in range(world)]
    plain = _bank(knh, w)
    ranges = [knh.shard_voice_range(n, r, world) for r in range(world)]
    assert ranges == [(r * 8192, 8192) for r in range(world)]
    own = {}
    for r in (0, 3, 7):  # three of the ranges as plain banks of their own
        lo, cnt = ranges[r]
        ws = configs.config("C4", n_voices=n, block_size=bs)
        ws.n_voices = cnt
        ws.ctor = {s: a[lo:lo + cnt] for s, a in w.ctor.items()}
        own[r] = _bank(knh, ws)
    v = np.arange(n, dtype=np.uint32)
    cut = 300.0 + (v[::3] % 1000)  # new SvfFilter cutoffs for every third voice (below Nyquist, unlike _script's at this size)
    for launch in range(2):
        for b in ranks + [plain]:  # the same GLOBAL parameter stream to every rank
            if launch == 0:
                fire_all(b, n, *w.restart)
                b.param_apply_many(v[::3], 2, 0, L.VALUE_FLOAT, cut, block_offset=1)
            if launch == 1:
                b.param_apply_many(v, w.release[0], w.release[1], L.VALUE_TRIGGER, block_offset=2)
                b.param_apply(n - 1, 0, 0, 777.0)
        for r, b in own.items():  # the same events in local indices
            lo, cnt = ranges[r]
            vl = np.arange(cnt, dtype=np.uint32)
            if launch == 0:
                fire_all(b, cnt, *w.restart)
                g = v[::3]
                g = g[(g >= lo) & (g < lo + cnt)]
                b.param_apply_many(g - lo, 2, 0, L.VALUE_FLOAT, 300.0 + (g % 1000), block_offset=1)
            if launch == 1:
                b.param_apply_many(vl, w.release[0], w.release[1], L.VALUE_TRIGGER, block_offset=2)
                if lo <= n - 1 < lo + cnt:
                    b.param_apply(n - 1 - lo, 0, 0, 777.0)
        outs = [b.process_blocks(3)[0] for b in ranks]
        want, _ = plain.process_blocks(3)
        for r, b in own.items():
            assert_bit_equal(outs[r], b.process_blocks(3)[0], f"rank {r} launch {launch}: its own range, nothing else")
        total = np.sum(np.stack(outs).astype(np.float64), axis=0)
        assert np.isfinite(want).all() and np.max(np.abs(total - want)) <= 1e-12 and np.abs(want).max() > 1e-4
    assert len(seen) == 2 * world and all(s == (3 * 2 * bs, 1, 0) for s in seen)
    assert all(b.ranks() == world for b in ranks)
    for b in ranks + [plain] + list(own.values()):
        b.close()


def test_rccl_communicator_with_one_rank(knh):
    """The RCCL path itself, with the one rank a one-GPU box allows: id, ncclCommInitRank, ncclCommCount, an in-place
    ncclReduce on the communicator's stream ordered after the producer stream, the waits."""
    lib = L.load()
    assert lib.knh_comm_rccl_version() > 0
    cid = knh.comm_unique_id()
    assert len(cid) == L.COMM_ID_BYTES and any(cid)
    comm = C.c_void_p()
    assert lib.knh_comm_create(0, 1, cid, -1, C.byref(comm)) == L.OK, lib.knh_comm_last_error(None)
    assert lib.knh_comm_world(comm) == 1
    n = 4096
    buf = lib.knh_device_malloc(n * 4, -1)
    assert buf
    assert lib.knh_comm_reduce_sum(comm, buf, n, L.F32, 0, None) == L.OK, lib.knh_comm_last_error(comm)
    assert lib.knh_comm_wait_buffer(comm, buf, None) == L.OK and lib.knh_comm_wait(comm, None) == L.OK
    assert lib.knh_comm_synchronize(comm) == L.OK
    host = np.ones(n, dtype=np.float32)
    assert lib.knh_device_read(host.ctypes.data_as(C.c_void_p), buf, n * 4, None) == L.OK
    assert not host.any()  # zero-initialised, summed over one rank
    assert lib.knh_comm_reduce_sum(comm, buf, n, L.F32, 1, None) != L.OK  # no such root
    lib.knh_device_free(buf)
    lib.knh_comm_destroy(comm)
    # and a rank bank that goes through it: world 1 never needs it, so ask for the id path explicitly with world 1
    w = configs.config("C3", n_voices=256, block_size=64)
    b = _bank(knh, w, rank=0, world=1, comm_id=cid)
    fire_all(b, 256, *w.restart)
    out, _ = b.process_blocks(2)
    assert np.abs(out).max() > 0
    b.close()


def test_two_ranks_two_processes_one_gpu(knh, tmp_path):
    """knh_bank_create_rank_custom with world = 2 for real: two processes, each with its share of 1000 voices on the same GPU,
    the sum of their blocks carried to rank 0 by gloo.  Rank 0 checks the result against one plain bank of all the voices."""
    port = str(_free_port())
    procs, outs = [], []
    for r in range(2):
        out = str(tmp_path / f"rank{r}.json")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multi_rank_worker.py"), str(r), "2", port, "C3", "1000", "64", out],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log
    res = [json.load(open(o)) for o in outs]
    assert [r["lo"] for r in res] == [0, 512] and [r["count"] for r in res] == [512, 488]
    assert all(r["reduce_calls"] == 3 and r["ranks"] == 2 for r in res)
    assert res[0]["worst"] <= 1e-5 and res[0]["peak"] > 1e-3


def test_bench_two_ranks_rehearsal(knh):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one rank per process), both ranks on this one GPU
    with KNH_BENCH_REHEARSE=1 (the library's reduce goes through gloo): the whole N > 1 control flow -- per-rank banks,
    global voice indices, a collective per launch, the agreed pre-warm, max over ranks -- ends in one JSON line."""
    env = dict(os.environ, KNH_BENCH_REHEARSE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--voices-per-gpu", "2048", "--c4-voices", "4096"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0 and d["output_finite"]
    assert d["config"]["voices_total"] == 4096 and d["config"]["voices_per_gpu"] == 2048 and d["config"]["ranks_seen_by_rccl"] == 2
    assert "REHEARSAL" in d["data"]
    assert d["c4_strong"]["scaling"] == "strong" and d["c4_strong"]["voices_total"] == 4096 and d["c4_strong"]["voices_per_gpu"] == 2048
    assert d["c4_strong"]["dtype"] == "f64" and d["c4_strong"]["value"] > 0 and d["c4_strong"]["output_finite"]
    # every rank's own kernel time and reduce time per launch (what a first real multi-GPU run is read with)
    for pr in (d["config"]["per_rank"], d["c4_strong"]["per_rank"]):
        assert len(pr["kernel_ms_per_launch"]) == 2 and all(x > 0 for x in pr["kernel_ms_per_launch"])
        assert len(pr["reduce_ms_per_launch"]) == 2 and all(x > 0 for x in pr["reduce_ms_per_launch"])
        assert pr["voices"] == [2048, 2048]


def test_bench_two_ranks_rehearsal_c4_headline(knh):
    """The same with BASELINE.json's multi-GPU configuration as the headline (`--config C4`: f64, the voices of ONE bank split
    over the ranks, strong scaling)."""
    env = dict(os.environ, KNH_BENCH_REHEARSE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--config", "C4", "--c4-voices", "8192"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["dtype"] == "f64" and d["value"] > 0 and d["output_finite"]
    assert d["config"]["voices_total"] == 8192 and d["config"]["voices_per_gpu"] == 4096 and d["config"]["ranks_seen_by_rccl"] == 2
    assert d["config"]["per_rank"]["voices"] == [4096, 4096] and "c4_strong" not in d and "REHEARSAL" in d["data"]
