"""One rank of a two-process run of the library's sharded bank (knh_bank_create_rank_custom) on ONE GPU, with the sum of
the ranks' blocks done through gloo on the host (RCCL refuses two ranks on one device).  Started twice by
tests/test_gpu_multi.py.  Rank 0 also renders the same voices as one plain bank and compares."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, name, n_voices, block_size, out_path = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]),
                                                               int(sys.argv[6]), sys.argv[7])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    import torch
    import torch.distributed as dist

    import knaster_amd
    from knaster_amd import _lib as L
    from knaster_amd import configs

    dist.init_process_group("gloo", rank=rank, world_size=world)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    calls = [0]

    def reduce(_user, buf, count, sample_type, root, stream):
        host = np.empty(count, dtype=np.float64 if sample_type == 1 else np.float32)
        if hip.hipStreamSynchronize(stream) != 0 or hip.hipMemcpy(host.ctypes.data, buf, host.nbytes, 2) != 0:
            return 4
        t = torch.from_numpy(host)
        dist.reduce(t, dst=root, op=dist.ReduceOp.SUM)
        if rank == root and hip.hipMemcpy(buf, host.ctypes.data, host.nbytes, 1) != 0:
            return 4
        calls[0] += 1
        return 0

    w = configs.config(name, n_voices=n_voices, block_size=block_size)
    bank = knaster_amd.VoiceBank(w.stages, n_voices, w.sample_type, w.out_channels, L.MIX_TREE, 0, False, rank=rank, world=world, reduce_fn=reduce)
    for s, a in w.ctor.items():
        bank.set_ctor_args(s, a)
    bank.init(48000, block_size)
    everyone = np.arange(n_voices, dtype=np.uint32)  # every rank is handed the same parameter stream
    lo, cnt = knaster_amd.shard_voice_range(n_voices, rank, world)
    K = 6
    outs = []
    for launch in range(3):
        if launch == 0:
            bank.param_apply_many(everyone, w.restart[0], w.restart[1], L.VALUE_TRIGGER)
            bank.param_apply_many(everyone[::5], 0, 0, L.VALUE_FLOAT, 200.0 + everyone[::5], block_offset=2)
        if launch == 1:
            bank.param_apply_many(everyone, w.release[0], w.release[1], L.VALUE_TRIGGER, block_offset=3)
        out, flags = bank.process_blocks(K)
        outs.append(out)
    result = {"rank": rank, "lo": lo, "count": cnt, "reduce_calls": calls[0], "ranks": bank.ranks()}
    if rank == 0:
        ref = knaster_amd.VoiceBank(w.stages, n_voices, w.sample_type, w.out_channels, L.MIX_TREE, 0, False)
        for s, a in w.ctor.items():
            ref.set_ctor_args(s, a)
        ref.init(48000, block_size)
        worst, peak = 0.0, 0.0
        for launch in range(3):
            if launch == 0:
                ref.param_apply_many(everyone, w.restart[0], w.restart[1], L.VALUE_TRIGGER)
                ref.param_apply_many(everyone[::5], 0, 0, L.VALUE_FLOAT, 200.0 + everyone[::5], block_offset=2)
            if launch == 1:
                ref.param_apply_many(everyone, w.release[0], w.release[1], L.VALUE_TRIGGER, block_offset=3)
            want, _ = ref.process_blocks(K)
            worst = max(worst, float(np.max(np.abs(want.astype(np.float64) - outs[launch]))))
            peak = max(peak, float(np.max(np.abs(want))))
        ref.close()
        result.update(worst=worst, peak=peak)
    bank.close()
    with open(out_path, "w") as f:
        json.dump(result, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
