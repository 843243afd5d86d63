"""tests/test_multirank_gpu.py's world-2 run (every rank filters its contiguous shard with the tiled kernel, the stats all-reduce,
with rmdup the (hash, global index) all-to-all with owner-side marking and the flags back) on the emulated library, gloo as the
wire: reduced block, concatenated records and duplicate flags equal the unsharded oracle run bit for bit."""
import pytest

import test_multirank_gpu as MR

CORE = ["test_two_ranks_emulated[True]"]


def _worker(rank, world, port, rmdup, tmp):
    import simt_lib as S
    mp = pytest.MonkeyPatch()
    S.torch_on_host(mp)
    MR._worker(rank, world, port, "gloo", rmdup, False, tmp)


@pytest.mark.parametrize("rmdup", [False, True])
def test_two_ranks_emulated(tmp_path, rmdup, monkeypatch):
    import simt_lib as S
    S.lib()                                                   # (built once, before the ranks start)
    monkeypatch.setattr(MR, "_worker", MR._worker)            # (keeps the module imported for the spawned ranks)
    real_spawn = MR.mp.spawn

    def spawn(fn, args, nprocs, join):
        world, port, backend, rm, c_abi, tmp = args
        return real_spawn(_worker, args=(world, port, rm, tmp), nprocs=nprocs, join=join)

    monkeypatch.setattr(MR.mp, "spawn", spawn)
    MR._run(tmp_path, "gloo", rmdup)
