"""Host-side mirror of the reference's per-patch seam for the `filter` hot path.

`FilterContext.filter_batch()` == filter_pe_fqs/filter_se_fqs + stat_*_fqs("raw")
+ stat_*_fqs("clean") of the reference (src/peprocess.cpp:1424,1076;
src/seprocess.cpp:871,632) on one device-resident patch.  torch is used for
device memory, streams and torch.distributed only; all work happens behind the
C ABI of include/snk_filter.h in hand-written HIP.  There is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import abi


class FilterError(RuntimeError):
    pass


class FilterContext:
    def __init__(self, params, device=0, lib=None):
        import torch
        self.torch = torch
        self.lib = lib or abi.load_library()
        self.params = params
        self.device = device
        torch.cuda.set_device(device)
        self.ctx = self.lib.snk_create(C.byref(params), device)
        if not self.ctx:
            raise FilterError(self.lib.snk_last_error().decode())
        lcap, nq, n = C.c_int32(), C.c_int32(), C.c_int64()
        self.lib.snk_stats_geometry(self.ctx, C.byref(lcap), C.byref(nq), C.byref(n))
        self.lcap, self.nq, self.sum_u64 = lcap.value, nq.value, n.value
        dev = torch.device("cuda", device)
        # accumulators live in torch tensors so that torch.distributed (RCCL) can reduce them
        self.sum = torch.zeros(self.sum_u64, dtype=torch.int64, device=dev)
        self.max = torch.zeros(abi.SNK_MAX_N, dtype=torch.int64, device=dev)
        self._check(self.lib.snk_bind_stats(self.ctx, self.sum.data_ptr(), self.max.data_ptr()))
        self.clear()

    def bind(self, sum_t, max_t):
        """Accumulate the following launches into other (zeroed, same-shape int64 cuda) tensors -- e.g. one
        block per virtual reference thread (snk_bind_stats)."""
        self.sum, self.max = sum_t, max_t
        self._check(self.lib.snk_bind_stats(self.ctx, sum_t.data_ptr(), max_t.data_ptr()))

    def _check(self, rc):
        if rc != 0:
            raise FilterError(f"snk error {rc}: {self.lib.snk_last_error().decode()}")

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.snk_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        self._check(self.lib.snk_stats_clear(self.ctx, self._stream()))

    def upload(self, data):
        """numpy batch (soapnuke_amd.synth.make_batch layout) -> dict of cuda tensors."""
        t, dev = self.torch, self.torch.device("cuda", self.device)
        out = {"n": data["n"], "L": data["L"], "pitch": data["pitch"], "seq": [], "qual": [], "len": []}
        for m in range(len(data["seq"])):
            out["seq"].append(t.from_numpy(np.ascontiguousarray(data["seq"][m])).to(dev))
            out["qual"].append(t.from_numpy(np.ascontiguousarray(data["qual"][m])).to(dev))
            ln = data["len"][m]
            out["len"].append(None if ln is None else t.from_numpy(ln.astype(np.int16)).to(dev))
        return out

    def make_batch(self, dev_data, first_index=0, dup=None):
        b = abi.Batch()
        b.n = dev_data["n"]
        b.pitch = dev_data["pitch"]
        for m in range(len(dev_data["seq"])):
            b.fixed_len[m] = dev_data["L"]
            b.seq[m] = dev_data["seq"][m].data_ptr()
            b.qual[m] = dev_data["qual"][m].data_ptr()
            if dev_data["len"][m] is not None:
                b.len[m] = dev_data["len"][m].data_ptr()
        if dup is not None:
            b.dup = dup.data_ptr()
        b.first_index = first_index
        return b

    def alloc_records(self, n):
        t, dev = self.torch, self.torch.device("cuda", self.device)
        return [t.empty((n, 16), dtype=t.uint8, device=dev) for _ in range(2)]

    def filter_batch(self, batch, records, kernel=0):
        """Asynchronous on the current torch stream."""
        self._check(self.lib.snk_filter_batch_device(self.ctx, C.byref(batch), records[0].data_ptr(),
                                                     records[1].data_ptr(), self._stream(), kernel))

    def set_timing(self, on=True):
        self._check(self.lib.snk_set_timing(self.ctx, 1 if on else 0))

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(self.lib.snk_last_kernel_ms(self.ctx, C.byref(ms)))
        return ms.value

    def finalize(self):
        self._check(self.lib.snk_stats_finalize(self.ctx, self._stream()))

    def fetch(self):
        """-> (sum uint64 ndarray, max uint64 ndarray, (code, mate, index))"""
        s = np.zeros(self.sum_u64, dtype=np.uint64)
        mx = np.zeros(abi.SNK_MAX_N, dtype=np.uint64)
        err = abi.Error()
        self._check(self.lib.snk_stats_fetch(self.ctx, s.ctypes.data, mx.ctypes.data, C.byref(err), self._stream()))
        return s, mx, (err.code, err.mate, err.index)

    # ---- rmdup pre-pass (include/snk_rmdup.h) ----------------------------------------------
    def hash_batch(self, batch, out=None):
        """std::hash<std::string>(mate1 ++ mate2) of every pair of a device batch -> int64 cuda tensor
        (the uint64 bit patterns; src/peprocess.cpp:3665,3680).  Asynchronous on the current stream."""
        t, dev = self.torch, self.torch.device("cuda", self.device)
        if out is None:
            out = t.empty(int(batch.n), dtype=t.int64, device=dev)
        self._check(self.lib.snk_rmdup_hash_device(self.ctx, C.byref(batch), out.data_ptr(), self._stream()))
        return out

    def bucket_count(self, hashes, total_n):
        """Population of the bucket of 2**64-1 among `hashes` (sentinel quirk, multi-GPU only) -> int"""
        t, dev = self.torch, self.torch.device("cuda", self.device)
        cnt = t.zeros(1, dtype=t.int64, device=dev)
        self._check(self.lib.snk_rmdup_bucket_count_device(self.ctx, hashes.data_ptr(), hashes.numel(), int(total_n),
                                                           cnt.data_ptr(), self._stream()))
        return cnt

    def mark_dups(self, hashes, index=None, total_n=None, sentinel_bucket_total=-1):
        """rmdup::markDup (src/rmdup.cpp:14): uint8 flag per hash, 1 = an equal hash has a smaller index.
        index: optional int32/uint32 cuda tensor of global input-order indices (default: position)."""
        t, dev = self.torch, self.torch.device("cuda", self.device)
        n = hashes.numel()
        dup = t.zeros(n, dtype=t.uint8, device=dev)
        self._check(self.lib.snk_rmdup_mark_device(self.ctx, hashes.data_ptr(), 0 if index is None else index.data_ptr(), n,
                                                   int(n if total_n is None else total_n), int(sentinel_bucket_total),
                                                   dup.data_ptr(), self._stream()))
        return dup

    def allreduce(self):
        """Sum/max all-reduce of the accumulators over torch.distributed (RCCL on GPUs):
        the only collective of this path."""
        from .shard import allreduce_stats
        self.finalize()
        allreduce_stats(self.sum, self.max)


def records_to_numpy(rec):
    return rec.cpu().numpy().view(abi.record_dtype()).reshape(-1)
