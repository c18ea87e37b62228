"""dist.py -- torch.distributed plumbing for the per-level exchange of dsm_mine (one all-gather per frontier
level).  The library asks its host for an all-gather through a C callback; here the buffers are
torch tensors so that backend "nccl" (= RCCL over xGMI) moves them device to device.  Backend "gloo" (CPU
tests, or several ranks sharing one GPU) stages through host memory.  No compute happens here."""
import torch
import torch.distributed as dist


class TurnGate:
    """Deterministic interleaving of the collectives of several lanes (threads) of one process.

    Every rank runs the same lanes over the same prefixes, so each lane issues the same number of collectives on every
    rank; strict round-robin between the lanes that are still active therefore enqueues collectives of different
    communicators in the same order on all ranks (required for NCCL/RCCL communicators used concurrently)."""

    def __init__(self, nlanes):
        import threading
        self.cv = threading.Condition()
        self.active = [True] * nlanes
        self.turn = 0
        self.log = [] if __import__('os').environ.get('DSM_TRACE_EXCHANGE') else None

    def _advance(self):
        n = len(self.active)
        for k in range(1, n + 1):
            j = (self.turn + k) % n
            if self.active[j]:
                self.turn = j
                return

    def acquire(self, lane):
        with self.cv:
            self.cv.wait_for(lambda: self.turn == lane)

    def release(self, lane):
        with self.cv:
            self._advance()
            self.cv.notify_all()

    def begin(self, lane):
        with self.cv:
            self.active[lane] = True

    def retire(self, lane):
        with self.cv:
            self.active[lane] = False
            if self.turn == lane:
                self._advance()
            self.cv.notify_all()

    def reset(self):
        with self.cv:
            self.active = [True] * len(self.active)
            self.turn = 0


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class Exchange:
    """Owns the send / recv buffers handed to dsm_params.exchange_* and implements dsm_allgather_fn."""

    def __init__(self, nbytes, world_size, device, group=None, stream=None, gate=None, lane=0):
        self.stream = stream  # torch.cuda.Stream the library's kernels run on (None = current stream)
        self.gate, self.lane = gate, lane
        self.nbytes = int(nbytes) & ~15
        self.world = int(world_size)
        self.group = group
        self.device = torch.device(device)
        self.send = torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device)
        # two halves: consecutive frontier levels alternate (the previous level's columns stay readable)
        self.recv = torch.zeros(2 * self.nbytes * self.world, dtype=torch.uint8, device=self.device)
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.calls = 0
        self.trace = [] if __import__('os').environ.get('DSM_TRACE_EXCHANGE') else None
        self.bytes_moved = 0

    def params(self):
        """(exchange_send, exchange_recv, exchange_bytes) for pydsm.Miner / pydsm.mine."""
        return (self.send.data_ptr(), self.recv.data_ptr(), self.nbytes)

    def half(self, recv_ptr):
        off = recv_ptr - self.recv.data_ptr()
        if off not in (0, self.nbytes * self.world):
            raise ValueError("recv pointer is not one of the two exchange halves")
        return off

    def allgather(self, send_ptr, recv_ptr, nbytes, stream=None):
        """recv[half][r * nbytes : (r+1) * nbytes] = rank r's send[:nbytes]  (ncclAllGather layout)."""
        if send_ptr != self.send.data_ptr():
            raise ValueError("send pointer is not the exchange send buffer")
        if nbytes > self.nbytes:
            raise ValueError("level larger than the exchange buffers")
        off = self.half(recv_ptr)
        if self.gate is not None:
            self.gate.acquire(self.lane)
            try:
                if self.gate.log is not None:
                    self.gate.log.append((self.lane, int(nbytes)))
                self._allgather(nbytes, off)
            finally:
                self.gate.release(self.lane)
        else:
            self._allgather(nbytes, off)
        self.calls += 1
        self.bytes_moved += nbytes * self.world

    def _allgather(self, nbytes, off):
        if self.trace is not None:
            self.trace.append(int(nbytes))
        out = self.recv[off: off + nbytes * self.world]
        src = self.send[:nbytes]
        if self.device.type == "cpu":
            dist.all_gather_into_tensor(out, src, group=self.group)
        elif self.backend == "nccl":
            # the collective is ordered after the library's kernels on its stream, and later kernels after the collective
            with torch.cuda.stream(self.stream) if self.stream is not None else _null():
                dist.all_gather_into_tensor(out, src, group=self.group)
        else:  # gloo with device buffers: stage through the host
            st = self.stream if self.stream is not None else torch.cuda.current_stream(self.device)
            with torch.cuda.stream(st):
                st.synchronize()
                h_src = src.cpu()
                h_out = torch.empty(nbytes * self.world, dtype=torch.uint8)
                dist.all_gather_into_tensor(h_out, h_src, group=self.group)
                out.copy_(h_out)
                st.synchronize()


    # ---- owner mode (dsm_gather_fn / dsm_bcast_fn): the columns go to the prefix's owner, the union's child planes come back ----
    def _turn(self, fn):
        if self.gate is not None:
            self.gate.acquire(self.lane)
            try:
                fn()
            finally:
                self.gate.release(self.lane)
        else:
            fn()
        self.calls += 1

    def gather(self, root, send_ptr, recv_ptr, nbytes, stream=None):
        """recv[half][r * nbytes : (r+1) * nbytes] on rank `root` = rank r's send[:nbytes]; the others only send.
        `stream`: the HIP stream the library's kernels that wrote `send` run on; the transfers are ordered on it."""
        self._lib_stream = stream
        if send_ptr != self.send.data_ptr():
            raise ValueError("send pointer is not the exchange send buffer")
        if nbytes > self.nbytes:
            raise ValueError("level larger than the exchange buffers")
        off = self.half(recv_ptr)
        self._turn(lambda: self._gather(root, nbytes, off))

    def _global(self, r):
        """torch.distributed's point-to-point and rooted calls take GLOBAL ranks, the library speaks in ranks of this exchange's group"""
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def _gather(self, root, nbytes, off):
        me = dist.get_rank(self.group)
        src = self.send[:nbytes]
        out = self.recv[off: off + nbytes * self.world]
        if self.world == 1:
            out.copy_(src)
            return
        st = self.stream if self.stream is not None else (torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None)
        if self.stream is None and self.device.type == "cuda" and getattr(self, "_lib_stream", None):
            st = torch.cuda.ExternalStream(int(self._lib_stream), device=self.device)  # (the library's stream, not torch's current one)
        if self.backend == "nccl":
            with torch.cuda.stream(st):
                if me == root:
                    out[root * nbytes:(root + 1) * nbytes].copy_(src)
                    ops = [dist.P2POp(dist.irecv, out[r * nbytes:(r + 1) * nbytes], self._global(r), self.group) for r in range(self.world) if r != root]
                else:
                    ops = [dist.P2POp(dist.isend, src, self._global(root), self.group)]
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            self.bytes_moved += nbytes * (self.world - 1 if me == root else 1)
            return
        # gloo: through the host
        if st is not None:
            st.synchronize()
        h_src = src.cpu()
        if me == root:
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
            dist.gather(h_src, parts, dst=self._global(root), group=self.group)
            out.copy_(torch.cat(parts))
            if st is not None:
                st.synchronize()
        else:
            dist.gather(h_src, None, dst=self._global(root), group=self.group)
        self.bytes_moved += nbytes * (self.world - 1 if me == root else 1)

    def bcast(self, root, ptr, nbytes, stream=None):
        """nbytes at device pointer `ptr` (a buffer of the library) on rank `root` reach `ptr` on every rank."""
        self._turn(lambda: self._bcast(root, ptr, nbytes, stream))

    def _bcast(self, root, ptr, nbytes, stream):
        import ctypes as C
        from . import lib
        if self.world == 1:
            return
        me = dist.get_rank(self.group)
        if self.device.type == "cpu":
            raise ValueError("owner mode needs device buffers")
        h = torch.empty(nbytes, dtype=torch.uint8)
        if me == root and lib().dsm_copy_from_device(h.data_ptr(), ptr, nbytes, stream):
            raise RuntimeError("dsm_copy_from_device failed")
        if self.backend == "nccl":
            d = h.to(self.device)
            dist.broadcast(d, src=self._global(root), group=self.group)
            h = d.cpu()
        else:
            dist.broadcast(h, src=self._global(root), group=self.group)
        if me != root and lib().dsm_copy_to_device(ptr, h.data_ptr(), nbytes, stream):
            raise RuntimeError("dsm_copy_to_device failed")
        self.bytes_moved += nbytes
