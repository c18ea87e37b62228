"""dist.py -- torch.distributed plumbing for the per-level exchange of dsm_mine (one all-gather per frontier
level).  The library asks its host for an all-gather through a C callback; here the buffers are
torch tensors so that backend "nccl" (= RCCL over xGMI) moves them device to device.  Backend "gloo" (CPU
tests, or several ranks sharing one GPU) stages through host memory.  No compute happens here."""
import torch
import torch.distributed as dist


class Exchange:
    """Owns the send / recv buffers handed to dsm_params.exchange_* and implements dsm_allgather_fn."""

    def __init__(self, nbytes, world_size, device, group=None):
        self.nbytes = int(nbytes)
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
        if self.trace is not None:
            self.trace.append(int(nbytes))
        out = self.recv[off: off + nbytes * self.world]
        src = self.send[:nbytes]
        if self.backend == "nccl" or self.device.type == "cpu":
            dist.all_gather_into_tensor(out, src, group=self.group)
        else:  # gloo with device buffers: stage through the host
            torch.cuda.current_stream(self.device).synchronize()
            h_out = torch.empty(nbytes * self.world, dtype=torch.uint8)
            dist.all_gather_into_tensor(h_out, src.cpu(), group=self.group)
            out.copy_(h_out)
            torch.cuda.current_stream(self.device).synchronize()
        self.calls += 1
        self.bytes_moved += nbytes * self.world
